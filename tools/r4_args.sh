#!/bin/bash
# round-4: one bench per argument set, after a parity subset on the build in the tree.  usage: r4_args.sh "LABEL args..." ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/args_test.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/args_test.log)"
[ $rc -ne 0 ] && exit $rc
for spec in "$@"; do
  set -- $spec; label=$1; shift
  timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli --steps 100 "$@" > gpurun_out/args_$label.log 2> gpurun_out/args_$label.err || { echo "$label failed"; tail -3 gpurun_out/args_$label.err; exit 1; }
  python3 - "$label" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads(open(f'gpurun_out/args_{v}.log').read().strip().splitlines()[-1])
l=d.get('variants',{}).get('cfg3lite')
o=d.get('one_batch_in_flight',{})
print(v, 'cfg3', d['value'], d['ms_per_step'], 'serial', o.get('ms_per_step'), d['kernels_ms']['huff_lanes'], d['kernels_ms']['idct_colour'], '| lite', (l['value'], l.get('one_batch_in_flight',{}).get('ms_per_step')) if l else None)
PY
done
