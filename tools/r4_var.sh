#!/bin/bash
# round-4: library variants built beforehand (pim-jpeg-decoder_amd/lib/var/libpjd_<name>.so), each: parity subset first, then the bench
# usage: r4_var.sh name [name ...]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=pim-jpeg-decoder_amd/lib
cp $L/libpjd.so /tmp/libpjd_keep.so
for v in "$@"; do
  cp $L/var/libpjd_$v.so $L/libpjd.so || exit 1
  timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/var_$v.test 2>&1; rc=$?
  echo "$v: parity subset rc=$rc $(tail -1 gpurun_out/var_$v.test)"
  [ $rc -ne 0 ] && { cp /tmp/libpjd_keep.so $L/libpjd.so; exit $rc; }
  timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli ${VAR_BENCH_ARGS} > gpurun_out/var_$v.log 2> gpurun_out/var_$v.err || { tail -3 gpurun_out/var_$v.err; cp /tmp/libpjd_keep.so $L/libpjd.so; exit 1; }
  python3 - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads(open(f'gpurun_out/var_{v}.log').read().strip().splitlines()[-1])
l=d.get('variants',{}).get('cfg3lite')
print(v, 'cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms']['huff_lanes'], d['kernels_ms']['idct_colour'], '| lite', (l['value'], l['one_batch_in_flight']['ms_per_step']) if l else None)
PY
done
cp /tmp/libpjd_keep.so $L/libpjd.so
