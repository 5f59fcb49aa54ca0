#!/bin/bash
# round-4: the PCIe-inclusive rate (pipelined batcher) under different settings.  usage: r4_pcie.sh "LABEL VAR=val" ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
for spec in "$@"; do
  set -- $spec; label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-cli --no-variants --steps 30 > gpurun_out/pcie_$label.log 2> gpurun_out/pcie_$label.err || { echo "$label failed"; tail -3 gpurun_out/pcie_$label.err; exit 1; }
  python3 - "$label" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads(open(f'gpurun_out/pcie_{v}.log').read().strip().splitlines()[-1])
p=d['pcie_inclusive']
print(v, 'pcie', p['value'], p['d2h_GBps'], p['wall_ms'], p['worker_ms'], '| value', d['value'])
PY
done
