#!/bin/bash
# round 3, GPU call 3: full GPU suite on the restructured Huffman loops, then bench (in flight + serial) and cfg2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3_tests.log
[ $rc -eq 0 ] || { grep -E "^(E|FAILED)" gpurun_out/r3_tests.log | head -30; exit $rc; }
timeout -k 10 400 python bench.py --steps 40 --warmup 8 --e2e-batches 0 --no-cpu-baseline > gpurun_out/r3_bench.log 2> gpurun_out/r3_bench.err || { echo bench failed; tail -5 gpurun_out/r3_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench.log').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], 'lite', d['variants']['cfg3lite']['value'], d['variants']['cfg3lite']['one_batch_in_flight'])
PY
timeout -k 10 120 python bench.py --workload cfg2 --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 50 --no-variants > gpurun_out/r3_cfg2.log 2> gpurun_out/r3_cfg2.err && python3 -c "
import json
d=json.loads(open('gpurun_out/r3_cfg2.log').read().strip().splitlines()[-1])
print('cfg2 ms/step', d['ms_per_step'], d['kernels_ms'])
"
