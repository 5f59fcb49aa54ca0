#!/bin/bash
# round 3: GPU suite with the walker switch test, parser slot statistics (PJD_DEBUG_STATS), default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3_tests.log
[ $rc -eq 0 ] || { grep -E "^(E|FAILED)" gpurun_out/r3_tests.log | head -30; exit $rc; }
PJD_DEBUG_STATS=1 timeout -k 10 200 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10 --no-variants > gpurun_out/r3_w2_dbg.log 2> gpurun_out/r3_w2_dbg.err; echo "dbg rc=$?"
grep "back end\]" gpurun_out/r3_w2_dbg.err | head -2
timeout -k 10 400 python bench.py --steps 100 --warmup 8 --e2e-batches 0 --no-cpu-baseline > gpurun_out/r3_w2_bench.log 2> gpurun_out/r3_w2_bench.err || { echo bench failed; tail -5 gpurun_out/r3_w2_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_w2_bench.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], d['config']['sync'], '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'])
PY
