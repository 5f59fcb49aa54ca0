#!/bin/bash
# round-4: the GPU suite on the build in the tree, then the default bench (parity first, numbers second)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu ${TESTS:+-k "$TESTS"} > gpurun_out/r4_test.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r4_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], 'fb', d['config']['exact_kernel_images'], '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'])
print(d['config']['sync'])
PY
