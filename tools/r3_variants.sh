#!/bin/bash
# build variants of the kernels on the box (HIPFLAGS_EXTRA), check each against the oracle on a test subset, run the default bench
#   VARIANTS="flags;flags;..."  TESTS="pytest -k expression" (empty: no tests)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
IFS=';' read -ra VS <<< "${VARIANTS:-;}"
k=0
for v in "${VS[@]}"; do
  k=$((k+1))
  touch pim-jpeg-decoder_amd/csrc/pjd_internal.h
  make -s -C pim-jpeg-decoder_amd HIPFLAGS_EXTRA="$v" all > gpurun_out/var_build.log 2>&1 || { tail -5 gpurun_out/var_build.log; exit 1; }
  if [ -n "$TESTS" ]; then
    timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "$TESTS" > gpurun_out/var_test_$k.log 2>&1; echo "[$v] tests rc=$? $(tail -1 gpurun_out/var_test_$k.log)"
  fi
  timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 40 > gpurun_out/var_$k.log 2> gpurun_out/var_$k.err || { echo "variant $v failed"; tail -3 gpurun_out/var_$k.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/var_$k.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('[$v] cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], 'fb', d['config']['exact_kernel_images'], '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'])"
  timeout -k 10 120 python bench.py --workload cfg2 --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 50 --no-variants > gpurun_out/var_cfg2_$k.log 2>/dev/null && python3 -c "
import json
d=json.loads(open('gpurun_out/var_cfg2_$k.log').read().strip().splitlines()[-1])
print('[$v] cfg2 ms/step', d['ms_per_step'], d['kernels_ms']['huff_lanes'])"
done
# leave the default build behind
touch pim-jpeg-decoder_amd/csrc/pjd_internal.h; make -s -C pim-jpeg-decoder_amd all > /dev/null 2>&1
