#!/bin/bash
# quick GPU check: fixtures through the probe, then a serialised bench and the wave timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_probe.py > gpurun_out/probe.log 2>&1; echo "probe rc=$?"; tail -1 gpurun_out/probe.log; grep -c "^ok" gpurun_out/probe.log; grep FAIL gpurun_out/probe.log | head
for sb in ${SBS:-0}; do
  PJD_SUB_BYTES=$sb PJD_DEBUG_STATS=1 timeout -k 10 120 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10 --verify --no-variants ${BENCH_ARGS:-} > gpurun_out/q_$sb.log 2> gpurun_out/q_$sb.err
  echo "bench sb=$sb rc=$?"
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/q_$sb.log').read().strip().splitlines()[-1])
print('sb', '$sb', 'ms/step', d['ms_per_step'], d['kernels_ms'], 'verified', d.get('verified_against_oracle'), d['config']['exact_kernel_images'])
"
  grep "pjd waves" gpurun_out/q_$sb.err | head -2
done
