#!/bin/bash
# how many batches in flight pay, with more hardware queues than the runtime's default of 4 (GPU_MAX_HW_QUEUES)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in ${QCFG:-4:4 8:4 8:6 8:8 16:8 16:12}; do
  set -- ${cfg/:/ }
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python bench.py --in-flight $2 --e2e-batches 0 --no-cpu-baseline --steps 48 --warmup 12 --no-variants > gpurun_out/r3_q_$1_$2.log 2> gpurun_out/r3_q_$1_$2.err || { echo "queues $1 in-flight $2 failed"; tail -2 gpurun_out/r3_q_$1_$2.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3_q_$1_$2.log').read().strip().splitlines()[-1])
print('hw queues $1 in flight $2: value', d['value'], 'ms', d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'])"
done
