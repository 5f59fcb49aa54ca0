#!/bin/bash
# round-4: the other workloads of BASELINE.json on the build in the tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
one() { # label, args
  label=$1; shift
  timeout -k 10 400 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli --no-variants "$@" > gpurun_out/wl.log 2> gpurun_out/wl.err || { echo "$label failed"; tail -3 gpurun_out/wl.err; return; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/wl.log').read().strip().splitlines()[-1])
print('$label', 'in flight', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], 'fb', d['config']['exact_kernel_images'])"
}
one cfg2 --workload cfg2 --steps 200
one cfg2rst --workload cfg2rst --steps 200
one cfg5_8192 --workload cfg5 --steps 100
one cfg4size --workload cfg3 --images 8192 --steps 20
one cfg3lite --workload cfg3lite --steps 200
