#!/bin/bash
# walker threshold after the B cache (PJD_WALK_MAX overrides the planner's per-picture choice), three repetitions of the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli --no-variants --steps 150 > gpurun_out/wm.log 2> gpurun_out/wm.err || { echo "$label failed"; tail -3 gpurun_out/wm.err; return; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/wm.log').read().strip().splitlines()[-1])
print('$label', 'in flight', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'])"
}
run "default" A=1
run "walk 4" PJD_WALK_MAX=4
run "walk 6" PJD_WALK_MAX=6
run "walk 12" PJD_WALK_MAX=12
run "walk 16" PJD_WALK_MAX=16
run "walk 24" PJD_WALK_MAX=24
run "default again" A=1
