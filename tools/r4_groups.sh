#!/bin/bash
# picture groups: adaptive choice (groups when the device is idle), cut fractions
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams or idempotent or two_batches or replayed" > gpurun_out/grp_test.log 2>&1; echo "parity subset rc=$? $(tail -1 gpurun_out/grp_test.log)"
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli --steps 100 > gpurun_out/grp.log 2> gpurun_out/grp.err || { echo "$label failed"; tail -3 gpurun_out/grp.err; return; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/grp.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('$label', 'in flight', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], '| lite', v['value'], 'serial', v['one_batch_in_flight']['ms_per_step'])"
}
run "default (20,55)" A=1



run "cuts 30" PJD_GROUP_CUTS=30
run "groups off" PJD_GROUPS=1
