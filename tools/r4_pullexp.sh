#!/bin/bash
# pull back end: persistent workers (PJD_PULL_WORKERS), parity subset first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random or idempotent or two_batches or replayed or default_bench or dense_optimised or lane" > gpurun_out/pe_test.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/pe_test.log)"
[ $rc -ne 0 ] && { tail -30 gpurun_out/pe_test.log; exit $rc; }
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli --steps 150 > gpurun_out/pe.log 2> gpurun_out/pe.err || { echo "$label failed"; tail -3 gpurun_out/pe.err; return; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/pe.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('$label', 'in flight', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], '| lite', v['value'], 'serial', v['one_batch_in_flight']['ms_per_step'], 'fb', d['config']['exact_kernel_images'])"
}
run "pull 512" PJD_IDLE_FORM=pull
run "pull 256" PJD_IDLE_FORM=pull PJD_PULL_WORKERS=256
run "pull 1024" PJD_IDLE_FORM=pull PJD_PULL_WORKERS=1024
run "pull 128" PJD_IDLE_FORM=pull PJD_PULL_WORKERS=128
run "groups" PJD_IDLE_FORM=groups
run "chain" PJD_IDLE_FORM=chain
