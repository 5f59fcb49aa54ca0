#!/bin/bash
# what a batch alone looks like on an idle device: PJD_IDLE_FORM = chain | groups | pull (the default); parity first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/forms_test.log 2>&1; rc=$?; echo "tests rc=$rc $(tail -1 gpurun_out/forms_test.log)"
[ $rc -ne 0 ] && { tail -30 gpurun_out/forms_test.log; exit $rc; }
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli --steps 150 > gpurun_out/forms.log 2> gpurun_out/forms.err || { echo "$label failed"; tail -3 gpurun_out/forms.err; return; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/forms.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('$label', 'in flight', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], '| lite', v['value'], 'serial', v['one_batch_in_flight']['ms_per_step'], 'fb', d['config']['exact_kernel_images'])"
}
run "pull" PJD_IDLE_FORM=pull
run "groups" PJD_IDLE_FORM=groups
run "chain" PJD_IDLE_FORM=chain
run "pull again" PJD_IDLE_FORM=pull
