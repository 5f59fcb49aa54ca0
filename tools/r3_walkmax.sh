#!/bin/bash
# walker threshold by environment (PJD_WALK_MAX) after the pair tables; wave timeline of the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PJD_DEBUG_STATS=1 timeout -k 10 200 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 20 --no-variants 2>&1 >/dev/null | grep "pjd waves\] n \|last to finish\|rounds (lanes" | head -6
for t in dflt 3 4 6 8 12; do
  if [ $t = dflt ]; then unset PJD_WALK_MAX; else export PJD_WALK_MAX=$t; fi
  timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 60 > gpurun_out/r3_wm_$t.log 2> gpurun_out/r3_wm_$t.err || { echo "walk max $t failed"; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3_wm_$t.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('walk max $t: cfg3', d['value'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'], '| lite', v['value'], 'serial', v['one_batch_in_flight']['ms_per_step'])"
done
