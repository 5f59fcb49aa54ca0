#!/bin/bash
# subsequence size for the default batch after walker / pairs (PJD_SUB_BYTES overrides the planner's choice)
cd "$GRAFT_REPO_ROOT" || exit 1
for sb in 0 512 768 1024; do
  PJD_SUB_BYTES=$sb timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 60 --no-variants 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sub_bytes $sb ->', d['config']['sub_bytes'], 'cfg3', d['value'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'], 'idct', d['kernels_ms']['idct_colour'])"
done
