#!/bin/bash
# what extra LDS per Huffman workgroup costs now (PJD_EXTRA_LDS): decides whether a second table per AC table (symbol pairs) can pay
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for x in 0 2048 4096 0; do
  PJD_EXTRA_LDS=$x timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 60 > gpurun_out/r3_lds_$x.log 2> gpurun_out/r3_lds_$x.err || { echo "extra $x failed"; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3_lds_$x.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('extra LDS $x: cfg3', d['value'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'], '| lite', v['value'], 'serial', v['one_batch_in_flight']['ms_per_step'])"
done
