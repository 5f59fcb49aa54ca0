#!/bin/bash
# in-flight throughput of the default bench against extra (unused) LDS per Huffman workgroup: is the kernel occupancy-bound?
cd "$GRAFT_REPO_ROOT" || exit 1
for x in ${XS:-0 8192 16384 32768}; do
  PJD_EXTRA_LDS=$x timeout -k 10 200 python bench.py --e2e-batches 0 --no-cpu-baseline --no-variants --steps 20 > gpurun_out/lds_$x.log 2> gpurun_out/lds_$x.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/lds_$x.log').read().strip().splitlines()[-1])
print('extra LDS', $x, 'value', d['value'], 'ms/step', d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'])"
done
