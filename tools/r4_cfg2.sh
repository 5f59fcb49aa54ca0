#!/bin/bash
# round-4: BASELINE config 2 (one 3840x2160 picture) against the subsequence size
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "4k or config2 or cfg2" > gpurun_out/cfg2_test.log 2>&1; echo "parity subset rc=$? $(tail -1 gpurun_out/cfg2_test.log)"
for sb in 0 128 192 256 320 384 512; do
  PJD_SUB_BYTES=$sb timeout -k 10 200 python bench.py --workload cfg2 --e2e-batches 0 --no-cpu-baseline --no-cli --no-variants --steps 200 > gpurun_out/cfg2_$sb.log 2> gpurun_out/cfg2_$sb.err || { echo "$sb failed"; tail -3 gpurun_out/cfg2_$sb.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/cfg2_$sb.log').read().strip().splitlines()[-1])
print('S=$sb ->', d['one_batch_in_flight']['sub_bytes'], 'in flight', d['value'], 'serial', d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['kernels_ms']['huff_lanes'], 'lanes', d['one_batch_in_flight']['huffman_lanes'])"
done
