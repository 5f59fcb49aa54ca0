#!/bin/bash
# round-4: config 2 with 64-byte lanes (a build with PJD_SUB_BYTES_MIN=64 in lib/var)
cd "$GRAFT_REPO_ROOT" || exit 1
L=pim-jpeg-decoder_amd/lib
cp $L/libpjd.so /tmp/keep.so; cp $L/var/libpjd_min64.so $L/libpjd.so
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "4k or reference_hashes or wrap" > gpurun_out/cfg2b_test.log 2>&1; echo "parity subset rc=$? $(tail -1 gpurun_out/cfg2b_test.log)"
for sb in 64 128; do
  PJD_SUB_BYTES=$sb timeout -k 10 200 python bench.py --workload cfg2 --e2e-batches 0 --no-cpu-baseline --no-cli --no-variants --steps 200 > gpurun_out/cfg2b_$sb.log 2> gpurun_out/cfg2b_$sb.err || { echo "$sb failed"; tail -3 gpurun_out/cfg2b_$sb.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/cfg2b_$sb.log').read().strip().splitlines()[-1])
print('S=$sb ->', d['one_batch_in_flight']['sub_bytes'], 'in flight', d['value'], 'serial', d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['kernels_ms']['huff_lanes'], 'lanes', d['one_batch_in_flight']['huffman_lanes'])"
done
PJD_SUB_BYTES=64 timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams or 4k" > gpurun_out/cfg2b_test64.log 2>&1; echo "parity with S=64 rc=$? $(tail -1 gpurun_out/cfg2b_test64.log)"
cp /tmp/keep.so $L/libpjd.so
