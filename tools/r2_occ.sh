#!/bin/bash
# occupancy experiments: rebuild the kernels on the box with other build options and run the default bench
#   VARIANTS="flags;flags;..."   e.g. "-DPJD_NCHK=4 -DPJD_STAGE_ENTRIES=16;-DPJD_HUFF_WAVES=2"
cd "$GRAFT_REPO_ROOT" || exit 1
IFS=';' read -ra VS <<< "${VARIANTS:--DPJD_HUFF_WAVES=4;-DPJD_HUFF_WAVES=2;-DPJD_HUFF_WAVES=2 -DPJD_NCHK=4 -DPJD_STAGE_ENTRIES=16;-DPJD_HUFF_WAVES=1 -DPJD_NCHK=4 -DPJD_STAGE_ENTRIES=16}"
k=0
for v in "${VS[@]}"; do
  k=$((k+1))
  touch pim-jpeg-decoder_amd/csrc/pjd_internal.h
  make -s -C pim-jpeg-decoder_amd HIPFLAGS_EXTRA="$v" all > gpurun_out/occ_build.log 2>&1 || { tail -5 gpurun_out/occ_build.log; exit 1; }
  timeout -k 10 200 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 20 > gpurun_out/occ_$k.log 2> gpurun_out/occ_$k.err || { echo "variant $v failed"; tail -3 gpurun_out/occ_$k.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/occ_$k.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('$v: cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'], 'fb', d['config']['exact_kernel_images'], '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'])"
done
