#!/bin/bash
# occupancy experiments: rebuild the kernels on the box with other checkpoint / staging sizes and run the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${VARIANTS:-"8 32" "6 16" "4 16" "4 8"}; do
  set -- $v
  touch pim-jpeg-decoder_amd/csrc/pjd_internal.h
  make -s -C pim-jpeg-decoder_amd HIPFLAGS_EXTRA="-DPJD_NCHK=$1 -DPJD_STAGE_ENTRIES=$2" all > gpurun_out/occ_build.log 2>&1 || { tail -5 gpurun_out/occ_build.log; exit 1; }
  timeout -k 10 200 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 20 > gpurun_out/occ_$1_$2.log 2> gpurun_out/occ_$1_$2.err || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/occ_$1_$2.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('NCHK $1 STAGE $2: cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], 'huff', d['kernels_ms']['huff_lanes'], 'fb', d['config']['exact_kernel_images'], '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'])"
done
