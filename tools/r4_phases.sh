#!/bin/bash
# back end by phase: the kernel cut short after set-up / parse / DC + rows 1..7 / row 0 + columns (pictures are wrong: timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first, on the product build (the builds timed below stop the back end early: their pictures are wrong by design): a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
for k in 0 1 2 3 none; do
  touch pim-jpeg-decoder_amd/csrc/pjd_internal.h
  if [ $k = none ]; then F=""; else F="-DPJD_IDCT_STOP_AFTER=$k"; fi
  make -s -C pim-jpeg-decoder_amd HIPFLAGS_EXTRA="$F" all > gpurun_out/ph_build.log 2>&1 || { tail -5 gpurun_out/ph_build.log; exit 1; }
  timeout -k 10 200 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --no-variants --steps 30 > gpurun_out/ph_$k.log 2> gpurun_out/ph_$k.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/ph_$k.log').read().strip().splitlines()[-1])
print('stop after $k: idct_colour', d['kernels_ms']['idct_colour'], 'huff', d['kernels_ms']['huff_lanes'])"
done
