#!/bin/bash
# round-4 baseline: the default bench, then the wave timeline of one batch decoded alone (dump kept for offline analysis)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --e2e-batches 0 --no-cpu-baseline --no-cli > gpurun_out/r4_base.log 2> gpurun_out/r4_base.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r4_base.log
timeout -k 10 300 python tools/r3_tail_probe.py > gpurun_out/r4_tail.log 2>&1; echo "probe rc=$?"
cp /tmp/pjd_dbg.bin gpurun_out/r4_dbg.bin
tail -12 gpurun_out/r4_tail.log
