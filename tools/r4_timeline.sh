#!/bin/bash
# wave timeline of one batch decoded alone (dump kept for offline analysis)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/r3_tail_probe.py > gpurun_out/r4_tail.log 2>&1; echo "probe rc=$?"
cp /tmp/pjd_dbg.bin gpurun_out/r4_dbg.bin
tail -5 gpurun_out/r4_tail.log
