#!/bin/bash
# round 3, GPU call 2: the split-image tests, wave timelines of cfg2 and cfg3 (one batch at a time)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "split" > gpurun_out/r3_split.log 2>&1; rc=$?; echo "split tests rc=$rc"; tail -3 gpurun_out/r3_split.log
[ $rc -eq 0 ] || grep -E "^(E|FAILED)" gpurun_out/r3_split.log | head -40
PJD_DEBUG_STATS=1 timeout -k 10 120 python bench.py --workload cfg2 --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 20 --no-variants > gpurun_out/r3_cfg2_dbg.log 2> gpurun_out/r3_cfg2_dbg.err; echo "cfg2 rc=$?"
grep "pjd waves" gpurun_out/r3_cfg2_dbg.err | head -12
PJD_DEBUG_STATS=1 timeout -k 10 200 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10 --no-variants > gpurun_out/r3_cfg3_dbg.log 2> gpurun_out/r3_cfg3_dbg.err; echo "cfg3 rc=$?"
grep "pjd waves" gpurun_out/r3_cfg3_dbg.err | head -12
