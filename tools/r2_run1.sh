#!/bin/bash
# round-2 GPU session 1: full GPU suite, default bench, subsequence-size sweep, wave timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
tail -3 gpurun_out/t1.log
timeout -k 10 300 python bench.py > gpurun_out/b1.log 2> gpurun_out/b1.err; echo "bench rc=$?"
for sb in 128 256 512 1024; do
  PJD_SUB_BYTES=$sb timeout -k 10 120 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10 > gpurun_out/sb_$sb.log 2> gpurun_out/sb_$sb.err
  echo "sb $sb rc=$?"
done
PJD_DEBUG_STATS=1 timeout -k 10 120 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/dbg.log 2> gpurun_out/dbg.err
echo done
