#!/bin/bash
# round-2 GPU session 2: wave timelines at S=256/128, rocprof kernel stats of the default bench (serialised)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for sb in 256 128; do
  PJD_SUB_BYTES=$sb PJD_DEBUG_STATS=1 timeout -k 10 120 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/dbg_$sb.log 2> gpurun_out/dbg_$sb.err
  echo "dbg $sb rc=$?"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r2a" -o r2a -- python3 "$GRAFT_REPO_ROOT/bench.py" --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 20 > "$GRAFT_REPO_ROOT/gpurun_out/prof_r2a.log" 2>&1
echo "prof rc=$?"
ls "$GRAFT_REPO_ROOT/gpurun_out/prof_r2a" | head
