#!/bin/bash
# wave timeline + shader clock of the entropy decoder: a decode issued with four batches in flight vs one alone (PJD_DEBUG_STATS)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PJD_DEBUG_STATS=1 timeout -k 10 200 python bench.py --in-flight 4 --e2e-batches 0 --no-cpu-baseline --steps 40 --no-variants > gpurun_out/r3_clk_4.log 2> gpurun_out/r3_clk_4.err; echo "rc=$?"
grep "bench\]\|shader clock\|pjd waves\] n \|last to finish" gpurun_out/r3_clk_4.err | head -12
