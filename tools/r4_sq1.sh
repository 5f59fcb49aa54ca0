#!/bin/bash
# round-4: one SQ counter pass (waits, instruction counts) over the serialised default bench, latency and throughput plan
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-sq1}"; mkdir -p "$OUT"
export PJD_GROUPS=1
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --steps 10"
for m in latency throughput; do
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/$m" -- $CMD --plan-mode $m > "$OUT/$m.log" 2>&1; echo "$m rc=$?"
done
find "$OUT" -name "*.db" -delete
