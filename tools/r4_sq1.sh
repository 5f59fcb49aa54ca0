#!/bin/bash
# round-4: one SQ counter pass (waits, instruction counts) over the serialised default bench, latency and throughput plan
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="$GRAFT_REPO_ROOT/gpurun_out/${1:-sq1}"; mkdir -p "$OUT"
export PJD_GROUPS=1
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out"
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > "$GRAFT_REPO_ROOT"/gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 "$GRAFT_REPO_ROOT"/gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --steps 10"
for m in latency throughput; do
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/$m" -- $CMD --plan-mode $m > "$OUT/$m.log" 2>&1; echo "$m rc=$?"
done
find "$OUT" -name "*.db" -delete
