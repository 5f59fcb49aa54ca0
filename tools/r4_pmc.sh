#!/bin/bash
# SQ counter passes over the serialised default bench (program directly after `--`, counters in their own runs)
# usage: bash tools/r4_pmc.sh TAG [SUB_BYTES]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-pmc}; SB=${2:-0}
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
export PJD_SUB_BYTES=$SB
# the counter passes decode the batch in ONE chain of launches (as with several batches in flight): per-kernel figures are those of
# whole-batch launches; on an idle device the library issues the same work as two chains (picture groups), traced at the end
export PJD_GROUPS=1
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out"
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > "$GRAFT_REPO_ROOT"/gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 "$GRAFT_REPO_ROOT"/gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1
CMD="python3 $GRAFT_REPO_ROOT/bench.py --workload ${WORKLOAD:-cfg3} --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --steps 10"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/sq1" -- $CMD > "$OUT/sq1.log" 2>&1; echo "sq1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq2" -- $CMD > "$OUT/sq2.log" 2>&1; echo "sq2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_INSTS_SENDMSG --kernel-trace --output-format csv -d "$OUT/sq3" -- $CMD > "$OUT/sq3.log" 2>&1; echo "sq3 rc=$?"
# the same counters for the throughput plan (longer lanes: what the batches kept in flight are planned with)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/sq1_thr" -- $CMD --plan-mode throughput > "$OUT/sq1_thr.log" 2>&1; echo "sq1_thr rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_thr" -- $CMD --plan-mode throughput > "$OUT/stats_thr.log" 2>&1; echo "stats_thr rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1; echo "write rc=$?"
# kernel trace of the default mode (four batches in flight): which kernels overlap (tools/inflight_overlap.py)
CMD3="python3 $GRAFT_REPO_ROOT/bench.py --workload ${WORKLOAD:-cfg3} --no-variants --in-flight 4 --plan-mode throughput --e2e-batches 0 --no-cpu-baseline --no-cli --steps 16"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/inflight" -- $CMD3 > "$OUT/inflight.log" 2>&1; echo "inflight rc=$?"
# one batch alone with the picture groups on (the library's choice on an idle device): which kernels overlap
unset PJD_GROUPS
CMD4="python3 $GRAFT_REPO_ROOT/bench.py --workload ${WORKLOAD:-cfg3} --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --steps 10"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/groups" -- $CMD4 > "$OUT/groups.log" 2>&1; echo "groups rc=$?"
# keep the merge small: the per-dispatch CSVs are enough
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
