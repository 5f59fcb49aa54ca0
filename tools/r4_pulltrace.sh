#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
# kernel trace of one batch alone in the form PJD_IDLE_FORM selects (used for the pull back end, profiles/r04_experiments.md #16)
OUT="$GRAFT_REPO_ROOT/gpurun_out/pulltrace"; mkdir -p "$OUT"
# parity first: a subset of the GPU suite on the build (and form) that is about to be traced
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > "$OUT/parity_first.log" 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 "$OUT/parity_first.log")"; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --steps 10 > "$OUT/log" 2>&1; echo rc=$?
find "$OUT" -name "*.db" -delete
