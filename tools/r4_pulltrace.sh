#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="$GRAFT_REPO_ROOT/gpurun_out/pulltrace"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --no-cli --steps 10 > "$OUT/log" 2>&1; echo rc=$?
find "$OUT" -name "*.db" -delete
