#!/bin/bash
# parity subset + FETCH_SIZE / WRITE_SIZE of the decode kernels on cfg3 (one counter per pass) + default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu  > gpurun_out/tr_tests.log 2>&1; rc=$?; echo "tests rc=$rc $(tail -1 gpurun_out/tr_tests.log)"
[ $rc -eq 0 ] || exit $rc
OUT="$GRAFT_REPO_ROOT/gpurun_out/tr"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-variants --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1; echo "write rc=$?"
find "$OUT" -name "*.db" -delete
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
for kind in ("fetch", "write"):
    f = sorted(glob.glob(f"gpurun_out/tr/{kind}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    print(kind, {k: round(sorted(v)[len(v)//2] / 1024, 1) for k, v in per.items() if k.startswith("pjd_k_") and sorted(v)[len(v)//2] > 1024}, "MiB raw (reads x2 on gfx950)")
PY
timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 60 --no-variants 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3', d['value'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'])"
