#!/bin/bash
# round 3, GPU call 1: the GPU suite, the graph-memset probe (once), a default bench line, cfg2 with several subsequence sizes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3_tests.log
[ $rc -eq 0 ] || { grep -E "^(E|FAILED)" gpurun_out/r3_tests.log | head -30; exit $rc; }
timeout -k 10 240 bin/graph_memset_probe 2000 > gpurun_out/r03_graph_memset_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r03_graph_memset_probe.log
timeout -k 10 400 python bench.py --steps 40 --warmup 8 > gpurun_out/r3_bench.log 2> gpurun_out/r3_bench.err || { echo bench failed; tail -5 gpurun_out/r3_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench.log').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], 'pcie', d.get('pcie_inclusive',{}).get('value'), 'lite', d['variants']['cfg3lite']['value'])
PY
for sb in 0 128 512; do
  PJD_SUB_BYTES=$sb timeout -k 10 120 python bench.py --workload cfg2 --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 50 --no-variants > gpurun_out/r3_cfg2_$sb.log 2> gpurun_out/r3_cfg2_$sb.err || { echo "cfg2 sb=$sb failed"; tail -3 gpurun_out/r3_cfg2_$sb.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/r3_cfg2_$sb.log').read().strip().splitlines()[-1])
print('cfg2 sb', '$sb', 'ms/step', d['ms_per_step'], d['kernels_ms'], d['config']['sub_bytes'], d['config']['sync'])
"
done
