#!/bin/bash
# the numbers of DESIGN.md's table: every workload of bench.py, one line each
cd "$GRAFT_REPO_ROOT" || exit 1
for w in cfg2 cfg2rst cfg5 cfg5split; do
  timeout -k 10 300 python bench.py --workload $w --e2e-batches 0 --no-cpu-baseline > gpurun_out/t_$w.log 2> gpurun_out/t_$w.err || echo "$w failed"
done
timeout -k 10 300 python bench.py --images 8192 --no-variants --e2e-batches 0 --no-cpu-baseline --steps 6 > gpurun_out/t_cfg3x8.log 2> gpurun_out/t_cfg3x8.err || echo "x8 failed"
timeout -k 10 400 python bench.py > gpurun_out/t_default.log 2> gpurun_out/t_default.err || echo "default failed"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/t_*.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f, d['value'], d['ms_per_step'], d.get('one_batch_in_flight', {}).get('value'), d.get('one_batch_in_flight', {}).get('ms_per_step'), d['kernels_ms'], d.get('pcie_inclusive', {}).get('value'))
PY
