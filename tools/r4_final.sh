#!/bin/bash
# round-4: the default bench exactly as the driver runs it (N = 1), twice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for k in 1 2; do
  timeout -k 10 560 python bench.py > gpurun_out/r4_final_$k.log 2> gpurun_out/r4_final_$k.err; echo "bench $k rc=$?"
done
python3 - <<'PY'
import json
for k in (1,2):
    d=json.loads(open(f'gpurun_out/r4_final_{k}.log').read().strip().splitlines()[-1])
    print(k,'value', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'])
    print('  lite', d['variants']['cfg3lite']['value'], d['variants']['cfg3lite']['one_batch_in_flight'])
    print('  cli', json.dumps(d.get('cli_end_to_end'))[:600])
    print('  pcie', d.get('pcie_inclusive',{}).get('value'), d.get('pcie_inclusive',{}).get('d2h_GBps'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('all_cores'))
    print('  roofline', d['roofline'], 'sym/s', d['huffman_symbols_per_s'], 'ecs', d['ecs_GBps'])
PY
