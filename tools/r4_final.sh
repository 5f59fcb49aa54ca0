#!/bin/bash
# round-4: the default bench exactly as the driver runs it (N = 1), twice
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
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
