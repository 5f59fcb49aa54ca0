#!/bin/bash
# round-4: the bench's extras on a one-GPU box: cli_end_to_end at N=1, and the N=2 control flow over gloo (PJD_BENCH_BACKEND) with both ranks on device 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
# parity first: a subset of the GPU suite on the build that is about to be timed (a sweep without it once reported a faster kernel that decoded garbage)
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reference_hashes or wrap or random_streams" > gpurun_out/parity_first.log 2>&1; rc=$?; echo "parity subset rc=$rc $(tail -1 gpurun_out/parity_first.log)"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python bench.py --steps 40 > gpurun_out/r4_full.log 2> gpurun_out/r4_full.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_full.log').read().strip().splitlines()[-1])
print('value', d['value'], 'serial', d['one_batch_in_flight']['value'])
print('cli', json.dumps(d.get('cli_end_to_end')))
print('pcie', d.get('pcie_inclusive',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
PJD_BENCH_BACKEND=gloo PJD_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --images 256 --tile 4096 > gpurun_out/r4_n2.log 2> gpurun_out/r4_n2.err; echo "n2 rc=$?"
tail -c 2500 gpurun_out/r4_n2.log; tail -5 gpurun_out/r4_n2.err
