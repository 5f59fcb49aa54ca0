#!/bin/bash
# the bound of the exact one-lane kernel: cfg2 (one 3840x2160 picture) and a 64-picture cfg3 batch decoded with PJD_F_FORCE_SEQUENTIAL
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload cfg2 --force-exact --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 2 --warmup 1 --no-variants > gpurun_out/r3_exact_cfg2.log 2> gpurun_out/r3_exact_cfg2.err; echo "cfg2 rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_exact_cfg2.log').read().strip().splitlines()[-1])
print('exact kernel, cfg2: value', d['value'], 'MPix/s, ms/step', d['ms_per_step'], d['kernels_ms'])"
timeout -k 10 300 python bench.py --workload cfg3 --images 64 --force-exact --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 3 --warmup 1 --no-variants > gpurun_out/r3_exact_cfg3.log 2> gpurun_out/r3_exact_cfg3.err; echo "cfg3x64 rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_exact_cfg3.log').read().strip().splitlines()[-1])
print('exact kernel, 64 pictures of cfg3: value', d['value'], 'MPix/s, ms/step', d['ms_per_step'], d['kernels_ms'])"
