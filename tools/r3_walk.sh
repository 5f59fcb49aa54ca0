#!/bin/bash
# round 3: the cooperative walker -- wave timeline (PJD_DEBUG_STATS) alone and with four batches in flight, then the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for n in 1 4; do
  PJD_DEBUG_STATS=1 timeout -k 10 200 python bench.py --in-flight $n --e2e-batches 0 --no-cpu-baseline --steps 40 --no-variants > gpurun_out/r3_walk_clk_$n.log 2> gpurun_out/r3_walk_clk_$n.err; echo "in flight $n rc=$?"
  grep "shader clock\|pjd waves\] n \|last to finish\|rounds (lanes" gpurun_out/r3_walk_clk_$n.err | head -8
done
timeout -k 10 400 python bench.py --steps 100 --warmup 8 --e2e-batches 0 --no-cpu-baseline > gpurun_out/r3_walk_bench.log 2> gpurun_out/r3_walk_bench.err || { echo bench failed; tail -5 gpurun_out/r3_walk_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_walk_bench.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], d.get('huffman_passes'), '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'])
PY
