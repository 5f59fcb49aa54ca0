#!/bin/bash
# round 3: back-end parser variants (PJD_PARSE_SEGS): parity subset, parser slot statistics, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
IFS=';' read -ra VS <<< "${VARIANTS:-;}"
k=0
for v in "${VS[@]}"; do
  k=$((k+1))
  touch pim-jpeg-decoder_amd/csrc/pjd_internal.h
  make -s -C pim-jpeg-decoder_amd HIPFLAGS_EXTRA="$v" all > gpurun_out/be_build.log 2>&1 || { tail -5 gpurun_out/be_build.log; exit 1; }
  timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "${TESTS:-matches_oracle or coefficients or routing or cfg4 or config4 or corrupt or long_tail}" > gpurun_out/be_test_$k.log 2>&1; rc=$?; echo "[$v] tests rc=$rc $(tail -1 gpurun_out/be_test_$k.log)"
  [ $rc -eq 0 ] || { grep -E "^(E|FAILED)" gpurun_out/be_test_$k.log | head -10; continue; }
  PJD_DEBUG_STATS=1 timeout -k 10 200 python bench.py --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 10 --no-variants > /dev/null 2> gpurun_out/be_dbg_$k.err; grep "back end\]" gpurun_out/be_dbg_$k.err | head -1
  timeout -k 10 300 python bench.py --e2e-batches 0 --no-cpu-baseline --steps 60 > gpurun_out/be_$k.log 2> gpurun_out/be_$k.err || { echo "variant $v failed"; tail -3 gpurun_out/be_$k.err; continue; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/be_$k.log').read().strip().splitlines()[-1])
v=d['variants']['cfg3lite']
print('[$v] cfg3', d['value'], d['ms_per_step'], 'serial', d['one_batch_in_flight']['ms_per_step'], d['kernels_ms'], '| lite', v['value'], v['ms_per_step'], 'serial', v['one_batch_in_flight']['ms_per_step'], v.get('kernels_ms'))"
  timeout -k 10 120 python bench.py --workload cfg2 --in-flight 1 --e2e-batches 0 --no-cpu-baseline --steps 50 --no-variants > gpurun_out/be_cfg2_$k.log 2>/dev/null && python3 -c "
import json
d=json.loads(open('gpurun_out/be_cfg2_$k.log').read().strip().splitlines()[-1])
print('[$v] cfg2 ms/step', d['ms_per_step'], d['kernels_ms'])"
done
touch pim-jpeg-decoder_amd/csrc/pjd_internal.h; make -s -C pim-jpeg-decoder_amd all > /dev/null 2>&1
