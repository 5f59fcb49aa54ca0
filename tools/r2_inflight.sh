#!/bin/bash
# throughput of the default bench against the number of batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
for nf in 2 3 4 6; do
  PJD_SUB_BYTES=${SB:-0} timeout -k 10 200 python bench.py --in-flight $nf --e2e-batches 0 --no-cpu-baseline --no-variants --steps 24 > gpurun_out/if_$nf.log 2> gpurun_out/if_$nf.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/if_$nf.log').read().strip().splitlines()[-1])
print('in-flight', $nf, 'value', d['value'], 'ms/step', d['ms_per_step'], 'serial', d.get('one_batch_in_flight',{}).get('ms_per_step'))
"
done
