#!/bin/bash
# pipeline download experiments, all on one box: copy kernel on the shared download stream, by workgroups / slots
cd "$GRAFT_REPO_ROOT" || exit 1
export PJD_PIPE_TRACE=1
for w in ${WGS:-8 16 32 64}; do
  PJD_DOWNLOAD=kernel PJD_COPY_WGS=$w timeout -k 10 200 python tools/pipe_probe.py --slots ${SLOTS:-3,4} > gpurun_out/pp_k$w.json 2> gpurun_out/pp_k$w.err || exit 1
  echo "== wgs $w"; cat gpurun_out/pp_k$w.json
done
timeout -k 10 200 python tools/pipe_probe.py --slots 3 > gpurun_out/pp_sdma.json 2> gpurun_out/pp_sdma.err || exit 1
echo "== sdma"; cat gpurun_out/pp_sdma.json
