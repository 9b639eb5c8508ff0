#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
for cfg in "ORBX_D2H=1" "ORBX_D2H=2" "ORBX_D2H=2 ORBX_SPLIT=0" "ORBX_D2H=2 ORBX_SPLIT_MIN_MPX=1000" "ORBX_D2H=1 ORBX_SPLIT=0"; do
  echo "[$cfg]"; env $cfg HOST_RATE_BATCHES=224,248,256,320,512 timeout -k 10 200 python tools/host_path_rate.py 2>/dev/null | sed 's/sync.*pipelined/pipelined/'
done
} 2>&1 | tee gpurun_out/exp_r6_d2h3.log
