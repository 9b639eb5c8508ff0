#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
for cfg in "" "HOST_RATE_LAP=1" "HOST_RATE_TORCH=1" "HOST_RATE_TORCH=2" "HOST_RATE_TORCH=1 GPU_MAX_HW_QUEUES=8"; do
  for rep in 1 2; do echo "[$cfg]"; env $cfg HOST_RATE_BATCHES=64 timeout -k 10 200 python tools/host_path_rate.py 2>/dev/null | sed 's/sync.*pipelined/pipelined/'; done
done
} 2>&1 | tee gpurun_out/exp_r6_host2.log
