#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
for cfg in "HOST_RATE_TORCH=3" "HOST_RATE_TORCH=4" "HOST_RATE_TORCH=1 ORBX_SPIN_WAIT=1" "HOST_RATE_TORCH=3 ORBX_SPIN_WAIT=1" "ORBX_SPIN_WAIT=1" "HOST_RATE_TORCH=3 OMP_NUM_THREADS=1" "HOST_RATE_TORCH=3 HIP_FORCE_DEV_KERNARG=0"; do
  echo "[$cfg]"; env $cfg HOST_RATE_BATCHES=64 timeout -k 10 200 python tools/host_path_rate.py 2>/dev/null | sed 's/sync.*pipelined/pipelined/'
done
python - <<'PY'
import os
import torch
print({k: v for k, v in os.environ.items() if k.startswith(("HSA", "HIP", "GPU_", "AMD", "ROC"))})
PY
} 2>&1 | tee gpurun_out/exp_r6_host3.log
