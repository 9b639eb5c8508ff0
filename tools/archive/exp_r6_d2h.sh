#!/bin/bash
# Round 6: the result copy of host-buffer batches - hipMemcpyAsync at _begin (0), at _end (1), a copy kernel at _begin (2) - with two handles alternating;
# and the queued quad-tree's scratch-free build at 128 frames per call.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
for cfg in "ORBX_D2H=0" "ORBX_D2H=1" "ORBX_D2H=2"; do
  for rep in 1 2; do echo "[$cfg]"; env $cfg HOST_RATE_BATCHES=8,64,256 timeout -k 10 120 python tools/host_path_rate.py 2>/dev/null; done
done
bash tools/ab_env.sh "--steps 60 --warmup 5 --batch 128" "" "ORBX_OCT_ROOMY=1"
bash tools/ab_env.sh "--steps 40 --warmup 5 --batch 256" "" "ORBX_OCT_ROOMY=1"
bash tools/ab_env.sh "--steps 40 --warmup 5 --batch 128 --workload hd720" "" "ORBX_OCT_ROOMY=1"
bash tools/ab_env.sh "--steps 40 --warmup 5 --batch 64 --workload hd1080" "" "ORBX_OCT_ROOMY=1"
} 2>&1 | tee gpurun_out/exp_r6_d2h.log
