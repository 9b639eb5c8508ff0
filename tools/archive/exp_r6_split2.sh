#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
bash tools/ab_env.sh "--steps 40 --warmup 5 --workload stereo640" "" "ORBX_PATCH_BLUR=1" "ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=3" "ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=1"
bash tools/ab_env.sh "--steps 40 --warmup 5 --workload hd720" "" "ORBX_BLUR_SPLIT=0" "ORBX_BLUR_SPLIT=6"
bash tools/ab_env.sh "--steps 40 --warmup 5 --batch 256" "" "ORBX_BLUR_SPLIT=0"
bash tools/ab_env.sh "--steps 60 --warmup 5 --batch 128" "" "ORBX_PATCH_BLUR=1" "ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=0"
bash tools/ab_env.sh "--steps 100 --warmup 5 --batch 64" "" "ORBX_PATCH_BLUR=1"
} 2>&1 | tee gpurun_out/exp_r6_split2.log
