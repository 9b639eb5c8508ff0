#!/bin/bash
# Round 6: the two description launches of a split blur side by side (ORBX_DESC_SIDE=1) against one after the other; the stagger on / off (ORBX_SPLIT_MIN_MPX=170: big-batch overlap, no stagger at 512 frames)
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
bash tools/ab_env.sh "--steps 40 --warmup 5" "" "ORBX_DESC_SIDE=1" "ORBX_SPLIT_MIN_MPX=170" "ORBX_DESC_SIDE=1 ORBX_SPLIT_MIN_MPX=170" "ORBX_DESC_SIDE=1 ORBX_BLUR_SPLIT=2" "ORBX_DESC_SIDE=1 ORBX_BLUR_SPLIT=4"
bash tools/ab_env.sh "--steps 40 --warmup 5 --batch 256" "" "ORBX_DESC_SIDE=1"
bash tools/ab_env.sh "--steps 40 --warmup 5 --workload stereo640" "" "ORBX_DESC_SIDE=1"
} 2>&1 | tee gpurun_out/exp_r6_descside.log
