#!/bin/bash
# Round 6: A/B of the blur split by level (ORBX_BLUR_SPLIT=L) against the unsplit patch blur and the separate k_blur, inside ONE gpurun call.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
bash tools/ab_env.sh "--steps 40 --warmup 5" "" "ORBX_BLUR_SPLIT=2" "ORBX_BLUR_SPLIT=3" "ORBX_BLUR_SPLIT=4" "ORBX_BLUR_SPLIT=5" "ORBX_PATCH_BLUR=0"
bash tools/ab_env.sh "--steps 20 --warmup 3 --workload hd1080" "" "ORBX_BLUR_SPLIT=4" "ORBX_BLUR_SPLIT=6" "ORBX_BLUR_SPLIT=7" "ORBX_PATCH_BLUR=0"
} 2>&1 | tee gpurun_out/ab_split.log
