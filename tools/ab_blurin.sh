#!/bin/bash
# A/B of the blur inside the region-major pyramid (ORBX_BLUR_IN_COLS=1, the finest ORBX_BLUR_IN_LEVELS levels) against the separate k_blur, per
# batch size, alternating inside ONE call (memory-bound kernels are bimodal between processes).
# usage (GPU box): CFGS="in,levels,shape ..." bash tools/ab_blurin.sh [workload] [batches]
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}"
WL=${1:-mono640}; BS=${2:-"512 128 32"}
for b in $BS; do for cfg in ${CFGS:-0,5,-1 1,3,-1 1,5,-1 0,5,-1}; do IFS=, read a l v <<< "$cfg"
  ORBX_SPLIT=0 ORBX_BLUR_IN_COLS=$a ORBX_BLUR_IN_LEVELS=$l ORBX_PYR_COLS_VARIANT=$v python bench.py --workload $WL --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('$WL batch $b blur-in $a levels $l shape $v: %.1f us/step  pyramid %.1f  k_blur %.1f  fast %.1f  sum %.1f' % (j['ms_per_step']*1e3, (r.get('k_resize',0)+r.get('k_pyr_first',0))*1e3, r.get('k_blur',0)*1e3, r.get('k_fast',0)*1e3, sum(r.values())*1e3))"
done; done
