#!/bin/bash
# small-batch A/B of the leaf tables (ORBX_LEAF_FRAMES=0 off / default on) inside ONE gpurun call: ms per call at 1, 2, 4, 8 frames
cd $GRAFT_REPO_ROOT
for b in 1 2 4 8; do for lf in 0 8; do
  ORBX_LEAF_FRAMES=$lf python bench.py --batch $b --steps 400 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('batch %d leaf_frames %s: %.1f us/call  kernels(us): %s' % ($b, '$lf', j['ms_per_step']*1e3, {k: round(v*1e3,1) for k,v in r.items()}))"
done; done
