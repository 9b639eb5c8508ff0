#!/bin/bash
# A/B of the small-batch quad-tree workgroup size (ORBX_OCT_SMALL_T) per workload and batch.  usage (GPU box): bash tools/ab_smallT.sh
cd $GRAFT_REPO_ROOT
for wl in mono640 stereo640; do for b in 1 2 4 8; do for t in 0 256 512 1024; do
  ORBX_OCT_SMALL_T=$t python bench.py --workload $wl --batch $b --steps 400 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('$wl batch $b smallT $t: %.1f us/call  k_octree %.1f' % (j['ms_per_step']*1e3, r.get('k_octree',0)*1e3))"
done; done; done
