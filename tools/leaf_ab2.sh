#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in stereo640 mono640; do for lf in 0 8; do
  ORBX_LEAF_FRAMES=$lf python bench.py --workload $wl --batch 2 --steps 400 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('$wl batch 2 leaf_frames $lf: %.1f us/call  kernels(us): %s' % (j['ms_per_step']*1e3, {k: round(v*1e3,1) for k,v in r.items()}))"
done; done
