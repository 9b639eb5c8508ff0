#!/bin/bash
# A/B of the region-major pyramid (ORBX_PYR_COLS=1, region side ORBX_PYR_COL_PX) against the tile forms per batch size.  usage (GPU box): bash tools/ab_cols.sh [workload]
cd $GRAFT_REPO_ROOT
WL=${1:-mono640}
for b in 1 2 4 8 16 32; do for cfg in "0 0" "1 40" "1 56" "1 80" "1 112"; do set -- $cfg
  ORBX_PYR_COLS=$1 ORBX_PYR_COL_PX=$2 python bench.py --workload $WL --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('$WL batch $b cols $1 px $2: %.1f us/call  pyramid %.1f' % (j['ms_per_step']*1e3, (r.get('k_resize',0)+r.get('k_pyr_first',0))*1e3))"
done; done
