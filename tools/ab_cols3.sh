#!/bin/bash
# workgroup shapes of k_pyr_cols (ORBX_PYR_COLS_VARIANT) x region size per batch size.  usage (GPU box): bash tools/ab_cols3.sh
cd $GRAFT_REPO_ROOT
for b in 1 2 4 8 16 32 64; do for v in 0 1 2 3; do for px in 40 56 80 112; do
  ORBX_PYR_COLS=1 ORBX_PYR_COLS_VARIANT=$v ORBX_PYR_COL_PX=$px python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b variant $v px $px: %.1f us/call' % (j['ms_per_step']*1e3))"
done; done; done
