#!/bin/bash
# the region-major pyramid at large batches.  usage (GPU box): bash tools/ab_cols2.sh
cd $GRAFT_REPO_ROOT
for b in 64 128 512; do for cfg in "0 0" "1 80" "1 112" "1 128"; do set -- $cfg
  ORBX_PYR_COLS=$1 ORBX_PYR_COL_PX=$2 python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('batch $b cols $1 px $2: %.1f us/call  %.0f fps  pyramid %.1f' % (j['ms_per_step']*1e3, j['value'], (r.get('k_resize',0)+r.get('k_pyr_first',0))*1e3), {k: round(v*1e3) for k,v in r.items()})"
done; done
