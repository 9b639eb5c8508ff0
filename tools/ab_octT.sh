#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for T in default 512 256; do
  if [ $T = default ]; then E=""; else E="ORBX_OCT_THREADS=$T ORBX_OCT_ROOMY=1"; fi
  env $E python bench.py --batch 1 --steps 400 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$T', round(d['ms_per_step']*1e3,1), 'us', {k: round(v * 1e3,1) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
done; done
