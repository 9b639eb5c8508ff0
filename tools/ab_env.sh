#!/bin/bash
# Generic A/B of environment switches inside ONE gpurun call (alternating; memory-bound kernels are bimodal between processes).
# usage (GPU box): bash tools/ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...      (each quoted group = one configuration; "" = defaults)
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}"
ARGS=$1; shift
for rep in 1 2; do for cfg in "$@"; do
  env $cfg python bench.py $ARGS --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']['kernel_ms_per_step']
print('[%s] %s: %.1f us/step %.0f fps | %s' % ('$cfg', '$ARGS', j['ms_per_step']*1e3, j['value'], ' '.join('%s %.0f' % (k, v*1e3) for k, v in r.items())))"
done; done
