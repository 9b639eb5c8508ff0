#!/bin/bash
# A/B of library builds inside ONE gpurun call (memory-bound kernels are bimodal between processes: alternate and repeat)
# usage: tools/ab_lib.sh "<lib1> <lib2> ..."  ("" = the default build)
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}"
for rep in 1 2; do for lib in $1; do [ "$lib" = default ] && lib=""; for args in "--steps 30" "--steps 300 --batch 1" "--steps 100 --batch 64" "--steps 30 --workload hd1080" "--steps 30 --variant natural"; do
  ORBX_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${lib:-default}', '$args', d['value'], d['ms_per_step'], {k: round(v * 1e3) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
done; done; done
