#!/bin/bash
# A/B of library builds on the large-batch workloads (alternating, two rounds).  usage: tools/ab_lib2.sh "<lib1> <lib2> ..."   ("default" = the in-tree build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in $1; do [ "$lib" = default ] && lib=""; for args in "--steps 30" "--steps 30 --workload hd1080"; do
  ORBX_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-verify $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${lib:-default}', '$args', d['value'], d['ms_per_step'], {k: round(v * 1e3) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
done; done; done
