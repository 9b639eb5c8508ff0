#!/bin/bash
# one-frame A/B of library builds inside ONE gpurun call: tools/ab_b1.sh "<lib1> <lib2> ..." ("default" = the in-tree build)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in $1; do l=$lib; [ "$lib" = default ] && l=""; ORBX_LIBRARY=$l python bench.py --batch 1 --steps 400 --warmup 20 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', round(d['ms_per_step']*1e3,1), 'us', {k: round(v * 1e3,1) for k, v in d['roofline']['kernel_ms_per_step'].items()})"; done; done
