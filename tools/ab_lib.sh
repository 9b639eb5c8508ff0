#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" extractorb_amd/liborbx_w6.so extractorb_amd/liborbx_w5.so extractorb_amd/liborbx_w4.so; do for args in "--steps 30" "--steps 30 --workload hd1080" "--steps 30 --variant natural"; do
  ORBX_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', '$args', d['value'], d['ms_per_step'], {k: round(v * 1e3) for k, v in d['roofline']['kernel_ms_per_step'].items()})"
done; done; done
