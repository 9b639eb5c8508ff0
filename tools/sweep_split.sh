#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 ./tools/ubench/valu_rate csv > gpurun_out/r02_valu_rate2.csv 2>/dev/null
: > gpurun_out/sweep_split.txt
for B in 512 256 128; do for M in 0 1 2 3; do
  echo -n "mono640 B=$B split=$M  " >> gpurun_out/sweep_split.txt
  ORBX_SPLIT=$M ORBX_SPLIT_MIN_MPX=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --batch $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep_split.txt
done; done
for B in 128 32; do for M in 0 1 2 3; do
  echo -n "hd1080 B=$B split=$M  " >> gpurun_out/sweep_split.txt
  ORBX_SPLIT=$M ORBX_SPLIT_MIN_MPX=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --workload hd1080 --batch $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep_split.txt
done; done
cat gpurun_out/sweep_split.txt
