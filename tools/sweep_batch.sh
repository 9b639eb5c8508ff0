#!/bin/bash
# bench.py over frames-per-call, handles and the two-half overlap (one gpurun call: same box for every row)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/sweep_batch.txt
for B in 256 512 768 1024; do for H in 1 2; do for S in 1 0; do
  echo -n "B=$B handles=$H split=$S  " >> gpurun_out/sweep_batch.txt
  ORBX_SPLIT=$S timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 30 --batch $B --handles $H 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep_batch.txt
done; done; done
cat gpurun_out/sweep_batch.txt
