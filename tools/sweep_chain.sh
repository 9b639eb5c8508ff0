#!/bin/bash
cd $GRAFT_REPO_ROOT
for B in 2 4 8 16 32; do for W in 1 1000000; do
  echo -n "B=$B chain_wgs=$W  "
  ORBX_PYR_CHAIN_WGS=$W timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 200 --batch $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done; done
