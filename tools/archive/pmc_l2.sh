#!/bin/bash
# GPU box: L1 / L2 request counters per kernel (is k_describe's patch gather bound by L2 -> L1 line traffic?).   usage: tools/pmc_l2.sh [bench args]
cd /tmp && export TMPDIR=/tmp ORBX_SPLIT=0
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_l2_$tag
  timeout -k 10 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_l2_$tag --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify "$@" > $R/gpurun_out/pmc_l2_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $R/gpurun_out/pmc_l2_$tag.log; continue; }
  (cd $R && python3 tools/pmc_summary.py $(find gpurun_out/pmc_l2_$tag -name "*counter_collection.csv") | grep -v packedSelf)
done
