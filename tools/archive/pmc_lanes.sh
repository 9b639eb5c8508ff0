#!/bin/bash
# GPU box: lane utilisation and instruction counts per kernel (SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)), optionally for other builds of the
# library (diagnostic knock-outs).   usage: tools/pmc_lanes.sh [library ...]
cd /tmp && export TMPDIR=/tmp ORBX_SPLIT=0
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}
for lib in "" "$@"; do
  tag=$(echo "${lib:-default}" | tr '/.' '__')
  [ -n "$lib" ] && export ORBX_LIBRARY=$R/$lib
  rm -rf $R/gpurun_out/pmc_lanes_$tag
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_lanes_$tag --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-verify $PMC_BENCH_ARGS > $R/gpurun_out/pmc_lanes_$tag.log 2>&1
  echo "## ${lib:-default}"
  (cd $R && python3 tools/pmc_summary.py $(find gpurun_out/pmc_lanes_$tag -name "*counter_collection.csv") | grep -v packedSelf)
done
