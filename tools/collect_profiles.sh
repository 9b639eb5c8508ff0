#!/bin/bash
# Runs on the GPU box (gpurun): kernel stats, HBM traffic and SQ counter passes for the default bench workload,
# summaries into gpurun_out/prof_<tag>/.   usage: tools/collect_profiles.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one launch = one batch: the two-half overlap (default for large batches) is switched off, as in bench.py's own event-profiled pass
export ORBX_SPLIT=0
python3 -c "import sys; sys.path.insert(0, '$R'); import extractorb_amd as X; print(X.source_hash())" > $OUT/source_hash.txt
CMD="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- $CMD > $OUT/stats.log 2>&1 || exit 1
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- $CMD > $OUT/write.log 2>&1 || exit 1
echo "traffic done"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/sqA --output-format csv -- $CMD > $OUT/sqA.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $OUT/sqB --output-format csv -- $CMD > $OUT/sqB.log 2>&1 || exit 1
echo "sq done"
cd $R
python3 tools/pmc_summary.py $(find $OUT/sqA $OUT/sqB -name "*counter_collection.csv") > $OUT/sq_summary.md
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $OUT/fetch.csv
cp $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/write.csv
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sqA $OUT/sqB
tail -1 $OUT/stats.log
cat $OUT/sq_summary.md
