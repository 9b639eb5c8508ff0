#!/bin/bash
# The long form of tools/soak_r4.sh (about ten minutes of GPU time).  Totals -> gpurun_out/r4_soak_long.txt
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}"; OUT=gpurun_out/r4_soak_long.txt; : > $OUT
run() { echo "## $1 :: $2" >> $OUT; env $1 timeout -k 10 900 python $2 2>&1 | tail -1 >> $OUT || echo "FAILED" >> $OUT; }
run "" "tools/fuzz_parity.py 700 501"
run "ORBX_PATCH_BLUR=1" "tools/fuzz_parity.py 500 502"
run "ORBX_PYR_COLS=1 ORBX_BLUR_IN_COLS=1" "tools/fuzz_parity.py 300 503"
run "" "tools/fuzz_batches.py 200 506"
run "ORBX_SPLIT_MIN_MPX=0 ORBX_SPLIT=3" "tools/fuzz_batches.py 120 508"
run "ORBX_PATCH_BLUR=1 ORBX_SPLIT_MIN_MPX=0" "tools/fuzz_batches.py 120 509"
cat $OUT
