#!/bin/bash
# Longer fuzz soaks of round 4's forms on the GPU box (one gpurun call): single frames and batch shapes under the default policy, the patch blur,
# the blurring pyramid, the copy-back host path and the forced overlap forms.  Totals -> gpurun_out/r4_soak.txt
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}"; OUT=gpurun_out/r4_soak.txt; : > $OUT
run() { echo "## $1 :: $2" >> $OUT; env $1 timeout -k 10 600 python $2 2>&1 | tail -1 >> $OUT || echo "FAILED" >> $OUT; }
run "" "tools/fuzz_parity.py 150 401"
run "ORBX_PATCH_BLUR=1" "tools/fuzz_parity.py 120 402"
run "ORBX_PYR_COLS=1 ORBX_BLUR_IN_COLS=1" "tools/fuzz_parity.py 80 403"
run "ORBX_PYR_COLS=1 ORBX_BLUR_IN_COLS=1 ORBX_BLUR_IN_LEVELS=3 ORBX_PYR_COL_PX=56" "tools/fuzz_parity.py 60 404"
run "ORBX_ZERO_COPY=0" "tools/fuzz_parity.py 60 405"
run "" "tools/fuzz_batches.py 50 406"
run "ORBX_SPLIT_MIN_MPX=0" "tools/fuzz_batches.py 40 407"
run "ORBX_SPLIT_MIN_MPX=0 ORBX_SPLIT=3" "tools/fuzz_batches.py 40 408"
run "ORBX_PATCH_BLUR=1 ORBX_SPLIT_MIN_MPX=0" "tools/fuzz_batches.py 40 409"
cat $OUT
