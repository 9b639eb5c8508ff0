#!/bin/bash
# Fuzz soak of the second half of round 4 (one polarity per pixel in k_fast's score pass, pair ballots in its NMS (the prefilter
# variant was still there when this ran: rows 602 / 603 / 614 forced it on and off), k_pyr_cols' bank row records / host-made dealing / straight writer runs, the blur's dot4 horizontal sums and saturating pack) on the GPU
# box: every FAST variant forced, both pyramid forms, every blur form.  Totals -> gpurun_out/r4b_soak.txt
cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}"; OUT=gpurun_out/r4b_soak.txt; : > $OUT
run() { echo "## $1 :: $2" >> $OUT; env $1 timeout -k 10 600 python $2 2>&1 | tail -1 >> $OUT || echo "FAILED" >> $OUT; }
run "" "tools/fuzz_parity.py 300 601"
run "" "tools/fuzz_parity.py 200 602"
run "" "tools/fuzz_parity.py 200 603"
run "ORBX_FAST_WIDE=1" "tools/fuzz_parity.py 150 604"
run "ORBX_FAST_WIDE=0" "tools/fuzz_parity.py 150 605"
run "ORBX_PYR_COLS=0" "tools/fuzz_parity.py 150 606"
run "ORBX_PYR_COLS=1 ORBX_PYR_COLS_VARIANT=2" "tools/fuzz_parity.py 120 607"
run "ORBX_PYR_COLS=1 ORBX_PYR_COL_PX=40 ORBX_PYR_COLS_VARIANT=6" "tools/fuzz_parity.py 120 608"
run "ORBX_PATCH_BLUR=1" "tools/fuzz_parity.py 150 609"
run "ORBX_PYR_COLS=1 ORBX_BLUR_IN_COLS=1" "tools/fuzz_parity.py 120 610"
run "" "tools/fuzz_batches.py 120 611"
run "ORBX_SPLIT_MIN_MPX=0 ORBX_SPLIT=3" "tools/fuzz_batches.py 80 612"
run "ORBX_PATCH_BLUR=1 ORBX_SPLIT_MIN_MPX=0" "tools/fuzz_batches.py 80 613"
run "ORBX_SPLIT_MIN_MPX=0" "tools/fuzz_batches.py 80 614"
cat $OUT
