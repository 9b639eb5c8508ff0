#!/bin/bash
# Round 6 soak on the final sources (one part per gpurun call, each bounded): the default policy, the blur split forced at every level, the
# scratch-free queued quad-tree, frames beyond 4096 px, test aids (poisoned arenas, polluted LDS), batches under every overlap policy.
# usage (GPU box): bash tools/soak_r6.sh <part 1..4>
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
PART=${1:-1}
OFF=${2:-0}      # added to every seed: a second pass draws other cases
LOG=gpurun_out/soak_r6_part${PART}_off$OFF.log
: > $LOG
run() { echo "== [$1] $2" | tee -a $LOG; env $1 timeout -k 10 ${3:-300} python $2 2>&1 | tail -1 | tee -a $LOG; }
if [ "$PART" = 1 ]; then
  run "" "tools/fuzz_parity.py 500 $((1101 + OFF))" 400
  run "ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=3" "tools/fuzz_parity.py 300 $((1102 + OFF))" 300
  run "ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=1 FUZZ_AIDS=lds_pollute=119" "tools/fuzz_parity.py 200 $((1103 + OFF))" 250
elif [ "$PART" = 2 ]; then
  run "ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=6 FUZZ_AIDS=poison=201" "tools/fuzz_parity.py 200 $((1104 + OFF))" 250
  run "ORBX_OCT_ROOMY=1 ORBX_OCT_THREADS=256" "tools/fuzz_parity.py 300 $((1105 + OFF))" 300
  run "FUZZ_AIDS=pyr_cols_shape=6,poison=90" "tools/fuzz_parity.py 200 $((1106 + OFF))" 250
  run "FUZZ_AIDS=pyr_cols_shape=4,lds_pollute=33" "tools/fuzz_parity.py 200 $((1107 + OFF))" 250
elif [ "$PART" = 3 ]; then
  run "FUZZ_BIG=1" "tools/fuzz_parity.py 60 $((1108 + OFF))" 500
  run "FUZZ_BIG=1 FUZZ_AIDS=poison=165,lds_pollute=77" "tools/fuzz_parity.py 40 $((1109 + OFF))" 400
elif [ "$PART" = 4 ]; then
  run "" "tools/fuzz_batches.py 120 $((1110 + OFF))" 400
  run "ORBX_SPLIT_MIN_MPX=0 ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=3" "tools/fuzz_batches.py 80 $((1111 + OFF))" 300
  run "ORBX_SPLIT_MIN_MPX=0 ORBX_SPLIT=3 ORBX_PATCH_BLUR=1 ORBX_BLUR_SPLIT=2 FUZZ_AIDS=lds_pollute=201" "tools/fuzz_batches.py 60 $((1112 + OFF))" 250
  run "ORBX_SPLIT_MIN_MPX=0 FUZZ_AIDS=shared_upload_bytes=0" "tools/fuzz_batches.py 60 $((1113 + OFF))" 250
fi
