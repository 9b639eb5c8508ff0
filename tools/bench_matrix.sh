#!/bin/bash
# Runs on the GPU box: bench.py over workloads, variants and batch sizes; JSON lines into gpurun_out/matrix.jsonl
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; : > gpurun_out/matrix.jsonl
run() { echo "# $*" >> gpurun_out/matrix.jsonl; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl; }
timeout -k 10 400 python bench.py --steps 30 2>/dev/null | tail -1 > gpurun_out/matrix_default.json
run --steps 30
run --steps 30 --variant textured
run --steps 30 --variant sparse
run --steps 30 --variant natural
run --steps 30 --workload stereo640
run --steps 30 --workload stereo640_match
run --steps 30 --workload mono640_init
run --steps 30 --workload hd720
run --steps 30 --workload hd1080
run --steps 30 --workload hd1080 --variant natural
run --steps 300 --batch 1
run --steps 200 --batch 8
run --steps 100 --batch 64
run --steps 30 --batch 512
echo "# ORBX_SPLIT_BATCHES=1 --steps 30" >> gpurun_out/matrix.jsonl; ORBX_SPLIT_BATCHES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# --steps 30 --handles 2" >> gpurun_out/matrix.jsonl; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --handles 2 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
run --steps 30 --workload mono640_bgr
timeout -k 10 300 python tools/host_path_rate.py > gpurun_out/host_path.txt 2>&1
echo done
