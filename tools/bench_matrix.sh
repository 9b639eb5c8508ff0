#!/bin/bash
# Runs on the GPU box: bench.py over workloads, variants and batch sizes; JSON lines into gpurun_out/matrix.jsonl
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; : > gpurun_out/matrix.jsonl
run() { echo "# $*" >> gpurun_out/matrix.jsonl; timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl; }
timeout -k 10 400 python bench.py --steps 30 2>/dev/null | tail -1 > gpurun_out/matrix_default.json
run --steps 30
run --steps 30 --variant textured
run --steps 30 --variant sparse
run --steps 30 --variant natural
run --steps 30 --workload stereo640
run --steps 30 --workload stereo640_match
run --steps 30 --workload mono640_init
run --steps 30 --workload mono640_track
run --steps 30 --workload mono640_bow
run --steps 30 --workload mono640_refkf
run --steps 30 --workload mono640_bgr
run --steps 30 --workload hd720
run --steps 30 --workload hd1080
run --steps 30 --workload hd1080 --variant natural
run --steps 300 --batch 1
run --steps 300 --batch 2 --workload stereo640
run --steps 200 --batch 2 --workload stereo640_match
run --steps 200 --batch 2 --workload mono640_track
run --steps 200 --batch 2 --workload mono640_refkf
run --steps 200 --batch 2 --workload mono640_init
run --steps 200 --batch 8
run --steps 100 --batch 64
run --steps 60 --batch 128
run --steps 50 --batch 256
run --steps 15 --batch 1024
run --steps 10 --batch 2048
echo "# ORBX_SPLIT=0 --steps 30" >> gpurun_out/matrix.jsonl; ORBX_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_FUSE_SMALL=0 --steps 300 --batch 1" >> gpurun_out/matrix.jsonl; ORBX_FUSE_SMALL=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 300 --batch 1 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_PATCH_BLUR=0 --steps 30 --workload hd1080" >> gpurun_out/matrix.jsonl; ORBX_PATCH_BLUR=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 --workload hd1080 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_PATCH_BLUR=0 --steps 30 --workload hd720" >> gpurun_out/matrix.jsonl; ORBX_PATCH_BLUR=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 --workload hd720 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_SPLIT=3 --steps 30 --batch 256" >> gpurun_out/matrix.jsonl; ORBX_SPLIT=3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 --batch 256 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_PATCH_BLUR=0 --steps 30" >> gpurun_out/matrix.jsonl; ORBX_PATCH_BLUR=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_BLUR_SPLIT=0 --steps 30" >> gpurun_out/matrix.jsonl; ORBX_BLUR_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
echo "# ORBX_BLUR_SPLIT=4 --steps 30" >> gpurun_out/matrix.jsonl; ORBX_BLUR_SPLIT=4 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 >> gpurun_out/matrix.jsonl
run --steps 30 --handles 2
HOST_RATE_BATCHES=1,8,64,256,512 timeout -k 10 300 python tools/host_path_rate.py > gpurun_out/host_path.txt 2>&1
echo done
