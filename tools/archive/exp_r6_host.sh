#!/bin/bash
# Round 6 experiments (one gpurun call): (1) the queued quad-tree in its scratch-free build (ORBX_OCT_ROOMY=1) against the default;
# (2) the host-fed path (64 frames per call, two handles) under runtime settings that change how streams map to hardware queues / copy engines,
# and a trace of it (kernels + memory copies with timestamps) to see what overlaps.
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
bash tools/ab_env.sh "--steps 40 --warmup 5" "" "ORBX_OCT_ROOMY=1"
bash tools/ab_env.sh "--steps 20 --warmup 3 --workload hd1080" "" "ORBX_OCT_ROOMY=1"
bash tools/ab_env.sh "--steps 40 --warmup 5 --workload stereo640" "" "ORBX_OCT_ROOMY=1"
for cfg in "" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "HSA_ENABLE_SDMA=0" "ORBX_ZERO_COPY=0"; do
  for rep in 1 2; do echo "[$cfg]"; env $cfg HOST_RATE_BATCHES=64 timeout -k 10 120 python tools/host_path_rate.py 2>/dev/null; done
done
} 2>&1 | tee gpurun_out/exp_r6_host.log
cd /tmp && export TMPDIR=/tmp
HOST_RATE_BATCHES=64 timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/host_trace --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/host_path_rate.py > $GRAFT_REPO_ROOT/gpurun_out/host_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os
root = "gpurun_out/host_trace"
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-40:], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?")), r.get("Stream_Id", "?")))
ev.sort()
t0 = ev[0][0] if ev else 0
tail = ev[-260:]
with open("gpurun_out/host_trace_tail.txt", "w") as o:
    for s, e, n, q in tail:
        o.write("%10.1f %10.1f %8.1f us  q%s  %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
print("events", len(ev))
PY
rm -rf gpurun_out/host_trace
