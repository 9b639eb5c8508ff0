#!/bin/bash
cd "${GRAFT_REPO_ROOT:?run on the GPU box}"
mkdir -p gpurun_out
{
for cfg in "ORBX_D2H=1" "ORBX_D2H=2"; do
  echo "[$cfg]"; env $cfg HOST_RATE_BATCHES=16,32,64,96,128,192,256 timeout -k 10 200 python tools/host_path_rate.py 2>/dev/null
done
} 2>&1 | tee gpurun_out/exp_r6_d2h2.log
cd /tmp && export TMPDIR=/tmp
ORBX_D2H=2 HOST_RATE_BATCHES=256 timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/host_trace --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/host_path_rate.py > $GRAFT_REPO_ROOT/gpurun_out/host_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
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
with open("gpurun_out/host_trace_tail_d2h2_b256.txt", "w") as o:
    for s, e, n, q in ev[-120:]:
        o.write("%10.1f %10.1f %8.1f us  q%s  %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
PY
rm -rf gpurun_out/host_trace
