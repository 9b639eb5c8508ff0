#!/bin/bash
# kernel-level timeline of one step of a "next row" workload at a small batch.  usage: tools/trace_next.sh <workload> [batch]
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}; cd /tmp; export TMPDIR=/tmp ORBX_SPLIT=0
W=$1; B=${2:-2}
rm -rf $R/gpurun_out/trace_next; mkdir -p $R/gpurun_out/trace_next
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_next --output-format csv -- python3 $R/bench.py --steps 50 --warmup 5 --batch $B --workload $W --no-cpu-baseline --no-extras > $R/gpurun_out/trace_next/log.txt 2>&1
cd $R; f=$(find gpurun_out/trace_next -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("orbx::", "").replace("void ", "")
    if float(r["TotalDurationNs"]) > 50e3:
        print("%-40s calls %5s avg %9.2f us" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/trace_next/*/ 2>/dev/null
