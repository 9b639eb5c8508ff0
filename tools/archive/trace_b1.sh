#!/bin/bash
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}; cd /tmp; export TMPDIR=/tmp ORBX_SPLIT=0
rm -rf $R/gpurun_out/trace_b1; mkdir -p $R/gpurun_out/trace_b1
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_b1 --output-format csv -- python3 $R/bench.py --steps 50 --warmup 5 --batch 1 --no-cpu-baseline --no-extras > $R/gpurun_out/trace_b1/log.txt 2>&1
cd $R; f=$(find gpurun_out/trace_b1 -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take a window in the middle of the timed region: find sequences starting with k_pyr_first
names = [r["Kernel_Name"].split("(")[0].replace("orbx::","").replace("void ","").split("<")[0] for r in rows]
starts = [i + 1 for i, n in enumerate(names[:-1]) if n.startswith("k_describe")]      # a call's first kernel follows the previous call's last
i0 = starts[len(starts)//2]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for i in range(i0, starts[len(starts)//2 + 1] + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("%-16s start %7.2f us  dur %6.2f us  gap %6.2f us  grid %s wg %s" % (names[i][:16], (s - t0)/1e3, (e - s)/1e3, 0 if prev_end is None else (s - prev_end)/1e3, rows[i].get("Grid_Size_X","")+"x"+rows[i].get("Grid_Size_Y","")+"x"+rows[i].get("Grid_Size_Z",""), rows[i].get("Workgroup_Size_X","")))
    prev_end = e
PY
rm -rf gpurun_out/trace_b1/*/ 2>/dev/null
