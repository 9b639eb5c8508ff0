#!/bin/bash
# Launch boundaries of a single-frame call at several frame sizes (rocprofv3 --kernel-trace): the gap between a kernel's end and the next one's
# start against the bytes the first one wrote.  usage (GPU box): bash tools/trace_gaps.sh
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}; cd /tmp; export TMPDIR=/tmp ORBX_SPLIT=0
for sz in "240 320" "480 640" "720 1280" "1080 1920"; do set -- $sz
  rm -rf $R/gpurun_out/trace_gap; mkdir -p $R/gpurun_out/trace_gap
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_gap --output-format csv -- python3 $R/tools/gap_probe.py $1 $2 60 > $R/gpurun_out/trace_gap/log.txt 2>&1
  f=$(find $R/gpurun_out/trace_gap -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $1 $2 <<'PY'
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("orbx::", "").replace("void ", "").split("<")[0] for r in rows]
gaps, durs = {}, {}
for i in range(len(rows) // 2, len(rows) - 1):
    a, b = names[i], names[i + 1]
    g = (int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3
    gaps.setdefault((a, b), []).append(g)
    durs.setdefault(a, []).append((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3)
print("%sx%s:" % (sys.argv[3], sys.argv[2]), "  ".join("%s %.1f us -> [gap %.2f] -> %s" % (a, statistics.median(durs[a]), statistics.median(v), b) for (a, b), v in gaps.items() if len(v) > 5))
PY
done
rm -rf $R/gpurun_out/trace_gap
