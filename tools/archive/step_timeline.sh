#!/bin/bash
# GPU box: kernel start / end times of ONE overlapped step of the default bench (rocprofv3 --kernel-trace), relative to the step's first kernel.
# usage: tools/step_timeline.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT) or export it}
rm -rf $R/gpurun_out/step_tl
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/step_tl --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --prime-steps 8 --no-cpu-baseline --no-extras --no-verify "$@" > $R/gpurun_out/step_tl.log 2>&1
python3 - "$(find $R/gpurun_out/step_tl -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("void orbx::k_", "orbx::k_")) and "SelfTest" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    n = r["Kernel_Name"].replace("void ", "").replace("orbx::", "").split("(")[0]; return n.split("<")[0] + ("<" + n.split("<")[1][:5] if n.startswith("k_describe") else "")
# steps start with the pyramid kernel; take the one before the last (the last may be the profiled serial pass)
starts = [i for i, r in enumerate(rows) if name(r) in ("k_pyr_cols", "k_pyr_first")]
for which in (12, 13):
    i0, i1 = starts[which], starts[which + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    print("step starting at row %d:" % i0)
    for r in rows[i0:i1]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print("  %-16s queue %-3s start %8.1f us  end %8.1f us  (%7.1f)  grid %s" % (name(r), r.get("Queue_Id", "?"), s, e, e - s, r.get("Grid_Size_X", "?")))
PY
