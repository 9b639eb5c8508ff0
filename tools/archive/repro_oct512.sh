#!/bin/bash
# Rebuilds the one revision in which a quad-tree variant returned wrong lists (docs/history/DESIGN_rounds_1-5.md §4i (a)) and runs the two failing fuzz cases.
# Part 1 (build container): tools/repro_oct512.sh build        -> build/repro/{tree, lib_fail.so, lib_*.so}
# Part 2 (GPU box, one gpurun call): tools/repro_oct512.sh run -> per library: the levels that differ from the oracle, three runs each
# What it showed in round 3: lib_fail fails cases 1 and 4 of fuzz seed 103 with ORBX_OCT_THREADS=512 (64 VGPRs), differently from run to run;
# -mllvm -amdgpu-waitcnt-forcezero does not cure it; -mllvm -enable-misched=false, -O1 and -mllvm -amdgpu-prealloc-sgpr-spill-vgprs do.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build/repro
if [ "$1" = build ]; then
  rm -rf $D; mkdir -p $D/tree
  git -C $R archive e6617bb | tar -x -C $D/tree
  sed -i 's/^#define OCT_SHORT_PHASE2 (OCT_W <= 4)/#define OCT_SHORT_PHASE2 1/' $D/tree/extractorb_amd/csrc/k_octree.hip     # the draft of 08:13: short phase-2 pass in every variant
  cd $D/tree/extractorb_amd/csrc
  F="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I../../include"
  for s in $(grep "^SRC" Makefile | sed 's/SRC = //'); do [ $s = k_octree.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c -x hip $s -o $D/${s%.*}.o 2>/dev/null & done; wait
  mk() { n=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 $F "$@" -c -x hip k_octree.hip -o $D/oct_$n.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$n.so $D/k_*.o $D/orbx_api.o $D/oct_$n.o 2>/dev/null && echo built lib_$n.so; }
  mk fail & mk forcezero -mllvm -amdgpu-waitcnt-forcezero & mk nomisched -mllvm -enable-misched=false & mk O1 -O1 & mk prealloc -mllvm -amdgpu-prealloc-sgpr-spill-vgprs & wait
  make -C $D/tree/oracle >/dev/null
  cat > $D/tree/tools/repro_case.py <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, fuzz_parity, extractorb_amd as X
from test_gpu_parity import oracle_run
rng = np.random.default_rng(103)
cases = [fuzz_parity.draw_case(rng, t) for t in range(5)]
os.environ["ORBX_OCT_THREADS"] = "512"
out = []
for t in (1, 4):
    c = cases[t]
    ex = X.ORBextractor(c["nf"], c["sf"], c["nlevels"], c["ini"], c["mn"], max_width=c["cols"], max_height=c["rows"])
    o, want = oracle_run(c["img"], c["nf"], c["lap"], c["nlevels"], c["sf"], c["ini"], c["mn"])
    for rep in range(3):
        mono, k, d, lvl = ex(c["img"], None, c["lap"])
        out.append("case %d run %d: differing levels %s, keypoints per level %s" % (t, rep, [l for l in range(c["nlevels"]) if lvl[l].tobytes() != o.level_keypoints(l).tobytes()], [len(x) for x in lvl]))
print(os.path.basename(os.environ["ORBX_LIBRARY"]), " | ".join(out))
PY
  echo "now: gpurun -- 'bash tools/repro_oct512.sh run'"
elif [ "$1" = run ]; then
  cd $D/tree
  for l in fail forcezero nomisched O1 prealloc; do ORBX_LIBRARY=$D/lib_$l.so timeout -k 10 120 python tools/repro_case.py 2>&1 | tail -1; done
else
  echo "usage: $0 build|run"; exit 2
fi
