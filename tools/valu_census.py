#!/usr/bin/env python3
"""Static census of the vector instructions of the product kernels by issue class (profiles/r02_valu_issue_rate.md).

gfx950 issues v_fma_f32 / v_add_f32 / v_mul_f32 / v_add_u32 / v_sub_u32 / v_and_b32 / v_or_b32 / v_mov_b32 in ~2 cycles per wave64
instruction and everything else measured (packed-16, 3-input min/max, v_perm, v_alignbyte, dot products, 24-bit multiplies, shifts,
conversions, f64) in ~4.  SQ_INSTS_VALU x 4 cycles therefore over-states a kernel's issue time by half of its full-rate share; this
tool compiles a kernel source to ISA (same flags as the Makefile) and reports that share per kernel, whole body, every instruction
counted once (the hot loops of these kernels are straight-line bodies that dominate the static count as they dominate the dynamic one).

usage: valu_census.py [--json]      -> table (or JSON {kernel: {valu, full_rate, share}}) for the kernels of the hot path"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "extractorb_amd", "csrc")
FULL = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32"}      # (v_xor_b32: profiles/r04_valu_issue_rate_additions.md)
# kernel (as bench.py / rocprof name it) -> (source file, mangled-name fragment of the variant the default workload runs)
KERNELS = {"k_fast": ("k_fast.hip", "k_fastILi48ELi45ELb0ELb0E"), "k_blur": ("k_blur.hip", "k_blur"), "k_describe<plain>": ("k_describe.hip", "k_describeILb0E"),
           "k_describe<PB>": ("k_describe.hip", "k_describeILb1E"), "k_octree_256r": ("k_octree.hip", "k_octree_256rE"),
           "k_pyr_first": ("k_pyramid.hip", "k_pyr_firstILb1E"), "k_resize": ("k_pyramid.hip", "k_resizeILb1E"),
           "k_pyr_cols": ("k_pyramid.hip", "k_pyr_colsILb1ELi512ELi256E"),
           "k_octree_256": ("k_octree.hip", "k_octree_256E")}


def census(src, frag):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, "-x", "hip", os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r"^(_ZN4orbx\d+%s[^:\n]*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(frag), text, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found in %s" % (frag, src))
    valu = full = 0
    for line in m.group(2).splitlines():
        op = line.strip().split(" ")[0].split("\t")[0]
        if not op.startswith("v_"):
            continue
        valu += 1
        base = re.sub(r"_(e32|e64)$", "", op)
        if base in FULL:
            full += 1
    return dict(valu=valu, full_rate=full, share=round(full / max(valu, 1), 4))


if __name__ == "__main__":
    res = {k: census(*v) for k, v in KERNELS.items()}
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1, sort_keys=True))
    else:
        print("| kernel | VALU instructions (static) | full-rate class (2 cycles) | share | issue time vs SQ_INSTS_VALU x 4 cycles |\n|---|---|---|---|---|")
        for k, r in res.items():
            print("| %s | %d | %d | %.3f | x %.3f |" % (k, r["valu"], r["full_rate"], r["share"], 1 - r["share"] / 2))
