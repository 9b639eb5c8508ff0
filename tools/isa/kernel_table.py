#!/usr/bin/env python3
"""Register / scratch / memory-instruction table of every kernel in the SHIPPED extractorb_amd/liborbx.so, read from the code objects embedded in
the library itself (clang offload bundles in .hip_fatbin; llvm-readelf --notes for the metadata, llvm-objdump -d for the instructions).

Why (docs/history/DESIGN_rounds_1-5.md §4i): the one wrong-result fault this project has seen lived in a 64-VGPR quad-tree variant whose node arrays
were reached with FLAT instructions through generic pointers reloaded from scratch, beside 147 SGPR spills; the source no longer contains that
construct, and this table is what stops it coming back unnoticed: tests/test_kernel_table.py rebuilds it from the library under test and fails if
  * a kernel whose arrays are LDS or global by construction contains a flat_load / flat_store / flat_atomic,
  * a kernel's scratch bytes or spill counts exceed what extractorb_amd/csrc/kernel_table.json (checked in next to the kernels) allows,
  * a kernel appears or disappears without the table saying so.
A codegen shift therefore shows up as a diff of that file.  usage: kernel_table.py [--write] [--library path]"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
TABLE = os.path.join(ROOT, "extractorb_amd", "csrc", "kernel_table.json")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """the gfx950 ELF images embedded in the library, in file order"""
    data = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        num = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(num):
            off, size, tsz = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tsz].decode()
            p += tsz
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(MAGIC)


def short(name):
    """_ZN4orbx10k_pyr_colsILb1ELi512ELi256ELb0ELi512EEEv... -> k_pyr_cols<1,512,256,0,512>"""
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dem = name
    dem = re.sub(r"^void ", "", dem)
    dem = re.sub(r"\(.*$", "", dem)            # drop the argument list
    dem = dem.replace("orbx::", "").replace("(anonymous namespace)::", "")
    dem = re.sub(r"\b(true|false)\b", lambda m: "1" if m.group(1) == "true" else "0", dem).replace(" ", "")
    return dem or name


def table(lib):
    rows = {}
    with tempfile.TemporaryDirectory() as td:
        for n, elf in enumerate(code_objects(lib)):
            path = os.path.join(td, "co_%d.elf" % n)
            open(path, "wb").write(elf)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True).stdout
            meta = {}
            for blk in re.split(r"\n  - ", notes)[1:]:
                m = re.search(r"\.name:\s+(\S+)", blk)
                if not m or ".symbol:" not in blk:
                    continue
                g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else 0
                meta[m.group(1)] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), sgpr_spill=g("sgpr_spill_count"),
                                        vgpr_spill=g("vgpr_spill_count"), scratch_bytes=g("private_segment_fixed_size"), lds_static_bytes=g("group_segment_fixed_size"))
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1) if m.group(1) in meta else None
                    if cur:
                        meta[cur].update(flat=0, scratch_ops=0, instructions=0)
                    continue
                if not cur:
                    continue
                op = line.strip().split(" ")[0].split("\t")[0]
                if not op or op.startswith("//"):
                    continue
                meta[cur]["instructions"] += 1
                if re.match(r"flat_(load|store|atomic)", op):
                    meta[cur]["flat"] += 1
                if op.startswith("scratch_") or (op.startswith("buffer_") and "offen" in line and "s[0:3]" in line):
                    meta[cur]["scratch_ops"] += 1
            for name, v in meta.items():
                rows[short(name)] = v
    return dict(sorted(rows.items()))


# Kernels that address one array through a generic pointer BY DESIGN, with no scratch at all beside it (they stay in the table; only the no-FLAT
# rule is waived): k_bow_reduce keeps a frame's word sums in LDS when they fit and in an HBM scratch row otherwise (k_bow.hip: `wsum`), k_search_bow
# reads descriptors from LDS when they were staged and from L2 otherwise (k_bow_match.hip).  Everything on the extraction path proper
# addresses LDS and global memory through pointers whose address space the compiler knows.
FLAT_ALLOWED = {"k_bow_reduce", "k_search_bow<0>", "k_search_bow<1>"}


def check(current, allowed):
    """violations of `current` (this build) against `allowed` (the checked-in table)"""
    bad = []
    for k, v in current.items():
        if v.get("flat", 0) and (k not in FLAT_ALLOWED or v["scratch_bytes"]):
            bad.append("%s: %d FLAT memory instructions (generic pointers: the construct behind docs/history/DESIGN_rounds_1-5.md §4i's fault)" % (k, v["flat"]))
        a = allowed.get(k)
        if a is None:
            bad.append("%s: kernel not in kernel_table.json (regenerate it: python tools/isa/kernel_table.py --write, and look at the diff)" % k)
            continue
        for key in ("scratch_bytes", "sgpr_spill", "vgpr_spill"):
            if v[key] > a[key]:
                bad.append("%s: %s %d > %d in kernel_table.json" % (k, key, v[key], a[key]))
    for k in allowed:
        if k not in current:
            bad.append("%s: in kernel_table.json but not in the library" % k)
    return bad


if __name__ == "__main__":
    lib = sys.argv[sys.argv.index("--library") + 1] if "--library" in sys.argv else os.path.join(ROOT, "extractorb_amd", "liborbx.so")
    t = table(lib)
    if "--write" in sys.argv:
        json.dump(t, open(TABLE, "w"), indent=1, sort_keys=True)
        print("wrote %s: %d kernels" % (TABLE, len(t)))
    else:
        print("| kernel | VGPR | SGPR | scratch B | SGPR spills | VGPR spills | FLAT | instructions |\n|---|---|---|---|---|---|---|---|")
        for k, v in t.items():
            print("| %s | %d | %d | %d | %d | %d | %d | %d |" % (k, v["vgpr"], v["sgpr"], v["scratch_bytes"], v["sgpr_spill"], v["vgpr_spill"], v.get("flat", 0), v.get("instructions", 0)))
        if os.path.exists(TABLE):
            bad = check(t, json.load(open(TABLE)))
            print("\n".join(bad) if bad else "within kernel_table.json")
            sys.exit(1 if bad else 0)
