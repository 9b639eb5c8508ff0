#!/usr/bin/env python3
"""Forward dataflow over a gfx9 kernel's ISA text (hipcc -S): which SGPR-spill slots (VGPR, lane) written by v_writelane_b32 can be
READ by v_readlane_b32 on some control-flow path before any write reached them ("maybe-uninitialised lane read").

The compiler spills scalar registers (lane masks of divergent control flow, uniform values) into lanes of reserved VGPRs.  A reload on
a path that bypassed the spill — e.g. the spill sits in a block a wave skips through `s_cbranch_execz` while the reload sits after the
join — returns whatever the lane held: a result that depends on which waves happened to be fully inactive, i.e. on the data.

usage: spill_lane_dataflow.py file.s [kernel-name-substring]"""
import re
import sys


def blocks_of(lines):
    """Basic blocks: (label, [instructions]); edges from s_branch / s_cbranch_* / fallthrough."""
    blocks, cur, name = [], [], "entry"
    for l in lines:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur)); name, cur = m.group(1), []
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t)
        if re.match(r"s_(c?branch|endpgm|setpc)", t):
            blocks.append((name, cur)); name, cur = "%s+%d" % (name, len(blocks)), []
    blocks.append((name, cur))
    return [b for b in blocks if b[1] or b[0].startswith(".LBB")]


def analyse(lines):
    bl = blocks_of(lines)
    idx = {n: i for i, (n, _) in enumerate(bl)}
    succ = [[] for _ in bl]
    for i, (n, ins) in enumerate(bl):
        last = ins[-1] if ins else ""
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", last)
        if last.startswith("s_endpgm"):
            continue
        if m and m.group(1) in idx:
            succ[i].append(idx[m.group(1)])
        if not last.startswith("s_branch") and i + 1 < len(bl):
            succ[i].append(i + 1)
    pred = [[] for _ in bl]
    for i, ss in enumerate(succ):
        for s in ss:
            pred[s].append(i)
    slots = set()
    gen = []
    for n, ins in bl:
        g = set()
        for t in ins:
            m = re.match(r"v_writelane_b32 (v\d+), \S+ (\d+)$", t)
            if m:
                g.add((m.group(1), int(m.group(2)))); slots.add((m.group(1), int(m.group(2))))
        gen.append(g)
    # must-be-written sets: IN[b] = intersection of OUT[p]; OUT[b] = IN[b] | gen[b]; entry IN = {}
    IN = [set(slots) for _ in bl]
    IN[0] = set()
    changed = True
    while changed:
        changed = False
        for i in range(len(bl)):
            if i == 0:
                new = set()
            else:
                ps = [IN[p] | gen[p] for p in pred[i]]
                new = set.intersection(*ps) if ps else set(slots)
            if new != IN[i]:
                IN[i] = new; changed = True
    findings = []
    for i, (n, ins) in enumerate(bl):
        have = set(IN[i])
        for t in ins:
            m = re.match(r"v_writelane_b32 (v\d+), \S+ (\d+)$", t)
            if m:
                have.add((m.group(1), int(m.group(2))))
            m = re.match(r"v_readlane_b32 (s\d+), (v\d+), (\d+)$", t)
            if m and (m.group(2), int(m.group(3))) in slots and (m.group(2), int(m.group(3))) not in have:
                findings.append((n, t))
    return bl, findings, slots


def main():
    text = open(sys.argv[1]).read().splitlines()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    # split the file into kernels
    start = None
    for i, l in enumerate(text):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            start, name = i, m.group(1)
        if ".end_amdhsa_kernel" in l and start is not None:
            if want in name:
                bl, f, slots = analyse(text[start:i])
                print("%s: %d blocks, %d spill slots, %d maybe-uninitialised lane reads" % (name[:40], len(bl), len(slots), len(f)))
                for n, t in f[:40]:
                    print("    %-14s %s" % (n, t))
            start = None


if __name__ == "__main__":
    main()
