#!/usr/bin/env python3
"""valu_rate csv (tools/ubench/valu_rate.hip, run on the GPU box) -> markdown table for profiles/.
usage: valu_rate_table.py gpurun_out/r03_valu_rate.csv > profiles/r03_valu_issue_rate.md"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ops = []
for r in rows:
    if r["instruction"] not in ops:
        ops.append(r["instruction"])
cell = {(r["instruction"], int(r["waves_per_simd"])): r for r in rows}
f = lambda r, k: float(r[k])
print("""# Vector-instruction issue rate on MI355X (gfx950) — `tools/ubench/valu_rate.hip`, round 3: both clocks, the clock each cell ran at, residency

`hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate csv` on the GPU box (1x MI355X); raw numbers:
`profiles/r03_valu_issue_rate.csv`.  Every wave issues 128 000 instructions of one kind (inline asm, no memory traffic); 256 x W workgroups of 4
waves, W = nominal waves per SIMD.  Per cell (16 accumulator chains per wave; the csv also has 8 chains and write-only destinations):

* **ev** — SIMD cycles per wave64 instruction from HIP-event time, priced at the nominal 2.4 GHz;
* **GHz** — the clock the cell actually ran at: delta `s_memtime` / delta `s_memrealtime` x 100 MHz inside the same waves (MI355X_MICROARCH.md, DVFS (6));
* **res** — waves resident per SIMD on average while the kernel ran (sum of wave lifetimes / (first start .. last end) / 1024 SIMDs);
* **mt** — SIMD cycles per instruction from the waves' own `s_memtime` ticks, spread over the waves that were actually co-resident (`res`).

Round 2 published only `ev` and its csv carried an `mt` column computed with the NOMINAL W, which disagreed by up to 1.8x (4.21 vs 2.37 at W = 8).
The gap is neither a clock far below 2.4 GHz (the cells ran at 2.25-2.42 GHz) nor a different issue rate: a grid of 256 x W workgroups does not
run as one resident set — at W = 8 only ~4.5 waves share a SIMD on average, at W = 4 ~2.6 — so a wave's own ticks must be spread over `res`
waves, not W.  With that, `mt` and `ev x GHz / 2.4` agree to a few per cent in every cell.

| instruction | class | W=1: ev / mt / GHz / res | W=2 | W=4 | W=8 |
|---|---|---|---|---|---|""")
full, half = [], []
for op in ops:
    c8 = f(cell[(op, 8)], "acc16_ev") * f(cell[(op, 8)], "acc16_ghz") / 2.4
    cls = "full rate (~2.3 cycles)" if c8 < 3.0 else "half rate (~4.1 cycles)"
    (full if c8 < 3.0 else half).append(op)
    cols = []
    for w in (1, 2, 4, 8):
        r = cell[(op, w)]
        res = f(r, "acc16_resident")
        cols.append("%.2f / %.2f / %.2f / %.2f" % (f(r, "acc16_ev"), f(r, "acc16_mt") * w / res, f(r, "acc16_ghz"), res))
    print("| `%s` | %s | %s |" % (op, cls, " | ".join(cols)))
h8 = [f(cell[(op, 8)], "acc16_ev") * f(cell[(op, 8)], "acc16_ghz") / 2.4 for op in half]
print("""
Reading:

* **Two issue classes** (unchanged from round 2, now in true cycles): `%s` issue one wave64 instruction per ~2.2-2.4 cycles once a SIMD holds two or
  more waves (`v_fma_f32` included: round 2's table had it at 3.4-3.8 and labelled it "half rate" while the text said "2-cycle class"; measured
  again it is full rate, 2.2-2.3 cycles at 2.26-2.30 GHz).
* `%s` stay at **%.2f-%.2f true cycles at W = 8** however many waves share the SIMD and however many accumulators a wave uses.  These are the
  instructions the kernels of this repository are made of.
* **The ceiling for a kernel made of half-rate instructions** = 1024 SIMDs x clock / cycles per instruction.  k_fast's own clock in the benchmark's
  launch shape (512 x 640x480 frames per launch), from a `-DORBX_FAST_CLOCK` build that stamps one wave in 64 (`tools/fast_clock.py`): **2.375 GHz**
  (step time unchanged by the stamps: 2.06 ms).  At 4.15 cycles that is **586 G wave-instructions/s** (bench.py prices 4.0 cycles at 2.4 GHz = 614 G/s,
  an upper bound of the ceiling, so its `valu_issue.frac` is a lower bound of the occupancy).
""" % ("`, `".join(full), "`, `".join(half), min(h8), max(h8)))
