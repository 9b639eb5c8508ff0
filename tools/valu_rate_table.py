#!/usr/bin/env python3
"""valu_rate csv (tools/ubench/valu_rate.hip, run on the GPU box) -> markdown table for profiles/.
usage: valu_rate_table.py gpurun_out/r02_valu_rate.csv > profiles/r02_valu_issue_rate.md"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ops = []
for r in rows:
    if r["instruction"] not in ops:
        ops.append(r["instruction"])
cell = {(r["instruction"], int(r["waves_per_simd"])): r for r in rows}
print("""# Vector-instruction issue rate on MI355X (gfx950), with f32 calibration lines — `tools/ubench/valu_rate.hip`

`hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate csv` on the GPU box (1x MI355X).
Every wave issues 128 000 instructions of one kind (inline asm, no memory traffic); 256 workgroups x waves/SIMD, 4 waves each.
Entries: SIMD cycles per wave64 instruction, from HIP-event time at the nominal 2.4 GHz.  `acc8` / `acc16`: 8 / 16 accumulator
chains per wave (each instruction reads its own previous result); `indep`: write-only destinations (no read-after-write at all).

| instruction | class | 1 wave/SIMD acc8 / acc16 / indep | 2 waves/SIMD | 4 waves/SIMD | 8 waves/SIMD |
|---|---|---|---|---|---|""")
full, half = [], []
for op in ops:
    c8 = min(float(cell[(op, 8)][k]) for k in ("acc8_ev", "acc16_ev", "indep_ev"))
    cls = "full rate (2 cycles)" if c8 < 3.0 else "half rate (4 cycles)"
    (full if c8 < 3.0 else half).append(op)
    cols = []
    for w in (1, 2, 4, 8):
        r = cell[(op, w)]
        cols.append("%.2f / %.2f / %.2f" % (float(r["acc8_ev"]), float(r["acc16_ev"]), float(r["indep_ev"])))
    print("| `%s` | %s | %s |" % (op, cls, " | ".join(cols)))
print("""
Reading:

* **Two issue classes exist on gfx950, and the guide's 2-cycle figure is the f32 one.**  `%s` issue one wave64 instruction per
  ~2.3 cycles once a SIMD holds two or more waves (4.4-5.2 for a wave alone) - this reproduces `MI355X_MICROARCH.md:54,473`
  (`v_fma_f32` 2 cycles, 4 for a lone wave).
* `%s` stay at **4.1-4.4 cycles however many waves share the SIMD and however many accumulators a wave uses** (acc16 and the
  write-only form are no faster than acc8, so the figure is not a dependent-chain artefact).  These are the instructions the kernels of this
  repository are made of (`v_pk_minimum3_f16` / `v_pk_maximum3_f16`, `v_perm_b32`, `v_alignbyte_b32`, `v_dot4_u32_u8`, `v_mad_u32_u24`, `v_bfe_u32`,
  packed-16 arithmetic).  Note that the packed / 3-input forms lose nothing against the full-rate class per unit of work: `v_pk_fma_f32` (two FMAs,
  4 cycles) equals two `v_fma_f32`; `v_pk_minimum3_f16` does four 2-input minima in 4 cycles where the full-rate class has no min/max at all
  (`v_min_u32` and `v_max3_f32` are themselves half rate).
* Pricing rule used by bench.py and the SQ-counter tables: **issue time = SQ_ACTIVE_INST_VALU x 4 cycles** (the counter counts quad-cycles of
  VALU issue, so a 2-cycle instruction contributes half a count and the mix is priced by the hardware itself), against 1024 SIMDs x 2.4 GHz.
  For a kernel made of half-rate instructions this equals `SQ_INSTS_VALU` / (614 G wave-instr/s), the round-1 figure.
""" % ("`, `".join(full), "`, `".join(half)))
