// k_fast_body.hpp - the FAST cell's arithmetic and the one-cell-per-wave body as device functions: k_fast.hip launches it as a kernel of its
// own (and has the workgroup-per-cell form of single frames).  See k_fast.hip for the
// algorithm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"
#include "k_blur_body.hpp"

namespace orbx {


// Two pixels per register: u16 halves holding 0..255.  As IEEE binary16 bit patterns those are non-negative
// denormals, whose order is the integer order, and minimum/maximum return one operand unchanged (the kernel runs
// with FP16 denormals preserved, the HIP default), so the packed 3-input f16 ops ARE integer min3/max3 on both halves.
__device__ __forceinline__ unsigned pkmin3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pkmax3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pksub(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// bytes O, O+1 of the 12-byte string L|C|R (three consecutive dwords of a tile row), zero-extended to two u16
template <int O>
__device__ __forceinline__ unsigned pairAt(unsigned L, unsigned C, unsigned R) {
    static_assert(O >= 0 && O <= 10, "pair outside the three dwords");
    if constexpr (O <= 6) return __builtin_amdgcn_perm(C, L, 0x0C000C00u | O | ((O + 1) << 16));
    else return __builtin_amdgcn_perm(R, C, 0x0C000C00u | (O - 4) | ((O - 3) << 16));
}

// Ring of the two pixels at bytes P, P+1 of the centre dword (P = 0 or 2) in cv::FAST order, r[16] = the centre pair:
// L/C/R[d] = the three dwords of tile row (centre row + d - 3).
template <int P>
__device__ __forceinline__ void pairRing(const unsigned (&L)[7], const unsigned (&C)[7], const unsigned (&R)[7], unsigned (&r)[17]) {
    constexpr int B = 4 + P;      // byte of the first pixel inside L|C|R
    r[0] = pairAt<B>(L[6], C[6], R[6]);       r[1] = pairAt<B + 1>(L[6], C[6], R[6]);   r[2] = pairAt<B + 2>(L[5], C[5], R[5]);
    r[3] = pairAt<B + 3>(L[4], C[4], R[4]);   r[4] = pairAt<B + 3>(L[3], C[3], R[3]);   r[5] = pairAt<B + 3>(L[2], C[2], R[2]);
    r[6] = pairAt<B + 2>(L[1], C[1], R[1]);   r[7] = pairAt<B + 1>(L[0], C[0], R[0]);   r[8] = pairAt<B>(L[0], C[0], R[0]);
    r[9] = pairAt<B - 1>(L[0], C[0], R[0]);   r[10] = pairAt<B - 2>(L[1], C[1], R[1]);  r[11] = pairAt<B - 3>(L[2], C[2], R[2]);
    r[12] = pairAt<B - 3>(L[3], C[3], R[3]);  r[13] = pairAt<B - 3>(L[4], C[4], R[4]);  r[14] = pairAt<B - 2>(L[5], C[5], R[5]);
    r[15] = pairAt<B - 1>(L[6], C[6], R[6]);
    r[16] = pairAt<B>(L[3], C[3], R[3]);
}
// BRIGHT: max over the 16 arcs of the arc minimum (and the centre); else min over the arcs of the arc maximum (and the
// centre).  Arcs k and k+1 (k even) share the 8 ring pixels k+1..k+8, and
//     max(min(core, r_k), min(core, r_k+9)) = min(core, max(r_k, r_k+9)),
// so 8 "arc pairs" replace 16 arcs.  The cores are two 4-runs starting at odd positions, each 4-run two 2-runs:
// 8 + 8 + 8 (end points) + 8 (3-input) + 4 (reduction) = 36 instructions per polarity.
template <bool BRIGHT>
__device__ __forceinline__ unsigned arcExtreme(const unsigned (&r)[17]) {
    unsigned x2[8], x4[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {            // x2[i] = extreme of ring pixels 2i+1, 2i+2
        const int p = 2 * i + 1;
        x2[i] = BRIGHT ? pkmin3(r[p], r[(p + 1) & 15], r[(p + 1) & 15]) : pkmax3(r[p], r[(p + 1) & 15], r[(p + 1) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++)              // x4[i] = extreme of ring pixels 2i+1 .. 2i+4
        x4[i] = BRIGHT ? pkmin3(x2[i], x2[(i + 1) & 7], x2[(i + 1) & 7]) : pkmax3(x2[i], x2[(i + 1) & 7], x2[(i + 1) & 7]);
#pragma unroll
    for (int i = 0; i < 8; i++) {            // arcs 2i and 2i+1: core = pixels 2i+1 .. 2i+8, end points 2i and 2i+9
        const int k = 2 * i;
        const unsigned e = BRIGHT ? pkmax3(r[k], r[(k + 9) & 15], r[(k + 9) & 15]) : pkmin3(r[k], r[(k + 9) & 15], r[(k + 9) & 15]);
        g[i] = BRIGHT ? pkmin3(x4[i], x4[(i + 2) & 7], e) : pkmax3(x4[i], x4[(i + 2) & 7], e);
    }
    return BRIGHT ? pkmax3(pkmax3(g[0], g[1], g[2]), pkmax3(g[3], g[4], g[5]), pkmax3(g[6], g[7], r[16]))
                  : pkmin3(pkmin3(g[0], g[1], g[2]), pkmin3(g[3], g[4], g[5]), pkmin3(g[6], g[7], r[16]));
}
// One polarity per pixel (round 4).  A pixel is a corner in at most one polarity, and which one is decided exactly by the opposite ring pairs: a
// 9-arc holds one pixel of EVERY pair (k, k+8), so a dark corner has min(r_k, r_k+8) < v for all eight pairs; a bright corner's arc k..k+8
// holds BOTH pixels of pair k, so for it that test fails.  "max over the pairs of the pair minimum < v" therefore selects dark for every
// dark corner and bright for every bright one; where it picks the wrong side of a non-corner the score comes out as that side's S, which is
// <= the true S <= minThFAST - and the NMS pass counts everything <= minThFAST alike (thPair).  For a corner the other side's S is 0 (any
// other arc shares >= 2 pixels with the corner's arc), so the stored score IS S.  A dark pixel's ring and centre are complemented (x ^ 255
// reverses the order and keeps the halves inside 0..255), after which dark is bright: 12 + 2 + 17 (full-rate v_xor) + 36 + 1 instructions
// per pixel pair instead of 75.
__device__ __forceinline__ unsigned pairScore(const unsigned (&r)[17]) {
    unsigned m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = pkmin3(r[k], r[k + 8], r[k + 8]);
    const unsigned M = pkmax3(pkmax3(m[0], m[1], m[2]), pkmax3(m[3], m[4], m[5]), pkmax3(m[6], m[7], m[7]));
    unsigned flip;                                   // per half: M < v  <=>  M - v is negative  <=>  its high byte is 0xFF (|M - v| <= 255)
    asm("v_pk_lshrrev_b16 %0, %2, %1" : "=v"(flip) : "v"(pksub(M, r[16])), "s"(0x00080008u));
    unsigned x[17];
#pragma unroll
    for (int i = 0; i < 17; i++) x[i] = r[i] ^ flip;
    __builtin_amdgcn_sched_barrier(0);
    return pksub(arcExtreme<true>(x), x[16]);        // the centre took part in the reduction: never negative
}

#if defined(ORBX_FAST_CLOCK) && defined(ORBX_FAST_TU)      // (diagnostic builds stamp the kernel of k_fast.hip only)
// diagnostic build only (tools/fast_clock.py): the clock k_fast's waves actually run at = sum of delta s_memtime / sum of delta s_memrealtime
// x 100 MHz over every cell-wave (MI355X_MICROARCH.md, DVFS give-back (6)); the stamps go to a buffer nothing else reads
constexpr int kFastClockSlots = 4096;      // one sampled wave per slot: plain stores (an atomic per wave would BE the load)
__device__ unsigned long long g_fastClock[2 * kFastClockSlots];
extern "C" int orbx_debug_fast_clock(unsigned long long* out, int reset) {
    if (reset) return (int)hipMemset((void*)nullptr, 0, 0) + (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fastClock), out, sizeof(unsigned long long) * 2 * kFastClockSlots);
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastClock), sizeof(unsigned long long) * 2 * kFastClockSlots);
}
// ... and the span of every wave of frame 0 (tools/fast_spans.py): [2 * (chunk * 4 + wave)] = start, end in s_memrealtime ticks
__device__ unsigned long long g_fastSpans[2 * 4096], g_fastMid[4 * 4096];      // (mid: staged, scored, counted, ... of a FAST wave)
extern "C" int orbx_debug_fast_spans(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastSpans), sizeof(g_fastSpans)); }
extern "C" int orbx_debug_fast_mid(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastMid), sizeof(g_fastMid)); }
#define FAST_MID(which) do { const int fsI = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6); \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.z == 0 && fsI < 4096) g_fastMid[4 * fsI + (which)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FAST_SPAN(which) do { const int fsI = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6); \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.z == 0 && fsI < 4096) g_fastSpans[2 * fsI + (which)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FAST_CLOCK_BEGIN const unsigned long long fcR0 = __builtin_amdgcn_s_memrealtime(), fcT0 = __builtin_amdgcn_s_memtime();
#define FAST_CLOCK_END do { const unsigned long long fcT1 = __builtin_amdgcn_s_memtime(), fcR1 = __builtin_amdgcn_s_memrealtime(); \
        const unsigned fcW = (unsigned)(f * nCells + ci); \
        if (lane == 0 && (fcW & 63u) == 0u) { g_fastClock[2 * ((fcW >> 6) & (kFastClockSlots - 1))] = fcT1 - fcT0; \
                                              g_fastClock[2 * ((fcW >> 6) & (kFastClockSlots - 1)) + 1] = fcR1 - fcR0; } } while (0)
#else
#define FAST_CLOCK_BEGIN
#define FAST_CLOCK_END do {} while (0)
#define FAST_SPAN(which) do {} while (0)
#define FAST_MID(which) do {} while (0)
#endif
constexpr int kFastWaves = 4;
#ifndef ORBX_FAST_WAVES
#define ORBX_FAST_WAVES 4   // waves per SIMD the kernel is compiled for
#endif
#ifndef ORBX_FAST_SKIP
#define ORBX_FAST_SKIP 0   // diagnostic builds (tools/fast_breakdown.py): 1 = no score pass, 2 = stop after the score pass, 4 = no staging loads
#endif

// Atomics on an LDS word whose address the compiler may see as a generic pointer (a body inlined behind pointer arguments): the cast names the
// address space, so the instruction is a ds_* whatever the optimiser merged around it.
typedef __attribute__((address_space(3))) unsigned LdsU32;
__device__ __forceinline__ void ldsAtomicAdd(unsigned* p, unsigned v) { (void)__hip_atomic_fetch_add((LdsU32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ldsAtomicMax(unsigned* p, unsigned v) { (void)__hip_atomic_fetch_max((LdsU32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// LDS operations of one wave execute in issue order, so lanes of a wave only need the COMPILER to keep the
// order of the stores before and the loads after this point.
__device__ __forceinline__ void waveLdsSync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// What the fused form needs to run the blur's lanes behind the FAST cells (small batches: the two kernels are independent, both only
// read the pyramid, and a launch costs ~6 us of latency whatever its size)
struct BlurTail { const BlurItem* items; const unsigned short* laneItem; int nLanes; uint8_t* blur; int fastChunks; };

// LDS of one FAST workgroup (four cell-waves): the pixel tiles behind 16 bytes of padding (the packed score pass reads the dword left of every
// row's first interior dword), the score tiles as an array of their own (their base is then one scalar, and the NMS pass's nine reads are
// immediate offsets of ONE address register; inside one array they sat 2 KB behind the pixel tile's base, past the reach of ds_read2's
// offsets: four v_add per trip), and the cells' x / y path codes (small batches: leaf tables).
template <int TS, int ROWS>
struct FastLds {
    static constexpr int kTileBytes = TS * ROWS, kScoreBytes = TS * (ROWS - 3);
    static constexpr int kTiles = 16 + kFastWaves * kTileBytes, kScores = kFastWaves * kScoreBytes, kCodes = kFastWaves * 2 * 64;
    static constexpr int kScoreOff = (kTiles + 15) & ~15, kCodeOff = (kScoreOff + kScores + 15) & ~15, kBytes = kCodeOff + kCodes;
};

// One cell on one wave (the whole of k_fast's work).  smem / scoreS / codeL: the workgroup's
// three LDS arrays (FastLds), chunk: the workgroup's group of four cells, f: the frame.  No workgroup barrier.
// BIG: the candidates are written in the two-dword format of frames beyond 4096 px (orbx_device.hpp: CandFmt).
template <int TS, int ROWS, bool BIG = false>
__device__ __forceinline__ void fastCell(const CellDesc* __restrict__ cells, int nCells, const LevelGeom* __restrict__ lv,
                                         const uint8_t* __restrict__ pyr, int iniTh, int minTh,
                                         typename CandFmt<BIG>::T* __restrict__ candSeg, unsigned* __restrict__ cellCount, LeafTables lt,
                                         uint8_t* smem, uint8_t* scoreS, uint8_t (*codeL)[2][64], int chunk, int f) {
    constexpr int kTileBytes = TS * ROWS;             // pixel tile
    constexpr int kScoreBytes = TS * (ROWS - 3);      // score tile: (ch + 2) rows <= ROWS - 4, and one more zero row for the NMS lanes past the last item
    constexpr int DW = TS / 4;                        // dwords per tile row
    constexpr int LPR = DW <= 16 ? 8 : 16;            // lanes per row while staging: a lane moves TWO dwords (one 8-byte load, one 8-byte LDS store)
    constexpr int RPI = 64 / LPR;                     // rows per staging step
    constexpr int STEPS = (ROWS + RPI - 1) / RPI;
    static_assert(DW % 2 == 0 && TS % 8 == 0, "dword pairs per tile row");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // scalar: the cell and its geometry load through the scalar unit
    const int ci = chunk * kFastWaves + wave;
    if (ci >= nCells) return;   // wave-uniform; the kernel has no workgroup barrier
    FAST_CLOCK_BEGIN
    const CellDesc c = cells[ci];
    const LevelGeom g = lv[c.level];
    uint8_t* tile = smem + 16 + wave * kTileBytes;
    uint8_t* score = scoreS + wave * kScoreBytes;
    const int roiW = c.roiW, roiH = c.roiH, cw = roiW - 6, ch = roiH - 6;
    // small batches (leaf tables): the x / y path codes of the cell's interior columns and rows are fetched WITH the ROI (lane i: column i and
    // row i) and kept in LDS: read from memory where the emit needs them they were two dependent L2 round trips at the end of every cell
    const bool leaf = lt.hist != nullptr && g.leafOK && f < lt.frames;      // wave-uniform
    uint8_t myXc = 0, myYc = 0;
    if (leaf) {
        myXc = lt.xcode[c.level * lt.XT + min(c.shiftX + 3 + lane, g.rectW - 1)];
        myYc = lt.ycode[c.level * lt.XT + min(c.shiftY + 3 + lane, g.rectH - 1)];
    }

    // ---- stage the ROI, re-aligned: tile byte k of row r is ROI pixel (k - 1, r), whatever the ROI's alignment in
    //      HBM, so the interior (ROI pixels 3 ..) always starts on a dword of the tile.  The packed passes walk the
    //      interior dword by dword: a 31- or 32-pixel cell is then 8 dwords wide, not 9, and its 8 x 32 four-pixel
    //      items are exactly 4 wave iterations.  Cost: the lane's right neighbour's dword (DPP) and one v_alignbyte. ----
    const int gx1 = kPadL + c.x0 - 1;                   // byte column of tile byte 0 in the bordered row
    const int gsh = gx1 & 3;                            // its offset inside the aligned dword the lanes load
    constexpr int mis = kFastTileShift;                              // tile byte of ROI pixel 0 (the passes below are written for any value)
    {
        // wave-uniform base + 32-bit lane offsets; rows / dword columns past the ROI are clamped, not predicated (their
        // tile bytes are never read by an interior pixel)
        const uint8_t* sp = pyr + c.pyrOff + (long long)f * c.pyrFrameBytes + (long long)(kEdge + c.y0) * c.pyrStride + (gx1 - gsh);
        const int dcol = lane & (LPR - 1), rsub = lane / LPR;                    // dcol: the lane's dword PAIR of the row
        const unsigned colOff = 8u * (unsigned)min(dcol, (gsh + roiW) >> 3);    // last pair holding a needed byte (its second dword still lies inside the bordered row)
        const unsigned off0 = (unsigned)__mul24(rsub, c.pyrStride) + colOff, offMax = (unsigned)__mul24(roiH - 1, c.pyrStride) + colOff;
        const unsigned stepOff = (unsigned)(RPI * c.pyrStride);
        uint2 w[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            if (ORBX_FAST_SKIP & 4) w[s] = uint2{0u, 0u};
            else __builtin_memcpy(&w[s], sp + min(off0 + s * stepOff, offMax), 8);      // (4-byte aligned: one global_load_dwordx2)
        }
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const int r = s * RPI + rsub;
            const unsigned next = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w[s].x, 0x130 /* wave_shl:1: lane i <- lane i+1 */, 0xF, 0xF, true);
            if (dcol < DW / 2 && r < ROWS)
                *(uint2*)(tile + r * TS + 8 * dcol) = uint2{__builtin_amdgcn_alignbyte(w[s].y, w[s].x, (unsigned)gsh), __builtin_amdgcn_alignbyte(next, w[s].y, (unsigned)gsh)};
        }
    }
    if (leaf) { codeL[wave][0][lane] = myXc; codeL[wave][1][lane] = myYc; }
    // zero the score tile (its 1-px apron stands for "outside the ROI interior")
#pragma unroll
    for (int i = 0; i < (kScoreBytes + 255) / 256; i++)
        if (lane * 4 + i * 256 < kScoreBytes) *(unsigned*)(score + lane * 4 + i * 256) = 0u;
    waveLdsSync();
    FAST_MID(0);

    // "items" of the packed passes: one tile dword (4 pixels) of an interior row; q0..q1 are the dwords that touch it
    const int q0 = (mis + 3) >> 2, q1 = (mis + 2 + cw) >> 2, nq = q1 - q0 + 1;
    const int nItems = nq * ch;
    // ---- pass 1: scores ----
    if (!(ORBX_FAST_SKIP & 1)) {
        // one lane = the four pixels of one tile dword (two packed pairs); 21 dword reads feed 4 scores.  Score row y+1
        // keeps the tile's column alignment, so the four scores are one dword store; bytes outside the interior
        // (first / last dword of a row) are written as 0 = "outside the ROI interior".
        const int lo = mis + 3 - 4 * q0, hi = mis + 3 + cw - 4 * q1;            // first valid byte of dword q0 / valid bytes of q1
        const unsigned maskFirst = 0xFFFFFFFFu << (8 * lo), maskLast = hi >= 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8 * hi));
        const int sy = (64 * c.itemRecip) >> 16, sx = 64 - sy * nq;           // 64 / nq, 64 % nq (exact: CellDesc::itemRecip)
        int y = (lane * c.itemRecip) >> 16, qi = lane - y * nq;                  // lane / nq, lane % nq
        // An LDS read is served in two groups of 32 lanes (banks = dword index mod 32).  With 8 items per row a group is four rows of 8 dwords, and
        // four CONSECUTIVE rows at 12 dwords per row start at banks 0, 12, 24, 4: the fourth collides with the first (every read of the pass
        // 2-way).  Rows 0, 2, 4, 6 start at 0, 24, 16, 8 (and 1, 3, 5, 7 at 12, 4, 28, 20): disjoint.  The order of the items does not matter here.
        if (DW == 12 && nq == 8) y = ((y & 3) << 1) | (y >> 2);
        auto scoreItem = [&](const uint8_t* base, uint8_t* dst, const unsigned m) {      // base: the item's dword in tile row y (= centre row - 3); dst: in score row y + 1
            unsigned L[7], C[7], R[7];
#pragma unroll
            for (int d = 0; d < 7; d++) {
                L[d] = *(const unsigned*)(base + d * TS - 4);
                C[d] = *(const unsigned*)(base + d * TS);
                R[d] = *(const unsigned*)(base + d * TS + 4);
            }
            unsigned rA[17], rB[17];
            pairRing<0>(L, C, R, rA);
            pairRing<2>(L, C, R, rB);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned sA = pairScore(rA);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned sB = pairScore(rB);
            *(unsigned*)dst = __builtin_amdgcn_perm(sB, sA, 0x06040200u) & m;
        };
        if (sx == 0) {
            // the items of a row divide the wave (8 items: the usual 29..32-px cells; 1, 2, 4, 16): a lane keeps its item column, so its edge mask is
            // fixed and a trip is "sy rows further": two pointer steps and a compare instead of re-deriving (item, row), the masks and the address
            const unsigned m = (qi == 0 ? maskFirst : 0xFFFFFFFFu) & (qi == nq - 1 ? maskLast : 0xFFFFFFFFu);
            const uint8_t* base = tile + y * TS + 4 * (q0 + qi);
            uint8_t* dst = score + (y + 1) * TS + 4 * (q0 + qi);
            for (; y < ch; y += sy, base += sy * TS, dst += sy * TS) scoreItem(base, dst, m);
        } else {
            for (; y < ch;) {
                unsigned m = qi == 0 ? maskFirst : 0xFFFFFFFFu;
                m = qi == nq - 1 ? (m & maskLast) : m;
                scoreItem(tile + y * TS + 4 * (q0 + qi), score + (y + 1) * TS + 4 * (q0 + qi), m);
                qi += sx; y += sy;
                if (qi >= nq) { qi -= nq; y++; }
            }
        }
    }
    waveLdsSync();
    if (ORBX_FAST_SKIP & 2) { if (lane == 0) cellCount[(long long)f * nCells + ci] = 0u; return; }

    FAST_MID(1);
    // ---- pass 2: strict local maxima, four pixels per lane (the score rows keep the tile's dword grid): nine dword
    //      reads, the 3 x 5 neighbour pairs by v_perm, packed 3-input maxima with minThFAST folded in, so
    //      "keep" is simply S > max.  Survivors are appended in raster order (lane order, then pixel order inside the
    //      lane) to a list that reuses the pixel tile (no longer needed) ----
    unsigned* list = (unsigned*)tile;                   // entry: x | y << 6 | S << 12
    int nMin = 0;
    {
        const unsigned thPair = (unsigned)minTh | ((unsigned)minTh << 16);
        const int sy = (64 * c.itemRecip) >> 16, sx = 64 - sy * nq;
        int y = (lane * c.itemRecip) >> 16, qi = lane - y * nq;
        // one trip of 64 items: base = the item's dword in score row y (the row above the centre row), xy = x | y << 6 of its pixel 0 (x may be
        // "negative": only kept pixels are used).  Lanes past the last item (y >= ch) take centre row ch + 1, the zero row below the interior: S = 0
        // keeps nothing, so the pass needs no "active" predicate
        auto nmsTrip = [&](const uint8_t* base, const unsigned xy) {
            unsigned U[3], M[3], D[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                U[k] = *(const unsigned*)(base + 4 * k - 4);
                M[k] = *(const unsigned*)(base + TS + 4 * k - 4);
                D[k] = *(const unsigned*)(base + 2 * TS + 4 * k - 4);
            }
            // pairs a..e = bytes (3,4) (4,5) (5,6) (6,7) (7,8) of L|C|R; up/down maxima per pair column, threshold folded in
            const unsigned Wa = pkmax3(pairAt<3>(U[0], U[1], U[2]), pairAt<3>(D[0], D[1], D[2]), thPair);
            const unsigned Wb = pkmax3(pairAt<4>(U[0], U[1], U[2]), pairAt<4>(D[0], D[1], D[2]), thPair);
            const unsigned Wc = pkmax3(pairAt<5>(U[0], U[1], U[2]), pairAt<5>(D[0], D[1], D[2]), thPair);
            const unsigned Wd = pkmax3(pairAt<6>(U[0], U[1], U[2]), pairAt<6>(D[0], D[1], D[2]), thPair);
            const unsigned We = pkmax3(pairAt<7>(U[0], U[1], U[2]), pairAt<7>(D[0], D[1], D[2]), thPair);
            const unsigned Ma = pairAt<3>(M[0], M[1], M[2]), Mc = pairAt<5>(M[0], M[1], M[2]), Me = pairAt<7>(M[0], M[1], M[2]);
            const unsigned sA = pairAt<4>(M[0], M[1], M[2]), sB = pairAt<6>(M[0], M[1], M[2]);
            const unsigned mA = pkmax3(pkmax3(Wa, Wb, Wc), Ma, Mc), mB = pkmax3(pkmax3(Wc, Wd, We), Mc, Me);
            unsigned dA, dB;                             // per half: S - max, saturated at 0: nonzero <=> strict maximum above minThFAST
            asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dA) : "v"(sA), "v"(mA));
            asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dB) : "v"(sB), "v"(mB));
            // two horizontal neighbours are never both strict maxima, so a pixel PAIR keeps at most one pixel: one ballot per pair, not per pixel
            const bool fA = dA != 0u, fB = dB != 0u;
            const unsigned long long bA = __ballot(fA), bB = __ballot(fB);
            if (bA | bB) {
                int at = nMin;                           // + kept pixels of the lower lanes: one v_mbcnt pair per ballot
                at = __builtin_amdgcn_mbcnt_hi((unsigned)(bA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bA, at));
                at = __builtin_amdgcn_mbcnt_hi((unsigned)(bB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bB, at));
                if (fA) { const bool hi = dA > 0xFFFFu; list[at++] = (xy + (hi ? 1u : 0u)) | ((hi ? sA >> 16 : sA & 0xFFFFu) << 12); }
                if (fB) { const bool hi = dB > 0xFFFFu; list[at] = (xy + (hi ? 3u : 2u)) | ((hi ? sB >> 16 : sB & 0xFFFFu) << 12); }
                nMin += __popcll(bA) + __popcll(bB);
            }
        };
        const int trips = (nItems + 63) >> 6;
        if (sx == 0) {      // a lane keeps its item column (scoreItem has the reason): a trip is sy rows further
            const uint8_t* base = score + y * TS + 4 * (q0 + qi);
            const uint8_t* const baseEnd = score + ch * TS + 4 * (q0 + qi);
            unsigned xy = (unsigned)(4 * (q0 + qi) - (mis + 3)) + ((unsigned)y << 6);
            for (int t = 0; t < trips; t++, base += sy * TS, xy += (unsigned)sy << 6) nmsTrip(base < baseEnd ? base : baseEnd, xy);
        } else {
            for (int t = 0; t < trips; t++) {
                nmsTrip(score + min(y, ch) * TS + 4 * (q0 + qi), (unsigned)(4 * (q0 + qi) - (mis + 3)) + ((unsigned)y << 6));
                qi += sx; y += sy;
                if (qi >= nq) { qi -= nq; y++; }
            }
        }
    }
    waveLdsSync();
    // the reference retries the cell at minThFAST only when the first call returned nothing (:835-838)
    int nIni = 0;
    for (int i0 = 0; i0 < nMin; i0 += 64)
        nIni += __popcll(__ballot(i0 + lane < nMin && (int)(list[i0 + lane] >> 12) > iniTh));
    const bool useIni = nIni > 0;
    const int total = useIni ? nIni : nMin;
    // every cell owns a fixed, exactly sized segment of the level's candidate arena (no two 8-adjacent NMS survivors
    // => at most ceil(cw/2)*ceil(ch/2) of them), so the emit needs no atomic; the quad-tree kernel compacts the
    // segments in cell order, which is the reference's vToDistributeKeys order (cell row, cell column, y, x)
    if (lane == 0) cellCount[(long long)f * nCells + ci] = (unsigned)total;
    FAST_CLOCK_END;
    FAST_SPAN(1);
    FAST_MID(2);
    if (total == 0) return;
    unsigned base = 0;
    typename CandFmt<BIG>::T* outPos = candSeg + g.candOff + (long long)f * g.candCap + c.segOff;
    const unsigned segCap = (unsigned)(((cw + 1) >> 1) * ((ch + 1) >> 1));
    const int th = useIni ? iniTh : minTh;
    // Small batches: the quad-tree's first sweep is done here, by 800 waves instead of one workgroup per level (k_octree_body.inc,
    // preCounted): every kept key adds 1 to its leaf cell of its root's 32 x 32 grid and offers (response, then smallest segment slot = first
    // in the reference's vector) as the leaf's best key.  Both operations commute, so the order of the cells' waves does not matter.  A cell
    // touches only a small rectangle of leaf cells (x / y path codes are monotone), so the wave first collects them in LDS (the score tile
    // is dead by now) and sends ONE pair of L2 atomics per touched leaf instead of one per key (100 k keys per frame otherwise: the L2
    // atomics on a few hundred hot lines cost more than the sweep they replace).
    constexpr int kLeafCap = kScoreBytes / 8;
    unsigned *tHist = (unsigned*)score, *tBest = tHist + kLeafCap;
    int xc0 = 0, yc0 = 0, nxl = 1, nLeafLocal = 0;
    const uint8_t *xcL = codeL[wave][0], *ycL = codeL[wave][1];      // code of interior column x / row y of the cell (pixel shift + 3 + x, clamped to the rectangle)
    if (leaf) {
        xc0 = xcL[0]; yc0 = ycL[0];
        nxl = (int)xcL[cw - 1] - xc0 + 1;
        const int nyl = (int)ycL[ch - 1] - yc0 + 1;
        nLeafLocal = nxl * nyl <= kLeafCap ? nxl * nyl : 0;      // 0: a cell over too many leaf cells (tiny leaves) goes to L2 key by key
        for (int e = lane; e < nLeafLocal; e += 64) { tHist[e] = 0u; tBest[e] = 0u; }
        waveLdsSync();
    }
    for (int i0 = 0; i0 < nMin; i0 += 64) {
        const unsigned e = i0 + lane < nMin ? list[i0 + lane] : 0u;
        const int s = (int)(e >> 12), x = (int)(e & 63), y = (int)((e >> 6) & 63);
        const bool keep = s > th;                       // entries hold S > minTh; 0 marks "past the end"
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const unsigned at = base + __popcll(m & ((1ull << lane) - 1));
            if (at < segCap) {
                const unsigned kx = (unsigned)(c.shiftX + x + 3), ky = (unsigned)(c.shiftY + y + 3);
                const typename CandFmt<BIG>::T w = CandFmt<BIG>::make(kx, ky, (unsigned)(s - 1));   // response = S - 1
                outPos[at] = w;
                if (leaf) {
                    const int xc = xcL[x], yc = ycL[y];
                    const unsigned val = CandFmt<BIG>::respKey(w) | (0xffffffu - ((unsigned)c.segOff + at));
                    if (nLeafLocal) {
                        const int li = (yc - yc0) * nxl + (xc - xc0);
                        // (LDS by type: with the tables reached through the body's pointer arguments the compiler merged this pair with the
                        // global pair of the other branch into ONE atomic on a generic pointer - a flat_atomic, docs/history/DESIGN_rounds_1-5.md §4i's construct)
                        ldsAtomicAdd(&tHist[li], 1u);
                        ldsAtomicMax(&tBest[li], val);
                    } else {
                        const unsigned cellOfRoot = leafTableEntry(lt, f, c.level, xc, yc);
                        atomicAdd(lt.hist + cellOfRoot, 1);
                        atomicMax(lt.best + cellOfRoot, val);
                    }
                }
            }
        }
        base += __popcll(m);
    }
    if (nLeafLocal) {
        waveLdsSync();
        for (int e = lane; e < nLeafLocal; e += 64) {
            const unsigned n = tHist[e];
            if (n) {
                const int er = (int)(((float)e + 0.5f) * __frcp_rn((float)nxl));      // e / nxl (exact: e < 1024, the quotient is >= 0.5 / nxl away from an integer)
                const int xc = xc0 + (e - er * nxl), yc = yc0 + er;
                const unsigned cellOfRoot = leafTableEntry(lt, f, c.level, xc, yc);
                atomicAdd(lt.hist + cellOfRoot, (int)n);
                atomicMax(lt.best + cellOfRoot, tBest[e]);
            }
        }
    }
    FAST_SPAN(1);      // (diagnostic builds: the wave's end including the emit; the stamp above stays as the end of a cell without keys)
}


}  // namespace orbx
