// k_blur_body.hpp — the 7x7 blur's arithmetic as device functions: k_blur.hip launches the lanes as a kernel of their own, k_fast.hip
// appends them to the FAST grid in small batches (one launch fewer on the latency path), and k_pyramid.hip blurs level l-1 from the LDS
// tile the resize of level l has staged anyway (large batches: the level is not read from HBM a second time).  See k_blur.hip for the
// arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

__device__ __forceinline__ unsigned hsum4(unsigned lo, unsigned hi) {
    // lo = pixels x-3..x, hi = pixels x+1..x+4 (the last one weighted 0)
    const unsigned KLO = 18u | (34u << 8) | (49u << 16) | (55u << 24), KHI = 49u | (34u << 8) | (18u << 16);
    return __builtin_amdgcn_udot4(lo, KLO, __builtin_amdgcn_udot4(hi, KHI, 0u, false), false);
}

// One block of kBlurRows output rows x 4 columns.  load(i, d0, d1, d2): the three aligned dwords (pixels x0-4 .. x0+7) of input row i,
// i = 0 .. kBlurRows + 5 (input row i is output row i - 3); store(orow, word): the four blurred pixels of output row orow.
// Vertical pass on row sums packed two rows to a register (u16 halves: a row sum is <= 255 * 257 = 65535): rows (2k, 2k + 1) of the
// block form pair k, a 7-row window is four pairs (one of them half used), and v_dot2_u32_u16 against the matching weight pair
// adds two taps per instruction — four instead of three adds and four multiply-adds per pixel.  The loop is unrolled, so the
// parity of the output row (which halves of which pairs it uses) is static.  The sums are the same integers as tap by tap.
template <int kBlurRows, class Load, class Store>
__device__ __forceinline__ void blurBlock(Load load, Store store) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto dot2 = [](unsigned pair, unsigned short w0, unsigned short w1, unsigned acc) {
        return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), u16x2{w0, w1}, acc, false);
    };
    unsigned P[4][4] = {}, lo[4] = {};      // the four youngest complete pairs (oldest first); the even row waiting for its partner
#pragma unroll
    for (int i = 0; i < kBlurRows + 6; i++) {
        unsigned d0, d1, d2;
        load(i, d0, d1, d2);
        unsigned hn[4];
        hn[0] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 1), __builtin_amdgcn_alignbyte(d2, d1, 1));
        hn[1] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d2, d1, 2));
        hn[2] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 3), __builtin_amdgcn_alignbyte(d2, d1, 3));
        hn[3] = hsum4(d1, d2);
        if (i & 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                P[0][j] = P[1][j]; P[1][j] = P[2][j]; P[2][j] = P[3][j];
                P[3][j] = (hn[j] << 16) | lo[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) lo[j] = hn[j];
        }
        if (i >= 6) {
            unsigned t[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                unsigned sacc;
                if (i & 1)      // rows i-6 .. i = high half of P[0], P[1], P[2], P[3]
                    sacc = dot2(P[3][j], 34, 18, dot2(P[2][j], 55, 49, dot2(P[1][j], 34, 49, dot2(P[0][j], 0, 18, 32768u))));
                else            // rows i-6 .. i = P[1], P[2], P[3], the waiting even row
                    sacc = dot2(lo[j], 18, 0, dot2(P[3][j], 49, 34, dot2(P[2][j], 49, 55, dot2(P[1][j], 18, 34, 32768u))));
                t[j] = min(sacc, 0x00FFFFFFu);      // (s + 32768) >> 16 clamped to 255 is byte 2 of this
            }
            const unsigned p01 = __builtin_amdgcn_perm(t[1], t[0], 0x0C0C0602u), p23 = __builtin_amdgcn_perm(t[3], t[2], 0x0C0C0602u);
            store(i - 6, (p23 << 16) | p01);
        }
    }
}

// The same arithmetic for a run of nOut output rows whose length is only known at run time (k_pyr_cols: a region's share of a level, dealt over
// the threads that blur): input rows i = 0 .. nOut + 5 through load(i, ...), output row r through store(r, word).  Two rows per trip, so that
// which halves of which pairs an output row uses stays static (an even row first: the run starts on an even i); an odd nOut computes one
// output more than it stores.
template <class Load, class Store>
__device__ __forceinline__ void blurRun(const int nOut, Load load, Store store) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto dot2 = [](unsigned pair, unsigned short w0, unsigned short w1, unsigned acc) {
        return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), u16x2{w0, w1}, acc, false);
    };
    unsigned P[4][4] = {}, lo[4] = {};
    auto hsums = [&](int i, unsigned (&hn)[4]) {
        unsigned d0, d1, d2;
        load(i, d0, d1, d2);
        hn[0] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 1), __builtin_amdgcn_alignbyte(d2, d1, 1));
        hn[1] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d2, d1, 2));
        hn[2] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 3), __builtin_amdgcn_alignbyte(d2, d1, 3));
        hn[3] = hsum4(d1, d2);
    };
    auto pack = [](const unsigned (&t)[4]) {
        const unsigned p01 = __builtin_amdgcn_perm(t[1], t[0], 0x0C0C0602u), p23 = __builtin_amdgcn_perm(t[3], t[2], 0x0C0C0602u);
        return (p23 << 16) | p01;
    };
    auto even = [&](int i) {          // row i (even) arrives: it waits for its partner
        unsigned hn[4];
        hsums(i, hn);
#pragma unroll
        for (int j = 0; j < 4; j++) lo[j] = hn[j];
    };
    auto odd = [&](int i) {           // row i (odd) arrives: the pair (i - 1, i) is complete
        unsigned hn[4];
        hsums(i, hn);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            P[0][j] = P[1][j]; P[1][j] = P[2][j]; P[2][j] = P[3][j];
            P[3][j] = (hn[j] << 16) | lo[j];
        }
    };
#pragma unroll
    for (int i = 0; i < 6; i += 2) { even(i); odd(i + 1); }
    for (int i = 6; i < nOut + 6; i += 2) {
        unsigned t[4];
        even(i);                      // rows i-6 .. i = P[1], P[2], P[3], the waiting even row
#pragma unroll
        for (int j = 0; j < 4; j++)
            t[j] = min(dot2(lo[j], 18, 0, dot2(P[3][j], 49, 34, dot2(P[2][j], 49, 55, dot2(P[1][j], 18, 34, 32768u)))), 0x00FFFFFFu);
        store(i - 6, pack(t));
        odd(i + 1);                   // rows i-5 .. i+1 = high half of P[0], P[1], P[2], P[3]
#pragma unroll
        for (int j = 0; j < 4; j++)
            t[j] = min(dot2(P[3][j], 34, 18, dot2(P[2][j], 55, 49, dot2(P[1][j], 34, 49, dot2(P[0][j], 0, 18, 32768u)))), 0x00FFFFFFu);
        if (i - 5 < nOut) store(i - 5, pack(t));
    }
}

// items: one per (level, row block); lanes of the whole grid.x enumerate (item, column group) pairs; laneItem[lane]
// names the lane's item (a per-thread binary search would start every workgroup with eight dependent loads).
template <int kBlurRows>
__device__ __forceinline__ void blurLanes(const BlurItem* __restrict__ items, const unsigned short* __restrict__ laneItem,
                                          int nLanes, const LevelGeom* __restrict__ lv,
                                          const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int chunk, int f) {
    const int gl = chunk * 256 + threadIdx.x;
    if (gl >= nLanes) return;
    const BlurItem it = items[laneItem[gl]];      // host table: the (level, row block) this lane works on
    const LevelGeom g = lv[it.level];
    const int grp = gl - it.firstLane;          // column group inside the row block
    const int x0 = 4 * grp, y0 = it.y0;
    // dword containing pixels x0-4..x0-1 of row y0-3 (kPadL keeps x0 dword-aligned)
    const uint8_t* sp = pyr + g.pyrOff + (long long)f * g.pyrFrameBytes + (long long)(kEdge + y0 - 3) * g.pyrStride + kPadL + x0 - 4;
    uint8_t* dp = blur + g.blurOff + (long long)f * g.blurFrameBytes + (long long)y0 * g.blurStride + x0;
    const int rowsValid = min(kBlurRows, g.h - y0);     // output rows this block really owns
    const int lastIn = g.h + kEdge - 1 - (y0 - 3);      // input rows below the bordered buffer are clamped (their outputs are not stored)
    blurBlock<kBlurRows>(
        [&](int i, unsigned& d0, unsigned& d1, unsigned& d2) {
            const int r = i < lastIn ? i : lastIn;
            const unsigned* row = (const unsigned*)(sp + (long long)r * g.pyrStride);
            d0 = row[0]; d1 = row[1]; d2 = row[2];
        },
        [&](int orow, unsigned w) {
            if (orow < rowsValid) *(unsigned*)(dp + (long long)orow * g.blurStride) = w;
        });
}

}  // namespace orbx
