// k_blur_body.hpp — the 7x7 blur's arithmetic as device functions: k_blur.hip launches the lanes as a kernel of their own, k_fast.hip
// appends them to the FAST grid in small batches (one launch fewer on the latency path), and k_pyramid.hip blurs level l-1 from the LDS
// tile the resize of level l has staged anyway (large batches: the level is not read from HBM a second time).  See k_blur.hip for the
// arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

// The horizontal 7-tap sums of the four pixels x0 .. x0 + 3 out of the three aligned dwords d0 | d1 | d2 = pixels x0 - 4 .. x0 + 7: pixel j reads
// bytes j + 1 .. j + 7 of that 12-byte string, so its taps are a v_dot4_u32_u8 per dword it touches against the kernel shifted to the pixel's
// place (2 + 3 + 3 + 2 multiply-adds; shifting the string to each pixel first - three v_alignbyte pairs and 2 x 4 multiply-adds - was 14).
// The sums are the same integers whatever the order.
__device__ __forceinline__ void hsums4(unsigned d0, unsigned d1, unsigned d2, unsigned (&hn)[4]) {
    constexpr unsigned K0 = 18, K1 = 34, K2 = 49, K3 = 55, K4 = 49, K5 = 34, K6 = 18;      // getGaussianKernel(7, 2) x 256, rounded as cv::GaussianBlur's fixed point does
    auto w4 = [](unsigned b0, unsigned b1, unsigned b2, unsigned b3) constexpr { return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24); };
    hn[0] = __builtin_amdgcn_udot4(d0, w4(0, K0, K1, K2), __builtin_amdgcn_udot4(d1, w4(K3, K4, K5, K6), 0u, false), false);
    hn[1] = __builtin_amdgcn_udot4(d0, w4(0, 0, K0, K1), __builtin_amdgcn_udot4(d1, w4(K2, K3, K4, K5), __builtin_amdgcn_udot4(d2, w4(K6, 0, 0, 0), 0u, false), false), false);
    hn[2] = __builtin_amdgcn_udot4(d0, w4(0, 0, 0, K0), __builtin_amdgcn_udot4(d1, w4(K1, K2, K3, K4), __builtin_amdgcn_udot4(d2, w4(K5, K6, 0, 0), 0u, false), false), false);
    hn[3] = __builtin_amdgcn_udot4(d1, w4(K0, K1, K2, K3), __builtin_amdgcn_udot4(d2, w4(K4, K5, K6, 0), 0u, false), false);
}
// (s + 32768) >> 16 saturated to 255 for four sums (a white patch reaches 257): the high halves of two sums side by side (v_perm), v_sat_pk_u8_i16
// on each pair, the two byte pairs joined (v_perm: nothing depends on what the saturating pack leaves in its upper half) - five instructions
// where a v_min_u32 per sum and the byte gather were seven.
__device__ __forceinline__ unsigned satPack4(const unsigned (&t)[4]) {
    unsigned p01, p23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(p01) : "v"(__builtin_amdgcn_perm(t[1], t[0], 0x07060302u)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(p23) : "v"(__builtin_amdgcn_perm(t[3], t[2], 0x07060302u)));
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
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
        hsums4(d0, d1, d2, hn);
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
                t[j] = sacc;
            }
            store(i - 6, satPack4(t));
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
        hsums4(d0, d1, d2, hn);
    };
    auto pack = [](const unsigned (&t)[4]) { return satPack4(t); };
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
            t[j] = dot2(lo[j], 18, 0, dot2(P[3][j], 49, 34, dot2(P[2][j], 49, 55, dot2(P[1][j], 18, 34, 32768u))));
        store(i - 6, pack(t));
        odd(i + 1);                   // rows i-5 .. i+1 = high half of P[0], P[1], P[2], P[3]
#pragma unroll
        for (int j = 0; j < 4; j++)
            t[j] = dot2(P[3][j], 34, 18, dot2(P[2][j], 55, 49, dot2(P[1][j], 34, 49, dot2(P[0][j], 0, 18, 32768u))));
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
