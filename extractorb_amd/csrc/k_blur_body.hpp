// k_blur_body.hpp — the lanes of the 7x7 blur as a device function: k_blur.hip launches them as a kernel of their own, k_fast.hip
// appends them to the FAST grid in small batches (one launch fewer on the latency path).  See k_blur.hip for the arithmetic.
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

    unsigned h[7][4] = {};
#pragma unroll
    for (int i = 0; i < kBlurRows + 6; i++) {
        const int r = i < lastIn ? i : lastIn;
        const unsigned* row = (const unsigned*)(sp + (long long)r * g.pyrStride);
        const unsigned d0 = row[0], d1 = row[1], d2 = row[2];
        unsigned hn[4];
        hn[0] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 1), __builtin_amdgcn_alignbyte(d2, d1, 1));
        hn[1] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 2), __builtin_amdgcn_alignbyte(d2, d1, 2));
        hn[2] = hsum4(__builtin_amdgcn_alignbyte(d1, d0, 3), __builtin_amdgcn_alignbyte(d2, d1, 3));
        hn[3] = hsum4(d1, d2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int t = 0; t < 6; t++) h[t][j] = h[t + 1][j];
            h[6][j] = hn[j];
        }
        if (i >= 6) {
            unsigned outw = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // 24-bit multiplies (row sums <= 2 * 65535): a 32-bit v_mul_lo_u32 issues at a quarter of the rate
                const unsigned s = __umul24(18u, h[0][j] + h[6][j]) + __umul24(34u, h[1][j] + h[5][j]) + __umul24(49u, h[2][j] + h[4][j]) +
                                   __umul24(55u, h[3][j]);
                unsigned v = (s + 32768u) >> 16;
                v = v > 255u ? 255u : v;
                outw |= v << (8 * j);
            }
            const int orow = i - 6;
            if (orow < rowsValid) *(unsigned*)(dp + (long long)orow * g.blurStride) = outw;
        }
    }
}

}  // namespace orbx
