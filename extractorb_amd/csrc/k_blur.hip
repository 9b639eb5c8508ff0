// k_blur.hip — 7x7 sigma-2 integer Gaussian of every pyramid level (reference ORBextractor.cc:1126-1127).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {
// ================================================================================================
// Gaussian blur 7x7, sigma 2: integer taps {18,34,49,55,49,34,18} per axis, (sum + 2^15) >> 16, saturate
// (SURVEY.md A.2).  The bordered pyramid already holds the REFLECT_101 frame the blur needs.
// One workgroup = one 64x32 output tile; tiles of all levels are enumerated by a table.
// ================================================================================================
constexpr int kBlurTW = 64, kBlurTH = 32;

__global__ __launch_bounds__(256) void k_blur(const BlurTile* __restrict__ tiles, const LevelGeom* __restrict__ lv,
                                               const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
    __shared__ uint8_t in[kBlurTH + 6][kBlurTW + 8];       // 38 x 72
    __shared__ uint16_t hs[kBlurTH + 6][kBlurTW];          // row sums <= 257*255 = 65535
    const BlurTile t = tiles[blockIdx.x];
    const LevelGeom g = lv[t.level];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int x0 = t.tx * kBlurTW, y0 = t.ty * kBlurTH;
    const uint8_t* sp = pyr + g.pyrOff + (long long)f * g.pyrFrameBytes + (long long)kEdge * g.pyrStride + kPadL;
    // the bordered buffer has 19 valid pixels on each side of the interior; clamp reads to it
    const int xLo = -kEdge, xHi = g.w + kEdge - 1, yLo = -kEdge, yHi = g.h + kEdge - 1;
    for (int i = tid; i < (kBlurTH + 6) * (kBlurTW + 6); i += 256) {
        const int r = i / (kBlurTW + 6), c = i - r * (kBlurTW + 6);
        int gx = x0 + c - 3, gy = y0 + r - 3;
        gx = gx < xLo ? xLo : (gx > xHi ? xHi : gx);
        gy = gy < yLo ? yLo : (gy > yHi ? yHi : gy);
        in[r][c] = sp[(long long)gy * g.pyrStride + gx];
    }
    __syncthreads();
    for (int i = tid; i < (kBlurTH + 6) * kBlurTW; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t* p = &in[r][c];
        const int s = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 49 * (p[2] + p[4]) + 55 * p[3];
        hs[r][c] = (uint16_t)s;
    }
    __syncthreads();
    uint8_t* dp = blur + g.blurOff + (long long)f * g.blurFrameBytes;
    for (int i = tid; i < kBlurTH * kBlurTW; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int gx = x0 + c, gy = y0 + r;
        if (gx >= g.w || gy >= g.h) continue;
        const int s = 18 * (hs[r][c] + hs[r + 6][c]) + 34 * (hs[r + 1][c] + hs[r + 5][c]) +
                      49 * (hs[r + 2][c] + hs[r + 4][c]) + 55 * hs[r + 3][c];
        int v = (s + 32768) >> 16;
        v = v > 255 ? 255 : v;
        dp[(long long)gy * g.blurStride + gx] = (uint8_t)v;
    }
}

void launchBlur(hipStream_t st, const BlurTile* tiles, int nTiles, const LevelGeom* lv, const uint8_t* pyr,
                uint8_t* blur, int B) {
    hipLaunchKernelGGL(k_blur, dim3(nTiles, B), dim3(256), 0, st, tiles, lv, pyr, blur);
}

}  // namespace orbx
