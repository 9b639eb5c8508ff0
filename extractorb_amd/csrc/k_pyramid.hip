// k_pyramid.hip — image pyramid: bordered level 0 copy and the fixed-point bilinear resize chain
// (reference ORBextractor.cc:1164-1219; cv::resize / copyMakeBorder semantics: SURVEY.md A.1, A.4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {
__device__ __forceinline__ int reflect101(int p, int n) {
    // BORDER_REFLECT_101 for |overshoot| < n (the border is 19 px, every level is wider)
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

// ================================================================================================
// Pyramid
// ================================================================================================
// grid (ceil((w+38)/256), h+38, B).  Thread = one byte of the bordered level-0 buffer.
__global__ __launch_bounds__(256) void k_level0(const uint8_t* __restrict__ src, long long stride,
                                                 long long frameStride, LevelGeom g, uint8_t* __restrict__ pyr) {
    const int bx = blockIdx.x * 256 + threadIdx.x, by = blockIdx.y, f = blockIdx.z;
    if (bx >= g.w + 2 * kEdge) return;
    const int sx = reflect101(bx - kEdge, g.w), sy = reflect101(by - kEdge, g.h);
    uint8_t* dst = pyr + g.pyrOff + (long long)f * g.pyrFrameBytes;
    dst[(long long)by * g.pyrStride + bx + (kPadL - kEdge)] = src[f * frameStride + sy * stride + sx];
}

// cv::resize(INTER_LINEAR) 8u fixed point + copyMakeBorder(REFLECT_101), fused: border
// bytes recompute the interior pixel they mirror, so the level needs no second pass.
__global__ __launch_bounds__(256) void k_resize(LevelGeom s, LevelGeom d, const ResizeX* __restrict__ xt,
                                                 const ResizeX* __restrict__ yt, uint8_t* __restrict__ pyr) {
    const int bx = blockIdx.x * 256 + threadIdx.x, by = blockIdx.y, f = blockIdx.z;
    if (bx >= d.w + 2 * kEdge) return;
    const int ix = reflect101(bx - kEdge, d.w), iy = reflect101(by - kEdge, d.h);
    const ResizeX cx = xt[ix], cy = yt[iy];
    const uint8_t* sp = pyr + s.pyrOff + (long long)f * s.pyrFrameBytes + (long long)kEdge * s.pyrStride + kPadL;
    const uint8_t* r0 = sp + (long long)cy.sx0 * s.pyrStride;
    const uint8_t* r1 = sp + (long long)cy.sx1 * s.pyrStride;
    const int h0 = r0[cx.sx0] * cx.a0 + r0[cx.sx1] * cx.a1;
    const int h1 = r1[cx.sx0] * cx.a0 + r1[cx.sx1] * cx.a1;
    const int v = (((cy.a0 * (h0 >> 4)) >> 16) + ((cy.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
    uint8_t* dst = pyr + d.pyrOff + (long long)f * d.pyrFrameBytes;
    dst[(long long)by * d.pyrStride + bx + (kPadL - kEdge)] = (uint8_t)v;
}

void launchLevel0(hipStream_t st, const uint8_t* src, long long stride, long long frameStride, const LevelGeom& g,
                  uint8_t* pyr, int B) {
    dim3 grid((g.w + 2 * kEdge + 255) / 256, g.h + 2 * kEdge, B);
    hipLaunchKernelGGL(k_level0, grid, dim3(256), 0, st, src, stride, frameStride, g, pyr);
}
void launchResize(hipStream_t st, const LevelGeom& s, const LevelGeom& d, const ResizeX* xt, const ResizeX* yt,
                  uint8_t* pyr, int B) {
    dim3 grid((d.w + 2 * kEdge + 255) / 256, d.h + 2 * kEdge, B);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, st, s, d, xt, yt, pyr);
}

}  // namespace orbx
