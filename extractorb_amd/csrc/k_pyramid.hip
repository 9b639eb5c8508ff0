// k_pyramid.hip — image pyramid: bordered level-0 copy and the fixed-point bilinear resize chain
// (reference ORBextractor.cc:1164-1219; cv::resize / copyMakeBorder semantics: SURVEY.md A.1, A.4).
//
// Every thread produces one aligned dword (4 pixels) of a bordered destination row, so stores are coalesced
// 256-B wave stores.  copyMakeBorder(REFLECT_101) is fused: a border byte recomputes the interior pixel it
// mirrors, so a level never needs a second pass.  Level l depends on the rounded u8 pixels of level l-1
// (a 7-deep chain the reference defines), hence one launch per level; the first launch builds level 0
// (copy) AND level 1 (resize straight from the caller's image), which are independent of each other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

__device__ __forceinline__ int reflect101(int p, int n) {
    // BORDER_REFLECT_101 for |overshoot| < n (the border is 19 px, every level is wider)
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

// cv::resize(INTER_LINEAR) 8u: horizontal pass in 11-bit fixed point, vertical pass with the (x>>4, >>16, +2>>2) rounding
__device__ __forceinline__ unsigned bilinear(const uint8_t* r0, const uint8_t* r1, ResizeX cx, int b0, int b1) {
    const int h0 = r0[cx.sx0] * cx.a0 + r0[cx.sx1] * cx.a1;
    const int h1 = r1[cx.sx0] * cx.a0 + r1[cx.sx1] * cx.a1;
    return (unsigned)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
}

constexpr int kPyrRows = 8;   // destination rows per thread: the x coefficients are loaded once and the loads of all
                              // rows are in flight together (the kernels are latency-, not bandwidth-bound)

static __host__ __device__ inline int rowDwords(const LevelGeom& g) { return (kPadL - kEdge + g.w + 2 * kEdge + 3) / 4; }
static __host__ __device__ inline int rowGroups(const LevelGeom& g) { return (g.pyrRows + kPyrRows - 1) / kPyrRows; }

// Thread t of a level: dword column t % rowDwords, rows [kPyrRows * (t / rowDwords), +kPyrRows) of the bordered buffer.
// src: interior pixel (0,0) of the source image of frame 0, rows `srcStride` apart, frames `srcFrame` apart.
__device__ __forceinline__ void resizeDwordColumn(const uint8_t* __restrict__ src, long long srcStride, long long srcFrame,
                                                  const LevelGeom& d, const ResizeX* __restrict__ xt,
                                                  const ResizeX* __restrict__ yt, uint8_t* __restrict__ pyr, int t, int f) {
    const int nd = rowDwords(d);
    const int rg = t / nd, dw = t - rg * nd;
    const int wB = d.w + 2 * kEdge, bc0 = 4 * dw;
    ResizeX cx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int bx = bc0 + j - (kPadL - kEdge);               // bordered x of this byte
        bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);    // bytes of the dword outside the bordered row are padding
        cx[j] = xt[reflect101(bx - kEdge, d.w)];
    }
    const uint8_t* sp = src + (long long)f * srcFrame;
    uint8_t* dst = pyr + d.pyrOff + (long long)f * d.pyrFrameBytes + bc0;
    ResizeX cy[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const int by = min(rg * kPyrRows + r, d.pyrRows - 1);
        cy[r] = yt[reflect101(by - kEdge, d.h)];
    }
    unsigned out[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const uint8_t* r0 = sp + (long long)cy[r].sx0 * srcStride;
        const uint8_t* r1 = sp + (long long)cy[r].sx1 * srcStride;
        unsigned o = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) o |= bilinear(r0, r1, cx[j], cy[r].a0, cy[r].a1) << (8 * j);
        out[r] = o;
    }
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const int by = rg * kPyrRows + r;
        if (by < d.pyrRows) *(unsigned*)(dst + (long long)by * d.pyrStride) = out[r];
    }
}

// 1-D grid over threads [0, T0 + T1): the first T0 copy the caller's image into bordered level 0, the rest build
// bordered level 1 from the caller's image (== level 0's interior).  blockIdx.y = frame.
__global__ __launch_bounds__(256) void k_pyr_first(const uint8_t* __restrict__ img, long long stride, long long frameStride,
                                                    LevelGeom g0, LevelGeom g1, int T0, int T1, const ResizeX* __restrict__ xt,
                                                    const ResizeX* __restrict__ yt, uint8_t* __restrict__ pyr) {
    int t = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (t < T0) {
        const int nd = rowDwords(g0);
        const int rg = t / nd, dw = t - rg * nd;
        const int wB = g0.w + 2 * kEdge, bc0 = 4 * dw;
        int sx[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int bx = bc0 + j - (kPadL - kEdge);
            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
            sx[j] = reflect101(bx - kEdge, g0.w);
        }
        const uint8_t* sp = img + (long long)f * frameStride;
        uint8_t* dst = pyr + g0.pyrOff + (long long)f * g0.pyrFrameBytes + bc0;
        unsigned out[kPyrRows];
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const int by = min(rg * kPyrRows + r, g0.pyrRows - 1);
            const uint8_t* row = sp + (long long)reflect101(by - kEdge, g0.h) * stride;
            out[r] = (unsigned)row[sx[0]] | ((unsigned)row[sx[1]] << 8) | ((unsigned)row[sx[2]] << 16) | ((unsigned)row[sx[3]] << 24);
        }
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const int by = rg * kPyrRows + r;
            if (by < g0.pyrRows) *(unsigned*)(dst + (long long)by * g0.pyrStride) = out[r];
        }
    } else if (t < T0 + T1) {
        resizeDwordColumn(img, stride, frameStride, g1, xt, yt, pyr, t - T0, f);
    }
}

// 1-D grid over the threads of level d (built from level s of the pyramid); blockIdx.y = frame.
__global__ __launch_bounds__(256) void k_resize(LevelGeom s, LevelGeom d, int T, const ResizeX* __restrict__ xt,
                                                 const ResizeX* __restrict__ yt, uint8_t* __restrict__ pyr) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (t >= T) return;
    const uint8_t* src = pyr + s.pyrOff + (long long)kEdge * s.pyrStride + kPadL;
    resizeDwordColumn(src, s.pyrStride, s.pyrFrameBytes, d, xt, yt, pyr, t, f);
}

void launchPyrFirst(hipStream_t st, const uint8_t* img, long long stride, long long frameStride, const LevelGeom& g0,
                    const LevelGeom* g1, const ResizeX* xt, const ResizeX* yt, uint8_t* pyr, int B) {
    const int T0 = rowDwords(g0) * rowGroups(g0), T1 = g1 ? rowDwords(*g1) * rowGroups(*g1) : 0;
    hipLaunchKernelGGL(k_pyr_first, dim3((T0 + T1 + 255) / 256, B), dim3(256), 0, st, img, stride, frameStride, g0,
                       g1 ? *g1 : g0, T0, T1, xt, yt, pyr);
}
void launchResize(hipStream_t st, const LevelGeom& s, const LevelGeom& d, const ResizeX* xt, const ResizeX* yt,
                  uint8_t* pyr, int B) {
    const int T = rowDwords(d) * rowGroups(d);
    hipLaunchKernelGGL(k_resize, dim3((T + 255) / 256, B), dim3(256), 0, st, s, d, T, xt, yt, pyr);
}

}  // namespace orbx
