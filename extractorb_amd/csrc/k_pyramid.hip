// k_pyramid.hip — image pyramid: bordered level-0 copy and the fixed-point bilinear resize chain
// (reference ORBextractor.cc:1164-1219; cv::resize / copyMakeBorder semantics: SURVEY.md A.1, A.4).
//
// One workgroup = one 256 x 32 tile of a bordered destination level; a thread owns one aligned dword column
// (4 pixels) over 8 rows, so stores are coalesced 256-B wave stores.  The tile's source footprint — a rectangle
// the host derives from the coefficient tables (they are monotonic and REFLECT_101 only folds indices back inside)
// — is staged in LDS with coalesced dword loads; the 4 taps of a pixel are then LDS byte reads (global byte
// gathers run at a quarter of the dword rate, and unaligned 16-bit LDS reads at half rate).
// copyMakeBorder(REFLECT_101) is fused: a border byte recomputes the interior pixel it mirrors.
// Level l depends on the rounded u8 pixels of level l-1 (the reference's 7-deep chain), hence one launch per
// level; the first launch builds level 0 (copy) AND level 1 (resized straight from the caller's image).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

__device__ __forceinline__ int reflect101(int p, int n) {
    // BORDER_REFLECT_101 for |overshoot| < n (the border is 19 px, every level is wider)
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

constexpr int kPyrRows = 8;      // destination rows per thread
constexpr int kTileCols = 64;    // dword columns per workgroup tile (256 pixels)
constexpr int kTileRowGroups = 4;
constexpr int kTileRows = kPyrRows * kTileRowGroups;   // 32 destination rows per workgroup tile

static __host__ __device__ inline int rowDwords(const LevelGeom& g) { return (kPadL - kEdge + g.w + 2 * kEdge + 3) / 4; }

struct SrcView {            // where a level's source pixels live
    const uint8_t* p;       // interior pixel (0,0) of frame 0
    long long frame;        // bytes between frames
    int stride;             // bytes between rows
    int readableCols;       // bytes of a row that may be read starting at pixel 0 (dword loads may overshoot the footprint by 3)
    int aligned;            // p, stride and frame are multiples of 4
};

// cv::resize(INTER_LINEAR) 8u: horizontal pass in 11-bit fixed point, vertical pass with the (x>>4, >>16, +2>>2)
// rounding; all products fit 24-bit multiplies.
__device__ __forceinline__ void resizeTile(const SrcView& sv, const LevelGeom& d, const ResizeX* __restrict__ xt,
                                           const ResizeX* __restrict__ yt, const TileFoot ft, uint8_t* __restrict__ pyr,
                                           int tileX, int tileY, int f, uint8_t* tile, int ldsStride) {
    const int tid = threadIdx.x, col = tid & (kTileCols - 1), rgrp = tid / kTileCols;
    const int nd = rowDwords(d), wB = d.w + 2 * kEdge;
    const int dw = tileX * kTileCols + col;
    const bool valid = dw < nd;
    const int bc0 = 4 * (valid ? dw : nd - 1);
    const int fx0 = ft.fx0, fy0 = ft.fy0, nDw = ft.nDw, nRows = ft.nRows;
    const uint8_t* sp = sv.p + (long long)f * sv.frame;

    // ---- stage the footprint: 128 dword columns x 2 rows per step ----
    {
        const int c = tid & 127;
        int srcOff = (fy0 + (tid >> 7)) * sv.stride + fx0 + 4 * c;
        int ldsOff = (tid >> 7) * ldsStride + 4 * c;
        for (int r = tid >> 7; r < nRows; r += 2, srcOff += 2 * sv.stride, ldsOff += 2 * ldsStride) {
            for (int cc = c, so = srcOff, lo = ldsOff; cc < nDw; cc += 128, so += 512, lo += 512) {
                unsigned w;
                if (sv.aligned && fx0 + 4 * cc + 3 < sv.readableCols) {
                    w = *(const unsigned*)(sp + so);
                } else {
                    w = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (fx0 + 4 * cc + b < sv.readableCols) w |= (unsigned)sp[so + b] << (8 * b);
                }
                *(unsigned*)(tile + lo) = w;
            }
        }
    }
    // ---- this thread's 4 columns: LDS byte offsets of the two taps and their 11-bit weights ----
    int c0[4], c1[4], a0[4], a1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int bx = bc0 + j - (kPadL - kEdge);               // bordered x of this byte
        bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);    // bytes of the dword outside the bordered row are padding
        const ResizeX cx = xt[reflect101(bx - kEdge, d.w)];
        c0[j] = cx.sx0 - fx0; c1[j] = cx.sx1 - fx0; a0[j] = cx.a0; a1[j] = cx.a1;
#ifdef ORBX_RESIZE_EXP   // diagnostic: conflict-free tap addresses (wrong pixels, same instruction count)
        c0[j] = 4 * col + j; c1[j] = 4 * col + j;
#endif
    }
    const int by0 = tileY * kTileRows + rgrp * kPyrRows;
    // the 32 rows' vertical coefficients go through LDS (a runtime-indexed register array would live in scratch)
    __shared__ ResizeX ycoef[kTileRows];
    if (tid < kTileRows) ycoef[tid] = yt[reflect101(min(tileY * kTileRows + tid, d.pyrRows - 1) - kEdge, d.h)];
    __syncthreads();
    const ResizeX* cy = ycoef + rgrp * kPyrRows;
    uint8_t* dst = pyr + d.pyrOff + (long long)f * d.pyrFrameBytes + (long long)by0 * d.pyrStride + bc0;
    // two rows per trip: enough LDS reads in flight without holding all 128 taps of the thread in registers
#pragma unroll 2
    for (int r = 0; r < kPyrRows; r++) {
        const uint8_t* r0 = tile + __mul24(cy[r].sx0 - fy0, ldsStride);
        const uint8_t* r1 = tile + __mul24(cy[r].sx1 - fy0, ldsStride);
        const int b0 = cy[r].a0, b1 = cy[r].a1;
        unsigned o = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int h0 = __mul24(r0[c0[j]], a0[j]) + __mul24(r0[c1[j]], a1[j]);
            const int h1 = __mul24(r1[c0[j]], a0[j]) + __mul24(r1[c1[j]], a1[j]);
            const int v = ((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2;
            o |= (unsigned)v << (8 * j);
        }
        if (valid && by0 + r < d.pyrRows) *(unsigned*)(dst + r * d.pyrStride) = o;
    }
}

// bordered level 0 = the caller's image with a 19-px REFLECT_101 frame (:1213-1215)
__device__ __forceinline__ void copyTile(const SrcView& sv, const LevelGeom& g0, uint8_t* __restrict__ pyr, int tileX,
                                         int tileY, int f) {
    const int tid = threadIdx.x, col = tid & (kTileCols - 1), rgrp = tid / kTileCols;
    const int dw = tileX * kTileCols + col;
    if (dw >= rowDwords(g0)) return;
    const int wB = g0.w + 2 * kEdge, bc0 = 4 * dw;
    const int x0 = bc0 - kPadL;                           // interior x of the dword's first byte (a multiple of 4)
    const bool inner = sv.aligned && x0 >= 0 && x0 + 3 < g0.w;   // no reflection inside this dword
    int sx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int bx = bc0 + j - (kPadL - kEdge);
        bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
        sx[j] = reflect101(bx - kEdge, g0.w);
    }
    const uint8_t* sp = sv.p + (long long)f * sv.frame;
    const int by0 = tileY * kTileRows + rgrp * kPyrRows;
    unsigned out[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const uint8_t* row = sp + reflect101(min(by0 + r, g0.pyrRows - 1) - kEdge, g0.h) * sv.stride;
        if (inner) out[r] = *(const unsigned*)(row + x0);
        else out[r] = (unsigned)row[sx[0]] | ((unsigned)row[sx[1]] << 8) | ((unsigned)row[sx[2]] << 16) | ((unsigned)row[sx[3]] << 24);
    }
    uint8_t* dst = pyr + g0.pyrOff + (long long)f * g0.pyrFrameBytes + (long long)by0 * g0.pyrStride + bc0;
#pragma unroll
    for (int r = 0; r < kPyrRows; r++)
        if (by0 + r < g0.pyrRows) *(unsigned*)(dst + r * g0.pyrStride) = out[r];
}

// grid (nTiles0 + nTiles1, B): the first tiles copy the caller's image into bordered level 0, the rest build
// bordered level 1 from the caller's image (== level 0's interior).
__global__ __launch_bounds__(256) void k_pyr_first(SrcView img, LevelGeom g0, LevelGeom g1, int tilesX0, int nTiles0,
                                                    int tilesX1, const ResizeX* __restrict__ xt, const ResizeX* __restrict__ yt,
                                                    const TileFoot* __restrict__ foot, uint8_t* __restrict__ pyr, int ldsStride) {
    extern __shared__ __align__(16) uint8_t tile[];
    int t = blockIdx.x;
    if (t < nTiles0) {
        const int tileY = t / tilesX0;
        copyTile(img, g0, pyr, t - tileY * tilesX0, tileY, blockIdx.y);
    } else {
        t -= nTiles0;
        const int tileY = t / tilesX1;
        resizeTile(img, g1, xt, yt, foot[t], pyr, t - tileY * tilesX1, tileY, blockIdx.y, tile, ldsStride);
    }
}

// grid (tilesX*tilesY, B): level d from level s of the pyramid.
__global__ __launch_bounds__(256) void k_resize(LevelGeom s, LevelGeom d, int tilesX, const ResizeX* __restrict__ xt,
                                                 const ResizeX* __restrict__ yt, const TileFoot* __restrict__ foot,
                                                 uint8_t* __restrict__ pyr, int ldsStride) {
    extern __shared__ __align__(16) uint8_t tile[];
    SrcView sv;
    sv.p = pyr + s.pyrOff + (long long)kEdge * s.pyrStride + kPadL;
    sv.stride = s.pyrStride; sv.frame = s.pyrFrameBytes; sv.readableCols = s.w + kEdge; sv.aligned = 1;
    const int tileY = blockIdx.x / tilesX;
    resizeTile(sv, d, xt, yt, foot[blockIdx.x], pyr, blockIdx.x - tileY * tilesX, tileY, blockIdx.y, tile, ldsStride);
}

void launchPyrFirst(hipStream_t st, const uint8_t* img, long long stride, long long frameStride, const LevelGeom& g0,
                    const LevelGeom* g1, int tilesX0, int tilesY0, int tilesX1, int tilesY1, const ResizeX* xt,
                    const ResizeX* yt, const TileFoot* foot, uint8_t* pyr, int ldsStride, int ldsRows, int B) {
    SrcView sv;
    sv.p = img; sv.stride = (int)stride; sv.frame = frameStride; sv.readableCols = g0.w;
    sv.aligned = (((uintptr_t)img | (uintptr_t)stride | (uintptr_t)frameStride) & 3) == 0;
    const int n0 = tilesX0 * tilesY0, n1 = g1 ? tilesX1 * tilesY1 : 0;
    hipLaunchKernelGGL(k_pyr_first, dim3(n0 + n1, B), dim3(256), (size_t)ldsStride * ldsRows, st, sv, g0, g1 ? *g1 : g0,
                       tilesX0, n0, g1 ? tilesX1 : 1, xt, yt, foot, pyr, ldsStride);
}
void launchResize(hipStream_t st, const LevelGeom& s, const LevelGeom& d, int tilesX, int tilesY, const ResizeX* xt,
                  const ResizeX* yt, const TileFoot* foot, uint8_t* pyr, int ldsStride, int ldsRows, int B) {
    hipLaunchKernelGGL(k_resize, dim3(tilesX * tilesY, B), dim3(256), (size_t)ldsStride * ldsRows, st, s, d, tilesX, xt, yt,
                       foot, pyr, ldsStride);
}

}  // namespace orbx
