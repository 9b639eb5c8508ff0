// k_pyramid.hip — image pyramid: bordered level-0 copy and the fixed-point bilinear resize chain
// (reference ORBextractor.cc:1164-1219; cv::resize / copyMakeBorder semantics: SURVEY.md A.1, A.4).
//
// Every thread produces aligned dwords (4 pixels) of a bordered destination level, so stores are coalesced 256-B
// wave stores; the source footprint of a 256 x 32 tile is staged in LDS (global byte gathers run at a quarter
// of the dword rate).  copyMakeBorder(REFLECT_101) is fused: a border byte recomputes the interior pixel it
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

constexpr int kPyrRows = 8;      // destination rows per thread
constexpr int kTileCols = 64;    // dword columns per workgroup tile (256 pixels)
constexpr int kTileRowGroups = 4;
constexpr int kTileRows = kPyrRows * kTileRowGroups;   // 32 destination rows per workgroup tile

static __host__ __device__ inline int rowDwords(const LevelGeom& g) { return (kPadL - kEdge + g.w + 2 * kEdge + 3) / 4; }

struct SrcView {            // where a level's source pixels live
    const uint8_t* p;       // interior pixel (0,0) of frame 0
    long long stride, frame;
    int readableCols;       // bytes of a row that may be read starting at pixel 0 (dword loads may overshoot the footprint by 3)
    int aligned;            // p, stride and frame are multiples of 4
};

// One workgroup = one 256 x 32 tile of a bordered destination level.  The source footprint of the tile (a
// contiguous rectangle: the coefficient tables are monotonic and REFLECT_101 only folds indices back inside) is
// staged in LDS with coalesced dword loads; the 4 taps per pixel are then LDS byte reads.
__device__ __forceinline__ void resizeTile(const SrcView& sv, const LevelGeom& d, const ResizeX* __restrict__ xt,
                                           const ResizeX* __restrict__ yt, uint8_t* __restrict__ pyr, int tileX, int tileY,
                                           int f, uint8_t* tile, int* rng, int ldsStride, int ldsRows) {
    const int tid = threadIdx.x, col = tid & (kTileCols - 1), rgrp = tid / kTileCols;
    const int nd = rowDwords(d), wB = d.w + 2 * kEdge;
    const int dw = tileX * kTileCols + col;
    const bool valid = dw < nd;
    const int bc0 = 4 * (valid ? dw : nd - 1);
    ResizeX cx[4];
    int sxmin = 1 << 30, sxmax = -1, symin = 1 << 30, symax = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int bx = bc0 + j - (kPadL - kEdge);               // bordered x of this byte
        bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);    // bytes of the dword outside the bordered row are padding
        cx[j] = xt[reflect101(bx - kEdge, d.w)];
        sxmin = min(sxmin, (int)cx[j].sx0); sxmax = max(sxmax, (int)cx[j].sx1);
    }
    ResizeX cy[kPyrRows];
    const int by0 = tileY * kTileRows + rgrp * kPyrRows;
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const int by = min(by0 + r, d.pyrRows - 1);
        cy[r] = yt[reflect101(by - kEdge, d.h)];
        symin = min(symin, (int)cy[r].sx0); symax = max(symax, (int)cy[r].sx1);
    }
    if (tid < 4) rng[tid] = (tid & 1) ? -1 : (1 << 30);
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sxmin = min(sxmin, __shfl_xor(sxmin, o)); sxmax = max(sxmax, __shfl_xor(sxmax, o));
        symin = min(symin, __shfl_xor(symin, o)); symax = max(symax, __shfl_xor(symax, o));
    }
    if ((tid & 63) == 0) { atomicMin(&rng[0], sxmin); atomicMax(&rng[1], sxmax); atomicMin(&rng[2], symin); atomicMax(&rng[3], symax); }
    __syncthreads();
    const int fx0 = rng[0] & ~3, fy0 = rng[2];
    const int nDw = (rng[1] - fx0 + 4) >> 2, nRows = rng[3] - fy0 + 1;
    const uint8_t* sp = sv.p + (long long)f * sv.frame;
    uint8_t* dst = pyr + d.pyrOff + (long long)f * d.pyrFrameBytes + bc0;
    const bool staged = nDw * 4 <= ldsStride && nRows <= ldsRows;   // always true for the sizes the host computes
    if (staged) {
        for (int r = tid >> 7; r < nRows; r += 2) {
            const uint8_t* row = sp + (long long)(fy0 + r) * sv.stride + fx0;
            for (int c = tid & 127; c < nDw; c += 128) {
                unsigned w;
                if (sv.aligned && fx0 + 4 * c + 3 < sv.readableCols) {
                    w = *(const unsigned*)(row + 4 * c);
                } else {
                    w = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (fx0 + 4 * c + b < sv.readableCols) w |= (unsigned)row[4 * c + b] << (8 * b);
                }
                *(unsigned*)(tile + r * ldsStride + 4 * c) = w;
            }
        }
    }
    __syncthreads();
    unsigned out[kPyrRows];
    if (staged) {
        // column parts of the LDS addresses and the two 11-bit weights of each of the 4 pixels, fixed for all rows
        // (two independent byte offsets on purpose: a merged 16-bit LDS read at an odd address is slow)
        int co[4], c1[4], a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { co[j] = cx[j].sx0 - fx0; c1[j] = cx[j].sx1 - fx0; a0[j] = cx[j].a0; a1[j] = cx[j].a1; }
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const uint8_t* r0 = tile + (cy[r].sx0 - fy0) * ldsStride;
            const uint8_t* r1 = tile + (cy[r].sx1 - fy0) * ldsStride;
            const int b0 = cy[r].a0, b1 = cy[r].a1;
            unsigned o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int h0 = __mul24(r0[co[j]], a0[j]) + __mul24(r0[c1[j]], a1[j]);
                const int h1 = __mul24(r1[co[j]], a0[j]) + __mul24(r1[c1[j]], a1[j]);
                const int v = ((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2;
                o |= (unsigned)v << (8 * j);
            }
            out[r] = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const uint8_t* r0 = sp + (long long)cy[r].sx0 * sv.stride;
            const uint8_t* r1 = sp + (long long)cy[r].sx1 * sv.stride;
            unsigned o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) o |= bilinear(r0, r1, cx[j], cy[r].a0, cy[r].a1) << (8 * j);
            out[r] = o;
        }
    }
    if (valid) {
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const int by = by0 + r;
            if (by < d.pyrRows) *(unsigned*)(dst + (long long)by * d.pyrStride) = out[r];
        }
    }
}

// grid (tilesX0*tilesY0 + tilesX1*tilesY1, B): the first tiles copy the caller's image into bordered level 0, the
// rest build bordered level 1 from the caller's image (== level 0's interior).
__global__ __launch_bounds__(256) void k_pyr_first(SrcView img, LevelGeom g0, LevelGeom g1, int tilesX0, int nTiles0,
                                                    int tilesX1, const ResizeX* __restrict__ xt, const ResizeX* __restrict__ yt,
                                                    uint8_t* __restrict__ pyr, int ldsStride, int ldsRows) {
    extern __shared__ __align__(16) uint8_t tile[];
    __shared__ int rng[4];
    const int f = blockIdx.y, tid = threadIdx.x;
    int t = blockIdx.x;
    if (t < nTiles0) {
        const int tileY = t / tilesX0, tileX = t - tileY * tilesX0;
        const int col = tid & (kTileCols - 1), rgrp = tid / kTileCols;
        const int dw = tileX * kTileCols + col;
        if (dw >= rowDwords(g0)) return;
        const int wB = g0.w + 2 * kEdge, bc0 = 4 * dw;
        int sx[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int bx = bc0 + j - (kPadL - kEdge);
            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
            sx[j] = reflect101(bx - kEdge, g0.w);
        }
        const uint8_t* sp = img.p + (long long)f * img.frame;
        uint8_t* dst = pyr + g0.pyrOff + (long long)f * g0.pyrFrameBytes + bc0;
        const int by0 = tileY * kTileRows + rgrp * kPyrRows;
        unsigned out[kPyrRows];
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const int by = min(by0 + r, g0.pyrRows - 1);
            const uint8_t* row = sp + (long long)reflect101(by - kEdge, g0.h) * img.stride;
            out[r] = (unsigned)row[sx[0]] | ((unsigned)row[sx[1]] << 8) | ((unsigned)row[sx[2]] << 16) | ((unsigned)row[sx[3]] << 24);
        }
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const int by = by0 + r;
            if (by < g0.pyrRows) *(unsigned*)(dst + (long long)by * g0.pyrStride) = out[r];
        }
    } else {
        t -= nTiles0;
        const int tileY = t / tilesX1, tileX = t - tileY * tilesX1;
        resizeTile(img, g1, xt, yt, pyr, tileX, tileY, f, tile, rng, ldsStride, ldsRows);
    }
}

// grid (tilesX*tilesY, B): level d from level s of the pyramid.
__global__ __launch_bounds__(256) void k_resize(LevelGeom s, LevelGeom d, int tilesX, const ResizeX* __restrict__ xt,
                                                 const ResizeX* __restrict__ yt, uint8_t* __restrict__ pyr, int ldsStride,
                                                 int ldsRows) {
    extern __shared__ __align__(16) uint8_t tile[];
    __shared__ int rng[4];
    SrcView sv;
    sv.p = pyr + s.pyrOff + (long long)kEdge * s.pyrStride + kPadL;
    sv.stride = s.pyrStride; sv.frame = s.pyrFrameBytes; sv.readableCols = s.w + kEdge; sv.aligned = 1;
    const int tileY = blockIdx.x / tilesX, tileX = blockIdx.x - tileY * tilesX;
    resizeTile(sv, d, xt, yt, pyr, tileX, tileY, blockIdx.y, tile, rng, ldsStride, ldsRows);
}

static inline int tilesX(const LevelGeom& g) { return (rowDwords(g) + kTileCols - 1) / kTileCols; }
static inline int tilesY(const LevelGeom& g) { return (g.pyrRows + kTileRows - 1) / kTileRows; }
// LDS footprint of one tile of level d resampled from a source sw x sh: generous bound of the real footprint
static inline void tileLds(int sw, int sh, const LevelGeom& d, int* stride, int* rows) {
    *stride = (int)(4.0 * kTileCols * sw / d.w) + 16;
    *stride = (*stride + 15) / 16 * 16;
    *rows = (int)((double)kTileRows * sh / d.h) + 4;
}

void launchPyrFirst(hipStream_t st, const uint8_t* img, long long stride, long long frameStride, const LevelGeom& g0,
                    const LevelGeom* g1, const ResizeX* xt, const ResizeX* yt, uint8_t* pyr, int B) {
    SrcView sv;
    sv.p = img; sv.stride = stride; sv.frame = frameStride; sv.readableCols = g0.w;
    sv.aligned = (((uintptr_t)img | (uintptr_t)stride | (uintptr_t)frameStride) & 3) == 0;
    const int n0 = tilesX(g0) * tilesY(g0), n1 = g1 ? tilesX(*g1) * tilesY(*g1) : 0;
    int ls = 16, lr = 1;
    if (g1) tileLds(g0.w, g0.h, *g1, &ls, &lr);
    hipLaunchKernelGGL(k_pyr_first, dim3(n0 + n1, B), dim3(256), (size_t)ls * lr, st, sv, g0, g1 ? *g1 : g0, tilesX(g0), n0,
                       g1 ? tilesX(*g1) : 1, xt, yt, pyr, ls, lr);
}
void launchResize(hipStream_t st, const LevelGeom& s, const LevelGeom& d, const ResizeX* xt, const ResizeX* yt,
                  uint8_t* pyr, int B) {
    int ls, lr;
    tileLds(s.w, s.h, d, &ls, &lr);
    hipLaunchKernelGGL(k_resize, dim3(tilesX(d) * tilesY(d), B), dim3(256), (size_t)ls * lr, st, s, d, tilesX(d), xt, yt, pyr,
                       ls, lr);
}

}  // namespace orbx
