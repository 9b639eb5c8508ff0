// k_pyramid.hip — image pyramid: bordered level-0 copy and the fixed-point bilinear resize chain
// (reference ORBextractor.cc:1164-1219; cv::resize / copyMakeBorder semantics: SURVEY.md A.1, A.4).
//
// Two forms, chosen by the host (orbx_api.cpp: enqueueBatch), bit-identical results:
//  * k_pyr_cols (the default): ONE launch; a workgroup takes a region of the image through EVERY level in LDS (end of this file);
//  * one launch per level (large batches of frames above half a megapixel, and geometries whose column quads do not fit the packed
//    step: scale factors above 2): k_pyr_first builds level 0 (bordered copy) and level 1 (resized straight from the caller's image),
//    k_resize level l from level l - 1.  One workgroup = one 256 x 32 tile of a bordered destination level; a thread owns one
//    aligned dword column (4 pixels) over 8 rows, so stores are coalesced 256-B wave stores.  The tile's source footprint — a
//    rectangle the host derives from the coefficient tables — is staged in LDS with coalesced dword loads (12 in flight per thread);
//    a thread then reads the three dwords that hold the taps of its four pixels and cuts the tap pairs out with v_alignbyte / v_perm
//    (packed form), or reads single bytes (any scale factor).  copyMakeBorder(REFLECT_101) is fused: a border byte recomputes the
//    interior pixel it mirrors.
// Level l depends on the rounded u8 pixels of level l-1 (the reference's 7-deep chain) in both forms.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

__device__ __forceinline__ int reflect101(int p, int n) {
    // BORDER_REFLECT_101 for |overshoot| < n (the border is 19 px, every level is wider)
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

constexpr int kPyrRows = kResizeTileRows / 4;   // destination rows per thread
constexpr int kTileCols = 64;    // dword columns per workgroup tile (256 pixels)
constexpr int kTileRowGroups = 4;
constexpr int kTileRows = kPyrRows * kTileRowGroups;   // destination rows per workgroup tile

static __host__ __device__ inline int rowDwords(const LevelGeom& g) { return (kPadL - kEdge + g.w + 2 * kEdge + 3) / 4; }

struct SrcView {            // where a level's source pixels live
    const uint8_t* p;       // interior pixel (0,0) of frame 0
    long long frame;        // bytes between frames
    int stride;             // bytes between rows
    int readableCols;       // bytes of a row that may be read starting at pixel 0 (dword loads may overshoot the footprint by 3)
    int aligned;            // p, stride and frame are multiples of 4
};

__device__ __forceinline__ unsigned mulHi24(unsigned a, unsigned b) {     // (a[23:0] * b[23:0]) >> 32
    unsigned r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pkLshr2(unsigned a) {                  // both u16 halves >> 2
    unsigned r;
    // the shift count is per half too: an inline constant 2 would shift only the low half
    asm("v_pk_lshrrev_b16 %0, %2, %1" : "=v"(r) : "v"(a), "s"(0x00020002u));
    return r;
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// cv::resize(INTER_LINEAR) 8u: horizontal pass in 11-bit fixed point, vertical pass with the (x>>4, >>16, +2>>2)
// rounding; all products fit 24-bit multiplies.
// PACKED (the host checked that the 8 taps of every dword column lie inside 8 consecutive source bytes): a source
// row costs three LDS dword reads and two v_alignbyte for the thread's four pixels, then per pixel one v_perm
// (the tap pair as two u16) and one v_dot2_u32_u16 against the packed weight pair; the vertical pass uses
// (b << 12) * (h & ~15) >> 32 == (b * (h >> 4)) >> 16 in one v_mul_hi_u32_u24.
template <bool PACKED>
__device__ __forceinline__ void resizeTile(const SrcView& sv, const LevelGeom& d, const ResizeX* __restrict__ xt, const QuadRec* __restrict__ xq,
                                           const ResizeX* __restrict__ yt, const TileFoot ft, uint8_t* __restrict__ pyr,
                                           int tileX, int tileY, int f, uint8_t* tile, int ldsStride) {
    const int tid = threadIdx.x, col = tid & (kTileCols - 1), rgrp = tid / kTileCols;
    const int nd = rowDwords(d), wB = d.w + 2 * kEdge;
    const int dw = tileX * kTileCols + col;
    const bool valid = dw < nd;
    const int bc0 = 4 * (valid ? dw : nd - 1);
    const int fx0 = ft.fx0, fy0 = ft.fy0, nDw = ft.nDw, nRows = ft.nRows;
    const uint8_t* sp = sv.p + (long long)f * sv.frame;

    // ---- this thread's 4 columns: LDS byte offsets of the two taps and their 11-bit weights (loads issued first,
    //      so their latency overlaps the staging loads') ----
    int c0[4], c1[4], a0[4], a1[4];
    uint4 q0{}, q1{};
    int qlo = 0;
    if constexpr (PACKED) {      // the host's record of this dword column (orbx_geometry.hpp: xq): selectors, weight pairs, the window's first source byte
        const uint4* qr = (const uint4*)(xq + (valid ? dw : nd - 1));
        q0 = qr[0]; q1 = qr[1];
        qlo = ((const int*)qr)[9] - fx0;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int bx = bc0 + j - (kPadL - kEdge);               // bordered x of this byte
            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);    // bytes of the dword outside the bordered row are padding
            const ResizeX cx = xt[reflect101(bx - kEdge, d.w)];
            c0[j] = cx.sx0 - fx0; c1[j] = cx.sx1 - fx0; a0[j] = cx.a0; a1[j] = cx.a1;
        }
    }
    // the 32 rows' vertical coefficients go through LDS (a runtime-indexed register array would live in scratch), as bank records (RowRec)
    __shared__ __align__(16) RowRec ycoef[kTileRows];
    RowRec myY{};
    if (tid < kTileRows) myY = makeRowRec(yt[reflect101(min(tileY * kTileRows + tid, d.pyrRows - 1) - kEdge, d.h)]);

    // ---- stage the footprint: 128 dword columns x 2 rows per step, kStageRows loads in flight per thread (a
    //      load -> store loop would pay one memory latency per row pair) ----
    {
        constexpr int kStageRows = 12;
        const int c = tid & 127, rsub = tid >> 7;
        for (int cc = c; cc < nDw; cc += 128) {           // one trip unless a tile's footprint is wider than 512 bytes
            const bool whole = sv.aligned && fx0 + 4 * cc + 3 < sv.readableCols;
            const int colOff = fx0 + 4 * cc;                 // (negative inside the source level's left border)
            for (int rb = rsub; rb < nRows; rb += 2 * kStageRows) {
                unsigned w[kStageRows];
#pragma unroll
                for (int k = 0; k < kStageRows; k++) {
                    const int r = min(rb + 2 * k, nRows - 1);     // clamped: every lane loads, only valid rows are stored
                    const uint8_t* q = sp + (__mul24(fy0 + r, sv.stride) + colOff);      // 24-bit: full-rate multiply; signed: rows / columns of the border
                    if (whole) {
                        w[k] = *(const unsigned*)q;
                    } else {
                        w[k] = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            if (fx0 + 4 * cc + b < sv.readableCols) w[k] |= (unsigned)q[b] << (8 * b);
                    }
                }
#pragma unroll
                for (int k = 0; k < kStageRows; k++)
                    if (rb + 2 * k < nRows) *(unsigned*)(tile + (rb + 2 * k) * ldsStride + 4 * cc) = w[k];
            }
        }
    }
    if (tid < kTileRows) ycoef[tid] = myY;
    __syncthreads();
    const int by0 = tileY * kTileRows + rgrp * kPyrRows;
    const RowRec* cy = ycoef + rgrp * kPyrRows;
    uint8_t* dst = pyr + d.pyrOff + (long long)f * d.pyrFrameBytes + (long long)by0 * d.pyrStride + bc0;
    if constexpr (PACKED) {
        const int base = qlo & ~3;
        const unsigned sh = (unsigned)(qlo & 3);
        const unsigned sel[4] = {q0.x, q0.y, q0.z, q0.w};
        const u16x2 wt[4] = {__builtin_bit_cast(u16x2, q1.x), __builtin_bit_cast(u16x2, q1.y), __builtin_bit_cast(u16x2, q1.z), __builtin_bit_cast(u16x2, q1.w)};
        // Horizontal pass of ONE source row for the thread's four pixels (masked for the vertical pass's >> 4).  Consecutive destination
        // rows share source rows (at scale 1.2 eight destination rows use ten distinct source rows, not sixteen), and the rows a
        // destination row uses are the same for the whole wave (a wave = one row group), so the choice "reuse / compute" is a scalar
        // branch: the horizontal pass runs once per distinct source row.
        auto hrow = [&](int srow, unsigned (&h)[4]) {
            const unsigned* rp = (const unsigned*)(tile + __mul24(srow - fy0, ldsStride) + base);
            const unsigned p0 = rp[0], p1 = rp[1], p2 = rp[2];
            const unsigned P0 = __builtin_amdgcn_alignbyte(p1, p0, sh), P1 = __builtin_amdgcn_alignbyte(p2, p1, sh);
#pragma unroll
            for (int j = 0; j < 4; j++)
                h[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(P1, P0, sel[j])), wt[j], 0u, false) & ~15u;
        };
        // two banks of horizontal-pass results (RowRec: the host-side rule that deals a row's two source rows to them, weights pre-shifted)
        unsigned HA[4] = {0, 0, 0, 0}, HB[4] = {0, 0, 0, 0};
        int haveA = -(1 << 20), haveB = -(1 << 20);
#pragma unroll
        for (int r = 0; r < kPyrRows; r++) {
            const RowRec rr = cy[r];
            const int sA = __builtin_amdgcn_readfirstlane(rr.sA), sB = __builtin_amdgcn_readfirstlane(rr.sB);
            if (sA != haveA) { hrow(sA, HA); haveA = sA; }
            if (sB != haveB) { hrow(sB, HB); haveB = sB; }
            unsigned t[4];
#pragma unroll
            for (int j = 0; j < 4; j++) t[j] = mulHi24(rr.bA, HA[j]) + mulHi24(rr.bB, HB[j]) + 2u;
            const unsigned u01 = pkLshr2(t[0] | (t[1] << 16)), u23 = pkLshr2(t[2] | (t[3] << 16));
            if (valid && by0 + r < d.pyrRows) *(unsigned*)(dst + r * d.pyrStride) = __builtin_amdgcn_perm(u23, u01, 0x06040200u);
        }
    } else {
    // two rows per trip: enough LDS reads in flight without holding all 128 taps of the thread in registers
#pragma unroll 2
    for (int r = 0; r < kPyrRows; r++) {
        const uint8_t* r0 = tile + __mul24(cy[r].sA - fy0, ldsStride);      // (bank A / bank B: either order is the same sum)
        const uint8_t* r1 = tile + __mul24(cy[r].sB - fy0, ldsStride);
        const int b0 = (int)(cy[r].bA >> 12), b1 = (int)(cy[r].bB >> 12);
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
        const uint8_t* row = sp + __mul24(reflect101(min(by0 + r, g0.pyrRows - 1) - kEdge, g0.h), sv.stride);
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
template <bool PACKED>
__global__ __launch_bounds__(256) void k_pyr_first(SrcView img, LevelGeom g0, LevelGeom g1, int tilesX0, int nTiles0,
                                                    int tilesX1, const ResizeX* __restrict__ xt, const QuadRec* __restrict__ xq, const ResizeX* __restrict__ yt,
                                                    const TileFoot* __restrict__ foot, uint8_t* __restrict__ pyr, int ldsStride, int f0, int nFrames) {
    extern __shared__ __align__(16) uint8_t tile[];
    int t, fr;
    if (!xcdChunkFrame(nFrames, t, fr)) return;                      // all tiles of a frame on one XCD: the copy tile and the resize tiles that read the same source rows share its L2
    if (t < nTiles0) {
        const int tileY = t / tilesX0;
        copyTile(img, g0, pyr, t - tileY * tilesX0, tileY, f0 + fr);
    } else {
        t -= nTiles0;
        const int tileY = t / tilesX1;
        resizeTile<PACKED>(img, g1, xt, xq, yt, foot[t], pyr, t - tileY * tilesX1, tileY, f0 + fr, tile, ldsStride);
    }
}

// grid (tilesX*tilesY, B): level d from level s of the pyramid.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_resize(LevelGeom s, LevelGeom d, int tilesX, const ResizeX* __restrict__ xt, const QuadRec* __restrict__ xq,
                                                 const ResizeX* __restrict__ yt, const TileFoot* __restrict__ foot,
                                                 uint8_t* __restrict__ pyr, int ldsStride, int f0, int nFrames) {
    extern __shared__ __align__(16) uint8_t tile[];
    SrcView sv;
    sv.p = pyr + s.pyrOff + (long long)kEdge * s.pyrStride + kPadL;
    sv.stride = s.pyrStride; sv.frame = s.pyrFrameBytes; sv.readableCols = s.w + kEdge; sv.aligned = 1;
    int t, fr;
    if (!xcdChunkFrame(nFrames, t, fr)) return;   // neighbouring tiles of a frame (overlapping source footprints) on one XCD
    const int tileY = t / tilesX;
    resizeTile<PACKED>(sv, d, xt, xq, yt, foot[t], pyr, t - tileY * tilesX, tileY, f0 + fr, tile, ldsStride);
}

// ---- stamped builds (-DORBX_CHAIN_STAMPS, tools/cols_stamps.py): s_memrealtime ticks of one region's stages and every region's span ----
#ifdef ORBX_CHAIN_STAMPS
__device__ unsigned long long g_chainStamps[32];
#ifndef ORBX_CHAIN_STAMP_T
#define ORBX_CHAIN_STAMP_T 0
#endif
#define CSTAMP(i) do { if (tid == 0 && t == ORBX_CHAIN_STAMP_T) g_chainStamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int orbx_debug_chain_stamps(unsigned long long* out32) { return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_chainStamps), sizeof(unsigned long long) * 32); }
// every region's (start, end, 0)
__device__ unsigned long long g_chainSpans[3 * 2048];
extern "C" int orbx_debug_chain_spans(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chainSpans), sizeof(g_chainSpans)); }
#define CSPAN_BEGIN do { if (tid == 0 && t < 2048) g_chainSpans[3 * t] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CSPAN_END(lvl) do { if (tid == 0 && t < 2048) { g_chainSpans[3 * t + 1] = __builtin_amdgcn_s_memrealtime(); g_chainSpans[3 * t + 2] = (unsigned long long)(lvl); } } while (0)
#else
#define CSTAMP(i) do {} while (0)
#define CSPAN_BEGIN do {} while (0)
#define CSPAN_END(lvl) do {} while (0)
#endif

// One level step of a region (k_pyr_cols): the rectangle rd of level j (LDS, row stride ds) resized from the rectangle rs of level j - 1 (LDS,
// row stride ss); qrs = one host-made QuadRec per four adjacent columns of rd (tap window, v_perm selectors, weight pairs: orbx_geometry.hpp),
// cys = the y records of rd's rows.  (The host checked that the 8 taps of ANY four adjacent columns lie inside 8 consecutive source bytes.)
// A thread owns four adjacent columns over a block of consecutive rows, so that the horizontal pass of a source row (three LDS dwords, two
// v_alignbyte, per pixel one v_perm + one v_dot2) is shared by the destination rows that use it.  Threads are dealt quad-major (thread = row
// block * quads + quad): every step keeps most of the T threads busy, each with few rows.
template <int T>
__device__ __forceinline__ void chainStep(const uint8_t* S, uint8_t* D, const ChainRegion rs, const ChainRegion rd, const ChainDeal dl, const ResizeX* qrs, const ResizeX* cys,
                                          const int ss, const int ds, const int tid) {
    static_assert(T == 256 || T == 512, "PyrColumn::deal holds the dealing for 256 and 512 deriving threads");
    const int nq = (rd.w + 3) >> 2, nb = dl.nb, blk = (int)(((unsigned)tid * dl.recip) >> 20), x4 = 4 * (tid - blk * nq);      // (orbx_device.hpp: ChainDeal)
    const int per = dl.per, yb = blk * per, ye = min(yb + per, (int)rd.h);
    if (blk < nb && yb < ye) {
        u16x2 wt[4];
        unsigned sel[4];
        const uint4* qr = (const uint4*)((const QuadRec*)qrs + (x4 >> 2));      // three 16-byte LDS reads
        const uint4 q0 = qr[0], q1 = qr[1];
        const unsigned bs = ((const unsigned*)qr)[8];
        sel[0] = q0.x; sel[1] = q0.y; sel[2] = q0.z; sel[3] = q0.w;
        wt[0] = __builtin_bit_cast(u16x2, q1.x); wt[1] = __builtin_bit_cast(u16x2, q1.y); wt[2] = __builtin_bit_cast(u16x2, q1.z); wt[3] = __builtin_bit_cast(u16x2, q1.w);
        const int base = (int)(bs & 0xffffu);
        const unsigned sh = bs >> 16;
        auto hrow = [&](int srow, unsigned (&h)[4]) {
            const unsigned* rp = (const unsigned*)(S + __mul24(srow - rs.y0, ss) + base);
            const unsigned p0 = rp[0], p1 = rp[1], p2 = rp[2];      // (up to 11 bytes past the last tap: the next row, or the buffers' tail padding)
            const unsigned P0 = __builtin_amdgcn_alignbyte(p1, p0, sh), P1 = __builtin_amdgcn_alignbyte(p2, p1, sh);
#pragma unroll
            for (int k = 0; k < 4; k++)
                h[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(P1, P0, sel[k])), wt[k], 0u, false) & ~15u;
        };
        // two banks of horizontal-pass results; the row records say which source row each bank must hold and the weight it gets (RowRec)
        unsigned HA[4] = {0, 0, 0, 0}, HB[4] = {0, 0, 0, 0};
        int haveA = -(1 << 20), haveB = -(1 << 20);
        const RowRec* rrs = (const RowRec*)cys;
        for (int y = yb; y < ye; y++) {
            const RowRec rr = rrs[y];                      // (per-lane: the lanes of a wave sit in different row blocks)
            if (rr.sA != haveA) { hrow(rr.sA, HA); haveA = rr.sA; }
            if (rr.sB != haveB) { hrow(rr.sB, HB); haveB = rr.sB; }
            unsigned t[4];
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = mulHi24(rr.bA, HA[k]) + mulHi24(rr.bB, HB[k]) + 2u;
            const unsigned u01 = pkLshr2(t[0] | (t[1] << 16)), u23 = pkLshr2(t[2] | (t[3] << 16));
            *(unsigned*)(D + y * ds + x4) = __builtin_amdgcn_perm(u23, u01, 0x06040200u);      // (columns past the region's width are padding of the 4-aligned stride)
        }
    }
}

// ---- region-major pyramid (orbx_device.hpp: PyrColumn): one workgroup = one region of the image through EVERY level.  The region's rectangle
//      of the caller's image is loaded once; level j + 1's rectangle is resized from level j's in LDS (chainStep, the reference's chain of
//      rounded u8 levels, ORBextractor.cc:1164-1219) while the bordered bytes the region owns of level j — interior dwords straight out of the
//      LDS rectangle, border bytes mirrored (copyMakeBorder REFLECT_101, :1213-1215) — go to HBM.  Nothing is read back from HBM. ----
// A level = two jobs that only READ its rectangle: resizing the next level's rectangle out of it (the chain every later level waits for) and
// writing the owned bordered bytes to HBM.  The first TD of the T threads do the former, the rest the latter, side by side (one after the
// other in every thread, a 40-px region's level took 1.0 us, of which 0.36 the write).  TD == T: every thread does both (more frames than
// the chip has room for at once: no thread should idle).
template <bool PACKED, int T, int TD>
__global__ __launch_bounds__(T) void k_pyr_cols(SrcView img, const PyrColumn* __restrict__ cols, const ColLevels* __restrict__ lvp, int nlevels,
                                                            const ResizeX* __restrict__ colCoef, int coefSlot,
                                                            uint8_t* __restrict__ pyr, int bufEvenBytes, int f0, int nFrames) {
    static_assert(TD <= T && TD % 64 == 0 && (T - TD) % 64 == 0, "roles are whole waves");
    extern __shared__ __align__(16) uint8_t lds[];
    __shared__ __align__(16) ResizeX coef[kChainCoefMax];
    // what writing a level needs, fetched with the up-front loads: read per level from memory, every level would start with an L2 round trip
    struct LevelOut { ColOwn own; int w, h, stride; long long off; };
    __shared__ LevelOut outOf[kMaxLevels];
    int t, fr;
    if (!xcdChunkFrame(nFrames, t, fr)) return;
    const int f = f0 + fr, tid = threadIdx.x;
    const PyrColumn& pc = cols[t];
    const ColLevels& lv = *lvp;
    const int top = nlevels - 1;
    LevelOut myOut{};
    if (tid < nlevels) myOut = LevelOut{pc.own[tid], lv.w[tid], lv.h[tid], lv.pyrStride[tid], lv.pyrOff[tid] + (long long)f * lv.pyrFrameBytes[tid]};
    CSTAMP(0);
    CSPAN_BEGIN;
    auto bufOf = [&](int j) { return lds + ((j & 1) ? bufEvenBytes : 0); };      // level j's rectangle lives in buffer j & 1 (an LDS offset, never a generic pointer)
    // the rectangles of the first nine levels are read with STATIC indices (a run-time index into the by-value record would go through scratch): one batch of wave-uniform loads
    constexpr int kStatic = 8;
    ChainRegion rj[kStatic + 1];
#pragma unroll
    for (int j = 0; j <= kStatic; j++) rj[j] = pc.region[j];
    {
        // ---- every load of the workgroup is issued up front: the image rectangle (aligned dwords) and the coefficient records of ALL steps
        //      (x records of level j's rectangle, then its y records, level after level) ----
        const ChainRegion r = rj[0];
        const int nDw = r.w >> 2, total = nDw * r.h;
        const uint8_t* src = img.p + (long long)f * img.frame + (long long)r.y0 * img.stride + r.x0;
        unsigned* dst = (unsigned*)bufOf(0);
        constexpr int kLoads = (kChainMaxH0 * kChainMaxW / 4 + T - 1) / T;
        const float inv = __frcp_rn((float)nDw);
        unsigned w[kLoads];
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            w[i] = 0;
            if (i * T < total) {                                  // workgroup-uniform: a small rectangle issues (and computes the addresses of) few loads
            const int idx = min(tid + i * T, total - 1);          // clamped: every lane loads a valid address
            const int row = (int)(((float)idx + 0.5f) * inv), c = idx - row * nDw;      // exact: idx < 6144, the quotient is >= 0.5 / nDw away from an integer
            const uint8_t* q = src + ((unsigned)__mul24(row, img.stride) + 4u * (unsigned)c);
            const int x = r.x0 + 4 * c;
            if (img.aligned && x >= 0 && x + 3 < img.readableCols) w[i] = *(const unsigned*)q;
            else {                                                          // a row's last dword or an unaligned image: no byte outside the row is read
#pragma unroll
                for (int b = 0; b < 4; b++) w[i] |= (unsigned)q[reflect101(x + b, img.readableCols) - x] << (8 * b);
            }
            }
        }
        constexpr int kPerThread = (kChainCoefMax + T - 1) / T;
        const int nCoef = pc.nCoef;
        const ResizeX* mine = colCoef + (size_t)t * (size_t)coefSlot;
        ResizeX cv[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; k++) cv[k] = tid + k * T < nCoef ? mine[tid + k * T] : ResizeX{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < kPerThread; k++)
            if (tid + k * T < nCoef) coef[tid + k * T] = cv[k];
        if (tid < nlevels) outOf[tid] = myOut;
#pragma unroll
        for (int i = 0; i < kLoads; i++)
            if (tid + i * T < total) dst[tid + i * T] = w[i];
    }
    __syncthreads();
    CSTAMP(1);
    int off = 0;
    // one level: the next level's rectangle first (the critical path), then this level's owned bytes; both only read S
    auto doLevel = [&](const int j, const ChainRegion rs, const ChainRegion rd) {
        const int ss = (rs.w + 3) & ~3;
        const uint8_t* S = bufOf(j);
        constexpr int TW = TD < T ? T - TD : T;      // threads of the writing role
        if (j < top) {
            const ResizeX* cxs = coef + off;                        // the level's quad records, then its y records (PyrColumn's layout)
            const int nqUnits = 6 * ((rd.w + 3) >> 2);
            if (tid < TD) chainStep<TD>(S, bufOf(j + 1), rs, rd, pc.deal[TD == 512][j + 1 < kMaxLevels ? j + 1 : j], cxs, cxs + nqUnits, ss, (rd.w + 3) & ~3, tid);      // wave-uniform
            off += nqUnits + 2 * rd.h;
        }
        // (the last level has nothing to derive: every thread writes)
        const bool everyone = TD == T || j == top;
        const int wtid = everyone ? tid : tid - TD, wstep = everyone ? T : TW;
        if (wtid >= 0) {
        const LevelOut lo = outOf[j];
        const ColOwn own = lo.own;
        const int w = lo.w, h = lo.h, wB = w + 2 * kEdge;
        uint8_t* out = pyr + lo.off;
        const int stride = lo.stride, nrows = own.r1 - own.r0;
        // the owned dword columns in two passes, so that no wave runs both bodies: first the columns that lie inside the level's rows (aligned
        // dwords of the LDS rectangle: nearly everything; an inner region owns nothing else), then the frame's columns of an outer region
        // (mirrored / clamped byte by byte).  In one pass an outer region's waves ran the byte-by-byte body for every trip of the loop — three
        // trips for a corner's level 0, which made the corners the launch's last workgroups.
        const int fa = max((int)own.dw0, kPadL / 4), fb = max(fa, min((int)own.dw1, kPadL / 4 + (w >> 2)));      // [fa, fb): x0 >= 0 and x0 + 3 < w
        if (fb > fa && (fb - fa) * nrows <= 2 * wstep) {             // workgroup-uniform: a small part (a fine cut's region; the deep levels)
            // ... is dealt dword by dword: at most two per thread, no chain of dependent rows
            const int ndw = fb - fa, total = ndw * nrows;
            const float inv = __frcp_rn((float)ndw);
            for (int i = wtid; i < total; i += wstep) {
                // (exact: the quotient's error, ~i / ndw * 2^-22, stays below the 0.5 / ndw that separates it from an integer while total < 2^21)
                const int rr = (int)(((float)i + 0.5f) * inv), dw = fa + (i - rr * ndw), row = own.r0 + rr;
                const uint8_t* srow = S + __mul24(reflect101(row - kEdge, h) - rs.y0, ss) - rs.x0;      // interior pixel x of that row at srow[x]
                *(unsigned*)(out + (long long)row * stride + 4 * dw) = *(const unsigned*)(srow + (4 * dw - kPadL));      // (a multiple of 4, as rs.x0 is)
            }
        } else if (fb > fa) {
            // a thread takes ONE dword column and walks down a block of rows (a row of the rectangle = one coalesced run of stores across
            // the lanes): per dword a row reflection, an LDS read and a store — no index arithmetic (dealt dword by dword through a division it
            // was ~22 instructions per dword, a fifth of the kernel's at 512 frames)
            // (quotients of small integers through the reciprocal: exact while the dividend stays below 2^12 — an integer division is ~40 instructions)
            const int ndw = fb - fa;
            const float rdw = __frcp_rn((float)ndw);
            const int ngrp = max((int)(((float)wstep + 0.5f) * rdw), 1);
            const int per = (int)(((float)(nrows + ngrp - 1) + 0.5f) * __frcp_rn((float)ngrp));
            const int grp = (int)(((float)wtid + 0.5f) * rdw), c = wtid - grp * ndw;
            if (wtid >= 0 && grp < ngrp) {
                const int rA = own.r0 + grp * per, rB = min(rA + per, (int)own.r1);
                const uint8_t* scol = S + (4 * (fa + c) - kPadL - rs.x0);      // the column's first byte in rectangle row 0 (a multiple of 4, as rs.x0 is)
                uint8_t* ocol = out + 4 * (fa + c);
                // bordered rows [0, kEdge) mirror level rows kEdge .. 1, [kEdge, kEdge + h) are rows 0 .. h - 1, the rest mirror h - 2 .. (reflect101):
                // three runs whose source row moves by one rectangle row per bordered row - an LDS read, a store and two pointer steps per dword
                // (with the reflection, the row's product and a 64-bit address worked out per dword it was ~14 instructions)
                auto run = [&](const int r0, const int r1, const int src0, const int sstep) {
                    const uint8_t* sp = scol + __mul24(src0 - rs.y0, ss);
                    uint8_t* op = ocol + (long long)r0 * stride;
                    for (int row = r0; row < r1; row++, op += stride, sp += sstep) *(unsigned*)op = *(const unsigned*)sp;
                };
                const int t1 = min(rB, kEdge), m0 = max(rA, kEdge), m1 = min(rB, kEdge + h), b0 = max(rA, kEdge + h);
                if (rA < t1) run(rA, t1, kEdge - rA, -ss);
                if (m0 < m1) run(m0, m1, m0 - kEdge, ss);
                if (b0 < rB) run(b0, rB, 2 * (h - 1) - (b0 - kEdge), -ss);
            }
            // (columns beyond ngrp * ndw threads: a rectangle wider than the role has threads — the regions are at most 64 dwords wide)
        }
        const int nL = fa - own.dw0, nO = nL + (own.dw1 - fb);      // frame columns left / in all
        if (nO > 0) {                                                // workgroup-uniform
            const int total = nO * nrows;
            const float inv = __frcp_rn((float)nO);
            for (int i = wtid; i < total; i += wstep) {
                const int rr = (int)(((float)i + 0.5f) * inv), jj = i - rr * nO, dw = jj < nL ? own.dw0 + jj : fb + (jj - nL), row = own.r0 + rr;
                const uint8_t* srow = S + __mul24(reflect101(row - kEdge, h) - rs.y0, ss) - rs.x0;
                const int bc0 = 4 * dw;
                unsigned o = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int bx = bc0 + k - (kPadL - kEdge);
                    bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);      // bytes of a dword outside the bordered row are padding
                    o |= (unsigned)srow[reflect101(bx - kEdge, w)] << (8 * k);
                }
                *(unsigned*)(out + (long long)row * stride + bc0) = o;
            }
        }
        }
        __syncthreads();
        CSTAMP(j + 2);
    };
#pragma unroll
    for (int j = 0; j < kStatic; j++)
        if (j <= top) doLevel(j, rj[j], rj[j + 1]);            // workgroup-uniform
    for (int j = kStatic; j <= top; j++) doLevel(j, pc.region[j], pc.region[j + 1 < kMaxLevels ? j + 1 : j]);      // (more than eight levels)
    CSPAN_END(0);
}

// shape: 1 = 512 threads (256 derive + 256 write: every launch that fills the chip), 4 = 768 (256 + 512) and 6 = 1024 (512 + 512): while every
// workgroup has a CU to itself (orbx_api.cpp: colsShape).  Rounds 3-5 carried four more shapes (every thread both jobs at 256 / 512 threads,
// 768 = 512 + 256, 1024 = 256 + 768) that never won at any batch size (docs/history/r04.md); removed in round 6.
void launchPyrCols(hipStream_t st, const uint8_t* img, long long stride, long long frameStride, int imgW, const PyrColumn* cols, int nCols,
                   const ColLevels* lv, int nlevels, const ResizeX* colCoef, int coefSlot, uint8_t* pyr, int ldsBytes, int bufEvenBytes,
                   bool packed, int shape, int f0, int B) {
    SrcView sv;
    sv.p = img; sv.stride = (int)stride; sv.frame = frameStride; sv.readableCols = imgW;
    sv.aligned = (((uintptr_t)img | (uintptr_t)stride | (uintptr_t)frameStride) & 3) == 0;
#define ORBX_COLS_LAUNCH(P, T, TD) hipLaunchKernelGGL((k_pyr_cols<P, T, TD>), xcdGrid(nCols, B), dim3(T), (size_t)ldsBytes, st, sv, cols, lv, nlevels, colCoef, coefSlot, \
                                                  pyr, bufEvenBytes, f0, B)
    (void)packed;      // (the host only takes this form when the taps of every column quad fit the packed step's 8-byte window)
    if (shape == 6) ORBX_COLS_LAUNCH(true, 1024, 512);
    else if (shape == 4) ORBX_COLS_LAUNCH(true, 768, 256);
    else ORBX_COLS_LAUNCH(true, 512, 256);
#undef ORBX_COLS_LAUNCH
}

void launchPyrFirst(hipStream_t st, const uint8_t* img, long long stride, long long frameStride, const LevelGeom& g0,
                    const LevelGeom* g1, int tilesX0, int tilesY0, int tilesX1, int tilesY1, const ResizeX* xt, const QuadRec* xq,
                    const ResizeX* yt, const TileFoot* foot, uint8_t* pyr, int ldsStride, int ldsRows, bool packed, int f0, int B) {
    SrcView sv;
    sv.p = img; sv.stride = (int)stride; sv.frame = frameStride; sv.readableCols = g0.w;
    sv.aligned = (((uintptr_t)img | (uintptr_t)stride | (uintptr_t)frameStride) & 3) == 0;
    const int n0 = tilesX0 * tilesY0, n1 = g1 ? tilesX1 * tilesY1 : 0;
    // + 16: the packed path reads three dwords from the first tap's dword, i.e. up to 8 bytes past a row's footprint
    if (packed) hipLaunchKernelGGL(k_pyr_first<true>, xcdGrid(n0 + n1, B), dim3(256), (size_t)ldsStride * ldsRows + 16, st, sv, g0,
                                   g1 ? *g1 : g0, tilesX0, n0, g1 ? tilesX1 : 1, xt, xq, yt, foot, pyr, ldsStride, f0, B);
    else hipLaunchKernelGGL(k_pyr_first<false>, xcdGrid(n0 + n1, B), dim3(256), (size_t)ldsStride * ldsRows + 16, st, sv, g0,
                            g1 ? *g1 : g0, tilesX0, n0, g1 ? tilesX1 : 1, xt, xq, yt, foot, pyr, ldsStride, f0, B);
}
void launchResize(hipStream_t st, const LevelGeom& s, const LevelGeom& d, int tilesX, int tilesY, const ResizeX* xt, const QuadRec* xq,
                  const ResizeX* yt, const TileFoot* foot, uint8_t* pyr, int ldsStride, int ldsRows, bool packed, int f0, int B) {
    if (packed) hipLaunchKernelGGL(k_resize<true>, xcdGrid(tilesX * tilesY, B), dim3(256), (size_t)ldsStride * ldsRows + 16, st, s, d,
                                   tilesX, xt, xq, yt, foot, pyr, ldsStride, f0, B);
    else hipLaunchKernelGGL(k_resize<false>, xcdGrid(tilesX * tilesY, B), dim3(256), (size_t)ldsStride * ldsRows + 16, st, s, d,
                            tilesX, xt, xq, yt, foot, pyr, ldsStride, f0, B);
}

}  // namespace orbx
