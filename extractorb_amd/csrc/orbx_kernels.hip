// orbx_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the ORB extraction hot path.
//
// One batch of B frames runs as a short, fixed launch sequence; every launch covers all frames:
//   k_level0      copy the input into the bordered level-0 buffer            (ORBextractor.cc:1213-1215)
//   k_resize x7   level l from level l-1, border fused                       (ORBextractor.cc:1183-1197)
//   k_blur        7x7 sigma-2 integer Gaussian of every level                (ORBextractor.cc:1126-1127)
//   k_fast        one wave per 30-px cell: FAST-9 score, cell threshold, NMS (ORBextractor.cc:797-864)
//   k_octree      one workgroup per (frame, level): DistributeOctTree        (ORBextractor.cc:544-771)
//   k_describe    one wave per kept keypoint: IC_Angle + rBRIEF + output     (ORBextractor.cc:75-145,1137-1158)
//
// All arithmetic that decides bit-exactness is integer, or float with explicitly separate roundings
// (this file is compiled with -ffp-contract=off and uses __fmul_rn/__fadd_rn where the reference's
// x86-64 build rounds twice).  No MFMA: nothing here is a contraction.
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

// ================================================================================================
// FAST-9/16 per cell.  One wave64 == one cell == one cv::FAST call of the reference (two when the first
// is empty).  S(p) = max(S_dark, S_bright) with S_dark = max over the 16 arcs of min(v - ring) and
// S_bright the mirror image; "corner at threshold t" <=> S > t and the reference's response is S-1
// (SURVEY.md A.3), so one score serves iniThFAST and the minThFAST retry.  NMS is a strict 3x3 maximum
// of S with everything outside the cell interior counted as 0; since a surviving centre has S > t,
// neighbours below t can never suppress it, so NMS is threshold independent.
// ================================================================================================
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }

__device__ __forceinline__ int fastScore(const uint8_t* c, int st) {
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * st];          d[1] = v - c[3 * st + 1];   d[2] = v - c[2 * st + 2];   d[3] = v - c[st + 3];
    d[4] = v - c[3];               d[5] = v - c[-st + 3];      d[6] = v - c[-2 * st + 2];  d[7] = v - c[-3 * st + 1];
    d[8] = v - c[-3 * st];         d[9] = v - c[-3 * st - 1];  d[10] = v - c[-2 * st - 2]; d[11] = v - c[-st - 3];
    d[12] = v - c[-3];             d[13] = v - c[st - 3];      d[14] = v - c[2 * st - 2];  d[15] = v - c[3 * st - 1];
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        hi3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int sDark = -256, sBrightNeg = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int lo9 = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);   // min of d over the arc k..k+8
        const int hi9 = max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);   // max of d over the arc
        sDark = max(sDark, lo9);
        sBrightNeg = min(sBrightNeg, hi9);
    }
    const int s = max(sDark, -sBrightNeg);
    return s < 0 ? 0 : s;   // <= 255
}

constexpr int kFastWaves = 4;

// dynamic LDS per wave: tile[tileRows*tileStride] + score[(maxCh+2)*scoreStride]
__global__ __launch_bounds__(256) void k_fast(const CellDesc* __restrict__ cells, int nCells,
                                               const LevelGeom* __restrict__ lv, int nlevels,
                                               const uint8_t* __restrict__ pyr, int iniTh, int minTh,
                                               uint2* __restrict__ cand, unsigned* __restrict__ candCount,
                                               int tileStride, int tileBytes, int scoreStride, int scoreBytes) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ci = blockIdx.x * kFastWaves + wave, f = blockIdx.y;
    if (ci >= nCells) return;   // wave-uniform; the kernel has no workgroup barrier
    const CellDesc c = cells[ci];
    const LevelGeom g = lv[c.level];
    uint8_t* tile = smem + wave * (tileBytes + scoreBytes);
    uint8_t* score = tile + tileBytes;
    const int roiW = c.roiW, roiH = c.roiH, cw = roiW - 6, ch = roiH - 6;

    const uint8_t* sp = pyr + g.pyrOff + (long long)f * g.pyrFrameBytes + (long long)(kEdge + c.y0) * g.pyrStride +
                        kPadL + c.x0;
    for (int r = 0; r < roiH; r++) {
        if (lane < roiW) tile[r * tileStride + lane] = sp[(long long)r * g.pyrStride + lane];
        if (lane + 64 < roiW) tile[r * tileStride + lane + 64] = sp[(long long)r * g.pyrStride + lane + 64];
    }
    // zero the score tile (its 1-px apron stands for "outside the ROI interior")
    for (int i = lane * 4; i < scoreBytes; i += 256) *(uint32_t*)(score + i) = 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave are done

    const int npix = cw * ch;
    // pass 1: scores
    {
        int x = lane % cw, y = lane / cw;
        for (int p = lane; p < npix; p += 64) {
            const int s = fastScore(tile + (y + 3) * tileStride + x + 3, tileStride);
            score[(y + 1) * scoreStride + x + 1] = (uint8_t)s;
            x += 64;
            while (x >= cw) { x -= cw; y++; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // pass 2: strict local maxima; lane i keeps the ballots of sweep i
    unsigned long long myIni = 0, myMin = 0;
    int nIni = 0, nMin = 0;
    {
        int x = lane % cw, y = lane / cw, it = 0;
        for (int base = 0; base < npix; base += 64, it++) {
            bool lm = false;
            int s = 0;
            if (base + lane < npix) {
                const uint8_t* q = score + (y + 1) * scoreStride + x + 1;
                s = q[0];
                lm = s > q[-1] && s > q[1] && s > q[-scoreStride - 1] && s > q[-scoreStride] &&
                     s > q[-scoreStride + 1] && s > q[scoreStride - 1] && s > q[scoreStride] && s > q[scoreStride + 1];
            }
            const unsigned long long bIni = __ballot(lm && s > iniTh);
            const unsigned long long bMin = __ballot(lm && s > minTh);
            if (lane == it) { myIni = bIni; myMin = bMin; }
            nIni += __popcll(bIni);
            nMin += __popcll(bMin);
            x += 64;
            while (x >= cw) { x -= cw; y++; }
        }
    }
    // the reference retries the cell at minThFAST only when the first call returned nothing (:835-838)
    const bool useIni = nIni > 0;
    const unsigned long long mine = useIni ? myIni : myMin;
    const int total = useIni ? nIni : nMin;
    if (total == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&candCount[f * nlevels + c.level], (unsigned)total);
    base = __builtin_amdgcn_readfirstlane(base);
    uint2* out = cand + g.candOff + (long long)f * g.candCap;
    {
        int x = lane % cw, y = lane / cw, it = 0;
        for (int b0 = 0; b0 < npix; b0 += 64, it++) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)mine, it);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(mine >> 32), it);
            const unsigned long long m = ((unsigned long long)hi << 32) | lo;
            if ((m >> lane) & 1) {
                const int s = score[(y + 1) * scoreStride + x + 1];
                const unsigned before = __popcll(m & ((1ull << lane) - 1));
                const unsigned px = (unsigned)(c.shiftX + x + 3), py = (unsigned)(c.shiftY + y + 3);
                uint2 e;
                e.x = px | (py << 12) | ((unsigned)(s - 1) << 24);        // response = S - 1
                e.y = ((unsigned)c.cellId << 12) | ((unsigned)y << 6) | (unsigned)x;   // reference list order
                const unsigned at = base + before;
                if (at < (unsigned)g.candCap) out[at] = e;
            }
            base += __popcll(m);
            x += 64;
            while (x >= cw) { x -= cw; y++; }
        }
    }
}

// ================================================================================================
// DistributeOctTree.  One 256-thread workgroup per (frame, level).
//
// The reference's std::list is modelled as an array in list order.  A key never needs its position
// inside a node: the final pick per node is max response, first in candidate order on ties
// (ORBextractor.cc:751-767), and DivideNode's partition is stable, so "first" == smallest candidate
// order word.  Each pass therefore only (1) counts keys per child quadrant, (2) decides which nodes
// split and where their children land in the new list, (3) renames every key's node.
//   pass, phase 1 (:611-670): every multi-key node splits, in list order; children are pushed to the
//     front one by one, so the new list is reverse(creation order) followed by the untouched nodes.
//   pass, phase 2 (:681-742): multi-key nodes split in order (size desc, newest first) and the pass
//     stops right after the split that reaches N nodes.  All multi-key nodes were created in the
//     previous pass, where creation order is the reverse of list order, so "newest first" == smallest
//     list position: the sort key is (size desc, position asc).
// ================================================================================================
constexpr int kOctThreads = 256;

struct OctShared {
    unsigned scanTmp[kOctThreads / 64];
    int size, prevSize, phase2, nToExpand, done, nChildren, nKept, breakRank, nCand;
};

// exclusive prefix sum of data[0..n) in place, returns the total; all threads of the workgroup call it
__device__ int blockExclusiveScan(int* data, int n, unsigned* tmp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + kOctThreads - 1) / kOctThreads;
    const int b = tid * per, e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += data[i];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    __syncthreads();   // tmp may still be read from a previous call
    if (lane == 63) tmp[wave] = (unsigned)incl;
    __syncthreads();
    int waveOff = 0, total = 0;
    for (int w = 0; w < kOctThreads / 64; w++) {
        if (w < wave) waveOff += (int)tmp[w];
        total += (int)tmp[w];
    }
    int run = waveOff + incl - sum;
    for (int i = b; i < e; i++) { const int v = data[i]; data[i] = run; run += v; }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int quadrantOf(int x, int y, short4 b /* x0,x1,y0,y1 */) {
    const int cx = b.x + ((b.y - b.x + 1) >> 1);   // UL.x + ceil((UR.x-UL.x)/2)
    const int cy = b.z + ((b.w - b.z + 1) >> 1);
    return (x < cx ? 0 : 1) + (y < cy ? 0 : 2);    // n1,n2,n3,n4 of DivideNode (:517-531)
}
__device__ __forceinline__ short4 childBox(short4 b, int q) {
    const short cx = (short)(b.x + ((b.y - b.x + 1) >> 1)), cy = (short)(b.z + ((b.w - b.z + 1) >> 1));
    short4 r;
    r.x = (q & 1) ? cx : b.x; r.y = (q & 1) ? b.y : cx;
    r.z = (q & 2) ? cy : b.z; r.w = (q & 2) ? b.w : cy;
    return r;
}

__global__ __launch_bounds__(kOctThreads) void k_octree(const LevelGeom* __restrict__ lv, int nlevels,
                                                         const uint2* __restrict__ cand,
                                                         const unsigned* __restrict__ candCount,
                                                         unsigned short* __restrict__ nodeOf,
                                                         uint2* __restrict__ sel, int selPerFrame,
                                                         int* __restrict__ levelCount, int* __restrict__ levelLap,
                                                         const int* __restrict__ lapArea, int M, int P) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ OctShared sh;
    const int level = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const LevelGeom g = lv[level];
    // carve the dynamic LDS
    short4* box[2];
    int* cnt[2];
    box[0] = (short4*)smem;              box[1] = box[0] + M;
    cnt[0] = (int*)(box[1] + M);         cnt[1] = cnt[0] + M;
    int* childCnt = cnt[1] + M;                       // [M][4]; reused as u64 best[M] at the end
    unsigned short* mapChild = (unsigned short*)(childCnt + 4 * M);   // [M][4]
    unsigned short* mapKeep = mapChild + 4 * M;       // [M]
    int* fwd = (int*)(mapKeep + M);                   // [M] forward (creation) offset of a split node's first child
    int* keepIdx = fwd + M;                           // [M]
    unsigned long long* sortKey = (unsigned long long*)(keepIdx + M);   // [P]   (M is a multiple of 8)

    int nC = (int)candCount[f * nlevels + level];
    nC = nC > g.candCap ? g.candCap : nC;
    const uint2* keys = cand + g.candOff + (long long)f * g.candCap;
    unsigned short* nof = nodeOf + g.candOff + (long long)f * g.candCap;
    uint2* selOut = sel + (long long)f * selPerFrame + g.selOff;
    const int N = g.quota;

    // ---- roots (:548-590) ----
    if (tid < g.nIni) {
        short4 b;
        b.x = (short)(int)(g.hX * (float)tid);
        b.y = (short)(int)(g.hX * (float)(tid + 1));
        b.z = 0; b.w = (short)g.rectH;
        box[1][tid] = b;
        cnt[1][tid] = 0;
    }
    __syncthreads();
    for (int k = tid; k < nC; k += kOctThreads) {
        const int x = keys[k].x & 0xfff;
        int r = (int)__fdiv_rn((float)x, g.hX);
        r = r > g.nIni - 1 ? g.nIni - 1 : r;
        nof[k] = (unsigned short)r;
        atomicAdd(&cnt[1][r], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int r = 0; r < g.nIni; r++) {
            if (cnt[1][r] > 0) { box[0][n] = box[1][r]; cnt[0][n] = cnt[1][r]; mapKeep[r] = (unsigned short)n; n++; }
        }
        sh.size = n; sh.phase2 = 0; sh.done = 0;
    }
    __syncthreads();
    if (g.nIni > 1)
        for (int k = tid; k < nC; k += kOctThreads) nof[k] = mapKeep[nof[k]];
    __syncthreads();

    int cur = 0, passes = 0;
    // ---- refinement passes (:599-744) ----
    while (sh.size > 0) {
        const int size = sh.size, phase2 = sh.phase2;
        short4* bx = box[cur];
        int* cn = cnt[cur];
        for (int i = tid; i < 4 * size; i += kOctThreads) childCnt[i] = 0;
        if (tid == 0) { sh.nToExpand = 0; sh.breakRank = 0x7fffffff; }
        __syncthreads();
        for (int k = tid; k < nC; k += kOctThreads) {
            const int n = nof[k];
            if (cn[n] > 1) {
                const unsigned w = keys[k].x;
                atomicAdd(&childCnt[4 * n + quadrantOf(w & 0xfff, (w >> 12) & 0xfff, bx[n])], 1);
            }
        }
        __syncthreads();
        // number of non-empty children of every multi-key node
        auto nch = [&](int n) {
            return (childCnt[4 * n] > 0) + (childCnt[4 * n + 1] > 0) + (childCnt[4 * n + 2] > 0) + (childCnt[4 * n + 3] > 0);
        };
        int C;   // children created in this pass
        if (!phase2) {
            for (int n = tid; n < size; n += kOctThreads) fwd[n] = cn[n] > 1 ? nch(n) : 0;
            __syncthreads();
            C = blockExclusiveScan(fwd, size, sh.scanTmp);
            for (int n = tid; n < size; n += kOctThreads) keepIdx[n] = cn[n] > 1 ? 0 : 1;
            __syncthreads();
        } else {
            // sort multi-key nodes by (size desc, list position asc)
            for (int i = tid; i < P; i += kOctThreads) {
                unsigned long long key = ~0ull;
                if (i < size && cn[i] > 1) key = ((unsigned long long)(0xffffffffu - (unsigned)cn[i]) << 32) | (unsigned)i;
                sortKey[i] = key;
            }
            __syncthreads();
            for (int k2 = 2; k2 <= P; k2 <<= 1) {
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < P; i += kOctThreads) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = sortKey[i], b = sortKey[ixj];
                            const bool up = (i & k2) == 0;
                            if ((a > b) == up) { sortKey[i] = b; sortKey[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            }
            // rank r -> node; inc[r] = children - 1; the pass breaks after the first r with
            // prevSize + sum_{i<=r} inc[i] >= N  (:735-736)
            int* inc = keepIdx;   // scratch: rank-indexed
            for (int r = tid; r < size; r += kOctThreads) {
                const unsigned long long key = sortKey[r];
                inc[r] = key == ~0ull ? 0 : nch((int)(unsigned)key) - 1;
            }
            __syncthreads();
            blockExclusiveScan(inc, size, sh.scanTmp);   // inc[r] = sum_{i<r}
            for (int r = tid; r < size; r += kOctThreads) {
                const unsigned long long key = sortKey[r];
                if (key != ~0ull) {
                    const int n = (int)(unsigned)key;
                    if (size + inc[r] + nch(n) - 1 >= N) atomicMin(&sh.breakRank, r);
                }
            }
            __syncthreads();
            const int br = sh.breakRank;
            // forward offset of rank r's first child = sum_{i<r} (inc_i + 1) = inc[r] + r
            for (int n = tid; n < size; n += kOctThreads) fwd[n] = -1;
            __syncthreads();
            for (int r = tid; r < size; r += kOctThreads) {
                const unsigned long long key = sortKey[r];
                if (key != ~0ull && r <= br) fwd[(int)(unsigned)key] = inc[r] + r;
            }
            __syncthreads();
            if (tid == 0) {
                // total children = offset of the last processed rank + its children
                int last = -1;
                // ranks of candidates are 0..nCand-1 (keys ~0 sort last)
                int lo = 0, hi = size;   // first rank whose key is ~0
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (sortKey[mid] == ~0ull) hi = mid; else lo = mid + 1; }
                const int nCand = lo;
                last = nCand - 1 < br ? nCand - 1 : br;
                sh.nChildren = last < 0 ? 0 : inc[last] + last + nch((int)(unsigned)sortKey[last]);
            }
            __syncthreads();
            C = sh.nChildren;
            for (int n = tid; n < size; n += kOctThreads) keepIdx[n] = fwd[n] < 0 ? 1 : 0;
            __syncthreads();
        }
        // in phase 1 a node splits iff it has several keys; in phase 2 iff fwd >= 0
        const int nKept = blockExclusiveScan(keepIdx, size, sh.scanTmp);
        const int newSize = C + nKept;
        if (newSize > M || ++passes > 64) {   // cannot happen for a valid geometry (host sizes M); never write out of bounds
            if (tid == 0) sh.size = 0;
            __syncthreads();
            break;
        }
        short4* nbx = box[cur ^ 1];
        int* ncn = cnt[cur ^ 1];
        for (int n = tid; n < size; n += kOctThreads) {
            const bool split = phase2 ? fwd[n] >= 0 : cn[n] > 1;
            if (split) {
                int j = fwd[n], expand = 0;
                for (int q = 0; q < 4; q++) {
                    const int c = childCnt[4 * n + q];
                    if (c > 0) {
                        const int pos = C - 1 - j;   // pushed to the front in creation order
                        nbx[pos] = childBox(bx[n], q);
                        ncn[pos] = c;
                        mapChild[4 * n + q] = (unsigned short)pos;
                        expand += c > 1;
                        j++;
                    }
                }
                if (expand) atomicAdd(&sh.nToExpand, expand);
            } else {
                const int pos = C + keepIdx[n];
                nbx[pos] = bx[n];
                ncn[pos] = cn[n];
                mapKeep[n] = (unsigned short)pos;
            }
        }
        __syncthreads();
        for (int k = tid; k < nC; k += kOctThreads) {
            const int n = nof[k];
            const bool split = phase2 ? fwd[n] >= 0 : cn[n] > 1;
            if (split) {
                const unsigned w = keys[k].x;
                nof[k] = mapChild[4 * n + quadrantOf(w & 0xfff, (w >> 12) & 0xfff, bx[n])];
            } else {
                nof[k] = mapKeep[n];
            }
        }
        __syncthreads();
        if (tid == 0) {
            sh.prevSize = size;
            sh.size = newSize;
            if (newSize >= N || newSize == size) sh.done = 1;
            else if (!phase2 && newSize + 3 * sh.nToExpand > N) sh.phase2 = 1;
        }
        cur ^= 1;
        __syncthreads();
        if (sh.done) break;
    }

    // ---- best key per node (:751-767) ----
    const int size = sh.size;
    unsigned long long* best = (unsigned long long*)childCnt;
    for (int i = tid; i < size; i += kOctThreads) best[i] = 0;
    __syncthreads();
    for (int k = tid; k < nC; k += kOctThreads) {
        const uint2 e = keys[k];
        const unsigned long long v = ((unsigned long long)(e.x >> 24) << 56) | ((unsigned long long)(~e.y) << 24) |
                                     (unsigned long long)(e.x & 0xffffff);
        atomicMax(&best[nof[k]], v);
    }
    __syncthreads();
    // lapping flags: the reference tests the level-0 x against vLappingArea (:1143-1151)
    const int lap0 = lapArea[2 * f], lap1 = lapArea[2 * f + 1];
    int* lapFlag = fwd;
    for (int i = tid; i < size; i += kOctThreads) {
        const unsigned long long v = best[i];
        const int x = (int)(v & 0xfff) + kMinBorder;
        float xs = (float)x;
        if (level != 0) xs = __fmul_rn(xs, g.scale);
        lapFlag[i] = (xs >= (float)lap0 && xs <= (float)lap1) ? 1 : 0;
        keepIdx[i] = lapFlag[i];
    }
    __syncthreads();
    const int nLap = blockExclusiveScan(keepIdx, size, sh.scanTmp);
    for (int i = tid; i < size; i += kOctThreads) {
        const unsigned long long v = best[i];
        uint2 o;
        const unsigned x = (unsigned)(v & 0xfff) + kMinBorder, y = (unsigned)((v >> 12) & 0xfff) + kMinBorder;
        o.x = x | (y << 12) | ((unsigned)(v >> 56) << 24);
        o.y = (unsigned)keepIdx[i] | ((unsigned)lapFlag[i] << 31);   // rank among lapping keys of this level
        selOut[i] = o;
    }
    if (tid == 0) {
        levelCount[f * nlevels + level] = size;
        levelLap[f * nlevels + level] = nLap;
    }
}

// ================================================================================================
// IC_Angle + rotated BRIEF + final placement.  One wave64 per kept keypoint.
// ================================================================================================
__constant__ int8_t c_pattern[1024] = {
#include "orbx_brief_pattern.inc"
};
__constant__ int c_umax[16];

// cv::fastAtan2: every operation rounded separately in binary32.
__device__ __forceinline__ float fastAtan2Deg(float y, float x) {
    const float sc = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc, p5 = 0.1555786518463281f * sc,
                p7 = -0.04432655554792128f * sc;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sinf/cosf as glibc >= 2.28 evaluates them for 0 <= y < 120: double-precision minimax polynomials after
// a quadrant reduction (constants of the published algorithm; tests compare the CPU twin of this routine
// with the host libm over every float in [0, 2*pi]).  Doubles, no contraction.
__device__ __forceinline__ void sincosGlibc(float y, float* s_out, float* c_out) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    double x = (double)y;
    const unsigned top12 = (__float_as_uint(y) >> 20) & 0x7ff;
    int n = 0;
    if (top12 < 0x3f4) {
        if (top12 < 0x398) { *s_out = y; *c_out = 1.0f; return; }
    } else {
        const double r = __dmul_rn(x, hpi_inv);
        n = ((int)r + 0x800000) >> 24;
        x = __dsub_rn(x, __dmul_rn((double)n, hpi));
    }
    const double x2 = __dmul_rn(x, x);
    auto polySin = [&](double xx) {
        const double x3 = __dmul_rn(xx, x2), s1 = __dadd_rn(S2, __dmul_rn(x2, S3)), x7 = __dmul_rn(x3, x2),
                     s = __dadd_rn(xx, __dmul_rn(x3, S1));
        return __dadd_rn(s, __dmul_rn(x7, s1));
    };
    auto polyCos = [&](double sg) {
        const double x4 = __dmul_rn(x2, x2), c2 = __dadd_rn(sg * C3, __dmul_rn(x2, sg * C4)),
                     c1 = __dadd_rn(sg * C0, __dmul_rn(x2, sg * C1)), x6 = __dmul_rn(x4, x2),
                     c = __dadd_rn(c1, __dmul_rn(x4, sg * C2));
        return __dadd_rn(c, __dmul_rn(x6, c2));
    };
    {
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign[n & 3]
        *s_out = (n & 1) ? (float)polyCos((n & 2) ? -1.0 : 1.0) : (float)polySin(x * sgn);
    }
    {
        const int m = n + 1;
        const double sgn = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
        *c_out = (m & 1) ? (float)polyCos((m & 2) ? -1.0 : 1.0) : (float)polySin(x * sgn);
    }
}

constexpr int kDescWaves = 4;

__global__ __launch_bounds__(256) void k_describe(const LevelGeom* __restrict__ lv, int nlevels,
                                                   const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                   const uint2* __restrict__ sel, int selPerFrame,
                                                   const int* __restrict__ levelCount, const int* __restrict__ levelLap,
                                                   Keypoint* __restrict__ outK, uint8_t* __restrict__ outD, int capacity,
                                                   int* __restrict__ nOut, int* __restrict__ monoOut,
                                                   Keypoint* __restrict__ outLevelK, int* __restrict__ outLevelCounts) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.x * kDescWaves + wave, f = blockIdx.y;
    if (slot >= selPerFrame) return;
    // totals of this frame
    int total = 0, totalLap = 0, level = 0, seqBase = 0, lapBase = 0;
    for (int l = 0; l < nlevels; l++) {
        const int c = levelCount[f * nlevels + l], lp = levelLap[f * nlevels + l];
        if (slot >= lv[l].selOff) { level = l; seqBase = total; lapBase = totalLap; }
        total += c;
        totalLap += lp;
    }
    if (slot == 0 && lane == 0) {
        nOut[f] = total;
        monoOut[f] = total - totalLap;   // monoIndex after the loop (:1161)
    }
    if (slot == 0 && outLevelCounts && lane < nlevels) outLevelCounts[f * nlevels + lane] = levelCount[f * nlevels + lane];
    const LevelGeom g = lv[level];
    const int i = slot - g.selOff;
    if (i >= levelCount[f * nlevels + level]) return;
    const uint2 e = sel[(long long)f * selPerFrame + slot];
    int kx = e.x & 0xfff, ky = (e.x >> 12) & 0xfff;
    const float response = (float)(e.x >> 24);
    // the quad-tree only emits points of the FAST rectangle; clamp anyway so a corrupted entry can never
    // turn into an out-of-bounds gather
    kx = min(max(kx, kEdge), g.w - kEdge - 1);
    ky = min(max(ky, kEdge), g.h - kEdge - 1);

    // ---- IC_Angle (:75-102): integer moments over the radius-15 disc of the unblurred level ----
    const uint8_t* center = pyr + g.pyrOff + (long long)f * g.pyrFrameBytes + (long long)(kEdge + ky) * g.pyrStride + kPadL + kx;
    int m10 = 0, m01 = 0;
    {
        const int col = lane & 31, u = col - kHalfPatch, half = lane >> 5;
        if (col <= 2 * kHalfPatch) {
            const int au = u < 0 ? -u : u;
            // half 0: rows v = 0..15, half 1: rows v = -1..-15
            for (int a = half; a <= kHalfPatch; a++) {
                if (au <= c_umax[a]) {
                    const int v = half ? -a : a;
                    const int val = center[(long long)v * g.pyrStride + u];
                    m10 += u * val;
                    m01 += v * val;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = fastAtan2Deg((float)m01, (float)m10);

    // ---- computeOrbDescriptor (:106-145) on the blurred level ----
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);   // (float)(CV_PI/180.f)
    float a, b;
    sincosGlibc(__fmul_rn(angle, factorPI), &b, &a);
    const uint8_t* bc = blur + g.blurOff + (long long)f * g.blurFrameBytes + (long long)ky * g.blurStride + kx;
    unsigned long long word[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = lane + 64 * j;   // test pair index; bit (p & 7) of descriptor byte (p >> 3)
        const float x0 = (float)c_pattern[4 * p], y0 = (float)c_pattern[4 * p + 1];
        const float x1 = (float)c_pattern[4 * p + 2], y1 = (float)c_pattern[4 * p + 3];
        const int r0 = (int)rintf(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = (int)rintf(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = (int)rintf(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = (int)rintf(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[(long long)r0 * g.blurStride + q0], t1 = bc[(long long)r1 * g.blurStride + q1];
        word[j] = __ballot(t0 < t1);
    }

    // ---- placement (:1137-1158): non-lapping keys fill from the front, lapping keys from the back ----
    const int lapRank = (int)(e.y & 0x7fffffff), isLap = (int)(e.y >> 31);
    const int lapBefore = lapBase + lapRank;
    const int monoBefore = (seqBase - lapBase) + (i - lapRank);
    const int at = isLap ? total - 1 - lapBefore : monoBefore;
    float ox = (float)kx, oy = (float)ky;
    if (level != 0) { ox = __fmul_rn(ox, g.scale); oy = __fmul_rn(oy, g.scale); }
    if (at < capacity) {
        if (lane == 0) {
            Keypoint k;
            k.x = ox; k.y = oy; k.size = (float)g.patchSize; k.angle = angle; k.response = response;
            k.octave = level; k.class_id = -1;
            outK[(long long)f * capacity + at] = k;
        }
        if (lane < 4) {
            unsigned long long w = word[0];
            w = lane == 1 ? word[1] : w;
            w = lane == 2 ? word[2] : w;
            w = lane == 3 ? word[3] : w;
            ((unsigned long long*)(outD + ((long long)f * capacity + at) * 32))[lane] = w;
        }
    }
    if (outLevelK && lane == 0 && seqBase + i < capacity) {
        Keypoint k;
        k.x = (float)kx; k.y = (float)ky; k.size = (float)g.patchSize; k.angle = angle; k.response = response;
        k.octave = level; k.class_id = -1;
        outLevelK[(long long)f * capacity + seqBase + i] = k;
    }
}

// ================================================================================================
// Launch wrappers (host)
// ================================================================================================
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
void launchBlur(hipStream_t st, const BlurTile* tiles, int nTiles, const LevelGeom* lv, const uint8_t* pyr,
                uint8_t* blur, int B) {
    hipLaunchKernelGGL(k_blur, dim3(nTiles, B), dim3(256), 0, st, tiles, lv, pyr, blur);
}
void fastLdsLayout(int maxRoiW, int maxRoiH, int* tileStride, int* tileBytes, int* scoreStride, int* scoreBytes) {
    *tileStride = (maxRoiW + 3) / 4 * 4 + 4;
    *tileBytes = (*tileStride * maxRoiH + 15) / 16 * 16;
    *scoreStride = (maxRoiW - 6 + 2 + 3) / 4 * 4 + 4;
    *scoreBytes = (*scoreStride * (maxRoiH - 6 + 2) + 15) / 16 * 16;
}
void launchFast(hipStream_t st, const CellDesc* cells, int nCells, const LevelGeom* lv, int nlevels,
                const uint8_t* pyr, int iniTh, int minTh, uint2* cand, unsigned* candCount, int maxRoiW, int maxRoiH,
                int B) {
    int ts, tb, ss, sb;
    fastLdsLayout(maxRoiW, maxRoiH, &ts, &tb, &ss, &sb);
    const size_t lds = (size_t)kFastWaves * (tb + sb);
    hipLaunchKernelGGL(k_fast, dim3((nCells + kFastWaves - 1) / kFastWaves, B), dim3(256), lds, st, cells, nCells, lv,
                       nlevels, pyr, iniTh, minTh, cand, candCount, ts, tb, ss, sb);
}
size_t octreeLdsBytes(int M, int P) {
    size_t b = 0;
    b += 2 * (size_t)M * sizeof(short4);            // box
    b += 2 * (size_t)M * sizeof(int);               // cnt
    b += 4 * (size_t)M * sizeof(int);               // childCnt / best
    b += 4 * (size_t)M * sizeof(unsigned short);    // mapChild
    b += (size_t)M * sizeof(unsigned short);        // mapKeep
    b += (size_t)M * sizeof(int);                   // fwd
    b += (size_t)M * sizeof(int);                   // keepIdx
    b += (size_t)P * sizeof(unsigned long long);    // sortKey
    return b + 64;
}
void launchOctree(hipStream_t st, const LevelGeom* lv, int nlevels, const uint2* cand, const unsigned* candCount,
                  unsigned short* nodeOf, uint2* sel, int selPerFrame, int* levelCount, int* levelLap,
                  const int* lapArea, int M, int P, int B) {
    hipLaunchKernelGGL(k_octree, dim3(nlevels, B), dim3(kOctThreads), octreeLdsBytes(M, P), st, lv, nlevels, cand,
                       candCount, nodeOf, sel, selPerFrame, levelCount, levelLap, lapArea, M, P);
}
void launchDescribe(hipStream_t st, const LevelGeom* lv, int nlevels, const uint8_t* pyr, const uint8_t* blur,
                    const uint2* sel, int selPerFrame, const int* levelCount, const int* levelLap, Keypoint* outK,
                    uint8_t* outD, int capacity, int* nOut, int* monoOut, Keypoint* outLevelK, int* outLevelCounts,
                    int B) {
    hipLaunchKernelGGL(k_describe, dim3((selPerFrame + kDescWaves - 1) / kDescWaves, B), dim3(256), 0, st, lv, nlevels,
                       pyr, blur, sel, selPerFrame, levelCount, levelLap, outK, outD, capacity, nOut, monoOut, outLevelK,
                       outLevelCounts);
}
hipError_t uploadUmax(const int* umax16) { return hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int)); }

// unpack one level's candidates into reference KeyPoints (introspection for tests)
__global__ void k_unpackCandidates(const uint2* __restrict__ keys, int n, Keypoint* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned w = keys[i].x;
    Keypoint k;
    k.x = (float)(w & 0xfff); k.y = (float)((w >> 12) & 0xfff); k.size = 7.f; k.angle = -1.f;
    k.response = (float)(w >> 24); k.octave = 0; k.class_id = -1;
    out[i] = k;
}
void launchUnpackCandidates(hipStream_t st, const uint2* keys, int n, Keypoint* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpackCandidates, dim3((n + 255) / 256), dim3(256), 0, st, keys, n, out);
}

}  // namespace orbx
