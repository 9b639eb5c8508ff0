// k_fast.hip — per-cell FAST-9/16 detection with threshold retry and NMS (reference ORBextractor.cc:797-864).
//
// One wave64 == one 30-px cell == one cv::FAST call of the reference (two when the first returns nothing).
//
// Score.  With ring pixels r_k and centre v:  S_dark = v - min_arcs(max_arc r),  S_bright = max_arcs(min_arc r) - v,
// S = max(S_dark, S_bright, 0) over the 16 arcs of 9 contiguous ring pixels.  "corner at threshold t" <=> S > t
// and the reference's response is S - 1, so ONE score serves iniThFAST and the minThFAST retry.
// Arcs k and k+1 (k even) share 8 ring pixels, and max(min(core, r_k), min(core, r_k+9)) = min(core, max(r_k, r_k+9)):
// 8 "arc pairs" instead of 16 arcs, 36 three-input min/max instructions per polarity.
//
// Packed form (the variant used on dense content): one lane = the four pixels of one tile dword, as two pixel PAIRS held
// as 2 x u16 per register.  The bit patterns 0..255 are non-negative binary16 denormals, whose order is the integer
// order, so gfx950's packed v_pk_minimum3_f16 / v_pk_maximum3_f16 are exact integer min3 / max3 on both pixels; ring
// pairs are cut out of the tile's dwords with v_perm_b32.  (Rounds 1-3 also kept a byte-per-lane variant that first rejected pixels with
// a cheap exact test and scored the survivors, for natural / sparse content; once the packed pass evaluated one polarity per pixel it was
// the faster one on every content measured - noise, textured, natural, sparse, 640x480 and 1920x1080 - and the variant was removed.)
//
// NMS is a strict 3x3 maximum of S with everything outside the cell interior counted as 0; a surviving centre has
// S > t, so neighbours below t can never suppress it: NMS is threshold independent and runs once (also four pixels
// per lane, minThFAST folded into the neighbour maximum).
//
// LDS: each wave stages its ROI (<= 45 x 45 for the usual 31..37-px cells) re-aligned so that the cell interior starts
// on a tile dword (a 31/32-px cell is then 8 dwords wide = 4 wave iterations per pass), then keeps the 8-bit score
// tile next to it on the same dword grid.  Strides are compile-time so ring offsets are instruction immediates.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "orbx_device.hpp"
#include "k_blur_body.hpp"
#define ORBX_FAST_TU 1
#include "k_fast_body.hpp"

namespace orbx {

// TS: LDS row stride of the pixel tile and of the score tile (bytes).  ROWS: max ROI rows.  FUSE_BLUR: chunks past tail.fastChunks run
// blur lanes (short-chain form) instead of FAST cells.
template <int TS, int ROWS, bool FUSE_BLUR = false, bool BIG = false>
__global__ __launch_bounds__(256, ORBX_FAST_WAVES) void k_fast(const CellDesc* __restrict__ cells, int nCells,
                                               const LevelGeom* __restrict__ lv, int nlevels,
                                               const uint8_t* __restrict__ pyr, int iniTh, int minTh,
                                               typename CandFmt<BIG>::T* __restrict__ candSeg, unsigned* __restrict__ cellCount, int f0, int nFrames,
                                               BlurTail tail, LeafTables lt) {
    using L = FastLds<TS, ROWS>;
    __shared__ __align__(16) uint8_t smem[L::kTiles];
    __shared__ __align__(16) uint8_t scoreS[L::kScores];
    __shared__ uint8_t codeL[kFastWaves][2][64];
    int chunk, fr;
    if (!xcdChunkFrame(nFrames, chunk, fr)) return;   // all cells of a frame on one XCD: the 6-px ROI overlap of neighbouring cells hits its L2
    FAST_SPAN(0);
    if constexpr (FUSE_BLUR) {
        if (chunk >= tail.fastChunks) {                // workgroup-uniform
            blurLanes<kBlurBlockRowsSmall>(tail.items, tail.laneItem, tail.nLanes, lv, pyr, tail.blur, chunk - tail.fastChunks, f0 + fr);
            FAST_SPAN(1);
            return;
        }
    }
    fastCell<TS, ROWS, BIG>(cells, nCells, lv, pyr, iniTh, minTh, candSeg, cellCount, lt, smem, scoreS, codeL, chunk, f0 + fr);
}

// ---- the small-batch form: one WORKGROUP (four waves) per cell.  A cell on one wave is a chain of ~1500 dependent-ish instructions and LDS
//      round trips — 6.6 us on a wave that has its SIMD to itself (tools/fast_spans.py: staged 1.05, score pass 2.3, NMS + count 1.9, emit 1.3)
//      although the instructions are 2.6 us of issue — and while a call holds fewer cells than the chip has SIMDs that latency IS the launch.
//      Here the four waves share the cell's dword items (256 per trip instead of 64); the raster order of the candidates (cv::FAST's output
//      order, ORBextractor.cc:797-864) comes from per-wave counts and ONE exchange of wave totals: every kept pixel learns its place in the
//      minThFAST list and its place among the iniThFAST ones, the retry rule (:835-838) picks one of the two, and the candidates go straight
//      from registers to the segment (no list in LDS).  Four barriers.  Same arithmetic as k_fast. ----
template <int TS, int ROWS, bool FUSE_BLUR>
__global__ __launch_bounds__(256) void k_fast_wide(const CellDesc* __restrict__ cells, int nCells,
                                                    const LevelGeom* __restrict__ lv, int nlevels,
                                                    const uint8_t* __restrict__ pyr, int iniTh, int minTh,
                                                    unsigned* __restrict__ candSeg, unsigned* __restrict__ cellCount, int f0, int nFrames,
                                                    BlurTail tail, LeafTables lt) {
    constexpr int kTileBytes = TS * ROWS, kScoreBytes = TS * (ROWS - 4), DW = TS / 4;
    constexpr int LPR = DW <= 16 ? 8 : 16, RPI = 256 / LPR, STEPS = (ROWS + RPI - 1) / RPI;      // lanes per row while staging, rows per step
    static_assert(DW % 2 == 0 && TS % 8 == 0, "dword pairs per tile row");
    constexpr int kRounds = (DW * (ROWS - 6) + 255) / 256;      // trips of 256 items over the largest cell
    __shared__ __align__(16) uint8_t smem[16 + kTileBytes + kScoreBytes];
    __shared__ int wMin[kRounds][4], wIni[kRounds][4];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int chunk, fr;
    if (!xcdChunkFrame(nFrames, chunk, fr)) return;
    FAST_SPAN(0);
    if constexpr (FUSE_BLUR) {
        if (chunk >= nCells) {                         // workgroup-uniform
            blurLanes<kBlurBlockRowsSmall>(tail.items, tail.laneItem, tail.nLanes, lv, pyr, tail.blur, chunk - nCells, f0 + fr);
            FAST_SPAN(1);
            return;
        }
    }
    const int ci = chunk, f = f0 + fr;
    if (ci >= nCells) return;
    const CellDesc c = cells[ci];
    const LevelGeom g = lv[c.level];
    uint8_t* tile = smem + 16;
    uint8_t* score = tile + kTileBytes;
    const int roiW = c.roiW, roiH = c.roiH, cw = roiW - 6, ch = roiH - 6;
    // the x / y path codes of the cell's interior columns and rows, fetched with the ROI (k_fast has the reasons)
    __shared__ uint8_t codeL[2][64];
    const bool leaf = lt.hist != nullptr && g.leafOK && f < lt.frames;      // workgroup-uniform
    uint8_t myCode = 0;
    if (leaf && tid < 128)
        myCode = tid < 64 ? lt.xcode[c.level * lt.XT + min(c.shiftX + 3 + tid, g.rectW - 1)] : lt.ycode[c.level * lt.XT + min(c.shiftY + 3 + tid - 64, g.rectH - 1)];
    // ---- stage the ROI, re-aligned (k_fast has the reasons): tile byte k of row r is ROI pixel (k - 1, r) ----
    const int gx1 = kPadL + c.x0 - 1, gsh = gx1 & 3;
    constexpr int mis = kFastTileShift;
    {
        const uint8_t* sp = pyr + c.pyrOff + (long long)f * c.pyrFrameBytes + (long long)(kEdge + c.y0) * c.pyrStride + (gx1 - gsh);
        const int dcol = tid & (LPR - 1), rsub = tid / LPR;
        const unsigned colOff = 8u * (unsigned)min(dcol, (gsh + roiW) >> 3);
        const unsigned off0 = (unsigned)__mul24(rsub, c.pyrStride) + colOff, offMax = (unsigned)__mul24(roiH - 1, c.pyrStride) + colOff;
        const unsigned stepOff = (unsigned)(RPI * c.pyrStride);
        uint2 w[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; s++) __builtin_memcpy(&w[s], sp + min(off0 + s * stepOff, offMax), 8);
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const int r = s * RPI + rsub;
            const unsigned next = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w[s].x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);      // (a row's lanes sit in one wave)
            if (dcol < DW / 2 && r < ROWS)
                *(uint2*)(tile + r * TS + 8 * dcol) = uint2{__builtin_amdgcn_alignbyte(w[s].y, w[s].x, (unsigned)gsh), __builtin_amdgcn_alignbyte(next, w[s].y, (unsigned)gsh)};
        }
    }
    if (leaf && tid < 128) codeL[tid >> 6][tid & 63] = myCode;
#pragma unroll
    for (int i = 0; i < (kScoreBytes + 1023) / 1024; i++)
        if (tid * 4 + i * 1024 < kScoreBytes) *(unsigned*)(score + tid * 4 + i * 1024) = 0u;
    __syncthreads();
    FAST_MID(0);
    const int q0 = (mis + 3) >> 2, q1 = (mis + 2 + cw) >> 2, nq = q1 - q0 + 1;
    const int nItems = nq * ch;
    const int sy = (256 * c.itemRecip) >> 16, sx = 256 - sy * nq;                // 256 / nq, 256 % nq (exact: CellDesc::itemRecip)
    const int y0w = (tid * c.itemRecip) >> 16, qi0w = tid - y0w * nq;             // tid / nq, tid % nq
    // ---- pass 1: scores (one thread = the four pixels of one tile dword) ----
    {
        const int lo = mis + 3 - 4 * q0, hi = mis + 3 + cw - 4 * q1;
        const unsigned maskFirst = 0xFFFFFFFFu << (8 * lo), maskLast = hi >= 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8 * hi));
        int qi = qi0w, y = y0w;
        for (int item = tid; item < nItems; item += 256) {
            const uint8_t* base = tile + y * TS + 4 * (q0 + qi);
            unsigned L[7], C[7], R[7];
#pragma unroll
            for (int d = 0; d < 7; d++) {
                L[d] = *(const unsigned*)(base + d * TS - 4);
                C[d] = *(const unsigned*)(base + d * TS);
                R[d] = *(const unsigned*)(base + d * TS + 4);
            }
            unsigned rA[17], rB[17];
            pairRing<0>(L, C, R, rA);
            pairRing<2>(L, C, R, rB);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned sA = pairScore(rA);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned sB = pairScore(rB);
            unsigned m = qi == 0 ? maskFirst : 0xFFFFFFFFu;
            m = qi == nq - 1 ? (m & maskLast) : m;
            *(unsigned*)(score + (y + 1) * TS + 4 * (q0 + qi)) = __builtin_amdgcn_perm(sB, sA, 0x06040200u) & m;
            qi += sx; y += sy;
            if (qi >= nq) { qi -= nq; y++; }
        }
    }
    __syncthreads();
    FAST_MID(1);
    // ---- pass 2: strict local maxima (k_fast's arithmetic); per trip every wave counts what it keeps, at minThFAST and at iniThFAST ----
    constexpr int kLeafCap = kTileBytes / 8;                                 // (the pixel tile is dead after pass 1)
    unsigned *tHist = (unsigned*)tile, *tBest = tHist + kLeafCap;
    int xc0 = 0, yc0 = 0, nxl = 1, nLeafLocal = 0;
    const uint8_t *xcL = codeL[0], *ycL = codeL[1];
    if (leaf) {
        xc0 = xcL[0]; yc0 = ycL[0];
        nxl = (int)xcL[cw - 1] - xc0 + 1;
        const int nyl = (int)ycL[ch - 1] - yc0 + 1;
        nLeafLocal = nxl * nyl <= kLeafCap ? nxl * nyl : 0;      // 0: a cell over too many leaf cells goes to L2 key by key
        for (int e = tid; e < nLeafLocal; e += 256) { tHist[e] = 0u; tBest[e] = 0u; }
    }
    unsigned keepS[kRounds][4];      // score of each of the item's four pixels, 0 = not kept
    unsigned xyOf[kRounds];
    int preMin[kRounds], preIni[kRounds];      // kept pixels of the lower lanes of this wave in this trip
    {
        const unsigned thPair = (unsigned)minTh | ((unsigned)minTh << 16);
        int qi = qi0w, y = y0w;
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const bool act = r * 256 + tid < nItems;
            const uint8_t* base = score + (act ? y * TS + 4 * (q0 + qi) : 4);
            unsigned U[3], M[3], D[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                U[k] = *(const unsigned*)(base + 4 * k - 4);
                M[k] = *(const unsigned*)(base + TS + 4 * k - 4);
                D[k] = *(const unsigned*)(base + 2 * TS + 4 * k - 4);
            }
            const unsigned Wa = pkmax3(pairAt<3>(U[0], U[1], U[2]), pairAt<3>(D[0], D[1], D[2]), thPair);
            const unsigned Wb = pkmax3(pairAt<4>(U[0], U[1], U[2]), pairAt<4>(D[0], D[1], D[2]), thPair);
            const unsigned Wc = pkmax3(pairAt<5>(U[0], U[1], U[2]), pairAt<5>(D[0], D[1], D[2]), thPair);
            const unsigned Wd = pkmax3(pairAt<6>(U[0], U[1], U[2]), pairAt<6>(D[0], D[1], D[2]), thPair);
            const unsigned We = pkmax3(pairAt<7>(U[0], U[1], U[2]), pairAt<7>(D[0], D[1], D[2]), thPair);
            const unsigned Ma = pairAt<3>(M[0], M[1], M[2]), Mc = pairAt<5>(M[0], M[1], M[2]), Me = pairAt<7>(M[0], M[1], M[2]);
            const unsigned sA = pairAt<4>(M[0], M[1], M[2]), sB = pairAt<6>(M[0], M[1], M[2]);
            const unsigned mA = pkmax3(pkmax3(Wa, Wb, Wc), Ma, Mc), mB = pkmax3(pkmax3(Wc, Wd, We), Mc, Me);
            unsigned dA, dB;
            asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dA) : "v"(sA), "v"(mA));
            asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(dB) : "v"(sB), "v"(mB));
            keepS[r][0] = act && (dA & 0xFFFFu) != 0 ? (sA & 0xFFFFu) : 0u;
            keepS[r][1] = act && dA > 0xFFFFu ? (sA >> 16) : 0u;
            keepS[r][2] = act && (dB & 0xFFFFu) != 0 ? (sB & 0xFFFFu) : 0u;
            keepS[r][3] = act && dB > 0xFFFFu ? (sB >> 16) : 0u;
            xyOf[r] = (unsigned)(4 * (q0 + qi) - (mis + 3)) + ((unsigned)y << 6);
            int pm = 0, pi = 0, nm = 0, ni = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned long long bm = __ballot(keepS[r][k] != 0u), bi = __ballot((int)keepS[r][k] > iniTh);
                pm = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, pm));
                pi = __builtin_amdgcn_mbcnt_hi((unsigned)(bi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bi, pi));
                nm += __popcll(bm); ni += __popcll(bi);
            }
            preMin[r] = pm; preIni[r] = pi;
            if (lane == 0) { wMin[r][wave] = nm; wIni[r][wave] = ni; }
            qi += sx; y += sy;
            if (qi >= nq) { qi -= nq; y++; }
        }
    }
    __syncthreads();
    // ---- place: trips in order, waves in order inside a trip, lanes inside a wave, pixels inside a lane = raster order ----
    int nMin = 0, nIni = 0, baseMin[kRounds], baseIni[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w == wave) { baseMin[r] = nMin; baseIni[r] = nIni; }
            nMin += wMin[r][w]; nIni += wIni[r][w];
        }
    // the reference retries the cell at minThFAST only when the first call returned nothing (:835-838)
    const bool useIni = nIni > 0;
    const int total = useIni ? nIni : nMin;
    if (tid == 0) cellCount[(long long)f * nCells + ci] = (unsigned)total;
    FAST_MID(2);
    FAST_SPAN(1);
    if (total == 0) return;                             // workgroup-uniform
    unsigned* outPos = candSeg + g.candOff + (long long)f * g.candCap + c.segOff;
    const unsigned segCap = (unsigned)(((cw + 1) >> 1) * ((ch + 1) >> 1));
    const int th = useIni ? iniTh : minTh;
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        unsigned at = (unsigned)(useIni ? baseIni[r] + preIni[r] : baseMin[r] + preMin[r]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = (int)keepS[r][k];
            if (s > th) {                               // (kept pixels hold S > minThFAST)
                if (at < segCap) {
                    const unsigned e = xyOf[r] + (unsigned)k;
                    const int x = (int)(e & 63), y = (int)((e >> 6) & 63);
                    const unsigned kx = (unsigned)(c.shiftX + x + 3), ky = (unsigned)(c.shiftY + y + 3);
                    const unsigned w = kx | (ky << 12) | ((unsigned)(s - 1) << 24);   // response = S - 1
                    outPos[at] = w;
                    if (leaf) {
                        const int xc = xcL[x], yc = ycL[y];
                        const unsigned val = (w & 0xff000000u) | (0xffffffu - ((unsigned)c.segOff + at));
                        if (nLeafLocal) {
                            const int li = (yc - yc0) * nxl + (xc - xc0);
                            atomicAdd(&tHist[li], 1u);
                            atomicMax(&tBest[li], val);
                        } else {
                            const unsigned cellOfRoot = leafTableEntry(lt, f, c.level, xc, yc);
                            atomicAdd(lt.hist + cellOfRoot, 1);
                            atomicMax(lt.best + cellOfRoot, val);
                        }
                    }
                }
                at++;
            }
        }
    }
    if (nLeafLocal) {                                   // workgroup-uniform
        __syncthreads();
        for (int e = tid; e < nLeafLocal; e += 256) {
            const unsigned n = tHist[e];
            if (n) {
                const int er = (int)(((float)e + 0.5f) * __frcp_rn((float)nxl));      // e / nxl (exact: e < 1024, the quotient is >= 0.5 / nxl away from an integer)
                const int xc = xc0 + (e - er * nxl), yc = yc0 + er;
                const unsigned cellOfRoot = leafTableEntry(lt, f, c.level, xc, yc);
                atomicAdd(lt.hist + cellOfRoot, (int)n);
                atomicMax(lt.best + cellOfRoot, tBest[e]);
            }
        }
    }
    FAST_SPAN(1);
}

void launchFast(hipStream_t st, const CellDesc* cells, int nCells, const LevelGeom* lv, int nlevels,
                const uint8_t* pyr, int iniTh, int minTh, unsigned* candSeg, unsigned* cellCount, int maxRoiW, int maxRoiH,
                int f0, int B, const BlurItem* blurItems, const unsigned short* blurLaneItem, int blurLanes, uint8_t* blur,
                LeafTables lt, bool wide, bool big) {
    const int fastChunks = (nCells + kFastWaves - 1) / kFastWaves;
    const dim3 block(256);
    BlurTail tail{blurItems, blurLaneItem, blurLanes, blur, fastChunks};
    if (big) {      // frames beyond 4096 px: two-dword candidates (the host keeps the blur in a launch of its own and the leaf tables off for them)
        typedef CandFmt<true>::T* BigSeg;
        const dim3 grid = xcdGrid(fastChunks, B);
        if (maxRoiW <= 45 && maxRoiH <= 45)
            hipLaunchKernelGGL((k_fast<48, 45, false, true>), grid, block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh, minTh, (BigSeg)candSeg, cellCount, f0, B, tail, lt);
        else
            hipLaunchKernelGGL((k_fast<72, 69, false, true>), grid, block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh, minTh, (BigSeg)candSeg, cellCount, f0, B, tail, lt);
        return;
    }
    // ROI of w pixels at any dword misalignment needs (3 + w + 3) / 4 dwords
    if (wide && maxRoiW <= 45 && maxRoiH <= 45) {      // few cells: a workgroup per cell
        if (blurItems)
            hipLaunchKernelGGL((k_fast_wide<48, 45, true>), xcdGrid(nCells + (blurLanes + 255) / 256, B), block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh,
                               minTh, candSeg, cellCount, f0, B, tail, lt);
        else hipLaunchKernelGGL((k_fast_wide<48, 45, false>), xcdGrid(nCells, B), block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh, minTh, candSeg, cellCount, f0, B, tail, lt);
        return;
    }
    if (maxRoiW <= 45 && maxRoiH <= 45) {
        if (blurItems) {      // small batch: the blur's lanes ride in the same launch
            const dim3 grid = xcdGrid(fastChunks + (blurLanes + 255) / 256, B);
            hipLaunchKernelGGL((k_fast<48, 45, true>), grid, block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh, minTh, candSeg, cellCount, f0, B, tail, lt);
            return;
        }
        const dim3 grid = xcdGrid(fastChunks, B);
        hipLaunchKernelGGL((k_fast<48, 45, false>), grid, block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh, minTh, candSeg, cellCount, f0, B, tail, lt);
    } else {   // cells up to 63 px (the geometry code rejects larger ones)
        const dim3 grid = xcdGrid(fastChunks, B);
        hipLaunchKernelGGL((k_fast<72, 69, false>), grid, block, 0, st, cells, nCells, lv, nlevels, pyr, iniTh, minTh, candSeg, cellCount, f0, B, tail, lt);
    }
}
bool fastCanCarryBlur(int maxRoiW, int maxRoiH) { return maxRoiW <= 45 && maxRoiH <= 45; }

// Create-time check of the assumption the packed passes rest on: v_pk_minimum3_f16 / v_pk_maximum3_f16 / v_pk_sub_u16
// behave as integer min3 / max3 / subtract on u16 halves holding 0..255 (FP16 denormals preserved).  One wave; lane l
// tests the triple (l*4+1, 255-l*3, (l*37) & 255) in the low half and a rotated triple in the high half.
__global__ void k_packedSelfTest(unsigned* __restrict__ bad) {
    const unsigned l = threadIdx.x;
    const unsigned a0 = (l * 4 + 1) & 255, b0 = (255 - l * 3) & 255, c0 = (l * 37) & 255;
    const unsigned a = a0 | (b0 << 16), b = b0 | (c0 << 16), c = c0 | (a0 << 16);
    const unsigned mn = pkmin3(a, b, c), mx = pkmax3(a, b, c), df = pksub(mx, mn), m2 = pkmax(a, b);
    const unsigned emn = min(a0, min(b0, c0)), emx = max(a0, max(b0, c0));
    const bool ok = mn == (emn | (emn << 16)) && mx == (emx | (emx << 16)) && df == ((emx - emn) | ((emx - emn) << 16)) &&
                    m2 == (max(a0, b0) | (max(b0, c0) << 16)) && pkmin3(0u, 0x00010001u, 0x00ff00ffu) == 0u;
    if (!ok) atomicAdd(bad, 1u);
}
hipError_t runPackedSelfTest(hipStream_t st, unsigned* d_scratch, unsigned* h_bad) {
    hipError_t e = hipMemsetAsync(d_scratch, 0, sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_packedSelfTest, dim3(1), dim3(64), 0, st, d_scratch);
    e = hipMemcpyAsync(h_bad, d_scratch, sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

}  // namespace orbx
