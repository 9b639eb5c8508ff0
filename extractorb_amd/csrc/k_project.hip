// k_project.hip — "next" row SURVEY.md §8f-2, second half: ORBmatcher::SearchByProjection on device-resident frames.
//   (a) frame-to-frame, Tracking::TrackWithMotionModel   reference src/ORBmatcher.cc:1961-2177
//   (b) map-to-frame,   Tracking::SearchLocalPoints       reference src/ORBmatcher.cc:44-267
//   (c) keyframe-to-frame, Tracking::Relocalization        reference src/ORBmatcher.cc:2179-2300 (the search of (a) with the caller's
//       distance bound ORBdist, every keypoint that already holds a MapPoint closed, no stereo column test)
// all for Nleft == -1 (one descriptor set per frame: monocular, rectified stereo, RGB-D), with Frame::GetFeaturesInArea
// (src/Frame.cc:655-724), ORBmatcher::DescriptorDistance (:2349-2365), ComputeThreeMaxima (:2303-2344) and Pinhole::project
// (src/CameraModels/Pinhole.cpp:30-33).
//
// Two kernels.  k_project_last is the embarrassingly parallel front half of (a): one thread per keypoint of the last frame turns
// its MapPoint (world position) into a search request ("query": projected position, predicted right-eye column, window radius,
// level range) under the current frame's pose.  For (b) the tracker's frustum test already produced those numbers
// (MapPoint::mTrackProjX / mTrackProjY / mTrackProjXR / mnTrackScaleLevel / mTrackViewCos), so the caller passes queries directly.
// k_search_proj is the matching proper.  A keypoint that receives a MapPoint with observations is closed to later requests
// (:64-66, :2035-2037), so the requests of one frame form a sequential chain: ONE WORKGROUP PER FRAME walks them in order and its
// 256 lanes share each search, exactly as k_search_init does (k_match.hip):
//   * the current frame's keypoints are staged once into LDS in grid order (cell x*48+y, push_back order inside a cell) with
//     descriptor words transposed; GetFeaturesInArea visits cells in ascending grid position, so the strict "<" of the
//     reference's running minimum is "smallest (distance, position)" and the cell columns of a window are one slot range;
//   * a lane keeps its two smallest keys (distance << 16 | position); the reference's (bestDist, bestLevel, bestDist2, bestLevel2)
//     after its sequential scan are the smallest and the second smallest key of the window (the element that ends up providing
//     bestDist2 is the first in traversal order among those with the second smallest distance — see DESIGN.md §4f);
//   * acceptance is uniform over the workgroup; every wave records "occupied" for its own next search, wave 0 owns the match
//     tables; one barrier per request (merge scratch double-buffered).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

struct ProjQuery { float u, v, ur, radius; int minLevel, maxLevel, flags; float angle; };   // == orbx_proj_query
static_assert(sizeof(ProjQuery) == 32, "orbx_proj_query layout");

struct ProjectParams {
    float fx, fy, cx, cy, minX, maxX, minY, maxY;
    float scale[kMaxLevels];
    float mbf, mb, th;
    int mono, capacity, lastFirst, lastStep, curFirst, curStep;
};

struct ProjSearchParams {
    float minX, minY, wInv, hInv, nnRatio;
    int ratioMode, checkOrientation, capacity, queryCapacity, curFirst, curStep, descFirst, descStep, maxDist;
};

namespace {
constexpr int kCols = 64, kRows = 48, kCells = kCols * kRows;
constexpr int kHistoLength = 30;                          // ORBmatcher.cc:38 (the distance bound, TH_HIGH = 100 or the caller's ORBdist, is a parameter)
constexpr int kNoneKey = (256 << 16) | 0xFFFF;            // bestDist = 256, no position
constexpr int kThreads = 256, kWaves = kThreads / 64;

__device__ __forceinline__ int bcast(int v, int srcLane) { return __builtin_amdgcn_readlane(v, srcLane); }
__device__ __forceinline__ float bcastf(float v, int srcLane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srcLane)); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dppMin(int v) { return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xF, false)); }
__device__ __forceinline__ int waveMin(int v) {          // minimum over the 64 lanes, wave-uniform
    v = dppMin<0xB1, 0xF>(v);
    v = dppMin<0x4E, 0xF>(v);
    v = dppMin<0x141, 0xF>(v);
    v = dppMin<0x140, 0xF>(v);
    v = dppMin<0x142, 0xA>(v);
    v = dppMin<0x143, 0xC>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

// one row of cv::gemm on 3x3 * 3x1 float data: products and sums in double (each rounded), scaled, C added, rounded to float once
__device__ __forceinline__ float gemmRow(float a0, float a1, float a2, const float (&b)[3], double alpha, float c, bool hasC) {
    double s = __dmul_rn((double)a0, (double)b[0]);
    s = __dadd_rn(s, __dmul_rn((double)a1, (double)b[1]));
    s = __dadd_rn(s, __dmul_rn((double)a2, (double)b[2]));
    s = __dmul_rn(s, alpha);
    if (hasC) s = __dadd_rn(s, (double)c);
    return (float)s;
}
}  // namespace

// grid (ceil(capacity / 256), n_pairs).  Tcw: one 3x4 row-major pose per FRAME.
__global__ __launch_bounds__(256) void k_project_last(const Keypoint* __restrict__ kps, const Keypoint* __restrict__ kpsUn,
                                                      const int* __restrict__ nOut, const uint8_t* __restrict__ mpFlags,
                                                      const float* __restrict__ world, const float* __restrict__ poses, ProjectParams p,
                                                      ProjQuery* __restrict__ queries) {
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.capacity) return;
    const int fl = p.lastFirst + pair * p.lastStep, fc = p.curFirst + pair * p.curStep;
    ProjQuery q{0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0.f};
    ProjQuery* out = queries + (long long)pair * p.capacity + i;
    const int NL = min(nOut[fl], p.capacity);
    const uint8_t fl8 = i < NL ? mpFlags[(long long)fl * p.capacity + i] : (uint8_t)0;
    if (!(fl8 & 1)) { *out = q; return; }                                        // ORBmatcher.cc:1987-1990
    const float* C = poses + (long long)fc * 12;
    const float* L = poses + (long long)fl * 12;
    const float tcw[3] = {C[3], C[7], C[11]};
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) twc[r] = gemmRow(C[r], C[4 + r], C[8 + r], tcw, -1.0, 0.f, false);      // -Rcw.t()*tcw (:1975)
    for (int r = 0; r < 3; r++) tlc[r] = gemmRow(L[4 * r], L[4 * r + 1], L[4 * r + 2], twc, 1.0, L[4 * r + 3], true);   // Rlw*twc+tlw (:1980)
    const bool bForward = tlc[2] > p.mb && !p.mono, bBackward = -tlc[2] > p.mb && !p.mono;                // :1982-1983
    const float* X = world + ((long long)fl * p.capacity + i) * 3;
    const float xw[3] = {X[0], X[1], X[2]};
    float xc[3];
    for (int r = 0; r < 3; r++) xc[r] = gemmRow(C[4 * r], C[4 * r + 1], C[4 * r + 2], xw, 1.0, C[4 * r + 3], true);   // Rcw*x3Dw+tcw (:1994)
    const float invzc = (float)__ddiv_rn(1.0, (double)xc[2]);                     // :1998
    if (invzc < 0) { *out = q; return; }
    const float u = __fadd_rn(__fdiv_rn(__fmul_rn(p.fx, xc[0]), xc[2]), p.cx);   // Pinhole::project
    const float v = __fadd_rn(__fdiv_rn(__fmul_rn(p.fy, xc[1]), xc[2]), p.cy);
    if (u < p.minX || u > p.maxX || v < p.minY || v > p.maxY) { *out = q; return; }   // :2005-2008
    const int oct = min(max(kps[(long long)fl * p.capacity + i].octave, 0), kMaxLevels - 1);   // LastFrame.mvKeys[i].octave (:2010)
    q.u = u; q.v = v;
    q.radius = __fmul_rn(p.th, p.scale[oct]);                                     // :2014
    q.ur = __fsub_rn(u, __fmul_rn(p.mbf, invzc));                                // :2043
    if (bForward) { q.minLevel = oct; q.maxLevel = -1; }                          // :2018-2023
    else if (bBackward) { q.minLevel = 0; q.maxLevel = oct; }
    else { q.minLevel = oct - 1; q.maxLevel = oct + 1; }
    q.angle = kpsUn[(long long)fl * p.capacity + i].angle;                        // kpLF = LastFrame.mvKeysUn[i] (:2067)
    q.flags = 1 | (fl8 & 2);
    *out = q;
}

size_t projSearchLdsBytes(int capacity) {
    const size_t c = (size_t)((capacity + 3) & ~3);
    return c * (32 + 4 + 4 + 4 + 4 + 4 + 4 + 2 + 2 + 1 + 1 + 2) + (kCols + 2) * sizeof(int) + 2 * kWaves * 2 * sizeof(int) + 2 * kWaves * sizeof(int) + 64;
}

// grid: n_pairs; 256 threads.
__global__ __launch_bounds__(kThreads) void k_search_proj(const ProjQuery* __restrict__ queries, const uint8_t* __restrict__ qdesc,
                                                          const int* __restrict__ nQueries, const Keypoint* __restrict__ kpsUn,
                                                          const uint8_t* __restrict__ desc, const int* __restrict__ nOut,
                                                          const int* __restrict__ gridOff, const int* __restrict__ gridIdx,
                                                          const float* __restrict__ uRight, uint8_t* __restrict__ occupied,
                                                          ProjSearchParams p, int* __restrict__ matches, int* __restrict__ nMatches) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int cap = p.capacity, capA = (cap + 3) & ~3;
    uint32_t* d2w = (uint32_t*)smem;                       // [8][capA] descriptor word k of slot s
    float* x2 = (float*)(d2w + 8 * capA);                  // [capA]
    float* y2 = x2 + capA;
    float* a2 = y2 + capA;                                 // angle
    float* ur2 = a2 + capA;                                // mvuRight (<= 0: no stereo observation)
    int* m2q = (int*)(ur2 + capA);                         // query whose MapPoint the keypoint holds, -1 = none
    unsigned* binMask = (unsigned*)(m2q + capA);           // rotHist bins the keypoint was pushed to
    int* colStart = (int*)(binMask + capA);                // [66] first slot of cell column c (c = 64, 65: n2)
    int* merge = colStart + kCols + 2;                     // [2][kWaves][2] per-wave (best key, second key), double-buffered
    int* wcnt = merge + 2 * kWaves * 2;                    // [2][kWaves] staging counts
    unsigned short* cell2 = (unsigned short*)(wcnt + 2 * kWaves);  // ix << 8 | iy
    unsigned short* idx2 = cell2 + capA;                   // keypoint index in the frame
    uint8_t* oct2 = (uint8_t*)(idx2 + capA);               // octave
    uint8_t* occ = oct2 + capA;                            // holds a MapPoint with Observations() > 0

    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f2 = p.curFirst + pair * p.curStep;
    const int N2 = min(nOut[f2], cap);
    const Keypoint* K2 = kpsUn + (long long)f2 * cap;
    const uint32_t* D2 = (const uint32_t*)(desc + (long long)f2 * cap * 32);
    const int* off2 = gridOff + (long long)f2 * (kCells + 1);
    const int* gi2 = gridIdx + (long long)f2 * cap;
    const float* UR = uRight ? uRight + (long long)f2 * cap : nullptr;
    uint8_t* occIO = occupied ? occupied + (long long)pair * cap : nullptr;
    const ProjQuery* Q = queries + (long long)pair * p.queryCapacity;
    const uint32_t* QD = (const uint32_t*)(qdesc + (long long)(p.descFirst + pair * p.descStep) * p.queryCapacity * 32);
    const int NQ = nQueries ? min(nQueries[pair], p.queryCapacity) : p.queryCapacity;
    int* out = matches + (long long)pair * cap;
    const int nIn2 = min(off2[kCells], N2);

    // ---- stage the frame's keypoints in grid order (every octave: the level window differs per request) ----
    int n2 = 0;
    for (int base = 0, it = 0; base < nIn2; base += kThreads, it ^= 1) {
        const int pos = base + tid;
        const bool keep = pos < nIn2;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcnt[it * kWaves + wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { const int c = wcnt[it * kWaves + w]; all += c; before += w < wave ? c : 0; }
        const int slot = n2 + before + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            const int i2 = gi2[pos];
            const Keypoint k = K2[i2];
            const int posX = (int)roundf(__fmul_rn(__fsub_rn(k.x, p.minX), p.wInv));   // the cell AssignFeaturesToGrid put it in
            const int posY = (int)roundf(__fmul_rn(__fsub_rn(k.y, p.minY), p.hInv));   // (PosInGrid, Frame.cc:728-729)
            x2[slot] = k.x; y2[slot] = k.y; a2[slot] = k.angle;
            ur2[slot] = UR ? UR[i2] : -1.0f;
            cell2[slot] = (unsigned short)((posX << 8) | posY);
            idx2[slot] = (unsigned short)i2;
            oct2[slot] = (uint8_t)min(max(k.octave, 0), 255);
            occ[slot] = occIO ? occIO[i2] : (uint8_t)0;
            m2q[slot] = -1; binMask[slot] = 0u;
            const uint4 lo = *(const uint4*)(D2 + (long long)i2 * 8), hi = *(const uint4*)(D2 + (long long)i2 * 8 + 4);
            d2w[0 * capA + slot] = lo.x; d2w[1 * capA + slot] = lo.y; d2w[2 * capA + slot] = lo.z; d2w[3 * capA + slot] = lo.w;
            d2w[4 * capA + slot] = hi.x; d2w[5 * capA + slot] = hi.y; d2w[6 * capA + slot] = hi.z; d2w[7 * capA + slot] = hi.w;
        }
        n2 += all;
    }
    for (int i = tid; i < cap; i += kThreads) out[i] = -1;     // keypoints outside the grid can never match
    __syncthreads();
    if (tid < kCols + 2) {      // first slot whose cell column is >= tid (slots are sorted by column)
        int lo = 0, hi = n2;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((cell2[mid] >> 8) < tid) lo = mid + 1; else hi = mid; }
        colStart[tid] = lo;
    }
    __syncthreads();

    int nm = 0;
    int histCnt = 0;        // lane b counts rotHist[b].size()
    int parity = 0;
    const float factor = 1.0f / kHistoLength;
    for (int base1 = 0; base1 < NQ; base1 += 64) {
        // one request per lane (every wave holds the same 64)
        const int mine = base1 + lane;
        ProjQuery q{0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0.f};
        uint4 dlo = make_uint4(0, 0, 0, 0), dhi = dlo;
        if (mine < NQ) {
            q = Q[mine];
            dlo = *(const uint4*)(QD + (long long)mine * 8); dhi = *(const uint4*)(QD + (long long)mine * 8 + 4);
        }
        const float r = q.radius;
        // GetFeaturesInArea's cell window (Frame.cc:666-688); an empty window is "no candidates"
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.u, p.minX), r), p.wInv)));
        const int maxCX = min(kCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.u, p.minX), r), p.wInv)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.v, p.minY), r), p.hInv)));
        const int maxCY = min(kRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.v, p.minY), r), p.hInv)));
        const bool window = !(minCX >= kCols || maxCX < 0 || minCY >= kRows || maxCY < 0 || minCX > maxCX || minCY > maxCY);
        const int myBeg = window ? colStart[minCX] : 0, myEnd = window ? colStart[maxCX + 1] : 0;
        const int myCY = (minCY << 8) | (maxCY & 255);
        // level filter of GetFeaturesInArea (:690, :705-712) as an inclusive range; without bCheckLevels everything passes
        const bool checkLevels = q.minLevel > 0 || q.maxLevel >= 0;
        const int loL = checkLevels ? max(q.minLevel, 0) : 0, hiL = checkLevels && q.maxLevel >= 0 ? min(q.maxLevel, 255) : 255;
        const int myLv = (loL << 8) | hiL;
        unsigned long long todo = __ballot((q.flags & 1) && myBeg < myEnd);
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int iq = base1 + j;
            const float x = bcastf(q.u, j), y = bcastf(q.v, j), rr = bcastf(r, j), urq = bcastf(q.ur, j);
            const int sBeg = bcast(myBeg, j), sEnd = bcast(myEnd, j), cyr = bcast(myCY, j), lvr = bcast(myLv, j), qflags = bcast(q.flags, j);
            const int loCY = cyr >> 8, hiCY = cyr & 255, loLv = lvr >> 8, hiLv = lvr & 255;
            const uint32_t w0 = bcast(dlo.x, j), w1 = bcast(dlo.y, j), w2 = bcast(dlo.z, j), w3 = bcast(dlo.w, j);
            const uint32_t w4 = bcast(dhi.x, j), w5 = bcast(dhi.y, j), w6 = bcast(dhi.z, j), w7 = bcast(dhi.w, j);
            int key = kNoneKey, second = kNoneKey;             // key = distance << 16 | slot
            for (int s = sBeg + tid; s < sEnd; s += kThreads) {
                const int cy = cell2[s] & 255, lv = oct2[s];
                const float distx = __fsub_rn(x2[s], x), disty = __fsub_rn(y2[s], y), us = ur2[s];
                const int dist = __popc(w0 ^ d2w[s]) + __popc(w1 ^ d2w[capA + s]) + __popc(w2 ^ d2w[2 * capA + s]) +
                                 __popc(w3 ^ d2w[3 * capA + s]) + __popc(w4 ^ d2w[4 * capA + s]) + __popc(w5 ^ d2w[5 * capA + s]) +
                                 __popc(w6 ^ d2w[6 * capA + s]) + __popc(w7 ^ d2w[7 * capA + s]);
                const bool stereoOut = us > 0.0f && fabsf(__fsub_rn(urq, us)) > rr;                                 // :68-73, :2039-2046
                const bool in = (int)(cy >= loCY) & (int)(cy <= hiCY) & (int)(lv >= loLv) & (int)(lv <= hiLv) &
                                (int)(fabsf(distx) < rr) & (int)(fabsf(disty) < rr) & (int)!occ[s] & (int)!stereoOut;  // Frame.cc:717; :64-66
                if (in) {
                    const int k = (dist << 16) | s;            // slots ascend per lane: a later equal distance never displaces
                    if (k < key) { second = key; key = k; }
                    else if (k < second) second = k;
                }
            }
            const int wkey = waveMin(key);
            const int wsecond = waveMin(key == wkey ? second : key);
            if (lane == 0) { merge[(parity * kWaves + wave) * 2] = wkey; merge[(parity * kWaves + wave) * 2 + 1] = wsecond; }
            __syncthreads();
            int bkey = merge[parity * kWaves * 2], bsecond = merge[parity * kWaves * 2 + 1];
#pragma unroll
            for (int w = 1; w < kWaves; w++) {
                const int ok = merge[(parity * kWaves + w) * 2], os = merge[(parity * kWaves + w) * 2 + 1];
                bsecond = min(max(bkey, ok), min(bsecond, os));
                bkey = min(bkey, ok);
            }
            parity ^= 1;
            const int bestDist = bkey >> 16, bs = bkey & 0xFFFF;
            if (bestDist <= p.maxDist) {                                                                       // :98 / :2058 / :2246
                bool accept = true;
                if (p.ratioMode) {      // only when best and second lie on the same level does the ratio apply (:100-104)
                    const int bestDist2 = bsecond >> 16;
                    const int bestLevel = oct2[bs], bestLevel2 = bestDist2 < 256 ? (int)oct2[bsecond & 0xFFFF] : -1;
                    accept = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(p.nnRatio, (float)bestDist2));
                }
                if (accept) {
                    const float ang2 = a2[bs];
                    // every wave records the occupancy itself (its own next search reads it: no barrier needed); the match
                    // tables belong to wave 0 alone
                    if (lane == 0) occ[bs] = (uint8_t)((qflags >> 1) & 1);
                    if (tid == 0) m2q[bs] = iq;                                                                // F.mvpMapPoints[bestIdx] = pMP
                    nm++;
                    if (!p.ratioMode && p.checkOrientation) {                                                  // :2064-2080
                        float rot = __fsub_rn(bcastf(q.angle, j), ang2);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, factor));
                        if (bin == kHistoLength) bin = 0;
                        if (tid == 0) binMask[bs] |= 1u << bin;
                        histCnt += lane == bin;
                    }
                }
            }
        }
    }
    __syncthreads();

    unsigned dropBins = 0u;
    int droppedCount = 0;
    if (!p.ratioMode && p.checkOrientation) {                                                                  // ComputeThreeMaxima
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < kHistoLength; i++) {
            const int s = bcast(histCnt, i);
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        for (int i = 0; i < kHistoLength; i++)
            if (i != ind1 && i != ind2 && i != ind3) { dropBins |= 1u << i; droppedCount += bcast(histCnt, i); }   // :2166-2170: one nmatches-- per entry
    }
    for (int s = tid; s < n2; s += kThreads) {
        const bool dropped = (binMask[s] & dropBins) != 0u;
        out[idx2[s]] = dropped ? -1 : m2q[s];
        if (occIO) occIO[idx2[s]] = dropped ? (uint8_t)0 : occ[s];
    }
    if (tid == 0) nMatches[pair] = nm - droppedCount;
}

void launchProjectLast(hipStream_t st, const Keypoint* kps, const Keypoint* kpsUn, const int* nOut, const uint8_t* mpFlags,
                       const float* world, const float* poses, const ProjectParams& p, ProjQuery* queries, int nPairs) {
    hipLaunchKernelGGL(k_project_last, dim3((p.capacity + 255) / 256, nPairs), dim3(256), 0, st, kps, kpsUn, nOut, mpFlags, world, poses, p, queries);
}

void launchSearchProj(hipStream_t st, const ProjQuery* queries, const uint8_t* qdesc, const int* nQueries, const Keypoint* kpsUn,
                      const uint8_t* desc, const int* nOut, const int* gridOff, const int* gridIdx, const float* uRight,
                      uint8_t* occupied, const ProjSearchParams& p, int* matches, int* nMatches, int nPairs) {
    hipLaunchKernelGGL(k_search_proj, dim3(nPairs), dim3(kThreads), projSearchLdsBytes(p.capacity), st, queries, qdesc, nQueries, kpsUn,
                       desc, nOut, gridOff, gridIdx, uRight, occupied, p, matches, nMatches);
}

}  // namespace orbx
