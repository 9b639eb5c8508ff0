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
// (:64-66, :2035-2037), so the requests of one frame form a sequential chain in the reference — but a request only looks at the few
// keypoints of its window, so ONE WORKGROUP PER FRAME settles all requests as a parallel fixed point (see the kernel), one request per
// thread per round:
//   * the current frame's keypoints are staged once into LDS in mGrid's CSR order (cell x*48+y, push_back order inside a cell) with
//     descriptor words transposed; GetFeaturesInArea visits cells in ascending grid position, so the strict "<" of the
//     reference's running minimum is "smallest (distance, slot)" and a window is one slot range per cell column;
//   * a request keeps its two smallest keys (distance << 16 | slot); the reference's (bestDist, bestLevel, bestDist2, bestLevel2)
//     after its sequential scan are the smallest and the second smallest key of the window (the element that ends up providing
//     bestDist2 is the first in traversal order among those with the second smallest distance — see docs/history/DESIGN_rounds_1-5.md §4f).
// One frame pair: 868 us as a walk (one barrier per request) -> see docs/history/DESIGN_rounds_1-5.md §4f for the fixed point's figure.
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
constexpr int kTop = 4;                                   // best keys a request remembers between rounds
constexpr unsigned kNoDecision = 0xFFFFFFFFu;             // the request is inactive, finds nothing or is rejected
constexpr int kThreads = 1024;     // one request per thread per round at ~1000 requests; 16 waves hide the LDS latency of the window scans (256 threads: 111 us for the first scan)

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

size_t projSearchLdsBytes(int capacity, int queryCapacity, bool topList) {
    const size_t c = (size_t)((capacity + 3) & ~3);
    return c * (32 + 16 + 4 + 4 + 2 + 1 + 1 + 2) + (kHistoLength + 4) * sizeof(int) + (kCells + 2) * sizeof(unsigned short) +
           (size_t)queryCapacity * (4 + 1 + (topList ? 4 * kTop : 0)) + 64;
}
__device__ int g_searchRounds[4];      // diagnostics: rounds the last launch's pair 0 needed (projection search, initialisation search)
extern "C" int orbx_debug_search_rounds(int* out4) { return (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_searchRounds), sizeof(int) * 4); }

// grid: n_pairs; 256 threads.  TOPLIST: the requests' best-key lists fit in LDS next to the staged frame.
template <bool TOPLIST>
__global__ __launch_bounds__(kThreads) void k_search_proj(const ProjQuery* __restrict__ queries, const uint8_t* __restrict__ qdesc,
                                                          const int* __restrict__ nQueries, const Keypoint* __restrict__ kpsUn,
                                                          const uint8_t* __restrict__ desc, const int* __restrict__ nOut,
                                                          const int* __restrict__ gridOff, const int* __restrict__ gridIdx,
                                                          const float* __restrict__ uRight, uint8_t* __restrict__ occupied,
                                                          ProjSearchParams p, int* __restrict__ matches, int* __restrict__ nMatches) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int cap = p.capacity, capA = (cap + 3) & ~3;
    uint4* d2 = (uint4*)smem;                              // [capA][2] descriptor of slot s (two 128-bit LDS reads per candidate)
    float4* rec = (float4*)(d2 + 2 * capA);                // [capA] {x, y, mvuRight (<= 0: no stereo observation), octave as bits}: one read per visited slot
    float* a2 = (float*)(rec + capA);                      // [capA] angle
    int* m2q = (int*)d2;                                   // (after the rounds, over the then dead descriptors) query whose MapPoint the keypoint holds, -1 = none
    unsigned* binMask = (unsigned*)(m2q + capA);           // (same) rotHist bins the keypoint was pushed to
    int* closedBy = (int*)(a2 + capA);                    // [capA] first request that closes the keypoint (-1: closed on entry, INT_MAX: nobody)
    int* top = closedBy + capA;                            // [queryCapacity][kTop] (TOPLIST; 16-byte aligned) the requests' smallest keys, ascending
    unsigned* dec = (unsigned*)(top + (TOPLIST ? (long long)p.queryCapacity * kTop : 0));      // [queryCapacity] decision of every request: slot | closes << 16, or kNoDecision
    int* hist = (int*)(dec + p.queryCapacity);             // [30] rotHist sizes
    int* flags = hist + kHistoLength;                      // [3] "a decision changed" (two alternating slots), number of accepted requests
    unsigned short* cellOff = (unsigned short*)(flags + 4);      // [kCells + 2] slot range of every grid cell (mGrid's CSR offsets)
    unsigned short* idx2 = cellOff + kCells + 2;           // keypoint index in the frame
    uint8_t* oct2 = (uint8_t*)(idx2 + capA);               // octave
    uint8_t* occ = oct2 + capA;                            // holds a MapPoint with Observations() > 0
    uint8_t* qflag = occ + capA;                           // [queryCapacity] orbx_proj_query::flags

    const int pair = blockIdx.x, tid = threadIdx.x;
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
    const unsigned long long tStart = __builtin_amdgcn_s_memrealtime();
    const int n2 = nIn2;                                   // slot = position in mGrid's CSR order (cell x*48+y, push_back order inside a cell)
    for (int slot = tid; slot < n2; slot += kThreads) {
        const int i2 = min(max(gi2[slot], 0), cap - 1);      // (clamped: a corrupt grid must not index past the frame)
        const Keypoint k = K2[i2];
        const int oct = min(max(k.octave, 0), 255);
        rec[slot] = make_float4(k.x, k.y, UR ? UR[i2] : -1.0f, __int_as_float(oct));
        a2[slot] = k.angle;
        idx2[slot] = (unsigned short)i2;
        oct2[slot] = (uint8_t)oct;
        occ[slot] = occIO ? occIO[i2] : (uint8_t)0;
        const uint4 lo = *(const uint4*)(D2 + (long long)i2 * 8), hi = *(const uint4*)(D2 + (long long)i2 * 8 + 4);
        d2[2 * slot] = lo; d2[2 * slot + 1] = hi;
    }
    for (int i = tid; i < cap; i += kThreads) out[i] = -1;     // keypoints outside the grid can never match
    for (int c = tid; c <= kCells; c += kThreads) cellOff[c] = (unsigned short)min(off2[c], n2);      // mGrid's CSR offsets: slot range of every cell
    for (int i = tid; i < NQ; i += kThreads) dec[i] = kNoDecision;
    if (tid < kHistoLength) hist[tid] = 0;
    if (tid == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; }
    __syncthreads();
    for (int s = tid; s < n2; s += kThreads) closedBy[s] = occ[s] ? -1 : 0x7fffffff;
    __syncthreads();

    // ---- the searches.  In the reference request i sees the keypoints that requests j < i closed (:64-66, :2035-2037), a sequential
    //      chain; but a request only ever looks at the handful of keypoints in its window, so almost all of the chain is independent.
    //      Fixed point instead of a walk: every request decides in parallel against closedBy[s] = the FIRST request that closes keypoint
    //      s under the current decisions (request i sees s closed iff closedBy[s] < i), closedBy is rebuilt, and the round repeats until
    //      no decision changes.  By induction the decisions of requests 0..k are final after round k + 1 (whether some j < i closes s
    //      depends only on decisions of requests < i), so the fixed point IS the sequential result; real frames settle in 3-6 rounds
    //      instead of a thousand dependent steps. ----
    const unsigned long long tStaged = __builtin_amdgcn_s_memrealtime();
    const float factor = 1.0f / kHistoLength;
    // the kTop smallest keys (distance << 16 | slot) of a request's window, ascending, among the keypoints that pass the static tests
    // (cell window, level, box, stereo column, not closed on entry) and - if `live` - are not closed for this request right now
    auto scan = [&](int iq, bool live, int (&keys)[kTop]) -> int {
#pragma unroll
        for (int k = 0; k < kTop; k++) keys[k] = kNoneKey;
        const ProjQuery q = Q[iq];
        if (!(q.flags & 1)) return 0;
        const float r = q.radius;
        // GetFeaturesInArea's cell window (Frame.cc:666-688); an empty window is "no candidates"
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.u, p.minX), r), p.wInv)));
        const int maxCX = min(kCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.u, p.minX), r), p.wInv)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.v, p.minY), r), p.hInv)));
        const int maxCY = min(kRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.v, p.minY), r), p.hInv)));
        if (minCX >= kCols || maxCX < 0 || minCY >= kRows || maxCY < 0 || minCX > maxCX || minCY > maxCY) return q.flags;
        // level filter of GetFeaturesInArea (:690, :705-712) as an inclusive range; without bCheckLevels everything passes
        const bool checkLevels = q.minLevel > 0 || q.maxLevel >= 0;
        const int loLv = checkLevels ? max(q.minLevel, 0) : 0, hiLv = checkLevels && q.maxLevel >= 0 ? min(q.maxLevel, 255) : 255;
        const uint4 dlo = *(const uint4*)(QD + (long long)iq * 8), dhi = *(const uint4*)(QD + (long long)iq * 8 + 4);
        for (int cx = minCX; cx <= maxCX; cx++) {          // ascending cells, push_back order inside a cell = ascending slots (Frame.cc:690-720)
            const int sEnd = cellOff[cx * kRows + maxCY + 1];
            for (int s = cellOff[cx * kRows + minCY]; s < sEnd; s++) {
                const float4 rc = rec[s];
                const int lv = __float_as_int(rc.w);
                const float distx = __fsub_rn(rc.x, q.u), disty = __fsub_rn(rc.y, q.v), us = rc.z;
                const bool stereoOut = us > 0.0f && fabsf(__fsub_rn(q.ur, us)) > r;                                    // :68-73, :2039-2046
                const int cb = closedBy[s];
                const bool in = (int)(lv >= loLv) & (int)(lv <= hiLv) & (int)(fabsf(distx) < r) & (int)(fabsf(disty) < r) &
                                (int)(live ? !(cb < iq) : cb != -1) & (int)!stereoOut;                                 // Frame.cc:717; :64-66
                if (in) {
                    const uint4 e = d2[2 * s], f = d2[2 * s + 1];
                    const int dist = __popc(dlo.x ^ e.x) + __popc(dlo.y ^ e.y) + __popc(dlo.z ^ e.z) + __popc(dlo.w ^ e.w) + __popc(dhi.x ^ f.x) +
                                     __popc(dhi.y ^ f.y) + __popc(dhi.z ^ f.z) + __popc(dhi.w ^ f.w);
                    int k = (dist << 16) | s;              // slots ascend: a later equal distance never displaces
#pragma unroll
                    for (int t = 0; t < kTop; t++) { const int lo = min(keys[t], k); k = max(keys[t], k); keys[t] = lo; }      // sorted insert
                }
            }
        }
        return q.flags;
    };
    // the reference's acceptance tests on (best, second best) of the keypoints a request sees
    auto judge = [&](int key, int second, int qflags) -> unsigned {
        const int bestDist = key >> 16, bs = key & 0xFFFF;
        if (key == kNoneKey || bestDist > p.maxDist) return kNoDecision;                                           // :98 / :2058 / :2246
        if (p.ratioMode) {      // only when best and second lie on the same level does the ratio apply (:100-104)
            const int bestDist2 = second >> 16;
            const int bestLevel = oct2[bs], bestLevel2 = bestDist2 < 256 ? (int)oct2[second & 0xFFFF] : -1;
            if (bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(p.nnRatio, (float)bestDist2)) return kNoDecision;
        }
        return (unsigned)bs | (((unsigned)qflags >> 1) & 1u) << 16;      // accepted: keypoint slot, bit 16 = the MapPoint closes it
    };
    // round 0: every request scans its window once and keeps its kTop best keys; later rounds only look at those (a full re-scan
    // is needed only if closed keypoints have used up a truncated list)
    for (int iq = tid; iq < NQ; iq += kThreads) {
        int keys[kTop];
        const int qf = scan(iq, false, keys);
        qflag[iq] = (uint8_t)qf;
        if (TOPLIST) *(int4*)(top + (long long)iq * kTop) = make_int4(keys[0], keys[1], keys[2], keys[3]);
        dec[iq] = (qf & 1) ? judge(keys[0], keys[1], qf) : kNoDecision;
    }
    __syncthreads();
    const unsigned long long tScanned = __builtin_amdgcn_s_memrealtime();
    int rounds = 1;
    for (int round = 1; round <= NQ + 1; round++, rounds++) {
        // closedBy[s] = first request that closes keypoint s under the current decisions
        for (int iq = tid; iq < NQ; iq += kThreads) {
            const unsigned d = dec[iq];
            if (d != kNoDecision && (d >> 16)) atomicMin(&closedBy[d & 0xFFFFu], iq);
        }
        __syncthreads();
        bool mineChanged = false;
        for (int iq = tid; iq < NQ; iq += kThreads) {
            const int qf = qflag[iq];
            if (!(qf & 1)) continue;
            int key = kNoneKey, second = kNoneKey;
            bool rescan = !TOPLIST;
            if (TOPLIST) {
                const int4 t4 = *(const int4*)(top + (long long)iq * kTop);
                const int t[kTop] = {t4.x, t4.y, t4.z, t4.w};
                int nvis = 0;
#pragma unroll
                for (int k = 0; k < kTop; k++)
                    if (t[k] != kNoneKey && !(closedBy[t[k] & 0xFFFF] < iq)) { if (nvis == 0) key = t[k]; else if (nvis == 1) second = t[k]; nvis++; }
                rescan = t[kTop - 1] != kNoneKey && nvis < (p.ratioMode ? 2 : 1);      // truncated list used up
            }
            if (rescan) {
                int keys[kTop];
                scan(iq, true, keys);
                key = keys[0]; second = keys[1];
            }
            const unsigned d = judge(key, second, qf);
            if (d != dec[iq]) { dec[iq] = d; mineChanged = true; }
        }
        if (mineChanged) flags[round & 1] = 1;
        __syncthreads();
        const bool any = flags[round & 1] != 0;
        if (tid == 0) flags[(round & 1) ^ 1] = 0;          // the other slot is read again only after the next barriers
        if (!any) break;
        for (int s = tid; s < n2; s += kThreads) closedBy[s] = occ[s] ? -1 : 0x7fffffff;
        __syncthreads();
    }
    if (tid == 0 && pair == 0) {      // diagnostics (100 MHz ticks): staging, first scan, rounds
        g_searchRounds[0] = rounds; g_searchRounds[1] = (int)(tStaged - tStart); g_searchRounds[2] = (int)(tScanned - tStaged);
        g_searchRounds[3] = (int)(__builtin_amdgcn_s_memrealtime() - tScanned);
    }
    // ---- the tables the walk would have left: F.mvpMapPoints[bestIdx] = pMP is overwritten by every later accepted request
    //      (a MapPoint without observations does not close its keypoint), nmatches and rotHist count every acceptance ----
    for (int s = tid; s < n2; s += kThreads) { m2q[s] = -1; binMask[s] = 0u; }      // (nobody reads the descriptor words any more)
    __syncthreads();
    int nm = 0;
    for (int iq = tid; iq < NQ; iq += kThreads) {
        const unsigned d = dec[iq];
        if (d == kNoDecision) continue;
        const int bs = (int)(d & 0xFFFFu);
        atomicMax(&m2q[bs], iq);
        nm++;
        if (!p.ratioMode && p.checkOrientation) {                                                                  // :2064-2080
            float rot = __fsub_rn(Q[iq].angle, a2[bs]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == kHistoLength) bin = 0;
            atomicOr(&binMask[bs], 1u << bin);
            atomicAdd(&hist[bin], 1);
        }
    }
    if (nm) atomicAdd(&flags[2], nm);
    __syncthreads();

    unsigned dropBins = 0u;
    int droppedCount = 0;
    if (!p.ratioMode && p.checkOrientation) {                                                                  // ComputeThreeMaxima
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < kHistoLength; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        for (int i = 0; i < kHistoLength; i++)
            if (i != ind1 && i != ind2 && i != ind3) { dropBins |= 1u << i; droppedCount += hist[i]; }             // :2166-2170: one nmatches-- per entry
    }
    for (int s = tid; s < n2; s += kThreads) {
        const bool dropped = (binMask[s] & dropBins) != 0u;
        out[idx2[s]] = dropped ? -1 : m2q[s];
        if (occIO) occIO[idx2[s]] = dropped ? (uint8_t)0 : (uint8_t)(closedBy[s] != 0x7fffffff);
    }
    if (tid == 0) nMatches[pair] = flags[2] - droppedCount;
}

void launchProjectLast(hipStream_t st, const Keypoint* kps, const Keypoint* kpsUn, const int* nOut, const uint8_t* mpFlags,
                       const float* world, const float* poses, const ProjectParams& p, ProjQuery* queries, int nPairs) {
    hipLaunchKernelGGL(k_project_last, dim3((p.capacity + 255) / 256, nPairs), dim3(256), 0, st, kps, kpsUn, nOut, mpFlags, world, poses, p, queries);
}

void launchSearchProj(hipStream_t st, const ProjQuery* queries, const uint8_t* qdesc, const int* nQueries, const Keypoint* kpsUn,
                      const uint8_t* desc, const int* nOut, const int* gridOff, const int* gridIdx, const float* uRight,
                      uint8_t* occupied, const ProjSearchParams& p, int* matches, int* nMatches, int nPairs) {
    // the best-key lists ride in LDS when they fit (640x480 x 1000 features: 103 KB); otherwise every round re-scans the windows
    if (projSearchLdsBytes(p.capacity, p.queryCapacity, true) <= 160 * 1024 - 512)
        hipLaunchKernelGGL(k_search_proj<true>, dim3(nPairs), dim3(kThreads), projSearchLdsBytes(p.capacity, p.queryCapacity, true), st, queries, qdesc,
                           nQueries, kpsUn, desc, nOut, gridOff, gridIdx, uRight, occupied, p, matches, nMatches);
    else
        hipLaunchKernelGGL(k_search_proj<false>, dim3(nPairs), dim3(kThreads), projSearchLdsBytes(p.capacity, p.queryCapacity, false), st, queries, qdesc,
                           nQueries, kpsUn, desc, nOut, gridOff, gridIdx, uRight, occupied, p, matches, nMatches);
}

}  // namespace orbx
