// k_match.hip — "next" row SURVEY.md §8f-2: ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:706-821)
// on device-resident frames: mvKeysUn + mDescriptors of both frames and frame 2's mGrid (the CSR k_frame_finish
// leaves), with Frame::GetFeaturesInArea (src/Frame.cc:655-724), ORBmatcher::DescriptorDistance (:2349-2365) and
// ComputeThreeMaxima (:2303-2344).
//
// The reference loop carries state from one keypoint of frame 1 to the next (vMatchedDistance filters later
// searches, vnMatches21 lets a later keypoint steal a match), so one pair is one sequential chain: ONE WORKGROUP (4
// waves) PER PAIR walks frame 1's level-0 keypoints in index order, and its 256 lanes share each search:
//   * frame 2's level-0 keypoints are staged once into LDS in grid order (cell x*48+y, push_back order inside a
//     cell).  GetFeaturesInArea visits cells ix-major / iy-minor, i.e. in ascending grid position, so "first
//     candidate wins a distance tie" (strict < at :748) is "smallest grid position wins", and the cell columns
//     [nMinCellX, nMaxCellX] of a search are one contiguous slot range (colStart[]);
//   * per frame-1 keypoint every lane tests its slots of that range against the cell window (:666-688) and the
//     |dx|,|dy| < r box (:717), computes the 256-bit Hamming distance and applies the vMatchedDistance filter (:744),
//     keeping (best, second best); two DPP min-reductions per wave and a 4-entry merge through LDS give
//     best = min over (distance, position) and second = the second order statistic of the multiset, exactly what the
//     sequential update of :748-757 yields;
//   * the decision (:760-786) is uniform over the workgroup; every wave records vMatchedDistance for its own next
//     search and wave 0 owns the match tables, so one barrier per frame-1 keypoint suffices (the merge scratch is
//     double-buffered).
// Frame-1 data are loaded 64 keypoints at a time, one per lane, and broadcast with v_readlane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

struct InitMatchParams {
    float minX, minY, wInv, hInv;   // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv of frame 2
    float r, nnRatio;               // windowSize as float (Frame.cc:660-661), mfNNratio
    int checkOrientation, capacity;
    int slotCapacity;               // level-0 keypoints of frame 2 the LDS tables hold (a multiple of 4)
    int f1First, f1Step, f2First, f2Step;
};

namespace {
constexpr int kCols = 64, kRows = 48, kCells = kCols * kRows;
constexpr int kThLow = 50, kHistoLength = 30;           // ORBmatcher.cc:37-38
constexpr int kNone = 0x7FFF;                            // "INT_MAX" of the 16-bit distance fields
constexpr int kThreads = 256, kWaves = kThreads / 64;

__device__ __forceinline__ int bcast(int v, int srcLane) { return __builtin_amdgcn_readlane(v, srcLane); }
__device__ __forceinline__ float bcastf(float v, int srcLane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srcLane)); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dppMin(int v) { return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xF, false)); }
// minimum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ int waveMin(int v) {
    v = dppMin<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dppMin<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dppMin<0x141, 0xF>(v);    // row_half_mirror
    v = dppMin<0x140, 0xF>(v);    // row_mirror: every lane of a row holds the row's minimum
    v = dppMin<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v = dppMin<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
}  // namespace

// LDS: per staged (level-0) keypoint of frame 2 its descriptor, position, angle, cell, index, matched distance and back pointer (52 B);
// per keypoint of frame 1 vnMatches12 and its rotHist bin (4 B).  Only level-0 keypoints are candidates (:722-726), about a fifth of a
// frame, so the slot tables are sized apart from `capacity`: the initialisation extractor's frames (ORBextractor(5 * nFeatures),
// Tracking.cc:774: 5000-10 000 keypoints) fit although 56 B x capacity would not.
size_t initMatchLdsBytes(int capacity, int slotCapacity) {
    const size_t c = (size_t)((capacity + 3) & ~3), sc = (size_t)((slotCapacity + 3) & ~3);
    return sc * (32 + 4 + 4 + 4 + 2 + 2 + 2 + 2) + c * (2 + 2) + (kCols + 2) * sizeof(int) + 2 * kWaves * 2 * sizeof(int) + 2 * kWaves * sizeof(int) + 16;
}
// the largest slot table that fits next to the frame-1 tables (0: not even those fit)
int initMatchSlotCapacity(int capacity) {
    const long long fixed = (long long)((capacity + 3) & ~3) * 4 + (kCols + 2) * sizeof(int) + 4 * kWaves * sizeof(int) + 2 * kWaves * sizeof(int) + 16;
    const long long room = 160LL * 1024 - 512 - fixed;
    if (room < 52 * 64) return 0;
    const long long sc = room / 52;
    return (int)(sc < capacity ? sc & ~3LL : (capacity + 3) & ~3);
}

// grid: n_pairs; 256 threads.
__global__ __launch_bounds__(kThreads) void k_search_init(const Keypoint* __restrict__ kpsUn, const uint8_t* __restrict__ desc,
                                                          const int* __restrict__ nOut, const int* __restrict__ gridOff,
                                                          const int* __restrict__ gridIdx, InitMatchParams p,
                                                          float* __restrict__ prevMatched, int* __restrict__ matches12,
                                                          int* __restrict__ nMatches) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int cap = p.capacity, capA = p.slotCapacity, capF = (cap + 3) & ~3;      // capA: slots (level-0 keypoints of frame 2); capF: keypoints of frame 1
    uint32_t* d2w = (uint32_t*)smem;                       // [8][capA] descriptor word k of slot s
    float* x2 = (float*)(d2w + 8 * capA);                  // [capA]
    float* y2 = x2 + capA;
    float* a2 = y2 + capA;                                 // angle
    int* colStart = (int*)(a2 + capA);                     // [66] first slot of cell column c (c = 64, 65: n2)
    int* merge = colStart + kCols + 2;                     // [2][kWaves][2] per-wave (key, second), double-buffered
    int* wcnt = merge + 2 * kWaves * 2;                    // [2][kWaves] staging counts
    unsigned short* cell2 = (unsigned short*)(wcnt + 2 * kWaves);  // ix << 8 | iy
    unsigned short* idx2 = cell2 + capA;                   // keypoint index in frame 2
    unsigned short* mdist = idx2 + capA;                   // vMatchedDistance (kNone = INT_MAX)
    short* m21 = (short*)(mdist + capA);                   // vnMatches21
    short* m12 = m21 + capA;                               // [capF] vnMatches12 (frame 1 index space)
    short* rbin = m12 + capF;                              // [capF] rotHist bin the keypoint was pushed to, -1 = none

    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f1 = p.f1First + pair * p.f1Step, f2 = p.f2First + pair * p.f2Step;
    const int N1 = min(nOut[f1], cap), N2 = min(nOut[f2], cap);
    const Keypoint* K1 = kpsUn + (long long)f1 * cap;
    const Keypoint* K2 = kpsUn + (long long)f2 * cap;
    const uint32_t* D1 = (const uint32_t*)(desc + (long long)f1 * cap * 32);
    const uint32_t* D2 = (const uint32_t*)(desc + (long long)f2 * cap * 32);
    const int* off2 = gridOff + (long long)f2 * (kCells + 1);
    const int* gi2 = gridIdx + (long long)f2 * cap;
    float* prev = prevMatched + (long long)pair * cap * 2;
    const int nIn2 = min(off2[kCells], N2);

    // ---- stage frame 2's level-0 keypoints in grid order (level1 is always 0: ORBmatcher.cc:722-726) ----
    int n2 = 0;
    for (int base = 0, it = 0; base < nIn2; base += kThreads, it ^= 1) {
        const int pos = base + tid;
        bool keep = false;
        int i2 = 0;
        Keypoint k{};
        if (pos < nIn2) { i2 = gi2[pos]; k = K2[i2]; keep = k.octave == 0; }      // Frame.cc:705-712 with minLevel = maxLevel = 0
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcnt[it * kWaves + wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { const int c = wcnt[it * kWaves + w]; all += c; before += w < wave ? c : 0; }
        const int slot = n2 + before + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && slot < capA) {
            const int posX = (int)roundf(__fmul_rn(__fsub_rn(k.x, p.minX), p.wInv));   // the cell AssignFeaturesToGrid put it in
            const int posY = (int)roundf(__fmul_rn(__fsub_rn(k.y, p.minY), p.hInv));   // (PosInGrid, Frame.cc:728-729)
            x2[slot] = k.x; y2[slot] = k.y; a2[slot] = k.angle;
            cell2[slot] = (unsigned short)((posX << 8) | posY);
            idx2[slot] = (unsigned short)i2;
            mdist[slot] = kNone; m21[slot] = -1;
            const uint4 lo = *(const uint4*)(D2 + (long long)i2 * 8), hi = *(const uint4*)(D2 + (long long)i2 * 8 + 4);
            d2w[0 * capA + slot] = lo.x; d2w[1 * capA + slot] = lo.y; d2w[2 * capA + slot] = lo.z; d2w[3 * capA + slot] = lo.w;
            d2w[4 * capA + slot] = hi.x; d2w[5 * capA + slot] = hi.y; d2w[6 * capA + slot] = hi.z; d2w[7 * capA + slot] = hi.w;
        }
        n2 += all;
    }
    if (n2 > capA) {      // more level-0 keypoints than the slot tables hold (not with a pyramid of several levels): report, match nothing
        for (int i = tid; i < N1; i += kThreads) matches12[(long long)pair * cap + i] = -1;
        if (tid == 0) nMatches[pair] = -1;
        return;
    }
    for (int i = tid; i < N1; i += kThreads) { m12[i] = -1; rbin[i] = -1; }
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
    for (int base1 = 0; base1 < N1; base1 += 64) {
        // one frame-1 keypoint per lane (every wave holds the same 64): octave, angle, search centre, descriptor
        const int mine = base1 + lane;
        int oct = 1;
        float ang1 = 0.f, px = 0.f, py = 0.f;
        uint4 dlo = make_uint4(0, 0, 0, 0), dhi = dlo;
        if (mine < N1) {
            const Keypoint k = K1[mine];
            oct = k.octave; ang1 = k.angle;
            px = prev[2 * mine]; py = prev[2 * mine + 1];
            dlo = *(const uint4*)(D1 + (long long)mine * 8); dhi = *(const uint4*)(D1 + (long long)mine * 8 + 4);
        }
        // GetFeaturesInArea's cell window (Frame.cc:666-688) of this lane's keypoint; an empty window is "no candidates"
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(px, p.minX), p.r), p.wInv)));
        const int maxCX = min(kCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(px, p.minX), p.r), p.wInv)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(py, p.minY), p.r), p.hInv)));
        const int maxCY = min(kRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(py, p.minY), p.r), p.hInv)));
        const bool window = !(minCX >= kCols || maxCX < 0 || minCY >= kRows || maxCY < 0 || minCX > maxCX || minCY > maxCY);
        const int myBeg = window ? colStart[minCX] : 0, myEnd = window ? colStart[maxCX + 1] : 0;   // the columns' slot range
        const int myCY = (minCY << 8) | (maxCY & 255);
        unsigned long long todo = __ballot(oct <= 0 && myBeg < myEnd);                  // level1 > 0: continue (:722-723)
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int i1 = base1 + j;
            const float x = bcastf(px, j), y = bcastf(py, j);
            const int sBeg = bcast(myBeg, j), sEnd = bcast(myEnd, j), cyr = bcast(myCY, j);
            const int loCY = cyr >> 8, hiCY = cyr & 255;
            const uint32_t w0 = bcast(dlo.x, j), w1 = bcast(dlo.y, j), w2 = bcast(dlo.z, j), w3 = bcast(dlo.w, j);
            const uint32_t w4 = bcast(dhi.x, j), w5 = bcast(dhi.y, j), w6 = bcast(dhi.z, j), w7 = bcast(dhi.w, j);
            int key = (kNone << 16) | 0xFFFF, second = kNone;     // key = best distance << 16 | slot
            for (int s = sBeg + tid; s < sEnd; s += kThreads) {
                const int cy = cell2[s] & 255, md = mdist[s];
                const float distx = __fsub_rn(x2[s], x), disty = __fsub_rn(y2[s], y);
                const int dist = __popc(w0 ^ d2w[s]) + __popc(w1 ^ d2w[capA + s]) + __popc(w2 ^ d2w[2 * capA + s]) +
                                 __popc(w3 ^ d2w[3 * capA + s]) + __popc(w4 ^ d2w[4 * capA + s]) + __popc(w5 ^ d2w[5 * capA + s]) +
                                 __popc(w6 ^ d2w[6 * capA + s]) + __popc(w7 ^ d2w[7 * capA + s]);
                const bool in = (int)(cy >= loCY) & (int)(cy <= hiCY) & (int)(fabsf(distx) < p.r) & (int)(fabsf(disty) < p.r)    // Frame.cc:717
                                & (int)!(md <= dist);                                                            // :744-745
                if (in) {
                    if (dist < (key >> 16)) { second = key >> 16; key = (dist << 16) | s; }   // :747-752 (slots ascend per lane)
                    else if (dist < second) second = dist;                                     // :753-756
                }
            }
            // wave: best = min key; second = min over the other candidates (the winner lane contributes its own second)
            const int wkey = waveMin(key);
            const int wsecond = waveMin(key == wkey ? second : (key >> 16));
            if (lane == 0) { merge[(parity * kWaves + wave) * 2] = wkey; merge[(parity * kWaves + wave) * 2 + 1] = wsecond; }
            __syncthreads();
            int bkey = merge[parity * kWaves * 2], bsecond = merge[parity * kWaves * 2 + 1];
#pragma unroll
            for (int w = 1; w < kWaves; w++) {
                const int ok = merge[(parity * kWaves + w) * 2], os = merge[(parity * kWaves + w) * 2 + 1];
                bsecond = min(max(bkey >> 16, ok >> 16), min(bsecond, os));
                bkey = min(bkey, ok);
            }
            parity ^= 1;
            const int bestDist = bkey >> 16, bs = bkey & 0xFFFF;
            if (bestDist <= kThLow) {                                                                          // :759
                const float second_f = bsecond == kNone ? 2147483648.0f : (float)bsecond;                      // (float)INT_MAX
                if ((float)bestDist < __fmul_rn(second_f, p.nnRatio)) {                                        // :761
                    const int i2 = idx2[bs], old = m21[bs];
                    const float ang2 = a2[bs];
                    // every wave records the matched distance itself (its next search reads it: no barrier needed); the
                    // match tables belong to wave 0 alone, so its read of vnMatches21 never sees another wave's update
                    if (lane == 0) mdist[bs] = (unsigned short)bestDist;                                       // :770
                    if (tid == 0) {
                        if (old >= 0) m12[old] = -1;                                                           // :763-767
                        m12[i1] = (short)i2; m21[bs] = (short)i1;                                              // :768-769
                    }
                    nm += 1 - (old >= 0);                                                                      // (wave 0's count is the result)
                    if (p.checkOrientation) {                                                                  // :773-783
                        float rot = __fsub_rn(bcastf(ang1, j), ang2);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, factor));
                        if (bin == kHistoLength) bin = 0;
                        if (tid == 0) rbin[i1] = (short)bin;
                        histCnt += lane == bin;
                    }
                }
            }
        }
    }
    __syncthreads();

    int ind1 = -1, ind2 = -1, ind3 = -1;
    if (p.checkOrientation) {                                                                                  // ComputeThreeMaxima
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < kHistoLength; i++) {
            const int s = bcast(histCnt, i);
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
    }
    int dropped = 0;
    int* out = matches12 + (long long)pair * cap;
    for (int i = tid; i < N1; i += kThreads) {
        int m = m12[i];
        const int b = rbin[i];
        if (p.checkOrientation && b >= 0 && b != ind1 && b != ind2 && b != ind3 && m >= 0) { m = -1; dropped++; }   // :797-808
        out[i] = m;
        if (m >= 0) { const Keypoint k = K2[m]; prev[2 * i] = k.x; prev[2 * i + 1] = k.y; }                        // :815-817
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dropped += __shfl_xor(dropped, o);
    if (lane == 0) wcnt[wave] = dropped;
    __syncthreads();
    if (tid == 0) nMatches[pair] = nm - (wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
}

void launchSearchInit(hipStream_t st, const Keypoint* kpsUn, const uint8_t* desc, const int* nOut, const int* gridOff,
                      const int* gridIdx, const InitMatchParams& p, float* prevMatched, int* matches12, int* nMatches, int nPairs) {
    hipLaunchKernelGGL(k_search_init, dim3(nPairs), dim3(kThreads), initMatchLdsBytes(p.capacity, p.slotCapacity), st, kpsUn, desc, nOut, gridOff,
                       gridIdx, p, prevMatched, matches12, nMatches);
}

}  // namespace orbx
