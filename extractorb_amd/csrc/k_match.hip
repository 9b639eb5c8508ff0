// k_match.hip — "next" row SURVEY.md §8f-2: ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:706-821)
// on device-resident frames: mvKeysUn + mDescriptors of both frames and frame 2's mGrid (the CSR k_frame_finish
// leaves), with Frame::GetFeaturesInArea (src/Frame.cc:655-724), ORBmatcher::DescriptorDistance (:2349-2365) and
// ComputeThreeMaxima (:2303-2344).
//
// The reference loop carries state from one keypoint of frame 1 to the next: vMatchedDistance[i2] (the distance of the match a
// keypoint of frame 2 holds) hides it from later keypoints whose distance is not smaller (:744), and vnMatches21 lets a later keypoint
// steal a match (:763-767).  Round 1 walked that chain (one window search shared by 256 lanes + one barrier per keypoint: 201 us for
// one pair).  A keypoint only looks at the level-0 keypoints of frame 2 inside its window, so — as in k_search_proj (k_project.hip) —
// the chain is settled as a PARALLEL FIXED POINT, one request (level-0 keypoint of frame 1, in index order) per thread per round:
//   * frame 2's level-0 keypoints are staged once into LDS in grid order (cell x*48+y, push_back order inside a cell).
//     GetFeaturesInArea visits cells ix-major / iy-minor, i.e. in ascending grid position, so a request's sequential scan over its
//     slot range IS the reference's traversal: "first candidate wins a distance tie" (strict < at :748) needs no key trick;
//   * every round each request decides (best slot, best distance) against md(s, r) = the smallest distance among the requests
//     r' < r that currently hold slot s — in the reference vMatchedDistance[s] only ever decreases, so the value request r sees is that
//     minimum; the holders of a slot are kept as a short list per slot (rebuilt every round; a slot with more holders than the list
//     has room for is answered by a scan over the decisions).  Decisions of requests 0..k are final after round k + 1, so the fixed
//     point is the sequential result; frames settle in a handful of rounds;
//   * the tables the walk would have left follow from the final decisions: a slot belongs to its LAST holder (earlier ones were
//     stolen from: vnMatches12 = -1, nmatches--), rotHist holds every acceptance (:773-783), its clean-up skips stolen entries (:797-808).
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
constexpr int kThreads = 1024, kWaves = kThreads / 64;
constexpr int kAcc = 2;                                  // holders of a slot kept per round (more: answered by a scan of the decisions)
constexpr unsigned kNoDec = 0xFFFFFFFFu;

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

// LDS: per staged (level-0) keypoint of frame 2 its descriptor, position, cell, index and the holder list (32 + 8 + 2 + 2 + 4 + 8 = 56 B);
// per request (level-0 keypoint of frame 1) its index and two decision words (10 B).  Only level-0 keypoints take part (:722-726), about a
// fifth of a frame, so both tables are sized apart from `capacity`: the initialisation extractor's frames (ORBextractor(5 * nFeatures),
// Tracking.cc:774: 5000-10 000 keypoints) fit although 66 B x capacity would not.
size_t initMatchLdsBytes(int capacity, int slotCapacity) {
    (void)capacity;
    const size_t sc = (size_t)((slotCapacity + 3) & ~3);
    return sc * (32 + 8 + 2 + 2 + 4 + 4 * kAcc + 2 + 4 + 4) + (kCols + 2 + kHistoLength + 8) * sizeof(int) + 2 * kWaves * sizeof(int) + 64;
}
// the largest table (slots = requests) that fits a workgroup's LDS
int initMatchSlotCapacity(int capacity) {
    const long long room = 160LL * 1024 - 1024 - (long long)((kCols + 2 + kHistoLength + 8) * sizeof(int) + 2 * kWaves * sizeof(int) + 64);
    const long long sc = room / (32 + 8 + 2 + 2 + 4 + 4 * kAcc + 2 + 4 + 4);
    return (int)(sc < capacity ? sc & ~3LL : (capacity + 3) & ~3);
}

// grid: n_pairs; 1024 threads.
__global__ __launch_bounds__(kThreads) void k_search_init(const Keypoint* __restrict__ kpsUn, const uint8_t* __restrict__ desc,
                                                          const int* __restrict__ nOut, const int* __restrict__ gridOff,
                                                          const int* __restrict__ gridIdx, InitMatchParams p,
                                                          float* __restrict__ prevMatched, int* __restrict__ matches12,
                                                          int* __restrict__ nMatches) {
    extern __shared__ __align__(16) uint8_t smem[];
    const int cap = p.capacity, capA = p.slotCapacity;     // capA: slots (level-0 keypoints of frame 2) and requests (of frame 1)
    uint4* d2 = (uint4*)smem;                              // [capA][2] descriptor of slot s
    float2* xy2 = (float2*)(d2 + 2 * capA);                // [capA] position
    unsigned* acc = (unsigned*)(xy2 + capA);               // [capA][kAcc] holders of slot s this round: request << 16 | distance
    int* accCnt = (int*)(acc + kAcc * capA);               // [capA] number of holders (may exceed kAcc); after the rounds: the slot's last holder
    unsigned* decA = (unsigned*)(accCnt + capA);           // [capA] decision of request r: distance << 16 | slot, or kNoDec
    unsigned* decB = decA + capA;                          // [capA] (the other buffer: decisions are read by others while new ones are written)
    int* colStart = (int*)(decB + capA);                   // [66] first slot of cell column c (c = 64, 65: n2)
    int* hist = colStart + kCols + 2;                      // [30] rotHist sizes
    int* flags = hist + kHistoLength;                      // [8] changed (two alternating), matches, dropped, requests
    int* wcnt = flags + 8;                                 // [2][kWaves] compaction counts
    unsigned short* cell2 = (unsigned short*)(wcnt + 2 * kWaves);  // [capA] ix << 8 | iy
    unsigned short* idx2 = cell2 + capA;                   // [capA] keypoint index in frame 2
    unsigned short* req = idx2 + capA;                     // [capA] keypoint index in frame 1 of request r

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
    int* out = matches12 + (long long)pair * cap;
    const int nIn2 = min(off2[kCells], N2);
    if (tid < kHistoLength) hist[tid] = 0;
    if (tid < 8) flags[tid] = 0;

    // stable compaction of `keep` over the workgroup, chunk by chunk: returns the element's position, advances `n`
    auto place = [&](bool keep, int& n, int it) -> int {
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wcnt[it * kWaves + wave] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { const int c = wcnt[it * kWaves + w]; all += c; before += w < wave ? c : 0; }
        const int at = n + before + __popcll(m & ((1ull << lane) - 1ull));
        n += all;
        return at;
    };
    // ---- stage frame 2's level-0 keypoints in grid order (level1 is always 0: ORBmatcher.cc:722-726) ----
    int n2 = 0, it = 0;      // `it`: which of the two count buffers the next compaction step uses (alternating: one barrier between reuse)
    for (int base = 0; base < nIn2; base += kThreads, it ^= 1) {
        const int pos = base + tid;
        bool keep = false;
        int i2 = 0;
        Keypoint k{};
        if (pos < nIn2) { i2 = min(max(gi2[pos], 0), cap - 1); k = K2[i2]; keep = k.octave == 0; }      // Frame.cc:705-712 with minLevel = maxLevel = 0
        const int slot = place(keep, n2, it);
        if (keep && slot < capA) {
            const int posX = (int)roundf(__fmul_rn(__fsub_rn(k.x, p.minX), p.wInv));   // the cell AssignFeaturesToGrid put it in
            const int posY = (int)roundf(__fmul_rn(__fsub_rn(k.y, p.minY), p.hInv));   // (PosInGrid, Frame.cc:728-729)
            xy2[slot] = make_float2(k.x, k.y);
            cell2[slot] = (unsigned short)((posX << 8) | posY);
            idx2[slot] = (unsigned short)i2;
            d2[2 * slot] = *(const uint4*)(D2 + (long long)i2 * 8); d2[2 * slot + 1] = *(const uint4*)(D2 + (long long)i2 * 8 + 4);
        }
    }
    // ---- the requests: frame 1's level-0 keypoints in index order (level1 > 0: continue, :722-723) ----
    int R = 0;
    for (int base = 0; base < N1; base += kThreads, it ^= 1) {
        const int i1 = base + tid;
        const bool keep = i1 < N1 && K1[i1].octave <= 0;
        const int r = place(keep, R, it);
        if (keep && r < capA) req[r] = (unsigned short)i1;
    }
    for (int i = tid; i < N1; i += kThreads) out[i] = -1;
    if (n2 > capA || R > capA) {      // more level-0 keypoints than the tables hold (not with a pyramid of several levels): report, match nothing
        if (tid == 0) nMatches[pair] = -1;
        return;
    }
    for (int i = tid; i < capA; i += kThreads) { accCnt[i] = 0; decA[i] = kNoDec; decB[i] = kNoDec; }
    __syncthreads();
    if (tid < kCols + 2) {      // first slot whose cell column is >= tid (slots are sorted by column)
        int lo = 0, hi = n2;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((cell2[mid] >> 8) < tid) lo = mid + 1; else hi = mid; }
        colStart[tid] = lo;
    }
    __syncthreads();

    unsigned* decCur = decA;      // decisions of the previous round (read by everybody), ...
    unsigned* decNew = decB;      // ... of this round (each request writes its own)
    // vMatchedDistance[s] as request r sees it: the smallest distance among the requests before r that hold s
    auto mdBefore = [&](int s, int r) -> int {
        const int c = accCnt[s];
        int md = kNone;
        if (c <= kAcc) {
#pragma unroll
            for (int k = 0; k < kAcc; k++)
                if (k < c) { const unsigned e = acc[s * kAcc + k]; if ((int)(e >> 16) < r) md = min(md, (int)(e & 0xFFFFu)); }
        } else {
            for (int j = 0; j < r; j++) { const unsigned d = decCur[j]; if (d != kNoDec && (int)(d & 0xFFFFu) == s) md = min(md, (int)(d >> 16)); }
        }
        return md;
    };
    auto decide = [&](int r) -> unsigned {
        const int i1 = req[r];
        const float px = prev[2 * i1], py = prev[2 * i1 + 1];
        // GetFeaturesInArea's cell window (Frame.cc:666-688); an empty window is "no candidates"
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(px, p.minX), p.r), p.wInv)));
        const int maxCX = min(kCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(px, p.minX), p.r), p.wInv)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(py, p.minY), p.r), p.hInv)));
        const int maxCY = min(kRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(py, p.minY), p.r), p.hInv)));
        if (minCX >= kCols || maxCX < 0 || minCY >= kRows || maxCY < 0 || minCX > maxCX || minCY > maxCY) return kNoDec;
        const uint4 a = *(const uint4*)(D1 + (long long)i1 * 8), b = *(const uint4*)(D1 + (long long)i1 * 8 + 4);
        int bestDist = kNone, bestDist2 = kNone, bs = 0;
        const int sEnd = colStart[maxCX + 1];
        for (int s = colStart[minCX]; s < sEnd; s++) {          // the columns' slot range, in the reference's traversal order
            const int cy = cell2[s] & 255;
            const float2 q = xy2[s];
            if (cy < minCY || cy > maxCY || !(fabsf(__fsub_rn(q.x, px)) < p.r) || !(fabsf(__fsub_rn(q.y, py)) < p.r)) continue;      // Frame.cc:717
            const uint4 e = d2[2 * s], g = d2[2 * s + 1];
            const int dist = __popc(a.x ^ e.x) + __popc(a.y ^ e.y) + __popc(a.z ^ e.z) + __popc(a.w ^ e.w) + __popc(b.x ^ g.x) +
                             __popc(b.y ^ g.y) + __popc(b.z ^ g.z) + __popc(b.w ^ g.w);
            if (mdBefore(s, r) <= dist) continue;                                                                // :744-745
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bs = s; }                              // :747-752
            else if (dist < bestDist2) bestDist2 = dist;                                                         // :753-756
        }
        if (bestDist > kThLow) return kNoDec;                                                                    // :759
        const float second_f = bestDist2 == kNone ? 2147483648.0f : (float)bestDist2;                            // (float)INT_MAX
        if (!((float)bestDist < __fmul_rn(second_f, p.nnRatio))) return kNoDec;                                  // :761
        return ((unsigned)bestDist << 16) | (unsigned)bs;
    };
    for (int round = 0; round <= R + 1; round++) {
        bool mineChanged = false;
        for (int r = tid; r < R; r += kThreads) {
            const unsigned d = decide(r);
            decNew[r] = d;
            mineChanged |= d != decCur[r];
        }
        if (mineChanged) flags[round & 1] = 1;
        __syncthreads();
        const bool any = flags[round & 1] != 0;
        if (tid == 0) flags[(round & 1) ^ 1] = 0;          // the other slot is read again only after the next barriers
        { unsigned* t = decCur; decCur = decNew; decNew = t; }
        if (!any) break;
        for (int s = tid; s < n2; s += kThreads) accCnt[s] = 0;
        __syncthreads();
        for (int r = tid; r < R; r += kThreads) {
            const unsigned d = decCur[r];
            if (d != kNoDec) {
                const int s = (int)(d & 0xFFFFu), k = atomicAdd(&accCnt[s], 1);
                if (k < kAcc) acc[s * kAcc + k] = ((unsigned)r << 16) | (d >> 16);
            }
        }
        __syncthreads();
    }
    // ---- the tables of the walk from the final decisions (decCur): a slot belongs to its last holder; rotHist holds every acceptance ----
    int* own = accCnt;
    for (int s = tid; s < n2; s += kThreads) own[s] = -1;
    __syncthreads();
    const float factor = 1.0f / kHistoLength;
    for (int r = tid; r < R; r += kThreads) {
        const unsigned d = decCur[r];
        decNew[r] = 0xFFu;                                  // (now: the request's rotHist bin, 255 = none)
        if (d == kNoDec) continue;
        const int s = (int)(d & 0xFFFFu);
        atomicMax(&own[s], r);
        if (p.checkOrientation) {                                                                                // :773-783
            float rot = __fsub_rn(K1[req[r]].angle, K2[idx2[s]].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == kHistoLength) bin = 0;
            decNew[r] = (unsigned)bin;
            atomicAdd(&hist[bin], 1);
        }
    }
    __syncthreads();
    int ind1 = -1, ind2 = -1, ind3 = -1;
    if (p.checkOrientation) {                                                                                  // ComputeThreeMaxima
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < kHistoLength; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
    }
    int kept = 0;
    for (int r = tid; r < R; r += kThreads) {
        const unsigned d = decCur[r];
        if (d == kNoDec) continue;
        const int s = (int)(d & 0xFFFFu), i1 = req[r];
        if (own[s] != r) continue;                                                         // stolen later (:763-767): stays -1
        const int b = (int)decNew[r];
        if (p.checkOrientation && b != ind1 && b != ind2 && b != ind3) continue;           // :797-808
        const int m = idx2[s];
        out[i1] = m;
        const Keypoint k = K2[m];
        prev[2 * i1] = k.x; prev[2 * i1 + 1] = k.y;                                        // :815-817
        kept++;
    }
    if (kept) atomicAdd(&flags[2], kept);
    __syncthreads();
    if (tid == 0) nMatches[pair] = flags[2];
}

void launchSearchInit(hipStream_t st, const Keypoint* kpsUn, const uint8_t* desc, const int* nOut, const int* gridOff,
                      const int* gridIdx, const InitMatchParams& p, float* prevMatched, int* matches12, int* nMatches, int nPairs) {
    hipLaunchKernelGGL(k_search_init, dim3(nPairs), dim3(kThreads), initMatchLdsBytes(p.capacity, p.slotCapacity), st, kpsUn, desc, nOut, gridOff,
                       gridIdx, p, prevMatched, matches12, nMatches);
}

}  // namespace orbx
