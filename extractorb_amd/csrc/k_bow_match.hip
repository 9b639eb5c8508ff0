// k_bow_match.hip — the consumer of ComputeBoW's FeatureVector: ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches)
// (reference src/ORBmatcher.cc:269-471; Tracking::TrackReferenceKeyFrame and Relocalization), Nleft == -1, with
// ORBmatcher::DescriptorDistance (:2349-2365) and ComputeThreeMaxima (:2303-2344), on device-resident frames.
//
// The reference walks the two FeatureVectors (std::map<NodeId, vector<feature index>>) in step and matches only inside a vocabulary
// node both frames have: for every keyframe feature of the node that holds a good MapPoint, in list order, the nearest and the second
// nearest of the frame's features of the node that have not been given a MapPoint yet (:318-319), accepted under TH_LOW and the ratio
// test (:375-377); an accepted feature is closed to later keyframe features.  A feature lies in exactly one node, so the chain of
// "already matched" dependencies never leaves a node: NODES ARE INDEPENDENT, only the keyframe features inside a node are sequential.
//   k_search_bow: one workgroup per (keyframe, frame) pair; both node columns in LDS; the starts of the keyframe's node segments are
//   collected (any order), and each 16-lane row of the workgroup takes segments: binary search of the frame's segment, then for each
//   keyframe feature in order its 16 lanes share the frame's features of the node (distance << 16 | position keys, the two
//   smallest per lane, a 16-lane DPP minimum for best and second best: the reference's running (bestDist1, bestDist2) after its scan
//   are the smallest and second smallest of the multiset, the best index the first position with the smallest distance).
//   The rotation histogram (:384-401) is 30 LDS counters; ComputeThreeMaxima and the clean-up (:452-468) close the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

struct BowMatchParams {
    float nnRatio;
    int thLow, checkOrientation, capacity, kfFirst, kfStep, curFirst, curStep;
    int twoKeyFrames;      // SearchByBoW(pKF1, pKF2, vpMatches12): candidates must hold a MapPoint, strict threshold, result indexed by pKF1's keypoints
};

namespace {
constexpr int kHistoLength = 30;                          // ORBmatcher.cc:38
constexpr unsigned kNoneKey = (256u << 16) | 0xFFFFu;     // bestDist = 256, no position
constexpr int kThreads = 1024;     // 64 rows of 16 lanes: ~100 node segments of a frame are two sequential segments per row

template <int CTRL>
__device__ __forceinline__ unsigned dppMinU(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
    return o < v ? o : v;
}
// minimum over each row of 16 lanes, returned in every lane of the row
__device__ __forceinline__ unsigned rowMin16(unsigned v) {
    v = dppMinU<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dppMinU<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dppMinU<0x141>(v);     // row_half_mirror
    v = dppMinU<0x140>(v);     // row_mirror
    return v;
}
}  // namespace

size_t bowMatchLdsBytes(int capacity, bool stageDesc) { return (size_t)((capacity + 15) & ~15) * (4 + 4 + 4 + 4 + 1 + 2 + 2 + 1 + (stageDesc ? 64 : 0)) + 64; }

// grid n_pairs; 1024 threads; dynamic LDS bowMatchLdsBytes(capacity, STAGE).  STAGE: both frames' descriptors fit in LDS next to the
// tables (a row's walk over its keyframe features is a chain of dependent descriptor reads: from L2 it was 169 us per pair)
template <bool STAGE>
__global__ __launch_bounds__(kThreads) void k_search_bow(const uint32_t* __restrict__ featNodes, const uint32_t* __restrict__ featIdx,
                                                         const int* __restrict__ nFeat, const uint8_t* __restrict__ kfFlags,
                                                         const uint8_t* __restrict__ curFlags, const Keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                         const int* __restrict__ nOut, BowMatchParams p, int* __restrict__ matches,
                                                         int* __restrict__ nMatches) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int sSeg, sHist[kHistoLength], sCount;
    const int cap = p.capacity, capA = (cap + 15) & ~15, pair = blockIdx.x, tid = threadIdx.x, sub = tid & 15, row = tid >> 4;
    const int fK = p.kfFirst + pair * p.kfStep, fC = p.curFirst + pair * p.curStep;
    uint4* sDescK = (uint4*)smem;                         // (STAGE) [capA][2] the keyframe's descriptors
    uint4* sDescC = sDescK + (STAGE ? 2 * capA : 0);      // (STAGE) [capA][2] the frame's
    uint32_t* nodeK = (uint32_t*)(sDescC + (STAGE ? 2 * capA : 0));      // [capA] node column of the keyframe's FeatureVector
    uint32_t* nodeC = nodeK + capA;                       // [capA] ... of the frame's
    int* segList = (int*)(nodeC + capA);                  // [capA] first entry of every keyframe node segment
    volatile int* takenBy = (volatile int*)(segList + capA);  // [capA] frame keypoint -> keyframe keypoint whose MapPoint it got (vpMapPointMatches)
    unsigned short* sIdxK = (unsigned short*)(takenBy + capA);      // [capA] feature-index column of the keyframe's FeatureVector
    unsigned short* sIdxC = sIdxK + capA;                 // [capA] ... of the frame's
    uint8_t* binOf = (uint8_t*)(sIdxC + capA);            // [capA] rotHist bin the frame keypoint was pushed to
    uint8_t* sFlag = binOf + capA;                        // [capA] the keyframe's MapPoint flags
    const int MK = min(nFeat[fK], cap), MC = min(nFeat[fC], cap), NC = min(nOut[fC], cap), NK = min(nOut[fK], cap);
    const uint32_t *gNodeK = featNodes + (long long)fK * cap, *gNodeC = featNodes + (long long)fC * cap;
    const uint32_t *idxK = featIdx + (long long)fK * cap, *idxC = featIdx + (long long)fC * cap;
    const uint8_t* flags = kfFlags + (long long)pair * cap;
    const uint32_t *descK = (const uint32_t*)(desc + (long long)fK * cap * 32), *descC = (const uint32_t*)(desc + (long long)fC * cap * 32);
    const Keypoint *kpK = kps + (long long)fK * cap, *kpC = kps + (long long)fC * cap;
    if (tid == 0) { sSeg = 0; sCount = 0; }
    if (tid < kHistoLength) sHist[tid] = 0;
    // (indices clamped: a corrupt FeatureVector must not index past the tables)
    for (int i = tid; i < MK; i += kThreads) { nodeK[i] = gNodeK[i]; sIdxK[i] = (unsigned short)min(idxK[i], (uint32_t)(cap - 1)); }
    for (int i = tid; i < MC; i += kThreads) { nodeC[i] = gNodeC[i]; sIdxC[i] = (unsigned short)min(idxC[i], (uint32_t)(cap - 1)); }
    // (the second keyframe's keypoints without a good MapPoint are never candidates (:879-886): closed from the start, marked -2)
    for (int i = tid; i < cap; i += kThreads) {
        takenBy[i] = p.twoKeyFrames && !(curFlags[(long long)pair * cap + i] & 1) ? -2 : -1;
        binOf[i] = 255; sFlag[i] = flags[i];
    }
    if constexpr (STAGE) {
        for (int i = tid; i < 2 * NK; i += kThreads) sDescK[i] = ((const uint4*)descK)[i];
        for (int i = tid; i < 2 * NC; i += kThreads) sDescC[i] = ((const uint4*)descC)[i];
    }
    auto loadK = [&](int idx, uint4& a, uint4& b) {
        if constexpr (STAGE) { a = sDescK[2 * idx]; b = sDescK[2 * idx + 1]; }
        else { a = *(const uint4*)(descK + (long long)idx * 8); b = *(const uint4*)(descK + (long long)idx * 8 + 4); }
    };
    auto loadC = [&](int idx, uint4& a, uint4& b) {
        if constexpr (STAGE) { a = sDescC[2 * idx]; b = sDescC[2 * idx + 1]; }
        else { a = *(const uint4*)(descC + (long long)idx * 8); b = *(const uint4*)(descC + (long long)idx * 8 + 4); }
    };
    __syncthreads();
    for (int i = tid; i < MK; i += kThreads)
        if (i == 0 || nodeK[i] != nodeK[i - 1]) segList[atomicAdd(&sSeg, 1)] = i;      // (any order: nodes are independent)
    __syncthreads();
    const int nSeg = sSeg;
    const float factor = 1.0f / kHistoLength;
    for (int s = row; s < nSeg; s += kThreads / 16) {
        const int k0 = segList[s];
        const uint32_t node = nodeK[k0];
        // the frame's segment of the same node: [c0, c1) (the maps' lower_bound walk, :432-439, meets exactly the common keys)
        int lo = 0, hi = MC;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (nodeC[mid] < node) lo = mid + 1; else hi = mid; }
        const int c0 = lo;
        if (c0 >= MC || nodeC[c0] != node) continue;
        int c1 = c0 + 1;
        while (c1 < MC && nodeC[c1] == node) c1++;
        for (int k = k0; k < MK && nodeK[k] == node; k++) {                                 // vIndicesKF in list order (:297)
            const int realIdxKF = (int)sIdxK[k];
            if (!(sFlag[realIdxKF] & 1)) continue;                                          // no MapPoint, or a bad one (:303-307)
            uint4 a, b;
            loadK(realIdxKF, a, b);
            unsigned key = kNoneKey, second = kNoneKey;
            for (int c = c0 + sub; c < c1; c += 16) {
                const int realIdxF = (int)sIdxC[c];
                if (takenBy[realIdxF] != -1) continue;                                      // :318-319 (:884-888)
                uint4 x, y;
                loadC(realIdxF, x, y);
                const int dist = __popc(a.x ^ x.x) + __popc(a.y ^ x.y) + __popc(a.z ^ x.z) + __popc(a.w ^ x.w) + __popc(b.x ^ y.x) +
                                 __popc(b.y ^ y.y) + __popc(b.z ^ y.z) + __popc(b.w ^ y.w);
                const unsigned kk = ((unsigned)dist << 16) | (unsigned)(c - c0);            // positions ascend per lane: a later equal distance never displaces (:325, :331)
                if (kk < key) { second = key; key = kk; }
                else if (kk < second) second = kk;
            }
            const unsigned best = rowMin16(key);
            const unsigned best2 = rowMin16(key == best ? second : key);
            const int bestDist1 = (int)(best >> 16), bestDist2 = (int)(best2 >> 16);
            if ((p.twoKeyFrames ? bestDist1 < p.thLow : bestDist1 <= p.thLow) && (float)bestDist1 < __fmul_rn(p.nnRatio, (float)bestDist2)) {   // :375-377 (:908-910)
                const int bestIdxF = (int)sIdxC[c0 + (int)(best & 0xFFFFu)];
                if (sub == 0) {
                    takenBy[bestIdxF] = realIdxKF;                                          // vpMapPointMatches[bestIdxF] = pMP
                    if (p.checkOrientation) {                                               // :384-401
                        float rot = __fsub_rn(kpK[realIdxKF].angle, kpC[bestIdxF].angle);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, factor));
                        if (bin == kHistoLength) bin = 0;
                        binOf[bestIdxF] = (uint8_t)bin;
                        atomicAdd(&sHist[bin], 1);
                    }
                }
                __builtin_amdgcn_wave_barrier();      // the row's next keyframe feature must see the closed keypoint (same wave: LDS is in order)
            }
        }
    }
    __syncthreads();
    unsigned dropBins = 0u;
    if (p.checkOrientation) {                                                               // ComputeThreeMaxima (:2303-2344), then :452-468
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < kHistoLength; i++) {
            const int s = sHist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
        for (int i = 0; i < kHistoLength; i++)
            if (i != ind1 && i != ind2 && i != ind3) dropBins |= 1u << i;
    }
    int mine = 0;
    int* out = matches + (long long)pair * cap;
    if (p.twoKeyFrames) {      // vpMatches12[idx1] = vpMapPoints2[bestIdx2]: the table is indexed by the FIRST keyframe's keypoints
        for (int i = tid; i < cap; i += kThreads) out[i] = -1;
        __syncthreads();
    }
    for (int i = tid; i < cap; i += kThreads) {
        int m = i < NC ? takenBy[i] : -1;
        if (m >= 0 && binOf[i] < kHistoLength && ((dropBins >> binOf[i]) & 1u)) m = -1;
        if (!p.twoKeyFrames) out[i] = m;
        else if (m >= 0) out[m] = i;
        mine += m >= 0;
    }
    if (mine) atomicAdd(&sCount, mine);
    __syncthreads();
    if (tid == 0) nMatches[pair] = sCount;
}

void launchSearchBow(hipStream_t st, const uint32_t* featNodes, const uint32_t* featIdx, const int* nFeat, const uint8_t* kfFlags,
                     const uint8_t* curFlags, const Keypoint* kps, const uint8_t* desc, const int* nOut, const BowMatchParams& p, int* matches, int* nMatches, int nPairs) {
    if (bowMatchLdsBytes(p.capacity, true) <= 150 * 1024)
        hipLaunchKernelGGL(k_search_bow<true>, dim3(nPairs), dim3(kThreads), bowMatchLdsBytes(p.capacity, true), st, featNodes, featIdx, nFeat, kfFlags,
                           curFlags, kps, desc, nOut, p, matches, nMatches);
    else
        hipLaunchKernelGGL(k_search_bow<false>, dim3(nPairs), dim3(kThreads), bowMatchLdsBytes(p.capacity, false), st, featNodes, featIdx, nFeat, kfFlags,
                           curFlags, kps, desc, nOut, p, matches, nMatches);
}

}  // namespace orbx
