// k_bow.hip — "next" row SURVEY.md §8f-4: Frame::ComputeBoW (reference src/Frame.cc:739-746) =
// DBoW2::TemplatedVocabulary<FORB>::transform(features, BowVector, FeatureVector, levelsup)
// (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1196 and :1218-1262, BowVector.cpp:34-83, FeatureVector.cpp:31-46,
// FORB.cpp distance = 256-bit Hamming) on device-resident descriptors.
//
// k_bow_words: every descriptor descends the vocabulary tree; 16 lanes share one descriptor, one child each (k <= 16 in one
// round, more in several), the nearest child — first one on ties, as the strict "<" of :1243 — by a 16-lane DPP minimum over
// (distance << 8 | child rank).  Output per feature: word id, word weight, node at level L - levelsup.
// k_bow_reduce: one workgroup (1024 threads) per frame builds the two std::maps the reference builds: features sorted by (word, index) and
// summed per word IN FEATURE ORDER (BowVector::addWeight adds one weight at a time, and floating-point addition is not
// associative), the L1 / L2 norm accumulated over the words in ascending order by ONE lane (the map iteration order of
// BowVector::normalize), then the division; and features sorted by (node, index) for the FeatureVector.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

struct VocabDevice {            // plain arrays of the tree (TemplatedVocabulary::m_nodes)
    const int* childOff;        // [nNodes + 1] children of node n: childList[childOff[n] .. childOff[n + 1])  (in m_nodes[n].children order)
    const int* childList;
    const uint32_t* desc;       // [nNodes][8]  node descriptors
    const double* weight;       // [nNodes]
    const uint32_t* wordId;     // [nNodes]     valid for leaves
    int nNodes, k, L, scoring, weighting;
};

namespace {
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dppMinU(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false);
    return o < v ? o : v;
}
// minimum over each row of 16 lanes, returned in every lane of the row
__device__ __forceinline__ unsigned rowMin16(unsigned v) {
    v = dppMinU<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dppMinU<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dppMinU<0x141, 0xF>(v);    // row_half_mirror
    v = dppMinU<0x140, 0xF>(v);    // row_mirror
    return v;
}
}  // namespace

// grid (ceil(capacity / 16), n_frames); 256 threads = 16 features x 16 lanes.
__global__ __launch_bounds__(256) void k_bow_words(VocabDevice V, const uint8_t* __restrict__ desc, const int* __restrict__ nOut,
                                                   int capacity, int levelsUp, uint32_t* __restrict__ featWord,
                                                   double* __restrict__ featWeight, uint32_t* __restrict__ featNode) {
    const int f = blockIdx.y, i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int N = min(nOut[f], capacity);
    if (i >= N) return;                                   // the 16 lanes of a feature leave together
    const uint32_t* d = (const uint32_t*)(desc + ((long long)f * capacity + i) * 32);
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = d[j];
    const int nidLevel = V.L - levelsUp;                  // TemplatedVocabulary.h:1226
    unsigned nid = 0;                                     // root if nid_level <= 0 (:1227); stays 0 if the leaf is shallower
    int node = 0, level = 0;
    for (int guard = 0; guard < 64; guard++) {            // do { ... } while (!isLeaf)  (:1232-1255); 64 levels bound a corrupt table
        const int c0 = V.childOff[node], c1 = V.childOff[node + 1];
        if (c1 <= c0) break;                              // (a childless root: the reference would dereference nodes[0])
        ++level;
        unsigned best = 0xFFFFFFFFu;
        for (int cb = c0; cb < c1; cb += 16) {            // 16 children per round
            unsigned key = 0xFFFFFFFFu;
            if (cb + sub < c1) {
                const uint32_t* cd = V.desc + (long long)V.childList[cb + sub] * 8;
                int dist = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) dist += __popc(w[j] ^ cd[j]);
                key = ((unsigned)dist << 8) | (unsigned)(cb - c0 + sub);      // first child wins a distance tie (d < best_d, :1243)
            }
            best = min(best, rowMin16(key));
        }
        node = V.childList[c0 + (int)(best & 255u)];
        if (level == nidLevel) nid = (unsigned)node;      // :1251-1252
    }
    if (sub == 0) {
        const long long o = (long long)f * capacity + i;
        featWord[o] = V.wordId[node];                     // :1258-1259
        featWeight[o] = V.weight[node];
        featNode[o] = nid;
    }
}

namespace {
constexpr int kReduceThreads = 1024;      // one key per thread at ~1000 features: a bitonic step is one compare-exchange + one barrier
// in-place ascending bitonic sort of n = power of two u64 keys in LDS, all threads of the workgroup
template <int T>
__device__ void bitonicSort(unsigned long long* key, int n) {
    for (int k2 = 2; k2 <= n; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key[i], b = key[ixj];
                    if ((a > b) == ((i & k2) == 0)) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
}
}  // namespace

// grid n_frames; 1024 threads; dynamic LDS: P u64 keys + P doubles (P = next power of two >= capacity).
// (One frame used to take 193 us here with 256 threads, a rank loop of O(i) per word and the norm summed from global memory; the
// reference's floating-point order leaves exactly one sequential piece, the norm, and that now runs on LDS values.)
__global__ __launch_bounds__(kReduceThreads) void k_bow_reduce(const uint32_t* __restrict__ featWord, const double* __restrict__ featWeight,
                                                    const uint32_t* __restrict__ featNode, const int* __restrict__ nOut, int capacity, int Pmax,
                                                    int scoring, int weighting, uint32_t* __restrict__ wordIds,
                                                    double* wordWeights /* beyond 8192 slots also the workgroup's wsum scratch (thread to thread): not __restrict__ */,
                                                    int* __restrict__ nWords, uint32_t* __restrict__ fvNodes, uint32_t* __restrict__ fvIdx,
                                                    int* __restrict__ nFeat) {
    constexpr int T = kReduceThreads;
    extern __shared__ __align__(16) uint8_t smem[];
    unsigned long long* key = (unsigned long long*)smem;
    // [Pmax] summed weight of word rank r: in LDS while 16 Pmax bytes fit a workgroup, else in the output array itself
    double* wsum = Pmax <= 8192 ? (double*)(key + Pmax) : wordWeights + (long long)blockIdx.x * capacity;
    __shared__ int sCount, sWaveSum[T / 64];
    __shared__ double sNorm;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = min(nOut[f], capacity);
    const long long base = (long long)f * capacity;
    int P = 64;                                           // the sorts run on the next power of two above THIS frame's keypoint count
    while (P < N) P <<= 1;                                // (<= Pmax, the next power of two above the capacity, which sizes the LDS)
    // ---- BowVector: features with a positive weight, sorted by (word id, feature index) ----
    if (tid == 0) sCount = 0;
    for (int i = tid; i < P; i += T) {
        unsigned long long k = ~0ull;
        if (i < N && featWeight[base + i] > 0) k = ((unsigned long long)featWord[base + i] << 32) | (unsigned)i;      // if (w > 0) (:1161)
        key[i] = k;
    }
    __syncthreads();
    bitonicSort<T>(key, P);
    for (int i = tid; i < P; i += T)
        if (key[i] != ~0ull && (i + 1 == P || key[i + 1] == ~0ull)) sCount = i + 1;
    __syncthreads();
    const int M = sCount;
    if (tid == 0) nFeat[f] = M;
    // word segments: the thread at a segment's first entry walks it, adding the feature weights one by one in feature order
    // (BowVector::addWeight, BowVector.cpp:34-46; TF / TF_IDF), or takes the first (addIfNotExist, :50-58; IDF / BINARY).  The
    // segment's rank (= position of the word in the map) is a prefix count of segment starts: every thread owns a block of `per`
    // consecutive entries, a wave-level DPP scan + the wave totals give the number of starts before the block.
    const bool accumulate = weighting == 0 || weighting == 1;      // TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3
    const int per = (P + T - 1) / T, b = tid * per, e = min(b + per, M);
    int starts = 0;
    for (int i = b; i < e; i++) starts += i == 0 || (unsigned)(key[i] >> 32) != (unsigned)(key[i - 1] >> 32);
    const int incl = waveInclusiveScan(starts);
    if (lane == 63) sWaveSum[wave] = incl;
    __syncthreads();
    int rank = incl - starts, W = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) { const int t = sWaveSum[w]; rank += w < wave ? t : 0; W += t; }
    for (int i = b; i < e; i++) {
        const unsigned word = (unsigned)(key[i] >> 32);
        if (i > 0 && (unsigned)(key[i - 1] >> 32) == word) continue;
        double s = featWeight[base + (unsigned)key[i]];
        if (accumulate)
            for (int j = i + 1; j < M && (unsigned)(key[j] >> 32) == word; j++) s = __dadd_rn(s, featWeight[base + (unsigned)key[j]]);
        wordIds[base + rank] = word;
        wsum[rank] = s;
        rank++;
    }
    if (tid == 0) nWords[f] = W;
    __syncthreads();
    // normalisation (:1170-1176 when the scoring needs none: divide by the number of words; else BowVector::normalize, BowVector.cpp:62-83)
    const bool must = scoring != 5;                       // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    const bool l2 = scoring == 1;                         // L2_NORM uses L2, the rest L1
    if (tid == 0) {
        double norm = 0.0;
        if (must) {
            int j = 0;
            for (; j + 8 <= W; j += 8) {                  // ascending word order = the map's iteration order; eight LDS reads in flight
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = wsum[j + u];
#pragma unroll
                for (int u = 0; u < 8; u++) norm = l2 ? __dadd_rn(norm, __dmul_rn(v[u], v[u])) : __dadd_rn(norm, fabs(v[u]));
            }
            for (; j < W; j++) { const double v = wsum[j]; norm = l2 ? __dadd_rn(norm, __dmul_rn(v, v)) : __dadd_rn(norm, fabs(v)); }
            if (l2) norm = __dsqrt_rn(norm);
        } else if (accumulate) {
            norm = (double)W;
        }
        sNorm = norm;
    }
    __syncthreads();
    const double norm = sNorm;
    for (int j = tid; j < W; j += T) wordWeights[base + j] = norm > 0.0 ? __ddiv_rn(wsum[j], norm) : wsum[j];
    __syncthreads();
    // ---- FeatureVector: the same features sorted by (node at level L - levelsup, feature index) (FeatureVector.cpp:31-46) ----
    for (int i = tid; i < P; i += T) {
        unsigned long long k = ~0ull;
        if (i < N && featWeight[base + i] > 0) k = ((unsigned long long)featNode[base + i] << 32) | (unsigned)i;
        key[i] = k;
    }
    __syncthreads();
    bitonicSort<T>(key, P);
    for (int i = tid; i < M; i += T) { fvNodes[base + i] = (unsigned)(key[i] >> 32); fvIdx[base + i] = (unsigned)key[i]; }
}

void launchBow(hipStream_t st, const VocabDevice& V, const uint8_t* desc, const int* nOut, int capacity, int levelsUp, uint32_t* featWord,
               double* featWeight, uint32_t* featNode, uint32_t* wordIds, double* wordWeights, int* nWords, uint32_t* fvNodes, uint32_t* fvIdx,
               int* nFeat, int B) {
    hipLaunchKernelGGL(k_bow_words, dim3((capacity + 15) / 16, B), dim3(256), 0, st, V, desc, nOut, capacity, levelsUp, featWord, featWeight, featNode);
    int P = 64;                       // the kernel never sorts fewer than 64 keys: the LDS block and the wsum offset follow the same minimum
    while (P < capacity) P <<= 1;
    hipLaunchKernelGGL(k_bow_reduce, dim3(B), dim3(kReduceThreads), (size_t)P * (P <= 8192 ? 16 : 8), st, featWord, featWeight, featNode, nOut, capacity, P, V.scoring, V.weighting,
                       wordIds, wordWeights, nWords, fvNodes, fvIdx, nFeat);
}

}  // namespace orbx
