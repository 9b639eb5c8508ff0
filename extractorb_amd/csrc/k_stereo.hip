// k_stereo.hip — "next" row SURVEY.md §8f-1: Frame::ComputeStereoMatches (reference src/Frame.cc:813-991)
// on the device-resident outputs of one batched extraction (frame 2p = left eye, frame 2p+1 = right eye).
//
//   k_stereo_rows    per pair: the reference's vRowIndices — for every image row the right keypoints whose
//                    band [floor(y - 2s), ceil(y + 2s)] (s = scale of the keypoint's octave) covers it  (:826-840)
//   k_stereo_match   one wave per left keypoint: 256-bit Hamming search over the row's candidates
//                    (ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2349-2365; octave within +-1, u inside
//                    [uL - bf/b, uL]), then the 11-shift SAD of an 11x11 window on the LEFT/RIGHT PYRAMIDS — which
//                    never leave HBM — and the parabola sub-pixel fit  (:849-978)
//   k_stereo_filter  per pair: median of the SAD distances, matches at >= 1.5*1.4*median dropped  (:981-996)
//
// Ties: the reference keeps the first minimum while walking candidates in increasing right index, so the search
// minimises the pair (distance, right index).  All float steps use explicit single roundings.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

struct StereoParams {
    float scale[kMaxLevels], invScale[kMaxLevels];
    float bf, b;
    int nlevels, capacity, rowCap;   // rowCap: entries of one pair's row-list arena
};

constexpr int kThHigh = 100, kThLow = 50;   // ORBmatcher.cc:36-37

// ---- row tables -----------------------------------------------------------------------------------------
// grid: n_pairs; 1024 threads.  rowOff: [pair][rows+1]; rowList: [pair][rowCap] right indices.
__global__ __launch_bounds__(1024) void k_stereo_rows(const Keypoint* __restrict__ kps, const int* __restrict__ nOut,
                                                       StereoParams sp, int rows, int* __restrict__ rowOff,
                                                       unsigned short* __restrict__ rowList) {
    extern __shared__ int cnt[];   // [rows + 1] counts, then cursors
    __shared__ int wsum[16];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Keypoint* kr = kps + (long long)(2 * pair + 1) * sp.capacity;
    const int Nr = min(nOut[2 * pair + 1], sp.capacity);
    for (int i = tid; i <= rows; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int iR = tid; iR < Nr; iR += 1024) {
        const float kpY = kr[iR].y, r = __fmul_rn(2.0f, sp.scale[min(max(kr[iR].octave, 0), sp.nlevels - 1)]);
        const int maxr = min((int)ceilf(__fadd_rn(kpY, r)), rows - 1), minr = max((int)floorf(__fsub_rn(kpY, r)), 0);
        for (int y = minr; y <= maxr; y++) atomicAdd(&cnt[y], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0..rows) by 1024 threads
    const int per = (rows + 1023) / 1024, b0 = tid * per, e0 = min(b0 + per, rows);
    int sum = 0;
    for (int i = b0; i < e0; i++) sum += cnt[i];
    const int incl = waveInclusiveScan(sum);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; w++) off += wsum[w];
    int run = off + incl - sum;
    int* ro = rowOff + (long long)pair * (rows + 1);
    for (int i = b0; i < e0; i++) { const int c = cnt[i]; ro[i] = run; cnt[i] = run; run += c; }   // cnt becomes the fill cursor
    if (tid == 1023) ro[rows] = run;
    __syncthreads();
    unsigned short* rl = rowList + (long long)pair * sp.rowCap;
    for (int iR = tid; iR < Nr; iR += 1024) {
        const float kpY = kr[iR].y, r = __fmul_rn(2.0f, sp.scale[min(max(kr[iR].octave, 0), sp.nlevels - 1)]);
        const int maxr = min((int)ceilf(__fadd_rn(kpY, r)), rows - 1), minr = max((int)floorf(__fsub_rn(kpY, r)), 0);
        for (int y = minr; y <= maxr; y++) {
            const int at = atomicAdd(&cnt[y], 1);
            if (at < sp.rowCap) rl[at] = (unsigned short)iR;
        }
    }
}

// ---- matching ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int waveSumI(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dppAdd(unsigned v) { return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false); }
// sum over the 64 lanes without the LDS crossbar, returned wave-uniform
__device__ __forceinline__ unsigned waveSumDpp(unsigned v) {
    v = dppAdd<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v = dppAdd<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v = dppAdd<0x141, 0xF>(v);    // row_half_mirror
    v = dppAdd<0x140, 0xF>(v);    // row_mirror: every lane of a row holds the row's sum
    v = dppAdd<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v = dppAdd<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dppMinU(unsigned v) { return min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false)); }
__device__ __forceinline__ unsigned waveMinDpp(unsigned v) {
    v = dppMinU<0xB1, 0xF>(v); v = dppMinU<0x4E, 0xF>(v); v = dppMinU<0x141, 0xF>(v); v = dppMinU<0x140, 0xF>(v);
    v = dppMinU<0x142, 0xA>(v); v = dppMinU<0x143, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// grid (ceil(capacity/4), n_pairs), 256 threads: one wave per left keypoint.
__global__ __launch_bounds__(256) void k_stereo_match(const LevelGeom* __restrict__ lv, const uint8_t* __restrict__ pyr,
                                                       const Keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                       const int* __restrict__ nOut, StereoParams sp, int rows,
                                                       const int* __restrict__ rowOff, const unsigned short* __restrict__ rowList,
                                                       float* __restrict__ uRight, float* __restrict__ depth,
                                                       int* __restrict__ sadDist) {
    const int lane = threadIdx.x & 63, iL = blockIdx.x * 4 + (threadIdx.x >> 6), pair = blockIdx.y;
    const int fL = 2 * pair, fR = 2 * pair + 1;
    const int N = min(nOut[fL], sp.capacity), Nr = min(nOut[fR], sp.capacity);
    if (iL >= N) return;
    const long long outIdx = (long long)pair * sp.capacity + iL;
    if (lane == 0) { uRight[outIdx] = -1.0f; depth[outIdx] = -1.0f; sadDist[outIdx] = -1; }
    const Keypoint* kl = kps + (long long)fL * sp.capacity;
    const Keypoint* kr = kps + (long long)fR * sp.capacity;
    const float uL = kl[iL].x, vL = kl[iL].y;
    const int levelL = min(max(kl[iL].octave, 0), sp.nlevels - 1);
    const int row = (int)vL;
    if (row < 0 || row >= rows) return;
    const int* ro = rowOff + (long long)pair * (rows + 1);
    const int c0 = ro[row], c1 = min(ro[row + 1], sp.rowCap);
    if (c1 <= c0) return;                                              // vCandidates.empty()  (:857-858)
    const float minZ = sp.b, minD = 0.f, maxD = __fdiv_rn(sp.bf, minZ);
    const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
    if (maxU < 0) return;
    // ---- descriptor search (:867-893) ----
    const unsigned* dl = (const unsigned*)(desc + ((long long)fL * sp.capacity + iL) * 32);
    unsigned dL[8];
#pragma unroll
    for (int k = 0; k < 8; k++) dL[k] = dl[k];
    const unsigned short* rl = rowList + (long long)pair * sp.rowCap;
    unsigned best = ((unsigned)kThHigh << 16);                         // (distance << 16 | right index): lexicographic minimum
    for (int c = c0 + lane; c < c1; c += 64) {
        const int iR = rl[c];
        if (iR >= Nr) continue;
        const int oR = kr[iR].octave;
        const float uR = kr[iR].x;
        if (oR < levelL - 1 || oR > levelL + 1) continue;
        if (!(uR >= minU && uR <= maxU)) continue;
        const unsigned* dr = (const unsigned*)(desc + ((long long)fR * sp.capacity + iR) * 32);
        int dist = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) dist += __popc(dL[k] ^ dr[k]);
        const unsigned key = ((unsigned)dist << 16) | (unsigned)iR;
        best = dist < kThHigh && key < best ? key : best;
    }
    best = waveMinDpp(best);
    const int bestDist = (int)(best >> 16);
    if (bestDist >= (kThHigh + kThLow) / 2) return;                    // thOrbDist  (:896)
    const int bestIdxR = (int)(best & 0xffff);
    // ---- sub-pixel refinement by correlation on the pyramids (:898-978) ----
    const float uR0 = kr[bestIdxR].x;
    const float scaleFactor = sp.invScale[levelL];
    const float scaleduL = roundf(__fmul_rn(uL, scaleFactor)), scaledvL = roundf(__fmul_rn(vL, scaleFactor));
    const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
    const int w = 5, L = 5;
    const LevelGeom g = lv[levelL];
    const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)L), (float)w), endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)L), (float)w), 1.f);
    if (iniu < 0 || endu >= (float)g.w) return;
    const int y0 = (int)__fsub_rn(scaledvL, (float)w), xl0 = (int)__fsub_rn(scaleduL, (float)w);
    // guard the gathers: the reference's windows stay inside the bordered level for every keypoint it can produce
    if (y0 < -kEdge || y0 + 10 >= g.h + kEdge || xl0 < -kEdge || xl0 + 10 >= g.w + kEdge) return;
    const uint8_t* pl = pyr + g.pyrOff + (long long)fL * g.pyrFrameBytes + (long long)kEdge * g.pyrStride + kPadL;
    const uint8_t* pr = pyr + g.pyrOff + (long long)fR * g.pyrFrameBytes + (long long)kEdge * g.pyrStride + kPadL;
    // The right strip (11 rows x 21 columns: the 11x11 window at all 11 shifts) goes through LDS once; the early return
    // above guarantees scaleduR0 - 10 >= -10 and scaleduR0 + 10 < g.w, so every shift's window lies inside the bordered level.
    __shared__ uint8_t strips[4][11 * 21 + 25];
    uint8_t* R = strips[threadIdx.x >> 6];
    const int xrBase = (int)__fsub_rn(__fadd_rn(scaleduR0, -5.f), (float)w);          // xr0 of incR = -5; xr0(incR) = xrBase + incR + 5
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int p = lane + 64 * t;
        if (p < 11 * 21) {
            const int ry = p / 21, rx = p - ry * 21;
            R[p] = pr[(long long)(y0 + ry) * g.pyrStride + xrBase + rx];
        }
    }
    // lane owns window pixels p = lane and lane + 64 (121 in all)
    int il[2], oy[2], ox[2];
    bool in[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int p = lane + 64 * t;
        in[t] = p < 121;
        oy[t] = in[t] ? p / 11 : 0;
        ox[t] = in[t] ? p - oy[t] * 11 : 0;
        il[t] = pl[(long long)(y0 + oy[t]) * g.pyrStride + xl0 + ox[t]];
    }
    const int cL = pl[(long long)(y0 + w) * g.pyrStride + xl0 + w];
    il[0] -= cL; il[1] -= cL;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    // per-lane partial SADs of all 11 shifts (<= 2 * 510 each: two shifts share a register, 16 bits apiece, and the wave
    // totals (<= 65 280) still fit), then 6 DPP wave sums instead of 11 shuffle reductions
    unsigned packed[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 11; s++) {
        const int cR = R[w * 21 + w + s];
        int part = 0;
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (in[t]) part += abs(il[t] - ((int)R[oy[t] * 21 + ox[t] + s] - cR));
        packed[s >> 1] |= (unsigned)part << (16 * (s & 1));
    }
    float vDists[11];
    int bestS = 2147483647, bestincR = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const unsigned tot = waveSumDpp(packed[j]);
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const int s = 2 * j + hlf;
            if (s < 11) {
                const float dist = (float)((tot >> (16 * hlf)) & 0xffffu);
                if (dist < (float)bestS) { bestS = (int)dist; bestincR = s - 5; }
                vDists[s] = dist;
            }
        }
    }
    if (bestincR == -L || bestincR == L) return;
    float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
    for (int k = 1; k < 10; k++)
        if (k == bestincR + 5) { dist1 = vDists[k - 1]; dist2 = vDists[k]; dist3 = vDists[k + 1]; }
    const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                   __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = __fmul_rn(sp.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
    float disparity = __fsub_rn(uL, bestuR);
    if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
        if (lane == 0) {
            depth[outIdx] = __fdiv_rn(sp.bf, disparity);
            uRight[outIdx] = bestuR;
            sadDist[outIdx] = bestS;
        }
    }
}

// ---- outlier removal (:981-996) ---------------------------------------------------------------------------
// grid n_pairs, 1024 threads.  median = the (cnt/2)-th element of the (distance, left index)-sorted matches.
__global__ __launch_bounds__(1024) void k_stereo_filter(const int* __restrict__ nOut, StereoParams sp, float* __restrict__ uRight,
                                                         float* __restrict__ depth, const int* __restrict__ sadDist,
                                                         int* __restrict__ nMatched) {
    extern __shared__ __align__(16) int d[];   // [capacity rounded up to 4] SAD distances, -1 = unmatched
    __shared__ int sCnt, sMedian, sKept, hist[256], sSel[2];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int N = min(nOut[2 * pair], sp.capacity);
    const long long base = (long long)pair * sp.capacity;
    if (tid == 0) { sCnt = 0; sMedian = 0; sKept = 0; }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < ((N + 3) & ~3); i += 1024) { d[i] = i < N ? sadDist[base + i] : -1; mine += d[i] >= 0; }      // padded to whole int4s (unmatched)
    if (mine) atomicAdd(&sCnt, mine);
    __syncthreads();
    const int cnt = sCnt;
    if (cnt == 0) { if (tid == 0) nMatched[pair] = 0; return; }       // the reference reads vDistIdx[0] here (undefined)
    const int kth = cnt / 2;
    // the median is the kth smallest distance VALUE (the index only orders equal distances): radix select, most significant byte first —
    // per byte a 256-bin LDS histogram of the distances that share the prefix found so far, and one wave's scan for the bin that holds
    // rank kth.  (Counting every keypoint's rank against every other is O(N^2) on ONE CU: 75 of this kernel's 85 us for one pair.)
    {
        unsigned prefix = 0u;
        int remaining = kth;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned himask = shift == 24 ? 0u : ~0u << (shift + 8);
            for (int i = tid; i < N; i += 1024) {
                const int di = d[i];
                if (di >= 0 && ((unsigned)di & himask) == prefix) atomicAdd(&hist[((unsigned)di >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {      // wave 0: bins 4 lane .. 4 lane + 3, inclusive scan of the lane sums, first bin whose running count exceeds `remaining`
                const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                const int incl = waveInclusiveScan(c0 + c1 + c2 + c3), before = incl - (c0 + c1 + c2 + c3);
                if (before <= remaining && remaining < incl) {      // exactly one lane
                    int b = 0, cum = before;
                    if (remaining >= cum + c0) { cum += c0; b = 1; if (remaining >= cum + c1) { cum += c1; b = 2; if (remaining >= cum + c2) { cum += c2; b = 3; } } }
                    sSel[0] = 4 * tid + b; sSel[1] = cum;
                }
            }
            __syncthreads();
            prefix |= (unsigned)sSel[0] << shift;
            remaining -= sSel[1];
        }
        if (tid == 0) sMedian = (int)prefix;
    }
    __syncthreads();
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), (float)sMedian);
    int kept = 0;
    for (int i = tid; i < N; i += 1024) {
        const int di = d[i];
        if (di < 0) continue;
        if ((float)di < thDist) kept++;
        else { uRight[base + i] = -1.0f; depth[base + i] = -1.0f; }
    }
    if (kept) atomicAdd(&sKept, kept);
    __syncthreads();
    if (tid == 0) nMatched[pair] = sKept;
}

void launchStereo(hipStream_t st, const LevelGeom* lv, const uint8_t* pyr, const Keypoint* kps, const uint8_t* desc,
                  const int* nOut, const StereoParams& sp, int rows, int* rowOff, unsigned short* rowList, float* uRight,
                  float* depth, int* sadDist, int* nMatched, int nPairs) {
    hipLaunchKernelGGL(k_stereo_rows, dim3(nPairs), dim3(1024), (size_t)(rows + 1) * sizeof(int), st, kps, nOut, sp, rows,
                       rowOff, rowList);
    hipLaunchKernelGGL(k_stereo_match, dim3((sp.capacity + 3) / 4, nPairs), dim3(256), 0, st, lv, pyr, kps, desc, nOut, sp,
                       rows, rowOff, rowList, uRight, depth, sadDist);
    hipLaunchKernelGGL(k_stereo_filter, dim3(nPairs), dim3(1024), (size_t)((sp.capacity + 3) & ~3) * sizeof(int), st, nOut, sp, uRight,
                       depth, sadDist, nMatched);
}

}  // namespace orbx
