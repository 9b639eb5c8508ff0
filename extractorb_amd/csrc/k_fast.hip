// k_fast.hip — per-cell FAST-9/16 detection with threshold retry and NMS (reference ORBextractor.cc:797-864).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {
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
                                               unsigned* __restrict__ candPos, unsigned* __restrict__ candOrd,
                                               unsigned* __restrict__ candCount,
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
    unsigned* outPos = candPos + g.candOff + (long long)f * g.candCap;
    unsigned* outOrd = candOrd + g.candOff + (long long)f * g.candCap;
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
                const unsigned at = base + before;
                if (at < (unsigned)g.candCap) {
                    outPos[at] = px | (py << 12) | ((unsigned)(s - 1) << 24);                       // response = S - 1
                    outOrd[at] = ((unsigned)c.cellId << 12) | ((unsigned)y << 6) | (unsigned)x;    // reference list order
                }
            }
            base += __popcll(m);
            x += 64;
            while (x >= cw) { x -= cw; y++; }
        }
    }
}

void fastLdsLayout(int maxRoiW, int maxRoiH, int* tileStride, int* tileBytes, int* scoreStride, int* scoreBytes) {
    *tileStride = (maxRoiW + 3) / 4 * 4 + 4;
    *tileBytes = (*tileStride * maxRoiH + 15) / 16 * 16;
    *scoreStride = (maxRoiW - 6 + 2 + 3) / 4 * 4 + 4;
    *scoreBytes = (*scoreStride * (maxRoiH - 6 + 2) + 15) / 16 * 16;
}
void launchFast(hipStream_t st, const CellDesc* cells, int nCells, const LevelGeom* lv, int nlevels,
                const uint8_t* pyr, int iniTh, int minTh, unsigned* candPos, unsigned* candOrd, unsigned* candCount,
                int maxRoiW, int maxRoiH, int B) {
    int ts, tb, ss, sb;
    fastLdsLayout(maxRoiW, maxRoiH, &ts, &tb, &ss, &sb);
    const size_t lds = (size_t)kFastWaves * (tb + sb);
    hipLaunchKernelGGL(k_fast, dim3((nCells + kFastWaves - 1) / kFastWaves, B), dim3(256), lds, st, cells, nCells, lv,
                       nlevels, pyr, iniTh, minTh, candPos, candOrd, candCount, ts, tb, ss, sb);
}
// unpack one level's candidates into reference KeyPoints (introspection for tests)
__global__ void k_unpackCandidates(const unsigned* __restrict__ keys, int n, Keypoint* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned w = keys[i];
    Keypoint k;
    k.x = (float)(w & 0xfff); k.y = (float)((w >> 12) & 0xfff); k.size = 7.f; k.angle = -1.f;
    k.response = (float)(w >> 24); k.octave = 0; k.class_id = -1;
    out[i] = k;
}
void launchUnpackCandidates(hipStream_t st, const unsigned* keys, int n, Keypoint* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpackCandidates, dim3((n + 255) / 256), dim3(256), 0, st, keys, n, out);
}


}  // namespace orbx
