// k_describe_body.hpp - IC_Angle, rotated BRIEF and the final placement of one workgroup's eight keypoints as a device function; k_describe.hip
// launches it.  See k_describe.hip for the algorithm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"
#include "k_blur_body.hpp"      // blurRun: the patch-blur form blurs a keypoint's 37 x 37 patch out of its raw 43 x 43 tile

namespace orbx {
// ================================================================================================
// IC_Angle + rotated BRIEF + final placement.  One wave64 per kept keypoint.
// ================================================================================================
// Both tables are the same for every extractor (HALF_PATCH_SIZE = 15 and the learned pattern are compile-time constants of the reference,
// ORBextractor.cc:71, 148-406), so they are initialised statically: no upload, nothing for a second handle's creation to overwrite while a
// first handle's kernels read them.  checkUmax compares the static c_umax with the table the host derives by the reference's formula (:459-474).
constexpr int kUmaxStatic[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
static __constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
static __constant__ __attribute__((aligned(16))) float c_patternF[1024] = {   // the rBRIEF pattern as floats
#include "orbx_brief_pattern.inc"
};

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
    // sin(y) and cos(y) = "sin" of quadrants n and n + 1; a quadrant takes the sine polynomial when even and the cosine
    // polynomial when odd, so exactly one of each is evaluated and the parity of n says which result is which.
    const bool odd = (n & 1) != 0;
    const int ns = odd ? n + 1 : n, nc = odd ? n : n + 1;
    const double sgnS = ((ns & 3) == 1 || (ns & 3) == 2) ? -1.0 : 1.0;   // sign[ns & 3]
    const float ps = (float)polySin(x * sgnS), pc = (float)polyCos((nc & 2) ? -1.0 : 1.0);
    *s_out = odd ? pc : ps;
    *c_out = odd ? ps : pc;
}

constexpr int kDescWaves = 4;                         // waves per workgroup; each half-wave (32 lanes) owns one keypoint
constexpr int kBriefReach = 18;                       // |rounded rotated pattern coordinate| <= 18 (max radius 18.385)
constexpr int kRawRows = 2 * kHalfPatch + 1;          // 31
constexpr int kRawStride = 40;                        // 3 + 31 bytes -> 9 dwords, staged as five 8-byte pairs
constexpr int kBlurRows = 2 * kBriefReach + 1;        // 37
constexpr int kBlurStride = 40;                       // 3 + 37 bytes -> 10 dwords
constexpr int kPatchLds = kRawRows * kRawStride + kBlurRows * kBlurStride;   // 2596 bytes per keypoint (dword multiple)
// Patch-blur form (PB): ONE raw tile of 43 rows (v = -21 .. 21) x 44 bytes (byte t <-> u = t - 22: one spare column in front, so that the blur's
// aligned dword triples x0 - 4 .. x0 + 7 of output group g start at dword g), re-aligned to the patch when it is staged; the 37 x 37 blurred
// patch the descriptor reads is computed from it into the blurred tile (same layout as above, no misalignment).
constexpr int kPbRows = 2 * (kBriefReach + 3) + 1;    // 43
constexpr int kPbStride = 44;                         // 11 dwords
constexpr int kPbRawBytes = (kPbRows * kPbStride + 15) & ~15;               // 1904
constexpr int kPbLds = kPbRawBytes + kBlurRows * kBlurStride;               // 3384 bytes per keypoint
// The patch blur's lanes (round 5).  The rotated pattern only reaches a DISC of the 37 x 37 blurred patch: a sample (row r, column q) is the
// rounding of a point at most sqrt(338) = 18.385 from the keypoint (the largest |(x, y)| of the pattern, ORBextractor.cc:148-406), so column
// group g (columns q = 4 g - 18 .. 4 g - 15) is read at |r| <= rmax(g) = floor(1/2 + sqrt(338.5 - max(min|q| - 1/2, 0)^2)) only: 11, 15, 17, 18,
// 18, 18, 18, 16, 13, 6 -> 310 of the box's 370 (group, row) items; tests/test_tables.py derives the bounds from the pattern and sweeps every
// 0.0005 degrees for samples outside them.  The 32 lanes of a keypoint take (group, first row, rows) runs of at most 12 rows (six trips of the
// two-row blur loop; the box dealt as 10 groups x 3 runs of 13 was seven): entry = group | first row << 8 | rows << 16.  WHERE the runs of a group are cut
// matters: the 32 lanes read tile rows (first row + i) x 11 dwords + group, and a first cut (runs of 10 / 12 rows from the top) had them collide in the LDS
// banks 4.4-fold on average (SQ_LDS_BANK_CONFLICT: 514 cycles per wave, the old 10 x 3 layout: 2-fold) - which ate the instructions the disc saves.  The cuts
// below come from a search (simulated annealing over the partitions; cost = sum over the loop's LDS instructions of the largest number of lanes on one
// bank, loads of the raw tile and stores of the blurred one): 143 against 259 for the first cut and 66 for a conflict-free walk.
static __constant__ unsigned c_pbRun[32] = {
    0x0b0700, 0x0c1200, 0x080301, 0x0b0b01, 0x0c1601, 0x0c0102, 0x0c0d02, 0x0b1902, 0x050003, 0x0c0503, 0x091103, 0x0b1a03, 0x050004, 0x0c0504, 0x081104,
    0x0c1904, 0x060005, 0x0c0605, 0x091205, 0x0a1b05, 0x050006, 0x0c0506, 0x0c1106, 0x081d06, 0x0b0207, 0x0b0d07, 0x0b1807, 0x0c0508, 0x0c1108, 0x031d08, 0x010c09, 0x0c0d09};

// One half-wave (32 lanes) per kept keypoint; the two keypoints of a wave share a level (selOff is even):
//   * both patches are staged in LDS with aligned dword loads that are all in flight at once: a half-wave covers
//     three patch rows per step (lane = (row % 3, dword column)), so the address of step s is one add away from
//     step 0's and the LDS destination is an immediate offset;
//   * IC_Angle: lane = patch row; the row's 31 pixels are byte-aligned with v_alignbyte and reduced with
//     v_dot4_u32_u8 against per-row weight words (u+16 inside the disc, 0 outside) — sum(u*I) = dot(I, u+16) - 16*dot(I, 1);
//   * rBRIEF: lane = 8 of the 256 test pairs; a ballot per group of 32 pairs packs 4 descriptor bytes of each keypoint.
#if defined(ORBX_DESC_STAMPS) && defined(ORBX_DESCRIBE_TU)      // (diagnostic builds stamp the kernel of k_describe.hip only)
// diagnostic build (tools/desc_spans.py): stage stamps of every wave of frame 0, s_memrealtime ticks
__device__ unsigned long long g_descStamps[6 * 1024];
extern "C" int orbx_debug_desc_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_descStamps), sizeof(g_descStamps)); }
#define DSTAMP(i) do { const int dsW = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6); \
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.z == 0 && dsW < 1024) g_descStamps[6 * dsW + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DSTAMP(i) do {} while (0)
#endif
// PB (patch blur, large frames with few features per pixel: orbx_api.cpp): the blurred LEVELS are never made - a half-wave stages the raw
// 43 x 43 patch around its keypoint (the bordered pyramid's REFLECT_101 frame is the reflection of the level itself, which is what the
// reference's border-less clone blurs against: ORBextractor.cc:1126-1127), blurs the 37 x 37 patch the rotated pattern can reach with the same
// integer arithmetic (blurRun) and describes from that.  1920x1080 x 2000 features: 2000 x 43 x 37 = 3.2 M pixels of horizontal pass against
// 6.4 M for the whole pyramid, no 6.4-MB write + 5.5-MB sparse re-read of blurred levels per frame.
// LDS of one description workgroup (eight keypoints): the patches, then the weight words of IC_Angle's rows.
template <bool PB>
struct DescLds {
    static constexpr int kPatches = 2 * kDescWaves * (PB ? kPbLds : kPatchLds), kWtabWords = 2 * 16 * (PB ? 12 : 8);
    static constexpr int kWtabOff = (kPatches + 15) & ~15, kBytes = kWtabOff + 4 * kWtabWords;
};

// One workgroup = eight keypoint slots [8 chunk, 8 chunk + 8) of frame f.  smem: the patches (DescLds::kPatches bytes), wtab: the weight words.
// One workgroup barrier (behind the weight table).  A wave whose level lies outside [levelLo, levelHi) belongs to another launch and leaves.
template <bool PB>
__device__ __forceinline__ void describeBlock(const LevelGeom* __restrict__ lv, int nlevels,
                                              const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                              const uint2* __restrict__ sel, int selPerFrame,
                                              const int* __restrict__ levelCount, const int* __restrict__ levelLap,
                                              Keypoint* __restrict__ outK, uint8_t* __restrict__ outD, int capacity,
                                              int* __restrict__ nOut, int* __restrict__ monoOut,
                                              Keypoint* __restrict__ outLevelK, int* __restrict__ outLevelCounts, int fewWaves,
                                              uint8_t* smem, unsigned (*wtab)[16][PB ? 12 : 8], int chunk, int f, int levelLo, int levelHi) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, hl = lane & 31;
    DSTAMP(0);
    if constexpr (PB) {   // weight words of row |v| = a over the re-aligned tile row: byte t = 4 j + b <-> u = t - 22
        for (int e = tid; e < 2 * 16 * 12; e += 256) {
            const int which = e / 192, a = (e / 12) & 15, j = e % 12;
            unsigned w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int u = 4 * j + b - (kBriefReach + 4), au = u < 0 ? -u : u;
                if (au <= kHalfPatch && au <= c_umax[a]) w |= (unsigned)(which ? 1 : u + 16) << (8 * b);
            }
            wtab[which][a][j] = w;
        }
    } else {   // weight words of row |v| = a, bytes k = 0..31 <-> u = k - 15
        const int which = tid >> 7, a = (tid >> 3) & 15, j = tid & 7;
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int u = 4 * j + b - kHalfPatch, au = u < 0 ? -u : u;
            if (au <= c_umax[a]) w |= (unsigned)(which ? 1 : u + 16) << (8 * b);
        }
        wtab[which][a][j] = w;
    }
    __syncthreads();
    const int slot0 = __builtin_amdgcn_readfirstlane((chunk * kDescWaves + wave) * 2);   // wave-uniform
    const int slot = slot0 + half;
    // totals of this frame and the level this wave belongs to.  Two forms, by the launch (wave-uniform):
    //  * few waves on the chip (small batches: the wave's own latency is the launch's): lane l < nlevels loads level l's first slot, count and
    //    lapping count - one memory round trip for all levels -, running sums by four DPP steps inside the first row of 16 lanes (nlevels <=
    //    kMaxLevels = 16), the wave's level by a ballot, its sums by v_readlane: one frame 37.0 -> 36.2 us;
    //  * a full chip: a scalar loop - nlevels dependent round trips through the scalar cache, hidden behind the other waves, while three more
    //    vector loads per wave cost L1 look-ups, which is what this kernel is short of (512 frames: 344 against 355 us with the form above).
    int total = 0, totalLap = 0, level = 0, seqBase = 0, lapBase = 0, levelN, selOff;
    if (fewWaves) {
        int myOff = 0x7fffffff, myCnt = 0, myLap = 0;
        if (lane < nlevels) { myOff = lv[lane].selOff; myCnt = levelCount[f * nlevels + lane]; myLap = levelLap[f * nlevels + lane]; }
        int incC = myCnt, incL = myLap;
        static_assert(kMaxLevels == 16, "the scan below covers one DPP row of 16 lanes");
        incC += __builtin_amdgcn_update_dpp(0, incC, 0x111, 0xF, 0xF, true); incL += __builtin_amdgcn_update_dpp(0, incL, 0x111, 0xF, 0xF, true);      // row_shr:1
        incC += __builtin_amdgcn_update_dpp(0, incC, 0x112, 0xF, 0xF, true); incL += __builtin_amdgcn_update_dpp(0, incL, 0x112, 0xF, 0xF, true);      // row_shr:2
        incC += __builtin_amdgcn_update_dpp(0, incC, 0x114, 0xF, 0xF, true); incL += __builtin_amdgcn_update_dpp(0, incL, 0x114, 0xF, 0xF, true);      // row_shr:4
        incC += __builtin_amdgcn_update_dpp(0, incC, 0x118, 0xF, 0xF, true); incL += __builtin_amdgcn_update_dpp(0, incL, 0x118, 0xF, 0xF, true);      // row_shr:8
        total = __builtin_amdgcn_readlane(incC, kMaxLevels - 1); totalLap = __builtin_amdgcn_readlane(incL, kMaxLevels - 1);      // (lanes >= nlevels add 0)
        level = __popcll(__ballot(slot0 >= myOff)) - 1;      // (level 0's first slot is 0: level >= 0; the first slots ascend)
        seqBase = __builtin_amdgcn_readlane(incC - myCnt, level); lapBase = __builtin_amdgcn_readlane(incL - myLap, level);
        levelN = __builtin_amdgcn_readlane(myCnt, level); selOff = __builtin_amdgcn_readlane(myOff, level);
        if (slot0 == 0 && outLevelCounts && lane < nlevels) outLevelCounts[f * nlevels + lane] = myCnt;
    } else {
        for (int l = 0; l < nlevels; l++) {
            const int c = levelCount[f * nlevels + l], lp = levelLap[f * nlevels + l];
            if (slot0 >= lv[l].selOff) { level = l; seqBase = total; lapBase = totalLap; }
            total += c;
            totalLap += lp;
        }
        selOff = lv[level].selOff;
        levelN = levelCount[f * nlevels + level];
        if (slot0 == 0 && outLevelCounts && lane < nlevels) outLevelCounts[f * nlevels + lane] = levelCount[f * nlevels + lane];
    }
    if (slot0 == 0 && lane == 0) { nOut[f] = total; monoOut[f] = total - totalLap; }      // monoIndex after the loop (:1161)
    if (level < levelLo || level >= levelHi) return;      // (wave-uniform: the other launch of a blur split by level describes this level)
    const int i = slot - selOff;
    const bool active = slot < selPerFrame && i < levelN;
    if (__ballot(active) == 0) return;
    DSTAMP(1);
    const int gw = lv[level].w, gh = lv[level].h, pyrStride = lv[level].pyrStride, blurStride = lv[level].blurStride;
    const uint8_t* pyrL = pyr + lv[level].pyrOff + (long long)f * lv[level].pyrFrameBytes;      // wave-uniform bases:
    const uint8_t* blurL = blur + lv[level].blurOff + (long long)f * lv[level].blurFrameBytes;   // lanes add 32-bit offsets
    const uint2 e = active ? sel[(long long)f * selPerFrame + slot] : make_uint2(0u, 0u);
    int kx = e.x & 0xffff, ky = e.x >> 16;      // (orbx_device.hpp: the selection entry)
    const float response = (float)(e.y >> 24);
    // the quad-tree only emits points of the FAST rectangle; clamp anyway so a corrupted entry (or an idle half)
    // can never turn into an out-of-bounds gather
    kx = min(max(kx, kEdge), gw - kEdge - 1);
    ky = min(max(ky, kEdge), gh - kEdge - 1);

    DSTAMP(2);
    int m10 = 0, m01 = 0;
    uint8_t* blurT;
    int blurMis;
    unsigned pbRun = 0;
    if constexpr (PB) pbRun = c_pbRun[hl];      // (requested ahead of the tile's loads)
    if constexpr (PB) {
        // ---- stage the raw 43 x 44 tile, re-aligned: 8 lanes per tile row (6 load an 8-byte pair of source dwords - 4-byte aligned: one
        //      global_load_dwordx2 -, 11 re-aligned dwords stored), four rows per step: 11 load instructions per wave (dword loads: 22; the gather is bound
        //      by the L1's look-up rate: profiles/r04_l1_l2_counters.md) ----
        uint8_t* rawT = smem + (wave * 2 + half) * kPbLds;
        blurT = rawT + kPbRawBytes;
        blurMis = 0;
        constexpr int kSteps = (kPbRows + 3) / 4;            // 11
        const int l8 = hl & 7, rsub = hl >> 3;
        const int col0 = kPadL + kx - (kBriefReach + 4), mis = col0 & 3;
        const int rOff = __mul24(kEdge + ky - (kBriefReach + 3) + rsub, pyrStride) + (col0 - mis) + 8 * l8;
        uint2 wr[kSteps];
#pragma unroll
        for (int s = 0; s < kSteps; s++) {
            wr[s] = uint2{0u, 0u};
            if (l8 < 6 && rsub + 4 * s < kPbRows) __builtin_memcpy(&wr[s], pyrL + ((unsigned)rOff + (unsigned)(4 * s * pyrStride)), 8);
        }
        uint8_t* rdst = rawT + rsub * kPbStride + 8 * l8;
#pragma unroll
        for (int s = 0; s < kSteps; s++) {
            const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wr[s].x, 0x101, 0xF, 0xF, false);      // row_shl:1: the next pair's first dword (same tile row for l8 < 5)
            const unsigned v0 = __builtin_amdgcn_alignbyte(wr[s].y, wr[s].x, (unsigned)mis), v1 = __builtin_amdgcn_alignbyte(nxt, wr[s].y, (unsigned)mis);
            if (l8 < 6 && rsub + 4 * s < kPbRows) {
                *(unsigned*)(rdst + 4 * s * kPbStride) = v0;
                if (l8 < 5) *(unsigned*)(rdst + 4 * s * kPbStride + 4) = v1;      // (a tile row is 11 dwords: the sixth pair stores its first only)
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        DSTAMP(3);
        // ---- IC_Angle (:75-102): lane = patch row v = hl - 15 = tile row hl + 6; the weight words know the layout (u = t - 22) ----
        if (hl < kRawRows) {
            const int v = hl - kHalfPatch, a = v < 0 ? -v : v;
            const unsigned* row = (const unsigned*)(rawT + (hl + 6) * kPbStride);
            unsigned s1 = 0, s0 = 0;
#pragma unroll
            for (int j = 1; j <= 9; j++) {          // bytes 7 .. 37 lie in dwords 1 .. 9
                const unsigned px = row[j];
                s1 = __builtin_amdgcn_udot4(px, wtab[0][a][j], s1, false);
                s0 = __builtin_amdgcn_udot4(px, wtab[1][a][j], s0, false);
            }
            m10 = (int)s1 - 16 * (int)s0;     // sum u*I
            m01 = v * (int)s0;                // v * sum I
        }
        // ---- the 7x7 blur of the part of the 37 x 37 patch the rotated pattern can reach (:1126-1127): lane = one (column group, run of rows) of
        //      c_pbRun; output row o reads tile rows o .. o + 6 ----
        {
            const int g4 = (int)(pbRun & 0xff), o0 = (int)((pbRun >> 8) & 0xff), nOut = (int)(pbRun >> 16);
            const uint8_t* src = rawT + o0 * kPbStride + 4 * g4;
            uint8_t* dst = blurT + o0 * kBlurStride + 4 * g4;
            blurRun(nOut,
                    [&](int i, unsigned& d0, unsigned& d1, unsigned& d2) {
                        const unsigned* row = (const unsigned*)(src + min(i, nOut + 5) * kPbStride);
                        d0 = row[0]; d1 = row[1]; d2 = row[2];
                    },
                    [&](int rr, unsigned word) { *(unsigned*)(dst + rr * kBlurStride) = word; });
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    } else {
    // ---- stage both patches: raw level rows/cols +-15 (IC_Angle), blurred level rows/cols +-18 (rBRIEF) ----
    uint8_t* rawT = smem + (wave * 2 + half) * kPatchLds;
    blurT = rawT + kRawRows * kRawStride;
    const int rawCol0 = kPadL + kx - kHalfPatch, rawMis = rawCol0 & 3;
    const int blurCol0 = kx - kBriefReach;
    blurMis = blurCol0 & 3;
    {
        // 8-byte loads (4-byte aligned: one global_load_dwordx2 each): a half-wave covers SIX tile rows per step (lane = (row % 6, dword pair)), 6 + 7
        // load instructions per wave instead of 11 + 13 with dword loads - the patch gather is bound by the L1's rate of line look-ups (0.9 per cycle
        // and CU in this kernel: profiles/r04_l1_l2_counters.md), and a wave instruction's look-ups go with its 16-lane groups x the lines each touches
        constexpr int kPairs = 5, kRowsPerStep = 6;
        static_assert(kRawStride == 8 * kPairs && kBlurStride == 8 * kPairs, "tile rows are five 8-byte pairs");
        constexpr int kRawSteps = (kRawRows + kRowsPerStep - 1) / kRowsPerStep, kBlurSteps = (kBlurRows + kRowsPerStep - 1) / kRowsPerStep;   // 6, 7
        const int rr = (hl * 52) >> 8, rc = hl - rr * kPairs;     // hl / 5 for hl < 32: lanes 0..29: row (mod 6) and dword pair of both tiles
        const bool stLane = hl < kRowsPerStep * kPairs;
        // (every aligned dword that starts inside a level's padded row lies inside it: strides are multiples of 64; the pairs reach at most 9 bytes
        //  past the raw patch - inside the 19-px border - and 3 + 3 bytes past the blurred one - inside the row, kx <= w - 20)
        const int rOff = __mul24(kEdge + ky - kHalfPatch + rr, pyrStride) + (rawCol0 - rawMis) + 8 * rc;   // 24-bit: full-rate multiplies
        const int bOff = __mul24(ky - kBriefReach + rr, blurStride) + (blurCol0 - blurMis) + 8 * rc;
        uint2 wr[kRawSteps], wb[kBlurSteps];
#pragma unroll
        for (int s = 0; s < kRawSteps; s++) {
            wr[s] = uint2{0u, 0u};
            if (stLane && rr + kRowsPerStep * s < kRawRows) __builtin_memcpy(&wr[s], pyrL + ((unsigned)rOff + (unsigned)(kRowsPerStep * s * pyrStride)), 8);   // uniform base + u32 offset
        }
#pragma unroll
        for (int s = 0; s < kBlurSteps; s++) {
            wb[s] = uint2{0u, 0u};
            if (stLane && rr + kRowsPerStep * s < kBlurRows) __builtin_memcpy(&wb[s], blurL + ((unsigned)bOff + (unsigned)(kRowsPerStep * s * blurStride)), 8);
        }
        uint8_t* rdst = rawT + rr * kRawStride + 8 * rc;
        uint8_t* bdst = blurT + rr * kBlurStride + 8 * rc;
#pragma unroll
        for (int s = 0; s < kRawSteps; s++)
            if (stLane && rr + kRowsPerStep * s < kRawRows) *(uint2*)(rdst + kRowsPerStep * s * kRawStride) = wr[s];
#pragma unroll
        for (int s = 0; s < kBlurSteps; s++)
            if (stLane && rr + kRowsPerStep * s < kBlurRows) *(uint2*)(bdst + kRowsPerStep * s * kBlurStride) = wb[s];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    DSTAMP(3);
    // ---- IC_Angle (:75-102): integer moments over the radius-15 disc of the unblurred level ----
    if (hl < kRawRows) {
        const int v = hl - kHalfPatch, a = v < 0 ? -v : v;
        const unsigned* row = (const unsigned*)(rawT + hl * kRawStride);
        unsigned d[9];
#pragma unroll
        for (int j = 0; j < 9; j++) d[j] = row[j];
        unsigned s1 = 0, s0 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // bytes u = 4j-15 .. 4j-12 of the row: shift the dword pair by the patch's misalignment
            const unsigned px = __builtin_amdgcn_alignbyte(d[j + 1], d[j], (unsigned)rawMis);   // the shift comes from a register
            s1 = __builtin_amdgcn_udot4(px, wtab[0][a][j], s1, false);
            s0 = __builtin_amdgcn_udot4(px, wtab[1][a][j], s0, false);
        }
        m10 = (int)s1 - 16 * (int)s0;     // sum u*I
        m01 = v * (int)s0;                // v * sum I
    }
    }
    // reduce inside the half-wave: four DPP steps cover a row of 16 lanes, one cross-lane exchange joins the two rows
    auto rowSum = [](int v) {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);    // row_half_mirror
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);    // row_mirror
        return v;
    };
    m10 = rowSum(m10); m01 = rowSum(m01);
    m10 += __shfl_xor(m10, 16);
    m01 += __shfl_xor(m01, 16);
    const float angle = fastAtan2Deg((float)m01, (float)m10);

    // ---- computeOrbDescriptor (:106-145) on the blurred level ----
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);   // (float)(CV_PI/180.f)
    float a, b;
    sincosGlibc(__fmul_rn(angle, factorPI), &b, &a);
    const uint8_t* bc = blurT + kBriefReach * kBlurStride + blurMis + kBriefReach;
    unsigned myWord = 0;                   // lane hl < 8 ends up holding descriptor bytes 4*hl .. 4*hl+3 of its keypoint
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = hl + 32 * j;        // test pair index; bit (p & 7) of descriptor byte (p >> 3)
        const float4 pt = ((const float4*)c_patternF)[p];
        const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
        const int r0 = (int)rintf(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = (int)rintf(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = (int)rintf(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = (int)rintf(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[__mul24(r0, kBlurStride) + q0], t1 = bc[__mul24(r1, kBlurStride) + q1];
        const unsigned long long m = __ballot(t0 < t1);
        const unsigned mine = half ? (unsigned)(m >> 32) : (unsigned)m;
        myWord = hl == j ? mine : myWord;
    }
    DSTAMP(4);
    if (!active) return;

    // ---- placement (:1137-1158): non-lapping keys fill from the front, lapping keys from the back ----
    const int lapRank = (int)(e.y & 0x7fffff), isLap = (int)((e.y >> 23) & 1);
    const int lapBefore = lapBase + lapRank;
    const int monoBefore = (seqBase - lapBase) + (i - lapRank);
    const int at = isLap ? total - 1 - lapBefore : monoBefore;
    float ox = (float)kx, oy = (float)ky;
    const float scale = lv[level].scale;
    if (level != 0) { ox = __fmul_rn(ox, scale); oy = __fmul_rn(oy, scale); }
    const float patchSize = (float)lv[level].patchSize;
    if (at < capacity) {
        if (hl == 0) {
            Keypoint k;
            k.x = ox; k.y = oy; k.size = patchSize; k.angle = angle; k.response = response;
            k.octave = level; k.class_id = -1;
            outK[(long long)f * capacity + at] = k;
        }
        if (hl < 8) ((unsigned*)(outD + ((long long)f * capacity + at) * 32))[hl] = myWord;
    }
    if (outLevelK && hl == 0 && seqBase + i < capacity) {
        Keypoint k;
        k.x = (float)kx; k.y = (float)ky; k.size = patchSize; k.angle = angle; k.response = response;
        k.octave = level; k.class_id = -1;
        outLevelK[(long long)f * capacity + seqBase + i] = k;
    }
}


}  // namespace orbx
