// k_describe.hip — IC_Angle orientation, rotated BRIEF and final placement
// (reference ORBextractor.cc:75-145, 1137-1158).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"
#define ORBX_DESCRIBE_TU 1
#include "k_describe_body.hpp"

namespace orbx {

template <bool PB>
__global__ __launch_bounds__(256) void k_describe(const LevelGeom* __restrict__ lv, int nlevels,
                                                   const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                   const uint2* __restrict__ sel, int selPerFrame,
                                                   const int* __restrict__ levelCount, const int* __restrict__ levelLap,
                                                   Keypoint* __restrict__ outK, uint8_t* __restrict__ outD, int capacity,
                                                   int* __restrict__ nOut, int* __restrict__ monoOut,
                                                   Keypoint* __restrict__ outLevelK, int* __restrict__ outLevelCounts, int f0, int nFrames, int fewWaves,
                                                   int chunk0, int levelLo, int levelHi) {
    __shared__ __align__(16) uint8_t smem[DescLds<PB>::kPatches];
    __shared__ __align__(16) unsigned wtab[2][16][PB ? 12 : 8];   // [m10 weights | disc mask][|v|][dword of the aligned row]
    int chunk, fr;
    if (!xcdChunkFrame(nFrames, chunk, fr)) return;   // all keypoints of a frame on one XCD: overlapping patches share its L2
    describeBlock<PB>(lv, nlevels, pyr, blur, sel, selPerFrame, levelCount, levelLap, outK, outD, capacity, nOut, monoOut, outLevelK, outLevelCounts, fewWaves,
                      smem, wtab, chunk0 + chunk, f0 + fr, levelLo, levelHi);
}

// The keypoints of levels [levelLo, levelHi) of B frames: their selection slots are [slotLo, slotHi) (LevelGeom::selOff).  One launch covers every
// level; two launches (levels below the split with the blur per keypoint, the rest from the blurred levels) cover a blur that is split by level
// (orbx_api.cpp) - a workgroup that straddles the boundary appears in both, and each of its waves runs in the launch its level belongs to.
void launchDescribe(hipStream_t st, const LevelGeom* lv, int nlevels, const uint8_t* pyr, const uint8_t* blur,
                    const uint2* sel, int selPerFrame, const int* levelCount, const int* levelLap, Keypoint* outK,
                    uint8_t* outD, int capacity, int* nOut, int* monoOut, Keypoint* outLevelK, int* outLevelCounts,
                    bool patchBlur, int levelLo, int levelHi, int slotLo, int slotHi, int f0, int B) {
    const int perBlock = 2 * kDescWaves, chunk0 = slotLo / perBlock, groups = (slotHi + perBlock - 1) / perBlock - chunk0;
    if (groups <= 0) return;
    const int fewWaves = (long long)groups * B <= 8 * 256;      // at most eight workgroups per CU in the whole launch: latency form of the level sums
    if (patchBlur)
        hipLaunchKernelGGL(k_describe<true>, xcdGrid(groups, B), dim3(256), 0, st, lv, nlevels,
                           pyr, blur, sel, selPerFrame, levelCount, levelLap, outK, outD, capacity, nOut, monoOut, outLevelK,
                           outLevelCounts, f0, B, fewWaves, chunk0, levelLo, levelHi);
    else
        hipLaunchKernelGGL(k_describe<false>, xcdGrid(groups, B), dim3(256), 0, st, lv, nlevels,
                           pyr, blur, sel, selPerFrame, levelCount, levelLap, outK, outD, capacity, nOut, monoOut, outLevelK,
                           outLevelCounts, f0, B, fewWaves, chunk0, levelLo, levelHi);
}
bool checkUmax(const int* umax16) {      // the static device table is the one the reference's constructor computes
    for (int i = 0; i < 16; i++) if (umax16[i] != kUmaxStatic[i]) return false;
    return true;
}


}  // namespace orbx
