// k_blur.hip — 7x7 sigma-2 integer Gaussian of every pyramid level (reference ORBextractor.cc:1126-1127).
//
// Arithmetic: taps {18,34,49,55,49,34,18} per axis, row sums first (<= 257*255 = 65535),
// then (column sum + 2^15) >> 16, saturated to 255.  The bordered pyramid already holds the
// BORDER_REFLECT_101 frame of the level (19 px >= the 3 px the blur reaches over the edge), so the
// kernel never special-cases an image edge.
//
// HBM-bound design: a lane owns 4 adjacent output columns and walks down a block of kBlurBlockRows rows, keeping the
// last seven rows of horizontal sums in registers.  Per row it reads three aligned dwords (12 pixels, two of
// the three served by L1), forms the four horizontal sums with v_alignbyte + v_dot4_u32_u8 (2 dot products
// per sum), the four vertical sums with 24-bit mads, and stores one dword: no LDS, every HBM byte of the
// level is read once (plus the 6-row halo per block) and every blurred byte is written once, coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"
#include "k_blur_body.hpp"      // blurLanes<ROWS>: the lanes themselves (also appended to the FAST grid in small batches, k_fast.hip)

namespace orbx {

// output rows per lane: kBlurBlockRows (32) for throughput, kBlurBlockRowsSmall for small batches, where the grid cannot fill the
// chip anyway and a lane's chain of dependent row loads (ROWS + 6 of them) is the kernel's duration

template <int kBlurRows>
__global__ __launch_bounds__(256) void k_blur(const BlurItem* __restrict__ items, const unsigned short* __restrict__ laneItem,
                                               int nLanes, const LevelGeom* __restrict__ lv,
                                               const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int f0, int nFrames) {
    int chunk, fr;
    if (!xcdChunkFrame(nFrames, chunk, fr)) return;   // all row blocks of a frame on one XCD: their 6 halo rows hit its L2
    blurLanes<kBlurRows>(items, laneItem, nLanes, lv, pyr, blur, chunk, f0 + fr);
}

void launchBlur(hipStream_t st, const BlurItem* items, const unsigned short* laneItem, int nLanes, int blockRows, const LevelGeom* lv,
                const uint8_t* pyr, uint8_t* blur, int f0, int B) {
    if (blockRows == kBlurBlockRows)
        hipLaunchKernelGGL(k_blur<kBlurBlockRows>, xcdGrid((nLanes + 255) / 256, B), dim3(256), 0, st, items, laneItem, nLanes, lv, pyr, blur, f0, B);
    else
        hipLaunchKernelGGL(k_blur<kBlurBlockRowsSmall>, xcdGrid((nLanes + 255) / 256, B), dim3(256), 0, st, items, laneItem, nLanes, lv, pyr, blur, f0, B);
}

}  // namespace orbx
