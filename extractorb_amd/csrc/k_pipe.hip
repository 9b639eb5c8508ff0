// k_pipe.hip — the stages behind the pyramid as ONE launch per pipeline step: a software pipeline over chunks of the batch
// (reference ORBextractor.cc:1078-1162: one call = all stages).
//
// Why.  Per 512 frames of 640x480 the five launches issue ~1.35-1.5 ms of vector instructions and take 1.70 ms: the difference is the idle share
// of the latency-bound stages (the quad-tree's chain of barriers: ~60 % idle; the description's patch gathers: ~40 %; the blur's row loads:
// ~50 %).  Separate launches cannot win it back: a launch that becomes ready while k_fast fills every wave slot of the chip is starved until
// FAST's grid drains (profiles/r04_step_timeline.md: 477 us for a 57-us quad-tree launch).  What shares the chip with FAST is work that is
// DISPATCHED together with it.  So the batch is cut into chunks of frames and step t of the pipeline is one launch whose workgroups are dealt,
// interleaved by a host-made table, four roles:
//     F  FAST cells of chunk t            (k_fast_body.hpp: fastCell, a wave per cell)
//     B  blur rows of chunk t             (k_blur_body.hpp: blurLanes)
//     O  quad-tree levels of chunk t - 1  (k_octree_body.inc as a device function: one workgroup per (frame, level))
//     D  keypoints of chunk t - 2         (k_describe_body.hpp: describeBlock, eight keypoints per workgroup)
// Every dependency (F -> O -> D, B -> D) points to an EARLIER launch of the same stream: stream order is the only synchronisation, no
// workgroup ever waits for another.  The bodies are the stand-alone kernels' own (same device functions, bit-identical results: the parity
// suite runs the whole batch matrix under ORBX_PIPE=1 and =0); the LDS of a workgroup is the union of the roles' needs and the register
// budget the quad-tree's (80 VGPRs: six workgroups per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"
#include "k_blur_body.hpp"
#include "k_fast_body.hpp"
#include "k_describe_body.hpp"
#include "k_octree_common.hpp"

namespace orbx {

#define STAMP(id) do {} while (0)
#define OCT_AS_ROLE 1
#define OCT_WAVE_PHASE2 1
#define OCT_W 6
#define OCT_SHORT_PHASE2 (OCT_W <= 4)
#define OCT_T 256
#define OCT_NAME(x) x##_pipe
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#undef OCT_W

#ifndef ORBX_PIPE_ROLES
#define ORBX_PIPE_ROLES 15      // (experiments: roles compiled into the launch, bit = role)
#endif
#ifndef ORBX_PIPE_WAVES
#define ORBX_PIPE_WAVES 6       // waves per SIMD the launch is compiled for (6: the quad-tree role's 80 VGPRs)
#endif
constexpr int kPipeOctShared = (sizeof(OctShared_pipe) + 255) & ~255;      // the quad-tree's shared record, in front of its node arrays

template <int TS, int ROWS>
__global__ __launch_bounds__(256, ORBX_PIPE_WAVES) void k_pipe(const PipeArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const PipeRole r = a.roles[blockIdx.y];                          // workgroup-uniform: a scalar load
    const int fr = gridDim.x * blockIdx.z + blockIdx.x;              // the XCD-aware grid of every "items x frames" kernel (orbx_device.hpp: xcdGrid)
    const int role = __builtin_amdgcn_readfirstlane((int)r.role), idx = __builtin_amdgcn_readfirstlane((int)r.index);
    if ((ORBX_PIPE_ROLES & 1) && role == kPipeF) {
        if (fr >= a.fFn) return;
        using L = FastLds<TS, ROWS>;
        LeafTables none{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
        fastCell<TS, ROWS>(a.cells, a.nCells, a.lv, a.pyr, a.iniTh, a.minTh, a.candSeg, a.cellCount, none, lds, lds + L::kScoreOff,
                           (uint8_t(*)[2][64])(lds + L::kCodeOff), idx, a.fF0 + fr);
    } else if ((ORBX_PIPE_ROLES & 2) && role == kPipeB) {
        if (fr >= a.bFn) return;
        blurLanes<kBlurBlockRows>(a.blurItems, a.laneItem, a.nBlurLanes, a.lv, a.pyr, a.blur, idx, a.bF0 + fr);
    } else if ((ORBX_PIPE_ROLES & 4) && role == kPipeO) {
        if (fr >= a.oFn) return;
        LeafTables none{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
        octreeRole_pipe(a.lv, a.nlevels, a.cells, a.nCells, a.candSeg, a.cellCount, a.cellOff, a.candPos, a.candCount, a.nodeOf, a.sel, a.selPerFrame,
                        a.levelCount, a.levelLap, a.lapArea, a.M, a.P, a.R, a.XT, 0, 0, nullptr, 0ull, none,
                        lds + kPipeOctShared, *(OctShared_pipe*)lds, idx, a.oF0 + fr);
    } else if ((ORBX_PIPE_ROLES & 8) && role == kPipeD) {
        if (fr >= a.dFn) return;
        using L = DescLds<false>;
        describeBlock<false>(a.lv, a.nlevels, a.pyr, a.blur, a.sel, a.selPerFrame, a.levelCount, a.levelLap, a.outK, a.outD, a.capacity, a.nOut, a.monoOut,
                             a.outLevelK, a.outLevelCounts, 0, lds, (unsigned(*)[16][8])(lds + L::kWtabOff), idx, a.dF0 + fr);
    }
}

// LDS bytes of a workgroup of the pipelined launch: the largest role's
size_t octreeLdsBytes(int M, int P, int R, int XT);
size_t pipeLdsBytes(int M, int P, int R, int XT) {
    size_t b = (size_t)FastLds<48, 45>::kBytes;
    b = std::max(b, (size_t)DescLds<false>::kBytes);
    if (ORBX_PIPE_ROLES & 4) b = std::max(b, (size_t)kPipeOctShared + octreeLdsBytes(M, P, R, XT));
    return (b + 255) & ~(size_t)255;
}
bool pipeCanRun(int maxRoiW, int maxRoiH) { return maxRoiW <= 45 && maxRoiH <= 45; }

// itemsPerFrame = entries of the role table; frames = the largest frame count among the roles of this step
void launchPipe(hipStream_t st, const PipeArgs& a, int itemsPerFrame, int frames, size_t ldsBytes) {
    hipLaunchKernelGGL((k_pipe<48, 45>), xcdGrid(itemsPerFrame, frames), dim3(256), ldsBytes, st, a);
}

}  // namespace orbx
