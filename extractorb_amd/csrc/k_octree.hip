// k_octree.hip — DistributeOctTree quad-tree culling (reference ORBextractor.cc:544-771).
//
// One workgroup per (frame, level) — 256, 512 or 1024 threads, chosen by the host (k_octree_body.inc is compiled per
// size); the node list lives in LDS.  While no node deeper than a 32x32 leaf grid per root has to split ("dense
// phase", the normal case) the keys are read once from k_fast's per-cell segments and only per-leaf counts and
// per-leaf best keys are kept; otherwise they are compacted to HBM and swept once per pass as described next.
//
// The reference's std::list is modelled as an array in list order.  A key never needs its position
// inside a node: the final pick per node is max response, first in candidate order on ties
// (:751-767), and DivideNode's partition is stable, so "first" == smallest candidate order word.
// A refinement pass is therefore (a) node-level work on <= N+3 nodes: decide which nodes split and
// where their children land in the new list; (b) ONE sweep over the keys that renames every key's
// node and, in the same visit, counts it into the child quadrant of its new node for the next pass.
// The last sweep takes the per-node arg-max instead.  passes+1 sweeps in total.
//   phase 1 (:611-670): every multi-key node splits, in list order; children are pushed to the
//     front one by one, so the new list is reverse(creation order) followed by the untouched nodes.
//   phase 2 (:681-742): multi-key nodes split in order (size desc, newest first) and the pass stops
//     right after the split that reaches N nodes.  All multi-key nodes were created in the previous
//     pass, where creation order is the reverse of list order, so "newest first" == smallest list
//     position: the sort key is (size desc, position asc)  [the oracle's declared tie rule; the
//     reference compares heap addresses there, SURVEY.md §8c].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "orbx_device.hpp"
#include "k_octree_common.hpp"

namespace orbx {

#ifdef ORBX_OCT_STAMPS
__device__ unsigned long long g_octStamps[128];
__device__ unsigned long long g_octSpans[2 * 16];      // (start, end) of frame 0's workgroup of every level
#ifndef ORBX_OCT_STAMP_LEVEL
#define ORBX_OCT_STAMP_LEVEL 0
#endif
#define STAMP(id) do { if (tid == 0 && blockIdx.x == 0 && blockIdx.y == ORBX_OCT_STAMP_LEVEL && nst < 126) { stampId[nst] = (id); stampT[nst++] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(id) do {} while (0)
#endif

// The kernel proper is compiled for three workgroup sizes.  A workgroup is bound by its chain of barriers and memory
// round trips, not by arithmetic, so a level whose sweeps are short (a 640x480 image: a few thousand keys) runs as
// fast with 256 threads as with 1024 but then holds a quarter of a CU's thread slots: four times as many levels run
// side by side.  Levels with tens of thousands of keys (1080p) still want 512 or 1024 threads.
// OCT_W = waves per SIMD the variant is compiled for, i.e. its register budget: 8 -> 64 VGPRs (about 64 values live in
// scratch, but 6 workgroups of 256 threads fit a CU: best when a large batch queues many more workgroups than fit the chip),
// 4 -> 128 VGPRs, no scratch (best while every workgroup of the launch is resident anyway: small batches, one frame).
// Queued variants (large batches): 1024 and 512 threads are compiled for 8 waves per SIMD (64 VGPRs: two 1024-thread or four
// 512-thread workgroups per CU), 256 threads for 6 (80 VGPRs, fewer values in scratch; LDS allows six such workgroups per CU
// anyway): 179 -> 169 us per 512 frames at 640x480; the same budget on the 1024-thread variant halves its residency (1080p: 282 -> 310 us).
#ifndef OCT_WAVE_PHASE2
#define OCT_WAVE_PHASE2 1      // a sorted pass over at most 64 nodes runs on one wave (k_octree_body.inc)
#endif
#ifndef OCT_SHORT_PHASE2
#define OCT_SHORT_PHASE2 (OCT_W <= 4)
#endif
// the candidate format of a build (orbx_device.hpp: CandFmt): one dword; the three ..b builds at the end read the two-dword format of frames beyond 4096 px
#define OCT_FMT CandFmt<false>
#define OCT_CAND CandFmt<false>::T
#define OCT_W 8
#define OCT_T 1024
#define OCT_NAME(x) x##_1024
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#define OCT_T 512
#define OCT_NAME(x) x##_512
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#undef OCT_W
#define OCT_W 6
#define OCT_T 256
#define OCT_NAME(x) x##_256
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#undef OCT_W
#define OCT_W 4
#define OCT_T 1024
#define OCT_NAME(x) x##_1024r
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
// (256 and 512 resident threads: three waves per SIMD = 168 VGPRs, where the body needs no scratch at all - 144 - against 9 / 7 spilled
// registers at 128; the host counts residency for these two at 768 threads per CU, launchOctree's caller.  1024 threads are four waves per SIMD
// by themselves: 128 VGPRs is their ceiling.)
#undef OCT_W
#define OCT_W 3
#define OCT_T 256
#define OCT_NAME(x) x##_256r
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#define OCT_T 512
#define OCT_NAME(x) x##_512r
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#undef OCT_W
#define OCT_W 4
#define OCT_GLOBAL 1
#define OCT_T 1024
#define OCT_NAME(x) x##_1024g
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#undef OCT_GLOBAL
#undef OCT_W
// ---- frames beyond 4096 px (round 6): the same body over two-dword candidates, in the three 1024-thread forms such frames take (queued,
//      resident, HBM node arena); the host keeps the leaf tables off for them ----
#undef OCT_FMT
#undef OCT_CAND
#define OCT_FMT CandFmt<true>
#define OCT_CAND CandFmt<true>::T
#define OCT_W 8
#define OCT_T 1024
#define OCT_NAME(x) x##_1024b
#include "k_octree_body.inc"
#undef OCT_NAME
#undef OCT_W
#define OCT_W 4
#define OCT_NAME(x) x##_1024rb
#include "k_octree_body.inc"
#undef OCT_NAME
#define OCT_GLOBAL 1
#define OCT_NAME(x) x##_1024gb
#include "k_octree_body.inc"
#undef OCT_T
#undef OCT_NAME
#undef OCT_GLOBAL
#undef OCT_W
#undef OCT_FMT
#undef OCT_CAND

#ifdef ORBX_OCT_STAMPS
extern "C" int orbx_debug_oct_stamps(unsigned long long* out128) {
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_octStamps), sizeof(unsigned long long) * 128);
}
extern "C" int orbx_debug_oct_spans(unsigned long long* out32) { return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_octSpans), sizeof(unsigned long long) * 32); }
#endif

// bytes of the node arrays of one workgroup (LDS, or a slice of the HBM arena of the _1024g variant); layout: k_octree_body.inc
size_t octreeLdsBytes(int M, int P, int R, int XT) {
    (void)P;                                        // the sort keys live in the idle node group (8 P <= 16 M)
    const size_t tables = ((size_t)2 * XT + 7) & ~(size_t)7;
    size_t b = 2 * std::max((size_t)16 * M, tables);    // two groups {box (8), cnt (4), ncode (4)} x M | sort keys | xcode, ycode
    b += 4 * (size_t)M * sizeof(int);               // childCnt / best
    b += 4 * (size_t)M * sizeof(unsigned short);    // mapChild
    b += (size_t)M * sizeof(unsigned short);        // mapKeep
    b += (size_t)M * sizeof(int);                   // fwd
    b += (size_t)M * sizeof(int);                   // keepIdx
    b += (size_t)R * kHistPerRoot * sizeof(int);    // hist
    b += (size_t)R * kLeaves * sizeof(unsigned);    // leafBest | cell
    b += 9 * kOctPad;                               // (host emulation only: red zones behind the sub-arrays)
    return b + 64;
}
void launchOctree(hipStream_t st, const LevelGeom* lv, int nlevels, const CellDesc* cells, int nCellsTotal,
                  const unsigned* candSeg, const unsigned* cellCount, int* cellOff, unsigned* candPos, unsigned* candCount,
                  unsigned short* nodeOf, uint2* sel, int selPerFrame, int* levelCount, int* levelLap, const int* lapArea,
                  int M, int P, int R, int XT, const int* threadsOfLevel, bool roomy, int f0, int B, uint8_t* nodeArena, LeafTables lt, bool big) {
    const size_t bytes = octreeLdsBytes(M, P, R, XT);
    if (big) {      // two-dword candidates: one launch, 1024 threads per (frame, level)
        typedef CandFmt<true>::T Big;
        if (nodeArena)
            hipLaunchKernelGGL(k_octree_1024gb, dim3(B, nlevels), dim3(1024), 0, st, lv, nlevels, cells, nCellsTotal, (const Big*)candSeg, cellCount,
                               cellOff, (Big*)candPos, candCount, nodeOf, sel, selPerFrame, levelCount, levelLap, lapArea, M, P, R, XT, 0, f0,
                               nodeArena, (unsigned long long)((bytes + 255) & ~(size_t)255), lt);
        else
            hipLaunchKernelGGL(roomy ? k_octree_1024rb : k_octree_1024b, dim3(B, nlevels), dim3(1024), bytes, st, lv, nlevels, cells, nCellsTotal,
                               (const Big*)candSeg, cellCount, cellOff, (Big*)candPos, candCount, nodeOf, sel, selPerFrame, levelCount, levelLap,
                               lapArea, M, P, R, XT, 0, f0, (uint8_t*)nullptr, 0ull, lt);
        return;
    }
    if (nodeArena) {      // node arrays in HBM: one launch, 1024 threads per (frame, level)
        hipLaunchKernelGGL(k_octree_1024g, dim3(B, nlevels), dim3(1024), 0, st, lv, nlevels, cells, nCellsTotal, candSeg, cellCount,
                           cellOff, candPos, candCount, nodeOf, sel, selPerFrame, levelCount, levelLap, lapArea, M, P, R, XT, 0, f0,
                           nodeArena, (unsigned long long)((bytes + 255) & ~(size_t)255), lt);
        return;
    }
    // consecutive levels with the same workgroup size share a launch; the smallest levels (many short workgroups) go
    // first so that the long workgroups of the large levels form the tail
    int hi = nlevels;
    while (hi > 0) {
        int lo = hi - 1;
        const int T = threadsOfLevel[lo];
        while (lo > 0 && threadsOfLevel[lo - 1] == T) lo--;
        auto kern = roomy ? (T == 256 ? k_octree_256r : (T == 512 ? k_octree_512r : k_octree_1024r))
                          : (T == 256 ? k_octree_256 : (T == 512 ? k_octree_512 : k_octree_1024));
        hipLaunchKernelGGL(kern, dim3(B, hi - lo), dim3(T), bytes, st, lv, nlevels, cells, nCellsTotal,
                           candSeg, cellCount, cellOff, candPos, candCount, nodeOf, sel, selPerFrame, levelCount, levelLap,
                           lapArea, M, P, R, XT, lo, f0, (uint8_t*)nullptr, 0ull, lt);
        hi = lo;
    }
}

}  // namespace orbx
