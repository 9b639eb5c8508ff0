// k_octree_common.hpp - helpers of the quad-tree kernel body (k_octree_body.inc): k_octree.hip compiles the body as kernels of three workgroup
// sizes.  See k_octree.hip for the algorithm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "orbx_device.hpp"

namespace orbx {

constexpr int kOctUnroll = 4;   // keys per thread per sweep iteration (memory-level parallelism)

// Dense phase.  DivideNode's boxes depend only on the root box, and the x and y split decisions are independent
// of each other, so a key's quadrant path of length kD0 below its root is two table look-ups: xcode[x] (root and
// 5 left/right decisions) and ycode[y] (5 up/down decisions), both built once per workgroup in LDS.  The first
// sweep stores the leaf cell (root, ypath, xpath) per key and histograms the 4^kD0 leaf cells of every root;
// summing 4:1 gives the key count of every possible node down to depth kD0.
// While every node that may split is shallower than kD0, a refinement pass needs NO sweep over the keys: its
// child counts are table look-ups.  The keys are visited again only to take the final arg-max (node found through
// a leaf-cell -> node table), or — when a node at depth kD0 must split — once to materialise their node ids,
// after which the passes continue with one rename+count sweep each.
constexpr int kD0 = kOctDepth;
constexpr int kLeaves = kOctLeaves;                           // 1024 leaf cells per root
constexpr int kHistPerRoot = (4 * kLeaves - 4) / 3;           // 4 + 16 + ... + 4^kD0 = 1364 counters per root
__device__ __forceinline__ int histOff(int depth) { return ((1 << (2 * depth)) - 4) / 3; }   // depth 1..kD0
// node descriptor in the dense phase: root << 24 | depth << 20 | ypath << 10 | xpath  (paths hold `depth` bits)
__device__ __forceinline__ unsigned nodeKey(int root, int depth, int yp, int xp) {
    return ((unsigned)root << 24) | ((unsigned)depth << 20) | ((unsigned)yp << 10) | (unsigned)xp;
}
// left/right (or up/down) decisions of DivideNode along one axis for coordinate v in the box [b0, b1): orbx_device.hpp
__device__ __forceinline__ int axisPath(int v, int b0, int b1) { return octAxisPath(v, b0, b1); }

__device__ __forceinline__ int quadrantOf(int x, int y, short4 b /* x0,x1,y0,y1 */) {
    const int cx = b.x + ((b.y - b.x + 1) >> 1);   // UL.x + ceil((UR.x-UL.x)/2)   (:488)
    const int cy = b.z + ((b.w - b.z + 1) >> 1);
    return (x < cx ? 0 : 1) + (y < cy ? 0 : 2);    // n1,n2,n3,n4 of DivideNode (:517-531)
}
__device__ __forceinline__ short4 childBox(short4 b, int q) {
    const short cx = (short)(b.x + ((b.y - b.x + 1) >> 1)), cy = (short)(b.z + ((b.w - b.z + 1) >> 1));
    short4 r;
    r.x = (q & 1) ? cx : b.x; r.y = (q & 1) ? b.y : cx;
    r.z = (q & 2) ? cy : b.z; r.w = (q & 2) ? b.w : cy;
    return r;
}

// Adds 1 to counter[4*node + q] for every active lane.  Candidates arrive cell by cell, so a wave's keys
// usually sit in one node: then four ballots replace up to 64 serialised same-address LDS atomics.
__device__ __forceinline__ void countQuadrant(int* childCnt, bool active, int node, int q) {
    const unsigned long long act = __ballot(active);
    if (act == 0) return;
    const int leader = __ffsll((long long)act) - 1;
    const int n0 = __shfl(node, leader);
    if (__all(!active || node == n0)) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const int c = __popcll(__ballot(active && q == qq));
            if (lane == leader && c) atomicAdd(&childCnt[4 * n0 + qq], c);
        }
    } else if (active) {
        atomicAdd(&childCnt[4 * node + q], 1);
    }
}

// Per-node arg-max: a 64-bit LDS atomic per key (lanes of a wave spread over only a few nodes, but a wave-level
// pre-reduction costs more instructions than the serialised same-address atomics it would save).
__device__ __forceinline__ void maxPerNode(unsigned long long* best, bool active, int node, unsigned long long v) {
    if (active) atomicMax(&best[node], v);
}


}  // namespace orbx
