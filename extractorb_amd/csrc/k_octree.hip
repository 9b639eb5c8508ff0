// k_octree.hip — DistributeOctTree quad-tree culling (reference ORBextractor.cc:544-771).
//
// One 1024-thread workgroup per (frame, level); the node list lives in LDS, the keys stay in HBM/L2.
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

#include "orbx_device.hpp"

namespace orbx {

#ifdef ORBX_OCT_STAMPS
__device__ unsigned long long g_octStamps[128];
#define STAMP(id) do { if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && nst < 126) { stampId[nst] = (id); stampT[nst++] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(id) do {} while (0)
#endif

constexpr int kOctThreads = 1024;
constexpr int kOctWaves = kOctThreads / 64;
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
constexpr int kD0 = 5;
constexpr int kLeaves = 1 << (2 * kD0);                       // 1024 leaf cells per root
constexpr int kHistPerRoot = (4 * kLeaves - 4) / 3;           // 4 + 16 + ... + 4^kD0 = 1364 counters per root
__device__ __forceinline__ int histOff(int depth) { return ((1 << (2 * depth)) - 4) / 3; }   // depth 1..kD0
// node descriptor in the dense phase: root << 24 | depth << 20 | ypath << 10 | xpath  (paths hold `depth` bits)
__device__ __forceinline__ unsigned nodeKey(int root, int depth, int yp, int xp) {
    return ((unsigned)root << 24) | ((unsigned)depth << 20) | ((unsigned)yp << 10) | (unsigned)xp;
}
// left/right (or up/down) decisions of DivideNode along one axis for coordinate v in the box [b0, b1)
__device__ __forceinline__ int axisPath(int v, int b0, int b1) {
    int path = 0;
#pragma unroll
    for (int d = 0; d < kD0; d++) {
        const int c = b0 + ((b1 - b0 + 1) >> 1);    // UL + ceil(extent/2)  (:488-489)
        const int bit = v < c ? 0 : 1;               // kp.pt.x < n1.UR.x  (:520)
        path = 2 * path + bit;
        if (bit) b0 = c; else b1 = c;
    }
    return path;
}

struct OctShared {
    unsigned scanTmp[kOctWaves];
    int size, prevSize, phase2, nToExpand, done, nChildren, breakRank, deep;
};

// exclusive prefix sum of data[0..n) in place, returns the total; every thread of the workgroup calls it
__device__ int blockExclusiveScan(int* data, int n, unsigned* tmp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + kOctThreads - 1) / kOctThreads;
    const int b = tid * per, e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += data[i];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    __syncthreads();   // tmp may still be read from a previous call
    if (lane == 63) tmp[wave] = (unsigned)incl;
    __syncthreads();
    int waveOff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kOctWaves; w++) {
        const int t = (int)tmp[w];
        if (w < wave) waveOff += t;
        total += t;
    }
    int run = waveOff + incl - sum;
    for (int i = b; i < e; i++) { const int v = data[i]; data[i] = run; run += v; }
    __syncthreads();
    return total;
}

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

__global__ __launch_bounds__(kOctThreads, 8) void k_octree(const LevelGeom* __restrict__ lv, int nlevels,
                                                         const CellDesc* __restrict__ cells, int nCellsTotal,
                                                         const unsigned* __restrict__ candSeg,
                                                         const unsigned* __restrict__ cellCount,
                                                         int* __restrict__ cellOff, unsigned* __restrict__ candPos,
                                                         unsigned* __restrict__ candCount,
                                                         unsigned short* __restrict__ nodeOf,
                                                         uint2* __restrict__ sel, int selPerFrame,
                                                         int* __restrict__ levelCount, int* __restrict__ levelLap,
                                                         const int* __restrict__ lapArea, int M, int P, int R, int XT) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ OctShared sh;
    // blockIdx.x = frame: consecutive workgroups are dealt round-robin to the 8 XCDs, so the heavy level-0
    // problems (dispatched first) spread over the whole chip instead of piling onto one XCD
    const int level = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
#ifdef ORBX_OCT_STAMPS
    unsigned long long stampT[126]; int stampId[126]; int nst = 0;
#endif
    STAMP(0);
    const LevelGeom g = lv[level];
    // carve the dynamic LDS (M is a multiple of 8)
    short4* box[2];
    int* cnt[2];
    box[0] = (short4*)smem;              box[1] = box[0] + M;
    cnt[0] = (int*)(box[1] + M);         cnt[1] = cnt[0] + M;
    int* childCnt = cnt[1] + M;                       // [M][4]; reused as u64 best[M] by the last sweep
    unsigned short* mapChild = (unsigned short*)(childCnt + 4 * M);   // [M][4] new position of child q
    unsigned short* mapKeep = mapChild + 4 * M;       // [M]    new position of an unsplit node
    int* fwd = (int*)(mapKeep + M);                   // [M]    creation offset of a split node's first child
    int* keepIdx = fwd + M;                           // [M]
    unsigned long long* sortKey = (unsigned long long*)(keepIdx + M);   // [P]
    unsigned* ncode[2];                                // [M] x2 dense-phase node descriptors
    ncode[0] = (unsigned*)(sortKey + P);  ncode[1] = ncode[0] + M;
    int* hist = (int*)(ncode[1] + M);                 // [R][kHistPerRoot] key counts of every node down to depth kD0
    unsigned short* cell = (unsigned short*)(hist + R * kHistPerRoot);   // [R][kLeaves] leaf cell -> node position
    uint8_t* xcode = (uint8_t*)(cell + R * kLeaves);  // [XT] root << kD0 | x path of every x of the rectangle
    uint8_t* ycode = xcode + XT;                      // [XT] y path of every y

    // ---- compaction plan: exclusive scan of this level's per-cell candidate counts (cells are in the reference's
    //      loop order, so the compacted array IS vToDistributeKeys, :786-864) ----
    const long long base = g.candOff + (long long)f * g.candCap;
    const unsigned* seg = candSeg + base;             // per-cell segments written by k_fast
    unsigned* pos = candPos + base;                   // compacted (x | y << 12 | response << 24), written below
    unsigned short* nof = nodeOf + base;
    uint2* selOut = sel + (long long)f * selPerFrame + g.selOff;
    const int N = g.quota;
    const int cN = g.cellCount;
    const CellDesc* cd0 = cells + g.cellFirst;
    const unsigned* cc = cellCount + (long long)f * nCellsTotal + g.cellFirst;
    int* co = cellOff + (long long)f * nCellsTotal + g.cellFirst;
    int nC;
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int per = (cN + kOctThreads - 1) / kOctThreads, b = tid * per, e = min(b + per, cN);
        int sum = 0;
        for (int i = b; i < e; i++) sum += (int)cc[i];
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) sh.scanTmp[wave] = (unsigned)incl;
        __syncthreads();
        int waveOff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kOctWaves; w++) { const int t = (int)sh.scanTmp[w]; if (w < wave) waveOff += t; total += t; }
        int run = waveOff + incl - sum;
        for (int i = b; i < e; i++) { co[i] = run; run += (int)cc[i]; }
        nC = total > g.candCap ? 0 : total;           // the segments are exactly sized: total <= candCap always
        if (tid == 0) candCount[f * nlevels + level] = (unsigned)nC;
    }

    // ---- roots (:548-575); empty roots are dropped by the first pass (:577-590) ----
    if (tid < g.nIni) {
        short4 b;
        b.x = (short)(int)(g.hX * (float)tid);
        b.y = (short)(int)(g.hX * (float)(tid + 1));
        b.z = 0; b.w = (short)g.rectH;
        box[0][tid] = b;
        cnt[0][tid] = 0;
    }
    bool dense = g.nIni <= R && g.rectW <= XT && g.rectH <= XT;    // workgroup-uniform
    if (tid < g.nIni) ncode[0][tid] = nodeKey(tid, 0, 0, 0);
    if (dense) {
        for (int x = tid; x < g.rectW; x += kOctThreads) {
            int r = (int)__fdiv_rn((float)x, g.hX);                // vpIniNodes[kp.pt.x/hX] (:574)
            r = r > g.nIni - 1 ? g.nIni - 1 : r;
            xcode[x] = (uint8_t)((r << kD0) | axisPath(x, (int)(g.hX * (float)r), (int)(g.hX * (float)(r + 1))));
        }
        for (int y = tid; y < g.rectH; y += kOctThreads) ycode[y] = (uint8_t)axisPath(y, 0, g.rectH);
    }
    for (int i = tid; i < 4 * g.nIni; i += kOctThreads) childCnt[i] = 0;
    if (dense)
        for (int i = tid; i < g.nIni * kHistPerRoot; i += kOctThreads) hist[i] = 0;
    __syncthreads();
    // ---- first sweep: a wave per cell gathers the cell's segment into the compacted array and, in the same visit,
    //      assigns the key to its root / dense-phase leaf cell ----
    constexpr int kCellsPerTrip = 8;   // small enough that 16 waves share even a 100-cell level evenly
    for (int c0 = (tid >> 6) * kCellsPerTrip; c0 < cN && nC > 0; c0 += kOctWaves * kCellsPerTrip) {
        // The wave takes 8 consecutive cells.  Their keys are consecutive in the compacted array, so lanes walk the
        // concatenation of the 8 segments at full width; which segment a key sits in is 7 compares against
        // wave-uniform prefix counts held in scalar registers.
        const int lane = tid & 63, cj = c0 + (lane & (kCellsPerTrip - 1));
        const int nj = cj < cN ? (int)cc[cj] : 0, sj = cj < cN ? cd0[cj].segOff : 0, oj = cj < cN ? co[cj] : 0;
        int pre[kCellsPerTrip + 1], D[kCellsPerTrip];
        pre[0] = 0;
#pragma unroll
        for (int j = 0; j < kCellsPerTrip; j++) {
            pre[j + 1] = pre[j] + __builtin_amdgcn_readlane(nj, j);
            D[j] = __builtin_amdgcn_readlane(sj, j) - pre[j];       // source slot = D[j] + (index inside the 8 cells)
        }
        const int k0 = __builtin_amdgcn_readlane(oj, 0), T8 = pre[kCellsPerTrip];
        for (int i0 = 0; i0 < T8; i0 += 64) {
            const int i = i0 + lane;
            const bool active = i < T8;
            int r = 0, q = 0;
            if (active) {
                int src = i + D[0];
#pragma unroll
                for (int j = 1; j < kCellsPerTrip; j++) src = i >= pre[j] ? i + D[j] : src;
                const unsigned w = seg[src];
                const int k = k0 + i;
                pos[k] = w;
                const int x = w & 0xfff, y = (w >> 12) & 0xfff;
                if (dense) {
                    const int xc = xcode[min(x, g.rectW - 1)], yc = ycode[min(y, g.rectH - 1)];
                    const int leaf = (yc << kD0) | (xc & ((1 << kD0) - 1));      // row-major leaf cell of the root
                    r = xc >> kD0;
                    nof[k] = (unsigned short)(r * kLeaves + leaf);
                    atomicAdd(&hist[r * kHistPerRoot + histOff(kD0) + leaf], 1);
                } else {
                    r = (int)__fdiv_rn((float)x, g.hX);            // vpIniNodes[kp.pt.x/hX] (:574)
                    r = r > g.nIni - 1 ? g.nIni - 1 : r;
                    nof[k] = (unsigned short)r;
                    q = quadrantOf(x, y, box[0][r]);
                }
            }
            if (!dense) countQuadrant(childCnt, active, r, q);
        }
    }
    __syncthreads();
    if (dense) {
        for (int d = kD0 - 1; d >= 1; d--) {      // 4:1 sums: counts of every node at depth d
            const int nd = 1 << (2 * d);
            for (int i = tid; i < g.nIni * nd; i += kOctThreads) {
                const int r = i >> (2 * d), c = i & (nd - 1), yp = c >> d, xp = c & ((1 << d) - 1);
                const int* ch = hist + r * kHistPerRoot + histOff(d + 1) + ((2 * yp) << (d + 1)) + 2 * xp;   // 2x2 block below
                hist[r * kHistPerRoot + histOff(d) + c] = ch[0] + ch[1] + ch[1 << (d + 1)] + ch[(1 << (d + 1)) + 1];
            }
            __syncthreads();
        }
        if (tid < g.nIni) {
            const int* h1 = hist + tid * kHistPerRoot;
            cnt[0][tid] = h1[0] + h1[1] + h1[2] + h1[3];
        }
    } else if (tid < g.nIni) {
        cnt[0][tid] = childCnt[4 * tid] + childCnt[4 * tid + 1] + childCnt[4 * tid + 2] + childCnt[4 * tid + 3];
    }
    if (tid == 0) { sh.size = nC > 0 ? g.nIni : 0; sh.phase2 = 0; sh.done = 0; }
    __syncthreads();
    if (tid == 0) {
        int ne = 0;
        for (int r = 0; r < g.nIni; r++) ne += cnt[0][r] > 0;
        sh.prevSize = ne;
    }
    __syncthreads();

    int cur = 0, passes = 0;
    unsigned long long* best = (unsigned long long*)childCnt;
    STAMP(1);
    // ---- refinement passes (:599-744) ----
    while (sh.size > 0) {
        const int size = sh.size, phase2 = sh.phase2, prevSize = sh.prevSize;
        short4* bx = box[cur];
        int* cn = cnt[cur];
        if (tid == 0) { sh.nToExpand = 0; sh.breakRank = 0x7fffffff; sh.deep = 0; }
        if (dense) {
            const unsigned* cd = ncode[cur];
            __syncthreads();
            for (int n = tid; n < size; n += kOctThreads)
                if (cn[n] > 1 && ((cd[n] >> 20) & 15) >= (unsigned)kD0) sh.deep = 1;
            __syncthreads();
            if (!sh.deep) {
                // child counts of every node that may split: look-ups in the count pyramid
                for (int i = tid; i < 4 * size; i += kOctThreads) {
                    const int n = i >> 2, q = i & 3;
                    const unsigned c = cd[n];
                    const int root = c >> 24, depth = (c >> 20) & 15, yp = (c >> 10) & 1023, xp = c & 1023;
                    const int cy = 2 * yp + (q >> 1), cx = 2 * xp + (q & 1);      // child q of DivideNode
                    childCnt[i] = cn[n] > 1 ? hist[root * kHistPerRoot + histOff(depth + 1) + (cy << (depth + 1)) + cx] : 0;
                }
                __syncthreads();
            } else {
                // a node at depth kD0 has to split: give every key its node id once, counting the next split;
                // from here on the passes sweep the keys
                for (int n = tid; n < size; n += kOctThreads) {
                    const unsigned c = cd[n];
                    const int root = c >> 24, depth = (c >> 20) & 15, yp = (c >> 10) & 1023, xp = c & 1023;
                    if (cn[n] > 0 && depth <= kD0) {
                        const int span = 1 << (kD0 - depth);      // leaf cells per side under this node
                        unsigned short* dst = cell + root * kLeaves + ((yp * span) << kD0) + xp * span;
                        for (int jy = 0; jy < span; jy++)
                            for (int jx = 0; jx < span; jx++) dst[(jy << kD0) + jx] = (unsigned short)n;
                    }
                }
                for (int i = tid; i < 4 * size; i += kOctThreads) childCnt[i] = 0;
                __syncthreads();
                for (int k0 = 0; k0 < nC; k0 += kOctUnroll * kOctThreads) {
                    unsigned w[kOctUnroll];
                    int nd[kOctUnroll];
#pragma unroll
                    for (int u = 0; u < kOctUnroll; u++) {
                        const int k = k0 + u * kOctThreads + tid;
                        w[u] = k < nC ? pos[k] : 0u;
                        nd[u] = k < nC ? (int)nof[k] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < kOctUnroll; u++) {
                        const int k = k0 + u * kOctThreads + tid;
                        const bool active = k < nC;
                        int nn = 0, q = 0;
                        bool counts = false;
                        if (active) {
                            nn = cell[nd[u]];
                            nof[k] = (unsigned short)nn;
                            counts = cn[nn] > 1;
                            if (counts) q = quadrantOf(w[u] & 0xfff, (w[u] >> 12) & 0xfff, bx[nn]);
                        }
                        countQuadrant(childCnt, counts, nn, q);
                    }
                }
                dense = false;
                __syncthreads();
            }
        }
        auto nch = [&](int n) {   // non-empty children of node n
            return (childCnt[4 * n] > 0) + (childCnt[4 * n + 1] > 0) + (childCnt[4 * n + 2] > 0) + (childCnt[4 * n + 3] > 0);
        };
        int C;   // children created in this pass
        if (!phase2) {
            for (int n = tid; n < size; n += kOctThreads) { fwd[n] = cn[n] > 1 ? nch(n) : 0; keepIdx[n] = cn[n] == 1 ? 1 : 0; }
            __syncthreads();
            C = blockExclusiveScan(fwd, size, sh.scanTmp);
        } else {
            // sort multi-key nodes by (size desc, list position asc)
            for (int i = tid; i < P; i += kOctThreads) {
                unsigned long long key = ~0ull;
                if (i < size && cn[i] > 1) key = ((unsigned long long)(0xffffffffu - (unsigned)cn[i]) << 32) | (unsigned)i;
                sortKey[i] = key;
            }
            __syncthreads();
            for (int k2 = 2; k2 <= P; k2 <<= 1) {
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < P; i += kOctThreads) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = sortKey[i], b = sortKey[ixj];
                            const bool up = (i & k2) == 0;
                            if ((a > b) == up) { sortKey[i] = b; sortKey[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            }
            // rank r -> node; inc[r] = children - 1; the pass breaks after the first r with
            // size + sum_{i<=r} inc[i] >= N  (:735-736)
            int* inc = keepIdx;   // scratch, rank-indexed
            for (int r = tid; r < size; r += kOctThreads) {
                const unsigned long long key = sortKey[r];
                inc[r] = key == ~0ull ? 0 : nch((int)(unsigned)key) - 1;
            }
            __syncthreads();
            blockExclusiveScan(inc, size, sh.scanTmp);   // inc[r] = sum_{i<r}
            for (int r = tid; r < size; r += kOctThreads) {
                const unsigned long long key = sortKey[r];
                if (key != ~0ull && size + inc[r] + nch((int)(unsigned)key) - 1 >= N) atomicMin(&sh.breakRank, r);
            }
            for (int n = tid; n < size; n += kOctThreads) fwd[n] = -1;
            __syncthreads();
            const int br = sh.breakRank;
            // creation offset of rank r's first child = sum_{i<r} (inc_i + 1) = inc[r] + r
            for (int r = tid; r < size; r += kOctThreads) {
                const unsigned long long key = sortKey[r];
                if (key != ~0ull && r <= br) fwd[(int)(unsigned)key] = inc[r] + r;
            }
            if (tid == 0) {
                int lo = 0, hi = size;   // first rank whose key is ~0 == number of multi-key nodes
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (sortKey[mid] == ~0ull) hi = mid; else lo = mid + 1; }
                const int last = lo - 1 < br ? lo - 1 : br;
                sh.nChildren = last < 0 ? 0 : inc[last] + last + nch((int)(unsigned)sortKey[last]);
            }
            __syncthreads();
            C = sh.nChildren;
            for (int n = tid; n < size; n += kOctThreads) keepIdx[n] = fwd[n] < 0 ? 1 : 0;
            __syncthreads();
        }
        STAMP(2 + phase2);
        const int nKept = blockExclusiveScan(keepIdx, size, sh.scanTmp);
        const int newSize = C + nKept;
        if (newSize > M || ++passes > 64) {   // cannot happen for a valid geometry (the host sizes M); never write out of bounds
            if (tid == 0) sh.size = 0;
            __syncthreads();
            break;
        }
        // a node splits iff it has several keys (phase 1) / iff it was reached before the break (phase 2)
        short4* nbx = box[cur ^ 1];
        int* ncn = cnt[cur ^ 1];
        const unsigned* ocd = ncode[cur];
        unsigned* ncd = ncode[cur ^ 1];
        for (int n = tid; n < size; n += kOctThreads) {
            const int c0 = cn[n];
            const bool split = phase2 ? fwd[n] >= 0 : c0 > 1;
            if (split) {
                int j = fwd[n], expand = 0;
                const unsigned cdn = ocd[n];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = childCnt[4 * n + q];
                    if (c > 0) {
                        const int p = C - 1 - j;   // pushed to the front in creation order
                        nbx[p] = childBox(bx[n], q);
                        ncn[p] = c;
                        if (dense) ncd[p] = nodeKey(cdn >> 24, ((cdn >> 20) & 15) + 1, 2 * ((cdn >> 10) & 1023) + (q >> 1), 2 * (cdn & 1023) + (q & 1));
                        mapChild[4 * n + q] = (unsigned short)p;
                        expand += c > 1;
                        j++;
                    }
                }
                if (expand) atomicAdd(&sh.nToExpand, expand);
            } else if (c0 > 0) {
                const int p = C + keepIdx[n];
                nbx[p] = bx[n];
                ncn[p] = c0;
                if (dense) ncd[p] = ocd[n];
                mapKeep[n] = (unsigned short)p;
            }
        }
        __syncthreads();
        if (tid == 0) {
            sh.size = newSize;
            if (newSize >= N || newSize == prevSize) sh.done = 1;              // :674, :739
            else if (!phase2 && newSize + 3 * sh.nToExpand > N) sh.phase2 = 1;  // :678
            sh.prevSize = newSize;
        }
        __syncthreads();
        const bool last = sh.done != 0;
        STAMP(4);
        if (dense) {
            // no sweep between dense passes; after the last one the keys find their node through the leaf-cell table
            if (last) {
                for (int n = tid; n < newSize; n += kOctThreads) {
                    const unsigned c = ncd[n];
                    const int root = c >> 24, depth = (c >> 20) & 15, yp = (c >> 10) & 1023, xp = c & 1023;
                    const int span = 1 << (kD0 - depth);
                    unsigned short* dst = cell + root * kLeaves + ((yp * span) << kD0) + xp * span;
                    for (int jy = 0; jy < span; jy++)
                        for (int jx = 0; jx < span; jx++) dst[(jy << kD0) + jx] = (unsigned short)n;
                    best[n] = 0;
                }
                __syncthreads();
                for (int k0 = 0; k0 < nC; k0 += kOctUnroll * kOctThreads) {
                    unsigned w[kOctUnroll];
                    int nd[kOctUnroll];
#pragma unroll
                    for (int u = 0; u < kOctUnroll; u++) {
                        const int k = k0 + u * kOctThreads + tid;
                        const bool active = k < nC;
                        w[u] = active ? pos[k] : 0u;
                        nd[u] = active ? (int)nof[k] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < kOctUnroll; u++) {
                        const int k = k0 + u * kOctThreads + tid;
                        const bool active = k < nC;
                        const int nn = active ? (int)cell[nd[u]] : 0;
                        // max response, then smallest candidate index (= first in the reference's vector)
                        const unsigned long long v = ((unsigned long long)(w[u] >> 24) << 56) | ((unsigned long long)(~(unsigned)k) << 24) |
                                                     (unsigned long long)(w[u] & 0xffffff);
                        maxPerNode(best, active, nn, v);
                    }
                }
            }
        } else {
        // the child counters are consumed; clear them for the new list (or the arg-max slots if this was the last pass)
        for (int i = tid; i < 4 * newSize; i += kOctThreads) childCnt[i] = 0;
        __syncthreads();
        // ---- the sweep: rename every key's node; count it for the next pass, or take the arg-max ----
        for (int k0 = 0; k0 < nC; k0 += kOctUnroll * kOctThreads) {
            unsigned w[kOctUnroll];
            int nd[kOctUnroll];
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {   // all loads of the iteration in flight together
                const int k = k0 + u * kOctThreads + tid;
                const bool active = k < nC;
                w[u] = active ? pos[k] : 0u;
                nd[u] = active ? (int)nof[k] : 0;
            }
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                const int k = k0 + u * kOctThreads + tid;
                const bool active = k < nC;
                int nn = 0, q = 0;
                bool counts = false;
                unsigned long long v = 0;
                if (active) {
                    const int n = nd[u];
                    const int x = w[u] & 0xfff, y = (w[u] >> 12) & 0xfff;
                    const bool split = phase2 ? fwd[n] >= 0 : cn[n] > 1;
                    nn = split ? mapChild[4 * n + quadrantOf(x, y, bx[n])] : mapKeep[n];
                    if (last) {
                        // max response, then smallest candidate index (= first in the reference's vector)
                        v = ((unsigned long long)(w[u] >> 24) << 56) | ((unsigned long long)(~(unsigned)k) << 24) |
                            (unsigned long long)(w[u] & 0xffffff);
                    } else {
                        nof[k] = (unsigned short)nn;
                        counts = ncn[nn] > 1;
                        if (counts) q = quadrantOf(x, y, nbx[nn]);
                    }
                }
                if (last) maxPerNode(best, active, nn, v);
                else countQuadrant(childCnt, counts, nn, q);
            }
        }
        }
        cur ^= 1;
        __syncthreads();
        STAMP(5);
        if (last) break;
    }

    // ---- one keypoint per node, in list order (:751-767), plus the lapping rank used for placement ----
    const int size = sh.size;
    const int lap0 = lapArea[2 * f], lap1 = lapArea[2 * f + 1];
    int* lapFlag = fwd;
    for (int i = tid; i < size; i += kOctThreads) {
        const unsigned long long v = best[i];
        const int x = (int)(v & 0xfff) + kMinBorder;
        float xs = (float)x;
        if (level != 0) xs = __fmul_rn(xs, g.scale);      // keypoint->pt *= scale (:1143-1145)
        lapFlag[i] = (xs >= (float)lap0 && xs <= (float)lap1) ? 1 : 0;   // :1147
        keepIdx[i] = lapFlag[i];
    }
    __syncthreads();
    const int nLap = blockExclusiveScan(keepIdx, size, sh.scanTmp);
    for (int i = tid; i < size; i += kOctThreads) {
        const unsigned long long v = best[i];
        uint2 o;
        const unsigned x = (unsigned)(v & 0xfff) + kMinBorder, y = (unsigned)((v >> 12) & 0xfff) + kMinBorder;
        o.x = x | (y << 12) | ((unsigned)(v >> 56) << 24);
        o.y = (unsigned)keepIdx[i] | ((unsigned)lapFlag[i] << 31);   // rank among this level's lapping keys
        selOut[i] = o;
    }
    if (tid == 0) {
        levelCount[f * nlevels + level] = size;
        levelLap[f * nlevels + level] = nLap;
    }
    STAMP(6);
#ifdef ORBX_OCT_STAMPS
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        g_octStamps[0] = nst;
        for (int i = 0; i < nst; i++) g_octStamps[1 + i] = (stampT[i] << 8) | (unsigned)stampId[i];
    }
#endif
}

#ifdef ORBX_OCT_STAMPS
extern "C" int orbx_debug_oct_stamps(unsigned long long* out128) {
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_octStamps), sizeof(unsigned long long) * 128);
}
#endif

size_t octreeLdsBytes(int M, int P, int R, int XT) {
    size_t b = 0;
    b += 2 * (size_t)M * sizeof(short4);            // box
    b += 2 * (size_t)M * sizeof(int);               // cnt
    b += 4 * (size_t)M * sizeof(int);               // childCnt / best
    b += 4 * (size_t)M * sizeof(unsigned short);    // mapChild
    b += (size_t)M * sizeof(unsigned short);        // mapKeep
    b += (size_t)M * sizeof(int);                   // fwd
    b += (size_t)M * sizeof(int);                   // keepIdx
    b += (size_t)P * sizeof(unsigned long long);    // sortKey
    b += 2 * (size_t)M * sizeof(unsigned);          // ncode
    b += (size_t)R * kHistPerRoot * sizeof(int);    // hist
    b += (size_t)R * kLeaves * sizeof(unsigned short);   // cell
    b += 2 * (size_t)XT;                            // xcode, ycode
    return b + 64;
}
void launchOctree(hipStream_t st, const LevelGeom* lv, int nlevels, const CellDesc* cells, int nCellsTotal,
                  const unsigned* candSeg, const unsigned* cellCount, int* cellOff, unsigned* candPos, unsigned* candCount,
                  unsigned short* nodeOf, uint2* sel, int selPerFrame, int* levelCount, int* levelLap, const int* lapArea,
                  int M, int P, int R, int XT, int B) {
    hipLaunchKernelGGL(k_octree, dim3(B, nlevels), dim3(kOctThreads), octreeLdsBytes(M, P, R, XT), st, lv, nlevels, cells,
                       nCellsTotal, candSeg, cellCount, cellOff, candPos, candCount, nodeOf, sel, selPerFrame, levelCount, levelLap, lapArea, M, P, R, XT);
}

}  // namespace orbx
