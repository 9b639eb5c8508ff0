// HOST EMULATION SHIM for k_octree.hip — TEST INFRASTRUCTURE (tools/octree_emu), never part of the product build.
//
// Lets g++ compile the quad-tree kernel source UNCHANGED and run one workgroup as T real host threads:
//   * __syncthreads()                     -> a pthread barrier over the workgroup (a happens-before edge ThreadSanitizer sees)
//   * __ballot/__any/__all/__shfl/readlane/update_dpp -> wave collectives: every lane of the 64-lane wave deposits its value, one
//     barrier over the wave, every lane reads what it needs (a lane that never arrives = divergent collective = deadlock, reported)
//   * atomicAdd/Max/Min on LDS or global  -> relaxed __atomic builtins
//   * __shared__                          -> static storage (one workgroup runs at a time); dynamic LDS -> an exactly sized heap block
// Built three ways by tools/octree_emu/Makefile: plain, -fsanitize=address,undefined (out-of-range LDS / global indices: every array
// is an exactly sized heap block, and ORBX_OCT_EMU_PAD puts poisoned red zones between the LDS sub-arrays), -fsanitize=thread (two
// accesses to one location, one of them a write, not separated by a barrier = a data race, whatever order the threads happened to run in).
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __align__(n) alignas(n)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct short4 { short x, y, z, w; };
struct uint2 { unsigned x, y; };
typedef void* hipStream_t;
using std::min;
using std::max;
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

namespace emu {
constexpr int kWave = 64;
struct Wave {
    pthread_barrier_t bar;
    unsigned long long slot[2][kWave];
};
struct Block {
    int nThreads = 0;
    pthread_barrier_t bar;
    std::vector<Wave> waves;
    uint8_t* dynShared = nullptr;
    size_t dynBytes = 0;
};
extern Block* g_block;
extern thread_local dim3 t_threadIdx, t_blockIdx;
extern thread_local int t_parity;
extern dim3 g_gridDim, g_blockDim;

// one wave collective: deposit, wave barrier, return the wave's slots of this round
inline const unsigned long long* exchange(unsigned long long v) {
    const int tid = (int)t_threadIdx.x, lane = tid & 63;
    Wave& w = g_block->waves[tid >> 6];
    const int p = t_parity;
    t_parity ^= 1;
    __atomic_store_n(&w.slot[p][lane], v, __ATOMIC_RELAXED);
    pthread_barrier_wait(&w.bar);
    return w.slot[p];       // rewritten two collectives later at the earliest, i.e. after every lane passed the NEXT wave barrier
}
inline unsigned long long peek(const unsigned long long* s, int lane) { return __atomic_load_n(&s[lane], __ATOMIC_RELAXED); }
}  // namespace emu

#define threadIdx emu::t_threadIdx
#define blockIdx emu::t_blockIdx
#define gridDim emu::g_gridDim
#define blockDim emu::g_blockDim

inline void __syncthreads() { pthread_barrier_wait(&emu::g_block->bar); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline unsigned long long __ballot(int pred) {
    const unsigned long long* s = emu::exchange(pred ? 1ull : 0ull);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) m |= emu::peek(s, l) << l;
    return m;
}
// lanes are threads here: "LDS operations of a wave execute in issue order" becomes a barrier of the wave's threads (every lane calls it)
#define ORBX_WAVE_LDS_SYNC() do { __atomic_thread_fence(__ATOMIC_SEQ_CST); (void)emu::exchange(0); __atomic_thread_fence(__ATOMIC_SEQ_CST); } while (0)
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return __ballot(pred) == ~0ull; }
inline int __shfl(int v, int srcLane) { return (int)(unsigned)emu::peek(emu::exchange((unsigned)v), srcLane & 63); }
inline int __shfl_up(int v, unsigned delta) {      // a lane with no source keeps its own value
    const unsigned long long* s = emu::exchange((unsigned)v);
    const int lane = (int)threadIdx.x & 63;
    return lane >= (int)delta ? (int)(unsigned)emu::peek(s, lane - (int)delta) : v;
}
inline unsigned __shfl_up(unsigned v, unsigned delta) { return (unsigned)__shfl_up((int)v, delta); }
inline int __shfl_xor(int v, int laneMask) {
    const unsigned long long* s = emu::exchange((unsigned)v);
    return (int)(unsigned)emu::peek(s, ((int)threadIdx.x & 63) ^ (laneMask & 63));
}
inline int emu_readlane(int v, int lane) { return __shfl(v, lane); }
#define __builtin_amdgcn_readlane emu_readlane
// v_mbcnt_lo / _hi: base + bits of the mask below this lane (low / high half of the wave)
inline int emu_mbcnt_lo(unsigned mask, int base) { const int l = (int)threadIdx.x & 63; return base + __builtin_popcount(mask & (l >= 32 ? 0xffffffffu : ((1u << l) - 1u))); }
inline int emu_mbcnt_hi(unsigned mask, int base) { const int l = (int)threadIdx.x & 63; return base + (l > 32 ? __builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0); }
#define __builtin_amdgcn_mbcnt_lo emu_mbcnt_lo
#define __builtin_amdgcn_mbcnt_hi emu_mbcnt_hi
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
// v_mov_b32_dpp: row_shr:n (0x110 + n), row_bcast:15 (0x142), row_bcast:31 (0x143); a lane whose row / bank is masked off or whose
// source does not exist keeps `old` (bound_ctrl off) or reads 0 (bound_ctrl on)
inline int emu_update_dpp(int old, int src, int ctrl, int rowMask, int bankMask, bool boundCtrl) {
    const unsigned long long* s = emu::exchange((unsigned)src);
    const int lane = (int)threadIdx.x & 63, row = lane >> 4, inRow = lane & 15, bank = inRow >> 2;
    if (!((rowMask >> row) & 1) || !((bankMask >> bank) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; if (inRow >= n) from = lane - n; }
    else if (ctrl == 0x142) { if (row >= 1) from = (row - 1) * 16 + 15; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    else { fprintf(stderr, "emu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
    if (from < 0) return boundCtrl ? 0 : old;
    return (int)(unsigned)emu::peek(s, from);
}
#define __builtin_amdgcn_update_dpp emu_update_dpp
inline unsigned long long emu_memrealtime() { return 0; }
#define __builtin_amdgcn_s_memrealtime emu_memrealtime

inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T emuAtomicMax(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> inline T emuAtomicMin(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
inline unsigned atomicMax(unsigned* p, unsigned v) { return emuAtomicMax(p, v); }
inline int atomicMax(int* p, int v) { return emuAtomicMax(p, v); }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { return emuAtomicMax(p, v); }
inline int atomicMin(int* p, int v) { return emuAtomicMin(p, v); }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }

// hipLaunchKernelGGL -> run the grid one workgroup at a time, a workgroup as blockDim.x host threads
namespace emu {
template <class K, class... A>
void launch(K kern, dim3 grid, dim3 block, size_t shmem, A... args);
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu::launch(kern, grid, block, shmem, __VA_ARGS__)
#define HIP_SYMBOL(x) x
