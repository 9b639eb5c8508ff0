// k_copy.hip — the result slab of a host-buffer batch copied to pinned host memory BY A KERNEL on the handle's own stream.
//
// Why not hipMemcpyAsync: the runtime's copy queue (SDMA) is served in the order the copies were ENQUEUED.  orbx_extract_batch_begin enqueues
// the input copy, the kernels and the result copy of one batch at once; with two handles used alternately the result copy of batch A - which
// cannot start before A's kernels end - then sits in that queue IN FRONT of the input copy of batch B, and B's input waits for A's kernels:
// nothing overlaps (round 6 trace, profiles/r06_host_path.md: input copy 351 us + kernels 265 us + result copy 81 us = 0.70 ms per 64 frames,
// strictly one after the other).  A copy done by a kernel never enters that queue.  Coalesced 16-byte loads from HBM, 16-byte stores to
// device-visible host memory (posted PCIe writes); a few waves per CU are enough to fill the link.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbx {

__global__ __launch_bounds__(256) void k_copy_out(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// bytes: a multiple of 16 (the slab sections are); src / dst 16-byte aligned
void launchCopyOut(hipStream_t st, const void* src, void* dst, size_t bytes, int numCUs) {
    const long long n16 = (long long)(bytes / 16);
    if (n16 <= 0) return;
    const long long want = (n16 + 255) / 256;
    const int grid = (int)(want < (long long)2 * numCUs ? want : (long long)2 * numCUs);
    hipLaunchKernelGGL(k_copy_out, dim3(grid), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n16);
}

}  // namespace orbx
