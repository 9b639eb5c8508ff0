// When do the XCDs of an MI355X start the workgroups of one launch?  Every workgroup stamps s_memrealtime (100 MHz, one clock for the
// whole device) and its XCC_ID; per launch the first stamp of each XCD relative to the earliest one is printed.  The launches follow a
// short kernel on the same stream (the product's situation: four dependent kernels per call).
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o xcd_start tools/ubench/xcd_start.hip && ./xcd_start
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_prev(unsigned* sink) { if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1; }
__global__ void k_stamp(unsigned long long* t, unsigned* xcc, int spin) {
    if (threadIdx.x == 0) {
        t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xf;
    }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(1);
}
int main() {
    const int n = 256;
    unsigned long long* d_t; unsigned *d_x, *d_s;
    hipMalloc(&d_t, n * 8); hipMalloc(&d_x, n * 4); hipMalloc(&d_s, 4);
    std::vector<unsigned long long> t(n); std::vector<unsigned> x(n);
    for (int rep = 0; rep < 12; rep++) {
        const int wg = rep < 6 ? 256 : 512;
        hipLaunchKernelGGL(k_prev, dim3(64), dim3(256), 0, 0, d_s);
        hipLaunchKernelGGL(k_stamp, dim3(n), dim3(wg), 0, 0, d_t, d_x, 200);
        hipDeviceSynchronize();
        hipMemcpy(t.data(), d_t, n * 8, hipMemcpyDeviceToHost); hipMemcpy(x.data(), d_x, n * 4, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, first[16]; for (auto& f : first) f = ~0ull;
        for (int i = 0; i < n; i++) { if (t[i] < t0) t0 = t[i]; if (t[i] < first[x[i]]) first[x[i]] = t[i]; }
        printf("launch %2d (%d threads): blockIdx 0..7 on XCC", rep, wg);
        for (int i = 0; i < 8; i++) printf(" %u", x[i]);
        printf(" | first start per XCC (us):");
        for (int i = 0; i < 8; i++) printf(" %.2f", (first[i] - t0) / 100.0);
        printf("\n");
    }
    return 0;
}
