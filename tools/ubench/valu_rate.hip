// Microbenchmark: issue rate of the VALU instructions the FAST kernel is built from (gfx950).
// Every wave runs ITER x 64 independent instructions of one kind (8 accumulators, no memory traffic);
// reports SIMD cycles per wave-instruction at a given occupancy.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    unsigned a[8], b = threadIdx.x * 2654435761u, c = blockIdx.x * 40503u + 7;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (threadIdx.x + i * 977u) & 0x00ff00ffu;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define ONE(i)                                                                                                          \
    if (OP == 0) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                            \
    else if (OP == 1) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));               \
    else if (OP == 2) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
    else if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                       \
    else if (OP == 4) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                    \
    else if (OP == 5) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
    else if (OP == 6) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
    else if (OP == 7) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                       \
    else if (OP == 8) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
    else if (OP == 9) asm volatile("v_min3_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                       \
    else if (OP == 10) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));                                          \
    else if (OP == 11) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(ONE)
#undef ONE
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int OP>
double run(unsigned* d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    unsigned* d;
    hipMalloc(&d, 4096);
    const char* names[] = {"v_min3_u32", "v_pk_minimum3_f16", "v_pk_min_u16", "v_perm_b32", "v_min_u32", "v_pk_max_u16",
                           "v_pk_min_f16", "v_max3_u32", "v_pk_sub_u16", "v_min3_i16", "v_bfe_u32", "v_pk_maximum3_f16"};
    const int iters = 2000;
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wavesPerSimd;     // 256 CUs x (4 waves per block = 1 per SIMD) x wavesPerSimd
        double ms[12];
        ms[0] = run<0>(d, blocks, iters); ms[1] = run<1>(d, blocks, iters); ms[2] = run<2>(d, blocks, iters); ms[3] = run<3>(d, blocks, iters);
        ms[4] = run<4>(d, blocks, iters); ms[5] = run<5>(d, blocks, iters); ms[6] = run<6>(d, blocks, iters); ms[7] = run<7>(d, blocks, iters);
        ms[8] = run<8>(d, blocks, iters); ms[9] = run<9>(d, blocks, iters); ms[10] = run<10>(d, blocks, iters); ms[11] = run<11>(d, blocks, iters);
        for (int i = 0; i < 12; i++) {
            const double instrPerWave = (double)iters * 64, ns = ms[i] * 1e6;
            // time per wave-instruction per SIMD (ns) = total time / (instructions per wave * waves per SIMD)
            printf("waves/SIMD %d  %-20s %8.3f ms  %6.3f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", wavesPerSimd, names[i],
                   ms[i], ns / (instrPerWave * wavesPerSimd), ns / (instrPerWave * wavesPerSimd) * 2.4);
        }
    }
    return 0;
}
