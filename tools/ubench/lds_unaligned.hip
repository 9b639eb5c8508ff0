// Microbenchmark: what a ds_read_b64 / ds_read_b32 costs on gfx950 when its address is only 2- or 4-byte aligned (the guide's LDS table is for
// naturally aligned accesses).  Question behind it (k_fast): a pixel tile kept as u16 per pixel would hand the score pass its
// ring pairs (2 x u16 per register) straight from one ds_read_b64 per ring position, without the 34 v_perm_b32 per item - but ten of the sixteen
// ring offsets are odd, i.e. the 8-byte read starts 2 bytes off a dword.
//
// Every wave issues ITER x 16 reads of one kind (addresses differ per read so nothing is merged), xors the results and stores one dword.
// Shapes: "row" = the lanes of a wave read 8 bytes each, contiguous (lane * 8); "tile" = k_fast's item layout, 8 lanes per tile row of 96 bytes.
// Reported: LDS cycles per wave-instruction per CU (HIP events, 2.4 GHz nominal, 16 waves per CU).
//   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <vector>

constexpr int ITER = 2000;

template <int BYTES, int OFF, bool TILE>
__global__ __launch_bounds__(256) void k(unsigned* out) {
    __shared__ __align__(16) unsigned char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += 256) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (TILE ? (lane >> 3) * 96 + (lane & 7) * 8 : lane * 8) + OFF + wave * 3072;
    unsigned acc = 0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned a = base + ((j * 96 + it * 8) & 1023);
            if constexpr (BYTES == 8) {
                uint2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc ^= v.x ^ v.y;
            } else {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc ^= v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int BYTES, int OFF, bool TILE>
static void run(unsigned* d, const char* name) {
    const int grid = 256 * 4;      // 4 workgroups of 4 waves per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BYTES, OFF, TILE>), dim3(grid), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BYTES, OFF, TILE>), dim3(grid), dim3(256), 0, 0, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instrPerCU = 16.0 * ITER * 16;      // 16 waves per CU
    printf("| %-28s | %6.2f |\n", name, ms * 1e-3 * 2.4e9 / instrPerCU);
}

int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    printf("| read | LDS cycles per wave-instruction per CU |\n|---|---|\n");
    run<8, 0, false>(d, "ds_read_b64 row  +0");
    run<8, 2, false>(d, "ds_read_b64 row  +2");
    run<8, 4, false>(d, "ds_read_b64 row  +4");
    run<8, 6, false>(d, "ds_read_b64 row  +6");
    run<8, 0, true>(d, "ds_read_b64 tile +0");
    run<8, 2, true>(d, "ds_read_b64 tile +2");
    run<8, 4, true>(d, "ds_read_b64 tile +4");
    run<8, 6, true>(d, "ds_read_b64 tile +6");
    run<4, 0, false>(d, "ds_read_b32 row(8 B stride) +0");
    run<4, 2, false>(d, "ds_read_b32 row(8 B stride) +2");
    run<4, 0, true>(d, "ds_read_b32 tile +0");
    run<4, 2, true>(d, "ds_read_b32 tile +2");
    hipFree(d);
    return 0;
}
