// Microbenchmark: issue rate of VALU instructions on gfx950 — the integer / packed-16 instructions the kernels are built
// from, next to f32 calibration lines (v_fma_f32, v_add_f32, v_max3_f32, v_pk_fma_f32, v_cvt_f32_ubyte0), so that the
// "vector-instruction issue" roofline is priced against the same yardstick as the hardware guide's 2-cycle v_fma_f32.
//
// Every wave runs ITER x (8 x NACC) instructions of one kind, no memory traffic.  Three dependency shapes:
//   acc8   8 accumulators, each instruction reads its own previous result (a dependent chain per accumulator)
//   acc16  16 accumulators (twice the distance between dependent instructions)
//   indep  the destination is write-only (no read-after-write at all): pure issue
// Reported per (instruction, shape, waves per SIMD): SIMD cycles per wave64 instruction two ways —
//   "ev"  from HIP-event time at the nominal 2.4 GHz, and
//   "mt"  from s_memtime ticks measured inside the waves (the guide: one tick = one shader cycle), which does not
//         depend on the clock the chip actually ran at,
// and "ghz", the clock each cell actually ran at: delta s_memtime / delta s_memrealtime x 100 MHz inside the same waves
// (MI355X_MICROARCH.md, DVFS give-back (6)).  ev x ghz / 2.4 must then equal mt: if it does, the ev / mt gap is the chip
// holding its clock down under a VALU-dense load, not an accounting error.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate [csv [first instruction index]]
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

enum Op {
    OP_MIN3_U32, OP_PK_MINIMUM3_F16, OP_PK_MIN_U16, OP_PERM_B32, OP_MIN_U32, OP_PK_SUB_U16, OP_BFE_U32, OP_ALIGNBYTE,
    OP_DOT4_U32_U8, OP_MAD_U32_U24, OP_ADD_U32, OP_AND_B32, OP_LSHLREV_B32, OP_MUL_LO_U32,
    OP_FMA_F32, OP_ADD_F32, OP_MAX3_F32, OP_PK_FMA_F32, OP_CVT_F32_UBYTE0, OP_PK_ADD_F16, OP_MOV_B32,
    OP_SAT_PK_U8_I16, OP_MIN_U32_SDWA, OP_LSHL_OR_B32, OP_ADD3_U32, OP_MUL_U32_U24, OP_OR_B32, OP_CNDMASK, OP_DOT2_U32_U16, OP_MUL_HI_U32_U24,
    OP_MUL_F32, OP_RNDNE_F32, OP_CVT_I32_F32, OP_FMA_F64, OP_ADD_F64, OP_MBCNT_LO, OP_MOV_DPP, OP_MAX_F32, OP_SUB_U32, OP_LSHL_ADD_U32,
    OP_MAD_U64_U32, OP_LSHL_ADD_U64, OP_MUL_F64, OP_LSHRREV_B64, OP_MUL_I32_I24, OP_RCP_F32, OP_CVT_F64_F32, OP_CVT_F32_F64, OP_DIV_FIXUP_F32, OP_BCNT, OP_CMP_CNDMASK, OP_XOR_B32, OP_PK_LSHRREV_B16, OP_COUNT
};
static const char* kNames[OP_COUNT] = {
    "v_min3_u32", "v_pk_minimum3_f16", "v_pk_min_u16", "v_perm_b32", "v_min_u32", "v_pk_sub_u16", "v_bfe_u32", "v_alignbyte_b32",
    "v_dot4_u32_u8", "v_mad_u32_u24", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mul_lo_u32",
    "v_fma_f32", "v_add_f32", "v_max3_f32", "v_pk_fma_f32", "v_cvt_f32_ubyte0", "v_pk_add_f16", "v_mov_b32",
    "v_sat_pk_u8_i16", "v_min_u32_sdwa(WORD_1)", "v_lshl_or_b32", "v_add3_u32", "v_mul_u32_u24", "v_or_b32", "v_cndmask_b32", "v_dot2_u32_u16", "v_mul_hi_u32_u24",
    "v_mul_f32", "v_rndne_f32", "v_cvt_i32_f32", "v_fma_f64", "v_add_f64", "v_mbcnt_lo_u32_b32", "v_mov_b32_dpp(quad_perm)", "v_max_f32", "v_sub_u32", "v_lshl_add_u32",
    "v_mad_u64_u32", "v_lshl_add_u64", "v_mul_f64", "v_lshrrev_b64", "v_mul_i32_i24", "v_rcp_f32", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_div_fixup_f32", "v_bcnt_u32_b32", "v_cmp_lt_u32+v_cndmask_b32 (pair)", "v_xor_b32", "v_pk_lshrrev_b16"};

// DEP: 1 = the instruction reads its own destination (accumulator chain), 0 = destination is write-only
template <int OP, int DEP>
__device__ __forceinline__ void one(unsigned& a, unsigned long long& a2, unsigned b, unsigned c, unsigned long long b2) {
#define I3(NAME)                                                                                         \
    if (DEP) asm volatile(NAME " %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));                          \
    else asm volatile(NAME " %0, %1, %2, %1" : "=v"(a) : "v"(b), "v"(c));
#define I2(NAME)                                                                                         \
    if (DEP) asm volatile(NAME " %0, %0, %1" : "+v"(a) : "v"(b));                                      \
    else asm volatile(NAME " %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
    if constexpr (OP == OP_MIN3_U32) { I3("v_min3_u32") }
    else if constexpr (OP == OP_PK_MINIMUM3_F16) { I3("v_pk_minimum3_f16") }
    else if constexpr (OP == OP_PK_MIN_U16) { I2("v_pk_min_u16") }
    else if constexpr (OP == OP_PERM_B32) { I3("v_perm_b32") }
    else if constexpr (OP == OP_MIN_U32) { I2("v_min_u32") }
    else if constexpr (OP == OP_PK_SUB_U16) { I2("v_pk_sub_u16") }
    else if constexpr (OP == OP_BFE_U32) {
        if (DEP) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a)); else asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_ALIGNBYTE) { I3("v_alignbyte_b32") }
    else if constexpr (OP == OP_DOT4_U32_U8) {
        if (DEP) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
        else asm volatile("v_dot4_u32_u8 %0, %1, %2, %1" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_MAD_U32_U24) {
        if (DEP) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
        else asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_ADD_U32) { I2("v_add_u32") }
    else if constexpr (OP == OP_AND_B32) { I2("v_and_b32") }
    else if constexpr (OP == OP_LSHLREV_B32) {
        if (DEP) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a)); else asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_MUL_LO_U32) { I2("v_mul_lo_u32") }
    else if constexpr (OP == OP_FMA_F32) {
        if (DEP) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
        else asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_ADD_F32) { I2("v_add_f32") }
    else if constexpr (OP == OP_MAX3_F32) { I3("v_max3_f32") }
    else if constexpr (OP == OP_PK_FMA_F32) {
        if (DEP) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a2) : "v"(b2));
        else asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(a2) : "v"(b2));
    }
    else if constexpr (OP == OP_CVT_F32_UBYTE0) {
        if (DEP) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a)); else asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_PK_ADD_F16) { I2("v_pk_add_f16") }
    else if constexpr (OP == OP_MOV_B32) {
        if (DEP) asm volatile("v_mov_b32 %0, %0" : "+v"(a)); else asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_SAT_PK_U8_I16) {
        if (DEP) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(a)); else asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_MIN_U32_SDWA) {
        if (DEP) asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a) : "v"(b));
        else asm volatile("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_LSHL_OR_B32) {
        if (DEP) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a) : "v"(b)); else asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_ADD3_U32) { I3("v_add3_u32") }
    else if constexpr (OP == OP_MUL_U32_U24) { I2("v_mul_u32_u24") }
    else if constexpr (OP == OP_OR_B32) { I2("v_or_b32") }
    else if constexpr (OP == OP_XOR_B32) { I2("v_xor_b32") }
    else if constexpr (OP == OP_PK_LSHRREV_B16) { I2("v_pk_lshrrev_b16") }
    else if constexpr (OP == OP_CNDMASK) {
        if (DEP) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); else asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_DOT2_U32_U16) {
        if (DEP) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
        else asm volatile("v_dot2_u32_u16 %0, %1, %2, %1" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_MUL_HI_U32_U24) { I2("v_mul_hi_u32_u24") }
    else if constexpr (OP == OP_MUL_F32) { I2("v_mul_f32") }
    else if constexpr (OP == OP_RNDNE_F32) {
        if (DEP) asm volatile("v_rndne_f32 %0, %0" : "+v"(a)); else asm volatile("v_rndne_f32 %0, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_CVT_I32_F32) {
        if (DEP) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a)); else asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_FMA_F64) {
        if (DEP) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(a2) : "v"(b2)); else asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(a2) : "v"(b2));
    }
    else if constexpr (OP == OP_ADD_F64) {
        if (DEP) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(b2)); else asm volatile("v_add_f64 %0, %1, %1" : "=v"(a2) : "v"(b2));
    }
    else if constexpr (OP == OP_MBCNT_LO) {
        if (DEP) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b)); else asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_MOV_DPP) {
        if (DEP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
        else asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_MAX_F32) { I2("v_max_f32") }
    else if constexpr (OP == OP_SUB_U32) { I2("v_sub_u32") }
    else if constexpr (OP == OP_LSHL_ADD_U32) {
        if (DEP) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b)); else asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_MAD_U64_U32) {
        if (DEP) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c) : "s20", "s21");
        else asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %3" : "=v"(a2) : "v"(b), "v"(c), "v"(b2) : "s20", "s21");
    }
    else if constexpr (OP == OP_LSHL_ADD_U64) {
        if (DEP) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a2) : "v"(b2)); else asm volatile("v_lshl_add_u64 %0, %1, 1, %1" : "=v"(a2) : "v"(b2));
    }
    else if constexpr (OP == OP_MUL_F64) {
        if (DEP) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(b2)); else asm volatile("v_mul_f64 %0, %1, %1" : "=v"(a2) : "v"(b2));
    }
    else if constexpr (OP == OP_LSHRREV_B64) {
        if (DEP) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a2)); else asm volatile("v_lshrrev_b64 %0, 1, %1" : "=v"(a2) : "v"(b2));
    }
    else if constexpr (OP == OP_MUL_I32_I24) { I2("v_mul_i32_i24") }
    else if constexpr (OP == OP_RCP_F32) {
        if (DEP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); else asm volatile("v_rcp_f32 %0, %1" : "=v"(a) : "v"(b));
    }
    else if constexpr (OP == OP_CVT_F64_F32) {
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a2) : "v"(DEP ? (unsigned)a2 : b));
    }
    else if constexpr (OP == OP_CVT_F32_F64) {
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(b2));
        if (DEP) a2 ^= a;
    }
    else if constexpr (OP == OP_DIV_FIXUP_F32) { I3("v_div_fixup_f32") }
    else if constexpr (OP == OP_BCNT) {
        if (DEP) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b)); else asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
    }
    else if constexpr (OP == OP_CMP_CNDMASK) {      // a compare that writes vcc and the select that reads it: two instructions per count
        if (DEP) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
        else asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : "vcc");
    }
#undef I3
#undef I2
}

template <int OP, int NACC, int DEP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* ticks, int iters) {
    unsigned a[NACC], b = (threadIdx.x * 2654435761u) & 0x00ff00ffu, c = (blockIdx.x * 40503u + 7) & 0x00ff00ffu;
    unsigned long long a2[NACC], b2 = ((unsigned long long)__float_as_uint(1.0f) << 32) | __float_as_uint(0.5f);
#pragma unroll
    for (int i = 0; i < NACC; i++) { a[i] = (threadIdx.x + i * 977u) & 0x00ff00ffu; a2[i] = b2 + i; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) one<OP, DEP>(a[i], a2[i], b, c, b2);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) r ^= a[i] ^ (unsigned)a2[i] ^ (unsigned)(a2[i] >> 32);
    if (r == 0x12345678u) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6), n = gridDim.x * 4;
        ticks[w] = t1 - t0; ticks[n + w] = r1 - r0; ticks[2 * n + w] = r0; ticks[3 * n + w] = r1;      // (absolute stamps: when was the wave resident?)
    }
}

struct Res { double evCycles, mtCycles, ghz, resident; };

template <int OP, int NACC, int DEP>
Res run(unsigned* d, unsigned long long* dT, int blocks, int wavesPerSimd) {
    const int iters = 16000 / NACC;                  // 8 * NACC * iters = 128000 instructions per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, NACC, DEP>), dim3(blocks), dim3(256), 0, 0, d, dT, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, NACC, DEP>), dim3(blocks), dim3(256), 0, 0, d, dT, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    std::vector<unsigned long long> t((size_t)blocks * 16);
    hipMemcpy(t.data(), dT, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0, sumReal = 0;
    for (size_t i = 0; i < (size_t)blocks * 4; i++) { sum += (double)t[i]; sumReal += (double)t[(size_t)blocks * 4 + i]; }
    const double instrPerWave = 8.0 * NACC * iters;
    Res r;
    r.evCycles = ms * 1e6 / (instrPerWave * wavesPerSimd) * 2.4;                 // ns per wave-instr per SIMD x 2.4 GHz
    r.mtCycles = sum / ((double)blocks * 4) / (instrPerWave * wavesPerSimd);     // a wave's ticks cover wavesPerSimd waves' instructions
    r.ghz = sum / sumReal * 0.1;                                                 // s_memrealtime ticks at 100 MHz
    // how many waves were resident per SIMD on average while the kernel ran: sum of wave lifetimes / (first start .. last end) / 1024 SIMDs.
    // Below the nominal waves per SIMD the grid did not run as one resident set (the dispatcher placed it in rounds), and "mt" - a wave's
    // own ticks spread over the NOMINAL number of co-resident waves - under-states the cycles a SIMD spent per instruction
    unsigned long long first = ~0ull, last = 0;
    for (size_t i = 0; i < (size_t)blocks * 4; i++) { first = std::min(first, t[(size_t)blocks * 8 + i]); last = std::max(last, t[(size_t)blocks * 12 + i]); }
    r.resident = sumReal / (double)(last - first) / 1024.0;
    return r;
}

template <int OP>
void runOp(unsigned* d, unsigned long long* dT, bool csv) {
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wavesPerSimd;     // 256 CUs x (4 waves per block = 1 per SIMD) x wavesPerSimd
        const Res a8 = run<OP, 8, 1>(d, dT, blocks, wavesPerSimd), a16 = run<OP, 16, 1>(d, dT, blocks, wavesPerSimd),
                  in = run<OP, 8, 0>(d, dT, blocks, wavesPerSimd);
        if (csv) printf("%s,%d,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%.2f,%.2f,%.2f\n", kNames[OP], wavesPerSimd, a8.evCycles, a8.mtCycles, a16.evCycles, a16.mtCycles, in.evCycles, in.mtCycles, a8.ghz, a16.ghz, in.ghz, a8.resident, a16.resident, in.resident);
        else printf("%-20s waves/SIMD %d   acc8 ev %6.2f mt %6.2f   acc16 ev %6.2f mt %6.2f   indep ev %6.2f mt %6.2f  cycles per wave64 instr per SIMD; clock %.2f / %.2f / %.2f GHz; waves resident per SIMD %.2f / %.2f / %.2f\n",
                    kNames[OP], wavesPerSimd, a8.evCycles, a8.mtCycles, a16.evCycles, a16.mtCycles, in.evCycles, in.mtCycles, a8.ghz, a16.ghz, in.ghz, a8.resident, a16.resident, in.resident);
        fflush(stdout);
    }
}

template <int OP>
void runAll(unsigned* d, unsigned long long* dT, bool csv, int first) {
    if constexpr (OP < OP_COUNT) { if (OP >= first) runOp<OP>(d, dT, csv); runAll<OP + 1>(d, dT, csv, first); }
}

int main(int argc, char** argv) {
    const bool csv = argc > 1 && !strcmp(argv[1], "csv");
    unsigned* d; unsigned long long* dT;
    hipMalloc(&d, 4096);
    hipMalloc(&dT, 256 * 8 * 4 * 4 * sizeof(unsigned long long));
    if (csv) printf("instruction,waves_per_simd,acc8_ev,acc8_mt,acc16_ev,acc16_mt,indep_ev,indep_mt,acc8_ghz,acc16_ghz,indep_ghz,acc8_resident,acc16_resident,indep_resident\n");
    const int first = argc > 2 ? atoi(argv[2]) : 0;      // (./valu_rate csv N: only the instructions from index N on - additions to the table)
    runAll<0>(d, dT, csv, first);
    return 0;
}
