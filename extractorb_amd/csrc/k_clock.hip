// k_clock.hip — the clock the chip's waves actually run at while the extraction kernels are running beside them.
//
// Not on the extraction path: bench.py's sustained-load figure (secondary.sustained) launches this probe on a side stream every few hundred
// steps and prices the vector-issue ceiling at the clock it reports, instead of assuming 2.4 GHz (DVFS lowers the clock of a kernel that sits
// at 0.99 of the issue rate for seconds).  One wave per CU reads s_memtime (shader-clock cycles) and s_memrealtime (100 MHz) around a sleep of
// `ticks` real-time ticks: clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md; the method of tools/fast_clock.py,
// which stamps k_fast's own waves in a diagnostic build and agrees with this probe).  The wave sleeps (s_sleep) between polls: it takes a wave
// slot for the window, not issue cycles.  Exit: the real-time counter always advances, and the loop is bounded as well.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbx {

__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* __restrict__ out, int slot, int nWgs, unsigned ticks) {
    // The two counters are read in the SAME order at both ends (shader clock, then real time): each read is a scalar-memory round trip, and with
    // the same order the two windows are shifted against each other by one round trip at both ends instead of one containing the other (ADVICE
    // round 5: real time read first at the start and last at the end made the real-time window two round trips longer than the shader-clock
    // one, i.e. the clock came out low by ~2 latencies per 50 us).  The residual is the DIFFERENCE of two round-trip latencies, not their sum.
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    for (int i = 0; i < (1 << 20) && r1 - r0 < (unsigned long long)ticks; i++) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned long long* o = out + 2 * ((size_t)slot * nWgs + blockIdx.x);
        o[0] = t1 - t0;
        o[1] = r1 - r0;
    }
}

void launchClockProbe(hipStream_t st, unsigned long long* out, int slot, int nWgs, unsigned ticks) {
    hipLaunchKernelGGL(k_clock_probe, dim3(nWgs), dim3(64), 0, st, out, slot, nWgs, ticks);
}

}  // namespace orbx
