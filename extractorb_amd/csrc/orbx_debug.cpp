// orbx_debug.cpp - introspection, event profiling and test aids of liborbx.so (include/orbx.h, "introspection used by tests and bench.py"):
// nothing here is on the extraction path.
#include "orbx_internal.hpp"

extern "C" {

// ORBX_HOST_TIMING=1: where a one-frame host call spends its wall time (tools/host_call_anatomy.py): seconds accumulated per phase
int orbx_debug_host_timing(double* out8, long* calls) {
    if (!out8 || !calls) return ORBX_ERR_BAD_ARGUMENT;
    for (int i = 0; i < 8; i++) { out8[i] = g_hostT[i]; g_hostT[i] = 0; }
    *calls = g_hostN; g_hostN = 0;
    return ORBX_OK;
}

int orbx_debug_set_option(const char* name, int value) {
    if (!name) return ORBX_ERR_BAD_ARGUMENT;
    const std::string n(name);
    if (n == "poison") g_aids.poison = value;
    else if (n == "lds_pollute") g_aids.ldsPollute = value;
    else if (n == "fail_after_fast") g_aids.failAfterFast = value;
    else if (n == "pyr_cols_shape") g_aids.colsShape = value;
    else if (n == "shared_upload_bytes") g_aids.sharedUploadBytes = value;
    else return ORBX_ERR_BAD_ARGUMENT;
    return ORBX_OK;
}

const char* orbx_debug_policy(const orbx_handle* h) { return h ? h->policy.c_str() : ""; }

// The shader clock while the handle's work is running: one sleeping wave per CU on a stream of its own (k_clock.hip), asynchronous.
enum { kClockSlots = 64, kClockTicks = 5000 };      // 5000 ticks of 100 MHz = 50 us per probe
int orbx_debug_clock_probe(orbx_handle* h, int slot) {
    if (!h || slot < 0 || slot >= kClockSlots) return ORBX_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->probeStream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->probeStream, hipStreamNonBlocking));
        HIP_TRY(h, hipMalloc(&h->d_clock, sizeof(unsigned long long) * 2 * kClockSlots * h->numCUs));
        HIP_TRY(h, hipMemsetAsync(h->d_clock, 0, sizeof(unsigned long long) * 2 * kClockSlots * h->numCUs, h->probeStream));
    }
    launchClockProbe(h->probeStream, h->d_clock, slot, h->numCUs, kClockTicks);
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}
int orbx_debug_clock_read(orbx_handle* h, int n_slots, double* ghz) {
    if (!h || !ghz || n_slots < 1 || n_slots > kClockSlots) return ORBX_ERR_BAD_ARGUMENT;
    if (!h->probeStream) return fail(h, ORBX_ERR_BAD_ARGUMENT, "orbx_debug_clock_read: no probe was launched");
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<unsigned long long> v((size_t)2 * n_slots * h->numCUs);
    HIP_TRY(h, hipMemcpyAsync(v.data(), h->d_clock, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->probeStream));
    HIP_TRY(h, hipStreamSynchronize(h->probeStream));
    for (int s = 0; s < n_slots; s++) {
        double dt = 0, dr = 0;
        for (int w = 0; w < h->numCUs; w++) { dt += (double)v[2 * ((size_t)s * h->numCUs + w)]; dr += (double)v[2 * ((size_t)s * h->numCUs + w) + 1]; }
        ghz[s] = dr > 0 ? 0.1 * dt / dr : 0.0;      // cycles per 10-ns tick -> GHz
    }
    return ORBX_OK;
}

int orbx_debug_last_forms(const orbx_handle* h, int* pyramid_form, int* pyramid_cut_px, int* blur_form) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (pyramid_form) *pyramid_form = h->lastPyrForm;
    if (pyramid_cut_px) *pyramid_cut_px = h->lastPyrCut;
    if (blur_form) *blur_form = h->lastBlurForm;
    return ORBX_OK;
}

int orbx_debug_last_split_level(const orbx_handle* h) { return h && h->lastBlurForm == 5 ? h->lastSplitLevel : 0; }

int orbx_debug_num_candidates(orbx_handle* h, int frame, int level, int* n) {
    if (!h || !n) return ORBX_ERR_BAD_ARGUMENT;
    if (h->geom.nlevels == 0 || frame < 0 || frame >= h->lastB || level < 0 || level >= h->nlevels)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "no such frame/level in the last batch");
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned v = 0;
    HIP_TRY(h, hipMemcpyAsync(&v, h->d_candCount + frame * h->nlevels + level, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *n = (int)v;
    return ORBX_OK;
}

int orbx_debug_get_candidates(orbx_handle* h, int frame, int level, orbx_keypoint* out, int capacity) {
    int n = 0;
    int rc = orbx_debug_num_candidates(h, frame, level, &n);
    if (rc != ORBX_OK) return rc;
    const LevelGeom& L = h->geom.lv[level];
    if (n > L.candCap) return fail(h, ORBX_ERR_CAPACITY, "candidate arena overflow (internal bound violated)");
    if (n > capacity) return fail(h, ORBX_ERR_CAPACITY, "capacity too small");
    if (n == 0) return ORBX_OK;
    // The quad-tree compacts the keys into candPos only when it has to sweep them more than once; the per-cell segments
    // k_fast wrote always hold them, in the reference's order (cell by cell, raster order inside a cell).
    const FrameGeom& g = h->geom;
    const int nCells = (int)g.cells.size();
    const size_t eb = g.big ? sizeof(CandFmt<true>::T) : sizeof(CandFmt<false>::T);      // one dword per candidate, two for frames beyond 4096 px (orbx_device.hpp)
    std::vector<unsigned> counts(L.cellCount);
    std::vector<unsigned long long> seg(L.candCap);      // (room for either format)
    HIP_TRY(h, hipMemcpyAsync(counts.data(), h->d_cellCount + (long long)frame * nCells + L.cellFirst, sizeof(unsigned) * L.cellCount,
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(seg.data(), (const uint8_t*)h->d_candSeg + (size_t)(L.candOff + (long long)frame * L.candCap) * eb, eb * L.candCap,
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int at = 0;
    for (int c = 0; c < L.cellCount; c++) {
        const int so = g.cells[L.cellFirst + c].segOff;
        for (unsigned i = 0; i < counts[c] && at < n; i++) {
            orbx_keypoint& k = out[at++];
            if (g.big) {
                const unsigned long long w = seg[so + i];
                k.x = (float)((unsigned)w & 0xffffu); k.y = (float)((unsigned)w >> 16); k.response = (float)(unsigned)(w >> 32);
            } else {
                const unsigned w = ((const unsigned*)seg.data())[so + i];
                k.x = (float)(w & 0xfff); k.y = (float)((w >> 12) & 0xfff); k.response = (float)(w >> 24);
            }
            k.size = 7.f; k.angle = -1.f; k.octave = 0; k.class_id = -1;
        }
    }
    if (at != n) return fail(h, ORBX_ERR_HIP, "candidate count and per-cell counts disagree (internal)");
    return ORBX_OK;
}

int orbx_debug_get_blurred(orbx_handle* h, int frame, int level, uint8_t* dst, ptrdiff_t dst_stride) {
    if (!h || !dst) return ORBX_ERR_BAD_ARGUMENT;
    if (h->geom.nlevels == 0 || frame < 0 || frame >= h->lastB || level < 0 || level >= h->nlevels)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "no such frame/level in the last batch");
    HIP_TRY(h, hipSetDevice(h->device));
    const LevelGeom& L = h->geom.lv[level];
    if (dst_stride < L.w) return fail(h, ORBX_ERR_BAD_ARGUMENT, "dst_stride too small");
    if (h->lastBlurForm == 3 || (h->lastBlurForm == 5 && level < h->lastSplitLevel))
        return fail(h, ORBX_ERR_UNSUPPORTED, "the last call blurred this level per keypoint inside k_describe: no blurred level exists (ORBX_PATCH_BLUR=0 keeps k_blur)");
    HIP_TRY(h, hipMemcpy2DAsync(dst, dst_stride, h->d_blur + L.blurOff + (long long)frame * L.blurFrameBytes, L.blurStride, L.w,
                                L.h, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ORBX_OK;
}

int orbx_profile_enable(orbx_handle* h, int enable) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    h->profiling = enable != 0;
    return ORBX_OK;
}
int orbx_profile_reset(orbx_handle* h) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (auto& ev : h->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    h->pending.clear();
    for (int i = 0; i < ORBX_NUM_KERNELS; i++) { h->profMs[i] = 0; h->profN[i] = 0; }
    return ORBX_OK;
}
int orbx_profile_read(orbx_handle* h, double* total_ms, long* launches) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (auto& ev : h->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { h->profMs[ev.slot] += ms; h->profN[ev.slot]++; }
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b);
    }
    h->pending.clear();
    for (int i = 0; i < ORBX_NUM_KERNELS; i++) {
        if (total_ms) total_ms[i] = h->profMs[i];
        if (launches) launches[i] = h->profN[i];
    }
    return ORBX_OK;
}
const char* orbx_profile_kernel_name(int slot) { return slot >= 0 && slot < ORBX_NUM_KERNELS ? kSlotNames[slot] : ""; }
const char* orbx_profile_kernel_name_of(const orbx_handle* h, int slot) {
    if (!h || slot < 0 || slot >= ORBX_NUM_KERNELS) return "";
    return h->lastKernel[slot].empty() ? kSlotNames[slot] : h->lastKernel[slot].c_str();
}

long orbx_algorithmic_bytes(const orbx_handle* h, int rows, int cols, int n_out) {
    if (!h) return 0;
    FrameGeom g;
    if (!makeFrameGeom(h->tabs, rows, cols, g).empty()) return 0;
    return (long)rows * cols + 2 * (long)g.sumPixels + 60L * n_out;
}

}  // extern "C"
