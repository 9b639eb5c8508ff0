// orbx_internal.hpp - what the three host files of liborbx.so share: the launch wrappers the kernel files define, the handle, the error / profiling
// helpers.  orbx_api.cpp holds the extraction path (create, geometry installation, the launch sequence, the extract entry points, the pyramid),
// orbx_rows.cpp the rows behind it (stereo matching, Frame finishing, the window searches, the vocabulary and ComputeBoW / SearchByBoW),
// orbx_debug.cpp the introspection, profiling and test aids.  Not installed: callers see include/orbx.h only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "orbx.h"
#include "orbx_geometry.hpp"

namespace orbx {
// launch wrappers, defined in the k_*.hip files
void launchPyrFirst(hipStream_t, const uint8_t*, long long, long long, const LevelGeom&, const LevelGeom*, int, int, int, int,
                    const ResizeX*, const QuadRec*, const ResizeX*, const TileFoot*, uint8_t*, int, int, bool, int, int);
void launchResize(hipStream_t, const LevelGeom&, const LevelGeom&, int, int, const ResizeX*, const QuadRec*, const ResizeX*, const TileFoot*,
                  uint8_t*, int, int, bool, int, int);
struct BowMatchParams { float nnRatio; int thLow, checkOrientation, capacity, kfFirst, kfStep, curFirst, curStep, twoKeyFrames; };
size_t bowMatchLdsBytes(int capacity, bool stageDesc);
void launchSearchBow(hipStream_t, const uint32_t*, const uint32_t*, const int*, const uint8_t*, const uint8_t*, const Keypoint*, const uint8_t*, const int*, const BowMatchParams&, int*, int*, int);
void launchLdsPollute(hipStream_t, int, int, unsigned*);
void launchPyrCols(hipStream_t, const uint8_t*, long long, long long, int, const PyrColumn*, int, const ColLevels*, int, const ResizeX*, int, uint8_t*, int, int, bool, int, int, int);
void launchBlur(hipStream_t, const BlurItem*, const unsigned short*, int, int, const LevelGeom*, const uint8_t*, uint8_t*, int, int);
void launchFast(hipStream_t, const CellDesc*, int, const LevelGeom*, int, const uint8_t*, int, int, unsigned*, unsigned*,
                int, int, int, int, const BlurItem*, const unsigned short*, int, uint8_t*, LeafTables, bool, bool);
bool fastCanCarryBlur(int, int);
size_t octreeLdsBytes(int M, int P, int R, int XT);
void launchOctree(hipStream_t, const LevelGeom*, int, const CellDesc*, int, const unsigned*, const unsigned*, int*, unsigned*,
                  unsigned*, unsigned short*, uint2*, int, int*, int*, const int*, int, int, int, int, const int*, bool, int, int, uint8_t*, LeafTables, bool);
void launchDescribe(hipStream_t, const LevelGeom*, int, const uint8_t*, const uint8_t*, const uint2*, int, const int*,
                    const int*, Keypoint*, uint8_t*, int, int*, int*, Keypoint*, int*, bool, int, int, int, int, int, int);
bool checkUmax(const int* umax16);
hipError_t runPackedSelfTest(hipStream_t, unsigned*, unsigned*);
struct StereoParams {
    float scale[kMaxLevels], invScale[kMaxLevels];
    float bf, b;
    int nlevels, capacity, rowCap;
};
struct CameraParams { float fx, fy, cx, cy, k1, k2, p1, p2, k3; };
struct FrameFinishParams { CameraParams cam; float minX, minY, wInv, hInv; int capacity, rawGrid; };      // (k_frame.hip holds the same layout)
void launchFrameFinish(hipStream_t, const Keypoint*, const int*, const FrameFinishParams&, Keypoint*, int*, int*, int*, int);
void launchStereo(hipStream_t, const LevelGeom*, const uint8_t*, const Keypoint*, const uint8_t*, const int*, const StereoParams&,
                  int, int*, unsigned short*, float*, float*, int*, int*, int);
struct InitMatchParams {
    float minX, minY, wInv, hInv, r, nnRatio;
    int checkOrientation, capacity, slotCapacity, f1First, f1Step, f2First, f2Step;
};
size_t initMatchLdsBytes(int capacity, int slotCapacity);
int initMatchSlotCapacity(int capacity);
void launchSearchInit(hipStream_t, const Keypoint*, const uint8_t*, const int*, const int*, const int*, const InitMatchParams&,
                      float*, int*, int*, int);
struct ProjQuery { float u, v, ur, radius; int minLevel, maxLevel, flags; float angle; };
struct ProjectParams {
    float fx, fy, cx, cy, minX, maxX, minY, maxY;
    float scale[kMaxLevels];
    float mbf, mb, th;
    int mono, capacity, lastFirst, lastStep, curFirst, curStep;
};
struct ProjSearchParams {
    float minX, minY, wInv, hInv, nnRatio;
    int ratioMode, checkOrientation, capacity, queryCapacity, curFirst, curStep, descFirst, descStep, maxDist;
};
size_t projSearchLdsBytes(int capacity, int queryCapacity, bool topList);
void launchProjectLast(hipStream_t, const Keypoint*, const Keypoint*, const int*, const uint8_t*, const float*, const float*, const ProjectParams&,
                       ProjQuery*, int);
void launchSearchProj(hipStream_t, const ProjQuery*, const uint8_t*, const int*, const Keypoint*, const uint8_t*, const int*, const int*,
                      const int*, const float*, uint8_t*, const ProjSearchParams&, int*, int*, int);
struct VocabDevice {
    const int* childOff; const int* childList; const uint32_t* desc; const double* weight; const uint32_t* wordId;
    int nNodes, k, L, scoring, weighting;
};
void launchBow(hipStream_t, const VocabDevice&, const uint8_t*, const int*, int, int, uint32_t*, double*, uint32_t*, uint32_t*, double*, int*,
               uint32_t*, uint32_t*, int*, int);
struct RgbdParams { int capacity, rows, cols, isU16, scale; long long stride, frame; float factor, mbf; };
void launchStereoFromRgbd(hipStream_t, const Keypoint*, const Keypoint*, const int*, const uint8_t*, const RgbdParams&, float*, float*, int);
struct GrayParams { int rows, cols, channels, redFirst, aligned; long long srcStride, srcFrame, dstStride, dstFrame; };
void launchGray(hipStream_t, const uint8_t*, uint8_t*, const GrayParams&, int);
void launchClockProbe(hipStream_t, unsigned long long*, int, int, unsigned);
void launchCopyOut(hipStream_t, const void*, void*, size_t, int);
}  // namespace orbx

using namespace orbx;

extern double g_hostT[8];
extern long g_hostN;
extern const bool g_hostTiming;      // ORBX_HOST_TIMING (tools/host_call_anatomy.py)
static inline double nowSec() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

// Test aids (tests/ only).  They are NOT reachable from the environment: a test sets them through orbx_debug_set_option() before orbx_create,
// so nothing in a deployed process's environment can poison an extractor's memory or make a call fail.
struct TestAids {
    int poison = -1;          // "poison": byte every device allocation of orbx_create is filled with (no kernel may depend on what hipMalloc returns)
    int ldsPollute = -1;      // "lds_pollute": byte every CU's LDS is filled with in front of every kernel
    int colsShape = -1;       // "pyr_cols_shape": pins the workgroup shape of k_pyr_cols (1, 4, 6) so that the parity tests reach every one
    long long sharedUploadBytes = -1;      // "shared_upload_bytes": input copies of at least this size go through the device's shared copy queue (-1: 16 MiB)
    int failAfterFast = 0;    // "fail_after_fast": the next handle's first call with leaf tables returns between k_fast and k_octree (one shot)
};
extern TestAids g_aids;
enum Slot { S_LEVEL0 = 0, S_RESIZE, S_BLUR, S_FAST, S_OCTREE, S_DESCRIBE, S_MISC, S_TOTAL, S_STEREO, S_FRAME };
extern const char* const kSlotNames[ORBX_NUM_KERNELS];
extern thread_local std::string g_createError;

struct EventPair { hipEvent_t a, b; int slot; };

struct orbx_handle {
    int device = 0;
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0;
    float scaleFactor = 0;
    int maxW = 0, maxH = 0, maxB = 0;
    ScaleTables tabs;
    FrameGeom geom;        // geometry of the image size of the last call
    FrameGeom maxGeom;     // geometry of max_width x max_height (sizes the arenas)
    hipStream_t stream = nullptr;
    bool ownStream = false;
    std::string err;

    // arenas (sized once)
    size_t pyrBytes = 0, blurBytes = 0, candEntries = 0, selEntries = 0, cellCap = 0, rxCap = 0, tileCap = 0;
    uint8_t *d_input = nullptr, *d_pyr = nullptr, *d_blur = nullptr;
    unsigned *d_candPos = nullptr, *d_candSeg = nullptr;   // packed (x,y,response): compacted / per-cell segments
    unsigned* d_cellCount = nullptr;                      // [frame][cell] candidates emitted by k_fast
    int* d_cellOff = nullptr;                             // [frame][cell] offset of the cell in the compacted array
    unsigned short* d_nodeOf = nullptr;
    unsigned* d_candCount = nullptr;
    unsigned* d_sink = nullptr;         // written by the LDS polluter (test aid)
    uint2* d_sel = nullptr;
    int *d_levelCount = nullptr, *d_levelLap = nullptr, *d_lap = nullptr;
    LevelGeom* d_lv = nullptr;
    CellDesc* d_cells = nullptr;
    ResizeX *d_rx = nullptr, *d_ry = nullptr;
    QuadRec* d_xq = nullptr;            // per level: the tile resize's dword-column records (FrameGeom::xq)
    size_t xqCap = 0, xqOff[kMaxLevels] = {};
    PyrColumn* d_cols = nullptr;        // regions of the region-major pyramid (k_pyr_cols)
    size_t colsCap = 0, colsOff[4] = {};   // first column of each cut of the geometry in d_cols
    ResizeX* d_colCoef = nullptr;       // the regions' coefficient lists, cut after cut
    size_t colCoefCap = 0, colCoefOff[4] = {};
    ColLevels* d_colLevels = nullptr;
    int pyrCols = -1;                   // ORBX_PYR_COLS: 1 = the region-major pyramid for every batch it fits, 0 = never, default: the smallest batches
    int colPx = 0;                      // ORBX_PYR_COL_PX: side of its regions in level-0 pixels (0: default)
    int fastWide = -1;                  // ORBX_FAST_WIDE: 1 = a workgroup per FAST cell (k_fast_wide) whatever the batch, 0 = never, default: while a call holds few cells
    int patchBlur = -1;                 // ORBX_PATCH_BLUR: 1 = k_describe blurs each keypoint's patch itself (no blurred levels), 0 = never, default: by features per pixel (enqueueBatch)
    int blurSplit = -1;                 // ORBX_BLUR_SPLIT=L: where the blur is per keypoint, only levels < L are (k_describe<PB>); levels >= L are blurred by k_blur and
                                        // described from the blurred level (the patches of the coarse levels hold more pixels than the levels); 0 = no split; default: enqueueBatch
    int colsShape = -1;                 // test aid "pyr_cols_shape": workgroup shape of k_pyr_cols (launchPyrCols: 1, 4 or 6; default: by the grid size)
    TileFoot* d_foot = nullptr;
    size_t footCap = 0, footOff[kMaxLevels] = {};
    BlurItem* d_tiles = nullptr;   // row-block items of the blur kernel: [0] blocks of kBlurBlockRows rows, [1] of kBlurBlockRowsSmall rows
    unsigned short* d_laneItem = nullptr;   // item of every lane of the blur grid, both tables
    size_t laneCap = 0;
    int nBlurLanes[3] = {0, 0, 0};     // [0]: 32-row blocks of every level, [1]: 8-row blocks (small batches), [2]: 32-row blocks of the levels >= splitLevel (the blur split by level)
    size_t blurItemOff[3] = {0, 0, 0}, blurLaneOff[3] = {0, 0, 0};
    int splitLevel = 0;                // the level table [2] starts at (installGeometry; 0: no such table)
    size_t rxOff[kMaxLevels] = {}, ryOff[kMaxLevels] = {};
    size_t octArenaSlice = 0;      // > 0: node arrays of the quad-tree live in d_octArena (too large for LDS)
    uint8_t* d_octArena = nullptr;
    int octM = 0, octP = 0, octR = 0, octXT = 0;   // quad-tree LDS: max nodes, sort size, roots covered by the dense phase
    // small batches: per-(frame, level, root) leaf counters / best keys filled by k_fast's emit, consumed and cleared by k_octree
    // (orbx_device.hpp: LeafTables); d_leafCode = [2][nlevels][octXT] host-built x / y path codes of the current geometry
    int leafFrames = 0;            // frames covered (ORBX_LEAF_FRAMES, default 128; 0 = off)
    int* d_leafHist = nullptr;
    unsigned* d_leafBest = nullptr;
    uint8_t* d_leafCode = nullptr;
    // outputs of the host path: ONE result slab [n | mono | level counts | keypoints | descriptors | per-level keypoints], section-major for the
    // frames of the call (outLayout), on the device (d_out) and in pinned host memory (h_out).  A batch comes back with ONE D2H copy of the part the
    // caller asked for; ONE frame per call (the reference's call shape, Frame.cc:419-427) has no copy at all: the kernels write the slab in
    // pinned host memory themselves (tools/host_zero_copy.py: +3.8 us on the kernels against 4-6 copy commands of ~10 us each)
    int outCap = 0;
    uint8_t *d_out = nullptr, *h_out = nullptr;
    size_t outBytes = 0;
    bool zeroCopy = true;              // ORBX_ZERO_COPY=0: a single frame also goes through d_out and the D2H copy
    // pinned staging
    int* h_lap = nullptr;
    std::vector<int> lapCached;
    uint8_t* h_in = nullptr;           // pageable input of a few frames is gathered here (tight rows), then ONE asynchronous H2D copy
    size_t hInBytes = 0;
    uint8_t* h_pyr = nullptr;          // orbx_fetch_pyramid: one frame's bordered levels (allocated on first use)
    size_t hPyrBytes = 0;
    uint8_t* h_tab = nullptr;          // installGeometry: the geometry's tables, packed, copied to the device arenas on the handle's stream
    size_t hTabBytes = 0;
    // where the results of the last host-buffer batch are: as the kernels see them (dev: d_out, or h_out when written zero-copy) and on the host
    struct OutView { Keypoint* k = nullptr; uint8_t* d = nullptr; int* n = nullptr; int* mono = nullptr; Keypoint* lk = nullptr; int* lc = nullptr; };
    OutView dev, host;
    bool pyramidOnly = false;          // the last call was orbx_compute_pyramid: the handle holds a pyramid (and blurred levels) but no results
    bool blurOwed = false;             // the pyramid part of a call left the blur to the FAST launch that follows
    bool testFailAfterFast = false;    // test aid "fail_after_fast" (orbx_debug_set_option)
    bool leafDirty = false;            // k_fast has filled the leaf tables and k_octree has not been enqueued to clear them
    int lastPyrForm = -1, lastPyrCut = 0, lastBlurForm = -1, lastSplitLevel = 0;      // orbx_debug_last_forms
    std::string lastKernel[ORBX_NUM_KERNELS];                      // the kernel (rocprofv3's name, without template arguments) that last ran in each profile slot
    int lastB = 0;
    int lastHostB = 0;           // frames whose results the handle itself holds (the result slab: h->dev / h->host): set by the host-buffer path only
    int pendingB = 0;            // frames of the batch begun with orbx_extract_batch_begin and not yet ended
    bool pendingLevels = false;
    int octThreads[kMaxLevels] = {};   // quad-tree workgroup size per level (installGeometry: one size for all levels,
                                       // chosen by the image area — separate launches per size measured slower)
    int octThreadsForced = 0;          // ORBX_OCT_THREADS
    int ldsPollute = -1;               // test aid "lds_pollute" (orbx_debug_set_option): every CU's LDS is filled with the byte in front of every kernel
    std::string policy;                // the launch-policy switches as read at orbx_create (orbx_debug_policy)
    bool octRoomyForced = false;       // ORBX_OCT_ROOMY: the 128-VGPR variants whatever the batch (tests reach every variant with it)
    int numCUs = 256;
    bool resizeBytewise = false;    // ORBX_RESIZE_BYTEWISE: force the byte-gather resize (diagnostic)
    // the internal stream and the events of the two-half overlap (enqueueBatch)
    hipStream_t aux = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    long long sharedUploadBytes = -1;      // test aid "shared_upload_bytes" as orbx_create read it (-1: the default limit, uploadFrames)
    hipEvent_t evUp[2] = {nullptr, nullptr};      // uploadFrames: the handle's stream -> the device's shared input-copy queue -> back
    hipStream_t aux2 = nullptr;        // the blur's side stream (pyramid -> {blur, FAST -> quad-tree} -> description), events per half-batch
    hipEvent_t evPyr[2] = {nullptr, nullptr}, evBlur[2] = {nullptr, nullptr};
    int splitMode = 1;                 // ORBX_SPLIT: 1 (default) = large batches overlap their blur with FAST + quad-tree on a side stream (the largest also stagger
                                       // their tails: enqueueBatch); 0 = no overlap of any kind; 3 = staggered tails for every large batch
    bool fuseSmall = true;             // ORBX_FUSE_SMALL=0: small batches keep the blur as a launch of its own
    long long splitMinPixels = 0;      // ORBX_SPLIT_MIN_MPX: smallest half (pyramid pixels) worth its own kernels
    // ComputeBoW scratch (allocated on first use): per-feature word id / weight / node
    size_t bowEntries = 0;
    uint32_t *d_bowWord = nullptr, *d_bowNode = nullptr;
    double* d_bowWeight = nullptr;
    // stereo matching (allocated on first use)
    int stereoPairs = 0, stereoCap = 0, stereoRows = 0;
    int *d_rowOff = nullptr, *d_sadDist = nullptr, *d_nMatched = nullptr;
    unsigned short* d_rowList = nullptr;
    float *d_uRight = nullptr, *d_depth = nullptr;
    // clock probe (orbx_debug_clock_probe: bench.py's sustained-load figure), allocated on first use
    hipStream_t probeStream = nullptr;
    unsigned long long* d_clock = nullptr;
    // profiling
    bool profiling = false;
    std::vector<EventPair> pending;
    double profMs[ORBX_NUM_KERNELS] = {};
    long profN[ORBX_NUM_KERNELS] = {};
};


#define HIP_TRY(h, expr)                                                                                 \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                \
            return ORBX_ERR_HIP;                                                                         \
        }                                                                                                \
    } while (0)

inline int fail(orbx_handle* h, int code, const std::string& msg) {
    h->err = msg;
    return code;
}

inline int nextPow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Byte offsets of the sections of the result slab for a call of B frames (capacity cap per frame).  What every caller wants comes first, so
// that the D2H copy of a batch is one contiguous range: [0, noLevels) without the per-level arrays, [0, all) with them.
struct OutLayout { size_t n, mono, counts, kps, desc, levelK, noLevels, all; };
inline OutLayout outLayout(int B, int cap, int nlevels) {
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    OutLayout o;
    o.n = 0;
    o.mono = up(sizeof(int) * (size_t)B);
    o.counts = o.mono + up(sizeof(int) * (size_t)B);
    o.kps = o.counts + up(sizeof(int) * (size_t)B * nlevels);
    o.desc = o.kps + up(sizeof(Keypoint) * (size_t)cap * B);
    o.noLevels = o.desc + up((size_t)32 * cap * B);
    o.levelK = o.noLevels;
    o.all = o.levelK + up(sizeof(Keypoint) * (size_t)cap * B);
    return o;
}
inline orbx_handle::OutView outView(uint8_t* base, const OutLayout& o) {
    orbx_handle::OutView v;
    v.n = (int*)(base + o.n); v.mono = (int*)(base + o.mono); v.lc = (int*)(base + o.counts);
    v.k = (Keypoint*)(base + o.kps); v.d = base + o.desc; v.lk = (Keypoint*)(base + o.levelK);
    return v;
}

struct Prof {
    orbx_handle* h; int slot; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Prof(orbx_handle* h_, int s, hipStream_t st_ = nullptr) : h(h_), slot(s), st(st_ ? st_ : h_->stream) {
        if (h->profiling) {
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, st);
        }
    }
    ~Prof() {
        if (h->profiling) {
            (void)hipEventRecord(b, st);
            h->pending.push_back(EventPair{a, b, slot});
        }
    }
};

