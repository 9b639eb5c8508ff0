// orbx_device.hpp — plain-old-data shared by the host geometry code and the HIP kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace orbx {

constexpr int kEdge = 19;        // EDGE_THRESHOLD, ORBextractor.cc:72
constexpr int kPadL = 32;        // bytes in front of each interior row in HBM (>= kEdge, keeps rows 32-B aligned)
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, ORBextractor.cc:71
constexpr int kPatch = 31;       // PATCH_SIZE, ORBextractor.cc:70
constexpr int kMaxLevels = 16;
constexpr int kCellW = 30;       // W, ORBextractor.cc:777
constexpr int kMinBorder = kEdge - 3;   // minBorderX/Y, ORBextractor.cc:781-782


struct Keypoint {             // == orbx_keypoint == cv::KeyPoint (28 bytes)
    float x, y, size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(Keypoint) == 28, "cv::KeyPoint layout");

// ---- packed formats between the kernels ----
// A FAST candidate (k_fast -> k_octree; rectangle coordinates, ORBextractor.cc:857-859) is packed two ways:
//   narrow (default): one dword  x:12 | y:12 | response:8   - every level of the geometry at most 4096 + 2 x 16 px wide and high;
//   big   (round 6):  two dwords [x:16 | y:16] [response]   - any larger frame (the reference has no size limit, :1171).
// The selection entry (k_octree -> k_describe, level coordinates: + kMinBorder) has one format: .x = x:16 | y:16, .y = response:8 << 24 |
// lapping flag << 23 | rank among the level's lapping keys (23 bits).
template <bool BIG> struct CandFmt;
template <> struct CandFmt<false> {
    typedef unsigned T;
#if defined(__HIPCC__) || defined(ORBX_OCT_EMU)
    static __host__ __device__ inline T make(unsigned x, unsigned y, unsigned resp) { return x | (y << 12) | (resp << 24); }
    static __host__ __device__ inline int x(T w) { return (int)(w & 0xfff); }
    static __host__ __device__ inline int y(T w) { return (int)((w >> 12) & 0xfff); }
    static __host__ __device__ inline unsigned xy32(T w) { return (w & 0xfff) | (((w >> 12) & 0xfff) << 16); }
    static __host__ __device__ inline unsigned respKey(T w) { return w & 0xff000000u; }      // response << 24
#endif
};
template <> struct CandFmt<true> {
    typedef unsigned long long T;
#if defined(__HIPCC__) || defined(ORBX_OCT_EMU)
    static __host__ __device__ inline T make(unsigned x, unsigned y, unsigned resp) { return (T)(x | (y << 16)) | ((T)resp << 32); }
    static __host__ __device__ inline int x(T w) { return (int)((unsigned)w & 0xffffu); }
    static __host__ __device__ inline int y(T w) { return (int)((unsigned)w >> 16); }
    static __host__ __device__ inline unsigned xy32(T w) { return (unsigned)w; }
    static __host__ __device__ inline unsigned respKey(T w) { return (unsigned)(w >> 32) << 24; }
#endif
};
constexpr int kNarrowMax = 4096;      // the narrow format's level size limit (rectangle coordinates < 4096)
constexpr int kBigMax = 16384;        // the big format's: box coordinates are shorts, slots 24 bits (makeFrameGeom checks both)

// One kBlurBlockRows-row block of one level for the blur kernel: lanes [firstLane, firstLane + ceil(w/4)) own its 4-column groups.
// items[0].count holds the number of items.
struct BlurItem { int firstLane; int count; short level; short y0; };

// ---- per-level geometry, shared verbatim with the device (plain ints/floats) -------------------
struct LevelGeom {
    int w, h;                 // level image size
    int pyrStride;            // bytes per row of the bordered level buffer (multiple of 64)
    int pyrRows;              // h + 2*kEdge
    long long pyrOff;         // byte offset of frame 0's bordered buffer inside the pyramid arena
    long long pyrFrameBytes;  // bytes per frame at this level (level-major arena: all frames of a level are adjacent)
    int blurStride;           // bytes per row of the blurred level (multiple of 64)
    long long blurOff, blurFrameBytes;
    int nCols, nRows, wCell, hCell;   // FAST cell grid (ORBextractor.cc:789-795)
    int rectW, rectH;         // maxBorder - minBorder (ORBextractor.cc:783-790)
    int quota;                // mnFeaturesPerLevel[level]
    int nIni;                 // quad-tree roots (ORBextractor.cc:548)
    float hX;                 // root width (ORBextractor.cc:550)
    int candCap;              // exact upper bound of FAST candidates of one frame at this level
    long long candOff;        // entry offset of frame 0 in the candidate arena (level-major)
    int selCap;               // upper bound of keypoints kept at this level
    int selOff;               // entry offset of this level inside one frame's selection slab
    float scale;              // mvScaleFactor[level]
    int patchSize;            // keypoint size written to the output
    int cellFirst, cellCount; // this level's cells inside the per-frame cell table
    int rxOff, ryOff;         // this level's resize coefficient records inside the handle's x / y tables (level >= 1)
    int leafOK;               // 1: the quad-tree's dense phase covers this level, so in small batches k_fast may histogram its keys (LeafTables)
};

struct CellDesc {             // one FAST cell == one cv::FAST call of the reference (ORBextractor.cc:818-819)
    short level;
    short roiW, roiH;         // maxX-iniX, maxY-iniY
    short x0, y0;             // iniX, iniY in level pixel coordinates
    short shiftX, shiftY;     // j*wCell, i*hCell (ORBextractor.cc:857-858)
    short pad;
    int cellId;               // i*nCols + j: raster rank of the cell inside its level
    int segOff;               // first slot of this cell's candidate segment inside the level's per-frame arena
    // where the cell's level lives in the pyramid arena (copied from LevelGeom by installGeometry): k_fast's staging loads then depend on
    // ONE wave-uniform load (this record), not on a second look-up in the level table — a memory round trip off a single frame's critical path
    int pyrStride;
    int itemRecip;            // ceil(65536 / fastItemsPerRow(roiW - 6)): k_fast deals lanes to (row, item) with a multiply instead of a division
    long long pyrOff, pyrFrameBytes;
};
// k_fast's packed passes: four-pixel items per interior row of a cell cw pixels wide (the interior starts on the tile's dword 1)
constexpr int kFastTileShift = 1;      // tile byte of ROI pixel 0 (k_fast.hip: mis)
inline constexpr int fastItemsPerRow(int cw) { return ((kFastTileShift + 2 + cw) >> 2) - ((kFastTileShift + 3) >> 2) + 1; }
static_assert(sizeof(CellDesc) == 48, "CellDesc layout");

// ---- quad-tree dense phase: leaf grid of a root (k_octree.hip) -------------------------------------------------------------------
// DivideNode's boxes depend only on the root box, and its x and y decisions are independent, so a key's quadrant path of length
// kOctDepth below its root is two table look-ups.  Shared by the kernels (k_octree builds the tables in LDS; k_fast reads the host-built
// copies in small batches) and the host.
constexpr int kOctDepth = 5;
constexpr int kOctLeaves = 1 << (2 * kOctDepth);      // 1024 leaf cells per root
#ifdef __HIPCC__
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif
// left/right (or up/down) decisions of DivideNode along one axis for coordinate v in the box [b0, b1)
ORBX_HD inline int octAxisPath(int v, int b0, int b1) {
    int path = 0;
    for (int d = 0; d < kOctDepth; d++) {
        const int c = b0 + ((b1 - b0 + 1) >> 1);    // UL + ceil(extent/2)  (ORBextractor.cc:488-489)
        const int bit = v < c ? 0 : 1;               // kp.pt.x < n1.UR.x  (:520)
        path = 2 * path + bit;
        if (bit) b0 = c; else b1 = c;
    }
    return path;
}
// root << kOctDepth | x path of rectangle column x (vpIniNodes[kp.pt.x / hX], :574; IEEE single division on host and device alike)
ORBX_HD inline int octXCode(int x, float hX, int nIni) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r = (int)__fdiv_rn((float)x, hX);
#else
    int r = (int)((float)x / hX);
#endif
    r = r > nIni - 1 ? nIni - 1 : r;
    return (r << kOctDepth) | octAxisPath(x, (int)(hX * (float)r), (int)(hX * (float)(r + 1)));
}
// Small batches (one (frame, level) quad-tree problem per CU cannot hide its first sweep over ~60 k keys): k_fast's emit adds every kept key
// to its leaf's counter and best key in these L2-resident tables, and k_octree starts from them instead of sweeping the segments.
//   hist / best : [frame][level][R][kOctLeaves]   (zero between calls: k_octree clears what it loads)
//   xcode / ycode : [level][XT] host-built octXCode / octAxisPath of every rectangle column / row
struct LeafTables { int* hist; unsigned* best; const uint8_t* xcode; const uint8_t* ycode; int R, XT, nlevels, frames; };
#ifdef __HIPCC__
// Entry of leaf cell (x code xc, y code yc) of frame f, level `level` in hist / best (32 bits: orbx_create keeps frames x levels x roots x 1024 below 2^30)
__device__ __forceinline__ unsigned leafTableEntry(const LeafTables& lt, int f, int level, int xc, int yc) {
    return ((unsigned)(f * lt.nlevels + level) * (unsigned)lt.R + (unsigned)(xc >> kOctDepth)) * kOctLeaves +
           (unsigned)((yc << kOctDepth) | (xc & ((1 << kOctDepth) - 1)));
}
#endif


constexpr int kBlurBlockRows = 32;    // output rows one lane of k_blur walks (plus a 6-row halo)
#ifndef ORBX_BLUR_SMALL_ROWS
#define ORBX_BLUR_SMALL_ROWS 8
#endif
constexpr int kBlurBlockRowsSmall = ORBX_BLUR_SMALL_ROWS;   // ... in small batches (latency, not throughput)
#ifndef ORBX_RESIZE_TILE_ROWS
#define ORBX_RESIZE_TILE_ROWS 32
#endif
constexpr int kResizeTileRows = ORBX_RESIZE_TILE_ROWS;   // destination rows per workgroup tile of k_pyr_first / k_resize (256 pixels wide)

// Source footprint of one 256 x kResizeTileRows destination tile of the resize kernel (host-computed from the coefficient tables)
// A resize tile's staged source rectangle: first column (a multiple of 4; negative = inside the level's border), dwords per row, first
// row, rows: the tile's tap footprint.
struct TileFoot { short fx0, nDw, fy0, nRows; };

struct ResizeX { short sx0, sx1, a0, a1; };   // two source columns (or rows) and their 11-bit weights for one output column (row)

// Largest rectangle (any level / the loaded one) and most coefficient records (all steps of one region) k_pyr_cols' LDS staging holds
constexpr int kChainMaxW = 256, kChainMaxH0 = 96, kChainCoefMax = 2048;
struct ChainRegion { short x0, y0, w, h; };

// Region-major pyramid (k_pyr_cols): the image is cut into RX x RY regions; ONE workgroup builds its region of EVERY level, level after level
// in LDS (level l + 1 is resized from the rounded pixels of level l, as the reference's chain does), and writes the bordered bytes it owns of
// each level as it goes.  region[l] = the rectangle of level l's interior the workgroup holds (what it owns of level l, the pixels its border
// bytes mirror, and the taps of region[l + 1]; x0 a multiple of 4); own[l] = the dword columns x rows of the BORDERED level l it writes (the
// own rectangles of the columns partition every level).  Nothing is derived twice except the regions' overlap: 640x480, levels 1-7 = 0.64 M pixels,
// the four cuts derive 1.40 / 1.10 / 0.86 / 0.81 M per frame (round 2's per-tile chains, removed in round 4, re-derived 6.6 M).
// What a thread of the packed resize step needs of its four adjacent columns, worked out by the host per region and level (k_pyr_cols): the
// aligned start and the byte shift of the 8-byte tap window inside the region's row, the v_perm selectors that cut each column's two taps out of
// it, the weight pairs.  (Derived in the kernel from four ResizeX records it was ~45 of the ~100 vector instructions of a one-row step.)
struct QuadRec { unsigned sel[4]; unsigned wt[4]; int baseSh; int pad[3]; };      // baseSh = base | shift << 16;  48 bytes = six 8-byte units
// A destination row of the same step: its two source rows dealt to two BANKS of horizontal-pass results (A: the even source row, B: the odd
// one; a row whose taps coincide or share a parity: tap 0 in A, tap 1 in B), each with its 11-bit weight already shifted for the vertical
// multiply.  A thread that walks down consecutive destination rows keeps a bank while its source row stays: at scale 1.2 five rows in six
// re-use one of the two, with no register copies and no search ("does the row I need sit in the other set?").  16 bytes = two 8-byte units.
struct RowRec { unsigned bA, bB; int sA, sB; };
ORBX_HD inline RowRec makeRowRec(const ResizeX& cy) {
    const bool swap = ((cy.sx0 ^ cy.sx1) & 1) && (cy.sx0 & 1);      // different parities and tap 0 is the odd row
    const unsigned b0 = (unsigned)(unsigned short)cy.a0 << 12, b1 = (unsigned)(unsigned short)cy.a1 << 12;
    return swap ? RowRec{b1, b0, cy.sx1, cy.sx0} : RowRec{b0, b1, cy.sx0, cy.sx1};
}
struct ColOwn { short dw0, dw1, r0, r1; };
// How the deriving role's TD threads are dealt over level l's rectangle (thread = row block * quads + quad), worked out by the host: in the kernel
// it was two integer divisions per thread and level (~45 vector instructions, a tenth of k_pyr_cols).  recip = ceil(2^20 / quads): tid / quads =
// tid * recip >> 20 (exact while tid * quads < 2^20); nb = row blocks, per = rows per block.  deal[0] is for TD = 256, deal[1] for TD = 512.
struct ChainDeal { unsigned recip; short nb, per; };
inline ChainDeal makeChainDeal(int w, int h, int TD) {
    const int nq = (w + 3) >> 2;
    if (nq <= 0 || h <= 0) return ChainDeal{0u, 0, 0};
    const int nb = TD / nq;
    return ChainDeal{(unsigned)(((1u << 20) + nq - 1) / nq), (short)nb, (short)(nb > 0 ? (h + nb - 1) / nb : 0)};
}
struct PyrColumn { ChainRegion region[kMaxLevels]; ColOwn own[kMaxLevels]; ChainDeal deal[2][kMaxLevels]; int nCoef, pad; };   // nCoef: coefficient records of all its steps
// (the records - per level the quad records of its rectangle's column quads (QuadRec), then its row records (RowRec) - are laid out per
// region by the host in 8-byte units: the kernel copies region t's list from coef[t * slot ..], one coalesced pass)
struct ColLevels {      // what the kernel needs of the level tables, by value (kernel argument: scalar loads)
    int nlevels, pad;
    int w[kMaxLevels], h[kMaxLevels], pyrStride[kMaxLevels], rxOff[kMaxLevels], ryOff[kMaxLevels];
    long long pyrOff[kMaxLevels], pyrFrameBytes[kMaxLevels];
};

#ifdef __HIPCC__
// The dynamically sized LDS block of a kernel.  (tools/octree_emu compiles k_octree.hip for the HOST to run it under sanitizers; its
// shim defines this as a pointer to an exactly sized heap block, and ORBX_OCT_EMU_PAD > 0 puts poisoned red zones between the
// quad-tree's LDS sub-arrays.  In the product build the pad is 0 and the macro is the HIP declaration.)
#ifndef ORBX_DYNAMIC_LDS
#define ORBX_DYNAMIC_LDS(name) extern __shared__ __align__(16) uint8_t name[]
#endif
#ifndef ORBX_OCT_EMU_PAD
#define ORBX_OCT_EMU_PAD 0
#define ORBX_OCT_REDZONE(ptr) do {} while (0)
#endif
constexpr size_t kOctPad = ORBX_OCT_EMU_PAD;

// LDS operations of one wave execute in issue order, so the lanes of a wave only need the COMPILER to keep the order of the stores before and
// the loads after this point (no s_barrier).  (The host emulation of k_octree runs lanes as threads: its shim defines this as a wave barrier.)
#ifndef ORBX_WAVE_LDS_SYNC
#define ORBX_WAVE_LDS_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif

// Inclusive prefix sum over the 64 lanes of a wave, every lane active: row_shr 1/2/4/8 inside the rows of 16, then lane 15 of
// rows 0 and 2 broadcast into rows 1 and 3, lane 31 into rows 2 and 3 (DPP: six vector adds, no LDS round trip — a __shfl_up
// ladder is six dependent ds_bpermute).  A lane with no source takes 0 (`old` of v_mov_dpp with bound_ctrl off).
__device__ __forceinline__ int waveInclusiveScan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}
#endif

#ifdef __HIPCC__
// XCD-aware launch shape for "chunks x frames" grids: grid = (8, chunks, ceil(frames / 8)).
// Workgroups are dealt round-robin to the 8 XCDs by linear id (MI355X_MICROARCH.md, Workgroup dispatch): ids b and b + 8
// share an XCD and its private 4-MiB L2.  With chunks fastest, the chunks of one frame are sprayed over all eight L2s, so
// pixels that neighbouring chunks share (FAST ROI overlap, blur halo rows, overlapping keypoint patches, resize
// footprints) are fetched from HBM once per XCD.  Here blockIdx.x (fastest) is the frame inside a group of eight, so every
// chunk of a frame lands on the same XCD and the shared lines hit its L2.  No division in the kernel; the workgroups of
// the last group that name a frame >= nFrames exit at once.  A speed choice only: results never depend on placement.
static inline dim3 xcdGrid(int chunks, int nFrames) { return nFrames < 8 ? dim3(nFrames, chunks, 1) : dim3(8, chunks, (nFrames + 7) / 8); }
__device__ __forceinline__ bool xcdChunkFrame(int nFrames, int& chunk, int& frame) {
    chunk = blockIdx.y;
    frame = gridDim.x * blockIdx.z + blockIdx.x;    // fewer than 8 frames: plain (frame, chunk) grid, chunks spread over all XCDs
    return frame < nFrames;
}
#endif


}  // namespace orbx
