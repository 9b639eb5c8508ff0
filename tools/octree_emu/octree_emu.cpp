// octree_emu.cpp — runs the quad-tree kernel SOURCE (extractorb_amd/csrc/k_octree.hip + k_octree_body.inc, unchanged) on the HOST,
// one workgroup as T real threads, so that sanitizers can see it.  TEST INFRASTRUCTURE: built by tools/octree_emu/Makefile, driven by
// tests/test_octree_emulation.py and tools/octree_emu/soak.py; never part of liborbx.so.
//
// What this catches that a GPU run cannot show:
//   -fsanitize=address,undefined : an LDS or global index out of its array (every array is an exactly sized heap block; the LDS
//                                  sub-arrays are separated by poisoned red zones, ORBX_OCT_EMU_PAD), signed overflow, bad shifts
//   -fsanitize=thread            : two threads touching one location, one writing, with no barrier in between — in ANY schedule
//   poison byte (header field)   : the LDS block and every scratch array start filled with it; two runs with different bytes that
//                                  disagree = a read of memory nobody wrote
//   --trace                      : the node list (box, count) after the kernel, per level, for diffing against the oracle's list
//
// usage: octree_emu <case.bin> <out.bin>       (formats: tests/test_octree_emulation.py)
#include <functional>
#define EMU_HOST 1

#define ORBX_DYNAMIC_LDS(name) uint8_t* const name = emu::g_block->dynShared
#ifndef ORBX_OCT_EMU_PAD
#define ORBX_OCT_EMU_PAD 64
#endif
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#define ORBX_OCT_REDZONE(ptr) ASAN_POISON_MEMORY_REGION((const void*)(ptr), ORBX_OCT_EMU_PAD)
#else
#define ORBX_OCT_REDZONE(ptr) do {} while (0)
#endif

#include <hip/hip_runtime.h>      // tools/octree_emu/shim

namespace emu {
Block* g_block = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx;
thread_local int t_parity = 0;
dim3 g_gridDim, g_blockDim;
int g_poison = 0xA5;

static void runBlock(const std::function<void()>& body, dim3 b, int T, size_t shmem) {
    Block blk;
    blk.nThreads = T;
    pthread_barrier_init(&blk.bar, nullptr, (unsigned)T);
    blk.waves.resize((size_t)(T + kWave - 1) / kWave);
    for (auto& w : blk.waves) {
        pthread_barrier_init(&w.bar, nullptr, kWave);
        memset(w.slot, 0, sizeof(w.slot));
    }
    blk.dynBytes = shmem;
    blk.dynShared = (uint8_t*)aligned_alloc(16, (shmem + 15) & ~(size_t)15);
    memset(blk.dynShared, g_poison, shmem);
    g_block = &blk;
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            t_threadIdx = dim3((unsigned)t, 0, 0);
            t_blockIdx = b;
            t_parity = 0;
            body();
        });
    for (auto& x : th) x.join();
    g_block = nullptr;
#if defined(__SANITIZE_ADDRESS__)
    ASAN_UNPOISON_MEMORY_REGION(blk.dynShared, (shmem + 15) & ~(size_t)15);
#endif
    free(blk.dynShared);
    for (auto& w : blk.waves) pthread_barrier_destroy(&w.bar);
    pthread_barrier_destroy(&blk.bar);
}

template <class K, class... A>
void launch(K kern, dim3 grid, dim3 block, size_t shmem, A... args) {
    if (block.x % kWave) { fprintf(stderr, "emu: workgroup size %u is not a multiple of 64\n", block.x); abort(); }
    g_gridDim = grid;
    g_blockDim = block;
    for (unsigned y = 0; y < grid.y; y++)
        for (unsigned x = 0; x < grid.x; x++) runBlock([&]() { kern(args...); }, dim3(x, y, 0), (int)block.x, shmem);
}
}  // namespace emu

#ifndef EMU_SRC
#define EMU_SRC ../../extractorb_amd/csrc      // (another directory: an older revision of the kernel under investigation)
#endif
#define EMU_STR2(x) #x
#define EMU_STR(x) EMU_STR2(x)
#include EMU_STR(EMU_SRC/k_octree.hip)
#include EMU_STR(EMU_SRC/orbx_geometry.hpp)

using namespace orbx;

namespace {
template <class T>
struct Exact {     // exactly sized heap array, pre-filled with the poison byte: one element past the end is an ASan report
    T* p;
    size_t n;
    explicit Exact(size_t count) : p((T*)malloc(std::max<size_t>(count, 1) * sizeof(T))), n(count) { memset(p, emu::g_poison, std::max<size_t>(count, 1) * sizeof(T)); }
    ~Exact() { free(p); }
};
struct CaseHeader {
    int magic, nfeatures, nlevels, rows, cols, maxRows, maxCols, threads, roomy, lap0, lap1, poison;
    float scaleFactor;
    int reserved[3];
};
static_assert(sizeof(CaseHeader) == 64, "case header");
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s case.bin out.bin\n", argv[0]); return 2; }
    FILE* fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 2; }
    CaseHeader H;
    if (fread(&H, sizeof(H), 1, fi) != 1 || H.magic != 0x4f435445) { fprintf(stderr, "bad case file\n"); return 2; }
    emu::g_poison = H.poison & 255;
    // geometry and sizing exactly as orbx_create / installGeometry compute them
    const ScaleTables tabs = makeScaleTables(H.nfeatures, H.scaleFactor, H.nlevels);
    FrameGeom mg, g;
    std::string why = makeFrameGeom(tabs, H.maxRows, H.maxCols, mg);
    if (why.empty()) why = makeFrameGeom(tabs, H.rows, H.cols, g);
    if (!why.empty()) { fprintf(stderr, "geometry rejected: %s\n", why.c_str()); return 3; }
    layoutArenas(mg, 1);
    layoutArenas(g, 1);
    const OctSizing z = octreeSizing(mg, H.nlevels, octreeLdsBytes);
    if (z.err || g.maxNodes > z.M) { fprintf(stderr, "sizing rejected\n"); return 3; }

    const int nCells = (int)g.cells.size();
    Exact<LevelGeom> lv((size_t)H.nlevels);
    memcpy(lv.p, g.lv, sizeof(LevelGeom) * (size_t)H.nlevels);
    Exact<CellDesc> cells((size_t)nCells);
    memcpy(cells.p, g.cells.data(), sizeof(CellDesc) * (size_t)nCells);
    Exact<unsigned> candSeg((size_t)g.candPerFrame), cellCount((size_t)nCells), candPos((size_t)g.candPerFrame), candCount((size_t)H.nlevels);
    Exact<int> cellOff((size_t)nCells), levelCount((size_t)H.nlevels), levelLap((size_t)H.nlevels), lapArea(2);
    Exact<unsigned short> nodeOf((size_t)g.candPerFrame);
    Exact<uint2> sel((size_t)g.selPerFrame);
    lapArea.p[0] = H.lap0; lapArea.p[1] = H.lap1;
    // candidates per level in the reference's order (cell row, cell column, raster inside the cell) -> k_fast's per-cell segments
    for (int l = 0; l < H.nlevels; l++) {
        int n = 0;
        if (fread(&n, 4, 1, fi) != 1) { fprintf(stderr, "short case file\n"); return 2; }
        std::vector<unsigned> w((size_t)n);
        if (n && fread(w.data(), 4, (size_t)n, fi) != (size_t)n) { fprintf(stderr, "short case file\n"); return 2; }
        const LevelGeom& L = g.lv[l];
        int k = 0;
        for (int c = 0; c < L.cellCount; c++) {
            const CellDesc& cd = g.cells[(size_t)(L.cellFirst + c)];
            const int segCap = ((cd.roiW - 6 + 1) / 2) * ((cd.roiH - 6 + 1) / 2);
            int cnt = 0;
            while (k < n) {
                const int x = (int)(w[(size_t)k] & 0xfff), y = (int)((w[(size_t)k] >> 12) & 0xfff);
                const bool in = x >= cd.shiftX + 3 && x < cd.shiftX + cd.roiW - 3 && y >= cd.shiftY + 3 && y < cd.shiftY + cd.roiH - 3;
                if (!in) break;
                if (cnt >= segCap) { fprintf(stderr, "level %d cell %d: more candidates than its segment holds\n", l, c); return 3; }
                candSeg.p[(size_t)L.candOff + (size_t)cd.segOff + (size_t)cnt] = w[(size_t)k];
                cnt++; k++;
            }
            cellCount.p[(size_t)(L.cellFirst + c)] = (unsigned)cnt;
        }
        if (k != n) { fprintf(stderr, "level %d: candidate %d of %d is not in cell order\n", l, k, n); return 3; }
    }
    fclose(fi);

    // reserved[0] != 0: the small-batch form — the leaf counters / best keys k_fast's emit would have left in L2 (orbx_device.hpp:
    // LeafTables), built here from the same segments with the host-side code tables of installGeometry
    LeafTables lt{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
    Exact<int> leafHist(H.reserved[0] ? (size_t)H.nlevels * (size_t)z.R * kOctLeaves : 0);
    Exact<unsigned> leafBest(H.reserved[0] ? (size_t)H.nlevels * (size_t)z.R * kOctLeaves : 0);
    Exact<uint8_t> leafCode(H.reserved[0] ? (size_t)2 * H.nlevels * (size_t)z.XT : 0);
    if (H.reserved[0] && z.R > 0 && !z.arena) {
        memset(leafHist.p, 0, leafHist.n * sizeof(int));
        memset(leafBest.p, 0, leafBest.n * sizeof(unsigned));
        for (int l = 0; l < H.nlevels; l++) {
            LevelGeom& L = lv.p[l];
            L.leafOK = L.nIni <= z.R && L.rectW <= z.XT && L.rectH <= z.XT ? 1 : 0;
            if (!L.leafOK) continue;
            uint8_t *xc = leafCode.p + (size_t)l * z.XT, *yc = leafCode.p + (size_t)(H.nlevels + l) * z.XT;
            for (int x = 0; x < L.rectW; x++) xc[x] = (uint8_t)octXCode(x, L.hX, L.nIni);
            for (int y = 0; y < L.rectH; y++) yc[y] = (uint8_t)octAxisPath(y, 0, L.rectH);
            for (int c = 0; c < L.cellCount; c++) {
                const CellDesc& cd = g.cells[(size_t)(L.cellFirst + c)];
                for (unsigned k = 0; k < cellCount.p[(size_t)(L.cellFirst + c)]; k++) {
                    const unsigned slot = (unsigned)cd.segOff + k, w = candSeg.p[(size_t)L.candOff + slot];
                    const unsigned x = w & 0xfff, y = (w >> 12) & 0xfff;
                    const unsigned cx = xc[std::min((int)x, L.rectW - 1)], cy = yc[std::min((int)y, L.rectH - 1)];
                    const size_t at = ((size_t)l * z.R + (cx >> kOctDepth)) * kOctLeaves + ((cy << kOctDepth) | (cx & ((1u << kOctDepth) - 1u)));
                    leafHist.p[at]++;
                    leafBest.p[at] = std::max(leafBest.p[at], (w & 0xff000000u) | (0xffffffu - slot));
                }
            }
        }
        lt = LeafTables{leafHist.p, leafBest.p, leafCode.p, leafCode.p + (size_t)H.nlevels * z.XT, z.R, z.XT, H.nlevels, 1};
    }
    int threadsOfLevel[kMaxLevels];
    for (int l = 0; l < H.nlevels; l++) threadsOfLevel[l] = H.threads;
    Exact<uint8_t> arena(z.arena ? z.arenaSlice * (size_t)H.nlevels : 0);
    launchOctree(nullptr, lv.p, H.nlevels, cells.p, nCells, candSeg.p, cellCount.p, cellOff.p, candPos.p, candCount.p, nodeOf.p, sel.p,
                 g.selPerFrame, levelCount.p, levelLap.p, lapArea.p, z.M, z.P, z.R, z.XT, threadsOfLevel, H.roomy != 0, 0, 1,
                 z.arena ? arena.p : nullptr, lt, false);
    if (lt.hist)      // the kernel clears what it loads: the tables are zero again for the next call
        for (size_t i = 0; i < leafHist.n; i++)
            if (leafHist.p[i] != 0 || leafBest.p[i] != 0) { fprintf(stderr, "leaf table entry %zu not cleared\n", i); return 4; }

#ifdef ORBX_OCT_TRACE
    if (argc > 3) {      // (investigation builds only: the per-pass node lists of one level, written by a patched kernel copy)
        FILE* ft = fopen(argv[3], "wb");
        fwrite(g_octTrace, sizeof(unsigned), (size_t)1 << 18, ft);
        fclose(ft);
    }
#endif
    if (argc > 4) {      // investigation aid: the kernel's global work arrays after the run (per level: capacity, compacted keys, node ids)
        FILE* fa = fopen(argv[4], "wb");
        for (int l = 0; l < H.nlevels; l++) {
            const int cap = g.lv[l].candCap;
            fwrite(&cap, 4, 1, fa);
            fwrite(candPos.p + g.lv[l].candOff, 4, (size_t)cap, fa);
            fwrite(nodeOf.p + g.lv[l].candOff, 2, (size_t)cap, fa);
        }
        fclose(fa);
    }
    FILE* fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 2; }
    for (int l = 0; l < H.nlevels; l++) {
        const int n = levelCount.p[l];
        fwrite(&n, 4, 1, fo);
        fwrite(&levelLap.p[l], 4, 1, fo);
        if (n < 0 || n > g.lv[l].selCap) { fprintf(stderr, "level %d: %d kept keypoints, capacity %d\n", l, n, g.lv[l].selCap); fclose(fo); return 4; }
        fwrite(sel.p + g.lv[l].selOff, sizeof(uint2), (size_t)n, fo);
    }
    fclose(fo);
    return 0;
}
