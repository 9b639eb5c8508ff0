// orbx_api.cpp — the C ABI of liborbx.so (include/orbx.h): handle, device arenas, launch sequence.
//
// This is the product path.  It has no CPU fallback: without a HIP device orbx_create fails with
// ORBX_ERR_NO_DEVICE, and nothing here includes, links or calls anything under oracle/.
#include "orbx_internal.hpp"

static_assert(sizeof(orbx_keypoint) == sizeof(Keypoint), "orbx_keypoint layout");
static_assert(sizeof(orbx_proj_query) == sizeof(ProjQuery), "orbx_proj_query layout");

double g_hostT[8];
long g_hostN;
const bool g_hostTiming = getenv("ORBX_HOST_TIMING") && atoi(getenv("ORBX_HOST_TIMING")) != 0;      // tools/host_call_anatomy.py
TestAids g_aids;
const char* const kSlotNames[ORBX_NUM_KERNELS] = {"k_pyr_first", "k_resize", "k_blur", "k_fast",
                                            "k_octree", "k_describe", "memset+copies", "batch_total",
                                            "k_stereo_rows+match+filter", "k_frame_finish+k_search_init+k_search_proj"};
thread_local std::string g_createError;

namespace {
// The automatic split level of the per-keypoint blur (installGeometry): k_blur takes the levels whose patches hold more than kSplitRatioNum /
// kSplitRatioDen of the level's pixels.  Measured (round 6, A/B in one call, us per step): 512 x 640x480 x 1000 - no split 1696-1699, split at level
// 2 (patches 1.36 of the level) 1667-1668, at 3 (1.63) 1656-1666, at 4 (1.96) 1664-1674, at 5 1674-1680, k_blur for every level 1707-1710; 128 x
// 1080p x 2000, where the coarsest level's patches reach 0.99 of it: no split 2290, at 7 2316-2321, at 6 2318-2331, at 4 2355-2367 - so the
// levels past 1.5.
constexpr int kSplitRatioNum = 3, kSplitRatioDen = 2;

void freeAll(orbx_handle* h) {
    void* dev[] = {h->d_input, h->d_pyr, h->d_blur, h->d_candPos, h->d_candSeg, h->d_cellCount, h->d_cellOff, h->d_nodeOf, h->d_candCount, h->d_sink, h->d_sel, h->d_levelCount,
                   h->d_levelLap, h->d_lap, h->d_lv, h->d_cells, h->d_rx, h->d_ry, h->d_xq, h->d_cols, h->d_colLevels, h->d_colCoef, h->d_foot, h->d_tiles, h->d_laneItem, h->d_out,
                   h->d_octArena, h->d_leafHist, h->d_leafBest, h->d_leafCode, h->d_bowWord, h->d_bowNode, h->d_bowWeight, h->d_rowOff, h->d_sadDist,
                   h->d_nMatched, h->d_rowList, h->d_uRight, h->d_depth, h->d_clock};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->aux2) (void)hipStreamDestroy(h->aux2);
    if (h->probeStream) { (void)hipStreamSynchronize(h->probeStream); (void)hipStreamDestroy(h->probeStream); }
    for (int i = 0; i < 2; i++) { if (h->evPyr[i]) (void)hipEventDestroy(h->evPyr[i]); if (h->evBlur[i]) (void)hipEventDestroy(h->evBlur[i]); if (h->evUp[i]) (void)hipEventDestroy(h->evUp[i]); }
    void* host[] = {h->h_lap, h->h_out, h->h_in, h->h_pyr, h->h_tab};
    for (void* p : host) if (p) (void)hipHostFree(p);
    for (auto& ev : h->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    if (h->ownStream && h->stream) (void)hipStreamDestroy(h->stream);
}

// Installs the geometry of a rows x cols image: builds its tables, checks that they fit the arenas, and uploads them.
//
// Ordering (docs/history/DESIGN_rounds_1-5.md §4j): every table is packed into ONE pinned staging block and copied to its device arena with hipMemcpyAsync ON THE
// HANDLE'S STREAM, the stream every kernel of the handle is launched on (the internal side streams only ever start behind an event recorded on
// it).  So "the tables have landed before the first kernel reads them" is stream order - stated, not assumed - and no call in this path waits
// for the device as a whole: another handle's work on the same device is not stalled by this handle meeting a new image size.  The staging
// block is reused by the next geometry change, which starts by waiting for this stream.
int installGeometry(orbx_handle* h, int rows, int cols) {
    FrameGeom g;
    std::string why = makeFrameGeom(h->tabs, rows, cols, g, h->colPx);
    if (!why.empty()) return fail(h, why.find("small") != std::string::npos ? ORBX_ERR_IMAGE_TOO_SMALL : ORBX_ERR_UNSUPPORTED, why);
    layoutArenas(g, h->maxB);
    const LevelGeom& last = g.lv[g.nlevels - 1];
    if ((size_t)(last.pyrOff + last.pyrFrameBytes * h->maxB) > h->pyrBytes ||
        (size_t)(last.blurOff + last.blurFrameBytes * h->maxB) > h->blurBytes ||
        (size_t)(last.candOff + (long long)last.candCap * h->maxB) > h->candEntries ||
        (size_t)g.selPerFrame * h->maxB > h->selEntries || g.cells.size() > h->cellCap || g.maxNodes > h->octM || (g.big && !h->maxGeom.big))
        return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "image geometry does not fit the arenas sized at orbx_create");
    // ---- pass 1 (host only): pack every table, note where it goes.  A table that does not fit returns before anything is uploaded. ----
    struct Upload { void* dst; size_t at, bytes; };
    std::vector<uint8_t> image;
    std::vector<Upload> ups;
    auto stage = [&](void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        const size_t at = (image.size() + 15) & ~(size_t)15;
        image.resize(at + bytes);
        std::memcpy(image.data() + at, src, bytes);
        ups.push_back(Upload{dst, at, bytes});
    };
    size_t rxOff[kMaxLevels] = {}, ryOff[kMaxLevels] = {}, xqOff[kMaxLevels] = {}, footOff[kMaxLevels] = {}, colsOff[4] = {}, colCoefOff[4] = {};
    size_t xo = 0, yo = 0;
    for (int l = 1; l < g.nlevels; l++) {
        rxOff[l] = xo; ryOff[l] = yo;
        g.lv[l].rxOff = (int)xo; g.lv[l].ryOff = (int)yo;
        if (xo + g.rx[l].size() > h->rxCap || yo + g.ry[l].size() > h->rxCap)
            return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "resize tables do not fit");
        stage(h->d_rx + xo, g.rx[l].data(), sizeof(ResizeX) * g.rx[l].size());
        stage(h->d_ry + yo, g.ry[l].data(), sizeof(ResizeX) * g.ry[l].size());
        xo += g.rx[l].size(); yo += g.ry[l].size();
    }
    {
        size_t qo = 0;
        for (int l = 1; l < g.nlevels; l++) {
            xqOff[l] = qo;
            if (g.xq[l].empty()) continue;      // (taps not packed: the byte-gather form reads the plain tables)
            if (qo + g.xq[l].size() > h->xqCap) return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "resize column records do not fit");
            stage(h->d_xq + qo, g.xq[l].data(), sizeof(QuadRec) * g.xq[l].size());
            qo += g.xq[l].size();
        }
    }
    if (h->leafFrames) {      // x / y path codes of every level the quad-tree's dense phase covers (k_octree_body.inc: `dense`)
        std::vector<uint8_t> code((size_t)2 * g.nlevels * h->octXT, 0);
        for (int l = 0; l < g.nlevels; l++) {
            LevelGeom& L = g.lv[l];
            L.leafOK = L.nIni <= h->octR && L.rectW <= h->octXT && L.rectH <= h->octXT ? 1 : 0;
            if (!L.leafOK) continue;
            for (int x = 0; x < L.rectW; x++) code[(size_t)l * h->octXT + x] = (uint8_t)octXCode(x, L.hX, L.nIni);
            for (int y = 0; y < L.rectH; y++) code[(size_t)(g.nlevels + l) * h->octXT + y] = (uint8_t)octAxisPath(y, 0, L.rectH);
        }
        stage(h->d_leafCode, code.data(), code.size());
    }
    stage(h->d_lv, g.lv, sizeof(LevelGeom) * g.nlevels);
    stage(h->d_cells, g.cells.data(), sizeof(CellDesc) * g.cells.size());
    size_t fo = 0;
    for (int l = 1; l < g.nlevels; l++) {
        footOff[l] = fo;
        if (fo + g.foot[l].size() > h->footCap) return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "tile footprint table does not fit");
        stage(h->d_foot + fo, g.foot[l].data(), sizeof(TileFoot) * g.foot[l].size());
        fo += g.foot[l].size();
    }
    {
        size_t n = 0, nc = 0;
        for (FrameGeom::ColumnSet& cs : g.colSets) {
            if (n + cs.columns.size() > h->colsCap || nc + cs.coef.size() > h->colCoefCap) cs.fit = false;      // (a region size far below the default ones: the other forms serve)
            if (!cs.fit) continue;
            stage(h->d_cols + n, cs.columns.data(), sizeof(PyrColumn) * cs.columns.size());
            stage(h->d_colCoef + nc, cs.coef.data(), sizeof(ResizeX) * cs.coef.size());
            const size_t slot = &cs - g.colSets.data();      // (at most four cuts per geometry: kColPx)
            colsOff[slot] = n;
            colCoefOff[slot] = nc;
            n += cs.columns.size();
            nc += cs.coef.size();
        }
        ColLevels c{};
        c.nlevels = g.nlevels;
        for (int l = 0; l < g.nlevels; l++) {
            c.w[l] = g.lv[l].w; c.h[l] = g.lv[l].h; c.pyrStride[l] = g.lv[l].pyrStride; c.rxOff[l] = g.lv[l].rxOff; c.ryOff[l] = g.lv[l].ryOff;
            c.pyrOff[l] = g.lv[l].pyrOff; c.pyrFrameBytes[l] = g.lv[l].pyrFrameBytes;
        }
        stage(h->d_colLevels, &c, sizeof(c));
    }
    // The blur split by level (round 6): where the blur is done per keypoint (k_describe<PB>), the patches of a coarse level hold more pixels than the
    // level itself - quota x 37 x 36 (the disc of the 43 x 37 horizontal pass + the 37 x 37 vertical one) against w x h; 640x480 x 1000: 0.94 of the
    // level's pixels at level 0, 1.63 at level 3, 3.4 at level 7 - so from the first level whose patches exceed kSplitRatio x its pixels on, k_blur
    // blurs the level (table [2] below) and k_describe reads the blurred level.  ORBX_BLUR_SPLIT=L pins the level, 0 turns the split off.
    int splitLevel = 0;
    if (h->blurSplit > 0) splitLevel = h->blurSplit;
    else if (h->blurSplit < 0 && kSplitRatioNum > 0) {
        for (int l = 1; l < g.nlevels && !splitLevel; l++)
            if ((long long)g.lv[l].quota * 37 * 36 * kSplitRatioDen > (long long)kSplitRatioNum * g.lv[l].w * g.lv[l].h) splitLevel = l;
        // ... if those levels hold a quarter of the features: two more launches for the coarsest level alone cost more than its patches (128 x
        // 1280x720 x 1500, where only level 7 - 6 % of the features - is past 1.5: 1146 us per step split, 1128 unsplit)
        int tail = 0;
        for (int l = splitLevel; splitLevel && l < g.nlevels; l++) tail += g.lv[l].quota;
        if (4 * tail < h->nfeatures) splitLevel = 0;
    }
    if (splitLevel >= g.nlevels) splitLevel = 0;
    int nBlurLanes[3] = {0, 0, 0};
    size_t blurItemOff[3] = {0, 0, 0}, blurLaneOff[3] = {0, 0, 0};
    {   // blur tables for both row-block sizes, and the 32-row blocks of the levels from splitLevel on
        std::vector<BlurItem> tiles;
        std::vector<unsigned short> laneItem;
        const int blockRows[3] = {kBlurBlockRows, kBlurBlockRowsSmall, kBlurBlockRows};
        for (int v = 0; v < 3; v++) {
            const size_t t0 = tiles.size(), l0 = laneItem.size();
            int lanes = 0;
            if (v == 2 && splitLevel == 0) { nBlurLanes[v] = 0; blurItemOff[v] = t0; blurLaneOff[v] = l0; continue; }
            for (int l = 0; l < g.nlevels; l++) {
                if (v == 2 && l < splitLevel) continue;               // the finest levels are blurred per keypoint
                for (int y0 = 0; y0 < g.lv[l].h; y0 += blockRows[v]) {
                    laneItem.insert(laneItem.end(), (size_t)(g.lv[l].w + 3) / 4, (unsigned short)(tiles.size() - t0));
                    tiles.push_back(BlurItem{lanes, 0, (short)l, (short)y0});
                    lanes += (g.lv[l].w + 3) / 4;
                }
            }
            tiles[t0].count = (int)(tiles.size() - t0);
            if (tiles.size() - t0 > 65535) return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "blur item table does not fit");
            nBlurLanes[v] = lanes; blurItemOff[v] = t0; blurLaneOff[v] = l0;
        }
        if (tiles.size() > h->tileCap || laneItem.size() > h->laneCap) return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "blur item table does not fit");
        stage(h->d_tiles, tiles.data(), sizeof(BlurItem) * tiles.size());
        stage(h->d_laneItem, laneItem.data(), sizeof(unsigned short) * laneItem.size());
    }
    // ---- pass 2: the uploads, in stream order ----
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // the device tables and the staging block may still be in use by queued work
    h->geom = FrameGeom();                          // a failure below must not leave half-written tables looking valid
    h->lastB = 0;
    h->lastHostB = 0;
    if (image.size() > h->hTabBytes) {
        if (h->h_tab) (void)hipHostFree(h->h_tab);
        h->h_tab = nullptr; h->hTabBytes = 0;
        const size_t want = image.size() + image.size() / 4 + 4096;
        HIP_TRY(h, hipHostMalloc(&h->h_tab, want));
        h->hTabBytes = want;
    }
    std::memcpy(h->h_tab, image.data(), image.size());
    for (const Upload& u : ups) HIP_TRY(h, hipMemcpyAsync(u.dst, h->h_tab + u.at, u.bytes, hipMemcpyHostToDevice, h->stream));
    for (int l = 0; l < kMaxLevels; l++) { h->rxOff[l] = rxOff[l]; h->ryOff[l] = ryOff[l]; h->xqOff[l] = xqOff[l]; h->footOff[l] = footOff[l]; }
    for (int i = 0; i < 4; i++) { h->colsOff[i] = colsOff[i]; h->colCoefOff[i] = colCoefOff[i]; }
    h->splitLevel = splitLevel;
    for (int v = 0; v < 3; v++) { h->nBlurLanes[v] = nBlurLanes[v]; h->blurItemOff[v] = blurItemOff[v]; h->blurLaneOff[v] = blurLaneOff[v]; }
    {
        const long long px = (long long)g.lv[0].w * g.lv[0].h;
        int T = px <= 500000 ? 256 : (px <= 1200000 ? 512 : 1024);   // measured: 640x480 -> 256, 1280x720 -> 512, 1920x1080 -> 1024 (512 equal)
        if (h->octThreadsForced == 256 || h->octThreadsForced == 512 || h->octThreadsForced == 1024) T = h->octThreadsForced;
        for (int l = 0; l < g.nlevels; l++) h->octThreads[l] = T;
    }
    h->geom = g;
    return ORBX_OK;
}

int checkFrameArgs(orbx_handle* h, int n_frames, int rows, int cols) {
    if (n_frames < 1) return fail(h, ORBX_ERR_BAD_ARGUMENT, "n_frames < 1");
    if (n_frames > h->maxB) return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "n_frames exceeds max_batch");
    if (cols > h->maxW || rows > h->maxH) return fail(h, ORBX_ERR_IMAGE_TOO_LARGE, "image exceeds max_width/max_height");
    return ORBX_OK;
}

// The launch sequence of one batch.  Everything is enqueued on h->stream; nothing synchronises.
enum { kStageFront = 1, kStageBack = 2 };      // front: pyramid (+ blur) = ComputePyramid; back: FAST, quad-tree, orientation, descriptors
int enqueueBatch(orbx_handle* h, int B, const uint8_t* d_imgs, int rows, int cols, long long stride,
                 long long frameStride, const int* lap, Keypoint* d_kps, uint8_t* d_desc, int capacity, int* d_nOut,
                 int* d_monoOut, Keypoint* d_levelK, int* d_levelCounts, int stages = kStageFront | kStageBack) {
    HIP_TRY(h, hipSetDevice(h->device));
    if (rows != h->geom.rows || cols != h->geom.cols) {
        int rc = installGeometry(h, rows, cols);
        if (rc != ORBX_OK) return rc;
    }
    const FrameGeom& g = h->geom;
    hipStream_t st = h->stream;
    if (h->leafDirty) {      // an earlier call returned between k_fast and k_octree: the leaf tables still hold its counts
        const size_t n = (size_t)h->leafFrames * h->nlevels * h->octR * kOctLeaves;
        HIP_TRY(h, hipMemsetAsync(h->d_leafHist, 0, n * sizeof(int), st));
        HIP_TRY(h, hipMemsetAsync(h->d_leafBest, 0, n * sizeof(unsigned), st));
        h->leafDirty = false;
    }
    Prof total(h, S_TOTAL);
    // lapping areas: upload only when they change (they are per-camera constants in the reference)
    {
        std::vector<int> want(2 * B);
        for (int f = 0; f < B; f++) {
            want[2 * f] = lap ? lap[2 * f] : 0;
            want[2 * f + 1] = lap ? lap[2 * f + 1] : 1000;   // Frame.cc:307 passes {0,1000}
        }
        bool same = h->lapCached.size() >= want.size();
        for (size_t i = 0; same && i < want.size(); i++) same = h->lapCached[i] == want[i];
        if (!same) {
            HIP_TRY(h, hipStreamSynchronize(st));   // previous async copy out of h_lap is done
            std::memcpy(h->h_lap, want.data(), want.size() * sizeof(int));
            HIP_TRY(h, hipMemcpyAsync(h->d_lap, h->h_lap, want.size() * sizeof(int), hipMemcpyHostToDevice, st));
            h->lapCached = want;
        }
    }
    // the launch sequence of frames [f0, f0 + Bn) on stream st, in two parts: front = pyramid + blur (HBM / latency bound),
    // back = FAST (vector-issue bound) + quad-tree (barrier-latency bound) + description
    auto blurVariant = [&](int Bn) { return (long long)h->nBlurLanes[0] * Bn >= 64LL * 2 * 4 * h->numCUs ? 0 : 1; };   // two waves per SIMD of 32-row lanes
    auto blurRidesWithFast = [&](int Bn) { return blurVariant(Bn) == 1 && !h->profiling && h->fuseSmall && fastCanCarryBlur(g.maxRoiW, g.maxRoiH) && !g.big; };
    // Patch blur (k_describe<PB>): no blurred levels at all - every keypoint's 37 x 37 patch is blurred out of its raw 43 x 43 tile.  It pays where
    // the keypoints' patches are not a much larger job than the whole pyramid (nfeatures x 43 x 37 pixels of horizontal pass against the pyramid's
    // pixels) and the blur would be a launch of its own (not the small-batch form that rides in the FAST launch) - k_describe is bound by its sparse
    // patch reads at large frames, and the one raw tile is fewer bytes than a raw + a blurred one.  Measured (us per step, k_blur beside FAST ->
    // patch blur): 128 x 1920x1080 x 2000 (ratio 0.50) 3041-3052 -> 2717-2723; 128 x 1280x720 x 1500 (0.84) 1448-1455 -> 1358-1364; 512 x 640x480 x
    // 1000 (1.67) 1929-1937 -> 1993-2000: taken up to a ratio of 1.25.  ORBX_PATCH_BLUR=1 / 0 forces / forbids it.
    // Round 5: ... and up to a ratio of 1.75 once the batch is large (>= 240 Mpx of pyramid: 256 frames of 640x480 x 1000).  Since round 4's
    // instruction work the two forms are level there (512 x 640x480 x 1000: 1725-1729 us with k_blur beside FAST, 1735 with the patch blur, A/B in
    // one gpurun call, -0.4 ... -0.6 %; 256 / 1024 frames the same; 64 frames -1.5 %, 1200 features per frame -3.5 %: not taken there), and the
    // patch blur moves 4.75 MB per frame through HBM instead of 8.0 (no blurred level is written or re-read) and leaves the
    // blurred arena (1 MB per frame) untouched: where the clock says "equal", the bytes decide.
    // A back-only pass (orbx_compute_keypoints_octree) describes from what the front part of the EARLIER call left: if that call blurred per
    // keypoint (form 3) no blurred level exists and this pass must do the same, whatever its own frame count would choose; otherwise the blurred
    // levels exist (or are owed to the FAST launch: blurOwed) and are used.
    // (round 6: "big" from ORBX_SPLIT_MIN_MPX = 120 Mpx of pyramid per call, the same threshold as the per-keypoint blur below - rounds 4-5: twice
    // that -: 128 x 640x480 with the coarse levels' blur aside and staggered tails 457 -> 450 us)
    const bool bigBatch = (long long)g.sumPixels * B >= h->splitMinPixels;
    const long long patchPx4 = (long long)h->nfeatures * 43 * 37 * 4;      // 4 x the patches' pixels of horizontal pass
    // Round 6: with the coarse levels split off to k_blur (installGeometry: splitLevel) the per-keypoint form also wins earlier and further - from
    // 120 Mpx of pyramid per call (128 x 640x480: 468 -> 459 us; 64 frames: 232 -> 248, not taken) and up to a ratio of 2.25 (512 x 640x480 x 1200
    // features, the stereo stream, 2.0: 1758 -> 1743 us).
    const bool midBatch = (long long)g.sumPixels * B >= h->splitMinPixels;
    const bool patchBlur = !(stages & kStageFront) ? (h->lastBlurForm == 3 || h->lastBlurForm == 5)
                         : (h->patchBlur > 0 || (h->patchBlur < 0 && !blurRidesWithFast(B) && (patchPx4 <= 5LL * g.sumPixels || (midBatch && patchPx4 <= 9LL * g.sumPixels))));
    // ... split by level: levels >= splitL are blurred by k_blur and described from the blurred level (installGeometry: splitLevel)
    const int splitL = !patchBlur ? 0 : (!(stages & kStageFront) ? (h->lastBlurForm == 5 ? h->lastSplitLevel : 0)
                                                                   : (h->splitLevel > 0 && h->splitLevel < g.nlevels && h->nBlurLanes[2] > 0 ? h->splitLevel : 0));
    auto pollute = [&](hipStream_t st) { if (h->ldsPollute >= 0) launchLdsPollute(st, h->numCUs, h->ldsPollute, h->d_sink); };
    // Overlap inside one call (round 4): the blurred levels are read by k_description only, the LAST launch, so the blur of a large batch runs on
    // a side stream beside FAST and the quad-tree: pyramid -> {blur | FAST -> quad-tree} -> description.  k_blur is the one HBM-bound kernel of the
    // path (0.53 of the issue rate), k_fast sits at the issue ceiling with idle memory pipes: 512 x 640x480 2016-2040 us unsplit, 1981-1988 as two
    // halves side by side (round 2's form), 1915-1933 with the blur aside (two halves AND the blur aside: 1932-1937); 256 frames 1035-1055 / 1001-1014 /
    // 999-1007; 128 x 1080p 3094-3136 (halves) -> 3021-3029; 128 frames of 640x480 and fewer: no difference.  A low-priority side stream only starts
    // the blur when everything else is done (2022-2030).  ORBX_SPLIT=0: no overlap of any kind (profiling runs: one kernel at a time).  (Round 2's two
    // half-batches side by side - removed in round 4 - are the "halves" figures above; profiles/r02_split_sweep.md.)
    const bool blurSide = h->splitMode != 0 && bigBatch && !h->profiling && (stages & kStageFront) && (stages & kStageBack);
    bool blurJoin[2] = {false, false};
    int blurF0[2] = {0, 0}, blurBn[2] = {0, 0};
    auto front = [&](hipStream_t st, int f0, int Bn) {
        // The pyramid region by region: one workgroup takes a region of the image through every level (k_pyr_cols; the coarsest cut that still
        // gives the chip ~3/4 workgroup per CU, else the finest).  Frames up to half a megapixel: for every batch size (512 frames of 640x480:
        // 2016 -> 1979 us per call against one launch per level, whose tiles are poorly filled on such small levels); larger frames: while the
        // coarsest cut stays below ~12 workgroups per CU - 1280x720: 16 frames 236 -> 226 us, 32: equal, 64: 767 vs 788; 1920x1080: 16 frames
        // 431 -> 422, 64: 1530 vs 1582, 128: 3119 vs 3222.  (Round 2's per-tile chains, k_pyr_chain, lost to it at every size and were removed in
        // round 4; a geometry whose regions do not fit the kernel's staging - none of the tested ones - takes one launch per level.)
        const FrameGeom::ColumnSet* cs = nullptr;
        auto pickCut = [&](const std::vector<FrameGeom::ColumnSet>& sets) {
            const FrameGeom::ColumnSet* best = nullptr;
            for (const FrameGeom::ColumnSet& c : sets) {
                if (!c.fit || c.ldsBytes > 60 * 1024) continue;
                if (!best || (long long)c.columns.size() * Bn >= 3LL * h->numCUs / 4) best = &c;
            }
            return best;
        };
        if (h->pyrCols != 0 && g.colsPacked && !h->resizeBytewise) {     // (its steps are the packed ones: the regions carry quad records)
            cs = pickCut(g.colSets);
        }
        const bool smallFrame = (long long)g.rows * g.cols <= 512 * 1024;
        if (cs && h->pyrCols < 0 && !smallFrame && (long long)cs->columns.size() * Bn > 12LL * h->numCUs) cs = nullptr;
        h->lastPyrCut = cs ? cs->px : 0;
        h->lastPyrForm = cs ? 0 : 1;
        if (cs) {
            // workgroup shape (launchPyrCols): while every workgroup has a CU to itself, more threads shorten its levels - 1024 (512 derive, 512
            // write) for the fine cuts, 768 (256 + 512) for the coarse ones, whose levels write more than they derive; else 512 (256 + 256).
            // One frame 40.9 -> 39.8 us, two 45.8 -> 44.6, four 58.3 -> 55.7, eight 71.5 -> 67.1; from 32 frames on the small shape wins
            const int colsShape = h->colsShape > 0 ? h->colsShape : ((long long)cs->columns.size() * Bn <= h->numCUs ? (cs->px <= 56 ? 6 : 4) : 1);
            Prof p(h, S_RESIZE, st);
            h->lastKernel[S_RESIZE] = "k_pyr_cols";
            pollute(st);
            const size_t slot = cs - g.colSets.data();
            launchPyrCols(st, d_imgs, stride, frameStride, g.lv[0].w, h->d_cols + h->colsOff[slot], (int)cs->columns.size(), h->d_colLevels, g.nlevels,
                          h->d_colCoef + h->colCoefOff[slot], cs->coefSlot, h->d_pyr, cs->ldsBytes, cs->evenBytes,
                          g.colsPacked && !h->resizeBytewise, colsShape, f0, Bn);
        } else {
            {   // level 0 (bordered copy) and level 1 (resized straight from the caller's image) in one launch
                Prof p(h, S_LEVEL0, st);
                pollute(st);
                launchPyrFirst(st, d_imgs, stride, frameStride, g.lv[0], g.nlevels > 1 ? &g.lv[1] : nullptr, g.tilesX[0], g.tilesY[0],
                               g.tilesX[1], g.tilesY[1], h->d_rx + h->rxOff[1], h->d_xq + h->xqOff[1], h->d_ry + h->ryOff[1], h->d_foot + h->footOff[1], h->d_pyr,
                               g.tileLdsStride, g.tileLdsRows, g.packedTaps[1] && !h->resizeBytewise, f0, Bn);
            }
            for (int l = 2; l < g.nlevels; l++) {      // levels 2..: one launch per level (each level is resized from the rounded pixels of the one above)
                Prof p(h, S_RESIZE, st);
                h->lastKernel[S_RESIZE] = "k_resize";
                pollute(st);
                launchResize(st, g.lv[l - 1], g.lv[l], g.tilesX[l], g.tilesY[l], h->d_rx + h->rxOff[l], h->d_xq + h->xqOff[l], h->d_ry + h->ryOff[l],
                             h->d_foot + h->footOff[l], h->d_pyr, g.tileLdsStride, g.tileLdsRows, g.packedTaps[l] && !h->resizeBytewise, f0, Bn);
            }
        }
        // blur: throughput form (32-row blocks) once the grid fills the chip, else the short-chain form (8-row blocks), which in
        // unprofiled small batches rides in the FAST launch (back) instead of being a launch of its own
        h->lastBlurForm = patchBlur ? (splitL ? 5 : 3) : (blurRidesWithFast(Bn) ? 1 : 0);
        h->lastSplitLevel = splitL;
        h->blurOwed = !patchBlur && blurRidesWithFast(Bn);      // (the back part of this call - or orbx_compute_keypoints_octree later - brings the blur)
        // the blurred levels are only read by k_describe, the last launch: a big batch blurs on the internal stream, beside k_fast and the
        // quad-tree (pyramid -> {blur, FAST -> quad-tree} -> description)
        auto blurLaunch = [&](int v, int blockRows) {
            Prof p(h, S_BLUR, st);
            pollute(st);
            hipStream_t bs = st;
            const int half = f0 ? 1 : 0;
            if (blurSide) {
                (void)hipEventRecord(h->evPyr[half], st);
                (void)hipStreamWaitEvent(h->aux2, h->evPyr[half], 0);
                bs = h->aux2;
            }
            launchBlur(bs, h->d_tiles + h->blurItemOff[v], h->d_laneItem + h->blurLaneOff[v], h->nBlurLanes[v], blockRows, h->d_lv, h->d_pyr, h->d_blur, f0, Bn);
            if (blurSide) { (void)hipEventRecord(h->evBlur[half], h->aux2); blurJoin[half] = true; blurF0[half] = f0; blurBn[half] = Bn; }
        };
        if (patchBlur) {
            if (splitL) blurLaunch(2, kBlurBlockRows);      // the coarse levels; levels < splitL: k_describe blurs per keypoint, no blurred level is written
        } else if (!blurRidesWithFast(Bn)) {
            // throughput form (32-row blocks) once the grid fills the chip, else the short-chain form (8-row blocks)
            const int v = blurVariant(Bn);
            blurLaunch(v, v == 1 ? kBlurBlockRowsSmall : kBlurBlockRows);
        }
    };
    bool injected = false;
    // small batches: k_fast's emit does the quad-tree's first sweep (leaf counters and best keys in L2); k_octree loads and clears them
    auto leafTables = [&](int f0, int Bn) {
        LeafTables lt{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
        if (h->leafFrames && f0 + Bn <= h->leafFrames && !g.big)
            lt = LeafTables{h->d_leafHist, h->d_leafBest, h->d_leafCode, h->d_leafCode + (size_t)g.nlevels * h->octXT, h->octR, h->octXT, g.nlevels, h->leafFrames};
        return lt;
    };
    auto backFast = [&](hipStream_t st, int f0, int Bn) {
        const LeafTables lt = leafTables(f0, Bn);
        {
            Prof p(h, S_FAST, st);
            const bool carry = h->blurOwed && blurRidesWithFast(Bn);
            if (h->blurOwed && !carry) {      // (a front part that counted on this launch, and a back part that cannot carry the blur: event profiling switched on in between)
                launchBlur(st, h->d_tiles + h->blurItemOff[1], h->d_laneItem + h->blurLaneOff[1], h->nBlurLanes[1], kBlurBlockRowsSmall, h->d_lv, h->d_pyr, h->d_blur, f0, Bn);
            }
            h->blurOwed = false;
            pollute(st);
            if (lt.hist) h->leafDirty = true;
            if (h->testFailAfterFast && lt.hist) { injected = true; h->testFailAfterFast = false; }      // (test aid: a call that dies between k_fast and k_octree)
            const bool wide = !g.big && (h->fastWide > 0 || (h->fastWide < 0 && (long long)g.cells.size() * Bn <= 4LL * h->numCUs));
            h->lastKernel[S_FAST] = wide && g.maxRoiW <= 45 && g.maxRoiH <= 45 ? "k_fast_wide" : "k_fast";
            launchFast(st, h->d_cells, (int)g.cells.size(), h->d_lv, g.nlevels, h->d_pyr, h->iniTh, h->minTh, h->d_candSeg,
                       h->d_cellCount, g.maxRoiW, g.maxRoiH, f0, Bn, carry ? h->d_tiles + h->blurItemOff[1] : nullptr,
                       h->d_laneItem + h->blurLaneOff[1], h->nBlurLanes[1], h->d_blur, lt, wide, g.big);
        }
    };
    auto backTail = [&](hipStream_t st, int f0, int Bn) {
        if (injected) return;
        const LeafTables lt = leafTables(f0, Bn);
        {
            Prof p(h, S_OCTREE, st);
            // small batches: while every (frame, level) workgroup is resident at once, the largest workgroup that still lets
            // them all be resident finishes a level soonest (640x480, one frame: 73 us with 1024 threads, 104 us with 256)
            int octT[kMaxLevels];
            // (the "resident" variants are compiled for 4 waves per SIMD = 128 VGPRs, no scratch: 1024 threads per CU-SIMD set)
            // (... the 1024-thread one; the 512- and 256-thread resident variants are compiled for 3 waves per SIMD - no scratch -: a CU then holds
            // THREE 256-thread workgroups (one wave per SIMD each) but only ONE of 512 threads (two waves per SIMD each: a second would need four) -
            // counted as 768 threads per CU, a third of the "resident" 512-thread workgroups queued)
            const long long wgs = (long long)Bn * g.nlevels;
            // one 1024-thread workgroup per CU (the 512-thread build is reached from here through `need` below, also one per CU), else three of 256
            int residentT = wgs <= h->numCUs ? 1024 : (wgs * 256 <= 768LL * h->numCUs ? 256 : 0);
            if (residentT < h->octThreads[0] || h->octThreadsForced) residentT = 0;   // never fewer threads than the image size asks for
            if (g.big) residentT = wgs <= h->numCUs ? 1024 : 0;      // (frames beyond 4096 px: the 1024-thread builds are the ones that read two-dword candidates)
            // ... but with the first sweep done by k_fast (leaf tables) what is left of a level is a chain of barriers over a list of at most
            // quota + 3 nodes: when 256 threads still give every node of the list its own thread (the short pass forms) the small workgroup
            // has the cheapest barriers (one frame: 1024 threads 16.7 us, 256 threads 13.6)
            if (residentT && lt.hist) {
                int q = 0;
                for (int l = 0; l < g.nlevels; l++) q = g.lv[l].quota > q ? g.lv[l].quota : q;
                // (nfeatures 1200, level-0 quota 261: 512 threads 14.1 us, 1024 threads 15.1 - its last pass ranks up to 256 multi-key nodes
                // against each other, one packed key per node; before the packed key the larger workgroup won, 18.0 vs 19.3)
                int need = q + 4 <= 256 ? 256 : (q + 4 <= 512 ? 512 : residentT);
                if (need < residentT) residentT = need;
            }
            // queued launches whose first sweep k_fast has done: the same chain of barriers, and the 256-thread workgroup wins at every frame size
            // (128 x 1080p: 87 us with 1024 threads, 57 with 512, 50 with 256; 64 x 1080p: 49 / 34 / 28; 128 x 720p: 87 / 57 / 46) - the sizes by
            // image area were measured with the sweep inside the kernel, where a level-0 workgroup reads ~60 k candidates
            const int queuedT = lt.hist && !h->octThreadsForced ? 256 : 0;
            for (int l = 0; l < g.nlevels; l++) octT[l] = g.big ? 1024 : (residentT ? residentT : (queuedT ? queuedT : h->octThreads[l]));
            // ... in its scratch-free build (three workgroups per CU instead of six) while the launch is at most four workgroups per CU: little queues
            // behind the first round, and no spilled register is touched.  Round 6, A/B in one call (k_octree_256 -> _256r, us per launch; step): 128 x
            // 1080p 51 -> 33 (2291-2294 -> 2271-2278), 64 x 1080p 28 -> 20, 128 x 720p 47 -> 32 (1167-1171 -> 1151-1157), 128 x 640x480 30 -> 28.5
            // (472-473 -> 468); 256 x 640x480 equal; 512 x 640x480 112 -> 118 (1663 -> 1699: the spilling build keeps its place there, six per CU)
            const bool roomy = residentT != 0 || h->octRoomyForced || (queuedT == 256 && wgs <= 4LL * h->numCUs);
            pollute(st);
            h->lastKernel[S_OCTREE] = (h->d_octArena ? "k_octree_1024g" : "k_octree_" + std::to_string(octT[0]) + (roomy ? "r" : "")) + (g.big ? "b" : "");
            launchOctree(st, h->d_lv, g.nlevels, h->d_cells, (int)g.cells.size(), h->d_candSeg, h->d_cellCount, h->d_cellOff,
                         h->d_candPos, h->d_candCount, h->d_nodeOf, h->d_sel, g.selPerFrame,
                         h->d_levelCount, h->d_levelLap, h->d_lap, h->octM, h->octP, h->octR, h->octXT, octT, roomy, f0, Bn, h->d_octArena, lt, g.big);
            h->leafDirty = false;
        }
        auto joinBlur = [&]() {
            for (int i = 0; i < 2; i++)      // every blur still on the side stream that covers frames of this part
                if (blurJoin[i] && blurF0[i] < f0 + Bn && f0 < blurF0[i] + blurBn[i]) (void)hipStreamWaitEvent(st, h->evBlur[i], 0);
        };
        {
            Prof p(h, S_DESCRIBE, st);
            auto describe = [&](bool pb, int l0, int l1) {      // the keypoints of levels [l0, l1): their slots are [selOff(l0), selOff(l1))
                pollute(st);
                launchDescribe(st, h->d_lv, g.nlevels, h->d_pyr, h->d_blur, h->d_sel, g.selPerFrame, h->d_levelCount,
                               h->d_levelLap, d_kps, d_desc, capacity, d_nOut, d_monoOut, d_levelK, d_levelCounts, pb, l0, l1,
                               g.lv[l0].selOff, l1 < g.nlevels ? g.lv[l1].selOff : g.selPerFrame, f0, Bn);
            };
            if (splitL) {
                // (side by side on two streams - the plain launch behind the blur on ITS stream, starting together with the per-keypoint one - measured
                // slower: 1673-1677 vs 1656-1659 us per 512 frames, the stereo stream 1740 vs 1721; docs/history/r06.md)
                describe(true, 0, splitL);      // (needs no blurred level: in front of the join)
                joinBlur();
                describe(false, splitL, g.nlevels);
            } else {
                joinBlur();
                describe(patchBlur, 0, g.nlevels);
            }
        }
    };
    auto back = [&](hipStream_t st, int f0, int Bn) { backFast(st, f0, Bn); backTail(st, f0, Bn); };
    struct BlurJoin {      // a blur left on the side stream by an early return is still joined into the caller's stream
        orbx_handle* h; hipStream_t st; bool* pending;
        ~BlurJoin() { for (int i = 0; i < 2; i++) if (pending[i]) (void)hipStreamWaitEvent(st, h->evBlur[i], 0); }
    } blurGuard{h, st, blurJoin};
    const bool doFront = (stages & kStageFront) != 0, doBack = (stages & kStageBack) != 0;
    // Staggered tails (the largest batches: 1.5 x the pixels of `bigBatch` - round 5: 384 x 640x480 1315-1324 -> 1301-1308 us, 256 frames no difference;
    // rounds 3-4: twice; ORBX_SPLIT=3: every big batch): FAST in two halves back to back on the
    // caller's stream; the first half's quad-tree and description (barrier- and latency-bound: 0.4-0.6 of the issue rate) run on the internal
    // stream under the second half's FAST.  512 x 640x480: 1913-1914 -> 1888-1894 us; 256 frames and 128 x 1080p: no difference (k_fast fills the chip;
    // what runs beside it mostly adds its own issue time)
    // (never with the blur riding in the FAST launch, the form of SMALL batches, which an ORBX_SPLIT_MIN_MPX=0 test run also counts as big: the
    // whole-batch pyramid would leave the blur to the FIRST half's FAST launch only — found by the batch-shape fuzz)
    // (round 6, with the blur split by level: every big batch of small frames - 256 x 640x480 880 -> 870 us, 320 frames 1085 -> 1063-1069; the stereo
    // stream at 256 frames equal; rounds 3-5 started at 1.5-2 x the pixels of `bigBatch`)
    const bool stagger = (h->splitMode == 3 || (h->splitMode == 1 && bigBatch && (long long)g.rows * g.cols <= 512 * 1024)) &&
                         !h->profiling && B >= 2 && bigBatch && doFront && doBack && !blurRidesWithFast(B);
    if (stagger) {
        struct Join {
            orbx_handle* h; hipStream_t st; bool armed = false;
            ~Join() { if (armed) { (void)hipEventRecord(h->evJoin, h->aux); (void)hipStreamWaitEvent(st, h->evJoin, 0); } }
        } join{h, st};
        const int B0 = (B + 1) / 2;
        front(st, 0, B);
        backFast(st, 0, B0);
        HIP_TRY(h, hipEventRecord(h->evFork, st));
        HIP_TRY(h, hipStreamWaitEvent(h->aux, h->evFork, 0));
        join.armed = true;
        backTail(h->aux, 0, B0);
        backFast(st, B0, B - B0);
        backTail(st, B0, B - B0);
    } else {
        if (doFront) front(st, 0, B);
        if (doBack) back(st, 0, B);
    }
    if (injected) return fail(h, ORBX_ERR_HIP, "injected failure (test aid fail_after_fast): returned between k_fast and k_octree");
    HIP_TRY(h, hipGetLastError());
    if (doFront) h->lastB = B;   // (a back-only pass leaves the pyramid of the earlier call, all its frames, in place)
    h->lastHostB = 0;            // the caller's buffers hold this batch; orbx_extract_batch_begin sets it again for its own
    h->pyramidOnly = !doBack;
    return ORBX_OK;
}


// The frames of a host-buffer call into the tight cols x rows x B device slab d_input, enqueued on the handle's stream.  Pinned input
// (orbx_host_alloc) is copied asynchronously as it lies; a few frames of pageable input are first gathered row by row in the handle's pinned
// staging (a hipMemcpyAsync from pageable memory stages inside the runtime and waits: 640x480 ~40 us against ~3 + ~15 here), which also turns a
// strided cv::Mat into one copy; larger pageable batches are left to the runtime.
// One input-copy queue per device, shared by every handle on it (batches of two or more frames).  The host-to-device link is ONE resource: two
// handles that copy their inputs at the same time each get half of it and then compute at the same time - the ping-pong of two handles locks
// into that in-phase state (both copy, both compute: 2.6 ms per 256 frames) as easily as into the anti-phase one (one copies while the other
// computes: 1.5 ms), round 6, profiles/r06_host_path.md.  Through one queue the copies run one after the other at the full rate, so the
// handles' kernels start staggered by a copy time and the anti-phase order is the only one there is.
hipStream_t sharedUploadStream(int device) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); streams[device] = nullptr; }
    return streams[device];      // (lives as long as the process: handles come and go)
}

// Measured (two handles alternating, 640x480, frames/s; own stream + result DMA at _begin -> this policy): 8 frames per call 56 k -> 105 k, 64: 91 k ->
// 163 k, 256: 98-124 k -> 173 k, 512: 111 k -> 174 k; the bare copy runs at 55 GB/s = 180 k frames/s.  Small copies keep the handle's own stream
// (the two event hops cost more than they order: 8 frames 105 k against 97 k) and bring their results back with a copy kernel instead (k_copy.hip).
bool usesSharedUpload(const orbx_handle* h, int B, int rows, int cols) {
    const long long limit = h->sharedUploadBytes >= 0 ? h->sharedUploadBytes : 16LL << 20;
    return B >= 2 && (long long)B * rows * cols >= limit;
}

int uploadFrames(orbx_handle* h, int B, const uint8_t* imgs, int rows, int cols, ptrdiff_t stride, ptrdiff_t frame_stride) {
    hipStream_t own = h->stream;
    hipStream_t st = own;
    if (usesSharedUpload(h, B, rows, cols)) {
        if (hipStream_t up = sharedUploadStream(h->device)) {
            st = up;
            if (hipStreamQuery(own) != hipSuccess) {      // work still queued on the handle's stream may read d_input: the copy waits for it
                (void)hipGetLastError();
                HIP_TRY(h, hipEventRecord(h->evUp[0], own));
                HIP_TRY(h, hipStreamWaitEvent(st, h->evUp[0], 0));
            }
        }
    }
    struct Rejoin {      // the handle's stream continues behind the copy (on every return path)
        orbx_handle* h; hipStream_t own, st;
        ~Rejoin() { if (st != own) { (void)hipEventRecord(h->evUp[1], st); (void)hipStreamWaitEvent(own, h->evUp[1], 0); } }
    } rejoin{h, own, st};
    const size_t tight = (size_t)rows * cols, total = tight * B;
    bool pinned = false;
    const double ta = g_hostTiming ? nowSec() : 0;
    {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, imgs) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
        else (void)hipGetLastError();      // (older runtimes report unregistered memory as an error)
    }
    const double tb = g_hostTiming ? nowSec() : 0;
    if (g_hostTiming) g_hostT[2] += tb - ta;
    struct Tail { double t; ~Tail() { if (g_hostTiming) g_hostT[4] += nowSec() - t; } } tail{tb};      // [3] staging memcpy, [4] staging + enqueue of the copy
    if (!pinned && total <= h->hInBytes) {
        for (int f = 0; f < B; f++) {
            const uint8_t* src = imgs + f * frame_stride;
            uint8_t* dst = h->h_in + f * tight;
            if (stride == cols) std::memcpy(dst, src, tight);
            else for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * cols, src + (ptrdiff_t)y * stride, (size_t)cols);
        }
        if (g_hostTiming) g_hostT[3] += nowSec() - tb;
        HIP_TRY(h, hipMemcpyAsync(h->d_input, h->h_in, total, hipMemcpyHostToDevice, st));
    } else if (stride == cols && frame_stride == (ptrdiff_t)tight) {
        HIP_TRY(h, hipMemcpyAsync(h->d_input, imgs, total, hipMemcpyHostToDevice, st));
    } else {
        for (int f = 0; f < B; f++)
            HIP_TRY(h, hipMemcpy2DAsync(h->d_input + (size_t)f * tight, cols, imgs + f * frame_stride, stride, cols, rows, hipMemcpyHostToDevice, st));
    }
    return ORBX_OK;
}
}  // namespace

extern "C" {

int orbx_abi_version(void) { return ORBX_ABI_VERSION; }

const char* orbx_last_error(const orbx_handle* h) { return h ? h->err.c_str() : g_createError.c_str(); }

int orbx_compute_tables(int nfeatures, float scale_factor, int nlevels, float* sf, float* isf, float* s2, float* is2,
                        int* quota, int* umax16) {
    if (nlevels < 1 || nlevels > kMaxLevels || nfeatures < 1 || !(scale_factor > 1.0f)) return ORBX_ERR_BAD_ARGUMENT;
    ScaleTables t = makeScaleTables(nfeatures, scale_factor, nlevels);
    for (int i = 0; i < nlevels; i++) {
        if (sf) sf[i] = t.scale[i];
        if (isf) isf[i] = t.invScale[i];
        if (s2) s2[i] = t.sigma2[i];
        if (is2) is2[i] = t.invSigma2[i];
        if (quota) quota[i] = t.quota[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = t.umax[i];
    return ORBX_OK;
}

int orbx_compute_level_sizes(float scale_factor, int nlevels, int rows, int cols, int* widths, int* heights) {
    if (nlevels < 1 || nlevels > kMaxLevels || !(scale_factor > 1.0f)) return ORBX_ERR_BAD_ARGUMENT;
    ScaleTables t = makeScaleTables(1000, scale_factor, nlevels);
    for (int l = 0; l < nlevels; l++) {
        widths[l] = roundHalfEven((float)cols * t.invScale[l]);
        heights[l] = roundHalfEven((float)rows * t.invScale[l]);
    }
    return ORBX_OK;
}

int orbx_compute_cell_grid(float scale_factor, int nlevels, int rows, int cols, int level, int* n_cols, int* n_rows,
                           int* w_cell, int* h_cell, int* n_cells, int* n_ini, int* cand_cap) {
    if (nlevels < 1 || nlevels > kMaxLevels || level < 0 || level >= nlevels || !(scale_factor > 1.0f))
        return ORBX_ERR_BAD_ARGUMENT;
    ScaleTables t = makeScaleTables(1000, scale_factor, nlevels);
    FrameGeom g;
    std::string why = makeFrameGeom(t, rows, cols, g);
    if (!why.empty()) return why.find("small") != std::string::npos ? ORBX_ERR_IMAGE_TOO_SMALL : ORBX_ERR_UNSUPPORTED;
    const LevelGeom& L = g.lv[level];
    if (n_cols) *n_cols = L.nCols;
    if (n_rows) *n_rows = L.nRows;
    if (w_cell) *w_cell = L.wCell;
    if (h_cell) *h_cell = L.hCell;
    if (n_cells) *n_cells = L.cellCount;
    if (n_ini) *n_ini = L.nIni;
    if (cand_cap) *cand_cap = L.candCap;
    return ORBX_OK;
}

int orbx_create(orbx_handle** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                int max_width, int max_height, int max_batch, int device) {
    if (!out) return ORBX_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (nfeatures < 1 || nlevels < 1 || nlevels > kMaxLevels || !(scale_factor > 1.0f) || ini_th < 1 || min_th < 1 ||
        ini_th > 254 || min_th > 254 || max_width < 1 || max_height < 1 || max_batch < 1) {
        g_createError = "orbx_create: bad argument (need nfeatures>=1, 1<=nlevels<=16, scale>1, 1<=thresholds<=254)";
        return ORBX_ERR_BAD_ARGUMENT;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        g_createError = "orbx_create: no HIP device (this library has no CPU path)";
        return ORBX_ERR_NO_DEVICE;
    }
    orbx_handle* h = new orbx_handle();
    auto bail = [&](int code) {
        g_createError = h->err;
        freeAll(h);
        delete h;
        return code;
    };
#define CREATE_TRY(expr)                                                                 \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            h->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return bail(ORBX_ERR_HIP);                                                   \
        }                                                                                \
    } while (0)
    if (device < 0) CREATE_TRY(hipGetDevice(&device));
    if (device >= ndev) { h->err = "orbx_create: device index out of range"; return bail(ORBX_ERR_BAD_ARGUMENT); }
    h->device = device;
    CREATE_TRY(hipSetDevice(device));
    h->nfeatures = nfeatures; h->nlevels = nlevels; h->iniTh = ini_th; h->minTh = min_th; h->scaleFactor = scale_factor;
    h->maxW = max_width; h->maxH = max_height; h->maxB = max_batch;
    h->tabs = makeScaleTables(nfeatures, scale_factor, nlevels);
    // Launch-policy switches: read ONCE, here, and kept as a string (orbx_debug_policy; bench.py prints it as config.policy).  They choose between
    // result-identical launch forms (tests run every one); none of them can change a result or make a call fail.
    auto envInt = [&](const char* name, int dflt) {
        const char* e = getenv(name);
        const int v = e ? atoi(e) : dflt;
        h->policy += std::string(h->policy.empty() ? "" : " ") + (name + 5) + "=" + std::to_string(v) + (e ? "(env)" : "");
        return v;
    };
    std::string why = makeFrameGeom(h->tabs, max_height, max_width, h->maxGeom);
    if (!why.empty()) { h->err = "orbx_create: " + why; return bail(why.find("small") != std::string::npos ? ORBX_ERR_IMAGE_TOO_SMALL : ORBX_ERR_UNSUPPORTED); }
    layoutArenas(h->maxGeom, max_batch);
    const FrameGeom& mg = h->maxGeom;
    // a smaller image can need slightly more of a rounded quantity (cell grid, strides): 12.5 % + slack head-room
    auto roomy = [](size_t v) { return v + v / 8 + 4096; };
    h->pyrBytes = roomy((size_t)mg.pyrBytesPerFrame * max_batch);
    h->blurBytes = roomy((size_t)mg.blurBytesPerFrame * max_batch);
    h->candEntries = roomy((size_t)mg.candPerFrame * max_batch);
    h->selEntries = (size_t)(mg.selPerFrame + 8 * nlevels) * max_batch;
    h->cellCap = roomy(mg.cells.size());
    h->rxCap = (size_t)(max_width > max_height ? max_width : max_height) * nlevels + 64;
    h->tileCap = roomy((size_t)(3 * ((max_height + 31) / 32) + (max_height + kBlurBlockRowsSmall - 1) / kBlurBlockRowsSmall + 4) * nlevels);
    h->laneCap = roomy((size_t)((max_width + 3) / 4 + 1) * (3 * ((max_height + 31) / 32) + (max_height + kBlurBlockRowsSmall - 1) / kBlurBlockRowsSmall + 4) * nlevels);
    // quad-tree node arrays: LDS, or an HBM arena when the per-level quotas run into the thousands (orbx_geometry.hpp: octreeSizing)
    {
        const OctSizing z = octreeSizing(mg, nlevels, octreeLdsBytes);
        if (z.err) { h->err = std::string("orbx_create: ") + z.err; return bail(ORBX_ERR_UNSUPPORTED); }
        h->octM = z.M; h->octP = z.P; h->octR = z.R; h->octXT = z.XT;
        if (z.arena) {
            h->octArenaSlice = z.arenaSlice;
            if ((double)h->octArenaSlice * max_batch * nlevels > 16e9) {
                h->err = "orbx_create: quad-tree node arena for this nfeatures x max_batch exceeds 16 GB; lower max_batch";
                return bail(ORBX_ERR_UNSUPPORTED);
            }
        }
    }
    h->outCap = mg.selPerFrame + 8 * nlevels;

// device allocation of orbx_create; the test aid "poison" fills it (tests run with it: no kernel may depend on what hipMalloc returns,
// which is zeroed pages in a fresh process and another test's leftovers later).  Every fill of this function is enqueued on the handle's OWN
// stream - the stream the kernels run on - so "filled before first use" is stream order (docs/history/DESIGN_rounds_1-5.md §4j), not an assumption about the null stream.
#define CREATE_ALLOC(ptr, bytes)                                                         \
    do {                                                                                 \
        const size_t n_ = (size_t)(bytes);                                               \
        CREATE_TRY(hipMalloc(&(ptr), n_));                                               \
        if (poison >= 0) CREATE_TRY(hipMemsetAsync((ptr), poison, n_, h->stream));       \
    } while (0)

    const int poison = g_aids.poison >= 0 ? g_aids.poison & 255 : -1;
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->ownStream = true;
    CREATE_ALLOC(h->d_input, (size_t)max_width * max_height * max_batch);
    CREATE_ALLOC(h->d_pyr, h->pyrBytes);
    CREATE_ALLOC(h->d_blur, h->blurBytes);
    const size_t candBytes = mg.big ? sizeof(CandFmt<true>::T) : sizeof(CandFmt<false>::T);      // (a handle for frames beyond 4096 px: two dwords per candidate)
    CREATE_ALLOC(h->d_candPos, h->candEntries * candBytes);
    CREATE_ALLOC(h->d_candSeg, h->candEntries * candBytes);
    CREATE_ALLOC(h->d_cellCount, sizeof(unsigned) * h->cellCap * max_batch);
    CREATE_ALLOC(h->d_cellOff, sizeof(int) * h->cellCap * max_batch);
    CREATE_ALLOC(h->d_nodeOf, h->candEntries * sizeof(unsigned short));
    CREATE_ALLOC(h->d_candCount, sizeof(unsigned) * max_batch * nlevels);
    CREATE_ALLOC(h->d_sink, 64);
    CREATE_ALLOC(h->d_sel, h->selEntries * sizeof(uint2));
    if (h->octArenaSlice) CREATE_ALLOC(h->d_octArena, h->octArenaSlice * max_batch * nlevels);
    // (k_fast pays ~6 % for the tables, k_octree loses its first sweep: 640x480, us per call without -> with: 12 frames 110 -> 101, 32: 169 -> 163,
    // 64: 306 -> 296, 128: 562 -> 556, 256: 1040 -> 1046, 512: 2040 -> 2053; 1920x1080: 16 frames 519 -> 431, 32: 904 -> 827, 128: 3108 -> 3092)
    h->leafFrames = envInt("ORBX_LEAF_FRAMES", 128);
    if (h->leafFrames > max_batch) h->leafFrames = max_batch;
    if (h->octR < 1 || h->octArenaSlice || h->leafFrames < 0) h->leafFrames = 0;      // no dense phase, or node arrays in HBM: the first sweep stays in k_octree
    // (k_fast indexes the tables with 32 bits: cap the frames so that frames x levels x roots x 1024 stays below 2^30 entries)
    if (h->leafFrames && (size_t)h->leafFrames * nlevels * h->octR * kOctLeaves >= ((size_t)1 << 30)) h->leafFrames = (int)((((size_t)1 << 30) - 1) / ((size_t)nlevels * h->octR * kOctLeaves));
    if (h->leafFrames) {
        const size_t n = (size_t)h->leafFrames * nlevels * h->octR * kOctLeaves;
        CREATE_ALLOC(h->d_leafHist, n * sizeof(int));
        CREATE_ALLOC(h->d_leafBest, n * sizeof(unsigned));
        CREATE_TRY(hipMemsetAsync(h->d_leafHist, 0, n * sizeof(int), h->stream));          // the tables are zero between calls (also under the poison aid)
        CREATE_TRY(hipMemsetAsync(h->d_leafBest, 0, n * sizeof(unsigned), h->stream));
        CREATE_ALLOC(h->d_leafCode, (size_t)2 * nlevels * h->octXT);
    }
    CREATE_ALLOC(h->d_levelCount, sizeof(int) * max_batch * nlevels);
    CREATE_ALLOC(h->d_levelLap, sizeof(int) * max_batch * nlevels);
    CREATE_ALLOC(h->d_lap, sizeof(int) * 2 * max_batch);
    CREATE_ALLOC(h->d_lv, sizeof(LevelGeom) * kMaxLevels);
    CREATE_ALLOC(h->d_cells, sizeof(CellDesc) * h->cellCap);
    CREATE_ALLOC(h->d_rx, sizeof(ResizeX) * h->rxCap);
    CREATE_ALLOC(h->d_ry, sizeof(ResizeX) * h->rxCap);
    h->xqCap = ((size_t)(max_width + 2 * kEdge + kPadL) / 4 + 2) * nlevels;
    CREATE_ALLOC(h->d_xq, sizeof(QuadRec) * h->xqCap);
    CREATE_ALLOC(h->d_tiles, sizeof(BlurItem) * h->tileCap);
    CREATE_ALLOC(h->d_laneItem, sizeof(unsigned short) * h->laneCap);
    h->pyrCols = envInt("ORBX_PYR_COLS", -1);
    h->colPx = envInt("ORBX_PYR_COL_PX", 0);
    h->fastWide = envInt("ORBX_FAST_WIDE", -1);
    h->patchBlur = envInt("ORBX_PATCH_BLUR", -1);
    h->blurSplit = envInt("ORBX_BLUR_SPLIT", -1);
    h->sharedUploadBytes = g_aids.sharedUploadBytes;
    h->colsShape = g_aids.colsShape == 1 || g_aids.colsShape == 4 || g_aids.colsShape == 6 ? g_aids.colsShape : -1;
    h->colsCap = 0;
    for (int px : kColPx) h->colsCap += (size_t)((max_width + px / 2) / px + 1) * ((max_height + px / 2) / px + 1);
    h->colCoefCap = (h->colsCap + h->colsCap / 8 + 16) * (size_t)kChainCoefMax * 5 / 8;      // (a region's list is 0.4 - 0.8 of the kernel's limit; a geometry past this keeps the tile forms)
    h->colsCap = roomy(h->colsCap);
    CREATE_ALLOC(h->d_colCoef, sizeof(ResizeX) * h->colCoefCap);
    CREATE_ALLOC(h->d_cols, sizeof(PyrColumn) * h->colsCap);
    CREATE_ALLOC(h->d_colLevels, sizeof(ColLevels));
    h->footCap = roomy((size_t)((max_width + 38 + 255) / 256 + 1) * ((max_height + 38 + 31) / 32 + 1) * nlevels);
    CREATE_ALLOC(h->d_foot, sizeof(TileFoot) * h->footCap);
    h->outBytes = outLayout(max_batch, h->outCap, nlevels).all;
    CREATE_ALLOC(h->d_out, h->outBytes);
    h->testFailAfterFast = g_aids.failAfterFast != 0;
    g_aids.failAfterFast = 0;      // one shot: this handle's first call with leaf tables
    h->zeroCopy = envInt("ORBX_ZERO_COPY", 1) != 0;
    CREATE_TRY(hipHostMalloc(&h->h_lap, sizeof(int) * 2 * max_batch));
    CREATE_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&h->aux2, hipStreamNonBlocking));      // (default priority: a low-priority blur only starts when everything else is done - 1928 -> 2022 us; high = default)
    for (int i = 0; i < 2; i++) { CREATE_TRY(hipEventCreateWithFlags(&h->evPyr[i], hipEventDisableTiming)); CREATE_TRY(hipEventCreateWithFlags(&h->evBlur[i], hipEventDisableTiming)); }
    for (int i = 0; i < 2; i++) CREATE_TRY(hipEventCreateWithFlags(&h->evUp[i], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming));
    h->splitMode = envInt("ORBX_SPLIT", 1);
    h->fuseSmall = envInt("ORBX_FUSE_SMALL", 1) != 0;
    {
        const char* e = getenv("ORBX_SPLIT_MIN_MPX");
        h->splitMinPixels = (long long)((e ? atof(e) : 120.0) * 1e6);
        h->policy += std::string(" SPLIT_MIN_MPX=") + (e ? e : "120") + (e ? "(env)" : "");
    }
    h->resizeBytewise = getenv("ORBX_RESIZE_BYTEWISE") != nullptr;      // diagnostic: the byte-gather form of k_resize
    h->octRoomyForced = getenv("ORBX_OCT_ROOMY") != nullptr;
    h->policy += std::string(" RESIZE_BYTEWISE=") + (h->resizeBytewise ? "1(env)" : "0") + " OCT_ROOMY=" + (h->octRoomyForced ? "1(env)" : "0");
    h->octThreadsForced = envInt("ORBX_OCT_THREADS", 0);   // tuning switch: 256, 512 or 1024
    if (g_aids.ldsPollute >= 0) h->ldsPollute = g_aids.ldsPollute & 255;
    if (poison >= 0 || h->ldsPollute >= 0 || h->testFailAfterFast || h->colsShape > 0 || h->sharedUploadBytes >= 0)
        h->policy += " test_aids=poison:" + std::to_string(poison) + ",lds_pollute:" + std::to_string(h->ldsPollute) + ",fail_after_fast:" + std::to_string((int)h->testFailAfterFast) +
                     ",pyr_cols_shape:" + std::to_string(h->colsShape) + ",shared_upload_bytes:" + std::to_string(h->sharedUploadBytes);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->numCUs = cus;
    }
    CREATE_TRY(hipHostMalloc(&h->h_out, h->outBytes));
    std::memset(h->h_out, 0, h->outBytes);
    // pageable input up to this size is gathered in pinned memory first (a hipMemcpyAsync from pageable memory stages and waits inside the call)
    h->hInBytes = std::min((size_t)max_width * max_height * max_batch, (size_t)8 << 20);
    CREATE_TRY(hipHostMalloc(&h->h_in, h->hInBytes));
    // the border of the pyramid arena is only ever written by the kernels, but the padding bytes between
    // rows are never written: clear once so introspection copies are deterministic
    CREATE_TRY(hipMemsetAsync(h->d_pyr, 0, h->pyrBytes, h->stream));
    CREATE_TRY(hipMemsetAsync(h->d_blur, 0, h->blurBytes, h->stream));
    if (!checkUmax(h->tabs.umax)) { h->err = "orbx_create: the patch-radius table differs from the device's static copy (internal)"; return bail(ORBX_ERR_UNSUPPORTED); }
    {   // k_fast's packed passes assume the 3-input packed f16 min/max act as integer min/max on u16 halves 0..255
        unsigned bad = 1;
        CREATE_TRY(runPackedSelfTest(h->stream, (unsigned*)h->d_candCount, &bad));
        if (bad) {
            h->err = "packed min/max self-test failed on this device (FP16 denormals not preserved?): the FAST kernel would not be exact";
            return bail(ORBX_ERR_UNSUPPORTED);
        }
    }
    // (runPackedSelfTest ended with a wait for h->stream: the fills above, enqueued on that stream before it, are over as well.  orbx_set_stream
    // waits for the old stream before it adopts the caller's.)
#undef CREATE_TRY
    h->geom = FrameGeom();   // installed on first use
    *out = h;
    return ORBX_OK;
}

void orbx_destroy(orbx_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    freeAll(h);
    delete h;
}

int orbx_get_levels(const orbx_handle* h) { return h ? h->nlevels : 0; }
float orbx_get_scale_factor(const orbx_handle* h) { return h ? h->scaleFactor : 0.f; }
int orbx_get_tables(const orbx_handle* h, float* sf, float* isf, float* s2, float* is2, int* quota, int* umax16) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    return orbx_compute_tables(h->nfeatures, h->scaleFactor, h->nlevels, sf, isf, s2, is2, quota, umax16);
}
int orbx_max_keypoints(const orbx_handle* h) { return h ? h->outCap : 0; }

int orbx_set_stream(orbx_handle* h, void* hip_stream) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->stream) HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->ownStream && h->stream) { (void)hipStreamDestroy(h->stream); h->ownStream = false; }
    h->stream = (hipStream_t)hip_stream;
    return ORBX_OK;
}
void* orbx_get_stream(const orbx_handle* h) { return h ? (void*)h->stream : nullptr; }
int orbx_synchronize(orbx_handle* h) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ORBX_OK;
}

int orbx_extract_batch_device(orbx_handle* h, int n_frames, const uint8_t* d_imgs, int rows, int cols,
                              ptrdiff_t stride, ptrdiff_t frame_stride, const int* lap, orbx_keypoint* d_kps,
                              uint8_t* d_desc, int capacity, int* d_n_out, int* d_mono_out, orbx_keypoint* d_level_kps,
                              int* d_level_counts) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_imgs || rows <= 0 || cols <= 0) return fail(h, ORBX_ERR_EMPTY_IMAGE, "empty image");
    if (!d_kps || !d_desc || !d_n_out || !d_mono_out || capacity < 1 || stride < cols)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null output pointer, capacity < 1 or stride < cols");
    int rc = checkFrameArgs(h, n_frames, rows, cols);
    if (rc != ORBX_OK) return rc;
    return enqueueBatch(h, n_frames, d_imgs, rows, cols, (long long)stride, (long long)frame_stride, lap,
                        (Keypoint*)d_kps, d_desc, capacity, d_n_out, d_mono_out, (Keypoint*)d_level_kps, d_level_counts);
}

int orbx_extract_batch_begin(orbx_handle* h, int n_frames, const uint8_t* imgs, int rows, int cols, ptrdiff_t stride,
                             ptrdiff_t frame_stride, const int* lap, int want_levels) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (h->pendingB) return fail(h, ORBX_ERR_BAD_ARGUMENT, "a batch is already in flight on this handle: call orbx_extract_batch_end first");
    if (!imgs || rows <= 0 || cols <= 0) return fail(h, ORBX_ERR_EMPTY_IMAGE, "empty image");
    if (stride < cols) return fail(h, ORBX_ERR_BAD_ARGUMENT, "stride < cols");
    int rc = checkFrameArgs(h, n_frames, rows, cols);
    if (rc != ORBX_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int B = n_frames;
    rc = uploadFrames(h, B, imgs, rows, cols, stride, frame_stride);
    if (rc != ORBX_OK) return rc;
    const int cap = h->outCap;
    const OutLayout lo = outLayout(B, cap, h->nlevels);
    // one frame per call: the kernels write the results into the pinned slab themselves (no copy command at all)
    const bool zc = B == 1 && h->zeroCopy;
    h->dev = outView(zc ? h->h_out : h->d_out, lo);
    h->host = outView(h->h_out, lo);
    rc = enqueueBatch(h, B, h->d_input, rows, cols, cols, (long long)rows * cols, lap, h->dev.k, h->dev.d, cap, h->dev.n, h->dev.mono,
                      want_levels ? h->dev.lk : nullptr, want_levels ? h->dev.lc : nullptr);
    if (rc != ORBX_OK) return rc;
    if (!zc) {
        // the results: by DMA where the input went through the device's shared copy queue (large batches), else by a kernel on the handle's
        // stream - a result DMA enqueued now would sit in the copy queue in front of the NEXT handle's input copy and hold it until this batch's
        // kernels are done (k_copy.hip; uploadFrames)
        const size_t bytes = want_levels ? lo.all : lo.noLevels;
        if (usesSharedUpload(h, B, rows, cols)) HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, bytes, hipMemcpyDeviceToHost, st));
        else launchCopyOut(st, h->d_out, h->h_out, bytes, h->numCUs);
    }
    h->pendingB = B;
    h->lastHostB = B;
    h->pendingLevels = want_levels != 0;
    return ORBX_OK;
}

int orbx_extract_batch_end(orbx_handle* h, orbx_keypoint* kps, uint8_t* desc, int capacity, int* n_out, int* mono_out,
                           orbx_keypoint* level_kps, int* level_counts) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!h->pendingB) return fail(h, ORBX_ERR_BAD_ARGUMENT, "no batch in flight: call orbx_extract_batch_begin first");
    const int B = h->pendingB, cap = h->outCap;
    h->pendingB = 0;
    if (!kps || !desc || !n_out || !mono_out || capacity < 1) return fail(h, ORBX_ERR_BAD_ARGUMENT, "null output pointer or capacity < 1");
    if ((level_kps || level_counts) && !h->pendingLevels)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "per-level outputs requested but the batch was begun with want_levels = 0");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const orbx_handle::OutView& r = h->host;
    int status = ORBX_OK;
    for (int f = 0; f < B; f++) {
        const int n = r.n[f];
        n_out[f] = n;
        mono_out[f] = r.mono[f];
        if (n > capacity || n > cap) {
            status = fail(h, ORBX_ERR_CAPACITY, "keypoint count exceeds the caller's capacity");
            continue;
        }
        std::memcpy(kps + (size_t)f * capacity, r.k + (size_t)f * cap, (size_t)n * sizeof(Keypoint));
        std::memcpy(desc + (size_t)f * capacity * 32, r.d + (size_t)f * cap * 32, (size_t)n * 32);
        if (level_kps) std::memcpy(level_kps + (size_t)f * capacity, r.lk + (size_t)f * cap, (size_t)n * sizeof(Keypoint));
        if (level_counts) std::memcpy(level_counts + (size_t)f * h->nlevels, r.lc + (size_t)f * h->nlevels, sizeof(int) * h->nlevels);
    }
    return status;
}

int orbx_extract_batch_end_view(orbx_handle* h, const orbx_keypoint** kps, const uint8_t** desc, int* capacity,
                                const int** n_out, const int** mono_out) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!h->pendingB) return fail(h, ORBX_ERR_BAD_ARGUMENT, "no batch in flight: call orbx_extract_batch_begin first");
    h->pendingB = 0;
    if (!kps || !desc || !capacity || !n_out || !mono_out) return fail(h, ORBX_ERR_BAD_ARGUMENT, "null output pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *kps = (const orbx_keypoint*)h->host.k; *desc = h->host.d; *capacity = h->outCap;
    *n_out = h->host.n; *mono_out = h->host.mono;
    return ORBX_OK;
}

int orbx_extract_batch(orbx_handle* h, int n_frames, const uint8_t* imgs, int rows, int cols, ptrdiff_t stride,
                       ptrdiff_t frame_stride, const int* lap, orbx_keypoint* kps, uint8_t* desc, int capacity,
                       int* n_out, int* mono_out, orbx_keypoint* level_kps, int* level_counts) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!imgs || rows <= 0 || cols <= 0) return fail(h, ORBX_ERR_EMPTY_IMAGE, "empty image");
    if (!kps || !desc || !n_out || !mono_out || capacity < 1 || stride < cols)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null output pointer, capacity < 1 or stride < cols");
    int rc = orbx_extract_batch_begin(h, n_frames, imgs, rows, cols, stride, frame_stride, lap, level_kps || level_counts);
    if (rc != ORBX_OK) return rc;
    return orbx_extract_batch_end(h, kps, desc, capacity, n_out, mono_out, level_kps, level_counts);
}

void* orbx_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes) == hipSuccess ? p : nullptr;
}
void orbx_host_free(void* p) { if (p) (void)hipHostFree(p); }

int orbx_extract(orbx_handle* h, const uint8_t* img, int rows, int cols, ptrdiff_t stride, int lap0, int lap1,
                 orbx_keypoint* kps, uint8_t* desc, int capacity, int* n_out, int* mono_out, orbx_keypoint* level_kps,
                 int* level_counts) {
    const int lap[2] = {lap0, lap1};
    return orbx_extract_batch(h, 1, img, rows, cols, stride, (ptrdiff_t)rows * stride, lap, kps, desc, capacity, n_out,
                              mono_out, level_kps, level_counts);
}

int orbx_extract_view(orbx_handle* h, const uint8_t* img, int rows, int cols, ptrdiff_t stride, int lap0, int lap1, int want_levels,
                      const orbx_keypoint** kps, const uint8_t** desc, int* n_out, int* mono_out, const orbx_keypoint** level_kps,
                      const int** level_counts) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!img || rows <= 0 || cols <= 0) return fail(h, ORBX_ERR_EMPTY_IMAGE, "empty image");
    if (!kps || !desc || !n_out || !mono_out || stride < cols || (want_levels && (!level_kps || !level_counts)))
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null output pointer or stride < cols");
    const int lap[2] = {lap0, lap1};
    const double t0 = g_hostTiming ? nowSec() : 0;
    int rc = orbx_extract_batch_begin(h, 1, img, rows, cols, stride, (ptrdiff_t)rows * stride, lap, want_levels);
    if (rc != ORBX_OK) return rc;
    h->pendingB = 0;
    const double t1 = g_hostTiming ? nowSec() : 0;
    HIP_TRY(h, hipStreamSynchronize(h->stream));      // (polling hipStreamQuery instead returns no sooner: 71.7-71.9 against 68.3-72.9 us per call)
    if (g_hostTiming) { const double t2 = nowSec(); g_hostT[0] += t1 - t0; g_hostT[1] += t2 - t1; g_hostN++; }
    const int n = h->host.n[0];
    if (n < 0 || n > h->outCap) return fail(h, ORBX_ERR_CAPACITY, "keypoint count exceeds the handle's capacity (internal bound violated)");
    *kps = (const orbx_keypoint*)h->host.k; *desc = h->host.d; *n_out = n; *mono_out = h->host.mono[0];
    if (want_levels) { *level_kps = (const orbx_keypoint*)h->host.lk; *level_counts = h->host.lc; }
    return ORBX_OK;
}

int orbx_compute_pyramid(orbx_handle* h, const uint8_t* img, int rows, int cols, ptrdiff_t stride) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (h->pendingB) return fail(h, ORBX_ERR_BAD_ARGUMENT, "a batch is already in flight on this handle: call orbx_extract_batch_end first");
    if (!img || rows <= 0 || cols <= 0) return fail(h, ORBX_ERR_EMPTY_IMAGE, "empty image");
    if (stride < cols) return fail(h, ORBX_ERR_BAD_ARGUMENT, "stride < cols");
    int rc = checkFrameArgs(h, 1, rows, cols);
    if (rc != ORBX_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = uploadFrames(h, 1, img, rows, cols, stride, (ptrdiff_t)rows * stride);
    if (rc != ORBX_OK) return rc;
    rc = enqueueBatch(h, 1, h->d_input, rows, cols, cols, (long long)rows * cols, nullptr, nullptr, nullptr, h->outCap, nullptr, nullptr, nullptr,
                      nullptr, kStageFront);
    if (rc != ORBX_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));      // ComputePyramid returns with mvImagePyramid filled (ORBextractor.cc:1164-1219)
    return ORBX_OK;
}

int orbx_compute_keypoints_octree(orbx_handle* h, orbx_keypoint* level_kps, int capacity, int* level_counts) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!level_kps || !level_counts || capacity < 1) return fail(h, ORBX_ERR_BAD_ARGUMENT, "null output pointer or capacity < 1");
    if (h->pendingB) return fail(h, ORBX_ERR_BAD_ARGUMENT, "a batch is already in flight on this handle: call orbx_extract_batch_end first");
    if (h->geom.nlevels == 0 || h->lastB < 1)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "orbx_compute_keypoints_octree: the handle holds no pyramid (call orbx_compute_pyramid or an extract call first)");
    HIP_TRY(h, hipSetDevice(h->device));
    const int cap = h->outCap;
    const OutLayout lo = outLayout(1, cap, h->nlevels);
    h->dev = outView(h->zeroCopy ? h->h_out : h->d_out, lo);
    h->host = outView(h->h_out, lo);
    // the stages after the pyramid on frame 0 of what the handle holds; the final arrays (level-0 coordinates, descriptors) are a by-product
    int rc = enqueueBatch(h, 1, h->d_input, h->geom.rows, h->geom.cols, h->geom.cols, (long long)h->geom.rows * h->geom.cols, nullptr, h->dev.k,
                          h->dev.d, cap, h->dev.n, h->dev.mono, h->dev.lk, h->dev.lc, kStageBack);
    if (rc != ORBX_OK) return rc;
    if (!h->zeroCopy) HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, lo.all, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->lastHostB = 1;
    const int n = h->host.n[0];
    if (n < 0 || n > cap) return fail(h, ORBX_ERR_CAPACITY, "keypoint count exceeds the handle's capacity (internal bound violated)");
    if (n > capacity) return fail(h, ORBX_ERR_CAPACITY, "keypoint count exceeds the caller's capacity");
    std::memcpy(level_kps, h->host.lk, (size_t)n * sizeof(Keypoint));
    std::memcpy(level_counts, h->host.lc, sizeof(int) * h->nlevels);
    return ORBX_OK;
}

int orbx_fetch_pyramid(orbx_handle* h, int frame, const uint8_t** base, size_t* level_offset, int* level_stride, int* widths, int* heights) {
    if (!h || !base || !level_offset || !level_stride || !widths || !heights) return ORBX_ERR_BAD_ARGUMENT;
    if (h->geom.nlevels == 0 || frame < 0 || frame >= h->lastB)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "orbx_fetch_pyramid: no such frame in the last batch");
    HIP_TRY(h, hipSetDevice(h->device));
    const FrameGeom& g = h->geom;
    size_t total = 0;
    for (int l = 0; l < g.nlevels; l++) total += (size_t)g.lv[l].pyrFrameBytes;
    if (total > h->hPyrBytes) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->h_pyr) (void)hipHostFree(h->h_pyr);
        h->h_pyr = nullptr; h->hPyrBytes = 0;
        size_t want = 0;      // room for the largest geometry of the handle, so that this happens once
        for (int l = 0; l < h->maxGeom.nlevels; l++) want += (size_t)h->maxGeom.lv[l].pyrStride * h->maxGeom.lv[l].pyrRows;
        want = std::max(want + want / 8 + 4096, total);
        HIP_TRY(h, hipHostMalloc(&h->h_pyr, want));
        h->hPyrBytes = want;
    }
    // a handle made for one frame per call holds the levels of its frame back to back (level-major arena): ONE copy
    bool contiguous = true;
    for (int l = 1; l < g.nlevels; l++) contiguous = contiguous && g.lv[l].pyrOff + (long long)frame * g.lv[l].pyrFrameBytes ==
                                                                     g.lv[l - 1].pyrOff + (long long)(frame + 1) * g.lv[l - 1].pyrFrameBytes;
    size_t at = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        if (!contiguous) HIP_TRY(h, hipMemcpyAsync(h->h_pyr + at, h->d_pyr + L.pyrOff + (long long)frame * L.pyrFrameBytes, (size_t)L.pyrFrameBytes, hipMemcpyDeviceToHost, h->stream));
        level_offset[l] = at + (size_t)kEdge * L.pyrStride + kPadL;      // interior pixel (0, 0); the REFLECT_101 frame of :1193-1215 lies around it
        level_stride[l] = L.pyrStride; widths[l] = L.w; heights[l] = L.h;
        at += (size_t)L.pyrFrameBytes;
    }
    if (contiguous) HIP_TRY(h, hipMemcpyAsync(h->h_pyr, h->d_pyr + g.lv[0].pyrOff + (long long)frame * g.lv[0].pyrFrameBytes, total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *base = h->h_pyr;
    return ORBX_OK;
}

int orbx_get_level(orbx_handle* h, int frame, int level, int bordered, uint8_t* dst, ptrdiff_t dst_stride, int* width,
                   int* height) {
    if (!h || !dst) return ORBX_ERR_BAD_ARGUMENT;
    if (h->geom.nlevels == 0 || frame < 0 || frame >= h->lastB || level < 0 || level >= h->nlevels)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "orbx_get_level: no such frame/level in the last batch");
    HIP_TRY(h, hipSetDevice(h->device));
    const LevelGeom& L = h->geom.lv[level];
    const int w = bordered ? L.w + 2 * kEdge : L.w, hh = bordered ? L.h + 2 * kEdge : L.h;
    if (dst_stride < w) return fail(h, ORBX_ERR_BAD_ARGUMENT, "orbx_get_level: dst_stride too small");
    const uint8_t* src = h->d_pyr + L.pyrOff + (long long)frame * L.pyrFrameBytes +
                         (bordered ? (kPadL - kEdge) : ((long long)kEdge * L.pyrStride + kPadL));
    HIP_TRY(h, hipMemcpy2DAsync(dst, dst_stride, src, L.pyrStride, w, hh, hipMemcpyDeviceToHost, h->stream));      // (behind the kernels, in stream order)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (width) *width = L.w;
    if (height) *height = L.h;
    return ORBX_OK;
}

}  // extern "C"
