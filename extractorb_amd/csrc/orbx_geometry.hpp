// orbx_geometry.hpp — host-side geometry and tables of the ORB extractor (pure C++, no HIP).
//
// Everything here is computed once per (handle, image size) on the host with the reference's own
// float/double sequence, so the device kernels only see integers and pre-rounded floats:
//   * scale tables / per-level quotas / umax      reference ORBextractor.cc:419-474
//   * pyramid level sizes                         reference ORBextractor.cc:1171
//   * bilinear resize coefficient tables          cv::resize INTER_LINEAR 8u
//   * FAST cell grid per level                    reference ORBextractor.cc:781-814
//   * quad-tree roots per level                   reference ORBextractor.cc:548-568
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "orbx_device.hpp"

namespace orbx {

inline int roundHalfEven(float v) { return (int)std::lrintf(v); }   // cvRound(float)
inline int roundHalfEven(double v) { return (int)std::lrint(v); }   // cvRound(double)
inline int roundUp(int v, int m) { return (v + m - 1) / m * m; }

// ---- tables of the constructor ---------------------------------------------------------------
struct ScaleTables {
    int nlevels = 0;
    float scale[kMaxLevels] = {}, invScale[kMaxLevels] = {}, sigma2[kMaxLevels] = {}, invSigma2[kMaxLevels] = {};
    int quota[kMaxLevels] = {};   // mnFeaturesPerLevel
    int patchSize[kMaxLevels] = {};   // (int)(PATCH_SIZE * scale[level]), ORBextractor.cc:872
    int umax[kHalfPatch + 1] = {};
};

inline ScaleTables makeScaleTables(int nfeatures, float scaleFactorF, int nlevels) {
    ScaleTables t;
    t.nlevels = nlevels;
    const double scaleFactor = scaleFactorF;   // the member is a double holding the float argument (ORBextractor.h:98)
    t.scale[0] = 1.0f;
    t.sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        t.scale[i] = (float)(t.scale[i - 1] * scaleFactor);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        t.invScale[i] = 1.0f / t.scale[i];
        t.invSigma2[i] = 1.0f / t.sigma2[i];
        t.patchSize[i] = (int)(kPatch * t.scale[i]);
    }
    // geometric split of nfeatures over the levels, remainder to the coarsest (ORBextractor.cc:439-451)
    const float factor = (float)(1.0f / scaleFactor);
    float want = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        t.quota[l] = roundHalfEven(want);
        sum += t.quota[l];
        want *= factor;
    }
    t.quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    // half-widths of the radius-15 disc rows (ORBextractor.cc:459-474)
    const float halfDiag = kHalfPatch * std::sqrt(2.f) / 2;
    const int vmax = (int)std::floor(halfDiag + 1), vmin = (int)std::ceil(halfDiag);
    const double r2 = (double)kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; v++) t.umax[v] = roundHalfEven(std::sqrt(r2 - (double)v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    return t;
}

inline short satShort(float v) {
    int i = roundHalfEven(v);
    return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

// cv::resize(INTER_LINEAR) coefficient set-up for one axis, 8u fixed-point path.
// zeroAtEdges: the x axis zeroes the fraction in clamped columns, the y axis only clamps the rows.
inline void resizeAxis(int srcN, int dstN, bool isX, std::vector<ResizeX>& out) {
    out.resize(dstN);
    const double scale = 1.0 / ((double)dstN / srcN);
    for (int d = 0; d < dstN; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        int s0, s1;
        if (isX) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= srcN - 1) { f = 0; s = srcN - 1; }
            s0 = s;
            s1 = s + 1 < srcN ? s + 1 : srcN - 1;   // weight of s1 is 0 whenever it was clamped
        } else {
            auto clip = [&](int v) { return v < 0 ? 0 : (v < srcN ? v : srcN - 1); };
            s0 = clip(s);
            s1 = clip(s + 1);
        }
        out[d] = ResizeX{(short)s0, (short)s1, satShort((1.f - f) * 2048), satShort(f * 2048)};
    }
}

struct FrameGeom {
    int rows = 0, cols = 0, nlevels = 0;
    LevelGeom lv[kMaxLevels];
    std::vector<CellDesc> cells;             // all levels, level 0 first, raster order inside a level
    std::vector<ResizeX> rx[kMaxLevels];     // level l from level l-1 (l >= 1)
    std::vector<ResizeX> ry[kMaxLevels];
    std::vector<TileFoot> foot[kMaxLevels];  // per 256 x kResizeTileRows tile of bordered level l (l >= 1), row-major over tiles
    int tilesX[kMaxLevels] = {}, tilesY[kMaxLevels] = {};
    int tileLdsStride = 16, tileLdsRows = 1; // LDS tile able to hold the largest footprint
    bool packedTaps[kMaxLevels] = {};        // level l: the 8 source taps of every aligned 4-pixel group span <= 8 bytes
    std::vector<QuadRec> xq[kMaxLevels];     // level l, packed taps: what a thread of the tile resize needs of its dword column (k_resize / k_pyr_first)
    long long pyrBytesPerFrame = 0, blurBytesPerFrame = 0;
    long long candPerFrame = 0;              // sum of candCap
    int selPerFrame = 0;                     // sum of selCap
    int maxRoiW = 0, maxRoiH = 0;            // largest FAST ROI over all cells
    int maxNodes = 0;                        // largest quad-tree node count over all levels
    long long sumPixels = 0;                 // S of SURVEY.md §8d
    // region-major pyramid (k_pyr_cols): the image cut into RX x RY regions (row-major) of about px x px level-0 pixels, for several px:
    // a launch takes the coarsest cut that still gives the chip enough workgroups (few frames: small regions, short chains of small steps;
    // more frames: large regions, less overlap)
    struct ColumnSet {
        int px = 0, RX = 0, RY = 0;
        std::vector<PyrColumn> columns;
        bool fit = false;
        int ldsBytes = 0, evenBytes = 0;
        int coefSlot = 0;                    // records per region in coef (the largest list, rounded up to 8)
        std::vector<ResizeX> coef;           // columns.size() * coefSlot
    };
    std::vector<ColumnSet> colSets;          // finest first
    bool colsPacked = false;
    bool big = false;                        // a level beyond 4096 px: candidates in the two-dword format (CandFmt<true>), 1024-thread quad-tree, no leaf tables
};
constexpr int kColPx[] = {40, 56, 80, 112};
#ifndef ORBX_COL_EDGE_NUM
#define ORBX_COL_EDGE_NUM 3
#define ORBX_COL_EDGE_DEN 5
#endif
constexpr int kColEdgeNum = ORBX_COL_EDGE_NUM, kColEdgeDen = ORBX_COL_EDGE_DEN;      // an outer region's share of the interior, as a fraction of an inner one's (fine cuts only)

// Returns an empty string on success, else the reason the geometry is unsupported.
// colPx: side (level-0 pixels) of the regions of the region-major pyramid (0: the sizes of kColPx)
inline std::string makeFrameGeom(const ScaleTables& t, int rows, int cols, FrameGeom& g, int colPx = 0) {
    g = FrameGeom();
    g.rows = rows; g.cols = cols; g.nlevels = t.nlevels;
    for (int l = 0; l < t.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        L.w = roundHalfEven((float)cols * t.invScale[l]);
        L.h = roundHalfEven((float)rows * t.invScale[l]);
        const int maxBorderX = L.w - kEdge + 3, maxBorderY = L.h - kEdge + 3;
        L.rectW = maxBorderX - kMinBorder;
        L.rectH = maxBorderY - kMinBorder;
        const float width = (float)L.rectW, height = (float)L.rectH;
        L.nCols = (int)(width / (float)kCellW);
        L.nRows = (int)(height / (float)kCellW);
        if (L.nCols < 1 || L.nRows < 1) return "image too small: a pyramid level is narrower than one 30-px FAST cell";
        // (the reference has no size limit, ORBextractor.cc:1171; here: 16-bit node boxes and the two-dword candidate format up to 16384 px, one-dword
        // candidates - x:12 | y:12 | response:8 - while every level stays within 4096 px: orbx_device.hpp, CandFmt)
        if (L.w > kBigMax || L.h > kBigMax) return "image too large: a pyramid level beyond 16384 px";
        if (L.w > kNarrowMax || L.h > kNarrowMax) g.big = true;
        L.wCell = (int)std::ceil(width / L.nCols);
        L.hCell = (int)std::ceil(height / L.nRows);
        if (L.wCell > 63 || L.hCell > 63) return "unsupported cell size (> 63 px)";
        L.pyrStride = roundUp(kPadL + L.w + kEdge, 64);
        L.pyrRows = L.h + 2 * kEdge;
        L.blurStride = roundUp(L.w, 64);
        L.quota = t.quota[l];
        L.nIni = (int)std::round((float)L.rectW / (float)L.rectH);
        if (L.nIni < 1) return "unsupported aspect ratio: the reference builds zero quad-tree roots (height > 2*width)";
        L.hX = (float)L.rectW / (float)L.nIni;
        L.scale = t.scale[l];
        L.patchSize = t.patchSize[l];
        // cells in the reference's loop order; a skipped cell (continue at :802-803, :811-812) is simply absent
        L.cellFirst = (int)g.cells.size();
        long long cap = 0;
        for (int i = 0; i < L.nRows; i++) {
            const int iniY = kMinBorder + i * L.hCell;
            int maxY = iniY + L.hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = maxBorderY;
            for (int j = 0; j < L.nCols; j++) {
                const int iniX = kMinBorder + j * L.wCell;
                int maxX = iniX + L.wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = maxBorderX;
                CellDesc c;
                c.level = (short)l;
                c.roiW = (short)(maxX - iniX);
                c.roiH = (short)(maxY - iniY);
                c.x0 = (short)iniX; c.y0 = (short)iniY;
                c.shiftX = (short)(j * L.wCell); c.shiftY = (short)(i * L.hCell);
                c.pad = 0; c.pyrStride = 0; c.pyrOff = 0; c.pyrFrameBytes = 0;      // (set by layoutArenas)
                { const int nq = fastItemsPerRow(c.roiW - 6); c.itemRecip = nq > 0 ? (65536 + nq - 1) / nq : 0; }
                c.cellId = i * L.nCols + j;
                c.segOff = (int)cap;     // cells own consecutive, exactly sized segments in the reference's loop order
                if (c.roiW < 7 || c.roiH < 7) continue;   // cv::FAST tests nothing on such an ROI
                g.cells.push_back(c);
                if (c.roiW > g.maxRoiW) g.maxRoiW = c.roiW;
                if (c.roiH > g.maxRoiH) g.maxRoiH = c.roiH;
                // strict 3x3 NMS: no two 8-adjacent survivors inside one cell
                cap += (long long)((c.roiW - 6 + 1) / 2) * ((c.roiH - 6 + 1) / 2);
            }
        }
        L.cellCount = (int)g.cells.size() - L.cellFirst;
        if (cap >= (1LL << 24)) return "image too large: a pyramid level with 2^24 or more candidate slots (the quad-tree's best keys hold 24-bit slots)";
        L.candCap = (int)cap;
        int nodes = L.quota + 3 > 4 * L.nIni ? L.quota + 3 : 4 * L.nIni;
        L.selCap = (nodes + 2) & ~1;   // even, so the two keypoints a wave of k_describe owns always share a level
        if (nodes + 1 > g.maxNodes) g.maxNodes = nodes + 1;
        L.selOff = g.selPerFrame;
        g.selPerFrame += L.selCap;
        g.candPerFrame += L.candCap;
        g.sumPixels += (long long)L.w * L.h;
        if (l > 0) {
            resizeAxis(g.lv[l - 1].w, L.w, true, g.rx[l]);
            resizeAxis(g.lv[l - 1].h, L.h, false, g.ry[l]);
        }
        // tiles of the bordered level: 64 dword columns x kResizeTileRows rows; their source footprints
        {
            const int nd = (kPadL - kEdge + L.w + 2 * kEdge + 3) / 4, wB = L.w + 2 * kEdge;
            g.tilesX[l] = (nd + 63) / 64;
            g.tilesY[l] = (L.pyrRows + kResizeTileRows - 1) / kResizeTileRows;
            auto refl = [](int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * (n - 1) - p : p; };
            if (l > 0) {
                bool packed = true;      // k_resize's packed path gathers a dword column's taps from one 8-byte window
                for (int dw = 0; dw < nd; dw++) {
                    int lo = 1 << 30, hi = -1;
                    for (int j = 0; j < 4; j++) {
                        int bx = 4 * dw + j - (kPadL - kEdge);
                        bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                        const ResizeX& c = g.rx[l][refl(bx - kEdge, L.w)];
                        lo = std::min(lo, (int)std::min(c.sx0, c.sx1)); hi = std::max(hi, (int)std::max(c.sx0, c.sx1));
                    }
                    if (hi - lo > 7) packed = false;
                }
                g.packedTaps[l] = packed;
                // the tile resize's per-thread setup, once per level instead of once per thread and tile (four table look-ups with their clamps
                // and reflections, the window's start, selectors and weight pairs: ~50 of a thread's ~800 vector instructions): QuadRec of the
                // bordered level's dword column dw, the window's first byte kept as an ABSOLUTE source column in pad[0] (tiles subtract their own
                // footprint origin, a multiple of 4)
                g.xq[l].clear();
                if (packed)
                    for (int dw = 0; dw < nd; dw++) {
                        QuadRec q{};
                        int c0[4], c1[4], lo = 1 << 30;
                        for (int j = 0; j < 4; j++) {
                            int bx = 4 * dw + j - (kPadL - kEdge);
                            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                            const ResizeX& c = g.rx[l][refl(bx - kEdge, L.w)];
                            c0[j] = c.sx0; c1[j] = c.sx1;
                            q.wt[j] = (unsigned)(unsigned short)c.a0 | ((unsigned)(unsigned short)c.a1 << 16);
                            lo = std::min(lo, std::min(c0[j], c1[j]));
                        }
                        for (int j = 0; j < 4; j++) q.sel[j] = 0x0C000C00u | (unsigned)(c0[j] - lo) | ((unsigned)(c1[j] - lo) << 16);
                        q.pad[0] = lo;
                        g.xq[l].push_back(q);
                    }
                // tap footprints per tile column / tile row (they factor: x taps depend on tx only, y taps on ty only)
                std::vector<int> fx0(g.tilesX[l]), fx1(g.tilesX[l]), fy0(g.tilesY[l]), fy1(g.tilesY[l]);
                for (int tx = 0; tx < g.tilesX[l]; tx++) {
                    int sx0 = 1 << 30, sx1 = -1;
                    for (int dw = tx * 64; dw < tx * 64 + 64; dw++) {
                        const int bc0 = 4 * (dw < nd ? dw : nd - 1);
                        for (int j = 0; j < 4; j++) {
                            int bx = bc0 + j - (kPadL - kEdge);
                            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                            const ResizeX& c = g.rx[l][refl(bx - kEdge, L.w)];
                            sx0 = c.sx0 < sx0 ? c.sx0 : sx0; sx1 = c.sx1 > sx1 ? c.sx1 : sx1;
                        }
                    }
                    fx0[tx] = sx0; fx1[tx] = sx1;
                }
                for (int ty = 0; ty < g.tilesY[l]; ty++) {
                    int sy0 = 1 << 30, sy1 = -1;
                    for (int by = ty * kResizeTileRows; by < (ty + 1) * kResizeTileRows; by++) {
                        const int b = by < L.pyrRows ? by : L.pyrRows - 1;
                        const ResizeX& c = g.ry[l][refl(b - kEdge, L.h)];
                        sy0 = c.sx0 < sy0 ? c.sx0 : sy0; sy1 = c.sx1 > sy1 ? c.sx1 : sy1;
                    }
                    fy0[ty] = sy0; fy1[ty] = sy1;
                }
                for (int ty = 0; ty < g.tilesY[l]; ty++)
                    for (int tx = 0; tx < g.tilesX[l]; tx++) {
                        TileFoot t{};
                        int x0 = fx0[tx] & ~3, x1 = fx1[tx] + 1, y0 = fy0[ty], y1 = fy1[ty] + 1;      // [x0, x1) x [y0, y1)
                        const LevelGeom& S = g.lv[l - 1];
                        // the bordered source holds columns -19 .. w+18 (dword-aligned reads: -16 .. ((w+19) & ~3) - 1) and rows -19 .. h+18
                        x0 = std::max(x0, -16); x1 = std::min(x1, (S.w + kEdge) & ~3);
                        y0 = std::max(y0, -kEdge); y1 = std::min(y1, S.h + kEdge);
                        t.fx0 = (short)x0; t.nDw = (short)((x1 - x0 + 3) >> 2);
                        t.fy0 = (short)y0; t.nRows = (short)(y1 - y0);
                        g.foot[l].push_back(t);
                        if (t.nDw * 4 > g.tileLdsStride) g.tileLdsStride = (t.nDw * 4 + 15) / 16 * 16;
                        if (t.nRows > g.tileLdsRows) g.tileLdsRows = t.nRows;
                    }
            }
        }
    }
    // ---- region-major pyramid (k_pyr_cols, orbx_device.hpp: PyrColumn): RX x RY regions; per region and level the owned part of the bordered
    //      level (the regions' parts partition its dword columns and rows) and the interior rectangle held in LDS, coarsest level first:
    //      what the owned bytes show (clamped and reflected exactly as the kernel does) joined with the taps of the next level's rectangle ----
    if (t.nlevels >= 2) {
        auto refl = [](int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * (n - 1) - p : p; };
        const int top = t.nlevels - 1;
        auto buildCols = [&](const int px, const bool evenCut) {
        FrameGeom::ColumnSet cs;
        const int RX = std::max(1, (cols + px / 2) / px), RY = std::max(1, (rows + px / 2) / px);
        cs.px = px; cs.RX = RX; cs.RY = RY;
        const int edgeNum = px <= 56 && !evenCut ? kColEdgeNum : 1, edgeDen = px <= 56 && !evenCut ? kColEdgeDen : 1;
        bool fits = true;
        int maxEven = 0, maxOdd = 0;
        for (int cy = 0; cy < RY; cy++)
            for (int cx = 0; cx < RX; cx++) {
                PyrColumn c{};
                int x0 = 0, x1 = -1, y0 = 0, y1 = -1;      // region of the level above the one in hand
                int coefs = 0;
                for (int l = top; l >= 0; l--) {
                    const LevelGeom& L = g.lv[l];
                    const int nd = (kPadL - kEdge + L.w + 2 * kEdge + 3) / 4, wB = L.w + 2 * kEdge;
                    // the INTERIOR is cut evenly (x cuts at multiples of 4: dword boundaries of the bordered rows), so that a region's parts of
                    // successive levels lie above each other; the border frame goes to the outer regions
                    // (the fine cuts - the launch of a few frames is as long as its slowest workgroup - give the OUTER regions less of the interior:
                    // they also write the 19-px frame and hold its mirrored pixels down to the coarsest level; edgeNum / edgeDen of an inner
                    // region's share.  One 640x480 frame 37.3 -> 36.6-36.7 us with 3/5; 1/2, 2/5 and 4/5 within 0.3 us of that - the corner regions'
                    // deep levels are set by the 19 mirrored pixels, not by their share, so this is all the cut can give)
                    auto frac = [&](int i, int R, long long extent) {
                        if (R < 3 || edgeNum == edgeDen) return (long long)i * extent / R;
                        return ((long long)edgeNum + (long long)(i - 1) * edgeDen) * extent / (2LL * edgeNum + (long long)(R - 2) * edgeDen);
                    };
                    auto cutX = [&](int i) { return i <= 0 ? 0 : (i >= RX ? nd : (kPadL + (int)(frac(i, RX, L.w) & ~3LL)) / 4); };
                    auto cutY = [&](int i) { return i <= 0 ? 0 : (i >= RY ? L.pyrRows : kEdge + (int)frac(i, RY, L.h)); };
                    const int dwA = cutX(cx), dwB = cutX(cx + 1), rA = cutY(cy), rB = cutY(cy + 1);
                    c.own[l] = ColOwn{(short)dwA, (short)dwB, (short)rA, (short)rB};
                    int nx0 = 1 << 30, nx1 = -1, ny0 = 1 << 30, ny1 = -1;
                    if (dwB > dwA && rB > rA) {
                        for (int b = 4 * dwA; b < 4 * dwB; b++) {
                            int bx = b - (kPadL - kEdge);
                            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                            const int v = refl(bx - kEdge, L.w);
                            nx0 = std::min(nx0, v); nx1 = std::max(nx1, v);
                        }
                        for (int r = rA; r < rB; r++) {
                            const int v = refl(r - kEdge, L.h);
                            ny0 = std::min(ny0, v); ny1 = std::max(ny1, v);
                        }
                    }
                    if (l < top && x1 >= x0 && y1 >= y0) {      // taps of the rectangle of level l + 1
                        const std::vector<ResizeX>& X = g.rx[l + 1];
                        const std::vector<ResizeX>& Y = g.ry[l + 1];
                        for (int xv = x0; xv <= x1; xv++) {
                            const int x = refl(xv, g.lv[l + 1].w);
                            nx0 = std::min(nx0, (int)std::min(X[x].sx0, X[x].sx1));
                            nx1 = std::max(nx1, (int)std::max(X[x].sx0, X[x].sx1));
                        }
                        for (int y = y0; y <= y1; y++) { ny0 = std::min(ny0, (int)std::min(Y[y].sx0, Y[y].sx1)); ny1 = std::max(ny1, (int)std::max(Y[y].sx0, Y[y].sx1)); }
                    }
                    if (nx1 < nx0 || ny1 < ny0) { nx0 = nx1 = 0; ny0 = ny1 = 0; }      // (a region that owns nothing of this level and feeds nothing)
                    nx0 &= ~3;                                // aligned dwords of the owned bytes are aligned dwords of the LDS rectangle
                    int w = nx1 - nx0 + 1;
                    if (l == 0) w = (w + 3) & ~3;             // the loaded rectangle: whole dwords (the kernel never reads past an image row)
                    const int hh = ny1 - ny0 + 1, bytes = ((w + 3) & ~3) * hh;
                    c.region[l] = ChainRegion{(short)nx0, (short)ny0, (short)w, (short)hh};
                    c.deal[0][l] = makeChainDeal(w, hh, 256); c.deal[1][l] = makeChainDeal(w, hh, 512);
                    if (w > kChainMaxW || (l == 0 && bytes > kChainMaxW * kChainMaxH0)) fits = false;
                    if (l & 1) maxOdd = std::max(maxOdd, bytes); else maxEven = std::max(maxEven, bytes);
                    if (l >= 1) coefs += 6 * ((w + 3) >> 2) + 2 * hh;      // 8-byte units: quad records (six each), row records (two each)
                    x0 = nx0; x1 = std::min(nx0 + w - 1, L.w - 1); y0 = ny0; y1 = ny1;
                    if (l == 0) x1 = nx1;
                }
                if (coefs > kChainCoefMax) fits = false;
                c.nCoef = coefs;
                if (coefs > cs.coefSlot) cs.coefSlot = coefs;
                cs.columns.push_back(c);
            }
        cs.coefSlot = (cs.coefSlot + 7) & ~7;
        if (fits) {
            cs.coef.assign(cs.columns.size() * (size_t)cs.coefSlot, ResizeX{0, 0, 0, 0});
            for (size_t i = 0; i < cs.columns.size(); i++) {
                ResizeX* q = cs.coef.data() + i * (size_t)cs.coefSlot;
                for (int l = 1; l <= top; l++) {
                    const ChainRegion& r = cs.columns[i].region[l];
                    const ChainRegion& rs = cs.columns[i].region[l - 1];
                    for (int x4 = 0; x4 < r.w; x4 += 4) {      // (chainStep's arithmetic, once per region and level instead of once per thread and step)
                        QuadRec qr{};
                        int c0[4], c1[4], lo = 1 << 30;
                        for (int k = 0; k < 4; k++) {
                            const ResizeX& cx = g.rx[l][refl(r.x0 + std::min(x4 + k, r.w - 1), g.lv[l].w)];
                            c0[k] = cx.sx0 - rs.x0; c1[k] = cx.sx1 - rs.x0;
                            qr.wt[k] = (unsigned)(unsigned short)cx.a0 | ((unsigned)(unsigned short)cx.a1 << 16);
                            lo = std::min(lo, std::min(c0[k], c1[k]));
                        }
                        for (int k = 0; k < 4; k++) {
                            if (c0[k] - lo > 7 || c1[k] - lo > 7) cs.fit = false, fits = false;      // (a mirrored quad whose taps leave the packed step's 8-byte window)
                            qr.sel[k] = 0x0C000C00u | (unsigned)(c0[k] - lo) | ((unsigned)(c1[k] - lo) << 16);
                        }
                        qr.baseSh = (lo & ~3) | ((lo & 3) << 16);
                        std::memcpy(q, &qr, sizeof(qr));
                        q += sizeof(qr) / sizeof(ResizeX);
                    }
                    for (int y = 0; y < r.h; y++) {
                        const RowRec rr = makeRowRec(g.ry[l][r.y0 + y]);
                        std::memcpy(q, &rr, sizeof(rr));
                        q += sizeof(rr) / sizeof(ResizeX);
                    }
                }
            }
        }
        cs.evenBytes = (maxEven + 15) & ~15;
        cs.ldsBytes = cs.evenBytes + ((maxOdd + 15) & ~15) + 32;
        cs.fit = fits;
        return cs;
        };
        // (a fine cut whose narrower outer regions make an inner region's records outgrow the kernel's staging - many levels at a small scale
        // factor - is rebuilt evenly)
        auto buildFit = [&](const int px) {
            FrameGeom::ColumnSet cs = buildCols(px, false);
            if (!cs.fit && px <= 56 && kColEdgeNum != kColEdgeDen) cs = buildCols(px, true);
            return cs;
        };
        if (colPx > 0) g.colSets.push_back(buildFit(colPx));
        else for (int px : kColPx) g.colSets.push_back(buildFit(px));
        bool packed = true;      // the packed horizontal pass: the 8 taps of any four adjacent columns (region columns start anywhere) within 8 source bytes
        for (int l = 1; l <= top && packed; l++) {
            const std::vector<ResizeX>& X = g.rx[l];
            const int w = g.lv[l].w;
            for (int x = 0; x < w; x++) {
                int lo = 1 << 30, hi = -1;
                for (int k = 0; k < 4; k++) {
                    const ResizeX& cc = X[std::min(x + k, w - 1)];
                    lo = std::min(lo, (int)std::min(cc.sx0, cc.sx1)); hi = std::max(hi, (int)std::max(cc.sx0, cc.sx1));
                }
                if (hi - lo > 7) { packed = false; break; }
            }
        }
        g.colsPacked = packed;
    }
    return std::string();
}

// Quad-tree node-array sizing of a handle (from its LARGEST geometry): M nodes (multiple of 8), P = next power of two (bitonic sort),
// R roots / XT coordinates covered by the dense phase (0 = no dense phase), arena = the arrays exceed a CU's LDS and live in HBM
// (per-level quotas in the thousands: the reference builds its initialisation extractor with 5 * nFeatures, Tracking.cc:774).
// ldsBytes = octreeLdsBytes of k_octree.hip.  Shared by orbx_create and the host emulation of the kernel (tools/octree_emu).
struct OctSizing { int M = 0, P = 0, R = 0, XT = 0; bool arena = false; size_t arenaSlice = 0; const char* err = nullptr; };
inline OctSizing octreeSizing(const FrameGeom& mg, int nlevels, size_t (*ldsBytes)(int, int, int, int)) {
    OctSizing z;
    int M = mg.maxNodes + 8;   // +8: tall/narrow sub-images may add a root
    M = (M + 7) / 8 * 8;
    z.M = M;
    z.P = 1;
    while (z.P < M) z.P <<= 1;
    // dense phase: count pyramids for the roots of the widest level and coordinate tables for the largest rectangle; dropped
    // (the kernel then sweeps the keys every pass) when LDS is short
    z.R = 1;
    for (int l = 0; l < nlevels; l++) z.R = mg.lv[l].nIni > z.R ? mg.lv[l].nIni : z.R;
    z.XT = ((mg.lv[0].rectW > mg.lv[0].rectH ? mg.lv[0].rectW : mg.lv[0].rectH) + 15) / 16 * 16;
    const int denseR = z.R <= 7 ? z.R : 0, denseXT = z.R <= 7 ? z.XT : 0;
    if (z.R > 7 || ldsBytes(z.M, z.P, z.R, z.XT) > 158 * 1024) { z.R = 0; z.XT = 0; }
    if (ldsBytes(z.M, z.P, z.R, z.XT) > 158 * 1024) {
        if (z.M > 65535) { z.err = "more than 65535 quad-tree nodes per level"; return z; }
        z.arena = true;
        z.R = denseR; z.XT = denseXT;
        z.arenaSlice = (ldsBytes(z.M, z.P, z.R, z.XT) + 255) & ~(size_t)255;
    }
    return z;
}

// Level-major arenas: all frames of level 0, then all frames of level 1, ...
inline void layoutArenas(FrameGeom& g, int maxBatch) {
    long long pyr = 0, blur = 0, cand = 0;
    for (int l = 0; l < g.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        L.pyrFrameBytes = (long long)L.pyrStride * L.pyrRows;
        L.pyrOff = pyr;
        pyr += L.pyrFrameBytes * maxBatch;
        L.blurFrameBytes = (long long)L.blurStride * L.h;
        L.blurOff = blur;
        blur += L.blurFrameBytes * maxBatch;
        L.candOff = cand;
        cand += (long long)L.candCap * maxBatch;
    }
    g.pyrBytesPerFrame = pyr / maxBatch;
    g.blurBytesPerFrame = blur / maxBatch;
    for (CellDesc& c : g.cells) {      // the cells carry their level's place in the pyramid arena (orbx_device.hpp: CellDesc)
        const LevelGeom& L = g.lv[c.level];
        c.pyrStride = L.pyrStride; c.pyrOff = L.pyrOff; c.pyrFrameBytes = L.pyrFrameBytes;
    }
}

}  // namespace orbx
