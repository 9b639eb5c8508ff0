// Host-side check of the region-major pyramid's tables (orbx_geometry.hpp: FrameGeom::colSets, what k_pyr_cols runs from), no GPU:
//  1. the regions' `own` rectangles partition every bordered level (each dword column x row exactly once);
//  2. a scalar replay of the kernel's data flow — per region: load region[0] of the image, resize region[l + 1] out of region[l] with the
//     level's coefficient records taken from the region's own list (cv::resize 8u bilinear arithmetic, ORBextractor.cc:1183-1197), write the
//     owned bytes of each level with the clamp / REFLECT_101 mapping of copyMakeBorder (:1213-1215) — assembles exactly the pyramid that a
//     plain level-by-level resize of whole levels with the same tables gives, and never reads outside a rectangle it holds;
//  3. the limits the kernel relies on (x0 % 4 == 0, widths, coefficient count, LDS bytes);
//  4. the WRITING role's thread dealing as the kernel computes it (k_pyramid.hip: doLevel) - the interior dword columns dealt dword by dword
//     or walked column-wise in three runs (top mirror, interior, bottom mirror), the frame's columns byte by byte, all through float
//     reciprocals - replayed for every writer count the launch shapes use (256 / 512 / 768 / 1024 threads): every owned (row, dword) is
//     written by exactly one thread, from the source row copyMakeBorder's REFLECT_101 names (round 5; docs/history/DESIGN_rounds_1-5.md §4j: the one miscompare of
//     a soak was a bordered level 0 of 1014 x 432, w % 4 == 2, whose cause was never found - this closes the host-side candidates);
//  5. a hash of every table the kernel reads (printed; tests/test_pyramid_columns.py runs the checker under two MALLOC_PERTURB_ values and
//     a dirtied heap and expects the same hash: no uninitialised byte reaches the tables).
// usage: pyr_columns_check <cols> <rows> <nlevels> <scaleFactor>      prints "ok ..." and exits 0, or the first violation and exits 1
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#include <vector>
#include <string>

#include "orbx_geometry.hpp"

using namespace orbx;

static int refl(int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * (n - 1) - p : p; }
static unsigned resizePx(int p00, int p01, int p10, int p11, const ResizeX& cx, const ResizeX& cy) {
    const int h0 = p00 * cx.a0 + p01 * cx.a1, h1 = p10 * cx.a0 + p11 * cx.a1;
    return (unsigned)((((cy.a0 * (h0 >> 4)) >> 16) + ((cy.a1 * (h1 >> 4)) >> 16) + 2) >> 2);
}
#define FAIL(...) do { printf(__VA_ARGS__); printf("\n"); return 1; } while (0)

// FNV-1a over the bytes of the tables the kernels read
static unsigned long long g_hash = 1469598103934665603ull;
static void hashBytes(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { g_hash ^= b[i]; g_hash *= 1099511628211ull; } }
template <class T> static void hashVec(const std::vector<T>& v) { if (!v.empty()) hashBytes(v.data(), v.size() * sizeof(T)); }

// The writing role of one level of one region exactly as k_pyr_cols deals it over `wstep` threads (k_pyramid.hip: doLevel; the same float
// reciprocals: __frcp_rn(x) is the correctly rounded 1.0f / x).  count[(row - o.r0) * (o.dw1 - o.dw0) + (dw - o.dw0)]++ per store; a store
// outside the owned rectangle, or a column-walk store from a source row other than REFLECT_101's, returns a message.
static const char* replayWriters(const ColOwn own, int w, int h, int wstep, std::vector<int>& count) {
    const int nrows = own.r1 - own.r0, ncols = own.dw1 - own.dw0;
    count.assign((size_t)std::max(nrows, 0) * std::max(ncols, 0), 0);
    if (nrows <= 0 || ncols <= 0) return nullptr;
    auto store = [&](int row, int dw) -> bool {
        if (row < own.r0 || row >= own.r1 || dw < own.dw0 || dw >= own.dw1) return false;
        count[(size_t)(row - own.r0) * ncols + (dw - own.dw0)]++;
        return true;
    };
    const int fa = std::max((int)own.dw0, kPadL / 4), fb = std::max(fa, std::min((int)own.dw1, kPadL / 4 + (w >> 2)));
    for (int wtid = 0; wtid < wstep; wtid++) {
        if (fb > fa && (fb - fa) * nrows <= 2 * wstep) {
            const int ndw = fb - fa, total = ndw * nrows;
            const float inv = 1.0f / (float)ndw;
            for (int i = wtid; i < total; i += wstep) {
                const int rr = (int)(((float)i + 0.5f) * inv), dw = fa + (i - rr * ndw), row = own.r0 + rr;
                if (!store(row, dw)) return "dword-by-dword store outside the owned rectangle";
            }
        } else if (fb > fa) {
            const int ndw = fb - fa;
            const float rdw = 1.0f / (float)ndw;
            const int ngrp = std::max((int)(((float)wstep + 0.5f) * rdw), 1);
            const int per = (int)(((float)(nrows + ngrp - 1) + 0.5f) * (1.0f / (float)ngrp));
            const int grp = (int)(((float)wtid + 0.5f) * rdw), c = wtid - grp * ndw;
            if (grp < ngrp) {
                if (c < 0 || c >= ndw) return "column-walk: column outside the interior columns";
                const int rA = own.r0 + grp * per, rB = std::min(rA + per, (int)own.r1);
                auto run = [&](int r0, int r1, int src0, int sstep) -> const char* {
                    int src = src0;
                    for (int row = r0; row < r1; row++, src += sstep) {
                        if (src != refl(row - kEdge, h)) return "column-walk: source row differs from REFLECT_101";
                        if (!store(row, fa + c)) return "column-walk store outside the owned rectangle";
                    }
                    return nullptr;
                };
                const int t1 = std::min(rB, kEdge), m0 = std::max(rA, kEdge), m1 = std::min(rB, kEdge + h), b0 = std::max(rA, kEdge + h);
                const char* e = nullptr;
                if (rA < t1 && (e = run(rA, t1, kEdge - rA, -1))) return e;
                if (m0 < m1 && (e = run(m0, m1, m0 - kEdge, 1))) return e;
                if (b0 < rB && (e = run(b0, rB, 2 * (h - 1) - (b0 - kEdge), -1))) return e;
            }
        }
        const int nL = fa - own.dw0, nO = nL + (own.dw1 - fb);
        if (nO > 0) {
            const int total = nO * nrows;
            const float inv = 1.0f / (float)nO;
            for (int i = wtid; i < total; i += wstep) {
                const int rr = (int)(((float)i + 0.5f) * inv), jj = i - rr * nO, dw = jj < nL ? own.dw0 + jj : fb + (jj - nL), row = own.r0 + rr;
                if (!store(row, dw)) return "frame-column store outside the owned rectangle";
            }
        }
    }
    return nullptr;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int cols = atoi(argv[1]), rows = atoi(argv[2]), nlevels = atoi(argv[3]);
    const float sf = (float)atof(argv[4]);
    ScaleTables t = makeScaleTables(1000, sf, nlevels);
    FrameGeom g;
    const std::string why = makeFrameGeom(t, rows, cols, g, 0);
    if (!why.empty()) { printf("rejected: %s\n", why.c_str()); return 0; }
    layoutArenas(g, 1);
    hashBytes(g.lv, sizeof(LevelGeom) * g.nlevels);
    hashVec(g.cells);
    for (int l = 0; l < nlevels; l++) { hashVec(g.rx[l]); hashVec(g.ry[l]); hashVec(g.foot[l]); hashVec(g.xq[l]); }
    for (const FrameGeom::ColumnSet& cs : g.colSets) { hashVec(cs.columns); hashVec(cs.coef); }
    // the image and the whole levels, resized level by level
    std::vector<std::vector<uint8_t>> lvl(nlevels);
    lvl[0].resize((size_t)cols * rows);
    unsigned seed = 12345u;
    for (auto& p : lvl[0]) { seed = seed * 1664525u + 1013904223u; p = (uint8_t)(seed >> 24); }
    for (int l = 1; l < nlevels; l++) {
        const int w = g.lv[l].w, h = g.lv[l].h, sw = g.lv[l - 1].w;
        lvl[l].resize((size_t)w * h);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const ResizeX cx = g.rx[l][x], cy = g.ry[l][y];
                const uint8_t* s = lvl[l - 1].data();
                lvl[l][(size_t)y * w + x] = (uint8_t)resizePx(s[(size_t)cy.sx0 * sw + cx.sx0], s[(size_t)cy.sx0 * sw + cx.sx1], s[(size_t)cy.sx1 * sw + cx.sx0], s[(size_t)cy.sx1 * sw + cx.sx1], cx, cy);
            }
    }
    // the FAST cells' item reciprocal (CellDesc::itemRecip): k_fast deals lanes, k_fast_wide threads, to (row, item) with it
    for (const CellDesc& c : g.cells) {
        const int nq = fastItemsPerRow(c.roiW - 6);
        if (nq < 1 || nq > 16) FAIL("cell %d of level %d: %d items per row", c.cellId, c.level, nq);
        for (int x = 0; x <= 256; x++)
            if ((x * c.itemRecip) >> 16 != x / nq) FAIL("cell %d of level %d: itemRecip %d gives %d / %d wrong", c.cellId, c.level, c.itemRecip, x, nq);
    }
    // the tile resize's dword-column records (FrameGeom::xq): taps and weights of the four bytes as the level tables have them
    for (int l = 1; l < nlevels; l++) {
        if (!g.packedTaps[l]) { if (!g.xq[l].empty()) FAIL("level %d: column records without packed taps", l); continue; }
        const int nd = (kPadL - kEdge + g.lv[l].w + 2 * kEdge + 3) / 4, wB = g.lv[l].w + 2 * kEdge;
        if ((int)g.xq[l].size() != nd) FAIL("level %d: %zu column records for %d dword columns", l, g.xq[l].size(), nd);
        for (int dw = 0; dw < nd; dw++)
            for (int j = 0; j < 4; j++) {
                int bx = 4 * dw + j - (kPadL - kEdge);
                bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                const ResizeX want = g.rx[l][refl(bx - kEdge, g.lv[l].w)];
                const QuadRec& q = g.xq[l][dw];
                const int t0 = (int)(q.sel[j] & 0xff), t1 = (int)((q.sel[j] >> 16) & 0xff);
                if ((q.sel[j] & 0xff00ff00u) != 0x0C000C00u || t0 > 7 || t1 > 7 || q.pad[0] + t0 != want.sx0 || q.pad[0] + t1 != want.sx1 ||
                    (short)(q.wt[j] & 0xffff) != want.a0 || (short)(q.wt[j] >> 16) != want.a1)
                    FAIL("level %d dword column %d byte %d: column record differs from the level table", l, dw, j);
            }
    }
    int checked = 0;
    for (const FrameGeom::ColumnSet& cs : g.colSets) {
        if (!cs.fit || !g.colsPacked) continue;      // (the kernel's steps are the packed ones: without packed taps the host never takes this form)
        if ((int)cs.columns.size() != cs.RX * cs.RY) FAIL("px %d: %zu regions for a %d x %d cut", cs.px, cs.columns.size(), cs.RX, cs.RY);
        if (cs.ldsBytes > 64 * 1024) FAIL("px %d: %d bytes of LDS", cs.px, cs.ldsBytes);
        std::vector<std::vector<int>> written(nlevels);      // per level: how often each (row, dword) was written
        std::vector<std::vector<uint8_t>> out(nlevels);
        std::vector<int> nd(nlevels);
        for (int l = 0; l < nlevels; l++) {
            nd[l] = (kPadL - kEdge + g.lv[l].w + 2 * kEdge + 3) / 4;
            written[l].assign((size_t)nd[l] * g.lv[l].pyrRows, 0);
            out[l].assign((size_t)nd[l] * 4 * g.lv[l].pyrRows, 0);
        }
        for (size_t ci = 0; ci < cs.columns.size(); ci++) {
            const PyrColumn& c = cs.columns[ci];
            const ResizeX* coef = cs.coef.data() + ci * (size_t)cs.coefSlot;
            int off = 0, total = 0;      // 8-byte units: per level the quad records (six units each), then the row records (two units each)
            for (int l = 1; l < nlevels; l++) total += 6 * ((c.region[l].w + 3) / 4) + 2 * c.region[l].h;
            if (total != c.nCoef || total > kChainCoefMax || total > cs.coefSlot) FAIL("px %d region %zu: %d coefficient records, nCoef %d, slot %d", cs.px, ci, total, c.nCoef, cs.coefSlot);
            // the host-made dealing of the deriving threads (ChainDeal): the reciprocal gives tid / quads exactly for every thread of the largest
            // workgroup, the row blocks times the quads fit the role's threads, and the blocks cover the rectangle's rows
            for (int l = 1; l < nlevels; l++)
                for (int v = 0; v < 2; v++) {
                    const ChainDeal d = c.deal[v][l];
                    const int TD = v ? 512 : 256, nq = (c.region[l].w + 3) / 4;
                    if (d.nb != TD / nq || d.nb < 1 || d.nb * nq > TD || (int)d.per * d.nb < c.region[l].h || ((int)d.per - 1) * d.nb >= c.region[l].h)
                        FAIL("px %d region %zu level %d: dealing for %d threads: %d blocks of %d rows over %d quads x %d rows", cs.px, ci, l, TD, d.nb, d.per, nq, c.region[l].h);
                    for (int tid = 0; tid < 1024; tid++)
                        if ((int)(((unsigned)tid * d.recip) >> 20) != tid / nq) FAIL("px %d region %zu level %d: reciprocal %u gives tid %d / %d wrong", cs.px, ci, l, d.recip, tid, nq);
                }
            std::vector<uint8_t> cur, nxt;
            for (int l = 0; l < nlevels; l++) {
                const ChainRegion r = c.region[l];
                const int w = g.lv[l].w, h = g.lv[l].h;
                if (r.x0 & 3) FAIL("px %d region %zu level %d: x0 %d not a multiple of 4", cs.px, ci, l, r.x0);
                if (r.w < 1 || r.h < 1 || r.w > kChainMaxW || r.x0 < 0 || r.y0 < 0 || r.y0 + r.h > h || (l > 0 && r.x0 + r.w > w)) FAIL("px %d region %zu level %d: rectangle %d,%d %dx%d outside the %dx%d level", cs.px, ci, l, r.x0, r.y0, r.w, r.h, w, h);
                const int stride = (r.w + 3) & ~3;
                if ((l & 1 ? cs.ldsBytes - cs.evenBytes - 32 : cs.evenBytes) < stride * r.h) FAIL("px %d region %zu level %d: %d bytes do not fit its LDS buffer", cs.px, ci, l, stride * r.h);
                if (l == 0) {
                    if (r.w & 3) FAIL("px %d region %zu: loaded width %d not whole dwords", cs.px, ci, r.w);
                    if (stride * r.h > kChainMaxW * kChainMaxH0) FAIL("px %d region %zu: %d loaded bytes", cs.px, ci, stride * r.h);
                    cur.assign((size_t)stride * r.h, 0xEE);
                    for (int y = 0; y < r.h; y++)
                        for (int x = 0; x < r.w; x++)      // (the kernel never reads outside an image row: a virtual column is the column it mirrors)
                            cur[(size_t)y * stride + x] = lvl[0][(size_t)(r.y0 + y) * w + refl(r.x0 + x, w)];
                }
                // the owned bytes of this level out of `cur`
                const ColOwn o = c.own[l];
                if (o.dw0 < 0 || o.dw1 > nd[l] || o.r0 < 0 || o.r1 > g.lv[l].pyrRows || o.dw1 < o.dw0 || o.r1 < o.r0) FAIL("px %d region %zu level %d: own rectangle", cs.px, ci, l);
                const int wB = w + 2 * kEdge;
                for (int wstep : {256, 512, 768, 1024}) {      // the writing role's thread counts of the launch shapes (launchPyrCols) and of the last level (everyone)
                    std::vector<int> cnt;
                    if (const char* e = replayWriters(o, w, h, wstep, cnt)) FAIL("px %d region %zu level %d, %d writers: %s", cs.px, ci, l, wstep, e);
                    for (size_t i = 0; i < cnt.size(); i++)
                        if (cnt[i] != 1) FAIL("px %d region %zu level %d, %d writers: owned dword (%d, %d) stored %d times by the kernel's dealing", cs.px, ci, l, wstep,
                                              o.r0 + (int)(i / (size_t)(o.dw1 - o.dw0)), o.dw0 + (int)(i % (size_t)(o.dw1 - o.dw0)), cnt[i]);
                }
                for (int row = o.r0; row < o.r1; row++)
                    for (int dw = o.dw0; dw < o.dw1; dw++) {
                        written[l][(size_t)row * nd[l] + dw]++;
                        const int iy = refl(row - kEdge, h);
                        for (int k = 0; k < 4; k++) {
                            int bx = 4 * dw + k - (kPadL - kEdge);
                            bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                            const int ix = refl(bx - kEdge, w);
                            if (ix < r.x0 || ix >= r.x0 + r.w || iy < r.y0 || iy >= r.y0 + r.h) FAIL("px %d region %zu level %d: owned byte (%d, %d) shows pixel (%d, %d) outside the held rectangle", cs.px, ci, l, row, 4 * dw + k, ix, iy);
                            out[l][((size_t)row * nd[l] + dw) * 4 + k] = cur[(size_t)(iy - r.y0) * stride + (ix - r.x0)];
                        }
                    }
                // the next level's rectangle out of `cur`, with the region's own coefficient list
                if (l + 1 < nlevels) {
                    const ChainRegion d = c.region[l + 1];
                    const int ds = (d.w + 3) & ~3;
                    nxt.assign((size_t)ds * d.h, 0xEE);
                    const int nq = (d.w + 3) / 4;
                    const QuadRec* qrs = (const QuadRec*)(coef + off);
                    const RowRec* rrs = (const RowRec*)(coef + off + 6 * nq);
                    for (int y = 0; y < d.h; y++)
                        for (int x = 0; x < d.w; x++) {
                            // the column's taps and weights as the kernel reads them out of its quad record: tap = row start + base + shift + selector byte
                            const QuadRec& qr = qrs[x / 4];
                            const int k = x & 3, base = qr.baseSh & 0xffff, sh = qr.baseSh >> 16;
                            if ((base & 3) || sh > 3 || (qr.sel[k] & 0xff00ff00u) != 0x0C000C00u) FAIL("px %d region %zu level %d: malformed quad record %d", cs.px, ci, l + 1, x / 4);
                            const int t0 = (int)(qr.sel[k] & 0xff), t1 = (int)((qr.sel[k] >> 16) & 0xff);
                            if (t0 > 7 || t1 > 7) FAIL("px %d region %zu level %d: a tap of column %d lies outside the 8-byte window", cs.px, ci, l + 1, x);
                            ResizeX cx;
                            cx.sx0 = (short)(r.x0 + base + sh + t0); cx.sx1 = (short)(r.x0 + base + sh + t1);
                            cx.a0 = (short)(qr.wt[k] & 0xffff); cx.a1 = (short)(qr.wt[k] >> 16);
                            // the row's taps back out of its bank record: bank A holds the even source row when the parities differ (either order is
                            // the same sum; the kernel adds the two products and the rounding constant as integers)
                            const RowRec rr = rrs[y];
                            ResizeX cy;
                            {
                                const ResizeX want = g.ry[l + 1][d.y0 + y];
                                const bool swapped = rr.sA != want.sx0 || (unsigned)(unsigned short)want.a0 << 12 != rr.bA;
                                cy.sx0 = (short)(swapped ? rr.sB : rr.sA); cy.sx1 = (short)(swapped ? rr.sA : rr.sB);
                                cy.a0 = (short)((swapped ? rr.bB : rr.bA) >> 12); cy.a1 = (short)((swapped ? rr.bA : rr.bB) >> 12);
                                if (((rr.sA ^ rr.sB) & 1) && (rr.sA & 1)) FAIL("px %d region %zu level %d row %d: bank A holds an odd source row although the parities differ", cs.px, ci, l + 1, y);
                            }
                            const ResizeX gx = g.rx[l + 1][refl(d.x0 + x, g.lv[l + 1].w)], gy = g.ry[l + 1][d.y0 + y];      // (a virtual column is derived with the taps of the column it mirrors)
                            if (cx.sx0 != gx.sx0 || cx.sx1 != gx.sx1 || cx.a0 != gx.a0 || cx.a1 != gx.a1 || cy.sx0 != gy.sx0 || cy.sx1 != gy.sx1 || cy.a0 != gy.a0 || cy.a1 != gy.a1)
                                FAIL("px %d region %zu level %d: coefficient list differs from the level tables at (%d, %d)", cs.px, ci, l + 1, x, y);
                            if (d.x0 + x >= g.lv[l + 1].w) continue;      // (padding columns of an aligned start are never shown)
                            const int xs[2] = {cx.sx0, cx.sx1}, ys[2] = {cy.sx0, cy.sx1};
                            int p[2][2];
                            for (int a = 0; a < 2; a++)
                                for (int b = 0; b < 2; b++) {
                                    if (xs[b] < r.x0 || xs[b] >= r.x0 + r.w || ys[a] < r.y0 || ys[a] >= r.y0 + r.h || xs[b] >= w)
                                        FAIL("px %d region %zu level %d: tap (%d, %d) of pixel (%d, %d) outside the held rectangle", cs.px, ci, l + 1, xs[b], ys[a], d.x0 + x, d.y0 + y);
                                    p[a][b] = cur[(size_t)(ys[a] - r.y0) * stride + (xs[b] - r.x0)];
                                }
                            nxt[(size_t)y * ds + x] = (uint8_t)resizePx(p[0][0], p[0][1], p[1][0], p[1][1], cx, cy);
                        }
                    off += 6 * nq + 2 * d.h;
                    cur.swap(nxt);
                }
            }
        }
        for (int l = 0; l < nlevels; l++) {
            const int w = g.lv[l].w, h = g.lv[l].h, wB = w + 2 * kEdge;
            for (int row = 0; row < g.lv[l].pyrRows; row++)
                for (int dw = 0; dw < nd[l]; dw++) {
                    if (written[l][(size_t)row * nd[l] + dw] != 1) FAIL("px %d level %d: dword (%d, %d) written %d times", cs.px, l, row, dw, written[l][(size_t)row * nd[l] + dw]);
                    for (int k = 0; k < 4; k++) {
                        int bx = 4 * dw + k - (kPadL - kEdge);
                        bx = bx < 0 ? 0 : (bx > wB - 1 ? wB - 1 : bx);
                        const uint8_t want = lvl[l][(size_t)refl(row - kEdge, h) * w + refl(bx - kEdge, w)];
                        if (out[l][((size_t)row * nd[l] + dw) * 4 + k] != want) FAIL("px %d level %d: byte (%d, %d) is %d, the level-by-level pyramid has %d", cs.px, l, row, 4 * dw + k, out[l][((size_t)row * nd[l] + dw) * 4 + k], want);
                    }
                }
        }
        checked++;
    }
    printf("ok %dx%d %d levels scale %.2f hash %016llx: %d cuts\n", cols, rows, nlevels, sf, g_hash, checked);
    return 0;
}
