// orb_oracle.cpp — CPU ORACLE for the ORB extraction hot path.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product path (extractorb_amd/, liborbx.so) never links, imports or calls it.
//
// PARITY: PINNED FOR DETECTION BY ONE REFERENCE-HELD COUNT, UNPINNED FOR THE REST.  The reference
// (/root/reference/src/orb_extractor/ORBextractor.cc) delegates all pixel arithmetic to OpenCV 3 (CMakeLists.txt:23), which is
// neither vendored in the reference nor installed in this image, and the reference ships no tests or golden vectors
// (SURVEY.md §4, §8c).  It holds one recorded result: img_folder/Screenshot.png, "ORB_SLAM3 has total 1420 keypoints", printed by
// src/orb_extractor/main_orb_extractor.cpp:34-53 (nFeatures 1500) on pic/TUM/dataset-room4_512_16/.../1520531124150444163.png.
// (An assumption rides on that: today's main constructs ORBextractor(5 * nFeatures) = 7500 features (:43), which gives 1547 on that frame, not
// 1420; the screenshot is taken to predate that line - with 1500 features the count is exact AND the 1420 positions sit on the circles the
// screenshot's image window shows.  tests/test_reference_pin.py asserts both readings.)
// This file returns exactly 1420 there (tests/test_reference_pin.py); `enum Mutation` below measures which restated semantics
// that count decides (resize rounding, level chain and sizes, FAST strictness, per-cell strict NMS, threshold retry, 30-px grid,
// the ">= N" stop rule) and which it does not.  PARITY UNPINNED for: keypoint order, GaussianBlur taps, fastAtan2, cosf/sinf, the
// rBRIEF rounding, undistortPoints and every "next"-row function — no reference-held vector exists for them.
// This file restates (a) the control flow, constants, float/int conversion points and
// container order of ORBextractor.cc, each function citing the lines it follows, and (b) the
// published OpenCV 3.4 generic-C++ semantics of the seven primitives the reference calls
// (SURVEY.md Appendix A).  The primitives are additionally checked against independent definitions in tests/
// (brute-force FAST-9 predicate/score, scipy mirror convolution for the blur, double-precision
// atan2, exhaustive glibc sinf/cosf comparison), not by a real OpenCV run.
//
// One declared divergence: the std::sort tie at ORBextractor.cc:689 is by heap address in the
// reference (allocator dependent); the oracle breaks ties by node creation order, newest first
// (SURVEY.md §8c).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off: the reference is built for baseline
// x86-64, i.e. without FMA contraction — CMakeLists.txt:4-5 has no -march flag).

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <thread>
#include <utility>
#include <map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// OpenCV scalar helpers
// ---------------------------------------------------------------------------------------------
inline int cvRoundF(float v) { return (int)lrintf(v); }    // round-half-even under default MXCSR
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
inline int cvCeilF(float v) { int i = (int)v; return i + (i < v); }
inline short saturateShort(float v) {
    int i = cvRoundF(v);
    return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

// ---------------------------------------------------------------------------------------------
// Sensitivity knob (tests/test_reference_pin.py, tools/pin_sensitivity.py): 0 = the faithful restatement.  A non-zero value
// swaps exactly ONE restated semantic for a plausible alternative, so that the reference-held known answer
// (img_folder/Screenshot.png: 1420 keypoints) can be shown to move — or not — with that semantic.  Set per
// extract() call from Oracle::mutation; never set by anything but those two files.
// ---------------------------------------------------------------------------------------------
enum Mutation {
    MUT_NONE = 0,
    MUT_RESIZE_TRUNC = 1,        // A.1: vertical pass without the "+2" rounding term
    MUT_RESIZE_FROM_LEVEL0 = 2,  // :1183 resizes level l-1; mutation: every level straight from level 0
    MUT_RESIZE_NO_RIGHT_CLAMP = 3,  // A.1: the sx >= sw-1 column keeps its fractional weight (reads the clamped neighbour)
    MUT_LEVEL_SIZE_FLOOR = 4,    // :1171 cvRound of the level size; mutation: truncation
    MUT_FAST_GE = 5,             // A.3: arc pixels strictly beyond v±t; mutation: >=
    MUT_NMS_GE = 6,              // A.3: strict 3x3 maximum; mutation: >= (ties survive)
    MUT_NMS_ACROSS_ROI = 7,      // A.3: NMS sees only the cell ROI's own scores; mutation: one score map per level
    MUT_NO_MIN_TH_RETRY = 8,     // :835-838 empty cell -> minThFAST; mutation: no retry
    MUT_SKIP_X_MINUS3 = 9,       // :802 iniX >= maxBorderX-6 skips the column; mutation: -3 as for rows
    MUT_STOP_GT_N = 10,          // :674 / :735 stop at size >= N; mutation: > N
    MUT_PHASE2_GE_N = 11,        // :678 enters the sorted phase when size + 3*nToExpand > N; mutation: >=
    MUT_TIE_OLDEST_FIRST = 12,   // :689 sort ties (heap address in the reference; oracle: newest first); mutation: oldest first
    MUT_HALF_FLOOR = 13,         // :488-489 ceil of the half extent; mutation: floor
    MUT_CELL_W_35 = 14,          // :777 W = 30; mutation: 35 (OpenCV-ORB-like grid)
    MUT_16BIT_SCALED = 15,       // (input decode, applied in Python) 16-bit PNG -> 8 bit by v*255/65535 instead of the high byte
    MUT_COUNT
};
thread_local int g_mut = MUT_NONE;

const int PATCH_SIZE = 31;        // ORBextractor.cc:70
const int HALF_PATCH_SIZE = 15;   // ORBextractor.cc:71
const int EDGE_THRESHOLD = 19;    // ORBextractor.cc:72

const int8_t kPattern[1024] = {
#include "../include/orbx_brief_pattern.inc"
};

// cv::KeyPoint layout: 28 bytes, no padding.
struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Gray {   // tightly described 8-bit view
    int w = 0, h = 0, stride = 0;
    const uint8_t* p = nullptr;
};

// ---------------------------------------------------------------------------------------------
// A.1  cv::resize(INTER_LINEAR), CV_8UC1, generic fixed-point path.
// Call site: ORBextractor.cc:1183-1188.
// ---------------------------------------------------------------------------------------------
void resizeLinear8u(const uint8_t* src, int sw, int sh, int sstride,
                    uint8_t* dst, int dw, int dh, int dstride) {
    const double scale_x = 1.0 / ((double)dw / sw);
    const double scale_y = 1.0 / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * dw), beta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { if (g_mut != MUT_RESIZE_NO_RIGHT_CLAMP) fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = saturateShort((1.f - fx) * 2048);
        alpha[2 * dx + 1] = saturateShort(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = saturateShort((1.f - fy) * 2048);
        beta[2 * dy + 1] = saturateShort(fy * 2048);
    }
    std::vector<int> row0(dw), row1(dw);
    auto hpass = [&](int sy, std::vector<int>& out) {
        const uint8_t* S = src + (size_t)sy * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (sx >= sw - 1) out[dx] = g_mut == MUT_RESIZE_NO_RIGHT_CLAMP ? S[sw - 1] * (alpha[2 * dx] + alpha[2 * dx + 1]) : S[sx] * 2048;   // right-clamped columns use one tap
            else out[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
        }
    };
    auto clip = [](int v, int n) { return v < 0 ? 0 : (v < n ? v : n - 1); };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], sh), sy1 = clip(yofs[dy] + 1, sh);
        hpass(sy0, row0);
        hpass(sy1, row1);
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + (g_mut == MUT_RESIZE_TRUNC ? 0 : 2)) >> 2);
    }
}

// A.4  BORDER_REFLECT_101 index map.
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

// ---------------------------------------------------------------------------------------------
// A.2  cv::GaussianBlur(7x7, sigma 2, REFLECT_101), CV_8UC1.  Call site: ORBextractor.cc:1127.
// Taps: getGaussianKernel(7, 2) in float, converted to 8-bit fixed point by rounding.
// ---------------------------------------------------------------------------------------------
void gaussTaps(int taps[7]) {
    // getGaussianKernel: exp in double, stored float, normalised by the double sum of the floats.
    float cf[7];
    double sum = 0, scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        double x = i - 3;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        taps[i] = cvRoundD((double)cf[i] * 256.0);
    }
}

void gaussianBlur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    int k[7];
    gaussTaps(k);
    std::vector<int> rows((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < 7; t++) s += k[t] * S[reflect101(x + t - 3, w)];
            rows[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < 7; t++) s += k[t] * rows[(size_t)reflect101(y + t - 3, h) * w + x];
            int v = (s + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// A.3  cv::FAST(img, kps, t, true) == TYPE_9_16 with cornerScore<16> and 3x3 strict NMS over the
// ROI only.  Call sites: ORBextractor.cc:818-819, 837-838.
// ---------------------------------------------------------------------------------------------
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// corner test + score, written from the definition: arc of 9 contiguous ring pixels all darker
// than v-t or all brighter than v+t; score = max(t, S_dark, S_bright) - 1.
inline int fastScoreAt(const uint8_t* p, int stride, int t, bool* isCorner) {
    int v = p[0];
    int d[25];
    for (int k = 0; k < 16; k++) d[k] = v - p[kRingDy[k] * stride + kRingDx[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int sDark = -1000, sBright = -1000;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
        sDark = std::max(sDark, mn);       // all d > t  <=> ring darker than v - t
        sBright = std::max(sBright, -mx);  // all -d > t <=> ring brighter than v + t
    }
    int s = std::max(sDark, sBright);
    *isCorner = g_mut == MUT_FAST_GE ? s >= t : s > t;
    return std::max(t, s) - 1;
}

void fastDetect(const Gray& roi, int threshold, bool nms, std::vector<KeyPoint>& out) {
    out.clear();
    const int W = roi.w, H = roi.h;
    if (W < 7 || H < 7) return;
    std::vector<uint8_t> score((size_t)W * H, 0);   // 0 outside the tested interior / non-corners
    std::vector<uint8_t> corner((size_t)W * H, 0);
    for (int y = 3; y < H - 3; y++)
        for (int x = 3; x < W - 3; x++) {
            bool c;
            int s = fastScoreAt(roi.p + (size_t)y * roi.stride + x, roi.stride, threshold, &c);
            if (c) { score[(size_t)y * W + x] = (uint8_t)s; corner[(size_t)y * W + x] = 1; }
        }
    for (int y = 3; y < H - 3; y++)
        for (int x = 3; x < W - 3; x++) {
            if (!corner[(size_t)y * W + x]) continue;
            int s = score[(size_t)y * W + x];
            if (nms) {
                const uint8_t* c = &score[(size_t)y * W + x];
                if (g_mut == MUT_NMS_GE) {
                    if (!(s >= c[-1] && s >= c[1] && s >= c[-W - 1] && s >= c[-W] && s >= c[-W + 1] &&
                          s >= c[W - 1] && s >= c[W] && s >= c[W + 1]))
                        continue;
                } else if (!(s > c[-1] && s > c[1] && s > c[-W - 1] && s > c[-W] && s > c[-W + 1] &&
                             s > c[W - 1] && s > c[W] && s > c[W + 1]))
                    continue;
            }
            out.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
        }
}

// ---------------------------------------------------------------------------------------------
// A.5  cv::fastAtan2 (degrees).  Call site: ORBextractor.cc:101.
// ---------------------------------------------------------------------------------------------
float fastAtan2f(float y, float x) {
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
                p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------------------------------------
// glibc >= 2.28 sinf/cosf algorithm restated (double-precision minimax polynomials after a
// quadrant reduction), valid for 0 <= x < 120.  The oracle itself calls the host's cosf/sinf as
// the reference does (ORBextractor.cc:111); this restatement is what the HIP kernel evaluates, and
// tests/test_oracle_primitives.py checks it against the host libm over every float in [0, 2*pi].
// ---------------------------------------------------------------------------------------------
void sincosRestated(float y, float* s_out, float* c_out) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    auto polySin = [&](double x, double x2) {
        double x3 = x * x2, s1 = S2 + x2 * S3, x7 = x3 * x2, s = x + x3 * S1;
        return s + x7 * s1;
    };
    auto polyCos = [&](double x2, double sgn) {   // sgn = -1 evaluates the negated table
        double x4 = x2 * x2, c2 = sgn * C3 + x2 * (sgn * C4), c1 = sgn * C0 + x2 * (sgn * C1),
               x6 = x4 * x2, c = c1 + x4 * (sgn * C2);
        return c + x6 * c2;
    };
    double x = y;
    // glibc compares only the top 12 bits (sign stripped) of y with those of pi/4, i.e. y < 0.75
    uint32_t yb;
    std::memcpy(&yb, &y, 4);
    const uint32_t top12 = (yb >> 20) & 0x7ff;
    if (top12 < 0x3f4) {
        double x2 = x * x;
        if (top12 < 0x398) { *s_out = y; *c_out = 1.0f; return; }   // y < 2^-12
        *s_out = (float)polySin(x, x2);
        *c_out = (float)polyCos(x2, 1.0);
        return;
    }
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * hpi;
    const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x2 = x * x;
    {   // sin
        double s = sign[n & 3];
        double sg = (n & 2) ? -1.0 : 1.0;
        *s_out = (n & 1) ? (float)polyCos(x2, sg) : (float)polySin(x * s, x2);
    }
    {   // cos
        double s = sign[(n + 1) & 3];
        double sg = ((n + 1) & 2) ? -1.0 : 1.0;
        *c_out = ((n ^ 1) & 1) ? (float)polyCos(x2, sg) : (float)polySin(x * s, x2);
    }
}

// ---------------------------------------------------------------------------------------------
// ExtractorNode / DivideNode: ORBextractor.h:31-42, ORBextractor.cc:486-542
// ---------------------------------------------------------------------------------------------
struct Pt2i { int x = 0, y = 0; };

struct Node {
    std::vector<KeyPoint> vKeys;
    Pt2i UL, UR, BL, BR;
    std::list<Node>::iterator lit;
    bool bNoMore = false;
    long seq = 0;   // creation order: the oracle's declared tie-break for the sort at :689

    void divide(Node& n1, Node& n2, Node& n3, Node& n4) const {
        const int halfX = g_mut == MUT_HALF_FLOOR ? (UR.x - UL.x) / 2 : (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
        const int halfY = g_mut == MUT_HALF_FLOOR ? (BR.y - UL.y) / 2 : (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
        n1.UL = UL;
        n1.UR = Pt2i{UL.x + halfX, UL.y};
        n1.BL = Pt2i{UL.x, UL.y + halfY};
        n1.BR = Pt2i{UL.x + halfX, UL.y + halfY};
        n2.UL = n1.UR; n2.UR = UR; n2.BL = n1.BR; n2.BR = Pt2i{UR.x, UL.y + halfY};
        n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = BL; n3.BR = Pt2i{n1.BR.x, BL.y};
        n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = BR;
        for (size_t i = 0; i < vKeys.size(); i++) {
            const KeyPoint& kp = vKeys[i];
            if (kp.x < n1.UR.x) {
                if (kp.y < n1.BR.y) n1.vKeys.push_back(kp);
                else n3.vKeys.push_back(kp);
            } else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
            else n4.vKeys.push_back(kp);
        }
        if (n1.vKeys.size() == 1) n1.bNoMore = true;
        if (n2.vKeys.size() == 1) n2.bNoMore = true;
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
};

// DistributeOctTree: ORBextractor.cc:544-771.
std::vector<KeyPoint> distributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX,
                                        int maxX, int minY, int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    std::vector<KeyPoint> vResultKeys;
    if (nIni < 1) return vResultKeys;   // reference indexes an empty vector here (undefined)
    const float hX = static_cast<float>(maxX - minX) / nIni;
    long seq = 0;

    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.UL = Pt2i{(int)(hX * static_cast<float>(i)), 0};
        ni.UR = Pt2i{(int)(hX * static_cast<float>(i + 1)), 0};
        ni.BL = Pt2i{ni.UL.x, maxY - minY};
        ni.BR = Pt2i{ni.UR.x, maxY - minY};
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KeyPoint& kp = vToDistributeKeys[i];
        vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
        else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
        else lit++;
    }

    bool bFinish = false;
    typedef std::pair<std::pair<int, long>, Node*> SizeSeqNode;   // (size, seq) then pointer payload
    std::vector<SizeSeqNode> vSizeAndPointerToNode;

    auto pushChildren = [&](Node& n1, Node& n2, Node& n3, Node& n4, int* nToExpand) {
        Node* ch[4] = {&n1, &n2, &n3, &n4};
        for (int c = 0; c < 4; c++) {
            if (ch[c]->vKeys.size() > 0) {
                ch[c]->seq = seq++;
                lNodes.push_front(*ch[c]);
                if (ch[c]->vKeys.size() > 1) {
                    if (nToExpand) (*nToExpand)++;
                    vSizeAndPointerToNode.push_back(
                        {{(int)ch[c]->vKeys.size(), lNodes.front().seq}, &lNodes.front()});
                    lNodes.front().lit = lNodes.begin();
                }
            }
        }
    };

    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) { lit++; continue; }
            Node n1, n2, n3, n4;
            lit->divide(n1, n2, n3, n4);
            pushChildren(n1, n2, n3, n4, &nToExpand);
            lit = lNodes.erase(lit);
        }
        const int stopAt = g_mut == MUT_STOP_GT_N ? N + 1 : N;
        if ((int)lNodes.size() >= stopAt || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > (g_mut == MUT_PHASE2_GE_N ? N - 1 : N)) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeSeqNode> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                // reference: sort by (size, heap pointer); oracle: by (size, creation seq)
                if (g_mut == MUT_TIE_OLDEST_FIRST)
                    std::sort(vPrev.begin(), vPrev.end(), [](const SizeSeqNode& a, const SizeSeqNode& b) {
                        return a.first.first != b.first.first ? a.first.first < b.first.first : a.first.second > b.first.second;
                    });
                else
                std::sort(vPrev.begin(), vPrev.end(),
                          [](const SizeSeqNode& a, const SizeSeqNode& b) { return a.first < b.first; });
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    vPrev[j].second->divide(n1, n2, n3, n4);
                    pushChildren(n1, n2, n3, n4, nullptr);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= stopAt) break;
                }
                if ((int)lNodes.size() >= stopAt || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    vResultKeys.reserve(lNodes.size());
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
        const std::vector<KeyPoint>& vNodeKeys = it->vKeys;
        const KeyPoint* pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++)
            if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

// ---------------------------------------------------------------------------------------------
// The extractor: ORBextractor.cc:408-475 (ctor), :773-888, :1078-1219.
// ---------------------------------------------------------------------------------------------
struct Level {
    int w = 0, h = 0;
    std::vector<uint8_t> buf;   // (w+38) x (h+38), interior at (19,19)
    int stride() const { return w + 2 * EDGE_THRESHOLD; }
    uint8_t* interior() { return buf.data() + (size_t)EDGE_THRESHOLD * stride() + EDGE_THRESHOLD; }
    const uint8_t* interior() const { return buf.data() + (size_t)EDGE_THRESHOLD * stride() + EDGE_THRESHOLD; }
};

struct Oracle {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Level> pyr;
    std::vector<std::vector<uint8_t>> blurred;
    std::vector<std::vector<KeyPoint>> candidates, levelKeys;
    int mutation = MUT_NONE;   // sensitivity knob, see enum Mutation

    Oracle(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf) {
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);   // float*double -> float
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRoundF(nDesired);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesired *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloorF(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeilF(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        pyr.resize(nlevels);
        blurred.resize(nlevels);
        candidates.resize(nlevels);
        levelKeys.resize(nlevels);
    }

    // ORBextractor.cc:1164-1219
    void computePyramid(const Gray& image) {
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            Level& L = pyr[level];
            L.w = g_mut == MUT_LEVEL_SIZE_FLOOR ? (int)((float)image.w * scale) : cvRoundF((float)image.w * scale);
            L.h = g_mut == MUT_LEVEL_SIZE_FLOOR ? (int)((float)image.h * scale) : cvRoundF((float)image.h * scale);
            L.buf.assign((size_t)(L.w + 2 * EDGE_THRESHOLD) * (L.h + 2 * EDGE_THRESHOLD), 0);
            if (level != 0) {
                const Level& P = pyr[g_mut == MUT_RESIZE_FROM_LEVEL0 ? 0 : level - 1];
                resizeLinear8u(P.interior(), P.w, P.h, P.stride(), L.interior(), L.w, L.h, L.stride());
            } else {
                for (int y = 0; y < L.h; y++)
                    std::memcpy(L.interior() + (size_t)y * L.stride(), image.p + (size_t)y * image.stride, L.w);
            }
            // copyMakeBorder(..., 19,19,19,19, BORDER_REFLECT_101) around the interior
            const int S = L.stride(), E = EDGE_THRESHOLD;
            for (int by = 0; by < L.h + 2 * E; by++) {
                int sy = reflect101(by - E, L.h);
                for (int bx = 0; bx < S; bx++) {
                    if (by >= E && by < E + L.h && bx >= E && bx < E + L.w) continue;
                    int sx = reflect101(bx - E, L.w);
                    L.buf[(size_t)by * S + bx] = L.interior()[(size_t)sy * S + sx];
                }
            }
        }
    }

    // ORBextractor.cc:75-102
    float icAngle(const Level& L, float ptx, float pty) const {
        int m_01 = 0, m_10 = 0;
        const int step = L.stride();
        const uint8_t* center = L.interior() + (size_t)cvRoundF(pty) * step + cvRoundF(ptx);
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fastAtan2f((float)m_01, (float)m_10);
    }

    // ORBextractor.cc:773-888
    void computeKeyPointsOctTree() {
        const float W = g_mut == MUT_CELL_W_35 ? 35 : 30;
        for (int level = 0; level < nlevels; ++level) {
            const Level& L = pyr[level];
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = L.w - EDGE_THRESHOLD + 3;
            const int maxBorderY = L.h - EDGE_THRESHOLD + 3;
            std::vector<KeyPoint>& vToDistributeKeys = candidates[level];
            vToDistributeKeys.clear();
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W);
            const int nRows = (int)(height / W);
            const int wCell = (int)std::ceil(width / nCols);
            const int hCell = (int)std::ceil(height / nRows);
            // MUT_NMS_ACROSS_ROI: one FAST + NMS over the whole active rectangle per threshold, dealt to the cells afterwards
            std::vector<KeyPoint> wholeIni, wholeMin;
            if (g_mut == MUT_NMS_ACROSS_ROI) {
                Gray all;
                all.w = maxBorderX - minBorderX; all.h = maxBorderY - minBorderY; all.stride = L.stride();
                all.p = L.interior() + (size_t)minBorderY * L.stride() + minBorderX;
                fastDetect(all, iniThFAST, true, wholeIni);
                fastDetect(all, minThFAST, true, wholeMin);
            }
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - (g_mut == MUT_SKIP_X_MINUS3 ? 3 : 6)) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    Gray roi;
                    roi.w = (int)maxX - (int)iniX;
                    roi.h = (int)maxY - (int)iniY;
                    roi.stride = L.stride();
                    roi.p = L.interior() + (size_t)(int)iniY * L.stride() + (int)iniX;
                    std::vector<KeyPoint> vKeysCell;
                    if (g_mut == MUT_NMS_ACROSS_ROI) {
                        // the cell's own pixels: ROI interior [3, w-3) x [3, h-3), in rectangle coordinates
                        const int x0 = j * wCell + 3, x1 = j * wCell + roi.w - 3, y0 = i * hCell + 3, y1 = i * hCell + roi.h - 3;
                        for (int pass = 0; pass < 2 && vKeysCell.empty(); pass++)
                            for (const KeyPoint& kp : (pass ? wholeMin : wholeIni))
                                if (kp.x >= x0 && kp.x < x1 && kp.y >= y0 && kp.y < y1) vKeysCell.push_back(kp);
                        for (auto& kp : vKeysCell) vToDistributeKeys.push_back(kp);
                        continue;
                    }
                    fastDetect(roi, iniThFAST, true, vKeysCell);
                    if (vKeysCell.empty() && g_mut != MUT_NO_MIN_TH_RETRY) fastDetect(roi, minThFAST, true, vKeysCell);
                    for (auto& kp : vKeysCell) {
                        kp.x += j * wCell;
                        kp.y += i * hCell;
                        vToDistributeKeys.push_back(kp);
                    }
                }
            }
            std::vector<KeyPoint>& keypoints = levelKeys[level];
            keypoints = distributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                          mnFeaturesPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (auto& kp : keypoints) {
                kp.x += minBorderX;
                kp.y += minBorderY;
                kp.octave = level;
                kp.size = (float)scaledPatchSize;
            }
        }
        for (int level = 0; level < nlevels; ++level)
            for (auto& kp : levelKeys[level]) kp.angle = icAngle(pyr[level], kp.x, kp.y);
    }

    // ORBextractor.cc:106-145
    static void computeOrbDescriptor(const KeyPoint& kpt, const uint8_t* img, int step, uint8_t* desc) {
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a = (float)cosf(angle), b = (float)sinf(angle);
        const uint8_t* center = img + (ptrdiff_t)cvRoundF(kpt.y) * step + cvRoundF(kpt.x);
        const int8_t* pattern = kPattern;
        auto GET = [&](int idx) -> int {
            float px = pattern[2 * idx], py = pattern[2 * idx + 1];
            return center[cvRoundF(px * b + py * a) * step + cvRoundF(px * a - py * b)];
        };
        for (int i = 0; i < 32; ++i, pattern += 32) {
            int val = 0;
            for (int k = 0; k < 8; k++) {
                int t0 = GET(2 * k), t1 = GET(2 * k + 1);
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    // ORBextractor.cc:1078-1162.  Returns nkeypoints (>= 0) or -1 for an empty image; *mono = return
    // value of the reference's operator().
    int extract(const Gray& image, int lap0, int lap1, KeyPoint* outK, uint8_t* outD, int capacity, int* mono) {
        if (image.w <= 0 || image.h <= 0 || !image.p) { *mono = -1; return -1; }
        struct MutScope { MutScope(int m) { g_mut = m; } ~MutScope() { g_mut = MUT_NONE; } } mutScope(mutation);
        computePyramid(image);
        computeKeyPointsOctTree();
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)levelKeys[level].size();
        if (nkeypoints > capacity) { *mono = -2; return -2; }
        int monoIndex = 0, stereoIndex = nkeypoints - 1;
        for (int level = 0; level < nlevels; ++level) {
            const Level& L = pyr[level];
            blurred[level].assign((size_t)L.w * L.h, 0);
            std::vector<KeyPoint> keypoints = levelKeys[level];   // copy: levelKeys keeps level coords
            if (keypoints.empty()) continue;
            // clone() drops the border, so REFLECT_101 mirrors the level itself
            gaussianBlur7(L.interior(), L.w, L.h, L.stride(), blurred[level].data(), L.w);
            float scale = mvScaleFactor[level];
            for (auto& kp : keypoints) {
                uint8_t d[32];
                computeOrbDescriptor(kp, blurred[level].data(), L.w, d);
                if (level != 0) { kp.x *= scale; kp.y *= scale; }
                int at;
                if (kp.x >= lap0 && kp.x <= lap1) at = stereoIndex--;
                else at = monoIndex++;
                outK[at] = kp;
                std::memcpy(outD + (size_t)at * 32, d, 32);
            }
        }
        *mono = monoIndex;
        return nkeypoints;
    }
};

}  // namespace

// =============================================================================================
// C API for ctypes (tests / smoke / bench cpu_baseline only)
// =============================================================================================
extern "C" {

void* oracle_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new Oracle(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void oracle_destroy(void* h) { delete (Oracle*)h; }
// sensitivity knob for the reference-held pin (enum Mutation); 0 = faithful.  Returns the number of defined mutations.
int oracle_set_mutation(void* h, int m) { ((Oracle*)h)->mutation = (m > 0 && m < MUT_COUNT) ? m : MUT_NONE; return MUT_COUNT; }

int oracle_extract(void* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                   void* kps, uint8_t* desc, int capacity, int* mono_index) {
    Gray g; g.w = cols; g.h = rows; g.stride = stride; g.p = img;
    return ((Oracle*)h)->extract(g, lap0, lap1, (KeyPoint*)kps, desc, capacity, mono_index);
}

void oracle_get_tables(void* h, float* sf, float* isf, float* s2, float* is2, int* nfeat, int* umax16) {
    Oracle* o = (Oracle*)h;
    for (int i = 0; i < o->nlevels; i++) {
        sf[i] = o->mvScaleFactor[i]; isf[i] = o->mvInvScaleFactor[i];
        s2[i] = o->mvLevelSigma2[i]; is2[i] = o->mvInvLevelSigma2[i];
        nfeat[i] = o->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = o->umax[i];
}
void oracle_level_size(void* h, int level, int* w, int* hh) {
    Oracle* o = (Oracle*)h; *w = o->pyr[level].w; *hh = o->pyr[level].h;
}
// bordered != 0: (w+38)x(h+38) tight; else w x h tight
void oracle_get_level(void* h, int level, int bordered, uint8_t* dst) {
    const Level& L = ((Oracle*)h)->pyr[level];
    if (bordered) { std::memcpy(dst, L.buf.data(), L.buf.size()); return; }
    for (int y = 0; y < L.h; y++) std::memcpy(dst + (size_t)y * L.w, L.interior() + (size_t)y * L.stride(), L.w);
}
void oracle_get_blurred(void* h, int level, uint8_t* dst) {
    Oracle* o = (Oracle*)h;
    std::memcpy(dst, o->blurred[level].data(), o->blurred[level].size());
}
int oracle_num_candidates(void* h, int level) { return (int)((Oracle*)h)->candidates[level].size(); }
void oracle_get_candidates(void* h, int level, void* out) {
    Oracle* o = (Oracle*)h;
    std::memcpy(out, o->candidates[level].data(), o->candidates[level].size() * sizeof(KeyPoint));
}
int oracle_num_level_keys(void* h, int level) { return (int)((Oracle*)h)->levelKeys[level].size(); }
void oracle_get_level_keys(void* h, int level, void* out) {
    Oracle* o = (Oracle*)h;
    std::memcpy(out, o->levelKeys[level].data(), o->levelKeys[level].size() * sizeof(KeyPoint));
}

// ---- primitives, exposed so tests can pin each one by an independent definition -------------
void oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    resizeLinear8u(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void oracle_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    gaussianBlur7(src, w, h, sstride, dst, dstride);
}
void oracle_gauss_taps(int* taps7) { gaussTaps(taps7); }
int oracle_fast(const uint8_t* img, int w, int h, int stride, int threshold, int nms, void* out, int cap) {
    Gray g; g.w = w; g.h = h; g.stride = stride; g.p = img;
    std::vector<KeyPoint> v;
    fastDetect(g, threshold, nms != 0, v);
    int n = (int)std::min<size_t>(v.size(), (size_t)cap);
    std::memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return (int)v.size();
}
float oracle_fast_atan2(float y, float x) { return fastAtan2f(y, x); }
void oracle_fast_atan2_array(const float* y, const float* x, float* out, int n) {
    for (int i = 0; i < n; i++) out[i] = fastAtan2f(y[i], x[i]);
}
int oracle_distribute(const void* keys, int n, int minX, int maxX, int minY, int maxY, int N, void* out, int cap) {
    std::vector<KeyPoint> v((const KeyPoint*)keys, (const KeyPoint*)keys + n);
    std::vector<KeyPoint> r = distributeOctTree(v, minX, maxX, minY, maxY, N);
    int m = (int)std::min<size_t>(r.size(), (size_t)cap);
    std::memcpy(out, r.data(), (size_t)m * sizeof(KeyPoint));
    return (int)r.size();
}
void oracle_sincos_restated(float x, float* s, float* c) { sincosRestated(x, s, c); }
// compares the restated sinf/cosf with the host libm over every float in [lo, hi]; returns #mismatches
long oracle_sincos_check(float lo, float hi, long* ntested) {
    uint32_t a, b;
    std::memcpy(&a, &lo, 4); std::memcpy(&b, &hi, 4);
    long bad = 0, n = 0;
    for (uint32_t u = a; u <= b; u++) {
        float x; std::memcpy(&x, &u, 4);
        float s, c; sincosRestated(x, &s, &c);
        if (s != sinf(x) || c != cosf(x)) bad++;
        n++;
    }
    *ntested = n;
    return bad;
}
void oracle_describe(const uint8_t* blurredImg, int step, float x, float y, float angle, uint8_t* desc32) {
    KeyPoint kp{x, y, 31.f, angle, 0.f, 0, -1};
    Oracle::computeOrbDescriptor(kp, blurredImg, step, desc32);
}

// ---------------------------------------------------------------------------------------------
// "Next" row: Frame::ComputeStereoMatches, reference src/Frame.cc:813-991, restated on the
// two oracle extractors that produced the left and right features.  Same test-infrastructure status as the rest.
// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2349-2365) is the 256-bit Hamming distance.
// ---------------------------------------------------------------------------------------------
static int descriptorDistance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// Returns the number of left keypoints that kept a stereo match.
int oracle_stereo_match(void* hL, void* hR, const void* kpsL_, const uint8_t* descL, int N, const void* kpsR_,
                        const uint8_t* descR, int Nr, float mbf, float mb, float* mvuRight, float* mvDepth) {
    const Oracle* oL = (const Oracle*)hL;
    const Oracle* oR = (const Oracle*)hR;
    const KeyPoint* mvKeys = (const KeyPoint*)kpsL_;
    const KeyPoint* mvKeysRight = (const KeyPoint*)kpsR_;
    const std::vector<float>& mvScaleFactors = oL->mvScaleFactor;
    const std::vector<float>& mvInvScaleFactors = oL->mvInvScaleFactor;
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int TH_HIGH = 100, TH_LOW = 50;                      // ORBmatcher.cc:36-37
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = oL->pyr[0].h;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const KeyPoint& kp = mvKeysRight[iR];
        const float kpY = kp.y;
        const float r = 2.0f * mvScaleFactors[kp.octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // (the reference does not guard; keypoints stay >= 17*scale inside)
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    auto levelAt = [](const Oracle* o, int level, int y, int x) -> int {
        const Level& L = o->pyr[level];
        return L.interior()[(ptrdiff_t)y * L.stride() + x];     // the Mats are views into the bordered buffers
    };
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = mvKeys[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        if ((int)vL < 0 || (int)vL >= nRows) continue;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const KeyPoint& kpR = mvKeysRight[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = descriptorDistance(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = mvKeysRight[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, L = 5;
            int IL[11][11];
            {
                const int y0 = (int)(scaledvL - w), x0 = (int)(scaleduL - w);
                for (int a = 0; a < 11; a++)
                    for (int b2 = 0; b2 < 11; b2++) IL[a][b2] = levelAt(oL, kpL.octave, y0 + a, x0 + b2);
                const int c = IL[w][w];
                for (int a = 0; a < 11; a++) for (int b2 = 0; b2 < 11; b2++) IL[a][b2] -= c;
            }
            int bestDistS = 2147483647, bestincR = 0;
            std::vector<float> vDists(2 * L + 1);
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= oR->pyr[kpL.octave].w) continue;
            for (int incR = -L; incR <= +L; incR++) {
                const int y0 = (int)(scaledvL - w), x0 = (int)(scaleduR0 + incR - w);
                int IR[11][11];
                for (int a = 0; a < 11; a++)
                    for (int b2 = 0; b2 < 11; b2++) IR[a][b2] = levelAt(oR, kpL.octave, y0 + a, x0 + b2);
                const int c = IR[w][w];
                double norm = 0;
                for (int a = 0; a < 11; a++) for (int b2 = 0; b2 < 11; b2++) norm += std::abs(IL[a][b2] - (IR[a][b2] - c));
                const float dist = (float)norm;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;       // the reference indexes vDistIdx[0] here (undefined for an empty vector)
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
        kept--;
    }
    return kept;
}
int oracle_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptorDistance(a, b); }

// ---------------------------------------------------------------------------------------------
// "Next" row: what the Frame constructor does with the extracted keypoints —
// Frame::UndistortKeyPoints (src/Frame.cc:748-782), ComputeImageBounds (:784-811), AssignFeaturesToGrid (:383-417)
// with PosInGrid (:726-736).  cv::undistortPoints is OpenCV again (not vendored): restated here as the 3.4.x
// cvUndistortPointsInternal path for a 4/5-coefficient model, no rectification, P = K: five fixed-point iterations
// in double, then re-projection.
// ---------------------------------------------------------------------------------------------
struct Camera { float fx, fy, cx, cy, k1, k2, p1, p2, k3; };

static void undistortPoint(const Camera& c, float xin, float yin, float* xo, float* yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy, ifx = 1. / fx, ify = 1. / fy;
    double k[12] = {c.k1, c.k2, c.p1, c.p2, c.k3, 0, 0, 0, 0, 0, 0, 0};
    double x = xin, y = yin;
    const double u = x, v = y;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww); *yo = (float)(yy * ww);
}

// bounds[4] = mnMinX, mnMaxX, mnMinY, mnMaxY
void oracle_image_bounds(const void* cam_, int cols, int rows, float* bounds) {
    const Camera& c = *(const Camera*)cam_;
    if (c.k1 != 0.0) {
        float m[4][2] = {{0.f, 0.f}, {(float)cols, 0.f}, {0.f, (float)rows}, {(float)cols, (float)rows}};
        for (int i = 0; i < 4; i++) undistortPoint(c, m[i][0], m[i][1], &m[i][0], &m[i][1]);
        bounds[0] = std::min(m[0][0], m[2][0]); bounds[1] = std::max(m[1][0], m[3][0]);
        bounds[2] = std::min(m[0][1], m[1][1]); bounds[3] = std::max(m[2][1], m[3][1]);
    } else {
        bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
    }
}

// mvKeysUn, and mGrid as CSR over cells x*48+y (x < 64, y < 48): gridOff[3073], gridIdx[N] (keypoint indices in
// increasing order inside a cell, as push_back leaves them).  Returns the number of keypoints inside the grid.
int oracle_frame_finish(const void* cam_, const void* kps_, int N, const float* bounds, void* kpsUn_, int* gridOff, int* gridIdx) {
    const Camera& c = *(const Camera*)cam_;
    const KeyPoint* mvKeys = (const KeyPoint*)kps_;
    KeyPoint* mvKeysUn = (KeyPoint*)kpsUn_;
    for (int i = 0; i < N; i++) {
        mvKeysUn[i] = mvKeys[i];
        if (c.k1 != 0.0) undistortPoint(c, mvKeys[i].x, mvKeys[i].y, &mvKeysUn[i].x, &mvKeysUn[i].y);
    }
    const int COLS = 64, ROWS = 48;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / static_cast<float>(mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / static_cast<float>(mnMaxY - mnMinY);
    std::vector<std::vector<int>> grid(COLS * ROWS);
    int inside = 0;
    for (int i = 0; i < N; i++) {
        const int posX = (int)std::round((mvKeysUn[i].x - mnMinX) * wInv), posY = (int)std::round((mvKeysUn[i].y - mnMinY) * hInv);
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        grid[posX * ROWS + posY].push_back(i);
        inside++;
    }
    int o = 0;
    for (int cell = 0; cell < COLS * ROWS; cell++) {
        gridOff[cell] = o;
        for (int i : grid[cell]) gridIdx[o++] = i;
    }
    gridOff[COLS * ROWS] = o;
    return inside;
}

// Frame::AssignFeaturesToGrid with Nleft != -1 (src/Frame.cc:383-417, the branch :404-414 of the two-camera constructor :1045-1122, which calls it
// BEFORE UndistortKeyPoints): kp = i < Nleft ? mvKeys[i] : mvKeysRight[i - Nleft] - the RAW keys -, left keys into mGrid with index i, right keys
// into mGridRight with index i - Nleft.  Both grids as the CSR of oracle_frame_finish.  Returns the keypoints inside mGrid; *insideRight those inside mGridRight.
int oracle_assign_features_two_eyes(const void* kpsLeft_, int Nleft, const void* kpsRight_, int Nright, const float* bounds, int* gridOff, int* gridIdx,
                                    int* gridOffRight, int* gridIdxRight, int* insideRight) {
    const KeyPoint* mvKeys = (const KeyPoint*)kpsLeft_;
    const KeyPoint* mvKeysRight = (const KeyPoint*)kpsRight_;
    const int COLS = 64, ROWS = 48, N = Nleft + Nright;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float wInv = static_cast<float>(COLS) / static_cast<float>(mnMaxX - mnMinX);
    const float hInv = static_cast<float>(ROWS) / static_cast<float>(mnMaxY - mnMinY);
    std::vector<std::vector<int>> mGrid(COLS * ROWS), mGridRight(COLS * ROWS);
    int nl = 0, nr = 0;
    for (int i = 0; i < N; i++) {
        const KeyPoint& kp = (i < Nleft) ? mvKeys[i] : mvKeysRight[i - Nleft];                                          // :405-407
        const int posX = (int)std::round((kp.x - mnMinX) * wInv), posY = (int)std::round((kp.y - mnMinY) * hInv);       // PosInGrid :728-729
        if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
        if (i < Nleft) { mGrid[posX * ROWS + posY].push_back(i); nl++; }                                                // :411-412
        else { mGridRight[posX * ROWS + posY].push_back(i - Nleft); nr++; }                                             // :413-414
    }
    int o = 0, orr = 0;
    for (int cell = 0; cell < COLS * ROWS; cell++) {
        gridOff[cell] = o; gridOffRight[cell] = orr;
        for (int i : mGrid[cell]) gridIdx[o++] = i;
        for (int i : mGridRight[cell]) gridIdxRight[orr++] = i;
    }
    gridOff[COLS * ROWS] = o; gridOffRight[COLS * ROWS] = orr;
    *insideRight = nr;
    return nl;
}

// ---------------------------------------------------------------------------------------------
// "Next" row: ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821) with
// Frame::GetFeaturesInArea (src/Frame.cc:655-724) and ComputeThreeMaxima (src/ORBmatcher.cc:2303-2344).
// Inputs are what the Frame constructor leaves behind: mvKeysUn + mDescriptors of both frames, frame 2's mGrid
// (as the CSR of oracle_frame_finish) and the image bounds.  TH_LOW = 50, HISTO_LENGTH = 30 (ORBmatcher.cc:37-38).
// ---------------------------------------------------------------------------------------------
static std::vector<size_t> getFeaturesInArea(const KeyPoint* mvKeysUn, const int* gridOff, const int* gridIdx, const float* bounds,
                                             float x, float y, float r, int minLevel, int maxLevel) {
    const int COLS = 64, ROWS = 48;
    const float mnMinX = bounds[0], mnMaxX = bounds[1], mnMinY = bounds[2], mnMaxY = bounds[3];
    const float mfGridElementWidthInv = static_cast<float>(COLS) / static_cast<float>(mnMaxX - mnMinX);
    const float mfGridElementHeightInv = static_cast<float>(ROWS) / static_cast<float>(mnMaxY - mnMinY);
    std::vector<size_t> vIndices;
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - factorX) * mfGridElementWidthInv));     // Frame.cc:666
    if (nMinCellX >= COLS) return vIndices;
    const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - mnMinX + factorX) * mfGridElementWidthInv));  // :672
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - factorY) * mfGridElementHeightInv));    // :678
    if (nMinCellY >= ROWS) return vIndices;
    const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - mnMinY + factorY) * mfGridElementHeightInv)); // :684
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);                                              // :690
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int j = gridOff[ix * ROWS + iy]; j < gridOff[ix * ROWS + iy + 1]; j++) {
                const KeyPoint& kpUn = mvKeysUn[gridIdx[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x, disty = kpUn.y - y;
                if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) vIndices.push_back((size_t)gridIdx[j]);   // :717
            }
    return vIndices;
}

static void computeThreeMaxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// prevMatched: N1 x (x, y), in/out (vbPrevMatched).  matches12: N1 ints out (vnMatches12).  Returns nmatches.
int oracle_search_for_initialization(const void* kpsUn1_, const uint8_t* desc1, int N1, const void* kpsUn2_, const uint8_t* desc2,
                                     int N2, const int* gridOff2, const int* gridIdx2, const float* bounds, float* prevMatched,
                                     int windowSize, float nnratio, int checkOrientation, int* matches12) {
    const KeyPoint* k1 = (const KeyPoint*)kpsUn1_;
    const KeyPoint* k2 = (const KeyPoint*)kpsUn2_;
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    int nmatches = 0;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(N2, INT_MAX), vnMatches21(N2, -1);
    for (int i1 = 0; i1 < N1; i1++) {
        const KeyPoint kp1 = k1[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<size_t> vIndices2 = getFeaturesInArea(k2, gridOff2, gridIdx2, bounds, prevMatched[2 * i1], prevMatched[2 * i1 + 1],
                                                          (float)windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            const int dist = descriptorDistance(d1, desc2 + i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOrientation) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        computeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < N1; i1++)
        if (matches12[i1] >= 0) { prevMatched[2 * i1] = k2[matches12[i1]].x; prevMatched[2 * i1 + 1] = k2[matches12[i1]].y; }
    return nmatches;
}

// Frame::GetFeaturesInArea alone (src/Frame.cc:655-724): indices in traversal order; returns their number.
int oracle_features_in_area(const void* kpsUn, const int* gridOff, const int* gridIdx, const float* bounds, float x, float y, float r,
                            int minLevel, int maxLevel, int* out, int capacity) {
    std::vector<size_t> v = getFeaturesInArea((const KeyPoint*)kpsUn, gridOff, gridIdx, bounds, x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size() && (int)i < capacity; i++) out[i] = (int)v[i];
    return (int)v.size();
}

// ---------------------------------------------------------------------------------------------
// "Next" row (second half): ORBmatcher::SearchByProjection.
//   (a) frame-to-frame, Tracking::TrackWithMotionModel: src/ORBmatcher.cc:1961-2177
//   (b) map-to-frame,   Tracking::SearchLocalPoints:     src/ORBmatcher.cc:44-267
// Both for Nleft == -1 (monocular, rectified stereo, RGB-D: one descriptor set per frame); the Nleft != -1 branches belong to the
// fisheye-stereo rig (KannalaBrandt8), whose camera model is out of scope.  MapPoints, poses and frustum
// results come from the tracker and the map (out of scope), so they enter as plain arrays.
// cv::Mat arithmetic restated: `Rcw*x3Dw+tcw` is ONE cv::gemm (MatExpr folds A*B+C): every element accumulates its three products
// in double, adds the double of tcw, and rounds to float once (OpenCV 3.x GEMMSingleMul<float,double>).  OpenCV is absent from this
// image, so this too is "parity unpinned" (DESIGN.md §2).
// ---------------------------------------------------------------------------------------------
static void gemm3(const float* A /*3x3 row-major, row stride sa*/, int sa, bool transA, const float* b, double alpha, const float* c, float* out) {
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(transA ? A[k * sa + r] : A[r * sa + k]) * (double)b[k];
        out[r] = (float)(s * alpha + (c ? (double)c[r] : 0.0));
    }
}

struct ProjQuery { float u, v, ur, radius; int minLevel, maxLevel, flags; float angle; };   // == orbx_proj_query (32 bytes)

// The per-keypoint front half of (a): :1989-2016 (+ the stereo prediction of :2043).  flags bit0 = search this one, bit1 = its
// MapPoint has Observations() > 0 (a later candidate check, :2035-2037, skips keypoints such a point already holds).
// mpFlags: bit0 = LastFrame.mvpMapPoints[i] != NULL && !LastFrame.mvbOutlier[i], bit1 = Observations() > 0.
void oracle_project_last_frame(const void* kpsLast_, const void* kpsUnLast_, int NL, const uint8_t* mpFlags, const float* world,
                               const float* Tcw /*3x4*/, const float* Tlw /*3x4*/, const float* cam4 /*fx fy cx cy*/, const float* bounds,
                               const float* scaleFactors, float mbf, float mb, float th, int bMono, void* queries_) {
    const KeyPoint* kL = (const KeyPoint*)kpsLast_;
    const KeyPoint* kUnL = (const KeyPoint*)kpsUnLast_;
    ProjQuery* q = (ProjQuery*)queries_;
    const float tcw[3] = {Tcw[3], Tcw[7], Tcw[11]}, tlw[3] = {Tlw[3], Tlw[7], Tlw[11]};
    float twc[3], tlc[3];
    gemm3(Tcw, 4, true, tcw, -1.0, nullptr, twc);            // twc = -Rcw.t()*tcw   (:1975)
    gemm3(Tlw, 4, false, twc, 1.0, tlw, tlc);                // tlc = Rlw*twc+tlw    (:1980)
    const bool bForward = tlc[2] > mb && !bMono;             // :1982-1983
    const bool bBackward = -tlc[2] > mb && !bMono;
    for (int i = 0; i < NL; i++) {
        q[i] = ProjQuery{0, 0, 0, 0, 0, 0, 0, 0};
        if (!(mpFlags[i] & 1)) continue;                     // :1987-1990
        float x3Dc[3];
        gemm3(Tcw, 4, false, world + 3 * i, 1.0, tcw, x3Dc); // :1994
        const float invzc = (float)(1.0 / x3Dc[2]);          // :1998 (double division, float variable)
        if (invzc < 0) continue;
        const float u = cam4[0] * x3Dc[0] / x3Dc[2] + cam4[2], v = cam4[1] * x3Dc[1] / x3Dc[2] + cam4[3];   // Pinhole::project (CameraModels/Pinhole.cpp)
        if (u < bounds[0] || u > bounds[1]) continue;        // :2005-2008
        if (v < bounds[2] || v > bounds[3]) continue;
        const int nLastOctave = kL[i].octave;                // :2010
        q[i].u = u; q[i].v = v;
        q[i].radius = th * scaleFactors[nLastOctave];        // :2014
        q[i].ur = u - mbf * invzc;                           // :2043
        if (bForward) { q[i].minLevel = nLastOctave; q[i].maxLevel = -1; }               // :2018-2023
        else if (bBackward) { q[i].minLevel = 0; q[i].maxLevel = nLastOctave; }
        else { q[i].minLevel = nLastOctave - 1; q[i].maxLevel = nLastOctave + 1; }
        q[i].angle = kUnL[i].angle;                          // kpLF = LastFrame.mvKeysUn[i] (:2067)
        q[i].flags = 1 | (mpFlags[i] & 2);
    }
}

// (c) SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:2179-2300, Tracking::Relocalization) is the search
// of (a) with maxDistance = ORBdist, occupied = "CurrentFrame.mvpMapPoints[i2] != NULL" (:2238-2239: any MapPoint closes a keypoint, so every
// request carries flags bit 1), no stereo column test; its projection (:2200-2226) reads MapPoint state and stays with the caller.
// The search half, shared by (a) and (b).  ratioMode 0 = (a): best candidate only, TH_HIGH (:2028-2062), rotation histogram
// (:2064-2080, :2155-2175).  ratioMode 1 = (b): best + second with the level rule and mfNNratio (:85-132), no histogram.
// occupied[i2] (in/out, may be NULL = all free): CurrentFrame.mvpMapPoints[i2] is set and has Observations() > 0.
// matches[i2] (out): index of the query whose MapPoint the keypoint ends up holding, -1 = none.  Returns nmatches.
int oracle_search_by_projection(const void* queries_, const uint8_t* qdesc, int NQ, const void* kpsUn_, const uint8_t* desc, int N,
                                const int* gridOff, const int* gridIdx, const float* bounds, const float* uRight, uint8_t* occupied,
                                int ratioMode, float nnratio, int maxDistance, int checkOrientation, int* matches) {
    const ProjQuery* q = (const ProjQuery*)queries_;
    const KeyPoint* kUn = (const KeyPoint*)kpsUn_;
    const int TH_HIGH = maxDistance, HISTO_LENGTH = 30;      // ORBmatcher::TH_HIGH = 100 (:36) in (a) and (b); ORBdist in the relocalisation form (:2246)
    std::vector<uint8_t> occ(N, 0);
    if (occupied) occ.assign(occupied, occupied + N);
    int nmatches = 0;
    for (int i = 0; i < N; i++) matches[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < NQ; i++) {
        if (!(q[i].flags & 1)) continue;
        const float radius = q[i].radius;
        const std::vector<size_t> vIndices = getFeaturesInArea(kUn, gridOff, gridIdx, bounds, q[i].u, q[i].v, radius, q[i].minLevel, q[i].maxLevel);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = qdesc + (size_t)i * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t idx : vIndices) {
            if (occ[idx]) continue;                                              // :64-66 / :2035-2037
            if (uRight && uRight[idx] > 0) {                                     // :68-73 / :2039-2046
                const float er = std::fabs(q[i].ur - uRight[idx]);
                if (er > radius) continue;
            }
            const int dist = descriptorDistance(dMP, desc + idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kUn[idx].octave; bestIdx = (int)idx; }
            else if (ratioMode && dist < bestDist2) { bestLevel2 = kUn[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (ratioMode && bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;          // :102-104
            matches[bestIdx] = i;                                                // F.mvpMapPoints[bestIdx] = pMP
            if (q[i].flags & 2) occ[bestIdx] = 1;                                // the point now sitting there has observations
            else occ[bestIdx] = 0;                                               // ... or has none (a temporal stereo point): later queries may take the keypoint over
            nmatches++;
            if (!ratioMode && checkOrientation) {
                float rot = q[i].angle - kUn[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx);
            }
        }
    }
    if (!ratioMode && checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        computeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { matches[idx] = -1; occ[idx] = 0; nmatches--; }   // :2166-2170: mvpMapPoints[idx] = NULL (no check that it
                                                                                        // is still set: the count may be decremented twice)
        }
    }
    if (occupied) std::copy(occ.begin(), occ.end(), occupied);
    return nmatches;
}

// ---------------------------------------------------------------------------------------------
// "Next" row: Frame::ComputeBoW (src/Frame.cc:739-746) = DBoW2::TemplatedVocabulary<FORB>::transform with
// levelsup = 4, restated from the DBoW2 sources vendored with the reference (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1196
// and 1218-1262, BowVector.cpp:34-83, FeatureVector.cpp:31-46, FORB.cpp distance).  The vocabulary arrives as the arrays the text
// loader builds (:1338-1423): node n >= 1 with parent[n] < n, children in node order, word ids to the leaves in node order.
// Vocabulary/ORBvoc.txt itself is absent from the reference (.MISSING_LARGE_BLOBS:1); the tests use synthetic trees.
// ---------------------------------------------------------------------------------------------
struct OracleVocab {
    int k, L, scoring, weighting;
    std::vector<std::vector<int>> children;
    std::vector<const uint8_t*> desc;
    std::vector<double> weight;
    std::vector<unsigned> wordId;
};

// returns the number of words; wordIds / wordWeights hold the BowVector in key order; fvNodes / fvIdx the FeatureVector flattened in
// (node, feature) order with *nFeat entries
int oracle_compute_bow(int k, int L, int scoring, int weighting, int nNodes, const int* parent, const uint8_t* isLeaf, const uint8_t* nodeDesc,
                       const double* weight, const uint8_t* desc, int N, int levelsup, unsigned* wordIds, double* wordWeights,
                       unsigned* fvNodes, unsigned* fvIdx, int* nFeat) {
    OracleVocab V{k, L, scoring, weighting, {}, {}, {}, {}};
    V.children.resize(nNodes); V.desc.resize(nNodes); V.weight.assign(weight, weight + nNodes); V.wordId.assign(nNodes, 0);
    unsigned words = 0;
    for (int n = 0; n < nNodes; n++) {
        V.desc[n] = nodeDesc + (size_t)n * 32;
        if (n > 0) { V.children[parent[n]].push_back(n); if (isLeaf[n]) V.wordId[n] = words++; }
    }
    std::map<unsigned, double> v;                              // BowVector
    std::map<unsigned, std::vector<unsigned>> fv;              // FeatureVector
    const bool must = scoring != 5;                            // ScoringObject.h:74-89: every scoring but DOT_PRODUCT normalises ...
    const bool l2 = scoring == 1;                              // ... L2_NORM with L2, the others with L1
    for (int i = 0; i < N; i++) {
        // transform(feature, word_id, weight, nid, levelsup)  (:1218-1262)
        const uint8_t* feature = desc + (size_t)i * 32;
        const int nid_level = L - levelsup;
        unsigned nid = 0;                                      // root if nid_level <= 0 (:1227); the reference leaves it unset when the
                                                               // leaf is shallower than nid_level (never in a full tree): 0 here
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const std::vector<int>& nodes = V.children[final_id];
            final_id = nodes[0];
            double best_d = (double)descriptorDistance(feature, V.desc[final_id]);
            for (size_t c = 1; c < nodes.size(); c++) {
                const double d = (double)descriptorDistance(feature, V.desc[nodes[c]]);
                if (d < best_d) { best_d = d; final_id = nodes[c]; }
            }
            if (current_level == nid_level) nid = (unsigned)final_id;
        } while (!V.children[final_id].empty());
        const unsigned id = V.wordId[final_id];
        const double w = V.weight[final_id];
        if (w > 0) {                                           // not stopped (:1161, :1187)
            if (weighting == 0 || weighting == 1) {            // TF_IDF, TF: BowVector::addWeight
                auto it = v.find(id);
                if (it != v.end()) it->second += w; else v[id] = w;
            } else if (!v.count(id)) v[id] = w;                // IDF, BINARY: addIfNotExist
            fv[nid].push_back((unsigned)i);
        }
    }
    if ((weighting == 0 || weighting == 1) && !v.empty() && !must) {      // :1170-1176
        const double nd = (double)v.size();
        for (auto& e : v) e.second /= nd;
    }
    if (must) {                                                // BowVector::normalize (BowVector.cpp:62-83)
        double norm = 0.0;
        if (!l2) for (auto& e : v) norm += std::fabs(e.second);
        else { for (auto& e : v) norm += e.second * e.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto& e : v) e.second /= norm;
    }
    int nw = 0;
    for (auto& e : v) { wordIds[nw] = e.first; wordWeights[nw] = e.second; nw++; }
    int nf = 0;
    for (auto& e : fv) for (unsigned i : e.second) { fvNodes[nf] = e.first; fvIdx[nf] = i; nf++; }
    *nFeat = nf;
    return nw;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:269-471), Nleft == -1
// (and pKF->mpCamera2 == NULL: kp = pKF->mvKeysUn[realIdxKF], whose angle is mvKeys' angle, src/Frame.cc:776-780).
// The FeatureVectors arrive flattened in (node, feature index) order, as oracle_compute_bow writes them, and are rebuilt into the
// std::map<NodeId, vector<unsigned>> the reference walks.  kfFlags[i] bit 0 = "pMP && !pMP->isBad()" (:301-307).
// matches[iF] (out, NF ints) = the keyframe keypoint whose MapPoint F's keypoint iF receives, -1 = NULL.  Returns nmatches.
int oracle_search_by_bow(const unsigned* kfNodes, const unsigned* kfIdx, int nKF, const unsigned* fNodes, const unsigned* fIdx, int nF,
                         const uint8_t* kfFlags, const void* kpsKF_, const uint8_t* descKF, const void* kpsF_, const uint8_t* descF, int NF,
                         float nnratio, int thLow, int checkOrientation, int* matches) {
    const KeyPoint* kpsKF = (const KeyPoint*)kpsKF_;
    const KeyPoint* kpsF = (const KeyPoint*)kpsF_;
    std::map<unsigned, std::vector<unsigned>> vFeatVecKF, vFeatVecF;
    for (int i = 0; i < nKF; i++) vFeatVecKF[kfNodes[i]].push_back(kfIdx[i]);
    for (int i = 0; i < nF; i++) vFeatVecF[fNodes[i]].push_back(fIdx[i]);
    for (int i = 0; i < NF; i++) matches[i] = -1;                                        // :273
    int nmatches = 0;
    const int HISTO_LENGTH = 30;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;                                            // :282
    auto KFit = vFeatVecKF.begin(), KFend = vFeatVecKF.end();
    auto Fit = vFeatVecF.begin(), Fend = vFeatVecF.end();
    while (KFit != KFend && Fit != Fend) {                                               // :290
        if (KFit->first == Fit->first) {
            const std::vector<unsigned>& vIndicesKF = KFit->second;
            const std::vector<unsigned>& vIndicesF = Fit->second;
            for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
                const unsigned realIdxKF = vIndicesKF[iKF];
                if (!(kfFlags[realIdxKF] & 1)) continue;                                 // :303-307
                const uint8_t* dKF = descKF + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
                    const unsigned realIdxF = vIndicesF[iF];
                    if (matches[realIdxF] >= 0) continue;                                // :318-319
                    const int dist = descriptorDistance(dKF, descF + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }     // :325-330
                    else if (dist < bestDist2) bestDist2 = dist;                                                    // :331-334
                }
                if (bestDist1 <= thLow) {                                                // :375
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {                 // :377
                        matches[bestIdxF] = (int)realIdxKF;
                        if (checkOrientation) {                                          // :384-401
                            float rot = kpsKF[realIdxKF].angle - kpsF[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++; Fit++;
        } else if (KFit->first < Fit->first) KFit = vFeatVecKF.lower_bound(Fit->first);  // :432-435
        else Fit = vFeatVecF.lower_bound(KFit->first);                                   // :436-439
    }
    if (checkOrientation) {                                                              // :445-468
        int ind1 = -1, ind2 = -1, ind3 = -1;
        computeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { matches[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:823-963), NLeft == -1.
// flags1 / flags2 bit 0 = "pMP && !pMP->isBad()" of the two keyframes' keypoints.  matches12[idx1] (out, N1 ints) = the keypoint of pKF2
// whose MapPoint vpMatches12[idx1] names, -1 = NULL.  Returns nmatches.
int oracle_search_by_bow_keyframes(const unsigned* nodes1, const unsigned* idx1_, int n1, const unsigned* nodes2, const unsigned* idx2_, int n2,
                                   const uint8_t* flags1, const uint8_t* flags2, const void* kps1_, const uint8_t* desc1, int N1, const void* kps2_,
                                   const uint8_t* desc2, int N2, float nnratio, int thLow, int checkOrientation, int* matches12) {
    const KeyPoint* vKeysUn1 = (const KeyPoint*)kps1_;
    const KeyPoint* vKeysUn2 = (const KeyPoint*)kps2_;
    std::map<unsigned, std::vector<unsigned>> vFeatVec1, vFeatVec2;
    for (int i = 0; i < n1; i++) vFeatVec1[nodes1[i]].push_back(idx1_[i]);
    for (int i = 0; i < n2; i++) vFeatVec2[nodes2[i]].push_back(idx2_[i]);
    for (int i = 0; i < N1; i++) matches12[i] = -1;                                      // :835
    std::vector<bool> vbMatched2(N2, false);                                             // :836
    const int HISTO_LENGTH = 30;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;                                            // :842
    int nmatches = 0;
    auto f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
    auto f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
    while (f1it != f1end && f2it != f2end) {                                             // :851
        if (f1it->first == f2it->first) {
            for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
                const size_t idx1 = f1it->second[i1];
                if (!(flags1[idx1] & 1)) continue;                                       // :862-866
                const uint8_t* d1 = desc1 + idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
                    const size_t idx2 = f2it->second[i2];
                    if (vbMatched2[idx2] || !(flags2[idx2] & 1)) continue;               // :884-888
                    const int dist = descriptorDistance(d1, desc2 + idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }     // :894-899
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < thLow) {                                                 // :906 (strict, unlike :375)
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        matches12[idx1] = bestIdx2;                                      // vpMatches12[idx1] = vpMapPoints2[bestIdx2]
                        vbMatched2[bestIdx2] = true;
                        if (checkOrientation) {
                            float rot = vKeysUn1[idx1].angle - vKeysUn2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back((int)idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++; f2it++;
        } else if (f1it->first < f2it->first) f1it = vFeatVec1.lower_bound(f2it->first);
        else f2it = vFeatVec2.lower_bound(f1it->first);
    }
    if (checkOrientation) {                                                              // :940-960
        int ind1 = -1, ind2 = -1, ind3 = -1;
        computeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// cv::cvtColor(RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) for 8-bit images as Tracking::GrabImage* calls it
// (src/Tracking.cc:915-941, 985-1001).  OpenCV 3.4 generic path: 14-bit fixed point with R2Y = 4899, G2Y = 9617,
// B2Y = 1868 and CV_DESCALE's rounding; alpha ignored.
void oracle_gray_from_color(const uint8_t* src, int rows, int cols, int channels, int redFirst, long srcStride, uint8_t* dst, long dstStride) {
    const int yuv_shift = 14, R2Y = 4899, G2Y = 9617, B2Y = 1868;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const uint8_t* p = src + (size_t)y * srcStride + (size_t)x * channels;
            const int r = redFirst ? p[0] : p[2], g = p[1], b = redFirst ? p[2] : p[0];
            dst[(size_t)y * dstStride + x] = (uint8_t)((b * B2Y + g * G2Y + r * R2Y + (1 << (yuv_shift - 1))) >> yuv_shift);
        }
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:994-1015) preceded by Tracking::GrabImageRGBD's depth conversion
// (src/Tracking.cc:1003-1004: convertTo(CV_32F, mDepthMapFactor) unless the map is float and the factor is 1).
void oracle_stereo_from_rgbd(const void* kps_, const void* kpsUn_, int N, const void* depth, int isU16, int rows, int cols, long strideBytes,
                             float mDepthMapFactor, float mbf, float* mvuRight, float* mvDepth) {
    const KeyPoint* mvKeys = (const KeyPoint*)kps_;
    const KeyPoint* mvKeysUn = (const KeyPoint*)kpsUn_;
    const bool convert = (std::fabs(mDepthMapFactor - 1.0f) > 1e-5) || isU16;
    (void)rows; (void)cols;
    for (int i = 0; i < N; i++) {
        mvuRight[i] = -1; mvDepth[i] = -1;
        const float v = mvKeys[i].y, u = mvKeys[i].x;
        const uint8_t* row = (const uint8_t*)depth + (size_t)(int)v * strideBytes;       // Mat::at<float>(int, int) from float arguments
        float d = isU16 ? (float)((const unsigned short*)row)[(int)u] : ((const float*)row)[(int)u];
        if (convert) d = d * mDepthMapFactor;                                            // cvtScale: one float multiply
        if (d > 0) { mvDepth[i] = d; mvuRight[i] = mvKeysUn[i].x - mbf / d; }
    }
}

// ---- CPU baseline: nframes extractions over nthreads host threads (one extractor per thread,
// the reference's own execution model per Frame.cc:109-112); returns wall seconds. ------------
double oracle_time_frames(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                          const uint8_t* frames, int nframes, int rows, int cols, int lap0, int lap1,
                          int nthreads, long* total_keypoints) {
    if (nthreads < 1) nthreads = 1;
    std::vector<long> counts(nthreads, 0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
        th.emplace_back([&, t]() {
            Oracle o(nfeatures, scaleFactor, nlevels, iniTh, minTh);
            int cap = nfeatures + 3 * nlevels + 64;
            std::vector<KeyPoint> k(cap);
            std::vector<uint8_t> d((size_t)cap * 32);
            for (int f = t; f < nframes; f += nthreads) {
                Gray g; g.w = cols; g.h = rows; g.stride = cols; g.p = frames + (size_t)f * rows * cols;
                int mono;
                int n = o.extract(g, lap0, lap1, k.data(), d.data(), cap, &mono);
                if (n > 0) counts[t] += n;
            }
        });
    }
    for (auto& x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    long tot = 0;
    for (long c : counts) tot += c;
    if (total_keypoints) *total_keypoints = tot;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
