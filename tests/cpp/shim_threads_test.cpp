// shim_threads_test.cpp — the stereo call shape of the reference (src/Frame.cc:109-112): two ORBextractor instances, one per eye,
// each called from its own std::thread at the same time, then joined:
//     thread threadLeft(&Frame::ExtractORB, this, 0, imLeft, 0, 0);  thread threadRight(&Frame::ExtractORB, this, 1, imRight, 0, 0);
// Compiled against the stand-in cv types of shim_test.cpp (this image has no OpenCV).
// usage: shim_threads_test <left.gray> <right.gray> <rows> <cols> <nfeatures> <rounds> <out.bin>
// out.bin, per eye: int32 mono, int32 n, n x 28-byte keypoints, n x 32 descriptor bytes   (results of the LAST round; every round
// must reproduce the first one bit for bit or the program exits with 5)
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "orbx_extractor.hpp"

namespace mini {
struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct Mat {
    int rows = 0, cols = 0; ptrdiff_t step = 0; std::vector<uint8_t> buf, border;
    bool empty() const { return rows == 0 || cols == 0; }
};
struct Traits {
    using KeyPoint = mini::KeyPoint;
    using Mat = mini::Mat;
    static bool empty(const Mat& m) { return m.empty(); }
    static const uint8_t* data(const Mat& m) { return m.buf.data(); }
    static int rows(const Mat& m) { return m.rows; }
    static int cols(const Mat& m) { return m.cols; }
    static ptrdiff_t step(const Mat& m) { return m.step; }
    static bool isU8C1(const Mat&) { return true; }
    static void createU8(Mat& m, int r, int c) { m.rows = r; m.cols = c; m.step = c; m.buf.assign((size_t)r * c, 0); }
    static void release(Mat& m) { m = Mat(); }
    static uint8_t* mutableData(Mat& m) { return m.buf.data(); }
    // mini::Mat has no views: the level's pixels are copied out of the bordered buffer, the border itself into `frame` (rows above, ..)
    static Mat wrapBordered(const uint8_t* s, int r, int c, ptrdiff_t step, int b) {
        Mat m; createU8(m, r, c);
        for (int y = 0; y < r; y++) std::memcpy(m.buf.data() + (size_t)y * c, s + (ptrdiff_t)y * step, c);
        m.border.assign((size_t)(r + 2 * b) * (c + 2 * b), 0);
        for (int y = -b; y < r + b; y++) std::memcpy(m.border.data() + (size_t)(y + b) * (c + 2 * b), s + (ptrdiff_t)y * step - b, c + 2 * b);
        return m;
    }
};
}  // namespace mini

using Extractor = orbx::BasicORBextractor<mini::Traits>;

struct Eye {                      // what Frame keeps per eye
    Extractor* extractor = nullptr;
    mini::Mat image;
    std::vector<mini::KeyPoint> mvKeys;
    mini::Mat mDescriptors;
    int monoIndex = 0;
    std::vector<std::vector<mini::KeyPoint>> allLevels;      // the reference shares ONE such vector between both threads (a data race,
                                                             // SURVEY.md §3.2); a per-eye vector is what a fixed caller would pass
};

static std::atomic<int> g_failed{0};

static void ExtractORB(Eye* e) {   // Frame::ExtractORB (src/Frame.cc:419-427)
    try {
        std::vector<int> vLapping = {0, 0};          // rectified stereo passes {0, 0} (Frame.cc:109-110)
        e->monoIndex = (*e->extractor)(e->image, mini::Mat(), e->mvKeys, e->mDescriptors, vLapping, e->allLevels);
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "error: %s\n", ex.what());
        g_failed = 1;
    }
}

int main(int argc, char** argv) {
    if (argc != 8) { std::fprintf(stderr, "usage: %s left right rows cols nfeatures rounds out\n", argv[0]); return 2; }
    const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), nf = std::atoi(argv[5]), rounds = std::atoi(argv[6]);
    Eye eye[2];
    for (int i = 0; i < 2; i++) {
        mini::Traits::createU8(eye[i].image, rows, cols);
        FILE* f = std::fopen(argv[1 + i], "rb");
        if (!f || std::fread(eye[i].image.buf.data(), 1, eye[i].image.buf.size(), f) != eye[i].image.buf.size()) { std::perror("input"); return 2; }
        std::fclose(f);
    }
    try {
        // mpORBextractorLeft / mpORBextractorRight (src/Tracking.cc:768-771); the arenas start SMALLER than the image, so the first
        // call of each thread also exercises the grow-on-demand path (the reference takes any image size)
        eye[0].extractor = new Extractor(nf, 1.2f, 8, 20, 7, 320, 240);
        eye[1].extractor = new Extractor(nf, 1.2f, 8, 20, 7, 320, 240);
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    std::vector<mini::KeyPoint> firstK[2]; std::vector<uint8_t> firstD[2]; int firstMono[2] = {0, 0};
    for (int r = 0; r < rounds; r++) {
        std::thread threadLeft(ExtractORB, &eye[0]);
        std::thread threadRight(ExtractORB, &eye[1]);
        threadLeft.join();
        threadRight.join();
        if (g_failed) return 1;
        for (int i = 0; i < 2; i++) {
            if (r == 0) { firstK[i] = eye[i].mvKeys; firstD[i] = eye[i].mDescriptors.buf; firstMono[i] = eye[i].monoIndex; continue; }
            if (eye[i].monoIndex != firstMono[i] || eye[i].mvKeys.size() != firstK[i].size() ||
                std::memcmp(eye[i].mvKeys.data(), firstK[i].data(), firstK[i].size() * sizeof(mini::KeyPoint)) != 0 ||
                eye[i].mDescriptors.buf != firstD[i]) {
                std::fprintf(stderr, "round %d eye %d differs from round 0\n", r, i);
                return 5;
            }
        }
    }
    FILE* o = std::fopen(argv[7], "wb");
    for (int i = 0; i < 2; i++) {
        int n = (int)eye[i].mvKeys.size();
        std::fwrite(&eye[i].monoIndex, 4, 1, o); std::fwrite(&n, 4, 1, o);
        std::fwrite(eye[i].mvKeys.data(), sizeof(mini::KeyPoint), n, o);
        std::fwrite(eye[i].mDescriptors.buf.data(), 1, (size_t)n * 32, o);
    }
    std::fclose(o);
    delete eye[0].extractor; delete eye[1].extractor;
    return 0;
}
