// orbx_frame_latency.cpp — the call a maintainer's Frame::ExtractORB makes (reference src/Frame.cc:419-427), timed in C++: the drop-in class
// (include/orbx_extractor.hpp, here over minimal cv-like types: this image has no OpenCV) called with a PAGEABLE host image, keypoints and
// descriptors delivered into the caller's std::vector / Mat, every call waited for.  No Python, no ctypes.
//
//   build:  g++ -O2 -std=c++17 -Iinclude examples/orbx_frame_latency.cpp -o orbx_frame_latency -Lextractorb_amd -lorbx -Wl,-rpath,$PWD/extractorb_amd
//   run:    ./orbx_frame_latency [rows=480] [cols=640] [nfeatures=1000] [calls=400]
// prints: median / 10th percentile microseconds per call, and n of the last call
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbx_extractor.hpp"

namespace mini {
struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct Mat {
    int rows = 0, cols = 0; ptrdiff_t step = 0; std::vector<uint8_t> buf;
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
    static void createU8(Mat& m, int r, int c) { m.rows = r; m.cols = c; m.step = c; m.buf.resize((size_t)r * c); }      // (like Mat::create: no reallocation when the size fits)
    static void release(Mat& m) { m = Mat(); }
    static uint8_t* mutableData(Mat& m) { return m.buf.data(); }
    static Mat wrapBordered(const uint8_t* s, int r, int c, ptrdiff_t step, int) {
        Mat m; createU8(m, r, c);
        for (int y = 0; y < r; y++) std::memcpy(m.buf.data() + (size_t)y * c, s + (ptrdiff_t)y * step, c);
        return m;
    }
};
}  // namespace mini

static uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? std::atoi(argv[1]) : 480, cols = argc > 2 ? std::atoi(argv[2]) : 640;
    const int nf = argc > 3 ? std::atoi(argv[3]) : 1000, calls = argc > 4 ? std::atoi(argv[4]) : 400;
    try {
        orbx::BasicORBextractor<mini::Traits> extractor(nf, 1.2f, 8, 20, 7, cols, rows);
        std::vector<mini::Mat> images(8);      // eight different pageable frames in turn, as a camera delivers them
        for (int f = 0; f < 8; f++) {
            mini::Traits::createU8(images[f], rows, cols);
            const uint64_t key = 20261003ull * 0x100000001B3ull + (uint64_t)f * 0xD1B54A32D192ED03ull;
            for (size_t i = 0; i < images[f].buf.size(); i++) images[f].buf[i] = (uint8_t)(splitmix64(key + i) >> 56);
        }
        std::vector<mini::KeyPoint> mvKeys;
        mini::Mat mDescriptors;
        std::vector<int> vLapping = {0, 1000};
        std::vector<std::vector<mini::KeyPoint>> allLevels;
        for (int i = 0; i < 50; i++) extractor(images[i & 7], mini::Mat(), mvKeys, mDescriptors, vLapping, allLevels);
        std::vector<double> us(calls);
        for (int i = 0; i < calls; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            extractor(images[i & 7], mini::Mat(), mvKeys, mDescriptors, vLapping, allLevels);
            us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        std::sort(us.begin(), us.end());
        std::printf("frame_call_us_median=%.1f frame_call_us_p10=%.1f frame_call_us_p90=%.1f keypoints=%zu calls=%d (%dx%d, %d features, pageable image in, host vectors out)\n",
                    us[calls / 2], us[calls / 10], us[calls * 9 / 10], mvKeys.size(), calls, cols, rows, nf);
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
