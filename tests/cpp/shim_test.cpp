// shim_test.cpp — compiles include/orbx_extractor.hpp against minimal cv-like types (this image has no
// OpenCV) and drives the reference call shape of Frame::ExtractORB (reference src/Frame.cc:419-427):
//     monoIndex = (*extractor)(im, Mat(), mvKeys, mDescriptors, vLapping, allLevelsKeypoints);
// usage: shim_test <in.gray> <rows> <cols> <nfeatures> <lap0> <lap1> <out.bin>
// out.bin: int32 mono, int32 n, n x 28-byte keypoints, n x 32 descriptor bytes, nlevels x int32 level counts,
//          then level 3 of mvImagePyramid (int32 w, int32 h, w*h bytes, (w+38)*(h+38) bytes of the bordered buffer around it)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

int main(int argc, char** argv) {
    if (argc != 8) { std::fprintf(stderr, "usage: %s in rows cols nfeatures lap0 lap1 out\n", argv[0]); return 2; }
    const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), nf = std::atoi(argv[4]);
    mini::Mat im; mini::Traits::createU8(im, rows, cols);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(im.buf.data(), 1, im.buf.size(), f) != im.buf.size()) { std::perror("input"); return 2; }
    std::fclose(f);
    try {
        using Extractor = orbx::BasicORBextractor<mini::Traits>;
        // ORBX_SHIM_TEST_SMALL_ARENAS: construct exactly as the reference does (five arguments, default arenas) so that a larger
        // image has to grow them inside operator()
        Extractor* mpORBextractorLeft = std::getenv("ORBX_SHIM_TEST_SMALL_ARENAS") ? new Extractor(nf, 1.2f, 8, 20, 7)
                                                                                   : new Extractor(nf, 1.2f, 8, 20, 7, cols, rows);
        std::vector<mini::KeyPoint> mvKeys;
        mini::Mat mDescriptors;
        std::vector<int> vLapping = {std::atoi(argv[5]), std::atoi(argv[6])};
        std::vector<std::vector<mini::KeyPoint>> allLevelsKeypoints;
        int monoIndex = (*mpORBextractorLeft)(im, mini::Mat(), mvKeys, mDescriptors, vLapping, allLevelsKeypoints);
        // an empty image returns -1 without touching the outputs (ORBextractor.cc:1083-1084)
        std::vector<mini::KeyPoint> k2; mini::Mat d2; std::vector<std::vector<mini::KeyPoint>> a2;
        if ((*mpORBextractorLeft)(mini::Mat(), mini::Mat(), k2, d2, vLapping, a2) != -1) return 3;
        if (mpORBextractorLeft->GetLevels() != 8 || mpORBextractorLeft->GetScaleFactors().size() != 8) return 4;
        // mvImagePyramid is indexed right after the call, with no call in between (Frame.cc:820,910): the member brings the levels itself
        if (mpORBextractorLeft->mvImagePyramid.size() != 8) return 5;
        const mini::Mat& l3 = mpORBextractorLeft->mvImagePyramid[3];
        FILE* o = std::fopen(argv[7], "wb");
        int n = (int)mvKeys.size();
        std::fwrite(&monoIndex, 4, 1, o); std::fwrite(&n, 4, 1, o);
        std::fwrite(mvKeys.data(), sizeof(mini::KeyPoint), n, o);
        std::fwrite(mDescriptors.buf.data(), 1, (size_t)n * 32, o);
        for (int l = 0; l < 8; l++) { int c = (int)allLevelsKeypoints[l].size(); std::fwrite(&c, 4, 1, o); }
        std::fwrite(&l3.cols, 4, 1, o); std::fwrite(&l3.rows, 4, 1, o);
        std::fwrite(l3.buf.data(), 1, l3.buf.size(), o);
        std::fwrite(l3.border.data(), 1, l3.border.size(), o);      // the (w + 38) x (h + 38) buffer the level is a view into
        std::fclose(o);
        delete mpORBextractorLeft;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
