// orbx_extractor.hpp — header-only C++ shim that rebuilds ORB_SLAM3::ORBextractor
// (reference inc/ORBextractor.h:44-111) on top of the C ABI of liborbx.so (include/orbx.h).
//
// With OpenCV available (the reference's own build), include this header INSTEAD of
// inc/ORBextractor.h and link liborbx.so instead of compiling src/orb_extractor/ORBextractor.cc:
// Frame::ExtractORB (src/Frame.cc:419-427) and every getter call site compile unchanged.
//
//     #include <opencv2/core/core.hpp>
//     #include "orbx_extractor.hpp"
//
// The shim only needs four things from the cv namespace, so it is a template over a small traits
// struct; `orbx::CvTraits` below binds it to real OpenCV types when <opencv2/core.hpp> was included
// first.  tests/cpp/shim_test.cpp binds it to tiny stand-in types so the shim itself is compiled
// and exercised in an image without OpenCV.
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"

namespace orbx {

// Traits contract:
//   using KeyPoint  : trivially copyable, layout-identical to orbx_keypoint (28 bytes)
//   using Mat       : image/descriptor container
//   static bool            empty(const Mat&)
//   static const uint8_t*  data(const Mat&); static int rows(const Mat&); static int cols(const Mat&);
//   static ptrdiff_t       step(const Mat&)
//   static bool            isU8C1(const Mat&)
//   static void            createU8(Mat&, int rows, int cols)   // like _descriptors.create(n, 32, CV_8U)
//   static void            release(Mat&)
//   static uint8_t*        mutableData(Mat&)
//   static Mat             wrapCopy(const uint8_t* src, int rows, int cols, ptrdiff_t step)  // owning copy
template <class Traits>
class BasicORBextractor {
public:
    using KeyPoint = typename Traits::KeyPoint;
    using Mat = typename Traits::Mat;
    static_assert(sizeof(KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be 28 bytes");

    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };   // inc/ORBextractor.h:48

    // inc/ORBextractor.h:50-51.  The reference takes images of any size; max_* only pre-size the device arenas (one
    // allocation for the usual camera): a larger image makes operator() re-create them at the new size (Reserve()).
    BasicORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_,
                      int max_width = 1920, int max_height = 1080, int device = -1)
        : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_),
          minThFAST(minThFAST_), device_(device) {
        Reserve(max_width, max_height);
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
        orbx_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                        mnFeaturesPerLevel.data(), umax.data());
        mvImagePyramid.resize(nlevels);
    }
    ~BasicORBextractor() { orbx_destroy(h_); }

    // (Re-)creates the device arenas for images up to width x height (never shrinks).  Called by the constructor and by
    // operator() when an image exceeds the current size; call it yourself to keep the allocation out of the first frame.
    void Reserve(int width, int height) {
        if (h_ && width <= maxW_ && height <= maxH_) return;
        const int w = width > maxW_ ? width : maxW_, hgt = height > maxH_ ? height : maxH_;
        orbx_handle* nh = nullptr;
        int rc = orbx_create(&nh, nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, w, hgt, 1, device_);
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_create: ") + orbx_last_error(nullptr));
        orbx_destroy(h_);
        h_ = nh; maxW_ = w; maxH_ = hgt;
        capacity_ = orbx_max_keypoints(h_);
        kbuf_.resize(capacity_); lbuf_.resize(capacity_); dbuf_.resize((size_t)capacity_ * 32);
        pyramidStale_ = true; lastRows_ = lastCols_ = 0;
    }
    BasicORBextractor(const BasicORBextractor&) = delete;
    BasicORBextractor& operator=(const BasicORBextractor&) = delete;

    // inc/ORBextractor.h:58-61 / ORBextractor.cc:1078-1162.  `mask` is ignored, as in the reference.
    int operator()(const Mat& image, const Mat& /*mask*/, std::vector<KeyPoint>& keypoints, Mat& descriptors,
                   std::vector<int>& vLappingArea, std::vector<std::vector<KeyPoint>>& allLevelsKeypoints) {
        if (Traits::empty(image)) return -1;                                     // :1083-1084
        if (!Traits::isU8C1(image)) throw std::invalid_argument("ORBextractor: image.type() != CV_8UC1");   // assert :1087
        Reserve(Traits::cols(image), Traits::rows(image));                       // the reference has no size limit
        int n = 0, mono = 0;
        std::vector<int> counts(nlevels);
        int rc = orbx_extract(h_, Traits::data(image), Traits::rows(image), Traits::cols(image), Traits::step(image),
                              vLappingArea.at(0), vLappingArea.at(1), reinterpret_cast<orbx_keypoint*>(kbuf_.data()),
                              dbuf_.data(), capacity_, &n, &mono, reinterpret_cast<orbx_keypoint*>(lbuf_.data()), counts.data());
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error(h_));
        keypoints.assign(kbuf_.begin(), kbuf_.begin() + n);                      // _keypoints = vector<KeyPoint>(nkeypoints) :1112
        if (n == 0) Traits::release(descriptors);                                // :1102-1103
        else {
            Traits::createU8(descriptors, n, 32);                                // :1106
            std::memcpy(Traits::mutableData(descriptors), dbuf_.data(), (size_t)n * 32);
        }
        allLevelsKeypoints.assign(nlevels, std::vector<KeyPoint>());             // :1094
        for (int l = 0, o = 0; l < nlevels; o += counts[l], l++)
            allLevelsKeypoints[l].assign(lbuf_.begin() + o, lbuf_.begin() + o + counts[l]);
        pyramidStale_ = true;
        lastRows_ = Traits::rows(image); lastCols_ = Traits::cols(image);
        return mono;                                                             // :1161
    }

    int GetLevels() { return nlevels; }                                          // inc/ORBextractor.h:63-83
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid is a public member of the reference (inc/ORBextractor.h:85) that
    // Frame::ComputeStereoMatches reads after the call (src/Frame.cc:820,910,929).  The pyramid lives in HBM;
    // call FetchImagePyramid() before touching the member (one D2H copy per level, only for callers that need it).
    std::vector<Mat> mvImagePyramid;
    void FetchImagePyramid() {
        if (!pyramidStale_ || lastRows_ == 0) return;
        std::vector<int> ws(nlevels), hs(nlevels);
        orbx_compute_level_sizes((float)scaleFactor, nlevels, lastRows_, lastCols_, ws.data(), hs.data());
        std::vector<uint8_t> tmp;
        for (int l = 0; l < nlevels; l++) {
            int w = 0, hgt = 0;
            tmp.resize((size_t)ws[l] * hs[l]);
            int rc = orbx_get_level(h_, 0, l, 0, tmp.data(), ws[l], &w, &hgt);
            if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_get_level: ") + orbx_last_error(h_));
            mvImagePyramid[l] = Traits::wrapCopy(tmp.data(), hgt, w, ws[l]);
        }
        pyramidStale_ = false;
    }

    // the remaining public data members of the reference class (inc/ORBextractor.h:95-110)
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

    orbx_handle* handle() { return h_; }

private:
    orbx_handle* h_ = nullptr;
    int device_ = -1, maxW_ = 0, maxH_ = 0;
    int lastRows_ = 0, lastCols_ = 0;
    int capacity_ = 0;
    bool pyramidStale_ = true;
    std::vector<KeyPoint> kbuf_, lbuf_;
    std::vector<uint8_t> dbuf_;
};

}  // namespace orbx

// real OpenCV was included before this header: bind the reference's names.  The include guard of <opencv2/core.hpp> is
// OPENCV_CORE_HPP from 3.2 on and __OPENCV_CORE_HPP__ in 3.0 / 3.1; the reference pins only "OpenCV 3" (CMakeLists.txt:23)
#if defined(OPENCV_CORE_HPP) || defined(__OPENCV_CORE_HPP__)
namespace orbx {
struct CvTraits {
    using KeyPoint = cv::KeyPoint;
    using Mat = cv::Mat;
    static bool empty(const Mat& m) { return m.empty(); }
    static const uint8_t* data(const Mat& m) { return m.data; }
    static int rows(const Mat& m) { return m.rows; }
    static int cols(const Mat& m) { return m.cols; }
    static ptrdiff_t step(const Mat& m) { return (ptrdiff_t)m.step; }
    static bool isU8C1(const Mat& m) { return m.type() == CV_8UC1; }
    static void createU8(Mat& m, int r, int c) { m.create(r, c, CV_8U); }
    static void release(Mat& m) { m.release(); }
    static uint8_t* mutableData(Mat& m) { return m.data; }
    static Mat wrapCopy(const uint8_t* s, int r, int c, ptrdiff_t step) { return Mat(r, c, CV_8UC1, (void*)s, (size_t)step).clone(); }
};
}  // namespace orbx
namespace ORB_SLAM3 {
using ORBextractor = orbx::BasicORBextractor<orbx::CvTraits>;   // drop-in name (inc/ORBextractor.h:44)
}
#endif
