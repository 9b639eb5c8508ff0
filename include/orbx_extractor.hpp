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
//   static Mat             wrapBordered(const uint8_t* src, int rows, int cols, ptrdiff_t step, int border)
//                          // owning copy of the (rows + 2*border) x (cols + 2*border) buffer around src (= pixel (0,0)), returned as the
//                          // rows x cols view into it: what mvImagePyramid[level] is in the reference (ORBextractor.cc:1173-1177)
// Optional (the reference's exact operator() signature, inc/ORBextractor.h:58-61; present in CvTraits):
//   using InputArray / OutputArray                                  // cv::InputArray, cv::OutputArray (references to the proxy classes)
//   static Mat             getMat(InputArray)                       // _image.getMat()                    ORBextractor.cc:1086
//   static uint8_t*        createOut(OutputArray, int rows, int cols)   // _descriptors.create(n, 32, CV_8U); the data of getMat()   :1106-1107
//   static void            releaseOut(OutputArray)                  // _descriptors.release()             :1103
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
        mvImagePyramid.resize(nlevels);
    }
    BasicORBextractor(const BasicORBextractor&) = delete;
    BasicORBextractor& operator=(const BasicORBextractor&) = delete;

    // inc/ORBextractor.h:58-61 / ORBextractor.cc:1078-1162.  `mask` is ignored, as in the reference.
    // One copy per output: the results come out of the handle's pinned slab straight into the caller's containers; no allocation in here
    // beyond what the caller's own vectors / Mat need to grow.
    int operator()(const Mat& image, const Mat& /*mask*/, std::vector<KeyPoint>& keypoints, Mat& descriptors,
                   std::vector<int>& vLappingArea, std::vector<std::vector<KeyPoint>>& allLevelsKeypoints) {
        return extract(image, keypoints, vLappingArea, &allLevelsKeypoints, [&](int n) -> uint8_t* {
            if (n == 0) { Traits::release(descriptors); return nullptr; }        // :1102-1103
            Traits::createU8(descriptors, n, 32);                                // :1106
            return Traits::mutableData(descriptors);
        });
    }
    // the tutorial clone's form (inc/ORBExtractor.h:55-56, src/orb_extractor/ORBExtractor.cpp): no allLevelsKeypoints
    int operator()(const Mat& image, const Mat& /*mask*/, std::vector<KeyPoint>& keypoints, Mat& descriptors, std::vector<int>& vLappingArea) {
        return extract(image, keypoints, vLappingArea, nullptr, [&](int n) -> uint8_t* {
            if (n == 0) { Traits::release(descriptors); return nullptr; }
            Traits::createU8(descriptors, n, 32);
            return Traits::mutableData(descriptors);
        });
    }
    // The reference's exact signature (inc/ORBextractor.h:58-61): cv::InputArray / cv::OutputArray, for callers that pass cv::noArray() as the
    // mask, a std::vector<uchar> or a cv::UMat.  A template so that it only exists for traits that name the proxy types (CvTraits does); with
    // cv::Mat arguments overload resolution still takes the Mat form above (exact match, no proxy objects).
    template <class T = Traits>
    int operator()(typename T::InputArray _image, typename T::InputArray /*_mask*/, std::vector<KeyPoint>& keypoints,
                   typename T::OutputArray _descriptors, std::vector<int>& vLappingArea,
                   std::vector<std::vector<KeyPoint>>& allLevelsKeypoints) {
        const Mat image = T::getMat(_image);                                     // Mat image = _image.getMat(); :1086
        return extract(image, keypoints, vLappingArea, &allLevelsKeypoints, [&](int n) -> uint8_t* {
            if (n == 0) { T::releaseOut(_descriptors); return nullptr; }         // :1102-1103
            return T::createOut(_descriptors, n, 32);                            // _descriptors.create(nkeypoints, 32, CV_8U); :1106
        });
    }
    template <class T = Traits>
    int operator()(typename T::InputArray _image, typename T::InputArray /*_mask*/, std::vector<KeyPoint>& keypoints,
                   typename T::OutputArray _descriptors, std::vector<int>& vLappingArea) {      // inc/ORBExtractor.h:55-56
        const Mat image = T::getMat(_image);
        return extract(image, keypoints, vLappingArea, nullptr, [&](int n) -> uint8_t* {
            if (n == 0) { T::releaseOut(_descriptors); return nullptr; }
            return T::createOut(_descriptors, n, 32);
        });
    }

    // inc/ORBextractor.h:87-90: the reference comments `protected:` out so that its demos can call the two stages themselves
    // (src/orb_extractor/main_orb_extractor.cpp:43-46).  ComputePyramid fills mvImagePyramid (here: the pyramid in HBM, fetched on first
    // access); ComputeKeyPointsOctTree works on the pyramid the extractor holds and hands out level coordinates with angles (:773-888).
    void ComputePyramid(Mat image) {
        if (Traits::empty(image)) throw std::invalid_argument("ORBextractor::ComputePyramid: empty image");
        if (!Traits::isU8C1(image)) throw std::invalid_argument("ORBextractor: image.type() != CV_8UC1");
        Reserve(Traits::cols(image), Traits::rows(image));
        int rc = orbx_compute_pyramid(h_, Traits::data(image), Traits::rows(image), Traits::cols(image), Traits::step(image));
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_compute_pyramid: ") + orbx_last_error(h_));
        mvImagePyramid.invalidate();
    }
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
        lbuf_.resize(capacity_);
        counts_.resize(nlevels);
        int rc = orbx_compute_keypoints_octree(h_, reinterpret_cast<orbx_keypoint*>(lbuf_.data()), capacity_, counts_.data());
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_compute_keypoints_octree: ") + orbx_last_error(h_));
        allKeypoints.resize(nlevels);                                            // :775
        for (int l = 0, o = 0; l < nlevels; o += counts_[l], l++) allKeypoints[l].assign(lbuf_.begin() + o, lbuf_.begin() + o + counts_[l]);
    }

    int GetLevels() { return nlevels; }                                          // inc/ORBextractor.h:63-83
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid is a public member of the reference (inc/ORBextractor.h:85) that Frame::ComputeStereoMatches indexes right after the
    // call (src/Frame.cc:820,910,924,929).  Here the pyramid lives in HBM, so the member is a view that behaves like the reference's
    // std::vector<cv::Mat> for its readers — operator[], at, size, begin / end, conversion to const std::vector<Mat>& — and brings the levels
    // to the host on the FIRST access after a call (one device-to-host copy of the whole bordered pyramid, orbx_fetch_pyramid; nothing is
    // copied for callers that never look).  Each level is, as in the reference (ORBextractor.cc:1173-1177), a w x h view into its own
    // (w + 38) x (h + 38) buffer with the BORDER_REFLECT_101 frame around it, and owns that buffer (a Mat taken from it stays valid).
    class ImagePyramid {
    public:
        explicit ImagePyramid(BasicORBextractor* owner) : owner_(owner) {}
        Mat& operator[](size_t l) { fetch(); return levels_[l]; }
        const Mat& operator[](size_t l) const { fetch(); return levels_[l]; }
        Mat& at(size_t l) { fetch(); return levels_.at(l); }
        const Mat& at(size_t l) const { fetch(); return levels_.at(l); }
        size_t size() const { return levels_.size(); }
        bool empty() const { return levels_.empty(); }
        typename std::vector<Mat>::iterator begin() { fetch(); return levels_.begin(); }
        typename std::vector<Mat>::iterator end() { fetch(); return levels_.end(); }
        typename std::vector<Mat>::const_iterator begin() const { fetch(); return levels_.begin(); }
        typename std::vector<Mat>::const_iterator end() const { fetch(); return levels_.end(); }
        operator const std::vector<Mat>&() const { fetch(); return levels_; }
        void resize(size_t n) { levels_.resize(n); }                             // ORBextractor.cc:437
        void invalidate() { stale_ = true; }
    private:
        void fetch() const { if (stale_) owner_->fetchLevels(levels_); stale_ = false; }
        BasicORBextractor* owner_;
        mutable std::vector<Mat> levels_;
        mutable bool stale_ = false;
    };
    ImagePyramid mvImagePyramid{this};
    void FetchImagePyramid() { (void)mvImagePyramid[0]; }                        // the explicit form (kept from the first version of this header)

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
    // operator() behind every signature: descOut(n) makes room for n descriptors in the caller's container and returns where they go
    template <class DescOut>
    int extract(const Mat& image, std::vector<KeyPoint>& keypoints, std::vector<int>& vLappingArea,
                std::vector<std::vector<KeyPoint>>* allLevelsKeypoints, DescOut descOut) {
        if (Traits::empty(image)) return -1;                                     // :1083-1084
        if (!Traits::isU8C1(image)) throw std::invalid_argument("ORBextractor: image.type() != CV_8UC1");   // assert :1087
        Reserve(Traits::cols(image), Traits::rows(image));                       // the reference has no size limit
        int n = 0, mono = 0;
        const orbx_keypoint *k = nullptr, *lk = nullptr;
        const uint8_t* d = nullptr;
        const int* counts = nullptr;
        int rc = orbx_extract_view(h_, Traits::data(image), Traits::rows(image), Traits::cols(image), Traits::step(image),
                                   vLappingArea.at(0), vLappingArea.at(1), allLevelsKeypoints ? 1 : 0, &k, &d, &n, &mono, &lk, &counts);
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error(h_));
        const KeyPoint* kp = reinterpret_cast<const KeyPoint*>(k);
        keypoints.assign(kp, kp + n);                                            // _keypoints = vector<KeyPoint>(nkeypoints) :1112
        if (uint8_t* out = descOut(n)) std::memcpy(out, d, (size_t)n * 32);
        if (allLevelsKeypoints) {
            const KeyPoint* lp = reinterpret_cast<const KeyPoint*>(lk);
            allLevelsKeypoints->resize(nlevels);                                 // :1094
            for (int l = 0, o = 0; l < nlevels; o += counts[l], l++) (*allLevelsKeypoints)[l].assign(lp + o, lp + o + counts[l]);
        }
        mvImagePyramid.invalidate();
        return mono;                                                             // :1161
    }
    void fetchLevels(std::vector<Mat>& levels) {
        const uint8_t* base = nullptr;
        size_t off[ORBX_MAX_LEVELS];
        int stride[ORBX_MAX_LEVELS], w[ORBX_MAX_LEVELS], hgt[ORBX_MAX_LEVELS];
        int rc = orbx_fetch_pyramid(h_, 0, &base, off, stride, w, hgt);
        if (rc != ORBX_OK) throw std::runtime_error(std::string("mvImagePyramid: ") + orbx_last_error(h_));
        levels.resize(nlevels);
        for (int l = 0; l < nlevels; l++) levels[l] = Traits::wrapBordered(base + off[l], hgt[l], w[l], stride[l], ORBX_EDGE_THRESHOLD);
    }
    orbx_handle* h_ = nullptr;
    int device_ = -1, maxW_ = 0, maxH_ = 0;
    int capacity_ = 0;
    std::vector<KeyPoint> lbuf_;
    std::vector<int> counts_;
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
    // the reference's own argument types (inc/ORBextractor.h:58-61)
    using InputArray = cv::InputArray;
    using OutputArray = cv::OutputArray;
    static Mat getMat(InputArray a) { return a.getMat(); }
    static uint8_t* createOut(OutputArray a, int r, int c) { a.create(r, c, CV_8U); return a.getMat().data; }
    static void releaseOut(OutputArray a) { a.release(); }
    static Mat wrapBordered(const uint8_t* s, int r, int c, ptrdiff_t step, int b) {
        Mat whole = Mat(r + 2 * b, c + 2 * b, CV_8UC1, (void*)(s - (ptrdiff_t)b * step - b), (size_t)step).clone();      // Mat temp(wholeSize, image.type()), :1174
        return whole(cv::Rect(b, b, c, r));                                                                              // temp(Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, sz.width, sz.height)), :1175
    }
};
}  // namespace orbx
namespace ORB_SLAM3 {
using ORBextractor = orbx::BasicORBextractor<orbx::CvTraits>;   // drop-in name (inc/ORBextractor.h:44)
}
#endif
