// STAND-IN for <opencv2/core/core.hpp> — TEST INFRASTRUCTURE, not OpenCV.
//
// This image has no OpenCV, so the `orbx::CvTraits` branch of include/orbx_extractor.hpp (the one a maintainer of the reference compiles)
// had never been seen by a compiler.  This header declares, with OpenCV 3's public names and signatures, exactly the parts of cv::Mat /
// cv::KeyPoint / the CV_* macros that branch and the reference's call site (src/Frame.cc:419-427) use, so that
// tests/cpp/cvtraits_typecheck.cpp type-checks (and, on the GPU box, runs) the drop-in name ORB_SLAM3::ORBextractor.  It implements
// just enough behaviour (a reference-counted byte buffer) for that program; it is not used to build or imitate the reference.
#pragma once
#define OPENCV_CORE_HPP          // the include guard of OpenCV >= 3.2, which orbx_extractor.hpp looks for
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)

namespace cv {
typedef unsigned char uchar;
struct Point2f { float x = 0, y = 0; };
class KeyPoint {          // modules/core/include/opencv2/core/types.hpp: pt, size, angle, response, octave, class_id — 28 bytes
public:
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
struct MatStep {           // Mat::step converts to size_t
    size_t v = 0;
    operator size_t() const { return v; }
};
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {} };
class Mat {
public:
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t stepBytes) : rows(r), cols(c), data((uchar*)ext), flags_(type) { step.v = stepBytes; }   // a view: no ownership
    void create(int r, int c, int type) {
        const int cn = (type >> CV_CN_SHIFT) + 1;
        own_ = std::make_shared<std::vector<uchar>>((size_t)r * c * cn);
        rows = r; cols = c; flags_ = type; step.v = (size_t)c * cn; data = own_->data();
    }
    void release() { own_.reset(); rows = cols = 0; data = nullptr; step.v = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return flags_; }
    Mat clone() const {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, flags_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step.v, data + (size_t)y * step.v, m.step.v);
        return m;
    }
    // views that share the buffer (and keep it alive), as Mat::operator()(Rect) / rowRange / colRange do
    Mat operator()(const Rect& r) const { Mat m(*this); m.data = data + (size_t)r.y * step.v + r.x; m.rows = r.height; m.cols = r.width; return m; }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
    template <class T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step.v + x * sizeof(T)); }
    template <class T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step.v + x * sizeof(T)); }
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    MatStep step;
private:
    int flags_ = 0;
    std::shared_ptr<std::vector<uchar>> own_;
};
// The argument proxies of OpenCV 3 (modules/core/include/opencv2/core/mat.hpp): `typedef const _InputArray& InputArray;`,
// `typedef const _OutputArray& OutputArray;`, cv::noArray().  Only what ORBextractor::operator() does with them (inc/ORBextractor.h:58-61,
// ORBextractor.cc:1086, 1103, 1106-1107): getMat(), create(rows, cols, type), release(); built from a Mat, a std::vector<uchar> or nothing.
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : mat_(const_cast<Mat*>(&m)) {}
    _InputArray(const std::vector<uchar>& v) : vec_(const_cast<std::vector<uchar>*>(&v)) {}
    Mat getMat(int = -1) const {
        if (mat_) return *mat_;
        if (vec_ && !vec_->empty()) return Mat(1, (int)vec_->size(), CV_8UC1, vec_->data(), vec_->size());      // a 1 x N view, as OpenCV gives
        return Mat();
    }
    bool empty() const { return getMat().empty(); }
protected:
    Mat* mat_ = nullptr;
    std::vector<uchar>* vec_ = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) { mat_ = &m; }
    _OutputArray(std::vector<uchar>& v) { vec_ = &v; }
    void create(int r, int c, int type) const {
        if (mat_) { if (mat_->rows != r || mat_->cols != c || mat_->type() != type || !mat_->data) mat_->create(r, c, type); }
        else if (vec_) vec_->resize((size_t)r * c);
    }
    void release() const { if (mat_) mat_->release(); else if (vec_) vec_->clear(); }
};
class _InputOutputArray : public _OutputArray {
public:
    _InputOutputArray() {}
    _InputOutputArray(Mat& m) : _OutputArray(m) {}
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _InputOutputArray& InputOutputArray;
inline InputOutputArray noArray() { static _InputOutputArray none; return none; }
}  // namespace cv
