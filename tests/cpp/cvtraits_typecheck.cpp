// cvtraits_typecheck.cpp — the drop-in name ORB_SLAM3::ORBextractor (include/orbx_extractor.hpp, CvTraits branch) compiled against a stand-in
// <opencv2/core/core.hpp> (tests/cpp/opencv_standin: OpenCV 3's public names for the members the shim touches) and driven exactly as
// Frame::ExtractORB does (reference src/Frame.cc:419-427, constructor call Tracking.cc:768-774, getters Frame.cc:285-291).
// usage: cvtraits_typecheck <in.gray> <rows> <cols> <nfeatures> <out.bin>      (out: int32 mono, int32 n, keypoints, descriptors)
#include <opencv2/core/core.hpp>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "orbx_extractor.hpp"

using namespace std;

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s in rows cols nfeatures out\n", argv[0]); return 2; }
    const int rows = atoi(argv[2]), cols = atoi(argv[3]);
    cv::Mat im(rows, cols, CV_8UC1);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(im.data, 1, (size_t)rows * cols, f) != (size_t)rows * cols) { perror("input"); return 2; }
    fclose(f);
    try {
        // Tracking.cc:768: mpORBextractorLeft = new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST);
        ORB_SLAM3::ORBextractor* mpORBextractorLeft = new ORB_SLAM3::ORBextractor(atoi(argv[4]), 1.2f, 8, 20, 7);
        // Frame.cc:285-291
        int mnScaleLevels = mpORBextractorLeft->GetLevels();
        float mfScaleFactor = mpORBextractorLeft->GetScaleFactor();
        vector<float> mvScaleFactors = mpORBextractorLeft->GetScaleFactors(), mvInvLevelSigma2 = mpORBextractorLeft->GetInverseScaleSigmaSquares();
        if (mnScaleLevels != 8 || mfScaleFactor != 1.2f || mvScaleFactors.size() != 8 || mvInvLevelSigma2.size() != 8) return 3;
        // Frame.cc:419-427
        std::vector<cv::KeyPoint> mvKeys;
        cv::Mat mDescriptors;
        vector<int> vLapping = {0, 1000};
        vector<vector<cv::KeyPoint>> allLevelsKeypoints;
        int monoLeft = (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping, allLevelsKeypoints);
        if (mDescriptors.rows != (int)mvKeys.size() || (mvKeys.size() && (mDescriptors.cols != 32 || mDescriptors.type() != CV_8U))) return 4;
        // a colour image is refused as the reference's assert(image.type() == CV_8UC1) refuses it (ORBextractor.cc:1087)
        bool refused = false;
        try { cv::Mat bgr(rows, cols, CV_8UC3); std::vector<cv::KeyPoint> k; cv::Mat d; vector<vector<cv::KeyPoint>> a; (*mpORBextractorLeft)(bgr, cv::Mat(), k, d, vLapping, a); }
        catch (const std::invalid_argument&) { refused = true; }
        if (!refused) return 5;
        // the reference's exact signature (inc/ORBextractor.h:58-61: cv::InputArray / cv::OutputArray): the mask as cv::noArray(), the
        // descriptors into an OutputArray that wraps a Mat - the same numbers as through the Mat overload
        {
            std::vector<cv::KeyPoint> k2;
            cv::Mat d2;
            vector<vector<cv::KeyPoint>> a2;
            cv::_InputArray in(im);
            cv::_OutputArray out(d2);
            int mono2 = (*mpORBextractorLeft)(in, cv::noArray(), k2, out, vLapping, a2);
            if (mono2 != monoLeft || k2.size() != mvKeys.size() || d2.rows != mDescriptors.rows || d2.cols != mDescriptors.cols) return 7;
            if (k2.size() && memcmp(k2.data(), mvKeys.data(), k2.size() * sizeof(cv::KeyPoint))) return 7;
            for (int r = 0; r < d2.rows; r++) if (memcmp(d2.data + (size_t)r * d2.step, mDescriptors.data + (size_t)r * mDescriptors.step, 32)) return 7;
            if (a2.size() != allLevelsKeypoints.size()) return 7;
            // ... mixed: a Mat image with cv::noArray() as the mask picks the proxy overload too
            std::vector<cv::KeyPoint> k3; cv::Mat d3; vector<vector<cv::KeyPoint>> a3;
            if ((*mpORBextractorLeft)(im, cv::noArray(), k3, d3, vLapping, a3) != monoLeft || k3.size() != mvKeys.size()) return 8;
            // the tutorial clone's five-argument form (inc/ORBExtractor.h:55-56), Mat and proxy arguments
            std::vector<cv::KeyPoint> k4, k5; cv::Mat d4, d5;
            if ((*mpORBextractorLeft)(im, cv::Mat(), k4, d4, vLapping) != monoLeft || k4.size() != mvKeys.size() || d4.rows != mDescriptors.rows) return 9;
            if ((*mpORBextractorLeft)(in, cv::noArray(), k5, cv::_OutputArray(d5), vLapping) != monoLeft || k5.size() != mvKeys.size() || d5.rows != mDescriptors.rows) return 9;
            // an empty InputArray returns -1 as an empty Mat does (ORBextractor.cc:1083-1084)
            std::vector<cv::KeyPoint> k6; cv::Mat d6; vector<vector<cv::KeyPoint>> a6;
            if ((*mpORBextractorLeft)(cv::noArray(), cv::noArray(), k6, cv::_OutputArray(d6), vLapping, a6) != -1) return 10;
        }
        // Frame::ComputeStereoMatches reads mvImagePyramid right after the call (Frame.cc:820): no call in between
        const int nRows = mpORBextractorLeft->mvImagePyramid[0].rows;
        if (mpORBextractorLeft->mvImagePyramid.size() != 8 || mpORBextractorLeft->mvImagePyramid[0].cols != cols || nRows != rows) return 6;
        FILE* o = fopen(argv[5], "wb");
        int n = (int)mvKeys.size();
        fwrite(&monoLeft, 4, 1, o); fwrite(&n, 4, 1, o);
        fwrite(mvKeys.data(), sizeof(cv::KeyPoint), mvKeys.size(), o);
        for (int r = 0; r < mDescriptors.rows; r++) fwrite(mDescriptors.data + (size_t)r * mDescriptors.step, 1, 32, o);
        fclose(o);
        delete mpORBextractorLeft;
    } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
