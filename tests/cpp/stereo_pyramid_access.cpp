// stereo_pyramid_access.cpp — what Frame::ComputeStereoMatches does with the extractors' public member mvImagePyramid right after the two
// ExtractORB calls (reference src/Frame.cc:820, 905-932), against include/orbx_extractor.hpp through the OpenCV-shaped stand-in header:
// no call is added between operator() and the indexing — the member itself brings the levels to the host.
//     const int nRows = mpORBextractorLeft->mvImagePyramid[0].rows;                                                       Frame.cc:820
//     cv::Mat IL = mpORBextractorLeft->mvImagePyramid[kpL.octave].rowRange(scaledvL-w,scaledvL+w+1).colRange(scaleduL-w,scaleduL+w+1);   :910
//     if(iniu<0 || endu >= mpORBextractorRight->mvImagePyramid[kpL.octave].cols)                                          :924
//     cv::Mat IR = mpORBextractorRight->mvImagePyramid[kpL.octave].rowRange(...).colRange(...);                           :929
// usage: stereo_pyramid_access <left.gray> <right.gray> <rows> <cols> <nfeatures> <out.bin>
// out.bin: int32 n (left keypoints with a full window), then per keypoint: int32 octave, int32 scaledvL, int32 scaleduL, 11x11 bytes of IL,
//          11x11 bytes of IR (same window in the right pyramid), and the byte 19 px left of level `octave`'s pixel (0, scaledvL) — the
//          REFLECT_101 frame the reference's views sit in (ORBextractor.cc:1173-1177), reached through the view's data pointer
#include <opencv2/core/core.hpp>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "orbx_extractor.hpp"

using namespace std;

static bool readGray(const char* path, cv::Mat& m) {
    FILE* f = fopen(path, "rb");
    const bool ok = f && fread(m.data, 1, (size_t)m.rows * m.cols, f) == (size_t)m.rows * m.cols;
    if (f) fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s left right rows cols nfeatures out\n", argv[0]); return 2; }
    const int rows = atoi(argv[3]), cols = atoi(argv[4]);
    cv::Mat imLeft(rows, cols, CV_8UC1), imRight(rows, cols, CV_8UC1);
    if (!readGray(argv[1], imLeft) || !readGray(argv[2], imRight)) { perror("input"); return 2; }
    try {
        ORB_SLAM3::ORBextractor* mpORBextractorLeft = new ORB_SLAM3::ORBextractor(atoi(argv[5]), 1.2f, 8, 20, 7);
        ORB_SLAM3::ORBextractor* mpORBextractorRight = new ORB_SLAM3::ORBextractor(atoi(argv[5]), 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
        cv::Mat mDescriptors, mDescriptorsRight;
        vector<int> vLapping = {0, 0};                    // Frame.cc:109-110
        vector<vector<cv::KeyPoint>> a, b;
        (*mpORBextractorLeft)(imLeft, cv::Mat(), mvKeys, mDescriptors, vLapping, a);
        (*mpORBextractorRight)(imRight, cv::Mat(), mvKeysRight, mDescriptorsRight, vLapping, b);
        const vector<float> mvInvScaleFactors = mpORBextractorLeft->GetInverseScaleFactors();
        // ---- Frame.cc:820 ----
        const int nRows = mpORBextractorLeft->mvImagePyramid[0].rows;
        if (nRows != rows) return 3;
        FILE* o = fopen(argv[6], "wb");
        int n = 0;
        fwrite(&n, 4, 1, o);
        const int w = 5;                                  // Frame.cc:909
        for (size_t iL = 0; iL < mvKeys.size(); iL++) {
            const cv::KeyPoint& kpL = mvKeys[iL];
            // ---- Frame.cc:903-910 ----
            const float scaleFactor = mvInvScaleFactors[kpL.octave];
            const float scaleduL = round(kpL.pt.x * scaleFactor);
            const float scaledvL = round(kpL.pt.y * scaleFactor);
            const cv::Mat& lvlL = mpORBextractorLeft->mvImagePyramid[kpL.octave];
            if (scaledvL - w < 0 || scaledvL + w + 1 > lvlL.rows || scaleduL - w < 0 || scaleduL + w + 1 > lvlL.cols) continue;
            cv::Mat IL = mpORBextractorLeft->mvImagePyramid[kpL.octave].rowRange(scaledvL-w,scaledvL+w+1).colRange(scaleduL-w,scaleduL+w+1);
            // ---- Frame.cc:924, 929 (zero disparity: the same window of the right pyramid) ----
            if (scaleduL + w + 1 >= mpORBextractorRight->mvImagePyramid[kpL.octave].cols) continue;
            cv::Mat IR = mpORBextractorRight->mvImagePyramid[kpL.octave].rowRange(scaledvL-w,scaledvL+w+1).colRange(scaleduL-w,scaleduL+w+1);
            const int oct = kpL.octave, v = (int)scaledvL, u = (int)scaleduL;
            fwrite(&oct, 4, 1, o); fwrite(&v, 4, 1, o); fwrite(&u, 4, 1, o);
            for (int y = 0; y < 2 * w + 1; y++) fwrite(IL.data + (size_t)y * IL.step, 1, 2 * w + 1, o);
            for (int y = 0; y < 2 * w + 1; y++) fwrite(IR.data + (size_t)y * IR.step, 1, 2 * w + 1, o);
            const unsigned char frame = *(lvlL.data + (size_t)v * lvlL.step - 19);      // inside the bordered buffer the view lies in
            fwrite(&frame, 1, 1, o);
            n++;
        }
        fseek(o, 0, SEEK_SET);
        fwrite(&n, 4, 1, o);
        fclose(o);
        delete mpORBextractorLeft; delete mpORBextractorRight;
    } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
