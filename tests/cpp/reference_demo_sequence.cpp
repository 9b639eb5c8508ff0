// reference_demo_sequence.cpp — the call sequence of the reference's own demo, /root/reference/src/orb_extractor/main_orb_extractor.cpp:43-53
// (the program whose console output, "ORB_SLAM3 has total 1420 keypoints", is the one result the reference holds: img_folder/Screenshot.png),
// compiled against include/orbx_extractor.hpp through the OpenCV-shaped stand-in header and run through the C ABI:
//     ORBextractor my_orb_extractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST);
//     my_orb_extractor.ComputePyramid(image);
//     vector<vector<KeyPoint>> allKeypoints;
//     my_orb_extractor.ComputeKeyPointsOctTree(allKeypoints);
//     ... image_total_keypoints += (int) allKeypoints[level].size(); ... cout << "ORB_SLAM3 has total " << image_total_keypoints << " keypoints"
// usage: reference_demo_sequence <in.gray> <rows> <cols> <nfeatures> <out.bin>
// out.bin: int32 total, nlevels x int32 per-level counts, total x 28-byte keypoints (level coordinates), then level 0 .. 7 of mvImagePyramid
//          as the demo's imshow loop would read them (int32 w, int32 h, w*h bytes each)
#include <opencv2/core/core.hpp>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "orbx_extractor.hpp"

using namespace std;
using namespace cv;
using ORB_SLAM3::ORBextractor;

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s in rows cols nfeatures out\n", argv[0]); return 2; }
    const int rows = atoi(argv[2]), cols = atoi(argv[3]);
    cv::Mat image(rows, cols, CV_8UC1);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(image.data, 1, (size_t)rows * cols, f) != (size_t)rows * cols) { perror("input"); return 2; }
    fclose(f);
    try {
        // main_orb_extractor.cpp:34-38
        int nFeatures = atoi(argv[4]);
        float fScaleFactor = 1.2;
        int nLevels = 8;
        int fIniThFAST = 20;
        int fMinThFAST = 7;
        // :43-46
        ORBextractor my_orb_extractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST);
        my_orb_extractor.ComputePyramid(image);
        vector<vector<KeyPoint>> allKeypoints;
        my_orb_extractor.ComputeKeyPointsOctTree(allKeypoints);
        // :48-53
        int image_total_keypoints = 0;
        for (int level = 0; level < nLevels; ++level) {
            image_total_keypoints += (int) allKeypoints[level].size();
        }
        cout << "ORB_SLAM3 has total " << image_total_keypoints << " keypoints" << endl;
        // :62-66: the scale restoration reads the public member mvScaleFactor
        float scale = my_orb_extractor.mvScaleFactor[1];
        if (!(scale > 1.19f && scale < 1.21f)) return 3;
        FILE* o = fopen(argv[5], "wb");
        fwrite(&image_total_keypoints, 4, 1, o);
        for (int level = 0; level < nLevels; ++level) { int c = (int)allKeypoints[level].size(); fwrite(&c, 4, 1, o); }
        for (int level = 0; level < nLevels; ++level) fwrite(allKeypoints[level].data(), sizeof(KeyPoint), allKeypoints[level].size(), o);
        // main_whole_orb_extractor.cpp shows the pyramid level by level out of the public member after ComputePyramid
        for (int level = 0; level < nLevels; ++level) {
            const cv::Mat& m = my_orb_extractor.mvImagePyramid[level];
            fwrite(&m.cols, 4, 1, o); fwrite(&m.rows, 4, 1, o);
            for (int y = 0; y < m.rows; y++) fwrite(m.data + (size_t)y * m.step, 1, m.cols, o);
        }
        fclose(o);
    } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
