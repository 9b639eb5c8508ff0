/* orbx.h — C ABI of the MI355X-native ORB extractor (liborbx.so).
 *
 * Drop-in boundary for the reference's ORB_SLAM3::ORBextractor (reference
 * inc/ORBextractor.h:44-111): every entry point below names the reference interface it replaces.
 * The C++ shim include/orbx_extractor.hpp rebuilds the reference class on top of these calls;
 * INTEGRATION.md shows the binding a maintainer adds at Frame::ExtractORB (reference
 * src/Frame.cc:419-427).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; no exceptions cross this boundary
 *   - return value: ORBX_OK (0) or a negative orbx_status; orbx_last_error() gives the text
 *   - one handle == one camera == one HIP stream + its device buffers (the reference keeps one
 *     ORBextractor per camera: src/Tracking.cc:768-774).  Calls on one handle are serialised by
 *     the caller, calls on different handles may run concurrently (as Frame.cc:109-112 does)
 *   - "device" pointers are HIP device pointers valid on the handle's GPU
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_ABI_VERSION 1
#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE_THRESHOLD 19 /* reference ORBextractor.cc:72 */

typedef enum orbx_status {
    ORBX_OK = 0,
    ORBX_ERR_EMPTY_IMAGE = -1,    /* reference operator() returns -1 (ORBextractor.cc:1083-1084) */
    ORBX_ERR_BAD_ARGUMENT = -2,
    ORBX_ERR_CAPACITY = -3,       /* caller's output capacity too small */
    ORBX_ERR_IMAGE_TOO_LARGE = -4,/* exceeds max_width/max_height/max_batch given at create */
    ORBX_ERR_IMAGE_TOO_SMALL = -5,/* coarsest level narrower than one 30-px FAST cell (reference divides by 0) */
    ORBX_ERR_HIP = -6,            /* HIP runtime error, see orbx_last_error */
    ORBX_ERR_NO_DEVICE = -7,
    ORBX_ERR_UNSUPPORTED = -8
} orbx_status;

/* Layout-identical to cv::KeyPoint (28 bytes; reference consumers read pt/octave/angle/size/response:
 * src/Frame.cc:383-417, 748-782).  The shim memcpy's arrays of these into std::vector<cv::KeyPoint>. */
typedef struct orbx_keypoint {
    float x, y;     /* pt: level-0 pixel coordinates in the final arrays; level coordinates in per-level arrays */
    float size;     /* (int)(31 * scale[level])                         ORBextractor.cc:872,881 */
    float angle;    /* IC_Angle, degrees in [0,360)                      ORBextractor.cc:75-102 */
    float response; /* FAST-9 corner score                                ORBextractor.cc:818-819 */
    int32_t octave; /* pyramid level                                      ORBextractor.cc:880 */
    int32_t class_id; /* always -1 */
} orbx_keypoint;

typedef struct orbx_handle orbx_handle; /* opaque */

/* Replaces ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (inc/ORBextractor.h:50-51, ORBextractor.cc:408-475).  All device memory for images up to
 * max_width x max_height and batches up to max_batch frames is allocated here, once.
 * device < 0 selects the current HIP device. */
int orbx_create(orbx_handle** out, int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                int min_th_fast, int max_width, int max_height, int max_batch, int device);
void orbx_destroy(orbx_handle* h);
const char* orbx_last_error(const orbx_handle* h); /* h may be NULL: error of the last failed orbx_create */
int orbx_abi_version(void);

/* Replaces the getters inc/ORBextractor.h:63-83 (GetLevels, GetScaleFactor, GetScaleFactors,
 * GetInverseScaleFactors, GetScaleSigmaSquares, GetInverseScaleSigmaSquares) plus the public members
 * mnFeaturesPerLevel / umax (:103-105).  Arrays hold nlevels (umax: 16) entries.  Host-only. */
int orbx_get_levels(const orbx_handle* h);
float orbx_get_scale_factor(const orbx_handle* h);
int orbx_get_tables(const orbx_handle* h, float* scale_factors, float* inv_scale_factors,
                    float* level_sigma2, float* inv_level_sigma2, int* features_per_level, int* umax16);
/* Max keypoints one frame can produce: sum over levels of max(quota+3, 4*nIni) for the widest
 * supported aspect; callers size kps/desc with it (nfeatures + 3*nlevels for ordinary aspects). */
int orbx_max_keypoints(const orbx_handle* h);

/* Handle-free host helpers (no GPU touched; usable on a machine without one).
 * orbx_compute_tables: the constructor's tables for (nfeatures, scaleFactor, nlevels), ORBextractor.cc:419-474.
 * orbx_compute_level_sizes: pyramid level sizes of a cols x rows image, ORBextractor.cc:1171.
 * orbx_compute_cell_grid: FAST cell grid of one level (nCols, nRows, wCell, hCell: ORBextractor.cc:789-795),
 *   the number of cv::FAST calls the reference makes on it, the quad-tree root count nIni (:548) and the
 *   exact upper bound of FAST candidates the level can produce. */
int orbx_compute_tables(int nfeatures, float scale_factor, int nlevels, float* scale_factors,
                        float* inv_scale_factors, float* level_sigma2, float* inv_level_sigma2,
                        int* features_per_level, int* umax16);
int orbx_compute_level_sizes(float scale_factor, int nlevels, int rows, int cols, int* widths, int* heights);
int orbx_compute_cell_grid(float scale_factor, int nlevels, int rows, int cols, int level, int* n_cols,
                           int* n_rows, int* w_cell, int* h_cell, int* n_cells, int* n_ini, int* cand_cap);

/* Replaces int ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea,
 * allLevelsKeypoints) (inc/ORBextractor.h:58-61, ORBextractor.cc:1078-1162) for one CV_8UC1 host image.
 *   img/rows/cols/stride : the cv::Mat (data, rows, cols, step); mask is ignored by the reference
 *   lap0, lap1           : vLappingArea[0], [1]
 *   kps, desc, capacity  : caller-owned outputs, desc is capacity x 32 bytes, row i <-> kps[i]
 *   *n_out               : number of keypoints == keypoints.size()
 *   *mono_out            : the reference's return value (monoIndex)
 *   level_kps, level_counts (optional, may be NULL): allLevelsKeypoints flattened level by level in
 *                          level coordinates (capacity entries), and the nlevels per-level counts
 * Returns ORBX_ERR_EMPTY_IMAGE (-1) for rows==0||cols==0||img==NULL exactly as the reference returns -1.
 * Synchronous: results are in host memory on return. */
int orbx_extract(orbx_handle* h, const uint8_t* img, int rows, int cols, ptrdiff_t stride, int lap0,
                 int lap1, orbx_keypoint* kps, uint8_t* desc, int capacity, int* n_out, int* mono_out,
                 orbx_keypoint* level_kps, int* level_counts);

/* The same call without the last copy: the results stay in the handle's pinned result slab and the caller gets pointers into it, valid
 * until the next result-producing call on this handle is MADE - its kernels write the same memory (the C++ shim copies them straight into the caller's std::vector / cv::Mat: one copy instead of
 * two).  want_levels != 0 also produces allLevelsKeypoints (level_kps flattened level by level, level_counts[nlevels]).  One frame per
 * call has NO copy command on the stream: the image goes through pinned staging in one H2D copy and the kernels write the slab in host
 * memory themselves. */
int orbx_extract_view(orbx_handle* h, const uint8_t* img, int rows, int cols, ptrdiff_t stride, int lap0, int lap1, int want_levels,
                      const orbx_keypoint** kps, const uint8_t** desc, int* n_out, int* mono_out, const orbx_keypoint** level_kps,
                      const int** level_counts);

/* Replaces the two public stage methods of the reference class (inc/ORBextractor.h:87-90: `protected:` is commented out so that callers
 * can use them; src/orb_extractor/main_orb_extractor.cpp:43-46 and main_whole_orb_extractor.cpp:44-46 do):
 *   void ORBextractor::ComputePyramid(cv::Mat image)                                        ORBextractor.cc:1164-1219
 *   void ORBextractor::ComputeKeyPointsOctTree(vector<vector<KeyPoint>>& allKeypoints)      ORBextractor.cc:773-888
 * orbx_compute_pyramid uploads one CV_8UC1 host image and builds the pyramid (mvImagePyramid: orbx_fetch_pyramid / orbx_get_level);
 * orbx_compute_keypoints_octree runs cell-grid FAST, DistributeOctTree, the (16,16) shift / octave / size fix-up and the orientation on
 * the pyramid the handle holds (from orbx_compute_pyramid or from any extract call: frame 0 of it) and returns allKeypoints flattened
 * level by level in LEVEL coordinates with angles set, exactly what operator() hands out as allLevelsKeypoints (:1094).  Both synchronous. */
int orbx_compute_pyramid(orbx_handle* h, const uint8_t* img, int rows, int cols, ptrdiff_t stride);
int orbx_compute_keypoints_octree(orbx_handle* h, orbx_keypoint* level_kps, int capacity, int* level_counts);

/* Batched form for a video stream or a stereo pair (the two std::threads of Frame.cc:109-112 become one
 * launch sequence).  All frames share rows/cols/stride; frame f starts at imgs + f*frame_stride.
 * Outputs are frame-major with a fixed per-frame capacity: kps[f*capacity + i], desc[(f*capacity+i)*32],
 * n_out[f], mono_out[f], level_counts[f*nlevels + l].  lap0/lap1 may be NULL (=> {0,1000}, the mono
 * default of Frame.cc:307) or hold one pair per frame: lap[2*f], lap[2*f+1]. */
int orbx_extract_batch(orbx_handle* h, int n_frames, const uint8_t* imgs, int rows, int cols,
                       ptrdiff_t stride, ptrdiff_t frame_stride, const int* lap, orbx_keypoint* kps,
                       uint8_t* desc, int capacity, int* n_out, int* mono_out, orbx_keypoint* level_kps,
                       int* level_counts);

/* Asynchronous host-buffer form.  _begin enqueues the H2D copy of the frames, the whole path and the D2H copy of
 * the results (into pinned staging owned by the handle) and returns without waiting; _end waits for that batch and
 * copies the results into the caller's arrays (same layout as orbx_extract_batch).  One batch in flight per handle:
 * two handles used alternately overlap the transfers of one batch with the kernels of the other (input copies of 16 MiB and more go through
 * ONE copy queue per device, shared by its handles, so that they run one after the other at the link's full rate: DESIGN.md).  The H2D copy is
 * only truly asynchronous from pinned memory (orbx_host_alloc).  want_levels != 0 also brings back the per-level
 * arrays.  orbx_extract_batch == _begin + _end. */
int orbx_extract_batch_begin(orbx_handle* h, int n_frames, const uint8_t* imgs, int rows, int cols, ptrdiff_t stride,
                             ptrdiff_t frame_stride, const int* lap, int want_levels);
int orbx_extract_batch_end(orbx_handle* h, orbx_keypoint* kps, uint8_t* desc, int capacity, int* n_out, int* mono_out,
                           orbx_keypoint* level_kps, int* level_counts);
/* Zero-copy variant of _end: waits, then hands out pointers into the handle's pinned result staging
 * (kps[f*capacity + i], desc[(f*capacity + i)*32], n_out[f], mono_out[f]).
 * Lifetime (same rule as orbx_extract_view): the pointers are valid until the NEXT CALL of any kind that produces results on this handle -
 * orbx_extract*, _begin, orbx_compute_keypoints_octree - is MADE, not until it completes: with one frame per call the kernels of that next
 * call write this very memory (there is no copy command whose completion would mark the overwrite).  Copy what you keep before calling again,
 * or alternate two handles. */
int orbx_extract_batch_end_view(orbx_handle* h, const orbx_keypoint** kps, const uint8_t** desc, int* capacity,
                                const int** n_out, const int** mono_out);
void* orbx_host_alloc(size_t bytes); /* pinned host memory (hipHostMalloc); NULL on failure */
void orbx_host_free(void* p);

/* Device-resident batched form: d_imgs and the outputs are device pointers; the work is enqueued on the
 * handle's stream and NOT synchronised (call orbx_synchronize or sync the stream yourself).  Same output
 * layout as orbx_extract_batch; d_level_kps/d_level_counts may be NULL.  lap is a HOST pointer (or NULL). */
int orbx_extract_batch_device(orbx_handle* h, int n_frames, const uint8_t* d_imgs, int rows, int cols,
                              ptrdiff_t stride, ptrdiff_t frame_stride, const int* lap,
                              orbx_keypoint* d_kps, uint8_t* d_desc, int capacity, int* d_n_out,
                              int* d_mono_out, orbx_keypoint* d_level_kps, int* d_level_counts);

/* Replaces reads of the public member mvImagePyramid (inc/ORBextractor.h:85; read by
 * Frame::ComputeStereoMatches, src/Frame.cc:820,910,929) after a call: copies level `level` of frame
 * `frame` of the last batch to host.  bordered==0: width x height pixels; bordered!=0: the
 * (width+38) x (height+38) buffer with the BORDER_REFLECT_101 frame of ORBextractor.cc:1193-1215. */
int orbx_get_level(orbx_handle* h, int frame, int level, int bordered, uint8_t* dst, ptrdiff_t dst_stride,
                   int* width, int* height);

/* All levels of frame `frame` of the last batch (or of orbx_compute_pyramid) in ONE device-to-host copy, for readers of mvImagePyramid
 * (Frame.cc:820,910,924,929): *base points into pinned staging owned by the handle (valid until the next orbx_fetch_pyramid on it);
 * level l is heights[l] rows of widths[l] pixels, pixel (0,0) at base[level_offset[l]], rows level_stride[l] bytes apart, and — as in the
 * reference, where mvImagePyramid[l] is a view into a bordered buffer (ORBextractor.cc:1173-1177) — the 19-px BORDER_REFLECT_101 frame
 * lies around it in the same buffer.  Arrays hold nlevels entries. */
int orbx_fetch_pyramid(orbx_handle* h, int frame, const uint8_t** base, size_t* level_offset, int* level_stride, int* widths, int* heights);

/* ---- next row beyond the extractor ------------------------------------------------------
 * Replaces Frame::ComputeStereoMatches() (reference src/Frame.cc:813-991) for stereo pairs extracted by ONE
 * batched call on this handle: frame 2p is the left eye, frame 2p+1 the right eye (n_frames = 2*n_pairs).
 * It consumes that call's keypoints/descriptors and the two eyes' image pyramids, which are still in HBM (the
 * reference reads mpORBextractorLeft/Right->mvImagePyramid for the 11x11 SAD refinement, Frame.cc:910,929).
 *   bf, b : Frame::mbf and Frame::mb (baseline*fx and baseline; minZ = b, maxD = bf/b, Frame.cc:843-845)
 *   u_right[p*capacity + i], depth[p*capacity + i] : mvuRight / mvDepth of left keypoint i (-1 = no match)
 *   n_matched[p] : matches that survive the median filter
 * orbx_stereo_match_device takes the device buffers an orbx_extract_batch_device call filled (and is asynchronous
 * on the handle's stream); orbx_stereo_match_last uses the results of the last orbx_extract_batch call, which the
 * handle keeps in HBM, and returns host arrays. */
int orbx_stereo_match_device(orbx_handle* h, int n_pairs, const orbx_keypoint* d_kps, const uint8_t* d_desc,
                             const int* d_n_out, int capacity, float bf, float b, float* d_u_right, float* d_depth,
                             int* d_n_matched);
int orbx_stereo_match_last(orbx_handle* h, int n_pairs, float bf, float b, float* u_right, float* depth, int capacity,
                           int* n_matched);

/* ---- the caller's side of the path: colour input ----------------------------------------------------------------
 * Replaces the cv::cvtColor(RGB2GRAY | BGR2GRAY | RGBA2GRAY | BGRA2GRAY) calls of Tracking::GrabImageMonocular / Stereo /
 * RGBD (src/Tracking.cc:915-941, 985-1001) for n_frames device-resident 8-bit frames of `channels` (3 or 4) interleaved
 * channels; red_first = Tracking::mbRGB.  gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14 (OpenCV's 14-bit fixed point).
 * The result feeds orbx_extract_batch_device.  Asynchronous on the handle's stream. */
int orbx_gray_from_color_device(orbx_handle* h, int n_frames, const uint8_t* d_src, int rows, int cols, int channels, int red_first,
                                ptrdiff_t src_stride, ptrdiff_t src_frame_stride, uint8_t* d_gray, ptrdiff_t gray_stride,
                                ptrdiff_t gray_frame_stride);

/* ---- RGB-D frames: Frame::ComputeStereoFromRGBD (src/Frame.cc:994-1015) with the depth conversion of
 * Tracking::GrabImageRGBD (imDepth.convertTo(CV_32F, mDepthMapFactor), src/Tracking.cc:1003-1004) fused in.
 * d_depth: n_frames depth maps, float32 or uint16 (depth_is_u16); a uint16 map is always scaled by depth_map_factor
 * (= 1 / DepthMapFactor of the settings file, Tracking.cc:680-684), a float map only when |factor - 1| > 1e-5.
 * For keypoint i of frame f (d_kps: mvKeys, d_kps_un: mvKeysUn): d = depth(int(y), int(x)); d > 0 gives
 * d_depth_out = d and d_u_right = kpUn.x - mbf / d, else both are -1.  Slots past n_out[f] are filled with -1. */
int orbx_stereo_from_rgbd_device(orbx_handle* h, int n_frames, const orbx_keypoint* d_kps, const orbx_keypoint* d_kps_un,
                                 const int* d_n_out, int capacity, const void* d_depth, int depth_is_u16, int rows, int cols,
                                 ptrdiff_t depth_stride_bytes, ptrdiff_t depth_frame_stride_bytes, float depth_map_factor, float mbf,
                                 float* d_u_right, float* d_depth_out);

/* ---- next row: the rest of the Frame constructor ------------------------------------------
 * Frame::mK and Frame::mDistCoef as plain floats (fx, fy, cx, cy: src/Frame.cc:342-345; k1, k2, p1, p2[, k3]). */
typedef struct orbx_camera { float fx, fy, cx, cy, k1, k2, p1, p2, k3; } orbx_camera;
#define ORBX_GRID_COLS 64 /* FRAME_GRID_COLS, inc/Frame.h:40 */
#define ORBX_GRID_ROWS 48 /* FRAME_GRID_ROWS, inc/Frame.h:39 */

/* Replaces Frame::ComputeImageBounds (src/Frame.cc:784-811): bounds[4] = mnMinX, mnMaxX, mnMinY, mnMaxY of a
 * cols x rows image (the four corners undistorted when k1 != 0).  Host-only, no GPU touched. */
int orbx_compute_image_bounds(const orbx_camera* cam, int cols, int rows, float* bounds4);

/* Replaces Frame::UndistortKeyPoints (src/Frame.cc:748-782) and Frame::AssignFeaturesToGrid (:383-417, the
 * Nleft == -1 case; the other one: orbx_frame_finish_two_eyes_device below) + PosInGrid (:726-736) for n_frames frames of device-resident extraction results:
 *   d_kps_un[f*capacity + i]            : mvKeysUn
 *   d_grid_off[f*(64*48+1) + x*48 + y]  : first slot of mGrid[x][y]; [.. + 64*48] = keypoints inside the grid
 *   d_grid_idx[f*capacity + slot]       : keypoint indices, increasing inside a cell (push_back order)
 *   d_n_inside[f]
 * bounds4 as returned by orbx_compute_image_bounds.  Asynchronous on the handle's stream. */
int orbx_frame_finish_device(orbx_handle* h, int n_frames, const orbx_keypoint* d_kps, const int* d_n_out, int capacity,
                             const orbx_camera* cam, const float* bounds4, orbx_keypoint* d_kps_un, int* d_grid_off,
                             int* d_grid_idx, int* d_n_inside);
/* The Nleft != -1 branch of Frame::AssignFeaturesToGrid (src/Frame.cc:404-414; the two-camera Frame constructor, :1045-1122, which assigns BEFORE
 * it undistorts): for n_pairs pairs of a device-resident batch - frame 2p = the left eye (mvKeys), frame 2p + 1 = the right eye (mvKeysRight) -
 * the cells come from each eye's RAW keypoints: the left frame's CSR is mGrid (indices i < Nleft), the right frame's is mGridRight (indices
 * i - Nleft, i.e. the right eye's own); d_kps_un still receives UndistortKeyPoints' mvKeysUn of both eyes.  Same array layout as above, for
 * 2 * n_pairs frames.  (The camera model, matcher twins and frustum tests of that rig are out of scope: SURVEY.md §2 #13.) */
int orbx_frame_finish_two_eyes_device(orbx_handle* h, int n_pairs, const orbx_keypoint* d_kps, const int* d_n_out, int capacity,
                                      const orbx_camera* cam, const float* bounds4, orbx_keypoint* d_kps_un, int* d_grid_off,
                                      int* d_grid_idx, int* d_n_inside);

/* ---- next row: monocular initialisation matching ------------------------------------------
 * Replaces ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:706-821; caller src/Tracking.cc:2065-2066 with
 * ORBmatcher(0.9, true) and windowSize 100) with Frame::GetFeaturesInArea (src/Frame.cc:655-724),
 * ORBmatcher::DescriptorDistance (:2349-2365) and ComputeThreeMaxima (:2303-2344), for n_pairs frame pairs of one
 * device-resident batch: pair p matches F1 = frame frame1_first + p*frame1_step against F2 = frame
 * frame2_first + p*frame2_step (a stream: 0,1,1,1; one initial frame against many: 0,0,1,1).
 *   d_kps_un, d_grid_off, d_grid_idx : as written by orbx_frame_finish_device (mvKeysUn, mGrid) for all frames
 *   d_desc, d_n_out                  : mDescriptors, N of all frames (orbx_extract_batch_device)
 *   d_prev_matched[(p*capacity + i)*2 + {0,1}] : vbPrevMatched of pair p, in/out (Tracking.cc:2029-2031 seeds it
 *                                      with F1.mvKeysUn[i].pt; matched entries become F2.mvKeysUn[match].pt, :815-817)
 *   d_matches12[p*capacity + i]      : vnMatches12, -1 = unmatched;  d_n_matches[p] : the return value
 * Frames of the initialisation extractor (ORBextractor(5 * nFeatures), src/Tracking.cc:774: capacity 5000 .. 10 000) are supported: only the
 * level-0 keypoints of F2 are candidates (:722-726) and only they are staged on chip (52 B each next to 4 B per keypoint of F1 in 160 KB of
 * LDS).  Should F2 hold more level-0 keypoints than fit (a one-level pyramid with thousands of features), the pair reports
 * d_n_matches[p] = -1 and an all -1 table.  Asynchronous on the handle's stream. */
int orbx_search_for_initialization_device(orbx_handle* h, int n_pairs, int frame1_first, int frame1_step, int frame2_first,
                                          int frame2_step, const orbx_keypoint* d_kps_un, const uint8_t* d_desc,
                                          const int* d_n_out, int capacity, const int* d_grid_off, const int* d_grid_idx,
                                          const float* bounds4, float* d_prev_matched, int window_size, float nn_ratio,
                                          int check_orientation, int* d_matches12, int* d_n_matches);

/* ---- next row (second half): ORBmatcher::SearchByProjection ---------------------------------------
 * For Nleft == -1 frames (monocular, rectified stereo, RGB-D).  MapPoints, poses and frustum tests belong to the tracker and
 * the map (out of scope), so they enter as plain arrays.  One search request per MapPoint: */
typedef struct orbx_proj_query {
    float u, v;          /* projected position in the current frame: uv (ORBmatcher.cc:2003) / MapPoint::mTrackProjX, mTrackProjY (:58) */
    float ur;            /* predicted right-eye column: uv.x - mbf*invzc (:2043) / mTrackProjXR (:70); used only where mvuRight > 0 */
    float radius;        /* window half-size: th*mvScaleFactors[octave] (:2014) / RadiusByViewingCos(cos)[*th]*mvScaleFactors[level] (:52-58) */
    int32_t min_level, max_level;   /* GetFeaturesInArea's level range: (oct-1, oct+1) | (oct, -1) | (0, oct) (:2018-2023) / (level-1, level) (:58) */
    int32_t flags;       /* bit 0: search this request; bit 1: its MapPoint has Observations() > 0 (it then closes the keypoint it takes) */
    float angle;         /* LastFrame.mvKeysUn[i].angle for the rotation histogram (:2067); unused in ratio mode */
} orbx_proj_query;

/* Front half of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1961-2023, 2043): for pair p the
 * keypoints of frame last_first + p*last_step that hold a MapPoint are projected into frame cur_first + p*cur_step.
 *   d_kps / d_kps_un    : mvKeys (octave) / mvKeysUn (angle) of all frames          d_n_out : N of all frames
 *   d_mp_flags[f*capacity + i] : bit 0 = LastFrame.mvpMapPoints[i] != NULL && !mvbOutlier[i], bit 1 = Observations() > 0
 *   d_world[(f*capacity + i)*3]: MapPoint::GetWorldPos()
 *   d_poses[f*12]       : Frame::mTcw, rows 0..2 (3x4, row-major) of every frame
 *   cam (fx, fy, cx, cy): Pinhole::project;  bounds4: mnMinX, mnMaxX, mnMinY, mnMaxY;  mbf, mb: Frame::mbf, mb;  th, mono as passed
 *   d_queries[p*capacity + i]  : out, one request per keypoint of the last frame (flags = 0 where the reference `continue`s)
 * Asynchronous on the handle's stream. */
int orbx_project_last_frame_device(orbx_handle* h, int n_pairs, int last_first, int last_step, int cur_first, int cur_step,
                                   const orbx_keypoint* d_kps, const orbx_keypoint* d_kps_un, const int* d_n_out, int capacity,
                                   const uint8_t* d_mp_flags, const float* d_world, const float* d_poses, const orbx_camera* cam,
                                   const float* bounds4, float mbf, float mb, float th, int mono, orbx_proj_query* d_queries);

/* The search itself, for n_pairs frames cur_first + p*cur_step of one device-resident batch.
 *   ratio_mode 0: SearchByProjection(CurrentFrame, LastFrame, ...) (src/ORBmatcher.cc:2025-2175): best candidate, TH_HIGH = 100, rotation
 *                 histogram when check_orientation;
 *                 With max_distance = ORBdist, d_occupied = "mvpMapPoints[i2] != NULL", every request's flags bit 1 set and d_u_right = NULL
 *                 this is also SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:2179-2300, Relocalization):
 *                 the caller projects pKF's MapPoints (:2200-2226: bounds, distance invariance, PredictScale) into requests.
 *   max_distance: the acceptance bound on the best descriptor distance: ORBmatcher::TH_HIGH = 100 for the first two forms (:98, :2058)
 *   ratio_mode 1: SearchByProjection(F, vpMapPoints, th, ...) (src/ORBmatcher.cc:44-135): best and second best, the ratio nn_ratio applies
 *                 only when both lie on the same pyramid level; the caller passes only MapPoints with mbTrackInView that are not bad and
 *                 pass the far-point test (:50-59), in vpMapPoints order.
 *   d_queries[p*query_capacity + i], d_n_queries[p] (NULL: query_capacity requests per frame, unused ones have flags = 0)
 *   d_query_desc[((desc_first + p*desc_step)*query_capacity + i)*32] : MapPoint::GetDescriptor() of request i (desc_first / desc_step index
 *                 blocks of query_capacity descriptors: for frame-to-frame matching the last frames' blocks of a per-frame array)
 *   d_kps_un, d_desc, d_n_out, d_grid_off, d_grid_idx : mvKeysUn, mDescriptors, N, mGrid of all frames (orbx_frame_finish_device)
 *   d_u_right[f*capacity + i] : mvuRight of all frames, or NULL (monocular)
 *   d_occupied[p*capacity + i] : in/out or NULL (= all free): the keypoint already holds a MapPoint with Observations() > 0
 *   d_matches[p*capacity + i]  : out, request whose MapPoint the keypoint holds afterwards, -1 = none;  d_n_matches[p] : the return value
 * Asynchronous on the handle's stream. */
int orbx_search_by_projection_device(orbx_handle* h, int n_pairs, int cur_first, int cur_step, const orbx_proj_query* d_queries,
                                     const uint8_t* d_query_desc, int desc_first, int desc_step, const int* d_n_queries, int query_capacity,
                                     const orbx_keypoint* d_kps_un, const uint8_t* d_desc, const int* d_n_out, int capacity,
                                     const int* d_grid_off, const int* d_grid_idx, const float* bounds4, const float* d_u_right,
                                     uint8_t* d_occupied, int ratio_mode, float nn_ratio, int max_distance, int check_orientation,
                                     int* d_matches, int* d_n_matches);

/* ---- next row: Frame::ComputeBoW (src/Frame.cc:739-746) --------------------------------------------------
 * = DBoW2::TemplatedVocabulary<FORB>::transform(features, BowVector, FeatureVector, levelsup = 4)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1196, 1218-1262; BowVector.cpp:34-83; FeatureVector.cpp:31-46).
 * The vocabulary is an object of its own (System.cc:81-82 loads one per process and shares it between all frames). */
typedef struct orbx_vocabulary orbx_vocabulary;
/* Replaces ORBVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1423): the ORBvoc.txt format ("k L scoring weighting", then one
 * node per line: parent, is-leaf, 32 descriptor bytes, weight).  device < 0: current HIP device.  Errors: ORBX_ERR_BAD_ARGUMENT (file). */
int orbx_vocabulary_load_text(orbx_vocabulary** out, const char* path, int device);
/* The same from arrays: node 0 is the root, node n >= 1 has parent[n] < n, is_leaf[n], desc[n*32 ..], weight[n]; word ids are given to the
 * leaves in node order, children are listed in node order (both as the loader does).  scoring: 0 L1_NORM .. 5 DOT_PRODUCT; weighting:
 * 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (BowVector.h:39-56). */
int orbx_vocabulary_create(orbx_vocabulary** out, int k, int L, int scoring, int weighting, int n_nodes, const int* parent,
                           const uint8_t* is_leaf, const uint8_t* desc, const double* weight, int device);
void orbx_vocabulary_destroy(orbx_vocabulary* v);
int orbx_vocabulary_info(const orbx_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words);
/* mBowVec / mFeatVec of n_frames device-resident frames (d_desc, d_n_out: mDescriptors, N of orbx_extract_batch_device):
 *   d_word_ids[f*capacity + j], d_word_weights[f*capacity + j], d_n_words[f] : the BowVector (std::map<WordId, WordValue>) in key order
 *   d_feat_nodes[f*capacity + j], d_feat_idx[f*capacity + j], d_n_feat[f]    : the FeatureVector (std::map<NodeId, vector<unsigned>>)
 *                                                                               flattened in (node, feature index) order
 * Asynchronous on the handle's stream; the vocabulary must live on the handle's device. */
int orbx_compute_bow_device(orbx_handle* h, const orbx_vocabulary* v, int n_frames, const uint8_t* d_desc, const int* d_n_out, int capacity,
                            int levels_up, uint32_t* d_word_ids, double* d_word_weights, int* d_n_words, uint32_t* d_feat_nodes,
                            uint32_t* d_feat_idx, int* d_n_feat);

/* ---- the consumer of the FeatureVector: ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) ------------------
 * (src/ORBmatcher.cc:269-471; Tracking::TrackReferenceKeyFrame, Tracking::Relocalization), Nleft == -1: for n_pairs pairs
 * (keyframe = frame kf_first + p*kf_step, current frame = frame cur_first + p*cur_step of one device-resident batch) the features of
 * the two frames that fell into the same vocabulary node are matched: nearest / second nearest by ORBmatcher::DescriptorDistance
 * among the frame's not yet matched features of the node, TH_LOW and the ratio test, rotation-histogram clean-up (ComputeThreeMaxima).
 *   d_feat_nodes, d_feat_idx, d_n_feat : mFeatVec of all frames as written by orbx_compute_bow_device (same capacity)
 *   d_kf_mp_flags[p*capacity + i]      : bit 0 = keypoint i of the keyframe holds a MapPoint that is not bad (:301-307)
 *   d_kps[f*capacity + i]              : mvKeys of all frames (only .angle is read: mvKeysUn keeps it, src/Frame.cc:776-780)
 *   d_desc, d_n_out                    : mDescriptors, N of all frames
 *   nn_ratio = mfNNratio, th_low = ORBmatcher::TH_LOW (50), check_orientation = mbCheckOrientation
 *   d_matches[p*capacity + i]          : out, the keyframe keypoint whose MapPoint keypoint i of the frame receives, -1 = none
 *   d_n_matches[p]                     : out, the return value
 * Asynchronous on the handle's stream.  Errors: ORBX_ERR_UNSUPPORTED if the tables (22 bytes per slot of the capacity rounded up to 16, + 64) do not fit a workgroup's LDS
 * (with 64 more bytes per slot the descriptors are staged in LDS too; without them they are read from L2). */
int orbx_search_by_bow_device(orbx_handle* h, int n_pairs, int kf_first, int kf_step, int cur_first, int cur_step,
                              const uint32_t* d_feat_nodes, const uint32_t* d_feat_idx, const int* d_n_feat, const uint8_t* d_kf_mp_flags,
                              const orbx_keypoint* d_kps, const uint8_t* d_desc, const int* d_n_out, int capacity, float nn_ratio,
                              int th_low, int check_orientation, int* d_matches, int* d_n_matches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (src/ORBmatcher.cc:823-963; LoopClosing.cc:624): as above between two
 * keyframes — a keypoint of pKF2 is a candidate only if it holds a good MapPoint and has not been matched (:879-888), the threshold is strict
 * (bestDist1 < TH_LOW, :908), and the result is indexed by pKF1's keypoints:
 *   d_kf1_mp_flags / d_kf2_mp_flags[p*capacity + i] : bit 0 = keypoint i of keyframe 1 / 2 holds a MapPoint that is not bad
 *   d_matches12[p*capacity + i]                     : out, the keypoint of pKF2 whose MapPoint vpMatches12[i] names, -1 = NULL */
int orbx_search_by_bow_keyframes_device(orbx_handle* h, int n_pairs, int kf1_first, int kf1_step, int kf2_first, int kf2_step,
                                        const uint32_t* d_feat_nodes, const uint32_t* d_feat_idx, const int* d_n_feat,
                                        const uint8_t* d_kf1_mp_flags, const uint8_t* d_kf2_mp_flags, const orbx_keypoint* d_kps,
                                        const uint8_t* d_desc, const int* d_n_out, int capacity, float nn_ratio, int th_low,
                                        int check_orientation, int* d_matches12, int* d_n_matches);

/* Stream control.  By default the handle owns a stream; orbx_set_stream adopts a caller stream
 * (hipStream_t passed as void*, e.g. torch.cuda.current_stream().cuda_stream) so the caller's events
 * and graphs see the work. */
int orbx_set_stream(orbx_handle* h, void* hip_stream);
void* orbx_get_stream(const orbx_handle* h);
int orbx_synchronize(orbx_handle* h);

/* ---- introspection used by tests and bench.py (not part of the reference surface) ------------- */
/* rounds the fixed-point projection search of the last launch needed for pair 0, and 100-MHz ticks of its staging / first scan / rounds */
int orbx_debug_search_rounds(int* out4);

/* Which launch forms the last call took (results never depend on them; the parity tests assert the form they mean to cover and the
 * published timings name theirs): pyramid_form 0 = k_pyr_cols (region-major, *pyramid_cut_px = side of its regions), 1 = k_pyr_first +
 * one k_resize per level; blur_form 0 = k_blur, 1 = lanes of the FAST launch, 3 = per keypoint inside k_describe (no blurred level exists:
 * orbx_debug_get_blurred has nothing to show), 5 = split by level: per keypoint below orbx_debug_last_split_level(), k_blur from that level on
 * (only those blurred levels exist).  (2 and 4 named forms that rounds 4 / 5 measured slower and round 6 removed.) */
int orbx_debug_last_forms(const orbx_handle* h, int* pyramid_form, int* pyramid_cut_px, int* blur_form);
/* the first level the last call blurred with k_blur when its blur form was 5; 0 otherwise */
int orbx_debug_last_split_level(const orbx_handle* h);

/* Test aids.  They cannot be set from the environment: a test calls this BEFORE orbx_create and the handles created afterwards carry the
 * setting.  name = "poison" (byte every device allocation of orbx_create is filled with; -1 = off), "lds_pollute" (byte every CU's LDS is
 * filled with in front of every kernel; -1 = off), "fail_after_fast" (1: the next handle's first small-batch call returns ORBX_ERR_HIP
 * between the FAST and the quad-tree launch, once), "pyr_cols_shape" (1, 4 or 6: pins the workgroup shape of k_pyr_cols; -1 = by the grid
 * size), "shared_upload_bytes" (host-buffer batches whose input is at least this large copy it through the device's shared copy queue
 * and bring the results back by DMA, smaller ones use the handle's stream and a copy kernel; -1 = 16 MiB).  Unknown name: ORBX_ERR_BAD_ARGUMENT. */
int orbx_debug_set_option(const char* name, int value);

/* The launch-policy switches as orbx_create read them, "NAME=value" separated by blanks, "(env)" behind a value that came from an ORBX_<NAME>
 * environment variable (read once, at orbx_create; they choose between result-identical launch forms), and the test aids in force, if any.
 * bench.py prints it as config.policy. */
const char* orbx_debug_policy(const orbx_handle* h);

/* The shader clock while the handle's work is running (bench.py: secondary.sustained).  orbx_debug_clock_probe enqueues, on a stream of
 * its own and without waiting, one sleeping wave per CU that reads the shader-clock and the 100-MHz real-time counters 50 us apart, into
 * slot 0..63; orbx_debug_clock_read waits for the probes launched so far and returns GHz per slot (0 for a slot never probed). */
int orbx_debug_clock_probe(orbx_handle* h, int slot);
int orbx_debug_clock_read(orbx_handle* h, int n_slots, double* ghz);

/* ORBX_HOST_TIMING=1 in the environment: wall seconds of the one-frame host call (orbx_extract_view) accumulated per phase since the last read:
 * out8[0] enqueue (staging + copy + launches), [1] wait, [2] pointer query, [3] staging memcpy, [4] staging + H2D enqueue; *calls = calls summed. */
int orbx_debug_host_timing(double* out8, long* calls);

/* Stage outputs of frame `frame` of the last batch, copied to host.  Candidates are the reference's
 * vToDistributeKeys of one level (ORBextractor.cc:786-864) in rectangle coordinates; their order is
 * unspecified (the octree result does not depend on it), sort before comparing. */
int orbx_debug_num_candidates(orbx_handle* h, int frame, int level, int* n);
int orbx_debug_get_candidates(orbx_handle* h, int frame, int level, orbx_keypoint* out, int capacity);
/* The 7x7 sigma-2 blurred level (ORBextractor.cc:1126-1127), width x height. */
int orbx_debug_get_blurred(orbx_handle* h, int frame, int level, uint8_t* dst, ptrdiff_t dst_stride);

/* Per-kernel timing with HIP events on the handle's stream.  enable=1 brackets every kernel launch of
 * subsequent extract calls with events (adds launch overhead: use for attribution, not for fps). */
#define ORBX_NUM_KERNELS 10
int orbx_profile_enable(orbx_handle* h, int enable);
int orbx_profile_reset(orbx_handle* h);
/* Resolves pending events; fills total milliseconds and launch counts per kernel slot. */
int orbx_profile_read(orbx_handle* h, double* total_ms, long* launches);
const char* orbx_profile_kernel_name(int slot);
/* ... and the kernel that last ran in that slot on this handle, by the name rocprofv3 prints (without template arguments): slot 1 is
 * k_pyr_cols or k_resize, slot 3 k_fast or k_fast_wide, slot 4 k_octree_256 / _512 / _1024 (+ r: the 128-VGPR build, g: node arrays in HBM).
 * bench.py keys roofline.kernel_ms_per_step with these, so that the line can be read next to profiles/ *_kernel_stats.md. */
const char* orbx_profile_kernel_name_of(const orbx_handle* h, int slot);

/* Algorithmic HBM bytes of one frame at this geometry (P0 + 2*S + 60*n_out). */
long orbx_algorithmic_bytes(const orbx_handle* h, int rows, int cols, int n_out);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
