// orbx_rows.cpp - the C ABI of the rows behind the extraction (include/orbx.h): ComputeStereoMatches, ComputeStereoFromRGBD, the
// caller's cvtColor, UndistortKeyPoints + AssignFeaturesToGrid, SearchForInitialization, SearchByProjection, the DBoW2 vocabulary, ComputeBoW and
// SearchByBoW.  Thin: argument checks, parameter structs, one launch wrapper each (the kernels are in k_stereo / k_frame / k_gray / k_match /
// k_project / k_bow / k_bow_match.hip).  No CPU path.
#include "orbx_internal.hpp"

extern "C" {

namespace {
// rows covered by one right keypoint's band [floor(y - 2s), ceil(y + 2s)], s = the coarsest level's scale
int bandRows(const orbx_handle* h) { return 2 * (int)std::ceil(2.0 * h->tabs.scale[h->nlevels - 1]) + 2; }

int stereoEnsure(orbx_handle* h, int nPairs, int capacity, int rows) {
    if (nPairs <= h->stereoPairs && capacity <= h->stereoCap && rows <= h->stereoRows) return ORBX_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    void* old[] = {h->d_rowOff, h->d_sadDist, h->d_nMatched, h->d_rowList, h->d_uRight, h->d_depth};
    for (void* p : old) if (p) (void)hipFree(p);
    h->d_rowOff = h->d_sadDist = h->d_nMatched = nullptr; h->d_rowList = nullptr; h->d_uRight = h->d_depth = nullptr;
    h->stereoPairs = nPairs > h->stereoPairs ? nPairs : h->stereoPairs;
    h->stereoCap = capacity > h->stereoCap ? capacity : h->stereoCap;
    h->stereoRows = rows > h->stereoRows ? rows : h->stereoRows;
    const size_t P = h->stereoPairs, C = h->stereoCap;
    HIP_TRY(h, hipMalloc(&h->d_rowOff, P * (h->stereoRows + 1) * sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->d_rowList, P * C * bandRows(h) * sizeof(unsigned short)));
    HIP_TRY(h, hipMalloc(&h->d_sadDist, P * C * sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->d_nMatched, P * sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->d_uRight, P * C * sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->d_depth, P * C * sizeof(float)));
    return ORBX_OK;
}

int stereoEnqueue(orbx_handle* h, int n_pairs, const Keypoint* d_kps, const uint8_t* d_desc, const int* d_n, int capacity,
                  float bf, float b, float* d_u, float* d_d, int* d_nm) {
    if (h->geom.nlevels == 0 || 2 * n_pairs > h->lastB || n_pairs < 1)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "stereo matching needs the 2*n_pairs frames of the last extract batch on this handle");
    if (!(b > 0.f) || !(bf > 0.f)) return fail(h, ORBX_ERR_BAD_ARGUMENT, "bf and b must be positive");
    if (capacity > 65535) return fail(h, ORBX_ERR_UNSUPPORTED, "capacity above 65535 keypoints per eye");
    HIP_TRY(h, hipSetDevice(h->device));
    const int rows = h->geom.rows;
    int rc = stereoEnsure(h, n_pairs, capacity, rows);
    if (rc != ORBX_OK) return rc;
    StereoParams sp;
    for (int l = 0; l < kMaxLevels; l++) { sp.scale[l] = l < h->nlevels ? h->tabs.scale[l] : 1.f; sp.invScale[l] = l < h->nlevels ? h->tabs.invScale[l] : 1.f; }
    sp.bf = bf; sp.b = b; sp.nlevels = h->nlevels; sp.capacity = capacity; sp.rowCap = h->stereoCap * bandRows(h);
    {
        Prof p(h, S_STEREO);
        launchStereo(h->stream, h->d_lv, h->d_pyr, d_kps, d_desc, d_n, sp, rows, h->d_rowOff, h->d_rowList, d_u, d_d, h->d_sadDist,
                     d_nm, n_pairs);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}
}  // namespace

int orbx_stereo_match_device(orbx_handle* h, int n_pairs, const orbx_keypoint* d_kps, const uint8_t* d_desc, const int* d_n_out,
                             int capacity, float bf, float b, float* d_u_right, float* d_depth, int* d_n_matched) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_kps || !d_desc || !d_n_out || !d_u_right || !d_depth || !d_n_matched || capacity < 1)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer or capacity < 1");
    return stereoEnqueue(h, n_pairs, (const Keypoint*)d_kps, d_desc, d_n_out, capacity, bf, b, d_u_right, d_depth, d_n_matched);
}

int orbx_stereo_match_last(orbx_handle* h, int n_pairs, float bf, float b, float* u_right, float* depth, int capacity,
                           int* n_matched) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!u_right || !depth || !n_matched || capacity < 1 || n_pairs < 1) return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity < 1 or n_pairs < 1");
    // the handle only holds results of the host-buffer path (orbx_extract / orbx_extract_batch / _begin + _end); after
    // orbx_extract_batch_device the results live in the caller's buffers: use orbx_stereo_match_device there
    if (h->pendingB) return fail(h, ORBX_ERR_BAD_ARGUMENT, "a batch is still in flight: call orbx_extract_batch_end first");
    if (2 * n_pairs > h->lastHostB)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "orbx_stereo_match_last needs the 2*n_pairs frames of the last orbx_extract_batch call on this handle "
                                              "(after orbx_extract_batch_device use orbx_stereo_match_device)");
    const int cap = h->outCap;
    for (int p = 0; p < n_pairs; p++) {     // counts of the left eyes, copied to the host by that call
        const int n = h->host.n[2 * p];
        if (n < 0 || n > cap) return fail(h, ORBX_ERR_HIP, "corrupt keypoint count in the handle's staging (internal)");
        if (n > capacity) return fail(h, ORBX_ERR_CAPACITY, "capacity smaller than the left keypoint count");
    }
    int rc = stereoEnsure(h, n_pairs, cap, h->geom.rows > 0 ? h->geom.rows : 1);
    if (rc != ORBX_OK) return rc;
    rc = stereoEnqueue(h, n_pairs, h->dev.k, h->dev.d, h->dev.n, cap, bf, b, h->d_uRight, h->d_depth, h->d_nMatched);
    if (rc != ORBX_OK) return rc;
    std::vector<float> hu((size_t)n_pairs * cap), hd((size_t)n_pairs * cap);
    HIP_TRY(h, hipMemcpyAsync(hu.data(), h->d_uRight, hu.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(hd.data(), h->d_depth, hd.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(n_matched, h->d_nMatched, sizeof(int) * n_pairs, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int p = 0; p < n_pairs; p++) {
        const int n = h->host.n[2 * p];
        std::memcpy(u_right + (size_t)p * capacity, hu.data() + (size_t)p * cap, sizeof(float) * n);
        std::memcpy(depth + (size_t)p * capacity, hd.data() + (size_t)p * cap, sizeof(float) * n);
    }
    return ORBX_OK;
}

namespace {
// cv::undistortPoints for one point, host twin of the device routine in k_frame.hip (same operation order, doubles;
// this file is compiled with -ffp-contract=off)
void undistortHost(const orbx_camera& c, float xin, float yin, float* xo, float* yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy, ifx = 1. / fx, ify = 1. / fy;
    const double k[12] = {c.k1, c.k2, c.p1, c.p2, c.k3, 0, 0, 0, 0, 0, 0, 0};
    double x = xin, y = yin;
    const double u = x, v = y;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww); *yo = (float)(yy * ww);
}
}  // namespace

int orbx_compute_image_bounds(const orbx_camera* cam, int cols, int rows, float* b) {
    if (!cam || !b || cols < 1 || rows < 1 || !(cam->fx > 0.f) || !(cam->fy > 0.f)) return ORBX_ERR_BAD_ARGUMENT;
    if (cam->k1 != 0.0f) {
        float m[4][2] = {{0.f, 0.f}, {(float)cols, 0.f}, {0.f, (float)rows}, {(float)cols, (float)rows}};
        for (int i = 0; i < 4; i++) undistortHost(*cam, m[i][0], m[i][1], &m[i][0], &m[i][1]);
        b[0] = m[0][0] < m[2][0] ? m[0][0] : m[2][0]; b[1] = m[1][0] > m[3][0] ? m[1][0] : m[3][0];
        b[2] = m[0][1] < m[1][1] ? m[0][1] : m[1][1]; b[3] = m[2][1] > m[3][1] ? m[2][1] : m[3][1];
    } else {
        b[0] = 0.0f; b[1] = (float)cols; b[2] = 0.0f; b[3] = (float)rows;
    }
    return ORBX_OK;
}

int orbx_stereo_from_rgbd_device(orbx_handle* h, int n_frames, const orbx_keypoint* d_kps, const orbx_keypoint* d_kps_un,
                                 const int* d_n_out, int capacity, const void* d_depth, int depth_is_u16, int rows, int cols,
                                 ptrdiff_t depth_stride_bytes, ptrdiff_t depth_frame_stride_bytes, float depth_map_factor, float mbf,
                                 float* d_u_right, float* d_depth_out) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    const ptrdiff_t elem = depth_is_u16 ? 2 : 4;
    if (!d_kps || !d_kps_un || !d_n_out || !d_depth || !d_u_right || !d_depth_out || capacity < 1 || n_frames < 1 || rows < 1 || cols < 1 ||
        depth_stride_bytes < (ptrdiff_t)cols * elem || (depth_stride_bytes % elem) != 0 || (depth_frame_stride_bytes % elem) != 0 ||
        ((uintptr_t)d_depth % elem) != 0 || n_frames > 65535)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/n_frames/rows/cols < 1, or a depth stride / pointer that is not a multiple of the element size");
    HIP_TRY(h, hipSetDevice(h->device));
    RgbdParams p;
    p.capacity = capacity; p.rows = rows; p.cols = cols; p.isU16 = depth_is_u16 != 0;
    p.scale = p.isU16 || std::fabs(depth_map_factor - 1.0f) > 1e-5f;      // Tracking.cc:1003
    p.stride = depth_stride_bytes; p.frame = depth_frame_stride_bytes; p.factor = depth_map_factor; p.mbf = mbf;
    launchStereoFromRgbd(h->stream, (const Keypoint*)d_kps, (const Keypoint*)d_kps_un, d_n_out, (const uint8_t*)d_depth, p, d_u_right,
                         d_depth_out, n_frames);
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

int orbx_gray_from_color_device(orbx_handle* h, int n_frames, const uint8_t* d_src, int rows, int cols, int channels, int red_first,
                                ptrdiff_t src_stride, ptrdiff_t src_frame_stride, uint8_t* d_gray, ptrdiff_t gray_stride,
                                ptrdiff_t gray_frame_stride) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_src || !d_gray || n_frames < 1 || rows < 1 || cols < 1 || (channels != 3 && channels != 4) ||
        src_stride < (ptrdiff_t)cols * channels || gray_stride < cols || rows > 65535 || n_frames > 65535)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, channels not 3 or 4, stride shorter than a row, or more than 65535 rows / frames");
    HIP_TRY(h, hipSetDevice(h->device));
    GrayParams p;
    p.rows = rows; p.cols = cols; p.channels = channels; p.redFirst = red_first != 0;
    p.srcStride = src_stride; p.srcFrame = src_frame_stride; p.dstStride = gray_stride; p.dstFrame = gray_frame_stride;
    p.aligned = (((uintptr_t)d_src | (uintptr_t)src_stride | (uintptr_t)src_frame_stride) & 3) == 0;
    launchGray(h->stream, d_src, d_gray, p, n_frames);
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

static int frameFinish(orbx_handle* h, int n_frames, const orbx_keypoint* d_kps, const int* d_n_out, int capacity, const orbx_camera* cam,
                       const float* bounds4, orbx_keypoint* d_kps_un, int* d_grid_off, int* d_grid_idx, int* d_n_inside, int rawGrid);

int orbx_frame_finish_device(orbx_handle* h, int n_frames, const orbx_keypoint* d_kps, const int* d_n_out, int capacity,
                             const orbx_camera* cam, const float* bounds4, orbx_keypoint* d_kps_un, int* d_grid_off,
                             int* d_grid_idx, int* d_n_inside) {
    return frameFinish(h, n_frames, d_kps, d_n_out, capacity, cam, bounds4, d_kps_un, d_grid_off, d_grid_idx, d_n_inside, 0);
}

int orbx_frame_finish_two_eyes_device(orbx_handle* h, int n_pairs, const orbx_keypoint* d_kps, const int* d_n_out, int capacity,
                                      const orbx_camera* cam, const float* bounds4, orbx_keypoint* d_kps_un, int* d_grid_off,
                                      int* d_grid_idx, int* d_n_inside) {
    if (n_pairs < 1 || n_pairs > (1 << 29)) return h ? fail(h, ORBX_ERR_BAD_ARGUMENT, "n_pairs < 1") : ORBX_ERR_BAD_ARGUMENT;
    return frameFinish(h, 2 * n_pairs, d_kps, d_n_out, capacity, cam, bounds4, d_kps_un, d_grid_off, d_grid_idx, d_n_inside, 1);
}

static int frameFinish(orbx_handle* h, int n_frames, const orbx_keypoint* d_kps, const int* d_n_out, int capacity, const orbx_camera* cam,
                       const float* bounds4, orbx_keypoint* d_kps_un, int* d_grid_off, int* d_grid_idx, int* d_n_inside, int rawGrid) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_kps || !d_n_out || !cam || !bounds4 || !d_kps_un || !d_grid_off || !d_grid_idx || !d_n_inside || capacity < 1 ||
        n_frames < 1 || !(cam->fx > 0.f) || !(cam->fy > 0.f) || !(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2]))
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/n_frames < 1, non-positive focal length or empty bounds");
    if (capacity > 32767) return fail(h, ORBX_ERR_UNSUPPORTED, "capacity above 32767 keypoints per frame");
    HIP_TRY(h, hipSetDevice(h->device));
    FrameFinishParams p;
    p.cam = CameraParams{cam->fx, cam->fy, cam->cx, cam->cy, cam->k1, cam->k2, cam->p1, cam->p2, cam->k3};
    p.minX = bounds4[0]; p.minY = bounds4[2];
    p.wInv = 64.0f / (bounds4[1] - bounds4[0]);      // mfGridElementWidthInv  (Frame.cc:339)
    p.hInv = 48.0f / (bounds4[3] - bounds4[2]);      // mfGridElementHeightInv (Frame.cc:340)
    p.capacity = capacity;
    p.rawGrid = rawGrid;
    {
        Prof pr(h, S_FRAME);
        launchFrameFinish(h->stream, (const Keypoint*)d_kps, d_n_out, p, (Keypoint*)d_kps_un, d_grid_off, d_grid_idx, d_n_inside, n_frames);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

int orbx_search_for_initialization_device(orbx_handle* h, int n_pairs, int frame1_first, int frame1_step, int frame2_first,
                                          int frame2_step, const orbx_keypoint* d_kps_un, const uint8_t* d_desc,
                                          const int* d_n_out, int capacity, const int* d_grid_off, const int* d_grid_idx,
                                          const float* bounds4, float* d_prev_matched, int window_size, float nn_ratio,
                                          int check_orientation, int* d_matches12, int* d_n_matches) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_kps_un || !d_desc || !d_n_out || !d_grid_off || !d_grid_idx || !bounds4 || !d_prev_matched || !d_matches12 ||
        !d_n_matches || capacity < 1 || n_pairs < 1 || frame1_first < 0 || frame2_first < 0 || frame1_step < 0 || frame2_step < 0 ||
        window_size < 0 || !(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2]))
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/n_pairs < 1, negative frame index/step/window or empty bounds");
    if (capacity > 32767) return fail(h, ORBX_ERR_UNSUPPORTED, "capacity above 32767 keypoints per frame");
    const int slotCap = initMatchSlotCapacity(capacity);
    if (slotCap < 64)
        return fail(h, ORBX_ERR_UNSUPPORTED, "capacity too large for the LDS-resident search (4 bytes per keypoint + 52 per level-0 keypoint of frame 2, 160 KB per CU)");
    HIP_TRY(h, hipSetDevice(h->device));
    InitMatchParams p;
    p.minX = bounds4[0]; p.minY = bounds4[2];
    p.wInv = 64.0f / (bounds4[1] - bounds4[0]);      // mfGridElementWidthInv  (Frame.cc:339)
    p.hInv = 48.0f / (bounds4[3] - bounds4[2]);      // mfGridElementHeightInv (Frame.cc:340)
    p.r = (float)window_size; p.nnRatio = nn_ratio; p.checkOrientation = check_orientation != 0; p.capacity = capacity; p.slotCapacity = slotCap;
    p.f1First = frame1_first; p.f1Step = frame1_step; p.f2First = frame2_first; p.f2Step = frame2_step;
    {
        Prof pr(h, S_FRAME);
        launchSearchInit(h->stream, (const Keypoint*)d_kps_un, d_desc, d_n_out, d_grid_off, d_grid_idx, p, d_prev_matched, d_matches12,
                         d_n_matches, n_pairs);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

int orbx_project_last_frame_device(orbx_handle* h, int n_pairs, int last_first, int last_step, int cur_first, int cur_step,
                                   const orbx_keypoint* d_kps, const orbx_keypoint* d_kps_un, const int* d_n_out, int capacity,
                                   const uint8_t* d_mp_flags, const float* d_world, const float* d_poses, const orbx_camera* cam,
                                   const float* bounds4, float mbf, float mb, float th, int mono, orbx_proj_query* d_queries) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_kps || !d_kps_un || !d_n_out || !d_mp_flags || !d_world || !d_poses || !cam || !bounds4 || !d_queries || capacity < 1 || n_pairs < 1 ||
        n_pairs > 65535 || last_first < 0 || cur_first < 0 || last_step < 0 || cur_step < 0 || !(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2]))
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/n_pairs < 1, more than 65535 pairs, negative frame index/step or empty bounds");
    HIP_TRY(h, hipSetDevice(h->device));
    ProjectParams p;
    p.fx = cam->fx; p.fy = cam->fy; p.cx = cam->cx; p.cy = cam->cy;
    p.minX = bounds4[0]; p.maxX = bounds4[1]; p.minY = bounds4[2]; p.maxY = bounds4[3];
    for (int l = 0; l < kMaxLevels; l++) p.scale[l] = l < h->nlevels ? h->tabs.scale[l] : h->tabs.scale[h->nlevels - 1];   // CurrentFrame.mvScaleFactors
    p.mbf = mbf; p.mb = mb; p.th = th; p.mono = mono != 0; p.capacity = capacity;
    p.lastFirst = last_first; p.lastStep = last_step; p.curFirst = cur_first; p.curStep = cur_step;
    {
        Prof pr(h, S_FRAME);
        launchProjectLast(h->stream, (const Keypoint*)d_kps, (const Keypoint*)d_kps_un, d_n_out, d_mp_flags, d_world, d_poses, p, (ProjQuery*)d_queries, n_pairs);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

int orbx_search_by_projection_device(orbx_handle* h, int n_pairs, int cur_first, int cur_step, const orbx_proj_query* d_queries,
                                     const uint8_t* d_query_desc, int desc_first, int desc_step, const int* d_n_queries, int query_capacity,
                                     const orbx_keypoint* d_kps_un, const uint8_t* d_desc, const int* d_n_out, int capacity,
                                     const int* d_grid_off, const int* d_grid_idx, const float* bounds4, const float* d_u_right,
                                     uint8_t* d_occupied, int ratio_mode, float nn_ratio, int max_distance, int check_orientation,
                                     int* d_matches, int* d_n_matches) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_queries || !d_query_desc || !d_kps_un || !d_desc || !d_n_out || !d_grid_off || !d_grid_idx || !bounds4 || !d_matches || !d_n_matches ||
        capacity < 1 || query_capacity < 1 || n_pairs < 1 || cur_first < 0 || cur_step < 0 || desc_first < 0 || desc_step < 0 || max_distance < 0 ||
        !(bounds4[1] > bounds4[0]) || !(bounds4[3] > bounds4[2]))
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/query_capacity/n_pairs < 1, negative frame index/step or empty bounds");
    if (capacity > 32767) return fail(h, ORBX_ERR_UNSUPPORTED, "capacity above 32767 keypoints per frame");
    if (projSearchLdsBytes(capacity, query_capacity, false) > 160 * 1024 - 512)
        return fail(h, ORBX_ERR_UNSUPPORTED, "capacity too large for the LDS-resident search (64 bytes per keypoint, 160 KB per CU)");
    HIP_TRY(h, hipSetDevice(h->device));
    ProjSearchParams p;
    p.minX = bounds4[0]; p.minY = bounds4[2];
    p.wInv = 64.0f / (bounds4[1] - bounds4[0]);      // mfGridElementWidthInv  (Frame.cc:339)
    p.hInv = 48.0f / (bounds4[3] - bounds4[2]);      // mfGridElementHeightInv (Frame.cc:340)
    p.nnRatio = nn_ratio; p.ratioMode = ratio_mode != 0; p.checkOrientation = check_orientation != 0;
    p.capacity = capacity; p.queryCapacity = query_capacity; p.curFirst = cur_first; p.curStep = cur_step;
    p.descFirst = desc_first; p.descStep = desc_step; p.maxDist = max_distance < 255 ? max_distance : 255;
    {
        Prof pr(h, S_FRAME);
        launchSearchProj(h->stream, (const ProjQuery*)d_queries, d_query_desc, d_n_queries, (const Keypoint*)d_kps_un, d_desc, d_n_out, d_grid_off,
                         d_grid_idx, d_u_right, d_occupied, p, d_matches, d_n_matches, n_pairs);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

}  // extern "C"

struct orbx_vocabulary {
    int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, nNodes = 0, nWords = 0;
    int *d_childOff = nullptr, *d_childList = nullptr;
    uint32_t *d_desc = nullptr, *d_wordId = nullptr;
    double* d_weight = nullptr;
};

extern "C" {

void orbx_vocabulary_destroy(orbx_vocabulary* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    void* dev[] = {v->d_childOff, v->d_childList, v->d_desc, v->d_wordId, v->d_weight};
    for (void* p : dev) if (p) (void)hipFree(p);
    delete v;
}

int orbx_vocabulary_create(orbx_vocabulary** out, int k, int L, int scoring, int weighting, int n_nodes, const int* parent,
                           const uint8_t* is_leaf, const uint8_t* desc, const double* weight, int device) {
    if (!out) return ORBX_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3 ||      // TemplatedVocabulary.h:1359
        n_nodes < 2 || !parent || !is_leaf || !desc || !weight) {
        g_createError = "orbx_vocabulary_create: bad argument (0<=k<=20, 1<=L<=10, scoring 0..5, weighting 0..3, at least a root and one node)";
        return ORBX_ERR_BAD_ARGUMENT;
    }
    // children in node order (m_nodes[pid].children.push_back(nid), :1392), word ids to the leaves in node order (:1409-1415)
    std::vector<int> cnt(n_nodes + 1, 0), off(n_nodes + 1, 0), list(n_nodes - 1);
    for (int n = 1; n < n_nodes; n++) {
        if (parent[n] < 0 || parent[n] >= n) { g_createError = "orbx_vocabulary_create: parent[n] must name an earlier node"; return ORBX_ERR_BAD_ARGUMENT; }
        cnt[parent[n]]++;
    }
    for (int n = 0; n < n_nodes; n++) off[n + 1] = off[n] + cnt[n];
    std::vector<int> fill(off.begin(), off.end() - 1);
    for (int n = 1; n < n_nodes; n++) list[fill[parent[n]]++] = n;
    std::vector<uint32_t> wid(n_nodes, 0);
    int words = 0;
    for (int n = 1; n < n_nodes; n++) {
        if (is_leaf[n]) { if (cnt[n]) { g_createError = "orbx_vocabulary_create: a leaf with children"; return ORBX_ERR_BAD_ARGUMENT; } wid[n] = (uint32_t)words++; }
        else if (!cnt[n]) { g_createError = "orbx_vocabulary_create: an inner node without children (the reference would treat it as a word without an id)"; return ORBX_ERR_BAD_ARGUMENT; }
    }
    if (!cnt[0]) { g_createError = "orbx_vocabulary_create: the root has no children"; return ORBX_ERR_BAD_ARGUMENT; }
    for (int n = 0; n < n_nodes; n++)      // k_bow_words packs (distance << 8 | child rank): a node's fan-out must fit the rank byte
        if (cnt[n] > 256) { g_createError = "orbx_vocabulary_create: a node with more than 256 children (the header's k allows at most 20)"; return ORBX_ERR_BAD_ARGUMENT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_createError = "orbx_vocabulary_create: no HIP device (this library has no CPU path)"; return ORBX_ERR_NO_DEVICE; }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return ORBX_ERR_HIP;
    if (device >= ndev) { g_createError = "orbx_vocabulary_create: device index out of range"; return ORBX_ERR_BAD_ARGUMENT; }
    orbx_vocabulary* v = new orbx_vocabulary();
    v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nNodes = n_nodes; v->nWords = words;
    // the uploads go through a stream of their own and the function returns when THAT stream has drained: the tables have landed before any
    // handle's stream can be given the vocabulary, and no other work on the device is waited for (docs/history/DESIGN_rounds_1-5.md §4j)
    hipStream_t us = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&us, hipStreamNonBlocking) != hipSuccess) {
        g_createError = "orbx_vocabulary_create: cannot create the upload stream";
        delete v;
        return ORBX_ERR_HIP;
    }
    auto up = [&](void** d, const void* src, size_t bytes) {
        return hipMalloc(d, bytes ? bytes : 4) == hipSuccess && (bytes == 0 || hipMemcpyAsync(*d, src, bytes, hipMemcpyHostToDevice, us) == hipSuccess);
    };
    std::vector<double> w0(weight, weight + n_nodes);
    if (!up((void**)&v->d_childOff, off.data(), sizeof(int) * (n_nodes + 1)) ||
        !up((void**)&v->d_childList, list.data(), sizeof(int) * list.size()) || !up((void**)&v->d_desc, desc, (size_t)n_nodes * 32) ||
        !up((void**)&v->d_wordId, wid.data(), sizeof(uint32_t) * n_nodes) || !up((void**)&v->d_weight, w0.data(), sizeof(double) * n_nodes)) {
        g_createError = "orbx_vocabulary_create: device allocation or copy failed";
        (void)hipStreamSynchronize(us);
        (void)hipStreamDestroy(us);
        orbx_vocabulary_destroy(v);
        return ORBX_ERR_HIP;
    }
    const hipError_t landed = hipStreamSynchronize(us);
    (void)hipStreamDestroy(us);
    if (landed != hipSuccess) { g_createError = "orbx_vocabulary_create: upload failed"; orbx_vocabulary_destroy(v); return ORBX_ERR_HIP; }
    *out = v;
    return ORBX_OK;
}

int orbx_vocabulary_load_text(orbx_vocabulary** out, const char* path, int device) {
    if (!out) return ORBX_ERR_BAD_ARGUMENT;
    *out = nullptr;
    FILE* f = path ? std::fopen(path, "r") : nullptr;
    if (!f) { g_createError = "orbx_vocabulary_load_text: cannot open the file"; return ORBX_ERR_BAD_ARGUMENT; }
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (std::fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4) { std::fclose(f); g_createError = "orbx_vocabulary_load_text: bad header"; return ORBX_ERR_BAD_ARGUMENT; }
    std::vector<int> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    for (;;) {      // one node per line: parent, is-leaf, FORB::L = 32 descriptor bytes, weight (TemplatedVocabulary.h:1378-1419)
        int pid = 0, isLeaf = 0;
        if (std::fscanf(f, "%d %d", &pid, &isLeaf) != 2) break;
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32 && ok; i++) { int b = 0; ok = std::fscanf(f, "%d", &b) == 1; d[i] = (uint8_t)b; }
        double w = 0;
        if (!ok || std::fscanf(f, "%lf", &w) != 1) { std::fclose(f); g_createError = "orbx_vocabulary_load_text: truncated node line"; return ORBX_ERR_BAD_ARGUMENT; }
        parent.push_back(pid); leaf.push_back(isLeaf > 0); weight.push_back(w);
        desc.insert(desc.end(), d, d + 32);
    }
    std::fclose(f);
    return orbx_vocabulary_create(out, k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), device);
}

int orbx_vocabulary_info(const orbx_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words) {
    if (!v) return ORBX_ERR_BAD_ARGUMENT;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = v->nNodes;
    if (n_words) *n_words = v->nWords;
    return ORBX_OK;
}

int orbx_compute_bow_device(orbx_handle* h, const orbx_vocabulary* v, int n_frames, const uint8_t* d_desc, const int* d_n_out, int capacity,
                            int levels_up, uint32_t* d_word_ids, double* d_word_weights, int* d_n_words, uint32_t* d_feat_nodes,
                            uint32_t* d_feat_idx, int* d_n_feat) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!v || !d_desc || !d_n_out || !d_word_ids || !d_word_weights || !d_n_words || !d_feat_nodes || !d_feat_idx || !d_n_feat || capacity < 1 ||
        n_frames < 1 || n_frames > 65535)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/n_frames < 1 or more than 65535 frames");
    if (v->device != h->device) return fail(h, ORBX_ERR_BAD_ARGUMENT, "the vocabulary lives on another device than the handle");
    if (capacity > 16384) return fail(h, ORBX_ERR_UNSUPPORTED, "capacity above 16384 keypoints per frame (the per-frame sort runs in LDS)");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t need = (size_t)n_frames * capacity;
    if (need > h->bowEntries) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        void* old[] = {h->d_bowWord, h->d_bowNode, h->d_bowWeight};
        for (void* p : old) if (p) (void)hipFree(p);
        h->d_bowWord = h->d_bowNode = nullptr; h->d_bowWeight = nullptr; h->bowEntries = 0;
        HIP_TRY(h, hipMalloc(&h->d_bowWord, need * sizeof(uint32_t)));
        HIP_TRY(h, hipMalloc(&h->d_bowNode, need * sizeof(uint32_t)));
        HIP_TRY(h, hipMalloc(&h->d_bowWeight, need * sizeof(double)));
        h->bowEntries = need;
    }
    VocabDevice V{v->d_childOff, v->d_childList, v->d_desc, v->d_weight, v->d_wordId, v->nNodes, v->k, v->L, v->scoring, v->weighting};
    {
        Prof pr(h, S_FRAME);
        launchBow(h->stream, V, d_desc, d_n_out, capacity, levels_up, h->d_bowWord, h->d_bowWeight, h->d_bowNode, d_word_ids, d_word_weights, d_n_words,
                  d_feat_nodes, d_feat_idx, d_n_feat, n_frames);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}

static int searchByBow(orbx_handle* h, int n_pairs, int kf_first, int kf_step, int cur_first, int cur_step, const uint32_t* d_feat_nodes,
                       const uint32_t* d_feat_idx, const int* d_n_feat, const uint8_t* d_kf_mp_flags, const uint8_t* d_cur_mp_flags,
                       const orbx_keypoint* d_kps, const uint8_t* d_desc, const int* d_n_out, int capacity, float nn_ratio, int th_low,
                       int check_orientation, int* d_matches, int* d_n_matches) {
    if (!h) return ORBX_ERR_BAD_ARGUMENT;
    if (!d_feat_nodes || !d_feat_idx || !d_n_feat || !d_kf_mp_flags || !d_kps || !d_desc || !d_n_out || !d_matches || !d_n_matches ||
        capacity < 1 || n_pairs < 1 || kf_first < 0 || cur_first < 0 || kf_first + (long long)(n_pairs - 1) * kf_step < 0 ||
        cur_first + (long long)(n_pairs - 1) * cur_step < 0)
        return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer, capacity/n_pairs < 1 or a negative frame index");
    if (capacity > 65535 || bowMatchLdsBytes(capacity, false) > 150 * 1024) return fail(h, ORBX_ERR_UNSUPPORTED, "capacity too large: the node columns and the match table of a pair live in LDS");
    HIP_TRY(h, hipSetDevice(h->device));
    BowMatchParams p{nn_ratio, th_low, check_orientation ? 1 : 0, capacity, kf_first, kf_step, cur_first, cur_step, d_cur_mp_flags ? 1 : 0};
    {
        Prof pr(h, S_FRAME);
        launchSearchBow(h->stream, d_feat_nodes, d_feat_idx, d_n_feat, d_kf_mp_flags, d_cur_mp_flags, (const Keypoint*)d_kps, d_desc, d_n_out, p, d_matches,
                        d_n_matches, n_pairs);
    }
    HIP_TRY(h, hipGetLastError());
    return ORBX_OK;
}
int orbx_search_by_bow_device(orbx_handle* h, int n_pairs, int kf_first, int kf_step, int cur_first, int cur_step,
                              const uint32_t* d_feat_nodes, const uint32_t* d_feat_idx, const int* d_n_feat, const uint8_t* d_kf_mp_flags,
                              const orbx_keypoint* d_kps, const uint8_t* d_desc, const int* d_n_out, int capacity, float nn_ratio,
                              int th_low, int check_orientation, int* d_matches, int* d_n_matches) {
    return searchByBow(h, n_pairs, kf_first, kf_step, cur_first, cur_step, d_feat_nodes, d_feat_idx, d_n_feat, d_kf_mp_flags, nullptr, d_kps, d_desc,
                       d_n_out, capacity, nn_ratio, th_low, check_orientation, d_matches, d_n_matches);
}
int orbx_search_by_bow_keyframes_device(orbx_handle* h, int n_pairs, int kf1_first, int kf1_step, int kf2_first, int kf2_step,
                                        const uint32_t* d_feat_nodes, const uint32_t* d_feat_idx, const int* d_n_feat,
                                        const uint8_t* d_kf1_mp_flags, const uint8_t* d_kf2_mp_flags, const orbx_keypoint* d_kps,
                                        const uint8_t* d_desc, const int* d_n_out, int capacity, float nn_ratio, int th_low,
                                        int check_orientation, int* d_matches12, int* d_n_matches) {
    if (h && !d_kf2_mp_flags) return fail(h, ORBX_ERR_BAD_ARGUMENT, "null pointer (the second keyframe's MapPoint flags)");
    return searchByBow(h, n_pairs, kf1_first, kf1_step, kf2_first, kf2_step, d_feat_nodes, d_feat_idx, d_n_feat, d_kf1_mp_flags, d_kf2_mp_flags, d_kps,
                       d_desc, d_n_out, capacity, nn_ratio, th_low, check_orientation, d_matches12, d_n_matches);
}

}  // extern "C"
