// k_frame.hip — "next" row SURVEY.md §8f-3: what the Frame constructor does with the freshly extracted keypoints,
// on the device-resident results: Frame::UndistortKeyPoints (reference src/Frame.cc:748-782) and
// AssignFeaturesToGrid (:383-417) with PosInGrid (:726-736).
//
// cv::undistortPoints (OpenCV 3.4 cvUndistortPointsInternal, 4/5-coefficient model, no rectification, P = K): five
// fixed-point iterations in double precision and the re-projection, every operation a separate IEEE rounding.
// The grid is the reference's mGrid[64][48] as CSR (cell = x*48 + y): one workgroup per frame counts keypoints
// per cell, scans, and ranks every keypoint among the earlier keypoints of its cell, which reproduces the
// push_back order (increasing keypoint index inside a cell).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.hpp"

namespace orbx {

struct CameraParams { float fx, fy, cx, cy, k1, k2, p1, p2, k3; };
struct FrameFinishParams {
    CameraParams cam;
    float minX, minY, wInv, hInv;   // mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
    int capacity;
    int rawGrid;                    // the Nleft != -1 branch of AssignFeaturesToGrid (:404-414): the cells come from mvKeys / mvKeysRight, not from mvKeysUn
};

constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;   // FRAME_GRID_COLS/ROWS, inc/Frame.h:39-40

__device__ __forceinline__ void undistortPoint(const CameraParams& c, float xin, float yin, float* xo, float* yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    const double ifx = __ddiv_rn(1.0, fx), ify = __ddiv_rn(1.0, fy);
    const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2, k4 = c.k3;
    const double u = xin, v = yin;
    double x = __dmul_rn(__dsub_rn(u, cx), ifx), y = __dmul_rn(__dsub_rn(v, cy), ify);
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
        // numerator 1 + ((k7*r2 + k6)*r2 + k5)*r2 with k5..k7 = 0 is exactly 1
        const double den = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k4, r2), k1), r2), k0), r2));
        const double icdist = __ddiv_rn(1.0, den);
        if (icdist < 0) { x = __dmul_rn(__dsub_rn(u, cx), ifx); y = __dmul_rn(__dsub_rn(v, cy), ify); break; }
        // deltaX = 2*k2*x*y + k3*(r2 + 2*x*x) + k8*r2 + k9*r2*r2  (k8..k11 = 0: the last two terms add +0.0)
        const double dX = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(__dmul_rn(__dmul_rn(2.0, k2), x), y),
                                                         __dmul_rn(k3, __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, x), x)))),
                                              __dmul_rn(0.0, r2)), __dmul_rn(__dmul_rn(0.0, r2), r2));
        const double dY = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(k2, __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, y), y))),
                                                         __dmul_rn(__dmul_rn(__dmul_rn(2.0, k3), x), y)),
                                              __dmul_rn(0.0, r2)), __dmul_rn(__dmul_rn(0.0, r2), r2));
        x = __dmul_rn(__dsub_rn(x0, dX), icdist);
        y = __dmul_rn(__dsub_rn(y0, dY), icdist);
    }
    const double xx = __dadd_rn(__dadd_rn(__dmul_rn(fx, x), __dmul_rn(0.0, y)), cx);
    const double yy = __dadd_rn(__dadd_rn(__dmul_rn(0.0, x), __dmul_rn(fy, y)), cy);
    const double ww = __ddiv_rn(1.0, __dadd_rn(__dadd_rn(__dmul_rn(0.0, x), __dmul_rn(0.0, y)), 1.0));
    *xo = (float)__dmul_rn(xx, ww); *yo = (float)__dmul_rn(yy, ww);
}

// grid: n_frames; 1024 threads.
__global__ __launch_bounds__(1024) void k_frame_finish(const Keypoint* __restrict__ kps, const int* __restrict__ nOut,
                                                        FrameFinishParams p, Keypoint* __restrict__ kpsUn,
                                                        int* __restrict__ gridOff, int* __restrict__ gridIdx,
                                                        int* __restrict__ nInside) {
    extern __shared__ __align__(16) int lds[];
    int* cnt = lds;                       // [kGridCells + 1]
    short* cellOf = (short*)(cnt + kGridCells + 4);   // [capacity] cell of every keypoint, -1 outside the grid (16-byte aligned: read eight at a time)
    __shared__ int wsum[16];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = min(nOut[f], p.capacity);
    const Keypoint* in = kps + (long long)f * p.capacity;
    Keypoint* un = kpsUn + (long long)f * p.capacity;
    for (int i = tid; i <= kGridCells; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        Keypoint k = in[i];
        const float rawX = k.x, rawY = k.y;
        if (p.cam.k1 != 0.0f) undistortPoint(p.cam, k.x, k.y, &k.x, &k.y);        // else mvKeysUn = mvKeys (:750-754)
        un[i] = k;
        const float gx = p.rawGrid ? rawX : k.x, gy = p.rawGrid ? rawY : k.y;      // (:405-407: mvKeysUn, or the eye's own raw keys)
        const int posX = (int)roundf(__fmul_rn(__fsub_rn(gx, p.minX), p.wInv));   // PosInGrid (:728-729)
        const int posY = (int)roundf(__fmul_rn(__fsub_rn(gy, p.minY), p.hInv));
        const bool ok = !(posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows);
        const int cell = ok ? posX * kGridRows + posY : -1;
        cellOf[i] = (short)cell;
        if (ok) atomicAdd(&cnt[cell], 1);
    }
    __syncthreads();
    // exclusive scan of the 3072 cell counts (3 per thread)
    const int per = (kGridCells + 1023) / 1024, b0 = tid * per, e0 = min(b0 + per, kGridCells);
    int sum = 0;
    for (int i = b0; i < e0; i++) sum += cnt[i];
    const int incl = waveInclusiveScan(sum);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; w++) off += wsum[w];
    int run = off + incl - sum;
    int* go = gridOff + (long long)f * (kGridCells + 1);
    for (int i = b0; i < e0; i++) { const int c = cnt[i]; go[i] = run; cnt[i] = run; run += c; }   // cnt now holds the cell's first slot
    if (tid == 1023) { go[kGridCells] = run; nInside[f] = run; }
    __syncthreads();
    int* gi = gridIdx + (long long)f * p.capacity;
    for (int i = tid; i < N; i += 1024) {
        const int cell = cellOf[i];
        if (cell < 0) continue;
        // earlier keypoints of the same cell: eight cells per LDS read, four reads in flight (one cell per trip was one LDS latency per
        // keypoint: most of the kernel's 33 us for one frame)
        int rank = 0;
        const unsigned cu = (unsigned)cell;
        const int4* c8 = (const int4*)cellOf;
        const int full = i >> 3;
#pragma unroll 4
        for (int j8 = 0; j8 < full; j8++) {
            const int4 v = c8[j8];
            rank += (int)(((unsigned)v.x & 0xffffu) == cu) + (int)(((unsigned)v.x >> 16) == cu) + (int)(((unsigned)v.y & 0xffffu) == cu) +
                    (int)(((unsigned)v.y >> 16) == cu) + (int)(((unsigned)v.z & 0xffffu) == cu) + (int)(((unsigned)v.z >> 16) == cu) +
                    (int)(((unsigned)v.w & 0xffffu) == cu) + (int)(((unsigned)v.w >> 16) == cu);
        }
        for (int j = full << 3; j < i; j++) rank += cellOf[j] == cell;
        gi[cnt[cell] + rank] = i;
    }
}

// Frame::ComputeStereoFromRGBD (reference src/Frame.cc:994-1015) with the depth conversion Tracking::GrabImageRGBD applies
// first (imDepth.convertTo(CV_32F, mDepthMapFactor), src/Tracking.cc:1003-1004): d = depth(int(v), int(u)) [* factor];
// d > 0: mvDepth = d, mvuRight = kpUn.x - mbf / d; else both -1.  One thread per keypoint slot.
struct RgbdParams {
    int capacity, rows, cols, isU16, scale;     // scale: convertTo runs (a 16-bit map always, a float map when factor != 1)
    long long stride, frame;                    // bytes
    float factor, mbf;
};
__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const Keypoint* __restrict__ kps, const Keypoint* __restrict__ kpsUn,
                                                           const int* __restrict__ nOut, const uint8_t* __restrict__ depth,
                                                           RgbdParams p, float* __restrict__ uRight, float* __restrict__ depthOut) {
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= p.capacity) return;
    float ur = -1.f, dz = -1.f;
    if (i < min(nOut[f], p.capacity)) {
        const Keypoint k = kps[(long long)f * p.capacity + i];
        const int v = min(max((int)k.y, 0), p.rows - 1), u = min(max((int)k.x, 0), p.cols - 1);   // Mat::at<float>(float, float) truncates
        const uint8_t* row = depth + (long long)f * p.frame + (long long)v * p.stride;
        float d = p.isU16 ? (float)((const unsigned short*)row)[u] : ((const float*)row)[u];
        if (p.scale) d = __fmul_rn(d, p.factor);
        if (d > 0.f) { dz = d; ur = __fsub_rn(kpsUn[(long long)f * p.capacity + i].x, __fdiv_rn(p.mbf, d)); }
    }
    uRight[(long long)f * p.capacity + i] = ur;
    depthOut[(long long)f * p.capacity + i] = dz;
}
void launchStereoFromRgbd(hipStream_t st, const Keypoint* kps, const Keypoint* kpsUn, const int* nOut, const uint8_t* depth,
                          const RgbdParams& p, float* uRight, float* depthOut, int nFrames) {
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((p.capacity + 255) / 256, nFrames), dim3(256), 0, st, kps, kpsUn, nOut, depth, p,
                       uRight, depthOut);
}

void launchFrameFinish(hipStream_t st, const Keypoint* kps, const int* nOut, const FrameFinishParams& p, Keypoint* kpsUn,
                       int* gridOff, int* gridIdx, int* nInside, int nFrames) {
    const size_t lds = (size_t)(kGridCells + 4) * sizeof(int) + (size_t)((p.capacity + 7) & ~7) * sizeof(short);
    hipLaunchKernelGGL(k_frame_finish, dim3(nFrames), dim3(1024), lds, st, kps, nOut, p, kpsUn, gridOff, gridIdx, nInside);
}

}  // namespace orbx
