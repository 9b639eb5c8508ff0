// k_gray.hip — the step in front of the extractor: Tracking::GrabImageMonocular / GrabImageStereo convert colour input
// with cv::cvtColor(RGB2GRAY | BGR2GRAY | RGBA2GRAY | BGRA2GRAY) before the Frame constructor calls ExtractORB
// (reference src/Tracking.cc:915-941, 985-1001).
//
// OpenCV 3.4 8-bit colour -> gray (generic path): 14-bit fixed point, gray = (R*4899 + G*9617 + B*1868 + 8192) >> 14
// (coefficients sum to 16384, so the result never exceeds 255); alpha is ignored.
//
// HBM-bound by construction (3-4 bytes in, 1 byte out per pixel): a thread converts 4 adjacent pixels — three or four
// aligned dword loads when the source rows are dword-aligned, byte loads otherwise — and stores one dword.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbx {

constexpr size_t kPolluteBytes = 158 * 1024;      // one workgroup per CU (160 KB of LDS each)


struct GrayParams {
    int rows, cols, channels, redFirst, aligned;
    long long srcStride, srcFrame, dstStride, dstFrame;
};

__device__ __forceinline__ unsigned grayOf(unsigned c0, unsigned c1, unsigned c2, bool redFirst) {
    const unsigned r = redFirst ? c0 : c2, b = redFirst ? c2 : c0;
    return (r * 4899u + c1 * 9617u + b * 1868u + 8192u) >> 14;
}

// grid (ceil(cols/4 / 256), rows, frames)
__global__ __launch_bounds__(256) void k_gray(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, GrayParams p) {
    const int g4 = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    const int x0 = 4 * g4;
    if (x0 >= p.cols) return;
    const uint8_t* s = src + (long long)f * p.srcFrame + (long long)y * p.srcStride + (long long)x0 * p.channels;
    uint8_t* d = dst + (long long)f * p.dstFrame + (long long)y * p.dstStride + x0;
    const bool red = p.redFirst != 0;
    unsigned v[4];
    if (x0 + 4 <= p.cols && p.aligned) {
        if (p.channels == 3) {
            const unsigned w0 = ((const unsigned*)s)[0], w1 = ((const unsigned*)s)[1], w2 = ((const unsigned*)s)[2];
            v[0] = grayOf(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, red);
            v[1] = grayOf(w0 >> 24, w1 & 255, (w1 >> 8) & 255, red);
            v[2] = grayOf((w1 >> 16) & 255, w1 >> 24, w2 & 255, red);
            v[3] = grayOf((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24, red);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned w = ((const unsigned*)s)[j];
                v[j] = grayOf(w & 255, (w >> 8) & 255, (w >> 16) & 255, red);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = 0;
            if (x0 + j < p.cols) v[j] = grayOf(s[j * p.channels], s[j * p.channels + 1], s[j * p.channels + 2], red);
        }
    }
    if (x0 + 4 <= p.cols && ((p.dstStride | p.dstFrame | (long long)(uintptr_t)dst) & 3) == 0) {
        *(unsigned*)d = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x0 + j < p.cols) d[j] = (uint8_t)v[j];
    }
}

void launchGray(hipStream_t st, const uint8_t* src, uint8_t* dst, const GrayParams& p, int nFrames) {
    hipLaunchKernelGGL(k_gray, dim3((p.cols / 4 + 256) / 256, p.rows, nFrames), dim3(256), 0, st, src, dst, p);
}

// Test aid (ORBX_LDS_POLLUTE=<byte>): LDS is not cleared between workgroups, so a kernel that reads LDS it has not written sees whatever
// the previous workgroup on that CU left — usually harmless leftovers of the same kernel.  This kernel takes a CU's whole LDS per workgroup
// and fills it with a byte pattern; the pipeline launches it in front of every kernel so that such a read becomes a deterministic failure.
__global__ __launch_bounds__(256) void k_lds_pollute(unsigned pattern, unsigned* __restrict__ sink) {
    extern __shared__ unsigned pl[];
    const int words = (int)(kPolluteBytes / 4);
    for (int i = threadIdx.x; i < words; i += 256) pl[i] = pattern;
    __syncthreads();
    if (pl[(threadIdx.x * 97 + blockIdx.x) % words] != pattern) sink[0] = 1;      // (keeps the stores alive)
}
void launchLdsPollute(hipStream_t st, int numCUs, int byte, unsigned* sink) {
    const unsigned b = (unsigned)byte & 255u;
    hipLaunchKernelGGL(k_lds_pollute, dim3(4 * numCUs), dim3(256), kPolluteBytes, st, b * 0x01010101u, sink);
}

}  // namespace orbx
