// orbx_stream.cpp — a C++ host program on the C ABI (no Python, no OpenCV): a camera stream fed frame batch by frame
// batch through orbx_extract_batch_device, the way a C++ SLAM front-end would use the library.
//
//   build:  hipcc -O2 -std=c++17 -Iinclude examples/orbx_stream.cpp -o orbx_stream -Lextractorb_amd -lorbx -Wl,-rpath,$PWD/extractorb_amd
//   run:    ./orbx_stream [frames_per_batch=64] [batches=20] [rows=480] [cols=640] [nfeatures=1000]
//
// Prints frames/s (device-resident input) and the keypoint count and a descriptor checksum of the first frame, which
// tests/test_examples.py compares with the CPU oracle.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_ORBX(x) do { int rc_ = (x); if (rc_ != ORBX_OK) { std::fprintf(stderr, "%s: %d %s\n", #x, rc_, orbx_last_error(h)); return 1; } } while (0)

// the stream generator of extractorb_amd/synth.py ("noise" variant): pix = splitmix64(key(seed, frame) + index) >> 56
static uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void noiseFrame(uint64_t frame, int rows, int cols, uint8_t* out) {
    const uint64_t key = 20261003ull * 0x100000001B3ull + frame * 0xD1B54A32D192ED03ull;
    for (size_t i = 0; i < (size_t)rows * cols; i++) out[i] = (uint8_t)(splitmix64(key + i) >> 56);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 64, batches = argc > 2 ? std::atoi(argv[2]) : 20;
    const int rows = argc > 3 ? std::atoi(argv[3]) : 480, cols = argc > 4 ? std::atoi(argv[4]) : 640;
    const int nfeatures = argc > 5 ? std::atoi(argv[5]) : 1000;
    orbx_handle* h = nullptr;
    int rc = orbx_create(&h, nfeatures, 1.2f, 8, 20, 7, cols, rows, B, -1);
    if (rc != ORBX_OK) { std::fprintf(stderr, "orbx_create: %d %s\n", rc, orbx_last_error(nullptr)); return 1; }
    const int cap = orbx_max_keypoints(h);

    std::vector<uint8_t> host((size_t)B * rows * cols);
    for (int f = 0; f < B; f++) noiseFrame((uint64_t)f, rows, cols, host.data() + (size_t)f * rows * cols);
    uint8_t *d_img = nullptr, *d_desc = nullptr;
    orbx_keypoint* d_kps = nullptr;
    int *d_n = nullptr, *d_mono = nullptr;
    CHECK_HIP(hipMalloc(&d_img, host.size()));
    CHECK_HIP(hipMalloc(&d_kps, sizeof(orbx_keypoint) * (size_t)B * cap));
    CHECK_HIP(hipMalloc(&d_desc, (size_t)B * cap * 32));
    CHECK_HIP(hipMalloc(&d_n, sizeof(int) * B));
    CHECK_HIP(hipMalloc(&d_mono, sizeof(int) * B));
    CHECK_HIP(hipMemcpy(d_img, host.data(), host.size(), hipMemcpyHostToDevice));

    auto run = [&]() { return orbx_extract_batch_device(h, B, d_img, rows, cols, cols, (ptrdiff_t)rows * cols, nullptr, d_kps, d_desc, cap, d_n, d_mono, nullptr, nullptr); };
    for (int i = 0; i < 3; i++) CHECK_ORBX(run());          // warm-up (first call installs the geometry tables)
    CHECK_ORBX(orbx_synchronize(h));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < batches; i++) CHECK_ORBX(run());
    CHECK_ORBX(orbx_synchronize(h));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<int> n(B);
    CHECK_HIP(hipMemcpy(n.data(), d_n, sizeof(int) * B, hipMemcpyDeviceToHost));
    std::vector<uint8_t> desc0((size_t)n[0] * 32);
    CHECK_HIP(hipMemcpy(desc0.data(), d_desc, desc0.size(), hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (size_t i = 0; i < desc0.size(); i++) sum = sum * 1099511628211ull + desc0[i];
    std::printf("frames_per_batch=%d batches=%d %dx%d nfeatures=%d\n", B, batches, cols, rows, nfeatures);
    std::printf("frames_per_sec=%.1f ms_per_batch=%.3f\n", (double)B * batches / sec, sec / batches * 1e3);
    std::printf("frame0_keypoints=%d frame0_descriptor_fnv=%llu\n", n[0], (unsigned long long)sum);
    orbx_destroy(h);
    (void)hipFree(d_img); (void)hipFree(d_kps); (void)hipFree(d_desc); (void)hipFree(d_n); (void)hipFree(d_mono);
    return 0;
}
