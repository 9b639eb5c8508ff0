// orbx_multi_gpu.cpp — ONE C++ process driving every visible MI355X through the C ABI (no Python, no torch, no RCCL): the host side
// SURVEY.md §8e describes — "one host thread per GPU, each with its own handle/stream", frames of a batched stream sharded in contiguous
// blocks, no data-path collective, and the only exchange a gather of the finished result slabs into device 0's memory
// (hipMemcpyPeerAsync over xGMI; staged through the host by the runtime where peer access is off).
//
//   build:  hipcc -O2 -std=c++17 -pthread -Iinclude examples/orbx_multi_gpu.cpp -o orbx_multi_gpu -Lextractorb_amd -lorbx -Wl,-rpath,$PWD/extractorb_amd
//   run:    ./orbx_multi_gpu [total_frames=512] [steps=20] [rows=480] [cols=640] [nfeatures=1000] [devices=all]
//
// Device d extracts frames [lo_d, hi_d) of the stream (the rule of extractorb_amd/sharding.py: shard_range) into a result slab
// [keypoints | descriptors | n | mono] (sharding.py: slab_layout) and copies the slab into its place on device 0.  Two slabs per device:
// while step k's slab travels, step k + 1 computes into the other one; a slab is only overwritten after its previous copy has finished
// (an event the handle's stream waits for).  A step of the whole node = every device's step; the timed region is bracketed by thread
// barriers + device syncs, and the rate is total frames x steps / the wall time of the slowest device.
//
// Prints frames/s, every device's own ms per step, and - read back from the GATHERED copy on device 0 - the keypoint count and a
// descriptor checksum of each device's first frame, which tests/test_examples.py compares with the CPU oracle on that frame.
#include <hip/hip_runtime.h>
#include <pthread.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "orbx.h"

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

// extractorb_amd/sharding.py: shard_range - contiguous blocks that differ by at most one frame
static void shardRange(int nFrames, int rank, int world, int& lo, int& hi) {
    const int base = nFrames / world, extra = nFrames % world;
    lo = rank * base + (rank < extra ? rank : extra);
    hi = lo + base + (rank < extra ? 1 : 0);
}
// extractorb_amd/sharding.py: slab_layout - [keypoints | descriptors | n | mono], 256-byte padded
struct Slab { size_t k, d, n, m, bytes; };
static Slab slabLayout(int frames, int cap) {
    Slab s;
    s.k = 0;
    s.d = s.k + (size_t)frames * cap * sizeof(orbx_keypoint);
    s.n = s.d + (size_t)frames * cap * 32;
    s.m = s.n + 4 * (size_t)frames;
    s.bytes = (s.m + 4 * (size_t)frames + 255) / 256 * 256;
    return s;
}

struct Shared {
    int D, total, steps, rows, cols, nfeatures, cap;
    std::vector<int> dev;                  // HIP device index of rank d
    std::vector<uint8_t*> gathered[2];     // on device dev[0]: rank d's slab of buffer k (allocated by rank 0)
    pthread_barrier_t bar;
    std::vector<std::string> err;
    std::vector<double> msPerStep;
    double wallSec = 0;
};

#define T_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { S.err[r] = std::string(#x) + ": " + hipGetErrorString(e_); failed = true; } } while (0)
#define T_ORBX(x) do { int rc_ = (x); if (rc_ != ORBX_OK) { S.err[r] = std::string(#x) + ": " + std::to_string(rc_) + " " + orbx_last_error(h); failed = true; } } while (0)

static void worker(Shared& S, int r) {
    bool failed = false;
    orbx_handle* h = nullptr;
    int lo, hi;
    shardRange(S.total, r, S.D, lo, hi);
    const int B = hi - lo;
    const Slab L = slabLayout(B, S.cap);
    uint8_t *d_img = nullptr, *slab[2] = {nullptr, nullptr};
    hipStream_t copyStream = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    T_HIP(hipSetDevice(S.dev[r]));
    if (!failed && B > 0) {
        int rc = orbx_create(&h, S.nfeatures, 1.2f, 8, 20, 7, S.cols, S.rows, B, S.dev[r]);
        if (rc != ORBX_OK) { S.err[r] = std::string("orbx_create: ") + std::to_string(rc) + " " + orbx_last_error(nullptr); failed = true; }
    }
    if (!failed && B > 0) {
        std::vector<uint8_t> host((size_t)B * S.rows * S.cols);
        for (int f = 0; f < B; f++) noiseFrame((uint64_t)(lo + f), S.rows, S.cols, host.data() + (size_t)f * S.rows * S.cols);
        T_HIP(hipMalloc(&d_img, host.size()));
        for (int k = 0; k < 2 && !failed; k++) {
            T_HIP(hipMalloc(&slab[k], L.bytes));
            T_HIP(hipMemset(slab[k], 0, L.bytes));
            T_HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            T_HIP(hipEventCreateWithFlags(&copied[k], hipEventDisableTiming));
        }
        T_HIP(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
        if (!failed) T_HIP(hipMemcpy(d_img, host.data(), host.size(), hipMemcpyHostToDevice));
        if (r != 0 && !failed) {      // peer access to the gather target where the topology offers it; without it the peer copy is staged by the runtime
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, S.dev[r], S.dev[0]) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(S.dev[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    if (r == 0 && !failed)      // the gather target: one place per rank and buffer, on rank 0's device
        for (int k = 0; k < 2; k++)
            for (int d = 0; d < S.D && !failed; d++) {
                int l2, h2;
                shardRange(S.total, d, S.D, l2, h2);
                T_HIP(hipMalloc(&S.gathered[k][d], slabLayout(h2 - l2, S.cap).bytes));
            }
    pthread_barrier_wait(&S.bar);      // every rank's setup (and rank 0's gather buffers) in place
    bool anyFailed = false;
    for (const std::string& e : S.err) anyFailed = anyFailed || !e.empty();
    hipStream_t st = h ? (hipStream_t)orbx_get_stream(h) : nullptr;
    int stepNo = 0;
    auto step = [&]() {
        if (B == 0) return;
        const int k = stepNo++ & 1;
        T_HIP(hipStreamWaitEvent(st, copied[k], 0));      // (a never-recorded event does not block) the slab's previous copy has left it
        T_ORBX(orbx_extract_batch_device(h, B, d_img, S.rows, S.cols, S.cols, (ptrdiff_t)S.rows * S.cols, nullptr, (orbx_keypoint*)(slab[k] + L.k),
                                         slab[k] + L.d, S.cap, (int*)(slab[k] + L.n), (int*)(slab[k] + L.m), nullptr, nullptr));
        T_HIP(hipEventRecord(done[k], st));
        T_HIP(hipStreamWaitEvent(copyStream, done[k], 0));
        if (S.dev[r] == S.dev[0]) T_HIP(hipMemcpyAsync(S.gathered[k][r], slab[k], L.bytes, hipMemcpyDeviceToDevice, copyStream));
        else T_HIP(hipMemcpyPeerAsync(S.gathered[k][r], S.dev[0], slab[k], S.dev[r], L.bytes, copyStream));
        T_HIP(hipEventRecord(copied[k], copyStream));
    };
    auto fence = [&]() { if (B > 0) { T_HIP(hipStreamSynchronize(st)); T_HIP(hipStreamSynchronize(copyStream)); } };
    if (!anyFailed) {
        for (int i = 0; i < 3 && !failed; i++) step();      // warm-up: the first call installs the geometry tables, the first peer copy maps the target
        fence();
    }
    pthread_barrier_wait(&S.bar);
    const auto t0 = std::chrono::steady_clock::now();
    if (!anyFailed) {
        for (int i = 0; i < S.steps && !failed; i++) step();
        fence();
    }
    const double mine = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    S.msPerStep[r] = mine / (S.steps > 0 ? S.steps : 1) * 1e3;
    pthread_barrier_wait(&S.bar);
    if (r == 0) S.wallSec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pthread_barrier_wait(&S.bar);      // rank 0 reads the gathered slabs (main) before anybody frees anything
    pthread_barrier_wait(&S.bar);
    if (h) orbx_destroy(h);
    (void)hipFree(d_img);
    for (int k = 0; k < 2; k++) { (void)hipFree(slab[k]); if (done[k]) (void)hipEventDestroy(done[k]); if (copied[k]) (void)hipEventDestroy(copied[k]); }
    if (copyStream) (void)hipStreamDestroy(copyStream);
}

int main(int argc, char** argv) {
    Shared S;
    S.total = argc > 1 ? std::atoi(argv[1]) : 512; S.steps = argc > 2 ? std::atoi(argv[2]) : 20;
    S.rows = argc > 3 ? std::atoi(argv[3]) : 480; S.cols = argc > 4 ? std::atoi(argv[4]) : 640;
    S.nfeatures = argc > 5 ? std::atoi(argv[5]) : 1000;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    S.D = argc > 6 && std::atoi(argv[6]) > 0 ? std::atoi(argv[6]) : ndev;
    if (S.total < 1 || S.steps < 1) { std::fprintf(stderr, "total_frames and steps must be positive\n"); return 1; }
    // (more ranks than devices - a rehearsal of the N > 1 control flow on a one-GPU box: ranks share devices round-robin, each with its own handle)
    for (int d = 0; d < S.D; d++) S.dev.push_back(d % ndev);
    S.cap = S.nfeatures + 3 * 8;      // the reference's bound: every level keeps at most quota + 3 keypoints
    S.err.assign(S.D, std::string());
    S.msPerStep.assign(S.D, 0.0);
    for (int k = 0; k < 2; k++) S.gathered[k].assign(S.D, nullptr);
    pthread_barrier_init(&S.bar, nullptr, (unsigned)S.D + 1);
    std::vector<std::thread> th;
    for (int d = 0; d < S.D; d++) th.emplace_back(worker, std::ref(S), d);
    pthread_barrier_wait(&S.bar);      // setup
    pthread_barrier_wait(&S.bar);      // start of the timed region
    pthread_barrier_wait(&S.bar);      // end of the timed region
    pthread_barrier_wait(&S.bar);      // wall time written
    int rc = 0;
    for (int d = 0; d < S.D; d++) if (!S.err[d].empty()) { std::fprintf(stderr, "rank %d (device %d): %s\n", d, S.dev[d], S.err[d].c_str()); rc = 1; }
    if (rc == 0) {
        std::printf("devices=%d ranks=%d total_frames=%d steps=%d %dx%d nfeatures=%d\n", ndev, S.D, S.total, S.steps, S.cols, S.rows, S.nfeatures);
        std::printf("frames_per_sec=%.1f ms_per_step=%.4f\n", (double)S.total * S.steps / S.wallSec, S.wallSec / S.steps * 1e3);
        std::printf("per_rank_ms_per_step=");
        for (int d = 0; d < S.D; d++) std::printf("%s%.4f", d ? "," : "", S.msPerStep[d]);
        std::printf("\n");
        // what ARRIVED on device 0: the first frame of every rank's slab of the last step
        (void)hipSetDevice(S.dev[0]);
        const int kl = (3 + S.steps - 1) & 1;
        for (int d = 0; d < S.D; d++) {
            int lo, hi;
            shardRange(S.total, d, S.D, lo, hi);
            if (hi == lo) continue;
            const Slab L = slabLayout(hi - lo, S.cap);
            std::vector<uint8_t> host(L.bytes);
            if (hipMemcpy(host.data(), S.gathered[kl][d], L.bytes, hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "read-back of rank %d failed\n", d); rc = 1; break; }
            const int n0 = *(const int*)(host.data() + L.n), m0 = *(const int*)(host.data() + L.m);
            uint64_t sum = 0;
            if (n0 >= 0 && n0 <= S.cap) for (size_t i = 0; i < (size_t)n0 * 32; i++) sum = sum * 1099511628211ull + host[L.d + i];
            std::printf("rank=%d device=%d frames=%d first_frame=%d keypoints=%d mono=%d descriptor_fnv=%llu\n", d, S.dev[d], hi - lo, lo, n0, m0, (unsigned long long)sum);
        }
    }
    pthread_barrier_wait(&S.bar);      // the workers may free now
    for (std::thread& t : th) t.join();
    (void)hipSetDevice(S.dev[0]);
    for (int k = 0; k < 2; k++) for (uint8_t* p : S.gathered[k]) (void)hipFree(p);
    pthread_barrier_destroy(&S.bar);
    return rc;
}
