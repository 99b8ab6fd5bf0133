// PCIe probe (not part of the product): what can host-pointer mode reach on this box?
//   pinned H2D, pageable H2D, hipHostRegister cost, multi-threaded memcpy into pinned staging, chunked pipeline.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pcopy(char* dst, const char* src, size_t n, int threads)
{
    std::vector<std::thread> th;
    size_t per = (n / threads + 4095) & ~size_t(4095);
    for (int t = 0; t < threads; ++t) {
        size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
        if (hi > lo) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& x : th) x.join();
}
int main()
{
    const size_t N = size_t(1) << 30;
    char* dev; CK(hipMalloc(&dev, N));
    char* pinned; CK(hipHostMalloc(&pinned, N, hipHostMallocDefault));
    char* pageable = static_cast<char*>(aligned_alloc(4096, N));
    memset(pinned, 1, N); memset(pageable, 2, N);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int r = 0; r < 3; ++r) {
        double t0 = now(); CK(hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double t1 = now(); printf("pinned   H2D 1 GiB: %.1f GB/s\n", N / (t1 - t0) / 1e9);
    }
    for (int r = 0; r < 3; ++r) {
        double t0 = now(); CK(hipMemcpy(dev, pageable, N, hipMemcpyHostToDevice));
        double t1 = now(); printf("pageable H2D 1 GiB: %.1f GB/s\n", N / (t1 - t0) / 1e9);
    }
    for (int r = 0; r < 2; ++r) {
        double t0 = now(); CK(hipHostRegister(pageable, N, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(dev, pageable, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        CK(hipHostUnregister(pageable)); double t3 = now();
        printf("hipHostRegister 1 GiB: %.1f ms, copy %.1f GB/s, unregister %.1f ms -> all-in %.1f GB/s\n", (t1 - t0) * 1e3,
               N / (t2 - t1) / 1e9, (t3 - t2) * 1e3, N / (t3 - t0) / 1e9);
    }
    for (int threads : {1, 2, 4, 8, 16}) {
        double t0 = now(); pcopy(pinned, pageable, N, threads); double t1 = now();
        printf("memcpy pageable -> pinned, %2d threads: %.1f GB/s\n", threads, N / (t1 - t0) / 1e9);
    }
    // chunked pipeline: worker threads stage chunk k+1 while chunk k is in flight
    for (size_t chunk : {size_t(8) << 20, size_t(32) << 20, size_t(128) << 20})
        for (int threads : {4, 8}) {
            const int slots = 3;
            hipEvent_t done[slots]; for (auto& e : done) CK(hipEventCreate(&e));
            double t0 = now();
            size_t k = 0;
            for (size_t off = 0; off < N; off += chunk, ++k) {
                const int sl = k % slots;
                if (k >= size_t(slots)) CK(hipEventSynchronize(done[sl]));
                const size_t len = std::min(chunk, N - off);
                pcopy(pinned + sl * chunk, pageable + off, len, threads);
                CK(hipMemcpyAsync(dev + off, pinned + sl * chunk, len, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(done[sl], s));
            }
            CK(hipStreamSynchronize(s));
            double t1 = now();
            printf("pipeline chunk %3zu MiB, %d copy threads, %d slots: %.1f GB/s\n", chunk >> 20, threads, slots, N / (t1 - t0) / 1e9);
        }
    {   // D2H of results: 5 B per 4 KiB string is nothing; just a sanity number
        double t0 = now(); CK(hipMemcpyAsync(pinned, dev, N / 64, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double t1 = now(); printf("pinned D2H 16 MiB: %.1f GB/s\n", N / 64 / (t1 - t0) / 1e9);
    }
    return 0;
}
