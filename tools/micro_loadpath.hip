// Microbenchmark: input-path ceilings for "one string per lane" over fixed-length records.
// Not part of the product; used to choose the tile shape of the scan kernel.
//   mode 0: fully coalesced streaming read (16 B/lane, grid-stride)            -- chip ceiling
//   mode 1: LDS-DMA tiles: wave loads 64 strings x TB bytes (TB/16 lanes per string), XOR-swizzled,
//           each lane then ds_read_b128's its own string's TB bytes                -- the design
//   mode 2: per-lane strided: lane reads TB bytes of its own string with dwordx4   -- no LDS
// Usage: micro_loadpath [nstrings_log2=20] [len=4096] [reps=5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void fill_kernel(uint32_t* p, size_t nwords) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < nwords; i += stride) { uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; p[i] = x; }
}

__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ in, size_t nvec, uint32_t* out, int nt) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    if (nt) { for (; i < nvec; i += stride) acc ^= __builtin_nontemporal_load(&in[i]); }
    else    { for (; i < nvec; i += stride) acc ^= in[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int TB, int WAVES, int AUX>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const uint8_t* __restrict__ text, uint64_t nstr, uint32_t len, uint32_t* out) {
    extern __shared__ __attribute__((aligned(128))) uint8_t lds[];
    constexpr int CH = TB / 16;            // 16-B chunks per string per tile
    constexpr int SPI = 64 / CH;           // strings per DMA instruction
    constexpr int NI = 64 / SPI;           // DMA instructions per tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* stage = lds + wave * (64 * TB);
    const uint64_t ntasks = nstr / 64;
    const int ntiles = len / TB;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t task = (uint64_t)blockIdx.x * WAVES + wave; task < ntasks; task += (uint64_t)gridDim.x * WAVES) {
        const uint64_t s0 = task * 64;
        // per-lane source geometry for the DMA: string (j*SPI + lane/CH), chunk (lane%CH) ^ f(string)
        const int f_rd = (lane >> 1) & (CH - 1);
        const uint32_t rd_base = lane * TB + (f_rd << 4);
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int srow = j * SPI + lane / CH;
                const int c = (lane % CH) ^ ((srow >> 1) & (CH - 1));
                const uint8_t* g = text + (s0 + srow) * (uint64_t)len + (uint32_t)t * TB + c * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(stage + j * 1024), 16, 0, AUX);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                u32x4 v = *(const u32x4*)(stage + (rd_base ^ (k << 4)));
                acc ^= v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// Same, but the tile is pulled into VGPRs first and the next tile's DMA is issued before "processing"
// (what the real kernel does: single LDS buffer, register-resident working tile).
template <int TB, int WAVES, int AUX, int WORK>
__global__ __launch_bounds__(WAVES * 64) void dma_pipe_kernel(const uint8_t* __restrict__ text, uint64_t nstr, uint32_t len, uint32_t* out) {
    extern __shared__ __attribute__((aligned(128))) uint8_t lds[];
    constexpr int CH = TB / 16, SPI = 64 / CH, NI = 64 / SPI;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* stage = lds + wave * (64 * TB);
    const uint64_t ntasks = nstr / 64;
    const int ntiles = len / TB;
    uint32_t acc = 0;
    const int f_rd = (lane >> 1) & (CH - 1);
    const uint32_t rd_base = lane * TB + (f_rd << 4);
    for (uint64_t task = (uint64_t)blockIdx.x * WAVES + wave; task < ntasks; task += (uint64_t)gridDim.x * WAVES) {
        const uint64_t s0 = task * 64;
        auto issue = [&](int t) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int srow = j * SPI + lane / CH;
                const int c = (lane % CH) ^ ((srow >> 1) & (CH - 1));
                const uint8_t* g = text + (s0 + srow) * (uint64_t)len + (uint32_t)t * TB + c * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(stage + j * 1024), 16, 0, AUX);
            }
        };
        issue(0);
        for (int t = 0; t < ntiles; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            u32x4 r[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) r[k] = *(const u32x4*)(stage + (rd_base ^ (k << 4)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 1 < ntiles) issue(t + 1);
            // fake per-byte work: WORK dependent VALU ops per byte (latency chain like the DFA step)
#pragma unroll
            for (int k = 0; k < CH; ++k) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t x = r[k][w];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        acc = __builtin_amdgcn_perm(acc, x, 0x0c0c0400u | b);
                        if (WORK) acc = acc * 33u + 1u;
                    }
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int TB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void strided_kernel(const uint8_t* __restrict__ text, uint64_t nstr, uint32_t len, uint32_t* out) {
    constexpr int CH = TB / 16;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t ntasks = nstr / 64;
    const int ntiles = len / TB;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t task = (uint64_t)blockIdx.x * WAVES + wave; task < ntasks; task += (uint64_t)gridDim.x * WAVES) {
        const u32x4* p = (const u32x4*)(text + (task * 64 + lane) * (uint64_t)len);
        for (int t = 0; t < ntiles; ++t) {
            u32x4 r[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) r[k] = p[t * CH + k];
#pragma unroll
            for (int k = 0; k < CH; ++k) acc ^= r[k];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// The tiled kernel's pattern: 8 adjacent lanes cover one whole 128-byte line, instruction j reads, in every group of 8
// lanes, string (lane & ~7) + (j % 8), line j / 8 of the visit.  LINES = 1: 128 bytes of each of the 64 strings per
// visit (the kernel as it is); LINES = 2 / 4: 256 / 512 contiguous bytes of each string per visit.  DEPTH = visits in
// flight per wave (1: issue, wait, consume; 2: the next visit is requested before the current one is consumed).
template <int LINES, int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void group_kernel(const uint8_t* __restrict__ text, uint64_t nstr, uint32_t len, uint32_t* out)
{
    constexpr int NI = 8 * LINES;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t ntasks = nstr / 64;
    const int nvisits = len / (128 * LINES);
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH][NI];
    for (uint64_t task = (uint64_t)blockIdx.x * WAVES + wave; task < ntasks; task += (uint64_t)gridDim.x * WAVES) {
        const uint8_t* base = text + (task * 64 + (lane & ~7)) * (uint64_t)len + (lane & 7) * 16;
        auto issue = [&](int v, int slot) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const uint8_t* g = base + (uint64_t)(j % 8) * len + (uint32_t)v * (128 * LINES) + (j / 8) * 128;
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[slot][j]) : "v"(g));
            }
        };
        issue(0, 0);
        for (int v = 0; v < nvisits; ++v) {
            if (DEPTH == 2) {
                if (v + 1 < nvisits) { issue(v + 1, (v + 1) & 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI) : "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                u32x4 x = r[DEPTH == 2 ? (v & 1) : 0][j];
                asm volatile("" : "+v"(x));
                acc ^= x;
            }
            if (DEPTH == 1 && v + 1 < nvisits) issue(v + 1, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <typename F>
static double time_it(const char* name, size_t bytes, int reps, F&& launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    double med = ms[ms.size() / 2];
    printf("%-44s  median %8.3f ms  min %8.3f ms  %8.1f GB/s (median)  %8.1f GB/s (best)\n", name, med, ms[0], bytes / med / 1e6, bytes / ms[0] / 1e6);
    fflush(stdout);
    return med;
}

int main(int argc, char** argv) {
    const bool groupOnly = argc > 1 && !strcmp(argv[1], "group");
    if (groupOnly) { argc--; argv++; }
    int lg = argc > 1 ? atoi(argv[1]) : 20;
    uint32_t len = argc > 2 ? atoi(argv[2]) : 4096;
    int reps = argc > 3 ? atoi(argv[3]) : 5;
    uint64_t nstr = 1ull << lg;
    size_t bytes = nstr * len;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  LDS/block %zu  bytes %.2f GiB\n", prop.name, cus, prop.sharedMemPerBlock, bytes / 1073741824.0);
    uint8_t* text; CK(hipMalloc(&text, bytes));
    uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 8 * 1024 * 4));
    fill_kernel<<<cus * 8, 256>>>((uint32_t*)text, bytes / 4); CK(hipDeviceSynchronize());

#define GRP(L, W, D) time_it("group pattern, " #L " line(s) per string and visit, waves=" #W " depth=" #D, bytes, reps, [&] { group_kernel<L, W, D><<<cus, W * 64>>>(text, nstr, len, out); })
    if (groupOnly) {
        time_it("stream coalesced nt", bytes, reps, [&] { stream_kernel<<<cus * 8, 256>>>((const u32x4*)text, bytes / 16, out, 1); });
        for (int rep = 0; rep < 2; ++rep) {
            GRP(1, 16, 1); GRP(1, 12, 1); GRP(1, 8, 1);   // (more than one line per visit: not debugged, faults)
        }
        return 0;
    }
    time_it("stream coalesced 16B/lane, 256x8 blocks x256", bytes, reps, [&] { stream_kernel<<<cus * 8, 256>>>((const u32x4*)text, bytes / 16, out, 0); });
    time_it("stream coalesced nt", bytes, reps, [&] { stream_kernel<<<cus * 8, 256>>>((const u32x4*)text, bytes / 16, out, 1); });
    time_it("stream coalesced 1024thr x 256", bytes, reps, [&] { stream_kernel<<<cus, 1024>>>((const u32x4*)text, bytes / 16, out, 0); });

#define DMA(TB, W, AUX) do { \
        auto k = dma_kernel<TB, W, AUX>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W * 64 * TB)); \
        time_it("dma tile TB=" #TB " waves=" #W " aux=" #AUX, bytes, reps, [&] { k<<<cus, W * 64, W * 64 * TB>>>(text, nstr, len, out); }); } while (0)
    DMA(128, 16, 0); DMA(128, 16, 2); DMA(64, 16, 0); DMA(256, 8, 0); DMA(128, 8, 0); DMA(64, 8, 0);
    DMA(64, 16, 2);

#define PIPE(TB, W, AUX, WORK) do { \
        auto k = dma_pipe_kernel<TB, W, AUX, WORK>; \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W * 64 * TB)); \
        time_it("dma+vgpr pipe TB=" #TB " waves=" #W " aux=" #AUX " work=" #WORK, bytes, reps, [&] { k<<<cus, W * 64, W * 64 * TB>>>(text, nstr, len, out); }); } while (0)
    PIPE(128, 16, 0, 0); PIPE(128, 16, 0, 1); PIPE(64, 16, 0, 0); PIPE(128, 12, 0, 0); PIPE(128, 8, 0, 0);

#define STR(TB, W) time_it("strided per-lane TB=" #TB " waves=" #W, bytes, reps, [&] { strided_kernel<TB, W><<<cus, W * 64>>>(text, nstr, len, out); })
    STR(128, 16); STR(64, 16); STR(256, 16); STR(128, 8);
    // 2 blocks per CU for the strided kernel (no LDS limit)
    time_it("strided per-lane TB=128 waves=16 x2 blocks/CU", bytes, reps, [&] { strided_kernel<128, 16><<<cus * 2, 1024>>>(text, nstr, len, out); });
    return 0;
}
