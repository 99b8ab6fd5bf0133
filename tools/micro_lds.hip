// Microbenchmark: throughput of the dependent LDS gather chain  st = table[(st<<8)|byte]  that the scan kernel is
// made of.  Not part of the product.  Answers: is ds_read_u8 slower than ds_read_b32?  how much do bank conflicts
// cost at realistic state spreads?  what does an unaligned ds_read_b32 return?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(3))) uint8_t* LdsB;
typedef const __attribute__((address_space(3))) uint32_t* LdsW;

__device__ __forceinline__ uint32_t rd8(uint32_t a) { return *reinterpret_cast<LdsB>(static_cast<uintptr_t>(a)); }
__device__ __forceinline__ uint32_t rd32(uint32_t a) { return *reinterpret_cast<LdsW>(static_cast<uintptr_t>(a)); }

// MODE 0: ds_read_u8.  MODE 1: ds_read_b32 at aligned address + v_alignbyte.  MODE 2: ds_read_b32 at the byte address.
// MODE 3: ds_read_u8 but TWO independent chains per lane (ILP 2).  MODE 4: chain of VALU only (no LDS) for reference.
template <int MODE>
__device__ __forceinline__ uint32_t step(uint32_t st, uint32_t x, uint32_t sel)
{
    const uint32_t a = __builtin_amdgcn_perm(st, x, sel);
    if (MODE == 0 || MODE == 3) return rd8(a);
    if (MODE == 1) { const uint32_t d = rd32(a & ~3u); return __builtin_amdgcn_alignbyte(d, d, a); }
    if (MODE == 2) return rd32(a);
    return a * 2654435761u >> 24;
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void chain_kernel(const uint8_t* __restrict__ table, uint32_t* out, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < 65536 / 16; i += blockDim.x)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(table)[i];
    __syncthreads();
    uint32_t r[16];
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; w |= (0x20u + (((h >> 8) & 0xFF) * 95 >> 8)) << (8 * b); }
        r[k] = w;
    }
    uint32_t st = 0, st2 = 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            st = step<MODE>(st, r[k], 0x0c0c0400u);
            if (MODE == 3) st2 = step<MODE>(st2, r[15 - k], 0x0c0c0403u);
            st = step<MODE>(st, r[k], 0x0c0c0401u);
            if (MODE == 3) st2 = step<MODE>(st2, r[15 - k], 0x0c0c0402u);
            st = step<MODE>(st, r[k], 0x0c0c0402u);
            if (MODE == 3) st2 = step<MODE>(st2, r[15 - k], 0x0c0c0401u);
            st = step<MODE>(st, r[k], 0x0c0c0403u);
            if (MODE == 3) st2 = step<MODE>(st2, r[15 - k], 0x0c0c0400u);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = st + st2;
}

__global__ void unaligned_probe(uint32_t* out)
{
    __shared__ uint32_t w[4];
    if (threadIdx.x == 0) { w[0] = 0x03020100; w[1] = 0x07060504; w[2] = 0x0b0a0908; w[3] = 0x0f0e0d0c; }
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(LdsW)w;
    if (threadIdx.x < 8) out[threadIdx.x] = rd32(base + threadIdx.x);
}

// table generators: spread = how many distinct states the chain wanders over
static std::vector<uint8_t> make_table(int spread, double stay0)
{
    std::vector<uint8_t> t(65536);
    uint32_t h = 12345;
    for (int s = 0; s < 256; ++s)
        for (int b = 0; b < 256; ++b) {
            h = h * 1664525u + 1013904223u;
            uint32_t r = h >> 8;
            uint8_t nx;
            if ((r & 0xFFFF) < stay0 * 65536.0) nx = 0;               // fall back to the hub state with prob stay0
            else nx = (uint8_t)((r >> 16) % spread);
            t[s * 256 + b] = nx;
        }
    return t;
}


// MODE u16: rows of 256 u16 entries, an entry is the LDS byte address of the next row; the step is
//   addr = v_dot4_u32_u8(x, 2 << 8k, row) = row + 2 * byte_k ;  row = ds_read_u16(addr)
typedef const __attribute__((address_space(3))) uint16_t* LdsH;
__device__ __forceinline__ uint32_t rd16(uint32_t a) { return *reinterpret_cast<LdsH>(static_cast<uintptr_t>(a)); }

template <int WAVES, int ILP>
__global__ __launch_bounds__(WAVES * 64) void chain16_kernel(const uint16_t* __restrict__ table, uint32_t words, uint32_t* out, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x)
        reinterpret_cast<uint32_t*>(lds)[i] = reinterpret_cast<const uint32_t*>(table)[i];
    __syncthreads();
    uint32_t r[16];
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; w |= (0x20u + (((h >> 8) & 0xFF) * 95 >> 8)) << (8 * b); }
        r[k] = w;
    }
    uint32_t st = 0, st2 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            st = rd16(__builtin_amdgcn_udot4(r[k], 0x00000002u, st, false));
            if (ILP == 2) st2 = rd16(__builtin_amdgcn_udot4(r[15 - k], 0x02000000u, st2, false));
            st = rd16(__builtin_amdgcn_udot4(r[k], 0x00000200u, st, false));
            if (ILP == 2) st2 = rd16(__builtin_amdgcn_udot4(r[15 - k], 0x00020000u, st2, false));
            st = rd16(__builtin_amdgcn_udot4(r[k], 0x00020000u, st, false));
            if (ILP == 2) st2 = rd16(__builtin_amdgcn_udot4(r[15 - k], 0x00000200u, st2, false));
            st = rd16(__builtin_amdgcn_udot4(r[k], 0x02000000u, st, false));
            if (ILP == 2) st2 = rd16(__builtin_amdgcn_udot4(r[15 - k], 0x00000002u, st2, false));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = st + st2;
}

// The same u16 rows with the address built by ONE full-rate VALU in the dependent chain: v_lshl_add_u32(byte, 1, row);
// the byte is extracted off the chain (v_bfe_u32, one more VALU per byte that depends on the text only).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void chain16s_kernel(const uint16_t* __restrict__ table, uint32_t words, uint32_t* out, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x)
        reinterpret_cast<uint32_t*>(lds)[i] = reinterpret_cast<const uint32_t*>(table)[i];
    __syncthreads();
    uint32_t r[16];
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t w = 0;
        for (int b = 0; b < 4; ++b) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; w |= (0x20u + (((h >> 8) & 0xFF) * 95 >> 8)) << (8 * b); }
        r[k] = w;
    }
    uint32_t st = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint32_t x = r[k];
            asm volatile("" : "+v"(x));   // keep the four extractions inside the loop (the real kernel pays them per byte)
            const uint32_t b0 = __builtin_amdgcn_ubfe(x, 0, 8), b1 = __builtin_amdgcn_ubfe(x, 8, 8), b2 = __builtin_amdgcn_ubfe(x, 16, 8), b3 = x >> 24;
            st = rd16((b0 << 1) + st);
            st = rd16((b1 << 1) + st);
            st = rd16((b2 << 1) + st);
            st = rd16((b3 << 1) + st);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = st;
}

static std::vector<uint16_t> make_table16(const std::vector<uint8_t>& t8, int rows, uint32_t pitch)
{
    std::vector<uint16_t> t(size_t(rows) * pitch / 2 + 8, 0);
    for (int s = 0; s < rows; ++s)
        for (int b = 0; b < 256; ++b) {
            uint32_t nx = t8[s * 256 + b];
            if (nx >= (uint32_t)rows) nx = 0;
            t[(size_t(s) * pitch) / 2 + b] = uint16_t(nx * pitch);
        }
    return t;
}

template <int WAVES, int ILP>
static void run16(const char* name, const uint16_t* dtab, uint32_t bytes, uint32_t* out, int cus, int iters)
{
    auto k = chain16_kernel<WAVES, ILP>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<cus, WAVES * 64, bytes>>>(dtab, bytes / 4, out, 4, 1); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); k<<<cus, WAVES * 64, bytes>>>(dtab, bytes / 4, out, iters, r); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double steps = (double)cus * WAVES * 64 * iters * 64.0 * ILP;
    const double tps = steps / (ms[2] * 1e-3);
    printf("%-52s %8.3f ms  %7.2f Tsteps/s  %6.2f steps/ns/CU\n", name, ms[2], tps / 1e12, tps / 1e9 / cus);
    fflush(stdout);
}

template <int WAVES>
static void run16s(const char* name, const uint16_t* dtab, uint32_t bytes, uint32_t* out, int cus, int iters)
{
    auto k = chain16s_kernel<WAVES>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<<<cus, WAVES * 64, bytes>>>(dtab, bytes / 4, out, 4, 1); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); k<<<cus, WAVES * 64, bytes>>>(dtab, bytes / 4, out, iters, r); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double steps = (double)cus * WAVES * 64 * iters * 64.0;
    const double tps = steps / (ms[2] * 1e-3);
    printf("%-52s %8.3f ms  %7.2f Tsteps/s  %6.2f steps/ns/CU\n", name, ms[2], tps / 1e12, tps / 1e9 / cus);
    fflush(stdout);
}

template <int MODE, int WAVES>
static void run(const char* name, const uint8_t* dtab, uint32_t* out, int cus, int blocksPerCu, int iters)
{
    auto k = chain_kernel<MODE, WAVES>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = cus * blocksPerCu;
    k<<<blocks, WAVES * 64, 65536>>>(dtab, out, 4, 1); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); k<<<blocks, WAVES * 64, 65536>>>(dtab, out, iters, r); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double chains = (MODE == 3 ? 2.0 : 1.0);
    const double steps = (double)blocks * WAVES * 64 * iters * 64.0 * chains;
    const double tps = steps / (ms[2] * 1e-3);
    printf("%-52s %8.3f ms  %7.2f Tsteps/s  %6.2f steps/ns/CU\n", name, ms[2], tps / 1e12, tps / 1e9 / cus);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 2 * 1024 * 4));
    uint32_t probe[8];
    unaligned_probe<<<1, 64>>>(out); CK(hipMemcpy(probe, out, sizeof(probe), hipMemcpyDeviceToHost));
    printf("unaligned ds_read_b32 at byte offsets 0..7:"); for (int i = 0; i < 8; ++i) printf(" %08x", probe[i]); printf("\n");
    uint8_t* dtab; CK(hipMalloc(&dtab, 65536));
    struct { const char* n; int spread; double stay0; } tabs[] = {
        {"all lanes state 0 (pure broadcast rows)", 1, 0.0},
        {"realistic: 76% hub state, 16 others", 16, 0.76},
        {"spread 16 uniform", 16, 0.0},
        {"spread 255 uniform (worst case)", 255, 0.0},
    };
    const int iters = 400;
    if (argc > 1 && !strcmp(argv[1], "shift")) {
        // u8 rows + v_perm (the library's step) against u16 rows + v_lshl_add_u32 at several pitches, 256 rows, 16 waves
        for (auto& tb : tabs) {
            auto t = make_table(tb.spread, tb.stay0);
            CK(hipMemcpy(dtab, t.data(), 65536, hipMemcpyHostToDevice));
            printf("--- table: %s\n", tb.n);
            run<0, 16>("u8 + v_perm, pitch 256, 16 waves/CU", dtab, out, cus, 1, iters);
            for (uint32_t pitch : {512u, 516u, 520u, 528u, 544u}) {
                auto t16 = make_table16(t, 256, pitch);
                uint16_t* d16;
                CK(hipMalloc(&d16, t16.size() * 2));
                CK(hipMemcpy(d16, t16.data(), t16.size() * 2, hipMemcpyHostToDevice));
                char nm[96];
                snprintf(nm, sizeof nm, "u16 + v_lshl_add, pitch %u, 16 waves/CU", pitch);
                run16s<16>(nm, d16, uint32_t(t16.size() * 2 + 15) & ~15u, out, cus, iters);
                if (pitch == 516u || pitch == 512u) {
                    snprintf(nm, sizeof nm, "u16 + v_dot4,     pitch %u, 16 waves/CU", pitch);
                    run16<16, 1>(nm, d16, uint32_t(t16.size() * 2 + 15) & ~15u, out, cus, iters);
                }
                CK(hipFree(d16));
            }
        }
        return 0;
    }
    for (auto& tb : tabs) {
        auto t = make_table(tb.spread, tb.stay0);
        CK(hipMemcpy(dtab, t.data(), 65536, hipMemcpyHostToDevice));
        printf("--- table: %s\n", tb.n);
        run<0, 16>("u8   16 waves/CU (1 block)", dtab, out, cus, 1, iters);
        run<0, 16>("u8   32 waves/CU (2 blocks)", dtab, out, cus, 2, iters);
        run<1, 16>("b32 aligned+alignbyte 16 waves/CU", dtab, out, cus, 1, iters);
        run<1, 16>("b32 aligned+alignbyte 32 waves/CU", dtab, out, cus, 2, iters);
        run<3, 16>("u8 ILP2 16 waves/CU", dtab, out, cus, 1, iters);
        run<3, 16>("u8 ILP2 32 waves/CU", dtab, out, cus, 2, iters);
        {
            uint16_t* d16; 
            for (uint32_t pitch : {512u, 544u, 576u, 608u}) {
                auto t16 = make_table16(t, 112, pitch);
                CK(hipMalloc(&d16, t16.size() * 2));
                CK(hipMemcpy(d16, t16.data(), t16.size() * 2, hipMemcpyHostToDevice));
                char nm[96];
                snprintf(nm, sizeof nm, "u16+dot4 pitch %u 16 waves/CU", pitch);
                run16<16, 1>(nm, d16, uint32_t(t16.size() * 2) & ~15u, out, cus, iters);
                snprintf(nm, sizeof nm, "u16+dot4 pitch %u 12 waves/CU ILP2", pitch);
                run16<12, 2>(nm, d16, uint32_t(t16.size() * 2) & ~15u, out, cus, iters);
                snprintf(nm, sizeof nm, "u16+dot4 pitch %u 16 waves/CU ILP2", pitch);
                run16<16, 2>(nm, d16, uint32_t(t16.size() * 2) & ~15u, out, cus, iters);
                snprintf(nm, sizeof nm, "u16+dot4 pitch %u 8 waves/CU", pitch);
                run16<8, 1>(nm, d16, uint32_t(t16.size() * 2) & ~15u, out, cus, iters);
                CK(hipFree(d16));
            }
        }
        run<0, 12>("u8   12 waves/CU (1 block)", dtab, out, cus, 1, iters);
        run<3, 12>("u8 ILP2 12 waves/CU", dtab, out, cus, 1, iters);
        run<3, 8>("u8 ILP2 8 waves/CU", dtab, out, cus, 1, iters);
        run<0, 8>("u8   8 waves/CU (1 block)", dtab, out, cus, 1, iters);
        run<0, 4>("u8   4 waves/CU (1 block)", dtab, out, cus, 1, iters);
    }
    run<4, 16>("VALU-only chain 16 waves/CU", dtab, out, cus, 1, iters);
    return 0;
}
