// micro_lds: what the LDS of an MI355X CU can do for the dense-row walk of a GIVEN table on a GIVEN text (round 5,
// "set_d at 0.73 -- prove the floor or beat it").
//
//   micro_lds <trace.bin> <rows.bin> <waves> <steps> [reps]
//
// trace.bin: u16 [waves][steps / 16][64 lanes][16] -- the LDS byte address (dense id << 8 | text byte) of every lookup
// the tiled kernel's walk makes for 64 consecutive strings of the corpus (tools/micro_lds.py builds it from the table's
// host accessors and the corpus generator).  rows.bin: the table's dense rows (u8 [(hot + 1) * 256]).
// One block of 16 waves per CU, 256 blocks, the rows in LDS like the product kernel; every wave replays one trace.
//   mode A "independent": the 16 addresses of a chunk are issued back to back (no lookup waits for another): the rate the
//       LDS sustains under THIS pattern of bank conflicts -- the ceiling of any kernel with this layout;
//   mode B "dependent": state = rows[state << 8 | byte], one chain per lane, 16 waves: the product kernel's walk without
//       its text loads and transposes (its bytes come from the trace);
//   mode C "uniform": mode A with every lane reading address (lane * 4) -- no conflict at all, the instruction's own rate.
// Each with ds_read_u8 (what the kernels use) and with ds_read_b32 of the aligned word that holds the byte.
// Prints lane-lookups per clock and CU (clock from s_memrealtime is 100 MHz: the figure is quoted at the 2.4 GHz peak
// clock and, via the elapsed time, as an equivalent GB/s of text: 1 byte per lookup).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) uint8_t* LdsBytePtr;

__device__ __forceinline__ uint32_t Look(uint32_t addr) { return *reinterpret_cast<LdsBytePtr>(static_cast<uintptr_t>(addr)); }
typedef const __attribute__((address_space(3))) uint32_t* LdsWordPtr;
__device__ __forceinline__ uint32_t LookWord(uint32_t addr) { return *reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(addr & ~3u)); }

template <int MODE>
__global__ __launch_bounds__(1024) void Replay(const uint16_t* trace, const uint8_t* rows, uint32_t rowBytes, uint32_t waves,
                                               uint32_t steps, uint32_t reps, uint32_t* sink)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	for (uint32_t i = threadIdx.x; i < rowBytes / 16; i += blockDim.x)
		reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(rows)[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t which = (blockIdx.x * 16 + wave) % waves;
	const u32x4* mine = reinterpret_cast<const u32x4*>(trace + size_t(which) * steps * 64) + lane * 2;
	uint32_t acc = 0, st = 0;
	for (uint32_t r = 0; r < reps; ++r)
		for (uint32_t c = 0; c < steps / 16; ++c) {
			const u32x4 a = mine[size_t(c) * 128], b = mine[size_t(c) * 128 + 1];   // 16 u16 addresses of this lane
			const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
			if (MODE == 0) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					acc += Look(w[k] & 0xFFFFu);
					acc += Look(w[k] >> 16);
				}
			} else if (MODE == 1) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					st = Look((st << 8) | (w[k] & 0xFFu));           // the byte of the trace, the state of the chain
					st = Look((st << 8) | ((w[k] >> 16) & 0xFFu));
				}
			} else if (MODE == 2) {
#pragma unroll
				for (int k = 0; k < 16; ++k)
					acc += Look(lane * 4 + ((w[k >> 1] + k) & 0x300u));
			} else if (MODE == 3) {
#pragma unroll
				for (int k = 0; k < 16; ++k)
					acc += LookWord(lane * 4 + ((w[k >> 1] + k) & 0x300u));
			} else if (MODE == 6) {
				// the class table held by the wave itself (lane i: dword i of a 256-byte table), fetched with ds_bpermute_b32
				// (the LDS crossbar, no bank access) and cut out with v_bfe: an alternative to the ds_read_u8 of cls8[byte]
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const uint32_t b0 = w[k] & 0xFFu, b1 = (w[k] >> 16) & 0xFFu;
					const uint32_t d0 = uint32_t(__builtin_amdgcn_ds_bpermute(int(b0 & 0xFCu), int(lane * 0x01010101u)));
					const uint32_t d1 = uint32_t(__builtin_amdgcn_ds_bpermute(int(b1 & 0xFCu), int(lane * 0x01010101u)));
					acc += (d0 >> ((b0 & 3u) * 8)) & 0xFFu;
					acc += (d1 >> ((b1 & 3u) * 8)) & 0xFFu;
				}
			} else if (MODE == 7) {
				// ... against the ds_read_u8 of a 256-byte table at LDS address 0 with the same (text) bytes
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					acc += Look(w[k] & 0xFFu);
					acc += Look((w[k] >> 16) & 0xFFu);
				}
			} else if (MODE == 4) {
				// the real addresses as aligned DWORD reads (the byte would be cut out of the word by a v_bfe)
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					acc += LookWord(w[k] & 0xFFFFu);
					acc += LookWord(w[k] >> 16);
				}
			} else {
				// the dependent walk with dword reads: state = byte (addr & 3) of the word at (state << 8 | byte) & ~3
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					uint32_t a0 = (st << 8) | (w[k] & 0xFFu);
					st = (LookWord(a0) >> ((a0 & 3u) * 8)) & 0xFFu;
					uint32_t a1 = (st << 8) | ((w[k] >> 16) & 0xFFu);
					st = (LookWord(a1) >> ((a1 & 3u) * 8)) & 0xFFu;
				}
			}
		}
	if (acc == 0x12345u || st == 77u + reps)   // (never true for reps > 200; the compiler cannot know)
		sink[0] = acc + st;
}

static std::vector<uint8_t> ReadFile(const char* path)
{
	FILE* f = fopen(path, "rb");
	if (!f) {
		fprintf(stderr, "cannot open %s\n", path);
		exit(1);
	}
	fseek(f, 0, SEEK_END);
	const long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> v(n);
	if (fread(v.data(), 1, n, f) != size_t(n))
		exit(1);
	fclose(f);
	return v;
}

template <int MODE>
static void Run(const char* name, const uint16_t* dTrace, const uint8_t* dRows, uint32_t rowBytes, uint32_t waves, uint32_t steps,
                uint32_t reps, uint32_t* sink, int cus)
{
	hipFuncSetAttribute(reinterpret_cast<const void*>(Replay<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(rowBytes));
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int warm = 0; warm < 3; ++warm)
		hipLaunchKernelGGL(Replay<MODE>, dim3(cus), dim3(1024), rowBytes, 0, dTrace, dRows, rowBytes, waves, steps, reps, sink);
	float best = 1e9f;
	for (int t = 0; t < 5; ++t) {
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(Replay<MODE>, dim3(cus), dim3(1024), rowBytes, 0, dTrace, dRows, rowBytes, waves, steps, reps, sink);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	const double lookups = double(cus) * 1024.0 * steps * reps;
	printf("%-12s %8.3f ms  %7.1f GB/s of text equivalent  %5.2f lane-lookups per clock and CU at 2.4 GHz\n", name, best,
	       lookups / (best * 1e-3) / 1e9, lookups / (best * 1e-3) / 2.4e9 / cus);
}

int main(int argc, char** argv)
{
	if (argc < 5) {
		fprintf(stderr, "usage: micro_lds trace.bin rows.bin waves steps [reps]\n");
		return 2;
	}
	const std::vector<uint8_t> trace = ReadFile(argv[1]), rows = ReadFile(argv[2]);
	const uint32_t waves = atoi(argv[3]), steps = atoi(argv[4]), reps = argc > 5 ? atoi(argv[5]) : 64;
	if (trace.size() != size_t(waves) * steps * 64 * 2 || steps % 16 || rows.size() % 256) {
		fprintf(stderr, "trace / rows size mismatch\n");
		return 2;
	}
	int cus = 0;
	hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	uint16_t* dTrace;
	uint8_t* dRows;
	uint32_t* sink;
	hipMalloc(reinterpret_cast<void**>(&dTrace), trace.size());
	hipMalloc(reinterpret_cast<void**>(&dRows), rows.size() + 16);
	hipMalloc(reinterpret_cast<void**>(&sink), 64);
	hipMemcpy(dTrace, trace.data(), trace.size(), hipMemcpyHostToDevice);
	hipMemcpy(dRows, rows.data(), rows.size(), hipMemcpyHostToDevice);
	const uint32_t rowBytes = uint32_t((rows.size() + 15) / 16 * 16);
	printf("micro_lds: %d CUs, %u traces of %u steps x 64 lanes, %u bytes of rows in LDS, %u repetitions\n", cus, waves, steps, rowBytes, reps);
	Run<2>("uniform u8", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<3>("uniform b32", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<0>("indep. u8", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<4>("indep. b32", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<7>("cls u8", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<6>("cls bpermute", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<1>("chain u8", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	Run<5>("chain b32", dTrace, dRows, rowBytes, waves, steps, reps, sink, cus);
	return 0;
}
