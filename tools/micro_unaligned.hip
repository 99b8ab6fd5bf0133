// Does the alignment of a lane's 16-byte loads matter?  The ragged kernel reads a string's first window from the
// string's first byte, whatever its alignment (8 x global_load_dwordx4 per lane, by groups of 8 lanes); this asks what
// the same loads cost when the window starts are rounded down to 4, 16 or 128 bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mu tools/micro_unaligned.hip && /tmp/mu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// consecutive lanes read consecutive "strings" of `pitch` bytes (URLs: ~110), one window of 128 bytes each; a wave
// moves on by 64 strings per iteration; `mask` rounds the window start down
template <bool GROUPS>
__global__ __launch_bounds__(1024) void Windows(const uint8_t* text, uint64_t size, uint32_t pitch, uint64_t mask, uint32_t iters,
                                                uint32_t* out)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t wave = uint64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const uint64_t waves = uint64_t(gridDim.x) * (blockDim.x >> 6);
	uint32_t acc = 0;
	for (uint32_t it = 0; it < iters; ++it) {
		const uint64_t s = (uint64_t(it) * waves + wave) * 64 + lane;
		uint64_t a = (s * pitch) % (size - 4096);
		a &= mask;
		u32x4 r[8];
		if (GROUPS) {
			// instruction j loads, in every group of 8 lanes, the window of lane 8g + j: lane 8g + c reads bytes [16c, 16c + 16)
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint64_t sj = (uint64_t(it) * waves + wave) * 64 + (lane & ~7u) + j;
				uint64_t aj = (sj * pitch) % (size - 4096);
				aj &= mask;
				r[j] = *reinterpret_cast<const u32x4*>(text + aj + (lane & 7u) * 16);
			}
		} else {
#pragma unroll
			for (int j = 0; j < 8; ++j)
				r[j] = *reinterpret_cast<const u32x4*>(text + a + j * 16);
		}
#pragma unroll
		for (int j = 0; j < 8; ++j)
			acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
	}
	if (acc == 0x12345678u)
		out[0] = acc;
}

int main()
{
	const uint64_t size = 1ull << 30;
	uint8_t* text;
	uint32_t* out;
	if (hipMalloc(&text, size) != hipSuccess || hipMalloc(&out, 64) != hipSuccess)
		return 1;
	(void)hipMemset(text, 1, size);
	hipEvent_t a, b;
	(void)hipEventCreate(&a);
	(void)hipEventCreate(&b);
	const uint32_t iters = 16;
	for (int groups = 0; groups < 2; ++groups)
		for (uint32_t pitch : {110u, 128u, 544u})
			for (uint64_t al : {1ull, 4ull, 16ull, 128ull}) {
				float best = 1e9f;
				for (int rep = 0; rep < 5; ++rep) {
					(void)hipEventRecord(a, nullptr);
					if (groups)
						hipLaunchKernelGGL(Windows<true>, dim3(256), dim3(1024), 0, nullptr, text, size, pitch, ~(al - 1), iters, out);
					else
						hipLaunchKernelGGL(Windows<false>, dim3(256), dim3(1024), 0, nullptr, text, size, pitch, ~(al - 1), iters, out);
					(void)hipEventRecord(b, nullptr);
					(void)hipEventSynchronize(b);
					float ms;
					(void)hipEventElapsedTime(&ms, a, b);
					best = ms < best ? ms : best;
				}
				const double req = 256.0 * 1024 * iters * 128;
				printf("%s loads, strings %3u B apart, window starts rounded down to %3llu: %.3f ms, %.0f GB/s requested\n",
				       groups ? "group" : "lane ", pitch, (unsigned long long)al, best, req / best / 1e6);
			}
	return 0;
}
