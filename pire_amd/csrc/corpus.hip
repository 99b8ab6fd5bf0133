// The synthetic corpus generator, identical to oracle/corpus.c.  DESIGN.md section 3.3.

#include "device_common.h"

namespace pirehip {

// ------------------------------------------------------------------------------------------ corpus generator
// Device twin of oracle/corpus.c (same integer arithmetic; tests/test_corpus.py pins equality).

struct DevPlants {
	uint32_t nplants;
	uint32_t len[16];
	uint32_t atTail[16];
	uint8_t bytes[16][64];
};

__device__ __forceinline__ uint64_t Mix64(uint64_t z)
{
	z ^= z >> 30;
	z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27;
	z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return z;
}

__global__ __launch_bounds__(256) void CorpusFillKernel(uint8_t* out, uint64_t seed, uint64_t first, uint64_t count,
                                                        uint64_t len, uint64_t stride, DevPlants plants)
{
	const uint64_t wordsPerString = (len + 7) / 8;
	const uint64_t total = count * wordsPerString;
	for (uint64_t g = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; g < total; g += uint64_t(gridDim.x) * blockDim.x) {
		const uint64_t i = g / wordsPerString, w = g % wordsPerString;
		const uint64_t s = first + i;
		const uint64_t x = Mix64(seed + s * 0x9E3779B97F4A7C15ull + (w + 1) * 0xD1B54A32D192ED03ull);
		uint64_t poff = ~0ull, plen = 0;
		uint32_t pid = 0;
		if (plants.nplants) {
			const uint64_t slot = s % (plants.nplants + 1);
			if (slot != 0) {
				pid = uint32_t(slot - 1);
				const uint64_t wl = plants.len[pid];
				if (wl <= len) {
					plen = wl;
					poff = plants.atTail[pid] ? len - wl : Mix64(seed ^ s ^ 0xA5A5A5A5ull) % (len - wl + 1);
				}
			}
		}
		uint8_t* dst = out + i * stride + w * 8;
		for (uint32_t k = 0; k < 8 && w * 8 + k < len; ++k) {
			const uint64_t pos = w * 8 + k;
			uint8_t v = uint8_t(0x20 + ((((x >> (8 * k)) & 0xFF) * 95) >> 8));
			if (plen && pos >= poff && pos < poff + plen)
				v = plants.bytes[pid][pos - poff];
			dst[k] = v;
		}
	}
}


int LaunchCorpusFill(uint8_t* out, uint64_t seed, uint64_t first, uint64_t count, uint64_t len, uint64_t stride,
                     const void* plantsHost, hipStream_t stream)
{
	DevPlants pl;
	memset(&pl, 0, sizeof(pl));
	if (plantsHost)
		memcpy(&pl, plantsHost, sizeof(pl));   // same layout as corpus_plants (oracle/corpus.h)
	if (pl.nplants > 16) {
		SetError("corpus: too many plants");
		return PIRE_HIP_EINVAL;
	}
	if (count == 0 || len == 0)
		return PIRE_HIP_OK;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const uint64_t total = count * ((len + 7) / 8);
	const unsigned blocks = unsigned(std::min<uint64_t>((total + 255) / 256, uint64_t(cus) * 32));
	hipLaunchKernelGGL(CorpusFillKernel, dim3(blocks), dim3(256), 0, stream, out, seed, first, count, len, stride, pl);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "corpus kernel launch");
	return PIRE_HIP_OK;
}


}  // namespace pirehip
