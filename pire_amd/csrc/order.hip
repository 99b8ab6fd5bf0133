// Strings in the order of their lengths, for the kernels that give a string to a lane for the whole of it.
//
// A wave of such a kernel is done when the LONGEST of its 64 strings is: on log lines of 64..1023 bytes the lanes idle
// 46 % of the time (the longest of 64 is ~1010 bytes, the mean 544; measured on the counting kernel: 941 GB/s on such a
// batch, 1 488 when every string is 544 bytes long -- profiles/r03_length_order.log).  The ragged kernel solves this by
// taking lanes off strings (a lane fetches the next string when its own ends); kernels whose per-byte work has state in
// registers and an action behind every step (the counting scanners) cannot follow it there without
// paying more than they win -- string switches are rare per lane but happen in every iteration of a 64-lane wave -- so
// they get the other remedy: a wave takes 64 strings of (nearly) the SAME length.  This file builds the permutation:
// a counting sort of the strings by length class on the device, three small launches on the caller's stream
// (histogram per block of strings, one-block scan, scatter), no host round trip.
//
// Length classes: 32-byte steps below 1 KiB, quarter octaves above (a class spans at most 32 bytes or 19 %); longest
// class first, so that what runs at the end of the launch -- when CUs go idle one by one -- is the short strings.
// (Tried: every run of 4 096 consecutive strings sorted on its own -- one launch, and a wave's strings from one stretch of
// the text -- 719 GB/s against 1 042 for the one order over the whole batch, 948 without: the blocks of a grid-stride
// loop then meet the same rank of every run, i.e. some only long strings.  profiles/r03_length_order.log.)
// Results do not depend on the order (every string is walked by one lane from its first byte to its last, as before).
//
// The kernels take string k of the order in a SERPENTINE over their passes (OrderedIndex below): a lane that took the
// t-th longest string of one pass takes the t-th shortest of the next, so that the lanes' totals are equal too (with a
// plain grid-stride loop and two strings per lane the first lanes got 1 024 + 544 bytes, the last 544 + 64).

#include <hip/hip_runtime.h>

#include <algorithm>

#include "internal.h"

namespace pirehip {

namespace {

constexpr uint32_t kOrderClasses = 64;
constexpr uint32_t kOrderThreads = 1024;
constexpr uint32_t kOrderMaxBlocks = 256;   // x 64 classes x 4 bytes = the scan kernel's 64 KB of LDS

__device__ __forceinline__ uint32_t LengthClass(uint64_t len)
{
	uint32_t c;
	if (len < 1024) {
		c = uint32_t(len >> 5);                                      // 0..31
	} else {
		const uint32_t lg = 63u - uint32_t(__clzll((long long)len));  // >= 10
		const uint32_t frac = uint32_t(len >> (lg - 2)) & 3u;        // the two bits below the leading one
		c = 32u + (lg - 10u) * 4u + frac;
		c = c < kOrderClasses - 1 ? c : kOrderClasses - 1;
	}
	return kOrderClasses - 1 - c;   // longest first
}

// hist[cls * blocks + block] = strings of the class in the block's range
__global__ __launch_bounds__(kOrderThreads) void OrderHistKernel(const uint64_t* offsets, uint64_t n, uint64_t perBlock, uint32_t* hist)
{
	__shared__ uint32_t local[kOrderClasses];
	if (threadIdx.x < kOrderClasses)
		local[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t lo = uint64_t(blockIdx.x) * perBlock, hi = lo + perBlock < n ? lo + perBlock : n;
	for (uint64_t s = lo + threadIdx.x; s < hi; s += kOrderThreads)
		atomicAdd(&local[LengthClass(offsets[s + 1] - offsets[s])], 1u);
	__syncthreads();
	if (threadIdx.x < kOrderClasses)
		hist[threadIdx.x * gridDim.x + blockIdx.x] = local[threadIdx.x];
}

// exclusive scan of hist in place (class-major: all blocks of class 0, then class 1, ...): one block, the whole array
// in LDS (at most kOrderMaxBlocks x 64 entries = 64 KB), every thread scans its own run of entries, a scan over the
// threads' sums in between (first version: entries read and written straight from memory, Hillis-Steele with two
// barriers a step: 20 us; this one 4)
__global__ __launch_bounds__(kOrderThreads) void OrderScanKernel(uint32_t* hist, uint32_t entries)
{
	extern __shared__ uint32_t all[];
	__shared__ uint32_t waveSum[kOrderThreads / 64];
	// padded: a thread's run starts 17 dwords after its neighbour's, not 16 (which would put 64 lanes on 4 banks)
	auto at = [](uint32_t i) { return i + (i >> 4); };
	for (uint32_t i = threadIdx.x; i < entries; i += kOrderThreads)
		all[at(i)] = hist[i];
	__syncthreads();
	const uint32_t per = (entries + kOrderThreads - 1) / kOrderThreads;
	const uint32_t lo = threadIdx.x * per, hi = lo + per < entries ? lo + per : entries;
	uint32_t sum = 0;
	for (uint32_t i = lo; i < hi; ++i)
		sum += all[at(i)];
	// inclusive scan of `sum` over the block: inside the wave by DPP-free shuffles, then over the 16 wave totals
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t incl = sum;
	for (uint32_t d = 1; d < 64; d <<= 1) {
		const uint32_t v = uint32_t(__shfl_up(int(incl), int(d), 64));
		if (lane >= d)
			incl += v;
	}
	if (lane == 63)
		waveSum[wave] = incl;
	__syncthreads();
	uint32_t before = 0;
	for (uint32_t w = 0; w < wave; ++w)
		before += waveSum[w];
	uint32_t run = before + incl - sum;
	for (uint32_t i = lo; i < hi; ++i) {
		const uint32_t v = all[at(i)];
		all[at(i)] = run;
		run += v;
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < entries; i += kOrderThreads)
		hist[i] = all[at(i)];
}

__global__ __launch_bounds__(kOrderThreads) void OrderScatterKernel(const uint64_t* offsets, uint64_t n, uint64_t perBlock,
                                                                   const uint32_t* base, uint32_t* perm)
{
	__shared__ uint32_t next[kOrderClasses];
	if (threadIdx.x < kOrderClasses)
		next[threadIdx.x] = base[threadIdx.x * gridDim.x + blockIdx.x];
	__syncthreads();
	const uint64_t lo = uint64_t(blockIdx.x) * perBlock, hi = lo + perBlock < n ? lo + perBlock : n;
	for (uint64_t s = lo + threadIdx.x; s < hi; s += kOrderThreads)
		perm[atomicAdd(&next[LengthClass(offsets[s + 1] - offsets[s])], 1u)] = uint32_t(s);
}

uint32_t OrderBlocks(uint64_t n)
{
	return uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(kOrderMaxBlocks, (n + 4095) / 4096)));
}

}  // namespace

bool LengthOrderWanted(uint64_t n)
{
	// below a few lane-fills of the chip the launches cost more than the idle lanes; indices are 32 bits
	return n >= 32768 && n < (1ull << 32) && !GetConfig().no_length_order;
}

size_t LengthOrderScratchBytes(uint64_t n)
{
	return ((size_t(n) * 4 + 255) & ~size_t(255)) + size_t(OrderBlocks(n)) * kOrderClasses * 4;
}

int BuildLengthOrder(const uint64_t* offsets, uint64_t n, void* scratch, hipStream_t stream, const uint32_t** perm, bool* serpentine)
{
	uint32_t* p = static_cast<uint32_t*>(scratch);
	*serpentine = true;
	uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(scratch) + ((size_t(n) * 4 + 255) & ~size_t(255)));
	const uint32_t blocks = OrderBlocks(n);
	const uint64_t perBlock = (n + blocks - 1) / blocks;
	hipLaunchKernelGGL(OrderHistKernel, dim3(blocks), dim3(kOrderThreads), 0, stream, offsets, n, perBlock, hist);
	const hipError_t le = SetDynamicLds(reinterpret_cast<const void*>(OrderScanKernel), kOrderMaxBlocks * kOrderClasses * 4 * 17 / 16 + 64);
	if (le != hipSuccess)
		return HipFail(le, "hipFuncSetAttribute(LDS)");
	hipLaunchKernelGGL(OrderScanKernel, dim3(1), dim3(kOrderThreads), blocks * kOrderClasses * 4 * 17 / 16 + 64, stream, hist, blocks * kOrderClasses);
	hipLaunchKernelGGL(OrderScatterKernel, dim3(blocks), dim3(kOrderThreads), 0, stream, offsets, n, perBlock, hist, p);
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "length order launch");
	*perm = p;
	return PIRE_HIP_OK;
}

}  // namespace pirehip
