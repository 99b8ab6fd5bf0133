// Byte feeder for one-string-per-lane exact kernels: hands the bytes of [ptr, end) to `step` in order, reading
// line-aligned tiles -- up to 8 x 16 bytes per lane in one go, from the current position to the end of its 128-byte
// line -- with aligned vector loads instead of byte loads.  Only 16-byte blocks that contain at least one byte of
// the string are read.  `step(byte)` returns false to stop early (the prefix searches do).
//
// Measured (round 1): it lifts the prefix kernel, which used to issue one global_load_ubyte per step, from 175 to
// 220 GB/s.  The generic / half-final / counting / slow kernels already read 16 bytes at a time and are bound by
// their per-byte instruction count, not by memory: with this feeder they were 0-30 % SLOWER (more control flow per
// byte), so they keep their simple loops.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pirehip {

typedef uint32_t walk_u32x4 __attribute__((ext_vector_type(4)));

// Block feeder: hands the 16-byte blocks that hold [ptr, end) to `block(v, skip, count)` in order -- bytes
// [skip, skip + count) of v belong to the string (only the first block can have skip > 0, only the last count < 16 -
// skip) -- reading line-aligned tiles as described above.  `block` returns false to stop early.
template <class Block>
__device__ __forceinline__ void WalkBlocks(const uint8_t* ptr, const uint8_t* end, Block&& block)
{
	if (ptr >= end)
		return;
	uint64_t base = reinterpret_cast<uint64_t>(ptr) & ~uint64_t(15);   // current 16-byte block
	uint32_t skip = uint32_t(reinterpret_cast<uint64_t>(ptr) - base);   // bytes of it that precede the string
	uint64_t remaining = uint64_t(end - ptr);
	while (remaining) {
		// blocks from `base` to the end of its 128-byte line, or to the block holding the last byte
		const uint64_t span = skip + remaining;
		const uint32_t toLine = (128u - uint32_t(base & 127)) >> 4;
		const uint64_t need = (span + 15) >> 4;
		const uint32_t nch = need < toLine ? uint32_t(need) : toLine;
		walk_u32x4 t[8];
#pragma unroll
		for (int k = 0; k < 8; ++k)
			if (uint32_t(k) < nch)
				t[k] = reinterpret_cast<const walk_u32x4*>(base)[k];
		uint64_t left = span;   // bytes from the start of block k to the end of the string
#pragma unroll 1
		for (uint32_t k = 0; k < nch; ++k) {
			const walk_u32x4 v = t[0];
#pragma unroll
			for (int j = 0; j < 7; ++j)   // static indices only: the tile is a shift register
				t[j] = t[j + 1];
			const uint32_t hi = left < 16 ? uint32_t(left) : 16u;
			const uint32_t sk = k == 0 ? skip : 0u;
			if (!block(v, sk, hi - sk))
				return;
			left -= hi;
		}
		const uint64_t consumed = uint64_t(nch) * 16 - skip;
		remaining = consumed < remaining ? remaining - consumed : 0;
		base += uint64_t(nch) * 16;
		skip = 0;
	}
}

// bytes [skip, skip + count) of a block, one at a time; false as soon as `step` says stop
template <class Step>
__device__ __forceinline__ bool BlockBytes(walk_u32x4 v, uint32_t skip, uint32_t count, Step&& step)
{
	for (uint32_t i = 0; i < skip; ++i) {
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
#pragma unroll 1
	for (uint32_t i = 0; i < count; ++i) {
		if (!step(v.x & 0xFFu))
			return false;
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return true;
}

template <class Step>
__device__ __forceinline__ void WalkBytes(const uint8_t* ptr, const uint8_t* end, Step&& step)
{
	WalkBlocks(ptr, end, [&](walk_u32x4 v, uint32_t skip, uint32_t count) { return BlockBytes(v, skip, count, step); });
}

}  // namespace pirehip
