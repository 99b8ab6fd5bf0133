// HIP kernels of the scan path for gfx950 (MI355X / CDNA4).  No MFMA: the path is a byte-gather DFA walk.
//
// What is computed (per input string, one string per lane):
//     st = start;  for each byte b:  st = Next(st, b);   [Begin/End marks around it]
// which is Pire::Run / Pire::Step of /root/reference/pire/run.h:50-57, 271-275 over the table of
// /root/reference/pire/scanners/multi.h:163-192 (Next = row[letters[ch]]).
//
// Device table layout (built in table.cpp, described in DESIGN.md section 3):
//   * states are renumbered "hot first" (perm ids); the reference's ids come back through origOfPerm[]
//   * hot rows: up to 255 states have a DENSE row of 256 u8 entries in LDS, indexed directly by the input
//     byte (the byte->letter-class translation of multi.h:163-166 is folded in).  One LDS gather per byte:
//         addr = v_perm_b32(st, word, sel)   = (st << 8) | byte_k(word)
//         st   = ds_read_u8(addr)
//     An entry is the next hot id, or the trap id H ("left the hot set"); row H maps every byte to H.
//   * everything else: nextPerm[perm * letters + cls[byte]] (u32) in HBM/L2 -- the exact, slow step.
// A lane that leaves the hot set is re-walked exactly through the slow step for the 16-byte chunk in which it
// trapped, so results never depend on which rows are hot.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <cstdio>

#include "internal.h"
#include "walk.h"

namespace pirehip {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t kDebugNoRefill = 1u << 30;   // internal, never set through the C ABI
constexpr uint32_t kDebugNoStep = 1u << 29;     // internal, never set through the C ABI
constexpr uint32_t kDebugNoColdCount = 1u << 28;
constexpr uint32_t kDebugNoHist = 1u << 27;
constexpr uint32_t kDebugNoPartial = 1u << 26;   // ragged kernel timing experiments only (results are wrong)
constexpr uint32_t kDebugNoFinish = 1u << 25;
constexpr uint32_t kDebugNoTrap = 1u << 24;

// Cooperative load of the LDS-resident part of the table.
__device__ inline void LoadTableToLds(const ScanParams& p, uint8_t* lds, const LdsLayout& L)
{
	const uint32_t tid = threadIdx.x, nthr = blockDim.x;
	// dense rows: 256-byte rows in HBM, `pitch`-byte rows in LDS (dword copies: 260 is only 4-byte aligned)
	const uint32_t* src = reinterpret_cast<const uint32_t*>(p.hotRows);
	uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
	const uint32_t pitchDw = L.pitch / 4;
	for (uint32_t i = tid; i < (p.hot + 1) * 64; i += nthr)
		dst[(i >> 6) * pitchDw + (i & 63)] = src[i];
	for (uint32_t i = tid; i < 256 / 4; i += nthr)
		reinterpret_cast<uint32_t*>(lds + L.flagsOff)[i] = reinterpret_cast<const uint32_t*>(p.hotFlags)[i];
	for (uint32_t i = tid; i < 264 / 2; i += nthr)
		reinterpret_cast<uint32_t*>(lds + L.clsOff)[i] = reinterpret_cast<const uint32_t*>(p.cls)[i];
	if (p.outCounts)
		for (uint32_t i = tid; i < p.regexps + 2; i += nthr)
			reinterpret_cast<uint32_t*>(lds + L.countsOff)[i] = 0;
	for (uint32_t i = tid; i < 256; i += nthr)
		reinterpret_cast<uint32_t*>(lds + L.histOff)[i] = 0;
	if (p.compact) {
		for (uint32_t i = tid; i < L.compactBytes / 16; i += nthr)
			reinterpret_cast<u32x4*>(lds + L.compactOff)[i] = reinterpret_cast<const u32x4*>(p.compactRows)[i];
		for (uint32_t i = tid; i < 256; i += nthr)
			lds[L.cls8Off + i] = uint8_t(2 * p.cls[i]);
	}
	__syncthreads();
}

// The exact step for any state: multi.h:169-192 on the perm-numbered table.
__device__ __forceinline__ uint32_t SlowStep(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                             uint32_t st, uint32_t byte)
{
	if (st < p.hot) {
		const uint32_t e = lds[st * L.pitch + byte];
		if (e != p.hot)
			return e;
	}
	const uint32_t c = reinterpret_cast<const uint16_t*>(lds + L.clsOff)[byte];
	return p.nextPerm[size_t(st) * p.letters + c];
}

__device__ __forceinline__ uint32_t SlowStepWord(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                                 uint32_t st, uint32_t w)
{
	st = SlowStep(p, lds, L, st, w & 0xFF);
	st = SlowStep(p, lds, L, st, (w >> 8) & 0xFF);
	st = SlowStep(p, lds, L, st, (w >> 16) & 0xFF);
	st = SlowStep(p, lds, L, st, w >> 24);
	return st;
}

// Start state of string s (perm id): Initialize() or the caller's resume state, then Begin() if asked.
__device__ __forceinline__ uint32_t StartState(const ScanParams& p, uint64_t s)
{
	if (!p.initIdx)
		return p.startPerm;   // host folded Initialize()+Begin() into one id
	uint32_t st = p.permOfOrig[p.initIdx[s]];
	if (p.flags & PIRE_HIP_RUN_BEGIN)
		st = p.nextPerm[size_t(st) * p.letters + p.beginCls];
	return st;
}

// End(), outputs and block-local match counters for one finished string.  One 16-byte record load replaces the
// chain  nextPerm[EndMark] -> flags -> origOfPerm -> acceptMask  of dependent lookups.
__device__ __forceinline__ void Finish(const ScanParams& p, uint8_t* lds, const LdsLayout& L, uint64_t s,
                                       bool active, uint32_t st)
{
	const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
	const u32x4 raw = *reinterpret_cast<const u32x4*>(&recs[st]);
	const uint32_t orig = raw.x, endPerm = raw.y & 0x0FFFFFFFu, fl = raw.y >> 28;
	const uint64_t mask = (uint64_t(raw.w) << 32) | raw.z;
	if (active) {
		if (p.outIdx)
			p.outIdx[s] = orig;
		if (p.outFinal)
			p.outFinal[s] = fl & kFinal;
	}
	if (p.outCounts) {
		uint32_t* cnt = reinterpret_cast<uint32_t*>(lds + L.countsOff);
		const int lane = threadIdx.x & 63;
		const unsigned long long finals = __ballot(active && (fl & kFinal));
		const unsigned long long actives = __ballot(active);
		if (lane == 0) {
			atomicAdd(&cnt[0], (uint32_t)__popcll(finals));
			atomicAdd(&cnt[1], (uint32_t)__popcll(actives));
		}
		if (p.acceptMaskPerm) {
			const uint64_t m = active ? mask : 0;
			for (uint32_t r = 0; r < p.regexps; ++r) {
				const unsigned long long b = __ballot((m >> r) & 1);
				if (lane == 0 && b)
					atomicAdd(&cnt[2 + r], (uint32_t)__popcll(b));
			}
		} else if (active) {
			for (uint64_t k = p.acceptOffPerm[endPerm]; k < p.acceptOffPerm[endPerm + 1]; ++k)
				atomicAdd(&cnt[2 + p.acceptIds[k]], 1u);
		}
	}
}

__device__ inline void FlushCounts(const ScanParams& p, uint8_t* lds, const LdsLayout& L)
{
	__syncthreads();
	const uint32_t* hist = reinterpret_cast<const uint32_t*>(lds + L.histOff);
	for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
		if (hist[i])
			atomicAdd(&p.visitHot[i], hist[i]);
	if (!p.outCounts)
		return;
	const uint32_t* cnt = reinterpret_cast<const uint32_t*>(lds + L.countsOff);
	for (uint32_t i = threadIdx.x; i < p.regexps + 2; i += blockDim.x)
		if (cnt[i])
			atomicAdd(&p.outCounts[i], (unsigned long long)cnt[i]);
}

}  // namespace

// ------------------------------------------------------------------------------------------ generic kernel
// Any offsets, any alignment, any length (including 0).  One string per lane, exact step per byte.

__global__ __launch_bounds__(256) void ScanGenericKernel(ScanParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, kRotPitch, CompactBytes(p));
	LoadTableToLds(p, lds, L);

	const uint64_t nrounds = (p.n + 63) / 64;
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t task = uint64_t(blockIdx.x) * wavesPerBlock + (threadIdx.x >> 6); task < nrounds;
	     task += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t s = task * 64 + lane;
		const bool active = s < p.n;
		uint32_t st = 0;
		if (active) {
			st = StartState(p, s);
			uint64_t b, e;
			if (p.offsets) {
				b = p.offsets[s];
				e = p.offsets[s + 1];
			} else {
				b = s * p.stride;
				e = b + p.len;
			}
			const uint8_t* ptr = p.text + b;
			const uint8_t* end = p.text + e;
			while (ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
				st = SlowStep(p, lds, L, st, *ptr);
				++ptr;
			}
			for (; ptr + 16 <= end; ptr += 16) {
				const u32x4 v = *reinterpret_cast<const u32x4*>(ptr);
				st = SlowStepWord(p, lds, L, st, v.x);
				st = SlowStepWord(p, lds, L, st, v.y);
				st = SlowStepWord(p, lds, L, st, v.z);
				st = SlowStepWord(p, lds, L, st, v.w);
			}
			for (; ptr < end; ++ptr)
				st = SlowStep(p, lds, L, st, *ptr);
		}
		Finish(p, lds, L, s, active, st);
	}
	FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ tiled kernel
// Fixed-length records, 16-byte aligned.  Each lane streams ITS OWN string straight from HBM in 128-byte
// tiles (8 x global_load_dwordx4 = exactly one cache line per lane per tile), double-buffered in VGPRs, and
// walks the tile out of registers with one LDS gather per byte.  No LDS staging: measured on MI355X
// (profiles/micro_loadpath_r01.log) the per-lane line-sized access streams at the same ~6.2 TB/s as a fully
// coalesced read, so all of the LDS is left for the table.

// ---- tile loads -------------------------------------------------------------------------------------------
// A tile is 128 bytes (one cache line) of each of the wave's 64 strings.  It is fetched with 8 x
// global_load_dwordx4 in which EIGHT ADJACENT LANES COVER ONE WHOLE LINE: instruction j, lane l reads
//     chunk (l & 7) of string  s0 + (l & ~7) + j        (16 bytes)
// so every instruction touches 8 full lines instead of 64 partial ones.  Measured on MI355X
// (profiles/r01_pmc_summary_strided_16w_nbuf3.txt): with one-line-per-lane loads the L1 (TCP) tag pipeline was the
// binding unit -- TA busy 75 %, TA stalled by TC 56 %, 0.63 lane-accesses/clk/CU -- and `nt` could not be used
// because each line was touched by 8 separate instructions.  With whole-line instructions the same bytes cost
// 1/8 of the L1 accesses and stream with `nt`.
// After the loads, lane 8g+k holds in register j chunk k of string 8g+j; an 8x8 transpose across each group of 8
// lanes (TransposeTile, DPP only, no LDS) leaves lane 8g+j with chunks 0..7 of its own string in registers 0..7.
//
// The loads are issued from inline asm and waited for with hand-counted s_waitcnt vmcnt(N).  Reason (measured,
// DESIGN.md section 6): hipcc's own wait insertion turns every loop-carried prefetch into `s_waitcnt vmcnt(0)` at
// the tile boundary, which collapses an N-deep register pipeline to depth 1.  Counting is safe with foreign VMEM
// ops in the queue: loads return in order among themselves, so "at most 8*k outstanding" implies every load issued
// before the last k tiles has landed; extra compiler-issued ops only make the wait stricter.
// "+v": the tile registers are updated IN PLACE, so the compiler has no reason to copy a slot that is in flight.
template <bool NT>
__device__ __forceinline__ void IssueTile(u32x4 (&r)[8], uint32_t voff, uint64_t tileBase, uint64_t stride)
{
	const uint64_t b0 = tileBase, b1 = b0 + stride, b2 = b1 + stride, b3 = b2 + stride, b4 = b3 + stride,
	               b5 = b4 + stride, b6 = b5 + stride, b7 = b6 + stride;
	if (NT)
		asm volatile(
			"global_load_dwordx4 %0, %8, %9 nt\n\t"
			"global_load_dwordx4 %1, %8, %10 nt\n\t"
			"global_load_dwordx4 %2, %8, %11 nt\n\t"
			"global_load_dwordx4 %3, %8, %12 nt\n\t"
			"global_load_dwordx4 %4, %8, %13 nt\n\t"
			"global_load_dwordx4 %5, %8, %14 nt\n\t"
			"global_load_dwordx4 %6, %8, %15 nt\n\t"
			"global_load_dwordx4 %7, %8, %16 nt"
			: "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
			: "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(b6), "s"(b7));
	else
		asm volatile(
			"global_load_dwordx4 %0, %8, %9\n\t"
			"global_load_dwordx4 %1, %8, %10\n\t"
			"global_load_dwordx4 %2, %8, %11\n\t"
			"global_load_dwordx4 %3, %8, %12\n\t"
			"global_load_dwordx4 %4, %8, %13\n\t"
			"global_load_dwordx4 %5, %8, %14\n\t"
			"global_load_dwordx4 %6, %8, %15\n\t"
			"global_load_dwordx4 %7, %8, %16"
			: "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
			: "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(b6), "s"(b7));
}

// Wait until at most TILES_BEHIND tiles issued after `r` are still in flight; names r so nothing reads it earlier.
template <int TILES_BEHIND>
__device__ __forceinline__ void WaitTile(u32x4 (&r)[8])
{
	asm volatile("s_waitcnt vmcnt(%8)"
	             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
	             : "n"(TILES_BEHIND * 8));
}

// One butterfly stage of the 8x8 transpose: exchange (register bit D) with (lane bit D) for the pair x = reg k,
// y = reg k|D:   x'[l] = (l & D) ? y[l ^ D] : x[l],    y'[l] = (l & D) ? y[l] : x[l ^ D].
// D = 4: two bank-masked DPP moves (row_shr:4 into banks 1,3; row_shl:4 into banks 0,2).
__device__ __forceinline__ void Butterfly4(uint32_t& x, uint32_t& y)
{
	const uint32_t nx = __builtin_amdgcn_update_dpp(x, y, 0x114, 0xF, 0xA, false);
	const uint32_t ny = __builtin_amdgcn_update_dpp(y, x, 0x104, 0xF, 0x5, false);
	x = nx;
	y = ny;
}

// D = 1 or 2: the partner lane sits in the same quad; one FUSED v_cndmask_b32_dpp per output (hipcc does not form
// it from v_mov_dpp + v_cndmask: the select mask would have to be inverted for half of them).  `lo` = lanes whose
// bit D is clear, `hi` = lanes whose bit D is set (64-bit wave masks).  Four pairs per statement, one column.
#define PIRE_BFLY_QUAD(PERM)                                                                                           \
	asm volatile("s_nop 1\n\t"                                                                                     \
	             "s_mov_b64 vcc, %16\n\t"                                                                           \
	             "v_cndmask_b32_dpp %0, %9, %8, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                       \
	             "v_cndmask_b32_dpp %2, %11, %10, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
	             "v_cndmask_b32_dpp %4, %13, %12, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
	             "v_cndmask_b32_dpp %6, %15, %14, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
	             "s_mov_b64 vcc, %17\n\t"                                                                           \
	             "v_cndmask_b32_dpp %1, %8, %9, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                       \
	             "v_cndmask_b32_dpp %3, %10, %11, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
	             "v_cndmask_b32_dpp %5, %12, %13, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
	             "v_cndmask_b32_dpp %7, %14, %15, vcc " PERM " row_mask:0xf bank_mask:0xf"                          \
	             : "=&v"(nx0), "=&v"(ny0), "=&v"(nx1), "=&v"(ny1), "=&v"(nx2), "=&v"(ny2), "=&v"(nx3), "=&v"(ny3)    \
	             : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "s"(lo), "s"(hi)          \
	             : "vcc")

// v_cndmask: D = vcc ? src1 : src0, DPP permutes src0.  With vcc = lo:  x' = lo ? x : perm(y);  with vcc = hi:
// y' = hi ? y : perm(x).
template <int D>
__device__ __forceinline__ void ButterflyQuad4(uint32_t& x0, uint32_t& y0, uint32_t& x1, uint32_t& y1, uint32_t& x2,
                                               uint32_t& y2, uint32_t& x3, uint32_t& y3, uint64_t lo, uint64_t hi)
{
	uint32_t nx0, ny0, nx1, ny1, nx2, ny2, nx3, ny3;
	if (D == 1)
		PIRE_BFLY_QUAD("quad_perm:[1,0,3,2]");
	else
		PIRE_BFLY_QUAD("quad_perm:[2,3,0,1]");
	x0 = nx0; y0 = ny0; x1 = nx1; y1 = ny1; x2 = nx2; y2 = ny2; x3 = nx3; y3 = ny3;
}

__device__ __forceinline__ void TransposeTile(u32x4 (&r)[8], uint32_t lane)
{
	(void)lane;
	const uint64_t lo1 = 0x5555555555555555ull, hi1 = 0xAAAAAAAAAAAAAAAAull;   // lane bit 0 clear / set
	const uint64_t lo2 = 0x3333333333333333ull, hi2 = 0xCCCCCCCCCCCCCCCCull;   // lane bit 1 clear / set
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		uint32_t d[8];
#pragma unroll
		for (int k = 0; k < 8; ++k)
			d[k] = r[k][w];
		ButterflyQuad4<1>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], lo1, hi1);
		ButterflyQuad4<2>(d[0], d[2], d[1], d[3], d[4], d[6], d[5], d[7], lo2, hi2);
		Butterfly4(d[0], d[4]); Butterfly4(d[1], d[5]); Butterfly4(d[2], d[6]); Butterfly4(d[3], d[7]);
#pragma unroll
		for (int k = 0; k < 8; ++k)
			r[k][w] = d[k];
		__builtin_amdgcn_sched_barrier(0);   // one column at a time: keeps the transpose's temporaries to ~10 VGPRs
	}
}

// The hot rows sit at LDS byte address 0 (the kernels declare no static __shared__, so the dynamic region
// starts at 0): the v_perm result IS the ds_read address, with no base add in the dependent chain.
typedef const __attribute__((address_space(3))) uint8_t* LdsBytePtr;
__device__ __forceinline__ uint32_t HotLookup(uint32_t addr)
{
	return *reinterpret_cast<LdsBytePtr>(static_cast<uintptr_t>(addr));
}
typedef const __attribute__((address_space(3))) uint16_t* LdsU16Ptr;
__device__ __forceinline__ uint32_t LdsU16(uint32_t addr)
{
	return *reinterpret_cast<LdsU16Ptr>(static_cast<uintptr_t>(addr));
}

// Compact tier: the exact walk of one 16-byte chunk for a lane whose state has a compact row, LDS
// only and branch free.  Row entries are the LDS address / 4 of the next state's row, so a step is
//   class2 = cls8[byte]            (off the dependent chain: 16 independent ds_read_u8)
//   row    = u16[row * 4 + class2] (the chain: v_lshl_add_u32 + ds_read_u16)
// Targets without a row lead to the absorbing escape row (id == p.compact): the caller then re-walks the chunk
// through the full table in HBM.  Returns the state id after the chunk (<= p.compact).
__device__ __forceinline__ uint32_t CompactChunk(const ScanParams& p, const LdsLayout& L, const u32x4 v, uint32_t st)
{
	const uint32_t pitch = CompactPitch(p.letters);
	uint32_t row = (L.compactOff >> 2) + st * (pitch >> 2);
	const uint32_t clsBase = L.cls8Off;   // multiple of 256: v_perm_b32 glues the byte under it
	// rolled on purpose: this code is instantiated once per unrolled chunk of the callers, and the kernels have to
	// stay well inside the 64 KiB instruction cache (measured: the unrolled form cost the tiled kernel 5%)
	u32x4 w = v;
#pragma unroll 1
	for (int i = 0; i < 4; ++i) {
		const uint32_t x = w.x;
		const uint32_t c0 = HotLookup(__builtin_amdgcn_perm(clsBase, x, 0x0c060500u));
		const uint32_t c1 = HotLookup(__builtin_amdgcn_perm(clsBase, x, 0x0c060501u));
		const uint32_t c2 = HotLookup(__builtin_amdgcn_perm(clsBase, x, 0x0c060502u));
		const uint32_t c3 = HotLookup(__builtin_amdgcn_perm(clsBase, x, 0x0c060503u));
		row = LdsU16((row << 2) + c0);
		row = LdsU16((row << 2) + c1);
		row = LdsU16((row << 2) + c2);
		row = LdsU16((row << 2) + c3);
		w.x = w.y;
		w.y = w.z;
		w.z = w.w;
	}
	return LdsU16((row << 2) + p.letters * 2);
}

// Same for the first `count` (1..15) bytes of v, rolled.
__device__ __forceinline__ uint32_t CompactPartial(const ScanParams& p, const LdsLayout& L, u32x4 v, uint32_t st,
                                                   uint32_t count)
{
	const uint32_t pitch = CompactPitch(p.letters);
	uint32_t row = (L.compactOff >> 2) + st * (pitch >> 2);
	const uint32_t clsBase = L.cls8Off;
#pragma unroll 1
	for (uint32_t i = 0; __any(i < count); ++i) {
		const uint32_t c = HotLookup(__builtin_amdgcn_perm(clsBase, v.x, 0x0c060500u));
		const uint32_t nr = LdsU16((row << 2) + c);
		row = i < count ? nr : row;
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return LdsU16((row << 2) + p.letters * 2);
}

// Exact re-walk of one 16-byte chunk for the lanes that trapped.  Deliberately a rolled loop (the chunk is shifted
// through as a 128-bit value): this is the cold path, and keeping it small keeps the hot loop dense in the I-cache.
__device__ __forceinline__ uint32_t SlowChunk(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, u32x4 v,
                                           uint32_t st)
{
#pragma unroll 1
	for (int i = 0; i < 16; ++i) {
		st = SlowStep(p, lds, L, st, v.x & 0xFF);
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return st;
}

// 16 bytes (one dwordx4) through the hot table; lanes that leave the hot set are re-walked exactly.
template <int ROT>
__device__ __forceinline__ void StepChunk(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                          const u32x4 v, uint32_t& hs, uint32_t& cold, uint32_t sampleLane)
{
	const uint32_t hs0 = hs;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		if (ROT == 1) {
			// rows are 65 dwords apart, so row r is rotated by r banks: lanes in different states reading the same
			// byte>>2 no longer hit the same bank.  The byte is extracted off the dependent chain; the chain
			// itself stays one VALU (v_mad_u32_u24) + one ds_read_u8.
			const uint32_t b0 = x & 0xFF, b1 = (x >> 8) & 0xFF, b2 = (x >> 16) & 0xFF, b3 = x >> 24;
			hs = HotLookup(__umul24(hs, kRotPitch) + b0);
			hs = HotLookup(__umul24(hs, kRotPitch) + b1);
			hs = HotLookup(__umul24(hs, kRotPitch) + b2);
			hs = HotLookup(__umul24(hs, kRotPitch) + b3);
		} else {
			hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0400u));
			hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0401u));
			hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0402u));
			hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0403u));
		}
	}
	if (hs == p.hot && !(p.flags & kDebugNoTrap)) {
		// left the dense rows somewhere in this chunk: exact re-walk from the chunk's start state, through the
		// compact rows in LDS when the state has one, through the full table in HBM when that escapes too
		const uint32_t st0 = hs0 != p.hot ? hs0 : cold;
		uint32_t f = p.compact;
		if (st0 < p.compact)
			f = CompactChunk(p, L, v, st0);
		if (f == p.compact)
			f = SlowChunk(p, lds, L, v, st0);
		if (f < p.hot) {
			hs = f;
		} else {
			hs = p.hot;
			cold = f;
			// Rare path: tell pire_hip_table_adapt() which rows deserve LDS.  SAMPLED (one rotating lane of 64):
			// un-sampled, the device-scope atomics of every trapped lane serialised on a few dozen addresses and
			// cost 4x the whole kernel (measured: 0.80 -> 3.45 ms on set_a).
			if ((threadIdx.x & 63) == sampleLane && !(p.flags & kDebugNoColdCount))
				atomicAdd(&p.visitCold[f], 1u);
		}
	}
}

template <int ROT>
__device__ __forceinline__ void StepTile(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                         const u32x4 (&r)[8], uint32_t& hs, uint32_t& cold, uint32_t tile)
{
#pragma unroll
	for (int k = 0; k < 8; ++k)
		StepChunk<ROT>(p, lds, L, r[k], hs, cold, (tile * 8 + k) & 63);
}

// Wave-wide early out (north_star: "wavefront ballot/any for early-out on dead states"): once every lane sits
// in a row whose every transition is a self loop, the rest of the text cannot change any state.  This is the
// GPU counterpart of the NO_EXIT_MASK return of multi.h:955-958, 979-982.
__device__ __forceinline__ bool AllAbsorbing(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint32_t hs)
{
	const bool a = hs != p.hot && (lds[L.flagsOff + hs] & kAbsorbing);
	return __all(a);
}

__device__ __forceinline__ void ZeroTile(u32x4 (&r)[8])
{
#pragma unroll
	for (int k = 0; k < 8; ++k)
		r[k] = u32x4{0, 0, 0, 0};
}

__device__ __forceinline__ uint64_t Uniform64(uint64_t v)
{
	const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v));
	const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
	return (uint64_t(hi) << 32) | lo;
}

// One pipeline phase of the register ring: refill the slot that was freed one phase ago with the tile NBUF-1
// ahead (index clamped to the last tile, so the steady-state loop has no conditional loads), wait until the
// current slot has landed, transpose it into lane-owns-string order, walk it.
template <int NBUF, bool NT, int ROT>
__device__ __forceinline__ void Phase(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint64_t rowBase,
                                      uint64_t chainBase, uint32_t voff, uint32_t lane, uint32_t t, uint32_t lastTile,
                                      u32x4 (&cur)[8], u32x4 (&refill)[8], uint32_t& hs, uint32_t& cold)
{
	// Refill target: the next tile of this task, or -- on the task's last tile -- tile 0 of the wave's NEXT task
	// (chainBase; equals this task's last tile when there is nothing to chain to), so that neither the HBM
	// latency of a task's first tile nor a duplicate load of its last tile is ever paid.
	const uint64_t ahead = t < lastTile ? rowBase + uint64_t(t + 1) * 128 : chainBase;
	if (!(p.flags & kDebugNoRefill))   // measurement knob only (PIRE_HIP_DEBUG_NOLOAD): walk stale registers
		IssueTile<NT>(refill, voff, ahead, p.stride);
	WaitTile<NBUF - 1>(cur);
	TransposeTile(cur, lane);
	if (lane == (t & 63) && !(p.flags & kDebugNoHist))   // visit sample: one lane per wave per tile, rotating
		atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + hs, 1u);
	if (p.flags & kDebugNoStep) {      // measurement knob only (PIRE_HIP_DEBUG_NOSTEP): stream + transpose, no walk
		hs ^= (cur[0].x ^ cur[7].w) & 1;
		return;
	}
	StepTile<ROT>(p, lds, L, cur, hs, cold, t);
}

// Fixed-length records, 16-byte aligned, whole tasks of 64 strings (the host routes the < 64-string remainder to
// the generic kernel).  NBUF register tiles per wave form a ring: tile t is walked out of registers -- one LDS
// gather per byte -- while tiles t+1 .. t+NBUF-1 stream in from HBM.
template <int WAVES, int NBUF, bool NT, int MINW, int ROT>
__global__ __launch_bounds__(WAVES * 64, MINW) void ScanTiledKernel(ScanParams p)
{
	// Depth 2 only: with three slots hipcc (ROCm 7.2) spills tile registers to scratch WHILE their loads are in
	// flight (profiles/ + DESIGN.md section 6) -- silently wrong data.  tests/test_build_audit.py pins "no scratch".
	static_assert(NBUF == 2, "ring depth");
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, ROT == 1 ? kRotPitch : 256u, CompactBytes(p));
	LoadTableToLds(p, lds, L);

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t ntasks = p.n / 64;                // whole tasks only
	const uint32_t ntiles = uint32_t(p.len / 128);   // >= 1 (TiledEligible)
	const uint32_t lastTile = ntiles - 1;
	const uint32_t groups = ntiles / NBUF;
	const uint32_t rem = ntiles % NBUF;
	// per-lane byte offset inside a task's tile: string (lane & ~7) [+ j per instruction], chunk (lane & 7)
	const uint32_t voff = (lane & ~7u) * uint32_t(p.stride) + (lane & 7u) * 16;

	// Ring slots: tile t lives in slot t % 2.
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);

	// With an even tile count every task starts in slot a, so the ring can run straight through task boundaries.
	const bool chain = rem == 0;
	bool primed = false;   // slot a already holds (or is receiving) tile 0 of the task about to start
	const uint64_t taskStep = uint64_t(gridDim.x) * WAVES;
	for (uint64_t task = uint64_t(blockIdx.x) * WAVES + wave; task < ntasks; task += taskStep) {
		const uint64_t s0 = task * 64;
		const uint64_t s = s0 + lane;
		const uint64_t rowBase = Uniform64(reinterpret_cast<uint64_t>(p.text) + s0 * p.stride);
		const bool hasNext = chain && task + taskStep < ntasks;
		const uint64_t chainBase = hasNext ? Uniform64(reinterpret_cast<uint64_t>(p.text) + (s0 + taskStep * 64) * p.stride)
		                                   : rowBase + uint64_t(lastTile) * 128;

		uint32_t cold = StartState(p, s);
		uint32_t hs = cold < p.hot ? cold : p.hot;

		bool done = false;
		if (!primed)
			IssueTile<NT>(a, voff, rowBase, p.stride);
		for (uint32_t g = 0; g < groups && !done; ++g) {
			const uint32_t t = g * 2;
			Phase<2, NT, ROT>(p, lds, L, rowBase, chainBase, voff, lane, t, lastTile, a, b, hs, cold);
			Phase<2, NT, ROT>(p, lds, L, rowBase, chainBase, voff, lane, t + 1, lastTile, b, a, hs, cold);
			done = AllAbsorbing(p, lds, L, hs);
		}
		primed = hasNext && !done;   // an early-out leaves some other tile in slot a: re-prime then
		if (!done && rem == 1) {
			WaitTile<0>(a);
			TransposeTile(a, lane);
			StepTile<ROT>(p, lds, L, a, hs, cold, lastTile);
		}

		uint32_t st = hs != p.hot ? hs : cold;
		// tail shorter than a tile: exact steps straight from memory
		if (!done) {
			const uint8_t* base = p.text + s * p.stride;
			for (uint64_t i = uint64_t(ntiles) * 128; i < p.len; ++i)
				st = SlowStep(p, lds, L, st, base[i]);
		}
		Finish(p, lds, L, s, true, st);
	}
	FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ ragged kernel
// Variable-length strings given by offsets -- the natural input of the reference's callers (URLs, log lines, one
// Runner per string: bench.cpp:244, pigrep.cpp:42).  One string per lane, but a lane is NOT tied to a string: when
// its string ends it takes the next one, so a wave stays full however uneven the lengths are.
//
//   * work distribution: blocks take ranges of `blockGrab` strings from one global counter (a few thousand atomics
//     per launch, not one per wave: same-address device atomics run at well under 100 per microsecond); waves take
//     64 strings at a time from their block's range in LDS; lanes take single strings from their wave's range by
//     ballot + mbcnt.
//   * every lane walks its string in windows of up to 128 bytes (8 x global_load_dwordx4 from its own address).  A
//     window starts at the string's current byte whatever its alignment, so a string of <= 128 bytes is ONE window; a
//     longer string cuts its first window at a 16-byte boundary and is aligned from then on.  The window of the NEXT iteration -- the same string's next 128 bytes, or the first window of
//     the lane's pending next string, whose offsets were fetched an iteration earlier -- is in flight while the
//     current one is walked; nothing on the common path makes the compiler wait for memory during the walk (the
//     end-of-string records of the hot states are in LDS for that reason).
//   * a window is walked as whole 16-byte chunks with the LDS fast path of the tiled kernel, then ONE pass for the
//     <= 15 bytes behind the last whole chunk of every lane that has some: each lane picks its chunk, walks all 16
//     bytes unrolled and keeps the state after its last real byte (no loop, no branches).
//   * nothing is read past the 16-byte block that holds the last byte of the text: a window that would reach further
//     is walked byte by byte from memory instead.

struct RaggedWork {
	unsigned long long next, end;   // the block's current range of string indices
	uint32_t lock, exhausted;
	uint32_t pad[2];
};
static_assert(kRaggedFinBytes + sizeof(RaggedWork) == kRaggedLdsExtra, "LDS budget of the warm rows (internal.h)");

__device__ __forceinline__ void IssueTileLane(u32x4 (&r)[8], uint64_t src)
{
	asm volatile(
		"global_load_dwordx4 %0, %8, off\n\t"
		"global_load_dwordx4 %1, %8, off offset:16\n\t"
		"global_load_dwordx4 %2, %8, off offset:32\n\t"
		"global_load_dwordx4 %3, %8, off offset:48\n\t"
		"global_load_dwordx4 %4, %8, off offset:64\n\t"
		"global_load_dwordx4 %5, %8, off offset:80\n\t"
		"global_load_dwordx4 %6, %8, off offset:96\n\t"
		"global_load_dwordx4 %7, %8, off offset:112"
		: "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
		: "v"(src));
}

__device__ __forceinline__ void WaitAllLoads(u32x4 (&r)[8])
{
	asm volatile("s_waitcnt vmcnt(0)"
	             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}

// Exact walk of the first `count` (0..15) bytes of v; lanes with a smaller count idle (one rolled loop per wave).
__device__ __forceinline__ uint32_t SlowPartial(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, u32x4 v,
                                                uint32_t st, uint32_t count)
{
#pragma unroll 1
	for (uint32_t i = 0; __any(i < count); ++i) {
		if (i < count)
			st = SlowStep(p, lds, L, st, v.x & 0xFF);
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return st;
}

// The first `count` (0..15) bytes of v through the LDS fast path: the whole chunk is walked, unrolled like StepChunk,
// and the state after byte `count` is kept (v_cmp + v_cndmask per byte, no loop, no branches); lanes with count == 0
// keep their state.  What the walk reads past `count` is ignored.  Exact re-walk on a trap like StepChunk.
__device__ __forceinline__ void StepPartial(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const u32x4 v,
                                            uint32_t count, uint32_t& hs, uint32_t& cold, uint32_t sampleLane)
{
	const uint32_t hs0 = hs;
	uint32_t h = hs, snap = hs;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0400u));
		snap = count == uint32_t(4 * w + 1) ? h : snap;
		h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0401u));
		snap = count == uint32_t(4 * w + 2) ? h : snap;
		h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0402u));
		snap = count == uint32_t(4 * w + 3) ? h : snap;
		if (w < 3) {
			h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0403u));
			snap = count == uint32_t(4 * w + 4) ? h : snap;
		}
	}
	hs = snap;
	if (count != 0 && hs == p.hot && !(p.flags & kDebugNoTrap)) {
		const uint32_t st0 = hs0 != p.hot ? hs0 : cold;
		uint32_t f = p.compact;
		if (st0 < p.compact)
			f = CompactPartial(p, L, v, st0, count);
		if (f == p.compact)
			f = SlowPartial(p, lds, L, v, st0, count);
		if (f < p.hot) {
			hs = f;
		} else {
			hs = p.hot;
			cold = f;
			if ((threadIdx.x & 63) == sampleLane && !(p.flags & kDebugNoColdCount))
				atomicAdd(&p.visitCold[f], 1u);
		}
	}
}

__device__ __forceinline__ void FinishRagged(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                             uint32_t s, bool active, uint32_t st)
{
	u32x4 raw = {0, 0, 0, 0};
	if (active) {
		if (st < p.hot) {
			raw = *reinterpret_cast<const u32x4*>(&finHot[st]);
		} else {
			const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
			raw = *reinterpret_cast<const u32x4*>(&recs[st]);
		}
	}
	const uint32_t orig = raw.x, endPerm = raw.y & 0x0FFFFFFFu, fl = raw.y >> 28;
	if (active) {
		if (p.outIdx)
			p.outIdx[s] = orig;
		if (p.outFinal)
			p.outFinal[s] = fl & kFinal;
	}
	if (p.outCounts) {
		uint32_t* cnt = reinterpret_cast<uint32_t*>(lds + L.countsOff);
		const int lane = threadIdx.x & 63;
		const unsigned long long finals = __ballot(active && (fl & kFinal));
		const unsigned long long actives = __ballot(active);
		if (lane == 0) {
			atomicAdd(&cnt[0], (uint32_t)__popcll(finals));
			atomicAdd(&cnt[1], (uint32_t)__popcll(actives));
		}
		if (p.acceptMaskPerm) {
			// only ended strings that accept anything get here: most ends accept nothing
			const uint64_t m = active ? ((uint64_t(raw.w) << 32) | raw.z) : 0;
			if (__any(m != 0))
				for (uint32_t r = 0; r < p.regexps; ++r) {
					const unsigned long long b = __ballot((m >> r) & 1);
					if (lane == 0 && b)
						atomicAdd(&cnt[2 + r], (uint32_t)__popcll(b));
				}
		} else if (active) {
			for (uint64_t k = p.acceptOffPerm[endPerm]; k < p.acceptOffPerm[endPerm + 1]; ++k)
				atomicAdd(&cnt[2 + p.acceptIds[k]], 1u);
		}
	}
}

// Per-lane walking state of the ragged kernel.
struct RaggedLane {
	uint64_t pos, end;   // absolute addresses of the unread part of the current string
	uint32_t sIdx;
	uint32_t hs, cold;
	bool busy;           // has a current string
	bool loaded;         // the current window is in the tile registers (else: walk it from memory)
	// pending next string: its offsets are fetched one iteration before it starts
	uint64_t pendPos, pendEnd;
	uint32_t sIdxN;
	bool pend;
};

// Wave-uniform range of strings still to hand out, refilled from the block's range, refilled from the global counter.
struct RaggedRange {
	uint64_t next, end;
	bool exhausted;
};

__device__ __forceinline__ void GrabWaveRange(const ScanParams& p, volatile RaggedWork* work,
                                              unsigned long long* workCounter, uint32_t blockGrab, RaggedRange& R)
{
	unsigned long long r0 = 0, r1 = 0;
	if ((threadIdx.x & 63) == 0) {
		while (atomicCAS(const_cast<uint32_t*>(&work->lock), 0u, 1u) != 0u)
			__builtin_amdgcn_s_sleep(2);
		unsigned long long nx = work->next, en = work->end;
		if (nx >= en && !work->exhausted) {
			const unsigned long long base = atomicAdd(workCounter, (unsigned long long)blockGrab);
			if (base >= p.n) {
				work->exhausted = 1;
			} else {
				nx = base;
				en = base + blockGrab < p.n ? base + blockGrab : p.n;
			}
		}
		const unsigned long long take = en - nx < 64 ? en - nx : 64;
		r0 = nx;
		r1 = nx + take;
		work->next = r1;
		work->end = en;
		__threadfence_block();
		atomicExch(const_cast<uint32_t*>(&work->lock), 0u);
	}
	R.next = Uniform64(r0);
	R.end = Uniform64(r1);
	R.exhausted = R.next >= R.end;
}

// Give every lane without a pending string the next unassigned one.  Returns (per lane) whether it got one.
__device__ __forceinline__ bool AssignPending(const ScanParams& p, volatile RaggedWork* work,
                                              unsigned long long* workCounter, uint32_t blockGrab, RaggedRange& R,
                                              RaggedLane& S)
{
	bool need = !S.pend, got = false;
	for (;;) {
		const unsigned long long mask = __ballot(need);
		if (!mask)
			break;
		if (R.next >= R.end) {
			if (R.exhausted)
				break;
			GrabWaveRange(p, work, workCounter, blockGrab, R);
			if (R.exhausted)
				break;
		}
		const uint64_t avail = R.end - R.next;
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0));
		if (need && rank < avail) {
			S.sIdxN = uint32_t(R.next) + rank;
			S.pend = true;
			need = false;
			got = true;
		}
		const uint64_t want = uint64_t(__popcll(mask));
		R.next += want < avail ? want : avail;
	}
	return got;
}

// One iteration: start fetching the next window into `nxt`, walk the current window held in `cur`.
// Returns false when the wave has nothing left to do.
__device__ __forceinline__ bool RaggedPhase(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                            volatile RaggedWork* work, unsigned long long* workCounter,
                                            uint32_t blockGrab, uint64_t textBase, uint64_t safeEnd, RaggedRange& R,
                                            RaggedLane& S, u32x4 (&cur)[8], u32x4 (&nxt)[8], uint32_t iter)
{
	WaitAllLoads(cur);

	// ---- this window: starts at the string's current byte, whatever its alignment.  A string that fits takes one
	// window; a longer one cuts its first window at a 16-byte boundary so that all the following ones are aligned.
	const uint64_t left = S.end - S.pos;
	const uint32_t nb = !S.busy ? 0u : left <= 128u ? uint32_t(left) : 128u - (uint32_t(S.pos) & 15u);
	const bool ends = S.busy && nb == left;

	// ---- the next window: the same string's next bytes, or the pending string's first window
	const bool cont = S.busy && !ends;
	const bool takeNew = !cont && S.pend;
	const uint64_t nPos = cont ? S.pos + nb : S.pendPos;
	const uint64_t nEnd = cont ? S.end : S.pendEnd;
	const uint32_t nIdx = cont ? S.sIdx : S.sIdxN;
	const bool nBusy = cont || takeNew;
	const bool nLoad = nBusy && nEnd > nPos && nPos + 128 <= safeEnd;
	// unconditional (idle lanes fetch a harmless valid line): a load under a per-lane condition could be turned into
	// load-to-a-copy + select by the compiler, and the select would read the register before the data arrives
	if (!(p.flags & kDebugNoRefill))
		IssueTileLane(nxt, nLoad ? nPos : reinterpret_cast<uint64_t>(p.hotRows));
	if (takeNew)
		S.pend = false;
	// the offsets of newly assigned strings: plain loads issued AFTER the tile loads and looked at only at the very
	// end of this iteration, so the one wait the compiler inserts for them sits behind the walk
	const bool got = AssignPending(p, work, workCounter, blockGrab, R, S);
	uint64_t offB = 0, offE = 0;
	if (__any(got)) {
		const uint64_t* offPtr = p.offsets + (got ? S.sIdxN : 0u);
		offB = offPtr[0];
		offE = offPtr[1];
	}

	// ---- walk the current window
	if ((threadIdx.x & 63) == (iter & 63) && nb != 0 && !(p.flags & kDebugNoHist))   // visit sample, as in the tiled kernel
		atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + S.hs, 1u);
	if (p.flags & kDebugNoStep) {
		// timing experiments: no walk at all
	} else if (__any(nb != 0)) {
		if (__any(nb != 0 && S.loaded)) {
			const uint32_t nbl = S.loaded ? nb : 0u;
			const uint32_t full = nbl >> 4, tail = nbl & 15u;
#pragma unroll
			for (int k = 0; k < 8; ++k)
				if (uint32_t(k) < full)
					StepChunk<0>(p, lds, L, cur[k], S.hs, S.cold, (iter * 8 + k) & 63);
			if (__any(tail != 0) && !(p.flags & kDebugNoPartial)) {
				// all the partial last chunks of the wave in ONE pass: pick each lane's chunk, walk it with a snapshot
				u32x4 v = cur[0];
#pragma unroll
				for (int k = 1; k < 8; ++k)
					if (full == uint32_t(k))
						v = cur[k];
				StepPartial(p, lds, L, v, tail, S.hs, S.cold, (iter + 32) & 63);
			}
		}
		if (nb != 0 && !S.loaded) {
			// the last bytes of the whole buffer: exact steps straight from memory
			uint32_t st = S.hs != p.hot ? S.hs : S.cold;
			const uint8_t* q = reinterpret_cast<const uint8_t*>(S.pos);
			for (uint32_t i = 0; i < nb; ++i)
				st = SlowStep(p, lds, L, st, q[i]);
			S.hs = st < p.hot ? st : p.hot;
			S.cold = st;
		}
	}
	if (__any(ends) && !(p.flags & kDebugNoFinish))
		FinishRagged(p, lds, L, finHot, S.sIdx, ends, S.hs != p.hot ? S.hs : S.cold);

	// ---- move on
	if (got) {
		S.pendPos = textBase + offB;
		S.pendEnd = textBase + offE;
	}
	if (takeNew) {
		const uint32_t st = StartState(p, nIdx);
		S.hs = st < p.hot ? st : p.hot;
		S.cold = st;
	}
	S.pos = nPos;
	S.end = nEnd;
	S.sIdx = nIdx;
	S.busy = nBusy;
	S.loaded = nLoad;
	return __any(nBusy || S.pend);
}

__global__ __launch_bounds__(1024) void ScanRaggedKernel(ScanParams p, unsigned long long* workCounter,
                                                         uint32_t blockGrab)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	FinRec* finHot = reinterpret_cast<FinRec*>(lds + L.total);
	volatile RaggedWork* work = reinterpret_cast<volatile RaggedWork*>(lds + L.total + kRaggedFinBytes);
	{
		const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			finHot[i] = recs[i];
		if (threadIdx.x == 0) {
			work->next = 0;
			work->end = 0;
			work->lock = 0;
			work->exhausted = 0;
		}
	}
	LoadTableToLds(p, lds, L);   // ends with a barrier

	const uint64_t textBase = reinterpret_cast<uint64_t>(p.text);
	const uint64_t safeEnd = (textBase + p.offsets[p.n] + 15) & ~uint64_t(15);

	RaggedRange R = {0, 0, false};
	RaggedLane S;
	S.pos = S.end = textBase;
	S.sIdx = 0;
	S.hs = S.cold = 0;
	S.busy = S.loaded = S.pend = false;
	S.sIdxN = 0;
	S.pendPos = S.pendEnd = textBase;
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);

	if (AssignPending(p, work, workCounter, blockGrab, R, S)) {
		S.pendPos = textBase + p.offsets[S.sIdxN];
		S.pendEnd = textBase + p.offsets[S.sIdxN + 1];
	}
	for (uint32_t iter = 0;; iter += 2) {
		if (!RaggedPhase(p, lds, L, finHot, work, workCounter, blockGrab, textBase, safeEnd, R, S, a, b, iter))
			break;
		if (!RaggedPhase(p, lds, L, finHot, work, workCounter, blockGrab, textBase, safeEnd, R, S, b, a, iter + 1))
			break;
	}
	FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ HalfFinalScanner
// Pire::HalfFinalScanner (scanners/half_final.h) is a Scanner whose Initialize and every Step end with TakeAction:
// if the new state is Final, every entry of its final list bumps the per-regexp match counter of the string
// (half_final.h:137-164).  Same table, same walk, plus State::Result(r) per string.  First version: the generic
// kernel's exact walk with a one-compare Final test per step -- the hot set is ordered non-final first, so
// "hot and Final" is `st >= hotFinalLo`, cold states look their flags up -- and, for up to 8 regexps, counters in
// registers fed from a packed increment word per state (hot states: LDS).

template <bool PACKED>
struct HalfCounters;

template <>
struct HalfCounters<true> {
	uint32_t c[8];
	__device__ __forceinline__ void Init(const ScanParams&, uint32_t*, uint64_t)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
			c[r] = 0;
	}
	__device__ __forceinline__ void Take(const ScanParams& p, const uint64_t* incHot, uint32_t st)
	{
		const uint64_t inc = st < p.hot ? incHot[st] : p.incPerm[st];
#pragma unroll
		for (int r = 0; r < 8; ++r)
			c[r] += uint32_t(inc >> (8 * r)) & 0xFFu;
	}
	__device__ __forceinline__ void Store(const ScanParams& p, uint32_t* out, uint64_t s)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)      // static indices only: a runtime index would put c[] into scratch
			if (uint32_t(r) < p.regexps)
				out[s * p.regexps + r] = c[r];
	}
};

template <>
struct HalfCounters<false> {
	uint32_t* row;
	__device__ __forceinline__ void Init(const ScanParams& p, uint32_t* out, uint64_t s)
	{
		row = out + s * p.regexps;
		for (uint32_t r = 0; r < p.regexps; ++r)
			row[r] = 0;
	}
	__device__ __forceinline__ void Take(const ScanParams& p, const uint64_t*, uint32_t st)
	{
		for (uint64_t k = p.acceptOffPerm[st]; k < p.acceptOffPerm[st + 1]; ++k)
			row[p.acceptIds[k]] += 1;   // the lane owns the row: plain read-modify-write
	}
	__device__ __forceinline__ void Store(const ScanParams&, uint32_t*, uint64_t) {}
};

__device__ __forceinline__ bool IsFinalState(const ScanParams& p, uint32_t st)
{
	return st >= p.hotFinalLo && (st < p.hot || (p.flagsPerm[st] & kFinal));
}

template <bool PACKED>
__global__ __launch_bounds__(256) void HalfFinalKernel(ScanParams p, uint32_t* outResults)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	uint64_t* incHot = reinterpret_cast<uint64_t*>(lds + L.total);
	if (PACKED)
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			incHot[i] = p.incPerm[i];
	LoadTableToLds(p, lds, L);

	const uint64_t nrounds = (p.n + 63) / 64;
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t initial = p.startPerm;   // Initialize(): the launcher passes flags without BEGIN to FillParams
	for (uint64_t task = uint64_t(blockIdx.x) * wavesPerBlock + (threadIdx.x >> 6); task < nrounds;
	     task += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t s = task * 64 + lane;
		if (s >= p.n)
			continue;
		HalfCounters<PACKED> cnt;
		cnt.Init(p, outResults, s);
		uint32_t st = initial;
		if (IsFinalState(p, st))
			cnt.Take(p, incHot, st);                       // Initialize ends with TakeAction, half_final.h:142
		if (p.flags & PIRE_HIP_RUN_BEGIN) {
			st = p.nextPerm[size_t(st) * p.letters + p.beginCls];
			if (IsFinalState(p, st))
				cnt.Take(p, incHot, st);
		}
		const uint8_t* ptr = p.text + p.offsets[s];
		const uint8_t* end = p.text + p.offsets[s + 1];
		auto step = [&](uint32_t byte) {
			st = SlowStep(p, lds, L, st, byte);
			if (IsFinalState(p, st))
				cnt.Take(p, incHot, st);
		};
		while (ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
			step(*ptr);
			++ptr;
		}
		for (; ptr + 16 <= end; ptr += 16) {
			u32x4 v = *reinterpret_cast<const u32x4*>(ptr);
#pragma unroll 1
			for (int i = 0; i < 16; ++i) {
				step(v.x & 0xFF);
				v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
				v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
				v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
				v.w >>= 8;
			}
		}
		for (; ptr < end; ++ptr)
			step(*ptr);
		if (p.flags & PIRE_HIP_RUN_END) {
			st = p.nextPerm[size_t(st) * p.letters + p.endCls];
			if (IsFinalState(p, st))
				cnt.Take(p, incHot, st);
		}
		cnt.Store(p, outResults, s);
		if (p.outIdx)
			p.outIdx[s] = p.origOfPerm[st];
		if (p.outFinal)
			p.outFinal[s] = p.flagsPerm[st] & kFinal;
	}
}

// ------------------------------------------------------------------------------------------ prefix searches
// Pire::LongestPrefix / ShortestPrefix (run.h:277-311) with LongestPrefixPred / ShortestPrefixPred (run.h:69-100):
// the same walk, but after every byte Final(state) records the position and Dead(state) (or, for the shortest
// prefix, the first Final) ends it.  One string per lane, exact step (dense row first); the early exit is per lane.

struct PrefixParams {
	ScanParams scan;
	uint32_t longest, throughEnd;
	long long* outLen;
};

__device__ __forceinline__ uint32_t StateFlags(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint32_t st)
{
	return st < p.hot ? lds[L.flagsOff + st] : p.flagsPerm[st];
}

__global__ __launch_bounds__(256) void PrefixKernel(PrefixParams q)
{
	const ScanParams& p = q.scan;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	LoadTableToLds(p, lds, L);
	for (uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; s < p.n; s += uint64_t(gridDim.x) * blockDim.x) {
		uint64_t b, e;
		if (p.offsets) {
			b = p.offsets[s];
			e = p.offsets[s + 1];
		} else {
			b = s * p.stride;
			e = b + p.len;
		}
		const uint8_t* text = p.text + b;
		const uint64_t len = e - b;
		uint32_t st = p.startPerm;                       // Initialize (+ BeginMark if throughBeginMark), run.h:280-283
		long long pos = -1;
		bool stop = false;
		uint32_t f = StateFlags(p, lds, L, st);
		if (f & kFinal) {
			pos = 0;                                     // run.h:284 / 301-302
			stop = !q.longest;
		}
		const bool foundAtStart = stop;
		if (!stop) {
			uint64_t i = 0;
			WalkBytes(text, text + len, [&](uint32_t byte) {   // line-aligned vector loads instead of byte loads
				st = SlowStep(p, lds, L, st, byte);
				f = StateFlags(p, lds, L, st);
				++i;
				if (f & kFinal) {
					pos = (long long)i;
					if (!q.longest)
						stop = true;                     // ShortestPrefixPred: Stop on the first Final
				}
				if (f & kDead)
					stop = true;                         // both predicates stop on a dead state
				return !stop;
			});
		}
		if (q.throughEnd && !foundAtStart) {
			st = p.nextPerm[size_t(st) * p.letters + p.endCls];
			if (StateFlags(p, lds, L, st) & kFinal) {
				if (q.longest || pos < 0)
					pos = (long long)len;                // run.h:286-290 / 305-309
			}
		}
		q.outLen[s] = pos;
	}
}

// ------------------------------------------------------------------------------------------ single Step()

__global__ __launch_bounds__(256) void StepKernel(ScanParams p, uint32_t* stateIdx, uint64_t n, uint32_t cls)
{
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) {
		const uint32_t st = p.permOfOrig[stateIdx[i]];
		stateIdx[i] = p.origOfPerm[p.nextPerm[size_t(st) * p.letters + cls]];
	}
}

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

// ------------------------------------------------------------------------------------------ launchers

namespace {

int DeviceCUs(int* cus)
{
	// asked on every launch: cache per device (hipGetDeviceProperties is far too slow for that)
	static std::atomic<int> cached[64];
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return HipFail(e, "hipGetDevice");
	const bool slot = dev >= 0 && dev < 64;
	int v = slot ? cached[dev].load(std::memory_order_relaxed) : 0;
	if (v == 0) {
		e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
		if (e != hipSuccess)
			return HipFail(e, "hipDeviceGetAttribute(multiprocessor count)");
		if (slot)
			cached[dev].store(v, std::memory_order_relaxed);
	}
	*cus = v;
	return PIRE_HIP_OK;
}

template <class K>
int LaunchScan(K kernel, const ScanParams& p, int threads, uint32_t ldsBytes, hipStream_t stream)
{
	int cus = 0;
	int rc = DeviceCUs(&cus);
	if (rc)
		return rc;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	int perCu = 0;
	e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, threads, ldsBytes);
	if (e != hipSuccess)
		return HipFail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
	if (getenv("PIRE_HIP_DEBUG_LAUNCH"))
		fprintf(stderr, "pire_hip: threads %d lds %u -> %d blocks/CU\n", threads, ldsBytes, perCu);
	if (perCu < 1)
		perCu = 1;
	const uint64_t ntasks = (p.n + 63) / 64;
	const uint64_t wavesPerBlock = uint64_t(threads) / 64;
	uint64_t blocks = (ntasks + wavesPerBlock - 1) / wavesPerBlock;
	blocks = std::min<uint64_t>(blocks, uint64_t(cus) * perCu);
	if (blocks == 0)
		blocks = 1;
	hipLaunchKernelGGL(kernel, dim3(unsigned(blocks)), dim3(threads), ldsBytes, stream, p);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "kernel launch");
	return PIRE_HIP_OK;
}

int CheckCounts(const ScanParams& p)
{
	if (p.outCounts && p.regexps > kMaxLdsCountRegexps) {
		SetError("out_counts is supported for scanners with at most 1024 regexps");
		return PIRE_HIP_EUNSUPPORTED;
	}
	return PIRE_HIP_OK;
}

}  // namespace

int LaunchGeneric(const ScanParams& p0, hipStream_t stream)
{
	if (int rc = CheckCounts(p0))
		return rc;
	ScanParams p = p0;
	p.compact = 0;   // small blocks, several per CU: no room (and no need) for the warm rows
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, kRotPitch, CompactBytes(p));
	return LaunchScan(ScanGenericKernel, p, 256, L.total, stream);
}

bool RaggedEligible(const ScanParams& p, uint64_t totalBytesHint)
{
	// one string per lane with dynamic re-assignment: worth it from a few waves' worth of strings
	return p.offsets != nullptr && p.n >= 256 && p.n < (1ull << 32) && totalBytesHint >= 4096;
}

int LaunchRagged(const ScanParams& p, unsigned long long* workCounter, hipStream_t stream)
{
	if (int rc = CheckCounts(p))
		return rc;
	hipError_t e = hipMemsetAsync(workCounter, 0, sizeof(unsigned long long), stream);
	if (e != hipSuccess)
		return HipFail(e, "hipMemsetAsync(work counter)");
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	const uint32_t ldsBytes = L.total + kRaggedLdsExtra;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	e = hipFuncSetAttribute(reinterpret_cast<const void*>(ScanRaggedKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
	                        int(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	// one string per lane: spread the waves over every CU before stacking them (4..16 waves per block, 1 block per CU)
	const uint64_t waves = (p.n + 63) / 64;
	const uint64_t wavesPerBlock = std::min<uint64_t>(16, std::max<uint64_t>(4, (waves + cus - 1) / cus));
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(cus), (waves + wavesPerBlock - 1) / wavesPerBlock));
	// strings a block takes from the global counter at a time: ~8 grabs per block keep the tail balanced; batches
	// that barely fill the lanes are simply split evenly
	const uint64_t perBlock = (p.n + blocks - 1) / blocks, lanes = wavesPerBlock * 64;
	uint64_t grab = std::min<uint64_t>(16384, std::max<uint64_t>(perBlock / 8, std::min(perBlock, lanes)));
	grab = (grab + 63) / 64 * 64;
	ScanParams q = p;
	if (const char* dbg = getenv("PIRE_HIP_DEBUG_RAGGED")) {   // timing experiments: 1 no partial passes, 2 no finish, 4 no traps
		const int m = atoi(dbg);
		q.flags |= (m & 1 ? kDebugNoPartial : 0) | (m & 2 ? kDebugNoFinish : 0) | (m & 4 ? kDebugNoTrap : 0) |
		           (m & 8 ? kDebugNoStep : 0) | (m & 16 ? kDebugNoRefill : 0);
	}
	hipLaunchKernelGGL(ScanRaggedKernel, dim3(unsigned(blocks)), dim3(unsigned(wavesPerBlock * 64)), ldsBytes, stream, q,
	                   workCounter, uint32_t(grab));
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "ragged kernel launch");
	return PIRE_HIP_OK;
}

bool TiledEligible(const ScanParams& p)
{
	return p.offsets == nullptr && p.n >= 64 && p.len >= 128 && (p.stride % 16) == 0 && p.stride * 64 < (1ull << 31) &&
	       (reinterpret_cast<uintptr_t>(p.text) % 16) == 0;
}

int LaunchTiled(const ScanParams& p, hipStream_t stream)
{
	if (int rc = CheckCounts(p))
		return rc;
	// Variant knob for A/B measurements; the default is the measured best.
	static const int variant = [] {
		const char* v = getenv("PIRE_HIP_TILED_VARIANT");
		return v ? atoi(v) : 0;
	}();
	static const bool noload = getenv("PIRE_HIP_DEBUG_NOLOAD") != nullptr;
	ScanParams q = p;
	static const bool nostep = getenv("PIRE_HIP_DEBUG_NOSTEP") != nullptr;
	if (noload)
		q.flags |= kDebugNoRefill;
	if (nostep)
		q.flags |= kDebugNoStep;
	if (getenv("PIRE_HIP_DEBUG_NOCOLDCOUNT"))
		q.flags |= kDebugNoColdCount;
	if (getenv("PIRE_HIP_DEBUG_NOHIST"))
		q.flags |= kDebugNoHist;
	if (variant == 1)
		q.compact = 0;   // the compact rows hold LDS addresses of the 256-byte-pitch layout
	q.n = p.n & ~uint64_t(63);   // whole 64-string tasks; the remainder goes to the generic kernel below
	int rc;
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, kRotPitch, CompactBytes(q));
	const LdsLayout L256 = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(q));
	switch (variant) {
	case 1:  rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 1>, q, 1024, L.total, stream); break;   // bank-rotated rows
	case 2:  rc = LaunchScan(ScanTiledKernel<16, 2, false, 5, 0>, q, 1024, L256.total, stream); break; // no nt
	default: rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 0>, q, 1024, L256.total, stream); break;
	}
	if (rc != PIRE_HIP_OK || q.n == p.n)
		return rc;
	ScanParams tail = p;
	tail.flags &= ~(kDebugNoRefill | kDebugNoStep | kDebugNoColdCount | kDebugNoHist);
	tail.n = p.n - q.n;
	tail.text = p.text + q.n * p.stride;
	if (p.initIdx)
		tail.initIdx = p.initIdx + q.n;
	if (p.outIdx)
		tail.outIdx = p.outIdx + q.n;
	if (p.outFinal)
		tail.outFinal = p.outFinal + q.n;
	return LaunchGeneric(tail, stream);
}

int LaunchPrefix(const ScanParams& p0, bool longest, bool throughEnd, long long* outLen, hipStream_t stream)
{
	if (p0.n == 0)
		return PIRE_HIP_OK;
	ScanParams p = p0;
	p.compact = 0;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(PrefixKernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, int(L.total));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	PrefixParams q;
	q.scan = p;
	q.longest = longest ? 1 : 0;
	q.throughEnd = throughEnd ? 1 : 0;
	q.outLen = outLen;
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + 255) / 256, uint64_t(cus) * 2)));
	hipLaunchKernelGGL(PrefixKernel, dim3(blocks), dim3(256), L.total, stream, q);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "prefix kernel launch");
	return PIRE_HIP_OK;
}

int LaunchHalfFinal(const ScanParams& p0, uint32_t* outResults, hipStream_t stream)
{
	if (p0.n == 0)
		return PIRE_HIP_OK;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	ScanParams p = p0;
	p.compact = 0;
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	const uint32_t ldsBytes = L.total + 256 * 8;
	const bool packed = p.incPerm != nullptr;
	const void* fn = packed ? reinterpret_cast<const void*>(HalfFinalKernel<true>) : reinterpret_cast<const void*>(HalfFinalKernel<false>);
	hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + 255) / 256, uint64_t(cus) * 2)));
	if (packed)
		hipLaunchKernelGGL(HalfFinalKernel<true>, dim3(blocks), dim3(256), ldsBytes, stream, p, outResults);
	else
		hipLaunchKernelGGL(HalfFinalKernel<false>, dim3(blocks), dim3(256), ldsBytes, stream, p, outResults);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "half-final kernel launch");
	return PIRE_HIP_OK;
}
int LaunchStep(const ScanParams& p, uint32_t* stateIdx, uint64_t n, uint32_t cls, hipStream_t stream)
{
	if (n == 0)
		return PIRE_HIP_OK;
	const unsigned blocks = unsigned((n + 255) / 256);
	hipLaunchKernelGGL(StepKernel, dim3(blocks), dim3(256), 0, stream, p, stateIdx, n, cls);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "step kernel launch");
	return PIRE_HIP_OK;
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
