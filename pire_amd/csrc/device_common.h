// Device-side pieces shared by the scan kernels (tiled.hip, ragged.hip, exact.hip): the LDS image of the table, the
// exact step, the 16-byte chunk walk with its trap / compact / full re-walk, end-of-string bookkeeping and the launch
// helpers.  Everything here is inline: the file is included by several translation units.
//
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

#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "internal.h"

namespace pirehip {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Timing-experiment knobs (walk stale registers, skip the walk, skip the end-of-string work ...): they make a kernel
// return WRONG results, so they exist only in a -DPIRE_HIP_TUNING build (tools/ab).  In the product library the masks
// are 0 and every test of them folds away.
#ifdef PIRE_HIP_TUNING
constexpr uint32_t kDebugNoRefill = 1u << 30;
constexpr uint32_t kDebugNoStep = 1u << 29;
constexpr uint32_t kDebugNoColdCount = 1u << 28;
constexpr uint32_t kDebugNoHist = 1u << 27;
constexpr uint32_t kDebugNoPartial = 1u << 26;
constexpr uint32_t kDebugNoFinish = 1u << 25;
constexpr uint32_t kDebugNoTrap = 1u << 24;
constexpr uint32_t kDebugNoTranspose = 1u << 22;
#else
constexpr uint32_t kDebugNoRefill = 0, kDebugNoStep = 0, kDebugNoColdCount = 0, kDebugNoHist = 0, kDebugNoPartial = 0,
                   kDebugNoFinish = 0, kDebugNoTrap = 0, kDebugNoTranspose = 0;
#endif
// internal (tiled.hip): tasks go round the blocks before they go round a block's waves -- set by the launcher for batches with
// fewer tasks than wave slots, which then put a few waves on every CU instead of sixteen on some
constexpr uint32_t kSpreadTasks = 1u << 21;
constexpr uint32_t kPermIds = 1u << 23;   // internal (segmented.hip): initIdx holds, outIdx receives, DEVICE state ids

// Block-wide copy of `count16` 16-byte units from global memory to LDS with up to BATCH loads per thread in flight
// before the first store: a copy loop that waits for every single load pays one memory latency per iteration, and at
// kernel start (cold caches, 256 blocks asking at once) that added up to tens of microseconds per launch.
template <int BATCH>
__device__ __forceinline__ void CopyToLds16(uint8_t* dst, const void* src, uint32_t count16)
{
	const u32x4* s = reinterpret_cast<const u32x4*>(src);
	u32x4* d = reinterpret_cast<u32x4*>(dst);
	for (uint32_t base = threadIdx.x; base < count16; base += blockDim.x * BATCH) {
		u32x4 v[BATCH];
#pragma unroll
		for (int k = 0; k < BATCH; ++k) {
			const uint32_t i = base + uint32_t(k) * blockDim.x;
			if (i < count16)
				v[k] = s[i];
		}
#pragma unroll
		for (int k = 0; k < BATCH; ++k) {
			const uint32_t i = base + uint32_t(k) * blockDim.x;
			if (i < count16)
				d[i] = v[k];
		}
	}
}

// Cooperative load of the LDS-resident part of the table.
__device__ inline void LoadTableToLds(const ScanParams& p, uint8_t* lds, const LdsLayout& L)
{
	const uint32_t tid = threadIdx.x, nthr = blockDim.x;
	// the small pieces: fetched first, stored last, so that their latency hides behind the big copies
	uint32_t flagsWord = 0, clsWord = 0, cls8 = 0;
	if (tid < 256 / 4)
		flagsWord = reinterpret_cast<const uint32_t*>(p.hotFlags)[tid];
	if (tid < 264 / 2)
		clsWord = reinterpret_cast<const uint32_t*>(p.cls)[tid];
	if (p.compact && tid < 256)
		cls8 = p.cls[tid];
	if (L.pitch == 256) {
		CopyToLds16<8>(lds, p.hotRows, (p.hot + 1) * 16);
	} else {
		// 260-byte pitch (bank-rotated variant): 256-byte rows in HBM; 16-byte loads, four in flight per thread, dword
		// stores (the rows are only 4-byte aligned in LDS).  Rounds 1-2 copied dword by dword, one load in flight: 25-45 us
		// for a 256-thread block, most of the time of a small call (kernel trace, round 3).
		const u32x4* src = reinterpret_cast<const u32x4*>(p.hotRows);
		uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
		const uint32_t pitchDw = L.pitch / 4;
		const uint32_t nvec = (p.hot + 1) * 16;
		for (uint32_t base = tid; base < nvec; base += nthr * 4) {
			u32x4 v[4];
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k)
				if (base + k * nthr < nvec)
					v[k] = src[base + k * nthr];
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t i = base + k * nthr;
				if (i < nvec) {
					uint32_t* d = dst + (i >> 4) * pitchDw + (i & 15) * 4;
					d[0] = v[k].x;
					d[1] = v[k].y;
					d[2] = v[k].z;
					d[3] = v[k].w;
				}
			}
		}
	}
	if (p.compact)
		CopyToLds16<8>(lds + L.compactOff, p.compactRows, L.compactBytes / 16);
	if (tid < 256 / 4)
		reinterpret_cast<uint32_t*>(lds + L.flagsOff)[tid] = flagsWord;
	if (tid < 264 / 2)
		reinterpret_cast<uint32_t*>(lds + L.clsOff)[tid] = clsWord;
	if (p.compact && tid < 256)
		lds[L.cls8Off + tid] = uint8_t(2 * cls8);
	if (p.outCounts)
		for (uint32_t i = tid; i < p.regexps + 2; i += nthr)
			reinterpret_cast<uint32_t*>(lds + L.countsOff)[i] = 0;
	for (uint32_t i = tid; i < 256 + 4; i += nthr)   // the visit samples and, behind them, the progress counter
		reinterpret_cast<uint32_t*>(lds + L.histOff)[i] = 0;
	__syncthreads();
}

// The exact step for any state: multi.h:169-192 on the perm-numbered table.
__device__ __forceinline__ uint32_t SlowStep(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                             uint32_t st, uint32_t byte)
{
	if (st < p.hot) {
		const uint32_t e = lds[st * L.pitch + (L.rot2 ? RotColumn(byte) : byte)];
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

// Sixteen bytes through the dense rows alone, from a state that has one: the state after them and the largest id on
// the way (the trap id p.hot, if the walk left the dense rows -- its row is absorbing).  The one-string-per-lane exact
// kernels use it the way the ragged kernel's walks with actions do: the dense rows are ordered plain,
// Dead, Final, so "largest id < threshold" says that none of the 16 steps needs a closer look.
__device__ __forceinline__ uint32_t DenseChunk(const uint8_t* lds, const LdsLayout& L, const u32x4 v, uint32_t& st)
{
	uint32_t mx = 0;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			st = lds[st * L.pitch + ((x >> (8 * b)) & 0xFFu)];
			mx = mx > st ? mx : st;
		}
	}
	return mx;
}

// The same for bytes [skip, skip + count) of a block (a string's ragged first or last block).
__device__ __forceinline__ uint32_t DenseBytes(const uint8_t* lds, const LdsLayout& L, u32x4 v, uint32_t skip, uint32_t count,
                                               uint32_t& st)
{
	for (uint32_t i = 0; i < skip; ++i) {
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	uint32_t mx = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < count; ++i) {
		st = lds[st * L.pitch + (v.x & 0xFFu)];
		mx = mx > st ? mx : st;
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return mx;
}

// Final(state) on a perm id: the hot set is ordered non-final first, so a hot state is Final iff id >= hotFinalLo.
__device__ __forceinline__ bool IsFinalState(const ScanParams& p, uint32_t st)
{
	return st >= p.hotFinalLo && (st < p.hot || (p.flagsPerm[st] & kFinal));
}

// Row flags (kFinal | kDead) of a perm id: LDS for the dense rows, memory otherwise.
__device__ __forceinline__ uint32_t StateFlags(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint32_t st)
{
	return st < p.hot ? lds[L.flagsOff + st] : p.flagsPerm[st];
}

// Start state of string s (perm id): Initialize() or the caller's resume state, then Begin() if asked.
__device__ __forceinline__ uint32_t StartStateFrom(const ScanParams& p, uint32_t init)
{
	init = init < p.states ? init : 0;   // a resume index the scanner does not have must not read outside the table
	uint32_t st = (p.flags & kPermIds) ? init : p.permOfOrig[init];
	if (p.flags & PIRE_HIP_RUN_BEGIN)
		st = p.nextPerm[size_t(st) * p.letters + p.beginCls];
	return st;
}

__device__ __forceinline__ uint32_t StartState(const ScanParams& p, uint64_t s)
{
	if (!p.initIdx)
		return p.startPerm;   // host folded Initialize()+Begin() into one id
	return StartStateFrom(p, p.initIdx[s]);
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
			p.outIdx[s] = (p.flags & kPermIds) ? endPerm : orig;
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

// `tid`: the caller's thread index (a kernel that is short of registers recomputes it instead of keeping it alive).
__device__ inline void FlushCounts(const ScanParams& p, uint8_t* lds, const LdsLayout& L, uint32_t tid)
{
	__syncthreads();
	const uint32_t* hist = reinterpret_cast<const uint32_t*>(lds + L.histOff);
	for (uint32_t i = tid; i < 256; i += blockDim.x)
		if (hist[i])
			atomicAdd(&p.visitHot[i], hist[i]);
	if (tid == 0 && hist[kLdsTrapSlot]) {
		// traps seen by this block: device total, and its new value into the host-visible word (internal.h DeviceTable)
		const uint32_t total = atomicAdd(&p.visitHot[kTrapSlot], hist[kLdsTrapSlot]) + hist[kLdsTrapSlot];
		if (p.trapSignal)
			__hip_atomic_store(p.trapSignal, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (!p.outCounts)
		return;
	const uint32_t* cnt = reinterpret_cast<const uint32_t*>(lds + L.countsOff);
	for (uint32_t i = tid; i < p.regexps + 2; i += blockDim.x)
		if (cnt[i])
			atomicAdd(&p.outCounts[i], (unsigned long long)cnt[i]);
}

__device__ inline void FlushCounts(const ScanParams& p, uint8_t* lds, const LdsLayout& L)
{
	FlushCounts(p, lds, L, threadIdx.x);
}


// ---- the 16-byte chunk: LDS fast path, trap, compact re-walk, full re-walk -----------------------------------------

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

// A lane left the dense rows somewhere in the chunk `v` (hs == p.hot after it): exact re-walk from the chunk's start
// state, through the compact rows in LDS when the state has one, through the full table in HBM when that escapes too.
__device__ __forceinline__ void TrapChunk(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const u32x4 v,
                                          uint32_t hs0, uint32_t& hs, uint32_t& cold, uint32_t sampleLane)
{
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
	}
	// Rare path: tell pire_hip_table_adapt() which rows deserve LDS.  SAMPLED (one rotating lane of 64): un-sampled, the
	// device-scope atomics of every trapped lane serialised on a few dozen addresses and cost 4x the whole kernel (measured:
	// 0.80 -> 3.45 ms on set_a).  WHAT is told (round 6): the state IN FRONT of one step of the chunk, the step
	// drawn from the chunk's place and the block's number, walked to exactly -- if it has no dense row.  (Rounds 1-5 told the
	// state the chunk ENDS in, if that has none: a state the walk passes through on its way back into the rows was never
	// seen, nor one that lives at a fixed offset of every record -- 4.0 % of the lookups of URL records outside the 255 rows
	// where 0.5 % need be, tools/ranking_quality_records.py.)  The trap signal counts chunks that end outside the rows, as before.
	if ((threadIdx.x & 63) == sampleLane && !(p.flags & kDebugNoColdCount)) {
		const uint32_t step = ((sampleLane + blockIdx.x * 0x632BE5ABu) * 0x9E3779B1u) >> 28;
		uint32_t s = st0;
		u32x4 w = v;
#pragma unroll 1
		for (uint32_t i = 0; i < step; ++i) {
			s = SlowStep(p, lds, L, s, w.x & 0xFF);
			w.x = __builtin_amdgcn_alignbit(w.y, w.x, 8);
			w.y = __builtin_amdgcn_alignbit(w.z, w.y, 8);
			w.z = __builtin_amdgcn_alignbit(w.w, w.z, 8);
			w.w >>= 8;
		}
		if (s >= p.hot)
			atomicAdd(&p.visitCold[s], 1u);
		if (f >= p.hot)
			atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + kLdsTrapSlot, 1u);
	}
}

// 16 bytes (one dwordx4) through the hot table; lanes that leave the hot set are re-walked exactly.
template <int ROT>
__device__ __forceinline__ void StepChunk(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                          const u32x4 v, uint32_t& hs, uint32_t& cold, uint32_t sampleLane)
{
	const uint32_t hs0 = hs;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		uint32_t x = v[w];
		if (ROT == 2)   // rotated columns: every byte of the word rotated left by 2 bits (3 VALU per 4 bytes, off the chain)
			x = ((x << 2) & 0xFCFCFCFCu) | ((x >> 6) & 0x03030303u);
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
	if (hs == p.hot && !(p.flags & kDebugNoTrap))
		TrapChunk(p, lds, L, v, hs0, hs, cold, sampleLane);
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

// ---- 8x8 transpose of a register tile across each group of 8 lanes (DPP only, no LDS) ------------------------------
// Before: lane 8g+k holds in register j chunk k of string 8g+j (whole-line loads: 8 adjacent lanes cover one line).
// After: lane 8g+j holds chunks 0..7 of its own string in registers 0..7.
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

// ---- window loads of the offset-batch kernels (ragged.hip, stream.hip) ------------------------------------------------
// Both kernels fetch, per lane and iteration, one window of 128 bytes at an address of the lane's own.
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

// The same 64 windows fetched by GROUPS of 8 lanes: instruction j loads, in every group, the window of lane 8g+j --
// lane 8g+c reads its bytes [16c, 16c+16) -- so that an instruction touches the 1-2 cache lines of 8 windows instead of
// one line of each of 64 strings (and of 64 different ones again 16 bytes further): the L1 tag pipeline was the binding
// unit of this kernel (profiles/r02_ragged_pmc_*_before.txt: 0.98 L1 accesses per clock and CU on URLs, HBM traffic
// 1.6-1.8 x the text because lines were evicted between the 8 instructions that touched them).  The tile then holds,
// in lane 8g+c, register j = chunk c of the window of lane 8g+j: TransposeTile() (the tiled kernel's) puts every
// lane's own window into its registers 0..7, and the walk is what it was.
// `src` = this lane's window address (any alignment); the addresses travel inside the group by DPP (GroupBroadcast).
// (Tried: per-lane loads and no transpose in the iterations in which every lane of the wave is in the middle of a long
// string -- one whole line per lane is the better pattern there: fixed 4 KiB strings 3.3 against 2.8 TB/s.  A transpose
// under a wave-uniform branch made hipcc spill 170-320 bytes per lane, tile registers included; not kept.  Fixed-length
// batches belong to the tiled kernel anyway.)
#ifndef PIRE_HIP_RAGGED_BPERMUTE
// lane j of every group of 8 lanes, to all 8 lanes of its group: quad broadcast, then the other quad of the group copies
// it (row_shr:4 into banks 1 and 3, or row_shl:4 into banks 0 and 2) -- VALU only, no trip through the LDS queue
template <int J>
__device__ __forceinline__ uint32_t GroupBroadcast(uint32_t x)
{
	constexpr int q = J & 3, quad = q | (q << 2) | (q << 4) | (q << 6);
	const int y = __builtin_amdgcn_update_dpp(0, int(x), quad, 0xF, 0xF, false);
	if constexpr (J < 4)
		return uint32_t(__builtin_amdgcn_update_dpp(y, y, 0x114, 0xF, 0xA, false));   // row_shr:4 -> quads 1, 3
	else
		return uint32_t(__builtin_amdgcn_update_dpp(y, y, 0x104, 0xF, 0x5, false));   // row_shl:4 -> quads 0, 2
}
#endif

template <int J>
__device__ __forceinline__ void IssueTileGroupOne(u32x4& r, uint32_t lo, uint32_t hi, uint32_t sel, uint32_t mine)
{
#ifndef PIRE_HIP_RAGGED_BPERMUTE
	const uint32_t l = GroupBroadcast<J>(lo);
	const uint32_t h = GroupBroadcast<J>(hi);
#else   // A/B: through the LDS queue, behind the other waves' table lookups (1-3 % slower, profiles/r02_ragged_clocks.log)
	const uint32_t l = uint32_t(__builtin_amdgcn_ds_bpermute(int(sel + 4 * J), int(lo)));
	const uint32_t h = uint32_t(__builtin_amdgcn_ds_bpermute(int(sel + 4 * J), int(hi)));
#endif
	const uint64_t a = ((uint64_t(h) << 32) | l) + mine;
	asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r) : "v"(a));
}

// (Tried: lanes whose chunk of a window lies behind the string's end ask for the window's first chunk instead -- a third
// fewer distinct chunks on URLs, 1-4 % SLOWER; consecutive strings to the lanes of one load instruction instead of to
// consecutive lanes -- half the cache lines per instruction, no difference.  The load path of this kernel is not bound
// by lines or requests: a window is simply on its way for 0.9 iterations, section clocks in profiles/r02_ragged_clocks.log.)
__device__ __forceinline__ void IssueTileGroup(u32x4 (&r)[8], uint64_t src, uint32_t lane)
{
	const uint32_t lo = uint32_t(src), hi = uint32_t(src >> 32);
	const uint32_t sel = (lane & ~7u) << 2;      // byte index of lane 8g for ds_bpermute
	const uint32_t mine = (lane & 7u) << 4;      // this lane's 16 bytes of every window of its group
	IssueTileGroupOne<0>(r[0], lo, hi, sel, mine);
	IssueTileGroupOne<1>(r[1], lo, hi, sel, mine);
	IssueTileGroupOne<2>(r[2], lo, hi, sel, mine);
	IssueTileGroupOne<3>(r[3], lo, hi, sel, mine);
	IssueTileGroupOne<4>(r[4], lo, hi, sel, mine);
	IssueTileGroupOne<5>(r[5], lo, hi, sel, mine);
	IssueTileGroupOne<6>(r[6], lo, hi, sel, mine);
	IssueTileGroupOne<7>(r[7], lo, hi, sel, mine);
}

__device__ __forceinline__ void WaitAllLoads(u32x4 (&r)[8])
{
	asm volatile("s_waitcnt vmcnt(0)"
	             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}


// ---- end of a string for the offset-batch kernels -------------------------------------------------------------------
// The end-of-string record of state `st`: LDS for a dense-row state, memory for any other.
__device__ __forceinline__ u32x4 FinRecordOf(const ScanParams& p, const FinRec* finHot, bool active, uint32_t st)
{
	// The record of a dense-row state comes from LDS, the record of any other state from memory -- under a wave-uniform
	// branch of its own, and waited for inside it.  Written as one if/else per lane the compiler merges the two into a
	// single FLAT load through a selected pointer, and the vmcnt(0) a flat load needs drained the window prefetched for
	// the next iteration every time a string ended: a third of the URL batch's time (profiles/r02_ragged_clocks.log).
	const bool cold = active && st >= p.hot;
	u32x4 raw = {0, 0, 0, 0};
	if (active && !cold)
		raw = *reinterpret_cast<const u32x4*>(&finHot[st]);
	if (__any(cold)) {
		if (cold) {
			const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
			raw = *reinterpret_cast<const u32x4*>(&recs[st]);
			asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));   // the wait belongs in here
		}
	}
	return raw;
}

template <bool EXT>
__device__ __forceinline__ void FinishWith(const ScanParams& p, uint8_t* lds, const LdsLayout& L, uint32_t s, bool active,
                                           const u32x4 raw);

template <bool EXT>
__device__ __forceinline__ void FinishRagged(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                             uint32_t s, bool active, uint32_t st)
{
	FinishWith<EXT>(p, lds, L, s, active, FinRecordOf(p, finHot, active, st));
}

// outputs and block-local counters of one finished string per lane, from its end-of-string record
template <bool EXT>
__device__ __forceinline__ void FinishWith(const ScanParams& p, uint8_t* lds, const LdsLayout& L, uint32_t s, bool active,
                                           const u32x4 raw)
{
	const uint32_t orig = raw.x, endPerm = raw.y & 0x0FFFFFFFu, fl = raw.y >> 28;
	if (active) {
		if (p.outIdx)
			p.outIdx[s] = (EXT && (p.flags & kPermIds)) ? endPerm : orig;
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


// ---- the same transpose cut into 24 SLOTS of four instructions (round 3) ---------------------------------------------
// A wave spends 6-9 % of its time transposing (112 VALU + waits per tile during which its lookup chain stands still:
// walk alone 0.527 ms, walk + transposes 0.586 ms).  The lookup chain leaves the wave idle for the ~64+
// cycles of every ds_read_u8, so the transpose of the NEXT tile is issued in those shadows: one slot after each of the
// last 24 lookups of the tile being walked (StepChunkShadow).  A slot is one half of one butterfly stage of one dword
// column: four independent v_cndmask_b32_dpp under one mask.  Slot S: column S / 6, stage (S % 6) / 2 (lane bit 0, 1,
// 2), half S & 1 (0: the x' = lo ? x : perm(y) outputs, kept in tmp[] because the other half still reads the old x;
// 1: the y' outputs, after which x', y' replace x, y).  Stage 2 (lane bit 2) is the same fused select with
// row_shr:4 / row_shl:4 and bound_ctrl:0 (lanes without a source read 0 and are the ones that select the other operand).
#define PIRE_SLOT_ASM(PERM)                                                                                              \
	asm volatile("s_nop 1\n\t"                                                                                       \
	             "s_mov_b64 vcc, %12\n\t"                                                                              \
	             "v_cndmask_b32_dpp %0, %4, %5, vcc " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
	             "v_cndmask_b32_dpp %1, %6, %7, vcc " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
	             "v_cndmask_b32_dpp %2, %8, %9, vcc " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                \
	             "v_cndmask_b32_dpp %3, %10, %11, vcc " PERM " row_mask:0xf bank_mask:0xf bound_ctrl:0"                   \
	             : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)                                                           \
	             : "v"(m0), "v"(k0), "v"(m1), "v"(k1), "v"(m2), "v"(k2), "v"(m3), "v"(k3), "s"(mask)                      \
	             : "vcc")

// o = mask ? k : PERM(m)   for four (m, k) pairs
template <int STAGE, bool UP>
__device__ __forceinline__ void SlotSelect(uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3, uint32_t m0, uint32_t k0,
                                           uint32_t m1, uint32_t k1, uint32_t m2, uint32_t k2, uint32_t m3, uint32_t k3,
                                           uint64_t mask)
{
	if (STAGE == 0)
		PIRE_SLOT_ASM("quad_perm:[1,0,3,2]");
	else if (STAGE == 1)
		PIRE_SLOT_ASM("quad_perm:[2,3,0,1]");
	else if (!UP)
		PIRE_SLOT_ASM("row_shr:4");   // lane l reads lane l - 4
	else
		PIRE_SLOT_ASM("row_shl:4");   // lane l reads lane l + 4
}

constexpr int kTransposeSlots = 24;

template <int S>
__device__ __forceinline__ void TransposeSlot(u32x4 (&r)[8], uint32_t (&tmp)[4])
{
	static_assert(S >= 0 && S < kTransposeSlots, "slot");
	constexpr int w = S / 6, stage = (S % 6) / 2, half = S & 1;
	constexpr int D = 1 << stage;
	// the four pairs (x = register k, y = register k | D) of this stage, k with bit D clear, in increasing k
	constexpr int x0 = (D == 1) ? 0 : (D == 2) ? 0 : 0, x1 = (D == 1) ? 2 : (D == 2) ? 1 : 1,
	              x2 = (D == 1) ? 4 : (D == 2) ? 4 : 2, x3 = (D == 1) ? 6 : (D == 2) ? 5 : 3;
	constexpr uint64_t lo = (D == 1) ? 0x5555555555555555ull : (D == 2) ? 0x3333333333333333ull : 0x0F0F0F0F0F0F0F0Full;
	if (half == 0) {
		// x' = (lane bit D clear) ? x : y[lane ^ D]      (the lanes with the bit set read the lane D below)
		SlotSelect<stage, false>(tmp[0], tmp[1], tmp[2], tmp[3], r[x0 | D][w], r[x0][w], r[x1 | D][w], r[x1][w], r[x2 | D][w],
		                         r[x2][w], r[x3 | D][w], r[x3][w], lo);
	} else {
		// y' = (lane bit D set) ? y : x[lane ^ D]        (the lanes with the bit clear read the lane D above)
		uint32_t n0, n1, n2, n3;
		SlotSelect<stage, true>(n0, n1, n2, n3, r[x0][w], r[x0 | D][w], r[x1][w], r[x1 | D][w], r[x2][w], r[x2 | D][w],
		                        r[x3][w], r[x3 | D][w], ~lo);
		r[x0][w] = tmp[0]; r[x1][w] = tmp[1]; r[x2][w] = tmp[2]; r[x3][w] = tmp[3];
		r[x0 | D][w] = n0; r[x1 | D][w] = n1; r[x2 | D][w] = n2; r[x3 | D][w] = n3;
	}
}

// The whole transpose through the slots, back to back (the prologue of a wave's ring, and the reference the tests of
// the shadowed form are checked against: it must equal TransposeTile).
template <int S = 0>
__device__ __forceinline__ void TransposeBySlots(u32x4 (&r)[8], uint32_t (&tmp)[4])
{
	if constexpr (S < kTransposeSlots) {
		TransposeSlot<S>(r, tmp);
		TransposeBySlots<S + 1>(r, tmp);
	}
}

// StepChunk with transpose slots FIRST, FIRST + 1, ... of `next` issued in the shadows of lookups SKIP .. 15 of this
// chunk: lookup, then -- while the LDS round trip is under way -- four independent VALU instructions on registers the
// walk does not touch.  WAIT: before the first slot the wave waits for `next` to have landed (requested most of a
// tile-time earlier).  The sched_barriers pin the order "ds_read, slot, s_waitcnt": nothing may drift between a
// lookup and its wait except the slot.
template <int I, int FIRST, int SKIP, bool WAIT>
__device__ __forceinline__ void ShadowLookups(const u32x4 v, uint32_t& hs, u32x4 (&next)[8], uint32_t (&tmp)[4])
{
	if constexpr (I < 16) {
		constexpr uint32_t sel = 0x0c0c0400u + (I & 3);
		hs = HotLookup(__builtin_amdgcn_perm(hs, v[I >> 2], sel));
		constexpr int slot = FIRST + I - SKIP;
		if constexpr (I >= SKIP && slot < kTransposeSlots) {
			__builtin_amdgcn_sched_barrier(0);
			if constexpr (WAIT && I == SKIP)
				asm volatile("s_waitcnt vmcnt(0)"
				             : "+v"(next[0]), "+v"(next[1]), "+v"(next[2]), "+v"(next[3]), "+v"(next[4]), "+v"(next[5]),
				               "+v"(next[6]), "+v"(next[7]));
			TransposeSlot<slot>(next, tmp);
			__builtin_amdgcn_sched_barrier(0);
		}
		ShadowLookups<I + 1, FIRST, SKIP, WAIT>(v, hs, next, tmp);
	}
}

template <int FIRST, int SKIP, bool WAIT>
__device__ __forceinline__ void StepChunkShadow(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const u32x4 v,
                                                uint32_t& hs, uint32_t& cold, uint32_t sampleLane, u32x4 (&next)[8],
                                                uint32_t (&tmp)[4])
{
	const uint32_t hs0 = hs;
	ShadowLookups<0, FIRST, SKIP, WAIT>(v, hs, next, tmp);
	if (hs == p.hot)
		TrapChunk(p, lds, L, v, hs0, hs, cold, sampleLane);
}

// One pipeline phase of the register ring: refill the slot that was freed one phase ago with the tile NBUF-1
// ahead (index clamped to the last tile, so the steady-state loop has no conditional loads), wait until the
// current slot has landed, transpose it into lane-owns-string order, walk it.

// ---- launch helpers --------------------------------------------------------------------------------------------------

inline int DeviceCUs(int* cus)
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
hipError_t CachedOccupancy(int* perCu, K kernel, int threads, uint32_t ldsBytes)
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return e;
	const LaunchCache::Key key = {dev, reinterpret_cast<const void*>(kernel), threads, ldsBytes};
	LaunchCache& c = LaunchCache::Get();
	{
		std::lock_guard<std::mutex> lock(c.mutex);
		for (auto& x : c.occupancy)
			if (x.key == key) {
				*perCu = x.value;
				return hipSuccess;
			}
	}
	e = hipOccupancyMaxActiveBlocksPerMultiprocessor(perCu, kernel, threads, ldsBytes);
	if (e != hipSuccess)
		return e;
	std::lock_guard<std::mutex> lock(c.mutex);
	if (c.occupancy.size() < 4096)
		c.occupancy.push_back({key, *perCu});
	return hipSuccess;
}

template <class K>
int LaunchScan(K kernel, const ScanParams& p, int threads, uint32_t ldsBytes, hipStream_t stream, int tasksPerBlock = 0)
{
	int cus = 0;
	int rc = DeviceCUs(&cus);
	if (rc)
		return rc;
	hipError_t e = SetDynamicLds(reinterpret_cast<const void*>(kernel), uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	int perCu = 0;
	e = CachedOccupancy(&perCu, kernel, threads, ldsBytes);
	if (e != hipSuccess)
		return HipFail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
#ifdef PIRE_HIP_TUNING
	if (getenv("PIRE_HIP_DEBUG_LAUNCH"))
		fprintf(stderr, "pire_hip: threads %d lds %u -> %d blocks/CU\n", threads, ldsBytes, perCu);
#endif
	if (perCu < 1)
		perCu = 1;
#ifdef PIRE_HIP_TUNING
	if (const char* cap = getenv("PIRE_HIP_BLOCKS_PER_CU"))   // knob: A/B measurements
		perCu = std::max(1, std::min(perCu, atoi(cap)));
#endif
	const uint64_t ntasks = (p.n + 63) / 64;
	const uint64_t wavesPerBlock = tasksPerBlock ? uint64_t(tasksPerBlock) : uint64_t(threads) / 64;
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

inline int CheckCounts(const ScanParams& p)
{
	if (p.outCounts && p.regexps > kMaxLdsCountRegexps) {
		SetError("out_counts is supported for scanners with at most 1024 regexps");
		return PIRE_HIP_EUNSUPPORTED;
	}
	return PIRE_HIP_OK;
}


}  // namespace pirehip
