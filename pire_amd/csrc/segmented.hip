// Long strings: a DFA walk is sequential, and one string per lane means that a few long strings leave the chip idle
// (one lane walks ~24 MB/s; a single 1 GiB string would take 45 s).  This file cuts long strings into segments and
// scans the segments in parallel -- speculatively, because the start state of a segment is the end state of the one
// before it -- and then resolves the chain of segments, accepting only what was computed from the state the chain
// is really in.
//
//   guess   Regexp automata mostly forget: after a few dozen bytes the state rarely depends on where the walk began.
//           So a segment guesses its start state by walking the W bytes in front of it (its warm-up) from the
//           string's start state.  What automata do NOT forget are their sticky modes ("hello\s+w" was seen, a
//           non-printable byte was seen): after the first trigger every such guess is wrong.  A MODE is therefore a
//           representative state r: under mode r a segment's guess is the warm-up walked from r.  Mode 0 is the
//           string's start state; further modes are learned from the states the chain finds itself in unexpectedly
//           (two modes cover 99.96-100 % of the segments of the benchmark tables).
//   scan    per mode one batch through the ordinary kernels: every segment from its guess -> its end state.
//   chain   in state `cur` at the start of segment k the chain looks for a mode whose guess for k IS cur and takes
//           that mode's end state -- an exact result, because a DFA step depends on nothing but the state and the
//           bytes.  On the device this is function composition: segment k maps "the mode whose guess is the true
//           state at k" to the same for k+1 (or to "none"); an inclusive scan composes the maps, and every segment
//           reads the mode it is really entered in.  Where the chain breaks (no mode predicted the true state),
//           the host is told the segment and the state: that segment is scanned from that state (all strings'
//           breaks in one small batch) and patched in as a one-segment mode; a state that keeps breaking chains
//           becomes a real mode; when the budget of such round trips is spent the rest of the string is walked the
//           plain way.  So the result is exact whatever the automaton; the speed depends on how well it forgets.
//   finish  End(), outputs and match counters exactly like the other kernels.
//
// Results are the reference's (run.h:271-275 walks the same bytes in the same order); only the schedule differs.
// The call synchronises its stream (the host reads, per round, whether and where chains broke).

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "device_common.h"

namespace pirehip {

namespace {

constexpr uint32_t kMaxModes = 6;          // learned modes; slot kMaxModes is the patch slot
constexpr uint32_t kSlots = kMaxModes + 1; // bytes 0..6 of a map; byte 7 = kResetMap or 0
constexpr uint64_t kResetMap = 1ull << 56; // the map ignores its input (first segment of a string)
constexpr uint32_t kNoSlot = 0xFF;
constexpr uint32_t kNoState = 0xFFFFFFFFu;

struct SegGeometry {
	uint64_t segBytes, warmBytes;
	uint64_t nStrings, nSeg;
	uint64_t segsPerString;        // strided batches: every string has this many segments
	const uint32_t* strFirst;      // offset batches: [nStrings + 1] first segment of every string (device)
};

struct SegArrays {
	uint64_t* segBegin;            // [nSeg] byte offsets into text
	uint64_t* segEnd;
	uint64_t* warmBegin;
	uint32_t* segStr;              // [nSeg] string of the segment
	uint32_t* segJ;                // [nSeg] index of the segment inside its string
	uint32_t* initSeg;             // [nSeg] caller's resume state of the segment's string (only with init states)
};

// guess / end state (DEVICE state ids: the batches run with kPermIds) of every segment under every slot
struct SlotArrays {
	uint32_t* guess[kSlots];
	uint32_t* end[kSlots];
	uint32_t count;                // slots in use below kMaxModes
};

struct ChainArrays {
	uint64_t* prefix;              // [nSeg] maps (slot at k -> slot at k+1, 7 x u8 + kResetMap), composed inclusively
	uint32_t* trueStart;           // nullable, [nSeg]: the state every segment is really entered in
	uint32_t* finalState;          // [nStrings]
	uint32_t* strDone;             // [nStrings] 1: finalState set by the plain walk
	uint32_t* breakSeg;            // [nStrings] first segment no slot predicted (kNoState: none)
	uint32_t* breakState;          // [nStrings] the state the chain is in there
	uint32_t* broken;              // [1]
	// the scan of the maps, two levels (SegmentMapScanKernel): `prefix` is inclusive inside blocks of kChainBlock
	// segments, blockPrefix[b] the inclusive composition of blocks 0..b
	uint64_t* blockAgg;            // [chain blocks]
	uint64_t* blockPrefix;         // [chain blocks]
	uint32_t* brokenHost;          // mapped host word: the kernel behind the resolve kernel copies *broken there
};

__device__ __forceinline__ void StringOfSegment(const SegGeometry& g, uint64_t seg, uint64_t* str, uint64_t* j)
{
	if (!g.strFirst) {
		*str = seg / g.segsPerString;
		*j = seg - *str * g.segsPerString;
		return;
	}
	uint64_t lo = 0, hi = g.nStrings;          // last string whose first segment is <= seg
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) / 2;
		if (g.strFirst[mid] <= seg)
			lo = mid;
		else
			hi = mid;
	}
	*str = lo;
	*j = seg - g.strFirst[lo];
}

__global__ void SegmentPrepKernel(ScanParams p, SegGeometry g, SegArrays a, uint32_t* patchGuess, uint32_t* patchEnd,
                                  uint32_t* strDone, uint32_t* breakSeg, uint32_t* broken)
{
	const uint64_t seg = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (seg >= g.nSeg)
		return;
	// what used to be five fills of their own (every launch costs ~6 us of a call that takes 400): the patch slot is
	// empty, no string is done or broken (every string has a segment: seg < nStrings covers them)
	patchGuess[seg] = kNoState;
	patchEnd[seg] = kNoState;
	if (seg < g.nStrings) {
		strDone[seg] = 0;
		breakSeg[seg] = kNoState;
	}
	if (seg == 0)
		*broken = 0;
	uint64_t s, j;
	StringOfSegment(g, seg, &s, &j);
	uint64_t B, E;
	if (p.offsets) {
		B = p.offsets[s];
		E = p.offsets[s + 1];
	} else {
		B = s * p.stride;
		E = B + p.len;
	}
	const uint64_t b = B + j * g.segBytes;
	const uint64_t e = b + g.segBytes < E ? b + g.segBytes : E;
	const uint64_t w = j == 0 ? 0 : (b - B < g.warmBytes ? b - B : g.warmBytes);
	a.segBegin[seg] = b;
	a.segEnd[seg] = e;
	a.warmBegin[seg] = b - w;
	a.segStr[seg] = uint32_t(s);
	a.segJ[seg] = uint32_t(j);
	if (a.initSeg)
		a.initSeg[seg] = p.permOfOrig[p.initIdx[s]];
}

__global__ void SegmentFillKernel(uint32_t* dst, uint32_t value, uint64_t n)
{
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n)
		dst[i] = value;
}

// The map of segment k: entered in slot i (i.e. in state guess[i][k]) it ends in end[i][k]; which slot of k+1 has
// that state as its guess?  The first segment of a string is entered in slot 0 whatever came before it, so its map
// is constant -- which also restarts the composition at every string.
__device__ __forceinline__ uint64_t MapOfSegment(const SegGeometry& g, const SegArrays& a, const SlotArrays& sl, uint64_t k)
{
	const bool last = k + 1 == g.nSeg || a.segJ[k + 1] == 0;
	uint64_t map = 0;
	if (!last) {
		uint32_t nextGuess[kSlots];
#pragma unroll
		for (uint32_t j = 0; j < kSlots; ++j)
			nextGuess[j] = (j < sl.count || j == kMaxModes) ? sl.guess[j][k + 1] : kNoState;
#pragma unroll
		for (uint32_t i = 0; i < kSlots; ++i) {
			uint32_t to = kNoSlot;
			if (i < sl.count || i == kMaxModes) {
				const uint32_t e = sl.end[i][k];
				const bool live = sl.guess[i][k] != kNoState;   // the patch slot is empty almost everywhere
#pragma unroll
				for (uint32_t j = kSlots; j-- > 0;)
					if (live && nextGuess[j] == e)
						to = j;
			}
			map |= uint64_t(to) << (8 * i);
		}
	}
	if (a.segJ[k] == 0) {
		// entered in slot 0 whatever came before -- even a broken chain of the string in front
		const uint64_t to0 = map & 0xFF;
		map = to0 * 0x0001010101010101ull | kResetMap;
	}
	return map;
}

struct ComposeMaps {
	__host__ __device__ __forceinline__ uint64_t operator()(const uint64_t& first, const uint64_t& then) const
	{
		if (then & kResetMap)
			return then;
		uint64_t out = first & kResetMap;   // a constant map stays constant
#pragma unroll
		for (uint32_t i = 0; i < kSlots; ++i) {
			const uint32_t mid = uint32_t(first >> (8 * i)) & 0xFF;
			const uint32_t to = mid == kNoSlot ? kNoSlot : uint32_t(then >> (8 * mid)) & 0xFF;
			out |= uint64_t(to) << (8 * i);
		}
		return out;
	}
};

constexpr uint32_t kChainBlock = 1024;
constexpr uint64_t kIdentityMap = 0x0006050403020100ull;

// The maps and their inclusive scan in two small launches (round 3; before: a map kernel and hipcub's two): a block
// composes its kChainBlock maps, one block then composes the blocks' aggregates, and the resolve kernel puts the two
// levels together for the one byte of the composition it needs.  (No "last block done" tricks: a release fence per
// block costs an L2 write-back on this chip -- eight L2s -- and made these kernels 25 us each.)
__global__ __launch_bounds__(kChainBlock) void SegmentMapScanKernel(SegGeometry g, SegArrays a, SlotArrays sl, ChainArrays c)
{
	using Scan = hipcub::BlockScan<uint64_t, kChainBlock>;
	__shared__ typename Scan::TempStorage temp;
	const uint64_t k = uint64_t(blockIdx.x) * kChainBlock + threadIdx.x;
	uint64_t map = kIdentityMap;
	if (k < g.nSeg)
		map = MapOfSegment(g, a, sl, k);
	uint64_t incl;
	Scan(temp).InclusiveScan(map, incl, ComposeMaps());
	if (k < g.nSeg)
		c.prefix[k] = incl;
	if (threadIdx.x == kChainBlock - 1)
		c.blockAgg[blockIdx.x] = incl;
}

__global__ __launch_bounds__(kChainBlock) void SegmentBlockScanKernel(ChainArrays c, uint32_t nblocks)
{
	using Scan = hipcub::BlockScan<uint64_t, kChainBlock>;
	__shared__ typename Scan::TempStorage temp;
	__shared__ uint64_t next;
	uint64_t carry = kIdentityMap;
	for (uint32_t base = 0; base < nblocks; base += kChainBlock) {
		const uint32_t b = base + threadIdx.x;
		const uint64_t agg = b < nblocks ? c.blockAgg[b] : kIdentityMap;
		uint64_t in;
		Scan(temp).InclusiveScan(agg, in, ComposeMaps());
		const uint64_t out = ComposeMaps()(carry, in);
		if (b < nblocks)
			c.blockPrefix[b] = out;
		if (threadIdx.x == kChainBlock - 1)
			next = out;
		__syncthreads();
		carry = next;
		__syncthreads();   // temp and next are reused
	}
}

// byte 0 of the inclusive composition up to and including segment k: the slot segment k+1 is entered in
__device__ __forceinline__ uint32_t ChainSlotAfter(const ChainArrays& c, uint64_t k)
{
	const uint64_t local = c.prefix[k];
	const uint64_t b = k / kChainBlock;
	if (b == 0 || (local & kResetMap))
		return uint32_t(local) & 0xFF;
	const uint32_t mid = uint32_t(c.blockPrefix[b - 1]) & 0xFF;
	return mid == kNoSlot ? kNoSlot : uint32_t(local >> (8 * mid)) & 0xFF;
}

// Every segment reads the slot it is really entered in; the first segment of a string nobody predicted reports
// itself; the last segment of a resolved string delivers the string's end state.
__global__ void SegmentResolveKernel(SegGeometry g, SegArrays a, SlotArrays sl, ChainArrays c)
{
	const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (k >= g.nSeg)
		return;
	const uint32_t str = a.segStr[k];
	if (c.strDone[str])
		return;
	const uint32_t j = a.segJ[k];
	const uint32_t slot = j == 0 ? 0u : ChainSlotAfter(c, k - 1);
	if (slot == kNoSlot) {
		const uint32_t before = j == 1 ? 0u : ChainSlotAfter(c, k - 2);
		if (before != kNoSlot) {   // the chain was intact up to segment k-1: this is where it breaks
			c.breakSeg[str] = uint32_t(k);
			c.breakState[str] = sl.end[before][k - 1];
			atomicAdd(c.broken, 1u);
		}
		return;
	}
	if (c.trueStart)
		c.trueStart[k] = sl.guess[slot][k];
	const bool last = k + 1 == g.nSeg || a.segJ[k + 1] == 0;
	if (last)
		c.finalState[str] = sl.end[slot][k];
}

// How many chains broke, for the host: a mapped host word instead of a copy to enqueue.  The finish kernels are
// launched behind the resolve kernel WITHOUT waiting for the host (round 3: one round trip less per call): they report
// the number and do their work only if it is zero -- the usual case; otherwise the host repairs and launches them again.
__device__ __forceinline__ bool ChainsIntact(const uint32_t* broken, uint32_t* brokenHost)
{
	if (!broken)
		return true;   // the host has already looked
	const uint32_t n = *broken;
	if (blockIdx.x == 0 && threadIdx.x == 0)
		__hip_atomic_store(brokenHost, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	return n == 0;
}

__global__ void SegmentReportKernel(const uint32_t* broken, uint32_t* brokenHost)
{
	(void)ChainsIntact(broken, brokenHost);
}

// Patch: segment breakSeg[i] of broken string i was scanned from breakState[i]; enter it as that segment's patch
// slot -- or, after the plain walk to the end of the string, as the string's end state.
__global__ void SegmentPatchKernel(const uint32_t* strings, const uint32_t* results, uint32_t m, uint32_t plain,
                                   SlotArrays sl, ChainArrays c)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m)
		return;
	const uint32_t str = strings[i];
	if (plain) {
		c.finalState[str] = results[i];
		c.strDone[str] = 1;
	} else {
		const uint32_t k = c.breakSeg[str];
		sl.guess[kMaxModes][k] = c.breakState[str];
		sl.end[kMaxModes][k] = results[i];
	}
}

// finish: one lane per string; endIdx[s] = the device state id string s ended in, before End().
__global__ __launch_bounds__(1024) void SegmentFinishKernel(ScanParams p, const uint32_t* endIdx, const uint32_t* broken,
                                                           uint32_t* brokenHost)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	if (!ChainsIntact(broken, brokenHost))
		return;
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, kRotPitch, 0);
	LoadTableToLds(p, lds, L);
	const uint64_t rounds = (p.n + 63) / 64;
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t task = uint64_t(blockIdx.x) * wavesPerBlock + (threadIdx.x >> 6); task < rounds;
	     task += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t s = task * 64 + lane;
		const bool active = s < p.n;
		Finish(p, lds, L, s, active, active ? endIdx[s] : 0u);
	}
	FlushCounts(p, lds, L);
}

// HalfFinalScanner counting (scanners/half_final.h:137-164) over segments whose TRUE start states are known from the
// resolved chain: nothing is speculative here, every segment counts the Final states of its own steps -- Initialize
// and Begin() belong to a string's first segment, End() to its last -- and the counts of a string's segments add up.
// One segment per lane, exact step per byte (the segments are equally long: no lane waits for another).
__global__ __launch_bounds__(1024) void SegmentHalfFinalKernel(ScanParams p, SegGeometry g, SegArrays a, const uint32_t* trueStart,
                                                              const uint32_t* strDone, uint32_t initialPerm, uint32_t* results)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, 256u, 0);   // dense rows at LDS address 0, 256 bytes apart (HotLookup)
	unsigned int* blockSum = reinterpret_cast<unsigned int*>(lds + L.total);   // [8] counts of the block's first string
	if (threadIdx.x < 8)
		blockSum[threadIdx.x] = 0;
	LoadTableToLds(p, lds, L);   // ends with a barrier
	const uint64_t k0 = uint64_t(blockIdx.x) * blockDim.x;
	const uint64_t k = k0 + threadIdx.x;
	const uint32_t firstStr = k0 < g.nSeg ? a.segStr[k0] : 0;
	if (k < g.nSeg && !strDone[a.segStr[k]]) {
		uint32_t c[8];
#pragma unroll
		for (int r = 0; r < 8; ++r)
			c[r] = 0;
		auto take = [&](uint32_t st) {
			if (IsFinalState(p, st)) {
				const uint64_t inc = p.incPerm[st];
#pragma unroll
				for (int r = 0; r < 8; ++r)
					c[r] += uint32_t(inc >> (8 * r)) & 0xFFu;
			}
		};
		uint32_t st = trueStart[k];
		if (a.segJ[k] == 0) {
			take(initialPerm);                                   // Initialize ends with TakeAction, half_final.h:142
			if (p.flags & PIRE_HIP_RUN_BEGIN)
				take(st);                                        // st IS the state after Begin()
		}
		const uint8_t* q = p.text + a.segBegin[k];
		const uint8_t* qe = p.text + a.segEnd[k];
		for (; q < qe && (reinterpret_cast<uintptr_t>(q) & 15); ++q) {
			st = SlowStep(p, lds, L, st, *q);
			take(st);
		}
		// whole chunks: the dense-row fast path keeps the largest id the 16 steps went through; the hot set is ordered
		// Final last and the trap id is the largest of all, so only a chunk that touched a Final state or left the
		// dense rows is walked again, exactly, with the action (the scheme of ragged.hip, "scans with actions")
		{
			uint32_t hs = st < p.hot ? st : p.hot, cold = st;
			u32x4 ahead = q + 16 <= qe ? *reinterpret_cast<const u32x4*>(q) : u32x4{0, 0, 0, 0};
			for (; q + 16 <= qe; q += 16) {
				const u32x4 v = ahead;
				if (q + 32 <= qe)   // the next chunk is on its way while this one is walked
					ahead = *reinterpret_cast<const u32x4*>(q + 16);
				const uint32_t hs0 = hs;
				uint32_t h = hs, m = 0;
#pragma unroll
				for (int w = 0; w < 4; ++w) {
					const uint32_t x = v[w];
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0400u));
					const uint32_t h1 = h;
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0401u));
					m = max(m, max(h1, h));
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0402u));
					const uint32_t h3 = h;
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0403u));
					m = max(m, max(h3, h));
				}
				hs = h;
				if (m >= p.hotFinalLo) {
					uint32_t w = hs0 != p.hot ? hs0 : cold;
					u32x4 u = v;
#pragma unroll 1
					for (int i = 0; i < 16; ++i) {
						w = SlowStep(p, lds, L, w, u.x & 0xFF);
						take(w);
						u.x = __builtin_amdgcn_alignbit(u.y, u.x, 8);
						u.y = __builtin_amdgcn_alignbit(u.z, u.y, 8);
						u.z = __builtin_amdgcn_alignbit(u.w, u.z, 8);
						u.w >>= 8;
					}
					hs = w < p.hot ? w : p.hot;
					cold = w;
				}
			}
			st = hs != p.hot ? hs : cold;
		}
		for (; q < qe; ++q) {
			st = SlowStep(p, lds, L, st, *q);
			take(st);
		}
		const bool last = k + 1 == g.nSeg || a.segJ[k + 1] == 0;
		if (last && (p.flags & PIRE_HIP_RUN_END)) {
			st = p.nextPerm[size_t(st) * p.letters + p.endCls];
			take(st);
		}
		const uint32_t str = a.segStr[k];
#pragma unroll
		for (int r = 0; r < 8; ++r)
			if (uint32_t(r) < p.regexps && c[r]) {
				if (str == firstStr)
					atomicAdd(&blockSum[r], c[r]);
				else
					atomicAdd(&results[size_t(str) * p.regexps + r], c[r]);
			}
	}
	__syncthreads();
	if (threadIdx.x < p.regexps && threadIdx.x < 8 && blockSum[threadIdx.x])
		atomicAdd(&results[size_t(firstStr) * p.regexps + threadIdx.x], blockSum[threadIdx.x]);
}

// The same for a handful of strings without the 66 KB table fill: the end-of-string record straight from memory, the
// counters straight to the caller's array (one atomic per wave and counter).
__global__ __launch_bounds__(256) void SegmentFinishSmallKernel(ScanParams p, const uint32_t* endIdx, const uint32_t* broken,
                                                               uint32_t* brokenHost)
{
	if (!ChainsIntact(broken, brokenHost))
		return;
	const uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	const bool active = s < p.n;
	const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
	u32x4 raw = {0, 0, 0, 0};
	if (active)
		raw = *reinterpret_cast<const u32x4*>(&recs[endIdx[s]]);
	const uint32_t orig = raw.x, endPerm = raw.y & 0x0FFFFFFFu, fl = raw.y >> 28;
	if (active) {
		if (p.outIdx)
			p.outIdx[s] = orig;
		if (p.outFinal)
			p.outFinal[s] = fl & kFinal;
	}
	if (!p.outCounts)
		return;
	const int lane = threadIdx.x & 63;
	const unsigned long long finals = __ballot(active && (fl & kFinal));
	const unsigned long long actives = __ballot(active);
	if (lane == 0 && actives) {
		atomicAdd(&p.outCounts[0], (unsigned long long)__popcll(finals));
		atomicAdd(&p.outCounts[1], (unsigned long long)__popcll(actives));
	}
	if (p.acceptMaskPerm) {
		const uint64_t m = active ? ((uint64_t(raw.w) << 32) | raw.z) : 0;
		if (__any(m != 0))
			for (uint32_t r = 0; r < p.regexps; ++r) {
				const unsigned long long b = __ballot((m >> r) & 1);
				if (lane == 0 && b)
					atomicAdd(&p.outCounts[2 + r], (unsigned long long)__popcll(b));
			}
	} else if (active) {
		for (uint64_t k = p.acceptOffPerm[endPerm]; k < p.acceptOffPerm[endPerm + 1]; ++k)
			atomicAdd(&p.outCounts[2 + p.acceptIds[k]], 1ull);
	}
}

// a pire_hip_config field: 0 = the default, the *_NONE sentinel = really zero
uint64_t Knob(uint64_t v, uint64_t fallback)
{
	return v == 0 ? fallback : v == ~uint64_t(0) ? 0 : v;
}

// Stream-ordered scratch: ONE allocation per call, made while the stream is still idle (allocating from the pool
// with kernels in flight cost about a millisecond per call), carved up as the call goes, freed on the stream when
// the call returns, i.e. after everything enqueued before.
// One mapped host word out of the pinned staging cache (api.cpp), for the duration of a call.
struct HostWord {
	void* host = nullptr;
	uint32_t* dev = nullptr;
	size_t block = 0;
	int Acquire()
	{
		if (int rc = StagingAcquireHost(64, &host, &block))
			return rc;
		void* d = nullptr;
		const hipError_t e = hipHostGetDevicePointer(&d, host, 0);
		if (e != hipSuccess)
			return HipFail(e, "hipHostGetDevicePointer");
		dev = static_cast<uint32_t*>(d);
		return PIRE_HIP_OK;
	}
	~HostWord()
	{
		if (host)
			StagingReleaseHost(host, block);
	}
};

struct StreamScratch {
	hipStream_t stream;
	uint8_t* base = nullptr;
	size_t size = 0, used = 0;
	explicit StreamScratch(hipStream_t s) : stream(s) {}
	~StreamScratch()
	{
		if (base)
			(void)hipFreeAsync(base, stream);
	}
	int Reserve(size_t bytes)
	{
		// keep freed scratch in the device's pool: by default it goes back to the driver at the next synchronize
		int dev = 0;
		hipMemPool_t pool;
		if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
			uint64_t keep = 1ull << 30;
			(void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
		}
		void* d = nullptr;
		hipError_t e = hipMallocAsync(&d, bytes, stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(segments)");
		base = static_cast<uint8_t*>(d);
		size = bytes;
		return PIRE_HIP_OK;
	}
	template <class T>
	int Alloc(T** out, size_t count)
	{
		const size_t bytes = (std::max<size_t>(count * sizeof(T), 16) + 255) & ~size_t(255);
		if (used + bytes > size) {
			SetError("segmented scan: scratch arena too small");
			return PIRE_HIP_ENOMEM;
		}
		*out = reinterpret_cast<T*>(base + used);
		used += bytes;
		return PIRE_HIP_OK;
	}
};

int ScanBatch(const ScanParams& q, pire_hip_table* t, hipStream_t stream)
{
	if (RaggedEligible(q, ~0ull))
		return LaunchRagged(q, WorkSlotOf(q.workBase, t->workSlot[q.workDevice].fetch_add(1)), stream);
	return LaunchGeneric(q, stream);
}

#define PIRE_TRY(expr)                                                                                                 \
	do {                                                                                                           \
		if (int rc_ = (expr))                                                                                      \
			return rc_;                                                                                            \
	} while (0)

int HipOk(hipError_t e, const char* what)
{
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, what);
}

}  // namespace

// Worth it when the lanes would starve.  One string per lane walks ~25 MB/s per lane however many lanes are busy
// (measured: 1 024 x 1 MiB 40 ms, 16 384 x 64 KiB 2.6 ms); the segmented scan does ~1.1 TB/s plus ~0.15 ms of fixed
// cost (profiles/r01_long_strings.log).  Either is exact; PIRE_HIP_SEGMENT_BYTES forces this one (tests).
// A mode DERIVED from mode 0 (round 3).  Walk a text from the start state a0 and from a mode's representative b0 at the
// same time: the pair of states moves through the product automaton.  If among ALL pairs reachable from (a0, b0) every
// first component comes with exactly one second component -- b = f(a) -- then the mode's walk is redundant: its guess
// and its end state of every segment are f of mode 0's, whatever the bytes.  That is the case for the sticky modes of
// the benchmark tables ("pattern 3 was seen": the same walk with one more bit set, until mode 0's walk sets it too),
// and then two modes cost ONE pass over the text instead of the pair kernel's two lookups per byte.  The search is a
// BFS over at most `states` pairs (it stops at the first state that turns up with two partners) on the host, once per
// (a0, b0), remembered in the table in reference numbering; the letters are those the 256 byte values map to (the
// marks of Begin() / End() never occur inside a segment).  Exact by construction: the BFS covers every text.
// (f in reference numbering; the device ids go through the arrays of the very image the walk used -- the host's copies
// may have been renumbered by an adaptation on another thread since this call took its parameters)
__global__ void SegmentDeriveKernel(const uint32_t* guess0, const uint32_t* end0, const uint32_t* f, uint32_t states,
                                    const uint32_t* origOfPerm, const uint32_t* permOfOrig, uint32_t* guess, uint32_t* end, uint64_t n)
{
	const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (k >= n)
		return;
	auto through = [&](uint32_t perm) {
		if (perm >= states)
			return kNoState;
		const uint32_t to = f[origOfPerm[perm]];
		return to < states ? permOfOrig[to] : kNoState;
	};
	guess[k] = through(guess0[k]);
	end[k] = through(end0[k]);
}

// f in reference numbering (kNoState where mode 0's walk never gets); false: no such function.  Only pairs reached by
// texts of at least `minSteps` bytes count -- a segment's guess is taken after its whole warm-up, its end state later
// still -- because on the first bytes the two walks have not forgotten where they started (the same mode-0 state then
// comes with several partners).  Levels of pairs up to minSteps (stopping early where a level repeats), then the closure.
bool ModeFunction(const HostTable& h, uint32_t a0, uint32_t b0, uint32_t minSteps, std::vector<uint32_t>* f)
{
	std::vector<uint32_t> letters;
	{
		std::vector<uint8_t> seen(h.letters, 0);
		for (uint32_t c = 0; c < 256; ++c)
			if (h.cls[c] < h.letters && !seen[h.cls[c]]) {
				seen[h.cls[c]] = 1;
				letters.push_back(h.cls[c]);
			}
	}
	// pairs per level we are willing to follow (a function has at most `states` of them in the end; the bound keeps the
	// search under ~0.2 s for the tables of this repository even when it fails late)
	const size_t cap = size_t(h.states) * 2 + 1024;
	std::vector<uint64_t> level{(uint64_t(a0) << 32) | b0}, next;
	for (uint32_t step = 0; step < minSteps; ++step) {
		next.clear();
		for (uint64_t pr : level) {
			const uint32_t a = uint32_t(pr >> 32), b = uint32_t(pr);
			for (uint32_t l : letters)
				next.push_back((uint64_t(h.next[size_t(a) * h.letters + l]) << 32) | h.next[size_t(b) * h.letters + l]);
		}
		std::sort(next.begin(), next.end());
		next.erase(std::unique(next.begin(), next.end()), next.end());
		if (next.size() > cap)
			return false;
		if (next == level)
			break;   // a fixed point: every later level is this one
		level.swap(next);
	}
	f->assign(h.states, kNoState);
	std::vector<uint32_t> todo;
	for (uint64_t pr : level) {
		const uint32_t a = uint32_t(pr >> 32), b = uint32_t(pr);
		if ((*f)[a] == kNoState) {
			(*f)[a] = b;
			todo.push_back(a);
		} else if ((*f)[a] != b) {
			return false;
		}
	}
	while (!todo.empty()) {
		const uint32_t a = todo.back();
		todo.pop_back();
		const uint32_t b = (*f)[a];
		for (uint32_t l : letters) {
			const uint32_t na = h.next[size_t(a) * h.letters + l], nb = h.next[size_t(b) * h.letters + l];
			if ((*f)[na] == kNoState) {
				(*f)[na] = nb;
				todo.push_back(na);
			} else if ((*f)[na] != nb) {
				return false;
			}
		}
	}
	return true;
}

// the same through the table's memory of earlier answers; *f stays valid while the caller holds nothing (a copy)
bool ModeFunctionCached(pire_hip_table* t, uint32_t a0, uint32_t b0, uint32_t minSteps, std::vector<uint32_t>* f)
{
	{
		std::lock_guard<std::mutex> lock(t->segMutex);
		for (const auto& e : t->segModeFns)
			if (e.a0 == a0 && e.b0 == b0 && e.minSteps == minSteps) {
				*f = e.f;
				return !f->empty();
			}
	}
	const bool ok = ModeFunction(t->host, a0, b0, minSteps, f);
	if (!ok)
		f->clear();
	std::lock_guard<std::mutex> lock(t->segMutex);
	if (t->segModeFns.size() < 64)
		t->segModeFns.push_back({a0, b0, minSteps, *f});
	return ok;
}

// Two modes that are NOT functions of each other, as ONE walk all the same: the product automaton of the table with
// itself, started in (a0, b0) -- Scanner::Glue's construction on two copies of the table that differ in their start
// states -- has one state per pair the two walks can be in together.  One
// lookup per byte in ITS dense rows instead of two in the table's (the pair kernel), and both modes' guess and end state
// are the two components of the product's.  Built once per table and (a0, b0), a table object of its own (ranked,
// uploaded and adapted like any other); none when the product outgrows 6 x the table.
struct ModeProduct {
	pire_hip_table* table = nullptr;
	const std::vector<uint32_t>* compA = nullptr;   // [product states, reference numbering] -> the table's state
	const std::vector<uint32_t>* compB = nullptr;
};

int EnsureModeProduct(pire_hip_table* t, uint32_t a0, uint32_t b0, ModeProduct* out)
{
	*out = ModeProduct();
	std::lock_guard<std::mutex> lock(t->segMutex);
	if (!t->segProductTried || t->segProductA0 != a0 || t->segProductB0 != b0) {
		if (t->segProductTried && t->segProduct)
			return PIRE_HIP_OK;   // one product per table: the first pair of modes it met (replacing it would free
			                      // device memory another thread's launch may still read)
		t->segProductTried = true;
		t->segProductA0 = a0;
		t->segProductB0 = b0;
		if (t->host.scannerType != 1 || t->host.empty)
			return PIRE_HIP_OK;
		// Glue's construction, restricted to what a segment can contain: the pairs reachable over the letters of the 256
		// byte values (the marks of Begin() / End() never occur inside a segment; their columns are self loops here).
		// For set_a that is 162 pairs -- every one of them gets a dense row.
		const HostTable& h = t->host;
		const uint32_t LC = h.letters;
		std::vector<uint8_t> isByteLetter(LC, 0);
		for (uint32_t c = 0; c < 256; ++c)
			if (h.cls[c] < LC)
				isByteLetter[h.cls[c]] = 1;
		const size_t cap = size_t(h.states) * 6 + 1024;
		std::vector<std::pair<uint32_t, uint32_t>> pairs{{a0, b0}};
		std::unordered_map<uint64_t, uint32_t> index{{(uint64_t(a0) << 32) | b0, 0u}};
		std::vector<uint32_t> next;
		for (size_t i = 0; i < pairs.size(); ++i) {
			const auto pr = pairs[i];   // (a copy: the vector grows)
			next.resize((i + 1) * size_t(LC));
			for (uint32_t l = 0; l < LC; ++l) {
				if (!isByteLetter[l]) {
					next[i * LC + l] = uint32_t(i);
					continue;
				}
				const uint32_t na = h.next[size_t(pr.first) * LC + l], nb = h.next[size_t(pr.second) * LC + l];
				const uint64_t key = (uint64_t(na) << 32) | nb;
				auto it = index.find(key);
				if (it == index.end()) {
					if (pairs.size() >= cap)
						return PIRE_HIP_OK;   // too large: the pair kernel it is
					it = index.emplace(key, uint32_t(pairs.size())).first;
					pairs.emplace_back(na, nb);
				}
				next[i * LC + l] = it->second;
			}
		}
		const uint32_t PN = uint32_t(pairs.size());
		HostTable prod;
		prod.scannerType = 1;
		prod.headerSize = h.headerSize;
		prod.rowStride = h.rowStride;
		prod.states = PN;
		prod.letters = LC;
		prod.regexps = 0;   // nobody asks the product what it accepts: its states are only ever taken apart
		prod.initial = 0;
		prod.cls = h.cls;
		prod.next.swap(next);
		prod.flags.assign(PN, 0);
		for (uint32_t i = 0; i < PN; ++i) {
			bool absorbing = true;
			for (uint32_t l = 0; l < LC; ++l)
				absorbing = absorbing && prod.next[size_t(i) * LC + l] == i;
			prod.flags[i] = absorbing ? uint8_t(kAbsorbing) : 0;
		}
		prod.acceptOff.assign(size_t(PN) + 1, 0);
		prod.refBufSize = 0;
		prod.blobBytes = 0;
		prod.ranked = false;
		t->segProductA.resize(PN);
		t->segProductB.resize(PN);
		for (uint32_t i = 0; i < PN; ++i) {
			t->segProductA[i] = pairs[i].first;
			t->segProductB[i] = pairs[i].second;
		}
		t->segProduct.reset(new pire_hip_table);
		t->segProduct->host = std::move(prod);
	}
	if (t->segProduct && t->segProductA0 == a0 && t->segProductB0 == b0) {
		out->table = t->segProduct.get();
		out->compA = &t->segProductA;
		out->compB = &t->segProductB;
	}
	return PIRE_HIP_OK;
}

// the product's states taken apart: slot A and slot B of every segment from the product walk's guess and end state
// (components in reference numbering, by the product's reference numbering; device ids go through the arrays of the two
// images the launches used -- either table may have been renumbered on the host by an adaptation since)
__global__ void SegmentSplitKernel(const uint32_t* guessP, const uint32_t* endP, const uint32_t* compA, const uint32_t* compB,
                                   uint32_t productStates, const uint32_t* productOrigOfPerm, const uint32_t* permOfOrig,
                                   uint32_t* guessA, uint32_t* endA, uint32_t* guessB, uint32_t* endB, uint64_t n)
{
	const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (k >= n)
		return;
	const uint32_t g = guessP[k], e = endP[k];
	const uint32_t go = g < productStates ? productOrigOfPerm[g] : kNoState, eo = e < productStates ? productOrigOfPerm[e] : kNoState;
	guessA[k] = go != kNoState ? permOfOrig[compA[go]] : kNoState;
	guessB[k] = go != kNoState ? permOfOrig[compB[go]] : kNoState;
	endA[k] = eo != kNoState ? permOfOrig[compA[eo]] : kNoState;
	endB[k] = eo != kNoState ? permOfOrig[compB[eo]] : kNoState;
}

bool SegmentedEligible(uint64_t n, uint64_t totalBytes)
{
	const pire_hip_config cfg = GetConfig();
	if (cfg.no_segments || n == 0 || n >= (1ull << 20))
		return false;
	if (cfg.segment_bytes)
		return true;
	const double mean = double(totalBytes) / double(n);
	if (mean < 8192.0)
		return false;
	const double lanes = 256.0 * 1024.0;
	const double plain = std::ceil(double(n) / lanes) * mean / 25e6;
	const double segmented = double(totalBytes) / 2.0e12 + 80e-6;   // (round 3's numbers; rounds 1-2: 1.1e12, 150e-6)
	return plain > 1.5 * segmented;
}

// p: the original batch with DEVICE pointers (strided, or offsets on the device); hostOffsets: the same offsets on
// the host (nullptr for strided batches) -- the host has to know the lengths to cut the strings up.
// halfFinalResults (nullable, device, [n][regexps], packed increments required): the table is walked as a
// HalfFinalScanner; p.startPerm must then be Initialize() with Begin() folded in (it is for every other caller too).
int RunSegmented(pire_hip_table* t, const ScanParams& p, const uint64_t* hostOffsets, hipStream_t stream,
                 uint32_t* halfFinalResults, bool* halfFinalIncomplete)
{
	PIRE_TRY(CheckCounts(p));
	// diagnostics (PIRE_HIP_SEGMENT_STATS): where the host's time goes, with the stream drained at every mark
	const pire_hip_config cfg = GetConfig();
	const bool wantStats = cfg.segment_stats != 0;
	auto t0 = std::chrono::steady_clock::now();
	std::string timeline;
	auto mark = [&](const char* what) {
		if (!wantStats)
			return;
		(void)hipStreamSynchronize(stream);
		const auto t1 = std::chrono::steady_clock::now();
		char buf[64];
		snprintf(buf, sizeof(buf), " %s %.0f us;", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
		timeline += buf;
		t0 = t1;
	};
	const uint64_t n = p.n;
	const uint64_t total = hostOffsets ? hostOffsets[n] - hostOffsets[0] : n * p.len;
	int cus = 0;
	PIRE_TRY(DeviceCUs(&cus));
	// segments: as many as there are lanes, a multiple of the 128-byte window, long against the warm-up
	const uint64_t wanted = uint64_t(cus) * 1024;
	// (floor 1 KiB since round 3 -- four times the warm-up: below 1 GiB of text 4 KiB segments left most lanes without
	// one, and a lane walks 25 MB/s however few are busy: 256 MiB took 0.23 ms, as long as 1 GiB)
	uint64_t segBytes = std::min<uint64_t>(1u << 20, std::max<uint64_t>(1024, (total / wanted + 255) / 256 * 256));
	segBytes = Knob(cfg.segment_bytes, segBytes);
	const uint64_t warmBytes = Knob(cfg.segment_warmup, 256);
	const uint32_t maxModes = uint32_t(std::min<uint64_t>(kMaxModes, std::max<uint64_t>(1, Knob(cfg.segment_modes, 6))));
	uint64_t budget = Knob(cfg.segment_budget, 32);   // round trips for segments no mode predicted
	if (segBytes == 0) {
		SetError("PIRE_HIP_SEGMENT_BYTES must be positive");
		return PIRE_HIP_EINVAL;
	}

	// ---- geometry
	SegGeometry g = {};
	g.segBytes = segBytes;
	g.warmBytes = warmBytes;
	g.nStrings = n;
	std::vector<uint32_t> strFirst(n + 1);
	{
		uint64_t acc = 0;
		for (uint64_t i = 0; i < n; ++i) {
			strFirst[i] = uint32_t(acc);
			const uint64_t len = hostOffsets ? hostOffsets[i + 1] - hostOffsets[i] : p.len;
			acc += std::max<uint64_t>(1, (len + segBytes - 1) / segBytes);
			if (acc >= (1ull << 31)) {
				SetError("too many segments");
				return PIRE_HIP_EINVAL;
			}
		}
		strFirst[n] = uint32_t(acc);
		g.nSeg = acc;
	}
	const uint64_t S = g.nSeg;
	StreamScratch scratch(stream);
	const uint64_t chainBlocks = (S + kChainBlock - 1) / kChainBlock;
	{
		const size_t perSeg = 3 * 8 + 3 * 4            // the cut
		                      + (size_t(maxModes) + 1) * 8 + 4   // guess + end per slot, the constant init array
		                      + 8                      // prefix
		                      + 4;                     // true start states (half-final counting)
		const size_t perString = 5 * 4 + 2 * 8 + 3 * 4 + 4;
		PIRE_TRY(scratch.Reserve(S * perSeg + n * perString + chainBlocks * 16 + size_t(t->host.states) * 4 * kMaxModes + 64 * 256));
	}
	if (hostOffsets) {
		uint32_t* d = nullptr;
		PIRE_TRY(scratch.Alloc(&d, n + 1));
		PIRE_TRY(HipOk(hipMemcpyAsync(d, strFirst.data(), (n + 1) * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(segment index)"));
		g.strFirst = d;
	} else {
		g.segsPerString = strFirst[1] - strFirst[0];
	}
	SegArrays a = {};
	PIRE_TRY(scratch.Alloc(&a.segBegin, S));
	PIRE_TRY(scratch.Alloc(&a.segEnd, S));
	PIRE_TRY(scratch.Alloc(&a.warmBegin, S));
	PIRE_TRY(scratch.Alloc(&a.segStr, S));
	PIRE_TRY(scratch.Alloc(&a.segJ, S));
	if (p.initIdx)
		PIRE_TRY(scratch.Alloc(&a.initSeg, S));
	const unsigned blocks = unsigned((S + 255) / 256);
	SlotArrays sl = {};
	PIRE_TRY(scratch.Alloc(&sl.guess[kMaxModes], S));   // the patch slot: empty
	PIRE_TRY(scratch.Alloc(&sl.end[kMaxModes], S));
	ChainArrays c = {};
	PIRE_TRY(scratch.Alloc(&c.prefix, S));
	PIRE_TRY(scratch.Alloc(&c.blockAgg, chainBlocks));
	PIRE_TRY(scratch.Alloc(&c.blockPrefix, chainBlocks));
	if (halfFinalResults)
		PIRE_TRY(scratch.Alloc(&c.trueStart, S));
	PIRE_TRY(scratch.Alloc(&c.finalState, n));
	PIRE_TRY(scratch.Alloc(&c.strDone, n));
	PIRE_TRY(scratch.Alloc(&c.breakSeg, n));
	PIRE_TRY(scratch.Alloc(&c.breakState, n));
	PIRE_TRY(scratch.Alloc(&c.broken, 1));
	// (the kernel that fills these -- the cut, the patch slot, the chain's per-string arrays -- is launched further down,
	// right in front of the first mode's pass: everything the host does in between would otherwise be a gap on the
	// stream between that 5 us kernel and the pass, 10-20 us in the kernel trace of a call)
	// the host's copy of the cut (same arithmetic), for the batches of broken chains
	auto segBeginOf = [&](uint64_t i, uint32_t k) {
		const uint64_t B = hostOffsets ? hostOffsets[i] : i * p.stride;
		return B + uint64_t(k - strFirst[i]) * segBytes;
	};
	auto stringEndOf = [&](uint64_t i) { return hostOffsets ? hostOffsets[i + 1] : i * p.stride + p.len; };

	ScanParams q = p;   // the segment batches: same table, same text
	q.len = q.stride = 0;
	q.textEnd = hostOffsets ? hostOffsets[n] : (n - 1) * p.stride + p.len;
	q.outFinal = nullptr;
	q.outCounts = nullptr;

	// how many leading segments lie on a regular grid of fixed-length records (see addMode)
	uint64_t gridSegs = 0;
	if (!hostOffsets && !cfg.segment_no_grid) {
		if (n == 1)
			gridSegs = p.len / segBytes;
		else if (p.stride == p.len && p.len % segBytes == 0)
			gridSegs = S;
		ScanParams probe = q;
		probe.offsets = nullptr;
		probe.n = gridSegs;
		probe.len = probe.stride = segBytes;
		if (!TiledEligible(probe))
			gridSegs = 0;
	}

	// ---- slots
	uint32_t* dConst = nullptr;
	// one mode: warm-up from its representative (mode 0: the string's own start state, and the first segment's
	// warm-up is empty, so its guess is the true start state), then the scan proper from the guesses
	// the grid segments of ONE mode with the warm-up inside the tiled pass (tiled.hip, ScanTiledSegKernel): whole
	// 64-segment tasks, whole pairs of tiles
	const uint64_t fusedSegs = (warmBytes % 256 == 0 && warmBytes <= segBytes && segBytes % 256 == 0 && !cfg.segment_no_pair)
	                               ? gridSegs & ~uint64_t(63) : 0;
	auto addMode = [&](bool first, uint32_t representative) -> int {
		const uint32_t m = sl.count;
		PIRE_TRY(scratch.Alloc(&sl.guess[m], S));
		PIRE_TRY(scratch.Alloc(&sl.end[m], S));
		if (fusedSegs) {
			// one launch for warm-up and scan of the grid's whole tasks; whatever is left (the segments beyond the
			// last whole 64, tails) gets a warm-up batch and a scan batch of its own
			ScanParams r = q;
			r.offsets = nullptr;
			r.ends = nullptr;
			r.n = fusedSegs;
			r.len = r.stride = segBytes;
			r.outIdx = sl.end[m];
			if (first) {
				r.initIdx = a.initSeg;                   // nullable: then startPerm (Initialize + Begin folded)
				r.flags = (p.flags & PIRE_HIP_RUN_BEGIN) | kPermIds;
			} else {
				r.initIdx = nullptr;
				r.startPerm = representative;
				r.flags = kPermIds;
			}
			PIRE_TRY(LaunchTiledSeg(r, warmBytes, a.segJ, sl.guess[m], stream));
			if (fusedSegs < S) {
				const uint64_t rest = S - fusedSegs;
				if (!first) {
					if (!dConst)
						PIRE_TRY(scratch.Alloc(&dConst, S));
					hipLaunchKernelGGL(SegmentFillKernel, dim3(unsigned((rest + 255) / 256)), dim3(256), 0, stream, dConst, representative, rest);
				}
				q.n = rest;
				q.offsets = a.warmBegin + fusedSegs;
				q.ends = a.segBegin + fusedSegs;
				q.initIdx = first ? (a.initSeg ? a.initSeg + fusedSegs : nullptr) : dConst;
				q.flags = (first ? (p.flags & PIRE_HIP_RUN_BEGIN) : 0u) | kPermIds;
				q.outIdx = sl.guess[m] + fusedSegs;
				PIRE_TRY(ScanBatch(q, t, stream));
				q.flags = kPermIds;
				q.offsets = a.segBegin + fusedSegs;
				q.ends = a.segEnd + fusedSegs;
				q.initIdx = sl.guess[m] + fusedSegs;
				q.outIdx = sl.end[m] + fusedSegs;
				PIRE_TRY(ScanBatch(q, t, stream));
			}
			sl.count = m + 1;
			return PIRE_HIP_OK;
		}
		q.n = S;
		q.offsets = a.warmBegin;
		q.ends = a.segBegin;
		if (first) {
			q.initIdx = a.initSeg;                       // nullable: then startPerm (Initialize + Begin folded)
			q.flags = (p.flags & PIRE_HIP_RUN_BEGIN) | kPermIds;
		} else {
			if (!dConst)
				PIRE_TRY(scratch.Alloc(&dConst, S));
			hipLaunchKernelGGL(SegmentFillKernel, dim3(blocks), dim3(256), 0, stream, dConst, representative, S);
			q.initIdx = dConst;
			q.flags = kPermIds;
		}
		q.outIdx = sl.guess[m];
		PIRE_TRY(ScanBatch(q, t, stream));
		// the scan proper: the leading `gridSegs` segments are fixed-length records on a regular grid (a single
		// string's full segments; several strings whose length is a multiple of the segment) -> the tiled kernel,
		// twice as fast as the ragged one; whatever is left (tails) goes to the ragged / generic kernel
		q.flags = kPermIds;
		if (gridSegs) {
			ScanParams r = q;
			r.offsets = nullptr;
			r.ends = nullptr;
			r.n = gridSegs;
			r.len = r.stride = segBytes;
			r.initIdx = sl.guess[m];
			r.outIdx = sl.end[m];
			PIRE_TRY(LaunchTiled(r, stream));
		}
		if (gridSegs < S) {
			q.n = S - gridSegs;
			q.offsets = a.segBegin + gridSegs;
			q.ends = a.segEnd + gridSegs;
			q.initIdx = sl.guess[m] + gridSegs;
			q.outIdx = sl.end[m] + gridSegs;
			PIRE_TRY(ScanBatch(q, t, stream));
		}
		sl.count = m + 1;
		return PIRE_HIP_OK;
	};
	// Two modes at once (round 3): the scan proper of mode 0 and of the first known mode read the same bytes, so the
	// grid segments take ONE pass of the fused pair kernel (pair.hip: the tiled load path, two lookups per byte -- here
	// the same table twice) instead of two tiled passes, and the two warm-ups that make the guesses are the first tiles
	// of that same pass (a lane starts `warmBytes` before its segment in the two representatives' states): one launch
	// instead of four, the text read 1.06 times instead of 2.12.  Segments beyond the last whole 64 (and tails) take the
	// separate warm-up and scan batches as before.
	auto addModePair = [&](uint32_t representativeB) -> int {
		const uint32_t m = sl.count;
		for (uint32_t k = m; k < m + 2; ++k) {
			PIRE_TRY(scratch.Alloc(&sl.guess[k], S));
			PIRE_TRY(scratch.Alloc(&sl.end[k], S));
		}
		const uint64_t paired = gridSegs & ~uint64_t(63);
		const uint64_t separate = paired;   // segments from here on: warm-up batches of their own
		if (separate < S) {
			if (!dConst)
				PIRE_TRY(scratch.Alloc(&dConst, S));
			hipLaunchKernelGGL(SegmentFillKernel, dim3(blocks), dim3(256), 0, stream, dConst, representativeB, S);
		}
		for (uint32_t k = 0; k < 2 && separate < S; ++k) {   // mode 0 from the strings' own start states, the other from its representative
			q.n = S - separate;
			q.offsets = a.warmBegin + separate;
			q.ends = a.segBegin + separate;
			q.initIdx = k == 0 ? (a.initSeg ? a.initSeg + separate : nullptr) : dConst;
			q.flags = (k == 0 ? (p.flags & PIRE_HIP_RUN_BEGIN) : 0u) | kPermIds;
			q.outIdx = sl.guess[m + k] + separate;
			PIRE_TRY(ScanBatch(q, t, stream));
		}
		q.flags = kPermIds;
		ScanParams ra = q;
		ra.offsets = nullptr;
		ra.ends = nullptr;
		ra.n = paired;
		ra.len = ra.stride = segBytes;
		ra.outIdx = sl.end[m];
		ScanParams rb = ra;
		ra.initIdx = a.initSeg;   // nullable: then startPerm (Initialize + Begin folded)
		ra.flags = (p.flags & PIRE_HIP_RUN_BEGIN) | kPermIds;
		rb.initIdx = nullptr;
		rb.startPerm = representativeB;
		PIRE_TRY(LaunchPairTiled(ra, rb, sl.end[m + 1], stream, warmBytes, a.segJ, sl.guess[m], sl.guess[m + 1]));
		for (uint32_t k = 0; k < 2 && paired < S; ++k) {
			q.n = S - paired;
			q.offsets = a.segBegin + paired;
			q.ends = a.segEnd + paired;
			q.initIdx = sl.guess[m + k] + paired;
			q.outIdx = sl.end[m + k] + paired;
			PIRE_TRY(ScanBatch(q, t, stream));
		}
		sl.count = m + 2;
		return PIRE_HIP_OK;
	};
	// Modes 0 and B as one walk of their product automaton (EnsureModeProduct) over the grid's whole tasks; the rest as
	// in addModePair.
	void* productScratch = nullptr;   // the product walk's own arrays: freed on the stream behind the kernels that use them
	struct ProductGuard {
		void*& q;
		hipStream_t s;
		~ProductGuard()
		{
			if (q)
				(void)hipFreeAsync(q, s);
		}
	} productGuard{productScratch, stream};
	void* pinnedComp = nullptr;
	size_t pinnedCompBytes = 0;
	struct PinnedCompGuard {
		void*& q;
		size_t& bytes;
		~PinnedCompGuard()
		{
			if (q)
				StagingReleaseHost(q, bytes);   // the call has synchronised its stream by the time it returns
		}
	} pinnedCompGuard{pinnedComp, pinnedCompBytes};
	bool productUsed = false;
	TableUse productUse;   // the product table's numbering stays put until this call returns
	auto addModeProduct = [&](const ModeProduct& mp, uint32_t representativeB) -> int {
		const uint32_t m = sl.count;
		for (uint32_t k = m; k < m + 2; ++k) {
			PIRE_TRY(scratch.Alloc(&sl.guess[k], S));
			PIRE_TRY(scratch.Alloc(&sl.end[k], S));
		}
		ScanParams pp;
		PIRE_TRY(PrepareScanParams(mp.table, &pp, 0, &productUse));   // the product on this device; startPerm = its initial state (a0, b0)
		const HostTable& ph = mp.table->host;
		const uint32_t PN = ph.states;
		const size_t bytes = ((fusedSegs * 4 + 255) & ~size_t(255)) * 2 + ((size_t(PN) * 4 + 255) & ~size_t(255)) * 2;
		PIRE_TRY(HipOk(hipMallocAsync(&productScratch, bytes, stream), "hipMallocAsync(product walk)"));
		uint8_t* base = static_cast<uint8_t*>(productScratch);
		uint32_t* guessP = reinterpret_cast<uint32_t*>(base);
		uint32_t* endP = reinterpret_cast<uint32_t*>(base + ((fusedSegs * 4 + 255) & ~size_t(255)));
		uint32_t* dCompA = reinterpret_cast<uint32_t*>(base + ((fusedSegs * 4 + 255) & ~size_t(255)) * 2);
		uint32_t* dCompB = dCompA + ((size_t(PN) + 63) & ~size_t(63));
		ScanParams r = pp;
		r.text = q.text;
		r.textEnd = q.textEnd;
		r.offsets = nullptr;
		r.ends = nullptr;
		r.n = fusedSegs;
		r.len = r.stride = segBytes;
		r.initIdx = nullptr;
		r.flags = kPermIds;
		r.outIdx = endP;
		r.outFinal = nullptr;
		r.outCounts = nullptr;
		PIRE_TRY(LaunchTiledSeg(r, warmBytes, a.segJ, guessP, stream));
		// (behind the launch: the host fills these while the pass runs)
		// the components of the product's states (the split kernel takes them through both images' own numberings)
		PIRE_TRY(StagingAcquireHost(size_t(PN) * 8, &pinnedComp, &pinnedCompBytes));
		uint32_t* hA = static_cast<uint32_t*>(pinnedComp);
		uint32_t* hB = hA + PN;
		memcpy(hA, mp.compA->data(), size_t(PN) * 4);   // reference numbering both: the kernel translates on the device
		memcpy(hB, mp.compB->data(), size_t(PN) * 4);
		PIRE_TRY(HipOk(hipMemcpyAsync(dCompA, hA, size_t(PN) * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(product components)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(dCompB, hB, size_t(PN) * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(product components)"));
		hipLaunchKernelGGL(SegmentSplitKernel, dim3(unsigned((fusedSegs + 255) / 256)), dim3(256), 0, stream, guessP, endP, dCompA, dCompB,
		                   PN, pp.origOfPerm, p.permOfOrig, sl.guess[m], sl.end[m], sl.guess[m + 1], sl.end[m + 1], fusedSegs);
		if (fusedSegs < S) {
			const uint64_t rest = S - fusedSegs;
			if (!dConst)
				PIRE_TRY(scratch.Alloc(&dConst, S));
			hipLaunchKernelGGL(SegmentFillKernel, dim3(unsigned((rest + 255) / 256)), dim3(256), 0, stream, dConst, representativeB, rest);
			for (uint32_t k = 0; k < 2; ++k) {
				q.n = rest;
				q.offsets = a.warmBegin + fusedSegs;
				q.ends = a.segBegin + fusedSegs;
				q.initIdx = k == 0 ? nullptr : dConst;
				q.flags = (k == 0 ? (p.flags & PIRE_HIP_RUN_BEGIN) : 0u) | kPermIds;
				q.outIdx = sl.guess[m + k] + fusedSegs;
				PIRE_TRY(ScanBatch(q, t, stream));
				q.flags = kPermIds;
				q.offsets = a.segBegin + fusedSegs;
				q.ends = a.segEnd + fusedSegs;
				q.initIdx = sl.guess[m + k] + fusedSegs;
				q.outIdx = sl.end[m + k] + fusedSegs;
				PIRE_TRY(ScanBatch(q, t, stream));
			}
			q.flags = kPermIds;
		}
		productUsed = true;
		sl.count = m + 2;
		return PIRE_HIP_OK;
	};
	mark("setup");
	// the modes earlier calls on this table learned: the automaton is the same, the text probably similar
	std::vector<uint32_t> known;
	{
		std::lock_guard<std::mutex> lock(t->segMutex);
		for (uint32_t r : t->segModes)   // kept as reference state indices: adapt() renumbers the device ids
			if (r < t->host.states)
				known.push_back(r);
	}
	size_t nextKnown = 0;
	bool pairedFirst = false;
	if (!known.empty() && maxModes >= 2 && gridSegs >= 64 && !cfg.segment_no_pair && warmBytes % 256 == 0 && warmBytes <= segBytes) {
		ScanParams probe = q;
		probe.offsets = nullptr;
		probe.n = gridSegs & ~uint64_t(63);
		probe.len = probe.stride = segBytes;
		pairedFirst = TiledEligible(probe) && segBytes % 256 == 0;
	}
	hipLaunchKernelGGL(SegmentPrepKernel, dim3(blocks), dim3(256), 0, stream, p, g, a, sl.guess[kMaxModes], sl.end[kMaxModes],
	                   c.strDone, c.breakSeg, c.broken);
	// Modes that are a function of mode 0 (ModeFunction above) are not walked: their slots are filled from mode 0's by a
	// table lookup per segment.  Only with one start state for every string (no caller's resume states): mode 0 is then
	// the walk from q.startPerm, slot 0.
	// ... and with segments no shorter than the warm-up: every segment but a string's first then has ALL of it behind it
	const bool derive = !p.initIdx && !cfg.segment_no_derive && segBytes >= warmBytes && warmBytes < (1u << 20);
	const uint32_t minSteps = uint32_t(warmBytes);
	// (the start state in reference numbering, from the table itself: Initialize(), then Begin() if asked -- not through
	// the host's copy of the device numbering, which another thread's adaptation may have moved on)
	uint32_t a0 = t->host.initial;
	if (p.flags & PIRE_HIP_RUN_BEGIN)
		a0 = t->host.next[size_t(a0) * t->host.letters + t->host.cls[kBeginMark]];
	uint32_t* dModeFn = nullptr;
	void* pinnedFn = nullptr;
	size_t pinnedFnBytes = 0;
	struct PinnedGuard {
		void*& q;
		size_t& bytes;
		~PinnedGuard()
		{
			if (q)
				StagingReleaseHost(q, bytes);   // the call has synchronised its stream by the time it returns
		}
	} pinnedGuard{pinnedFn, pinnedFnBytes};
	uint32_t derivedModes = 0;
	auto addDerived = [&](const std::vector<uint32_t>& fOrig) -> int {
		const uint32_t m = sl.count;
		const uint32_t N = t->host.states;
		PIRE_TRY(scratch.Alloc(&sl.guess[m], S));
		PIRE_TRY(scratch.Alloc(&sl.end[m], S));
		if (!pinnedFn) {
			PIRE_TRY(StagingAcquireHost(size_t(N) * 4 * kMaxModes, &pinnedFn, &pinnedFnBytes));
			PIRE_TRY(scratch.Alloc(&dModeFn, size_t(N) * kMaxModes));
		}
		uint32_t* host = static_cast<uint32_t*>(pinnedFn) + size_t(derivedModes) * N;
		uint32_t* dev = dModeFn + size_t(derivedModes) * N;
		memcpy(host, fOrig.data(), size_t(N) * 4);   // reference numbering: the kernel translates with the image's own arrays
		PIRE_TRY(HipOk(hipMemcpyAsync(dev, host, size_t(N) * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(mode function)"));
		hipLaunchKernelGGL(SegmentDeriveKernel, dim3(blocks), dim3(256), 0, stream, sl.guess[0], sl.end[0], dev, N, p.origOfPerm,
		                   p.permOfOrig, sl.guess[m], sl.end[m], S);
		++derivedModes;
		sl.count = m + 1;
		return PIRE_HIP_OK;
	};
	std::vector<std::vector<uint32_t>> derivable;
	std::vector<uint32_t> walked;   // known modes that need a walk of their own
	for (uint32_t r : known) {
		std::vector<uint32_t> f;
		if (derive && derivable.size() + 1 < kMaxModes && ModeFunctionCached(t, a0, r, minSteps, &f))
			derivable.push_back(std::move(f));
		else
			walked.push_back(r);
	}
	ModeProduct product;
	if (pairedFirst && !walked.empty() && fusedSegs && !p.initIdx && !cfg.segment_no_product)
		PIRE_TRY(EnsureModeProduct(t, a0, walked[0], &product));
	if (product.table) {
		PIRE_TRY(addModeProduct(product, p.hostPermOfOrig[walked[0]]));
		nextKnown = 1;
		pairedFirst = false;
		mark("modes 0+1 (product)");
	} else if (pairedFirst && !walked.empty()) {
		PIRE_TRY(addModePair(p.hostPermOfOrig[walked[0]]));
		nextKnown = 1;
		mark("modes 0+1 (fused)");
	} else {
		pairedFirst = false;
		PIRE_TRY(addMode(true, 0));
		mark("mode 0");
	}
	for (const auto& f : derivable)
		if (sl.count < maxModes)
			PIRE_TRY(addDerived(f));
	for (; nextKnown < walked.size(); ++nextKnown)
		if (sl.count < maxModes)
			PIRE_TRY(addMode(false, p.hostPermOfOrig[walked[nextKnown]]));

	mark("known modes");
	// ---- the chain
	// how many chains broke: a mapped host word the resolve kernel's last block writes (no copy to enqueue per round)
	HostWord brokenWord;
	PIRE_TRY(brokenWord.Acquire());
	c.brokenHost = brokenWord.dev;
	uint32_t *dExStr = nullptr, *dExInit = nullptr, *dExOut = nullptr;
	uint64_t *dExB = nullptr, *dExE = nullptr;
	std::vector<uint32_t> breakSeg(n), breakState(n), exStr, exInit;
	std::vector<uint64_t> exB, exE;
	std::unordered_map<uint32_t, uint32_t> surprises;
	uint64_t nWalked = 0, nPlain = 0, rounds = 0;
	// finish: End(), outputs and match counters.  With `broken` it runs only if no chain broke (see ChainsIntact).
	auto launchFinish = [&](const uint32_t* broken) -> int {
		if (n <= 4096) {
			hipLaunchKernelGGL(SegmentFinishSmallKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, p, c.finalState, broken,
			                   c.brokenHost);
		} else {
			ScanParams f = p;
			f.compact = 0;
			const LdsLayout L = MakeLayout(f.hot, f.outCounts ? f.regexps : 0, kRotPitch, 0);
			PIRE_TRY(HipOk(SetDynamicLds(reinterpret_cast<const void*>(SegmentFinishKernel), uint32_t(L.total)), "hipFuncSetAttribute(LDS)"));
			const unsigned threads = 1024;
			const uint64_t tasks = (n + 63) / 64;
			const unsigned cblocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((tasks * 64 + threads - 1) / threads, uint64_t(cus))));
			hipLaunchKernelGGL(SegmentFinishKernel, dim3(cblocks), dim3(threads), L.total, stream, f, c.finalState, broken, c.brokenHost);
		}
		return PIRE_HIP_OK;
	};
	for (;;) {
		++rounds;
		*static_cast<volatile uint32_t*>(brokenWord.host) = kNoState;
		hipLaunchKernelGGL(SegmentMapScanKernel, dim3(unsigned(chainBlocks)), dim3(kChainBlock), 0, stream, g, a, sl, c);
		if (chainBlocks > 1)
			hipLaunchKernelGGL(SegmentBlockScanKernel, dim3(1), dim3(kChainBlock), 0, stream, c, uint32_t(chainBlocks));
		hipLaunchKernelGGL(SegmentResolveKernel, dim3(blocks), dim3(256), 0, stream, g, a, sl, c);
		if (halfFinalResults)   // the counting needs the host's decision first
			hipLaunchKernelGGL(SegmentReportKernel, dim3(1), dim3(64), 0, stream, c.broken, c.brokenHost);
		else
			PIRE_TRY(launchFinish(c.broken));
		PIRE_TRY(HipOk(hipGetLastError(), "segmented scan: chain kernels"));
		PIRE_TRY(HipOk(hipStreamSynchronize(stream), "hipStreamSynchronize"));
		const uint32_t broken = *static_cast<volatile uint32_t*>(brokenWord.host);
		if (broken == kNoState) {
			SetError("segmented scan: the chain kernels did not report");
			return PIRE_HIP_EUNSUPPORTED;
		}
		if (!broken)
			break;
		PIRE_TRY(HipOk(hipMemsetAsync(c.broken, 0, 4, stream), "hipMemset"));
		// where and in which state: the host only orchestrates, the states stay on the device
		PIRE_TRY(HipOk(hipMemcpyAsync(breakSeg.data(), c.breakSeg, n * 4, hipMemcpyDeviceToHost, stream), "hipMemcpy(D2H)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(breakState.data(), c.breakState, n * 4, hipMemcpyDeviceToHost, stream), "hipMemcpy(D2H)"));
		PIRE_TRY(HipOk(hipStreamSynchronize(stream), "hipStreamSynchronize"));
		exStr.clear();
		for (uint64_t i = 0; i < n; ++i)
			if (breakSeg[i] != kNoState && breakSeg[i] > strFirst[i] && breakSeg[i] < strFirst[i + 1])
				exStr.push_back(uint32_t(i));
		// (breakSeg is cleared before every round, so only the strings that broke in THIS round have an entry)
		if (exStr.size() != broken) {
			SetError("segmented scan: inconsistent chain state");
			return PIRE_HIP_EUNSUPPORTED;
		}
		// A state that breaks chains round after round is a mode of the automaton the guesses do not know yet: a
		// sticky mode breaks the chain again at the very next segment, a transient state does it once.  (Counting
		// rounds, not strings: a mode costs two scans of everything, a round trip only the broken segments.)
		uint32_t best = 0, bestCount = 0;
		{
			std::vector<uint32_t> seen;
			for (uint32_t i : exStr)
				seen.push_back(breakState[i]);
			std::sort(seen.begin(), seen.end());
			seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
			for (uint32_t v : seen) {
				const uint32_t cnt = ++surprises[v];
				if (cnt > bestCount) {
					best = v;
					bestCount = cnt;
				}
			}
		}
		if (sl.count < maxModes && (bestCount >= 2 || budget == 0)) {
			std::vector<uint32_t> f;
			if (derive && derivedModes + 1 < kMaxModes && ModeFunctionCached(t, a0, p.hostOrigOfPerm[best], minSteps, &f))
				PIRE_TRY(addDerived(f));
			else
				PIRE_TRY(addMode(false, best));
			{
				std::lock_guard<std::mutex> lock(t->segMutex);
				const uint32_t orig = p.hostOrigOfPerm[best];
				if (std::find(t->segModes.begin(), t->segModes.end(), orig) == t->segModes.end() && t->segModes.size() < kMaxModes)
					t->segModes.push_back(orig);
			}
			surprises.erase(best);
			PIRE_TRY(HipOk(hipMemsetAsync(c.breakSeg, 0xFF, n * 4, stream), "hipMemset"));
			continue;
		}
		// scan the broken segments from the states the chains are in -- or, when the budget of such round trips
		// is spent, the whole rest of those strings, the plain sequential way
		const bool plain = budget == 0;
		if (!plain)
			--budget;
		const size_t m = exStr.size();
		exB.resize(m);
		exE.resize(m);
		exInit.resize(m);
		for (size_t j = 0; j < m; ++j) {
			const uint64_t i = exStr[j];
			exB[j] = segBeginOf(i, breakSeg[i]);
			exE[j] = plain ? stringEndOf(i) : std::min(stringEndOf(i), exB[j] + segBytes);
			exInit[j] = breakState[i];
		}
		if (!dExB) {
			PIRE_TRY(scratch.Alloc(&dExB, n));
			PIRE_TRY(scratch.Alloc(&dExE, n));
			PIRE_TRY(scratch.Alloc(&dExStr, n));
			PIRE_TRY(scratch.Alloc(&dExInit, n));
			PIRE_TRY(scratch.Alloc(&dExOut, n));
		}
		PIRE_TRY(HipOk(hipMemcpyAsync(dExB, exB.data(), m * 8, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(dExE, exE.data(), m * 8, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(dExStr, exStr.data(), m * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(dExInit, exInit.data(), m * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		q.n = m;
		q.offsets = dExB;
		q.ends = dExE;
		q.initIdx = dExInit;
		q.flags = kPermIds;
		q.outIdx = dExOut;
		PIRE_TRY(ScanBatch(q, t, stream));
		hipLaunchKernelGGL(SegmentPatchKernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, stream, dExStr, dExOut, uint32_t(m),
		                   plain ? 1u : 0u, sl, c);
		PIRE_TRY(HipOk(hipMemsetAsync(c.breakSeg, 0xFF, n * 4, stream), "hipMemset"));
		// the host vectors are sources of async copies: they are rewritten only after the next synchronize
		PIRE_TRY(HipOk(hipStreamSynchronize(stream), "hipStreamSynchronize"));
		(plain ? nPlain : nWalked) += m;
	}

	mark("chain");
	if (halfFinalIncomplete)
		*halfFinalIncomplete = nPlain != 0;
	if (halfFinalResults && !nPlain) {
		// every segment now knows the state it is really entered in (not so after a plain walk: the caller then
		// counts the whole batch with the one-string-per-lane kernel)
		PIRE_TRY(HipOk(hipMemsetAsync(halfFinalResults, 0, n * size_t(p.regexps) * 4, stream), "hipMemset"));
		ScanParams hp = p;
		hp.compact = 0;
		const LdsLayout L = MakeLayout(hp.hot, 0, 256u, 0);
		const uint32_t ldsBytes = L.total + 64;
		PIRE_TRY(HipOk(SetDynamicLds(reinterpret_cast<const void*>(SegmentHalfFinalKernel), uint32_t(ldsBytes)), "hipFuncSetAttribute(LDS)"));
		const uint32_t initialPerm = p.hostPermOfOrig[t->host.initial];
		// 1024-thread blocks: the dense rows take 66 KB of LDS, so this is 16 waves per CU instead of 8
		hipLaunchKernelGGL(SegmentHalfFinalKernel, dim3(unsigned((S + 1023) / 1024)), dim3(1024), ldsBytes, stream, hp, g, a,
		                   c.trueStart, c.strDone, initialPerm, halfFinalResults);
		mark("half-final counts");
	}
	// ---- finish: already done behind the last resolve kernel, except after the half-final counting
	if (halfFinalResults)
		PIRE_TRY(launchFinish(nullptr));
	PIRE_TRY(HipOk(hipGetLastError(), "segmented scan launch"));
	// "+plain": some strings ended in the sequential walk; the symbol says whether two modes shared one pass, or one
	// mode had its warm-up inside the tiled pass
	NoteKernel(nPlain ? "segmented+plain" : "segmented",
	           pairedFirst ? "pirehip::ScanPairTiledKernel"
	           : productUsed ? "pirehip::ScanTiledSegKernel+product"
	           : fusedSegs   ? (derivedModes ? "pirehip::ScanTiledSegKernel+derived" : "pirehip::ScanTiledSegKernel")
	                         : "");
	if (wantStats) {
		mark("finish");
		fprintf(stderr, "pire_hip segmented: %llu strings, %llu segments of %llu B (+%llu B warm-up), %u modes, %llu chain "
		                "rounds; segments scanned again from the chain's state %llu, strings left to the plain walk %llu;%s\n",
		        (unsigned long long)n, (unsigned long long)S, (unsigned long long)segBytes, (unsigned long long)warmBytes,
		        sl.count, (unsigned long long)rounds, (unsigned long long)nWalked, (unsigned long long)nPlain, timeline.c_str());
	}
	return PIRE_HIP_OK;
}

}  // namespace pirehip
