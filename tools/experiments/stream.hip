// NOT COMPILED INTO THE LIBRARY (round 2 experiment, profiles/r02_stream_kernel.log).  To try it again: copy to
// pire_amd/csrc/, add stream.hip to NAMES in the Makefile, declare StreamEligible / LaunchStream in internal.h, add the
// flag PIRE_HIP_RUN_SHORT = 1u << 5 to include/pire_hip.h (and to the mask at the top of RunImpl), route in Dispatch():
//     streamed = !tiled && StreamEligible(p) && (p.flags & PIRE_HIP_RUN_SHORT)  ->  LaunchStream(p, stream)
// State: the first version (two register tiles, end-of-string work per chunk) passed tools/experiments/test_stream.py
// on the GPU -- every border case, empty and tiny strings, the per-string fallback, a table that traps -- at 1.3-1.7
// TB/s on the URL batch (ragged kernel: 2.05).  This file is the second version (one walk instance, end-of-string
// queue, batched offset loads: 5.5 K instead of 13.6 K instructions, 123 VGPRs, no scratch): 1.8 / 2.07 TB/s on the
// 0.43 / 1.7 GiB URL batches against 2.06 / 2.48 -- and it has a RACE that was not found: ~0.5 % of the strings, always the
// first string of a lane when it starts in the lane's first tile, come back as if no byte had been walked, a different
// subset on every run.  Slower than what it was meant to replace, so it stops here.

// The stream kernel: offset batches of SHORT strings (URLs, queries, log lines) walked as what they are in memory --
// one contiguous text.  DESIGN.md section 7 (ragged kernel).
//
// The ragged kernel gives every lane a string and a 128-byte window per iteration; on strings of ~100 bytes a window is
// 58 % full and the cost of an iteration is per iteration, whatever is in it (profiles/r02_ragged_clocks.log).  Here the
// unit of work is not a string but a SPAN of text: the batch [offsets[0], offsets[n]) is cut into spans of `S` bytes
// (128-byte aligned addresses), a task is 64 consecutive spans, lane l of the wave that owns the task walks span l --
// with the tiled kernel's load path (whole lines, 8 lanes per line, `nt`, register transpose) -- and is responsible for
// the strings that START in its span: it skips the bytes in front of the first one, walks every string to its end
// (the last one usually runs into the next lane's span: the wave then walks a few tiles more), and resets its state
// to the start state at every boundary.
//
//   * boundaries: the wave finds the first string of its task with a 64-ary search over the offsets (4 dependent
//     loads for 16 M strings), copies the task's offsets -- relative to the task, 32 bits -- into its slice of LDS, and
//     every lane finds its own first string there (binary search in LDS).
//   * the walk of a 16-byte chunk in which SOME lane has a boundary is branch free: before byte r (the lane's distance
//     to its boundary) the state is saved and replaced by the start state -- one v_cmp + two v_cndmask per byte on top
//     of the v_perm -> ds_read_u8 chain.  The end-of-string work (outputs, counters, the next boundary from LDS) is done
//     once per chunk for the lanes that had one.  Chunks without any boundary take the ordinary StepChunk.
//   * anything irregular goes through ONE exact routine per lane (ExactChunk): a trap (the chunk left the dense rows),
//     two boundaries inside one chunk (strings shorter than 16 bytes, empty strings).
//   * a task with more string starts than the LDS slice holds (tiny or empty strings en masse), or whose last string
//     ends more than 4 GiB behind the task's start, is walked by the wave one string per lane, straight from memory
//     (TaskFallback): slow, exact -- the kernel is correct for every batch, the hint that selects it
//     (PIRE_HIP_RUN_SHORT) is about speed only.
//
// Everything here is the plain walk (no resume states, no actions); results as everywhere: StateIndex + Final per
// string, block-local match counters.

#include "device_common.h"

namespace pirehip {

namespace {

constexpr uint32_t kStreamListCap = 1024;                     // string starts of a task kept in LDS ...
constexpr uint32_t kStreamListWords = kStreamListCap + 64;    // ... loaded 64 at a time, one entry beyond the last start
constexpr uint32_t kStreamInf = 0xFFFFFFFFu;
constexpr uint32_t kStreamFar = 0xFFFFFFF0u;                  // relative positions from here up: not representable

typedef __attribute__((address_space(3))) uint32_t* LdsWordPtr;
__device__ __forceinline__ uint32_t ListAt(uint32_t base, uint32_t k)
{
	return *reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(base + 4u * k));
}
// (written through the same address space: a store through a generic pointer is a FLAT instruction, and nothing orders a
// flat store to LDS with the ds_read of another lane that follows it)
__device__ __forceinline__ void ListPut(uint32_t base, uint32_t k, uint32_t v)
{
	*reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(base + 4u * k)) = v;
}

// 8 x global_load_dwordx4 nt, 8 adjacent lanes per line (the tiled kernel's pattern): wave-uniform base, 32-bit
// per-lane offsets, clamped so that a span behind the end of the text re-reads the text's last line instead of touching
// memory that may not be there.
__device__ __forceinline__ void IssueSpanTile(u32x4 (&r)[8], uint64_t base, uint32_t tileOff, uint32_t stride, uint32_t room,
                                              bool hasText)
{
	const uint32_t lane = threadIdx.x & 63, col = (lane & 7u) * 16;
	const uint32_t first = hasText ? (lane & ~7u) * stride + col + tileOff : col;   // string (lane & ~7), chunk lane & 7
	const uint32_t last = room + col;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		uint32_t o = first + uint32_t(j) * stride;
		o = o < last ? o : last;
		asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "+v"(r[j]) : "v"(o), "s"(base));
	}
}

template <int BEHIND>
__device__ __forceinline__ void WaitSpanTile(u32x4 (&r)[8])
{
	asm volatile("s_waitcnt vmcnt(%8)"
	             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
	             : "n"(BEHIND * 8));
}

// First i in [0, n] with offsets[i] >= key (n if none); the wave searches together, 64 probes a round.
__device__ __forceinline__ uint64_t WaveLowerBound(const uint64_t* off, uint64_t n, long long key, uint32_t lane)
{
	uint64_t lo = 0, hi = n;   // the answer is in [lo, hi]
	while (hi - lo > 64) {
		const uint64_t step = (hi - lo + 63) / 64;
		const uint64_t q = lo + uint64_t(lane) * step;
		const bool less = q < hi && (long long)off[q] < key;
		const uint32_t c = uint32_t(__popcll(__ballot(less)));   // probes are sorted: the trues are a prefix
		if (c == 0) {
			hi = lo;
		} else {
			const uint64_t nhi = lo + uint64_t(c) * step;
			lo = lo + uint64_t(c - 1) * step + 1;
			hi = nhi < hi ? nhi : hi;
		}
	}
	const uint64_t q = lo + lane;
	const bool less = q < hi && (long long)off[q] < key;
	return lo + uint64_t(__popcll(__ballot(less)));
}

struct StreamLane {
	uint32_t nextB;    // position (relative to the task) of this lane's next boundary, kStreamInf when it has none
	uint32_t nextK;    // list index of the string that starts at nextB, if this lane owns it; the one being walked
	                   // (when !skip) is nextK - 1
	bool skip;         // the bytes in front of the boundary belong to nobody this lane answers for
	uint32_t q0, q1;   // strings finished in this tile and not yet written out: list index << 8 | end state (a dense-row
	                   // id), kStreamInf = empty slot.  Written out once per tile: the end-of-string work of a wave costs
	                   // about as much as walking a chunk, whoever of its lanes takes part
};

// Per lane, divergent: one finished string (record from LDS for a dense-row state, from memory otherwise).
__device__ __forceinline__ void FinishOne(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                          uint32_t s, uint32_t st)
{
	const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
	u32x4 raw;
	if (st < p.hot)
		raw = *reinterpret_cast<const u32x4*>(&finHot[st]);
	else
		raw = *reinterpret_cast<const u32x4*>(&recs[st]);
	const uint32_t orig = raw.x, endPerm = raw.y & 0x0FFFFFFFu, fl = raw.y >> 28;
	if (p.outIdx)
		p.outIdx[s] = orig;
	if (p.outFinal)
		p.outFinal[s] = fl & kFinal;
	if (p.outCounts) {
		uint32_t* cnt = reinterpret_cast<uint32_t*>(lds + L.countsOff);
		if (fl & kFinal)
			atomicAdd(&cnt[0], 1u);
		atomicAdd(&cnt[1], 1u);
		if (p.acceptMaskPerm) {
			uint64_t m = (uint64_t(raw.w) << 32) | raw.z;
			while (m) {
				atomicAdd(&cnt[2 + __builtin_ctzll(m)], 1u);
				m &= m - 1;
			}
		} else {
			for (uint64_t k = p.acceptOffPerm[endPerm]; k < p.acceptOffPerm[endPerm + 1]; ++k)
				atomicAdd(&cnt[2 + p.acceptIds[k]], 1u);
		}
	}
}

// The same for the lanes `active` of a wave at once (ballots for the counters); the record of a state outside the dense
// rows under a wave-uniform branch of its own, as in the ragged kernel.
__device__ __forceinline__ void FinishWave(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                           uint32_t s, bool active, uint32_t st)
{
	const bool cold = active && st >= p.hot;
	u32x4 raw = {0, 0, 0, 0};
	if (active && !cold)
		raw = *reinterpret_cast<const u32x4*>(&finHot[st]);
	if (__any(cold)) {
		if (cold) {
			const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
			raw = *reinterpret_cast<const u32x4*>(&recs[st]);
			asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));
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

// Write out what the lanes have queued (at most two strings each).
__device__ __forceinline__ void FlushFinished(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                              uint32_t firstString, StreamLane& S)
{
	if (__any(S.q0 != kStreamInf)) {
		const bool act = S.q0 != kStreamInf;
		FinishWave(p, lds, L, finHot, firstString + (S.q0 >> 8), act, S.q0 & 0xFFu);
		S.q0 = kStreamInf;
	}
	if (__any(S.q1 != kStreamInf)) {
		const bool act = S.q1 != kStreamInf;
		FinishWave(p, lds, L, finHot, firstString + (S.q1 >> 8), act, S.q1 & 0xFFu);
		S.q1 = kStreamInf;
	}
}

// The boundary at nextB has been reached: the string that starts there becomes the current one if this lane owns it
// (it starts inside the lane's span), else the lane has nothing more to do in this task.
__device__ __forceinline__ void Advance(StreamLane& S, uint32_t listBase, uint32_t strings, uint32_t spanHi)
{
	const bool own = S.nextK < strings && ListAt(listBase, S.nextK) < spanHi;
	S.nextK += 1;
	S.skip = !own;
	S.nextB = own ? ListAt(listBase, S.nextK) : kStreamInf;
}

// Exact walk of one chunk with every boundary in it, one lane at a time (divergent callers): `pos` = position of the
// chunk's first byte.  Handles what the branch-free walk cannot: a chunk that left the dense rows, several boundaries
// in one chunk.  A boundary at pos + 16 is handled here too (after the last byte), like the fast walk does.
__device__ __forceinline__ void ExactChunk(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                           u32x4 v, uint32_t pos, StreamLane& S, uint32_t listBase, uint32_t strings,
                                           uint32_t spanHi, uint32_t firstString, uint32_t& hs, uint32_t& cold,
                                           bool sample)
{
	uint32_t st = hs != p.hot ? hs : cold;
#pragma unroll 1
	for (uint32_t b = 0; b <= 16; ++b) {
		while (S.nextB == pos + b) {
			if (!S.skip)
				FinishOne(p, lds, L, finHot, firstString + S.nextK - 1, st);
			Advance(S, listBase, strings, spanHi);
			st = p.startPerm;
		}
		if (b < 16) {
			st = SlowStep(p, lds, L, st, v.x & 0xFF);
			v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
			v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
			v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
			v.w >>= 8;
		}
	}
	hs = st < p.hot ? st : p.hot;
	cold = st;
	if (sample && st >= p.hot && !S.skip && !(p.flags & kDebugNoColdCount))
		atomicAdd(&p.visitCold[st], 1u);   // feeds pire_hip_table_adapt(), sampled like everywhere
}

// StepChunk without the compact tier (its LDS is the offset slices here): 16 lookups, exact re-walk of a chunk that
// left the dense rows.
__device__ __forceinline__ void PlainChunk(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const u32x4 v,
                                           uint32_t& hs, uint32_t& cold, uint32_t sampleLane)
{
	const uint32_t hs0 = hs;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0400u));
		hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0401u));
		hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0402u));
		hs = HotLookup(__builtin_amdgcn_perm(hs, x, 0x0c0c0403u));
	}
	if (hs == p.hot && !(p.flags & kDebugNoTrap)) {
		const uint32_t f = SlowChunk(p, lds, L, v, hs0 != p.hot ? hs0 : cold);
		if (f < p.hot) {
			hs = f;
		} else {
			cold = f;
			if ((threadIdx.x & 63) == sampleLane && !(p.flags & kDebugNoColdCount))
				atomicAdd(&p.visitCold[f], 1u);
		}
	}
}

// One chunk for the whole wave.
__device__ __forceinline__ void StreamChunk(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                            const u32x4 v, uint32_t pos, StreamLane& S, uint32_t listBase,
                                            uint32_t strings, uint32_t spanHi, uint32_t firstString, uint32_t startHs,
                                            uint32_t& hs, uint32_t& cold, uint32_t sampleLane)
{
	const uint32_t r = S.nextB - pos;           // distance to this lane's boundary (huge when it has none)
	const bool hasB = r <= 16u;
	// a lane in front of its first string, or behind its last one, walks bytes it does not answer for: its state is
	// pinned to the start state so that it neither traps nor drags the exact routine in
	const bool dead = S.skip && !hasB;
	hs = dead ? startHs : hs;
	if (!__any(hasB)) {
		if (p.stride & 2)
			StepChunk<0>(p, lds, L, v, hs, cold, sampleLane);
		else
		PlainChunk(p, lds, L, v, hs, cold, sampleLane);   // nobody ends here: the ordinary step (with its own trap path)
		return;
	}
	// a second boundary of the same lane inside this chunk (a string shorter than the rest of the chunk)?
	uint32_t e2 = kStreamInf;
	bool own2 = false;
	if (hasB) {
		own2 = S.nextK < strings && ListAt(listBase, S.nextK) < spanHi;
		if (own2)
			e2 = ListAt(listBase, S.nextK + 1);
	}
	const bool multi = hasB && (e2 - pos) <= 16u;
	// the walk: state saved and reset in front of byte r
	uint32_t h = hs, snap = hs;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const bool at = r == uint32_t(4 * w + j);
			snap = at ? h : snap;
			h = at ? startHs : h;
			h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0400u + uint32_t(j)));
		}
	}
	if (r == 16u) {
		snap = h;
		h = startHs;
	}
	// (a lane that was outside the dense rows already shows up as h == hot or snap == hot: row `hot` is absorbing)
	const bool exact = !dead && (multi || h == p.hot || (hasB && !S.skip && snap == p.hot));
	if (__any(exact) && !(p.flags & kDebugNoTrap)) {
		if (exact)
			ExactChunk(p, lds, L, finHot, v, pos, S, listBase, strings, spanHi, firstString, hs, cold,
			           (threadIdx.x & 63) == sampleLane);
	}
	const bool fast = !exact;
	const bool done = fast && hasB;
	const bool push = done && !S.skip;
	if (__any(push && S.q1 != kStreamInf))   // a third end in one tile: write the queue out first (strings < 43 bytes)
		FlushFinished(p, lds, L, finHot, firstString, S);
	if (push) {
		const uint32_t e = ((S.nextK - 1) << 8) | snap;
		S.q1 = S.q0 != kStreamInf ? e : S.q1;
		S.q0 = S.q0 != kStreamInf ? S.q0 : e;
	}
	if ((p.stride & 1) && __any(push))   // debugging: no queue
		FlushFinished(p, lds, L, finHot, firstString, S);
	if (done) {
		S.nextK += 1;
		S.skip = !own2;
		S.nextB = e2;     // kStreamInf when the string that starts here is somebody else's
	}
	if (fast)
		hs = h;
}

// A task the LDS slice cannot describe: its strings one per lane, straight from memory, exact steps.
__device__ __forceinline__ void TaskFallback(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                             uint64_t first, uint64_t end, uint32_t lane)
{
	for (uint64_t s = first + lane; s < end; s += 64) {
		const uint8_t* q = p.text + p.offsets[s];
		const uint8_t* e = p.text + p.offsets[s + 1];
		uint32_t st = p.startPerm;
		for (; q < e; ++q)
			st = SlowStep(p, lds, L, st, *q);
		FinishOne(p, lds, L, finHot, uint32_t(s), st);
	}
}

}  // namespace

__global__ __launch_bounds__(1024) void ScanStreamKernel(ScanParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	FinRec* finHot = reinterpret_cast<FinRec*>(lds + L.total);
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t listBase = L.total + kRaggedFinBytes + wave * kStreamListWords * 4;   // LDS byte address
	{
		const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			finHot[i] = recs[i];
	}
	LoadTableToLds(p, lds, L);   // ends with a barrier

	const uint64_t T = reinterpret_cast<uint64_t>(p.text);
	const uint64_t o0 = Uniform64(p.offsets[0]), oN = Uniform64(p.offsets[p.n]);   // wave-uniform: keep them in SGPRs
	const uint32_t S = uint32_t(p.len);                 // bytes per span, a multiple of 128
	const uint32_t spanTiles = S / 128;
	const uint64_t taskBytes = 64ull * S;
	const uint64_t A0 = (T + o0) & ~uint64_t(127), Aend = T + oN;
	// a string that starts at the very end (empty strings behind the text) starts in the task that holds address Aend
	const uint64_t ntasks = oN > o0 ? (Aend - A0) / taskBytes + 1 : 1;
	const uint64_t lastChunkLine = oN > o0 ? ((Aend - 1) & ~uint64_t(127)) : 0;   // the line of the text's last byte
	const uint32_t startSt = p.startPerm;
	const uint32_t startHs = startSt < p.hot ? startSt : p.hot;

	u32x4 cur[8], nxt[8];
	ZeroTile(cur);
	ZeroTile(nxt);
	const uint64_t taskStep = uint64_t(gridDim.x) * (blockDim.x >> 6);
	for (uint64_t task = uint64_t(blockIdx.x) * (blockDim.x >> 6) + wave; task < ntasks; task += taskStep) {
		const uint64_t taskAbs = A0 + task * taskBytes;
		const long long relLo = (long long)(taskAbs - T), relHi = relLo + (long long)taskBytes;
		// the strings that START in [relLo, relHi): offsets are sorted, start(i) = offsets[i]
		const uint64_t f0 = WaveLowerBound(p.offsets, p.n, relLo, lane);
		if (f0 >= p.n)
			continue;   // (uniform) nothing starts here or later
		// their offsets, relative to the task, into this wave's LDS slice; one entry beyond the last start (its end)
		uint32_t starts = 0;       // list entries < taskBytes that are string starts (index < n)
		bool overflow = oN == o0;  // no text at all: only empty strings, the slice cannot be trusted to hold them
		bool listed = false;
		const uint32_t tripRounds = (p.stride & 4) ? 1u : 4u;   // debugging: one round per trip
		for (uint32_t base0 = 0; !overflow && !listed; base0 += 64 * tripRounds) {
			// four rounds of 64 offsets per trip: the loads of a trip are independent, a trip costs one memory latency
			long long v4[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const uint64_t idx = f0 + base0 + 64 * q + lane;
				v4[q] = idx <= p.n ? (long long)p.offsets[idx] - relLo : (long long)0x7FFFFFFFFFFFFFFFll;
			}
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				if (overflow || listed || uint32_t(q) >= tripRounds)
					break;
				const uint32_t base = base0 + 64 * q;
				const uint64_t idx = f0 + base + lane;
				const long long v = v4[q];
				const bool beyond = idx > p.n || v >= (long long)taskBytes;
				const bool isStart = !beyond && idx < p.n;
				ListPut(listBase, base + lane, v >= (long long)kStreamFar ? kStreamFar : uint32_t(v));
				starts += uint32_t(__popcll(__ballot(isStart)));
				const unsigned long long bb = __ballot(beyond || idx >= p.n);
				if (bb) {
					// entry `starts` ends the last string that starts here: it must be representable
					const long long endV = __shfl(v, int(starts - base));
					overflow = endV >= (long long)kStreamFar;
					listed = true;
				} else if (base + 64 >= kStreamListCap) {
					overflow = true;
				}
			}
		}
		if (starts == 0 && !overflow)
			continue;
		if (overflow) {
			// this task's strings one per lane; where they end: the first start of the next task
			const uint64_t f1 = WaveLowerBound(p.offsets, p.n, relHi, lane);
			TaskFallback(p, lds, L, finHot, f0, f1, lane);
			continue;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();

		// ---- this lane's span, its first string, how far its last string reaches
		const uint32_t spanLo = lane * S, spanHi = spanLo + S;
		uint32_t kf, kl;   // first list index with start >= spanLo / >= spanHi
		{
			uint32_t lo = 0, hi = starts;
			while (__any(lo < hi)) {
				const uint32_t mid = (lo + hi) >> 1;
				const bool less = lo < hi && ListAt(listBase, mid) < spanLo;
				const bool geq = lo < hi && !less;
				lo = less ? mid + 1 : lo;
				hi = geq ? mid : hi;
			}
			kf = lo;
			lo = kf;
			hi = starts;
			while (__any(lo < hi)) {
				const uint32_t mid = (lo + hi) >> 1;
				const bool less = lo < hi && ListAt(listBase, mid) < spanHi;
				const bool geq = lo < hi && !less;
				lo = less ? mid + 1 : lo;
				hi = geq ? mid : hi;
			}
			kl = lo;
		}
		const bool owns = kl > kf;
		StreamLane Sl;
		Sl.skip = true;
		Sl.nextK = kf;
		Sl.nextB = owns ? ListAt(listBase, kf) : kStreamInf;
		// tiles this lane needs: up to the end of its last string
		uint32_t need = 0;
		if (owns) {
			const uint32_t myEnd = ListAt(listBase, kl);           // end of string kl-1 (entry kl exists: one beyond)
			need = (myEnd - spanLo + 127u) / 128u;
			need = need ? need : 1u;                               // an empty string at the span's first byte still needs a visit
		}
		uint32_t tiles = need;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			const uint32_t o = uint32_t(__shfl_xor(int(tiles), off));
			tiles = o > tiles ? o : tiles;
		}
		tiles = uint32_t(__builtin_amdgcn_readfirstlane(int(tiles)));
		(void)spanTiles;

		// ---- the walk: tile t = bytes [128 t, 128 t + 128) of every span, two register tiles
		// (a task that starts exactly where the text ends holds nothing but empty strings: its loads go to the last line)
		const bool hasText = lastChunkLine >= taskAbs;
		const uint64_t loadBase = Uniform64(hasText ? taskAbs : lastChunkLine);
		const uint64_t room = hasText ? lastChunkLine - taskAbs : 0;
		const uint32_t room32 = uint32_t(room < 0xFFFF0000ull ? room : 0xFFFF0000ull);
		uint32_t hs = startHs, cold = startSt;
		Sl.q0 = Sl.q1 = kStreamInf;
		if (p.stride & 8)
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		IssueSpanTile(nxt, loadBase, 0u, S, room32, hasText);
#pragma unroll 1
		for (uint32_t t = 0; t < tiles; ++t) {
			// the tile that has landed moves into the walk registers, its slot takes the next tile: ONE instance of the
			// walk code (the kernel has to stay inside the instruction cache), 32 v_mov per tile
			WaitSpanTile<0>(nxt);
#pragma unroll
			for (int k = 0; k < 8; ++k)
				cur[k] = nxt[k];
			if (t + 1 < tiles)
				IssueSpanTile(nxt, loadBase, (t + 1) * 128, S, room32, hasText);
			TransposeTile(cur, lane);
			if (lane == (t & 63) && !Sl.skip && !(p.flags & kDebugNoHist))
				atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + hs, 1u);
#pragma unroll 1
			for (uint32_t half = 0; half < 2; ++half) {
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					StreamChunk(p, lds, L, finHot, cur[k], spanLo + t * 128 + half * 64 + 16 * k, Sl, listBase, starts, spanHi,
					            uint32_t(f0), startHs, hs, cold, (t * 8 + half * 4 + k) & 63);
					__builtin_amdgcn_sched_barrier(0);   // one chunk at a time: nothing of the next one is hoisted over this one
				}
#pragma unroll
				for (int k = 0; k < 4; ++k)
					cur[k] = cur[k + 4];
			}
			FlushFinished(p, lds, L, finHot, uint32_t(f0), Sl);
		}
		__builtin_amdgcn_wave_barrier();   // the slice is rewritten by the next task
	}
	FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ launcher

bool StreamEligible(const ScanParams& p)
{
	// plain walks over offsets, device memory; batches worth more than a few waves
	return p.offsets != nullptr && !p.ends && !p.initIdx && !(p.flags & kPermIds) && p.n >= 256 && p.n < (1ull << 32);
}

int LaunchStream(const ScanParams& p0, hipStream_t stream)
{
	if (int rc = CheckCounts(p0))
		return rc;
	ScanParams p = p0;
	p.compact = 0;   // the LDS behind the dense rows holds the waves' offset slices here, not the compact rows
	uint32_t span = 1024;
	if (const char* e = getenv("PIRE_HIP_STREAM_SPAN")) {   // tuning: bytes per lane and task (a multiple of 128)
		const int v = atoi(e);
		if (v >= 128 && v <= 65536 && v % 128 == 0)
			span = uint32_t(v);
	}
	p.len = span;
	p.stride = getenv("PIRE_HIP_STREAM_DEBUG") ? uint64_t(atoi(getenv("PIRE_HIP_STREAM_DEBUG"))) : 0;
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	const uint32_t ldsBytes = L.total + kRaggedFinBytes + 16 * kStreamListWords * 4;
	if (ldsBytes > kLdsPerBlock)
		return LaunchGeneric(p0, stream);   // (cannot happen with the compact tier sized as it is; be safe)
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ScanStreamKernel),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	NoteKernel("stream", "pirehip::ScanStreamKernel");
	hipLaunchKernelGGL(ScanStreamKernel, dim3(unsigned(cus)), dim3(1024), ldsBytes, stream, p);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "stream kernel launch");
	return PIRE_HIP_OK;
}

}  // namespace pirehip
