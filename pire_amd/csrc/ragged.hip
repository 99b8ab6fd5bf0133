// The ragged kernel (offset batches, pire_hip_run).  DESIGN.md section 4.4.

#include "device_common.h"
#include "wide_common.h"

namespace pirehip {

// ------------------------------------------------------------------------------------------ ragged kernel
// Variable-length strings given by offsets -- the natural input of the reference's callers (URLs, log lines, one
// Runner per string: bench.cpp:244, pigrep.cpp:42).  One string per lane, but a lane is NOT tied to a string: when
// its string ends it takes the next one, so a wave stays full however uneven the lengths are.
//
//   * work distribution: blocks take ranges of `grab.block` strings from one global counter (a few thousand atomics
//     per launch, not one per wave: same-address device atomics run at well under 100 per microsecond); waves take
//     64 strings at a time from their block's range in LDS; lanes take single strings from their wave's range by
//     ballot + mbcnt.
//   * every lane walks its string in windows of up to 128 bytes.  A window starts at the string's current byte
//     whatever its alignment, so a string of <= 128 bytes is ONE window; a longer string cuts its first window at the
//     next 128-byte line boundary and reads whole aligned lines from then on.  The 64 windows of a wave are fetched by
//     groups of 8 lanes (IssueTileGroup: 8 lanes x 16 bytes = one window per instruction and group) and transposed in
//     registers.  The window of the NEXT iteration -- the same string's next 128 bytes, or the first window of
//     the lane's pending next string, whose offsets were fetched an iteration earlier -- is in flight while the
//     current one is walked; nothing on the common path makes the compiler wait for memory during the walk (the
//     end-of-string records of the hot states are in LDS for that reason).
//   * a window is walked as whole 16-byte chunks with the LDS fast path of the tiled kernel, then ONE pass for the
//     <= 15 bytes behind the last whole chunk of every lane that has some: each lane picks its chunk, walks all 16
//     bytes unrolled and keeps the state after its last real byte (no loop, no branches).
//   * nothing is read past the 16-byte block that holds the last byte of the text: a window that would reach further
//     is walked byte by byte from memory instead.


// How many strings a block takes from the global counter at a time, and a wave from its block's range (one visit of
// the block lock each).
struct RaggedGrab {
	uint32_t block, wave;
};

struct RaggedWork {
	// the block's current range of string indices, in ONE word: end << 32 | next.  Waves take from it with a plain
	// ds_add_rtn_u64 (no lock: with short strings every wave comes here every iteration or two, and queueing 16 waves
	// behind a spin lock was the largest single cost of such batches); only the refill from the global counter, a few
	// times per block and launch, is serialised.
	unsigned long long range;
	unsigned long long pad0;
	uint32_t lock, exhausted;
	uint32_t pad[2];
};
static_assert(kRaggedFinBytes + sizeof(RaggedWork) == kRaggedLdsExtra, "LDS budget of the warm rows (internal.h)");

// (IssueTileLane / IssueTileGroup / WaitAllLoads: device_common.h, shared with the stream kernel.)

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
		}
		// (the sample: the state in front of a drawn one of the chunk's `count` steps, as TrapChunk's -- not the state behind
		// the last one: that is the state the STRING ends in, which nothing looks up; rounds 2-5 gave rows to those)
		if ((threadIdx.x & 63) == sampleLane && !(p.flags & kDebugNoColdCount)) {
			const uint32_t step = (((sampleLane + blockIdx.x * 0x632BE5ABu) * 0x9E3779B1u) >> 28) % count;
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
		}
	}
}

// ------------------------------------------------------------------------------------------ scans with actions
// HalfFinalScanner counting and the prefix searches are the same walk plus something to do whenever the walk is in a
// Final state.  The fast path does not look: it only keeps the MAXIMUM of the 16 dense-row ids a chunk went through
// (one v_max3_u32 per two bytes).  The hot set is ordered non-final first and the trap id is the largest of all, so
// "max >= hotFinalLo" says "this chunk visited a hot Final state or left the dense rows": only then is the chunk
// re-walked exactly, from its start state, with the action called after every step.  Scans that are rarely in a
// Final state run at nearly the speed of the plain ragged kernel; scans that always are degrade to the exact walk
// the one-string-per-lane kernels of exact.hip do all the time.
#ifdef PIRE_HIP_RAGGED_LANE_LOADS   // A/B builds (tools/ab): every kernel with the per-lane loads of round 1
constexpr bool kRaggedGroupLoads = false;
#else
constexpr bool kRaggedGroupLoads = true;
#endif

struct NoAct {
	static constexpr bool kActive = false;
	static constexpr bool kGroupLoads = kRaggedGroupLoads;
	struct Lane {};
};

// Pire::HalfFinalScanner (scanners/half_final.h:137-164): Initialize and every Step end with TakeAction -- a Final
// state bumps the match counter of every regexp in its final list.  Up to 8 regexps, counters in registers, fed
// from the packed increment word of the state (table.cpp, inc64).
struct HalfFinalAct {
	static constexpr bool kActive = true;
	static constexpr bool kGroupLoads = kRaggedGroupLoads;
	static constexpr uint32_t kWideMask = kFinal;   // what a chunk of the wide walk is walked again for (WideChunkAct)
	static constexpr bool kBulk = true;             // ... and Bulk() below is there for chunks that start in a Final state for good
	uint32_t* results;
	struct Lane {
		uint32_t c[8];
	};
	__device__ __forceinline__ bool Wants(const Lane&) const { return true; }
	__device__ __forceinline__ uint32_t Threshold(const ScanParams& p) const { return p.hotFinalLo; }
	// next to the table in LDS: the packed increments of the dense-row states (2 KiB), and the first half of their
	// end-of-string records (StateIndex, end state | flags: 2 KiB) -- the part of FinishRagged()'s records this walk needs
	__device__ __forceinline__ void LoadLds(const ScanParams& p, uint8_t* area) const
	{
		uint64_t* incHot = reinterpret_cast<uint64_t*>(area);
		uint2* endHot = reinterpret_cast<uint2*>(area + 2048);
		const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x) {
			incHot[i] = p.incPerm[i];
			endHot[i] = make_uint2(recs[i].orig, recs[i].permFlags);
		}
	}
	__device__ __forceinline__ void Take(Lane& al, uint64_t inc) const
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
			al.c[r] += uint32_t(inc >> (8 * r)) & 0xFFu;
	}
	// `steps` steps that all end in `st` again, a Final state whose every transition is a self loop (what a Surround()ed
	// dictionary is behind a match): the action of each of them at once (the wide walk, WideChunkAct)
	__device__ __forceinline__ void Bulk(const ScanParams& p, Lane& al, uint32_t st, uint32_t steps) const
	{
		const uint64_t inc = p.incPerm[st];
#pragma unroll
		for (int r = 0; r < 8; ++r)
			al.c[r] += steps * (uint32_t(inc >> (8 * r)) & 0xFFu);
	}
	// after a step inside the dense rows that ended in h >= Threshold()
	__device__ __forceinline__ void HotStep(const ScanParams&, const uint8_t*, const LdsLayout&, const uint8_t* area,
	                                        Lane& al, uint32_t h, uint64_t) const
	{
		Take(al, reinterpret_cast<const uint64_t*>(area)[h]);
	}
	__device__ __forceinline__ void Step(const ScanParams& p, const uint8_t*, const LdsLayout&, Lane& al, uint32_t st,
	                                     uint64_t) const
	{
		if (IsFinalState(p, st))
			Take(al, p.incPerm[st]);
	}
	__device__ __forceinline__ uint32_t Start(const ScanParams& p, const uint8_t*, const LdsLayout&, Lane& al, uint32_t,
	                                          uint64_t) const
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
			al.c[r] = p.hfStartC[r];                       // Initialize [+ Begin] end with TakeAction: walked by the host
		return p.hfStart;
	}
	// End of string: Step(EndMark) if asked -- it ends with TakeAction like every step -- then the outputs.  The end
	// state of a dense-row state comes from LDS; anything that needs memory sits under a wave-uniform branch of its
	// own and is waited for in there (a load on the common path would drain the prefetched window, see FinishRagged).
	__device__ __forceinline__ void Finish(const ScanParams& p, const uint8_t*, const LdsLayout&, const uint8_t* area,
	                                       Lane& al, uint32_t s, uint32_t st, uint64_t) const
	{
		const uint64_t* incHot = reinterpret_cast<const uint64_t*>(area);
		const uint2* endHot = reinterpret_cast<const uint2*>(area + 2048);
		const bool cold = st >= p.hot;
		uint32_t orig = 0, pf = 0;
		if (!cold) {
			const uint2 r = endHot[st];
			orig = r.x;
			pf = r.y;
		}
		if (__any(cold)) {
			if (cold) {
				const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
				const uint2 r = *reinterpret_cast<const uint2*>(&recs[st]);
				orig = r.x;
				pf = r.y;
				asm volatile("" : "+v"(orig), "+v"(pf));   // the wait belongs in here
			}
		}
		const uint32_t endSt = pf & 0x0FFFFFFFu, fl = pf >> 28;
		if (p.flags & PIRE_HIP_RUN_END) {
			const bool fin = (fl & kFinal) != 0;
			if (__any(fin)) {
				uint32_t lo = 0, hi = 0;
				if (fin && endSt < p.hot) {
					const uint64_t inc = incHot[endSt];
					lo = uint32_t(inc);
					hi = uint32_t(inc >> 32);
				}
				if (__any(fin && endSt >= p.hot)) {
					if (fin && endSt >= p.hot) {
						const uint64_t inc = p.incPerm[endSt];
						lo = uint32_t(inc);
						hi = uint32_t(inc >> 32);
						asm volatile("" : "+v"(lo), "+v"(hi));
					}
				}
				Take(al, (uint64_t(hi) << 32) | lo);
			}
		}
#pragma unroll
		for (int r = 0; r < 8; ++r)      // static indices only: a runtime index would put c[] into scratch
			if (uint32_t(r) < p.regexps)
				results[size_t(s) * p.regexps + r] = al.c[r];
		if (p.outIdx)
			p.outIdx[s] = orig;
		if (p.outFinal)
			p.outFinal[s] = fl & kFinal;
	}
};

// The same for tables whose counters do not pack (more than 8 regexps, or a state that bumps one counter more than
// 255 times): the lane owns row s of the result array and read-modify-writes it -- only in the exact re-walks, i.e.
// where a Final state really was visited.
struct HalfFinalWideAct {
	static constexpr bool kActive = true;
	static constexpr bool kGroupLoads = kRaggedGroupLoads;
	static constexpr uint32_t kWideMask = kFinal;
	static constexpr bool kBulk = false;
	uint32_t* results;
	struct Lane {
		uint32_t* row;
	};
	__device__ __forceinline__ bool Wants(const Lane&) const { return true; }
	__device__ __forceinline__ uint32_t Threshold(const ScanParams& p) const { return p.hotFinalLo; }
	__device__ __forceinline__ void LoadLds(const ScanParams&, uint8_t*) const {}
	__device__ __forceinline__ void Step(const ScanParams& p, const uint8_t*, const LdsLayout&, Lane& al, uint32_t st,
	                                     uint64_t) const
	{
		if (IsFinalState(p, st))
			for (uint64_t k = p.acceptOffPerm[st]; k < p.acceptOffPerm[st + 1]; ++k)
				al.row[p.acceptIds[k]] += 1;
	}
	__device__ __forceinline__ void HotStep(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const uint8_t*,
	                                        Lane& al, uint32_t h, uint64_t after) const
	{
		Step(p, lds, L, al, h, after);
	}
	__device__ __forceinline__ uint32_t Start(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, Lane& al,
	                                          uint32_t s, uint64_t addr) const
	{
		al.row = results + size_t(s) * p.regexps;
		for (uint32_t r = 0; r < p.regexps; ++r)
			al.row[r] = 0;
		uint32_t st = p.startPerm;                         // Initialize ends with TakeAction, half_final.h:142
		Step(p, lds, L, al, st, addr);
		if (p.flags & PIRE_HIP_RUN_BEGIN) {
			st = p.nextPerm[size_t(st) * p.letters + p.beginCls];
			Step(p, lds, L, al, st, addr);
		}
		return st;
	}
	__device__ __forceinline__ void Finish(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const uint8_t*,
	                                       Lane& al, uint32_t s, uint32_t st, uint64_t end) const
	{
		if (p.flags & PIRE_HIP_RUN_END) {
			st = p.nextPerm[size_t(st) * p.letters + p.endCls];
			Step(p, lds, L, al, st, end);
		}
		if (p.outIdx)
			p.outIdx[s] = p.origOfPerm[st];
		if (p.outFinal)
			p.outFinal[s] = p.flagsPerm[st] & kFinal;
	}
};

// Pire::LongestPrefix / ShortestPrefix (run.h:277-311, predicates 69-100): the position after the last (first) step
// that ended in a Final state; a Dead state ends the search.  A lane whose search is over stops re-walking (its
// state no longer matters) and idles through the rest of its string.
struct PrefixAct {
	static constexpr bool kActive = true;
	// group loads like the other walks since the window addresses travel by DPP (127 VGPRs; with ds_bpermute and its
	// address temporaries this instantiation needed 12 bytes of scratch and kept per-lane loads, which made the
	// wait for the window twice the walk: profiles/r02_ragged_clocks.log)
	static constexpr bool kGroupLoads = true;
	static constexpr uint32_t kWideMask = kFinal | kDead;   // (a Dead state ends the search: the rest of the string is skipped)
	static constexpr bool kBulk = false;                    // (Step() stops the search at a Final state that is one for good)
	long long* outLen;
	uint32_t longest, throughEnd;
	uint32_t startFlags;   // flags of the state every string starts in: the host knows them (a load per string otherwise)
	struct Lane {
		uint64_t begin;
		long long pos;
		uint32_t stop;   // 1 search over, 2 Final before the first byte (shortest: answered at once), 4 ended Dead, 8 Final for good
	};
	__device__ __forceinline__ bool Wants(const Lane& al) const { return !(al.stop & 1u); }
	__device__ __forceinline__ uint32_t Threshold(const ScanParams& p) const { return p.hotDeadLo; }
	// next to the table in LDS: for every dense-row state, the flags of the state Step(EndMark) leads to -- what the end of
	// a string asks when the search runs through End().  (Round 5: the end of a string read nextPerm and the flags from
	// memory, two dependent loads whose wait also drained the window on its way; log lines end in every iteration of a
	// wave, and the searches ran at two thirds of the plain scan, profiles/r05_prefix_sizes.log.)
	__device__ __forceinline__ void LoadLds(const ScanParams& p, uint8_t* area) const
	{
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			area[i] = uint8_t(p.finEnd[i].permFlags >> 28);
	}
	__device__ __forceinline__ void HotStep(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const uint8_t*,
	                                        Lane& al, uint32_t h, uint64_t after) const
	{
		Step(p, lds, L, al, h, after);
	}
	__device__ __forceinline__ void Step(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, Lane& al,
	                                     uint32_t st, uint64_t after) const
	{
		if (al.stop & 1u)
			return;
		const uint32_t f = StateFlags(p, lds, L, st);
		if (f & kFinal) {
			al.pos = (long long)(after - al.begin);
			if (!longest)
				al.stop |= 1u;                             // ShortestPrefixPred stops on the first Final
			else if (f & kAbsorbing)
				al.stop |= 9u;                             // Final for good (every transition a self loop, marks included: what a
				                                           // Surround()ed dictionary is behind a match): the longest prefix is the
				                                           // whole string -- Finish() says so, the rest of the string is not walked
		}
		if (f & kDead)
			al.stop |= 5u;                                 // both predicates stop on a dead state
	}
	__device__ __forceinline__ uint32_t Start(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, Lane& al,
	                                          uint32_t, uint64_t addr) const
	{
		al.begin = addr;
		al.pos = -1;
		al.stop = 0;
		const uint32_t st = p.startPerm;                   // Initialize (+ BeginMark), run.h:280-283
		// (the wide walk's launch has p.hot = 0 and the flags from the host; the dense one reads them where it always did)
		if ((p.hot ? StateFlags(p, lds, L, st) : startFlags) & kFinal) {
			al.pos = 0;                                    // run.h:284 / 301-302
			if (!longest)
				al.stop = 3u;
		}
		return st;
	}
	__device__ __forceinline__ void Finish(const ScanParams& p, const uint8_t*, const LdsLayout&, const uint8_t* area,
	                                       Lane& al, uint32_t s, uint32_t st, uint64_t end) const
	{
		// a search that ended Dead stays not-Final through EndMark; one that was answered before the first byte
		// does not look at EndMark at all; a shortest prefix already found is kept (run.h:286-290 / 305-309)
		const bool asks = throughEnd && !(al.stop & 14u) && (longest || al.pos < 0);
		const bool cold = asks && st >= p.hot;
		uint32_t fl = asks && !cold ? area[st] : 0u;   // flags of the state End() leads to: LDS for a dense-row state
		if (__any(cold)) {
			if (cold) {
				fl = p.finEnd[st].permFlags >> 28;
				asm volatile("" : "+v"(fl));   // the wait belongs in here
			}
		}
		if ((asks && (fl & kFinal)) || (al.stop & 8u))
			al.pos = (long long)(end - al.begin);
		outLen[s] = al.pos;
	}
};

// Pire::CapturingScanner (extra/capture.h:49-162) on this kernel.  Its table is a LoadedScanner whose TRANSITIONS carry
// the actions (BeginCapture / EndCapture); counting.hip expands it so that an action becomes a property of the STATE
// entered -- expanded state = (state, action of the transition that entered it), at most 4 x the states, which are
// few -- and flags those with an action as the states this walk cares about (they take the place Final states have for
// the half-final counting: ordered last among the dense rows, one compare per chunk on the largest id seen).  On text
// the capture's brackets are met in a few chunks per string, so nearly every chunk takes the plain fast path.
// info[e] for expanded state e (reference numbering of the expanded table): original state << 8 | Final tag << 2 | action.
struct CaptureAct {
	static constexpr bool kActive = true;
	static constexpr bool kGroupLoads = kRaggedGroupLoads;
	static constexpr uint32_t kWideMask = kFinal;
	static constexpr bool kBulk = false;
	const uint32_t* info;
	long long* outBegin;
	long long* outEnd;
	uint32_t beginStep;   // 1 when Begin() is a counted step of the string (PIRE_HIP_RUN_BEGIN)
	static constexpr uint32_t npos = ~uint32_t(0);
	struct Lane {
		uint64_t start;
		uint32_t begin, end;
	};
	__device__ __forceinline__ bool Wants(const Lane&) const { return true; }   // Final needs the whole string anyway
	__device__ __forceinline__ uint32_t Threshold(const ScanParams& p) const { return p.hotFinalLo; }
	// next to the table in LDS: the info word of every dense-row state (1 KiB), and of the state its end of string
	// leads to (Step(EndMark) when asked for; 1 KiB at +2048) -- no load from memory on the common path
	__device__ __forceinline__ void LoadLds(const ScanParams& p, uint8_t* area) const
	{
		uint32_t* infoHot = reinterpret_cast<uint32_t*>(area);
		uint32_t* endHot = reinterpret_cast<uint32_t*>(area + 2048);
		const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x) {
			infoHot[i] = info[p.origOfPerm[i]];
			endHot[i] = info[recs[i].orig];
		}
	}
	// TakeAction, capture.h:96-102, at value `at` = m_counter - 1
	__device__ __forceinline__ void Apply(Lane& al, uint32_t a, uint32_t at) const
	{
		const bool open = !(al.begin != npos && al.end != npos);
		const bool setBegin = (a & 1u) && open;
		const bool setEnd = !(a & 1u) && (a & 2u) && open;
		al.begin = setBegin ? at : al.begin;
		al.end = setEnd ? at : al.end;
	}
	// `after` = address behind the byte just consumed: that byte was step (after - start) + beginStep, counted from 1
	__device__ __forceinline__ void HotStep(const ScanParams&, const uint8_t*, const LdsLayout&, const uint8_t* area, Lane& al,
	                                        uint32_t h, uint64_t after) const
	{
		Apply(al, reinterpret_cast<const uint32_t*>(area)[h] & 3u, uint32_t(after - al.start) + beginStep - 1u);
	}
	__device__ __forceinline__ void Step(const ScanParams& p, const uint8_t*, const LdsLayout&, Lane& al, uint32_t st,
	                                     uint64_t after) const
	{
		if (IsFinalState(p, st))   // "Final" = entered by a transition with an action (counting.hip BuildCaptureTable)
			Apply(al, info[p.origOfPerm[st]] & 3u, uint32_t(after - al.start) + beginStep - 1u);
	}
	__device__ __forceinline__ uint32_t Start(const ScanParams& p, const uint8_t*, const LdsLayout&, Lane& al, uint32_t,
	                                          uint64_t addr) const
	{
		al.start = addr;
		al.begin = al.end = npos;                          // Initialize, capture.h:89-94
		const uint32_t st = p.startPerm;                   // ... [+ Step(BeginMark): m_counter = 1, its action at 0]
		if (beginStep && IsFinalState(p, st))
			Apply(al, info[p.origOfPerm[st]] & 3u, 0u);
		return st;
	}
	__device__ __forceinline__ void Finish(const ScanParams& p, const uint8_t*, const LdsLayout&, const uint8_t* area, Lane& al,
	                                       uint32_t s, uint32_t st, uint64_t end) const
	{
		const bool cold = st >= p.hot;
		uint32_t e = cold ? 0u : reinterpret_cast<const uint32_t*>(area + 2048)[st];
		if (__any(cold)) {
			if (cold) {
				const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
				e = info[recs[st].orig];
				asm volatile("" : "+v"(e));   // the wait belongs in here
			}
		}
		if (p.flags & PIRE_HIP_RUN_END)   // the End() step is a counted step with an action of its own
			Apply(al, e & 3u, uint32_t(end - al.start) + beginStep);
		if (p.outIdx)
			p.outIdx[s] = e >> 8;
		if (p.outFinal)
			p.outFinal[s] = (e >> 2) & 1u;
		outBegin[s] = al.begin == npos ? -1ll : (long long)al.begin;
		outEnd[s] = al.end == npos ? -1ll : (long long)al.end;
	}
};

// Exact walk of the first `count` (<= 16) bytes of v with the action after every step; `addr` is the address of
// byte 0.  Rolled: this is the cold path.
template <class Act>
__device__ __forceinline__ uint32_t ActBytes(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, u32x4 v,
                                             uint32_t st, uint32_t count, const Act& act, typename Act::Lane& al,
                                             uint64_t addr)
{
#pragma unroll 1
	for (uint32_t i = 0; i < count; ++i) {
		st = SlowStep(p, lds, L, st, v.x & 0xFF);
		act.Step(p, lds, L, al, st, addr + i + 1);
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return st;
}

// The same for a chunk that stayed inside the dense rows (the usual reason for a re-walk: it touched a Final state):
// LDS only, one compare per step on top of the lookup, the action only where the compare says so.
template <class Act>
__device__ __forceinline__ uint32_t ActHotBytes(const ScanParams& p, const uint8_t* lds, const LdsLayout& L,
                                                const uint8_t* area, u32x4 v, uint32_t h, uint32_t count,
                                                const Act& act, typename Act::Lane& al, uint64_t addr)
{
	const uint32_t thr = act.Threshold(p);
#pragma unroll 1
	for (uint32_t w = 0; w < 16; w += 4) {
		const uint32_t x = v.x;
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			const uint32_t nh = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0400u + j));
			if (w + j < count) {
				h = nh;
				if (h >= thr)
					act.HotStep(p, lds, L, area, al, h, addr + w + j + 1);
			}
		}
		v.x = v.y;
		v.y = v.z;
		v.z = v.w;
	}
	return h;
}

// StepChunk with the visit test: 16 bytes through the dense rows, keeping the largest id seen.
template <class Act>
__device__ __forceinline__ void StepChunkAct(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const u32x4 v,
                                             uint32_t& hs, uint32_t& cold, const uint8_t* area, const Act& act,
                                             typename Act::Lane& al, uint64_t addr)
{
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
	if (m >= act.Threshold(p) && act.Wants(al) && !(p.flags & kDebugNoTrap)) {
		if (m < p.hot) {
			(void)ActHotBytes(p, lds, L, area, v, hs0, 16u, act, al, addr);   // ends in h again
		} else {
			const uint32_t st0 = hs0 != p.hot ? hs0 : cold;
			uint32_t st;
			if (p.actDist && p.actDist[st0] > 16) {
				// the chunk left the dense rows but cannot reach a state the action cares about: the plain
				// trap path (compact rows in LDS first) moves the state on, there is nothing else to do
				st = p.compact;
				if (st0 < p.compact)
					st = CompactChunk(p, L, v, st0);
				if (st == p.compact)
					st = SlowChunk(p, lds, L, v, st0);
			} else {
				st = ActBytes(p, lds, L, v, st0, 16u, act, al, addr);
			}
			hs = st < p.hot ? st : p.hot;
			cold = st;
			// tell pire_hip_table_adapt() which rows deserve LDS, sampled like TrapChunk's.  (Round 5: the walks with actions
			// left no samples, so a caller of the prefix searches alone never saw its table adapt -- and 13 states of set_a
			// without a dense row on log lines cost the searches 40 %: 1.65 against 2.35 TB/s, profiles/r05_prefix_sizes*.log.)
			if (st >= p.hot && (threadIdx.x & 63) == ((uint32_t(addr) >> 4) & 63u)) {
				atomicAdd(&p.visitCold[st], 1u);
				atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + kLdsTrapSlot, 1u);
			}
		}
	}
}

// StepPartial with the visit test: only the first `count` steps count.
template <class Act>
__device__ __forceinline__ void StepPartialAct(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const u32x4 v,
                                               uint32_t count, uint32_t& hs, uint32_t& cold, const uint8_t* area,
                                               const Act& act, typename Act::Lane& al, uint64_t addr)
{
	const uint32_t hs0 = hs;
	uint32_t h = hs, snap = hs, m = 0;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (w == 3 && j == 3)
				break;
			h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0400u + uint32_t(j)));
			const bool in = count >= uint32_t(4 * w + j + 1);
			m = max(m, in ? h : 0u);
			snap = count == uint32_t(4 * w + j + 1) ? h : snap;
		}
	}
	hs = snap;
	if (count != 0 && m >= act.Threshold(p) && act.Wants(al) && !(p.flags & kDebugNoTrap)) {
		if (m < p.hot) {
			(void)ActHotBytes(p, lds, L, area, v, hs0, count, act, al, addr);
		} else {
			const uint32_t st0 = hs0 != p.hot ? hs0 : cold;
			uint32_t st;
			if (p.actDist && p.actDist[st0] > 16) {
				st = p.compact;
				if (st0 < p.compact)
					st = CompactPartial(p, L, v, st0, count);
				if (st == p.compact)
					st = SlowPartial(p, lds, L, v, st0, count);
			} else {
				st = ActBytes(p, lds, L, v, st0, count, act, al, addr);
			}
			hs = st < p.hot ? st : p.hot;
			cold = st;
		}
	}
}

// (FinishRagged: device_common.h, shared with the stream kernel.)

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
	uint32_t pendInit;   // its resume state (only with init states), fetched with the offsets
	bool pend;
};

// Wave-uniform range of strings still to hand out, refilled from the block's range, refilled from the global counter.
struct RaggedRange {
	uint64_t next, end;
	bool exhausted;
};

__device__ __forceinline__ void GrabWaveRange(const ScanParams& p, volatile RaggedWork* work,
                                              unsigned long long* workCounter, RaggedGrab grab, RaggedRange& R)
{
	unsigned long long r0 = 0, r1 = 0;
	if ((threadIdx.x & 63) == 0) {
		unsigned long long* range = const_cast<unsigned long long*>(&work->range);
		for (;;) {
			const unsigned long long old = atomicAdd(range, (unsigned long long)grab.wave);
			const unsigned long long nx = old & 0xFFFFFFFFull, en = old >> 32;
			if (nx < en) {
				r0 = nx;
				r1 = nx + grab.wave < en ? nx + grab.wave : en;
				break;
			}
			// the block's range is used up (the add overshot `next`: harmless, the refill rewrites the word; it cannot
			// wrap either: at most 16 waves overshoot by grab.wave <= 256 each, and n < 2^32 - 2^16 is checked by the
			// launcher)
			if (work->exhausted)
				break;
			if (atomicCAS(const_cast<uint32_t*>(&work->lock), 0u, 1u) == 0u) {
				const unsigned long long cur = *const_cast<volatile unsigned long long*>(range);
				if ((cur & 0xFFFFFFFFull) >= (cur >> 32) && !work->exhausted) {   // still empty: this wave refills
					const unsigned long long base = atomicAdd(workCounter, (unsigned long long)grab.block);
					if (base >= p.n) {
						work->exhausted = 1;
					} else {
						const unsigned long long e2 = base + grab.block < p.n ? base + grab.block : p.n;
						atomicExch(range, (e2 << 32) | base);
					}
				}
				__threadfence_block();
				atomicExch(const_cast<uint32_t*>(&work->lock), 0u);
			} else {
				__builtin_amdgcn_s_sleep(2);   // another wave is refilling
			}
		}
	}
	R.next = Uniform64(r0);
	R.end = Uniform64(r1);
	R.exhausted = R.next >= R.end;
}

// Give every lane without a pending string the next unassigned one.  Returns (per lane) whether it got one.
__device__ __forceinline__ bool AssignPending(const ScanParams& p, volatile RaggedWork* work,
                                              unsigned long long* workCounter, RaggedGrab grab, RaggedRange& R,
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
			GrabWaveRange(p, work, workCounter, grab, R);
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

#ifdef PIRE_HIP_TUNING
// timing experiments (PIRE_HIP_DEBUG_RAGGED_CLOCKS): shader-clock time of a wave per section of an iteration, summed
// over the waves into ScanParams::stamps: 0 wait for the window, 1 transpose + next window's loads, 2 assignment +
// offsets, 3 walk, 4 end of string, 5 move on (waits for the offsets), 6 before the first iteration, 7 iterations
struct RaggedClock {
	unsigned long long t, acc[8];
	bool on;
};
#define PIRE_RCLK(c, k)                                                 \
	do {                                                                \
		if ((c).on) {                                                   \
			const unsigned long long n_ = __builtin_readcyclecounter(); \
			(c).acc[k] += n_ - (c).t;                                   \
			(c).t = n_;                                                 \
		}                                                               \
	} while (0)
#else
struct RaggedClock {};
#define PIRE_RCLK(c, k) do { } while (0)
#endif

// ---- the walks with actions on the class-indexed walk (round 6) -----------------------------------------
// A dictionary scanner under LongestPrefix or as a HalfFinalScanner visits thousands of states: on the dense rows nearly every
// chunk leaves them and is walked again from memory (0.3 TB/s on log lines, profiles/r06_actions_wide.jsonl).  The same walk
// through the wide rows: the fast path asks of every state it enters whether it is Final or Dead -- plain rows: the flags
// halfword at the end of the state's row (a third LDS read per step, off the dependent chain); zipped image: bit 0 of the
// state's header, which the next step reads anyway -- and only a chunk that met such a state, or left the tier, is walked
// again with the action (WideActBytes: the rows in LDS, the table in memory for states without one, the action's own look at
// the state only where the image says "Final or Dead").  The host hands the kernel p.hot = 0: to the actions every state is one
// "without a dense row" (flags, increments and end-of-string records from memory, under wave-uniform branches of their own).
// What the image says about a state of the tier (or the escape state): plain rows -- the flags halfword at the end of its row;
// zipped -- its header, bit 0 = Final (Dead is not in there: a Dead state stays Dead, so the walks that care look at the
// flags of the state a chunk ENDS in, which a Dead state has -- absorbing states keep a row of their own, table.cpp PlanZip --
// and a Dead state they miss only costs them the walk to the end of the string, never the answer).
template <bool ZIP>
__device__ __forceinline__ uint32_t WideFlaggedBits(uint32_t st, const WideConst& K)
{
	if constexpr (ZIP)
		return *reinterpret_cast<LdsU32Ptr>(static_cast<uintptr_t>(K.hOff + (st << 2)));
	else
		return *reinterpret_cast<LdsU16Ptr>(static_cast<uintptr_t>(__umul24(st, K.pitch) + K.flagsOff + 256u));
}
template <bool ZIP, uint32_t MASK>
__device__ __forceinline__ bool WideIsFlagged(uint32_t bits, uint32_t endState, const WideConst& K)
{
	if constexpr (!ZIP)
		return (bits & MASK) != 0;
	bool f = (bits & 1u) != 0;
	if constexpr ((MASK & kDead) != 0)
		f = f || (WideFlags<true>(endState, K) & kDead) != 0;
	return f;
}

// The first `count` (<= 16) bytes of v exactly, device ids all the way, the action after every step that enters a state the
// image calls Final or Dead -- or a state outside the tier, of which the image says nothing.  Rolled: the cold path.
template <class Act, bool ZIP>
__device__ __forceinline__ uint32_t WideActBytes(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, const WideConst& K,
                                                 u32x4 v, uint32_t sid, uint32_t count, const Act& act, typename Act::Lane& al,
                                                 uint64_t addr, uint32_t sampleStep, uint32_t& sampleSid)
{
	sampleSid = 0;
#pragma unroll 1
	for (uint32_t i = 0; i < count; ++i) {
		if (i == sampleStep)
			sampleSid = sid;   // the state whose row (or table line) this step looks up: what the ranking counts
		const uint32_t c2 = HotLookup(v.x & 0xFFu);
		uint32_t next = WideEntry<ZIP>(sid < p.wide ? sid : p.wide, K, c2);
		bool look = true;
		if (next == p.wide) {
			next = WideNextC2<true>(p, sid, c2);
			asm volatile("" : "+v"(next));   // (the wait belongs in here)
		} else {
			look = WideIsFlagged<ZIP, Act::kWideMask>(WideFlaggedBits<ZIP>(next, K), next, K);
		}
		sid = next;
		if (look)
			act.Step(p, lds, L, al, sid, addr + i + 1);
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return sid;
}

// 16 bytes (count == 16) or the first `count` (0..15) of a string's last chunk through the wide rows with the visit test.
template <class Act, bool ZIP, bool PARTIAL>
__device__ __forceinline__ void WideChunkAct(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const WideLayout& W, const WideConst& K,
                                             const u32x4 v, uint32_t count, uint32_t& st, uint32_t& cold, const Act& act,
                                             typename Act::Lane& al, uint64_t addr)
{
	const uint32_t st0 = st;
	uint32_t h = st, snap = st, acc = 0;
	// A chunk that STARTS in a Final state that is one for good (every transition a self loop: its row says so, an absorbing state
	// keeps a row of its own in either image) ends there, and every step of it takes that state's action: once for all of them.
	bool absorbed = false;
	if constexpr (Act::kBulk) {
		absorbed = (WideFlags<ZIP>(st0, K) & (kFinal | kAbsorbing)) == (kFinal | kAbsorbing) && (!PARTIAL || count != 0);
		if (__any(absorbed)) {
			if (absorbed)
				act.Bulk(p, al, st0, PARTIAL ? count : 16u);
		}
	}
	if constexpr (ZIP) {
		// (a dword per trip, rolled: the zipped step's temporaries times sixteen did not fit beside the two line tiles and the
		// action's own registers -- 8 VGPR spills in the half-final instantiation, and a spill in this loop is a wrong result)
		u32x4 t = v;
#pragma unroll 1
		for (uint32_t w = 0; w < 4; ++w) {
			const uint32_t x = t.x;
			t.x = t.y;
			t.y = t.z;
			t.z = t.w;
#pragma unroll
			for (uint32_t j = 0; j < 4; ++j) {
				const uint32_t nh = WideEntry<ZIP>(h, K, HotLookup((x >> (8 * j)) & 0xFFu));
				const uint32_t bits = WideFlaggedBits<ZIP>(nh, K);
				const bool in = !PARTIAL || count >= 4 * w + j + 1;
				acc |= in ? bits : 0u;
				h = nh;
				snap = count == 4 * w + j + 1 ? h : snap;
			}
		}
	} else {
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			const uint32_t x = v[w];
			const uint32_t cl[4] = {HotLookup(x & 0xFFu), HotLookup((x >> 8) & 0xFFu), HotLookup((x >> 16) & 0xFFu), HotLookup(x >> 24)};
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (PARTIAL && w == 3 && j == 3)
					break;
				h = WideEntry<ZIP>(h, K, cl[j]);
				const uint32_t bits = WideFlaggedBits<ZIP>(h, K);
				if constexpr (PARTIAL) {
					const bool in = count >= uint32_t(4 * w + j + 1);
					acc |= in ? bits : 0u;
					snap = count == uint32_t(4 * w + j + 1) ? h : snap;
				} else {
					acc |= bits;
				}
			}
		}
	}
	st = PARTIAL ? snap : h;
	if ((!PARTIAL || count != 0) && !absorbed && (WideIsFlagged<ZIP, Act::kWideMask>(acc, st, K) || st == p.wide) && act.Wants(al)) {
		// tell pire_hip_table_adapt() which states deserve a place in the tier: one lane of 64 (by the chunk's address), the state
		// in front of ONE of the chunk's steps (by the address too: not the state the chunk ends in -- an offset batch's strings
		// start with their windows, and a state they are in at their 6th byte and nowhere else would never be seen)
		const uint32_t h = ((uint32_t(addr) >> 4) + blockIdx.x * 0x632BE5ABu) * 0x9E3779B1u;
		uint32_t seen;
		const uint32_t sid = WideActBytes<Act, ZIP>(p, lds, L, K, v, st0 < p.wide ? st0 : cold, count, act, al, addr, (h >> 22) & 15u, seen);
		st = sid < p.wide ? sid : p.wide;
		cold = sid;
		if (seen >= p.wide && (threadIdx.x & 63) == (h >> 26) && ((h >> 22) & 15u) < count) {
			atomicAdd(&p.visitCold[seen], 1u);
			atomicAdd(reinterpret_cast<uint32_t*>(lds + W.progOff) + 1, 1u);
		}
	}
}

// One iteration: start fetching the next window into `nxt`, walk the current window held in `cur`.
// Returns false when the wave has nothing left to do.
// EXT: the extensions segmented.hip needs (separate end offsets, resume states fetched with the offsets, device
// state ids in and out).  A separate instantiation: compiled into the plain kernel they cost it 2-8 % (measured
// A/B on one box: fixed 4 KiB strings 3 267 -> 3 026 GB/s), although none of it runs there.
// WIDE (round 5; 1: the exact table behind the rows has u32 entries, 2: u16, 3: u16 and the LDS image is zipped, round 6): the class-indexed walk of wide.hip instead of
// the dense rows -- S.hs is then a device id with a row or `wide` (the escape row), the state's end-of-string record comes
// from memory (the LDS is the rows'), everything else of the kernel is what it was.  Plain scans only (NoAct).
template <class Act, bool EXT, int WIDE = 0>
__device__ __forceinline__ bool RaggedPhase(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const FinRec* finHot,
                                            volatile RaggedWork* work, unsigned long long* workCounter,
                                            RaggedGrab grab, uint64_t textBase, uint64_t safeEnd, RaggedRange& R,
                                            RaggedLane& S, u32x4 (&cur)[8], u32x4 (&nxt)[8], uint32_t iter,
                                            const Act& act, typename Act::Lane& al, RaggedClock& clk,
                                            const WideLayout& W = WideLayout(), const WideConst& K = WideConst())
{
	static_assert(!(WIDE == 1 && Act::kActive), "the walks with actions take the wide walk with the u16 table only");
	WaitAllLoads(cur);
	PIRE_RCLK(clk, 0);
	if constexpr (Act::kGroupLoads)
		TransposeTile(cur, threadIdx.x & 63);

	// ---- this window: starts at the string's current byte, whatever its alignment.  A string that fits takes one
	// window; a longer one cuts its first window at a line boundary so that all the following ones are whole lines.
	uint64_t left = S.end - S.pos;
	if constexpr (Act::kActive)
		if (S.busy && !act.Wants(al))
			left = 0;   // the search is over: the rest of the string is not needed, the lane moves on
	// a string that does not fit one window cuts its first window at the next 128-byte LINE boundary: from then on it
	// reads whole aligned lines, each exactly once (cut at 16-byte boundaries only, every line of a long string was
	// fetched by two windows 5-10 us apart and often twice from HBM: 1.7 x the text, profiles/r02_ragged_pmc_*)
	const uint32_t nb = !S.busy ? 0u : left <= 128u ? uint32_t(left) : 128u - (uint32_t(S.pos) & 127u);
	const bool ends = S.busy && nb == left;

	// ---- the next window: the same string's next bytes, or the pending string's first window
	const bool cont = S.busy && !ends;
	const bool takeNew = !cont && S.pend;
	const uint64_t nPos = cont ? S.pos + nb : S.pendPos;
	const uint64_t nEnd = cont ? S.end : S.pendEnd;
	const uint32_t nIdx = cont ? S.sIdx : S.sIdxN;
	const bool nBusy = cont || takeNew;
	const bool nLoad = nBusy && nEnd > nPos && nPos + 128 <= safeEnd;
	// unconditional PER LANE (idle lanes fetch a harmless valid line): a load under a per-lane condition could be turned into
	// load-to-a-copy + select by the compiler, and the select would read the register before the data arrives.  But not when
	// NO lane has a window to fetch (round 6): that is the wave's last iteration -- nothing busy, nothing pending --, and what it
	// requested was still on its way when the kernel's epilogue took the registers (lesson 29; the waits on the way out stay,
	// but hipcc moved the epilogue's first vector instruction, `tid << 2`, in front of one of them in the dense prefix
	// instantiation: a fault in two runs of five of tests/test_random_scanners.py).  Nothing requested, nothing on its way.
	if (!(p.flags & kDebugNoRefill) && __any(nLoad)) {
		if constexpr (Act::kGroupLoads)
			IssueTileGroup(nxt, nLoad ? nPos : reinterpret_cast<uint64_t>(p.hotRows), threadIdx.x & 63);
		else
			IssueTileLane(nxt, nLoad ? nPos : reinterpret_cast<uint64_t>(p.hotRows));
	}
	if (takeNew)
		S.pend = false;
	PIRE_RCLK(clk, 1);
	// the offsets of newly assigned strings: plain loads issued AFTER the tile loads and looked at only at the very
	// end of this iteration, so the one wait the compiler inserts for them sits behind the walk
	const bool got = AssignPending(p, work, workCounter, grab, R, S);
	uint64_t offB = 0, offE = 0;
	uint32_t initV = 0;
	if (__any(got)) {
		const uint32_t which = got ? S.sIdxN : 0u;
		const uint64_t* offPtr = p.offsets + which;
		offB = offPtr[0];
		if constexpr (EXT) {
			offE = p.ends ? p.ends[which] : offPtr[1];
			if (p.initIdx)
				initV = p.initIdx[which];
		} else {
			offE = offPtr[1];
		}
	}

	PIRE_RCLK(clk, 2);
	// ---- walk the current window
	// (the walks with actions on the dense rows keep round 2's visit sample -- the state a window STARTS in, one lane per
	// iteration -- and the re-walks' own: their instantiations have no register left for the drawn byte's, below)
	constexpr bool kDrawn = WIDE != 0 || !Act::kActive;
	if (!kDrawn && (threadIdx.x & 63) == (iter & 63) && nb != 0 && !(p.flags & kDebugNoHist))
		atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + S.hs, 1u);
	if (p.flags & kDebugNoStep) {
		// timing experiments: no walk at all
	} else if (__any(nb != 0)) {
		if (__any(nb != 0 && S.loaded)) {
			const uint32_t nbl = S.loaded ? nb : 0u;
			const uint32_t full = nbl >> 4, tail = nbl & 15u;
			// The wide walk's visit sample: the state behind ONE byte of the window of ONE lane, both drawn per iteration -- every
			// lane-step of a wave's 64 x 128 with the same chance, whatever the strings' lengths.  (Round 5 took the state behind
			// one of the window's CHUNKS: a URL is one window that starts with the string, so only the states a URL is in behind
			// its 16th, 32nd, ... byte were ever seen, and the seven states of "http://", 1.4 % of all steps each, never -- in the
			// tier they went without samples, their mass halved with every adapt() until they dropped out, trapped, came back:
			// after eight rounds the 6th, 9th, 13th, 16th and 18th most visited states of a blacklist scanner had no row and
			// 4.1 % of the steps were outside the tier where 0.5 % need be; tools/ranking_quality.py.)  The walk does not keep
			// the states inside a chunk: the drawn lane walks the drawn chunk again from the state in front of it, up to 15 bytes,
			// exactly (the table in memory where the rows say "no row").  What is counted is the state IN FRONT of the drawn byte
			// -- the state whose row that step looks up: the state every string starts in is in front of a byte, never behind one.
			// (drawn per iteration AND wave: if every wave counted its iterations from 0 -- a launch of a URL batch has 32 -- then drawn
			// from the count alone the same 32 pairs of lane and byte are looked at by every wave of every launch; round 5's did, and the
			// state every string starts in, in front of byte 0, happened not to be among them: it sank through the ranking launch
			// by launch although every string looks it up, tools/sampler_probe.py)
			// -- so every wave starts its count somewhere else: ScanRaggedKernel)
			const uint32_t sampleHash = iter * 0x9E3779B1u;
			const bool sampleLaneHere = kDrawn && (threadIdx.x & 63) == (sampleHash >> 26) && !(p.flags & kDebugNoHist);
			const uint32_t lim = WIDE ? p.wide : p.hot;   // (S.hs == lim: the state has no row, S.cold is its id)
			const uint32_t sampleAt = (sampleHash >> 19) & 127u, sampleChunk = sampleAt >> 4;   // wave-uniform
			uint32_t sampleFrom = 0;
#pragma unroll
			for (int k = 0; k < 8; ++k)
				if (uint32_t(k) < full) {
					if (kDrawn && uint32_t(k) == sampleChunk)
						sampleFrom = S.hs != lim ? S.hs : S.cold;
					if constexpr (Act::kActive && WIDE != 0)
						WideChunkAct<Act, WIDE == 3, false>(p, lds, L, W, K, cur[k], 16u, S.hs, S.cold, act, al, S.pos + 16u * k);
					else if constexpr (Act::kActive)
						StepChunkAct(p, lds, L, cur[k], S.hs, S.cold, reinterpret_cast<const uint8_t*>(finHot), act, al,
						             S.pos + 16u * k);
					else if constexpr (WIDE != 0)
						WideChunk<WIDE >= 2, WIDE == 3>(p, lds, W, K, cur[k], S.hs, S.cold, (iter * 8 + k) & 63);
					else
						StepChunk<0>(p, lds, L, cur[k], S.hs, S.cold, (iter * 8 + k) & 63);
				}
			if (kDrawn && full == sampleChunk)
				sampleFrom = S.hs != lim ? S.hs : S.cold;   // (the drawn byte lies in the lane's partial last chunk)
			if (__any(tail != 0) && !(p.flags & kDebugNoPartial)) {
				// all the partial last chunks of the wave in ONE pass: pick each lane's chunk, walk it with a snapshot
				u32x4 v = cur[0];
#pragma unroll
				for (int k = 1; k < 8; ++k)
					if (full == uint32_t(k))
						v = cur[k];
				if constexpr (Act::kActive && WIDE != 0)
					WideChunkAct<Act, WIDE == 3, true>(p, lds, L, W, K, v, tail, S.hs, S.cold, act, al, S.pos + 16u * full);
				else if constexpr (Act::kActive)
					StepPartialAct(p, lds, L, v, tail, S.hs, S.cold, reinterpret_cast<const uint8_t*>(finHot), act, al,
					               S.pos + 16u * full);
				else if constexpr (WIDE != 0)
					WidePartial<WIDE >= 2, WIDE == 3>(p, K, v, tail, S.hs, S.cold);
				else
					StepPartial(p, lds, L, v, tail, S.hs, S.cold, (iter + 32) & 63);
			}
			if constexpr (WIDE == 0 && kDrawn) {
				// The dense rows' sample, drawn like the wide walk's (above): the state in front of one byte of one lane's window,
				// walked to exactly, if it has a dense row; which states are looked up WITHOUT one the re-walks say, fairly too since
				// round 6 (device_common.h TrapChunk), one sample per 1 024 lane-steps of theirs.  (Rounds 2-5: the state a window STARTS
				// in, and of a chunk that left the rows the state it ENDS in if that has no row -- on a URL batch the first is the
				// start state every time and the second misses every state the walk passes through on its way back into the rows:
				// a state with 4.6 % of all lookups stood outside the 255 rows for good, 17 % of the steps there where 2 % need be;
				// tools/ranking_quality.py ... dense.)
				// (every fourth iteration, four times the weight: the dense walk's iteration is short, and a lane walking up to 15
				// steps alone at its end was 6-8 % of it -- 2 140 -> 2 013 GB/s on set_a's URL batch when every iteration drew)
				if (sampleLaneHere && sampleAt < nbl && (iter & 3u) == 0) {
					u32x4 v = cur[0];
#pragma unroll
					for (int k = 1; k < 8; ++k)
						if (sampleChunk == uint32_t(k))
							v = cur[k];
					uint32_t st = sampleFrom;
#pragma unroll 1
					for (uint32_t i = 0; i < (sampleAt & 15u); ++i) {
						st = SlowStep(p, lds, L, st, v.x & 0xFFu);
						v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
						v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
						v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
						v.w >>= 8;
					}
					if (st < p.hot)   // (a state without a row: the re-walks say which, device_common.h TrapChunk)
						atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + st, 4u);
				}
			}
			if constexpr (WIDE != 0) {
				if (sampleLaneHere && sampleAt < nbl) {
					u32x4 v = cur[0];
#pragma unroll
					for (int k = 1; k < 8; ++k)
						if (sampleChunk == uint32_t(k))
							v = cur[k];
					uint32_t sid = sampleFrom;   // device id of the state in front of the chunk
#pragma unroll 1
					for (uint32_t i = 0; i < (sampleAt & 15u); ++i) {
						const uint32_t c2 = HotLookup(v.x & 0xFFu);
						uint32_t next = WideEntry<WIDE == 3>(sid < p.wide ? sid : p.wide, K, c2);
						if (next == p.wide) {
							next = WideNextC2<WIDE >= 2>(p, sid, c2);
							asm volatile("" : "+v"(next));   // (the wait belongs in here)
						}
						sid = next;
						v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
						v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
						v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
						v.w >>= 8;
					}
					// (a state outside the tier is counted as "outside" here; WHICH state it is the re-walks say -- WideTrapChunk, one
					// sample per 1 024 lane-steps outside the tier where this one stands for 8 192 --, and the walks with actions below)
					WideSample<WIDE == 3>(p, lds, W, sid < p.wide ? sid : p.wide);
				}
			}
		}
		if (nb != 0 && !S.loaded) {
			// the last bytes of the whole buffer: exact steps straight from memory
			const uint32_t lim = WIDE ? p.wide : p.hot;   // S.hs == lim: the state has no row, S.cold is its id
			uint32_t st = S.hs != lim ? S.hs : S.cold;
			const uint8_t* q = reinterpret_cast<const uint8_t*>(S.pos);
			for (uint32_t i = 0; i < nb; ++i) {
				if constexpr (WIDE != 0)
					st = WideNext<WIDE >= 2>(p, st, uint32_t(lds[q[i]]) >> 1);
				else
					st = SlowStep(p, lds, L, st, q[i]);
				if constexpr (Act::kActive)
					act.Step(p, lds, L, al, st, S.pos + i + 1);
			}
			S.hs = st < lim ? st : lim;
			S.cold = st;
		}
	}
	PIRE_RCLK(clk, 3);
	if constexpr (Act::kActive) {
		if (ends)
			act.Finish(p, lds, L, reinterpret_cast<const uint8_t*>(finHot), al, S.sIdx, S.hs != (WIDE ? p.wide : p.hot) ? S.hs : S.cold, S.end);
	} else if (__any(ends) && !(p.flags & kDebugNoFinish)) {
		if constexpr (WIDE != 0) {
			// the end-of-string record from memory, under a wave-uniform branch of its own and waited for inside it (FinRecordOf)
			const uint32_t st = S.hs != p.wide ? S.hs : S.cold;
			u32x4 raw = {0, 0, 0, 0};
			if (ends) {
				const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
				raw = *reinterpret_cast<const u32x4*>(&recs[st]);
				asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));
			}
			FinishWith<EXT>(p, lds, L, S.sIdx, ends, raw);
		} else {
			FinishRagged<EXT>(p, lds, L, finHot, S.sIdx, ends, S.hs != p.hot ? S.hs : S.cold);
		}
	}
	PIRE_RCLK(clk, 4);

	// ---- move on
	uint32_t startInit = 0;
	if constexpr (EXT)
		startInit = S.pendInit;   // of the string that starts now (takeNew), before the refill below
	if (got) {
		S.pendPos = textBase + offB;
		S.pendEnd = textBase + offE;
		if constexpr (EXT)
			S.pendInit = initV;
	}
	if (takeNew) {
		uint32_t st;
		if constexpr (Act::kActive)
			st = act.Start(p, lds, L, al, nIdx, nPos);
		else if constexpr (EXT)
			st = p.initIdx ? StartStateFrom(p, startInit) : p.startPerm;
		else
			st = p.startPerm;   // batches with resume states take the EXT instantiation
		const uint32_t lim = WIDE ? p.wide : p.hot;
		S.hs = st < lim ? st : lim;
		S.cold = st;
	}
	S.pos = nPos;
	S.end = nEnd;
	S.sIdx = nIdx;
	S.busy = nBusy;
	S.loaded = nLoad;
	PIRE_RCLK(clk, 5);
	if (__any(nBusy || S.pend))
		return true;
	// The wave's last window: the (dummy) loads into `nxt` are still on their way, and behind the loop the registers are the
	// epilogue's -- the flush of the visit counters starts with a barrier that waits for LDS only, and a line that landed late
	// overwrote the zero high word of its counter index: a memory fault once in a few hundred launches of a kernel whose
	// strings all die at their first byte, i.e. whose last iteration has nothing to walk (found by tools/stress_dict.py, round
	// 6; every instantiation of this kernel since round 2 had the window).  Waited for HERE, on the way out -- and without
	// naming the registers: named, the instantiations at the register limit carried them to this point through scratch,
	// stored while the loads were in flight (the build's audit refused three of them).
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	return false;
}

template <class Act, bool EXT, int WIDE = 0>
__global__ __launch_bounds__(1024) void ScanRaggedKernel(ScanParams p, unsigned long long* workCounter,
                                                         RaggedGrab grab, Act act)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	RaggedClock clk;
#ifdef PIRE_HIP_TUNING
	clk.on = p.stamps != nullptr;
	for (int k = 0; k < 8; ++k)
		clk.acc[k] = 0;
	clk.t = clk.on ? __builtin_readcyclecounter() : 0;
#endif
	const WideLayout W = WIDE ? MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, WIDE == 3 ? p.zipFull : 0) : WideLayout();
	const WideConst K = WIDE ? MakeWideConst(p, W) : WideConst();
	LdsLayout L = {};
	if constexpr (WIDE != 0)
		L.countsOff = W.countsOff;   // what FinishWith looks at
	else
		L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	FinRec* finHot = reinterpret_cast<FinRec*>(lds + L.total);
	volatile RaggedWork* work = reinterpret_cast<volatile RaggedWork*>(lds + (WIDE ? W.total : L.total + kRaggedFinBytes));
	if constexpr (WIDE != 0) {
		if (threadIdx.x == 0) {
			work->range = 0;
			work->lock = 0;
			work->exhausted = 0;
		}
		LoadWideToLds(p, lds, W);   // ends with a barrier
	} else {
		if constexpr (!Act::kActive) {
			const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
			for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
				finHot[i] = recs[i];
		} else {
			act.LoadLds(p, reinterpret_cast<uint8_t*>(finHot));   // the actions' own LDS data take that place
		}
		if (threadIdx.x == 0) {
			work->range = 0;
			work->lock = 0;
			work->exhausted = 0;
		}
		LoadTableToLds(p, lds, L);   // ends with a barrier
	}

	const uint64_t textBase = reinterpret_cast<uint64_t>(p.text);
	const uint64_t safeEnd = (textBase + ((EXT && p.ends) ? p.textEnd : p.offsets[p.n]) + 15) & ~uint64_t(15);

	RaggedRange R = {0, 0, false};
	RaggedLane S;
	S.pos = S.end = textBase;
	S.sIdx = 0;
	S.hs = S.cold = 0;
	S.busy = S.loaded = S.pend = false;
	S.sIdxN = 0;
	S.pendInit = 0;
	S.pendPos = S.pendEnd = textBase;
	typename Act::Lane al = {};
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);

	if (AssignPending(p, work, workCounter, grab, R, S)) {
		S.pendPos = textBase + p.offsets[S.sIdxN];
		S.pendEnd = textBase + ((EXT && p.ends) ? p.ends[S.sIdxN] : p.offsets[S.sIdxN + 1]);
		if (EXT && p.initIdx)
			S.pendInit = p.initIdx[S.sIdxN];
	}
	PIRE_RCLK(clk, 6);
	// (the iteration count seeds the visit samples -- which lane, which byte -- and every wave starts it somewhere else: RaggedPhase)
	const uint32_t iter0 = uint32_t(__builtin_amdgcn_readfirstlane(int((blockIdx.x * 16u + (threadIdx.x >> 6)) * 0x632BE5ABu)));
	uint32_t iter = iter0;
	for (;; iter += 2) {
		if (!RaggedPhase<Act, EXT, WIDE>(p, lds, L, finHot, work, workCounter, grab, textBase, safeEnd, R, S, a, b, iter, act, al, clk, W, K))
			break;
		if (!RaggedPhase<Act, EXT, WIDE>(p, lds, L, finHot, work, workCounter, grab, textBase, safeEnd, R, S, b, a, iter + 1, act, al, clk, W, K))
			break;
	}
	// (once more behind the loop, for the build's audit: it follows every way out of the window loop until all loads are waited
	// for -- tools/audit/inflight_registers.py check_exits -- and cannot know that the two ways out of RaggedPhase, merged by the
	// compiler behind a flag, each passed their wait; this one costs nothing, the loads have landed)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PIRE_HIP_TUNING
	if (clk.on && (threadIdx.x & 63) == 0) {
		clk.acc[7] = iter - iter0;
		for (int k = 0; k < 8; ++k)
			atomicAdd(&p.stamps[k], clk.acc[k]);
		atomicAdd(&p.stamps[8], 1ull);
	}
#endif
	if constexpr (WIDE != 0)
		FlushWide(p, lds, W);
	else
		FlushCounts(p, lds, L);
	// the last block out leaves the launch's slot {next string, blocks done} zeroed for whoever uses it next
	// (internal.h WorkSlotOf): every block is past its last grab when it counts itself done
	if (threadIdx.x == 0) {
		__threadfence();
		if (atomicAdd(workCounter + 1, 1ull) == gridDim.x - 1) {
			workCounter[0] = 0;
			workCounter[1] = 0;
		}
	}
}


// ------------------------------------------------------------------------------------------ launcher

bool RaggedEligible(const ScanParams& p, uint64_t totalBytesHint)
{
	// one string per lane with dynamic re-assignment: worth it from a few waves' worth of strings
	return p.offsets != nullptr && p.n >= 256 && p.n < (1ull << 32) - (1ull << 16) && totalBytesHint >= 4096;
}

namespace {

template <class Act, bool EXT, int WIDE = 0>
int LaunchRaggedT(const ScanParams& p, unsigned long long* workCounter, const Act& act, hipStream_t stream)
{
	// (the counter is zero: the last block of the launch that used the slot before put it back, internal.h WorkSlotOf)
	hipError_t e;
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	const uint32_t ldsBytes = WIDE ? MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, WIDE == 3 ? p.zipFull : 0).total + uint32_t(sizeof(RaggedWork))
	                               : L.total + kRaggedLdsExtra;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	e = SetDynamicLds(reinterpret_cast<const void*>(ScanRaggedKernel<Act, EXT, WIDE>), uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	// one string per lane: spread the waves over every CU before stacking them (4..16 waves per block, 1 block per CU)
	const uint64_t waves = (p.n + 63) / 64;
	const uint64_t wavesPerBlock = std::min<uint64_t>(16, std::max<uint64_t>(4, (waves + cus - 1) / cus));
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(cus), (waves + wavesPerBlock - 1) / wavesPerBlock));
	// strings a block takes from the global counter at a time: ~8 grabs per block keep the tail balanced; batches
	// that barely fill the lanes are simply split evenly
	const uint64_t perBlock = (p.n + blocks - 1) / blocks, lanes = wavesPerBlock * 64;
	uint64_t grabDiv = 8, grabCap = 16384;
#ifdef PIRE_HIP_TUNING
	if (const char* e1 = getenv("PIRE_HIP_RAGGED_GRABDIV"))
		grabDiv = std::max(1, atoi(e1));
	if (const char* e2 = getenv("PIRE_HIP_RAGGED_GRABCAP"))
		grabCap = std::max(64, atoi(e2));
#endif
	uint64_t blockGrab = std::min<uint64_t>(grabCap, std::max<uint64_t>(perBlock / grabDiv, std::min(perBlock, lanes)));
	blockGrab = (blockGrab + 63) / 64 * 64;
	// a wave takes several windows' worth of strings per visit of the block lock when there is plenty (otherwise the
	// 16 waves of a block queue up behind the lock every iteration), one lane-fill when a block grab barely feeds
	// its waves
	RaggedGrab grab;
	grab.block = uint32_t(blockGrab);
	grab.wave = uint32_t(std::min<uint64_t>(256, std::max<uint64_t>(64, blockGrab / (2 * wavesPerBlock) / 64 * 64)));
	ScanParams q = p;
#ifdef PIRE_HIP_TUNING
	if (const char* dbg = getenv("PIRE_HIP_DEBUG_RAGGED")) {   // timing experiments: 1 no partial passes, 2 no finish, 4 no traps
		const int m = atoi(dbg);
		q.flags |= (m & 1 ? kDebugNoPartial : 0) | (m & 2 ? kDebugNoFinish : 0) | (m & 4 ? kDebugNoTrap : 0) |
		           (m & 8 ? kDebugNoStep : 0) | (m & 16 ? kDebugNoRefill : 0);
	}
	static unsigned long long* clockBuf = nullptr;
	const bool clocks = getenv("PIRE_HIP_DEBUG_RAGGED_CLOCKS") != nullptr;
	q.stamps = nullptr;
	if (clocks) {
		if (!clockBuf)
			(void)hipMalloc(reinterpret_cast<void**>(&clockBuf), 16 * 8);
		(void)hipMemset(clockBuf, 0, 16 * 8);
		q.stamps = clockBuf;
	}
#endif
	hipLaunchKernelGGL((ScanRaggedKernel<Act, EXT, WIDE>), dim3(unsigned(blocks)), dim3(unsigned(wavesPerBlock * 64)), ldsBytes, stream,
	                   q, workCounter, grab, act);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "ragged kernel launch");
#ifdef PIRE_HIP_TUNING
	if (clocks) {
		(void)hipDeviceSynchronize();
		unsigned long long c[16];
		(void)hipMemcpy(c, clockBuf, sizeof c, hipMemcpyDeviceToHost);
		const double waves = double(c[8] ? c[8] : 1), iters = double(c[7] ? c[7] : 1);
		fprintf(stderr, "pire_hip ragged clocks: %llu waves, %.1f iterations per wave; shader clocks per wave and iteration: wait %.0f | "
		        "transpose+issue %.0f | assign+offsets %.0f | walk %.0f | end of string %.0f | move on %.0f ; before the first iteration %.0f "
		        "per wave\n", c[8], iters / waves, c[0] / iters, c[1] / iters, c[2] / iters, c[3] / iters, c[4] / iters, c[5] / iters,
		        c[6] / waves);
	}
#endif
	return PIRE_HIP_OK;
}

}  // namespace

int LaunchRagged(const ScanParams& p, unsigned long long* workCounter, hipStream_t stream)
{
	if (int rc = CheckCounts(p))
		return rc;
	if (p.ends || p.initIdx || (p.flags & kPermIds))
		return LaunchRaggedT<NoAct, true>(p, workCounter, NoAct(), stream);
	return LaunchRaggedT<NoAct, false>(p, workCounter, NoAct(), stream);
}

// Offset batches of a table whose scans keep leaving the dense rows (WideWanted, wide.hip): the same kernel on the
// class-indexed walk.
int LaunchRaggedWide(const ScanParams& p, unsigned long long* workCounter, hipStream_t stream)
{
	if (int rc = CheckCounts(p))
		return rc;
	const bool ext = p.ends || p.initIdx || (p.flags & kPermIds);
	if (p.zipFull) {   // (a zipped image implies the u16 table: table.cpp ChooseZip)
		NoteKernel("ragged_wide", "pirehip::ScanRaggedKernel<NoAct, wide walk, u16 table, zipped rows>");
		return ext ? LaunchRaggedT<NoAct, true, 3>(p, workCounter, NoAct(), stream) : LaunchRaggedT<NoAct, false, 3>(p, workCounter, NoAct(), stream);
	}
	if (p.next16) {
		NoteKernel("ragged_wide", "pirehip::ScanRaggedKernel<NoAct, wide walk, u16 table>");
		return ext ? LaunchRaggedT<NoAct, true, 2>(p, workCounter, NoAct(), stream) : LaunchRaggedT<NoAct, false, 2>(p, workCounter, NoAct(), stream);
	}
	NoteKernel("ragged_wide", "pirehip::ScanRaggedKernel<NoAct, wide walk, u32 table>");
	return ext ? LaunchRaggedT<NoAct, true, 1>(p, workCounter, NoAct(), stream) : LaunchRaggedT<NoAct, false, 1>(p, workCounter, NoAct(), stream);
}

// The walks with actions take the ragged kernel from a few waves' worth of strings (below that, and for tables
// whose counters do not pack, the one-string-per-lane kernels of exact.hip).
bool RaggedActEligible(const ScanParams& p)
{
	return p.offsets != nullptr && p.n >= 256 && p.n < (1ull << 32) - (1ull << 16) && !GetConfig().no_ragged_act;
}

// The walks with actions on the class-indexed walk (WideChunkAct above): for tables WideWanted() sends there, with the u16 table
// behind the rows.  To the actions every state is then one without a dense row (p.hot = 0: flags, increments, records from memory).
static bool WideActWanted(const ScanParams& p)
{
	return p.next16 && p.wide && p.wideRows && !p.ends && !p.initIdx && !(p.flags & kPermIds) && WideWanted(p, GetConfig());
}
static void ForWideAct(ScanParams* p)
{
	p->hot = 0;
	p->hotFinalLo = 0;
	p->hotDeadLo = 0;
	p->compact = 0;
	p->actDist = nullptr;
}

int LaunchRaggedHalfFinal(const ScanParams& p0, unsigned long long* workCounter, uint32_t* outResults, hipStream_t stream)
{
	ScanParams p = p0;    // with the compact rows: trapped chunks that cannot reach a Final state use them (actDist)
	if (!p.actDist)
		p.compact = 0;
	p.outCounts = nullptr;
	if (!p.incPerm) {     // counters that do not pack: rows of the result array
		HalfFinalWideAct wide;
		wide.results = outResults;
		return LaunchRaggedT<decltype(wide), false>(p, workCounter, wide, stream);
	}
	HalfFinalAct act;
	act.results = outResults;
	if (WideActWanted(p)) {   // a table whose scans keep leaving the dense rows: the same walk through the wide rows
		ForWideAct(&p);
		NoteKernel("ragged_half_final_wide", p.zipFull ? "pirehip::ScanRaggedKernel<HalfFinalAct, wide walk, zipped rows>" : "pirehip::ScanRaggedKernel<HalfFinalAct, wide walk>");
		return p.zipFull ? LaunchRaggedT<decltype(act), false, 3>(p, workCounter, act, stream) : LaunchRaggedT<decltype(act), false, 2>(p, workCounter, act, stream);
	}
	return LaunchRaggedT<decltype(act), false>(p, workCounter, act, stream);
}

int LaunchRaggedCapture(const ScanParams& p0, unsigned long long* workCounter, const uint32_t* info, long long* outBegin,
                        long long* outEnd, hipStream_t stream)
{
	ScanParams p = p0;
	if (!p.actDist)
		p.compact = 0;
	p.outCounts = nullptr;
	CaptureAct act;
	act.info = info;
	act.outBegin = outBegin;
	act.outEnd = outEnd;
	act.beginStep = (p0.flags & PIRE_HIP_RUN_BEGIN) ? 1u : 0u;
	NoteKernel("ragged_capture");
	return LaunchRaggedT<CaptureAct, false>(p, workCounter, act, stream);
}

int LaunchRaggedPrefix(const ScanParams& p0, unsigned long long* workCounter, bool longest, bool throughEnd,
                       long long* outLen, hipStream_t stream)
{
	ScanParams p = p0;
	if (!p.actDist)
		p.compact = 0;
	p.outCounts = nullptr;
	PrefixAct act;
	act.outLen = outLen;
	act.longest = longest ? 1 : 0;
	act.throughEnd = throughEnd ? 1 : 0;
	act.startFlags = p.owner ? p.owner->host.flags[p.hostOrigOfPerm[p.startPerm]] : 0;
	if (WideActWanted(p)) {
		ForWideAct(&p);
		NoteKernel("ragged_prefix_wide", p.zipFull ? "pirehip::ScanRaggedKernel<PrefixAct, wide walk, zipped rows>" : "pirehip::ScanRaggedKernel<PrefixAct, wide walk>");
		return p.zipFull ? LaunchRaggedT<decltype(act), false, 3>(p, workCounter, act, stream) : LaunchRaggedT<decltype(act), false, 2>(p, workCounter, act, stream);
	}
	return LaunchRaggedT<decltype(act), false>(p, workCounter, act, stream);
}


}  // namespace pirehip
