// The stream kernel (offset batches of many short strings, pire_hip_run).  DESIGN.md section 4.4.
//
// What the reference's callers pass is ragged lines -- one Runner(sc).Begin().Run(p, n).End() per line
// (/root/reference/samples/pigrep/pigrep.cpp:38-45, tools/bench/bench.cpp:241-254, pire/run.h:271-275).  The ragged
// kernel (ragged.hip) gives every lane ONE string at a time and a window of up to 128 bytes of it per iteration: on
// URL-sized strings a window is 58 % full, a line of the text is fetched by two windows iterations apart (L2 fetches
// 1.87 x the text, profiles/r02_ragged_pmc_urls_group_loads.txt), and every iteration pays assignment, offsets and
// end-of-string work for the whole wave (0.27-0.32 of HBM).
//
// Here the unit of work is not a string but a RUN OF CONSECUTIVE STRINGS.  The strings of an offset batch lie back to
// back in memory (string i = [offsets[i], offsets[i+1])), so the bytes behind the end of string i ARE string i+1: a
// lane that owns strings s0 .. s1-1 walks one contiguous piece of text in whole, 128-byte-aligned lines -- every line
// exactly once, every window full -- and what marks a string is a BOUNDARY inside a 16-byte chunk of the walk: at the
// step where the current string ends the lane keeps the state it is in (the string's end state) and continues from the
// start state (StepChunkB: one compare and two selects per byte on top of the lookup; no loop, no branch).
//
//   * cost key: key(i) = (offsets[i] - offsets[0]) + lambda * i -- a byte costs 1, a string boundary `lambda` -- is
//     strictly increasing, so "the first string at or behind key T" is well defined even among empty strings.  The
//     batch is cut into one TASK per wave at equal steps of the key (two cooperative 64-ary searches per wave over the
//     offsets: 4 rounds for 4 M strings), a task into SUB-TASKS of at most 1 280 strings (kStreamMaxStrings), and a sub-task among the 64
//     lanes at equal steps of the key again.  Equal work per wave and per lane by construction, whatever the lengths:
//     no work counters, no atomics, no tail of late long strings.
//   * a sub-task's string positions sit in LDS as 32-bit offsets from the line that holds its first byte (4 KiB per
//     wave: the compact tier's place, which this kernel does not use).  A lane finds its first string by a binary
//     search there, reads the end of a string when it starts it, and leaves a string's end state in the slot of its
//     end position (dead by then); when the sub-task is over the wave turns the slots into StateIndex / Final /
//     counters with coalesced stores (FinishRagged, 64 strings at a time).
//   * windows: the lane's current line in registers, the next one in flight (group loads + DPP transpose, the ragged
//     and tiled kernels' load path).  Nothing outside the 128-byte lines that hold the sub-task's text is read: a line
//     lies inside one page, so whatever holds the text holds the line.
//   * exactness: a chunk in which a lane leaves the dense rows, or in which a lane meets more than one boundary (strings
//     shorter than 16 bytes, empty strings), is walked again for that lane byte by byte with the exact step
//     (ExactRest).  Results never depend on which rows are dense, on lambda or on how the batch was cut.

#include "stream_common.h"
#include "wide_common.h"

namespace pirehip {

struct StreamLane {
	uint32_t wpos;      // start of the current window, relative to the sub-task's line base (multiple of 128)
	uint32_t E;         // end of the current string (relative); kStreamInf once the lane's strings are over
	uint32_t En;        // end of the string that starts at the next boundary, read one boundary ahead (LDS latency off the walk)
	uint32_t laneEnd;   // end of the lane's last string: windows up to the one that holds it are walked
	uint32_t dataEnd;   // the same, or 0 when all the lane's strings are empty: lines that hold none of its bytes are not fetched
	uint32_t nxt;       // the string that starts at the next boundary (index inside the sub-task)
	uint32_t sEnd;      // one past the lane's last string
	uint32_t hs, cold;  // walk state: dense-row id (p.hot = outside the dense rows, then `cold` is the state)
	bool live;          // the bytes being walked belong to a string of this lane (not the bytes in front of its first one)
};

// A string ends here in `state`: leave the state in the slot of the string's END position (read when the string
// started, dead since), start the next string of the lane if there is one.
__device__ __forceinline__ void StreamBoundary(LdsWordPtr eo, StreamLane& S, uint32_t state)
{
	if (S.live)
		eo[S.nxt] = state;
	if (S.nxt < S.sEnd) {
		S.E = S.En;
		S.nxt += 1;
		S.live = true;
		S.En = eo[S.nxt + 1];   // of the string after this one (a word of the padding / a neighbour's when there is none: unused)
	} else {
		S.E = kStreamInf;
		S.live = false;
	}
}

// Sixteen bytes through the dense rows with a string boundary in front of byte c (c >= 16: none): `snap` = the state the
// walk was in when it reached byte c, and the walk goes on from `start`.
// What a step costs on top of the plain kernel's v_perm + ds_read_u8 is the select that restarts the walk (on the
// dependent chain) and the select that keeps the end state (issued behind the lookup, in its shadow).  The sixteen lane
// masks "c == j" are NOT sixteen vector compares: five ballots (c < 16 and the four bits of c) and the scalar unit's
// and / andn2 give all of them, and the scalar unit has nothing else to do here (the first version spent a quarter of
// its vector instructions on those compares; 60 % VALU-busy, the walk's steps 150 cycles apart against 90 in the tiled
// kernel: profiles/r04_stream_pmc_*).
// START0 (the start state has dense id 0: table.cpp puts Begin()'s there when it can): the restart is not a select of the
// state in front of the lookup but a select of the v_perm SELECTOR -- the row byte of the address comes from the state
// register (0x04) or is the constant 0 (0x0c) -- which depends on nothing the chain computes: the dependent chain of a
// boundary chunk is the plain kernel's v_perm -> ds_read_u8 again (the first version's select in the chain cost every
// step ~50 cycles: 174 per step against the tiled kernel's 96, profiles/r04_stream_ablation.log).
template <bool START0>
__device__ __forceinline__ void StepChunkB(const u32x4 v, uint32_t c, uint32_t start, uint32_t& hs, uint32_t& snap)
{
	const unsigned long long any = __ballot(c < 16u), b0 = __ballot((c & 1u) != 0), b1 = __ballot((c & 2u) != 0),
	                         b2 = __ballot((c & 4u) != 0), b3 = __ballot((c & 8u) != 0);
	uint32_t h = hs, sn = hs;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int j = 4 * w + b;
			const unsigned long long m = any & ((j & 1) ? b0 : ~b0) & ((j & 2) ? b1 : ~b1) & ((j & 4) ? b2 : ~b2) & ((j & 8) ? b3 : ~b3);
			const bool at = __builtin_amdgcn_inverse_ballot_w64(m);
			uint32_t next;
			if constexpr (START0) {
				const uint32_t sel = at ? 0x0c0c0c00u + uint32_t(b) : 0x0c0c0400u + uint32_t(b);
				next = HotLookup(__builtin_amdgcn_perm(h, x, sel));
			} else {
				const uint32_t from = at ? start : h;
				next = HotLookup(__builtin_amdgcn_perm(from, x, 0x0c0c0400u + uint32_t(b)));
			}
			sn = at ? h : sn;   // behind the lookup: it needs the state in front of the step, not the lookup's result
			h = next;
		}
	}
	hs = h;
	snap = sn;
}

// ---- the same on the class-indexed walk (WIDE: 2 = rows of u16 entries in LDS, 3 = the zipped image; wide_common.h) -----------
// The state is a device id with a place in the tier or `p.wide` (the escape state: sticky, S.cold holds the id).  The classes
// of the sixteen bytes do not depend on the walk; the restart at a boundary is a select in front of the row's address.
template <bool ZIP>
__device__ __forceinline__ void WideChunkPlain(const WideConst& K, const u32x4 v, uint32_t& st)
{
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		const uint32_t c0 = HotLookup(x & 0xFFu);
		const uint32_t c1 = HotLookup((x >> 8) & 0xFFu);
		const uint32_t c2 = HotLookup((x >> 16) & 0xFFu);
		const uint32_t c3 = HotLookup(x >> 24);
		st = WideEntry<ZIP>(st, K, c0);
		st = WideEntry<ZIP>(st, K, c1);
		st = WideEntry<ZIP>(st, K, c2);
		st = WideEntry<ZIP>(st, K, c3);
	}
}

template <bool ZIP>
__device__ __forceinline__ void WideChunkB(const WideConst& K, const u32x4 v, uint32_t c, uint32_t start, uint32_t& st, uint32_t& snap)
{
	const unsigned long long any = __ballot(c < 16u), b0 = __ballot((c & 1u) != 0), b1 = __ballot((c & 2u) != 0),
	                         b2 = __ballot((c & 4u) != 0), b3 = __ballot((c & 8u) != 0);
	uint32_t h = st, sn = st;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		const uint32_t cl[4] = {HotLookup(x & 0xFFu), HotLookup((x >> 8) & 0xFFu), HotLookup((x >> 16) & 0xFFu), HotLookup(x >> 24)};
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int j = 4 * w + b;
			const unsigned long long m = any & ((j & 1) ? b0 : ~b0) & ((j & 2) ? b1 : ~b1) & ((j & 4) ? b2 : ~b2) & ((j & 8) ? b3 : ~b3);
			const bool at = __builtin_amdgcn_inverse_ballot_w64(m);
			const uint32_t next = WideEntry<ZIP>(at ? start : h, K, cl[b]);
			sn = at ? h : sn;
			h = next;
		}
	}
	st = h;
	snap = sn;
}

// Bytes from .. 15 of chunk k of the window, exactly, for one lane: boundaries as they come (any number, empty strings
// included), the exact step for the bytes of live strings.  `st` = the state in front of byte `from`.  Rolled: the cold path.
template <int WIDE>
__device__ __forceinline__ void ExactRest(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const WideLayout& W, const WideConst& K, LdsWordPtr eo,
                                          const u32x4& v, uint32_t k, uint32_t from, uint32_t st, StreamLane& S, uint32_t sampleLane)
{
	const uint32_t lim = WIDE ? p.wide : p.hot;
#pragma unroll 1
	for (uint32_t i = from; i < 16; ++i) {
		while (S.E - S.wpos == 16u * k + i) {
			StreamBoundary(eo, S, st);
			st = p.startPerm;
		}
		if (S.live) {
			const uint32_t word = i < 8 ? (i < 4 ? v.x : v.y) : (i < 12 ? v.z : v.w);
			const uint32_t byte = (word >> (8u * (i & 3u))) & 0xFFu;
			if constexpr (WIDE != 0) {
				// the row's entry in LDS; the table in memory only where that says "no row" (the state has none, or the target)
				const uint32_t c2 = HotLookup(byte);
				uint32_t next = WideEntry<WIDE == 3>(st < lim ? st : lim, K, c2);
				if (next == lim) {
					next = WideNextC2<true>(p, st, c2);
					asm volatile("" : "+v"(next));   // (the wait belongs in here)
				}
				st = next;
			} else {
				st = SlowStep(p, lds, L, st, byte);
			}
		}
	}
	S.hs = st < lim ? st : lim;
	S.cold = st;
	// tell pire_hip_table_adapt() which rows deserve LDS, sampled like TrapChunk's
	if (S.live && st >= lim && (threadIdx.x & 63) == sampleLane) {
		atomicAdd(&p.visitCold[st], 1u);
		if constexpr (WIDE != 0)
			atomicAdd(reinterpret_cast<uint32_t*>(lds + W.progOff) + 1, 1u);
		else
			atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + kLdsTrapSlot, 1u);
	}
}

// One window: start fetching the next line into `nxt`, walk the line held in `cur`.  Returns whether any lane of the wave
// has a further line.
template <bool START0, int WIDE>
__device__ __forceinline__ bool StreamPhase(const ScanParams& p, uint8_t* lds, const LdsLayout& L, const WideLayout& W, const WideConst& K,
                                            LdsWordPtr eo, uint64_t lineBase, StreamLane& S, u32x4 (&cur)[8], u32x4 (&nxt)[8],
                                            uint32_t iter, bool walk)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t lim = WIDE ? p.wide : p.hot;   // S.hs == lim: the state has no row, S.cold is its id
	const bool more = S.laneEnd > S.wpos + 128u;   // boundaries of this lane lie behind this window
	// The NEXT line is requested before this one is waited for (its address depends on nothing but the window counter),
	// as the tiled kernel does: its latency then runs behind the wait, the transpose and the walk of this line.  With the
	// wait first a wave had nothing on its way during the transpose, and the kernel without any walk still took 122 us on
	// the URL batch (3.75 TB/s, profiles/r04_stream_ablation.log).
	// unconditional (lanes without a further line fetch a harmless valid one), see ragged.hip
	// (the sum modulo 2^32 FIRST: in a sub-task's first phase wpos is -128 mod 2^32 for the lanes whose first line is line 0)
	IssueTileGroup(nxt, S.dataEnd > S.wpos + 128u ? lineBase + uint64_t(uint32_t(S.wpos + 128u)) : reinterpret_cast<uint64_t>(p.hotRows), lane);
	// at most the 8 loads just issued may still be out: loads return in order, whatever else is in the queue only makes
	// the wait stricter (tiled.hip)
	asm volatile("s_waitcnt vmcnt(8)"
	             : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
	TransposeTile(cur, lane);
	if (lane == (iter & 63) && S.live) {   // visit sample, as in the tiled kernel
		if constexpr (WIDE != 0)
			WideSample<WIDE == 3>(p, lds, W, S.hs);
		else
			atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + S.hs, 1u);
	}
	// (`walk` is false in a sub-task's first phase only: its window is the line in FRONT of the lanes' first lines, there
	// to get the first lines requested from inside the loop -- asynchronous asm loads issued in front of the loop end up
	// in registers the loop does not use, and the compiler copies them over while they are in flight)
#if defined(PIRE_EXP) && PIRE_EXP == 3   // timing experiment: no walk at all (loads, transposes, bookkeeping only)
	if (false)
#else
	if (walk)
#endif
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint32_t c = (S.E - S.wpos) - 16u * uint32_t(k);   // bytes of this chunk in front of the boundary (>= 16: none)
		const uint32_t hs0 = S.hs;
#if defined(PIRE_EXP) && PIRE_EXP == 5   // timing experiment: every chunk takes the plain step (boundaries ignored)
		if (true) {
#else
		if (!__any(c < 16u)) {
#endif
			// no string of the wave ends in this chunk: the tiled kernel's step
			if constexpr (WIDE != 0) {
				uint32_t h = S.hs;
				WideChunkPlain<WIDE == 3>(K, cur[k], h);
				S.hs = h;
				if (h == lim && S.live)   // (lanes between their strings walk bytes that are not theirs: wherever that leads)
					WideTrapChunk<true, WIDE == 3>(p, lds, W, K, cur[k], hs0, S.hs, S.cold, (iter * 8 + k) & 15u);
			} else {
				uint32_t h = S.hs;
#pragma unroll
				for (int w = 0; w < 4; ++w) {
					const uint32_t x = cur[k][w];
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0400u));
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0401u));
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0402u));
					h = HotLookup(__builtin_amdgcn_perm(h, x, 0x0c0c0403u));
				}
				S.hs = h;
				if (h == p.hot && S.live) {
					// TrapChunk without the compact tier (its LDS holds the string positions here): the chunk again through the
					// full table, the cold end state sampled for pire_hip_table_adapt()
					const uint32_t f = SlowChunk(p, lds, L, cur[k], hs0 != p.hot ? hs0 : S.cold);
					S.hs = f < p.hot ? f : p.hot;
					S.cold = f;
					if (f >= p.hot && lane == ((iter * 8 + k) & 63)) {
						atomicAdd(&p.visitCold[f], 1u);
						atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + kLdsTrapSlot, 1u);
					}
				}
			}
		} else {
			uint32_t snap;
			if constexpr (WIDE != 0)
				WideChunkB<WIDE == 3>(K, cur[k], c, p.startPerm, S.hs, snap);
			else
				StepChunkB<START0>(cur[k], c, p.startPerm, S.hs, snap);
			const bool isB = c < 16u;
			// the part of the chunk that belongs to the current string left the rows (or was outside them all along),
			// or the part that belongs to the string starting here did: this lane's chunk again, exactly
			const bool trapBefore = S.live && (isB ? snap == lim : S.hs == lim);
			const bool trapAfter = isB && S.hs == lim && S.nxt < S.sEnd;
			bool exact = trapBefore || trapAfter;
			uint32_t from = 0, st = hs0 != lim ? hs0 : S.cold;
#if defined(PIRE_EXP) && PIRE_EXP == 4   // timing experiment: boundary chunks walked, boundaries not processed
			if (false) {
#else
			if (!exact && isB) {
#endif
				StreamBoundary(eo, S, snap);
				if ((S.E - S.wpos) - 16u * uint32_t(k) < 16u) {   // the string that started here ends in this chunk as well
					exact = true;
					from = c;
					st = p.startPerm;
				}
			}
			if (exact)
				ExactRest<WIDE>(p, lds, L, W, K, eo, cur[k], uint32_t(k), from, st, S, (iter * 8 + k) & 63);
		}
	}
	// strings that end with the line: their boundary is here, not in a window of its own (which may not exist)
	while (S.E - S.wpos == 128u) {
		StreamBoundary(eo, S, S.hs != lim ? S.hs : S.cold);
		S.hs = p.startPerm;
		S.cold = p.startPerm;
	}
	S.wpos += 128u;
	if (__any(more))
		return true;
	// the sub-task's last window: the (dummy) loads into `nxt` are waited for HERE, where the compiler has the registers
	// at hand -- named behind the loop they were carried there through scratch, i.e. given to other values while the
	// loads were still on their way
	WaitAllLoads(nxt);
	return false;
}

#ifdef PIRE_HIP_TUNING
// timing experiments (PIRE_HIP_DEBUG_STREAM_CLOCKS): wall-clock (100 MHz) time of a wave per stage, summed over the waves
// into ScanParams::stamps: 0 search, 1 table copy, 2 positions into LDS + lane search, 3 window loop, 4 flush; 5 = waves,
// 6 = windows, 7 = the latest end of any wave, 8 = the earliest start (both against the kernel's first stamp)
#define PIRE_SCLK(k)                                                       \
	do {                                                                   \
		if (p.stamps) {                                                    \
			const unsigned long long n_ = wall_clock64();                  \
			sclk[k] += n_ - sclkT;                                         \
			sclkT = n_;                                                    \
		}                                                                  \
	} while (0)
#else
#define PIRE_SCLK(k) do { } while (0)
#endif

// WIDE (round 6): 0 = the dense rows; 2 / 3 = the class-indexed walk on the stream image of the table (internal.h
// StreamWideTier: plain rows / zipped), sub-tasks of kStreamWideStrings strings, end-of-string records from memory.
template <bool START0, int WIDE>
__global__ __launch_bounds__(1024) void ScanStreamKernel(ScanParams p, StreamGeom g)
{
#ifdef PIRE_HIP_TUNING
	unsigned long long sclk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sclkT = p.stamps ? wall_clock64() : 0;
	const unsigned long long sclkStart = sclkT;
#endif
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const WideLayout W = WIDE ? MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, WIDE == 3 ? p.zipFull : 0) : WideLayout();
	const WideConst K = WIDE ? MakeWideConst(p, W) : WideConst();
	LdsLayout L = {};
	if constexpr (WIDE != 0)
		L.countsOff = W.countsOff;   // what FinishWith looks at
	else
		L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, 0);
	FinRec* finHot = reinterpret_cast<FinRec*>(lds + L.total);
	const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
	LdsWordPtr eo = reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(
		WIDE ? W.total + wave * (g.maxStrings + 16u) * 4u : L.total + kRaggedFinBytes + wave * kStreamStageWords * 4u));
	const uint32_t kMaxStrings = WIDE ? g.maxStrings : kStreamMaxStrings;   // (wide walk: what the image leaves room for)

	// ---- this wave's task: strings [i0, i1), found before the table is copied (the searches' round trips overlap the
	// other waves' part of the copy)
	uint64_t i0, i1;
	const bool hasTask = StreamTaskOfWave(p.offsets, p.n, g, i0, i1);
	(void)hasTask;
	PIRE_SCLK(0);
	const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
	if constexpr (WIDE != 0) {
		LoadWideToLds(p, lds, W);   // ends with a barrier
	} else {
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			finHot[i] = recs[i];
		LoadTableToLds(p, lds, L);   // ends with a barrier
	}
	PIRE_SCLK(1);

	const uint64_t textBase = reinterpret_cast<uint64_t>(p.text);
	uint32_t iter = 0;
	const uint32_t subStrings = StreamSubStrings(i1 - i0, kMaxStrings);
	for (uint64_t sub = i0; sub < i1; sub += subStrings) {
		// (what the set-up derives from the lane number -- a dozen addresses and positions -- is cheap to compute and was
		// hoisted out of this loop and carried across the window loop through scratch: opaque here, so it stays inside)
		uint32_t lane = threadIdx.x & 63;
		asm volatile("" : "+v"(lane));
		const uint32_t m = uint32_t(__builtin_amdgcn_readfirstlane(int(i1 - sub < subStrings ? i1 - sub : subStrings)));
		uint64_t lineBase;
		uint32_t lead, spanBytes;
		const bool staged = StreamStage(p.offsets, sub, m, textBase, eo, lane, lineBase, lead, spanBytes);
		PIRE_SCLK(5);   // (tuning) positions into LDS
		if (!staged) {
			// positions that do not fit 32 bits (a string of 4 GiB among short ones): every lane takes whole strings and
			// walks them byte by byte from memory
			for (uint32_t base = 0; base < m; base += 64) {
				const uint32_t q = base + lane;
				uint32_t st = p.startPerm;
				if (q < m)
					for (uint64_t at = p.offsets[sub + q], end = p.offsets[sub + q + 1]; at < end; ++at) {
						if constexpr (WIDE != 0)
							st = WideNext<true>(p, st, uint32_t(lds[p.text[at]]) >> 1);
						else
							st = SlowStep(p, lds, L, st, p.text[at]);
					}
				if constexpr (WIDE != 0) {
					u32x4 raw = {0, 0, 0, 0};
					if (q < m)
						raw = *reinterpret_cast<const u32x4*>(&recs[st]);
					FinishWith<false>(p, lds, L, uint32_t(sub + q), q < m, raw);
				} else {
					FinishRagged<false>(p, lds, L, finHot, uint32_t(sub + q), q < m, st);
				}
			}
			continue;
		}
		uint32_t s0, s1;
		StreamLaneSplit(eo, m, lead, spanBytes, g.lambda, lane, s0, s1);
		PIRE_SCLK(7);   // (tuning) lane search
		StreamLane S;
		S.nxt = s0;
		S.sEnd = s1;
		S.live = false;
		const bool has = s0 < s1;
		S.E = has ? eo[s0] : kStreamInf;
		S.En = eo[s0 + 1];
		S.laneEnd = has ? eo[s1] : 0u;
		S.dataEnd = S.E < S.laneEnd ? S.laneEnd : 0u;
		// the window in front of the lane's first line (modulo 2^32; lanes without strings stay at 0, where "no boundary
		// ahead" really is far away): the first phase only requests the first line
		S.wpos = has ? (S.E & ~127u) - 128u : 0u;
		S.hs = p.startPerm;
		S.cold = p.startPerm;
		// the two line registers live for the walk of one sub-task only: kept across the set-up of the next one (64 of a
		// lane's 128 registers) they pushed the set-up's temporaries into scratch
		u32x4 a[8], b[8];
		ZeroTile(a);
		ZeroTile(b);
		bool walk = false;
		PIRE_SCLK(2);
		for (;; iter += 2) {
			if (!StreamPhase<START0, WIDE>(p, lds, L, W, K, eo, lineBase, S, a, b, iter, walk))
				break;
			walk = true;
			if (!StreamPhase<START0, WIDE>(p, lds, L, W, K, eo, lineBase, S, b, a, iter + 1, true))
				break;
		}
		PIRE_SCLK(3);
		// ---- the sub-task's results: End(), StateIndex, Final and the counters, 4 x 64 strings at a time (the slots and the
		// end-of-string records of four strings per lane are read before any is used: two dependent LDS reads per string
		// cost a microsecond per 64 strings one after the other)
#if defined(PIRE_EXP) && PIRE_EXP == 1   // timing experiment: no flush
		for (uint32_t base = m; base < m; base += 256) {
#else
		for (uint32_t base = 0; base < m; base += 256) {
#endif
			u32x4 rec[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t q = base + uint32_t(j) * 64 + lane;
				if constexpr (WIDE != 0) {   // every record from memory (the LDS is the rows'); the four loads on their way together
					rec[j] = u32x4{0, 0, 0, 0};
					if (q < m)
						rec[j] = *reinterpret_cast<const u32x4*>(&recs[eo[q + 1]]);
				} else {
					rec[j] = FinRecordOf(p, finHot, q < m, q < m ? eo[q + 1] : 0u);
				}
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t q = base + uint32_t(j) * 64 + lane;
				if (base + uint32_t(j) * 64 < m)
					FinishWith<false>(p, lds, L, uint32_t(sub + q), q < m, rec[j]);
			}
		}
	}
#ifdef PIRE_HIP_TUNING
	if (p.stamps && hasTask) {
		PIRE_SCLK(4);
		if ((threadIdx.x & 63) == 0) {
			for (int k = 0; k < 5; ++k)
				atomicAdd(&p.stamps[k], sclk[k]);
			atomicAdd(&p.stamps[5], 1ull);
			atomicAdd(&p.stamps[6], (unsigned long long)(iter));
			atomicAdd(&p.stamps[7], wall_clock64() - sclkStart);
			for (int k = 5; k < 8; ++k)
				atomicAdd(&p.stamps[4 + k], sclk[k]);
		}
	}
#endif
	if constexpr (WIDE != 0)
		FlushWide(p, lds, W);
	else
		FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ launcher

// pire_hip_config.ragged_variant: 0 = the stream kernel for offset batches of many strings, 1 = never (the ragged kernel
// of ragged.hip), 2 = whenever its results are defined (the tests' A/B).
bool StreamEligible(const ScanParams& p, uint64_t totalBytesHint)
{
	const uint32_t variant = GetConfig().ragged_variant;
	if (variant == 1)
		return false;
	if (!p.offsets || p.ends || p.initIdx || (p.flags & kPermIds) || p.startPerm >= p.hot)
		return false;   // resume states and the segmented scan's batches keep the ragged kernel
	if (p.n >= (1ull << 32) - (1ull << 16) || p.n < 64)
		return false;
	if (variant == 2)
		return true;
	// The stream kernel's fixed part (two searches, the table, the first lines: ~75 us from launch to the first byte
	// walked, against ~50 for the ragged kernel) pays from ~190 MB of URL-sized text on (T = 75 us + bytes / 4.5 TB/s
	// against 50 us + bytes / 2.8 TB/s, profiles/r04_ragged_cases.log); on long strings the two are even.  With device
	// offsets the host does not know the bytes (hint ~0): a million strings stand for them.
	if (totalBytesHint != ~0ull)
		return totalBytesHint >= (160ull << 20);
	return p.n >= (1ull << 20);
}

#ifdef PIRE_HIP_TUNING
static unsigned long long* g_streamClockBuf = nullptr;
// stage clocks: accumulated over launches (no synchronisation here: a drained GPU drops its clocks and the stamps
// would describe another machine); pire_hip_debug_stream_clocks() reads and clears them
static void StreamTuning(ScanParams& p, StreamGeom& g)
{
	p.stamps = nullptr;
	if (getenv("PIRE_HIP_DEBUG_STREAM_CLOCKS")) {
		if (!g_streamClockBuf) {
			(void)hipMalloc(reinterpret_cast<void**>(&g_streamClockBuf), 16 * 8);
			(void)hipMemset(g_streamClockBuf, 0, 16 * 8);
		}
		p.stamps = g_streamClockBuf;
	}
	if (const char* lam = getenv("PIRE_HIP_STREAM_LAMBDA"))
		g.lambda = uint32_t(std::max(1, atoi(lam)));
}
#endif

int LaunchStream(const ScanParams& p0, hipStream_t stream)
{
	if (int rc = CheckCounts(p0))
		return rc;
	ScanParams p = p0;
	p.compact = 0;   // the compact tier's LDS holds the string positions here
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, 0);
	const uint32_t ldsBytes = L.total + kRaggedFinBytes + kStreamWaves * kStreamStageWords * 4;
	const bool start0 = p.startPerm == 0;
	hipError_t e = SetDynamicLds(start0 ? reinterpret_cast<const void*>(ScanStreamKernel<true, 0>) : reinterpret_cast<const void*>(ScanStreamKernel<false, 0>),
	                             ldsBytes);
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	StreamGeom g;
	g.maxStrings = kStreamMaxStrings;
	g.lambda = 16;   // a boundary costs the wave a few lane-steps' worth of instructions (1 / 16 / 64 measured: 16 by a hair); what matters is that keys stay distinct among empty strings
	g.minTaskUnits = 64 * 256;
	// every CU (the kernel starts as many of a block's waves as the batch has work for, see perBlock there); the host
	// cannot size the grid by bytes: with device offsets it does not know them
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(cus), p.n / 64));
	NoteKernel("stream", start0 ? "pirehip::ScanStreamKernel<start0>" : "pirehip::ScanStreamKernel<any start>");
#ifdef PIRE_HIP_TUNING
	StreamTuning(p, g);
#endif
	if (start0)
		hipLaunchKernelGGL((ScanStreamKernel<true, 0>), dim3(unsigned(blocks)), dim3(kStreamWaves * 64), ldsBytes, stream, p, g);
	else
		hipLaunchKernelGGL((ScanStreamKernel<false, 0>), dim3(unsigned(blocks)), dim3(kStreamWaves * 64), ldsBytes, stream, p, g);
	e = hipGetLastError();
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "stream kernel launch");
}

// ---- the class-indexed walk (round 6) -------------------------------------------------------------------------------------
// Offset batches of a table whose scans keep leaving the dense rows took the ragged kernel on the wide walk: one string per
// lane, windows 44 % full on URL-sized strings and every chunk of the longest lane paid by the whole wave -- 0.83-1.03 TB/s
// on a blacklist scanner's URL batches whose working set FITS the rows (profiles/r06_wide_curve.jsonl), a quarter of what
// the same walk does on fixed-length records.  The stream kernel's cut (runs of consecutive strings, whole lines, boundaries
// inside the chunk walk) on the same step; the price is LDS: the strings' positions sit beside the image, so the image is the
// table's first StreamWideTier states (a zipped image: the same rows, fewer headers).
bool StreamWideEligible(const ScanParams& p, uint64_t totalBytesHint)
{
	const uint32_t variant = GetConfig().ragged_variant;
	if (variant == 1 || !p.wideRowsStream || !p.next16 || !p.wideStream)
		return false;
	if (!p.offsets || p.ends || p.initIdx || (p.flags & kPermIds) || p.startPerm >= p.wideStream)
		return false;
	if (p.n >= (1ull << 32) - (1ull << 16) || p.n < 64)
		return false;
	// Opt-in only (ragged_variant = 2).  Measured on URL batches of the blacklist scanners, 8 M strings (profiles/
	// r06_stream_wide_urls.jsonl): where every state of the table has a place in the zipped tier 923 against the ragged
	// kernel's 949 GB/s; wherever lanes leave the tier -- and this image's tier is the smaller one -- 0.33-0.52 against
	// 0.59-1.03 TB/s: in a URL batch every 16-byte chunk of a wave holds a string boundary, a chunk with a boundary AND a lane
	// outside the rows is walked again byte by byte, and with 0.6 % of the steps outside the rows that is every chunk.
	(void)totalBytesHint;
	return variant == 2;
}

int LaunchStreamWide(const ScanParams& p0, hipStream_t stream)
{
	if (int rc = CheckCounts(p0))
		return rc;
	ScanParams p = p0;
	p.wideOutSlot = p0.wide;   // (the table's counters: the slot behind ITS tier counts the samples outside)
	p.wide = p0.wideStream;
	p.wideRows = p0.wideRowsStream;
	const bool zip = p.zipFull != 0;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const WideLayout W = MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, p.zipFull);
	// sub-tasks as large as the image leaves room for (a zipped image of a few thousand states: most of the dense kernel's 1 280)
	StreamGeom g;
	g.maxStrings = W.total + 64 < kLdsPerBlock ? std::min<uint32_t>(kStreamMaxStrings, ((kLdsPerBlock - W.total - 64) / (kStreamWaves * 4) - 16) / 64 * 64) : 0;
	const uint32_t ldsBytes = W.total + kStreamWaves * (g.maxStrings + 16) * 4;
	if (g.maxStrings < kStreamWideStrings || ldsBytes > kLdsPerBlock) {
		SetError("stream kernel on the wide walk: the image does not leave room for the strings' positions");
		return PIRE_HIP_EINVAL;
	}
	hipError_t e = SetDynamicLds(zip ? reinterpret_cast<const void*>(ScanStreamKernel<false, 3>) : reinterpret_cast<const void*>(ScanStreamKernel<false, 2>),
	                             ldsBytes);
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	g.lambda = 16;
	g.minTaskUnits = 64 * 256;
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(cus), p.n / 64));
#ifdef PIRE_HIP_TUNING
	StreamTuning(p, g);
#endif
	NoteKernel("stream_wide", zip ? "pirehip::ScanStreamKernel<wide walk, zipped rows>" : "pirehip::ScanStreamKernel<wide walk>");
	if (zip)
		hipLaunchKernelGGL((ScanStreamKernel<false, 3>), dim3(unsigned(blocks)), dim3(kStreamWaves * 64), ldsBytes, stream, p, g);
	else
		hipLaunchKernelGGL((ScanStreamKernel<false, 2>), dim3(unsigned(blocks)), dim3(kStreamWaves * 64), ldsBytes, stream, p, g);
	e = hipGetLastError();
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "stream kernel launch");
}

}  // namespace pirehip

#ifdef PIRE_HIP_TUNING
// tuning build only: the stage clocks accumulated since the last call -- us per wave: search, table copy, positions + lane
// search, window loop, flush, whole wave; then phases per wave and the number of waves.  Drains the device.
extern "C" int pire_hip_debug_stream_clocks(double* out8)
{
	using namespace pirehip;
	if (!g_streamClockBuf)
		return -1;
	(void)hipDeviceSynchronize();
	unsigned long long c[16];
	(void)hipMemcpy(c, g_streamClockBuf, sizeof c, hipMemcpyDeviceToHost);
	(void)hipMemset(g_streamClockBuf, 0, 16 * 8);
	const double w = double(c[5] ? c[5] : 1);
	for (int k = 0; k < 5; ++k)
		out8[k] = c[k] / w / 100.0;
	out8[5] = c[7] / w / 100.0;
	out8[6] = double(c[6]) / w;
	out8[7] = w;
	fprintf(stderr, "pire_hip stream clocks, inside 'positions + lane search' (us per wave): [flush of the previous sub-task +] the two end "
	        "offsets %.2f | positions into LDS %.2f | lane search %.2f | (rest: shuffles, zeroing the line registers)\n", c[9] / w / 100.0,
	        c[10] / w / 100.0, c[11] / w / 100.0);
	return 0;
}
#endif
