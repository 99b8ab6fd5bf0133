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
//     offsets: 4 rounds for 4 M strings), a task into SUB-TASKS of at most 1 024 strings, and a sub-task among the 64
//     lanes at equal steps of the key again.  Equal work per wave and per lane by construction, whatever the lengths:
//     no work counters, no atomics, no tail of late long strings.
//   * a sub-task's string positions sit in LDS as 32-bit offsets from the line that holds its first byte (4 KiB per
//     wave: the compact tier's place, which this kernel does not use).  A lane finds its first string by a binary
//     search there, reads the end of a string when it starts it, and leaves a string's end state in the slot of its
//     end position (dead by then); when the sub-task is over the wave turns the 1 024 slots into StateIndex / Final /
//     counters with coalesced stores (FinishRagged, 64 strings at a time).
//   * windows: the lane's current line in registers, the next one in flight (group loads + DPP transpose, the ragged
//     and tiled kernels' load path).  Nothing outside the 128-byte lines that hold the sub-task's text is read: a line
//     lies inside one page, so whatever holds the text holds the line.
//   * exactness: a chunk in which a lane leaves the dense rows, or in which a lane meets more than one boundary (strings
//     shorter than 16 bytes, empty strings), is walked again for that lane byte by byte with the exact step
//     (ExactRest).  Results never depend on which rows are dense, on lambda or on how the batch was cut.

#include "device_common.h"

namespace pirehip {

constexpr uint32_t kStreamMaxStrings = 1024;                    // strings of one sub-task
constexpr uint32_t kStreamStageWords = kStreamMaxStrings + 16;  // their positions (m + 1 words) per wave, padded
constexpr uint32_t kStreamInf = 0xFFFFFFFFu;                    // "no boundary ahead": the lane's strings are over
constexpr uint32_t kStreamWaves = 16;

struct StreamGeom {
	uint32_t lambda;         // cost of a string boundary in bytes of walk
	uint32_t minTaskUnits;   // a wave is not started for less than this much key
};

typedef __attribute__((address_space(3))) uint32_t* LdsWordPtr;

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
			const uint32_t from = at ? start : h;
			const uint32_t next = HotLookup(__builtin_amdgcn_perm(from, x, 0x0c0c0400u + uint32_t(b)));
			sn = at ? h : sn;   // behind the lookup: it needs the state in front of the step, not the lookup's result
			h = next;
		}
	}
	hs = h;
	snap = sn;
}

// Bytes from .. 15 of chunk k of the window, exactly, for one lane: boundaries as they come (any number, empty strings
// included), the exact step for the bytes of live strings.  `st` = the state in front of byte `from`.  Rolled: the cold path.
__device__ __forceinline__ void ExactRest(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, LdsWordPtr eo,
                                          const u32x4& v, uint32_t k, uint32_t from, uint32_t st, StreamLane& S, uint32_t sampleLane)
{
#pragma unroll 1
	for (uint32_t i = from; i < 16; ++i) {
		while (S.E - S.wpos == 16u * k + i) {
			StreamBoundary(eo, S, st);
			st = p.startPerm;
		}
		if (S.live) {
			const uint32_t word = i < 8 ? (i < 4 ? v.x : v.y) : (i < 12 ? v.z : v.w);
			st = SlowStep(p, lds, L, st, (word >> (8u * (i & 3u))) & 0xFFu);
		}
	}
	S.hs = st < p.hot ? st : p.hot;
	S.cold = st;
	// tell pire_hip_table_adapt() which rows deserve LDS, sampled like TrapChunk's
	if (S.live && st >= p.hot && (threadIdx.x & 63) == sampleLane) {
		atomicAdd(&p.visitCold[st], 1u);
		atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + kLdsTrapSlot, 1u);
	}
}

// One window: start fetching the next line into `nxt`, walk the line held in `cur`.  Returns whether any lane of the wave
// has a further line.
__device__ __forceinline__ bool StreamPhase(const ScanParams& p, uint8_t* lds, const LdsLayout& L, LdsWordPtr eo,
                                            uint64_t lineBase, StreamLane& S, u32x4 (&cur)[8], u32x4 (&nxt)[8], uint32_t iter,
                                            bool walk)
{
	const uint32_t lane = threadIdx.x & 63;
	WaitAllLoads(cur);
	TransposeTile(cur, lane);
	const bool more = S.laneEnd > S.wpos + 128u;   // boundaries of this lane lie behind this window
	// unconditional (lanes without a further line fetch a harmless valid one), see ragged.hip
	// (the sum modulo 2^32 FIRST: in a sub-task's first phase wpos is -128 mod 2^32 for the lanes whose first line is line 0)
	IssueTileGroup(nxt, S.dataEnd > S.wpos + 128u ? lineBase + uint64_t(uint32_t(S.wpos + 128u)) : reinterpret_cast<uint64_t>(p.hotRows), lane);
	if (lane == (iter & 63) && S.live)   // visit sample, as in the tiled kernel
		atomicAdd(reinterpret_cast<uint32_t*>(lds + L.histOff) + S.hs, 1u);
	// (`walk` is false in a sub-task's first phase only: its window is the line in FRONT of the lanes' first lines, there
	// to get the first lines requested from inside the loop -- asynchronous asm loads issued in front of the loop end up
	// in registers the loop does not use, and the compiler copies them over while they are in flight)
	if (walk)
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint32_t c = (S.E - S.wpos) - 16u * uint32_t(k);   // bytes of this chunk in front of the boundary (>= 16: none)
		const uint32_t hs0 = S.hs;
		if (!__any(c < 16u)) {
			// no string of the wave ends in this chunk: the tiled kernel's step
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
			if (h == p.hot && S.live)
				TrapChunk(p, lds, L, cur[k], hs0, S.hs, S.cold, (iter * 8 + k) & 63);
		} else {
			uint32_t snap;
			StepChunkB(cur[k], c, p.startPerm, S.hs, snap);
			const bool isB = c < 16u;
			// the part of the chunk that belongs to the current string left the dense rows (or was outside them all along),
			// or the part that belongs to the string starting here did: this lane's chunk again, exactly
			const bool trapBefore = S.live && (isB ? snap == p.hot : S.hs == p.hot);
			const bool trapAfter = isB && S.hs == p.hot && S.nxt < S.sEnd;
			bool exact = trapBefore || trapAfter;
			uint32_t from = 0, st = hs0 != p.hot ? hs0 : S.cold;
			if (!exact && isB) {
				StreamBoundary(eo, S, snap);
				if ((S.E - S.wpos) - 16u * uint32_t(k) < 16u) {   // the string that started here ends in this chunk as well
					exact = true;
					from = c;
					st = p.startPerm;
				}
			}
			if (exact)
				ExactRest(p, lds, L, eo, cur[k], uint32_t(k), from, st, S, (iter * 8 + k) & 63);
		}
	}
	// strings that end with the line: their boundary is here, not in a window of its own (which may not exist)
	while (S.E - S.wpos == 128u) {
		StreamBoundary(eo, S, S.hs != p.hot ? S.hs : S.cold);
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

// smallest i in [0, n] with key(i) >= T for two targets at once; key(i) = offsets[i] - off0 + lambda * i is strictly
// increasing and key(n) >= T (the caller clamps T).  64 probes per round and target: the interval shrinks 64-fold.
__device__ __forceinline__ void StreamSearch2(const uint64_t* off, uint64_t off0, uint64_t n, uint32_t lambda, uint64_t T0,
                                              uint64_t T1, uint64_t& r0, uint64_t& r1)
{
	const uint32_t lane = threadIdx.x & 63;
	uint64_t lo[2] = {0, 0}, hi[2] = {n, n};
	const uint64_t T[2] = {T0, T1};
	while (lo[0] < hi[0] || lo[1] < hi[1]) {
		uint64_t pos[2], key[2], step[2];
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			step[j] = (hi[j] - lo[j]) / 64 + 1;
			pos[j] = lo[j] + uint64_t(lane) * step[j];
			key[j] = ~0ull;
			if (pos[j] <= hi[j])
				key[j] = off[pos[j]] - off0 + uint64_t(lambda) * pos[j];
		}
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const unsigned long long ge = __ballot(key[j] >= T[j]);   // monotone: 0...01...1
			const uint32_t f = ge ? uint32_t(__builtin_ctzll(ge)) : 64u;   // number of probes below the target
			if (f == 0) {
				hi[j] = lo[j];
			} else {
				const uint64_t lastBelow = lo[j] + uint64_t(f - 1) * step[j];
				const uint64_t firstAt = lo[j] + uint64_t(f) * step[j];
				lo[j] = lastBelow + 1;
				if (f < 64 && firstAt < hi[j])
					hi[j] = firstAt;
			}
			lo[j] = Uniform64(lo[j]);
			hi[j] = Uniform64(hi[j]);
		}
	}
	r0 = lo[0];
	r1 = lo[1];
}

__global__ __launch_bounds__(1024) void ScanStreamKernel(ScanParams p, StreamGeom g)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, 0);
	FinRec* finHot = reinterpret_cast<FinRec*>(lds + L.total);
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	LdsWordPtr eo = reinterpret_cast<LdsWordPtr>(
		static_cast<uintptr_t>(L.total + kRaggedFinBytes + wave * kStreamStageWords * 4u));

	// ---- this wave's task: strings [i0, i1), found before the table is copied (the searches' round trips overlap the
	// other waves' part of the copy)
	const uint64_t off0 = p.offsets[0], offN = p.offsets[p.n];
	const uint64_t totalKey = (offN - off0) + uint64_t(g.lambda) * p.n;
	const uint64_t W = uint64_t(gridDim.x) * wavesPerBlock;
	uint64_t K = totalKey / g.minTaskUnits;
	K = K < 1 ? 1 : K > W ? W : K;
	const uint64_t perTask = (totalKey + K - 1) / K;
	// fewer tasks than waves: every block takes its share of them (ceil(K / blocks) of its waves work), so that a batch
	// that does not fill the chip still uses every CU's LDS bandwidth instead of the first K / 16 CUs'
	const uint64_t perBlock = (K + gridDim.x - 1) / gridDim.x;
	const uint64_t gw = wave < perBlock ? uint64_t(blockIdx.x) * perBlock + wave : K;
	uint64_t i0 = 0, i1 = 0;
	if (gw < K) {
		const uint64_t T0 = gw * perTask, T1 = (gw + 1) * perTask;
		StreamSearch2(p.offsets, off0, p.n, g.lambda, T0 < totalKey ? T0 : totalKey, T1 < totalKey ? T1 : totalKey, i0, i1);
		if (gw == K - 1)
			i1 = p.n;
	}
	{
		const FinRec* recs = (p.flags & PIRE_HIP_RUN_END) ? p.finEnd : p.finSelf;
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			finHot[i] = recs[i];
	}
	LoadTableToLds(p, lds, L);   // ends with a barrier

	const uint64_t textBase = reinterpret_cast<uint64_t>(p.text);
	uint32_t iter = 0;
	for (uint64_t sub = i0; sub < i1; sub += kStreamMaxStrings) {
		const uint32_t m = uint32_t(i1 - sub < kStreamMaxStrings ? i1 - sub : kStreamMaxStrings);
		// ---- the sub-task's string positions into LDS, relative to the line that holds its first byte
		const uint64_t offA = p.offsets[sub], offZ = p.offsets[sub + m];
		const uint64_t firstByte = textBase + offA;
		const uint64_t lineBase = Uniform64(firstByte & ~uint64_t(127));
		const uint32_t lead = uint32_t(firstByte) & 127u;
		if (offZ - offA >= 0xFFFF0000ull) {
			// positions that do not fit 32 bits (a string of 4 GiB among short ones): every lane takes whole strings and
			// walks them byte by byte from memory
			for (uint32_t base = 0; base < m; base += 64) {
				const uint32_t q = base + lane;
				uint32_t st = p.startPerm;
				if (q < m)
					for (uint64_t at = p.offsets[sub + q], end = p.offsets[sub + q + 1]; at < end; ++at)
						st = SlowStep(p, lds, L, st, p.text[at]);
				FinishRagged<false>(p, lds, L, finHot, uint32_t(sub + q), q < m, st);
			}
			continue;
		}
		for (uint32_t base = 0; base <= m; base += 64 * 8) {
			uint64_t v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t q = base + uint32_t(j) * 64 + lane;
				v[j] = q <= m ? p.offsets[sub + q] : 0;
			}
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t q = base + uint32_t(j) * 64 + lane;
				if (q <= m)
					eo[q] = lead + uint32_t(v[j] - offA);
			}
		}
		// ---- the lane's strings: equal steps of the key over the 64 lanes
		const uint32_t keyAll = uint32_t(offZ - offA) + g.lambda * m;   // < 2^32: m <= 1024, the span is checked above
		const uint32_t perLane = (keyAll + 63) / 64;                    // >= 1: m >= 1
		uint32_t s0;
		{
			const uint32_t target = lane * perLane;
			uint32_t lo = 0, hi = m;
#pragma unroll 1
			for (int it = 0; it < 11; ++it) {   // 2^11 > kStreamMaxStrings + 1 candidates
				const uint32_t mid = (lo + hi) >> 1;
				const bool below = lo < hi && (eo[mid] - lead) + g.lambda * mid < target;
				const bool shrink = lo < hi && !below;
				lo = below ? mid + 1 : lo;
				hi = shrink ? mid : hi;
			}
			s0 = lo;
		}
		uint32_t s1 = uint32_t(__shfl_down(int(s0), 1));
		if (lane == 63)
			s1 = m;
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
		for (;; iter += 2) {
			if (!StreamPhase(p, lds, L, eo, lineBase, S, a, b, iter, walk))
				break;
			walk = true;
			if (!StreamPhase(p, lds, L, eo, lineBase, S, b, a, iter + 1, true))
				break;
		}
		// ---- the sub-task's results: End(), StateIndex, Final and the counters, 64 strings at a time
		for (uint32_t base = 0; base < m; base += 64) {
			const uint32_t q = base + lane;
			const bool act = q < m;
			const uint32_t st = act ? eo[q + 1] : 0u;
			FinishRagged<false>(p, lds, L, finHot, uint32_t(sub + q), act, st);
		}
	}
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
	// strings much longer than a lane's share of the batch leave most lanes without one: the ragged kernel, which deals
	// single strings out, is the better one there.  With device offsets the host does not know the lengths (hint ~0).
	if (totalBytesHint != ~0ull && totalBytesHint / p.n > 1024)
		return false;
	return p.n >= 16384;
}

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
	hipError_t e = SetDynamicLds(reinterpret_cast<const void*>(ScanStreamKernel), ldsBytes);
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	StreamGeom g;
	g.lambda = 4;   // a boundary costs the wave a few lane-steps' worth of instructions; what matters is that keys stay distinct among empty strings
	g.minTaskUnits = 64 * 256;
	// every CU (the kernel starts as many of a block's waves as the batch has work for, see perBlock there); the host
	// cannot size the grid by bytes: with device offsets it does not know them
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(cus), p.n / 64));
	NoteKernel("stream", "pirehip::ScanStreamKernel");
	hipLaunchKernelGGL(ScanStreamKernel, dim3(unsigned(blocks)), dim3(kStreamWaves * 64), ldsBytes, stream, p, g);
	e = hipGetLastError();
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "stream kernel launch");
}

}  // namespace pirehip
