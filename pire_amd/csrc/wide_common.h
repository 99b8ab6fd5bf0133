// Device-side pieces of the class-indexed walk shared by its kernels: wide.hip (fixed-length records) and the WIDE
// instantiation of the ragged kernel (ragged.hip, offset batches).  DESIGN.md 4.8; the LDS image: internal.h WideLayout.
#pragma once

#include "device_common.h"

namespace pirehip {

// wave-uniform constants of a launch
struct WideConst {
	uint32_t pitch;    // bytes per row
	uint32_t flagsOff; // byte offset of a row's flags (2 * letters)
	uint32_t full;     // states with a row of their own (== p.wide unless the image is zipped)
	uint32_t hOff;     // zipped image: LDS byte address of the headers
	uint32_t xRel;     // zipped image: xOff - 6 * full (state st's exception targets are at st * 6 + xRel)
};

__device__ __forceinline__ WideConst MakeWideConst(const ScanParams& p, const WideLayout& W)
{
	WideConst K;
	K.pitch = W.pitch;
	K.flagsOff = p.letters * 2;
	K.full = W.full;
	K.hOff = W.hOff;
	K.xRel = W.xOff - 2 * kZipExceptions * W.full;
	return K;
}

typedef const __attribute__((address_space(3))) uint32_t* LdsU32Ptr;

// A row entry: the u16 at byte `c2` (2 * letter class) of state `st`'s row.  The rows start at LDS byte 256: the DS
// instruction's immediate offset, so the address is ONE v_mad_u32_u24.
// ZIP (internal.h MakeWideLayout): the state's header first -- the row it leans on and the <= 3 letters it differs in --,
// then ONE u16 read: the exception's target if the letter is one of them, else that row's entry.  (c2 * 0x4081 puts the
// class into the three letter fields at once; a field of the xor that is zero is a match, and a state's letters are distinct.)
template <bool ZIP>
__device__ __forceinline__ uint32_t WideEntry(uint32_t st, const WideConst& K, uint32_t c2)
{
	if constexpr (!ZIP) {
		return *reinterpret_cast<LdsU16Ptr>(static_cast<uintptr_t>(__umul24(st, K.pitch) + c2 + 256u));
	} else {
		const uint32_t h = *reinterpret_cast<LdsU32Ptr>(static_cast<uintptr_t>(K.hOff + (st << 2)));
		const uint32_t x = h ^ __umul24(c2, kZipMulC2);
		const uint32_t xa = __umul24(st, 2 * kZipExceptions) + K.xRel;
		uint32_t a = __umul24(h >> 22, K.pitch) + c2 + 256u;
		a = (x & 0x3F8000u) ? a : xa + 4;
		a = (x & 0x007F00u) ? a : xa + 2;
		a = (x & 0x0000FEu) ? a : xa;
		return *reinterpret_cast<LdsU16Ptr>(static_cast<uintptr_t>(a));
	}
}

// The flags of a state of the tier (the escape state's: 0).  Zipped image: only states with a row of their own have any
// the walk looks at (an absorbing state keeps its row, table.cpp PlanZip).
template <bool ZIP>
__device__ __forceinline__ uint32_t WideFlags(uint32_t st, const WideConst& K)
{
	const uint32_t row = ZIP ? (st < K.full ? st : K.full) : st;
	return *reinterpret_cast<LdsU16Ptr>(static_cast<uintptr_t>(__umul24(row, K.pitch) + K.flagsOff + 256u));
}

// One visit sample of tier state `st` (or of the escape state: the lane is outside the tier).  The counters of the states
// with a row are in LDS (the hot ones: same-address atomics); the zipped states' go straight to memory (each carries little).
template <bool ZIP>
__device__ __forceinline__ void WideSample(const ScanParams& p, uint8_t* lds, const WideLayout& W, uint32_t st)
{
	if (!ZIP || st < W.full)
		atomicAdd(reinterpret_cast<uint32_t*>(lds + W.histOff) + st, 1u);
	else if (st == p.wide)
		atomicAdd(reinterpret_cast<uint32_t*>(lds + W.histOff) + W.full, 1u);
	else
		atomicAdd(&p.visitWide[st], 1u);
}

// The exact step for a state without a row (device ids in and out).
template <bool N16>
__device__ __forceinline__ uint32_t WideNext(const ScanParams& p, uint32_t st, uint32_t cls)
{
	if (N16)
		return p.next16[size_t(st) * p.letters + cls];
	return p.nextPerm[size_t(st) * p.letters + cls];
}

// ... with the byte's doubled class as the caller has it (cls8 holds 2 * class).  The u16 table's entry is a 32-bit byte
// offset from a wave-uniform base (65 536 states x 127 letters x 2 bytes < 2^24): one v_mad_u32_u24 and a load that adds
// the base itself, instead of 64-bit address arithmetic in every step of the re-walk.
template <bool N16>
__device__ __forceinline__ uint32_t WideNextC2(const ScanParams& p, uint32_t st, uint32_t c2)
{
	if (N16) {
		const uint32_t off = __umul24(st, p.letters * 2u) + c2;
		return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(p.next16) + off);
	}
	return p.nextPerm[size_t(st) * p.letters + (c2 >> 1)];
}

// A lane sits in the escape row after the 16 bytes `v`: walk them again from the state it was in before them, exactly,
// device ids all the way: the row's entry in LDS, and ONE load from the table in memory in the steps in which that entry
// says "no row" -- for a lane that leaves the rows with this step and for one that is outside them already alike (the
// table answers both from the state's id).  The class of the next byte is asked for before this byte's step.  Rolled on
// purpose (instantiated once per unrolled chunk of the tile walk).
// History (profiles/r05b..r05d_wide_curve.jsonl, dict_1k / k512: 2 738 states visited, 2 041 rows, 1.7 % of the steps
// outside them, EVERY wave-chunk with a lane outside): row / row's id / table as three dependent round trips per
// iteration 464 GB/s; every lane of the re-walk through the table in memory, one round trip 887 GB/s -- but 64 scattered
// accesses per step where all lanes re-walk (the L1 serves about one per clock and CU: dict_10k / k10000 454 GB/s); no
// re-walk at all, every step asking whether a lane is outside the rows, 644 GB/s here and 2.4 instead of 4.25 TB/s where
// the working set fits (a third form of the kernel, removed again); rows for the lanes that have one, the table for the
// others and -- in an arm of its own -- for those that leave with this step: 651 GB/s (two round trips in a row).
// What pire_hip_table_adapt() ranks the states beyond the rows by: every 64th re-walk leaves, at one rotating step, the
// state of the first lane that is outside the rows, counted once per lane that is outside them in that step -- 64 chunks x
// 16 steps per sample and lane: the 1 024 lane-steps a sample of the dense walk's trap path stands for (table.cpp
// AdaptTable), so that the measured share of the steps outside the rows is a share.  (The first form sampled one fixed lane of 64 at the chunk's end,
// like TrapChunk: a state that carries 1e-6 of the steps was never seen and stayed without a row, and although 2 148
// rows were there for 1 530 visited states 36 % of all wave-chunks were walked twice, profiles/r05_pmc_wide_first.txt.)
// CREDIT = false: no samples from here (an A/B switch of round 6; every kernel takes them).
template <bool N16, bool ZIP, bool CREDIT = true>
__device__ __forceinline__ void WideTrapChunk(const ScanParams& p, uint8_t* lds, const WideLayout& W, const WideConst& K, u32x4 v,
                                              uint32_t st0, uint32_t& st, uint32_t& cold, uint32_t sampleStep)
{
	uint32_t sid = st0 < p.wide ? st0 : cold;   // the device id of the state the chunk started in
	uint32_t c2 = HotLookup(v.x & 0xFFu);       // 2 * letter class
	// one wave-chunk more that is walked twice (exact count, block-local); every 64th of them leaves a sample
	const unsigned long long lanes = __ballot(true);
	uint32_t nth = 0;
	if ((threadIdx.x & 63) == uint32_t(__ffsll(lanes)) - 1u)
		nth = atomicAdd(reinterpret_cast<uint32_t*>(lds + W.progOff) + 1, 1u);
	const uint32_t nthU = uint32_t(__builtin_amdgcn_readfirstlane(int(nth)));
	const bool sampled = (nthU & 63u) == 0;
	// WHICH step of the sampled re-walk leaves the sample: drawn from the re-walk's number, not taken from the chunk's place in
	// its window (round 6: `sampleStep` was (8 * iteration + chunk) % 16 -- for chunk k of a window only step k or k + 8 -- and an
	// offset batch's strings start with their windows: a state a URL is in at its 6th byte and nowhere else -- "scheme:/",
	// 1.4 % of all steps -- was never seen once it was outside the tier, and every URL left the rows there for good;
	// tools/ranking_quality.py)
	// (... and from the block's number too: the count of re-walks restarts with every launch, a block of a URL batch makes a few
	// hundred -- from the count alone only the first handful of draws ever happened, steps 0, 9, 3, 13, 7, and nothing else)
	sampleStep = (((nthU >> 6) + blockIdx.x * 0x632BE5ABu) * 0x9E3779B1u) >> 28;
	// (a dword per trip, its four steps unrolled, the bytes as bit fields: see WideTrapChunk2)
#pragma unroll 1
	for (uint32_t w = 0; w < 4; ++w) {
		const uint32_t x = v.x;
		v.x = v.y;
		v.y = v.z;
		v.z = v.w;
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			const uint32_t c2n = HotLookup(k < 3 ? (x >> (8 * k + 8)) & 0xFFu : v.x & 0xFFu);   // the next byte's (behind the 16th: unused)
			// the sample: the state IN FRONT of the drawn step -- the state whose row (or table line) the step looks up.  (Round 5
			// took the state behind it: the state every string starts in is in front of a step and never behind one, so once it
			// was outside the tier it stayed there -- 1.8 % of a URL batch's steps, every string leaving the rows with its first
			// byte, and nothing to tell adapt() about it; tools/ranking_quality.py.)
			if (CREDIT && sampled && w * 4 + k == sampleStep) {
				const bool out = sid >= p.wide;
				const unsigned long long m = __ballot(out);
				if (out && (threadIdx.x & 63) == uint32_t(__ffsll(m)) - 1u)
					atomicAdd(&p.visitCold[sid], uint32_t(__popcll(m)));   // (see above: the step's lanes outside the rows)
			}
			// the row's entry (a state without a row reads the escape row: "no row") ...
			const uint32_t e = WideEntry<ZIP>(sid < p.wide ? sid : p.wide, K, c2);
			uint32_t next = e;
			if (e == p.wide) {   // ... and, for the lanes it sends outside the rows or that are there already, the table in memory
				next = WideNextC2<N16>(p, sid, c2);
				asm volatile("" : "+v"(next));   // (the wait belongs in here: left to the join it is a vmcnt(0) every lane passes)
			}
			sid = next;
			c2 = c2n;
		}
	}
	st = sid < p.wide ? sid : p.wide;
	cold = sid;
}

// 16 bytes through the rows in LDS; lanes that leave them are re-walked exactly.
// (WideChunk2's `direct` mode was tried here too, same box, same process: + 12 % where 3 % of the steps are outside the rows,
// - 4 % where 30 % are -- which is where this form of the kernel runs when the batch fills the chip -- and the ranking the
// next adapt() took from its samples cost the two-strings form 8 % on dict_10k / k512; with "two chunks in a row before a
// wave stops trying the rows" on top: the same.  Not kept.)
template <bool N16, bool ZIP, bool CREDIT = true>
__device__ __forceinline__ void WideChunk(const ScanParams& p, uint8_t* lds, const WideLayout& W, const WideConst& K, const u32x4 v,
                                          uint32_t& st, uint32_t& cold, uint32_t sampleLane)
{
	const uint32_t st0 = st;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		// the four classes first: independent of the walk, the LDS serves them while the chain below waits for its rows
		const uint32_t c0 = HotLookup(x & 0xFFu);
		const uint32_t c1 = HotLookup((x >> 8) & 0xFFu);
		const uint32_t c2 = HotLookup((x >> 16) & 0xFFu);
		const uint32_t c3 = HotLookup(x >> 24);
		st = WideEntry<ZIP>(st, K, c0);
		st = WideEntry<ZIP>(st, K, c1);
		st = WideEntry<ZIP>(st, K, c2);
		st = WideEntry<ZIP>(st, K, c3);
	}
	if (st == p.wide)
		WideTrapChunk<N16, ZIP, CREDIT>(p, lds, W, K, v, st0, st, cold, sampleLane & 15u);
}


// ---- two strings per lane (wide.hip ScanWide2Kernel) --------------------------------------------------------------------
// Beyond the rows the walk is bound by the latency of its loads -- an L2 hit is ~500 clocks against ~150 for an LDS step --
// and a CU holds 16 waves whatever happens (the image is its whole LDS).  What is left is more dependent chains per
// wave: every lane walks TWO strings, step by step in turn, so that the two lookups -- and, in the re-walk, the two
// loads from the table -- of a step are on their way together.
// (FOUR strings per lane on half-line tiles -- 64 bytes of each string per phase, fetched by groups of four lanes, 4 x 4
// transposes inside the quads, every line read from HBM twice -- was built and measured, profiles/
// r05j_wide_curve_four_chains.jsonl: 1.42 against 1.29 TB/s on dict_1k / k512, 1.40 against 1.30 on dict_10k / k32, no
// gain from 6 % of the steps outside the rows on, 3.55 against 4.28 where the working set fits.  With more chains in
// lock step every re-walk step has a lane outside the rows and more of them per load; not kept.  Once more with the re-walk
// loop of today -- a third of the instructions, waves that skip the attempt on the rows --, r05l_wide_curve_four_chains_lean_loop.jsonl:
// 1.86 against 1.82 on k512, 1.09 against 1.17 on k1000, 1.66 against 1.70 on dict_10k / k32: what bounds the walk beyond the
// rows now is the rate of its scattered table loads, ~0.26 per clock and CU whatever the number of chains.  The address unit is
// busy 81 % of the kernel there (r05_wide_pmc_dict_1k_k512.txt: 16.5 busy cycles per load instruction); ONE load instruction per step
// for the lanes that need an entry of either chain, a second only for lanes that need both, halves the instructions and changes
// nothing: 1.77 against 1.82 TB/s -- busy waiting for the L2, not issuing.)

// A lane of either string sits in the escape row after the chunk: both strings' 16 bytes again, exactly (WideTrapChunk
// for two chains; a chain that did not leave the rows is walked again as well -- it costs nothing in lock step and ends
// where it ended).  One round trip to the table per step serves both chains.
// `direct` (wave-uniform, != 0: the wave's count of such chunks): the wave did not try the rows alone first -- its last chunk
// left them --, all lanes are here and this IS the walk of the chunk; the return value says whether a lane left the rows
// in it (else: whether to come here directly next time, i.e. yes).
template <bool N16, bool ZIP>
__device__ __forceinline__ uint32_t WideTrapChunk2(const ScanParams& p, uint8_t* lds, const WideLayout& W, const WideConst& K, u32x4 va,
                                                   u32x4 vb, uint32_t sa0, uint32_t sb0, uint32_t& sa, uint32_t& sb, uint32_t& colda,
                                                   uint32_t& coldb, uint32_t sampleStep, uint32_t direct)
{
	uint32_t ia = sa0 < p.wide ? sa0 : colda, ib = sb0 < p.wide ? sb0 : coldb;
	uint32_t c2a = HotLookup(va.x & 0xFFu), c2b = HotLookup(vb.x & 0xFFu);
	const unsigned long long lanes = __ballot(true);
	uint32_t nth = direct << 1;
	if (!direct) {
		if ((threadIdx.x & 63) == uint32_t(__ffsll(lanes)) - 1u)
			nth = atomicAdd(reinterpret_cast<uint32_t*>(lds + W.progOff) + 1, 2u);   // wave-chunks walked twice (exact): two here
		nth = uint32_t(__builtin_amdgcn_readfirstlane(int(nth)));
	}
	const bool sampled = (nth & 126u) == 0;
	sampleStep = (((nth >> 7) + blockIdx.x * 0x632BE5ABu) * 0x9E3779B1u) >> 28;   // (drawn from the re-walk's and the block's number: see WideTrapChunk)
	bool left = false;
	// (a dword of each string per trip, its four steps unrolled: the bytes are bit fields of one register -- shifting the
	// 16 bytes down by one in every step was 10 of the step's ~35 vector instructions, and four waves share a SIMD)
#pragma unroll 1
	for (uint32_t w = 0; w < 4; ++w) {
		const uint32_t xa = va.x, xb = vb.x;
		va.x = va.y;
		va.y = va.z;
		va.z = va.w;
		vb.x = vb.y;
		vb.y = vb.z;
		vb.z = vb.w;
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			// (behind the chunk's last byte: whatever the register holds -- any byte value is a valid address of cls8)
			const uint32_t bna = k < 3 ? (xa >> (8 * k + 8)) & 0xFFu : va.x & 0xFFu;
			const uint32_t bnb = k < 3 ? (xb >> (8 * k + 8)) & 0xFFu : vb.x & 0xFFu;
			const uint32_t c2an = HotLookup(bna), c2bn = HotLookup(bnb);
			if (sampled && w * 4 + k == sampleStep) {   // (the states IN FRONT of the drawn step: see WideTrapChunk)
				const uint32_t out = ia >= p.wide ? ia : ib;
				const bool is = ia >= p.wide || ib >= p.wide;
				const unsigned long long m = __ballot(is);
				const uint32_t weight = uint32_t(__popcll(__ballot(ia >= p.wide)) + __popcll(__ballot(ib >= p.wide)));
				if (is && (threadIdx.x & 63) == uint32_t(__ffsll(m)) - 1u)
					atomicAdd(&p.visitCold[out], weight);
			}
			const uint32_t ea = WideEntry<ZIP>(ia < p.wide ? ia : p.wide, K, c2a);
			const uint32_t eb = WideEntry<ZIP>(ib < p.wide ? ib : p.wide, K, c2b);
			uint32_t na = ea, nb = eb;
			if (ea == p.wide || eb == p.wide) {
				if (ea == p.wide)
					na = WideNextC2<N16>(p, ia, c2a);
				if (eb == p.wide)
					nb = WideNextC2<N16>(p, ib, c2b);
				asm volatile("" : "+v"(na), "+v"(nb));   // both loads on their way, ONE wait, inside this arm
				left = true;
			}
			ia = na;
			ib = nb;
			c2a = c2an;
			c2b = c2bn;
		}
	}
	sa = ia < p.wide ? ia : p.wide;
	colda = ia;
	sb = ib < p.wide ? ib : p.wide;
	coldb = ib;
	if (!direct)
		return 1u;
	const unsigned long long out = __ballot(left);
	if (out && (threadIdx.x & 63) == 0)
		atomicAdd(reinterpret_cast<uint32_t*>(lds + W.progOff) + 1, 2u);   // (would have been walked twice)
	return uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(out) | uint32_t(out >> 32)))) ? direct + 1u : 0u;
}

// 16 bytes of each of the lane's two strings through the rows, the two chains' lookups in turn.
// `direct` (wave-uniform, kept by the caller from chunk to chunk): once a chunk left the rows the wave's next chunks skip
// the attempt on the rows alone -- with 1.7 % of the steps outside them EVERY wave-chunk has such a lane, and the first
// pass is a quarter of the time for nothing -- until a chunk stays inside them.
template <bool N16, bool ZIP>
__device__ __forceinline__ void WideChunk2(const ScanParams& p, uint8_t* lds, const WideLayout& W, const WideConst& K, const u32x4 va,
                                           const u32x4 vb, uint32_t& sa, uint32_t& sb, uint32_t& colda, uint32_t& coldb,
                                           uint32_t sampleLane, uint32_t& direct)
{
	const uint32_t sa0 = sa, sb0 = sb;
	if (direct) {
		direct = WideTrapChunk2<N16, ZIP>(p, lds, W, K, va, vb, sa0, sb0, sa, sb, colda, coldb, sampleLane & 15u, direct);
		return;
	}
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t xa = va[w], xb = vb[w];
		const uint32_t a0 = HotLookup(xa & 0xFFu), b0 = HotLookup(xb & 0xFFu);
		const uint32_t a1 = HotLookup((xa >> 8) & 0xFFu), b1 = HotLookup((xb >> 8) & 0xFFu);
		const uint32_t a2 = HotLookup((xa >> 16) & 0xFFu), b2 = HotLookup((xb >> 16) & 0xFFu);
		const uint32_t a3 = HotLookup(xa >> 24), b3 = HotLookup(xb >> 24);
		sa = WideEntry<ZIP>(sa, K, a0);
		sb = WideEntry<ZIP>(sb, K, b0);
		sa = WideEntry<ZIP>(sa, K, a1);
		sb = WideEntry<ZIP>(sb, K, b1);
		sa = WideEntry<ZIP>(sa, K, a2);
		sb = WideEntry<ZIP>(sb, K, b2);
		sa = WideEntry<ZIP>(sa, K, a3);
		sb = WideEntry<ZIP>(sb, K, b3);
	}
	const bool trapped = sa == p.wide || sb == p.wide;
	if (trapped)
		WideTrapChunk2<N16, ZIP>(p, lds, W, K, va, vb, sa0, sb0, sa, sb, colda, coldb, sampleLane & 15u, 0u);
	const unsigned long long any = __ballot(trapped);
	direct = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(any) | uint32_t(any >> 32)))) ? 1u : 0u;
}

// The first `count` (0..15) bytes of v through the rows: the whole chunk is walked, unrolled like WideChunk, and the
// state after byte `count` is kept; lanes with count == 0 keep their state (the ragged kernel's last, partial chunk of a
// string: StepPartial of ragged.hip for this walk).  A lane that left the rows inside its bytes is re-walked exactly,
// byte by byte through the table in memory (rolled; few lanes, few bytes).
template <bool N16, bool ZIP>
__device__ __forceinline__ void WidePartial(const ScanParams& p, const WideConst& K, const u32x4 v, uint32_t count, uint32_t& st,
                                            uint32_t& cold)
{
	const uint32_t st0 = st;
	uint32_t h = st, snap = st;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (w == 3 && j == 3)
				break;
			h = WideEntry<ZIP>(h, K, HotLookup((x >> (8 * j)) & 0xFFu));
			snap = count == uint32_t(4 * w + j + 1) ? h : snap;
		}
	}
	st = snap;
	if (count != 0 && st == p.wide) {
		uint32_t sid = st0 < p.wide ? st0 : cold;
		u32x4 t = v;
#pragma unroll 1
		for (uint32_t i = 0; __any(i < count); ++i) {
			if (i < count)
				sid = WideNext<N16>(p, sid, HotLookup(t.x & 0xFFu) >> 1);
			t.x = __builtin_amdgcn_alignbit(t.y, t.x, 8);
			t.y = __builtin_amdgcn_alignbit(t.z, t.y, 8);
			t.z = __builtin_amdgcn_alignbit(t.w, t.z, 8);
			t.w >>= 8;
		}
		st = sid < p.wide ? sid : p.wide;
		cold = sid;
	}
}

// Block-wide: the walk's image into LDS (cls8 at byte 0, the rows behind it), counters and samples zeroed.
__device__ inline void LoadWideToLds(const ScanParams& p, uint8_t* lds, const WideLayout& W)
{
	uint32_t cls8 = 0;
	if (threadIdx.x < 256)
		cls8 = p.cls[threadIdx.x];
	CopyToLds16<4>(lds + W.rowsOff, p.wideRows, (W.imageEnd - W.rowsOff) / 16);
	__syncthreads();   // the copy's last unit may reach past the image
	if (threadIdx.x < 256)
		lds[threadIdx.x] = uint8_t(2 * cls8);
	for (uint32_t i = threadIdx.x; i < W.rows; i += blockDim.x)
		reinterpret_cast<uint32_t*>(lds + W.histOff)[i] = 0;
	for (uint32_t i = threadIdx.x; i < (W.total - W.countsOff) / 4; i += blockDim.x)
		reinterpret_cast<uint32_t*>(lds + W.countsOff)[i] = 0;
	__syncthreads();
}

// Block-wide, at the end of a launch: visit samples, the count of wave-chunks walked twice, the match counters.
__device__ inline void FlushWide(const ScanParams& p, uint8_t* lds, const WideLayout& W)
{
	__syncthreads();
	const uint32_t* hist = reinterpret_cast<const uint32_t*>(lds + W.histOff);
	const uint32_t* prog = reinterpret_cast<const uint32_t*>(lds + W.progOff);
	for (uint32_t i = threadIdx.x; i <= W.full; i += blockDim.x)   // (the last slot -> `wide`: samples that found their lane outside the tier)
		if (hist[i])
			atomicAdd(&p.visitWide[i < W.full ? i : p.wideOutSlot], hist[i]);
	if (threadIdx.x == 0 && prog[1]) {
		atomicAdd(&p.visitHot[kWideTrapSlot], prog[1]);
		const uint32_t total = atomicAdd(&p.visitHot[kTrapSlot], prog[1]) + prog[1];
		if (p.trapSignal)
			__hip_atomic_store(p.trapSignal, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (p.outCounts) {
		const uint32_t* cnt = reinterpret_cast<const uint32_t*>(lds + W.countsOff);
		for (uint32_t i = threadIdx.x; i < p.regexps + 2; i += blockDim.x)
			if (cnt[i])
				atomicAdd(&p.outCounts[i], (unsigned long long)cnt[i]);
	}
}

}  // namespace pirehip
