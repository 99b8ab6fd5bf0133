// The wide walk (round 5): fixed-length records of tables whose scans keep leaving the 255 dense rows.  DESIGN.md 4.8.
//
// The reference's step costs the same in any state (multi.h:169-192: letter = m_letters[ch]; state = row[letter]).  The
// dense rows of tiled.hip are that step with the letter lookup folded in -- one LDS gather per byte -- but only 255
// states have one, and a 16-byte chunk in which a lane leaves them is walked a second time.  A dictionary automaton
// (samples/blacklist/blacklist.cpp:65-76) or a glued table whose text keeps matches alive visits thousands of states;
// nearly every wave-chunk then holds a lane outside the dense rows.  For such tables this kernel IS multi.h:169-192:
//
//     c   = cls8[byte]                       ds_read_u8, off the dependent chain (cls8 at LDS address 0: the byte is the address)
//     st  = u16[rows + st * pitch + c]       v_mad_u32_u24 + ds_read_u16 (offset: rows), the chain; entries are state ids
//
// with class-indexed u16 rows of the first `wide` states of the ranking filling the CU's whole LDS (internal.h
// WideLayout: 1 700 states at 44 letters, 2 040 at 34).  Targets without a row lead to an absorbing escape row (id ==
// wide); a lane found there after a chunk is re-walked from the chunk's first byte, one load per byte -- its row in LDS
// while its state has one, the exact table in memory (u16 entries when the ids fit: half the cache footprint) while it
// has none.  Bit-exact for any table and any ranking, like every kernel of the path.  Text path, task numbering, wave
// levelling: tiled.hip's.

#include "device_common.h"
#include "wide_common.h"

namespace pirehip {

// Whole-line loads of one 128-byte tile of 64 strings (tiled.hip IssueTile; a copy of its own, like pair.hip's, so that
// the headline kernel's translation unit stays what it was measured as)
__device__ __forceinline__ void WideIssueTile(u32x4 (&r)[8], uint32_t voff, uint64_t tileBase, uint64_t stride)
{
	const uint64_t b0 = tileBase, b1 = b0 + stride, b2 = b1 + stride, b3 = b2 + stride, b4 = b3 + stride,
	               b5 = b4 + stride, b6 = b5 + stride, b7 = b6 + stride;
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
}

template <int TILES_BEHIND>
__device__ __forceinline__ void WideWaitTile(u32x4 (&r)[8])
{
	asm volatile("s_waitcnt vmcnt(%8)"
	             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
	             : "n"(TILES_BEHIND * 8));
}

template <bool N16, bool ZIP>
__device__ __forceinline__ void WidePhase(const ScanParams& p, uint8_t* lds, const WideLayout& W, const WideConst& K, uint64_t rowBase,
                                          uint64_t chainBase, uint32_t voff, uint64_t istride, uint32_t lane, uint32_t t,
                                          uint32_t lastTile, u32x4 (&cur)[8], u32x4 (&refill)[8], uint32_t& st, uint32_t& cold,
                                          uint32_t* prog, uint32_t& myTiles)
{
	{   // the waves of a block kept in step (tiled.hip, EQ)
		uint32_t sum = 0;
		if (lane == 0)
			sum = atomicAdd(prog, 1u) + 1;
		sum = uint32_t(__builtin_amdgcn_readfirstlane(int(sum)));
		const uint32_t mine = ++myTiles;
		constexpr uint32_t margin = 4;
		if (mine * (blockDim.x >> 6) > sum + margin)
			__builtin_amdgcn_s_setprio(0);
		else if (mine * (blockDim.x >> 6) + margin < sum)
			__builtin_amdgcn_s_setprio(3);
		else
			__builtin_amdgcn_s_setprio(1);
	}
	const uint64_t ahead = t < lastTile ? rowBase + uint64_t(t + 1) * 128 : chainBase;
	WideIssueTile(refill, voff, ahead, istride);
	WideWaitTile<1>(cur);
	TransposeTile(cur, lane);
#pragma unroll
	for (int k = 0; k < 8; ++k)
		WideChunk<N16, ZIP>(p, lds, W, K, cur[k], st, cold, (t * 8 + k) & 63);
	// visit sample: one lane per wave per tile (the escape row counts into slot `wide`), the state BEHIND the tile -- in
	// front of a record's first tile every lane is in the start state, an eighth of the samples of 1 KiB records.  WHICH
	// lane: a hash of the wave's tile count -- `t & 63` sampled lane l at tile l of its string and nowhere else, and a batch
	// that repeats a base of a few thousand records (every benchmark here) was then seen at 2 048 places, over and over
	if (lane == (myTiles * 0x9E3779B1u) >> 26)
		WideSample<ZIP>(p, lds, W, st);
}

// Fixed-length records, 16-byte aligned, at least two 128-byte tiles per record (+ a tail shorter than a tile), whole tasks
// of 64 strings: tiled.hip's ring of two register tiles, chained through task boundaries when the tile count is even.
template <bool N16, bool ZIP>
__global__ __launch_bounds__(1024, 4) void ScanWideKernel(ScanParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const WideLayout W = MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, ZIP ? p.zipFull : 0);
	const WideConst K = MakeWideConst(p, W);
	LdsLayout L = {};           // what Finish() looks at: the block-local counters
	L.countsOff = W.countsOff;

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t ntasks = p.n / 64;
	const uint32_t ntiles = uint32_t(p.len / 128);   // >= 2 (the dispatcher); a tail shorter than a tile: exact steps below
	const uint32_t lastTile = ntiles - 1;
	const uint32_t paired = ntiles & ~1u;            // tiles walked two by two out of the ring; an odd last one after them
	// With an even tile count every task starts in slot a, so the ring runs straight through task boundaries (tiled.hip).
	const bool chain = (ntiles & 1u) == 0;
	const uint32_t voff = (lane & ~7u) * uint32_t(p.stride) + (lane & 7u) * 16;
	const uint64_t istride = p.stride;
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);
	uint32_t* prog = reinterpret_cast<uint32_t*>(lds + W.progOff);
	uint32_t myTiles = 0;
	// tasks go round the blocks before they go round a block's waves: a batch with fewer tasks than wave slots then puts a few
	// waves on every CU instead of sixteen on some (2^18 strings, two per lane: 2.1 TB/s on half the CUs)
	const uint64_t taskStep = uint64_t(gridDim.x) * 16;
	const uint64_t firstTask = uint64_t(wave) * gridDim.x + blockIdx.x;
	bool primed = firstTask < ntasks;
	if (primed)   // the first tile is on its way while the table is copied
		WideIssueTile(a, voff, Uniform64(reinterpret_cast<uint64_t>(p.text) + firstTask * 64 * p.stride), istride);
	LoadWideToLds(p, lds, W);
	for (uint64_t task = firstTask; task < ntasks; task += taskStep) {
		const uint64_t s0 = task * 64;
		const uint64_t s = s0 + lane;
		const uint64_t rowBase = Uniform64(reinterpret_cast<uint64_t>(p.text) + s0 * p.stride);
		const bool hasNext = chain && task + taskStep < ntasks;
		const uint64_t chainBase = hasNext ? Uniform64(reinterpret_cast<uint64_t>(p.text) + (s0 + taskStep * 64) * p.stride)
		                                   : rowBase + uint64_t(lastTile) * 128;
		uint32_t cold = StartState(p, s);
		uint32_t st = cold < p.wide ? cold : p.wide;   // the walk's state: a device id with a row, or `wide` = the escape row
		bool done = false;
		if (!primed)
			WideIssueTile(a, voff, rowBase, istride);
		for (uint32_t t = 0; t < paired && !done; t += 2) {
			WidePhase<N16, ZIP>(p, lds, W, K, rowBase, chainBase, voff, istride, lane, t, lastTile, a, b, st, cold, prog, myTiles);
			WidePhase<N16, ZIP>(p, lds, W, K, rowBase, chainBase, voff, istride, lane, t + 1, lastTile, b, a, st, cold, prog, myTiles);
			// wave-wide early out (multi.h:955-958): every lane in a row whose every transition is a self loop
			done = __all((WideFlags<ZIP>(st, K) & kAbsorbing) != 0);
			// An early-out leaves the tile the last phase asked for on its way into slot a.  It is waited for HERE, inside the
			// loop, where slot a still is the registers the load was issued into: round 5 waited behind the loop, hipcc copied
			// the slot on the loop's exit edge, and the load landed in registers that by then held something else -- a few
			// lanes of the wave's NEXT task walked stale text (found in round 6 by a corpus whose records nearly all reach the
			// absorbing state: 557 of 2^20 strings wrong; tests/test_wide.py test_early_out_between_chained_tasks).
#if !defined(PIRE_EXP) || PIRE_EXP != 2
			if (done)
				WideWaitTile<0>(a);
#endif
		}
		primed = hasNext && !done;   // an early-out left some other tile in slot a: re-prime then
#if defined(PIRE_EXP) && PIRE_EXP == 2   // (round 5's form, kept for the regression test's own test: make exp N=2)
		if (done)
			WideWaitTile<0>(a);
#endif
		if (!done && !chain) {
			// the odd last tile (ntiles >= 3 here): requested into slot a by the last phase of the loop, walked with nothing
			// on its way behind it
			WideWaitTile<0>(a);
			TransposeTile(a, lane);
#pragma unroll
			for (int k = 0; k < 8; ++k)
				WideChunk<N16, ZIP>(p, lds, W, K, a[k], st, cold, (lastTile * 8 + k) & 63);
		}
		uint32_t end = st < p.wide ? st : cold;
		if (!done) {   // the tail shorter than a tile: exact steps straight from memory
			const uint8_t* base = p.text + s * p.stride;
			for (uint64_t i = uint64_t(ntiles) * 128; i < p.len; ++i)
				end = WideNext<N16>(p, end, uint32_t(lds[base[i]]) >> 1);
		}
		Finish(p, lds, L, s, true, end);
	}
	WideWaitTile<0>(a);
	WideWaitTile<0>(b);
	// (for the build's audit, which follows every way out of the task loop until all loads are waited for: nothing is on its way
	// here -- the loops above wait for what they request before they are left -- and this says so where the walker can see it)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	FlushWide(p, lds, W);
}

// ---- two strings per lane: working sets beyond the rows (wide_common.h WideChunk2) --------------------------------------
// A task = 128 strings, lane l walks strings l and l + 64 of it.  Both tiles live in the 64 tile registers the ring of
// the kernel above uses for one string's two tiles, so there is no tile on its way during the walk: load, wait, walk
// (the walk is 10 x the load here; and a load on its way would be waited for by the first vmcnt(0) of a re-walk anyway).
template <bool N16, bool ZIP>
__global__ __launch_bounds__(1024, 4) void ScanWide2Kernel(ScanParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const WideLayout W = MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, ZIP ? p.zipFull : 0);
	const WideConst K = MakeWideConst(p, W);
	LdsLayout L = {};
	L.countsOff = W.countsOff;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t ntasks = p.n / 128;
	const uint32_t ntiles = uint32_t(p.len / 128);
	const uint32_t voff = (lane & ~7u) * uint32_t(p.stride) + (lane & 7u) * 16;
	const uint64_t istride = p.stride;
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);
	uint32_t* prog = reinterpret_cast<uint32_t*>(lds + W.progOff);
	uint32_t myTiles = 0;
	uint32_t direct = 0;   // wave-uniform: the last chunk left the rows, the next ones skip the attempt on the rows alone (WideChunk2)
	// tasks go round the blocks before they go round a block's waves: a batch with fewer tasks than wave slots then puts a few
	// waves on every CU instead of sixteen on some (2^18 strings, two per lane: 2.1 TB/s on half the CUs)
	const uint64_t taskStep = uint64_t(gridDim.x) * 16;
	const uint64_t firstTask = uint64_t(wave) * gridDim.x + blockIdx.x;
	LoadWideToLds(p, lds, W);
	for (uint64_t task = firstTask; task < ntasks; task += taskStep) {
		const uint64_t sA = task * 128 + lane, sB = sA + 64;
		const uint64_t baseA = Uniform64(reinterpret_cast<uint64_t>(p.text) + task * 128 * p.stride);
		const uint64_t baseB = baseA + 64 * p.stride;
		uint32_t colda = StartState(p, sA), coldb = StartState(p, sB);
		uint32_t sa = colda < p.wide ? colda : p.wide, sb = coldb < p.wide ? coldb : p.wide;
		bool done = false;
		for (uint32_t t = 0; t < ntiles && !done; ++t) {
			{   // the waves of a block kept in step (tiled.hip, EQ)
				uint32_t sum = 0;
				if (lane == 0)
					sum = atomicAdd(prog, 1u) + 1;
				sum = uint32_t(__builtin_amdgcn_readfirstlane(int(sum)));
				const uint32_t mine = ++myTiles;
				constexpr uint32_t margin = 4;
				if (mine * (blockDim.x >> 6) > sum + margin)
					__builtin_amdgcn_s_setprio(0);
				else if (mine * (blockDim.x >> 6) + margin < sum)
					__builtin_amdgcn_s_setprio(3);
				else
					__builtin_amdgcn_s_setprio(1);
			}
			WideIssueTile(a, voff, baseA + uint64_t(t) * 128, istride);
			WideIssueTile(b, voff, baseB + uint64_t(t) * 128, istride);
			WideWaitTile<0>(a);
			WideWaitTile<0>(b);
			TransposeTile(a, lane);
			TransposeTile(b, lane);
			// (written out: with the zipped step's larger body hipcc left the loop rolled and both tiles in scratch)
			WideChunk2<N16, ZIP>(p, lds, W, K, a[0], b[0], sa, sb, colda, coldb, (t * 8 + 0) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[1], b[1], sa, sb, colda, coldb, (t * 8 + 1) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[2], b[2], sa, sb, colda, coldb, (t * 8 + 2) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[3], b[3], sa, sb, colda, coldb, (t * 8 + 3) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[4], b[4], sa, sb, colda, coldb, (t * 8 + 4) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[5], b[5], sa, sb, colda, coldb, (t * 8 + 5) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[6], b[6], sa, sb, colda, coldb, (t * 8 + 6) & 63, direct);
			WideChunk2<N16, ZIP>(p, lds, W, K, a[7], b[7], sa, sb, colda, coldb, (t * 8 + 7) & 63, direct);
			if (lane == (myTiles * 0x9E3779B1u) >> 26) {   // visit samples: one lane per wave per tile, behind it (WidePhase)
				WideSample<ZIP>(p, lds, W, sa);
				WideSample<ZIP>(p, lds, W, sb);
			}
			if (t & 1)   // wave-wide early out (multi.h:955-958), every other tile
				done = __all(((WideFlags<ZIP>(sa, K) & WideFlags<ZIP>(sb, K)) & kAbsorbing) != 0);
		}
		uint32_t enda = sa < p.wide ? sa : colda, endb = sb < p.wide ? sb : coldb;
		if (!done) {   // the tail shorter than a tile: exact steps straight from memory
			const uint8_t* ta = p.text + sA * p.stride;
			const uint8_t* tb = p.text + sB * p.stride;
			for (uint64_t i = uint64_t(ntiles) * 128; i < p.len; ++i) {
				enda = WideNext<N16>(p, enda, uint32_t(lds[ta[i]]) >> 1);
				endb = WideNext<N16>(p, endb, uint32_t(lds[tb[i]]) >> 1);
			}
		}
		Finish(p, lds, L, sA, true, enda);
		Finish(p, lds, L, sB, true, endb);
	}
	// (for the build's audit, which follows every way out of the task loop until all loads are waited for: nothing is on its way
	// here -- the loops above wait for what they request before they are left -- and this says so where the walker can see it)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	FlushWide(p, lds, W);
}

// ------------------------------------------------------------------------------------------ launcher

// The choice between the dense rows and the wide walk (any choice is correct).  pire_hip_config.walk_variant: 0 by the
// share of the ranking's mass outside the dense rows -- measured by the scans so far once the table has adapted, the
// byte model's estimate before (which only a table far outside the dense rows acts on) --, 1 never, 2 whenever the
// table has states beyond the dense rows.
bool WideWanted(const ScanParams& p, const pire_hip_config& cfg)
{
	if (!p.wide || !p.wideRows || cfg.checked || cfg.tiled_variant != 0 || cfg.walk_variant == 1)
		return false;
	if (p.initIdx && (p.flags & kPermIds))
		return false;   // the segmented scan's passes stay on the kernels they were measured on
	if (p.outCounts && p.regexps > kMaxLdsCountRegexps)
		return false;
	if (cfg.walk_variant >= 2)
		return true;
	// Measured (profiles/r05_wide_curve.jsonl): with 0.1-0.2 % of the steps outside the dense rows 40 % of all wave-chunks
	// hold a lane that left them and the dense walk runs at 1.1-1.8 TB/s, the wide walk at 3.1; with nothing outside them
	// the dense walk's 6.5 TB/s against 4.3.  The wide walk wins from a few hundredths of a percent.
	return p.outsideDense > (p.massMeasured ? 0.0005f : 0.05f);
}

int LaunchWide(const ScanParams& p, hipStream_t stream)
{
	if (int rc = CheckCounts(p))
		return rc;
	ScanParams q = p;
	q.n = p.n & ~uint64_t(63);   // whole 64-string tasks; the remainder goes to the generic kernel below
	const WideLayout W = MakeWideLayout(p.wide, p.letters, p.outCounts ? p.regexps : 0, p.zipFull);
	int rc;
	// Two forms (same results): one string per lane and a ring of two tiles, or two strings per lane (ScanWide2Kernel).
	// A wave of the second walks its two strings in the time a wave of the first walks one and one more (3.6 against 4.0
	// TB/s with 2^18 strings, half the waves), so it is for batches
	// that give all 16 waves of every CU a task of 128 strings: there it is as fast where the working set fits the rows
	// (profiles/r05_wide_curve.jsonl: 4.29 against 4.27 TB/s) and faster beyond them -- 1.89 against 1.11 with 3 % of the
	// steps outside the rows, 1.36 against 1.04 with 12 %: the loads of the walk beyond the rows are what the time goes into
	// there, and two chains per lane have two of them on their way; 1.09 against 0.97 with 17 %, the same (0.60-0.65) from 29 % on.
	// Smaller batches: one string per lane, more waves (2^18 strings, 3 % outside the rows: 1.09 against 1.05,
	// r05m_wide_small_batches.txt).  walk_variant 2 / 3 force one; ScanParams::forceLanes (the first-use self-test) another.
	const pire_hip_config cfg = GetConfig();
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const bool two = p.forceLanes ? p.forceLanes == 2 : cfg.walk_variant == 3 || (cfg.walk_variant != 2 && p.n >= uint64_t(cus) * 16 * 128);
	if (two)
		q.n = p.n & ~uint64_t(127);   // whole 128-string tasks
	if (p.wideLaunched)
		p.wideLaunched->fetch_add(q.n / 64 * (p.len / 16), std::memory_order_relaxed);
	if (q.n == 0) {
		rc = PIRE_HIP_OK;
	} else if (p.zipFull) {   // (a zipped image implies the u16 table: table.cpp ChooseZip)
		NoteKernel("wide", two ? "pirehip::ScanWide2Kernel<u16 table, zipped rows>" : "pirehip::ScanWideKernel<u16 table, zipped rows>");
		rc = two ? LaunchScan(ScanWide2Kernel<true, true>, q, 1024, W.total, stream, 1) : LaunchScan(ScanWideKernel<true, true>, q, 1024, W.total, stream, 1);
	} else if (p.next16) {
		NoteKernel("wide", two ? "pirehip::ScanWide2Kernel<u16 table>" : "pirehip::ScanWideKernel<u16 table>");
		rc = two ? LaunchScan(ScanWide2Kernel<true, false>, q, 1024, W.total, stream, 1) : LaunchScan(ScanWideKernel<true, false>, q, 1024, W.total, stream, 1);
	} else {
		NoteKernel("wide", two ? "pirehip::ScanWide2Kernel<u32 table>" : "pirehip::ScanWideKernel<u32 table>");
		rc = two ? LaunchScan(ScanWide2Kernel<false, false>, q, 1024, W.total, stream, 1) : LaunchScan(ScanWideKernel<false, false>, q, 1024, W.total, stream, 1);
	}
	if (rc != PIRE_HIP_OK || q.n == p.n)
		return rc;
	ScanParams tail = p;
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

}  // namespace pirehip
