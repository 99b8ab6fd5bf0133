// The tiled kernel (fixed-length records): the benchmark kernel of the path.  DESIGN.md section 4.3.

#include "device_common.h"

namespace pirehip {

// ------------------------------------------------------------------------------------------ tiled kernel
// Fixed-length records, 16-byte aligned.  A wave streams 128-byte tiles (one cache line of each of its 64 strings)
// straight from HBM into VGPRs with whole-line loads, transposes them in registers so that every lane owns the
// bytes of its own string, and walks them with one LDS gather per byte.
// No LDS staging: all of the LDS is left for the table (profiles/micro_loadpath_r01.log: the register path streams
// as fast as a fully coalesced read).

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

// (Butterfly4 / ButterflyQuad4 / TransposeTile: device_common.h, shared with the ragged kernel.)

// The hot rows sit at LDS byte address 0 (the kernels declare no static __shared__, so the dynamic region
// starts at 0): the v_perm result IS the ds_read address, with no base add in the dependent chain.

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


// EQ: keep the waves of a block in step.  Every tile a wave adds 1 to a block-wide progress counter (LDS) and compares
// 16 x its own tile count with the sum: a wave ahead of the block's average by more than a quarter tile drops its issue
// priority, one behind raises it.  Without it the 16 waves of a block, which all do exactly the same amount of work,
// finish up to 30 us apart (profiles/r02_tiled_block_stamps.log: age-ordered arbitration lets some waves run ahead all
// the way) and the CU idles half empty at the end of a launch; with it 10 us.  Worth 1.3 % on the 2^20 x 4 KiB headline
// (9 of 9 alternating pairs), 4.8 % with the single-pattern table, nothing from 8 tasks per wave up, -0.7 % on C++ text
// (profiles/r02_tiled_equalise.log).
template <int NBUF, bool NT, int ROT, bool EQ = false>
__device__ __forceinline__ void Phase(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint64_t rowBase,
                                      uint64_t chainBase, uint32_t voff, uint64_t istride, uint32_t lane, uint32_t t,
                                      uint32_t lastTile, u32x4 (&cur)[8], u32x4 (&refill)[8], uint32_t& hs, uint32_t& cold,
                                      uint32_t* prog = nullptr, uint32_t* myTiles = nullptr)
{
	if (EQ) {
		uint32_t sum = 0;
		if (lane == 0)
			sum = atomicAdd(prog, 1u) + 1;
		sum = uint32_t(__builtin_amdgcn_readfirstlane(int(sum)));
		const uint32_t mine = ++*myTiles;
		constexpr uint32_t margin = 4;   // in sixteenths of a tile (0 / 4 / 8: equal; 16, 32: less effect)
		if (mine * (blockDim.x >> 6) > sum + margin)
			__builtin_amdgcn_s_setprio(0);
		else if (mine * (blockDim.x >> 6) + margin < sum)
			__builtin_amdgcn_s_setprio(3);
		else
			__builtin_amdgcn_s_setprio(1);
	}
	// Refill target: the next tile of this task, or -- on the task's last tile -- tile 0 of the wave's NEXT task
	// (chainBase; equals this task's last tile when there is nothing to chain to), so that neither the HBM
	// latency of a task's first tile nor a duplicate load of its last tile is ever paid.
	const uint64_t ahead = t < lastTile ? rowBase + uint64_t(t + 1) * 128 : chainBase;
	if (!(p.flags & kDebugNoRefill))   // measurement knob only (PIRE_HIP_DEBUG_NOLOAD): walk stale registers
		IssueTile<NT>(refill, voff, ahead, istride);
	WaitTile<NBUF - 1>(cur);
	if (!(p.flags & kDebugNoTranspose))
		TransposeTile(cur, lane);
	if (lane == (t & 63) && !(p.flags & kDebugNoHist))   // visit sample: one lane per wave per tile, rotating
		atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + hs, 1u);
	if (p.flags & kDebugNoStep) {      // measurement knob only (PIRE_HIP_DEBUG_NOSTEP): stream + transpose, no walk
		hs ^= (cur[0].x ^ cur[7].w) & 1;
		return;
	}
	StepTile<ROT>(p, lds, L, cur, hs, cold, t);
}

// The same phase with the transpose of the NEXT tile hidden in the walk of this one (round 3, SHADOW).  Invariant: `cur`
// holds tile t already transposed.  The refill slot `next` (walked one phase ago) is requested first; the walk of
// chunks 0..5 and half of chunk 6 gives it 6.5 / 8 of a tile-time to land; then one transpose slot (four independent
// v_cndmask_b32_dpp, device_common.h TransposeSlot) follows each of the last 24 lookups, in the shadow of the LDS round
// trip the wave would sit out anyway.  At the end `next` is transposed and the roles swap.
template <bool NT, bool EQ>
__device__ __forceinline__ void PhaseShadow(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint64_t rowBase,
                                            uint64_t chainBase, uint32_t voff, uint64_t istride, uint32_t lane, uint32_t t,
                                            uint32_t lastTile, u32x4 (&cur)[8], u32x4 (&next)[8], uint32_t& hs, uint32_t& cold,
                                            uint32_t* prog, uint32_t* myTiles)
{
	if (EQ) {
		uint32_t sum = 0;
		if (lane == 0)
			sum = atomicAdd(prog, 1u) + 1;
		sum = uint32_t(__builtin_amdgcn_readfirstlane(int(sum)));
		const uint32_t mine = ++*myTiles;
		constexpr uint32_t margin = 4;   // in sixteenths of a tile
		if (mine * (blockDim.x >> 6) > sum + margin)
			__builtin_amdgcn_s_setprio(0);
		else if (mine * (blockDim.x >> 6) + margin < sum)
			__builtin_amdgcn_s_setprio(3);
		else
			__builtin_amdgcn_s_setprio(1);
	}
	const uint64_t ahead = t < lastTile ? rowBase + uint64_t(t + 1) * 128 : chainBase;
	IssueTile<NT>(next, voff, ahead, istride);
	if (lane == (t & 63))   // visit sample: one lane per wave per tile, rotating
		atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + hs, 1u);
	uint32_t tmp[4];
#pragma unroll
	for (int k = 0; k < 6; ++k)
		StepChunk<0>(p, lds, L, cur[k], hs, cold, (t * 8 + k) & 63);
	StepChunkShadow<0, 8, true>(p, lds, L, cur[6], hs, cold, (t * 8 + 6) & 63, next, tmp);
	StepChunkShadow<8, 0, false>(p, lds, L, cur[7], hs, cold, (t * 8 + 7) & 63, next, tmp);
}

// Fixed-length records, 16-byte aligned, whole tasks of 64 strings (the host routes the < 64-string remainder to
// the generic kernel).  NBUF register tiles per wave form a ring: tile t is walked out of registers -- one LDS
// gather per byte -- while tiles t+1 .. t+NBUF-1 stream in from HBM.
template <int WAVES, int NBUF, bool NT, int MINW, int ROT, bool CHECKED = false, bool EQ = false, bool SHADOW = false>
__global__ __launch_bounds__(WAVES * 64, MINW) void ScanTiledKernel(ScanParams p)
{
	static_assert(!SHADOW || (ROT == 0 && !CHECKED), "the shadowed transpose exists for the shipped layout only");
	static_assert(ROT != 2 || !CHECKED, "rotated columns: shipped form only");
	// Depth 2 only: with three slots hipcc (ROCm 7.2) spills tile registers to scratch WHILE their loads are in
	// flight (profiles/ + DESIGN.md section 6) -- silently wrong data.  tests/test_build_audit.py pins "no scratch".
	static_assert(NBUF == 2, "ring depth");
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, ROT == 1 ? kRotPitch : 256u, CompactBytes(p));
	L.rot2 = ROT == 2 ? 1u : 0u;   // the launcher handed over the rows with rotated columns (ScanParams::hotRowsRot)

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t ntasks = p.n / 64;                // whole tasks only
	const uint32_t ntiles = uint32_t(p.len / 128);   // >= 1 (TiledEligible)
	const uint32_t lastTile = ntiles - 1;
	const uint32_t groups = ntiles / NBUF;
	const uint32_t rem = ntiles % NBUF;
	// per-lane byte offset inside a task's tile: string (lane & ~7) [+ j per instruction], chunk (lane & 7)
	const uint32_t voff = (lane & ~7u) * uint32_t(p.stride) + (lane & 7u) * 16;
	// (Tried: instruction j covering the 8 CONSECUTIVE strings 8j .. 8j+7 -- same transpose, lane 8g+j then owns string
	// 8j+g.  Identical times, profiles/r02_tiled_trmap.log.  What does matter is that a CU's 16 waves work on ADJACENT
	// tasks: wave-major task numbering costs 14 %, profiles/r02_tiled_taskmap.log -- lines of strings 64 KiB apart share
	// DRAM rows and the waves of one CU stay in phase.)
	const uint64_t istride = p.stride;
	const uint32_t mine = lane;   // this lane's string inside a task

	// Ring slots: tile t lives in slot t % 2.
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);
	uint32_t* prog = reinterpret_cast<uint32_t*>(lds + L.progOff);   // zeroed with the visit samples by LoadTableToLds
	uint32_t myTiles = 0;

	// With an even tile count every task starts in slot a, so the ring can run straight through task boundaries.
	const bool chain = rem == 0;
	const uint64_t taskStep = uint64_t(gridDim.x) * WAVES;
	const uint64_t firstTask = (p.flags & kSpreadTasks) ? uint64_t(wave) * gridDim.x + blockIdx.x   // (device_common.h)
	                                                     : uint64_t(blockIdx.x) * WAVES + wave;
	// The first tile of the wave's first task is requested BEFORE the table is copied into LDS: its HBM latency (a
	// few microseconds when all 4 096 waves of a launch ask at once) then hides behind the copy.
	bool primed = firstTask < ntasks;   // slot a already holds (or is receiving) tile 0 of the task about to start
	bool ready = false;                 // SHADOW: ... and it is transposed
#ifdef PIRE_HIP_TUNING
	if (p.stamps && threadIdx.x == 0)
		p.stamps[blockIdx.x * 4 + 0] = wall_clock64();
#endif
	if (primed)
		IssueTile<NT>(a, voff, Uniform64(reinterpret_cast<uint64_t>(p.text) + firstTask * 64 * p.stride), istride);
	LoadTableToLds(p, lds, L);
#ifdef PIRE_HIP_TUNING
	if (p.stamps && threadIdx.x == 0)
		p.stamps[blockIdx.x * 4 + 1] = wall_clock64();
#endif
	for (uint64_t task = firstTask; task < ntasks; task += taskStep) {
		const uint64_t s0 = task * 64;
		const uint64_t s = s0 + mine;
		const uint64_t rowBase = Uniform64(reinterpret_cast<uint64_t>(p.text) + s0 * p.stride);
		const bool hasNext = chain && task + taskStep < ntasks;
		const uint64_t chainBase = hasNext ? Uniform64(reinterpret_cast<uint64_t>(p.text) + (s0 + taskStep * 64) * p.stride)
		                                   : rowBase + uint64_t(lastTile) * 128;

		uint32_t cold = StartState(p, s);
		uint32_t hs = cold < p.hot ? cold : p.hot;

		bool done = false;
		// CHECKED (PIRE_HIP_CHECKED=1, the analogue of the reference's ValidateSkip, multi.h:925-934): the early-out is
		// only NOTED; the rest of the task is walked all the same, and a lane whose state moved after the wave was
		// declared absorbing is counted (pire_hip_table_check_failures).
		bool noted = false;
		uint32_t hsNoted = 0;
		if (!primed)
			IssueTile<NT>(a, voff, rowBase, istride);
		if (SHADOW && !ready) {
			// the ring's invariant -- slot a holds tile 0 TRANSPOSED -- does not hold yet (a wave's first task, or
			// the task after an early-out): this one transpose is not hidden
			WaitTile<0>(a);
			TransposeTile(a, lane);
		}
		for (uint32_t g = 0; g < groups && !done; ++g) {
			const uint32_t t = g * 2;
			if (SHADOW) {
				PhaseShadow<NT, EQ>(p, lds, L, rowBase, chainBase, voff, istride, lane, t, lastTile, a, b, hs, cold, prog, &myTiles);
				PhaseShadow<NT, EQ>(p, lds, L, rowBase, chainBase, voff, istride, lane, t + 1, lastTile, b, a, hs, cold, prog, &myTiles);
			} else {
				Phase<2, NT, ROT, EQ>(p, lds, L, rowBase, chainBase, voff, istride, lane, t, lastTile, a, b, hs, cold, prog, &myTiles);
				Phase<2, NT, ROT, EQ>(p, lds, L, rowBase, chainBase, voff, istride, lane, t + 1, lastTile, b, a, hs, cold, prog, &myTiles);
			}
			done = AllAbsorbing(p, lds, L, hs);
			if (CHECKED && done) {
				if (!noted)
					hsNoted = hs;
				noted = true;
				done = false;
			}
		}
		primed = hasNext && !done;   // an early-out leaves some other tile in slot a: re-prime then
		ready = primed;              // SHADOW: ... and when it is the right tile, it is transposed already
		if (!done && rem == 1) {
			if (!SHADOW) {   // SHADOW: the last phase of the loop (or, without one, the prologue) left it transposed
				WaitTile<0>(a);
				TransposeTile(a, lane);
			}
			StepTile<ROT>(p, lds, L, a, hs, cold, lastTile);
		}

		if (CHECKED && noted && hs != hsNoted)
			atomicAdd(&p.visitHot[kCheckSlot], 1u);
		uint32_t st = hs != p.hot ? hs : cold;
		// tail shorter than a tile: exact steps straight from memory
		if (!done) {
			const uint8_t* base = p.text + s * p.stride;
			for (uint64_t i = uint64_t(ntiles) * 128; i < p.len; ++i)
				st = SlowStep(p, lds, L, st, base[i]);
		}
		Finish(p, lds, L, s, true, st);
	}
#ifdef PIRE_HIP_TUNING
	if (p.stamps && lane == 0)
		atomicMax(&p.stamps[blockIdx.x * 4 + 2], wall_clock64());   // the block's last wave out of the walk
	if (p.stamps && lane == 0)
		atomicMin(&p.stamps[blockIdx.x * 4 + 3], wall_clock64());   // ... and its first
#endif
	FlushCounts(p, lds, L);
}


// ------------------------------------------------------------------------------------------ segmented scan, one mode
// The tiled kernel for the segmented scan's grid segments (segmented.hip) with the WARM-UP INSIDE THE PASS, as the pair
// kernel has it for two modes (pair.hip, SEG): a lane starts `warmTiles` 128-byte tiles before its segment in the
// mode's representative state (mode 0: the string's own start state), writes the state it is in at the segment's first
// byte to `guess` (device ids; what the chain compares) and walks on; a string's first segment (segJ == 0) keeps its
// start state, and the one record whose warm-up would lie in front of the text reads its own bytes for those tiles.
// One launch instead of a ragged warm-up batch plus the tiled pass: what a table with ONE mode -- e.g. the reference's
// benchmark corpus as one string -- still paid in round 3's first version of the fused scan.
// A kernel of its own (a trimmed copy of ScanTiledKernel's loop: shipped variant only, even tile counts only) so that
// the headline kernel's code is what it was.
struct TiledSegParams {
	ScanParams p;
	uint32_t warmTiles;
	const uint32_t* segJ;
	uint32_t* guess;
};

__device__ __forceinline__ void IssueTileLow(u32x4 (&r)[8], uint32_t voff, uint64_t tileBase, uint64_t stride, uint64_t low)
{
	// only the warm-up tiles of the very first record (the first load's lanes 0..7) lie below `low`, the first byte of
	// the text: those lanes read from the record itself instead, and what they read is never used
	uint32_t voff0 = voff;
	if (tileBase < low)
		voff0 += (threadIdx.x & 63) < 8 ? uint32_t(low - tileBase + 127) & ~127u : 0u;
	const uint64_t b0 = tileBase, b1 = b0 + stride, b2 = b1 + stride, b3 = b2 + stride, b4 = b3 + stride,
	               b5 = b4 + stride, b6 = b5 + stride, b7 = b6 + stride;
	asm volatile(
		"global_load_dwordx4 %0, %17, %9 nt\n\t"
		"global_load_dwordx4 %1, %8, %10 nt\n\t"
		"global_load_dwordx4 %2, %8, %11 nt\n\t"
		"global_load_dwordx4 %3, %8, %12 nt\n\t"
		"global_load_dwordx4 %4, %8, %13 nt\n\t"
		"global_load_dwordx4 %5, %8, %14 nt\n\t"
		"global_load_dwordx4 %6, %8, %15 nt\n\t"
		"global_load_dwordx4 %7, %8, %16 nt"
		: "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
		: "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(b6), "s"(b7), "v"(voff0));
}

__device__ __forceinline__ void PhaseSeg(const ScanParams& p, const uint8_t* lds, const LdsLayout& L, uint64_t rowBase,
                                         uint64_t chainBase, uint32_t voff, uint64_t istride, uint64_t low, uint32_t lane,
                                         uint32_t t, uint32_t lastTile, u32x4 (&cur)[8], u32x4 (&refill)[8], uint32_t& hs,
                                         uint32_t& cold, uint32_t* prog, uint32_t& myTiles)
{
	{   // the waves of a block kept in step (EQ, above)
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
	IssueTileLow(refill, voff, ahead, istride, low);
	WaitTile<1>(cur);
	TransposeTile(cur, lane);
	if (lane == (t & 63))   // visit sample: one lane per wave per tile, rotating
		atomicAdd(reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(lds) + L.histOff) + hs, 1u);
	StepTile<0>(p, lds, L, cur, hs, cold, t);
}

__global__ __launch_bounds__(1024, 4) void ScanTiledSegKernel(TiledSegParams q)
{
	const ScanParams& p = q.p;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, 256u, CompactBytes(p));
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t ntasks = p.n / 64;                          // whole tasks only
	const uint32_t warm = q.warmTiles;                         // even (the launcher checks)
	const uint32_t ntiles = uint32_t(p.len / 128) + warm;      // even (the launcher checks)
	const uint32_t lastTile = ntiles - 1;
	const uint32_t voff = (lane & ~7u) * uint32_t(p.stride) + (lane & 7u) * 16;
	const uint64_t istride = p.stride;
	const uint64_t low = reinterpret_cast<uint64_t>(p.text);
	const uint64_t text = low - uint64_t(warm) * 128;
	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);
	uint32_t* prog = reinterpret_cast<uint32_t*>(lds + L.progOff);
	uint32_t myTiles = 0;
	const uint32_t wavesPerBlock = blockDim.x >> 6;   // 16, or fewer when the batch has fewer tasks than 16 per CU
	const uint64_t taskStep = uint64_t(gridDim.x) * wavesPerBlock;
	const uint64_t firstTask = uint64_t(blockIdx.x) * wavesPerBlock + wave;
	bool primed = firstTask < ntasks;
	if (primed)
		IssueTileLow(a, voff, Uniform64(text + firstTask * 64 * p.stride), istride, low);
	LoadTableToLds(p, lds, L);
	for (uint64_t task = firstTask; task < ntasks; task += taskStep) {
		const uint64_t s0 = task * 64;
		const uint64_t s = s0 + lane;
		const uint64_t rowBase = Uniform64(text + s0 * p.stride);
		const bool hasNext = task + taskStep < ntasks;
		const uint64_t chainBase = hasNext ? Uniform64(text + (s0 + taskStep * 64) * p.stride) : rowBase + uint64_t(lastTile) * 128;
		uint32_t cold = StartState(p, s);   // the mode's representative (mode 0: the string's start state)
		uint32_t hs = cold < p.hot ? cold : p.hot;
		bool done = false;
		if (!primed)
			IssueTileLow(a, voff, rowBase, istride, low);
		for (uint32_t t = 0; t < ntiles && !done; t += 2) {
			if (t == warm) {   // the segment's first byte: this state is the guess
				uint32_t st = hs != p.hot ? hs : cold;
				if (q.segJ[s] == 0)
					st = StartState(p, s);
				q.guess[s] = st;
				cold = st;
				hs = st < p.hot ? st : p.hot;
			}
			PhaseSeg(p, lds, L, rowBase, chainBase, voff, istride, low, lane, t, lastTile, a, b, hs, cold, prog, myTiles);
			PhaseSeg(p, lds, L, rowBase, chainBase, voff, istride, low, lane, t + 1, lastTile, b, a, hs, cold, prog, myTiles);
			done = t >= warm && AllAbsorbing(p, lds, L, hs);
		}
		primed = hasNext && !done;   // an early-out leaves some other tile in slot a: re-prime then
		if (done)
			WaitTile<0>(a);
		uint32_t st = hs != p.hot ? hs : cold;
		if (!done) {   // tail shorter than a tile: exact steps straight from memory
			const uint8_t* base = p.text + s * p.stride;
			for (uint64_t i = uint64_t(ntiles - warm) * 128; i < p.len; ++i)
				st = SlowStep(p, lds, L, st, base[i]);
		}
		Finish(p, lds, L, s, true, st);
	}
	WaitTile<0>(a);
	WaitTile<0>(b);
	FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ launcher

bool TiledEligible(const ScanParams& p)
{
	return p.offsets == nullptr && p.n >= 64 && p.len >= 128 && (p.stride % 16) == 0 && p.stride * 64 < (1ull << 31) &&
	       (reinterpret_cast<uintptr_t>(p.text) % 16) == 0;
}

int LaunchTiled(const ScanParams& p, hipStream_t stream)
{
	if (int rc = CheckCounts(p))
		return rc;
	// Variant knob for A/B measurements (DESIGN.md section 5 ladder); every variant returns the same results.
	const pire_hip_config cfg = GetConfig();
	int variant = cfg.checked ? 4 : int(cfg.tiled_variant);
	// Tables whose traffic is spread over many states (no state carries 60 % of the steps: set_d 0.55 against set_a's 0.76,
	// set_b's 0.68, a single pattern's 0.87) are bound by LDS bank conflicts, not by the loads (PMC: the LDS is busy 98 % of
	// set_d's kernel, 5.6 cycles per lookup against 4.8, profiles/r04_set_d_pmc.txt).  Variant 23 gives them rows with
	// ROTATED columns -- bank = byte & 63 instead of (byte >> 2) & 63, simulated 5.9 -> 4.65 cycles -- at 3 more VALU per 4
	// bytes.  Measured (profiles/r04_rotated_columns_ab.log, two alternating rounds on one box): set_d 0.7367 / 0.7273 ms
	// against 0.7325 / 0.7240 with the plain rows, every other table 3-5 % slower -- the walk is as sensitive to VALU
	// issue as to the LDS, what round 1 found for set_a holds for the spread-out table too.  NOT selected by itself; kept
	// as an A/B variant (23).  With its lanes in ~11 different rows a half-wave's 32 lookups are 32 balls in 64 banks
	// (expected fullest bank ~3 -> ~6 cycles per wave lookup): set_d already sits at that floor of a 64-lane byte gather.
	if (variant == 24)
		variant = 0;
	ScanParams q = p;
	// a batch with fewer tasks than wave slots: a few waves on every CU instead of sixteen on some (kSpreadTasks)
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const int tpb = (p.n + 63) / 64 < uint64_t(cus) * 16 ? 1 : 0;
	if (tpb)
		q.flags |= kSpreadTasks;
#ifdef PIRE_HIP_TUNING
	q.stamps = nullptr;
	if (getenv("PIRE_HIP_DEBUG_NOLOAD"))
		q.flags |= kDebugNoRefill;
	if (getenv("PIRE_HIP_DEBUG_NOSTEP"))
		q.flags |= kDebugNoStep;
	if (getenv("PIRE_HIP_DEBUG_NOCOLDCOUNT"))
		q.flags |= kDebugNoColdCount;
	if (getenv("PIRE_HIP_DEBUG_NOHIST"))
		q.flags |= kDebugNoHist;
	if (getenv("PIRE_HIP_DEBUG_NOTRANSPOSE"))
		q.flags |= kDebugNoTranspose;
	if (getenv("PIRE_HIP_DEBUG_NOTRAP"))
		q.flags |= kDebugNoTrap;
	if (getenv("PIRE_HIP_DEBUG_TASKMAP"))
		q.flags |= kSpreadTasks;
	static unsigned long long* stampBuf = nullptr;
	const bool stamping = getenv("PIRE_HIP_DEBUG_STAMPS") != nullptr;
	if (stamping) {
		if (!stampBuf)
			(void)hipMalloc(reinterpret_cast<void**>(&stampBuf), 1024 * 4 * 8);
		std::vector<unsigned long long> init(1024 * 4, 0);
		for (int b = 0; b < 1024; ++b)
			init[b * 4 + 3] = ~0ull;
		(void)hipMemcpy(stampBuf, init.data(), init.size() * 8, hipMemcpyHostToDevice);
		q.stamps = stampBuf;
	}
#endif
	if (variant == 1)
		q.compact = 0;   // the compact rows hold LDS addresses of the 256-byte-pitch layout
	q.n = p.n & ~uint64_t(63);   // whole 64-string tasks; the remainder goes to the generic kernel below
	int rc;
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, kRotPitch, CompactBytes(q));
	const LdsLayout L256 = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(q));
	switch (variant) {
	case 1:
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,nt,5,rot>");
		rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 1>, q, 1024, L.total, stream, tpb);   // bank-rotated rows
		break;
	case 2:
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,plain,5>");
		rc = LaunchScan(ScanTiledKernel<16, 2, false, 5, 0>, q, 1024, L256.total, stream, tpb);   // no nt
		break;
	case 20:
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,nt,5,free-running>");
		rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 0>, q, 1024, L256.total, stream, tpb);   // waves not kept in step
		break;
	case 4:   // also chosen by PIRE_HIP_CHECKED=1
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,nt,5,checked>");
		rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 0, true>, q, 1024, L256.total, stream, tpb);
		break;
	case 23:
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,nt,5,rotated columns>");
		q.hotRows = p.hotRowsRot ? p.hotRowsRot : p.hotRows;
		if (!p.hotRowsRot) {
			rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 0, false, true>, q, 1024, L256.total, stream, tpb);
			break;
		}
		rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 2, false, true>, q, 1024, L256.total, stream, tpb);
		break;
	case 22:
		// The transpose of tile t+1 in the shadows of the last 24 lookups of tile t (PhaseShadow).  Measured in round 3
		// (profiles/r03_shadow_ab.log, one process, alternating bursts): +1.5-2 % while the clocks are low (the first
		// ~25 launches after an idle gap), -0.8 % at settled clocks, where the kernel sits 1.5 % above its load path
		// and the earlier deadline for the refill (6.5 / 8 of a tile-time) costs more than the hidden VALU saves.
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,nt,4,shadow>");
		rc = LaunchScan(ScanTiledKernel<16, 2, true, 4, 0, false, true, true>, q, 1024, L256.total, stream, tpb);
		break;
	default:
		NoteKernel("tiled", "pirehip::ScanTiledKernel<16,2,nt,5>");
		rc = LaunchScan(ScanTiledKernel<16, 2, true, 5, 0, false, true>, q, 1024, L256.total, stream, tpb);
		break;
	}
#ifdef PIRE_HIP_TUNING
	if (stamping && rc == PIRE_HIP_OK) {
		(void)hipDeviceSynchronize();
		std::vector<unsigned long long> st(1024 * 4);
		(void)hipMemcpy(st.data(), stampBuf, st.size() * 8, hipMemcpyDeviceToHost);
		unsigned long long t0 = ~0ull, tEnd = 0;
		std::vector<double> loaded, firstOut, lastOut;
		for (int b = 0; b < 1024; ++b)
			if (st[b * 4 + 0]) {
				t0 = std::min(t0, st[b * 4 + 0]);
				tEnd = std::max(tEnd, st[b * 4 + 2]);
			}
		for (int b = 0; b < 1024; ++b)
			if (st[b * 4 + 0]) {   // wall_clock64 ticks at 100 MHz
				loaded.push_back((st[b * 4 + 1] - t0) / 100.0);
				firstOut.push_back((st[b * 4 + 3] - t0) / 100.0);
				lastOut.push_back((st[b * 4 + 2] - t0) / 100.0);
			}
		if (const char* dump = getenv("PIRE_HIP_DEBUG_STAMPS_DUMP")) {   // raw per-block stamps (us): block start loaded first-out last-out
			if (FILE* f = fopen(dump, "w")) {
				for (int b = 0; b < 1024; ++b)
					if (st[b * 4 + 0])
						fprintf(f, "%d %.2f %.2f %.2f %.2f\n", b, (st[b * 4 + 0] - t0) / 100.0, (st[b * 4 + 1] - t0) / 100.0,
						        (st[b * 4 + 3] - t0) / 100.0, (st[b * 4 + 2] - t0) / 100.0);
				fclose(f);
			}
		}
		auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[size_t(q * (v.size() - 1))]; };
		fprintf(stderr, "pire_hip stamps (us from the first block's start): table loaded min %.1f median %.1f max %.1f | first wave "
		        "of a block done min %.1f median %.1f max %.1f | last wave of a block done min %.1f median %.1f p90 %.1f max %.1f\n",
		        pct(loaded, 0), pct(loaded, 0.5), pct(loaded, 1), pct(firstOut, 0), pct(firstOut, 0.5), pct(firstOut, 1),
		        pct(lastOut, 0), pct(lastOut, 0.5), pct(lastOut, 0.9), pct(lastOut, 1));
	}
#endif
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



int LaunchTiledSeg(const ScanParams& p, uint64_t warmBytes, const uint32_t* segJ, uint32_t* guess, hipStream_t stream)
{
	if (warmBytes % 256 != 0 || warmBytes > p.stride || (p.len / 128) % 2 != 0 || p.len < 256 || !segJ || !guess || !TiledEligible(p)) {
		SetError("tiled segment kernel: whole pairs of 128-byte tiles, a warm-up no longer than the segment");
		return PIRE_HIP_EINVAL;
	}
	TiledSegParams q = {};
	q.p = p;
	q.p.n = p.n & ~uint64_t(63);   // whole 64-segment tasks; the caller runs the rest as batches of their own
	q.p.outCounts = nullptr;
	q.p.outFinal = nullptr;
	q.warmTiles = uint32_t(warmBytes / 128);
	q.segJ = segJ;
	q.guess = guess;
	if (q.p.n == 0)
		return PIRE_HIP_OK;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const LdsLayout L = MakeLayout(p.hot, 0, 256u, CompactBytes(p));
	hipError_t e = SetDynamicLds(reinterpret_cast<const void*>(ScanTiledSegKernel), L.total);
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	// one block per CU (the table fills its LDS); fewer than 16 tasks per CU: smaller blocks on more CUs
	const uint64_t ntasks = q.p.n / 64;
	// (at least 4 waves: LoadTableToLds hands its small pieces to the first 256 threads)
	const uint64_t waves = std::max<uint64_t>(4, std::min<uint64_t>(16, (ntasks + cus - 1) / cus));
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((ntasks + waves - 1) / waves, uint64_t(cus)));
	NoteKernel("tiled_seg", "pirehip::ScanTiledSegKernel");
	hipLaunchKernelGGL(ScanTiledSegKernel, dim3(unsigned(blocks)), dim3(unsigned(waves * 64)), L.total, stream, q);
	e = hipGetLastError();
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "tiled segment kernel launch");
}

}  // namespace pirehip
