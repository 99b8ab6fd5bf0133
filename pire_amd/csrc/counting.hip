// Pire::CountingScanner / Pire::AdvancedCountingScanner on the GPU: count, per regexp, the
// occurrences of `re` separated by `sep` -- the first scanners on this path whose Action is not a no-op.
//
// Reference: /root/reference/pire/extra/count.h, /root/reference/pire/scanners/loaded.h
//   LoadedScanner table: u8 letters[264], Transition{u32 shift, u32 action}[states*letters], u8 tags   loaded.h:57-66, 235-243
//   serialised form (Save)                                                                            scanner_io.cpp:172-189
//   Next: state += SignExtend(x.shift); return x.action                                               count.h:148-153
//   CountingScanner::TakeActionImpl: increment, then reset                                            count.h:251-257
//   AdvancedCountingScanner::TakeActionImpl: reset, then increment                                    count.h:287-295
//   PerformIncrement / PerformReset, IncrementPerformer / ResetPerformer                              count.h:48-101, 175-192
//   State::Result(i) = max(current[i], total[i])                                                      count.h:206
// Device form: transitions repacked as {next state index, action} (8 bytes), kept in LDS when the table fits
// (counting scanners are small: tens of states); one string per lane, exact step per byte; the per-regexp counters
// live in registers (8 or 16 pairs); an action word is applied only when it is non-zero.

#include <hip/hip_runtime.h>

#include <type_traits>
#include <unordered_map>

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "internal.h"
#include "selftest.h"
#include "device_common.h"

namespace pirehip {

constexpr uint32_t kMaxReCount = 16;   // LoadedScanner::MAX_RE_COUNT, loaded.h:74

struct CountingHost {
	uint32_t states = 0, letters = 0, regexps = 0, initial = 0;
	std::vector<uint8_t> letterOf;     // [264] m_letters
	std::vector<uint64_t> trans;       // [states*letters] next state index | action << 32
	uint32_t type = 4;                 // ScannerIOTypes: 4 LoadedScanner, 5 NoGlueLimitCountingScanner
	std::vector<uint8_t> tags;         // [states] m_tags (CapturingScanner: bit 0 = Final, capture.h:56, 134)
	std::vector<uint32_t> actions;     // type 5: [0] = length; per action: resets count, ids, increments count, ids
	// the dense form of CountingPackedKernel (BASIC / ADVANCED, <= 255 states, <= 255 distinct actions); empty otherwise
	std::vector<uint16_t> dense;       // [states][256] next state | action id << 8, indexed by the input BYTE
	std::vector<uint16_t> denseMarks;  // [states][2] the same for BeginMark / EndMark
	std::vector<uint32_t> actWords;    // [256][2 * NREG]: per action id NREG packed increment words, then NREG reset masks
	uint32_t nreg = 0;                 // 32-bit words of packed 16-bit counters: ceil(regexps / 2) rounded up to 1, 2, 4, 8
	// the letter-indexed form of CountingRowKernel (BASIC / ADVANCED, any number of states whose rows fit a CU's LDS,
	// <= 255 distinct actions, <= 8 regexps); empty otherwise
	std::vector<uint32_t> lrows;       // [states][letters] next state | action id << 16
	std::vector<uint32_t> lactWords;   // [distinct actions + 1][2 * lnreg], as actWords
	uint32_t lnreg = 0;
};

struct CountingDevice {
	int device = -1;
	uint8_t* letterOf = nullptr;
	uint64_t* trans = nullptr;
	uint32_t* actions = nullptr;
	uint8_t* tags = nullptr;
	uint16_t* dense = nullptr;
	uint16_t* denseMarks = nullptr;
	uint32_t* actWords = nullptr;
	uint32_t* lrows = nullptr;
	uint32_t* lactWords = nullptr;
};

}  // namespace pirehip

struct pire_hip_counting_table {
	pirehip::CountingHost host;
	pirehip::CountingDevice devs[pirehip::kMaxDevices];   // one image per HIP device, as in pire_hip_table
	std::mutex uploadMutex;
	// CapturingScanner on the ragged kernel with actions (BuildCaptureTable): the expanded automaton as an ordinary
	// table (dense rows, adaptation and all), built when first needed, and its per-state info word on every device
	std::unique_ptr<pire_hip_table> captureTable;
	std::vector<uint32_t> captureInfo;                    // [expanded states] original state << 8 | Final tag << 2 | action
	uint32_t* captureInfoDev[pirehip::kMaxDevices] = {};
	bool captureTried = false;
	std::atomic<uint32_t> selfTested[pirehip::kMaxDevices] = {};   // per device: bit 0 counting, bit 1 capture passed their known-answer batches (selftest.h)
};

namespace pirehip {

struct CountingParams {
	const uint8_t* letterOf;
	const uint64_t* trans;
	uint32_t states, letters, regexps, initial, flags, transInLds;
	const uint8_t* text;
	const uint64_t* offsets;
	uint64_t n;
	uint32_t* outIdx;
	uint32_t* outResults;
	const uint32_t* actions;   // NoGlueLimitCountingScanner action lists, or null (single regexp: raw action bits)
	uint32_t* scratch;         // NoGlueLimit with more than 16 regexps: current[n][regexps]
	// CountingPackedKernel
	const uint16_t* dense;
	const uint16_t* denseMarks;
	const uint32_t* actWords;
	const uint32_t* lrows;     // CountingRowKernel, letter-indexed rows: [states][letters] next | action id << 16
	const uint32_t* lactWords;
	uint32_t lnreg;            // 0: no letter-indexed rows (or PIRE_HIP_RUN_GENERIC)
	uint32_t lactCount;        // rows of lactWords (distinct actions + 1)
	uint32_t initialAct;       // action id pending before the first step (HalfFinal: Initialize ends with TakeAction)
	uint32_t maxLen;           // CountingRowKernel: longest string its 16-bit counters hold (0: 65 000); longer ones go on `overflow`
	uint32_t* overflow;        // [0] = count, [1 ..] = strings too long for 16-bit counters: the 32-bit kernel takes them
	const uint32_t* order;     // nullable: string k of the launch is order[k] (order.hip: by length class)
	uint32_t serpentine;       // walk the order forwards and backwards in turn (the global order)
	uint32_t spreadWaves;      // row kernels: a wave's 64 strings of the order go to the next BLOCK, not to the block's next wave (batches
	                           // with fewer strings than the chip has lanes: a few waves on every CU instead of sixteen on some)
	// CapturingScanner run
	const uint8_t* tags;
	uint8_t* outFinal;
	long long* outBegin;
	long long* outEnd;
};

// CountingState minus the state pointer (count.h:204-234).  Only bits 16..31 of m_updatedMask are ever read
// (PerformReset masks a 32-bit action with it, count.h:187) and the reset clears everything above (count.h:190), so
// a 32-bit mask is exact.
template <int RMAX>
struct Counters {
	uint32_t current[RMAX], total[RMAX];
	uint32_t updated;

	__device__ __forceinline__ void Init()
	{
#pragma unroll
		for (int r = 0; r < RMAX; ++r)
			current[r] = total[r] = 0;
		updated = 0;
	}
	__device__ __forceinline__ void Increment(uint32_t a)   // PerformIncrement, count.h:175-182
	{
#pragma unroll
		for (int r = 0; r < RMAX; ++r)
			current[r] += (a >> r) & 1u;
		updated |= a << kMaxReCount;
	}
	__device__ __forceinline__ void Reset(uint32_t a)       // PerformReset, count.h:184-192
	{
		const uint32_t m = a & updated;
		if (m) {
#pragma unroll
			for (int r = 0; r < RMAX; ++r)
				if (((m >> (kMaxReCount + r)) & 1u) && current[r]) {
					total[r] = total[r] > current[r] ? total[r] : current[r];
					current[r] = 0;
				}
			updated &= ~m;
		}
	}
	// NoGlueLimitCountingState (count.h:306-325): Reset(id): current = 0;  Increment(id): ++current, total = max
	__device__ __forceinline__ void ResetId(uint32_t id)
	{
#pragma unroll
		for (int r = 0; r < RMAX; ++r)
			current[r] = uint32_t(r) == id ? 0u : current[r];
	}
	__device__ __forceinline__ void IncrementId(uint32_t id)
	{
#pragma unroll
		for (int r = 0; r < RMAX; ++r)
			if (uint32_t(r) == id) {
				++current[r];
				total[r] = total[r] > current[r] ? total[r] : current[r];
			}
	}
	// NoGlueLimitCountingScanner::TakeActionImpl, count.h:404-437: resets first, then increments
	__device__ __forceinline__ void TakeNoGlue(const uint32_t* actions, uint32_t a)
	{
		if (actions) {
			const uint32_t* act = actions + a;
			for (uint32_t n = *act++; n--;)
				ResetId(*act++);
			for (uint32_t n = *act++; n--;)
				IncrementId(*act++);
		} else {
			if (a & 2u)
				ResetId(0);
			if (a & 1u)
				IncrementId(0);
		}
	}
	template <bool ADVANCED>
	__device__ __forceinline__ void Take(uint32_t a)
	{
		constexpr uint32_t kInc = (1u << kMaxReCount) - 1u, kReset = kInc << kMaxReCount;   // loaded.h:223-224
		if (ADVANCED) {
			if (a & kReset)
				Reset(a);
			if (a & kInc)
				Increment(a);
		} else {
			if (a & kInc)
				Increment(a);
			if (a & kReset)
				Reset(a);
		}
	}
};

// ---- CountingScanner / AdvancedCountingScanner, dense rows and packed counters (round 3) -----------------------------
// On text 22-46 % of a counting scanner's steps carry an action, so -- unlike HalfFinal counting or capturing -- there
// is no skipping the chunks without one; what can shrink is the step itself.  The kernel above pays, per byte, two
// dependent LDS lookups (letter, then the 8-byte transition) and, whenever any lane of the wave has an action (always),
// ~30 VALU instructions of per-regexp bit tests (PerformIncrement / PerformReset over RMAX counters).  Here:
//   * ONE lookup per byte: dense rows indexed by the input byte, entry = next state | action id << 8 (<= 255 states,
//     <= 255 distinct actions: counting tables have tens of states);
//   * counters PACKED two to a register as 16-bit halves: an action is NREG packed increment words and NREG reset
//     masks from LDS, applied with v_pk_add_u16 / v_pk_max_u16 and two ANDs per register.  PerformReset's
//     "mask &= m_updatedMask" needs no mask: a counter is non-zero exactly when its updated bit is set (an increment
//     sets both, only a reset clears both), and resetting a zero counter changes nothing (count.h:87-99, 175-192).
// 16 bits hold any count of a string shorter than 65 000 bytes (a step bumps a counter by at most one); longer strings
// go onto the overflow list and the 32-bit kernel above walks them (CountingKernel with CountingParams::overflow).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int NREG, bool ADVANCED>
__global__ __launch_bounds__(1024) void CountingPackedKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint16_t* dense = reinterpret_cast<uint16_t*>(lds);                               // [states][256]
	uint32_t* act = reinterpret_cast<uint32_t*>(lds + size_t(p.states) * 512);        // [256][2 * NREG]
	for (uint32_t i = threadIdx.x; i < p.states * 128; i += blockDim.x)
		reinterpret_cast<uint32_t*>(dense)[i] = reinterpret_cast<const uint32_t*>(p.dense)[i];
	for (uint32_t i = threadIdx.x; i < 256 * 2 * NREG; i += blockDim.x)
		act[i] = p.actWords[i];
	__syncthreads();
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < p.n; ++pass) {
		// order.hip: a wave takes 64 strings of about the same length, a lane long and short ones in turn
		const uint64_t k = OrderedIndex(pass, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, uint64_t(gridDim.x) * blockDim.x, p.serpentine != 0);
		if (k >= p.n)
			continue;
		const uint64_t s = p.order ? p.order[k] : k;
		const uint64_t b = p.offsets[s], e = p.offsets[s + 1];
		if (e - b > 65000) {
			const uint32_t k = atomicAdd(&p.overflow[0], 1u);
			p.overflow[1 + k] = uint32_t(s);
			continue;
		}
		u16x2 cur[NREG], tot[NREG];
#pragma unroll
		for (int r = 0; r < NREG; ++r)
			cur[r] = tot[r] = u16x2{0, 0};
		uint32_t st = p.initial;
		auto take = [&](uint32_t id) {   // TakeActionImpl: count.h:251-257 (increment, reset) / 287-295 (reset, increment)
			const uint32_t* w = act + id * (2 * NREG);
#pragma unroll
			for (int r = 0; r < NREG; ++r) {
				const uint32_t incBits = w[r], rstBits = w[NREG + r];
				u16x2 inc, rst;
				__builtin_memcpy(&inc, &incBits, 4);
				__builtin_memcpy(&rst, &rstBits, 4);
				if (!ADVANCED)
					cur[r] += inc;
				tot[r] = __builtin_elementwise_max(tot[r], cur[r] & rst);
				cur[r] &= ~rst;
				if (ADVANCED)
					cur[r] += inc;
			}
		};
		// Software pipeline: the lookup of step i+1 needs only the STATE of step i, so it is issued first and the
		// action of step i (another LDS read + eight packed instructions) is applied while it is on its way -- one LDS
		// round trip per byte on the dependent chain instead of two.  `pend` = the action id not applied yet.
		uint32_t pend = 0;
		auto step = [&](uint32_t entry) {
			if (pend)
				take(pend);
			st = entry & 0xFFu;
			pend = entry >> 8;
		};
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			step(p.denseMarks[st * 2]);
		const uint8_t* ptr = p.text + b;
		const uint8_t* end = p.text + e;
		while (ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
			step(dense[st * 256 + *ptr]);
			++ptr;
		}
		// the next 16 bytes are requested before this block's 16 steps, not when they are needed
		uint4 ahead = ptr + 16 <= end ? *reinterpret_cast<const uint4*>(ptr) : uint4{0, 0, 0, 0};
		for (; ptr + 16 <= end; ptr += 16) {
			uint4 v = ahead;
			if (ptr + 32 <= end)
				ahead = *reinterpret_cast<const uint4*>(ptr + 16);
#pragma unroll 1
			for (int i = 0; i < 4; ++i) {
				const uint32_t x = v.x;
				step(dense[st * 256 + (x & 0xFF)]);
				step(dense[st * 256 + ((x >> 8) & 0xFF)]);
				step(dense[st * 256 + ((x >> 16) & 0xFF)]);
				step(dense[st * 256 + (x >> 24)]);
				v.x = v.y;
				v.y = v.z;
				v.z = v.w;
			}
		}
		for (; ptr < end; ++ptr)
			step(dense[st * 256 + *ptr]);
		if (p.flags & PIRE_HIP_RUN_END)
			step(p.denseMarks[st * 2 + 1]);
		if (pend)
			take(pend);
		if (p.outIdx)
			p.outIdx[s] = st;
		for (uint32_t r = 0; r < p.regexps; ++r) {
			uint32_t c = 0, t = 0;
#pragma unroll
			for (int k = 0; k < NREG; ++k)
				if (uint32_t(k) == (r >> 1)) {
					c = (r & 1) ? cur[k].y : cur[k].x;
					t = (r & 1) ? tot[k].y : tot[k].x;
				}
			p.outResults[s * p.regexps + r] = c > t ? c : t;   // Result(r), count.h:206
		}
	}
}

// ---- the same walk on whole cache lines, with the step cut to its data flow (round 4) --------------------------------
// The kernel above runs at 1.05 TB/s whatever the table (one regexp or three, 4 states or 45): it waits for its text.  A
// lane asks for 16 bytes of its own string at a time, one block ahead of its walk -- 64 lines touched per instruction,
// each of them eight times, and a round trip to memory that sixteen steps do not cover.  Making the step cheaper alone
// (entries that are LDS addresses, below) bought 6-10 % (profiles/r04_counting_rows.log).  So:
//   * the text arrives as in the offset-batch kernels of the Scanner path: per lane one 128-byte LINE per window, fetched by
//     groups of 8 lanes (IssueTileGroup's pattern: an instruction touches 8 lines, every line once) while the previous
//     line is walked, transposed in registers (TransposeTile);
//   * an entry IS the two addresses a step needs: 8 bytes per (state, byte) = { LDS offset of the next state's row,
//     LDS offset of the action's words }, expanded from the u16 rows while the block copies them.  A step is one v_bfe,
//     one v_lshl_add (row + byte * 8), one ds_read_b64, one ds_read_b128 of the pending action's words and three packed
//     operations per counter register (below): no unpacking, and no branch around the action -- action 0 is a row of
//     zeroes;
//   * the bytes of a line in front of the string's first or behind its last take a 257th entry of the row, { the row
//     itself, action 0 }: a step that changes nothing, chosen by an add, a compare and a select.  Windows that lie
//     inside the strings of all the lanes still at work take a copy of the walk without the three; lanes that are done
//     sit in a sink row meanwhile.
// Rows indexed by the byte cost 2 KB of LDS per state: tables of up to kCountingRowStates states (counting tables have
// tens).  Larger ones take rows indexed by the table's own LETTERS -- the byte is translated first, one more LDS read off
// the dependent chain -- and fit as long as (states + 1) x (letters + 1) entries and their distinct actions fit a CU's
// LDS (template parameter LETTERS; 573 states x 21 letters with 7 regexps: 0.99 TB/s against 0.34).  One block of 16
// waves per CU either way; up to eight regexps (four counter registers).  Same counters, same overflow list, same length
// order.  HalfFinalScanner's match counting rides the same kernel (MODE 2: counters that are only added to).
// The line on its way lands in ACCUMULATION registers a0..a31 -- by name: the load instructions write a[4j:4j+3] and
// LandTile() reads a0..a31, whatever the compiler thinks.  To the compiler they are eight values that the loads define
// in exactly those registers and LandTile() consumes from exactly those registers ("{a[0:3]}" constraints), alive
// during the walk: so it keeps its own spills out of them (it uses free accumulation registers as spill space: with
// clobber lists alone it parked values of the walk in a0..a22, under the loads), and it has no reason to move them --
// nothing else wants an accumulation register here; tests/test_build_audit.py checks that no instruction but these
// touches a0..a31.  The first form of this kernel kept the line on its way in a second tile of ordinary registers, as
// stream.hip and tiled.hip do: a live-range split copied that tile in front of its s_waitcnt, and 57 of 4 096 strings
// came out wrong, all in the lanes whose loads are issued last (profiles/r04_counting_rows_vmcnt.log).
// LandTile() waits for the line and reads it into the one ordinary tile: 32 v_accvgpr_read per 128 steps.
struct AccTile {
	u32x4 r[8];
};
#define PIRE_ACC_LOAD(J, RANGE)                                                                                        \
	if constexpr (J_ == J)                                                                                            \
		asm volatile("global_load_dwordx4 " RANGE ", %1, off" : "={" RANGE "}"(acc.r[J]) : "v"(a))
template <int J_>
__device__ __forceinline__ void IssueAccGroupOne(AccTile& acc, uint32_t lo, uint32_t hi, uint32_t mine)
{
	const uint32_t l = GroupBroadcast<J_>(lo);
	const uint32_t h = GroupBroadcast<J_>(hi);
	const uint64_t a = ((uint64_t(h) << 32) | l) + mine;
	PIRE_ACC_LOAD(0, "a[0:3]");
	PIRE_ACC_LOAD(1, "a[4:7]");
	PIRE_ACC_LOAD(2, "a[8:11]");
	PIRE_ACC_LOAD(3, "a[12:15]");
	PIRE_ACC_LOAD(4, "a[16:19]");
	PIRE_ACC_LOAD(5, "a[20:23]");
	PIRE_ACC_LOAD(6, "a[24:27]");
	PIRE_ACC_LOAD(7, "a[28:31]");
}
#undef PIRE_ACC_LOAD

// IssueTileGroup() (device_common.h) with a0..a31 as the destination: instruction j loads, in every group of 8 lanes, the
// line of lane 8g+j, lane 8g+c its bytes [16c, 16c+16).
__device__ __forceinline__ void IssueAccGroup(AccTile& acc, uint64_t src, uint32_t lane)
{
	const uint32_t lo = uint32_t(src), hi = uint32_t(src >> 32);
	const uint32_t mine = (lane & 7u) << 4;
	IssueAccGroupOne<0>(acc, lo, hi, mine);
	IssueAccGroupOne<1>(acc, lo, hi, mine);
	IssueAccGroupOne<2>(acc, lo, hi, mine);
	IssueAccGroupOne<3>(acc, lo, hi, mine);
	IssueAccGroupOne<4>(acc, lo, hi, mine);
	IssueAccGroupOne<5>(acc, lo, hi, mine);
	IssueAccGroupOne<6>(acc, lo, hi, mine);
	IssueAccGroupOne<7>(acc, lo, hi, mine);
}

#define PIRE_ACC_IN                                                                                                    \
	"{a[0:3]}"(acc.r[0]), "{a[4:7]}"(acc.r[1]), "{a[8:11]}"(acc.r[2]), "{a[12:15]}"(acc.r[3]), "{a[16:19]}"(acc.r[4]),   \
		"{a[20:23]}"(acc.r[5]), "{a[24:27]}"(acc.r[6]), "{a[28:31]}"(acc.r[7])
__device__ __forceinline__ void LandTile(const AccTile& acc, u32x4 (&r)[8])
{
	uint32_t w[32];
	asm volatile("s_waitcnt vmcnt(0)\n\t"
	             "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
	             "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\t"
	             "v_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\t"
	             "v_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15"
	             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]), "=&v"(w[8]),
	               "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11]), "=&v"(w[12]), "=&v"(w[13]), "=&v"(w[14]), "=&v"(w[15])
	             : PIRE_ACC_IN);
	asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\t"
	             "v_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\t"
	             "v_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\t"
	             "v_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31"
	             : "=&v"(w[16]), "=&v"(w[17]), "=&v"(w[18]), "=&v"(w[19]), "=&v"(w[20]), "=&v"(w[21]), "=&v"(w[22]), "=&v"(w[23]),
	               "=&v"(w[24]), "=&v"(w[25]), "=&v"(w[26]), "=&v"(w[27]), "=&v"(w[28]), "=&v"(w[29]), "=&v"(w[30]), "=&v"(w[31])
	             : PIRE_ACC_IN);
#pragma unroll
	for (int q = 0; q < 8; ++q)
		r[q] = u32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
}
#undef PIRE_ACC_IN

constexpr uint32_t kCountingRowStates = 64;
constexpr uint32_t kCountingRowPitch = 257 * 8;
constexpr uint32_t kCaptureRowPitch = 257 * 16;    // CaptureRowKernel: 16-byte entries
constexpr uint32_t kCaptureRowStates = 34;         // (34 + 1) x 4 112 bytes = 141 KB

// LETTERS: rows indexed by the table's own letters (a byte is translated first: one more LDS read, off the dependent
// chain) -- (states + 1) x (letters + 1) entries, so tables of hundreds of states fit; without it rows of 257 entries
// indexed by the byte itself (<= 64 states).
// MODE: 0 CountingScanner (increment, reset), 1 AdvancedCountingScanner (reset, increment), 2 counters that are only
// ever added to (HalfFinalScanner's match counts: one packed add per register and step, no reset words read).
template <int NREG, int MODE, bool LETTERS>
__global__ __launch_bounds__(1024) void CountingRowKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	typedef uint32_t Pair __attribute__((ext_vector_type(2)));
	typedef uint32_t Quad __attribute__((ext_vector_type(4)));
	// (LDS by absolute address: the dynamic segment starts at 0, this kernel has no static LDS)
	typedef const __attribute__((address_space(3))) Pair* LdsPair;
	typedef const __attribute__((address_space(3))) Quad* LdsQuad;
	typedef const __attribute__((address_space(3))) uint32_t* LdsU32;
	const uint32_t pitch = LETTERS ? (p.letters + 1u) * 8u : kCountingRowPitch;
	const uint32_t idleOff = pitch - 8u;                     // the row's last entry: { this row, no action }
	const uint32_t sinkRow = p.states * pitch;               // every entry: { the sink row, no action }
	const uint32_t actBase = ((p.states + 1) * pitch + 15u) & ~15u;
	const uint32_t actCount = LETTERS ? p.lactCount : 256u;
	const uint32_t lettab = actBase + actCount * 8u * NREG;  // LETTERS: u16[256], 8 * letter of every byte
	for (uint32_t i = threadIdx.x; i < pitch / 8u; i += blockDim.x)
		*reinterpret_cast<uint2*>(lds + sinkRow + i * 8u) = uint2{sinkRow, actBase};
	if (LETTERS) {
		for (uint32_t i = threadIdx.x; i < p.states * p.letters; i += blockDim.x) {
			const uint32_t e = p.lrows[i], st = i / p.letters, l = i - st * p.letters;
			*reinterpret_cast<uint2*>(lds + st * pitch + l * 8u) = uint2{(e & 0xFFFFu) * pitch, actBase + (e >> 16) * (8u * NREG)};
		}
		for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
			*reinterpret_cast<uint16_t*>(lds + lettab + i * 2u) = uint16_t(p.letterOf[i] * 8u);
	} else {
		for (uint32_t i = threadIdx.x; i < p.states * 256; i += blockDim.x) {
			const uint32_t e = p.dense[i];
			*reinterpret_cast<uint2*>(lds + (i >> 8) * pitch + (i & 255u) * 8u) = uint2{(e & 0xFFu) * pitch, actBase + (e >> 8) * (8u * NREG)};
		}
	}
	for (uint32_t i = threadIdx.x; i < p.states; i += blockDim.x)
		*reinterpret_cast<uint2*>(lds + i * pitch + idleOff) = uint2{i * pitch, actBase};
	for (uint32_t i = threadIdx.x; i < actCount * 2 * NREG; i += blockDim.x)
		reinterpret_cast<uint32_t*>(lds + actBase)[i] = (LETTERS ? p.lactWords : p.actWords)[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint64_t maxLen = p.maxLen ? p.maxLen : 65000u;
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < p.n; ++pass) {
		// (no lane leaves the pass early: the loads and the transpose below are the whole wave's)
		// (spreadWaves: four waves share a SIMD, and this kernel is bound by its issue -- 2^17 strings 788 instead of 514 GB/s)
		const uint64_t k = OrderedIndex(pass, p.spreadWaves ? (uint64_t(threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64 + lane : uint64_t(blockIdx.x) * blockDim.x + threadIdx.x,
		                               uint64_t(gridDim.x) * blockDim.x, p.serpentine != 0);
		uint64_t b = 0, e = 0;
		if (k < p.n) {
			const uint64_t s = p.order ? p.order[k] : k;
			b = p.offsets[s];
			e = p.offsets[s + 1];
			if (e - b > maxLen) {
				const uint32_t slot = atomicAdd(&p.overflow[0], 1u);
				p.overflow[1 + slot] = uint32_t(s);
			}
		}
		const bool ok = k < p.n && e - b <= maxLen;
		const uint32_t len = ok ? uint32_t(e - b) : 0u;
		const uint64_t first = reinterpret_cast<uint64_t>(p.text) + b;
		const uint64_t line0 = first & ~uint64_t(127);
		const uint32_t lead = uint32_t(first - line0);
		const uint32_t windows = len ? (lead + len + 127u) >> 7 : 0u;
		u16x2 cur[NREG], tot[NREG];
#pragma unroll
		for (int r = 0; r < NREG; ++r)
			cur[r] = tot[r] = u16x2{0, 0};
		uint32_t row = p.initial * pitch, pend = actBase + p.initialAct * (8u * NREG);   // pend: LDS offset of the words of the action not applied yet
		auto take = [&]() __attribute__((always_inline)) {   // TakeActionImpl: count.h:251-257 (increment, reset) / 287-295 (reset, increment)
			constexpr bool ADVANCED = MODE == 1, SUM = MODE == 2;
			uint32_t w[2 * NREG];
			if (SUM && NREG == 1) {
				w[0] = *reinterpret_cast<LdsU32>(static_cast<uintptr_t>(pend));
			} else if (NREG == 1 || (SUM && NREG == 2)) {
				const Pair q = *reinterpret_cast<LdsPair>(static_cast<uintptr_t>(pend));
				w[0] = q.x;
				w[1] = q.y;
			} else {
#pragma unroll
				for (int q4 = 0; q4 < (SUM ? NREG / 4 : NREG / 2); ++q4) {
					const Quad q = *reinterpret_cast<LdsQuad>(static_cast<uintptr_t>(pend + 16u * q4));
					w[4 * q4] = q.x;
					w[4 * q4 + 1] = q.y;
					w[4 * q4 + 2] = q.z;
					w[4 * q4 + 3] = q.w;
				}
			}
			if (SUM) {
#pragma unroll
				for (int r = 0; r < NREG; ++r) {
					u16x2 inc;
					__builtin_memcpy(&inc, &w[r], 4);
					cur[r] += inc;
					uint32_t pin;   // (a chain, as the maxima below: hipcc sums the increments of a window first, in a tree)
					__builtin_memcpy(&pin, &cur[r], 4);
					asm volatile("" : "+v"(pin));
					__builtin_memcpy(&cur[r], &pin, 4);
				}
				return;
			}
			// Between two resets a counter only grows, and the result is max(current, total) (count.h:206): so the largest
			// value `current` ever has IS the result, and the maximum can be taken at every step instead of under the
			// reset mask -- three packed operations per register (and-not, add, max) instead of four.  total then is
			// max(total, current) of count.h plus values it would only have taken in at the next reset.
#pragma unroll
			for (int r = 0; r < NREG; ++r) {
				const uint32_t incBits = w[r], rstBits = w[NREG + r];
				u16x2 inc, rst;
				__builtin_memcpy(&inc, &incBits, 4);
				__builtin_memcpy(&rst, &rstBits, 4);
				if (!ADVANCED)
					cur[r] += inc;
				tot[r] = __builtin_elementwise_max(tot[r], cur[r]);
				// (the maxima stay a chain: left alone, hipcc reassociates the 128 of a window into a tree and spills
				// its leaves -- 1.2-2.5 KB of scratch per lane)
				uint32_t pin;
				__builtin_memcpy(&pin, &tot[r], 4);
				asm volatile("" : "+v"(pin));
				__builtin_memcpy(&tot[r], &pin, 4);
				cur[r] &= ~rst;
				if (ADVANCED)
					cur[r] += inc;
			}
		};
		// as above: the lookup of step i+1 needs only the row of step i, the action of step i is applied under it
		auto step = [&](uint32_t entryOffset) __attribute__((always_inline)) {
			const Pair next = *reinterpret_cast<LdsPair>(static_cast<uintptr_t>(row + entryOffset));
			take();
			row = next.x;
			pend = next.y;
		};
		auto mark = [&](uint32_t which) __attribute__((always_inline)) {   // BeginMark / EndMark, twice per string
			if (LETTERS) {
				step(p.letterOf[which ? kEndMark : kBeginMark] * 8u);
			} else {
				const uint32_t m = p.denseMarks[(row / kCountingRowPitch) * 2 + which];   // the u16 entry from global memory
				take();
				row = (m & 0xFFu) * kCountingRowPitch;
				pend = actBase + (m >> 8) * (8u * NREG);
			}
		};
		// the entry a byte selects within a row
		auto at8 = [&](uint32_t byte) __attribute__((always_inline)) -> uint32_t {
			return LETTERS ? LdsU16(lettab + byte * 2u) : byte * 8u;
		};
		if (ok && (p.flags & PIRE_HIP_RUN_BEGIN))
			mark(0);
		// Window t of a lane = the line line0 + 128 t.  An iteration lands the line that was on its way (window t), asks
		// for the next one and walks; the first iteration (t = -1) lands nothing of value and does not walk.
		u32x4 tile[8];
		AccTile acc;
		// (whatever a0..a31 hold: the first iteration lands it and walks nothing)
		asm volatile("" : "={a[0:3]}"(acc.r[0]), "={a[4:7]}"(acc.r[1]), "={a[8:11]}"(acc.r[2]), "={a[12:15]}"(acc.r[3]),
		             "={a[16:19]}"(acc.r[4]), "={a[20:23]}"(acc.r[5]), "={a[24:27]}"(acc.r[6]), "={a[28:31]}"(acc.r[7]));
		for (uint32_t t = ~0u;;) {
			LandTile(acc, tile);
			const uint32_t tn = t + 1u;
			// (lanes without a further line fetch a harmless valid one: the table)
			IssueAccGroup(acc, tn < windows ? line0 + uint64_t(tn) * 128u : reinterpret_cast<uint64_t>(LETTERS ? static_cast<const void*>(p.lrows) : static_cast<const void*>(p.dense)), lane);
			if (t != ~0u) {
				TransposeTile(tile, lane);
				const uint32_t at = t * 128u - lead;   // byte j of this window is byte at + j of the string (mod 2^32)
				const bool inside = t * 128u >= lead && at + 128u <= len;
				const bool idle = t >= windows;
				if (__all(inside || idle)) {
					// every lane's line lies inside its string, or behind it: the walk without the compares.  The lanes
					// that are done spend it in the sink row (their pending action is applied by the first step).
					const uint32_t keep = row;
					row = idle ? sinkRow : row;
#pragma unroll
					for (int q = 0; q < 8; ++q)
#pragma unroll
						for (int w = 0; w < 4; ++w) {
							const uint32_t x = tile[q][w];
							step(at8(x & 0xFFu));
							step(at8((x >> 8) & 0xFFu));
							step(at8((x >> 16) & 0xFFu));
							step(at8(x >> 24));
							__builtin_amdgcn_sched_barrier(0);
						}
					row = idle ? keep : row;
				} else {
#pragma unroll
					for (int q = 0; q < 8; ++q)
#pragma unroll
						for (int w = 0; w < 4; ++w) {
							const uint32_t x = tile[q][w];
							const uint32_t j = 16u * q + 4u * w;
							step(at + j < len ? at8(x & 0xFFu) : idleOff);
							step(at + j + 1u < len ? at8((x >> 8) & 0xFFu) : idleOff);
							step(at + j + 2u < len ? at8((x >> 16) & 0xFFu) : idleOff);
							step(at + j + 3u < len ? at8(x >> 24) : idleOff);
							__builtin_amdgcn_sched_barrier(0);   // a dword at a time: hipcc otherwise hoists the compares of the whole window
						}
				}
			}
			t = tn;
			if (!__any(t < windows))
				break;
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the line asked for last (nobody's) has landed before a0..a31 are asked again
		if (ok && (p.flags & PIRE_HIP_RUN_END))
			mark(1);
		take();
		const uint64_t s = ok ? (p.order ? p.order[k] : k) : 0;   // read again: two registers less across the window loop
		if (ok && p.outIdx)
			p.outIdx[s] = row / pitch;
		if (ok && p.outFinal)
			p.outFinal[s] = p.tags[row / pitch] & 1u;   // (HalfFinal counting: Final of the end state)
		if (ok)
			for (uint32_t r = 0; r < p.regexps; ++r) {
				uint32_t c = 0, m = 0;
#pragma unroll
				for (int q = 0; q < NREG; ++q)
					if (uint32_t(q) == (r >> 1)) {
						c = (r & 1) ? cur[q].y : cur[q].x;
						m = (r & 1) ? tot[q].y : tot[q].x;
					}
				p.outResults[s * p.regexps + r] = c > m ? c : m;   // Result(r), count.h:206
			}
	}
}

// KIND: PIRE_HIP_COUNTING_BASIC / _ADVANCED / _NOGLUELIMIT (the latter with at most RMAX regexps: counters in registers)
// (blocks of up to 16 waves -- 8 for the sixteen-counter instantiations, whose counters need more than 128 registers --
// so that a table of up to 150 KB can sit in LDS: LaunchOne)
template <int RMAX, int KIND>
__global__ __launch_bounds__(RMAX > 8 ? 512 : 1024) void CountingKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t* letterOf = lds;                                            // 264 bytes
	uint64_t* transLds = reinterpret_cast<uint64_t*>(lds + 272);
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x)
		letterOf[i] = p.letterOf[i];
	if (p.transInLds)
		for (uint32_t i = threadIdx.x; i < p.states * p.letters; i += blockDim.x)
			transLds[i] = p.trans[i];
	__syncthreads();
	const uint64_t* trans = p.transInLds ? transLds : p.trans;

	// with an overflow list (the packed kernel ran first on this stream): only the strings it names
	const uint64_t todo = p.overflow ? p.overflow[0] : p.n;
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < todo; ++pass) {
		const uint64_t k = OrderedIndex(pass, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, uint64_t(gridDim.x) * blockDim.x,
		                                !p.overflow && p.serpentine);
		if (k >= todo)
			continue;
		const uint64_t s = p.overflow ? p.overflow[1 + k] : p.order ? p.order[k] : k;
		Counters<RMAX> c;
		c.Init();                                                       // Initialize, count.h:127-133
		uint32_t st = p.initial;
		auto step = [&](uint32_t ch) {                                  // Step = Next + TakeAction, run.h:50-57
			const uint64_t x = trans[st * p.letters + letterOf[ch]];
			st = uint32_t(x);
			const uint32_t a = uint32_t(x >> 32);
			if (a) {
				if (KIND == PIRE_HIP_COUNTING_NOGLUELIMIT)
					c.TakeNoGlue(p.actions, a);
				else
					c.template Take<KIND == PIRE_HIP_COUNTING_ADVANCED>(a);
			}
		};
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			step(kBeginMark);
		const uint8_t* ptr = p.text + p.offsets[s];
		const uint8_t* end = p.text + p.offsets[s + 1];
		while (ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
			step(*ptr);
			++ptr;
		}
		// the next 16 bytes are requested before this block's 16 steps, not when they are needed
		uint4 ahead = ptr + 16 <= end ? *reinterpret_cast<const uint4*>(ptr) : uint4{0, 0, 0, 0};
		for (; ptr + 16 <= end; ptr += 16) {
			uint4 v = ahead;
			if (ptr + 32 <= end)
				ahead = *reinterpret_cast<const uint4*>(ptr + 16);
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
		if (p.flags & PIRE_HIP_RUN_END)
			step(kEndMark);
		if (p.outIdx)
			p.outIdx[s] = st;
		for (uint32_t r = 0; r < p.regexps; ++r) {
			uint32_t cur = 0, tot = 0;
#pragma unroll
			for (int k = 0; k < RMAX; ++k)
				if (uint32_t(k) == r) {
					cur = c.current[k];
					tot = c.total[k];
				}
			p.outResults[s * p.regexps + r] = cur > tot ? cur : tot;   // Result(r), count.h:206
		}
	}
}

// NoGlueLimitCountingScanner with more than 16 regexps: the counters of a string live in its own rows of two global
// arrays (current: scratch, total: the result array itself -- Result(r) == total[r], because Increment keeps
// total >= current and Reset only clears current); the lane owns the rows, so plain read-modify-write.
__global__ __launch_bounds__(256) void CountingWideKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t* letterOf = lds;
	uint64_t* transLds = reinterpret_cast<uint64_t*>(lds + 272);
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x)
		letterOf[i] = p.letterOf[i];
	if (p.transInLds)
		for (uint32_t i = threadIdx.x; i < p.states * p.letters; i += blockDim.x)
			transLds[i] = p.trans[i];
	__syncthreads();
	const uint64_t* trans = p.transInLds ? transLds : p.trans;
	const uint32_t R = p.regexps;
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < p.n; ++pass) {
		// order.hip: a wave takes 64 strings of about the same length, a lane long and short ones in turn
		const uint64_t k = OrderedIndex(pass, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, uint64_t(gridDim.x) * blockDim.x, p.serpentine != 0);
		if (k >= p.n)
			continue;
		const uint64_t s = p.order ? p.order[k] : k;
		uint32_t* current = p.scratch + s * R;
		uint32_t* total = p.outResults + s * R;
		for (uint32_t r = 0; r < R; ++r)
			current[r] = total[r] = 0;
		uint32_t st = p.initial;
		auto step = [&](uint32_t ch) {
			const uint64_t x = trans[st * p.letters + letterOf[ch]];
			st = uint32_t(x);
			const uint32_t a = uint32_t(x >> 32);
			if (a) {
				const uint32_t* act = p.actions + a;
				for (uint32_t n = *act++; n--;)
					current[*act++] = 0;
				for (uint32_t n = *act++; n--;) {
					const uint32_t id = *act++;
					const uint32_t c = ++current[id];
					if (c > total[id])
						total[id] = c;
				}
			}
		};
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			step(kBeginMark);
		const uint8_t* ptr = p.text + p.offsets[s];
		const uint8_t* end = p.text + p.offsets[s + 1];
		for (; ptr < end; ++ptr)
			step(*ptr);
		if (p.flags & PIRE_HIP_RUN_END)
			step(kEndMark);
		if (p.outIdx)
			p.outIdx[s] = st;
	}
}

// Pire::CapturingScanner (extra/capture.h:49-162): the same LoadedScanner walk; the state carries the step counter and
// the begin / end of the one captured group (capture.h:59-87).  TakeAction, capture.h:96-102: a BeginCapture (1)
// action records counter - 1 as begin, otherwise an EndCapture (2) action records it as end, both only until the
// capture is complete.  Positions count steps, the BeginMark step included (tests/capture_ut.cpp:85-91 subtracts 1).
__global__ __launch_bounds__(256) void CaptureKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t* letterOf = lds;
	uint64_t* transLds = reinterpret_cast<uint64_t*>(lds + 272);
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x)
		letterOf[i] = p.letterOf[i];
	if (p.transInLds)
		for (uint32_t i = threadIdx.x; i < p.states * p.letters; i += blockDim.x)
			transLds[i] = p.trans[i];
	__syncthreads();
	const uint64_t* trans = p.transInLds ? transLds : p.trans;
	// 32-bit positions (a string is shorter than 4 GiB); widened to the reference's size_t / npos on output
	constexpr uint32_t npos = ~uint32_t(0);
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < p.n; ++pass) {
		// order.hip: a wave takes 64 strings of about the same length, a lane long and short ones in turn
		const uint64_t k = OrderedIndex(pass, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, uint64_t(gridDim.x) * blockDim.x, p.serpentine != 0);
		if (k >= p.n)
			continue;
		const uint64_t s = p.order ? p.order[k] : k;
		uint32_t st = p.initial;
		uint32_t begin = npos, end = npos, counter = 0;     // Initialize, capture.h:89-94
		auto step = [&](uint32_t ch) {
			const uint64_t x = trans[st * p.letters + letterOf[ch]];
			st = uint32_t(x);
			++counter;                                        // NextTranslated, capture.h:109-116
			const uint32_t a = uint32_t(x >> 32);
			// TakeAction, capture.h:96-102, as two selects (an if / else-if on two variables became a store through
			// a selected stack address, i.e. scratch)
			const bool open = !(begin != npos && end != npos);
			const bool setBegin = (a & 1u) && open;
			const bool setEnd = !(a & 1u) && (a & 2u) && open;
			begin = setBegin ? counter - 1 : begin;
			end = setEnd ? counter - 1 : end;
		};
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			step(kBeginMark);
		const uint8_t* ptr = p.text + p.offsets[s];
		const uint8_t* stop = p.text + p.offsets[s + 1];
		while (ptr < stop && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
			step(*ptr);
			++ptr;
		}
		// the next 16 bytes are requested before this block's 16 steps, not when they are needed
		uint4 ahead = ptr + 16 <= stop ? *reinterpret_cast<const uint4*>(ptr) : uint4{0, 0, 0, 0};
		for (; ptr + 16 <= stop; ptr += 16) {
			uint4 v = ahead;
			if (ptr + 32 <= stop)
				ahead = *reinterpret_cast<const uint4*>(ptr + 16);
#pragma unroll 1
			for (int i = 0; i < 16; ++i) {
				step(v.x & 0xFF);
				v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
				v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
				v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
				v.w >>= 8;
			}
		}
		for (; ptr < stop; ++ptr)
			step(*ptr);
		if (p.flags & PIRE_HIP_RUN_END)
			step(kEndMark);
		if (p.outIdx)
			p.outIdx[s] = st;
		if (p.outFinal)
			p.outFinal[s] = p.tags[st] & 1u;                  // Final, capture.h:134 (FinalFlag = 1)
		p.outBegin[s] = begin == npos ? -1ll : (long long)begin;   // npos -> -1
		p.outEnd[s] = end == npos ? -1ll : (long long)end;
	}
}

// The same walk with ONE lookup per byte: dense rows indexed by the input byte, entry = next state | action << 8
// (BuildDenseCounting: <= 255 states; the action is the raw 2-bit capture action, not an id), the action applied while
// the next lookup is on its way, the text requested 16 bytes ahead -- the recipe of CountingPackedKernel.  For the
// capture scanners that are in an action state on most bytes and therefore stay off the ragged kernel.
__global__ __launch_bounds__(1024) void CaptureDenseKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint16_t* dense = reinterpret_cast<uint16_t*>(lds);
	for (uint32_t i = threadIdx.x; i < p.states * 128; i += blockDim.x)
		reinterpret_cast<uint32_t*>(dense)[i] = reinterpret_cast<const uint32_t*>(p.dense)[i];
	__syncthreads();
	constexpr uint32_t npos = ~uint32_t(0);
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < p.n; ++pass) {
		// order.hip: a wave takes 64 strings of about the same length, a lane long and short ones in turn
		const uint64_t k = OrderedIndex(pass, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, uint64_t(gridDim.x) * blockDim.x, p.serpentine != 0);
		if (k >= p.n)
			continue;
		const uint64_t s = p.order ? p.order[k] : k;
		uint32_t st = p.initial;
		uint32_t begin = npos, end = npos, counter = 0, pend = 0, pendAt = 0;
		auto take = [&]() {   // TakeAction, capture.h:96-102, for the step that set `pend`, at m_counter - 1 = pendAt
			const bool open = !(begin != npos && end != npos);
			const bool setBegin = (pend & 1u) && open;
			const bool setEnd = !(pend & 1u) && (pend & 2u) && open;
			begin = setBegin ? pendAt : begin;
			end = setEnd ? pendAt : end;
		};
		auto step = [&](uint32_t entry) {
			if (pend)
				take();
			st = entry & 0xFFu;
			pend = entry >> 8;
			pendAt = counter++;
		};
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			step(p.denseMarks[st * 2]);
		const uint8_t* ptr = p.text + p.offsets[s];
		const uint8_t* stop = p.text + p.offsets[s + 1];
		while (ptr < stop && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
			step(dense[st * 256 + *ptr]);
			++ptr;
		}
		uint4 ahead = ptr + 16 <= stop ? *reinterpret_cast<const uint4*>(ptr) : uint4{0, 0, 0, 0};
		for (; ptr + 16 <= stop; ptr += 16) {
			uint4 v = ahead;
			if (ptr + 32 <= stop)
				ahead = *reinterpret_cast<const uint4*>(ptr + 16);
#pragma unroll 1
			for (int i = 0; i < 4; ++i) {
				const uint32_t x = v.x;
				step(dense[st * 256 + (x & 0xFF)]);
				step(dense[st * 256 + ((x >> 8) & 0xFF)]);
				step(dense[st * 256 + ((x >> 16) & 0xFF)]);
				step(dense[st * 256 + (x >> 24)]);
				v.x = v.y;
				v.y = v.z;
				v.z = v.w;
			}
		}
		for (; ptr < stop; ++ptr)
			step(dense[st * 256 + *ptr]);
		if (p.flags & PIRE_HIP_RUN_END)
			step(p.denseMarks[st * 2 + 1]);
		if (pend)
			take();
		if (p.outIdx)
			p.outIdx[s] = st;
		if (p.outFinal)
			p.outFinal[s] = p.tags[st] & 1u;                  // Final, capture.h:134 (FinalFlag = 1)
		p.outBegin[s] = begin == npos ? -1ll : (long long)begin;
		p.outEnd[s] = end == npos ? -1ll : (long long)end;
	}
}

namespace {

struct RefHeader {
	uint32_t magic, version, ptrSize, maxWordSize, type, hdrSize;
};
struct LoadedLocals {
	uint32_t statesCount, lettersCount, regexpsCount, pad;
	uint64_t initial;
};

int Bad(const char* msg)
{
	SetError(msg);
	return PIRE_HIP_EFORMAT;
}

// ---- CapturingScanner on whole text lines (round 4) --------------------------------------------------------------------
// CaptureDenseKernel with the text path and the entries of CountingRowKernel: one 128-byte line per lane and window,
// landing in a0..a31 while the previous one is walked; entries of 8 bytes = { LDS offset of the next state's row, what
// the step's action does: 1 = BeginCapture, 2 = EndCapture alone (capture.h:96-102: Begin wins when both are set) }; a
// 257th entry per row for the bytes outside the string, a sink row for lanes that are done.  The capture is kept as
// masks: `sel` = the action's kind as an all-ones word, and-ed with "capture not complete yet"; begin / end take the
// position through v_bfi.  Positions count steps (the BeginMark step included); a step's action is applied one step
// later, so what is stored is the position of the step that applies it, corrected by one at the end.
__global__ __launch_bounds__(1024) void CaptureRowKernel(CountingParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	typedef uint32_t Quad __attribute__((ext_vector_type(4)));
	typedef const __attribute__((address_space(3))) Quad* LdsQuad;
	constexpr uint32_t kPitch = kCaptureRowPitch, kIdle = 256u * 16u;
	const uint32_t sinkRow = p.states * kPitch;
	for (uint32_t i = threadIdx.x; i < 257; i += blockDim.x)
		*reinterpret_cast<uint4*>(lds + sinkRow + i * 16u) = uint4{sinkRow, 0u, 0u, 0u};
	for (uint32_t i = threadIdx.x; i < p.states * 256; i += blockDim.x) {
		const uint32_t e = p.dense[i], a = e >> 8;
		// the action as two masks: BeginCapture, EndCapture alone (capture.h:96-102: Begin wins when both are set)
		*reinterpret_cast<uint4*>(lds + (i >> 8) * kPitch + (i & 255u) * 16u) =
			uint4{(e & 0xFFu) * kPitch, (a & 1u) ? ~0u : 0u, (!(a & 1u) && (a & 2u)) ? ~0u : 0u, 0u};
	}
	for (uint32_t i = threadIdx.x; i < p.states; i += blockDim.x)
		*reinterpret_cast<uint4*>(lds + i * kPitch + kIdle) = uint4{i * kPitch, 0u, 0u, 0u};
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	constexpr uint32_t npos = ~uint32_t(0);
	const uint32_t beginStep = (p.flags & PIRE_HIP_RUN_BEGIN) ? 1u : 0u;
	for (uint64_t pass = 0; pass * gridDim.x * blockDim.x < p.n; ++pass) {
		// (no lane leaves the pass early: the loads and the transpose below are the whole wave's)
		const uint64_t k = OrderedIndex(pass, p.spreadWaves ? (uint64_t(threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64 + lane : uint64_t(blockIdx.x) * blockDim.x + threadIdx.x,   // (CountingRowKernel)
		                                uint64_t(gridDim.x) * blockDim.x, p.serpentine != 0);
		uint64_t b = 0, e = 0;
		if (k < p.n) {
			const uint64_t s = p.order ? p.order[k] : k;
			b = p.offsets[s];
			e = p.offsets[s + 1];
		}
		// (positions are 32-bit, as in the kernels above)
		const bool ok = k < p.n;
		const uint32_t len = ok ? uint32_t(e - b) : 0u;
		const uint64_t first = reinterpret_cast<uint64_t>(p.text) + b;
		const uint64_t line0 = first & ~uint64_t(127);
		const uint32_t lead = uint32_t(first - line0);
		const uint32_t windows = len ? (lead + len + 127u) >> 7 : 0u;
		uint32_t row = p.initial * kPitch, pendB = 0, pendE = 0;
		uint32_t begin = npos, end = npos, hasBegin = 0, hasEnd = 0;
		// TakeAction (capture.h:96-102) of the step before the one at position `at`
		auto take = [&](uint32_t at) __attribute__((always_inline)) {
			const uint32_t open = ~(hasBegin & hasEnd);
			const uint32_t selB = pendB & open;
			const uint32_t selE = pendE & open;
			begin = (at & selB) | (begin & ~selB);
			end = (at & selE) | (end & ~selE);
			hasBegin |= selB;
			hasEnd |= selE;
		};
		auto step = [&](uint32_t entryOffset, uint32_t at) __attribute__((always_inline)) {
			const Quad next = *reinterpret_cast<LdsQuad>(static_cast<uintptr_t>(row + entryOffset));
			take(at);
			row = next.x;
			pendB = next.y;
			pendE = next.z;
		};
		auto mark = [&](uint32_t which, uint32_t at) __attribute__((always_inline)) {
			const uint32_t m = p.denseMarks[(row / kPitch) * 2 + which];
			take(at);
			row = (m & 0xFFu) * kPitch;
			const uint32_t a = m >> 8;
			pendB = (a & 1u) ? ~0u : 0u;
			pendE = (!(a & 1u) && (a & 2u)) ? ~0u : 0u;
		};
		// `at` of a step = the string's byte index it consumes (BeginMark: -1, EndMark: len); its own position in steps
		// is at + beginStep, the position of the step whose action it applies one less
		if (ok && beginStep)
			mark(0, npos);
		u32x4 tile[8];
		AccTile acc;
		asm volatile("" : "={a[0:3]}"(acc.r[0]), "={a[4:7]}"(acc.r[1]), "={a[8:11]}"(acc.r[2]), "={a[12:15]}"(acc.r[3]),
		             "={a[16:19]}"(acc.r[4]), "={a[20:23]}"(acc.r[5]), "={a[24:27]}"(acc.r[6]), "={a[28:31]}"(acc.r[7]));
		for (uint32_t t = ~0u;;) {
			LandTile(acc, tile);
			const uint32_t tn = t + 1u;
			IssueAccGroup(acc, tn < windows ? line0 + uint64_t(tn) * 128u : reinterpret_cast<uint64_t>(p.dense), lane);
			if (t != ~0u) {
				TransposeTile(tile, lane);
				const uint32_t at = t * 128u - lead;
				const bool inside = t * 128u >= lead && at + 128u <= len;
				const bool idle = t >= windows;
				if (__all(inside || idle)) {
					const uint32_t keep = row;
					row = idle ? sinkRow : row;
#pragma unroll
					for (int q = 0; q < 8; ++q)
#pragma unroll
						for (int w = 0; w < 4; ++w) {
							const uint32_t x = tile[q][w];
							const uint32_t j = 16u * q + 4u * w;
							step((x & 0xFFu) * 16u, at + j);
							step(((x >> 8) & 0xFFu) * 16u, at + j + 1u);
							step(((x >> 16) & 0xFFu) * 16u, at + j + 2u);
							step((x >> 24) * 16u, at + j + 3u);
							__builtin_amdgcn_sched_barrier(0);
						}
					row = idle ? keep : row;
				} else {
#pragma unroll
					for (int q = 0; q < 8; ++q)
#pragma unroll
						for (int w = 0; w < 4; ++w) {
							const uint32_t x = tile[q][w];
							const uint32_t j = 16u * q + 4u * w;
							step(at + j < len ? (x & 0xFFu) * 16u : kIdle, at + j);
							step(at + j + 1u < len ? ((x >> 8) & 0xFFu) * 16u : kIdle, at + j + 1u);
							step(at + j + 2u < len ? ((x >> 16) & 0xFFu) * 16u : kIdle, at + j + 2u);
							step(at + j + 3u < len ? (x >> 24) * 16u : kIdle, at + j + 3u);
							__builtin_amdgcn_sched_barrier(0);
						}
				}
			}
			t = tn;
			if (!__any(t < windows))
				break;
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (ok && (p.flags & PIRE_HIP_RUN_END)) {
			mark(1, len);
			take(len + 1u);
		} else {
			take(len);
		}
		if (ok) {
			const uint64_t s = p.order ? p.order[k] : k;   // read again: two registers less across the window loop
			const uint32_t st = row / kPitch;
			if (p.outIdx)
				p.outIdx[s] = st;
			if (p.outFinal)
				p.outFinal[s] = p.tags[st] & 1u;                  // Final, capture.h:134 (FinalFlag = 1)
			// stored: the byte index of the step that applied the action = the action's own byte index + 1
			p.outBegin[s] = hasBegin ? (long long)(begin + beginStep - 1u) : -1ll;
			p.outEnd[s] = hasEnd ? (long long)(end + beginStep - 1u) : -1ll;
		}
	}
}

// The dense, packed form of CountingPackedKernel for LoadedScanner tables (CountingScanner / AdvancedCountingScanner:
// the action word is increment bits 0..15 | reset bits 16..31, loaded.h:223-224).  Left empty -- the 32-bit kernel
// stays -- for more than 255 states or distinct actions, or a table that would not leave room for several blocks per CU.
void BuildDenseCounting(CountingHost& t)
{
	t.dense.clear();
	t.nreg = 0;
	if (t.type != 4 || t.states == 0 || t.states > 255 || t.regexps == 0 || t.regexps > kMaxReCount)
		return;
	const uint32_t nreg = t.regexps <= 2 ? 1 : t.regexps <= 4 ? 2 : t.regexps <= 8 ? 4 : 8;
	if (size_t(t.states) * 512 + 256 * 2 * nreg * 4 > 150 * 1024)   // (more than 40 KB: one block of 16 waves per CU, LaunchPacked)
		return;
	std::vector<uint32_t> ids;   // distinct non-zero action words, id = index + 1
	// tables whose action words all fit a byte (CapturingScanner: 1 = BeginCapture, 2 = EndCapture) keep them as ids:
	// CaptureDenseKernel reads the action itself out of the entry
	uint32_t maxAction = 0;
	for (uint64_t x : t.trans)
		maxAction = std::max(maxAction, uint32_t(x >> 32));
	if (maxAction <= 255)
		for (uint32_t a = 1; a <= maxAction; ++a)
			ids.push_back(a);
	auto idOf = [&](uint32_t a) -> uint32_t {
		if (!a)
			return 0;
		for (size_t i = 0; i < ids.size(); ++i)
			if (ids[i] == a)
				return uint32_t(i + 1);
		ids.push_back(a);
		return uint32_t(ids.size());
	};
	std::vector<uint16_t> dense(size_t(t.states) * 256), marks(size_t(t.states) * 2);
	for (uint32_t st = 0; st < t.states; ++st) {
		for (uint32_t ch = 0; ch < 256 + 2; ++ch) {
			const uint32_t c = ch < 256 ? ch : ch == 256 ? kBeginMark : kEndMark;
			const uint64_t x = t.trans[size_t(st) * t.letters + t.letterOf[c]];
			const uint32_t id = idOf(uint32_t(x >> 32));
			if (id > 255 || uint32_t(x) > 255)
				return;
			const uint16_t e = uint16_t(uint32_t(x) | (id << 8));
			if (ch < 256)
				dense[size_t(st) * 256 + ch] = e;
			else
				marks[size_t(st) * 2 + (ch - 256)] = e;
		}
	}
	t.actWords.assign(size_t(256) * 2 * nreg, 0);
	for (size_t i = 0; i < ids.size(); ++i) {
		uint32_t* w = &t.actWords[(i + 1) * 2 * nreg];
		for (uint32_t r = 0; r < t.regexps; ++r) {
			if ((ids[i] >> r) & 1u)
				w[r >> 1] |= 1u << (16 * (r & 1));                    // +1 in the counter's 16-bit half
			if ((ids[i] >> (kMaxReCount + r)) & 1u)
				w[nreg + (r >> 1)] |= 0xFFFFu << (16 * (r & 1));      // the counter's reset mask
		}
	}
	t.dense.swap(dense);
	t.denseMarks.swap(marks);
	t.nreg = nreg;
}

// The letter-indexed rows of CountingRowKernel: what BuildDenseCounting does for tables of up to 255 states, without the
// expansion to 256 columns -- (states + 1) x (letters + 1) entries of 8 bytes in LDS, so hundreds of states fit.
constexpr size_t kCountingRowLds = 150 * 1024;
void BuildLetterRows(CountingHost& t)
{
	t.lrows.clear();
	t.lnreg = 0;
	if (t.type != 4 || t.states == 0 || t.states > 65535 || t.regexps == 0 || t.regexps > 8 || t.letters == 0 || t.letters > 255)
		return;
	const uint32_t nreg = t.regexps <= 2 ? 1 : t.regexps <= 4 ? 2 : 4;
	const size_t rowBytes = (size_t(t.states) + 1) * (t.letters + 1) * 8 + 16 + 512;
	if (rowBytes + 2 * 8 * nreg > kCountingRowLds)
		return;
	// as many distinct actions as fit behind the rows (8 * nreg bytes each), at most what 16 bits number
	const size_t maxIds = std::min<size_t>(65535, (kCountingRowLds - rowBytes) / (8 * nreg) - 1);
	std::vector<uint32_t> ids;   // distinct non-zero action words, id = index + 1
	std::unordered_map<uint32_t, uint32_t> idOf;
	std::vector<uint32_t> rows(size_t(t.states) * t.letters);
	for (size_t i = 0; i < rows.size(); ++i) {
		const uint64_t x = t.trans[i];
		const uint32_t a = uint32_t(x >> 32);
		uint32_t id = 0;
		if (a) {
			auto it = idOf.find(a);
			if (it == idOf.end()) {
				if (ids.size() == maxIds)
					return;
				ids.push_back(a);
				it = idOf.emplace(a, uint32_t(ids.size())).first;
			}
			id = it->second;
		}
		if (uint32_t(x) >= t.states)
			return;
		rows[i] = uint32_t(x) | (id << 16);
	}
	t.lactWords.assign((ids.size() + 1) * 2 * nreg, 0);
	for (size_t i = 0; i < ids.size(); ++i) {
		uint32_t* w = &t.lactWords[(i + 1) * 2 * nreg];
		for (uint32_t r = 0; r < t.regexps; ++r) {
			if ((ids[i] >> r) & 1u)
				w[r >> 1] |= 1u << (16 * (r & 1));                    // +1 in the counter's 16-bit half
			if ((ids[i] >> (kMaxReCount + r)) & 1u)
				w[nreg + (r >> 1)] |= 0xFFFFu << (16 * (r & 1));      // the counter's reset mask
		}
	}
	t.lrows.swap(rows);
	t.lnreg = nreg;
}

int BuildCountingHost(const void* blob, size_t len, CountingHost* out)
{
	const uint8_t* p = static_cast<const uint8_t*>(blob);
	if (!p || len < sizeof(RefHeader))
		return Bad("EOF reached while reading the scanner header");
	RefHeader h;
	memcpy(&h, p, sizeof(h));
	// Header::Validate, common.h:65-77; type LoadedScanner = 4, common.h:39
	// type LoadedScanner = 4, NoGlueLimitCountingScanner = 5 (common.h:39-40)
	if (h.magic != 0x45524950u || h.ptrSize != 8 || h.maxWordSize != 16 || (h.type != 4 && h.type != 5) ||
	    h.hdrSize != sizeof(LoadedLocals))
		return Bad("Serialized regexp incompatible with your system");
	if (h.version != 7 && h.version != 6)
		return Bad("You are trying to used an incompatible version of a serialized regexp");
	size_t pos = 24;
	LoadedLocals m;
	if (len < pos + sizeof(m))
		return Bad("EOF reached while reading the scanner locals");
	memcpy(&m, p + pos, sizeof(m));
	pos += sizeof(m);
	if (m.statesCount == 0 || m.lettersCount == 0 || m.lettersCount > 256 || (h.type == 4 && m.regexpsCount > kMaxReCount))
		return Bad("Corrupt scanner: bad state, letter or regexp count");
	const size_t njumps = size_t(m.statesCount) * m.lettersCount;
	if (len < pos + 264 + njumps * 8 + m.statesCount)
		return Bad("EOF reached while reading the scanner buffer");
	CountingHost& t = *out;
	t = CountingHost();
	t.states = m.statesCount;
	t.letters = m.lettersCount;
	t.regexps = m.regexpsCount;
	t.type = h.type;
	const uint64_t stateSize = uint64_t(m.lettersCount) * 8;   // StateSize(), loaded.h:171-174
	if (m.initial % stateSize != 0 || m.initial / stateSize >= m.statesCount)
		return Bad("Corrupt scanner: initial state out of range");
	t.initial = uint32_t(m.initial / stateSize);
	t.letterOf.assign(p + pos, p + pos + 264);
	for (uint32_t c = 0; c < 264; ++c)
		if (t.letterOf[c] >= m.lettersCount)
			return Bad("Corrupt scanner: letter out of range");
	pos += 264;   // already a multiple of 8 (AlignedSaveArray)
	t.trans.resize(njumps);
	for (uint32_t s = 0; s < m.statesCount; ++s)
		for (uint32_t l = 0; l < m.lettersCount; ++l) {
			uint32_t shift, action;
			memcpy(&shift, p + pos + (size_t(s) * m.lettersCount + l) * 8, 4);
			memcpy(&action, p + pos + (size_t(s) * m.lettersCount + l) * 8 + 4, 4);
			const int64_t dest = int64_t(s) * int64_t(stateSize) + int64_t(int32_t(shift));   // SignExtend, loaded.h:210
			if (dest < 0 || uint64_t(dest) % stateSize != 0 || uint64_t(dest) / stateSize >= m.statesCount)
				return Bad("Corrupt scanner: transition out of range");
			t.trans[size_t(s) * m.lettersCount + l] = uint64_t(dest) / stateSize | (uint64_t(action) << 32);
		}
	{
		size_t tpos = pos + njumps * 8;
		if (h.version == 6)
			tpos += (njumps * 4 + 7) / 8 * 8;   // the ignored per-transition action array of the old format
		if (len < tpos + m.statesCount)
			return Bad("EOF reached while reading the scanner tags");
		t.tags.assign(p + tpos, p + tpos + m.statesCount);
	}
	if (h.type == 5) {
		// NoGlueLimitCountingScanner::Load, count.cpp:1020-1035: behind the (8-byte padded) tags a u32 length
		// (0 = no table: one regexp, raw action bits), then length-1 more words
		pos += njumps * 8;
		if (h.version == 6)
			pos += (njumps * 4 + 7) / 8 * 8;
		pos += (size_t(m.statesCount) + 7) / 8 * 8;
		uint32_t size = 0;
		if (len < pos + 4)
			return Bad("EOF reached while reading the action table");
		memcpy(&size, p + pos, 4);
		if (size) {
			if (len < pos + size_t(size) * 4)
				return Bad("EOF reached while reading the action table");
			t.actions.resize(size);
			memcpy(t.actions.data(), p + pos, size_t(size) * 4);
			// every action word of the table must point at two well-formed lists
			for (uint64_t x : t.trans) {
				const uint32_t a = uint32_t(x >> 32);
				if (!a)
					continue;
				size_t q = a;
				for (int list = 0; list < 2; ++list) {
					if (q >= size)
						return Bad("Corrupt scanner: action out of range");
					const uint32_t cnt = t.actions[q++];
					if (q + cnt > size)
						return Bad("Corrupt scanner: action list out of range");
					for (uint32_t k = 0; k < cnt; ++k)
						if (t.actions[q + k] >= m.regexpsCount)
							return Bad("Corrupt scanner: regexp id out of range");
					q += cnt;
				}
			}
		} else if (m.regexpsCount > 1) {
			return Bad("Corrupt scanner: several regexps without an action table");
		}
	}
	BuildDenseCounting(t);
	BuildLetterRows(t);
	return PIRE_HIP_OK;
}

void FreeCountingDevice(CountingDevice* d)
{
	if (d->device < 0)
		return;
	if (d->letterOf)
		(void)hipFree(d->letterOf);
	if (d->trans)
		(void)hipFree(d->trans);
	if (d->actions)
		(void)hipFree(d->actions);
	if (d->tags)
		(void)hipFree(d->tags);
	if (d->dense)
		(void)hipFree(d->dense);
	if (d->denseMarks)
		(void)hipFree(d->denseMarks);
	if (d->actWords)
		(void)hipFree(d->actWords);
	if (d->lrows)
		(void)hipFree(d->lrows);
	if (d->lactWords)
		(void)hipFree(d->lactWords);
	*d = CountingDevice();
}

// Image of the current device (built on first use), copied out under the table's lock.
int UploadCounting(pire_hip_counting_table* t, CountingDevice* image)
{
	std::lock_guard<std::mutex> lock(t->uploadMutex);
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return HipFail(e, "hipGetDevice");
	if (dev < 0 || dev >= kMaxDevices) {
		SetError("HIP device ordinal out of range");
		return PIRE_HIP_EUNSUPPORTED;
	}
	if (t->devs[dev].device == dev) {
		*image = t->devs[dev];
		return PIRE_HIP_OK;
	}
	CountingDevice d;
	e = hipMalloc(reinterpret_cast<void**>(&d.letterOf), 272);
	if (e == hipSuccess)
		e = hipMalloc(reinterpret_cast<void**>(&d.trans), t->host.trans.size() * 8);
	if (e == hipSuccess)
		e = hipMemcpy(d.letterOf, t->host.letterOf.data(), 264, hipMemcpyHostToDevice);
	if (e == hipSuccess)
		e = hipMemcpy(d.trans, t->host.trans.data(), t->host.trans.size() * 8, hipMemcpyHostToDevice);
	if (e == hipSuccess && !t->host.actions.empty()) {
		e = hipMalloc(reinterpret_cast<void**>(&d.actions), t->host.actions.size() * 4);
		if (e == hipSuccess)
			e = hipMemcpy(d.actions, t->host.actions.data(), t->host.actions.size() * 4, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess) {
		e = hipMalloc(reinterpret_cast<void**>(&d.tags), t->host.tags.size() + 16);
		if (e == hipSuccess)
			e = hipMemcpy(d.tags, t->host.tags.data(), t->host.tags.size(), hipMemcpyHostToDevice);
	}
	if (e == hipSuccess && !t->host.dense.empty()) {
		const CountingHost& h = t->host;
		e = hipMalloc(reinterpret_cast<void**>(&d.dense), h.dense.size() * 2);
		if (e == hipSuccess)
			e = hipMalloc(reinterpret_cast<void**>(&d.denseMarks), h.denseMarks.size() * 2);
		if (e == hipSuccess)
			e = hipMalloc(reinterpret_cast<void**>(&d.actWords), h.actWords.size() * 4);
		if (e == hipSuccess)
			e = hipMemcpy(d.dense, h.dense.data(), h.dense.size() * 2, hipMemcpyHostToDevice);
		if (e == hipSuccess)
			e = hipMemcpy(d.denseMarks, h.denseMarks.data(), h.denseMarks.size() * 2, hipMemcpyHostToDevice);
		if (e == hipSuccess)
			e = hipMemcpy(d.actWords, h.actWords.data(), h.actWords.size() * 4, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess && !t->host.lrows.empty()) {
		const CountingHost& h = t->host;
		e = hipMalloc(reinterpret_cast<void**>(&d.lrows), h.lrows.size() * 4);
		if (e == hipSuccess)
			e = hipMalloc(reinterpret_cast<void**>(&d.lactWords), h.lactWords.size() * 4);
		if (e == hipSuccess)
			e = hipMemcpy(d.lrows, h.lrows.data(), h.lrows.size() * 4, hipMemcpyHostToDevice);
		if (e == hipSuccess)
			e = hipMemcpy(d.lactWords, h.lactWords.data(), h.lactWords.size() * 4, hipMemcpyHostToDevice);
	}
	d.device = dev;
	if (e != hipSuccess) {
		FreeCountingDevice(&d);
		return HipFail(e, "uploading the counting table");
	}
	t->devs[dev] = d;
	*image = d;
	return PIRE_HIP_OK;
}

template <int RMAX, int KIND>
void LaunchOne(const CountingParams& p, unsigned cus, uint32_t ldsBytes, hipStream_t stream, hipError_t* err)
{
	*err = SetDynamicLds(reinterpret_cast<const void*>(CountingKernel<RMAX, KIND>), uint32_t(ldsBytes));
	if (*err != hipSuccess)
		return;
	// (a table of more than 40 KB: blocks of 16 waves, as LaunchPacked)
	const unsigned threads = ldsBytes > 40 * 1024 ? (RMAX > 8 ? 512 : 1024) : 256;
	const uint64_t perCu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / threads, (160 * 1024) / ldsBytes));
	const uint64_t todo = p.n;   // (with an overflow list: at most this many)
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((todo + threads - 1) / threads, uint64_t(cus) * perCu)));
	hipLaunchKernelGGL((CountingKernel<RMAX, KIND>), dim3(blocks), dim3(threads), ldsBytes, stream, p);
	*err = hipGetLastError();
}

template <int NREG, bool LETTERS>
void LaunchRow(const CountingParams& p, int mode, unsigned blocks, uint32_t ldsBytes, hipStream_t stream, hipError_t* err)
{
	auto go = [&](auto m) {
		constexpr int MODE = decltype(m)::value;
		*err = SetDynamicLds(reinterpret_cast<const void*>(CountingRowKernel<NREG, MODE, LETTERS>), uint32_t(ldsBytes));
		if (*err != hipSuccess)
			return;
		hipLaunchKernelGGL((CountingRowKernel<NREG, MODE, LETTERS>), dim3(blocks), dim3(1024), ldsBytes, stream, p);
		*err = hipGetLastError();
	};
	if (mode == 2) {
		if constexpr (LETTERS)
			go(std::integral_constant<int, 2>());
	} else if (mode == 1) {
		go(std::integral_constant<int, 1>());
	} else {
		go(std::integral_constant<int, 0>());
	}
}

template <int NREG>
void LaunchPacked(const CountingParams& p, bool advanced, unsigned cus, uint32_t ldsBytes, hipStream_t stream, hipError_t* err)
{
	const void* fn = advanced ? reinterpret_cast<const void*>(CountingPackedKernel<NREG, true>)
	                          : reinterpret_cast<const void*>(CountingPackedKernel<NREG, false>);
	*err = SetDynamicLds(fn, uint32_t(ldsBytes));
	if (*err != hipSuccess)
		return;
	// small tables: several 256-thread blocks per CU; a table of more than 40 KB (up to 255 states: 128 KB of rows) leaves
	// room for one to three blocks, so these are blocks of 16 waves (round 4: such tables went to the 32-bit kernel, a
	// 143-state scanner of 7 regexps at 0.45 TB/s)
	const unsigned threads = ldsBytes > 40 * 1024 ? 1024 : 256;
	const uint64_t perCu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / threads, (160 * 1024) / ldsBytes));
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + threads - 1) / threads, uint64_t(cus) * perCu)));
	if (advanced)
		hipLaunchKernelGGL((CountingPackedKernel<NREG, true>), dim3(blocks), dim3(threads), ldsBytes, stream, p);
	else
		hipLaunchKernelGGL((CountingPackedKernel<NREG, false>), dim3(blocks), dim3(threads), ldsBytes, stream, p);
	*err = hipGetLastError();
}

int LaunchCounting(CountingParams p, int kind, hipStream_t stream, uint32_t nreg = 0)
{
	if (p.n == 0)
		return PIRE_HIP_OK;
	int dev = 0, cus = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e == hipSuccess)
		e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (e != hipSuccess)
		return HipFail(e, "device query");
	// the strings by length class (order.hip), so that a wave's 64 lanes finish together
	void* orderScratch = nullptr;
	struct OrderGuard {
		void*& q;
		hipStream_t s;
		~OrderGuard()
		{
			if (q)
				(void)hipFreeAsync(q, s);
		}
	} orderGuard{orderScratch, stream};
	p.order = nullptr;
	if (p.offsets && LengthOrderWanted(p.n)) {
		e = hipMallocAsync(&orderScratch, LengthOrderScratchBytes(p.n), stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(length order)");
		bool serp = false;
		if (int rc = BuildLengthOrder(p.offsets, p.n, orderScratch, stream, &p.order, &serp))
			return rc;
		p.serpentine = serp ? 1u : 0u;
	}
	// Dense rows + packed 16-bit counters first (CountingPackedKernel); the strings it leaves on the overflow list
	// (longer than 65 000 bytes) go through the 32-bit kernel below on the same stream.
	void* list = nullptr;
	struct ListGuard {
		void*& q;
		hipStream_t s;
		~ListGuard()
		{
			if (q)
				(void)hipFreeAsync(q, s);
		}
	} listGuard{list, stream};
	// letter-indexed rows (CountingRowKernel<.., LETTERS>): any table whose (states + 1) x (letters + 1) entries fit
	const uint32_t lnreg = p.lnreg;
	const int variant = GetConfig().counting_variant;
	const bool fills = variant == 2 || p.n >= uint64_t(cus) * 256;   // the one block of 16 waves a CU then holds
	const bool byteRows = nreg && p.dense && nreg <= 4 && p.states <= kCountingRowStates && variant != 1 && fills;
	const bool letterRows = !byteRows && lnreg && p.lrows && variant != 1 && fills;
	if (((nreg && p.dense) || letterRows) && kind != PIRE_HIP_COUNTING_NOGLUELIMIT && p.n < (1ull << 32) - 1) {
		e = hipMallocAsync(&list, (size_t(p.n) + 1) * 4, stream);
		if (e == hipSuccess)
			e = hipMemsetAsync(list, 0, 4, stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(counting overflow list)");
		p.overflow = static_cast<uint32_t*>(list);
		const bool adv = kind == PIRE_HIP_COUNTING_ADVANCED;
		// entries that are LDS addresses (CountingRowKernel) where the table leaves room for them and the batch fills the
		// GPU; pire_hip_config.counting_variant: 1 = never, 2 = whenever the table fits
		p.spreadWaves = p.n <= uint64_t(cus) * 1024 ? 1u : 0u;   // (one pass: the lengths of the order spread over the CUs as well)
		const unsigned rblocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>(p.spreadWaves ? (p.n + 63) / 64 : (p.n + 1023) / 1024, uint64_t(cus))));
		if (byteRows) {
			const uint32_t rowLds = uint32_t(((size_t(p.states + 1) * kCountingRowPitch + 15) & ~size_t(15)) + 256 * 2 * nreg * 4);
			switch (nreg) {
			case 1: LaunchRow<1, false>(p, adv ? 1 : 0, rblocks, rowLds, stream, &e); break;
			case 2: LaunchRow<2, false>(p, adv ? 1 : 0, rblocks, rowLds, stream, &e); break;
			default: LaunchRow<4, false>(p, adv ? 1 : 0, rblocks, rowLds, stream, &e); break;
			}
		} else if (letterRows) {
			const uint32_t rowLds = uint32_t(((size_t(p.states + 1) * (p.letters + 1) * 8 + 15) & ~size_t(15)) + size_t(p.lactCount) * 2 * lnreg * 4 + 512);
			switch (lnreg) {
			case 1: LaunchRow<1, true>(p, adv ? 1 : 0, rblocks, rowLds, stream, &e); break;
			case 2: LaunchRow<2, true>(p, adv ? 1 : 0, rblocks, rowLds, stream, &e); break;
			default: LaunchRow<4, true>(p, adv ? 1 : 0, rblocks, rowLds, stream, &e); break;
			}
		} else {
			const uint32_t packedLds = uint32_t(size_t(p.states) * 512 + 256 * 2 * nreg * 4);
			switch (nreg) {
			case 1: LaunchPacked<1>(p, adv, unsigned(cus), packedLds, stream, &e); break;
			case 2: LaunchPacked<2>(p, adv, unsigned(cus), packedLds, stream, &e); break;
			case 4: LaunchPacked<4>(p, adv, unsigned(cus), packedLds, stream, &e); break;
			default: LaunchPacked<8>(p, adv, unsigned(cus), packedLds, stream, &e); break;
			}
		}
		if (e != hipSuccess)
			return HipFail(e, "counting kernel launch");
		const bool rows = byteRows || letterRows;
		NoteKernel(byteRows ? "counting_rows" : letterRows ? "counting_letter_rows" : "counting_packed");
		(void)rows;
	} else {
		NoteKernel("counting");
	}
	const uint64_t tableBytes = uint64_t(p.states) * p.letters * 8;
	// the transitions in LDS while they fit a CU's 160 KB (round 4: up to 60 KB only -- a 573-state scanner of 7 regexps
	// read every transition from memory); above 40 KB the blocks are of 16 waves (LaunchOne)
	const bool wideKernel = kind == PIRE_HIP_COUNTING_NOGLUELIMIT && p.regexps > kMaxReCount;
	p.transInLds = tableBytes <= (wideKernel ? 60 : 150) * 1024 ? 1 : 0;
	const uint32_t ldsBytes = 272 + (p.transInLds ? uint32_t(tableBytes) : 0);
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + 255) / 256, uint64_t(cus) * 8)));
	if (kind == PIRE_HIP_COUNTING_NOGLUELIMIT && p.regexps > kMaxReCount) {
		e = SetDynamicLds(reinterpret_cast<const void*>(CountingWideKernel), uint32_t(ldsBytes));
		if (e == hipSuccess) {
			hipLaunchKernelGGL(CountingWideKernel, dim3(blocks), dim3(256), ldsBytes, stream, p);
			e = hipGetLastError();
		}
	} else {
		// counters in registers, as many as the scanner has regexps (rounded up to 1, 2, 4, 8, 16): TakeAction touches
		// every counter slot, so a single-regexp scanner should not pay for eight
		auto launch = [&](auto rmax) {
			constexpr int R = decltype(rmax)::value;
			switch (kind) {
			case PIRE_HIP_COUNTING_BASIC: LaunchOne<R, PIRE_HIP_COUNTING_BASIC>(p, unsigned(cus), ldsBytes, stream, &e); break;
			case PIRE_HIP_COUNTING_ADVANCED: LaunchOne<R, PIRE_HIP_COUNTING_ADVANCED>(p, unsigned(cus), ldsBytes, stream, &e); break;
			default: LaunchOne<R, PIRE_HIP_COUNTING_NOGLUELIMIT>(p, unsigned(cus), ldsBytes, stream, &e); break;
			}
		};
		if (p.regexps <= 1)
			launch(std::integral_constant<int, 1>());
		else if (p.regexps <= 2)
			launch(std::integral_constant<int, 2>());
		else if (p.regexps <= 4)
			launch(std::integral_constant<int, 4>());
		else if (p.regexps <= 8)
			launch(std::integral_constant<int, 8>());
		else
			launch(std::integral_constant<int, 16>());
	}
	if (e != hipSuccess)
		return HipFail(e, "counting kernel launch");
	return PIRE_HIP_OK;
}

}  // namespace

// ---- HalfFinalScanner counting on the row kernel (round 4) --------------------------------------------------------------
// HalfFinalScanner::TakeAction (half_final.h:137-164) bumps the counter of every regexp in the final list of the state a
// step ARRIVES in.  As a counting table: the action of (state, letter) = the increments of next(state, letter); Initialize
// ends with TakeAction too, which is an action pending before the first step.  No resets, so Result = the count.  Needs
// counters that pack (<= 8 regexps); a regexp that is m times in a state's final list bumps its 16-bit counter by m, so the
// row kernel takes strings of up to 65 000 / m bytes and leaves longer ones on its list.
namespace {
void BuildHalfRows(const HostTable& h, HalfRowsHost& r)
{
	r.tried = true;
	r.nreg = 0;
	if (!h.incPacked || h.states == 0 || h.states > 65535 || h.letters == 0 || h.letters > 255 || h.regexps == 0 || h.regexps > 8)
		return;
	uint32_t maxMult = 1;   // how often a regexp is in one state's final list (a step bumps its counter by that much)
	for (uint64_t inc : h.inc64)
		for (int b = 0; b < 8; ++b)
			maxMult = std::max(maxMult, uint32_t(inc >> (8 * b)) & 0xFFu);
	const uint32_t nreg = h.regexps <= 2 ? 1 : h.regexps <= 4 ? 2 : 4;
	const size_t rowBytes = (size_t(h.states) + 1) * (h.letters + 1) * 8 + 16 + 512;
	if (rowBytes + 2 * 8 * nreg > kCountingRowLds)
		return;
	const size_t maxIds = std::min<size_t>(65535, (kCountingRowLds - rowBytes) / (8 * nreg) - 1);
	std::vector<uint64_t> ids;   // distinct non-zero increment words, id = index + 1
	std::unordered_map<uint64_t, uint32_t> idOf;
	auto idFor = [&](uint32_t st, bool* ok) -> uint32_t {
		const uint64_t inc = (h.flags[st] & kFinal) ? h.inc64[st] : 0;
		if (!inc)
			return 0;
		auto it = idOf.find(inc);
		if (it == idOf.end()) {
			if (ids.size() == maxIds) {
				*ok = false;
				return 0;
			}
			ids.push_back(inc);
			it = idOf.emplace(inc, uint32_t(ids.size())).first;
		}
		return it->second;
	};
	bool ok = true;
	std::vector<uint32_t> rows(size_t(h.states) * h.letters);
	for (size_t i = 0; i < rows.size() && ok; ++i)
		rows[i] = h.next[i] | (idFor(h.next[i], &ok) << 16);
	const uint32_t initialAct = idFor(h.initial, &ok);
	if (!ok)
		return;
	r.lactWords.assign((ids.size() + 1) * 2 * nreg, 0);
	for (size_t i = 0; i < ids.size(); ++i)
		for (uint32_t q = 0; q < h.regexps; ++q)
			r.lactWords[(i + 1) * 2 * nreg + (q >> 1)] |= (uint32_t(ids[i] >> (8 * q)) & 0xFFu) << (16 * (q & 1));   // the counter's 16-bit half
	r.letterOf.resize(264);
	for (size_t c = 0; c < 264; ++c)
		r.letterOf[c] = uint8_t(h.cls[c]);
	r.finalTag.resize(h.states);
	for (uint32_t st = 0; st < h.states; ++st)
		r.finalTag[st] = (h.flags[st] & kFinal) ? 1 : 0;
	r.lrows.swap(rows);
	r.initialAct = initialAct;
	r.maxLen = 65000u / maxMult;   // (+ the two marks and Initialize: 65 002 x 1, 32 502 x 2, ... all below 65 536)
	r.nreg = nreg;
}

void FreeHalfRowsDevice(HalfRowsDevice* d)
{
	if (d->lrows) (void)hipFree(d->lrows);
	if (d->lactWords) (void)hipFree(d->lactWords);
	if (d->letterOf) (void)hipFree(d->letterOf);
	if (d->finalTag) (void)hipFree(d->finalTag);
	*d = HalfRowsDevice();
}

template <class T>
hipError_t PutHalf(T** dst, const std::vector<T>& v, size_t pad = 0)
{
	hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), v.size() * sizeof(T) + pad);
	if (e == hipSuccess)
		e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
	return e;
}
}  // namespace

void FreeHalfRows(pire_hip_table* t)
{
	int cur = -1;
	(void)hipGetDevice(&cur);
	for (int k = 0; k < kMaxDevices; ++k)
		if (t->halfRowsDev[k].device >= 0) {
			(void)hipSetDevice(k);
			FreeHalfRowsDevice(&t->halfRowsDev[k]);
		}
	if (cur >= 0)
		(void)hipSetDevice(cur);
}

// The letter-indexed rows of dense HalfFinal counting on the CURRENT device: built (host) and uploaded on first need.
// pire_hip_table_upload() calls this, so that an ON_DEVICE call of pire_hip_run_half_final after it allocates and copies
// nothing synchronously (the first such call on a device did, under halfRowsMutex).  A table that does not
// qualify (more than 8 regexps, rows that do not fit the LDS) costs the host build once and uploads nothing.
int UploadHalfRows(pire_hip_table* t)
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return HipFail(e, "hipGetDevice");
	if (dev < 0 || dev >= kMaxDevices)
		return PIRE_HIP_OK;
	std::lock_guard<std::mutex> lock(t->halfRowsMutex);
	if (!t->halfRows.tried)
		BuildHalfRows(t->host, t->halfRows);   // (reference numbering: nothing an adaptation changes)
	const HalfRowsHost& h = t->halfRows;
	HalfRowsDevice& d = t->halfRowsDev[dev];
	if (!h.nreg || d.device == dev)
		return PIRE_HIP_OK;
	e = PutHalf(&d.lrows, h.lrows, 128);
	if (e == hipSuccess)
		e = PutHalf(&d.lactWords, h.lactWords);
	if (e == hipSuccess)
		e = PutHalf(&d.letterOf, h.letterOf, 8);
	if (e == hipSuccess)
		e = PutHalf(&d.finalTag, h.finalTag, 16);
	if (e != hipSuccess) {
		FreeHalfRowsDevice(&d);
		return HipFail(e, "uploading the half-final rows");
	}
	d.device = dev;
	return PIRE_HIP_OK;
}

int LaunchHalfFinalRows(pire_hip_table* t, const uint8_t* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                        uint32_t* outIdx, uint8_t* outFinal, uint32_t* outResults, hipStream_t stream, bool* done,
                        uint32_t** overflow)
{
	*done = false;
	*overflow = nullptr;
	const int variant = GetConfig().counting_variant;
	if (variant == 1 || n == 0 || n >= (1ull << 32) - 1 || !offsets)
		return PIRE_HIP_OK;
	int dev = 0, cus = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e == hipSuccess)
		e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (e != hipSuccess)
		return HipFail(e, "device query");
	if (dev < 0 || dev >= kMaxDevices || (variant != 2 && n < uint64_t(cus) * 256))
		return PIRE_HIP_OK;
	HalfRowsDevice image;
	uint32_t nreg, initialAct, lactCount, maxLen;
	{
		// (the image of this device: uploaded here on the first call unless pire_hip_table_upload() did it -- UploadHalfRows)
		if (int rc = UploadHalfRows(t))
			return rc;
		std::lock_guard<std::mutex> lock(t->halfRowsMutex);
		const HalfRowsHost& h = t->halfRows;
		if (!h.nreg || t->halfRowsDev[dev].device != dev)
			return PIRE_HIP_OK;
		image = t->halfRowsDev[dev];
		nreg = h.nreg;
		initialAct = h.initialAct;
		maxLen = h.maxLen;
		lactCount = uint32_t(h.lactWords.size() / (2 * h.nreg));
	}
	CountingParams p;
	memset(&p, 0, sizeof(p));
	p.letterOf = image.letterOf;
	p.lrows = image.lrows;
	p.lactWords = image.lactWords;
	p.lnreg = nreg;
	p.lactCount = lactCount;
	p.initialAct = initialAct;
	p.maxLen = maxLen;
	p.tags = image.finalTag;
	p.states = t->host.states;
	p.letters = t->host.letters;
	p.regexps = t->host.regexps;
	p.initial = t->host.initial;
	p.flags = flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END);
	p.text = text;
	p.offsets = offsets;
	p.n = n;
	p.outIdx = outIdx;
	p.outFinal = outFinal;
	p.outResults = outResults;
	// strings by length class, as the counting scanners (order.hip)
	void* orderScratch = nullptr;
	if (LengthOrderWanted(n)) {
		e = hipMallocAsync(&orderScratch, LengthOrderScratchBytes(n), stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(length order)");
		bool serp = false;
		if (int rc = BuildLengthOrder(offsets, n, orderScratch, stream, &p.order, &serp)) {
			(void)hipFreeAsync(orderScratch, stream);
			return rc;
		}
		p.serpentine = serp ? 1u : 0u;
	}
	void* list = nullptr;
	e = hipMallocAsync(&list, (size_t(n) + 1) * 4, stream);
	if (e == hipSuccess)
		e = hipMemsetAsync(list, 0, 4, stream);
	if (e == hipSuccess) {
		p.overflow = static_cast<uint32_t*>(list);
		const uint32_t rowLds = uint32_t(((size_t(p.states + 1) * (p.letters + 1) * 8 + 15) & ~size_t(15)) + size_t(lactCount) * 2 * nreg * 4 + 512);
		p.spreadWaves = n <= uint64_t(cus) * 1024 ? 1u : 0u;
		const unsigned rblocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>(p.spreadWaves ? (n + 63) / 64 : (n + 1023) / 1024, uint64_t(cus))));
		switch (nreg) {
		case 1: LaunchRow<1, true>(p, 2, rblocks, rowLds, stream, &e); break;
		case 2: LaunchRow<2, true>(p, 2, rblocks, rowLds, stream, &e); break;
		default: LaunchRow<4, true>(p, 2, rblocks, rowLds, stream, &e); break;
		}
	}
	if (orderScratch)
		(void)hipFreeAsync(orderScratch, stream);
	if (e != hipSuccess) {
		if (list)
			(void)hipFreeAsync(list, stream);
		return HipFail(e, "half-final row kernel");
	}
	NoteKernel("half_final_rows");
	*overflow = static_cast<uint32_t*>(list);
	*done = true;
	return PIRE_HIP_OK;
}

}  // namespace pirehip

using namespace pirehip;

extern "C" {

int pire_hip_counting_table_create(const void* save_blob, size_t len, pire_hip_counting_table** out)
try {
	if (!out) {
		SetError("null out pointer");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	std::unique_ptr<pire_hip_counting_table> t(new (std::nothrow) pire_hip_counting_table);
	if (!t) {
		SetError("out of memory");
		return PIRE_HIP_ENOMEM;
	}
	if (int rc = BuildCountingHost(save_blob, len, &t->host))
		return rc;
	*out = t.release();
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

void pire_hip_counting_table_destroy(pire_hip_counting_table* t)
{
	if (!t)
		return;
	int cur = -1;
	(void)hipGetDevice(&cur);
	for (int k = 0; k < kMaxDevices; ++k)
		if (t->devs[k].device >= 0) {
			(void)hipSetDevice(k);
			FreeCountingDevice(&t->devs[k]);
		}
	for (int k = 0; k < kMaxDevices; ++k)
		if (t->captureInfoDev[k]) {
			(void)hipSetDevice(k);
			(void)hipFree(t->captureInfoDev[k]);
		}
	if (t->captureTable)
		pire_hip_table_destroy(t->captureTable.release());
	if (cur >= 0)
		(void)hipSetDevice(cur);
	delete t;
}

int pire_hip_counting_table_get_info(const pire_hip_counting_table* t, pire_hip_counting_info* out)
try {
	if (!t || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	memset(out, 0, sizeof(*out));
	out->states = t->host.states;
	out->letters = t->host.letters;
	out->regexps = t->host.regexps;
	out->initial = t->host.initial;
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_counting_table_forms(const pire_hip_counting_table* t, uint32_t out[8])
try {
	if (!t || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	const CountingHost& h = t->host;
	memset(out, 0, 8 * sizeof(uint32_t));
	out[0] = h.dense.empty() ? 0 : h.nreg;
	out[1] = out[0] ? uint32_t(size_t(h.states) * 512 + 256 * 2 * h.nreg * 4) : 0;
	out[2] = (out[0] && h.nreg <= 4 && h.states <= kCountingRowStates) ? 1 : 0;
	out[3] = out[2] ? uint32_t(((size_t(h.states + 1) * kCountingRowPitch + 15) & ~size_t(15)) + 256 * 2 * h.nreg * 4) : 0;
	out[4] = h.lrows.empty() ? 0 : h.lnreg;
	if (out[4]) {
		const uint32_t count = uint32_t(h.lactWords.size() / (2 * h.lnreg));
		out[5] = uint32_t(((size_t(h.states + 1) * (h.letters + 1) * 8 + 15) & ~size_t(15)) + size_t(count) * 2 * h.lnreg * 4 + 512);
		out[6] = count - 1;
	}
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();
}

// ---- first-use self-tests (selftest.h): the counting scanners' and the capturing scanner's kernels -------------------------
namespace {

struct CountOwnStream {
	hipStream_t s = nullptr;
	CountOwnStream() { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
	~CountOwnStream()
	{
		if (s) {
			(void)hipStreamSynchronize(s);
			(void)hipStreamDestroy(s);
		}
	}
};

bool CountingTested(pire_hip_counting_table* t, uint32_t bit, int* dev)
{
	*dev = -1;
	if (hipGetDevice(dev) != hipSuccess || *dev < 0 || *dev >= pirehip::kMaxDevices)
		return true;
	return (t->selfTested[*dev].load(std::memory_order_relaxed) & (1u << bit)) != 0;
}

// the host's copy of the scanner: state | action << 32 per (state, letter) -- count.h / loaded.h as ingested
struct CountWalk {
	const pirehip::CountingHost& h;
	uint64_t Trans(uint32_t st, uint32_t ch) const { return h.trans[size_t(st) * h.letters + h.letterOf[ch]]; }
};

pirehip::KnownBatch CountingBatch(const CountWalk& w, uint32_t n, uint32_t maxLen, uint32_t start, uint64_t seed)
{
	return pirehip::MakeKnownBatch(n, maxLen, start, seed ^ (uint64_t(w.h.states) << 24) ^ (uint64_t(w.h.letters) << 8),
	                               [&](uint32_t st, uint32_t ch) { return uint32_t(w.Trans(st, ch)); }, [](uint32_t) { return false; });
}

// CountingScanner / AdvancedCountingScanner / NoGlueLimitCountingScanner on the host: count.h:175-192 (PerformIncrement,
// PerformReset), 251-257 / 287-295 (the order of the two), 306-325 + 404-437 (NoGlueLimit), Result = max(current, total) 206
void HostCount(const CountWalk& w, int kind, uint32_t flags, const uint8_t* b, const uint8_t* e, uint32_t* outIdx, uint32_t* results)
{
	using namespace pirehip;
	const uint32_t R = w.h.regexps;
	std::vector<uint32_t> cur(std::max<uint32_t>(R, 1), 0), tot(std::max<uint32_t>(R, 1), 0);
	uint32_t updated = 0, st = w.h.initial;
	constexpr uint32_t kInc = (1u << kMaxReCount) - 1u, kReset = kInc << kMaxReCount;
	auto increment = [&](uint32_t a) {
		for (uint32_t r = 0; r < R && r < kMaxReCount; ++r)
			cur[r] += (a >> r) & 1u;
		updated |= a << kMaxReCount;
	};
	auto reset = [&](uint32_t a) {
		const uint32_t m = a & updated;
		if (!m)
			return;
		for (uint32_t r = 0; r < R && r < kMaxReCount; ++r)
			if (((m >> (kMaxReCount + r)) & 1u) && cur[r]) {
				tot[r] = std::max(tot[r], cur[r]);
				cur[r] = 0;
			}
		updated &= ~m;
	};
	auto take = [&](uint32_t a) {
		if (kind == PIRE_HIP_COUNTING_NOGLUELIMIT) {
			if (!w.h.actions.empty()) {
				const uint32_t* act = w.h.actions.data() + a;
				for (uint32_t k = *act++; k--;)
					cur[*act++] = 0;
				for (uint32_t k = *act++; k--;) {
					const uint32_t id = *act++;
					++cur[id];
					tot[id] = std::max(tot[id], cur[id]);
				}
			} else {
				if (a & 2u)
					cur[0] = 0;
				if (a & 1u) {
					++cur[0];
					tot[0] = std::max(tot[0], cur[0]);
				}
			}
		} else if (kind == PIRE_HIP_COUNTING_ADVANCED) {
			if (a & kReset)
				reset(a);
			if (a & kInc)
				increment(a);
		} else {
			if (a & kInc)
				increment(a);
			if (a & kReset)
				reset(a);
		}
	};
	auto step = [&](uint32_t ch) {
		const uint64_t x = w.Trans(st, ch);
		st = uint32_t(x);
		if (uint32_t(x >> 32))   // (0 = no action; a NoGlueLimit action is an offset into the lists, whose word 0 is their length)
			take(uint32_t(x >> 32));
	};
	if (flags & PIRE_HIP_RUN_BEGIN)
		step(kBeginMark);
	for (const uint8_t* p = b; p != e; ++p)
		step(*p);
	if (flags & PIRE_HIP_RUN_END)
		step(kEndMark);
	*outIdx = st;
	for (uint32_t r = 0; r < R; ++r)
		results[r] = std::max(cur[r], tot[r]);
}

int SelfTestCounting(pire_hip_counting_table* t, int kind, uint32_t flags, hipStream_t stream)
{
	using namespace pirehip;
	uint32_t mode = 0;
	int dev = -1;
	if (CountingTested(t, 0, &dev) || !EntrySelfTestDue(stream, &mode) || !t->host.states)
		return PIRE_HIP_OK;
	const CountWalk w{t->host};
	const uint32_t R = std::max<uint32_t>(t->host.regexps, 1);
	flags &= PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END;
	const uint32_t start = (flags & PIRE_HIP_RUN_BEGIN) ? uint32_t(w.Trans(t->host.initial, kBeginMark)) : t->host.initial;
	const KnownBatch kb = CountingBatch(w, 320, 200, start, 11);
	std::vector<uint32_t> wantIdx(kb.n), gotIdx(kb.n), wantRes(size_t(kb.n) * R, 0), gotRes(size_t(kb.n) * R);
	for (uint32_t i = 0; i < kb.n; ++i)
		HostCount(w, kind, flags, kb.text.data() + kb.offsets[i], kb.text.data() + kb.offsets[i + 1], &wantIdx[i], &wantRes[size_t(i) * R]);
	if (mode == 2)
		wantIdx[kb.n / 2] ^= 1;
	std::vector<std::function<void(pire_hip_config&)>> variants;
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 2; });   // whole lines per lane: rows by byte / by letter where the table has them
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; });   // 16 bytes at a time: packed 16-bit counters / the 32-bit kernel
	CountOwnStream own;
	uint32_t extra = 0;   // the third pass: PIRE_HIP_RUN_GENERIC = the 32-bit kernel alone
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; });
	size_t pass = 0;
	const int rc = RunSelfTestVariants(variants, [&]() -> int {
		extra = pass++ == 2 ? PIRE_HIP_RUN_GENERIC : 0u;
		std::fill(gotIdx.begin(), gotIdx.end(), ~0u);
		std::fill(gotRes.begin(), gotRes.end(), ~0u);
		const int r = pire_hip_counting_run(t, kind, kb.text.data(), kb.offsets.data(), kb.n, flags | extra, gotIdx.data(), gotRes.data(), own.s);
		if (r != PIRE_HIP_OK)
			return r;
		for (uint32_t i = 0; i < kb.n; ++i) {
			if (gotIdx[i] != wantIdx[i])
				return SelfTestMismatch("the counting scanner", i, "state " + std::to_string(gotIdx[i]), "state " + std::to_string(wantIdx[i]));
			for (uint32_t r2 = 0; r2 < t->host.regexps; ++r2)
				if (gotRes[size_t(i) * R + r2] != wantRes[size_t(i) * R + r2])
					return SelfTestMismatch("the counting scanner", i, "count " + std::to_string(gotRes[size_t(i) * R + r2]) + " for regexp " + std::to_string(r2),
					                        std::to_string(wantRes[size_t(i) * R + r2]));
		}
		return PIRE_HIP_OK;
	});
	if (rc != PIRE_HIP_OK)
		return rc;
	t->selfTested[dev].fetch_or(1u);
	return PIRE_HIP_OK;
}

// CapturingScanner on the host: capture.h:89-116 (the step counter, BeginCapture = 1 / EndCapture = 2 taken while nothing is
// captured yet), Final from the tags (capture.h:134)
int SelfTestCapture(pire_hip_counting_table* t, uint32_t flags, hipStream_t stream)
{
	using namespace pirehip;
	uint32_t mode = 0;
	int dev = -1;
	if (CountingTested(t, 1, &dev) || !EntrySelfTestDue(stream, &mode) || !t->host.states || t->host.type != 4)
		return PIRE_HIP_OK;
	const CountWalk w{t->host};
	flags &= PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END;
	const uint32_t start = (flags & PIRE_HIP_RUN_BEGIN) ? uint32_t(w.Trans(t->host.initial, kBeginMark)) : t->host.initial;
	const KnownBatch kb = CountingBatch(w, 320, 200, start, 12);
	std::vector<uint32_t> wantIdx(kb.n), gotIdx(kb.n);
	std::vector<uint8_t> wantFin(kb.n), gotFin(kb.n);
	std::vector<int64_t> wantB(kb.n), wantE(kb.n), gotB(kb.n), gotE(kb.n);
	for (uint32_t i = 0; i < kb.n; ++i) {
		uint32_t st = t->host.initial;
		int64_t begin = -1, end = -1, counter = 0;
		auto step = [&](uint32_t ch) {
			const uint64_t x = w.Trans(st, ch);
			const bool captured = begin >= 0 && end >= 0;
			st = uint32_t(x);
			++counter;
			const uint32_t a = uint32_t(x >> 32);
			if ((a & 1u) && !captured)
				begin = counter - 1;
			else if ((a & 2u) && !captured)
				end = counter - 1;
		};
		if (flags & PIRE_HIP_RUN_BEGIN)
			step(kBeginMark);
		for (uint64_t k = kb.offsets[i]; k < kb.offsets[i + 1]; ++k)
			step(kb.text[k]);
		if (flags & PIRE_HIP_RUN_END)
			step(kEndMark);
		wantIdx[i] = st;
		wantFin[i] = !t->host.tags.empty() && (t->host.tags[st] & 1u) ? 1 : 0;
		wantB[i] = begin;
		wantE[i] = end;
	}
	if (mode == 2)
		wantIdx[kb.n / 2] ^= 1;
	std::vector<std::function<void(pire_hip_config&)>> variants;
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 2; c.ragged_act_always = 1; c.no_ragged_act = 0; });   // the ragged kernel with actions (>= 256 strings) ...
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 2; c.no_ragged_act = 1; });   // ... whole lines per lane (CaptureRowKernel) ...
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; c.no_ragged_act = 1; });   // ... dense rows, 16 bytes at a time ...
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; c.no_ragged_act = 1; });   // ... and (PIRE_HIP_RUN_GENERIC) letter + transition
	CountOwnStream own;
	size_t pass = 0;
	const int rc = RunSelfTestVariants(variants, [&]() -> int {
		const uint32_t extra = pass++ == 3 ? PIRE_HIP_RUN_GENERIC : 0u;
		std::fill(gotIdx.begin(), gotIdx.end(), ~0u);
		std::fill(gotB.begin(), gotB.end(), int64_t(-77));
		const int r = pire_hip_capture_run(t, kb.text.data(), kb.offsets.data(), kb.n, flags | extra, gotIdx.data(), gotFin.data(), gotB.data(),
		                                   gotE.data(), own.s);
		if (r != PIRE_HIP_OK)
			return r;
		for (uint32_t i = 0; i < kb.n; ++i)
			if (gotIdx[i] != wantIdx[i] || gotFin[i] != wantFin[i] || gotB[i] != wantB[i] || gotE[i] != wantE[i])
				return SelfTestMismatch("the capturing scanner", i,
				                        "state " + std::to_string(gotIdx[i]) + " captured [" + std::to_string(gotB[i]) + ", " + std::to_string(gotE[i]) + ")",
				                        "state " + std::to_string(wantIdx[i]) + " captured [" + std::to_string(wantB[i]) + ", " + std::to_string(wantE[i]) + ")");
		return PIRE_HIP_OK;
	});
	if (rc != PIRE_HIP_OK)
		return rc;
	t->selfTested[dev].fetch_or(2u);
	return PIRE_HIP_OK;
}

}  // namespace

int pire_hip_counting_run(pire_hip_counting_table* t, int kind, const void* text, const uint64_t* offsets, uint64_t n,
                          uint32_t flags, uint32_t* out_state_idx, uint32_t* out_results, void* streamPtr)
try {
	if (!t || (n && (!offsets || !out_results)) || (kind != PIRE_HIP_COUNTING_BASIC && kind != PIRE_HIP_COUNTING_ADVANCED && kind != PIRE_HIP_COUNTING_NOGLUELIMIT)) {
		SetError("bad argument");
		return PIRE_HIP_EINVAL;
	}
	// the serialised form decides between the classes: a type-5 blob is a NoGlueLimitCountingScanner and nothing
	// else; a LoadedScanner blob run as NOGLUELIMIT is the reference's "AdvancedScannerCompatibilityMode"
	// (count.cpp:1036-1039), i.e. AdvancedCountingScanner semantics
	if ((t->host.type == 5) != (kind == PIRE_HIP_COUNTING_NOGLUELIMIT)) {
		if (t->host.type == 4 && kind == PIRE_HIP_COUNTING_NOGLUELIMIT) {
			kind = PIRE_HIP_COUNTING_ADVANCED;
		} else {
			SetError("this table was serialised by a NoGlueLimitCountingScanner: run it with PIRE_HIP_COUNTING_NOGLUELIMIT");
			return PIRE_HIP_EINVAL;
		}
	}
	if (n == 0)
		return PIRE_HIP_OK;
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	if (int rc = SelfTestCounting(t, kind, flags, stream))   // first use on this device: every kernel of this entry point, known answers (selftest.h)
		return rc;
	CountingDevice image;
	if (int rc = UploadCounting(t, &image))
		return rc;
	CountingParams p;
	memset(&p, 0, sizeof(p));
	p.actions = image.actions;
	// more than 16 regexps: per-string `current` rows in a temporary device array (freed after the stream is drained)
	void* scratch = nullptr;
	// stream-ordered: allocated and freed on the call's stream (the free is ordered after the kernel that uses it)
	struct ScratchGuard {
		void*& q;
		hipStream_t s;
		~ScratchGuard()
		{
			if (q)
				(void)hipFreeAsync(q, s);
		}
	} scratchGuard{scratch, stream};
	if (kind == PIRE_HIP_COUNTING_NOGLUELIMIT && t->host.regexps > kMaxReCount) {
		hipError_t se = hipMallocAsync(&scratch, size_t(n) * t->host.regexps * 4, stream);
		if (se != hipSuccess)
			return HipFail(se, "hipMallocAsync(counting scratch)");
		p.scratch = static_cast<uint32_t*>(scratch);
	}
	p.letterOf = image.letterOf;
	p.trans = image.trans;
	p.dense = image.dense;
	p.denseMarks = image.denseMarks;
	p.actWords = image.actWords;
	p.lrows = image.lrows;
	p.lactWords = image.lactWords;
	p.lnreg = (flags & PIRE_HIP_RUN_GENERIC) ? 0 : t->host.lnreg;
	p.lactCount = t->host.lnreg ? uint32_t(t->host.lactWords.size() / (2 * t->host.lnreg)) : 0;
	p.states = t->host.states;
	p.letters = t->host.letters;
	p.regexps = t->host.regexps;
	p.initial = t->host.initial;
	p.flags = flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END);
	p.n = n;
	// PIRE_HIP_RUN_GENERIC keeps the 32-bit kernel alone (the tests compare the two)
	const uint32_t nreg = (flags & PIRE_HIP_RUN_GENERIC) ? 0 : t->host.nreg;
	const uint32_t R = std::max<uint32_t>(t->host.regexps, 1);
	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		p.outIdx = out_state_idx;
		p.outResults = out_results;
		return LaunchCounting(p, kind, stream, nreg);
	}
	for (uint64_t i = 0; i < n; ++i)
		if (offsets[i] > offsets[i + 1]) {
			SetError("offsets must be non-decreasing");
			return PIRE_HIP_EINVAL;
		}
	const uint64_t textBytes = offsets[n];
	if (!text && textBytes) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	Staging stage(stream);
	const uint8_t* dText = nullptr;
	const uint64_t* dOffs = nullptr;
	void *dIdx = nullptr, *dRes = nullptr;
	int rc;
	if ((rc = stage.In(static_cast<const uint8_t*>(text), size_t(textBytes), &dText, stream)) ||
	    (rc = stage.In(offsets, size_t(n + 1), &dOffs, stream)) || (rc = stage.Alloc(&dIdx, n * 4)) ||
	    (rc = stage.Alloc(&dRes, n * R * 4)))
		return rc;
	p.text = dText;
	p.offsets = dOffs;
	p.outIdx = static_cast<uint32_t*>(dIdx);
	p.outResults = static_cast<uint32_t*>(dRes);
	if ((rc = LaunchCounting(p, kind, stream, nreg)))
		return rc;
	rc = stage.Out(out_state_idx, dIdx, n * 4);
	if (!rc && t->host.regexps)
		rc = stage.Out(out_results, dRes, n * t->host.regexps * 4);
	return rc ? rc : stage.Finish();
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

}  // extern "C"

namespace pirehip {
namespace {

// The capture scanner as a table the scan kernels understand: actions move from the transitions to the states.
// Expanded state = (state, action of the transition that entered it); the walk of the expanded automaton visits
// (state_i, action_i) exactly where the reference's walk is in state_i having just returned action_i from Next().
// Breadth-first from (initial, 0) over the letters, so only reachable pairs exist (<= 4 x the states).  States entered
// with an action carry kFinal -- the flag the ragged kernel with actions looks for -- the real Final tag travels in
// the info word.  Returns false when the expansion is not worth it / not possible (then the one-string-per-lane kernel
// stays): more than 60 000 expanded states.
bool BuildCaptureTable(const CountingHost& h, HostTable* out, std::vector<uint32_t>* info)
{
	const uint32_t C = h.letters;
	std::vector<uint32_t> idOf(size_t(h.states) * 4, UINT32_MAX), queue;
	auto get = [&](uint32_t st, uint32_t a) {
		uint32_t& id = idOf[size_t(st) * 4 + (a & 3u)];
		if (id == UINT32_MAX) {
			id = uint32_t(queue.size());
			queue.push_back(st * 4 + (a & 3u));
		}
		return id;
	};
	(void)get(h.initial, 0);
	std::vector<uint32_t> next;
	for (size_t head = 0; head < queue.size(); ++head) {
		if (queue.size() > 60000)
			return false;
		const uint32_t st = queue[head] >> 2;
		for (uint32_t c = 0; c < C; ++c) {
			const uint64_t x = h.trans[size_t(st) * C + c];
			next.push_back(get(uint32_t(x), uint32_t(x >> 32)));
		}
	}
	HostTable& t = *out;
	t = HostTable();
	t.states = uint32_t(queue.size());
	t.letters = C;
	t.regexps = 1;
	t.initial = 0;
	t.scannerType = 4;
	t.cls.assign(kMaxChar, 0);
	for (uint32_t c = 0; c < kMaxChar; ++c)
		t.cls[c] = h.letterOf[c];
	t.next = next;
	t.flags.assign(t.states, 0);
	info->assign(t.states, 0);
	for (uint32_t e = 0; e < t.states; ++e) {
		const uint32_t st = queue[e] >> 2, a = queue[e] & 3u;
		bool absorbing = true;
		for (uint32_t c = 0; c < C; ++c)
			absorbing = absorbing && t.next[size_t(e) * C + c] == e;
		t.flags[e] = uint8_t((a ? kFinal : 0) | (absorbing ? kAbsorbing : 0));
		(*info)[e] = (st << 8) | ((h.tags[st] & 1u) << 2) | a;
	}
	t.acceptOff.assign(size_t(t.states) + 1, 0);   // no AcceptedRegexps lists: the counters of pire_hip_run are not used
	ChooseHotAndPermuteExported(t);
	return true;
}

// The capture table and its info array on the current device (built / uploaded when first needed).
int EnsureCaptureTable(pire_hip_counting_table* t, pire_hip_table** table, const uint32_t** infoDev)
{
	std::lock_guard<std::mutex> lock(t->uploadMutex);
	*table = nullptr;
	if (!t->captureTried) {
		t->captureTried = true;
		if (t->host.states < (1u << 22) && t->host.letters <= 127) {
			std::unique_ptr<pire_hip_table> ct(new pire_hip_table);
			if (BuildCaptureTable(t->host, &ct->host, &t->captureInfo))
				t->captureTable = std::move(ct);
		}
	}
	if (!t->captureTable)
		return PIRE_HIP_OK;
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess || dev < 0 || dev >= kMaxDevices)
		return HipFail(e == hipSuccess ? hipErrorInvalidDevice : e, "hipGetDevice");
	if (!t->captureInfoDev[dev]) {
		e = hipMalloc(reinterpret_cast<void**>(&t->captureInfoDev[dev]), t->captureInfo.size() * 4);
		if (e == hipSuccess)
			e = hipMemcpy(t->captureInfoDev[dev], t->captureInfo.data(), t->captureInfo.size() * 4, hipMemcpyHostToDevice);
		if (e != hipSuccess)
			return HipFail(e, "hipMalloc(capture info)");
	}
	*table = t->captureTable.get();
	*infoDev = t->captureInfoDev[dev];
	return PIRE_HIP_OK;
}

}  // namespace
}  // namespace pirehip

extern "C" {

int pire_hip_capture_run(pire_hip_counting_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                         uint32_t* out_state_idx, uint8_t* out_final, int64_t* out_begin, int64_t* out_end, void* streamPtr)
try {
	if (!t || (n && (!offsets || !out_begin || !out_end))) {
		SetError("bad argument");
		return PIRE_HIP_EINVAL;
	}
	if (t->host.type != 4) {
		SetError("a CapturingScanner serialises as a LoadedScanner (type 4) table");
		return PIRE_HIP_EINVAL;
	}
	if (n == 0)
		return PIRE_HIP_OK;
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	if (int rc = SelfTestCapture(t, flags, stream))   // first use on this device (selftest.h)
		return rc;
	CountingDevice image;
	if (int rc = UploadCounting(t, &image))
		return rc;
	CountingParams p;
	memset(&p, 0, sizeof(p));
	p.letterOf = image.letterOf;
	p.trans = image.trans;
	p.tags = image.tags;
	p.states = t->host.states;
	p.letters = t->host.letters;
	p.regexps = t->host.regexps;
	p.initial = t->host.initial;
	p.flags = flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END);
	p.n = n;
	int dev = 0, cus = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e == hipSuccess)
		e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (e != hipSuccess)
		return HipFail(e, "device query");
	const uint64_t tableBytes = uint64_t(p.states) * p.letters * 8;
	p.transInLds = tableBytes <= 60 * 1024 ? 1 : 0;
	const uint32_t ldsBytes = 272 + (p.transInLds ? uint32_t(tableBytes) : 0);
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, uint64_t(cus) * 8)));
	e = SetDynamicLds(reinterpret_cast<const void*>(CaptureKernel), uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	// one string per lane: the dense-row kernel when the table has the dense form (<= 255 states), else letter + transition
	p.dense = image.dense;
	p.denseMarks = image.denseMarks;
	void* orderScratch = nullptr;   // freed on the stream, behind the kernel that reads it
	struct OrderGuard {
		void*& q;
		hipStream_t s;
		~OrderGuard()
		{
			if (q)
				(void)hipFreeAsync(q, s);
		}
	} orderGuard{orderScratch, stream};
	auto launchPerLane = [&]() -> int {
		hipError_t le;
		// (order.hip, measured on the capture walks: 1 025-1 045 GB/s in the caller's order, 941-977 by length -- the walk
		// has little to win, its text reads lose their neighbours; `capture_by_length` keeps the A/B)
		// whole text lines per lane (CaptureRowKernel) where the table leaves room for 2 KB rows and the batch fills the one
		// block of 16 waves a CU then holds; pire_hip_config.counting_variant as for the counting scanners.  That kernel
		// does take its strings by length: a line is a line wherever the neighbouring lanes read.
		const int variant = GetConfig().counting_variant;
		const bool rows = p.dense && !(flags & PIRE_HIP_RUN_GENERIC) && p.states <= kCaptureRowStates && variant != 1 &&
		                  (variant == 2 || p.n >= uint64_t(cus) * 256);
		if (p.offsets && LengthOrderWanted(p.n) && (GetConfig().capture_by_length || (rows && !GetConfig().no_length_order))) {
			le = hipMallocAsync(&orderScratch, LengthOrderScratchBytes(p.n), stream);
			if (le != hipSuccess)
				return HipFail(le, "hipMallocAsync(length order)");
			bool serp = false;
			if (int rc = BuildLengthOrder(p.offsets, p.n, orderScratch, stream, &p.order, &serp))
				return rc;
			p.serpentine = serp ? 1u : 0u;
		}
		if (rows) {
			const uint32_t rowLds = uint32_t(size_t(p.states + 1) * kCaptureRowPitch);
			le = SetDynamicLds(reinterpret_cast<const void*>(CaptureRowKernel), rowLds);
			if (le != hipSuccess)
				return HipFail(le, "hipFuncSetAttribute(LDS)");
			NoteKernel("capture_rows");
			p.spreadWaves = n <= uint64_t(cus) * 1024 ? 1u : 0u;
			const unsigned rblocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>(p.spreadWaves ? (n + 63) / 64 : (n + 1023) / 1024, uint64_t(cus))));
			hipLaunchKernelGGL(CaptureRowKernel, dim3(rblocks), dim3(1024), rowLds, stream, p);
		} else if (p.dense && !(flags & PIRE_HIP_RUN_GENERIC)) {
			const uint32_t denseLds = p.states * 512;
			le = SetDynamicLds(reinterpret_cast<const void*>(CaptureDenseKernel), uint32_t(denseLds));
			if (le != hipSuccess)
				return HipFail(le, "hipFuncSetAttribute(LDS)");
			NoteKernel("capture_dense");
			// (a table of more than 40 KB: blocks of 16 waves, as LaunchPacked)
			const unsigned dthreads = denseLds > 40 * 1024 ? 1024 : 256;
			const uint64_t perCu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / dthreads, (160 * 1024) / std::max<uint32_t>(denseLds, 1)));
			const unsigned dblocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((n + dthreads - 1) / dthreads, uint64_t(cus) * perCu)));
			hipLaunchKernelGGL(CaptureDenseKernel, dim3(dblocks), dim3(dthreads), denseLds, stream, p);
		} else {
			NoteKernel("capture");
			hipLaunchKernelGGL(CaptureKernel, dim3(blocks), dim3(256), ldsBytes, stream, p);
		}
		le = hipGetLastError();
		return le == hipSuccess ? PIRE_HIP_OK : HipFail(le, "capture kernel launch");
	};
	// Batches of >= 256 strings ride the ragged kernel with actions (the expanded table of BuildCaptureTable):
	// 2-3 x the one-string-per-lane kernel below, which keeps the small batches and PIRE_HIP_RUN_GENERIC.
	TableUse ctUse;   // the expanded capture table's numbering stays put until this call returns
	auto ragged = [&](const uint8_t* dText, const uint64_t* dOffs, uint32_t* dIdx, uint8_t* dFin, long long* dB,
	                  long long* dE, bool* done) -> int {
		*done = false;
		if ((flags & PIRE_HIP_RUN_GENERIC) || n < 256 || n >= (1ull << 32) - (1ull << 16) || GetConfig().no_ragged_act)
			return PIRE_HIP_OK;
		pire_hip_table* ct = nullptr;
		const uint32_t* infoDev = nullptr;
		if (int rc = EnsureCaptureTable(t, &ct, &infoDev))
			return rc;
		if (!ct)
			return PIRE_HIP_OK;
		ScanParams sp;
		if (int rc = PrepareScanParams(ct, &sp, flags & PIRE_HIP_RUN_BEGIN, &ctUse, /*wantDist=*/true, (flags & PIRE_HIP_RUN_ON_DEVICE) != 0))   // startPerm = Initialize [+ BeginMark]
			return rc;
		// Scanners whose walk is in an action state most of the time gain nothing from looking for the chunks that have
		// one: =(\d+)[^\d] re-arms BeginCapture on every byte in front of the match (99.5 % of the steps on the benchmark
		// text, measured), (/to-match-with) on 2 %, google_id\s*=... on none.  The share of the byte model's visits that
		// fall on action states (the expanded table's "Final" share, table.cpp) decides; either kernel is exact.
		if (sp.finalShare > 0.002f && !GetConfig().ragged_act_always)   // (2 % of the steps = a third of the chunks re-walked: 740 GB/s, the dense-row kernel does 1 000)
			return PIRE_HIP_OK;
		sp.flags = flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END);
		sp.n = n;
		sp.text = dText;
		sp.offsets = dOffs;
		sp.outIdx = dIdx;
		sp.outFinal = dFin;
		sp.actDist = sp.distFinalPerm;   // of the image in sp, taken under the table's lock
		*done = true;
		return LaunchRaggedCapture(sp, TakeWorkSlot(ct, sp), infoDev, dB, dE, stream);
	};
	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		p.outIdx = out_state_idx;
		p.outFinal = out_final;
		p.outBegin = reinterpret_cast<long long*>(out_begin);
		p.outEnd = reinterpret_cast<long long*>(out_end);
		bool done = false;
		if (int rc = ragged(p.text, p.offsets, p.outIdx, p.outFinal, p.outBegin, p.outEnd, &done))
			return rc;
		if (done)
			return PIRE_HIP_OK;
		return launchPerLane();
	}
	for (uint64_t i = 0; i < n; ++i)
		if (offsets[i] > offsets[i + 1]) {
			SetError("offsets must be non-decreasing");
			return PIRE_HIP_EINVAL;
		}
	const uint64_t textBytes = offsets[n];
	if (!text && textBytes) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	Staging stage(stream);
	const uint8_t* dText = nullptr;
	const uint64_t* dOffs = nullptr;
	void *dIdx = nullptr, *dFin = nullptr, *dB = nullptr, *dE = nullptr;
	int rc;
	if ((rc = stage.In(static_cast<const uint8_t*>(text), size_t(textBytes), &dText, stream)) ||
	    (rc = stage.In(offsets, size_t(n + 1), &dOffs, stream)) || (rc = stage.Alloc(&dIdx, n * 4)) ||
	    (rc = stage.Alloc(&dFin, n)) || (rc = stage.Alloc(&dB, n * 8)) || (rc = stage.Alloc(&dE, n * 8)))
		return rc;
	p.text = dText;
	p.offsets = dOffs;
	p.outIdx = static_cast<uint32_t*>(dIdx);
	p.outFinal = static_cast<uint8_t*>(dFin);
	p.outBegin = static_cast<long long*>(dB);
	p.outEnd = static_cast<long long*>(dE);
	bool done = false;
	if ((rc = ragged(p.text, p.offsets, p.outIdx, p.outFinal, p.outBegin, p.outEnd, &done)))
		return rc;
	if (!done && (rc = launchPerLane()))
		return rc;
	rc = stage.Out(out_state_idx, dIdx, n * 4);
	if (!rc)
		rc = stage.Out(out_final, dFin, n);
	if (!rc)
		rc = stage.Out(out_begin, dB, n * 8);
	if (!rc)
		rc = stage.Out(out_end, dE, n * 8);
	return rc ? rc : stage.Finish();
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

}  // extern "C"
