// Pire::SlowScanner on the GPU (BASELINE config 5b): NFA simulation, one string per lane, the state is the SET of
// active NFA states held as a bitset in registers.
//
// Reference: /root/reference/pire/scanners/slow.h
//   State{vector<unsigned> states; BitSet flags}   63-74     (the set is what Next/Final are defined on)
//   NextTranslated(cur, next, l)                    103-130   next = union of jump lists of every active state
//   Final(s) = any active state is final            152-158
//   Run<SlowScanner>                                436-451   Step per byte, double-buffered
//   serialised form (Save)                          scanner_io.cpp:71-111
// Device form: for every (state, letter) a K-word bitmask of the jump list (the CSR lists of the reference turned
// into rows), kept in LDS when it fits; a step is  next = OR over active s of mask[s][letter].  Work per byte is
// O(active states * K), as in the reference -- this is the scanner for automata whose DFA would not fit anywhere.
// That form holds up to 256 NFA states (8 words per lane).  Beyond it (SlowWideKernel, below) the reference's own
// sparse form is kept -- the CSR jump lists -- and ONE WAVE walks one string: the set is a bitset of any size in LDS
// (or in device memory when it does not fit), the wave finds the active states word by word and its lanes share the
// targets of every active state's jump list.  No limit on the number of states but the scanner's own (2^20 here).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "internal.h"

namespace pirehip {

struct SlowHost {
	uint32_t states = 0, letters = 0, start = 0, words = 0;
	bool empty = false;
	std::vector<uint8_t> letterOf;     // [264] letter of each Char (m_letters, slow.h:349)
	std::vector<uint32_t> masks;       // [states*letters][words]
	std::vector<uint32_t> single;      // [states*letters][2] the (up to two) targets of the jump list, or kSlowMulti
	std::vector<uint32_t> finals;      // [words] bitset of final states
	// the reference's sparse form (m_jumpPos / m_jumps, slow.h:351-352), kept for scanners of more than 256 states
	std::vector<uint32_t> jumpPos;     // [states*letters + 1]
	std::vector<uint32_t> jumps;       // targets
	bool wide = false;                 // more than 256 states: masks / single are not built
	// wide automata of fewer than 2^15 (state, letter) rows: the list form's table, [(states + 1) * letters] words
	// (first / second target as row offsets, layout at SlowListKernel); empty otherwise
	std::vector<uint32_t> pair16;
};

struct SlowDevice {
	int device = -1;
	uint8_t* letterOf = nullptr;
	uint32_t* masks = nullptr;
	uint32_t* single = nullptr;
	uint32_t* finals = nullptr;
	uint32_t* jumpPos = nullptr;
	uint32_t* jumps = nullptr;
	uint32_t* pair16 = nullptr;
};

constexpr uint32_t kSlowNone = 0xFFFFFFFFu;    // no target / empty list slot
constexpr uint32_t kSlowMulti = 0xFFFFFFFEu;   // more than two targets: the step takes the bitset form

}  // namespace pirehip

struct pire_hip_slow_table {
	pirehip::SlowHost host;
	pirehip::SlowDevice devs[pirehip::kMaxDevices];   // one image per HIP device, as in pire_hip_table
	std::mutex uploadMutex;
};

namespace pirehip {

struct SlowParams {
	const uint8_t* letterOf;
	const uint32_t* masks;
	const uint32_t* single;
	const uint32_t* finals;
	const uint32_t* jumpPos;   // wide form
	const uint32_t* jumps;
	const uint32_t* pair16;    // wide form, list kernel
	uint32_t* overflow;        // list kernel -> wide kernel: [0] = count, [1 ..] = the strings whose sets outgrew the list
	const uint32_t* order;     // list kernel, nullable: string k of the launch is order[k] (order.hip: by length class)
	uint32_t serpentine;
	uint32_t* scratch;         // wide form, sets that do not fit the LDS: [waves][2][words]
	uint32_t states, letters, start, words, flags, masksInLds, singleInLds;
	const uint8_t* text;
	const uint64_t* offsets;   // nullable: strided
	uint64_t n, len, stride;
	uint8_t* outFinal;
	uint32_t* outBits;
	unsigned long long* outCounts;
};

template <int K>
__device__ __forceinline__ void SlowStep(const uint32_t* masks, uint32_t letters, uint32_t (&cur)[K], uint32_t letter)
{
	uint32_t next[K];
#pragma unroll
	for (int k = 0; k < K; ++k)
		next[k] = 0;
#pragma unroll
	for (int w = 0; w < K; ++w) {
		uint32_t bits = cur[w];
		while (bits) {
			const uint32_t s = w * 32 + __builtin_ctz(bits);
			bits &= bits - 1;
			const uint32_t* row = masks + (size_t(s) * letters + letter) * K;
#pragma unroll
			for (int k = 0; k < K; ++k)
				next[k] |= row[k];
		}
	}
#pragma unroll
	for (int k = 0; k < K; ++k)
		cur[k] = next[k];
}

// The active set of one lane: on real text an NFA has one or two active states (measured: 1.3 - 1.8 on the fixtures),
// but WHICH states differs from lane to lane, so a bitset walk -- "for every word, for every set bit" -- makes a wave
// pay for the union of its lanes' words (7 - 8 iterations per byte for 1.4 active states).  The set is therefore kept
// as a LIST of up to four state ids while it fits: a step is four independent lookups, the same for every lane.
// Both forms hold exactly the reference's set (slow.h:63-74) -- a list may name a state twice, which changes
// nothing -- so the results do not depend on the form.
// An empty list slot holds the id E = `states`, one past the last state: the `single` table has a row for it that
// leads back to E, so that the step needs no "is this slot in use" test (round 2: 83 -> ~50 VALU per byte).
template <int K>
struct SlowLane {
	uint32_t lst[4];     // list form: state ids, E = empty slot (anywhere)
	uint32_t cur[K];     // bitset form
	uint32_t E;          // the empty marker (= number of states)
	bool bits;           // which form is valid

	__device__ __forceinline__ void Start(uint32_t start, uint32_t states)
	{
		E = states;
		lst[0] = start;
		lst[1] = lst[2] = lst[3] = E;
		bits = false;
#pragma unroll
		for (int k = 0; k < K; ++k)
			cur[k] = 0;
	}
	__device__ __forceinline__ void ListToBits()
	{
#pragma unroll
		for (int k = 0; k < K; ++k)
			cur[k] = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const uint32_t s = lst[i];
			if (s != E) {
#pragma unroll
				for (int k = 0; k < K; ++k)
					cur[k] |= (s >> 5) == uint32_t(k) ? (1u << (s & 31)) : 0u;
			}
		}
		bits = true;
	}
	// back to the list form when at most four states are active
	__device__ __forceinline__ void BitsToListIfSmall()
	{
		uint32_t count = 0;
#pragma unroll
		for (int k = 0; k < K; ++k)
			count += __popc(cur[k]);
		if (count > 4)
			return;
		uint32_t n[4] = {E, E, E, E};
		uint32_t used = 0;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			uint32_t b = cur[k];
			while (b) {
				const uint32_t s = uint32_t(k) * 32 + __builtin_ctz(b);
				b &= b - 1;
				n[0] = used == 0 ? s : n[0];
				n[1] = used == 1 ? s : n[1];
				n[2] = used == 2 ? s : n[2];
				n[3] = used == 3 ? s : n[3];
				++used;
			}
		}
#pragma unroll
		for (int i = 0; i < 4; ++i)
			lst[i] = n[i];
		bits = false;
	}
	__device__ __forceinline__ void Step(const uint32_t* masks, const uint2* single, uint32_t letters, uint32_t letter)
	{
		if (!bits) {
			// Every state moves to its (first) target IN PLACE -- no compaction, no duplicate check: a duplicate only
			// wastes a slot; an empty slot (E) moves to E.  A second target (the one row in a hundred where the set
			// grows, e.g. the start state meeting the pattern's first letter) is put into a free slot; two of them in one
			// step, no free slot, or a row with three targets send the lane to the bitset form.  Branch free.  Row
			// entries: x = first target, or kSlowMulti (the only value with the top bit set that x can take); y =
			// second target, or kSlowNone (top bit set).
			uint32_t n[4], extra = 0, extras = 0, multi = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const uint2 r = single[__umul24(lst[i], letters) + letter];
				n[i] = r.x;
				multi |= r.x;
				const bool has = int32_t(r.y) >= 0;
				extra = has ? r.y : extra;
				extras += has ? 1u : 0u;
			}
			uint32_t overflow = multi >> 31;
			if (__any(extras != 0)) {
				overflow |= uint32_t(extras > 1);
				uint32_t placed = uint32_t(extras != 1);
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint32_t take = uint32_t(n[j] == E) & (placed ^ 1u);
					n[j] = take ? extra : n[j];
					placed |= take;
				}
				overflow |= placed ^ 1u;
			}
			if (!overflow) {
#pragma unroll
				for (int i = 0; i < 4; ++i)
					lst[i] = n[i];
				return;
			}
			ListToBits();   // the OLD set; the step is redone in the bitset form
		}
		SlowStep<K>(masks, letters, cur, letter);
		BitsToListIfSmall();
	}
	__device__ __forceinline__ bool Final(const uint32_t* finals) const
	{
		bool fin = false;
		if (bits) {
#pragma unroll
			for (int k = 0; k < K; ++k)
				fin = fin || (cur[k] & finals[k]) != 0;
		} else {
#pragma unroll
			for (int i = 0; i < 4; ++i)
				fin = fin || (lst[i] != E && ((finals[lst[i] >> 5] >> (lst[i] & 31)) & 1u));
		}
		return fin;
	}
};

template <int K>
__global__ __launch_bounds__(1024) void SlowScanKernel(SlowParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t* ldsLetter = lds;                                        // 264 bytes
	uint32_t* ldsSingle = reinterpret_cast<uint32_t*>(lds + 272);
	const uint32_t entries = p.states * p.letters;
	const uint32_t singleEntries = (p.states + 1) * p.letters;   // + the row of the empty marker
	uint32_t* ldsMasks = ldsSingle + (p.singleInLds ? 2 * singleEntries : 0);
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x)
		ldsLetter[i] = p.letterOf[i];
	if (p.singleInLds)
		for (uint32_t i = threadIdx.x; i < 2 * singleEntries; i += blockDim.x)
			ldsSingle[i] = p.single[i];
	const uint32_t maskWords = entries * K;
	if (p.masksInLds)
		for (uint32_t i = threadIdx.x; i < maskWords; i += blockDim.x)
			ldsMasks[i] = p.masks[i];
	__syncthreads();
	const uint32_t* masks = p.masksInLds ? ldsMasks : p.masks;
	const uint2* single = reinterpret_cast<const uint2*>(p.singleInLds ? ldsSingle : p.single);

	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	unsigned long long finals = 0, strings = 0;
	// with an overflow list (the list kernel ran first on this stream): only the strings it names
	const uint64_t todo = p.overflow ? p.overflow[0] : p.n;
	for (uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; k < todo; k += stride) {
		const uint64_t s = p.overflow ? p.overflow[1 + k] : k;
		uint64_t b, e;
		if (p.offsets) {
			b = p.offsets[s];
			e = p.offsets[s + 1];
		} else {
			b = s * p.stride;
			e = b + p.len;
		}
		SlowLane<K> lane;
		lane.Start(p.start, p.states);                                   // Initialize, slow.h:89-95
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			lane.Step(masks, single, p.letters, ldsLetter[kBeginMark]);  // Begin(), run.h:375
		const uint8_t* ptr = p.text + b;
		const uint8_t* end = p.text + e;
		// Run<SlowScanner>, slow.h:436-451 -- byte by byte; 16-byte vector loads where the pointer allows
		for (; ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15); ++ptr)
			lane.Step(masks, single, p.letters, ldsLetter[*ptr]);
		// the next 16 bytes are requested before this block's 16 steps, not when they are needed
		uint4 ahead = ptr + 16 <= end ? *reinterpret_cast<const uint4*>(ptr) : uint4{0, 0, 0, 0};
		for (; ptr + 16 <= end; ptr += 16) {
			uint4 v = ahead;
			if (ptr + 32 <= end)
				ahead = *reinterpret_cast<const uint4*>(ptr + 16);
#pragma unroll 1
			for (int i = 0; i < 16; ++i) {
				lane.Step(masks, single, p.letters, ldsLetter[v.x & 0xFF]);
				v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
				v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
				v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
				v.w >>= 8;
			}
		}
		for (; ptr < end; ++ptr)
			lane.Step(masks, single, p.letters, ldsLetter[*ptr]);
		if (p.flags & PIRE_HIP_RUN_END)
			lane.Step(masks, single, p.letters, ldsLetter[kEndMark]);    // End(), run.h:376
		const bool fin = lane.Final(p.finals);                           // Final, slow.h:152-158
		if (p.outFinal)
			p.outFinal[s] = fin ? 1 : 0;
		if (p.outBits) {
			if (!lane.bits)
				lane.ListToBits();
#pragma unroll
			for (int k = 0; k < K; ++k)
				if (uint32_t(k) < p.words)
					p.outBits[s * p.words + k] = lane.cur[k];
		}
		finals += fin ? 1 : 0;
		strings += 1;
	}
	if (p.outCounts) {
		// wave reduce, one atomic pair per wave
		for (int off = 32; off > 0; off >>= 1) {
			finals += __shfl_down(finals, off);
			strings += __shfl_down(strings, off);
		}
		if ((threadIdx.x & 63) == 0 && strings) {
			atomicAdd(&p.outCounts[0], finals);
			atomicAdd(&p.outCounts[1], strings);
		}
	}
}

// ---- more than 256 NFA states, list form (round 3) ----------------------------------------------------------------------
// The automata this scanner exists for -- easy.h:155-161 falls back to it when determinisation explodes, the textbook
// case being x.{300}$ -- have hundreds or thousands of states of which a handful is active: one thread per 'x' seen in
// the last 300 bytes, 1 + 300/95 = 4.2 on printable text, and nearly every (state, letter) has ONE target.  So the set
// stays what SlowLane keeps it as, a list of states that move to their first target IN PLACE, only longer: 16 slots in
// registers.  Rows: row 0 is the EMPTY row (every letter leads back to it), state s has row s + 1; a row is `letters`
// words, and the word of (row, letter) is
//     bits  0..15  BYTE offset of the first target's row (0 = the empty row: the jump list is empty, the thread dies)
//     bits 16..30  index (row * letters) of the second target's row
//     bit  31      there is a second target   (second == kListMulti: there are three or more)
// A slot simply holds the word of its last lookup: its low half is where the state's row starts (no multiply, no mask:
// the address is one add with a 16-bit operand select), an empty slot is the word 0, and "four slots all empty" is one
// OR.  Step (branch free but for wave-uniform tests): up to 16 independent lookups; the OR of the 16 words says whether
// the lane spawned a thread; a spawned thread enters at slot 0 and everything else moves up one slot (16 selects, every
// step, whether or not anyone spawned -- cheaper than looking for a free slot only when someone did, which on text
// with one trigger letter in 95 is every other step of a wave).  When a state with a self loop spawns (the unanchored
// start state meeting the pattern's first letter) the table makes the SELF LOOP the spawned one: the persistent state
// stays at slot 0, the new thread takes over its old slot and drifts upwards as it ages, so threads of x.{n}, which die
// in the order they were born, die at the top and the list stays packed; groups of four slots that are empty in every
// lane of the wave are skipped.
// What the list cannot hold marks the STRING as overflowed -- two spawns in one step, a row with three targets, a live
// state pushed out of slot 15 -- and the string is walked again from its start by the wave-per-string kernel below,
// which knows no limits (SlowParams::overflow carries the list from one launch to the next on the stream).  A list may
// name a state twice (two threads that merged): that wastes a slot and changes nothing, as in SlowLane.
constexpr int kListSlots = 16;
constexpr uint32_t kListMulti = 0x7FFFu;
constexpr uint32_t kListMaxRows = 16383;   // (states + 1) * letters: 16-bit byte offsets of 4-byte words

typedef const __attribute__((address_space(3))) uint32_t* LdsWordPtr;

// Round 4: a step costs what the slots in use cost.  `top` (wave-uniform) bounds the groups of four slots that hold a
// thread in SOME lane of the wave: slots 4 * top .. 15 are empty in every lane.  ListStepT<T> walks the first T groups
// in straight-line code -- lookups, spawn bookkeeping and shift alike -- and the window loop picks the instance by
// `top` with one branch per step.  (Round 3 skipped the lookups of an empty group and still paid 16 selects, 16 ORs and,
// in the 80 % of the steps in which some lane spawns, a 16-slot count: 65 VALU instructions per step on x.{40}$, which
// bounded the kernel -- profiles/r04_slow_list_pmc.txt: 4.37 G instructions x 4 clocks on 1 024 SIMDs = 7.1 ms of the
// 6.8 ms launch.  A first form with a branch per group and phase traded 20 of them for 26 scalar instructions and 7
// branches per step, and was slower.)  The spawn words are SUMMED, not ORed: an entry's upper half is 0x8000 | the
// spawned row for a spawning entry and 0 for any other, so the sum of the upper halves is 0 (nobody spawned), below
// 0x10000 (exactly one did: the sum is its word) or not (two or more) -- one add per slot instead of an OR and a count.
template <int T>
__device__ __forceinline__ void ListStepT(uint32_t tabBase, uint32_t letter4, uint32_t (&lst)[kListSlots], bool& ovf)
{
	constexpr int N = 4 * T;
	uint32_t r[N];
	uint32_t sumHi = 0;
	const uint32_t base = letter4 + tabBase;
#pragma unroll
	for (int i = 0; i < N; ++i) {
		// the row is the slot's lower half (the upper one is what is left of a spawn word): mask and add in one instruction
		uint32_t addr;
		asm("v_mad_u32_u16 %0, %1, 1, %2" : "=v"(addr) : "v"(lst[i]), "v"(base));
		r[i] = *reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(addr));
		sumHi += r[i] >> 16;
	}
	const bool spawn = sumHi != 0;
	const uint32_t extra = sumHi & 0x7FFFu;   // the spawned thread's row -- if exactly one slot spawned
	// two spawns in one step do not fit this scheme (sum >= 0x10000), nor a row with three targets (its word is 0x8000 |
	// kListMulti = 0xFFFF, the largest a single one can be), nor a live thread pushed out of the last slot
	bool over = sumHi >= (0x8000u | kListMulti);
	if (N == kListSlots)
		over = over || (spawn && (r[N - 1] & 0xFFFFu) != 0);
	if (__any(over)) {   // rare: the string goes quiet (every slot empty from now on, so nothing spawns either)
		ovf = ovf || over;
#pragma unroll
		for (int i = 0; i < N; ++i)
			r[i] = over ? 0u : r[i];
	}
	const bool shift = spawn && !ovf;
	if (N < kListSlots)
		lst[N] = shift ? r[N - 1] : 0u;      // the first slot above: empty before, takes the thread that shifts up
#pragma unroll
	for (int i = N - 1; i > 0; --i)
		lst[i] = shift ? r[i - 1] : r[i];
	lst[0] = shift ? extra << 2 : r[0];
}

// the smallest `top` that holds for the wave (exact: used after the steps that do not keep it up to date)
__device__ __forceinline__ uint32_t ListTop(const uint32_t (&lst)[kListSlots])
{
	uint32_t top = 1;
#pragma unroll
	for (int g = 1; g < kListSlots / 4; ++g)
		top = __any((lst[4 * g] | lst[4 * g + 1] | lst[4 * g + 2] | lst[4 * g + 3]) != 0) ? uint32_t(g + 1) : top;
	return top;
}

// One step with `top` kept up to date: at most one group more (the thread that shifted into its first slot), one less
// when the highest group in use has emptied in every lane.
__device__ __forceinline__ void ListStepTop(uint32_t tabBase, uint32_t letter4, uint32_t (&lst)[kListSlots], bool& ovf, uint32_t& top)
{
	switch (top) {
	case 1:
		ListStepT<1>(tabBase, letter4, lst, ovf);
		top = __any(lst[4] != 0) ? 2u : 1u;
		break;
	case 2:
		ListStepT<2>(tabBase, letter4, lst, ovf);
		top = __any(lst[8] != 0) ? 3u : __any((lst[4] | lst[5] | lst[6] | lst[7]) != 0) ? 2u : 1u;
		break;
	case 3:
		ListStepT<3>(tabBase, letter4, lst, ovf);
		top = __any(lst[12] != 0) ? 4u : __any((lst[8] | lst[9] | lst[10] | lst[11]) != 0) ? 3u : 2u;
		break;
	default:
		ListStepT<4>(tabBase, letter4, lst, ovf);
		top = __any((lst[12] | lst[13] | lst[14] | lst[15]) != 0) ? 4u : 3u;
		break;
	}
}

// (the steps outside the window loop -- marks, the bytes in front of the first and behind the last whole block)
__device__ __forceinline__ void ListStep(uint32_t tabBase, uint32_t letter4, uint32_t (&lst)[kListSlots], bool& ovf)
{
	ListStepT<kListSlots / 4>(tabBase, letter4, lst, ovf);
}

__global__ __launch_bounds__(1024) void SlowListKernel(SlowParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t* ldsLetter = lds;                                        // 264 bytes: 4 * letter of every Char
	uint32_t* tab = reinterpret_cast<uint32_t*>(lds + 1056 + 16);
	const uint32_t tabBase = 1056 + 16;                              // LDS byte address of the table (dynamic LDS starts at 0)
	uint32_t* ldsLetter4 = reinterpret_cast<uint32_t*>(lds);         // [264] u32: 4 * letter, indexed by Char
	const uint32_t entries = (p.states + 1) * p.letters;
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x)
		ldsLetter4[i] = 4u * p.letterOf[i];
	for (uint32_t i = threadIdx.x; i < entries; i += blockDim.x)
		tab[i] = p.pair16[i];
	__syncthreads();
	(void)ldsLetter;
	const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
	unsigned long long finals = 0, strings = 0;
	for (uint64_t pass = 0; pass * stride < p.n; ++pass) {
		// order.hip: a wave takes 64 strings of about the same length, a lane long and short ones in turn
		const uint64_t k = OrderedIndex(pass, uint64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride, p.serpentine != 0);
		if (k >= p.n)
			continue;
		const uint64_t s = p.order ? p.order[k] : k;
		uint64_t b, e;
		if (p.offsets) {
			b = p.offsets[s];
			e = p.offsets[s + 1];
		} else {
#if defined(PIRE_EXP) && PIRE_EXP == 21   // timing experiment: every lane reads one of 64 strings (the text stays in the caches)
			b = (s & 63) * p.stride;
#else
			b = s * p.stride;
#endif
			e = b + p.len;
		}
		uint32_t lst[kListSlots];
		lst[0] = (p.start + 1) * p.letters * 4;                          // Initialize, slow.h:89-95
#pragma unroll
		for (int i = 1; i < kListSlots; ++i)
			lst[i] = 0;
		bool ovf = false;
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			ListStep(tabBase, ldsLetter4[kBeginMark], lst, ovf);          // Begin(), run.h:375
		const uint8_t* ptr = p.text + b;
		const uint8_t* end = p.text + e;
		for (; ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15); ++ptr)
			ListStep(tabBase, ldsLetter4[*ptr], lst, ovf);
		// the next 16 bytes are requested before this block's 16 steps, not when they are needed
		uint4 ahead = ptr + 16 <= end ? *reinterpret_cast<const uint4*>(ptr) : uint4{0, 0, 0, 0};
		uint32_t top = ListTop(lst);
		for (; ptr + 16 <= end; ptr += 16) {
			// (`top` is the wave's: a lane that has left this loop no longer counts, and what is left of the wave is
			// bounded by the same number or a smaller one)
			top = __builtin_amdgcn_readfirstlane(top);
			uint4 v = ahead;
			if (ptr + 32 <= end)
				ahead = *reinterpret_cast<const uint4*>(ptr + 16);
#pragma unroll 1
			for (int i = 0; i < 4; ++i) {   // a dword at a time: the byte is one v_bfe away, not four funnel shifts
				const uint32_t x = v.x;
				ListStepTop(tabBase, ldsLetter4[x & 0xFF], lst, ovf, top);
				ListStepTop(tabBase, ldsLetter4[(x >> 8) & 0xFF], lst, ovf, top);
				ListStepTop(tabBase, ldsLetter4[(x >> 16) & 0xFF], lst, ovf, top);
				ListStepTop(tabBase, ldsLetter4[x >> 24], lst, ovf, top);
				v.x = v.y;
				v.y = v.z;
				v.z = v.w;
			}
		}
		for (; ptr < end; ++ptr)
			ListStep(tabBase, ldsLetter4[*ptr], lst, ovf);
		if (p.flags & PIRE_HIP_RUN_END)
			ListStep(tabBase, ldsLetter4[kEndMark], lst, ovf);            // End(), run.h:376
		if (ovf) {
			// the wave-per-string kernel that follows on the stream walks this string from its start
			const uint32_t k = atomicAdd(&p.overflow[0], 1u);
			p.overflow[1 + k] = uint32_t(s);
			continue;
		}
		bool fin = false;                                                // Final, slow.h:152-158
#pragma unroll
		for (int i = 0; i < kListSlots; ++i) {
			const uint32_t row = (lst[i] & 0xFFFFu) >> 2;                 // rows back to state ids, once per string
			lst[i] = row ? row / p.letters - 1 : p.states;
			fin = fin || (lst[i] != p.states && ((p.finals[lst[i] >> 5] >> (lst[i] & 31)) & 1u));
		}
		if (p.outFinal)
			p.outFinal[s] = fin ? 1 : 0;
		if (p.outBits) {
			uint32_t* row = p.outBits + s * p.words;
			for (uint32_t w = 0; w < p.words; ++w) {
				uint32_t v = 0;
#pragma unroll
				for (int i = 0; i < kListSlots; ++i)
					v |= (lst[i] != p.states && (lst[i] >> 5) == w) ? 1u << (lst[i] & 31) : 0u;
				row[w] = v;
			}
		}
		finals += fin ? 1 : 0;
		strings += 1;
	}
	if (p.outCounts) {
		for (int off = 32; off > 0; off >>= 1) {
			finals += __shfl_down(finals, off);
			strings += __shfl_down(strings, off);
		}
		if ((threadIdx.x & 63) == 0 && strings) {
			atomicAdd(&p.outCounts[0], finals);
			atomicAdd(&p.outCounts[1], strings);
		}
	}
}

// ---- more than 256 NFA states: one wave per string, sparse jump lists ---------------------------------------------------
// cur / next are bitsets of `words` words (LDS, or device memory through the same flat pointers).  A step
//   next = {};  for every active state s:  for every t in jumps[jumpPos[s*letters+l] .. jumpPos[s*letters+l+1]):  next |= {t}
// is NextTranslated (slow.h:103-130) with the wave as the worker: lanes scan 64 words of `cur` at a time, the non-zero
// ones are broadcast one by one (readlane), every set bit is an active state, and its jump list is spread over the lanes
// (atomic OR into `next`: duplicates fall together as in the reference's flags.Test / Set).
// The sets are shared by the 64 lanes of ONE wave and nobody else; every phase (clear, scatter, read) ends with a fence
// + wave barrier.  LDS form: a workgroup-scope fence is a wait for the wave's own LDS operations, which the hardware
// keeps in order.  Device-memory form: words are read and written with relaxed device-scope atomics and the fence is
// device scope, so that a lane never meets a stale L1 line of a word another lane of its wave wrote.
template <bool SETS_LDS>
__device__ __forceinline__ void WideSync()
{
	if (SETS_LDS)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	else
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
	__builtin_amdgcn_wave_barrier();
}
template <bool SETS_LDS>
__device__ __forceinline__ uint32_t WideLoad(const uint32_t* q)
{
	if (SETS_LDS)
		return *q;
	return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SETS_LDS>
__device__ __forceinline__ void WideStore(uint32_t* q, uint32_t v)
{
	if (SETS_LDS)
		*q = v;
	else
		__hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct WideTables {
	const uint32_t* pos;     // jumpPos: LDS copy or device memory
	const uint32_t* jumps;
};

template <bool SETS_LDS>
__device__ __forceinline__ void WideStep(const SlowParams& p, const WideTables& T, uint32_t*& cur, uint32_t*& next,
                                         uint32_t letter, uint32_t lane)
{
	for (uint32_t w = lane; w < p.words; w += 64)
		WideStore<SETS_LDS>(&next[w], 0u);
	WideSync<SETS_LDS>();
	for (uint32_t w0 = 0; w0 < p.words; w0 += 64) {
		const uint32_t mine = w0 + lane < p.words ? WideLoad<SETS_LDS>(&cur[w0 + lane]) : 0u;
		unsigned long long nz = __ballot(mine != 0);
		while (nz) {
			const int j = __builtin_ctzll(nz);
			nz &= nz - 1;
			uint32_t bits = uint32_t(__builtin_amdgcn_readlane(int(mine), j));
			const uint32_t base = (w0 + uint32_t(j)) * 32;
			while (bits) {
				const uint32_t s = base + uint32_t(__builtin_ctz(bits));
				bits &= bits - 1;
				const uint32_t* pos = T.pos + size_t(s) * p.letters + letter;
				const uint32_t lo = pos[0], hi = pos[1];
				for (uint32_t k = lo + lane; k < hi; k += 64) {
					const uint32_t t = T.jumps[k];
					atomicOr(&next[t >> 5], 1u << (t & 31));
				}
			}
		}
	}
	WideSync<SETS_LDS>();
	uint32_t* tmp = cur;
	cur = next;
	next = tmp;
}

// LDS: [264 letters, padded to 272][jumpPos when posInLds][jumps when jumpsInLds][2 * words per wave when SETS_LDS]
template <bool SETS_LDS>
__global__ __launch_bounds__(1024) void SlowWideKernel(SlowParams p, uint32_t wavesPerBlock, uint32_t posInLds,
                                                       uint32_t jumpsInLds, uint32_t njumps)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t* ldsLetter = lds;
	uint32_t* ldsPos = reinterpret_cast<uint32_t*>(lds + 272);
	const uint32_t npos = p.states * p.letters + 1;
	uint32_t* ldsJumps = ldsPos + (posInLds ? npos : 0);
	uint32_t* ldsSets = ldsJumps + (jumpsInLds ? njumps : 0);
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x)
		ldsLetter[i] = p.letterOf[i];
	if (posInLds)
		for (uint32_t i = threadIdx.x; i < npos; i += blockDim.x)
			ldsPos[i] = p.jumpPos[i];
	if (jumpsInLds)
		for (uint32_t i = threadIdx.x; i < njumps; i += blockDim.x)
			ldsJumps[i] = p.jumps[i];
	__syncthreads();
	WideTables T;
	T.pos = posInLds ? ldsPos : p.jumpPos;
	T.jumps = jumpsInLds ? ldsJumps : p.jumps;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = threadIdx.x >> 6;
	const uint64_t gwave = uint64_t(blockIdx.x) * wavesPerBlock + wave;
	uint32_t* setA = SETS_LDS ? ldsSets + size_t(wave) * 2 * p.words : p.scratch + gwave * 2 * p.words;
	uint32_t* setB = setA + p.words;
	unsigned long long finals = 0, strings = 0;
	// with an overflow list (the list kernel ran first on this stream): only the strings it names
	const uint64_t todo = p.overflow ? p.overflow[0] : p.n;
	for (uint64_t k = gwave; k < todo; k += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t s = p.overflow ? p.overflow[1 + k] : k;
		uint64_t b, e;
		if (p.offsets) {
			b = p.offsets[s];
			e = p.offsets[s + 1];
		} else {
			b = s * p.stride;
			e = b + p.len;
		}
		uint32_t *cur = setA, *next = setB;
		for (uint32_t w = lane; w < p.words; w += 64)                    // Initialize, slow.h:89-95: { start }
			WideStore<SETS_LDS>(&cur[w], (p.start >> 5) == w ? 1u << (p.start & 31) : 0u);
		WideSync<SETS_LDS>();
		if (p.flags & PIRE_HIP_RUN_BEGIN)
			WideStep<SETS_LDS>(p, T, cur, next, ldsLetter[kBeginMark], lane);         // Begin(), run.h:375
		for (uint64_t i = b; i < e; i += 64) {                           // Run<SlowScanner>, slow.h:436-451
			const uint32_t mine = i + lane < e ? p.text[i + lane] : 0u;  // 64 bytes per load, one per lane
			const uint32_t cnt = e - i < 64 ? uint32_t(e - i) : 64u;
			for (uint32_t k = 0; k < cnt; ++k)
				WideStep<SETS_LDS>(p, T, cur, next, ldsLetter[__builtin_amdgcn_readlane(int(mine), int(k))], lane);
		}
		if (p.flags & PIRE_HIP_RUN_END)
			WideStep<SETS_LDS>(p, T, cur, next, ldsLetter[kEndMark], lane);           // End(), run.h:376
		bool fin = false;                                                // Final, slow.h:152-158
		for (uint32_t w = lane; w < p.words; w += 64) {
			const uint32_t v = WideLoad<SETS_LDS>(&cur[w]);
			fin = fin || (v & p.finals[w]) != 0;
			if (p.outBits)
				p.outBits[s * p.words + w] = v;
		}
		fin = __any(fin);
		if (p.outFinal && lane == 0)
			p.outFinal[s] = fin ? 1 : 0;
		finals += fin ? 1 : 0;
		strings += 1;
		WideSync<SETS_LDS>();
	}
	if (p.outCounts && lane == 0 && strings) {
		atomicAdd(&p.outCounts[0], finals);
		atomicAdd(&p.outCounts[1], strings);
	}
}

namespace {

size_t Up8(size_t v) { return (v + 7) & ~size_t(7); }

int BadSlow(const char* msg)
{
	SetError(msg);
	return PIRE_HIP_EFORMAT;
}

// SlowScanner::Load, scanner_io.cpp:113-170 (layout written by Save, 71-111).
int BuildSlowHost(const void* blob, size_t len, SlowHost* out)
{
	const uint8_t* p = static_cast<const uint8_t*>(blob);
	if (!p || len < 24 + 24 + 8)
		return BadSlow("EOF reached while reading the SlowScanner header");
	uint32_t hdr[6];
	memcpy(hdr, p, 24);
	if (hdr[0] != 0x45524950u || hdr[2] != 8 || hdr[3] != 16)
		return BadSlow("Serialized regexp incompatible with your system");
	if (hdr[1] != 7 && hdr[1] != 6)
		return BadSlow("You are trying to used an incompatible version of a serialized regexp");
	if (hdr[4] != 3 /* ScannerIOTypes::SlowScanner */ || hdr[5] != 24)
		return BadSlow("Serialized regexp incompatible with your system");
	uint64_t states, letters, start;
	memcpy(&states, p + 24, 8);
	memcpy(&letters, p + 32, 8);
	memcpy(&start, p + 40, 8);
	size_t pos = 48;
	const bool empty = p[pos] != 0;
	pos += 8;
	SlowHost& h = *out;
	h = SlowHost();
	h.empty = empty;
	if (empty) {
		// Null() = Fsm::MakeFalse() compiled (slow.h:425-429): one non-final state, nowhere to go
		h.states = 1;
		h.letters = 1;
		h.start = 0;
		h.words = 1;
		h.letterOf.assign(kMaxChar, 0);
		h.masks.assign(1, 0);
		h.single.assign(4, kSlowNone);
		h.single[0] = h.single[2] = 1;   // state 0 goes nowhere (its slot empties), and the empty marker E = 1 stays E
		h.finals.assign(1, 0);
		return PIRE_HIP_OK;
	}
	if (states == 0 || letters == 0 || letters > 256 || states > (1u << 20) || start >= states)
		return BadSlow("Corrupt SlowScanner: bad geometry");
	h.states = uint32_t(states);
	h.letters = uint32_t(letters);
	h.start = uint32_t(start);
	h.words = (h.states + 31) / 32;
	if (len < pos + size_t(kMaxChar) * 8)
		return BadSlow("EOF reached while reading SlowScanner letters");
	h.letterOf.assign(kMaxChar, 0);
	for (uint32_t c = 0; c < kMaxChar; ++c) {
		uint64_t v;
		memcpy(&v, p + pos + size_t(c) * 8, 8);
		if (v >= letters)
			return BadSlow("Corrupt SlowScanner: letter out of range");
		h.letterOf[c] = uint8_t(v);
	}
	pos += size_t(kMaxChar) * 8;
	if (len < pos + Up8(states))
		return BadSlow("EOF reached while reading SlowScanner finals");
	h.finals.assign(h.words, 0);
	for (uint32_t s = 0; s < h.states; ++s)
		if (p[pos + s])
			h.finals[s >> 5] |= 1u << (s & 31);
	pos += Up8(states);
	const size_t npos = size_t(states) * letters + 1;
	if (len < pos + npos * 8)
		return BadSlow("EOF reached while reading SlowScanner jump positions");
	std::vector<uint64_t> jumpPos(npos);
	memcpy(jumpPos.data(), p + pos, npos * 8);
	pos += npos * 8;
	const uint64_t njumps = jumpPos[npos - 1];
	if (len < pos + Up8(size_t(njumps) * 4))
		return BadSlow("EOF reached while reading SlowScanner jumps");
	if (njumps >= (1ull << 32))
		return BadSlow("Corrupt SlowScanner: too many jumps");
	for (size_t i = 0; i + 1 < npos; ++i)
		if (jumpPos[i] > jumpPos[i + 1] || jumpPos[i + 1] > njumps)
			return BadSlow("Corrupt SlowScanner: jump positions not monotone");
	for (uint64_t k = 0; k < njumps; ++k) {
		uint32_t tgt;
		memcpy(&tgt, p + pos + size_t(k) * 4, 4);
		if (tgt >= states)
			return BadSlow("Corrupt SlowScanner: jump target out of range");
	}
	// the list form's table (SlowListKernel; layout there): row 0 = the empty row, state s = row s + 1
	if ((size_t(states) + 1) * letters <= kListMaxRows) {
		h.pair16.assign((size_t(states) + 1) * letters, 0);   // "no target": the thread dies (slot -> empty row)
		for (size_t i = 0; i + 1 < npos; ++i) {
			const uint32_t self = uint32_t(i / letters);
			uint32_t t0 = kSlowNone, t1 = kSlowNone;
			bool multi = false;
			for (uint64_t k = jumpPos[i]; k < jumpPos[i + 1]; ++k) {
				uint32_t tgt;
				memcpy(&tgt, p + pos + size_t(k) * 4, 4);
				if (t0 == kSlowNone)
					t0 = tgt;
				else if (tgt == t0)
					continue;
				else if (t1 == kSlowNone)
					t1 = tgt;
				else if (tgt != t1)
					multi = true;
			}
			if (t1 != kSlowNone && t0 == self)
				std::swap(t0, t1);   // a self loop is the SPAWNED one: the persistent state re-enters at slot 0
			const auto rowIndex = [&](uint32_t t) { return uint32_t((size_t(t) + 1) * letters); };
			uint32_t& w = h.pair16[letters + i];
			if (multi)
				w = (1u << 31) | (kListMulti << 16);
			else if (t1 != kSlowNone)
				w = (1u << 31) | (rowIndex(t1) << 16) | (rowIndex(t0) * 4);
			else if (t0 != kSlowNone)
				w = rowIndex(t0) * 4;
		}
	}
	h.wide = h.words > 8;
	if (h.wide) {
		// more than 256 states: the sparse form as it is (a dense (state, letter) -> set matrix would be quadratic)
		h.jumpPos.resize(npos);
		for (size_t i = 0; i < npos; ++i)
			h.jumpPos[i] = uint32_t(jumpPos[i]);
		h.jumps.resize(size_t(njumps));
		memcpy(h.jumps.data(), p + pos, size_t(njumps) * 4);
		return PIRE_HIP_OK;
	}
	h.masks.assign(size_t(states) * letters * h.words, 0);
	for (size_t i = 0; i + 1 < npos; ++i)
		for (uint64_t k = jumpPos[i]; k < jumpPos[i + 1]; ++k) {
			uint32_t tgt;
			memcpy(&tgt, p + pos + size_t(k) * 4, 4);
			h.masks[i * h.words + (tgt >> 5)] |= 1u << (tgt & 31);
		}
	// the first two distinct targets of every jump list, for the list form of the walk: [2*i] = the first target (E =
	// `states`, the empty marker, when the list is empty), [2*i+1] = the second or kSlowNone; three or more distinct
	// targets: [2*i] = kSlowMulti
	h.single.assign(size_t(states + 1) * letters * 2, kSlowNone);
	for (uint64_t l = 0; l < letters; ++l)
		h.single[(size_t(states) * letters + l) * 2] = uint32_t(states);   // the empty marker E = states stays E
	for (size_t i = 0; i + 1 < npos; ++i) {
		uint32_t t0 = kSlowNone, t1 = kSlowNone;
		bool multi = false;
		for (uint64_t k = jumpPos[i]; k < jumpPos[i + 1]; ++k) {
			uint32_t tgt;
			memcpy(&tgt, p + pos + size_t(k) * 4, 4);
			if (t0 == kSlowNone)
				t0 = tgt;
			else if (tgt == t0)
				continue;
			else if (t1 == kSlowNone)
				t1 = tgt;
			else if (tgt != t1)
				multi = true;
		}
		h.single[2 * i] = multi ? kSlowMulti : t0 == kSlowNone ? uint32_t(states) : t0;   // no target: the slot empties (E)
		h.single[2 * i + 1] = multi ? kSlowNone : t1;
	}
	return PIRE_HIP_OK;
}

template <class T>
int PutSlow(T** dst, const std::vector<T>& src)
{
	*dst = nullptr;
	hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(src.size() * sizeof(T), 16));
	if (e != hipSuccess)
		return HipFail(e, "hipMalloc(slow table)");
	if (!src.empty()) {
		e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
		if (e != hipSuccess)
			return HipFail(e, "hipMemcpy(slow table)");
	}
	return PIRE_HIP_OK;
}

void FreeSlowDevice(SlowDevice* d)
{
	if (d->device < 0)
		return;
	if (d->letterOf) (void)hipFree(d->letterOf);
	if (d->masks) (void)hipFree(d->masks);
	if (d->single) (void)hipFree(d->single);
	if (d->finals) (void)hipFree(d->finals);
	if (d->jumpPos) (void)hipFree(d->jumpPos);
	if (d->jumps) (void)hipFree(d->jumps);
	if (d->pair16) (void)hipFree(d->pair16);
	*d = SlowDevice();
}

// The kernel is compiled for K in {1,2,4,8} words; the device rows are padded to that K.
int DeviceK(uint32_t words)
{
	return words <= 1 ? 1 : words <= 2 ? 2 : words <= 4 ? 4 : words <= 8 ? 8 : 0;
}

// Image of the current device (built on first use), copied out under the table's lock.
int UploadSlow(pire_hip_slow_table* t, SlowDevice* image)
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
	const SlowHost& h = t->host;
	if (h.wide) {
		SlowDevice d;
		int rc;
		if ((rc = PutSlow(&d.letterOf, h.letterOf)) || (rc = PutSlow(&d.finals, h.finals)) ||
		    (rc = PutSlow(&d.jumpPos, h.jumpPos)) || (rc = PutSlow(&d.jumps, h.jumps)) ||
		    (!h.pair16.empty() && (rc = PutSlow(&d.pair16, h.pair16)))) {
			d.device = dev;
			FreeSlowDevice(&d);
			return rc;
		}
		d.device = dev;
		t->devs[dev] = d;
		*image = d;
		return PIRE_HIP_OK;
	}
	const int K = DeviceK(h.words);
	std::vector<uint32_t> masks(size_t(h.states) * h.letters * K, 0), finals(K, 0);
	for (size_t i = 0; i < size_t(h.states) * h.letters; ++i)
		for (uint32_t w = 0; w < h.words; ++w)
			masks[i * K + w] = h.masks[i * h.words + w];
	for (uint32_t w = 0; w < h.words; ++w)
		finals[w] = h.finals[w];
	SlowDevice d;
	int rc;
	if ((rc = PutSlow(&d.letterOf, h.letterOf)) || (rc = PutSlow(&d.masks, masks)) || (rc = PutSlow(&d.single, h.single)) ||
	    (rc = PutSlow(&d.finals, finals)) || (!h.pair16.empty() && (rc = PutSlow(&d.pair16, h.pair16)))) {
		d.device = dev;
		FreeSlowDevice(&d);
		return rc;
	}
	d.device = dev;
	t->devs[dev] = d;
	*image = d;
	return PIRE_HIP_OK;
}

template <int K>
int LaunchSlowK(const SlowParams& p0, hipStream_t stream)
{
	SlowParams p = p0;
	const size_t maskBytes = size_t(p.states) * p.letters * K * 4;
	const size_t singleBytes = size_t(p.states + 1) * p.letters * 8;
	p.singleInLds = singleBytes <= 64 * 1024 ? 1 : 0;
	p.masksInLds = maskBytes + (p.singleInLds ? singleBytes : 0) <= 150 * 1024 ? 1 : 0;
	const uint32_t ldsBytes = uint32_t(272 + (p.singleInLds ? singleBytes : 0) + (p.masksInLds ? maskBytes : 0));
	hipError_t e = SetDynamicLds(reinterpret_cast<const void*>(SlowScanKernel<K>), uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	int dev = 0, cus = 0;
	if ((e = hipGetDevice(&dev)) != hipSuccess ||
	    (e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess)
		return HipFail(e, "device query");
	// big batches: 1024-thread blocks, as many per CU as the LDS copy of the masks allows (32 waves per CU at most);
	// small ones: 256-thread blocks so that more CUs take part
	const unsigned threads = p.n >= 128 * 1024 ? 1024 : 256;
	const uint64_t perCu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / threads, (160 * 1024) / std::max<uint32_t>(ldsBytes, 1)));
	const uint64_t want = (p.n + threads - 1) / threads;
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>(want, uint64_t(cus) * perCu)));
	hipLaunchKernelGGL(SlowScanKernel<K>, dim3(blocks), dim3(threads), ldsBytes, stream, p);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "slow kernel launch");
	return PIRE_HIP_OK;
}

// One wave per string.  LDS budget, in this order: the two sets of every wave (16 waves per block, fewer when the sets
// are large; device memory from the stream-ordered pool when not even one wave's fit), then jumpPos, then the jumps.
int LaunchSlowWide(const SlowParams& p0, uint32_t njumps, hipStream_t stream)
{
	SlowParams p = p0;
	int dev = 0, cus = 0;
	hipError_t e;
	if ((e = hipGetDevice(&dev)) != hipSuccess ||
	    (e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess)
		return HipFail(e, "device query");
	const size_t setBytes = size_t(p.words) * 8;   // cur + next of one wave
	const size_t ldsRoom = 158 * 1024 - 272;
	uint32_t waves = uint32_t(std::min<size_t>(16, ldsRoom / setBytes));
	const bool inLds = waves >= 1 && !GetConfig().slow_sets_in_memory;   // (the knob: tests run the device-memory form at any size)
	if (!inLds)
		waves = 16;
	size_t room = ldsRoom - (inLds ? waves * setBytes : 0);
	const size_t posBytes = (size_t(p.states) * p.letters + 1) * 4, jumpBytes = size_t(njumps) * 4;
	const uint32_t posInLds = posBytes <= room ? 1 : 0;
	if (posInLds)
		room -= posBytes;
	const uint32_t jumpsInLds = posInLds && jumpBytes <= room ? 1 : 0;
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((p.n + waves - 1) / waves, uint64_t(cus)));
	void* scratch = nullptr;
	if (!inLds) {
		e = hipMallocAsync(&scratch, blocks * waves * setBytes, stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(slow scanner sets)");
		p.scratch = static_cast<uint32_t*>(scratch);
	}
	const uint32_t ldsBytes = uint32_t(272 + (inLds ? waves * setBytes : 0) + (posInLds ? posBytes : 0) + (jumpsInLds ? jumpBytes : 0));
	const void* fn = inLds ? reinterpret_cast<const void*>(SlowWideKernel<true>) : reinterpret_cast<const void*>(SlowWideKernel<false>);
	e = SetDynamicLds(fn, uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	if (inLds)
		hipLaunchKernelGGL(SlowWideKernel<true>, dim3(unsigned(blocks)), dim3(waves * 64), ldsBytes, stream, p, waves, posInLds,
		                   jumpsInLds, njumps);
	else
		hipLaunchKernelGGL(SlowWideKernel<false>, dim3(unsigned(blocks)), dim3(waves * 64), ldsBytes, stream, p, waves, posInLds,
		                   jumpsInLds, njumps);
	e = hipGetLastError();
	if (scratch)
		(void)hipFreeAsync(scratch, stream);
	if (e != hipSuccess)
		return HipFail(e, "slow kernel launch");
	return PIRE_HIP_OK;
}

// List form first: SlowListKernel over the whole batch, then `fallback` -- the bitset kernel (<= 256 states) or the
// wave-per-string kernel (more) -- over the strings whose sets outgrew the list (usually none; the count stays on the
// device, an empty second launch costs a few microseconds).  The overflow list lives in stream-ordered scratch.
template <class Fallback>
int LaunchSlowListThen(const SlowParams& p0, hipStream_t stream, Fallback fallback)
{
	SlowParams p = p0;
	int dev = 0, cus = 0;
	hipError_t e;
	if ((e = hipGetDevice(&dev)) != hipSuccess ||
	    (e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess)
		return HipFail(e, "device query");
	void* list = nullptr;
	e = hipMallocAsync(&list, (size_t(p.n) + 1) * 4, stream);
	if (e == hipSuccess)
		e = hipMemsetAsync(list, 0, 4, stream);
	if (e != hipSuccess) {
		if (list)
			(void)hipFreeAsync(list, stream);
		return HipFail(e, "hipMallocAsync(slow scanner overflow list)");
	}
	p.overflow = static_cast<uint32_t*>(list);
	// ragged batches: the strings by length class (order.hip) -- this kernel is all VALU, a wave waiting for its longest
	// string is its one avoidable cost
	void* orderScratch = nullptr;
	p.order = nullptr;
	if (p.offsets && LengthOrderWanted(p.n)) {
		e = hipMallocAsync(&orderScratch, LengthOrderScratchBytes(p.n), stream);
		if (e != hipSuccess) {
			(void)hipFreeAsync(list, stream);
			return HipFail(e, "hipMallocAsync(length order)");
		}
		bool serp = false;
		const int orc = BuildLengthOrder(p.offsets, p.n, orderScratch, stream, &p.order, &serp);
		p.serpentine = serp ? 1u : 0u;
		if (orc) {
			(void)hipFreeAsync(orderScratch, stream);
			(void)hipFreeAsync(list, stream);
			return orc;
		}
	}
	const uint32_t ldsBytes = uint32_t(1056 + 16 + size_t(p.states + 1) * p.letters * 4);
	e = SetDynamicLds(reinterpret_cast<const void*>(SlowListKernel), uint32_t(ldsBytes));
	int rc = PIRE_HIP_OK;
	if (e != hipSuccess) {
		rc = HipFail(e, "hipFuncSetAttribute(LDS)");
	} else {
		// one string per lane and ~100 VALU instructions per byte: spread the waves over every SIMD of the chip before
		// stacking them (65 536 strings are 1 024 waves = one per SIMD: 256-thread blocks, one per CU)
		const unsigned threads = p.n >= uint64_t(cus) * 2048 ? 1024 : p.n >= uint64_t(cus) * 512 ? 512 : 256;
		const uint64_t perCu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / threads, (160 * 1024) / ldsBytes));
		const uint64_t want = (p.n + threads - 1) / threads;
		const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>(want, uint64_t(cus) * perCu)));
		hipLaunchKernelGGL(SlowListKernel, dim3(blocks), dim3(threads), ldsBytes, stream, p);
		e = hipGetLastError();
		if (e != hipSuccess)
			rc = HipFail(e, "slow kernel launch");
	}
	if (rc == PIRE_HIP_OK && GetConfig().slow_stats) {   // measurements: how many strings the 16-slot list could not hold
		uint32_t over = 0;
		if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(&over, list, 4, hipMemcpyDeviceToHost) == hipSuccess)
			fprintf(stderr, "pire_hip slow: %u of %llu strings left the list kernel for the %s one\n", over,
			        static_cast<unsigned long long>(p.n), p.words > 8 ? "wave-per-string" : "bitset");
	}
	if (rc == PIRE_HIP_OK)
		rc = fallback(p);   // p.overflow set: only the strings on the list
	if (orderScratch)
		(void)hipFreeAsync(orderScratch, stream);
	(void)hipFreeAsync(list, stream);
	return rc;
}

int RunSlow(pire_hip_slow_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint64_t len, uint64_t stride,
            uint32_t flags, uint8_t* outFinal, uint32_t* outBits, uint64_t* outCounts, void* streamPtr)
{
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	SlowDevice image;
	if (int rc = UploadSlow(t, &image))
		return rc;
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	const SlowHost& h = t->host;
	SlowParams p;
	memset(&p, 0, sizeof(p));
	p.letterOf = image.letterOf;
	p.masks = image.masks;
	p.single = image.single;
	p.finals = image.finals;
	p.jumpPos = image.jumpPos;
	p.jumps = image.jumps;
	p.pair16 = image.pair16;
	p.states = h.states;
	p.letters = h.letters;
	p.start = h.start;
	p.words = h.words;
	p.flags = flags;
	p.n = n;
	p.len = len;
	p.stride = stride;
	if (n == 0)
		return PIRE_HIP_OK;
	const int K = DeviceK(h.words);
	const bool listFirst = image.pair16 && n < (1ull << 32) - 1 && !GetConfig().slow_no_list;
	auto plain = [&](const SlowParams& q) {
		if (h.wide)
			return LaunchSlowWide(q, uint32_t(h.jumps.size()), stream);
		switch (K) {
		case 1: return LaunchSlowK<1>(q, stream);
		case 2: return LaunchSlowK<2>(q, stream);
		case 4: return LaunchSlowK<4>(q, stream);
		default: return LaunchSlowK<8>(q, stream);
		}
	};
	auto launch = [&](const SlowParams& q) {
		if (listFirst) {
			NoteKernel("slow_list", "pirehip::SlowListKernel");
			return LaunchSlowListThen(q, stream, plain);
		}
		if (h.wide)
			NoteKernel("slow_wide", "pirehip::SlowWideKernel");
		else
			NoteKernel("slow", K == 1 ? "pirehip::SlowScanKernel<1>" : K == 2 ? "pirehip::SlowScanKernel<2>" :
			                    K == 4 ? "pirehip::SlowScanKernel<4>" : "pirehip::SlowScanKernel<8>");
		return plain(q);
	};
	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		p.outFinal = outFinal;
		p.outBits = outBits;
		p.outCounts = reinterpret_cast<unsigned long long*>(outCounts);
		return launch(p);
	}
	// host-pointer mode: stage through HBM
	Staging stage(stream);
	auto alloc = [&](void** d, size_t bytes) { return stage.Alloc(d, bytes); };
	auto cleanup = [] {};   // the staging frees itself when the call returns
	uint64_t textBytes = offsets ? offsets[n] : (n - 1) * stride + len;
	if (!text && textBytes) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	void *dText = nullptr, *dOff = nullptr, *dFin = nullptr, *dBits = nullptr, *dCnt = nullptr;
	int rc = alloc(&dText, textBytes);
	if (!rc && textBytes)
		if (hipMemcpyAsync(dText, text, textBytes, hipMemcpyHostToDevice, stream) != hipSuccess)
			rc = HipFail(hipGetLastError(), "hipMemcpy(H2D)");
	if (!rc && offsets) {
		rc = alloc(&dOff, (n + 1) * 8);
		if (!rc && hipMemcpyAsync(dOff, offsets, (n + 1) * 8, hipMemcpyHostToDevice, stream) != hipSuccess)
			rc = HipFail(hipGetLastError(), "hipMemcpy(H2D)");
	}
	if (!rc && outFinal) rc = alloc(&dFin, n);
	if (!rc && outBits) rc = alloc(&dBits, n * h.words * 4);
	if (!rc && outCounts) {
		rc = alloc(&dCnt, 16);
		if (!rc && hipMemcpyAsync(dCnt, outCounts, 16, hipMemcpyHostToDevice, stream) != hipSuccess)
			rc = HipFail(hipGetLastError(), "hipMemcpy(H2D)");
	}
	if (!rc) {
		p.text = static_cast<const uint8_t*>(dText);
		p.offsets = static_cast<const uint64_t*>(dOff);
		p.outFinal = static_cast<uint8_t*>(dFin);
		p.outBits = static_cast<uint32_t*>(dBits);
		p.outCounts = static_cast<unsigned long long*>(dCnt);
		rc = launch(p);
	}
	hipError_t e = hipSuccess;
	if (!rc && outFinal) e = hipMemcpyAsync(outFinal, dFin, n, hipMemcpyDeviceToHost, stream);
	if (!rc && e == hipSuccess && outBits) e = hipMemcpyAsync(outBits, dBits, n * h.words * 4, hipMemcpyDeviceToHost, stream);
	if (!rc && e == hipSuccess && outCounts) e = hipMemcpyAsync(outCounts, dCnt, 16, hipMemcpyDeviceToHost, stream);
	if (!rc && e == hipSuccess) e = hipStreamSynchronize(stream);
	if (!rc && e != hipSuccess)
		rc = HipFail(e, "copy back / synchronize");
	cleanup();
	return rc;
}

}  // namespace
}  // namespace pirehip

using namespace pirehip;

extern "C" {

int pire_hip_slow_table_create(const void* save_blob, size_t len, pire_hip_slow_table** out)
try {
	if (!out) {
		SetError("null out pointer");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	std::unique_ptr<pire_hip_slow_table> t(new (std::nothrow) pire_hip_slow_table);
	if (!t) {
		SetError("out of memory");
		return PIRE_HIP_ENOMEM;
	}
	if (int rc = BuildSlowHost(save_blob, len, &t->host))
		return rc;
	*out = t.release();
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

void pire_hip_slow_table_destroy(pire_hip_slow_table* t)
{
	if (!t)
		return;
	int cur = -1;
	(void)hipGetDevice(&cur);
	for (int k = 0; k < kMaxDevices; ++k)
		if (t->devs[k].device >= 0) {
			(void)hipSetDevice(k);
			FreeSlowDevice(&t->devs[k]);
		}
	if (cur >= 0)
		(void)hipSetDevice(cur);
	delete t;
}

int pire_hip_slow_table_get_info(const pire_hip_slow_table* t, pire_hip_slow_info* out)
try {
	if (!t || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	memset(out, 0, sizeof(*out));
	out->states = t->host.states;
	out->letters = t->host.letters;
	out->start = t->host.start;
	out->words = t->host.words;
	out->empty = t->host.empty ? 1 : 0;
	out->mask_bytes = uint64_t(t->host.masks.size()) * 4;
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_slow_run(pire_hip_slow_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                      uint8_t* out_final, uint32_t* out_state_bits, uint64_t* out_counts, void* stream)
try {
	if (n && !offsets) {
		SetError("null offsets");
		return PIRE_HIP_EINVAL;
	}
	return RunSlow(t, text, offsets, n, 0, 0, flags, out_final, out_state_bits, out_counts, stream);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_slow_run_strided(pire_hip_slow_table* t, const void* text, uint64_t n, uint64_t len, uint64_t stride,
                              uint32_t flags, uint8_t* out_final, uint32_t* out_state_bits, uint64_t* out_counts,
                              void* stream)
try {
	if (stride < len) {
		SetError("stride smaller than len");
		return PIRE_HIP_EINVAL;
	}
	return RunSlow(t, text, nullptr, n, len, stride, flags, out_final, out_state_bits, out_counts, stream);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

}  // extern "C"
