// Internal structures of libpire_hip.so (not installed).  See DESIGN.md for the data layout.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/pire_hip.h"

namespace pirehip {

// pire/defs.h:59-73
enum : uint32_t {
	kEpsilon = 257,
	kBeginMark = 258,
	kEndMark = 259,
	kMaxCharUnaligned = 260,
	kMaxChar = 264,
};

enum : uint8_t {
	kFinal = 1,      // FinalFlag, multi.h:91
	kDead = 2,       // DeadFlag,  multi.h:92
	kAbsorbing = 4,  // ours: every transition of the row (all letter classes, marks included) is a self loop
};

constexpr uint32_t kMaxHotRows = 256;   // dense rows addressable by a u8 state id (255 hot + 1 trap)

// LDS carve-up shared by the scan kernels (single dynamic region, 16-byte aligned pieces).
struct LdsLayout {
	uint32_t pitch;        // bytes between dense rows: 260 = 65 dwords rotates row r by r banks, or 256
	uint32_t hotBytes;     // (hot+1)*pitch, rounded up to 16
	uint32_t flagsOff;     // 256 B of hot flags
	uint32_t cls8Off;      // 256 B: 2 * letter class of every byte (compact tier; 256-byte aligned when pitch == 256)
	uint32_t compactOff;   // compact tier: class-indexed u16 rows of the first `compact` states + the escape row
	uint32_t compactBytes;
	uint32_t clsOff;       // 264 u16 (generic kernel + slow step)
	uint32_t countsOff;    // (regexps+2) u32 block-local counters
	uint32_t histOff;      // 256 u32: sampled visits of hot ids (feeds pire_hip_table_adapt)
	uint32_t progOff;      // u32: tiles walked by the waves of the block (tiled kernel: keeps them in step)
	uint32_t total;
	uint32_t rot2;         // the dense rows' columns are in ROTATED byte order (column of byte b = rotl8(b, 2)): tiled kernel variant
};

// Column of input byte b in a dense row whose columns are rotated (LdsLayout::rot2): the LDS bank of a lookup is then
// b & 63 instead of (b >> 2) & 63 -- printable text spreads over all 64 banks instead of 24.
__host__ __device__ inline uint32_t RotColumn(uint32_t b) { return ((b << 2) | (b >> 6)) & 0xFFu; }

constexpr uint32_t kRotPitch = 260;
constexpr uint32_t kMaxLdsCountRegexps = 1024;
constexpr uint32_t kCheckSlot = 256;                     // visitHot[256]: failures seen by the checked kernel build
constexpr uint32_t kTrapSlot = 257;                      // visitHot[257]: sampled traps since the image was uploaded (device total)
constexpr uint32_t kWideTrapSlot = 258;                 // visitHot[258]: sampled chunks the WIDE walk left its rows in
constexpr uint32_t kVisitHotSlots = 260;
constexpr uint32_t kLdsTrapSlot = 257;                   // the same count inside a block: hist[257] in LDS (hist[256] = progress)
constexpr uint32_t kLdsPerBlock = 160 * 1024;            // gfx950: 160 KiB per CU, one block per CU may have it all
constexpr uint32_t kRaggedFinBytes = 256 * 16;           // ragged kernel: end-of-string records of the hot states
constexpr uint32_t kRaggedLdsExtra = kRaggedFinBytes + 32;

// The compact region sits right behind the dense rows so that its LDS addresses do not depend on the per-launch
// pieces (counters): the rows hold LDS addresses.
__host__ __device__ inline LdsLayout MakeLayout(uint32_t hot, uint32_t regexps, uint32_t pitch = kRotPitch,
                                                uint32_t compactBytes = 0)
{
	LdsLayout l;
	l.pitch = pitch;
	l.hotBytes = ((hot + 1) * pitch + 15) / 16 * 16;
	l.flagsOff = l.hotBytes;
	l.cls8Off = l.flagsOff + 256;
	l.compactOff = l.cls8Off + 256;
	l.compactBytes = (compactBytes + 15) / 16 * 16;
	l.clsOff = l.compactOff + l.compactBytes;
	l.countsOff = l.clsOff + 528;
	l.histOff = l.countsOff + ((regexps + 2) * 4 + 15) / 16 * 16;
	l.progOff = l.histOff + 1024;   // 16 B: the tiled kernel's block-wide progress counter
	l.total = l.progOff + 16;
	l.rot2 = 0;
	return l;
}

// Compact tier geometry: row = `letters` u16 entries + 1 u16 holding the row's own state id, padded to 4 bytes.
__host__ __device__ inline uint32_t CompactPitch(uint32_t letters) { return ((letters + 1) * 2 + 3) / 4 * 4; }

// How many states get a compact row next to everything else any kernel keeps in LDS (0 = tier off).
inline uint32_t CompactCapacity(uint32_t hot, uint32_t letters, uint32_t regexps, uint32_t states)
{
	if (letters > 127)
		return 0;   // the per-byte class table holds 2*class in a u8
	const uint32_t base = MakeLayout(hot, regexps <= kMaxLdsCountRegexps ? regexps : 0, 256u).total + kRaggedLdsExtra;
	if (base >= kLdsPerBlock)
		return 0;
	const uint32_t rows = (kLdsPerBlock - base) / CompactPitch(letters);
	if (rows < hot + 2 || states <= hot)
		return 0;   // no room beyond the hot states, or nothing beyond them
	return rows - 1 < states ? rows - 1 : states;
}

// ---- the wide walk (wide.hip, round 5) ---------------------------------------------------------------------------------
// Tables whose scans keep leaving the 255 dense rows take a walk of their own: NO dense rows, the whole LDS of the CU
// holds class-indexed u16 rows of the first `wide` states of the ranking (multi.h:169-192 as it stands: letter =
// m_letters[ch], state = row[letter]).  A row is
//     u16 next[letters]   device id of the target state; `wide` (the escape row's id) for targets without a row
//     u16 flags           kFinal | kDead | kAbsorbing of the state
// + 2 bytes where that makes a multiple of 8 (WidePitch); row `wide`, the escape row, leads to itself.  cls8 (2 * letter class of every byte
// value) sits at LDS address 0, so that the byte IS the address of its class.  Behind the rows: one u32 visit counter
// per row (what pire_hip_table_adapt() ranks from).
struct WideLayout {
	uint32_t pitch;      // bytes per row
	uint32_t rowsOff;    // 256
	uint32_t rows;       // states with a row + 1 (the last one is the escape row): wide + 1, zipped image: full + 1
	uint32_t full;       // zipped image (below): states with a row of their own; else == wide
	uint32_t hOff;       // zipped image: u32 header of every state of the tier and of the escape state (wide + 1 of them)
	uint32_t xOff;       // zipped image: 3 x u16 exception targets of every state WITHOUT a row of its own (wide - full of them)
	uint32_t imageEnd;   // end of what is copied from memory (rows [+ headers + exceptions]), 16-byte aligned
	uint32_t histOff;    // u32[rows]
	uint32_t countsOff;  // (regexps + 2) u32 block-local match counters
	uint32_t progOff;    // u32 progress counter of the block's waves + u32 trap samples
	uint32_t total;
};

// (a number of HALFWORDS per row that is not a multiple of four -- an odd number of dwords, or of halfwords: rows then start
// in every LDS bank in turn, in the second case at either half of its dword -- with 18
// dwords, 34 letters, the lanes of a wave that read the same letter's entry of different rows would share 16 of the 32
// banks.  Round 5 first padded to an odd number of DWORDS, 76 bytes for 34 letters; 70 do the same for the banks and
// leave room for 8 % more rows.)
__host__ __device__ inline uint32_t WidePitch(uint32_t letters)
{
	const uint32_t halfwords = letters + 1;   // the row's entries + its flags
	return (halfwords % 4 ? halfwords : halfwords + 1) * 2;
}

// ---- the zipped image (round 6) ----------------------------------------------------------------------------------------
// A dictionary automaton's rows are nearly all "the row of a shallower state, except for the one or two letters that
// continue a word" (the failure-link structure of the reference's determinised `word1|word2|...`, samples/blacklist/
// blacklist.cpp:65-76) -- a 70-byte row per state keeps 2 207 states of dict_10k's 30 202 in a CU's LDS, and from 3 % of the
// steps outside them on the walk is bound by the L2's request rate.  The zipped image keeps a full row only
// for `full` states (<= 1 023) and, for every other state of the tier, 10 bytes:
//     u32 header    bits 22..31  index of the full row this state's row is equal to except in <= 3 letters
//                   bits 1..7 / 8..14 / 15..21  those letter classes (127 = none)  [= 2 * class * 0x4081: ONE multiply of the
//                   byte's doubled class compares all three]
//     u16 target[3] where they lead
// States with a row of their own have a header too (base = themselves, no letters): the step is the same code for every
// lane --  h = header[st];  entry address = letter matches one of h's ? &target[st][k] : &row[h.base][letter];  st = u16 at it.
// Two dependent LDS reads per byte instead of one, 3 LDS instructions instead of 2: slower while the working set fits
// the plain rows, 4 x the states in LDS when it does not.  Device numbering: full states first, then the rest of the tier.
constexpr uint32_t kZipMaxFull = 1022;      // + the escape row: 10 bits of base
constexpr uint32_t kZipNoLetter = 127;
constexpr uint32_t kZipExceptions = 3;
constexpr uint32_t kZipMulC2 = 0x4081u;     // (2 * class) * 0x4081 = class at bits 1, 8, 15

__host__ __device__ inline WideLayout MakeWideLayout(uint32_t wide, uint32_t letters, uint32_t regexps, uint32_t zipFull = 0)
{
	WideLayout w;
	w.pitch = WidePitch(letters);
	w.rowsOff = 256;
	w.full = zipFull ? zipFull : wide;
	w.rows = w.full + 1;
	w.hOff = (w.rowsOff + w.rows * w.pitch + 3) / 4 * 4;
	w.xOff = w.hOff + (zipFull ? (wide + 1) * 4 : 0);
	w.imageEnd = (w.xOff + (zipFull ? (wide - zipFull) * 2 * kZipExceptions : 0) + 15) / 16 * 16;
	w.histOff = w.imageEnd;
	w.countsOff = (w.histOff + w.rows * 4 + 15) / 16 * 16;
	w.progOff = w.countsOff + ((regexps + 2) * 4 + 15) / 16 * 16;
	w.total = w.progOff + 16;
	return w;
}

// How many states get a wide row (0: the wide walk is not for this table).
inline uint32_t WideCapacity(uint32_t letters, uint32_t regexps, uint32_t states)
{
	if (letters > 127 || states < 2)
		return 0;   // cls8 holds 2 * class in a u8
	const uint32_t fixed = MakeWideLayout(0, letters, regexps <= kMaxLdsCountRegexps ? regexps : 0).total + 64;
	if (fixed >= kLdsPerBlock)
		return 0;
	const uint32_t rows = (kLdsPerBlock - fixed) / (WidePitch(letters) + 4);
	if (rows < 2)
		return 0;
	return rows - 1 < states ? rows - 1 : states;
}

// ---- the stream kernel on the wide walk (stream.hip, round 6) ------------------------------------------------------------
// The stream kernel keeps the positions of a sub-task's strings in LDS, per wave -- with the wide walk's image beside them:
// sub-tasks of 640 strings (42 KB for 16 waves) and an image of its own with a SMALLER TIER, the first `StreamWideTier` states
// of the same numbering (entries that lead beyond it are the escape state's; a zipped image keeps its rows and drops headers).
constexpr uint32_t kStreamWideStrings = 640;   // at least (a zipped image that leaves more room gets larger sub-tasks, stream.hip)
constexpr uint32_t kStreamWideStageBytes = 16 * (kStreamWideStrings + 16) * 4;   // kStreamWaves waves
inline uint32_t StreamWideTier(uint32_t letters, uint32_t regexps, uint32_t wide, uint32_t zipFull)
{
	if (!wide)
		return 0;
	const uint32_t r = regexps <= kMaxLdsCountRegexps ? regexps : 0;
	const uint32_t budget = kLdsPerBlock - kStreamWideStageBytes - 64;
	if (MakeWideLayout(zipFull ? zipFull : 1, letters, r, zipFull).total > budget)
		return 0;   // (a zipped image whose rows alone do not leave the room)
	uint32_t lo = zipFull ? zipFull : 1, hi = wide;   // the largest tier whose layout fits (the layout grows with the tier)
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) / 2;
		if (MakeWideLayout(mid, letters, r, zipFull).total <= budget)
			lo = mid;
		else
			hi = mid - 1;
	}
	return lo;
}

// Host-side, fully decoded scanner.  States are in the REFERENCE's numbering ("orig") unless a name says perm.
struct HostTable {
	// geometry (mirrors Scanner::Locals, multi.h:315-323)
	uint32_t states = 0, letters = 0, regexps = 0, initial = 0;
	bool empty = false;
	bool ranked = true;               // false: hot / permutation / dense rows not computed yet (EnsureRanked)
	uint32_t scannerType = 1;         // ScannerIOTypes (common.h:34-40): 1 Scanner, 2 SimpleScanner
	uint32_t headerSize = 0, rowStride = 0;
	uint64_t refBufSize = 0;
	uint64_t blobBytes = 0;           // bytes of the serialised image this scanner occupied (what Mmap() would consume)

	std::vector<uint16_t> cls;        // [264] letter class of each Char (Translate() - HEADER_SIZE)
	std::vector<uint32_t> next;       // [states * letters] next state index, orig numbering
	std::vector<uint8_t> flags;       // [states] kFinal|kDead|kAbsorbing
	std::vector<uint64_t> acceptOff;  // [states + 1] CSR into acceptIds
	std::vector<uint64_t> acceptIds;  // AcceptedRegexps lists (multi.h:149-158)

	// device numbering: hot states first ("perm" ids).  permOfOrig / origOfPerm are inverse permutations.
	std::vector<uint32_t> permOfOrig, origOfPerm;
	uint32_t hot = 0;                 // number of states with a dense LDS row; trap id == hot
	uint32_t hotFinalLo = 0;          // hot perm ids >= this are Final (the hot set is ordered non-final first)
	uint32_t hotDeadLo = 0;           // hot perm ids in [hotDeadLo, hotFinalLo) are Dead (ordered plain, Dead, Final)
	float deadShare = 0, finalShare = 0;   // share of the byte model's visits that fall on Dead / Final states
	float topShare = 1;               // share of the ranking's mass on its most visited state (1 = all lanes in one row)
	// steps (text bytes only, capped at 255) from every state to the nearest Final / Final-or-Dead state: the ragged
	// kernel with actions skips the exact re-walk of a trapped chunk that cannot reach one (table.cpp EnsureActDist)
	std::vector<uint8_t> distFinal, distFlagged;   // [states], reference numbering; empty until first needed
	bool incPacked = false;           // regexps <= 8 and every final-list multiplicity <= 255: inc64 is usable
	std::vector<uint64_t> inc64;      // [states] (orig numbering) byte r = how often regexp r is in the final list
	uint32_t compact = 0;             // perm ids [0, compact) also have a class-indexed u16 row in LDS (tiled/ragged kernels)
	uint32_t wide = 0;                // perm ids [0, wide) have a row in the wide walk's LDS image (wide.hip; 0 = no such image)
	uint32_t zipFull = 0;             // != 0: the wide image is ZIPPED (MakeWideLayout): perm ids [0, zipFull) have a row of their own,
	                                  // [zipFull, wide) a header + <= 3 exceptions against the row of zipBase[id - zipFull]
	std::vector<uint16_t> zipBase;    // [wide - zipFull] perm id (< zipFull) of that row
	float offsetBatchShare = 0;       // share of the bytes scanned since the last ranking that came as offset batches (one string per
	                                  // lane: the zipped step's 20 vector instructions bound that kernel, table.cpp ChooseZip)
	float zipPlainOutside = 0;        // what the plan that chose between the two images estimated: share of the ranking's mass outside
	float zipOutside = 0;             // the plain rows / outside the zipped tier (table.cpp ChooseZip)
	float outsideDense = 0;           // share of the ranking's mass on states WITHOUT a dense row / ...
	float outsideWide = 0;            // ... without a wide row (from the byte model until adapt() has seen scans, then measured)
	bool massMeasured = false;        // those shares come from visit counters, not from the a-priori byte model
	float wideTwiceShare = 0;         // share of the wide walk's 16-byte wave-chunks it walked twice (exact, between the two most
	                                  // recent adapt() calls)
	std::vector<uint8_t> hotRows;     // [(hot + 1) * 256] u8: next perm id (< hot) or `hot` (= leaves the hot set)
	std::vector<uint8_t> hotFlags;    // [256] flags of hot perm ids (kAbsorbing used for the early-out ballot)
	std::vector<double> seenMass;     // [states] what the scans so far visited (lane-steps, halved at every adapt(); orig numbering)
	std::vector<double> priorMass;    // [states] expected visits under the byte model, max-normalised (orig numbering)
	uint64_t lastTrapSamples = 0;     // cold-state samples seen by the most recent pire_hip_table_adapt()
	uint64_t lastWideTrapChunks = 0;  // 16-byte wave-chunks the wide walk walked twice, as seen by the most recent adapt() (exact)
	uint32_t adaptations = 0;
};

// Everything the end of a string needs, in ONE 16-byte load: reference StateIndex, device id + flags, regexp mask.
struct FinRec {
	uint32_t orig;        // StateIndex (multi.h:281-284) of the end state
	uint32_t permFlags;   // device id of the end state | flags << 28
	uint64_t acceptMask;  // bit r = regexp r in AcceptedRegexps() (regexps <= 64)
};

// Device image (one HIP device).
struct DeviceTable {
	int device = -1;
	uint8_t* hotRows = nullptr;       // [(hot+1)*256]
	uint8_t* hotRowsRot = nullptr;    // the same rows with their columns in rotated byte order (RotColumn)
	uint8_t* hotFlags = nullptr;      // [256]
	uint16_t* cls = nullptr;          // [264]
	uint32_t* nextPerm = nullptr;     // [states*letters], perm ids in, perm ids out
	uint8_t* flagsPerm = nullptr;     // [states]
	uint32_t* origOfPerm = nullptr;   // [states]
	uint32_t* permOfOrig = nullptr;   // [states]
	uint64_t* acceptMaskPerm = nullptr;  // [states] bit r = regexp r accepted (regexps <= 64), else null
	uint64_t* acceptOffPerm = nullptr;   // [states+1] CSR (regexps > 64)
	uint64_t* acceptIds = nullptr;
	struct FinRec* finSelf = nullptr; // [states] end-of-string record when End() is not requested
	struct FinRec* finEnd = nullptr;  // [states] end-of-string record after Step(EndMark)
	uint64_t* incPerm = nullptr;      // [states] packed per-regexp increments of HalfFinalScanner::TakeAction, or null
	uint8_t* distFinalPerm = nullptr;    // [states] HostTable::distFinal by device id (uploaded when first needed)
	uint8_t* distFlaggedPerm = nullptr;
	uint16_t* compactRows = nullptr;  // [(compact+1) rows] LDS address / 4 of the next state's row (last row = escape), padded
	uint16_t* wideRows = nullptr;     // [(wide+1) rows] the wide walk's LDS image (WideLayout), or null
	uint16_t* next16 = nullptr;       // [states*letters] nextPerm as u16 when states <= 65536 (half the L2 footprint), or null
	uint16_t* wideRowsStream = nullptr;   // the wide image once more with the stream kernel's smaller tier (StreamWideTier), or null
	uint32_t wideStream = 0;          // ... its tier
	uint32_t* visitWide = nullptr;    // [wide+1] sampled visits of the wide rows (one lane per wave per 128-byte tile)
	uint32_t* visitHot = nullptr;     // [256]    sampled visits of hot perm ids (one lane per wave per tile)
	uint32_t* visitCold = nullptr;    // [states] trapped chunks that ended in this (cold) perm id
	unsigned long long* workCounter = nullptr;   // [kWorkSlots] ragged kernel: next string range to hand out;
	                                             // one slot per launch so that launches on different streams
	                                             // never share a counter
	// Auto-adaptation signal: blocks that saw traps add them to visitHot[kTrapSlot] (device atomics) and store the new
	// total into this word of MAPPED HOST memory (a plain system-scope store: PCIe atomics are not needed, and a
	// late smaller value only delays the trigger), so the host can look at it at every launch without synchronising.
	volatile uint32_t* trapSignalHost = nullptr;
	uint32_t* trapSignalDev = nullptr;
	uint64_t bytes = 0;
};

}  // namespace pirehip

namespace pirehip {
constexpr uint32_t kWorkSlots = 1024;
// The ragged kernels take ranges of strings from a counter that must be zero when a launch starts.  Launch number k of a
// table on a device uses slot k % kWorkSlots = two words {next string, blocks done}: every block adds 1 to `done` when
// it leaves, and the block that finds itself last puts BOTH words back to zero -- so a slot is clean whenever no launch
// is using it, whatever kinds of launches (tiled, generic, failed, n == 0) took slot numbers in between, and no launch
// pays a memset dispatch of its own (5 us in front of every ragged kernel).  Round 2 had launch k clear the slot of
// launch k + kWorkSlots instead, which left a slot dirty when that later launch number went to a kernel that does not
// clear.  The array is zeroed when the image is uploaded.
inline unsigned long long* WorkSlotOf(unsigned long long* base, uint32_t k)
{
	return base + 2 * (k % kWorkSlots);
}
}  // namespace pirehip

namespace pirehip { constexpr int kMaxDevices = 64; }

namespace pirehip {
// HalfFinalScanner counting on CountingRowKernel (counting.hip): the table as letter-indexed rows with the increments
// of the TARGET state as the step's action, reference numbering (adaptation does not touch it); built when first needed.
struct HalfRowsHost {
	bool tried = false;
	uint32_t nreg = 0, initialAct = 0, maxLen = 65000;
	std::vector<uint32_t> lrows, lactWords;   // as CountingHost::lrows / lactWords
	std::vector<uint8_t> letterOf, finalTag;  // [264] letter of every Char; [states] bit 0 = Final
};
struct HalfRowsDevice {
	int device = -1;
	uint32_t* lrows = nullptr;
	uint32_t* lactWords = nullptr;
	uint8_t* letterOf = nullptr;
	uint8_t* finalTag = nullptr;
};
}  // namespace pirehip

struct pire_hip_table {
	pirehip::HostTable host;
	pirehip::HalfRowsHost halfRows;
	pirehip::HalfRowsDevice halfRowsDev[pirehip::kMaxDevices];
	std::mutex halfRowsMutex;
	// One image per HIP device (devs[d].device == d once uploaded), so that one handle serves every GPU of the node,
	// from one host thread or from several.  Guarded by uploadMutex; the run entry points COPY the image's pointers
	// while holding it (UploadTable) and never look at devs[] afterwards.
	pirehip::DeviceTable devs[pirehip::kMaxDevices];
	std::mutex uploadMutex;
	std::vector<pirehip::DeviceTable> retired;   // images an AUTOMATIC adaptation replaced: alive until destroy (captured graphs)
	std::atomic<uint32_t> workSlot[pirehip::kMaxDevices] = {};   // per device image: round-robin over its counter pairs
	std::mutex segMutex;
	std::vector<uint32_t> segModes;      // segmented.hip: mode representatives (state indices) earlier calls learned
	// segmented.hip, ModeFunction: for a pair (start state, mode representative), reference numbering, the function f
	// with "mode's state = f(mode 0's state)" after any text -- or an empty vector when there is none
	struct SegModeFn {
		uint32_t a0, b0, minSteps;
		std::vector<uint32_t> f;
	};
	std::vector<SegModeFn> segModeFns;
	// segmented.hip, EnsureModeProduct: the product automaton of this table started in (a0, b0) -- the walk of two
	// modes as ONE walk -- with the two components of every product state (reference numbering, both)
	std::unique_ptr<pire_hip_table> segProduct;
	uint32_t segProductA0 = 0, segProductB0 = 0;
	bool segProductTried = false;
	std::vector<uint32_t> segProductA, segProductB;
	// Adaptation (table.cpp AdaptTable) rewrites host.{hot, origOfPerm, permOfOrig, hotRows, ...} and replaces the images.
	// Run entry points hold adaptMutex SHARED from their first look at the table until they return (internal.h TableUse);
	// an adaptation -- the caller's or the automatic one -- holds it exclusively, drains the devices and frees the old
	// images: no call can be between "copied the pointers" and "enqueued its kernels" at that moment.
	std::shared_mutex adaptMutex;
	std::atomic<uint32_t> autoAdapts{0};
	std::atomic<uint64_t> wideLaunched{0};   // wave-chunks handed to the wide walk since the last adapt()
	std::atomic<uint64_t> bytesScanned{0};   // text bytes handed to the kernels of pire_hip_run[_strided] since the last ranking (where the host knows)
	std::atomic<uint64_t> bytesNominal{0};   // ... the same with 48 bytes a string where it does not (offsets on the device), and
	std::atomic<uint64_t> bytesOffsetBatches{0};   // those of them that came as offset batches (the ragged / stream kernels): what
	                                               // ChooseZip weighs the zipped image by (HostTable::offsetBatchShare)
	// the adaptation a call that only enqueues starts in the background (table.cpp BackgroundAdaptStep): state 0 idle, 1 the
	// worker is on its way, 2 `host` + `image` (device `device`) are ready to be swapped in at a launch boundary
	struct Background {
		std::mutex mutex;                 // start / join / swap, one at a time
		std::thread thread;
		std::atomic<int> state{0};
		std::unique_ptr<pirehip::HostTable> host;
		pirehip::DeviceTable image;
		int device = -1;
		uint64_t trapsAtLastLook = 0;     // the trap signal when a worker last found nothing to do
		std::atomic<uint32_t> swaps{0};
	} bg;
	// pire_hip_table_config_set(): this table's own configuration (a second user of the library in the same process -- another
	// component, another thread pool -- sets the process-wide one for itself).  Written under adaptMutex held
	// exclusively, read by the entry points under TableUse.
	bool hasConfig = false;
	pire_hip_config config = {};
	std::atomic<uint32_t> selfTested[pirehip::kMaxDevices] = {};   // per device, bit k: Dispatch's kernel kind k passed its known-answer
	                                                               // batch on this table there (api.cpp SelfTest)
};
namespace pirehip { constexpr uint32_t kMaxAutoAdapts = 6; }

namespace pirehip {

// Kernel parameter block (passed by value).
struct ScanParams {
	// table
	const uint8_t* hotRows;
	const uint8_t* hotRowsRot;   // host side only (LaunchTiled swaps it in for the rotated-column kernel)
	float topShare;              // host side only: share of the walk's steps the most visited state carries (ranking's estimate)
	const uint8_t* hotFlags;
	const uint16_t* cls;
	const uint32_t* nextPerm;
	const uint8_t* flagsPerm;
	const uint32_t* origOfPerm;
	const uint32_t* permOfOrig;
	const uint64_t* acceptMaskPerm;
	const uint64_t* acceptOffPerm;
	const uint64_t* acceptIds;
	const FinRec* finSelf;
	const FinRec* finEnd;
	uint32_t* visitHot;
	uint32_t* visitCold;
	uint32_t* trapSignal;   // mapped host word (device address) for the auto-adaptation policy, or null
	const uint16_t* compactRows;
	uint32_t compact;        // 0 = tier off
	const uint16_t* wideRows;   // nullable: the wide walk's LDS image
	const uint16_t* next16;     // nullable
	uint32_t* visitWide;
	uint32_t wide;              // states with a wide row; 0 = no image
	uint32_t zipFull;           // != 0: the image is zipped (internal.h MakeWideLayout), this many states have a row of their own
	uint32_t wideOutSlot;       // visitWide[wideOutSlot] counts the visit samples that found their lane outside the tier (== wide,
	                            // except under the stream kernel's image, whose tier is a prefix of the table's)
	const uint16_t* wideRowsStream;   // host side only: the stream kernel's image (LaunchStreamWide swaps it in) and its tier
	uint32_t wideStream;
	float outsideDense, outsideWide;   // host side only: LaunchTiled's choice between the dense and the wide walk
	uint32_t forceLanes;               // host side only: 1 / 2 = LaunchWide takes the kernel with that many strings per lane (the self-test)
	std::atomic<uint64_t>* wideLaunched;   // host side only: wave-chunks handed to the wide walk since the last adapt()
	bool massMeasured;                 // host side only
	const uint64_t* incPerm; // nullable
	uint32_t hotFinalLo;
	uint32_t hotDeadLo;
	float deadShare, finalShare;   // host-side hints for the choice of kernel (any choice is correct)
	const uint8_t* actDist;        // nullable: steps to the nearest state the action cares about (ragged kernel with actions)
	uint32_t states, letters, regexps, hot;
	uint32_t startPerm;      // perm id every string starts in (Initialize(), then Begin() if requested)
	uint32_t beginCls, endCls;
	uint32_t flags;          // PIRE_HIP_RUN_*
	// batch
	const uint8_t* text;
	const uint64_t* offsets; // nullable: then string i = [i*stride, i*stride+len)
	const uint64_t* ends;    // nullable: string i = [offsets[i], ends[i]) -- strings may overlap or leave gaps (the
	                         // segments of segmented.hip); then textEnd = bytes readable at `text`
	uint64_t textEnd;
	uint64_t n, len, stride;
	const uint32_t* initIdx; // nullable
	uint32_t* outIdx;        // nullable
	uint8_t* outFinal;       // nullable
	unsigned long long* outCounts;  // nullable
	unsigned long long* workBase;   // host side only: the ring of ragged work counters of the image in use
	int workDevice;                 // host side only: the device of that image (its launch counter, internal.h WorkSlotOf)
	// HalfFinalScanner on the ragged kernel: the state every string starts its text in (device id) and the counts
	// Initialize() [+ Step(BeginMark)] have already taken (half_final.h:137-164) -- the same for every string, so the
	// host walks those two steps once instead of every lane doing them (two dependent loads) for every string
	uint32_t hfStart;
	uint32_t hfStartC[8];
	// host side only, filled by FillParams under the entry point's TableUse (below) and valid while it is held: the
	// numbering of the image the pointers above belong to, and that image's distance tables when the entry point asked
	// for them.  What an entry point derives after FillParams (start states of the half-final / suffix / segmented
	// scans, mode representatives) goes through these.
	const uint32_t* hostPermOfOrig;
	const uint32_t* hostOrigOfPerm;
	pire_hip_table* owner;          // the table itself (the first-use self-test walks its host image)
	const uint8_t* distFinalPerm;
	const uint8_t* distFlaggedPerm;
#ifdef PIRE_HIP_TUNING
	unsigned long long* stamps;     // timing experiments: [blocks][4] wall-clock stamps (start, table loaded, walk done, end)
#endif
};

__host__ __device__ inline uint32_t CompactBytes(const ScanParams& p)
{
	return p.compact ? (p.compact + 1) * CompactPitch(p.letters) : 0;
}

// The library configuration (include/pire_hip.h pire_hip_config): a snapshot by value, taken once per call.
pire_hip_config GetConfig();
void SetError(const std::string& msg);
// Inside a catch (...) of an extern "C" entry point: std::bad_alloc / std::length_error (blob fields that ask for
// absurd sizes) -> PIRE_HIP_ENOMEM, anything else -> PIRE_HIP_EINVAL; the message goes to pire_hip_last_error().
int HandleException() noexcept;
int HipFail(hipError_t e, const char* what);   // sets the error, returns PIRE_HIP_ENODEVICE / ENOMEM

// Owns the temporary device buffers of a host-pointer call (PCIe-inclusive convenience mode).  Default: blocks from the
// per-device cache of api.cpp (StagingAcquire / StagingRelease: no allocation in steady state); they go back when the
// call returns -- after its stream has been drained, which the destructor sees to itself: an error path may leave
// copies or kernels in flight.  pire_hip_config.host_staging: 1 = hipMalloc + hipFree per call (round 2), 2 = the
// stream-ordered pool.
int StagingAcquire(size_t bytes, void** out, size_t* blockBytes);
void StagingRelease(void* p, size_t blockBytes);
// the same for PINNED host blocks: small inputs and results cross PCIe from / into pinned memory (a copy between
// pageable memory and the device goes through the runtime's own staging and costs 10-15 us a piece, which is most of
// a 10-string call), and the caller's pageable arrays are touched with plain memcpy
int StagingAcquireHost(size_t bytes, void** out, size_t* blockBytes);
void StagingReleaseHost(void* p, size_t blockBytes);
struct Staging {
	std::vector<void*> ptrs, hostPtrs;
	std::vector<size_t> sizes, hostSizes;
	struct Deferred {
		void* dst;
		const void* pinned;
		size_t bytes;
	};
	std::vector<Deferred> deferred;   // results waiting in pinned blocks for Finish()
	// Small calls (round 3): every input and result of the call is carved out of ONE device block and its pinned twin
	// (same offsets), so that the inputs cross PCIe in one copy and the results in one -- two copies enqueued per call
	// instead of four to six, at 4-5 us of stream time each (profiles/r03_small_call_timeline.log).  Inputs wait in the
	// pinned block until Flush() (called by the first Alloc() / Out() / Finish() at the latest: every user stages its
	// inputs first, then its result arrays, then launches); results are fetched by Finish().
	static constexpr size_t kArenaBytes = size_t(256) << 10;
	uint8_t* arenaDev = nullptr;
	uint8_t* arenaPin = nullptr;
	size_t arenaUsed = 0, arenaInputs = 0, arenaFlushed = 0;
	bool arenaTried = false;
	struct ArenaOut {
		void* dst;
		size_t off, bytes;
	};
	std::vector<ArenaOut> arenaOuts;
	hipStream_t stream = nullptr;
	uint32_t mode = 1;   // a default-constructed Staging is round 2's
	bool drained = false;
	static constexpr size_t kPinnedMax = size_t(1) << 20;   // larger pieces go straight from / to the caller's memory
	Staging() {}
	explicit Staging(hipStream_t s) : stream(s), mode(GetConfig().host_staging) {}
	~Staging()
	{
		if (mode == 0 && !drained && (!ptrs.empty() || !hostPtrs.empty()))
			(void)hipStreamSynchronize(stream);
		for (size_t i = 0; i < ptrs.size(); ++i) {
			if (mode == 0)
				StagingRelease(ptrs[i], sizes[i]);
			else if (mode == 2)
				(void)hipFreeAsync(ptrs[i], stream);
			else
				(void)hipFree(ptrs[i]);
		}
		for (size_t i = 0; i < hostPtrs.size(); ++i)
			StagingReleaseHost(hostPtrs[i], hostSizes[i]);
	}
	// a piece of the arena, or null when the call is not a small one (any more)
	uint8_t* Carve(size_t bytes)
	{
		if (mode != 0)
			return nullptr;
		if (!arenaTried) {
			arenaTried = true;
			void *d = nullptr, *h = nullptr;
			if (Alloc(&d, kArenaBytes, /*fromArena=*/false) != PIRE_HIP_OK || Pinned(&h, kArenaBytes) != PIRE_HIP_OK)
				return nullptr;
			arenaDev = static_cast<uint8_t*>(d);
			arenaPin = static_cast<uint8_t*>(h);
		}
		const size_t need = (std::max<size_t>(bytes, 16) + 255) & ~size_t(255);
		if (!arenaDev || !arenaPin || arenaUsed + need > kArenaBytes)
			return nullptr;
		uint8_t* p = arenaDev + arenaUsed;
		arenaUsed += need;
		return p;
	}
	// the inputs staged so far: one copy
	int Flush()
	{
		if (arenaInputs > arenaFlushed) {
			const hipError_t e = hipMemcpyAsync(arenaDev + arenaFlushed, arenaPin + arenaFlushed, arenaInputs - arenaFlushed,
			                                    hipMemcpyHostToDevice, stream);
			if (e != hipSuccess)
				return HipFail(e, "hipMemcpy(H2D)");
			arenaFlushed = arenaInputs;
		}
		return PIRE_HIP_OK;
	}
	int Alloc(void** out, size_t bytes, bool fromArena = true)
	{
		*out = nullptr;
		if (fromArena) {
			if (int rc = Flush())
				return rc;
			if (uint8_t* p = Carve(bytes)) {
				*out = p;
				return PIRE_HIP_OK;
			}
		}
		size_t block = bytes ? bytes : 16;
		if (mode == 0) {
			if (int rc = StagingAcquire(block, out, &block))
				return rc;
		} else {
			const hipError_t e = mode == 2 ? hipMallocAsync(out, block, stream) : hipMalloc(out, block);
			if (e != hipSuccess)
				return HipFail(e, "hipMalloc(staging)");
		}
		ptrs.push_back(*out);
		sizes.push_back(block);
		return PIRE_HIP_OK;
	}
	int Pinned(void** out, size_t bytes)
	{
		size_t block = 0;
		if (int rc = StagingAcquireHost(bytes, out, &block))
			return rc;
		hostPtrs.push_back(*out);
		hostSizes.push_back(block);
		return PIRE_HIP_OK;
	}
	template <class T>
	int In(const T* host, size_t count, const T** dev, hipStream_t s)
	{
		void* d;
		const size_t bytes = count * sizeof(T);
		if (s == stream && arenaInputs == arenaUsed)   // inputs only in front of everything else of the arena
			if (uint8_t* p = Carve(bytes)) {
				if (bytes)   // (an empty input may come with a null pointer: memcpy's arguments are declared non-null, UBSan r4)
					memcpy(arenaPin + (p - arenaDev), host, bytes);
				arenaInputs = arenaUsed;
				*dev = reinterpret_cast<const T*>(p);
				return PIRE_HIP_OK;
			}
		if (int rc = Alloc(&d, bytes, /*fromArena=*/false))
			return rc;
		if (count) {
			const void* src = host;
			if (mode == 0 && bytes <= kPinnedMax) {
				void* pin = nullptr;
				if (int rc = Pinned(&pin, bytes))
					return rc;
				memcpy(pin, host, bytes);
				src = pin;
			}
			hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s);
			if (e != hipSuccess)
				return HipFail(e, "hipMemcpy(H2D)");
		}
		*dev = static_cast<const T*>(d);
		return PIRE_HIP_OK;
	}
	// Result `bytes` from device memory into the caller's array: enqueued now, in the caller's array after Finish()
	int Out(void* hostDst, const void* dev, size_t bytes)
	{
		if (!bytes || !hostDst)
			return PIRE_HIP_OK;
		if (int rc = Flush())
			return rc;
		const uint8_t* dp = static_cast<const uint8_t*>(dev);
		if (arenaDev && dp >= arenaDev && dp + bytes <= arenaDev + kArenaBytes) {
			arenaOuts.push_back(ArenaOut{hostDst, size_t(dp - arenaDev), bytes});
			return PIRE_HIP_OK;
		}
		void* dst = hostDst;
		if (mode == 0 && bytes <= kPinnedMax) {
			void* pin = nullptr;
			if (int rc = Pinned(&pin, bytes))
				return rc;
			deferred.push_back(Deferred{hostDst, pin, bytes});
			dst = pin;
		}
		const hipError_t e = hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, stream);
		return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemcpy(D2H)");
	}
	// Drain the stream, then hand the results that waited in pinned blocks to the caller's arrays
	int Finish()
	{
		if (int rc = Flush())
			return rc;
		if (!arenaOuts.empty()) {
			size_t lo = kArenaBytes, hi = 0;
			for (const ArenaOut& o : arenaOuts) {
				lo = std::min(lo, o.off);
				hi = std::max(hi, o.off + o.bytes);
			}
			const hipError_t ce = hipMemcpyAsync(arenaPin + lo, arenaDev + lo, hi - lo, hipMemcpyDeviceToHost, stream);
			if (ce != hipSuccess)
				return HipFail(ce, "hipMemcpy(D2H)");
		}
		const hipError_t e = hipStreamSynchronize(stream);
		if (e != hipSuccess)
			return HipFail(e, "copy back / synchronize");
		drained = true;
		for (const Deferred& d : deferred)
			memcpy(d.dst, d.pinned, d.bytes);
		deferred.clear();
		for (const ArenaOut& o : arenaOuts)
			memcpy(o.dst, arenaPin + o.off, o.bytes);
		arenaOuts.clear();
		return PIRE_HIP_OK;
	}
};


// table.cpp
int BuildHostTable(const void* blob, size_t len, HostTable* out);
int UploadTable(pire_hip_table* t, DeviceTable* image);   // image of the CURRENT device (built on first use), copied out
int BuildDeviceImage(const HostTable& h, int dev, DeviceTable* out);   // allocations + copies of a ranked table's image, current device
void JoinBackgroundAdapt(pire_hip_table* t);   // waits for an adaptation in the background and drops what it prepared
void EnsureRanked(pire_hip_table* t);
std::vector<uint16_t> BuildWideRows(const HostTable& h, uint32_t tier = 0);   // the wide walk's LDS image (WideLayout), current numbering; tier != 0: of the first `tier` states only
// after UploadTable, current device: the per-state distance tables (built on first use)
int EnsureActDist(pire_hip_table* t, const uint8_t** distFinalPerm, const uint8_t** distFlaggedPerm);
void FreeAllDeviceTables(pire_hip_table* t);
struct GlueProduct {
	std::vector<std::pair<uint32_t, uint32_t>> states;   // numbered product states (lhs state, rhs state)
	std::vector<uint32_t> next;                          // [states * letters]
	bool failed = false;                                 // more than maxSize new states: the glue yields an empty scanner
};
int GlueBfsHost(const HostTable& a, const HostTable& b, const std::vector<uint32_t>& la, const std::vector<uint32_t>& lb,
                size_t maxSize, GlueProduct* out);
int GlueBfsDevice(const HostTable& a, const HostTable& b, const std::vector<uint32_t>& la,
                  const std::vector<uint32_t>& lb, size_t maxSize, GlueProduct* out);   // glue.hip
int GlueHostTables(const HostTable& a, const HostTable& b, size_t maxSize, HostTable* out, bool onDevice = false);
int AdaptTable(pire_hip_table* t, uint32_t* changedRows, bool automatic = false);
// the auto-adaptation policy (pire_hip_config.auto_adapt): called at every launch boundary, cheap when nothing is due
void MaybeAutoAdapt(pire_hip_table* t, bool enqueueOnly = false);
int CheckFailures(pire_hip_table* t, uint64_t* out);
void FreeDeviceTable(DeviceTable* d);

// tiled.hip / ragged.hip / exact.hip / corpus.hip
int LaunchGeneric(const ScanParams& p, hipStream_t stream);
int LaunchTiled(const ScanParams& p, hipStream_t stream);
// wide.hip: fixed-length records through the class-indexed walk (tables whose scans keep leaving the dense rows)
bool WideWanted(const ScanParams& p, const pire_hip_config& cfg);
int LaunchWide(const ScanParams& p, hipStream_t stream);
bool TiledEligible(const ScanParams& p);
bool RaggedEligible(const ScanParams& p, uint64_t totalBytesHint);
int LaunchRagged(const ScanParams& p, unsigned long long* workCounter, hipStream_t stream);
int LaunchRaggedWide(const ScanParams& p, unsigned long long* workCounter, hipStream_t stream);   // ... on the class-indexed walk
// stream.hip: offset batches of many short strings, every lane a run of consecutive strings
bool StreamEligible(const ScanParams& p, uint64_t totalBytesHint);
int LaunchStream(const ScanParams& p, hipStream_t stream);
bool StreamWideEligible(const ScanParams& p, uint64_t totalBytesHint);   // ... on the class-indexed walk (offset batches of a wide table)
int LaunchStreamWide(const ScanParams& p, hipStream_t stream);
// segmented.hip: few long strings, cut into segments that are scanned in parallel (speculatively; the chain of
// segments is then followed on the host, so the call synchronises its stream)
bool SegmentedEligible(uint64_t n, uint64_t totalBytes);
int RunSegmented(pire_hip_table* t, const ScanParams& p, const uint64_t* hostOffsets, hipStream_t stream,
                 uint32_t* halfFinalResults = nullptr,    // non-null: also HalfFinalScanner counts, [n][regexps] ...
                 bool* halfFinalIncomplete = nullptr);    // ... unless some string ended in the plain walk (then true)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the occupancy query cost the host several microseconds per
// launch; what they say does not change: remembered per device and kernel (the LDS limit as a high-water mark).
struct LaunchCache {
	struct Key {
		int dev;
		const void* fn;
		int threads;
		uint32_t lds;
		bool operator==(const Key& o) const { return dev == o.dev && fn == o.fn && threads == o.threads && lds == o.lds; }
	};
	struct Entry {
		Key key;
		int value;
	};
	std::mutex mutex;
	std::vector<Entry> ldsMax, occupancy;
	static LaunchCache& Get()
	{
		static LaunchCache c;
		return c;
	}
};

inline hipError_t SetDynamicLds(const void* fn, uint32_t ldsBytes)
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return e;
	LaunchCache& c = LaunchCache::Get();
	std::lock_guard<std::mutex> lock(c.mutex);
	LaunchCache::Entry* found = nullptr;
	for (auto& x : c.ldsMax)
		if (x.key.dev == dev && x.key.fn == fn)
			found = &x;
	if (found && uint32_t(found->value) >= ldsBytes)
		return hipSuccess;
	e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
	if (e != hipSuccess)
		return e;
	if (found)
		found->value = int(ldsBytes);
	else
		c.ldsMax.push_back({{dev, fn, 0, 0}, int(ldsBytes)});
	return hipSuccess;
}

// order.hip: the strings of an offset batch ordered by length class (longest first), built on the device, for the
// one-string-per-lane kernels with per-byte actions.  `scratch`: LengthOrderScratchBytes(n) of device memory that stays
// valid until the kernels that read *perm have run.
// The k of pass `pass` (k = pass * lanes + t, or the same range backwards in odd passes when `serpentine`), for the
// grid-stride loops of those kernels; >= n: nothing for this lane in this pass.
__host__ __device__ inline uint64_t OrderedIndex(uint64_t pass, uint64_t t, uint64_t lanes, bool serpentine)
{
	return pass * lanes + ((serpentine && (pass & 1)) ? lanes - 1 - t : t);
}
bool LengthOrderWanted(uint64_t n);
size_t LengthOrderScratchBytes(uint64_t n);
// *serpentine: whether the kernels should walk the order with OrderedIndex's serpentine (the global order) or plainly
int BuildLengthOrder(const uint64_t* offsets, uint64_t n, void* scratch, hipStream_t stream, const uint32_t** perm, bool* serpentine);
void NoteKernel(const char* name, const char* symbol = nullptr);   // what pire_hip_last_kernel[_symbol]() report (thread local)
bool RaggedActEligible(const ScanParams& p);
int LaunchRaggedHalfFinal(const ScanParams& p, unsigned long long* workCounter, uint32_t* outResults, hipStream_t stream);
// CapturingScanner on the ragged kernel with actions (counting.hip builds the table): `info` = per state of the
// EXPANDED automaton (reference numbering of that table): original state << 8 | Final tag << 2 | action that entered it
int LaunchRaggedCapture(const ScanParams& p, unsigned long long* workCounter, const uint32_t* info, long long* outBegin,
                        long long* outEnd, hipStream_t stream);
// api.cpp, for the other translation units: the scan parameters of a table on the current device (uploads the image,
// auto-adapts at the launch boundary), one ragged work slot of it, a table handle around a host table built elsewhere
// Holds a table still for ONE entry point: the automatic adaptation's turn first (no lock held), then the table's
// adaptMutex SHARED until the entry point returns, i.e. until its kernels are enqueued (host-pointer forms: until they
// have run).  pire_hip_table_adapt() and the automatic adaptation take the mutex exclusively, drain the devices and only
// then replace numbering and images -- so whatever an entry point derives from t->host after FillParams belongs to the
// image its ScanParams point at, and an adaptation may run concurrently with scans on other host threads (ADVICE r3:
// round 3 dropped the lock when FillParams returned).  Two tables (pire_hip_run_pair): acquire in address order.
extern thread_local const pire_hip_config* g_cfgOverride;   // api.cpp: what GetConfig() returns on this thread when set
struct TableUse {
	std::shared_lock<std::shared_mutex> lock;
	const pire_hip_config* outer = nullptr;
	bool pushed = false;
	TableUse() {}
	TableUse(pire_hip_table* t, bool enqueueOnly) { Acquire(t, enqueueOnly); }
	TableUse(const TableUse&) = delete;
	TableUse& operator=(const TableUse&) = delete;
	~TableUse()
	{
		if (pushed)
			g_cfgOverride = outer;
	}
	void Acquire(pire_hip_table* t, bool enqueueOnly)
	{
		if (lock.owns_lock())
			return;   // a second look at the same table inside one entry point (an adaptation would wait for this very lock)
		MaybeAutoAdapt(t, enqueueOnly);
		lock = std::shared_lock<std::shared_mutex>(t->adaptMutex);
		// the table's own configuration, if it has one, for everything this entry point does on this thread (a first-use
		// self-test in progress keeps its own: it already started from the table's)
		if (t->hasConfig && !g_cfgOverride) {
			outer = g_cfgOverride;
			g_cfgOverride = &t->config;
			pushed = true;
		}
	}
};
// for the other translation units: TableUse::Acquire(t) + FillParams
int PrepareScanParams(pire_hip_table* t, ScanParams* p, uint32_t flags, TableUse* use, bool wantDist = false, bool enqueueOnly = false);
unsigned long long* TakeWorkSlot(pire_hip_table* t, const ScanParams& p);
void ChooseHotAndPermuteExported(HostTable& t);
int LaunchRaggedPrefix(const ScanParams& p, unsigned long long* workCounter, bool longest, bool throughEnd,
                       long long* outLen, hipStream_t stream);
int LaunchStep(const ScanParams& p, uint32_t* stateIdx, uint64_t n, uint32_t cls, hipStream_t stream);
int LaunchHalfFinal(const ScanParams& p, uint32_t* outResults, hipStream_t stream, unsigned long long* workCounter,
                    const uint32_t* list = nullptr);
// counting.hip: dense HalfFinal counting on the row kernel.  *done = false: not this table / batch.  Strings the 16-bit
// counters cannot hold are left on `*overflow` (device: [0] = count, then string indices; stream-ordered memory the
// caller frees) for LaunchHalfFinal(.., list).
int LaunchHalfFinalRows(pire_hip_table* t, const uint8_t* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                        uint32_t* outIdx, uint8_t* outFinal, uint32_t* outResults, hipStream_t stream, bool* done,
                        uint32_t** overflow);
void FreeHalfRows(pire_hip_table* t);
int UploadHalfRows(pire_hip_table* t);   // counting.hip: the dense HalfFinal row image of the current device (first need / table_upload)
int LaunchPrefix(const ScanParams& p, bool longest, bool throughEnd, long long* outLen, hipStream_t stream,
                 unsigned long long* workCounter);
// pair.hip: two scanners in one pass over fixed-length records (run.h:229-241)
bool PairTiledEligible(const ScanParams& a, const ScanParams& b);
// the tiled kernel for the segmented scan's grid segments, warm-up inside the pass (tiled.hip, TiledSegParams)
int LaunchTiledSeg(const ScanParams& p, uint64_t warmBytes, const uint32_t* segJ, uint32_t* guess, hipStream_t stream);
// guessA != nullptr: the segmented scan's form (pair.hip, PairParams): warm-up inside the pass, the guesses written out
int LaunchPairTiled(const ScanParams& a, const ScanParams& b, uint32_t* outIdxB, hipStream_t stream, uint64_t warmBytes = 0,
                    const uint32_t* segJ = nullptr, uint32_t* guessA = nullptr, uint32_t* guessB = nullptr);
int LaunchOrFinal(uint8_t* fin, const uint8_t* other, uint64_t n, hipStream_t stream);
int LaunchSuffix(const ScanParams& p, bool longest, bool throughBegin, long long* outLen, hipStream_t stream);
int LaunchCorpusFill(uint8_t* out, uint64_t seed, uint64_t first, uint64_t count, uint64_t len, uint64_t stride,
                     const void* plantsHost, hipStream_t stream);

}  // namespace pirehip

