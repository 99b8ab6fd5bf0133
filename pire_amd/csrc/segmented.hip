// Long strings: a DFA walk is sequential, and one string per lane means that a few long strings leave the chip idle
// (one lane walks ~24 MB/s; a single 1 GiB string would take 45 s).  This file cuts long strings into segments and
// scans the segments in parallel -- speculatively, because the start state of a segment is the end state of the one
// before it -- and then follows the chain of segments on the host, accepting only what was computed from the state
// the chain is really in.
//
//   guess   Regexp automata mostly forget: after a few dozen bytes the state rarely depends on where the walk began.
//           So a segment guesses its start state by walking the W bytes in front of it (its warm-up) from the
//           string's start state.  What automata do NOT forget are their sticky modes ("hello\s+w" was seen, a
//           non-printable byte was seen): after the first trigger every such guess is wrong.  A MODE is therefore a
//           representative state r: under mode r a segment's guess is the warm-up walked from r.  Mode 0 is the
//           string's start state; further modes are learned from the states the chain finds itself in unexpectedly
//           (two modes cover 99.96-100 % of the segments of the benchmark tables).
//   scan    per mode one batch through the ordinary kernels: every segment from its guess -> its end state.
//   chain   the host walks each string's segments: in state `cur` at the start of segment k it looks for a mode
//           whose guess for k IS cur and takes that mode's end state -- an exact result, because a DFA step depends
//           on nothing but the state and the bytes.  No mode matches: segment k is scanned from `cur` on the
//           device (all strings' pending segments in one small batch), a state that keeps turning up becomes a
//           mode, and when the budget of such round trips is spent the rest of the string is walked the plain way.
//           So the result is exact whatever the automaton; the speed depends on how well it forgets.
//   finish  End(), outputs and match counters exactly like the other kernels.
//
// Results are the reference's (run.h:271-275 walks the same bytes in the same order); only the schedule differs.
// The call synchronises its stream (the chain runs on the host).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "device_common.h"

namespace pirehip {

namespace {

struct SegGeometry {
	uint64_t segBytes, warmBytes;
	uint64_t nStrings, nSeg;
	uint64_t segsPerString;        // strided batches: every string has this many segments
	const uint32_t* strFirst;      // offset batches: [nStrings + 1] first segment of every string (device)
};

struct SegArrays {
	uint64_t* segBegin;            // [nSeg] byte offsets into text
	uint64_t* segEnd;
	uint64_t* warmBegin;
	uint32_t* initSeg;             // [nSeg] caller's resume state of the segment's string (only with init states)
};

__device__ __forceinline__ void StringOfSegment(const SegGeometry& g, uint64_t seg, uint64_t* str, uint64_t* j)
{
	if (!g.strFirst) {
		*str = seg / g.segsPerString;
		*j = seg - *str * g.segsPerString;
		return;
	}
	uint64_t lo = 0, hi = g.nStrings;          // last string whose first segment is <= seg
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) / 2;
		if (g.strFirst[mid] <= seg)
			lo = mid;
		else
			hi = mid;
	}
	*str = lo;
	*j = seg - g.strFirst[lo];
}

__global__ void SegmentPrepKernel(ScanParams p, SegGeometry g, SegArrays a)
{
	const uint64_t seg = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (seg >= g.nSeg)
		return;
	uint64_t s, j;
	StringOfSegment(g, seg, &s, &j);
	uint64_t B, E;
	if (p.offsets) {
		B = p.offsets[s];
		E = p.offsets[s + 1];
	} else {
		B = s * p.stride;
		E = B + p.len;
	}
	const uint64_t b = B + j * g.segBytes;
	const uint64_t e = b + g.segBytes < E ? b + g.segBytes : E;
	const uint64_t w = j == 0 ? 0 : (b - B < g.warmBytes ? b - B : g.warmBytes);
	a.segBegin[seg] = b;
	a.segEnd[seg] = e;
	a.warmBegin[seg] = b - w;
	if (a.initSeg)
		a.initSeg[seg] = p.initIdx[s];
}

// finish: one lane per string; endIdx[s] = the state index (reference numbering) string s ended in, before End().
__global__ __launch_bounds__(1024) void SegmentFinishKernel(ScanParams p, const uint32_t* endIdx)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, kRotPitch, 0);
	LoadTableToLds(p, lds, L);
	const uint64_t rounds = (p.n + 63) / 64;
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t task = uint64_t(blockIdx.x) * wavesPerBlock + (threadIdx.x >> 6); task < rounds;
	     task += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t s = task * 64 + lane;
		const bool active = s < p.n;
		Finish(p, lds, L, s, active, active ? p.permOfOrig[endIdx[s]] : 0u);
	}
	FlushCounts(p, lds, L);
}

uint64_t EnvBytes(const char* name, uint64_t fallback)
{
	const char* v = getenv(name);
	return v && *v ? uint64_t(strtoull(v, nullptr, 10)) : fallback;
}

// Stream-ordered scratch: freed on the stream when the call returns, i.e. after everything enqueued before.
struct StreamScratch {
	hipStream_t stream;
	std::vector<void*> ptrs;
	explicit StreamScratch(hipStream_t s) : stream(s) {}
	~StreamScratch()
	{
		for (void* p : ptrs)
			(void)hipFreeAsync(p, stream);
	}
	template <class T>
	int Alloc(T** out, size_t count)
	{
		void* d = nullptr;
		hipError_t e = hipMallocAsync(&d, std::max<size_t>(count * sizeof(T), 16), stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(segments)");
		ptrs.push_back(d);
		*out = static_cast<T*>(d);
		return PIRE_HIP_OK;
	}
};

int ScanBatch(const ScanParams& q, pire_hip_table* t, hipStream_t stream)
{
	if (RaggedEligible(q, ~0ull))
		return LaunchRagged(q, t->dev.workCounter + t->workSlot.fetch_add(1) % kWorkSlots, stream);
	return LaunchGeneric(q, stream);
}

}  // namespace

// Worth it when the lanes would starve: few strings, long ones.  PIRE_HIP_SEGMENT_BYTES forces the mode (tests).
bool SegmentedEligible(uint64_t n, uint64_t totalBytes)
{
	if (getenv("PIRE_HIP_NO_SEGMENTS") || n == 0 || n >= (1ull << 31))
		return false;
	if (getenv("PIRE_HIP_SEGMENT_BYTES"))
		return true;
	return n <= 32768 && totalBytes / n >= 32768;
}

namespace {

struct Mode {
	uint32_t* dGuess = nullptr;
	uint32_t* dEnd = nullptr;
	std::vector<uint32_t> guess, end;   // host copies
};

#define PIRE_TRY(expr)                                                                                                 \
	do {                                                                                                           \
		if (int rc_ = (expr))                                                                                      \
			return rc_;                                                                                            \
	} while (0)

int HipOk(hipError_t e, const char* what)
{
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, what);
}

}  // namespace

// p: the original batch with DEVICE pointers (strided, or offsets on the device); hostOffsets: the same offsets on
// the host (nullptr for strided batches) -- the host has to know the lengths to cut the strings up.
int RunSegmented(pire_hip_table* t, const ScanParams& p, const uint64_t* hostOffsets, hipStream_t stream)
{
	PIRE_TRY(CheckCounts(p));
	const uint64_t n = p.n;
	const uint64_t total = hostOffsets ? hostOffsets[n] - hostOffsets[0] : n * p.len;
	int cus = 0;
	PIRE_TRY(DeviceCUs(&cus));
	// segments: a quarter of the lanes' worth of them (the chain on the host costs per segment), a multiple of the
	// 128-byte window, long against the warm-up
	const uint64_t wanted = uint64_t(cus) * 256;
	uint64_t segBytes = std::min<uint64_t>(1u << 20, std::max<uint64_t>(4096, (total / wanted + 127) / 128 * 128));
	segBytes = EnvBytes("PIRE_HIP_SEGMENT_BYTES", segBytes);
	const uint64_t warmBytes = EnvBytes("PIRE_HIP_SEGMENT_WARMUP", 256);
	const size_t maxModes = size_t(std::max<uint64_t>(1, EnvBytes("PIRE_HIP_SEGMENT_MODES", 6)));
	uint64_t budget = EnvBytes("PIRE_HIP_SEGMENT_BUDGET", 32);   // round trips for segments no mode predicted
	const bool wantStats = getenv("PIRE_HIP_SEGMENT_STATS") != nullptr;
	if (segBytes == 0) {
		SetError("PIRE_HIP_SEGMENT_BYTES must be positive");
		return PIRE_HIP_EINVAL;
	}

	// ---- geometry
	SegGeometry g = {};
	g.segBytes = segBytes;
	g.warmBytes = warmBytes;
	g.nStrings = n;
	std::vector<uint32_t> strFirst(n + 1);
	{
		uint64_t acc = 0;
		for (uint64_t i = 0; i < n; ++i) {
			strFirst[i] = uint32_t(acc);
			const uint64_t len = hostOffsets ? hostOffsets[i + 1] - hostOffsets[i] : p.len;
			acc += std::max<uint64_t>(1, (len + segBytes - 1) / segBytes);
			if (acc >= (1ull << 31)) {
				SetError("too many segments");
				return PIRE_HIP_EINVAL;
			}
		}
		strFirst[n] = uint32_t(acc);
		g.nSeg = acc;
	}
	const uint64_t S = g.nSeg;
	StreamScratch scratch(stream);
	if (hostOffsets) {
		uint32_t* d = nullptr;
		PIRE_TRY(scratch.Alloc(&d, n + 1));
		PIRE_TRY(HipOk(hipMemcpyAsync(d, strFirst.data(), (n + 1) * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(segment index)"));
		g.strFirst = d;
	} else {
		g.segsPerString = strFirst[1] - strFirst[0];
	}
	SegArrays a = {};
	PIRE_TRY(scratch.Alloc(&a.segBegin, S));
	PIRE_TRY(scratch.Alloc(&a.segEnd, S));
	PIRE_TRY(scratch.Alloc(&a.warmBegin, S));
	if (p.initIdx)
		PIRE_TRY(scratch.Alloc(&a.initSeg, S));
	const unsigned blocks = unsigned((S + 255) / 256);
	hipLaunchKernelGGL(SegmentPrepKernel, dim3(blocks), dim3(256), 0, stream, p, g, a);
	std::vector<uint64_t> segBegin(S), segEnd(S);   // the host's copy of the cut (same arithmetic)
	for (uint64_t i = 0; i < n; ++i) {
		const uint64_t B = hostOffsets ? hostOffsets[i] : i * p.stride;
		const uint64_t E = hostOffsets ? hostOffsets[i + 1] : B + p.len;
		for (uint32_t k = strFirst[i]; k < strFirst[i + 1]; ++k) {
			segBegin[k] = B + uint64_t(k - strFirst[i]) * segBytes;
			segEnd[k] = std::min(E, segBegin[k] + segBytes);
		}
	}

	ScanParams q = p;   // the segment batches: same table, same text
	q.len = q.stride = 0;
	q.textEnd = hostOffsets ? hostOffsets[n] : (n - 1) * p.stride + p.len;
	q.outFinal = nullptr;
	q.outCounts = nullptr;

	// ---- one mode: warm-up from its representative (mode 0: the string's own start state, and the first segment's
	// warm-up is empty, so its guess is the true start state), then the scan proper from the guesses
	std::vector<Mode> modes;
	uint32_t* dConst = nullptr;
	auto addMode = [&](bool first, uint32_t representative) -> int {
		Mode m;
		PIRE_TRY(scratch.Alloc(&m.dGuess, S));
		PIRE_TRY(scratch.Alloc(&m.dEnd, S));
		q.n = S;
		q.offsets = a.warmBegin;
		q.ends = a.segBegin;
		if (first) {
			q.initIdx = a.initSeg;                       // nullable: then startPerm (Initialize + Begin folded)
			q.flags = p.flags & PIRE_HIP_RUN_BEGIN;
		} else {
			if (!dConst)
				PIRE_TRY(scratch.Alloc(&dConst, S));
			PIRE_TRY(HipOk(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dConst), int(representative), S, stream), "hipMemsetD32"));
			q.initIdx = dConst;
			q.flags = 0;
		}
		q.outIdx = m.dGuess;
		PIRE_TRY(ScanBatch(q, t, stream));
		q.offsets = a.segBegin;
		q.ends = a.segEnd;
		q.initIdx = m.dGuess;
		q.flags = 0;
		q.outIdx = m.dEnd;
		PIRE_TRY(ScanBatch(q, t, stream));
		m.guess.resize(S);
		m.end.resize(S);
		PIRE_TRY(HipOk(hipMemcpyAsync(m.guess.data(), m.dGuess, S * 4, hipMemcpyDeviceToHost, stream), "hipMemcpy(D2H)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(m.end.data(), m.dEnd, S * 4, hipMemcpyDeviceToHost, stream), "hipMemcpy(D2H)"));
		PIRE_TRY(HipOk(hipStreamSynchronize(stream), "hipStreamSynchronize"));
		modes.push_back(std::move(m));
		return PIRE_HIP_OK;
	};
	PIRE_TRY(addMode(true, 0));
	// the modes earlier calls on this table learned: the automaton is the same, the text probably similar
	std::vector<uint32_t> known;
	{
		std::lock_guard<std::mutex> lock(t->segMutex);
		known = t->segModes;
	}
	for (uint32_t r : known)
		if (modes.size() < maxModes)
			PIRE_TRY(addMode(false, r));

	// ---- the chain
	std::vector<uint32_t> cur(n), at(n);          // per string: state at the start of segment at[i]
	std::vector<uint32_t> finalState(n);
	for (uint64_t i = 0; i < n; ++i) {
		at[i] = strFirst[i];
		cur[i] = modes[0].guess[strFirst[i]];
	}
	std::vector<uint64_t> pending;                // strings stopped at a segment no mode predicted
	std::unordered_map<uint32_t, uint32_t> surprises;
	uint64_t nPredicted = 0, nWalked = 0, nPlain = 0;
	uint64_t *dExB = nullptr, *dExE = nullptr;
	uint32_t *dExI = nullptr, *dExO = nullptr;
	std::vector<uint64_t> exB, exE;
	std::vector<uint32_t> exI, exO;
	std::vector<uint64_t> todo(n);
	for (uint64_t i = 0; i < n; ++i)
		todo[i] = i;
	while (!todo.empty()) {
		pending.clear();
		for (uint64_t i : todo) {
			uint32_t k = at[i], c = cur[i];
			const uint32_t last = strFirst[i + 1];
			while (k < last) {
				bool hit = false;
				for (const Mode& m : modes)
					if (m.guess[k] == c) {
						c = m.end[k];
						hit = true;
						break;
					}
				if (!hit)
					break;
				++k;
				++nPredicted;
			}
			at[i] = k;
			cur[i] = c;
			if (k < last)
				pending.push_back(i);
			else
				finalState[i] = c;
		}
		todo.clear();
		if (pending.empty())
			break;
		// a state that keeps surprising us is a mode of the automaton the guesses do not know yet
		uint32_t best = 0, bestCount = 0;
		for (uint64_t i : pending) {
			const uint32_t c = ++surprises[cur[i]];
			if (c > bestCount) {
				best = cur[i];
				bestCount = c;
			}
		}
		if (modes.size() < maxModes && (bestCount >= 2 || budget == 0)) {
			PIRE_TRY(addMode(false, best));
			{
				std::lock_guard<std::mutex> lock(t->segMutex);
				if (std::find(t->segModes.begin(), t->segModes.end(), best) == t->segModes.end() && t->segModes.size() < 8)
					t->segModes.push_back(best);
			}
			surprises.erase(best);
			todo = pending;
			continue;
		}
		// scan the pending segments from the states the chain is in -- or, when the budget of such round trips is
		// spent, the whole rest of those strings, the plain sequential way
		const bool plain = budget == 0;
		if (!plain)
			--budget;
		const size_t m = pending.size();
		exB.resize(m);
		exE.resize(m);
		exI.resize(m);
		exO.resize(m);
		for (size_t j = 0; j < m; ++j) {
			const uint64_t i = pending[j];
			exB[j] = segBegin[at[i]];
			exE[j] = plain ? segEnd[strFirst[i + 1] - 1] : segEnd[at[i]];
			exI[j] = cur[i];
		}
		if (!dExB) {
			PIRE_TRY(scratch.Alloc(&dExB, n));
			PIRE_TRY(scratch.Alloc(&dExE, n));
			PIRE_TRY(scratch.Alloc(&dExI, n));
			PIRE_TRY(scratch.Alloc(&dExO, n));
		}
		PIRE_TRY(HipOk(hipMemcpyAsync(dExB, exB.data(), m * 8, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(dExE, exE.data(), m * 8, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		PIRE_TRY(HipOk(hipMemcpyAsync(dExI, exI.data(), m * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
		q.n = m;
		q.offsets = dExB;
		q.ends = dExE;
		q.initIdx = dExI;
		q.flags = 0;
		q.outIdx = dExO;
		PIRE_TRY(ScanBatch(q, t, stream));
		PIRE_TRY(HipOk(hipMemcpyAsync(exO.data(), dExO, m * 4, hipMemcpyDeviceToHost, stream), "hipMemcpy(D2H)"));
		PIRE_TRY(HipOk(hipStreamSynchronize(stream), "hipStreamSynchronize"));
		for (size_t j = 0; j < m; ++j) {
			const uint64_t i = pending[j];
			cur[i] = exO[j];
			if (plain) {
				nPlain += strFirst[i + 1] - at[i];
				at[i] = strFirst[i + 1];
				finalState[i] = cur[i];
			} else {
				++nWalked;
				++at[i];
				if (at[i] == strFirst[i + 1])
					finalState[i] = cur[i];
				else
					todo.push_back(i);
			}
		}
	}

	// ---- finish
	uint32_t* dFinal = nullptr;
	PIRE_TRY(scratch.Alloc(&dFinal, n));
	PIRE_TRY(HipOk(hipMemcpyAsync(dFinal, finalState.data(), n * 4, hipMemcpyHostToDevice, stream), "hipMemcpy(H2D)"));
	{
		ScanParams f = p;
		f.compact = 0;
		const LdsLayout L = MakeLayout(f.hot, f.outCounts ? f.regexps : 0, kRotPitch, 0);
		PIRE_TRY(HipOk(hipFuncSetAttribute(reinterpret_cast<const void*>(SegmentFinishKernel),
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, int(L.total)), "hipFuncSetAttribute(LDS)"));
		const unsigned threads = n >= 4096 ? 1024 : 256;
		const uint64_t tasks = (n + 63) / 64;
		const unsigned cblocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((tasks * 64 + threads - 1) / threads, uint64_t(cus))));
		hipLaunchKernelGGL(SegmentFinishKernel, dim3(cblocks), dim3(threads), L.total, stream, f, dFinal);
	}
	PIRE_TRY(HipOk(hipGetLastError(), "segmented scan launch"));
	PIRE_TRY(HipOk(hipStreamSynchronize(stream), "hipStreamSynchronize"));   // finalState is the source of the copy above
	NoteKernel("segmented");
	if (wantStats)
		fprintf(stderr, "pire_hip segmented: %llu strings, %llu segments of %llu B (+%llu B warm-up), %zu modes; segments "
		                "predicted %llu, scanned again from the chain's state %llu, left to the plain walk %llu\n",
		        (unsigned long long)n, (unsigned long long)S, (unsigned long long)segBytes, (unsigned long long)warmBytes,
		        modes.size(), (unsigned long long)nPredicted, (unsigned long long)nWalked, (unsigned long long)nPlain);
	return PIRE_HIP_OK;
}

}  // namespace pirehip
