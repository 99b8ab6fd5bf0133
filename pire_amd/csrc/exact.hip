// The one-string-per-lane exact kernels: generic scan (any offsets / alignment / length), HalfFinalScanner counting,
// LongestPrefix / ShortestPrefix, single Step().  DESIGN.md sections 4.4 - 4.6.

#include "device_common.h"
#include "walk.h"

namespace pirehip {

// ------------------------------------------------------------------------------------------ generic kernel
// Any offsets, any alignment, any length (including 0).  One string per lane, exact step per byte.

__global__ __launch_bounds__(1024) void ScanGenericKernel(ScanParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	LoadTableToLds(p, lds, L);

	const uint64_t nrounds = (p.n + 63) / 64;
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t task = uint64_t(blockIdx.x) * wavesPerBlock + (threadIdx.x >> 6); task < nrounds;
	     task += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t s = task * 64 + lane;
		const bool active = s < p.n;
		uint32_t st = 0;
		if (active) {
			st = StartState(p, s);
			uint64_t b, e;
			if (p.offsets) {
				b = p.offsets[s];
				e = p.ends ? p.ends[s] : p.offsets[s + 1];
			} else {
				b = s * p.stride;
				e = b + p.len;
			}
			const uint8_t* ptr = p.text + b;
			const uint8_t* end = p.text + e;
			while (ptr < end && (reinterpret_cast<uintptr_t>(ptr) & 15)) {
				st = SlowStep(p, lds, L, st, *ptr);
				++ptr;
			}
			// whole 16-byte chunks: the dense-row fast path of the other kernels (one v_perm + one LDS byte per step,
			// exact re-walk of a chunk that leaves the dense rows); this halves the time a lone lane needs per byte
			uint32_t hs = st < p.hot ? st : p.hot, cold = st;
			for (; ptr + 16 <= end; ptr += 16)
				StepChunk<0>(p, lds, L, *reinterpret_cast<const u32x4*>(ptr), hs, cold, uint32_t(reinterpret_cast<uintptr_t>(ptr) >> 4) & 63);
			st = hs != p.hot ? hs : cold;
			for (; ptr < end; ++ptr)
				st = SlowStep(p, lds, L, st, *ptr);
		}
		Finish(p, lds, L, s, active, st);
	}
	FlushCounts(p, lds, L);
}

// ------------------------------------------------------------------------------------------ HalfFinalScanner
// Pire::HalfFinalScanner (scanners/half_final.h) is a Scanner whose Initialize and every Step end with TakeAction:
// if the new state is Final, every entry of its final list bumps the per-regexp match counter of the string
// (half_final.h:137-164).  Same table, same walk, plus State::Result(r) per string.  First version: the generic
// kernel's exact walk with a one-compare Final test per step -- the hot set is ordered non-final first, so
// "hot and Final" is `st >= hotFinalLo`, cold states look their flags up -- and, for up to 8 regexps, counters in
// registers fed from a packed increment word per state (hot states: LDS).

template <bool PACKED>
struct HalfCounters;

template <>
struct HalfCounters<true> {
	uint32_t c[8];
	__device__ __forceinline__ void Init(const ScanParams&, uint32_t*, uint64_t)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)
			c[r] = 0;
	}
	__device__ __forceinline__ void Take(const ScanParams& p, const uint64_t* incHot, uint32_t st)
	{
		const uint64_t inc = st < p.hot ? incHot[st] : p.incPerm[st];
#pragma unroll
		for (int r = 0; r < 8; ++r)
			c[r] += uint32_t(inc >> (8 * r)) & 0xFFu;
	}
	__device__ __forceinline__ void Store(const ScanParams& p, uint32_t* out, uint64_t s)
	{
#pragma unroll
		for (int r = 0; r < 8; ++r)      // static indices only: a runtime index would put c[] into scratch
			if (uint32_t(r) < p.regexps)
				out[s * p.regexps + r] = c[r];
	}
};

template <>
struct HalfCounters<false> {
	uint32_t* row;
	__device__ __forceinline__ void Init(const ScanParams& p, uint32_t* out, uint64_t s)
	{
		row = out + s * p.regexps;
		for (uint32_t r = 0; r < p.regexps; ++r)
			row[r] = 0;
	}
	__device__ __forceinline__ void Take(const ScanParams& p, const uint64_t*, uint32_t st)
	{
		for (uint64_t k = p.acceptOffPerm[st]; k < p.acceptOffPerm[st + 1]; ++k)
			row[p.acceptIds[k]] += 1;   // the lane owns the row: plain read-modify-write
	}
	__device__ __forceinline__ void Store(const ScanParams&, uint32_t*, uint64_t) {}
};

template <bool PACKED>
__global__ __launch_bounds__(1024) void HalfFinalKernel(ScanParams p, uint32_t* outResults, const uint32_t* list)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	uint64_t* incHot = reinterpret_cast<uint64_t*>(lds + L.total);
	if (PACKED)
		for (uint32_t i = threadIdx.x; i < p.hot; i += blockDim.x)
			incHot[i] = p.incPerm[i];
	LoadTableToLds(p, lds, L);

	// with a list (the row kernel ran first on this stream and left the strings its 16-bit counters cannot hold): only those
	const uint64_t todo = list ? list[0] : p.n;
	const uint64_t nrounds = (todo + 63) / 64;
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t initial = p.startPerm;   // Initialize(): the launcher passes flags without BEGIN to FillParams
	for (uint64_t task = uint64_t(blockIdx.x) * wavesPerBlock + (threadIdx.x >> 6); task < nrounds;
	     task += uint64_t(gridDim.x) * wavesPerBlock) {
		const uint64_t k = task * 64 + lane;
		if (k >= todo)
			continue;
		const uint64_t s = list ? list[1 + k] : k;
		HalfCounters<PACKED> cnt;
		cnt.Init(p, outResults, s);
		uint32_t st = initial;
		if (IsFinalState(p, st))
			cnt.Take(p, incHot, st);                       // Initialize ends with TakeAction, half_final.h:142
		if (p.flags & PIRE_HIP_RUN_BEGIN) {
			st = p.nextPerm[size_t(st) * p.letters + p.beginCls];
			if (IsFinalState(p, st))
				cnt.Take(p, incHot, st);
		}
		const uint8_t* ptr = p.text + p.offsets[s];
		const uint8_t* end = p.text + p.offsets[s + 1];
		auto step = [&](uint32_t byte) {
			st = SlowStep(p, lds, L, st, byte);
			if (IsFinalState(p, st))
				cnt.Take(p, incHot, st);
			return true;
		};
		// whole blocks first through the dense rows alone: no Final state among the 16 (largest id below the first
		// Final one, which also says the walk stayed in the dense rows) -> nothing to count, take the state; otherwise
		// the block again, step by step.  Line-aligned tile loads (walk.h) instead of a block per iteration and byte
		// loads at the ragged ends: round 3, from the kernel trace of 10-string calls (0.14 us per byte before).
		WalkBlocks(ptr, end, [&](walk_u32x4 v, uint32_t skip, uint32_t count) {
			if (st < p.hot) {
				uint32_t h = st;
				const uint32_t mx = count == 16 ? DenseChunk(lds, L, v, h) : DenseBytes(lds, L, v, skip, count, h);
				if (mx < p.hotFinalLo) {
					st = h;
					return true;
				}
			}
			return BlockBytes(v, skip, count, step);
		});
		if (p.flags & PIRE_HIP_RUN_END) {
			st = p.nextPerm[size_t(st) * p.letters + p.endCls];
			if (IsFinalState(p, st))
				cnt.Take(p, incHot, st);
		}
		cnt.Store(p, outResults, s);
		if (p.outIdx)
			p.outIdx[s] = p.origOfPerm[st];
		if (p.outFinal)
			p.outFinal[s] = p.flagsPerm[st] & kFinal;
	}
}

// ------------------------------------------------------------------------------------------ prefix searches
// Pire::LongestPrefix / ShortestPrefix (run.h:277-311) with LongestPrefixPred / ShortestPrefixPred (run.h:69-100):
// the same walk, but after every byte Final(state) records the position and Dead(state) (or, for the shortest
// prefix, the first Final) ends it.  One string per lane, exact step (dense row first); the early exit is per lane.

struct PrefixParams {
	ScanParams scan;
	uint32_t longest, throughEnd;
	long long* outLen;
};

__global__ __launch_bounds__(1024) void PrefixKernel(PrefixParams q)
{
	const ScanParams& p = q.scan;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	LoadTableToLds(p, lds, L);
	for (uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; s < p.n; s += uint64_t(gridDim.x) * blockDim.x) {
		uint64_t b, e;
		if (p.offsets) {
			b = p.offsets[s];
			e = p.offsets[s + 1];
		} else {
			b = s * p.stride;
			e = b + p.len;
		}
		const uint8_t* text = p.text + b;
		const uint64_t len = e - b;
		uint32_t st = p.startPerm;                       // Initialize (+ BeginMark if throughBeginMark), run.h:280-283
		long long pos = -1;
		bool stop = false;
		uint32_t f = StateFlags(p, lds, L, st);
		if (f & kFinal) {
			pos = 0;                                     // run.h:284 / 301-302
			stop = !q.longest;
		}
		const bool foundAtStart = stop;
		if (!stop) {
			uint64_t i = 0;
			auto step = [&](uint32_t byte) {
				st = SlowStep(p, lds, L, st, byte);
				f = StateFlags(p, lds, L, st);
				++i;
				if (f & kFinal) {
					pos = (long long)i;
					if (!q.longest)
						stop = true;                     // ShortestPrefixPred: Stop on the first Final
				}
				if (f & kDead)
					stop = true;                         // both predicates stop on a dead state
				return !stop;
			};
			// line-aligned vector loads instead of byte loads (walk.h); whole blocks first through the dense rows alone:
			// neither a Dead nor a Final state among the 16 (the rows are ordered plain, Dead, Final) -> nothing to record,
			// take the state; otherwise the block again, step by step (round 3: 0.25 us per byte before, one wave alone)
			WalkBlocks(text, text + len, [&](walk_u32x4 v, uint32_t skip, uint32_t count) {
				if (st < p.hot) {
					uint32_t h = st;
					const uint32_t mx = count == 16 ? DenseChunk(lds, L, v, h) : DenseBytes(lds, L, v, skip, count, h);
					if (mx < p.hotDeadLo) {
						st = h;
						i += count;
						return true;
					}
				}
				return BlockBytes(v, skip, count, step);
			});
		}
		if (q.throughEnd && !foundAtStart) {
			st = p.nextPerm[size_t(st) * p.letters + p.endCls];
			if (StateFlags(p, lds, L, st) & kFinal) {
				if (q.longest || pos < 0)
					pos = (long long)len;                // run.h:286-290 / 305-309
			}
		}
		q.outLen[s] = pos;
	}
}

// ------------------------------------------------------------------------------------------ suffix searches
// Pire::LongestSuffix / ShortestSuffix (run.h:313-362): the text is walked BACKWARDS from its last byte (the scanner
// comes from Fsm::Reverse()).  One string per lane; 16-byte blocks are read from the top down, only blocks that hold at
// least one byte of the string.  out = length of the suffix (the reference returns the pointer (last byte) - out), -1
// where the reference returns null.  startPerm: Initialize() (+ Step(EndMark) when throughEndMark), folded by the host.
struct SuffixParams {
	ScanParams scan;
	uint32_t longest, throughBegin;
	long long* outLen;
};

__global__ __launch_bounds__(1024) void SuffixKernel(SuffixParams q)
{
	const ScanParams& p = q.scan;
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	LoadTableToLds(p, lds, L);
	for (uint64_t s = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; s < p.n; s += uint64_t(gridDim.x) * blockDim.x) {
		uint64_t b, e;
		if (p.offsets) {
			b = p.offsets[s];
			e = p.offsets[s + 1];
		} else {
			b = s * p.stride;
			e = b + p.len;
		}
		const uint64_t lo = reinterpret_cast<uint64_t>(p.text) + b;   // address of the first byte
		uint64_t cur = reinterpret_cast<uint64_t>(p.text) + e;        // one past the next byte to take
		const uint64_t top = cur;
		uint32_t st = p.startPerm;
		uint32_t f = StateFlags(p, lds, L, st);
		long long pos = -1;
		// LongestSuffix: while (bytes left && !Dead) { if Final: pos = here; step }   run.h:326-333
		// ShortestSuffix: while (bytes left && !Final && !Dead) step                  run.h:354-357
		bool go = cur > lo && !(f & kDead) && (q.longest || !(f & kFinal));
		while (go) {
			const uint64_t block = (cur - 1) & ~uint64_t(15);
			u32x4 v = *reinterpret_cast<const u32x4*>(block);
			uint32_t k = uint32_t(cur - 1 - block);                          // index of the next byte inside the block
			const uint32_t kLow = block >= lo ? 0u : uint32_t(lo - block);   // first index that belongs to the string
			// A whole block of the string from a state with a dense row: its sixteen bytes top down through the dense rows alone
			// (round 6; the prefix kernel's DenseChunk, backwards).  Neither a Dead nor a Final state among the sixteen reached
			// (the rows are ordered plain, Dead, Final) -> nothing to record but what the state in front of them says, take the
			// state; otherwise the block byte by byte below.  The suffix searches were never measured before round 6: 0.52 TB/s of
			// log lines with the byte-wise walk alone (a SlowStep, a flag lookup and three funnel shifts per byte).
			if (k == 15 && kLow == 0 && st < p.hot) {
				uint32_t h = st, mx = 0;
#pragma unroll
				for (int w = 3; w >= 0; --w) {
					const uint32_t x = v[w];
#pragma unroll
					for (int b2 = 3; b2 >= 0; --b2) {
						h = lds[h * L.pitch + ((x >> (8 * b2)) & 0xFFu)];
						mx = mx > h ? mx : h;
					}
				}
				if (mx < p.hotDeadLo) {
					if (q.longest && (f & kFinal))
						pos = (long long)(top - cur);
					st = h;
					f = 0;   // a plain dense state: neither Final nor Dead
					cur -= 16;
					go = cur > lo;
					continue;
				}
			}
			// bring byte k to the top of the 128-bit value, then peel bytes off the top
			for (uint32_t sh = 15 - k; sh; --sh) {
				v.w = __builtin_amdgcn_alignbit(v.w, v.z, 24);
				v.z = __builtin_amdgcn_alignbit(v.z, v.y, 24);
				v.y = __builtin_amdgcn_alignbit(v.y, v.x, 24);
				v.x <<= 8;
			}
			for (;;) {
				if (q.longest && (f & kFinal))
					pos = (long long)(top - cur);
				st = SlowStep(p, lds, L, st, v.w >> 24);
				f = StateFlags(p, lds, L, st);
				--cur;
				go = cur > lo && !(f & kDead) && (q.longest || !(f & kFinal));
				if (!go || k == kLow)
					break;
				--k;
				v.w = __builtin_amdgcn_alignbit(v.w, v.z, 24);
				v.z = __builtin_amdgcn_alignbit(v.z, v.y, 24);
				v.y = __builtin_amdgcn_alignbit(v.y, v.x, 24);
				v.x <<= 8;
			}
		}
		const long long here = (long long)(top - cur);
		if (q.longest) {
			if (f & kFinal)
				pos = here;                                              // run.h:334-335
			if (q.throughBegin) {
				st = p.nextPerm[size_t(st) * p.letters + p.beginCls];
				if (StateFlags(p, lds, L, st) & kFinal)
					pos = here;                                          // run.h:336-340
			}
		} else {
			if (q.throughBegin)
				st = p.nextPerm[size_t(st) * p.letters + p.beginCls];    // run.h:358-359
			pos = (StateFlags(p, lds, L, st) & kFinal) ? here : -1;      // run.h:360
		}
		q.outLen[s] = pos;
	}
}

// ------------------------------------------------------------------------------------------ single Step()

__global__ __launch_bounds__(256) void StepKernel(ScanParams p, uint32_t* stateIdx, uint64_t n, uint32_t cls)
{
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) {
		const uint32_t in = stateIdx[i];
		const uint32_t st = p.permOfOrig[in < p.states ? in : 0];   // never read outside the table
		stateIdx[i] = p.origOfPerm[p.nextPerm[size_t(st) * p.letters + cls]];
	}
}


// ------------------------------------------------------------------------------------------ launchers

namespace {

// The dense rows take 66 KB of LDS (130 KB with the compact tier), and every block copies them in before it walks a
// byte: 1024-thread blocks throughout.  Rounds 1-2 gave batches below 128 Ki strings 256-thread blocks ("more CUs
// busy"); the kernel trace of 10-string calls (round 3) showed what that costs: the copy by four waves took 25-45 us
// of a call whose walk takes 3.  Sixteen waves copy it in ~5 us, and 16 waves per CU hide the lookup latency as well
// as 4 waves on each of four times as many CUs.
int ExactBlockThreads(uint64_t)
{
	return 1024;
}


}  // namespace

int LaunchGeneric(const ScanParams& p0, hipStream_t stream)
{
	if (int rc = CheckCounts(p0))
		return rc;
	ScanParams p = p0;
	p.compact = 0;   // small blocks, several per CU: no room (and no need) for the warm rows
	const LdsLayout L = MakeLayout(p.hot, p.outCounts ? p.regexps : 0, 256u, CompactBytes(p));
	return LaunchScan(ScanGenericKernel, p, ExactBlockThreads(p.n), L.total, stream);
}


int LaunchPrefix(const ScanParams& p0, bool longest, bool throughEnd, long long* outLen, hipStream_t stream,
                 unsigned long long* workCounter)
{
	if (p0.n == 0)
		return PIRE_HIP_OK;
	// A search that is over after a few bytes (a lexer's token scanner is Dead right behind the token; the first
	// Final state of a dense scanner comes at once) reads a 16-byte block or two per string in the kernel below and
	// a whole 128-byte window in the ragged one: measured 6.8 against 2.5 TB/s of text on log lines.  Everything that
	// walks on takes the ragged kernel (2.6 - 5.6 x faster).  The shares come from the byte model of table.cpp.
	const bool quick = (p0.deadShare > 0.5f || (!longest && p0.finalShare > 0.25f)) &&
	                   !GetConfig().ragged_act_always;   // knob: tests and A/B measurements
	if (workCounter && !quick && RaggedActEligible(p0)) {
		NoteKernel("ragged_prefix");
		return LaunchRaggedPrefix(p0, workCounter, longest, throughEnd, outLen, stream);
	}
	NoteKernel("prefix");
	ScanParams p = p0;
	p.compact = 0;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	hipError_t e = SetDynamicLds(reinterpret_cast<const void*>(PrefixKernel), uint32_t(L.total));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	PrefixParams q;
	q.scan = p;
	q.longest = longest ? 1 : 0;
	q.throughEnd = throughEnd ? 1 : 0;
	q.outLen = outLen;
	const unsigned threads = unsigned(ExactBlockThreads(p.n));
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + threads - 1) / threads, uint64_t(cus) * 2)));
	hipLaunchKernelGGL(PrefixKernel, dim3(blocks), dim3(threads), L.total, stream, q);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "prefix kernel launch");
	return PIRE_HIP_OK;
}

int LaunchSuffix(const ScanParams& p0, bool longest, bool throughBegin, long long* outLen, hipStream_t stream)
{
	if (p0.n == 0)
		return PIRE_HIP_OK;
	NoteKernel("suffix");
	ScanParams p = p0;
	p.compact = 0;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	hipError_t e = SetDynamicLds(reinterpret_cast<const void*>(SuffixKernel), uint32_t(L.total));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	SuffixParams q;
	q.scan = p;
	q.longest = longest ? 1 : 0;
	q.throughBegin = throughBegin ? 1 : 0;
	q.outLen = outLen;
	const unsigned threads = unsigned(ExactBlockThreads(p.n));
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + threads - 1) / threads, uint64_t(cus) * 2)));
	hipLaunchKernelGGL(SuffixKernel, dim3(blocks), dim3(threads), L.total, stream, q);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "suffix kernel launch");
	return PIRE_HIP_OK;
}

int LaunchHalfFinal(const ScanParams& p0, uint32_t* outResults, hipStream_t stream, unsigned long long* workCounter,
                    const uint32_t* list)
{
	if (p0.n == 0)
		return PIRE_HIP_OK;
	if (!list && workCounter && RaggedActEligible(p0)) {
		NoteKernel("ragged_half_final");
		return LaunchRaggedHalfFinal(p0, workCounter, outResults, stream);
	}
	if (!list)
		NoteKernel("half_final");
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	ScanParams p = p0;
	p.compact = 0;
	const LdsLayout L = MakeLayout(p.hot, 0, kRotPitch, CompactBytes(p));
	const uint32_t ldsBytes = L.total + 256 * 8;
	const bool packed = p.incPerm != nullptr;
	const void* fn = packed ? reinterpret_cast<const void*>(HalfFinalKernel<true>) : reinterpret_cast<const void*>(HalfFinalKernel<false>);
	hipError_t e = SetDynamicLds(fn, uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	const unsigned threads = unsigned(ExactBlockThreads(p.n));
	const unsigned blocks = unsigned(std::max<uint64_t>(1, std::min<uint64_t>((p.n + threads - 1) / threads, uint64_t(cus) * 2)));
	if (packed)
		hipLaunchKernelGGL(HalfFinalKernel<true>, dim3(blocks), dim3(threads), ldsBytes, stream, p, outResults, list);
	else
		hipLaunchKernelGGL(HalfFinalKernel<false>, dim3(blocks), dim3(threads), ldsBytes, stream, p, outResults, list);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "half-final kernel launch");
	return PIRE_HIP_OK;
}
int LaunchStep(const ScanParams& p, uint32_t* stateIdx, uint64_t n, uint32_t cls, hipStream_t stream)
{
	if (n == 0)
		return PIRE_HIP_OK;
	const unsigned blocks = unsigned((n + 255) / 256);
	hipLaunchKernelGGL(StepKernel, dim3(blocks), dim3(256), 0, stream, p, stateIdx, n, cls);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "step kernel launch");
	return PIRE_HIP_OK;
}


}  // namespace pirehip
