// Scanner::Glue's product construction on the GPU: the breadth-first state discovery of
// Impl::Determine (/root/reference/pire/determine.h:91-137) over pairs of states of two ingested tables
// (ScannerGlueCommon::Next, /root/reference/pire/glue.h:132-137), level by level, with EXACTLY the numbering of the
// sequential reference loop.
//
// Why a level-synchronous formulation numbers states like the sequential one: the reference visits states in index
// order and, per state, the letter classes in order; a pair seen for the first time gets the next free index.  All
// states of one BFS level [lo, hi) are visited before any state found in that level, so within a level the new pairs
// are numbered by the position pos = (i - lo) * letters + l of their FIRST occurrence.  Per level:
//   1. insert : every (state i, letter l) computes its target pair and atomicMin()s its position into the pair's
//               slot of an open-addressing hash table (already numbered pairs hold their index, which is smaller
//               than any tentative position and survives the min);
//   2. mark   : position pos is a "first occurrence" iff the slot holds exactly pos;
//   3. scan   : exclusive prefix sum of the marks (hipcub) = rank of every new pair in discovery order;
//   4. assign : new pair -> index hi + rank, written to the state list and into the slot;
//   5. fill   : next[i][l] = index of the target pair (now final for every pair).
// The host only reads the number of new states per level (and applies the reference's maxSize rule).

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "internal.h"

namespace pirehip {

namespace {

constexpr uint32_t kTentative = 0x80000000u;          // | position inside the level
constexpr uint64_t kEmptyKey = ~uint64_t(0);

struct GlueDev {
	const uint32_t* nextA;
	const uint32_t* nextB;
	const uint32_t* la;
	const uint32_t* lb;
	uint32_t lettersA, lettersB, LC;
	uint32_t* stA;          // [cap] pair components of the numbered states
	uint32_t* stB;
	unsigned long long* keys;   // [hashSize] pair (a << 32 | b) or kEmptyKey
	uint32_t* vals;             // [hashSize] final index, or kTentative | min position, or kEmptyVal
	uint32_t hashMask;
	uint32_t* marks;        // [positions] 0/1, then (after the scan) ranks
	uint32_t* ranks;
	uint32_t* nextOut;      // [cap * LC]
	uint32_t cap;
	uint32_t maxProbe;      // probes before a lookup gives up (the table is then far too full: more pairs than cap)
	uint32_t* overflow;     // set when that happened
};

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t HashPair(uint64_t key, uint32_t mask)
{
	key ^= key >> 33;
	key *= 0xff51afd7ed558ccdULL;
	key ^= key >> 29;
	return uint32_t(key) & mask;
}

__device__ __forceinline__ uint64_t TargetOf(const GlueDev& g, uint32_t i, uint32_t l)
{
	const uint32_t na = g.nextA[size_t(g.stA[i]) * g.lettersA + g.la[l]];   // Lhs().Next(state.first, letter), glue.h:134
	const uint32_t nb = g.nextB[size_t(g.stB[i]) * g.lettersB + g.lb[l]];   // Rhs().Next(state.second, letter), glue.h:135
	return (uint64_t(na) << 32) | nb;
}

// Finds the slot of `key`, claiming an empty one if it is not in the table yet.
__device__ __forceinline__ uint32_t SlotOf(const GlueDev& g, uint64_t key)
{
	// Bounded: one level may meet far more distinct pairs than the table has slots (an exploding product -- exactly
	// the case in which Glue is about to fail); unbounded probing of a full table would spin for ever.
	uint32_t h = HashPair(key, g.hashMask);
	for (uint32_t probes = 0; probes < g.maxProbe; ++probes) {
		const unsigned long long seen = atomicCAS(&g.keys[h], kEmptyKey, key);
		if (seen == kEmptyKey || seen == key)
			return h;
		h = (h + 1) & g.hashMask;
	}
	*g.overflow = 1;
	return kNoSlot;
}

__global__ void GlueInsert(GlueDev g, uint32_t lo, uint32_t count)
{
	const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
	if (pos >= count)
		return;
	const uint64_t key = TargetOf(g, lo + pos / g.LC, pos % g.LC);
	const uint32_t slot = SlotOf(g, key);
	if (slot != kNoSlot)
		atomicMin(&g.vals[slot], kTentative | pos);
}

__global__ void GlueMark(GlueDev g, uint32_t lo, uint32_t count)
{
	const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
	if (pos >= count)
		return;
	const uint64_t key = TargetOf(g, lo + pos / g.LC, pos % g.LC);
	const uint32_t slot = SlotOf(g, key);
	g.marks[pos] = slot != kNoSlot && g.vals[slot] == (kTentative | pos) ? 1u : 0u;
}

__global__ void GlueAssign(GlueDev g, uint32_t lo, uint32_t hi, uint32_t count)
{
	const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
	if (pos >= count || !g.marks[pos])
		return;
	const uint64_t key = TargetOf(g, lo + pos / g.LC, pos % g.LC);
	const uint32_t idx = hi + g.ranks[pos];
	if (idx < g.cap) {   // beyond the cap the glue fails anyway (host checks the count)
		g.stA[idx] = uint32_t(key >> 32);
		g.stB[idx] = uint32_t(key);
	}
	const uint32_t slot = SlotOf(g, key);
	if (slot != kNoSlot)
		g.vals[slot] = idx;
}

__global__ void GlueFill(GlueDev g, uint32_t lo, uint32_t count)
{
	const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
	if (pos >= count)
		return;
	const uint32_t i = lo + pos / g.LC, l = pos % g.LC;
	const uint32_t slot = SlotOf(g, TargetOf(g, i, l));
	g.nextOut[size_t(i) * g.LC + l] = slot != kNoSlot ? g.vals[slot] : 0u;
}

struct DevBuf {
	void* p = nullptr;
	~DevBuf()
	{
		if (p)
			(void)hipFree(p);
	}
	int Alloc(size_t bytes)
	{
		hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
		return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMalloc(glue)");
	}
	template <class T>
	T* As() const { return static_cast<T*>(p); }
};

template <class T>
int Upload(DevBuf* b, const std::vector<T>& v)
{
	if (int rc = b->Alloc(v.size() * sizeof(T)))
		return rc;
	hipError_t e = hipMemcpy(b->p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemcpy(glue)");
}

}  // namespace

int GlueBfsDevice(const HostTable& a, const HostTable& b, const std::vector<uint32_t>& la, const std::vector<uint32_t>& lb,
                  size_t maxSize, GlueProduct* out)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		(void)hipGetLastError();
		SetError("no HIP device: pire_hip_table_glue_gpu needs a GPU (pire_hip_table_glue is the host version)");
		return PIRE_HIP_ENODEVICE;
	}
	const uint32_t LC = uint32_t(la.size());
	// the glue fails once more than maxSize NEW states are needed (determine.h:112-113), i.e. beyond maxSize + 1 states
	if (maxSize > (1u << 22))
		return GlueBfsHost(a, b, la, lb, maxSize, out);   // beyond what the device buffers are sized for: host version
	const uint32_t cap = uint32_t(maxSize) + 1;
	uint32_t hashSize = 1;
	while (hashSize < 4u * (cap + 64))
		hashSize <<= 1;
	const size_t maxPositions = size_t(cap) * LC;

	DevBuf dNextA, dNextB, dLa, dLb, dStA, dStB, dKeys, dVals, dMarks, dRanks, dNext, dTemp, dOverflow;
	int rc;
	if ((rc = Upload(&dNextA, a.next)) || (rc = Upload(&dNextB, b.next)) || (rc = Upload(&dLa, la)) || (rc = Upload(&dLb, lb)) ||
	    (rc = dStA.Alloc(size_t(cap) * 4)) || (rc = dStB.Alloc(size_t(cap) * 4)) || (rc = dKeys.Alloc(size_t(hashSize) * 8)) ||
	    (rc = dVals.Alloc(size_t(hashSize) * 4)) || (rc = dMarks.Alloc(maxPositions * 4)) ||
	    (rc = dRanks.Alloc(maxPositions * 4)) || (rc = dNext.Alloc(maxPositions * 4)) || (rc = dOverflow.Alloc(4)))
		return rc;
	hipError_t e = hipMemset(dKeys.p, 0xFF, size_t(hashSize) * 8);
	if (e == hipSuccess)
		e = hipMemset(dOverflow.p, 0, 4);
	if (e == hipSuccess)
		e = hipMemset(dVals.p, 0xFF, size_t(hashSize) * 4);
	if (e != hipSuccess)
		return HipFail(e, "hipMemset(glue hash)");

	GlueDev g;
	g.nextA = dNextA.As<uint32_t>();
	g.nextB = dNextB.As<uint32_t>();
	g.la = dLa.As<uint32_t>();
	g.lb = dLb.As<uint32_t>();
	g.lettersA = a.letters;
	g.lettersB = b.letters;
	g.LC = LC;
	g.stA = dStA.As<uint32_t>();
	g.stB = dStB.As<uint32_t>();
	g.keys = dKeys.As<unsigned long long>();
	g.vals = dVals.As<uint32_t>();
	g.hashMask = hashSize - 1;
	g.marks = dMarks.As<uint32_t>();
	g.ranks = dRanks.As<uint32_t>();
	g.nextOut = dNext.As<uint32_t>();
	g.cap = cap;
	g.maxProbe = std::min<uint32_t>(hashSize, 1u << 14);
	g.overflow = dOverflow.As<uint32_t>();

	// state 0 = (lhs initial, rhs initial), numbered before the loop (determine.h:105-106)
	{
		const uint32_t ia = a.initial, ib = b.initial;
		const unsigned long long key = (uint64_t(ia) << 32) | ib;
		uint64_t kk = key;
		kk ^= kk >> 33;
		kk *= 0xff51afd7ed558ccdULL;
		kk ^= kk >> 29;
		const uint32_t h = uint32_t(kk) & g.hashMask;
		const uint32_t zero = 0;
		e = hipMemcpy(g.stA, &ia, 4, hipMemcpyHostToDevice);
		if (e == hipSuccess)
			e = hipMemcpy(g.stB, &ib, 4, hipMemcpyHostToDevice);
		if (e == hipSuccess)
			e = hipMemcpy(&g.keys[h], &key, 8, hipMemcpyHostToDevice);
		if (e == hipSuccess)
			e = hipMemcpy(&g.vals[h], &zero, 4, hipMemcpyHostToDevice);
		if (e != hipSuccess)
			return HipFail(e, "hipMemcpy(glue initial state)");
	}

	size_t tempBytes = 0;
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, g.marks, g.ranks, int(maxPositions));
	if ((rc = dTemp.Alloc(tempBytes)))
		return rc;

	out->failed = false;
	uint32_t lo = 0, hi = 1;
	while (lo < hi) {
		const uint32_t count = (hi - lo) * LC;
		const unsigned blocks = (count + 255) / 256;
		hipLaunchKernelGGL(GlueInsert, dim3(blocks), dim3(256), 0, nullptr, g, lo, count);
		hipLaunchKernelGGL(GlueMark, dim3(blocks), dim3(256), 0, nullptr, g, lo, count);
		e = hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, g.marks, g.ranks, int(count));
		if (e != hipSuccess)
			return HipFail(e, "hipcub::DeviceScan::ExclusiveSum");
		hipLaunchKernelGGL(GlueAssign, dim3(blocks), dim3(256), 0, nullptr, g, lo, hi, count);
		hipLaunchKernelGGL(GlueFill, dim3(blocks), dim3(256), 0, nullptr, g, lo, count);
		uint32_t lastRank = 0, lastMark = 0, overflow = 0;
		e = hipMemcpy(&lastRank, &g.ranks[count - 1], 4, hipMemcpyDeviceToHost);   // synchronises the level
		if (e == hipSuccess)
			e = hipMemcpy(&lastMark, &g.marks[count - 1], 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess)
			e = hipMemcpy(&overflow, g.overflow, 4, hipMemcpyDeviceToHost);
		if (e != hipSuccess)
			return HipFail(e, "glue level (kernels / copy back)");
		if (overflow)   // the pair table filled up inside a level: let the sequential version decide (it is exact)
			return GlueBfsHost(a, b, la, lb, maxSize, out);
		const uint64_t fresh = uint64_t(lastRank) + lastMark;
		if (uint64_t(hi) + fresh > cap) {   // more than maxSize new states in total: task.Failure()
			out->failed = true;
			out->states.clear();
			out->next.clear();
			return PIRE_HIP_OK;
		}
		lo = hi;
		hi += uint32_t(fresh);
	}
	const uint32_t N = hi;
	std::vector<uint32_t> sa(N), sb(N);
	out->next.resize(size_t(N) * LC);
	e = hipMemcpy(sa.data(), g.stA, size_t(N) * 4, hipMemcpyDeviceToHost);
	if (e == hipSuccess)
		e = hipMemcpy(sb.data(), g.stB, size_t(N) * 4, hipMemcpyDeviceToHost);
	if (e == hipSuccess)
		e = hipMemcpy(out->next.data(), g.nextOut, size_t(N) * LC * 4, hipMemcpyDeviceToHost);
	if (e != hipSuccess)
		return HipFail(e, "hipMemcpy(glue result)");
	out->states.resize(N);
	for (uint32_t i = 0; i < N; ++i)
		out->states[i] = std::make_pair(sa[i], sb[i]);
	return PIRE_HIP_OK;
}

}  // namespace pirehip
