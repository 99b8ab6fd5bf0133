// Two scanners over the same text in ONE pass: Pire::Run(scanner1, scanner2, state1, state2, begin, end)
// (/root/reference/pire/run.h:229-241), i.e. a Runner over Pire::ScannerPair (scanners/pair.h:33-94: Next = both
// Next()s, Final = either, StateIndex = the pair).  Fixed-length records, the tiled kernel's load path (whole-line loads
// by groups of 8 lanes, register transpose, ring of two tiles chained across tasks) with BOTH tables' dense rows in LDS
// and two lookups per byte and lane.  The text is read once instead of twice; the two dependent LDS chains of a lane
// overlap each other's latency.  Anything that is not tiled-eligible takes two ordinary passes (api.cpp).
//
// LDS: table A's dense rows at byte 0 (its v_perm result is the ds_read address, as everywhere), table B's at the fixed
// byte 65 280 so that its lookups are the same v_perm with the base in the instruction's 16-bit immediate offset
// (ds_read_u8 v, addr offset:65280).  A is therefore cut to 254 dense rows + the trap row (ids >= 254 become the trap
// id while the rows are copied in; such states are walked exactly like any state without a dense row).

#include "device_common.h"

namespace pirehip {

constexpr uint32_t kPairHotA = 254;            // dense rows of table A (ids 0..253), trap id 254
constexpr uint32_t kPairBaseB = 65280;         // = (kPairHotA + 1) * 256: where table B's rows start

struct PairSide {
	const uint8_t* rows;      // this table's dense rows in LDS (generic pointer)
	const uint16_t* cls;      // [264] letter classes in LDS
	const uint8_t* flags;     // [256] hot flags in LDS
	uint32_t hot;             // dense rows in LDS; == trap id
};

struct PairParams {
	ScanParams a, b;
	uint32_t* outIdxB;
	// the segmented scan's use (SEG, segmented.hip): the same table twice, every record a segment walked under two
	// guessed modes.  The warm-up that MAKES the guesses is part of the pass: a lane starts `warmTiles` 128-byte tiles
	// before its segment in the two representatives' states, notes the two states it is in at the segment's first byte
	// (guessA / guessB, device ids) and walks on.  A string's first segment (segJ == 0) has no warm-up: its guesses are
	// the start states themselves and the bytes before it are somebody else's.
	uint32_t warmTiles;
	const uint32_t* segJ;
	uint32_t* guessA;
	uint32_t* guessB;
};

__device__ __forceinline__ uint32_t PairSlowStep(const ScanParams& p, const PairSide& S, uint32_t st, uint32_t byte)
{
	if (st < S.hot) {
		const uint32_t e = S.rows[st * 256 + byte];
		if (e != S.hot)
			return e;
	}
	return p.nextPerm[size_t(st) * p.letters + S.cls[byte]];
}

__device__ __forceinline__ uint32_t PairSlowChunk(const ScanParams& p, const PairSide& S, u32x4 v, uint32_t st)
{
#pragma unroll 1
	for (int i = 0; i < 16; ++i) {
		st = PairSlowStep(p, S, st, v.x & 0xFF);
		v.x = __builtin_amdgcn_alignbit(v.y, v.x, 8);
		v.y = __builtin_amdgcn_alignbit(v.z, v.y, 8);
		v.z = __builtin_amdgcn_alignbit(v.w, v.z, 8);
		v.w >>= 8;
	}
	return st;
}

__device__ __forceinline__ uint32_t HotLookupB(uint32_t addr)
{
	return *reinterpret_cast<LdsBytePtr>(static_cast<uintptr_t>(addr) + kPairBaseB);   // base folded into the offset field
}

// 16 bytes through both tables; a side that leaves its dense rows is re-walked exactly for that chunk.
// SAME: both walks are in table A's rows (the segmented scan's two modes of one table).
template <bool SAME>
__device__ __forceinline__ void PairStepChunk(const PairParams& q, const PairSide& A, const PairSide& B, const u32x4 v,
                                              uint32_t& ha, uint32_t& ca, uint32_t& hb, uint32_t& cb)
{
	const uint32_t ha0 = ha, hb0 = hb;
#pragma unroll
	for (int w = 0; w < 4; ++w) {
		const uint32_t x = v[w];
		ha = HotLookup(__builtin_amdgcn_perm(ha, x, 0x0c0c0400u));
		hb = SAME ? HotLookup(__builtin_amdgcn_perm(hb, x, 0x0c0c0400u)) : HotLookupB(__builtin_amdgcn_perm(hb, x, 0x0c0c0400u));
		ha = HotLookup(__builtin_amdgcn_perm(ha, x, 0x0c0c0401u));
		hb = SAME ? HotLookup(__builtin_amdgcn_perm(hb, x, 0x0c0c0401u)) : HotLookupB(__builtin_amdgcn_perm(hb, x, 0x0c0c0401u));
		ha = HotLookup(__builtin_amdgcn_perm(ha, x, 0x0c0c0402u));
		hb = SAME ? HotLookup(__builtin_amdgcn_perm(hb, x, 0x0c0c0402u)) : HotLookupB(__builtin_amdgcn_perm(hb, x, 0x0c0c0402u));
		ha = HotLookup(__builtin_amdgcn_perm(ha, x, 0x0c0c0403u));
		hb = SAME ? HotLookup(__builtin_amdgcn_perm(hb, x, 0x0c0c0403u)) : HotLookupB(__builtin_amdgcn_perm(hb, x, 0x0c0c0403u));
	}
	if (ha == A.hot) {
		const uint32_t f = PairSlowChunk(q.a, A, v, ha0 != A.hot ? ha0 : ca);
		ha = f < A.hot ? f : A.hot;
		ca = f;
	}
	if (hb == B.hot) {
		const uint32_t f = PairSlowChunk(SAME ? q.a : q.b, B, v, hb0 != B.hot ? hb0 : cb);
		hb = f < B.hot ? f : B.hot;
		cb = f;
	}
}

template <bool NT>
__device__ __forceinline__ void PairIssueTile(u32x4 (&r)[8], uint32_t voff, uint64_t tileBase, uint64_t stride, uint64_t low = 0)
{
	// `low`: the first byte of the text.  Only the warm-up tiles of the very first record (the first load's lanes 0..7)
	// lie below it (SEG); those lanes read from the record itself instead, and what they read is never used.
	uint32_t voff0 = voff;
	if (tileBase < low)
		voff0 += (threadIdx.x & 63) < 8 ? uint32_t(low - tileBase + 127) & ~127u : 0u;
	const uint64_t b0 = tileBase, b1 = tileBase + stride, b2 = b1 + stride, b3 = b2 + stride, b4 = b3 + stride,
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

template <int BEHIND>
__device__ __forceinline__ void PairWaitTile(u32x4 (&r)[8])
{
	asm volatile("s_waitcnt vmcnt(%8)"
	             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
	             : "n"(BEHIND * 8));
}

// Copy a table's dense rows into LDS, cutting ids >= `hot` to the trap id `hot` (table A: 254 rows of its 255).
__device__ inline void PairLoadRows(uint8_t* dst, const uint8_t* src, uint32_t srcHot, uint32_t hot)
{
	const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
	uint32_t* d = reinterpret_cast<uint32_t*>(dst);
	for (uint32_t i = threadIdx.x; i < hot * 64; i += blockDim.x) {
		uint32_t w = s[i];
		if (srcHot > hot) {
			uint32_t o = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const uint32_t e = (w >> (8 * b)) & 0xFF;
				o |= (e < hot ? e : hot) << (8 * b);
			}
			w = o;
		}
		d[i] = w;
	}
	for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x)   // the trap row: absorbing
		d[hot * 64 + i] = hot * 0x01010101u;
}

template <bool SEG>
__global__ __launch_bounds__(1024, 4) void ScanPairTiledKernel(PairParams q)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const ScanParams& pa = q.a;
	const ScanParams& pb = SEG ? q.a : q.b;   // SEG: one table, walked twice (q.b only carries the second start state)
	PairSide A, B;
	A.hot = SEG ? pa.hot : (pa.hot < kPairHotA ? pa.hot : kPairHotA);
	A.rows = lds;
	uint8_t* tail = SEG ? lds + (A.hot + 1) * 256 : lds + kPairBaseB + (pb.hot + 1) * 256;
	A.flags = tail;
	A.cls = reinterpret_cast<const uint16_t*>(tail + 512);
	if (SEG) {
		B = A;
	} else {
		B.hot = pb.hot;
		B.rows = lds + kPairBaseB;
		B.flags = tail + 256;
		B.cls = reinterpret_cast<const uint16_t*>(tail + 512 + 528);
	}

	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t ntasks = pa.n / 64;
	const uint32_t warm = SEG ? q.warmTiles : 0;      // even (the launcher checks)
	const uint32_t ntiles = uint32_t(pa.len / 128) + warm;   // even, >= 2 (the launcher checks)
	const uint32_t lastTile = ntiles - 1;
	const uint32_t voff = (lane & ~7u) * uint32_t(pa.stride) + (lane & 7u) * 16;
	const uint64_t low = SEG ? reinterpret_cast<uint64_t>(pa.text) : 0;
	const uint64_t text = reinterpret_cast<uint64_t>(pa.text) - uint64_t(warm) * 128;
	// 16; the segmented scan's form: fewer when the batch has fewer tasks than 16 per CU (a compile-time 16 for the plain
	// pair: the run-time value cost it 28 bytes of scratch, tests/test_build_audit.py)
	const uint32_t wavesPerBlock = SEG ? blockDim.x >> 6 : 16u;
	const uint64_t taskStep = uint64_t(gridDim.x) * wavesPerBlock;
	const uint64_t firstTask = uint64_t(blockIdx.x) * wavesPerBlock + wave;

	u32x4 a[8], b[8];
	ZeroTile(a);
	ZeroTile(b);
	bool primed = firstTask < ntasks;
	if (primed)
		PairIssueTile<true>(a, voff, Uniform64(text + firstTask * 64 * pa.stride), pa.stride, low);
	PairLoadRows(lds, pa.hotRows, pa.hot, A.hot);
	if (!SEG)
		PairLoadRows(lds + kPairBaseB, pb.hotRows, pb.hot, B.hot);
	for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
		tail[i] = i < A.hot ? pa.hotFlags[i] : 0;
		if (!SEG)
			tail[256 + i] = pb.hotFlags[i];
	}
	for (uint32_t i = threadIdx.x; i < 264; i += blockDim.x) {
		reinterpret_cast<uint16_t*>(tail + 512)[i] = pa.cls[i];
		if (!SEG)
			reinterpret_cast<uint16_t*>(tail + 512 + 528)[i] = pb.cls[i];
	}
	// block-wide progress counter: the waves of a block are kept in step by issue priority, as in the tiled kernel
	uint32_t* prog = reinterpret_cast<uint32_t*>(tail + 512 + 2 * 528);   // (SEG leaves the B halves of the tail unused)
	if (threadIdx.x == 0)
		*prog = 0;
	uint32_t myTiles = 0;
	__syncthreads();

	for (uint64_t task = firstTask; task < ntasks; task += taskStep) {
		const uint64_t s0 = task * 64;
		const uint64_t s = s0 + lane;
		const uint64_t rowBase = Uniform64(text + s0 * pa.stride);
		const bool hasNext = task + taskStep < ntasks;
		const uint64_t chainBase = hasNext ? Uniform64(text + (s0 + taskStep * 64) * pa.stride) : rowBase + uint64_t(lastTile) * 128;
		// Initialize() (+ Begin()) of both, folded by the host.  SEG: the two modes' representatives (device ids), mode 0's
		// being the string's own start state
		uint32_t ca = SEG && pa.initIdx ? StartStateFrom(pa, pa.initIdx[s]) : pa.startPerm;
		uint32_t cb = q.b.startPerm;
		uint32_t ha = ca < A.hot ? ca : A.hot, hb = cb < B.hot ? cb : B.hot;
		bool done = false;
		if (!primed)
			PairIssueTile<true>(a, voff, rowBase, pa.stride, low);
		for (uint32_t t = 0; t < ntiles && !done; t += 2) {
			if (SEG && t == warm) {   // the segment's first byte: these two states are the guesses
				ca = ha != A.hot ? ha : ca;
				cb = hb != B.hot ? hb : cb;
				if (q.segJ[s] == 0) {
					ca = pa.initIdx ? StartStateFrom(pa, pa.initIdx[s]) : pa.startPerm;
					cb = q.b.startPerm;
				}
				ha = ca < A.hot ? ca : A.hot;
				hb = cb < B.hot ? cb : B.hot;
				q.guessA[s] = ca;
				q.guessB[s] = cb;
			}
			{
				uint32_t sum = 0;
				if (lane == 0)
					sum = atomicAdd(prog, 2u) + 2;
				sum = uint32_t(__builtin_amdgcn_readfirstlane(int(sum)));
				myTiles += 2;
				if (myTiles * wavesPerBlock > sum + 8)
					__builtin_amdgcn_s_setprio(0);
				else if (myTiles * wavesPerBlock + 8 < sum)
					__builtin_amdgcn_s_setprio(3);
				else
					__builtin_amdgcn_s_setprio(1);
			}
			PairIssueTile<true>(b, voff, rowBase + uint64_t(t + 1) * 128, pa.stride, low);
			PairWaitTile<1>(a);
			TransposeTile(a, lane);
#pragma unroll
			for (int k = 0; k < 8; ++k)
				PairStepChunk<SEG>(q, A, B, a[k], ha, ca, hb, cb);
			PairIssueTile<true>(a, voff, t + 2 < ntiles ? rowBase + uint64_t(t + 2) * 128 : chainBase, pa.stride, low);
			PairWaitTile<1>(b);
			TransposeTile(b, lane);
#pragma unroll
			for (int k = 0; k < 8; ++k)
				PairStepChunk<SEG>(q, A, B, b[k], ha, ca, hb, cb);
			// ScannerPair is absorbing when both sides are (the early-out of the single kernels, for the pair)
			const bool absA = ha != A.hot && (A.flags[ha] & kAbsorbing);
			const bool absB = hb != B.hot && (B.flags[hb] & kAbsorbing);
			done = __all(absA && absB) && t + 2 < ntiles && (!SEG || t >= warm);
		}
		primed = hasNext && !done;   // an early-out leaves some other tile in slot a: re-prime then
		if (done)
			PairWaitTile<0>(a);
		uint32_t sa = ha != A.hot ? ha : ca, sb = hb != B.hot ? hb : cb;
		if (!done) {
			const uint8_t* base = pa.text + s * pa.stride;
			for (uint64_t i = uint64_t(ntiles - warm) * 128; i < pa.len; ++i) {
				sa = PairSlowStep(pa, A, sa, base[i]);
				sb = PairSlowStep(pb, B, sb, base[i]);
			}
		}
		// End(), StateIndex of both, Final = either (pair.h:69-72, 79-82)
		const FinRec* ra = SEG ? pa.finSelf : (pa.flags & PIRE_HIP_RUN_END) ? pa.finEnd : pa.finSelf;
		const FinRec* rb = SEG ? pa.finSelf : (pb.flags & PIRE_HIP_RUN_END) ? pb.finEnd : pb.finSelf;
		const u32x4 rawA = *reinterpret_cast<const u32x4*>(&ra[sa]);
		const u32x4 rawB = *reinterpret_cast<const u32x4*>(&rb[sb]);
		if (pa.outIdx)
			pa.outIdx[s] = SEG ? (rawA.y & 0x0FFFFFFFu) : rawA.x;   // SEG: device ids out as well as in
		if (q.outIdxB)
			q.outIdxB[s] = SEG ? (rawB.y & 0x0FFFFFFFu) : rawB.x;
		if (!SEG && pa.outFinal)
			pa.outFinal[s] = ((rawA.y | rawB.y) >> 28) & kFinal;
	}
	PairWaitTile<0>(a);
	PairWaitTile<0>(b);
}

bool PairTiledEligible(const ScanParams& a, const ScanParams& b)
{
	const uint32_t need = kPairBaseB + (b.hot + 1) * 256 + 512 + 2 * 528 + 64;
	return TiledEligible(a) && a.len >= 256 && (a.len / 128) % 2 == 0 && need <= kLdsPerBlock;
}

int LaunchPairTiled(const ScanParams& a, const ScanParams& b, uint32_t* outIdxB, hipStream_t stream, uint64_t warmBytes,
                    const uint32_t* segJ, uint32_t* guessA, uint32_t* guessB)
{
	PairParams q = {};
	q.a = a;
	q.b = b;
	q.outIdxB = outIdxB;
	const bool seg = guessA != nullptr;
	if (seg && (warmBytes % 256 != 0 || warmBytes > a.stride || !segJ || !guessB || a.hotRows != b.hotRows)) {
		SetError("pair kernel: a segment's warm-up is whole pairs of 128-byte tiles, no longer than the segment");
		return PIRE_HIP_EINVAL;
	}
	q.warmTiles = uint32_t(warmBytes / 128);
	q.segJ = segJ;
	q.guessA = guessA;
	q.guessB = guessB;
	q.a.n = a.n & ~uint64_t(63);   // whole 64-string tasks; the caller runs the remainder as two ordinary passes
	if (q.a.n == 0)
		return PIRE_HIP_OK;
	const uint32_t ldsBytes = (seg ? (a.hot + 1) * 256 : kPairBaseB + (b.hot + 1) * 256) + 512 + 2 * 528 + 64;
	int cus = 0;
	if (int rc = DeviceCUs(&cus))
		return rc;
	const void* kernel = seg ? reinterpret_cast<const void*>(ScanPairTiledKernel<true>) : reinterpret_cast<const void*>(ScanPairTiledKernel<false>);
	hipError_t e = SetDynamicLds(kernel, uint32_t(ldsBytes));
	if (e != hipSuccess)
		return HipFail(e, "hipFuncSetAttribute(LDS)");
	// one block per CU (the tables fill its LDS); a batch of fewer than 16 tasks per CU gets smaller blocks on more CUs
	// (the segmented scan of 64-256 MiB: 1 024-4 096 tasks)
	const uint64_t ntasks = q.a.n / 64;
	const uint64_t waves = seg ? std::max<uint64_t>(4, std::min<uint64_t>(16, (ntasks + cus - 1) / cus)) : 16;
	const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((ntasks + waves - 1) / waves, uint64_t(cus)));
	NoteKernel("pair_tiled", "pirehip::ScanPairTiledKernel");
	if (seg)
		hipLaunchKernelGGL(ScanPairTiledKernel<true>, dim3(unsigned(blocks)), dim3(unsigned(waves * 64)), ldsBytes, stream, q);
	else
		hipLaunchKernelGGL(ScanPairTiledKernel<false>, dim3(unsigned(blocks)), dim3(unsigned(waves * 64)), ldsBytes, stream, q);
	e = hipGetLastError();
	if (e != hipSuccess)
		return HipFail(e, "pair kernel launch");
	return PIRE_HIP_OK;
}

// Final = either (pair.h:69-72) for the two-pass form: fin[i] |= other[i]
__global__ void OrFinalKernel(uint8_t* fin, const uint8_t* other, uint64_t n)
{
	const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n)
		fin[i] |= other[i];
}

int LaunchOrFinal(uint8_t* fin, const uint8_t* other, uint64_t n, hipStream_t stream)
{
	if (n == 0)
		return PIRE_HIP_OK;
	hipLaunchKernelGGL(OrFinalKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, fin, other, n);
	const hipError_t e = hipGetLastError();
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "or-final kernel launch");
}

}  // namespace pirehip
