// Shared by the kernels that walk an offset batch as RUNS OF CONSECUTIVE STRINGS (stream.hip: the plain scan; counting.hip:
// the counting scanners): how the batch is cut by cost key into one task per wave, a task into sub-tasks, a sub-task
// among the 64 lanes, and how a sub-task's string positions get into LDS.  DESIGN.md section 4.4.
#pragma once

#include "device_common.h"

namespace pirehip {

constexpr uint32_t kStreamMaxStrings = 1280;                    // strings of one sub-task (20 x 64; with the dense rows 155 of the 160 KiB)
constexpr uint32_t kStreamStageWords = kStreamMaxStrings + 16;  // their positions (m + 1 words) per wave, padded
constexpr uint32_t kStreamInf = 0xFFFFFFFFu;                    // "no boundary ahead": the lane's strings are over
constexpr uint32_t kStreamWaves = 16;

struct StreamGeom {
	uint32_t lambda;         // cost of a string boundary in bytes of walk
	uint32_t minTaskUnits;   // a wave is not started for less than this much key
	uint32_t maxStrings;     // strings of one sub-task (the dense rows: kStreamMaxStrings; the wide walk: what its image leaves room for)
};

typedef __attribute__((address_space(3))) uint32_t* LdsWordPtr;

// smallest i in [0, n] with key(i) >= T for two targets at once; key(i) = offsets[i] - off0 + lambda * i is strictly
// increasing and key(n) >= T (the caller clamps T).  64 probes per round and target: the interval shrinks 64-fold.
__device__ __forceinline__ void StreamSearch2(const uint64_t* off, uint64_t off0, uint64_t n, uint32_t lambda, uint64_t T0,
                                              uint64_t T1, uint64_t& r0, uint64_t& r1)
{
	const uint32_t lane = threadIdx.x & 63;
	uint64_t lo[2] = {0, 0}, hi[2] = {n, n};
	const uint64_t T[2] = {T0, T1};
	while (lo[0] < hi[0] || lo[1] < hi[1]) {
		uint64_t pos[2], key[2], step[2];
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			step[j] = (hi[j] - lo[j]) / 64 + 1;
			pos[j] = lo[j] + uint64_t(lane) * step[j];
			key[j] = ~0ull;
			if (pos[j] <= hi[j])
				key[j] = off[pos[j]] - off0 + uint64_t(lambda) * pos[j];
		}
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const unsigned long long ge = __ballot(key[j] >= T[j]);   // monotone: 0...01...1
			const uint32_t f = ge ? uint32_t(__builtin_ctzll(ge)) : 64u;   // number of probes below the target
			if (f == 0) {
				hi[j] = lo[j];
			} else {
				const uint64_t lastBelow = lo[j] + uint64_t(f - 1) * step[j];
				const uint64_t firstAt = lo[j] + uint64_t(f) * step[j];
				lo[j] = lastBelow + 1;
				if (f < 64 && firstAt < hi[j])
					hi[j] = firstAt;
			}
			lo[j] = Uniform64(lo[j]);
			hi[j] = Uniform64(hi[j]);
		}
	}
	r0 = lo[0];
	r1 = lo[1];
}

// This wave's task: strings [i0, i1) of the batch, equal steps of the cost key over the K waves that get one.  Returns
// false for a wave without a task.  (Two cooperative 64-ary searches over the offsets: issue it before the table copy.)
__device__ __forceinline__ bool StreamTaskOfWave(const uint64_t* offsets, uint64_t n, StreamGeom g, uint64_t& i0, uint64_t& i1)
{
	const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
	const uint32_t wavesPerBlock = blockDim.x >> 6;
	const uint64_t off0 = offsets[0], offN = offsets[n];
	const uint64_t totalKey = (offN - off0) + uint64_t(g.lambda) * n;
	const uint64_t W = uint64_t(gridDim.x) * wavesPerBlock;
	uint64_t K = totalKey / g.minTaskUnits;
	K = K < 1 ? 1 : K > W ? W : K;
	const uint64_t perTask = (totalKey + K - 1) / K;
	// fewer tasks than waves: every block takes its share of them (ceil(K / blocks) of its waves work), so that a batch
	// that does not fill the chip still uses every CU's LDS bandwidth instead of the first K / 16 CUs'
	const uint64_t perBlock = (K + gridDim.x - 1) / gridDim.x;
	const uint64_t gw = wave < perBlock ? uint64_t(blockIdx.x) * perBlock + wave : K;
	i0 = i1 = 0;
	if (gw < K) {
		const uint64_t T0 = gw * perTask, T1 = (gw + 1) * perTask;
		StreamSearch2(offsets, off0, n, g.lambda, T0 < totalKey ? T0 : totalKey, T1 < totalKey ? T1 : totalKey, i0, i1);
		if (gw == K - 1)
			i1 = n;
	}
	i0 = Uniform64(i0);   // wave-uniform by construction: keep them (and what is derived from them) in scalar registers
	i1 = Uniform64(i1);
	return gw < K;
}

// Sub-tasks of equal size (a task of 1 030 strings is not one of 1 024 and one of 6: the second would pay a whole pipeline
// start for six strings, and the launch ends with its slowest wave): strings per sub-task of a task of `taskStrings`.
__device__ __forceinline__ uint32_t StreamSubStrings(uint64_t taskStrings, uint32_t maxStrings = kStreamMaxStrings)
{
	const uint64_t subTasks = (taskStrings + maxStrings - 1) / maxStrings;
	return subTasks ? uint32_t((taskStrings + subTasks - 1) / subTasks) : 0;
}

// The positions of strings sub .. sub + m (m + 1 of them) into eo[0 .. m], relative to the 128-byte line that holds the
// first byte.  ONE round trip per batch of 11 x 64 positions: the offsets and the two the conversion needs are all
// requested before any is looked at (at kernel start every wave asks at once and a round trip is 4 us: six of them in a
// row were a sixth of the URL batch's time, profiles/r04_stream_stage_clocks.log).  Returns false when the positions do
// not fit 32 bits (a string of 4 GiB among short ones): nothing is written then.
__device__ __forceinline__ bool StreamStage(const uint64_t* offsets, uint64_t sub, uint32_t m, uint64_t textBase, LdsWordPtr eo,
                                            uint32_t lane, uint64_t& lineBase, uint32_t& lead, uint32_t& spanBytes)
{
	constexpr int kLoads = 11;   // per batch: two batches cover kStreamMaxStrings + 1 positions (all 21 at once: 42 registers)
	uint64_t v[kLoads];
#pragma unroll
	for (int j = 0; j < kLoads; ++j) {
		const uint32_t q = uint32_t(j) * 64 + lane;
		v[j] = offsets[sub + (q <= m ? q : m)];   // (clamped, not skipped: an unconditional load can be issued at once)
	}
	const uint64_t offA = offsets[sub], offZ = offsets[sub + m];
	const uint64_t firstByte = textBase + offA;
	lineBase = Uniform64(firstByte & ~uint64_t(127));
	lead = uint32_t(firstByte) & 127u;
	spanBytes = uint32_t(offZ - offA);
	if (offZ - offA >= 0xFFFF0000ull)
		return false;
#pragma unroll
	for (int j = 0; j < kLoads; ++j) {
		const uint32_t q = uint32_t(j) * 64 + lane;
		if (q <= m)
			eo[q] = lead + uint32_t(v[j] - offA);
	}
	if (m >= kLoads * 64) {   // the second batch (sub-tasks of more than 703 strings)
#pragma unroll
		for (int j = 0; j < kLoads; ++j) {
			const uint32_t q = uint32_t(kLoads + j) * 64 + lane;
			v[j] = offsets[sub + (q <= m ? q : m)];
		}
#pragma unroll
		for (int j = 0; j < kLoads; ++j) {
			const uint32_t q = uint32_t(kLoads + j) * 64 + lane;
			if (q <= m)
				eo[q] = lead + uint32_t(v[j] - offA);
		}
	}
	return true;
}

// The lane's strings [s0, s1) of the sub-task: equal steps of the key (position + lambda * index) over the 64 lanes, a
// binary search per lane in LDS.
__device__ __forceinline__ void StreamLaneSplit(LdsWordPtr eo, uint32_t m, uint32_t lead, uint32_t spanBytes, uint32_t lambda,
                                                uint32_t lane, uint32_t& s0, uint32_t& s1)
{
	const uint32_t keyAll = spanBytes + lambda * m;   // < 2^32: m <= kStreamMaxStrings, the span was checked by StreamStage
	const uint32_t perLane = (keyAll + 63) / 64;      // >= 1: m >= 1
	const uint32_t target = lane * perLane;
	uint32_t lo = 0, hi = m;
#pragma unroll 1
	for (int it = 0; it < 11; ++it) {   // 2^11 > kStreamMaxStrings + 1 candidates
		const uint32_t mid = (lo + hi) >> 1;
		const bool below = lo < hi && (eo[mid] - lead) + lambda * mid < target;
		const bool shrink = lo < hi && !below;
		lo = below ? mid + 1 : lo;
		hi = shrink ? mid : hi;
	}
	s0 = lo;
	// (Neighbouring lanes levelling their pieces out afterwards -- a lane hands its last string to the next one when that
	// shortens the longer of the two -- takes the longest lane of a wave from 23.8 to 22.2 windows on log lines and changes
	// nothing measurable: lanes that are done all read the same table line in the same state, one LDS dword for all of
	// them, and the walk is bound by the LDS reads of the lanes still at work.  profiles/r04_stream_lane_levelling.log.)
	s1 = uint32_t(__shfl_down(int(s0), 1));
	if (lane == 63)
		s1 = m;
}

}  // namespace pirehip
