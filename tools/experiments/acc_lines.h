// EXPERIMENT, not built (profiles/r04_row_kernels_no_vgpr_tile.log: no gain).  Text lines through accumulation registers: the loader of the one-string-per-lane kernels that walk with a per-step
// action (counting.hip).  Round 4.
//
// A lane reads its string as 128-byte lines.  The line it will walk NEXT is on its way while it walks the current one,
// and the registers a load still owes data to must not be read, moved or reused by anything -- which the compiler cannot
// be told about an ordinary register (tiled.hip / stream.hip keep such tiles in ordinary registers and are audited
// instruction by instruction, tools/audit/inflight_registers.py; a first form of CountingRowKernel did, and register
// allocation copied the tile in front of its s_waitcnt: profiles/r04_counting_rows_vmcnt.log).  So here
//   a0..a31   the line on its way: written by eight global_load_dwordx4 (one per lane of a group of 8, IssueTileGroup's
//             pattern: instruction j loads the line of lane 8g+j, lane 8g+c its bytes [16c, 16c+16)),
//   a32..a63  the line being walked, transposed (lane 8g+j: its own line, chunk q in a[32+4q .. 35+4q]),
// and the walk takes one dword at a time out of a32..a63 (v_accvgpr_read_b32) -- no ordinary register tile at all.  To
// the compiler the sixteen tuples are values that these asm statements define in, and consume from, exactly those
// registers ("{a[n:m]}" constraints): it keeps its own spills out of them and has no reason to move them, nothing else
// here wants an accumulation register.  tests/test_build_audit.py checks that no other instruction names a0..a63.
// (LLVM gives a kernel that names an accumulation register half of the lane's register budget as ordinary registers:
// 64 + 64 for a block of 16 waves.)
#pragma once

#include "device_common.h"

namespace pirehip {

struct AccLines {
	u32x4 land[8];   // a[4j : 4j+3]
	u32x4 cur[8];    // a[32+4q : 35+4q]
};

// (whatever the registers hold: nothing is walked before the first Arrive)
__device__ __forceinline__ void AccInit(AccLines& L)
{
	asm volatile("" : "={a[0:3]}"(L.land[0]), "={a[4:7]}"(L.land[1]), "={a[8:11]}"(L.land[2]), "={a[12:15]}"(L.land[3]),
	             "={a[16:19]}"(L.land[4]), "={a[20:23]}"(L.land[5]), "={a[24:27]}"(L.land[6]), "={a[28:31]}"(L.land[7]));
	asm volatile("" : "={a[32:35]}"(L.cur[0]), "={a[36:39]}"(L.cur[1]), "={a[40:43]}"(L.cur[2]), "={a[44:47]}"(L.cur[3]),
	             "={a[48:51]}"(L.cur[4]), "={a[52:55]}"(L.cur[5]), "={a[56:59]}"(L.cur[6]), "={a[60:63]}"(L.cur[7]));
}

template <int J>
__device__ __forceinline__ void AccRequestOne(AccLines& L, uint32_t lo, uint32_t hi, uint32_t mine)
{
	const uint32_t l = GroupBroadcast<J>(lo);
	const uint32_t h = GroupBroadcast<J>(hi);
	const uint64_t a = ((uint64_t(h) << 32) | l) + mine;
	if constexpr (J == 0)
		asm volatile("global_load_dwordx4 a[0:3], %1, off" : "={a[0:3]}"(L.land[0]) : "v"(a));
	if constexpr (J == 1)
		asm volatile("global_load_dwordx4 a[4:7], %1, off" : "={a[4:7]}"(L.land[1]) : "v"(a));
	if constexpr (J == 2)
		asm volatile("global_load_dwordx4 a[8:11], %1, off" : "={a[8:11]}"(L.land[2]) : "v"(a));
	if constexpr (J == 3)
		asm volatile("global_load_dwordx4 a[12:15], %1, off" : "={a[12:15]}"(L.land[3]) : "v"(a));
	if constexpr (J == 4)
		asm volatile("global_load_dwordx4 a[16:19], %1, off" : "={a[16:19]}"(L.land[4]) : "v"(a));
	if constexpr (J == 5)
		asm volatile("global_load_dwordx4 a[20:23], %1, off" : "={a[20:23]}"(L.land[5]) : "v"(a));
	if constexpr (J == 6)
		asm volatile("global_load_dwordx4 a[24:27], %1, off" : "={a[24:27]}"(L.land[6]) : "v"(a));
	if constexpr (J == 7)
		asm volatile("global_load_dwordx4 a[28:31], %1, off" : "={a[28:31]}"(L.land[7]) : "v"(a));
}

// `src` = this lane's line address (lanes without a further line pass any valid address)
__device__ __forceinline__ void AccRequest(AccLines& L, uint64_t src, uint32_t lane)
{
	const uint32_t lo = uint32_t(src), hi = uint32_t(src >> 32);
	const uint32_t mine = (lane & 7u) << 4;
	AccRequestOne<0>(L, lo, hi, mine);
	AccRequestOne<1>(L, lo, hi, mine);
	AccRequestOne<2>(L, lo, hi, mine);
	AccRequestOne<3>(L, lo, hi, mine);
	AccRequestOne<4>(L, lo, hi, mine);
	AccRequestOne<5>(L, lo, hi, mine);
	AccRequestOne<6>(L, lo, hi, mine);
	AccRequestOne<7>(L, lo, hi, mine);
}

// The line that was requested has arrived: transposed (TransposeTile's butterflies, a column of eight dwords at a
// time) into a32..a63.  Everything this wave has asked for is waited for -- that is this line.
__device__ __forceinline__ void AccArrive(AccLines& L)
{
	const uint64_t lo1 = 0x5555555555555555ull, hi1 = 0xAAAAAAAAAAAAAAAAull;   // lane bit 0 clear / set
	const uint64_t lo2 = 0x3333333333333333ull, hi2 = 0xCCCCCCCCCCCCCCCCull;   // lane bit 1 clear / set
	uint32_t d[8];
	// column 0
	asm volatile("s_waitcnt vmcnt(0)\n\t" "v_accvgpr_read_b32 %0, a0" : "=v"(d[0]) : "{a[0:3]}"(L.land[0]));
	asm volatile("v_accvgpr_read_b32 %0, a4" : "=v"(d[1]) : "{a[4:7]}"(L.land[1]));
	asm volatile("v_accvgpr_read_b32 %0, a8" : "=v"(d[2]) : "{a[8:11]}"(L.land[2]));
	asm volatile("v_accvgpr_read_b32 %0, a12" : "=v"(d[3]) : "{a[12:15]}"(L.land[3]));
	asm volatile("v_accvgpr_read_b32 %0, a16" : "=v"(d[4]) : "{a[16:19]}"(L.land[4]));
	asm volatile("v_accvgpr_read_b32 %0, a20" : "=v"(d[5]) : "{a[20:23]}"(L.land[5]));
	asm volatile("v_accvgpr_read_b32 %0, a24" : "=v"(d[6]) : "{a[24:27]}"(L.land[6]));
	asm volatile("v_accvgpr_read_b32 %0, a28" : "=v"(d[7]) : "{a[28:31]}"(L.land[7]));
	ButterflyQuad4<1>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], lo1, hi1);
	ButterflyQuad4<2>(d[0], d[2], d[1], d[3], d[4], d[6], d[5], d[7], lo2, hi2);
	Butterfly4(d[0], d[4]); Butterfly4(d[1], d[5]); Butterfly4(d[2], d[6]); Butterfly4(d[3], d[7]);
	asm volatile("v_accvgpr_write_b32 a32, %1" : "+{a[32:35]}"(L.cur[0]) : "v"(d[0]));
	asm volatile("v_accvgpr_write_b32 a36, %1" : "+{a[36:39]}"(L.cur[1]) : "v"(d[1]));
	asm volatile("v_accvgpr_write_b32 a40, %1" : "+{a[40:43]}"(L.cur[2]) : "v"(d[2]));
	asm volatile("v_accvgpr_write_b32 a44, %1" : "+{a[44:47]}"(L.cur[3]) : "v"(d[3]));
	asm volatile("v_accvgpr_write_b32 a48, %1" : "+{a[48:51]}"(L.cur[4]) : "v"(d[4]));
	asm volatile("v_accvgpr_write_b32 a52, %1" : "+{a[52:55]}"(L.cur[5]) : "v"(d[5]));
	asm volatile("v_accvgpr_write_b32 a56, %1" : "+{a[56:59]}"(L.cur[6]) : "v"(d[6]));
	asm volatile("v_accvgpr_write_b32 a60, %1" : "+{a[60:63]}"(L.cur[7]) : "v"(d[7]));
	// column 1
	asm volatile("v_accvgpr_read_b32 %0, a1" : "=v"(d[0]) : "{a[0:3]}"(L.land[0]));
	asm volatile("v_accvgpr_read_b32 %0, a5" : "=v"(d[1]) : "{a[4:7]}"(L.land[1]));
	asm volatile("v_accvgpr_read_b32 %0, a9" : "=v"(d[2]) : "{a[8:11]}"(L.land[2]));
	asm volatile("v_accvgpr_read_b32 %0, a13" : "=v"(d[3]) : "{a[12:15]}"(L.land[3]));
	asm volatile("v_accvgpr_read_b32 %0, a17" : "=v"(d[4]) : "{a[16:19]}"(L.land[4]));
	asm volatile("v_accvgpr_read_b32 %0, a21" : "=v"(d[5]) : "{a[20:23]}"(L.land[5]));
	asm volatile("v_accvgpr_read_b32 %0, a25" : "=v"(d[6]) : "{a[24:27]}"(L.land[6]));
	asm volatile("v_accvgpr_read_b32 %0, a29" : "=v"(d[7]) : "{a[28:31]}"(L.land[7]));
	ButterflyQuad4<1>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], lo1, hi1);
	ButterflyQuad4<2>(d[0], d[2], d[1], d[3], d[4], d[6], d[5], d[7], lo2, hi2);
	Butterfly4(d[0], d[4]); Butterfly4(d[1], d[5]); Butterfly4(d[2], d[6]); Butterfly4(d[3], d[7]);
	asm volatile("v_accvgpr_write_b32 a33, %1" : "+{a[32:35]}"(L.cur[0]) : "v"(d[0]));
	asm volatile("v_accvgpr_write_b32 a37, %1" : "+{a[36:39]}"(L.cur[1]) : "v"(d[1]));
	asm volatile("v_accvgpr_write_b32 a41, %1" : "+{a[40:43]}"(L.cur[2]) : "v"(d[2]));
	asm volatile("v_accvgpr_write_b32 a45, %1" : "+{a[44:47]}"(L.cur[3]) : "v"(d[3]));
	asm volatile("v_accvgpr_write_b32 a49, %1" : "+{a[48:51]}"(L.cur[4]) : "v"(d[4]));
	asm volatile("v_accvgpr_write_b32 a53, %1" : "+{a[52:55]}"(L.cur[5]) : "v"(d[5]));
	asm volatile("v_accvgpr_write_b32 a57, %1" : "+{a[56:59]}"(L.cur[6]) : "v"(d[6]));
	asm volatile("v_accvgpr_write_b32 a61, %1" : "+{a[60:63]}"(L.cur[7]) : "v"(d[7]));
	// column 2
	asm volatile("v_accvgpr_read_b32 %0, a2" : "=v"(d[0]) : "{a[0:3]}"(L.land[0]));
	asm volatile("v_accvgpr_read_b32 %0, a6" : "=v"(d[1]) : "{a[4:7]}"(L.land[1]));
	asm volatile("v_accvgpr_read_b32 %0, a10" : "=v"(d[2]) : "{a[8:11]}"(L.land[2]));
	asm volatile("v_accvgpr_read_b32 %0, a14" : "=v"(d[3]) : "{a[12:15]}"(L.land[3]));
	asm volatile("v_accvgpr_read_b32 %0, a18" : "=v"(d[4]) : "{a[16:19]}"(L.land[4]));
	asm volatile("v_accvgpr_read_b32 %0, a22" : "=v"(d[5]) : "{a[20:23]}"(L.land[5]));
	asm volatile("v_accvgpr_read_b32 %0, a26" : "=v"(d[6]) : "{a[24:27]}"(L.land[6]));
	asm volatile("v_accvgpr_read_b32 %0, a30" : "=v"(d[7]) : "{a[28:31]}"(L.land[7]));
	ButterflyQuad4<1>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], lo1, hi1);
	ButterflyQuad4<2>(d[0], d[2], d[1], d[3], d[4], d[6], d[5], d[7], lo2, hi2);
	Butterfly4(d[0], d[4]); Butterfly4(d[1], d[5]); Butterfly4(d[2], d[6]); Butterfly4(d[3], d[7]);
	asm volatile("v_accvgpr_write_b32 a34, %1" : "+{a[32:35]}"(L.cur[0]) : "v"(d[0]));
	asm volatile("v_accvgpr_write_b32 a38, %1" : "+{a[36:39]}"(L.cur[1]) : "v"(d[1]));
	asm volatile("v_accvgpr_write_b32 a42, %1" : "+{a[40:43]}"(L.cur[2]) : "v"(d[2]));
	asm volatile("v_accvgpr_write_b32 a46, %1" : "+{a[44:47]}"(L.cur[3]) : "v"(d[3]));
	asm volatile("v_accvgpr_write_b32 a50, %1" : "+{a[48:51]}"(L.cur[4]) : "v"(d[4]));
	asm volatile("v_accvgpr_write_b32 a54, %1" : "+{a[52:55]}"(L.cur[5]) : "v"(d[5]));
	asm volatile("v_accvgpr_write_b32 a58, %1" : "+{a[56:59]}"(L.cur[6]) : "v"(d[6]));
	asm volatile("v_accvgpr_write_b32 a62, %1" : "+{a[60:63]}"(L.cur[7]) : "v"(d[7]));
	// column 3
	asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(d[0]) : "{a[0:3]}"(L.land[0]));
	asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(d[1]) : "{a[4:7]}"(L.land[1]));
	asm volatile("v_accvgpr_read_b32 %0, a11" : "=v"(d[2]) : "{a[8:11]}"(L.land[2]));
	asm volatile("v_accvgpr_read_b32 %0, a15" : "=v"(d[3]) : "{a[12:15]}"(L.land[3]));
	asm volatile("v_accvgpr_read_b32 %0, a19" : "=v"(d[4]) : "{a[16:19]}"(L.land[4]));
	asm volatile("v_accvgpr_read_b32 %0, a23" : "=v"(d[5]) : "{a[20:23]}"(L.land[5]));
	asm volatile("v_accvgpr_read_b32 %0, a27" : "=v"(d[6]) : "{a[24:27]}"(L.land[6]));
	asm volatile("v_accvgpr_read_b32 %0, a31" : "=v"(d[7]) : "{a[28:31]}"(L.land[7]));
	ButterflyQuad4<1>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], lo1, hi1);
	ButterflyQuad4<2>(d[0], d[2], d[1], d[3], d[4], d[6], d[5], d[7], lo2, hi2);
	Butterfly4(d[0], d[4]); Butterfly4(d[1], d[5]); Butterfly4(d[2], d[6]); Butterfly4(d[3], d[7]);
	asm volatile("v_accvgpr_write_b32 a35, %1" : "+{a[32:35]}"(L.cur[0]) : "v"(d[0]));
	asm volatile("v_accvgpr_write_b32 a39, %1" : "+{a[36:39]}"(L.cur[1]) : "v"(d[1]));
	asm volatile("v_accvgpr_write_b32 a43, %1" : "+{a[40:43]}"(L.cur[2]) : "v"(d[2]));
	asm volatile("v_accvgpr_write_b32 a47, %1" : "+{a[44:47]}"(L.cur[3]) : "v"(d[3]));
	asm volatile("v_accvgpr_write_b32 a51, %1" : "+{a[48:51]}"(L.cur[4]) : "v"(d[4]));
	asm volatile("v_accvgpr_write_b32 a55, %1" : "+{a[52:55]}"(L.cur[5]) : "v"(d[5]));
	asm volatile("v_accvgpr_write_b32 a59, %1" : "+{a[56:59]}"(L.cur[6]) : "v"(d[6]));
	asm volatile("v_accvgpr_write_b32 a63, %1" : "+{a[60:63]}"(L.cur[7]) : "v"(d[7]));
}

// dword W of chunk Q of the line being walked
template <int Q, int W>
__device__ __forceinline__ uint32_t AccDword(const AccLines& L)
{
	uint32_t x;
	if constexpr (Q == 0 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a32" : "=v"(x) : "{a[32:35]}"(L.cur[0]));
	if constexpr (Q == 0 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a33" : "=v"(x) : "{a[32:35]}"(L.cur[0]));
	if constexpr (Q == 0 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a34" : "=v"(x) : "{a[32:35]}"(L.cur[0]));
	if constexpr (Q == 0 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a35" : "=v"(x) : "{a[32:35]}"(L.cur[0]));
	if constexpr (Q == 1 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a36" : "=v"(x) : "{a[36:39]}"(L.cur[1]));
	if constexpr (Q == 1 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a37" : "=v"(x) : "{a[36:39]}"(L.cur[1]));
	if constexpr (Q == 1 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a38" : "=v"(x) : "{a[36:39]}"(L.cur[1]));
	if constexpr (Q == 1 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a39" : "=v"(x) : "{a[36:39]}"(L.cur[1]));
	if constexpr (Q == 2 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a40" : "=v"(x) : "{a[40:43]}"(L.cur[2]));
	if constexpr (Q == 2 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a41" : "=v"(x) : "{a[40:43]}"(L.cur[2]));
	if constexpr (Q == 2 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a42" : "=v"(x) : "{a[40:43]}"(L.cur[2]));
	if constexpr (Q == 2 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a43" : "=v"(x) : "{a[40:43]}"(L.cur[2]));
	if constexpr (Q == 3 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a44" : "=v"(x) : "{a[44:47]}"(L.cur[3]));
	if constexpr (Q == 3 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a45" : "=v"(x) : "{a[44:47]}"(L.cur[3]));
	if constexpr (Q == 3 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a46" : "=v"(x) : "{a[44:47]}"(L.cur[3]));
	if constexpr (Q == 3 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a47" : "=v"(x) : "{a[44:47]}"(L.cur[3]));
	if constexpr (Q == 4 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a48" : "=v"(x) : "{a[48:51]}"(L.cur[4]));
	if constexpr (Q == 4 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a49" : "=v"(x) : "{a[48:51]}"(L.cur[4]));
	if constexpr (Q == 4 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a50" : "=v"(x) : "{a[48:51]}"(L.cur[4]));
	if constexpr (Q == 4 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a51" : "=v"(x) : "{a[48:51]}"(L.cur[4]));
	if constexpr (Q == 5 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a52" : "=v"(x) : "{a[52:55]}"(L.cur[5]));
	if constexpr (Q == 5 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a53" : "=v"(x) : "{a[52:55]}"(L.cur[5]));
	if constexpr (Q == 5 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a54" : "=v"(x) : "{a[52:55]}"(L.cur[5]));
	if constexpr (Q == 5 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a55" : "=v"(x) : "{a[52:55]}"(L.cur[5]));
	if constexpr (Q == 6 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a56" : "=v"(x) : "{a[56:59]}"(L.cur[6]));
	if constexpr (Q == 6 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a57" : "=v"(x) : "{a[56:59]}"(L.cur[6]));
	if constexpr (Q == 6 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a58" : "=v"(x) : "{a[56:59]}"(L.cur[6]));
	if constexpr (Q == 6 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a59" : "=v"(x) : "{a[56:59]}"(L.cur[6]));
	if constexpr (Q == 7 && W == 0)
		asm volatile("v_accvgpr_read_b32 %0, a60" : "=v"(x) : "{a[60:63]}"(L.cur[7]));
	if constexpr (Q == 7 && W == 1)
		asm volatile("v_accvgpr_read_b32 %0, a61" : "=v"(x) : "{a[60:63]}"(L.cur[7]));
	if constexpr (Q == 7 && W == 2)
		asm volatile("v_accvgpr_read_b32 %0, a62" : "=v"(x) : "{a[60:63]}"(L.cur[7]));
	if constexpr (Q == 7 && W == 3)
		asm volatile("v_accvgpr_read_b32 %0, a63" : "=v"(x) : "{a[60:63]}"(L.cur[7]));
	return x;
}

}  // namespace pirehip
