// Scanner::Save() blob  ->  compact host table  ->  device image.
//
// Reference layout facts used here (all under /root/reference/pire):
//   blob        = Header(24 B) | Locals(48 B) | bool empty (padded to 8) | BufSize() bytes   scanners/multi.h:557-573
//   Header      = {Magic "PIRE", Version 7, PtrSize 8, MaxWordSize 16, Type 1, HdrSize 48}     scanners/common.h:44-63
//   buffer      = m_letters[264] u16 | m_final[finalTableSize] u64 | m_finalIndex[states] u64
//                 | m_transitions[states * RowSize] u32                                         multi.h:381-388
//   row         = ScannerRowHeader (ExitMasks<N>: N*4 u64 masks + u64 Flags; NoShortcuts: u64 Flags)
//                 then one u32 per letter class = signed byte distance to the target row        multi.h:55-67, 704-767
//   m_letters[] = class + HEADER_SIZE                                                           multi.h:375
// The ExitMasks themselves are a CPU (SSE2) skipping device and are not carried over: they never change a
// result (multi.h:925-934 asserts it); the GPU analogue is the absorbing-row early-out.

#include <algorithm>
#include <map>
#include <unordered_map>
#include <cstdlib>
#include <cmath>
#include <iterator>
#include <cstring>
#include <functional>
#include <numeric>
#include <stdexcept>

#include "internal.h"

namespace pirehip {

namespace {

struct RefHeader {
	uint32_t magic, version, ptrSize, maxWordSize, type, hdrSize;
};

struct RefLocals {
	uint32_t statesCount, lettersCount, regexpsCount, pad0;
	uint64_t initial;
	uint32_t finalTableSize, pad1;
	uint64_t relocationSignature, shortcuttingSignature;
};
static_assert(sizeof(RefHeader) == 24, "Header layout");
static_assert(sizeof(RefLocals) == 48, "Locals layout");

constexpr uint32_t kMagic = 0x45524950u;

size_t AlignUp(size_t v, size_t b) { return (v + b - 1) & ~(b - 1); }

int Bad(const char* msg)
{
	SetError(msg);
	return PIRE_HIP_EFORMAT;
}

// Expected visits per state for text drawn from a byte model, used only to decide WHICH rows get the fast dense LDS
// representation.  Any choice is correct; a better choice is faster.
//
// A corpus is one KIND of text, so the prior is a mixture of separate chains -- uniform printable ASCII, prose-like
// ASCII, UTF-8 prose, raw bytes -- not one chain over a mixed byte distribution: with "90 % printable + 10 % any byte"
// per step (round 1) every walk met a non-printable byte within a few dozen steps, the sticky "seen a strange byte"
// modes of patterns with dots soaked up the mass, and states that real text sits in for thousands of bytes (a mode
// entered by a rare trigger) were ranked below the 255th place: 5 % of all steps of the benchmark corpus trapped on
// ONE such state of set_b / set_d before adapt() (profiles/r02_cold_ranking.txt).
struct ByteModel {
	double weight;
	double prob[256];
};

void AddRange(ByteModel& m, int lo, int hi, double total)
{
	for (int b = lo; b <= hi; ++b)
		m.prob[b] += total / double(hi - lo + 1);
}

std::vector<ByteModel> PriorModels()
{
	std::vector<ByteModel> out(4);
	for (auto& m : out)
		memset(m.prob, 0, sizeof(m.prob));
	out[0].weight = 0.40;   // uniform printable ASCII (random identifiers, base64, the synthetic corpus)
	AddRange(out[0], 0x20, 0x7E, 1.0);
	out[1].weight = 0.35;   // prose / source-like ASCII: letter frequencies, spaces, some digits and punctuation
	{
		ByteModel& m = out[1];
		static const char letters[] = "etaoinshrdlucmwfgypbvkxjqz";
		static const double freq[] = {.100, .075, .065, .060, .057, .057, .053, .050, .050, .035, .033, .023, .023,
		                              .020, .019, .018, .016, .016, .015, .012, .008, .006, .0015, .001, .001, .0006};
		for (int i = 0; i < 26; ++i)
			m.prob[uint8_t(letters[i])] += freq[i] * 0.85;
		m.prob[' '] += 0.15;
		m.prob['\n'] += 0.012;
		m.prob['\t'] += 0.004;
		AddRange(m, 'A', 'Z', 0.03);
		AddRange(m, '0', '9', 0.03);
		static const char punct[] = ".,;:'\"-_/()=<>{}[]!?@#$%&*+\\|~^`";
		for (const char* c = punct; *c; ++c)
			m.prob[uint8_t(*c)] += 0.06 / double(sizeof(punct) - 1);
		double sum = 0;
		for (double q : m.prob)
			sum += q;
		for (double& q : m.prob)
			q /= sum;
	}
	out[2].weight = 0.15;   // UTF-8 prose: ASCII plus two-byte (Cyrillic / Latin-1 supplement) and some three-byte sequences
	{
		ByteModel& m = out[2];
		AddRange(m, 0x20, 0x7E, 0.55);
		m.prob[0xD0] += 0.11;
		m.prob[0xD1] += 0.06;
		m.prob[0xC3] += 0.03;
		AddRange(m, 0xE2, 0xE9, 0.02);
		AddRange(m, 0x80, 0xBF, 0.23);
	}
	out[3].weight = 0.10;   // raw bytes
	AddRange(out[3], 0, 255, 1.0);
	return out;
}

// Visits per byte of text under `model`, walking from `start`.  The chain runs until `steps` bytes or until its
// work budget is spent (tables whose probability spreads over thousands of states); the rest of the string is then
// charged to the distribution reached so far.
std::vector<double> VisitMassOf(const HostTable& t, uint32_t start, const ByteModel& model, int steps)
{
	const uint32_t N = t.states, C = t.letters;
	std::vector<double> classProb(C, 0.0);
	for (uint32_t b = 0; b < 256; ++b)
		classProb[t.cls[b]] += model.prob[b];
	std::vector<uint32_t> usedClass;
	for (uint32_t c = 0; c < C; ++c)
		if (classProb[c] > 0.0)
			usedClass.push_back(c);
	std::vector<double> p(N, 0.0), np(N, 0.0), mass(N, 0.0);
	std::vector<uint32_t> live{start}, nlive;
	p[start] = 1.0;
	uint64_t budget = 3000000;   // (state, class) pairs: a few milliseconds
	for (int step = 0; step < steps; ++step) {
		const uint64_t cost = uint64_t(live.size()) * usedClass.size();
		if (cost > budget || live.size() > 65536) {
			for (uint32_t s : live)
				mass[s] += p[s] * double(steps - step);
			break;
		}
		budget -= cost;
		nlive.clear();
		for (uint32_t s : live) {
			const double ps = p[s];
			mass[s] += ps;
			if (ps < 1e-9)
				continue;   // pruned: the tail of the distribution cannot decide a <= 255-row choice
			const uint32_t* row = &t.next[size_t(s) * C];
			for (uint32_t c : usedClass) {
				const uint32_t d = row[c];
				if (np[d] == 0.0)
					nlive.push_back(d);
				np[d] += ps * classProb[c];
			}
		}
		for (uint32_t s : live)
			p[s] = 0.0;
		for (uint32_t s : nlive) {
			p[s] = np[s];
			np[s] = 0.0;
		}
		live.swap(nlive);
	}
	for (double& m : mass)
		m /= double(steps);
	return mass;
}

// People scan text because it sometimes matches.  A match of an unanchored pattern leaves the automaton in a sticky
// "seen r" copy of its idle states for the rest of the string -- states random text never reaches, so no byte model
// ranks them, yet a corpus with one match per string spends half its bytes there.  Breadth-first from `hub` over the
// classes printable text has: the first state met for every distinct AcceptedRegexps set is a MODE (shortest
// witnesses first: single matches, then texts that match two patterns, ...; the first kMaxModes of them); each gets the
// text chain continued from it, and its witness path one visit per string.
// Round 3: a `$`-anchored pattern only becomes Final on EndMark, which is not a text class, so the search above never
// met its matches and the states ALONG its witness ("...ABCDEFGHIJKLMNOPQRSTUVWXYZ" at the end of a line) had no
// mass at all: the benchmark corpus plants such tails, 40 of the 85 states it visits on set_a had no dense row and not
// even a compact one, and the two trapped chunks per string cost 15 % of the kernel before adapt()
// (profiles/r02_cold_ranking.txt, BENCH_r02.json value_before_adapt).  So a state whose End() is accepting is a
// mode as well (keyed by that set, tagged): its witness path gets the same one visit per string.
constexpr uint32_t kMaxModes = 24;
void AddMatchModes(const HostTable& t, uint32_t hub, const ByteModel& text, double weight, std::vector<double>& mass)
{
	const uint32_t N = t.states, C = t.letters;
	std::vector<uint8_t> textClass(C, 0);
	for (uint32_t b = 0x20; b <= 0x7E; ++b)
		textClass[t.cls[b]] = 1;
	const uint32_t endCls = t.cls[kEndMark];
	std::vector<uint32_t> parent(N, UINT32_MAX), queue, modes, endModes;
	queue.reserve(N);
	queue.push_back(hub);
	parent[hub] = hub;
	std::map<std::vector<uint64_t>, uint32_t> seen, seenAtEnd;
	auto keyOf = [&](uint32_t s) {
		return std::vector<uint64_t>(t.acceptIds.begin() + t.acceptOff[s], t.acceptIds.begin() + t.acceptOff[s + 1]);
	};
	for (size_t head = 0; head < queue.size() && (modes.size() < kMaxModes || endModes.size() < kMaxModes); ++head) {
		const uint32_t s = queue[head];
		if (modes.size() < kMaxModes && t.acceptOff[s + 1] > t.acceptOff[s] && seen.emplace(keyOf(s), s).second)
			modes.push_back(s);
		const uint32_t* row = &t.next[size_t(s) * C];
		const uint32_t e = row[endCls];
		if (endModes.size() < kMaxModes && t.acceptOff[s + 1] == t.acceptOff[s] && t.acceptOff[e + 1] > t.acceptOff[e] &&
		    seenAtEnd.emplace(keyOf(e), s).second)
			endModes.push_back(s);
		for (uint32_t c = 0; c < C; ++c)
			if (textClass[c] && parent[row[c]] == UINT32_MAX) {
				parent[row[c]] = s;
				queue.push_back(row[c]);
			}
	}
	const double w = weight / double(std::max<size_t>(1, modes.size() + endModes.size()));
	for (uint32_t f : modes) {
		const std::vector<double> after = VisitMassOf(t, f, text, 1024);
		for (uint32_t s = 0; s < N; ++s)
			mass[s] += w * after[s];
		for (uint32_t s = f; s != hub; s = parent[s])
			mass[s] += w / 1024.0;   // the witness itself: once per string
	}
	for (uint32_t f : endModes) {
		for (uint32_t s = f; s != hub; s = parent[s])
			mass[s] += w / 1024.0;   // a match at the end of the string: its witness once per string, nothing after it
		mass[t.next[size_t(f) * C + endCls]] += w / 1024.0;
	}
}

// Mixture over the corpus kinds; every chain runs kPriorSteps bytes (a mode entered by a rare trigger needs a long
// string to show its weight).
constexpr int kPriorSteps = 2048;
std::vector<double> VisitMass(const HostTable& t, uint32_t start)
{
	static const std::vector<ByteModel> models = PriorModels();
	std::vector<double> mass(t.states, 0.0);
	for (const ByteModel& m : models) {
		const std::vector<double> one = VisitMassOf(t, start, m, kPriorSteps);
		for (uint32_t s = 0; s < t.states; ++s)
			mass[s] += m.weight * one[s];
	}
	return mass;
}

// How much of the ranking's mass (`score`, reference numbering) lies on states without a dense row / without a wide row in
// the CURRENT numbering: what LaunchTiled chooses the walk by (any choice is correct).
void MeasureShares(HostTable& t, const std::vector<double>& score)
{
	double total = 0, dense = 0, wide = 0;
	for (uint32_t pid = 0; pid < t.states; ++pid) {
		const double v = std::max(0.0, score[t.origOfPerm[pid]]);
		total += v;
		if (pid < t.hot)
			dense += v;
		if (pid < t.wide)
			wide += v;
	}
	t.outsideDense = total > 0 ? float(std::max(0.0, 1.0 - dense / total)) : 0.0f;
	t.outsideWide = total > 0 && t.wide ? float(std::max(0.0, 1.0 - wide / total)) : t.outsideDense;
}

// ---- the zipped image of the wide walk (internal.h MakeWideLayout; round 6) ---------------------------------------------
// Which states of the ranking get a row of their own, which a header + <= 3 exceptions against one of those rows, in
// rank order until the CU's LDS is full: a state whose row differs from the row of a state that already has one in at most
// three letters costs 10 bytes, any other a row (pitch + 8 bytes) -- while `cap` rows are not used up; after that such a
// state stays outside the tier (74 bytes buy seven zipped states).  `order`: the ranking, the first `hot` states keep
// their places (they have rows: the dense-row kernels number them 0..hot-1).  Any outcome is correct: a state outside the
// tier is walked through the exact table in memory.
struct ZipPlan {
	std::vector<uint32_t> full, zipped, rest;   // reference state indices, in rank order
	std::vector<uint16_t> base;                 // per zipped state: position in `full` of the row it leans on
	double inside = 0;                          // mass of the ranking on the tier
};

// number of letters two rows differ in, given up at more than `limit`
inline uint32_t RowDistance(const uint32_t* a, const uint32_t* b, uint32_t letters, uint32_t limit)
{
	uint32_t d = 0, c = 0;
	for (; c + 8 <= letters; c += 8) {
		for (uint32_t k = 0; k < 8; ++k)
			d += a[c + k] != b[c + k];
		if (d > limit)
			return d;
	}
	for (; c < letters; ++c)
		d += a[c] != b[c];
	return d;
}

ZipPlan PlanZip(const HostTable& t, const std::vector<uint32_t>& order, const std::vector<double>& score, uint32_t hot, uint32_t cap)
{
	const uint32_t N = t.states, C = t.letters;
	ZipPlan z;
	const uint32_t pitch = WidePitch(C);
	const uint64_t fixed = MakeWideLayout(0, C, t.regexps <= kMaxLdsCountRegexps ? t.regexps : 0, 0).total + 64 + sizeof(uint32_t) * 8;
	const uint64_t rowCost = pitch + 4 + 4, zipCost = 4 + 2 * kZipExceptions;   // row + header + visit counter / header + targets
	uint64_t used = fixed + rowCost /* the escape row */ + 16;
	cap = std::min(cap, kZipMaxFull);
	auto mass = [&](uint32_t s) { return score[s] > 0 ? score[s] : 0.0; };
	uint32_t lastBase = 0;
	for (uint32_t i = 0; i < N; ++i) {
		const uint32_t s = order[i];
		if (used + std::max(rowCost, zipCost) > kLdsPerBlock || z.full.size() + z.zipped.size() >= 65000) {
			z.rest.insert(z.rest.end(), order.begin() + i, order.end());
			break;
		}
		const uint32_t* row = &t.next[size_t(s) * C];
		int found = -1;
		if (i >= hot && !(t.flags[s] & kAbsorbing)) {   // (an absorbing state keeps its row: the early-out reads the row's flags)
			// the row the previous zipped state leans on first: states of one neighbourhood follow each other in the ranking
			if (!z.full.empty() && RowDistance(row, &t.next[size_t(z.full[lastBase]) * C], C, kZipExceptions) <= kZipExceptions)
				found = int(lastBase);
			for (uint32_t f = 0; found < 0 && f < z.full.size(); ++f)
				if (RowDistance(row, &t.next[size_t(z.full[f]) * C], C, kZipExceptions) <= kZipExceptions)
					found = int(f);
		}
		if (found >= 0) {
			z.zipped.push_back(s);
			z.base.push_back(uint16_t(found));
			lastBase = uint32_t(found);
			used += zipCost;
			z.inside += mass(s);
		} else if (i < hot || z.full.size() < cap) {
			z.full.push_back(s);
			used += rowCost;
			z.inside += mass(s);
		} else {
			z.rest.push_back(s);
		}
	}
	return z;
}

// The choice between the plain rows and the zipped image (pire_hip_config.zip_variant: 0 by the ranking's masses once
// adapt() has measured them, 1 never, 2 always), and the zipped numbering: full states, zipped states, the rest.
void ChooseZip(HostTable& t, std::vector<uint32_t>& order, const std::vector<double>& score)
{
	const uint32_t N = t.states, C = t.letters;
	const bool was = t.zipFull != 0;
	t.zipFull = 0;
	t.zipBase.clear();
	t.zipPlainOutside = t.zipOutside = 0;
	const uint32_t variant = GetConfig().zip_variant;
	if (variant == 1 || !t.wide || t.wide >= N || N > 65000 || C > 126)
		return;   // the plain rows hold every state / ids or letter classes the image's fields cannot hold
	double total = 0, plainInside = 0;
	for (uint32_t i = 0; i < N; ++i) {
		const double v = score[order[i]] > 0 ? score[order[i]] : 0.0;
		total += v;
		if (i < t.wide)
			plainInside += v;
	}
	if (total <= 0)
		return;
	t.zipPlainOutside = float(std::max(0.0, 1.0 - plainInside / total));
	// Worth planning at all?  Without a measurement the byte model's masses say little about thousands of states; with one,
	// the plain rows are the faster walk while the working set fits them (two LDS instructions per byte, not three).
	// Offset batches (one string per lane) are another matter: that kernel is bound by its vector instructions, of which the zipped
	// step has 20 where the plain one has 5 -- URL batches of a blacklist scanner with 26 000 states, 2.6 % of the steps outside
	// the plain rows: 784 GB/s on them, 570 on the zipped image that leaves 0.5 % outside (profiles/r06_wide_curve.jsonl).  A table
	// whose bytes came mostly that way is zipped only from a share the plain rows really cannot live with.
	const bool offsetBatches = t.offsetBatchShare > 0.5f;
	const float enter = offsetBatches ? (was ? 0.05f : 0.08f) : (was ? 0.002f : 0.004f);   // (hysteresis: a zipped table stays zipped at a share that would not have made it one)
	if (variant != 2 && (!t.massMeasured || t.zipPlainOutside < enter))
		return;
	ZipPlan best;
	double bestInside = -1;
	for (uint32_t cap : {kZipMaxFull, 767u, 511u, 383u}) {   // (large alphabets: a row is 230 bytes, 23 zipped states)
		if (cap < t.hot)
			continue;
		ZipPlan z = PlanZip(t, order, score, t.hot, cap);
		// (caps in falling order, strictly more of the ranking's mass inside the tier to be taken: where every sampled state fits
		// anyway the first, with the most rows, stays.  The masses are samples of a long tail: the share they leave outside the tier
		// is several times too low -- the states no sample has hit are visited too -- and which cap wins is only roughly right; a
		// Good-Turing estimate for the unseen mass and a tie-break on the number of states were tried and changed no choice.  A BYTE
		// form of the header -- a letter compared by one v_cmp_eq_u32_sdwa, 13 vector instructions per step instead of 19, at most 255
		// rows to lean on -- was built too: + 3 %; the zipped walk is bound by its three LDS instructions per byte.  Not kept.)
		if (z.inside > bestInside) {
			bestInside = z.inside;
			best = std::move(z);
		}
	}
	if (bestInside < 0 || best.zipped.empty())
		return;
	const float zipOutside = float(std::max(0.0, 1.0 - bestInside / total));
	t.zipOutside = zipOutside;
	if (variant != 2 && !(zipOutside < 0.6f * t.zipPlainOutside))
		return;
	// (the plan counted bytes; what decides is the layout the kernels use, with the ragged kernel's work record behind it)
	const uint32_t regexps = t.regexps <= kMaxLdsCountRegexps ? t.regexps : 0;
	while (!best.zipped.empty() &&
	       MakeWideLayout(uint32_t(best.full.size() + best.zipped.size()), C, regexps, uint32_t(best.full.size())).total + 128 > kLdsPerBlock) {
		best.rest.insert(best.rest.begin(), best.zipped.back());
		best.zipped.pop_back();
		best.base.pop_back();
	}
	if (best.zipped.empty())
		return;
	order.clear();
	order.insert(order.end(), best.full.begin(), best.full.end());
	order.insert(order.end(), best.zipped.begin(), best.zipped.end());
	order.insert(order.end(), best.rest.begin(), best.rest.end());
	t.zipFull = uint32_t(best.full.size());
	t.wide = uint32_t(best.full.size() + best.zipped.size());
	t.zipBase = std::move(best.base);
}

// Renumber "hot first" by `score` (higher = hotter) and build the dense LDS rows.
void PermuteByScore(HostTable& t, const std::vector<double>& score)
{
	const uint32_t N = t.states, C = t.letters;
	std::vector<uint32_t> order(N);
	std::iota(order.begin(), order.end(), 0u);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return score[a] > score[b]; });
	{
		// how concentrated the traffic is: lanes in the SAME state that read the same dword of its row are served together
		// by the LDS, lanes in different states collide in the bank (byte >> 2) & 63
		double total = 0;
		for (uint32_t i = 0; i < N; ++i)
			total += score[i] > 0 ? score[i] : 0;
		t.topShare = total > 0 && N ? float((score[order[0]] > 0 ? score[order[0]] : 0) / total) : 1.0f;
	}

	t.hot = std::min<uint32_t>(N, kMaxHotRows - 1);
#ifdef PIRE_HIP_TUNING
	if (const char* cap = getenv("PIRE_HIP_MAX_HOT"))   // knob: A/B measurements (fewer dense rows = less LDS per block)
		t.hot = std::min<uint32_t>(t.hot, std::max(1, atoi(cap)));
#endif
	// inside the hot set the order is free: plain states first, then Dead ones, then Final ones, so that "hot and
	// Final" and "hot and Final or Dead" are one compare each (HalfFinalScanner's per-step TakeAction, the prefix
	// searches' stop conditions; ragged.hip tests the largest id a 16-byte chunk went through against them)
	auto rankOf = [&](uint32_t s) { return (t.flags[s] & kFinal) ? 2 : (t.flags[s] & kDead) ? 1 : 0; };
	std::stable_sort(order.begin(), order.begin() + t.hot, [&](uint32_t a, uint32_t b) { return rankOf(a) < rankOf(b); });
	t.hotDeadLo = 0;
	while (t.hotDeadLo < t.hot && rankOf(order[t.hotDeadLo]) == 0)
		++t.hotDeadLo;
	t.hotFinalLo = t.hotDeadLo;
	while (t.hotFinalLo < t.hot && rankOf(order[t.hotFinalLo]) == 1)
		++t.hotFinalLo;
	// Dense id 0 = the state strings start in with Begin() (Initialize(), then Step(BeginMark): run.h:375) whenever that
	// state is among the plain dense rows -- any order inside the group is as good as any other, and the stream kernel
	// restarts a lane's walk at a string boundary by selecting the CONSTANT 0 as the row byte of the lookup address
	// (v_perm_b32 selector 0x0c), which takes the restart off the dependent chain (stream.hip StepChunkB).
	{
		const uint32_t begin = t.next[size_t(t.initial) * C + t.cls[kBeginMark]];
		for (uint32_t i = 0; i < t.hotDeadLo; ++i)
			if (order[i] == begin) {
				std::swap(order[0], order[i]);
				break;
			}
	}
	t.compact = GetConfig().no_compact ? 0 : CompactCapacity(t.hot, C, t.regexps, N);   // knob: A/B measurements
	// the wide walk's image (wide.hip): only for tables with more states than dense rows -- a table that fits the dense
	// rows never leaves them
	t.wide = N > t.hot ? WideCapacity(C, t.regexps, N) : 0;
	ChooseZip(t, order, score);   // (may renumber everything behind the dense rows and raise t.wide)
	t.origOfPerm = order;
	MeasureShares(t, score);
	t.permOfOrig.assign(N, 0);
	for (uint32_t pid = 0; pid < N; ++pid)
		t.permOfOrig[order[pid]] = pid;

	// dense rows: entry = perm id of next state if that is hot, else the trap id (== hot)
	const uint32_t H = t.hot;
	t.hotRows.assign(size_t(H + 1) * 256, uint8_t(H));
	for (uint32_t pid = 0; pid < H; ++pid) {
		const uint32_t* row = &t.next[size_t(order[pid]) * C];
		uint8_t* out = &t.hotRows[size_t(pid) * 256];
		for (uint32_t b = 0; b < 256; ++b) {
			const uint32_t d = t.permOfOrig[row[t.cls[b]]];
			out[b] = d < H ? uint8_t(d) : uint8_t(H);
		}
	}
	t.hotFlags.assign(256, 0);
	for (uint32_t pid = 0; pid < H; ++pid)
		t.hotFlags[pid] = t.flags[order[pid]];
}

void ChooseHotAndPermute(HostTable& t)
{
	const uint32_t N = t.states, C = t.letters;
	// packed per-regexp increments of HalfFinalScanner::TakeAction (half_final.h:156-164): the final list of a
	// state may name a regexp several times (BuildFinals, half_final.h:202-213)
	t.inc64.assign(N, 0);
	t.incPacked = t.regexps <= 8;
	for (uint32_t s = 0; s < N && t.incPacked; ++s) {
		uint32_t mult[8] = {0};
		for (uint64_t k = t.acceptOff[s]; k < t.acceptOff[s + 1]; ++k)
			if (t.acceptIds[k] >= 8 || ++mult[t.acceptIds[k]] > 255)
				t.incPacked = false;
		for (uint32_t r = 0; r < 8; ++r)
			t.inc64[s] |= uint64_t(mult[r]) << (8 * r);
	}
	// every string starts at Initialize() and (normally) takes BeginMark first; rank from both
	std::vector<double> mass = VisitMass(t, t.initial);
	{
		const uint32_t afterBegin = t.next[size_t(t.initial) * C + t.cls[kBeginMark]];
		std::vector<double> m2 = VisitMass(t, afterBegin);
		for (uint32_t s = 0; s < N; ++s)
			mass[s] += m2[s];
		// the idle state of text = where the printable chain from Begin() sits most; matches start from there
		static const std::vector<ByteModel> models = PriorModels();
		const std::vector<double> idle = VisitMassOf(t, afterBegin, models[0], 256);
		const uint32_t hub = uint32_t(std::max_element(idle.begin(), idle.end()) - idle.begin());
		AddMatchModes(t, hub, models[0], 0.5, mass);
		mass[afterBegin] += 1.0;
		mass[t.initial] += 1.0;
	}
	// where the model's text spends its time: searches over scanners that are soon Dead (lexers) or soon Final end
	// after a few bytes, which decides between the kernels of the prefix searches (exact.hip LaunchPrefix)
	double all = 0, dead = 0, fin = 0;
	for (uint32_t s = 0; s < N; ++s) {
		all += mass[s];
		if (t.flags[s] & kDead)
			dead += mass[s];
		if (t.flags[s] & kFinal)
			fin += mass[s];
	}
	t.deadShare = all > 0 ? float(dead / all) : 0.0f;
	t.finalShare = all > 0 ? float(fin / all) : 0.0f;
	if (GetConfig().prior_flat)   // knob for the tests of adapt(): a prior that knows nothing (dense rows = the
		for (uint32_t s = 0; s < N; ++s)  // first 255 states by index), so that adapt() has everything to learn
			mass[s] = 1.0 / double(1 + s);
	const double top = *std::max_element(mass.begin(), mass.end());
	t.priorMass.resize(N);
	for (uint32_t s = 0; s < N; ++s)
		t.priorMass[s] = top > 0 ? mass[s] / top : 0.0;
	PermuteByScore(t, t.priorMass);
	if (GetConfig().prior_flat)
		t.outsideDense = t.outsideWide = 0;   // a prior that knows nothing makes no claim about where the walk will be
}

}  // namespace

namespace {

// Pire::SimpleScanner (scanners/simple.h): one regexp, rows of MaxChar + 1 size_t slots = [tag, byte shift per Char],
// no letter classes, never Dead.  Save format: scanner_io.cpp:35-49.  The dense 264-column rows are folded into the
// same HostTable as a Scanner: columns with identical transitions become one letter class, so every kernel, the
// compact tier and the accessors work unchanged; StateIndex (simple.h:154-157) is the row number.
int BuildSimpleHostTable(const uint8_t* p, size_t len, size_t pos, HostTable* out)
{
	struct SimpleLocals {
		uint64_t statesCount, initial;   // simple.h:171-174; initial on disk = byte offset from m_transitions
	} m;
	if (len < pos + sizeof(m) + 8)
		return Bad("EOF reached while reading the scanner locals");
	memcpy(&m, p + pos, sizeof(m));
	pos += sizeof(m);
	const bool empty = p[pos] != 0;
	pos += 8;
	constexpr uint64_t kRowSlots = kMaxChar + 1;   // STATE_ROW_SIZE, simple.h:43
	constexpr uint64_t kStride = kRowSlots * 8;

	HostTable& t = *out;
	t = HostTable();
	t.scannerType = 2;
	t.headerSize = 1;   // the tag slot in front of the transitions
	t.rowStride = uint32_t(kStride);
	t.empty = empty;
	if (m.statesCount == 0 || m.statesCount > (1u << 24))
		return Bad("Corrupt scanner: bad state count");
	if (m.initial % kStride != 8 || m.initial / kStride >= m.statesCount)
		return Bad("Corrupt scanner: initial state out of range");
	t.states = uint32_t(m.statesCount);
	t.initial = uint32_t(m.initial / kStride);
	t.regexps = empty ? 0 : 1;       // RegexpsCount(), simple.h:59
	t.cls.assign(kMaxChar, 0);

	if (empty) {
		// aliases Null() = Fsm::MakeFalse() compiled: zeroed rows (every shift 0), no final state (simple.h:187-191, 234)
		t.letters = 1;
		t.next.resize(t.states);
		for (uint32_t s = 0; s < t.states; ++s)
			t.next[s] = s;
		t.flags.assign(t.states, uint8_t(kAbsorbing));
		t.acceptOff.assign(size_t(t.states) + 1, 0);
		t.blobBytes = pos;
		ChooseHotAndPermute(t);
		return PIRE_HIP_OK;
	}

	const uint64_t bufSize = kStride * m.statesCount;   // BufSize(), simple.h:160-163
	if (len < pos + bufSize)
		return Bad("EOF reached while reading the scanner buffer");
	t.refBufSize = bufSize;
	t.blobBytes = pos + bufSize;
	const uint8_t* buf = p + pos;

	// decode every (state, Char) -> next state
	std::vector<uint32_t> dense(size_t(t.states) * kMaxChar);
	for (uint32_t s = 0; s < t.states; ++s)
		for (uint32_t c = 0; c < kMaxChar; ++c) {
			uint64_t shift;
			memcpy(&shift, buf + (uint64_t(s) * kRowSlots + 1 + c) * 8, 8);
			const uint64_t dest = uint64_t(s) * kStride + shift;   // state += shift (mod 2^64), simple.h:78-79
			if (dest % kStride != 0 || dest / kStride >= m.statesCount)
				return Bad("Corrupt scanner: transition out of range");
			dense[size_t(s) * kMaxChar + c] = uint32_t(dest / kStride);
		}
	// letter classes = distinct columns, numbered in order of first appearance
	std::map<std::vector<uint32_t>, uint16_t> classOf;
	std::vector<uint32_t> col(t.states);
	std::vector<uint32_t> repr;   // representative Char of each class
	for (uint32_t c = 0; c < kMaxChar; ++c) {
		for (uint32_t s = 0; s < t.states; ++s)
			col[s] = dense[size_t(s) * kMaxChar + c];
		auto it = classOf.find(col);
		if (it == classOf.end()) {
			it = classOf.emplace(col, uint16_t(repr.size())).first;
			repr.push_back(c);
		}
		t.cls[c] = it->second;
	}
	t.letters = uint32_t(repr.size());
	t.next.resize(size_t(t.states) * t.letters);
	t.flags.resize(t.states);
	t.acceptOff.assign(size_t(t.states) + 1, 0);
	for (uint32_t s = 0; s < t.states; ++s) {
		bool absorbing = true;
		for (uint32_t l = 0; l < t.letters; ++l) {
			const uint32_t d = dense[size_t(s) * kMaxChar + repr[l]];
			t.next[size_t(s) * t.letters + l] = d;
			absorbing = absorbing && d == s;
		}
		uint64_t tag;
		memcpy(&tag, buf + uint64_t(s) * kStride, 8);
		const bool fin = tag != 0;                       // Final(), simple.h:62; Dead() is always false (64)
		t.flags[s] = uint8_t((fin ? kFinal : 0) | (absorbing ? kAbsorbing : 0));
		t.acceptOff[s] = t.acceptIds.size();
		if (fin)
			t.acceptIds.push_back(0);                    // AcceptedRegexps() = {0} when Final, simple.h:66-68, 193-203
	}
	t.acceptOff[t.states] = t.acceptIds.size();
	ChooseHotAndPermute(t);
	return PIRE_HIP_OK;
}

}  // namespace

void ChooseHotAndPermuteExported(HostTable& t) { ChooseHotAndPermute(t); }

int BuildHostTable(const void* blob, size_t len, HostTable* out)
{
	const uint8_t* p = static_cast<const uint8_t*>(blob);
	if (!p || len < sizeof(RefHeader))
		return Bad("EOF reached while reading the scanner header");
	RefHeader h;
	memcpy(&h, p, sizeof(h));
	// Header::Validate, common.h:65-77
	if (h.magic != kMagic || h.ptrSize != 8 || h.maxWordSize != 16)
		return Bad("Serialized regexp incompatible with your system");
	if (h.version != 7 && h.version != 6)
		return Bad("You are trying to used an incompatible version of a serialized regexp");
	if (h.type == 2 /* ScannerIOTypes::SimpleScanner, common.h:37 */ && h.hdrSize == 16)
		return BuildSimpleHostTable(p, len, AlignUp(sizeof(RefHeader), 8), out);
	if (h.type != 1 /* ScannerIOTypes::Scanner */ || h.hdrSize != sizeof(RefLocals))
		return Bad("Serialized regexp incompatible with your system");
	size_t pos = AlignUp(sizeof(RefHeader), 8);

	if (len < pos + sizeof(RefLocals))
		return Bad("EOF reached while reading the scanner locals");
	RefLocals m;
	memcpy(&m, p + pos, sizeof(m));
	pos += AlignUp(sizeof(RefLocals), 8);
	if (m.relocationSignature != 1)
		return Bad("Type mismatch while mmapping Pire::Scanner");
	uint32_t maskCount;
	if (m.shortcuttingSignature == 0x1000)
		maskCount = 0;
	else if ((m.shortcuttingSignature >> 8) == 0x20 && (m.shortcuttingSignature & 0xFF) != 0)
		maskCount = uint32_t(m.shortcuttingSignature & 0xFF);
	else
		return Bad("This scanner has different shortcutting type");

	if (len < pos + 1)
		return Bad("EOF reached while reading the scanner");
	const bool empty = p[pos] != 0;
	pos += 8;

	HostTable& t = *out;
	t = HostTable();
	t.scannerType = 1;
	const uint32_t flagsOff = maskCount * 4 * 8;
	t.headerSize = (flagsOff + 8) / 4;
	t.empty = empty;

	if (empty) {
		// Scanner::Null() = Fsm::MakeFalse() compiled (multi.h:339-344): it never matches.  Model it as one
		// non-final state with a single letter class that loops (tests/pire_ut.cpp:760-830 pins the behaviour).
		t.states = 1;
		t.letters = 1;
		t.regexps = 0;
		t.initial = 0;
		t.rowStride = uint32_t(AlignUp(1 + t.headerSize, 4) * 4);
		t.cls.assign(kMaxChar, 0);
		t.next.assign(1, 0);
		t.flags.assign(1, uint8_t(kDead | kAbsorbing));
		t.acceptOff.assign(2, 0);
		t.blobBytes = pos;
		ChooseHotAndPermute(t);
		return PIRE_HIP_OK;
	}

	if (m.statesCount == 0 || m.lettersCount == 0 || m.lettersCount > kMaxChar)
		return Bad("Corrupt scanner: bad state or letter count");
	// The kernels pack a state id into 28 bits (FinRec) and index counters by regexp id: bound both here.  The
	// reference itself never builds more than 200 000 determinised / 80 000 glued states (fsm.cpp:1018, multi.h:1100).
	if (m.statesCount >= (1u << 28))
		return Bad("Corrupt scanner: state count out of range");
	if (m.regexpsCount > (1u << 20))
		return Bad("Corrupt scanner: regexp count out of range");
	const size_t rowSize = AlignUp(size_t(m.lettersCount) + t.headerSize, 16 / 4);
	const size_t bufSize = AlignUp(size_t(kMaxChar) * 2 + size_t(m.finalTableSize) * 8 + size_t(m.statesCount) * 8 +
	                                   rowSize * m.statesCount * 4,
	                               8);
	if (len < pos + bufSize)
		return Bad("EOF reached while reading the scanner buffer");

	t.states = m.statesCount;
	t.letters = m.lettersCount;
	t.regexps = m.regexpsCount;
	t.rowStride = uint32_t(rowSize * 4);
	t.refBufSize = bufSize;
	t.blobBytes = AlignUp(pos + bufSize, 8);

	const uint8_t* buf = p + pos;
	const uint8_t* letters = buf;
	const uint8_t* finalTab = letters + size_t(kMaxChar) * 2;
	const uint8_t* finalIndex = finalTab + size_t(m.finalTableSize) * 8;
	const uint8_t* trans = finalIndex + size_t(m.statesCount) * 8;
	const uint64_t stride = t.rowStride;
	const uint64_t tableBytes = stride * t.states;

	if (m.initial % stride != 0 || m.initial >= tableBytes)
		return Bad("Corrupt scanner: initial state out of range");
	t.initial = uint32_t(m.initial / stride);

	t.cls.assign(kMaxChar, 0);
	for (uint32_t c = 0; c < kMaxChar; ++c) {
		uint16_t v;
		memcpy(&v, letters + size_t(c) * 2, 2);
		if (c == kEpsilon) {
			t.cls[c] = 0;   // never fed to Next(); the reference leaves this slot zero (Init, multi.h:364-365, 374-376)
			continue;
		}
		if (v < t.headerSize || v >= t.headerSize + t.letters)
			return Bad("Corrupt scanner: letter class out of range");
		t.cls[c] = uint16_t(v - t.headerSize);
	}

	t.next.resize(size_t(t.states) * t.letters);
	t.flags.resize(t.states);
	for (uint32_t s = 0; s < t.states; ++s) {
		const uint8_t* row = trans + size_t(s) * stride;
		uint64_t fl;
		memcpy(&fl, row + flagsOff, 8);
		bool absorbing = true;
		for (uint32_t c = 0; c < t.letters; ++c) {
			int32_t shift;
			memcpy(&shift, row + size_t(t.headerSize + c) * 4, 4);
			const int64_t dest = int64_t(s) * int64_t(stride) + shift;   // Relocatable::Go, multi.h:65
			if (dest < 0 || uint64_t(dest) >= tableBytes || uint64_t(dest) % stride != 0)
				return Bad("Corrupt scanner: transition out of range");
			const uint32_t d = uint32_t(uint64_t(dest) / stride);
			t.next[size_t(s) * t.letters + c] = d;
			absorbing = absorbing && d == s;
		}
		t.flags[s] = uint8_t((fl & 1 ? kFinal : 0) | (fl & 2 ? kDead : 0) | (absorbing ? kAbsorbing : 0));
	}

	t.acceptOff.assign(size_t(t.states) + 1, 0);
	for (uint32_t s = 0; s < t.states; ++s) {
		uint64_t fi;
		memcpy(&fi, finalIndex + size_t(s) * 8, 8);
		t.acceptOff[s] = t.acceptIds.size();
		for (;; ++fi) {
			if (fi >= m.finalTableSize)
				return Bad("Corrupt scanner: final table is not terminated");
			uint64_t id;
			memcpy(&id, finalTab + fi * 8, 8);
			if (id == ~uint64_t(0))
				break;   // End sentinel, multi.h:96, 155
			// the kernels index counters with these ids (cnt[2 + id], row[id]): an id the scanner does not have would
			// be an out-of-bounds device write at run time
			if (id >= m.regexpsCount)
				return Bad("Corrupt scanner: regexp id in the final table out of range");
			// lists of different states may overlap in a mutated image, which would grow the decoded form
			// quadratically: a well-formed scanner has finalTableSize entries in total (BuildFinals, multi.h:477-528)
			if (t.acceptIds.size() >= size_t(m.finalTableSize) + t.states)
				return Bad("Corrupt scanner: final lists overlap");
			t.acceptIds.push_back(id);
		}
	}
	t.acceptOff[t.states] = t.acceptIds.size();

	ChooseHotAndPermute(t);
	return PIRE_HIP_OK;
}

// ------------------------------------------------------------------------------------------ device image

namespace {

// The stream image uploads go through: null = plain hipMemcpy (the caller's thread waits for the device's legacy stream,
// as every first run of a table always did); the background adaptation's worker sets a NON-BLOCKING stream of its own, so that
// its copies neither wait for the caller's kernels nor hold them up (thread local: only that thread's uploads).
thread_local hipStream_t g_putStream = nullptr;

template <class T>
int Put(T** dst, const std::vector<T>& src, uint64_t* total)
{
	*dst = nullptr;
	const size_t bytes = std::max<size_t>(src.size() * sizeof(T), 16);
	hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), bytes);
	if (e != hipSuccess)
		return HipFail(e, "hipMalloc(table)");
	if (!src.empty()) {
		if (g_putStream) {
			e = hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, g_putStream);
			if (e == hipSuccess)
				e = hipStreamSynchronize(g_putStream);   // (`src` is often a temporary)
		} else {
			e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
		}
		if (e != hipSuccess)
			return HipFail(e, "hipMemcpy(table)");
	}
	*total += bytes;
	return PIRE_HIP_OK;
}

}  // namespace

void FreeDeviceTable(DeviceTable* d)
{
	if (d->device < 0)
		return;
	void* ptrs[] = {d->hotRows, d->hotRowsRot, d->hotFlags, d->cls, d->nextPerm, d->flagsPerm, d->origOfPerm,
	                d->permOfOrig, d->acceptMaskPerm, d->acceptOffPerm, d->acceptIds, d->visitHot, d->visitCold,
	                d->finSelf,    d->finEnd,  d->workCounter, d->compactRows, d->incPerm,
	                d->distFinalPerm, d->distFlaggedPerm, d->wideRows, d->next16, d->visitWide, d->wideRowsStream};
	for (void* q : ptrs)
		if (q)
			(void)hipFree(q);
	if (d->trapSignalHost)
		(void)hipHostFree(const_cast<uint32_t*>(d->trapSignalHost));
	*d = DeviceTable();
}

void FreeAllDeviceTables(pire_hip_table* t)
{
	int cur = -1;
	(void)hipGetDevice(&cur);
	for (DeviceTable& r : t->retired)
		if (r.device >= 0 && hipSetDevice(r.device) == hipSuccess)
			FreeDeviceTable(&r);
	t->retired.clear();
	for (int k = 0; k < kMaxDevices; ++k)
		if (t->devs[k].device >= 0) {
			(void)hipSetDevice(k);
			FreeDeviceTable(&t->devs[k]);
		}
	if (cur >= 0)
		(void)hipSetDevice(cur);
}

// The wide walk's LDS image (internal.h WideLayout) in the table's current numbering, as it lies in LDS from rowsOff on:
// rows -- entry = device id of the target, `wide` (the escape state) for targets outside the tier; then the row's flags --
// and, zipped image (internal.h MakeWideLayout), the headers of all tier states + the escape state and the exception targets
// of the zipped ones.
std::vector<uint16_t> BuildWideRows(const HostTable& h, uint32_t tier)
{
	const uint32_t W = tier ? tier : h.wide, C = h.letters, F = h.zipFull ? h.zipFull : W;
	const WideLayout wl = MakeWideLayout(W, C, 0, h.zipFull);
	const uint32_t pitch2 = wl.pitch / 2;
	std::vector<uint16_t> img((wl.imageEnd - wl.rowsOff) / 2 + 8, 0);
	auto target = [&](uint32_t o, uint32_t c) { return uint16_t(std::min(h.permOfOrig[h.next[size_t(o) * C + c]], W)); };
	for (uint32_t pid = 0; pid <= F; ++pid) {
		uint16_t* row = &img[size_t(pid) * pitch2];
		const uint32_t o = pid < F ? h.origOfPerm[pid] : 0;
		for (uint32_t c = 0; c < C; ++c)
			row[c] = pid < F ? target(o, c) : uint16_t(W);
		row[C] = pid < F ? h.flags[o] : 0;
	}
	if (!h.zipFull)
		return img;
	uint16_t* hdr = &img[(wl.hOff - wl.rowsOff) / 2];
	uint16_t* exc = &img[(wl.xOff - wl.rowsOff) / 2];
	const uint32_t none = (kZipNoLetter << 1) | (kZipNoLetter << 8) | (kZipNoLetter << 15);
	auto put = [&](uint32_t pid, uint32_t word) {
		hdr[2 * pid] = uint16_t(word);
		hdr[2 * pid + 1] = uint16_t(word >> 16);
	};
	// bit 0 of a header (no part of the letter fields: 2 * class * 0x4081 has it clear): the state is Final -- what the walks
	// with actions look for in a chunk (ragged.hip WideChunkAct); the plain walks never look at it
	auto flagged = [&](uint32_t pid) { return (h.flags[h.origOfPerm[pid]] & kFinal) ? 1u : 0u; };
	for (uint32_t pid = 0; pid < F; ++pid)
		put(pid, (pid << 22) | none | flagged(pid));
	put(W, (F << 22) | none);   // the escape state leans on the escape row
	for (uint32_t pid = F; pid < W; ++pid) {
		const uint32_t o = h.origOfPerm[pid], b = h.zipBase[pid - F], bo = h.origOfPerm[b];
		uint32_t word = (b << 22), k = 0;
		uint16_t* x = exc + size_t(pid - F) * kZipExceptions;
		for (uint32_t c = 0; c < C; ++c)
			if (h.next[size_t(o) * C + c] != h.next[size_t(bo) * C + c]) {
				if (k == kZipExceptions)
					throw std::logic_error("zipped image: a state with more exceptions than the plan allowed");
				word |= c << (1 + 7 * k);
				x[k++] = target(o, c);
			}
		for (; k < kZipExceptions; ++k) {
			word |= kZipNoLetter << (1 + 7 * k);
			x[k] = uint16_t(W);
		}
		put(pid, word | flagged(pid));
	}
	return img;
}

int UploadTable(pire_hip_table* t, DeviceTable* image)
{
	EnsureRanked(t);
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
	DeviceTable d;
	if (int rc = BuildDeviceImage(t->host, dev, &d))
		return rc;
	t->devs[dev] = d;
	*image = d;
	return PIRE_HIP_OK;
}

// The device image of a ranked host table on the CURRENT device (`dev`): allocations + copies, nothing else.
int BuildDeviceImage(const HostTable& h, int dev, DeviceTable* out)
{
	const uint32_t N = h.states, C = h.letters;
	std::vector<uint32_t> nextPerm(size_t(N) * C);
	std::vector<uint8_t> flagsPerm(N);
	for (uint32_t pid = 0; pid < N; ++pid) {
		const uint32_t o = h.origOfPerm[pid];
		for (uint32_t c = 0; c < C; ++c)
			nextPerm[size_t(pid) * C + c] = h.permOfOrig[h.next[size_t(o) * C + c]];
		flagsPerm[pid] = h.flags[o];
	}
	DeviceTable d;
	int rc;
	// the rows with rotated columns: read by the A/B variant tiled_variant = 23 alone, which is never chosen by the library
	// (no gain measured, tiled.hip LaunchTiled) -- built and uploaded only for images made while that variant is asked
	// for (every upload and every re-upload after an adaptation paid a host loop and 64 KB for it); a table
	// uploaded earlier runs variant 23 with the plain rows (tiled.hip falls back when the pointer is null)
	std::vector<uint8_t> rowsRot;
	if (GetConfig().tiled_variant == 23) {
		rowsRot.resize(h.hotRows.size());
		for (size_t r = 0; r < h.hotRows.size() / 256; ++r)
			for (uint32_t b = 0; b < 256; ++b)
				rowsRot[r * 256 + RotColumn(b)] = h.hotRows[r * 256 + b];
	}
	if ((rc = Put(&d.hotRows, h.hotRows, &d.bytes)) || (!rowsRot.empty() && (rc = Put(&d.hotRowsRot, rowsRot, &d.bytes))) ||
	    (rc = Put(&d.hotFlags, h.hotFlags, &d.bytes)) ||
	    (rc = Put(&d.cls, h.cls, &d.bytes)) || (rc = Put(&d.nextPerm, nextPerm, &d.bytes)) ||
	    (rc = Put(&d.flagsPerm, flagsPerm, &d.bytes)) || (rc = Put(&d.origOfPerm, h.origOfPerm, &d.bytes)) ||
	    (rc = Put(&d.permOfOrig, h.permOfOrig, &d.bytes))) {
		d.device = dev;
		FreeDeviceTable(&d);
		return rc;
	}
	{
		// CSR final lists in device numbering (always: HalfFinalScanner needs the multiplicities)
		std::vector<uint64_t> off(size_t(N) + 1, 0), ids;
		for (uint32_t pid = 0; pid < N; ++pid) {
			const uint32_t o = h.origOfPerm[pid];
			off[pid] = ids.size();
			ids.insert(ids.end(), h.acceptIds.begin() + h.acceptOff[o], h.acceptIds.begin() + h.acceptOff[o + 1]);
		}
		off[N] = ids.size();
		ids.push_back(0);   // never empty
		if (!(rc = Put(&d.acceptOffPerm, off, &d.bytes)))
			rc = Put(&d.acceptIds, ids, &d.bytes);
		if (!rc && h.incPacked) {
			std::vector<uint64_t> inc(N);
			for (uint32_t pid = 0; pid < N; ++pid)
				inc[pid] = h.inc64[h.origOfPerm[pid]];
			rc = Put(&d.incPerm, inc, &d.bytes);
		}
		if (rc) {
			d.device = dev;
			FreeDeviceTable(&d);
			return rc;
		}
	}
	if (h.regexps <= 64) {
		std::vector<uint64_t> mask(N, 0);
		for (uint32_t pid = 0; pid < N; ++pid) {
			const uint32_t o = h.origOfPerm[pid];
			for (uint64_t k = h.acceptOff[o]; k < h.acceptOff[o + 1]; ++k)
				if (h.acceptIds[k] < 64)
					mask[pid] |= uint64_t(1) << h.acceptIds[k];
		}
		rc = Put(&d.acceptMaskPerm, mask, &d.bytes);
	}
	if (!rc) {
		// end-of-string records (one 16-byte load per string instead of a chain of dependent lookups)
		std::vector<FinRec> self(N), end(N);
		const uint32_t endCls = h.cls[kEndMark];
		auto fill = [&](FinRec& r, uint32_t pid) {
			const uint32_t o = h.origOfPerm[pid];
			r.orig = o;
			r.permFlags = pid | (uint32_t(h.flags[o]) << 28);
			r.acceptMask = 0;
			if (h.regexps <= 64)
				for (uint64_t k = h.acceptOff[o]; k < h.acceptOff[o + 1]; ++k)
					if (h.acceptIds[k] < 64)
						r.acceptMask |= uint64_t(1) << h.acceptIds[k];
		};
		for (uint32_t pid = 0; pid < N; ++pid) {
			fill(self[pid], pid);
			fill(end[pid], nextPerm[size_t(pid) * C + endCls]);
		}
		rc = Put(&d.finSelf, self, &d.bytes);
		if (!rc)
			rc = Put(&d.finEnd, end, &d.bytes);
	}
	if (!rc) {
		// compact tier: entry = LDS address / 4 of the next state's row; targets without a row go to the escape row
		const uint32_t W = h.compact, pitch = CompactPitch(C);
		const uint32_t base = MakeLayout(h.hot, 0, 256u).compactOff;
		std::vector<uint16_t> rows(W ? (size_t(W + 1) * pitch / 2 + 15) / 8 * 8 : 8, 0);
		if (W) {
			auto addr4 = [&](uint32_t pid) { return uint16_t((base + std::min(pid, W) * pitch) / 4); };
			for (uint32_t pid = 0; pid <= W; ++pid) {
				uint16_t* row = &rows[size_t(pid) * pitch / 2];
				for (uint32_t c = 0; c < C; ++c)
					row[c] = pid < W ? addr4(nextPerm[size_t(pid) * C + c]) : addr4(W);
				row[C] = uint16_t(pid);
			}
		}
		rc = Put(&d.compactRows, rows, &d.bytes);
	}
	if (!rc && h.wide) {
		rc = Put(&d.wideRows, BuildWideRows(h), &d.bytes);
		if (!rc)
			rc = Put(&d.visitWide, std::vector<uint32_t>(h.wide + 1, 0), &d.bytes);
	}
	if (!rc && h.wide && N <= 65536) {
		// the exact table once more as u16: what the wide walk reads for states without a row (half the cache footprint)
		std::vector<uint16_t> n16(nextPerm.size());
		for (size_t i = 0; i < nextPerm.size(); ++i)
			n16[i] = uint16_t(nextPerm[i]);
		rc = Put(&d.next16, n16, &d.bytes);
		// ... and the image for the stream kernel (a smaller tier: the strings' positions share the LDS with it) -- for images made
		// while that kernel is asked for: it is opt-in (stream.hip StreamWideEligible), nobody else pays for the second image
		const uint32_t tier = GetConfig().ragged_variant == 2 ? StreamWideTier(C, h.regexps, h.wide, h.zipFull) : 0;
		if (!rc && tier > h.hot && tier > h.zipFull) {
			rc = Put(&d.wideRowsStream, BuildWideRows(h, tier), &d.bytes);
			d.wideStream = tier;
		}
	}
	if (!rc)
		rc = Put(&d.visitHot, std::vector<uint32_t>(kVisitHotSlots, 0), &d.bytes);
	if (!rc)
		rc = Put(&d.visitCold, std::vector<uint32_t>(N, 0), &d.bytes);
	if (!rc)
		rc = Put(&d.workCounter, std::vector<unsigned long long>(2 * kWorkSlots, 0), &d.bytes);   // pairs: internal.h
	if (!rc) {
		// the auto-adaptation signal: one word of mapped host memory; without it (no mapped memory on this platform)
		// the table simply never adapts by itself
		void* hostWord = nullptr;
		void* devWord = nullptr;
		if (hipHostMalloc(&hostWord, 64, hipHostMallocMapped) == hipSuccess) {
			memset(hostWord, 0, 64);
			if (hipHostGetDevicePointer(&devWord, hostWord, 0) == hipSuccess) {
				d.trapSignalHost = static_cast<volatile uint32_t*>(hostWord);
				d.trapSignalDev = static_cast<uint32_t*>(devWord);
			} else {
				(void)hipHostFree(hostWord);
			}
		}
		(void)hipGetLastError();
	}
	if (rc) {
		d.device = dev;
		FreeDeviceTable(&d);
		return rc;
	}
	d.device = dev;
	*out = d;
	return PIRE_HIP_OK;
}

// Re-rank the dense LDS rows from what the kernels actually visited (see pire_hip_table_adapt in pire_hip.h).
// Scanner::Glue (multi.h:1092-1103) on two ingested tables: the product automaton, numbered exactly as the reference
// numbers it -- ScannerGlueCommon/LettersEquality (glue.h:35-159) for the letter classes, Impl::Determine
// (determine.h:91-137) for the breadth-first state numbering, ScannerGlueTask::AcceptStates (multi.h:1024-1043) for
// flags and final lists.  `out` is an empty scanner when more than maxSize new states are needed (Failure()).
// breadth-first product construction (determine.h:100-122), sequential host version
int GlueBfsHost(const HostTable& a, const HostTable& b, const std::vector<uint32_t>& la, const std::vector<uint32_t>& lb,
                size_t maxSize, GlueProduct* out)
{
	const uint32_t LC = uint32_t(la.size());
	std::vector<std::pair<uint32_t, uint32_t>>& states = out->states;
	std::vector<uint32_t>& next = out->next;
	std::unordered_map<uint64_t, uint32_t> index;
	auto keyOf = [](uint32_t x, uint32_t y) { return (uint64_t(x) << 32) | y; };
	states.clear();
	next.clear();
	states.emplace_back(a.initial, b.initial);
	index.emplace(keyOf(a.initial, b.initial), 0u);
	out->failed = false;
	for (size_t i = 0; i < states.size() && !out->failed; ++i) {
		const uint32_t sa = states[i].first, sb = states[i].second;
		next.resize((i + 1) * size_t(LC));
		for (uint32_t l = 0; l < LC; ++l) {
			const uint32_t na = a.next[size_t(sa) * a.letters + la[l]], nb = b.next[size_t(sb) * b.letters + lb[l]];
			auto ins = index.emplace(keyOf(na, nb), uint32_t(states.size()));
			if (ins.second) {
				if (!maxSize--) {   // determine.h:112-113
					out->failed = true;
					break;
				}
				states.emplace_back(na, nb);
			}
			next[i * size_t(LC) + l] = ins.first->second;
		}
	}
	return PIRE_HIP_OK;
}

int GlueHostTables(const HostTable& a, const HostTable& b, size_t maxSize, HostTable* out, bool onDevice)
{
	if (a.scannerType != 1 || b.scannerType != 1)
		return Bad("Glue is defined for Pire::Scanner tables");
	if (a.empty) {   // multi.h:1094-1097
		*out = b;
		return PIRE_HIP_OK;
	}
	if (b.empty) {
		*out = a;
		return PIRE_HIP_OK;
	}
	if (a.headerSize != b.headerSize)
		return Bad("Glue: the scanners have different shortcutting types");
	if (maxSize == 0)
		maxSize = 80000;   // DefMaxSize, multi.h:1099

	HostTable t;
	t.scannerType = 1;
	t.headerSize = a.headerSize;
	t.regexps = a.regexps + b.regexps;
	t.initial = 0;   // Init(..., size_t(0), ...), multi.h:1031

	// letter classes: Partition<Char, LettersEquality> filled with 0..MaxChar-1 except Epsilon, in that order
	// (glue.h:123-127): a class is numbered when its first (= smallest, = representative) Char arrives, and the
	// partition iterates in representative order (partition.h:40, 186-204), which is the same order
	std::map<std::pair<uint16_t, uint16_t>, uint16_t> classOf;
	std::vector<uint32_t> rep;
	t.cls.assign(kMaxChar, 0);
	for (uint32_t c = 0; c < kMaxChar; ++c) {
		if (c == kEpsilon)
			continue;
		const std::pair<uint16_t, uint16_t> key(a.cls[c], b.cls[c]);
		auto it = classOf.find(key);
		if (it == classOf.end()) {
			it = classOf.emplace(key, uint16_t(rep.size())).first;
			rep.push_back(c);
		}
		t.cls[c] = it->second;
	}
	const uint32_t LC = uint32_t(rep.size());
	t.letters = LC;

	std::vector<uint32_t> la(LC), lb(LC);
	for (uint32_t l = 0; l < LC; ++l) {
		la[l] = a.cls[rep[l]];
		lb[l] = b.cls[rep[l]];
	}
	GlueProduct prod;
	if (int rc = onDevice ? GlueBfsDevice(a, b, la, lb, maxSize, &prod) : GlueBfsHost(a, b, la, lb, maxSize, &prod))
		return rc;
	const bool failed = prod.failed;
	std::vector<std::pair<uint32_t, uint32_t>>& states = prod.states;
	t.next.swap(prod.next);
	std::vector<uint32_t>& next = t.next;
	if (failed) {
		// task.Failure() = Scanner(): the empty scanner (glue.h:146, multi.h:121)
		HostTable e;
		e.scannerType = 1;
		e.headerSize = a.headerSize;
		e.empty = true;
		e.states = 1;
		e.letters = 1;
		e.regexps = 0;
		e.initial = 0;
		e.rowStride = uint32_t(AlignUp(1 + e.headerSize, 4) * 4);
		e.cls.assign(kMaxChar, 0);
		e.next.assign(1, 0);
		e.flags.assign(1, uint8_t(kDead | kAbsorbing));
		e.acceptOff.assign(2, 0);
		ChooseHotAndPermute(e);
		*out = e;
		return PIRE_HIP_OK;
	}
	const uint32_t N = uint32_t(states.size());
	t.states = N;
	const size_t rowSize = AlignUp(size_t(LC) + t.headerSize, 16 / 4);   // RowSize(), multi.h:347
	t.rowStride = uint32_t(rowSize * 4);

	// AcceptStates, multi.h:1024-1043: lhs ids, then rhs ids shifted by lhs.RegexpsCount(); Final = either, Dead = both
	t.flags.resize(N);
	t.acceptOff.assign(size_t(N) + 1, 0);
	for (uint32_t i = 0; i < N; ++i) {
		const uint32_t sa = states[i].first, sb = states[i].second;
		t.acceptOff[i] = t.acceptIds.size();
		for (uint64_t k = a.acceptOff[sa]; k < a.acceptOff[sa + 1]; ++k)
			t.acceptIds.push_back(a.acceptIds[k]);
		for (uint64_t k = b.acceptOff[sb]; k < b.acceptOff[sb + 1]; ++k)
			t.acceptIds.push_back(b.acceptIds[k] + a.regexps);
		bool absorbing = true;
		for (uint32_t l = 0; l < LC; ++l)
			absorbing = absorbing && next[size_t(i) * LC + l] == i;
		const bool fin = (a.flags[sa] & kFinal) || (b.flags[sb] & kFinal);
		const bool dead = (a.flags[sa] & kDead) && (b.flags[sb] & kDead);
		t.flags[i] = uint8_t((fin ? kFinal : 0) | (dead ? kDead : 0) | (absorbing ? kAbsorbing : 0));
	}
	t.acceptOff[N] = t.acceptIds.size();
	// BufSize(), multi.h:297-305, with finalTableSize = ids + one sentinel per state (multi.h:362)
	t.refBufSize = AlignUp(size_t(kMaxChar) * 2 + (t.acceptIds.size() + N) * 8 + size_t(N) * 8 + rowSize * N * 4, 8);
	t.blobBytes = 0;
	t.ranked = false;   // the dense-row ranking (a Markov walk over the table) is only needed once the table is run
	                    // or inspected, not for the intermediates of a left-to-right glue
	*out = std::move(t);
	return PIRE_HIP_OK;
}

// Distance, in text bytes, from every state to the nearest Final (Final or Dead) state, capped: round r marks the
// states that reach a marked state of round r-1 on some byte.  Only distances up to one 16-byte chunk matter.
static std::vector<uint8_t> DistanceTo(const HostTable& h, uint8_t flagMask)
{
	const uint32_t N = h.states, C = h.letters;
	std::vector<uint8_t> isByteLetter(C, 0);
	for (uint32_t b = 0; b < 256; ++b)
		isByteLetter[h.cls[b]] = 1;
	std::vector<uint32_t> letters;
	for (uint32_t c = 0; c < C; ++c)
		if (isByteLetter[c])
			letters.push_back(c);
	std::vector<uint8_t> dist(N, 255);
	std::vector<uint32_t> open;
	for (uint32_t s = 0; s < N; ++s) {
		if (h.flags[s] & flagMask)
			dist[s] = 0;
		else
			open.push_back(s);
	}
	for (uint32_t r = 1; r <= 17 && !open.empty(); ++r) {
		std::vector<uint32_t> still, reached;
		for (uint32_t s : open) {
			bool hit = false;
			const uint32_t* row = &h.next[size_t(s) * C];
			for (uint32_t c : letters)
				if (dist[row[c]] == r - 1) {
					hit = true;
					break;
				}
			(hit ? reached : still).push_back(s);
		}
		for (uint32_t s : reached)
			dist[s] = uint8_t(r);
		open.swap(still);
	}
	return dist;
}

int EnsureActDist(pire_hip_table* t, const uint8_t** distFinalPerm, const uint8_t** distFlaggedPerm)
{
	std::lock_guard<std::mutex> lock(t->uploadMutex);
	HostTable& h = t->host;
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess)
		return HipFail(e, "hipGetDevice");
	if (dev < 0 || dev >= kMaxDevices || t->devs[dev].device != dev) {
		SetError("EnsureActDist before UploadTable");
		return PIRE_HIP_EINVAL;
	}
	DeviceTable& d = t->devs[dev];
	if (h.distFinal.empty()) {
		h.distFinal = DistanceTo(h, kFinal);
		h.distFlagged = DistanceTo(h, kFinal | kDead);
	}
	const uint32_t N = h.states;
	// the two arrays independently: a failed second upload must not leak or re-upload the first
	if (!d.distFinalPerm) {
		std::vector<uint8_t> a(N);
		for (uint32_t pid = 0; pid < N; ++pid)
			a[pid] = h.distFinal[h.origOfPerm[pid]];
		if (int rc = Put(&d.distFinalPerm, a, &d.bytes))
			return rc;
	}
	if (!d.distFlaggedPerm) {
		std::vector<uint8_t> b(N);
		for (uint32_t pid = 0; pid < N; ++pid)
			b[pid] = h.distFlagged[h.origOfPerm[pid]];
		if (int rc = Put(&d.distFlaggedPerm, b, &d.bytes))
			return rc;
	}
	*distFinalPerm = d.distFinalPerm;
	*distFlaggedPerm = d.distFlaggedPerm;
	return PIRE_HIP_OK;
}

void EnsureRanked(pire_hip_table* t)
{
	std::lock_guard<std::mutex> lock(t->uploadMutex);
	if (!t->host.ranked) {
		ChooseHotAndPermute(t->host);
		t->host.ranked = true;
	}
}

// Failures the checked kernel build counted on any device since the last call (and clears the counters).
int CheckFailures(pire_hip_table* t, uint64_t* out)
{
	*out = 0;
	std::lock_guard<std::mutex> lock(t->uploadMutex);
	int cur = -1;
	hipError_t e = hipGetDevice(&cur);
	if (e != hipSuccess)
		return HipFail(e, "hipGetDevice");
	for (int k = 0; k < kMaxDevices; ++k) {
		if (t->devs[k].device < 0)
			continue;
		uint32_t v = 0;
		if ((e = hipSetDevice(k)) == hipSuccess && (e = hipDeviceSynchronize()) == hipSuccess &&
		    (e = hipMemcpy(&v, t->devs[k].visitHot + kCheckSlot, 4, hipMemcpyDeviceToHost)) == hipSuccess)
			e = hipMemset(t->devs[k].visitHot + kCheckSlot, 0, 4);
		if (e != hipSuccess) {
			(void)hipSetDevice(cur);
			return HipFail(e, "check counters");
		}
		*out += v;
	}
	(void)hipSetDevice(cur);
	return PIRE_HIP_OK;
}

namespace {

// What the scans on a table left in the visit counters of its device images, summed.
struct SeenCounters {
	std::vector<uint64_t> hot, cold, wide;
	uint64_t wideTrapChunks = 0;       // wave-chunks the wide walk walked twice (exact)
	uint64_t wideOutsideSamples = 0;   // of the wide walk's visit samples (one lane per wave and tile): lanes found outside the tier
	bool any = false;
};

// Adds image `d`'s counters (current device) to `seen`.  stream == nullptr: plain copies behind a drained device (the caller
// synchronised); else asynchronous copies on that (non-blocking) stream, waited for here -- the kernels of other streams go
// on counting meanwhile: the counters are samples, a torn total is a sample too.
int ReadCounters(const DeviceTable& d, uint32_t N, uint32_t wideStates, hipStream_t stream, SeenCounters* seen)
{
	std::vector<uint32_t> bufHot(kVisitHotSlots), bufCold(N), bufWide(wideStates + 1, 0);
	hipError_t e;
	if (stream) {
		e = hipMemcpyAsync(bufHot.data(), d.visitHot, kVisitHotSlots * 4, hipMemcpyDeviceToHost, stream);
		if (e == hipSuccess)
			e = hipMemcpyAsync(bufCold.data(), d.visitCold, size_t(N) * 4, hipMemcpyDeviceToHost, stream);
		if (e == hipSuccess && d.visitWide && wideStates)
			e = hipMemcpyAsync(bufWide.data(), d.visitWide, size_t(wideStates + 1) * 4, hipMemcpyDeviceToHost, stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(stream);
	} else {
		e = hipMemcpy(bufHot.data(), d.visitHot, kVisitHotSlots * 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess)
			e = hipMemcpy(bufCold.data(), d.visitCold, size_t(N) * 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess && d.visitWide && wideStates)
			e = hipMemcpy(bufWide.data(), d.visitWide, size_t(wideStates + 1) * 4, hipMemcpyDeviceToHost);
	}
	if (e != hipSuccess)
		return HipFail(e, "visit counters");
	if (seen->hot.empty()) {
		seen->hot.assign(256, 0);
		seen->cold.assign(N, 0);
		seen->wide.assign(wideStates + 1, 0);
	}
	seen->any = true;
	for (uint32_t i = 0; i < 256; ++i)
		seen->hot[i] += bufHot[i];
	seen->wideTrapChunks += bufHot[kWideTrapSlot];
	for (uint32_t i = 0; i < N; ++i)
		seen->cold[i] += bufCold[i];
	if (d.visitWide) {
		for (uint32_t i = 0; i < wideStates; ++i)
			seen->wide[i] += bufWide[i];
		seen->wideOutsideSamples += bufWide[wideStates];
	}
	return PIRE_HIP_OK;
}

// The ranking's new scores from the counters (h keeps its numbering; h.seenMass and the statistics are updated).
// Returns false when nothing left the rows and nothing else asks for a new numbering: the current one covers the traffic.
struct Scored {
	std::vector<double> score;
	bool wideSeen = false;
	float wideSeenOutside = 0;
	bool rerank = false;
};

Scored ScoreFromCounters(HostTable& h, const SeenCounters& seen, uint64_t wideLaunched)
{
	const uint32_t N = h.states, H = h.hot;
	// hot ids are sampled once per wave per 128-byte tile (1 of 64*128 lane-steps), cold ids once per trapped
	// 16-byte chunk for one rotating lane of 64 (1 of 64*16 lane-steps): bring both to "lane-steps".
	// The estimates are remembered from one adapt() to the next (halved each time): the counters are samples, a state
	// that carries 1e-5 of the steps often has none in a given batch, and a ranking from the latest counters alone
	// dropped such rows at every other call only to see them trap again (URL batches: 35-43 rows changed at EVERY
	// adapt(), trap re-walks 19 % of the kernel time; profiles/r02_ragged_ablation.log).
	Scored out;
	out.score.resize(N);
	uint64_t coldSamples = 0;
	if (h.seenMass.size() != N)
		h.seenMass.assign(N, 0.0);
	for (uint32_t pid = 0; pid < N; ++pid) {
		const uint32_t o = h.origOfPerm[pid];
		double est = double(seen.cold[pid]) * 1024.0;
		if (pid < H)
			est += double(seen.hot[pid]) * 8192.0;
		if (pid < h.wide && pid < seen.wide.size())
			est += double(seen.wide[pid]) * 8192.0;   // the wide walk samples like the tiled kernel: one lane per wave per tile
		if (pid >= H)
			coldSamples += seen.cold[pid];
		h.seenMass[o] = 0.5 * h.seenMass[o] + est;
		out.score[o] = h.seenMass[o] + h.priorMass[o];   // prior (<= 1) only orders states nobody has visited yet
	}
	h.lastTrapSamples = coldSamples;
	h.lastWideTrapChunks = seen.wideTrapChunks;
	if (wideLaunched)   // (no wide launch since the last adapt(): the share stays what it was)
		h.wideTwiceShare = float(std::min(1.0, double(seen.wideTrapChunks) / double(wideLaunched)));
	h.massMeasured = true;
	// The share of the steps outside the wide rows, as the scans since the last adapt() SAW it: of the wide walk's visit
	// samples (one lane per wave and tile, whatever state it is in) those that found their lane outside the rows.  What
	// the masses give instead is the share the new ranking would leave outside -- of the states that were sampled:
	// on long-tailed tables most states outside the rows never are, and the figure is an order of magnitude too low.
	double wideSamples = double(seen.wideOutsideSamples);
	for (uint32_t i = 0; i < h.wide && i < seen.wide.size(); ++i)
		wideSamples += double(seen.wide[i]);
	out.wideSeen = wideSamples >= 4096.0;
	out.wideSeenOutside = out.wideSeen ? float(double(seen.wideOutsideSamples) / wideSamples) : 0.0f;
	// (a zipped table whose traffic no longer leaves the tier: would the plain rows hold it too?  They are the faster walk.)
	bool unzip = false;
	if (coldSamples == 0 && h.zipFull && GetConfig().zip_variant != 2) {
		std::vector<double> sorted(out.score);
		std::sort(sorted.begin(), sorted.end(), std::greater<double>());
		const uint32_t plain = WideCapacity(h.letters, h.regexps, N);
		double total = 0, inside = 0;
		for (uint32_t i = 0; i < N; ++i) {
			const double v = std::max(0.0, sorted[i]);
			total += v;
			if (i < plain)
				inside += v;
		}
		unzip = total > 0 && 1.0 - inside / total < (h.offsetBatchShare > 0.5f ? 0.05 : 0.002);   // (ChooseZip's own thresholds)
	}
	out.rerank = coldSamples != 0 || unzip;
	return out;
}

}  // namespace

// ---- adaptation in the background (round 6) -----------------------------------------------------------------------------
// A caller that only enqueues (PIRE_HIP_RUN_ON_DEVICE) must never wait for the device inside a call -- and until round 6 its
// table therefore never adapted unless it knew to call pire_hip_table_adapt() (0.3-0.7 TB/s for ever on tables the
// a-priori ranking knows little about).  Now such a call, at its launch boundary, may START an adaptation: a worker thread
// copies the visit counters on a non-blocking stream of its own (the caller's kernels go on), re-ranks a COPY of the host
// table, uploads the new image on that stream and reports "ready"; a later launch boundary SWAPS table and image in (a few
// pointer moves under the table's lock; the replaced images stay alive for launches still in flight and for captured graphs,
// like those of every automatic adaptation).  No call waits for the device, none makes a call that is illegal during stream
// capture.  pire_hip_table_adapt(), the synchronous automatic adaptation and pire_hip_table_destroy() first wait for a
// worker that is on its way (and drop what it prepared).
namespace {

void BackgroundWorker(pire_hip_table* t, int dev)
{
	pire_hip_table::Background& bg = t->bg;
	auto fail = [&]() { bg.state.store(0, std::memory_order_release); };
	if (hipSetDevice(dev) != hipSuccess)
		return fail();
	hipStream_t side = nullptr;
	if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess)
		return fail();
	std::unique_ptr<HostTable> h;
	DeviceTable cur;
	uint64_t launched = 0;
	{
		// the table as it is now (an adaptation of another kind would have waited for this thread before it took the lock)
		std::shared_lock<std::shared_mutex> stable(t->adaptMutex);
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		cur = t->devs[dev];
		if (cur.device == dev)
			h.reset(new HostTable(t->host));
		launched = t->wideLaunched.exchange(0, std::memory_order_relaxed);
		const uint64_t all = t->bytesNominal.load(std::memory_order_relaxed), off = t->bytesOffsetBatches.load(std::memory_order_relaxed);
		if (h && all)
			h->offsetBatchShare = float(std::min(1.0, double(off) / double(all)));
	}
	bool ok = h != nullptr;
	SeenCounters seen;
	if (ok)
		ok = ReadCounters(cur, h->states, h->wide, side, &seen) == PIRE_HIP_OK;
	DeviceTable image;
	if (ok) {
		try {
			Scored sc = ScoreFromCounters(*h, seen, launched);
			if (sc.rerank) {
				PermuteByScore(*h, sc.score);
				if (sc.wideSeen)
					h->outsideWide = sc.wideSeenOutside;
				h->adaptations++;
				g_putStream = side;
				ok = BuildDeviceImage(*h, dev, &image) == PIRE_HIP_OK;
				g_putStream = nullptr;
			} else {
				ok = false;   // nothing to do: the rows cover the traffic
				std::lock_guard<std::mutex> lock(t->uploadMutex);
				if (t->devs[dev].device == dev && t->devs[dev].trapSignalHost)
					bg.trapsAtLastLook = *t->devs[dev].trapSignalHost;
			}
		} catch (...) {
			g_putStream = nullptr;
			ok = false;
		}
	}
	(void)hipStreamDestroy(side);
	(void)hipGetLastError();
	if (!ok) {
		if (image.device >= 0 || image.hotRows)
			FreeDeviceTable(&image);
		return fail();
	}
	bg.host = std::move(h);
	bg.image = image;
	bg.device = dev;
	bg.state.store(2, std::memory_order_release);
}

}  // namespace

// Waits for a background adaptation that is on its way and drops what it prepared (the caller is about to rank the table itself,
// or to destroy it).
void JoinBackgroundAdapt(pire_hip_table* t)
{
	pire_hip_table::Background& bg = t->bg;
	std::lock_guard<std::mutex> one(bg.mutex);
	if (bg.thread.joinable())
		bg.thread.join();
	if (bg.state.load(std::memory_order_acquire) == 2) {
		int cur = -1;
		(void)hipGetDevice(&cur);
		if (bg.image.device >= 0 && hipSetDevice(bg.image.device) == hipSuccess)
			FreeDeviceTable(&bg.image);
		if (cur >= 0)
			(void)hipSetDevice(cur);
		bg.host.reset();
		bg.image = DeviceTable();
	}
	bg.state.store(0, std::memory_order_release);
}

namespace {

// Whether the scans since the last ranking ask for a new one: `threshold` sampled traps -- or, for a table nobody has ranked from
// measurements yet, a first look after 32 MB of text once a few dozen traps say the a-priori ranking is off: such a table may have
// taken the class-indexed walk on the strength of those few traps (api.cpp FillParams), traps no more, and would otherwise keep
// the a-priori rows -- and the slower walk -- for ever.
bool AdaptationDue(pire_hip_table* t, uint64_t traps, uint64_t threshold)
{
	if (traps >= threshold)
		return true;
	if (traps < 64 || t->bytesNominal.load(std::memory_order_relaxed) < (uint64_t(32) << 20))
		return false;
	std::shared_lock<std::shared_mutex> stable(t->adaptMutex);
	return !t->host.massMeasured;
}

// A launch boundary of an enqueue-only call: swap in what a finished worker prepared; start a worker when the scans since the
// last ranking left the rows often enough.  Never waits for the device.
void BackgroundAdaptStep(pire_hip_table* t, uint64_t threshold)
{
	pire_hip_table::Background& bg = t->bg;
	const int state = bg.state.load(std::memory_order_acquire);
	if (state == 1)
		return;   // on its way
	if (state == 2) {
		std::unique_lock<std::mutex> one(bg.mutex, std::try_to_lock);
		if (!one.owns_lock() || bg.state.load(std::memory_order_acquire) != 2)
			return;
		if (bg.thread.joinable())
			bg.thread.join();   // (it has stored "ready": it is past its last instruction that matters)
		std::unique_lock<std::shared_mutex> exclusive(t->adaptMutex);   // entry points between "copied the pointers" and "enqueued" finish first
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		// every image holds the old numbering: they stay alive (launches in flight, captured graphs) until the table is destroyed
		for (int k = 0; k < kMaxDevices; ++k)
			if (t->devs[k].device >= 0) {
				t->retired.push_back(t->devs[k]);
				t->devs[k] = DeviceTable();
			}
		t->host = std::move(*bg.host);
		bg.host.reset();
		t->devs[bg.device] = bg.image;
		bg.image = DeviceTable();
		bg.trapsAtLastLook = 0;
		t->bytesScanned.store(0, std::memory_order_relaxed);
		t->bytesNominal.store(0, std::memory_order_relaxed);
		t->bytesOffsetBatches.store(0, std::memory_order_relaxed);
		t->autoAdapts.fetch_add(1, std::memory_order_relaxed);
		bg.swaps.fetch_add(1, std::memory_order_relaxed);
		bg.state.store(0, std::memory_order_release);
		return;
	}
	if (t->autoAdapts.load(std::memory_order_relaxed) >= kMaxAutoAdapts)
		return;
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
		return;
	uint64_t traps = 0;
	{
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		if (t->devs[dev].device != dev || !t->devs[dev].trapSignalHost)
			return;
		traps = *t->devs[dev].trapSignalHost;
	}
	if (!AdaptationDue(t, traps, threshold) || (bg.trapsAtLastLook && traps < bg.trapsAtLastLook + threshold))
		return;
	std::unique_lock<std::mutex> one(bg.mutex, std::try_to_lock);
	if (!one.owns_lock() || bg.state.load(std::memory_order_acquire) != 0)
		return;
	if (bg.thread.joinable())
		bg.thread.join();
	bg.state.store(1, std::memory_order_release);
	try {
		bg.thread = std::thread(BackgroundWorker, t, dev);
	} catch (...) {
		bg.state.store(0, std::memory_order_release);
	}
}

}  // namespace

// The policy behind pire_hip_config.auto_adapt.  Every launch boundary looks at the trap totals the images' blocks have
// stored into mapped host memory (no synchronisation, no transfer: a read of a few host words); once the scans since the
// last ranking left the dense rows more than `auto_adapt_min_traps` sampled times (default 256 samples ~ 256 Ki
// lane-steps re-walked) the table is re-ranked.  In a call that synchronises anyway -- the host-pointer forms, and
// pire_hip_run with PIRE_HIP_RUN_HOST_OFFSETS -- right there: the device is drained (hipDeviceSynchronize), the counters
// read, the rows re-ranked, the images replaced, and the launch that noticed goes on with the new image.  In a call that only
// enqueues work (PIRE_HIP_RUN_ON_DEVICE: legal during stream capture, no stall of other streams) IN THE BACKGROUND (round 6,
// above): such a call never waits; it finds the new image a few launch boundaries later.  auto_adapt = 1: never; 2: every
// launch boundary drains and re-ranks (round 3's form); 3: round 5's default (never inside a call that only enqueues).
// A table adapts itself at most kMaxAutoAdapts times: the remembered estimates make the ranking converge within two or three
// (DESIGN.md 3.1), and a workload whose traffic does not fit the rows must not pay a re-ranking per call.
void MaybeAutoAdapt(pire_hip_table* t, bool enqueueOnly)
{
	pire_hip_config cfg = GetConfig();
	{
		std::shared_lock<std::shared_mutex> stable(t->adaptMutex);
		if (t->hasConfig && !g_cfgOverride)
			cfg = t->config;   // (pire_hip_table_config_set: the policy is the table's own)
	}
	if (cfg.auto_adapt == 1)
		return;
	const uint64_t threshold = cfg.auto_adapt_min_traps ? cfg.auto_adapt_min_traps : 256;
	if (enqueueOnly && cfg.auto_adapt != 2) {
		if (cfg.auto_adapt == 0)
			BackgroundAdaptStep(t, threshold);
		return;
	}
	if (t->autoAdapts.load(std::memory_order_relaxed) >= kMaxAutoAdapts)
		return;
	uint64_t traps = 0;
	{
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		for (int k = 0; k < kMaxDevices; ++k)
			if (t->devs[k].device >= 0 && t->devs[k].trapSignalHost)
				traps += *t->devs[k].trapSignalHost;
	}
	if (!AdaptationDue(t, traps, threshold))
		return;
	(void)AdaptTable(t, nullptr, true);   // a performance measure: a failure here surfaces in the launch that follows
}

int AdaptTable(pire_hip_table* t, uint32_t* changedRows, bool automatic)
{
	if (changedRows)
		*changedRows = 0;
	JoinBackgroundAdapt(t);   // (before the lock: the worker takes it shared)
	std::unique_lock<std::shared_mutex> exclusive(t->adaptMutex);
	if (automatic) {
		// re-check under the lock: another thread may just have done it
		uint64_t traps = 0;
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		for (int k = 0; k < kMaxDevices; ++k)
			if (t->devs[k].device >= 0 && t->devs[k].trapSignalHost)
				traps += *t->devs[k].trapSignalHost;
		if (traps == 0 || t->autoAdapts.load(std::memory_order_relaxed) >= kMaxAutoAdapts)
			return PIRE_HIP_OK;
		t->autoAdapts.fetch_add(1, std::memory_order_relaxed);
	}
	HostTable& h = t->host;
	const uint32_t H = h.hot;
	// what every device that ever ran this table saw, summed
	SeenCounters seen;
	{
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		int cur = -1;
		hipError_t e = hipGetDevice(&cur);
		if (e != hipSuccess)
			return HipFail(e, "hipGetDevice");
		for (int k = 0; k < kMaxDevices; ++k) {
			if (t->devs[k].device < 0)
				continue;
			int rc = PIRE_HIP_OK;
			if ((e = hipSetDevice(k)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess)
				rc = HipFail(e, "visit counters");
			else
				rc = ReadCounters(t->devs[k], h.states, h.wide, nullptr, &seen);
			if (rc != PIRE_HIP_OK) {
				(void)hipSetDevice(cur);
				return rc;
			}
		}
		(void)hipSetDevice(cur);
		if (!seen.any)
			return PIRE_HIP_OK;   // never ran: nothing observed
	}
	{
		t->bytesScanned.store(0, std::memory_order_relaxed);
		const uint64_t all = t->bytesNominal.exchange(0, std::memory_order_relaxed), off = t->bytesOffsetBatches.exchange(0, std::memory_order_relaxed);
		if (all)
			h.offsetBatchShare = float(std::min(1.0, double(off) / double(all)));
	}
	Scored sc = ScoreFromCounters(h, seen, t->wideLaunched.exchange(0, std::memory_order_relaxed));
	std::vector<uint32_t> before(h.origOfPerm.begin(), h.origOfPerm.begin() + H);
	std::sort(before.begin(), before.end());
	if (!sc.rerank) {
		MeasureShares(h, sc.score);   // same numbering, measured masses
		if (sc.wideSeen)
			h.outsideWide = sc.wideSeenOutside;
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		int cur = -1;
		(void)hipGetDevice(&cur);
		for (int k = 0; k < kMaxDevices; ++k)
			if (t->devs[k].device >= 0 && hipSetDevice(k) == hipSuccess) {
				(void)hipMemset(t->devs[k].visitHot, 0, 256 * 4);          // not the checked build's slot
				(void)hipMemset(t->devs[k].visitHot + kTrapSlot, 0, 4);
				(void)hipMemset(t->devs[k].visitHot + kWideTrapSlot, 0, 4);
				if (t->devs[k].visitWide)
					(void)hipMemset(t->devs[k].visitWide, 0, size_t(h.wide + 1) * 4);
				if (t->devs[k].trapSignalHost)
					*t->devs[k].trapSignalHost = 0;
			}
		(void)hipSetDevice(cur);
		return PIRE_HIP_OK;   // nothing trapped: the current rows already cover the traffic
	}
	PermuteByScore(h, sc.score);
	if (sc.wideSeen)
		h.outsideWide = sc.wideSeenOutside;
	std::vector<uint32_t> after(h.origOfPerm.begin(), h.origOfPerm.begin() + h.hot);
	std::sort(after.begin(), after.end());
	std::vector<uint32_t> diff;
	std::set_difference(after.begin(), after.end(), before.begin(), before.end(), std::back_inserter(diff));
	if (changedRows)
		*changedRows = uint32_t(diff.size());
	h.adaptations++;
	// every image holds the old numbering: drop them all (synchronised above), each device re-uploads on its next run
	{
		std::lock_guard<std::mutex> lock(t->uploadMutex);
		if (automatic) {
			// An AUTOMATIC adaptation keeps the replaced images until the table is destroyed: a graph captured
			// from enqueue-only calls has their pointers baked in, and its owner never asked for a re-ranking -- a replay
			// then walks the old image, which is as exact as the new one.  Bounded: kMaxAutoAdapts images per device.
			// pire_hip_table_adapt() is the caller's own act: it frees them (include/pire_hip.h says re-capture).
			for (int k = 0; k < kMaxDevices; ++k)
				if (t->devs[k].device >= 0) {
					t->retired.push_back(t->devs[k]);
					t->devs[k] = DeviceTable();
				}
		} else {
			FreeAllDeviceTables(t);   // drained above; every entry point that could hold their pointers has returned (TableUse)
		}
	}
	DeviceTable d;
	return UploadTable(t, &d);
}

}  // namespace pirehip
