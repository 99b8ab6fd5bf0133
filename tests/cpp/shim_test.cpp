// Drop-in check of include/pire_hip/batch_runner.hpp against the UNMODIFIED reference, in the reference's own
// vocabulary: for every (pattern set, string) the GPU BatchRunner must return exactly the Scanner::State that
//     Pire::Runner(sc).Begin().Run(str).End().State()            (run.h:365-392; call shape of tests/common.h:158-169)
// returns, so sc.Final(st) and sc.AcceptedRegexps(st) agree as well.  Patterns/strings are those of the reference's
// tests (tests/pire_ut.cpp: String, Boundaries, Primitives, Repetition, TestShortcuts, Glue, Aligned, EmptyScanner).
// Built only where /root/reference exists (tests/cpp/Makefile); the binary ships to the GPU box.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <pire/pire.h>
#include <pire/extra.h>
#include <pire_hip/batch_runner.hpp>

static int g_checks = 0, g_fail = 0;
#define CHECK(cond) do { ++g_checks; if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

static Pire::Fsm Parse(const char* re, bool surround = true)
{
	Pire::Fsm fsm = Pire::Lexer(re, re + strlen(re)).Parse();
	if (surround)
		fsm.Surround();
	return fsm;
}

template <class Scanner>
static void CompareAll(const Scanner& sc, const std::vector<Pire::ystring>& strings)
{
	Pire::Hip::BatchRunner<Scanner> gpu(sc);
	const std::vector<typename Scanner::State>& st = gpu.Begin().Run(strings).End().States();
	const std::vector<char>& fin = gpu.Finals();
	CHECK(st.size() == strings.size());
	uint64_t finals = 0;
	for (size_t i = 0; i < strings.size(); ++i) {
		typename Scanner::State want = Pire::Runner(sc).Begin().Run(strings[i]).End().State();
		CHECK(st[i] == want);                                  // the very same row address inside the host scanner
		CHECK((fin[i] != 0) == sc.Final(want));
		auto a = sc.AcceptedRegexps(st[i]);
		auto b = sc.AcceptedRegexps(want);
		CHECK((a.second - a.first) == (b.second - b.first));
		finals += sc.Final(want) ? 1 : 0;
	}
	CHECK(gpu.MatchCounts()[0] == finals);
	CHECK(gpu.MatchCounts()[1] == strings.size());
}

template <class Scanner>
static void TestSuite()
{
	std::vector<Pire::ystring> text = {
		"def abc ghi", "abc", "def abd ghi", "abc ghi", "def abc", "xaez", "xadddddddddddddddddddddddez", "xafez",
		"xx", "xxx", "xxxxxx", "xxxxxxx", "", "hello world", "say hello   wod", "hello world!",
		"......................................aaa.............", "......................................aab.............",
		"ZZZZZabcdeZZZZZZ", "ZZZZZZZZZZZZZabcdf", "aaabbb", "ccc", "HeadInnerInnerTail",
		Pire::ystring(5000, 'x') + "abc" + Pire::ystring(777, 'y'),
	};
	const char* patterns[] = {"abc", "^abc", "abc$", "ad*e", "^x{3,6}$", "hello\\s+w.+d$", "aaa", "[ab]{3}", "abcde",
	                          "Head(Inner)*Tail"};
	for (const char* re : patterns) {
		Scanner sc = Parse(re).template Compile<Scanner>();
		CompareAll(sc, text);
	}
	// Scanner::Glue -- pire_ut.cpp:648-705
	Scanner sc1 = Parse("aaa").template Compile<Scanner>();
	Scanner sc2 = Parse("bbb").template Compile<Scanner>();
	Scanner glued = Scanner::Glue(sc1, sc2);
	CHECK(glued.RegexpsCount() == 2);
	CompareAll(glued, text);
	{
		Pire::Hip::BatchRunner<Scanner> gpu(glued);
		std::vector<Pire::ystring> s = {"aaa", "bbb", "aaabbb", "ccc"};
		const auto& st = gpu.Begin().Run(s).End().States();
		auto r0 = glued.AcceptedRegexps(st[0]);
		CHECK(r0.second - r0.first == 1 && *r0.first == 0);
		auto r1 = glued.AcceptedRegexps(st[1]);
		CHECK(r1.second - r1.first == 1 && *r1.first == 1);
		auto r2 = glued.AcceptedRegexps(st[2]);
		CHECK(r2.second - r2.first == 2 && r2.first[0] == 0 && r2.first[1] == 1);
		auto r3 = glued.AcceptedRegexps(st[3]);
		CHECK(r3.second == r3.first);
		CHECK(gpu.MatchCounts()[2] == 2 && gpu.MatchCounts()[3] == 2);
	}
	Scanner sc3 = Parse("ccc").template Compile<Scanner>();
	Scanner glued3 = Scanner::Glue(sc3, glued);
	CHECK(glued3.RegexpsCount() == 3);
	CompareAll(glued3, text);
	// resume from saved states: Runner(sc, st) -- run.h:391-392
	{
		Scanner sc = Parse("hello\\s+w.+d$").template Compile<Scanner>();
		std::vector<Pire::ystring> head = {"say hel", "hello", "", "hello w"}, tail = {"lo   wod", " world!", "hello world", "orld"};
		Pire::Hip::Table<Scanner> table(sc);
		Pire::Hip::BatchRunner<Scanner> first(table);
		std::vector<typename Scanner::State> mid = first.Begin().Run(head).States();
		Pire::Hip::BatchRunner<Scanner> second(table);
		const auto& end = second.From(mid).Run(tail).End().States();
		for (size_t i = 0; i < head.size(); ++i) {
			typename Scanner::State want = Pire::Runner(sc).Begin().Run(head[i] + tail[i]).End().State();
			CHECK(end[i] == want);
		}
	}
	// Table::Adapt(): the dense-row ranking follows the data (explicitly here; the library also does it by itself),
	// and the states that come back are the same row addresses before and after
	{
		Scanner sc = glued3;
		Pire::Hip::Table<Scanner> table(sc);
		std::vector<Pire::ystring> many;
		for (int i = 0; i < 600; ++i)
			many.push_back(text[size_t(i) % text.size()] + (i % 3 ? " hello   world" : "abcabc"));
		Pire::Hip::BatchRunner<Scanner> before(table);
		const std::vector<typename Scanner::State> a = before.Begin().Run(many).End().States();
		const unsigned rows = table.Adapt();
		CHECK(rows <= 255);
		Pire::Hip::BatchRunner<Scanner> after(table);
		const std::vector<typename Scanner::State>& b = after.Begin().Run(many).End().States();
		CHECK(a.size() == b.size());
		for (size_t i = 0; i < a.size(); ++i) {
			CHECK(a[i] == b[i]);
			CHECK(a[i] == Pire::Runner(sc).Begin().Run(many[i]).End().State());
		}
		Pire::Hip::Table<Scanner>::FreezeRanking(true);
		Pire::Hip::Table<Scanner>::FreezeRanking(false);
	}
	// every GPU of the node behind the same surface (here: whatever devices there are; one is enough)
	{
		Pire::Hip::MultiBatchRunner<Scanner> gpus(glued3);
		CHECK(gpus.Devices() >= 1);
		const auto& st = gpus.Begin().Run(text).End().States();
		CHECK(st.size() == text.size());
		size_t finals = 0;
		for (size_t i = 0; i < text.size(); ++i) {
			typename Scanner::State want = Pire::Runner(glued3).Begin().Run(text[i]).End().State();
			CHECK(st[i] == want);
			finals += glued3.Final(want) ? 1 : 0;
		}
		CHECK(gpus.MatchCounts()[0] == finals && gpus.MatchCounts()[1] == text.size());
	}
	// empty scanner never matches and must not crash -- pire_ut.cpp:760-830
	{
		Scanner empty;
		CHECK(empty.Empty());
		Pire::Hip::BatchRunner<Scanner> gpu(empty);
		std::vector<Pire::ystring> s = {"a strin", ""};
		const auto& fin = gpu.Begin().Run(s).End().Finals();
		CHECK(!fin[0] && !fin[1]);
	}
}

// LongestPrefix / ShortestPrefix (run.h:277-311) and SlowScanner (pire_ut.cpp:707-714) through the shim.
static void TestPrefixAndSlow()
{
	std::vector<Pire::ystring> text = {"aaab", "", "fixed prefix", "a fixed nonexistent prefix", "aaabbb", "bbbbbb", "xaaay"};
	Pire::ystring flat;
	std::vector<uint64_t> offs(1, 0);
	for (auto& s : text) { flat += s; offs.push_back(flat.size()); }
	const char* patterns[] = {"a*", "a", "fixed", "aa*", "a+b", "aaa"};
	for (const char* re : patterns) {
		Pire::Scanner sc = Parse(re, false).Compile<Pire::Scanner>();
		Pire::Hip::Table<Pire::Scanner> table(sc);
		for (int tb = 0; tb < 2; ++tb)
			for (int te = 0; te < 2; ++te) {
				auto lp = Pire::Hip::BatchLongestPrefix(table, flat.data(), offs.data(), text.size(), tb, te);
				auto sp = Pire::Hip::BatchShortestPrefix(table, flat.data(), offs.data(), text.size(), tb, te);
				for (size_t i = 0; i < text.size(); ++i) {
					const char* b = flat.data() + offs[i];
					const char* e = flat.data() + offs[i + 1];
					CHECK(lp[i] == Pire::LongestPrefix(sc, b, e, tb, te));
					CHECK(sp[i] == Pire::ShortestPrefix(sc, b, e, tb, te));
				}
			}
	}
	Pire::SlowScanner slow = Parse("a.{30}$").Compile<Pire::SlowScanner>();
	Pire::Hip::SlowBatchRunner run(slow);
	std::vector<Pire::ystring> s = {"....a..............................", "....a...............................",
	                                "....a.............................", ""};
	Pire::ystring f2;
	std::vector<uint64_t> o2(1, 0);
	for (auto& x : s) { f2 += x; o2.push_back(f2.size()); }
	std::vector<char> m = run.Matches(f2.data(), o2.data(), s.size());
	for (size_t i = 0; i < s.size(); ++i)
		CHECK((m[i] != 0) == bool(Pire::Runner(slow).Begin().Run(s[i]).End()));
	CHECK(m[0] && !m[1] && !m[2]);
}

// Pire::SimpleScanner through the same shim: identical State values, Final, AcceptedRegexps (simple.h:62-68).
static void TestSimpleScanner()
{
	const char* patterns[] = {"hello\\s+w.+d$", "abc|def", "ad*e", "Head(Inner)*Tail"};
	std::vector<Pire::ystring> strings;
	const char* fixed[] = {"hello world", "Hello world", "say hello   wod", "", "abc", "xxdefyy", "addde", "HeadInnerInnerTail",
	                       "HeadInneTail", "hello wd"};
	for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i)
		strings.push_back(fixed[i]);
	unsigned seed = 12345;
	for (int i = 0; i < 600; ++i) {
		Pire::ystring s;
		seed = seed * 1103515245u + 12345u;
		const size_t len = (seed >> 16) % 90;
		for (size_t k = 0; k < len; ++k) {
			seed = seed * 1103515245u + 12345u;
			s.push_back("abcdefHeadInrTl w\thello"[(seed >> 16) % 23]);
		}
		strings.push_back(s);
	}
	for (size_t p = 0; p < sizeof(patterns) / sizeof(patterns[0]); ++p) {
		Pire::SimpleScanner sc = Parse(patterns[p]).Compile<Pire::SimpleScanner>();
		CompareAll(sc, strings);
	}
	Pire::SimpleScanner empty;
	CHECK(empty.Empty());
	CompareAll(empty, strings);
}

// Pire::HalfFinalScanner: State::Result(r) per string, built as tests/count_ut.cpp:503-527 builds its scanners.
static void TestHalfFinal()
{
	const char* regexps[] = {"ab+", "(ab)+", "ab+c|b", "a[a-z]+c|b"};
	std::vector<Pire::ystring> strings;
	const char* fixed[] = {"abbabbbabbbbbb", "ababbababbab", "abbbbbbbbbbc", "abeeeebeeeeeeeeeceeaeebeeeaeecceebeeaeebeeb", "", "b"};
	for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i)
		strings.push_back(fixed[i]);
	unsigned seed = 777;
	for (int i = 0; i < 500; ++i) {
		Pire::ystring s;
		seed = seed * 1103515245u + 12345u;
		const size_t len = (seed >> 16) % 70;
		for (size_t k = 0; k < len; ++k) {
			seed = seed * 1103515245u + 12345u;
			s.push_back("abcde "[(seed >> 16) % 6]);
		}
		strings.push_back(s);
	}
	for (size_t p = 0; p < sizeof(regexps) / sizeof(regexps[0]); ++p) {
		const Pire::Fsm re = Parse(regexps[p], false);
		Pire::HalfFinalScanner glued;
		for (int mode = 0; mode < 5; ++mode) {
			Pire::HalfFinalFsm fsm(re);
			switch (mode) {
			case 0: fsm.MakeGreedyCounter(true); break;
			case 1: fsm.MakeGreedyCounter(false); break;
			case 2: fsm.MakeNonGreedyCounter(true, true); break;
			case 3: fsm.MakeNonGreedyCounter(true, false); break;
			default: fsm.MakeNonGreedyCounter(false); break;
			}
			Pire::HalfFinalScanner one(fsm);
			glued = mode == 0 ? one : Pire::HalfFinalScanner::Glue(glued, one);
		}
		CHECK(glued.RegexpsCount() == 5);
		Pire::Hip::HalfFinalBatchRunner<> gpu(glued);
		gpu.Begin().Run(strings).End();
		for (size_t i = 0; i < strings.size(); ++i) {
			Pire::HalfFinalScanner::State st;
			glued.Initialize(st);
			Pire::Step(glued, st, Pire::BeginMark);
			Pire::Run(glued, st, strings[i].data(), strings[i].data() + strings[i].size());
			Pire::Step(glued, st, Pire::EndMark);
			for (size_t r = 0; r < 5; ++r)
				CHECK(gpu.Result(i, r) == st.Result(r));
			CHECK((gpu.Finals()[i] != 0) == glued.Final(st));
			CHECK(gpu.StateIndices()[i] == glued.StateIndex(st));
		}
	}
}

// Pire::ScannerPair (scanners/pair.h): the pair of states and Final = either.
static void TestScannerPair()
{
	Pire::Scanner s1 = Parse("hello\\s+w.+d$").Compile<Pire::Scanner>();
	Pire::SimpleScanner s2 = Parse("abc|def").Compile<Pire::SimpleScanner>();
	typedef Pire::ScannerPair<Pire::Scanner, Pire::SimpleScanner> Pair;
	Pair pair(s1, s2);
	std::vector<Pire::ystring> strings;
	const char* fixed[] = {"hello world", "abc", "say hello   wod def", "", "nothing", "xxdef", "hello wd"};
	for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i)
		strings.push_back(fixed[i]);
	Pire::Hip::PairBatchRunner<Pire::Scanner, Pire::SimpleScanner> gpu(s1, s2);
	gpu.Begin().Run(strings).End();
	const std::vector<Pair::State> st = gpu.States();
	const std::vector<char> fin = gpu.Finals();
	CHECK(st.size() == strings.size());
	for (size_t i = 0; i < strings.size(); ++i) {
		Pair::State want = Pire::Runner(pair).Begin().Run(strings[i]).End().State();
		CHECK(st[i].first == want.first && st[i].second == want.second);
		CHECK((fin[i] != 0) == pair.Final(want));
		CHECK(pair.StateIndex(st[i]) == pair.StateIndex(want));
	}
	CHECK(fin[0] && fin[1] && fin[2] && !fin[3] && !fin[4] && fin[5] && !fin[6]);

	// device-resident fixed-length records: ONE fused pass (pire_hip_run_pair_strided), against
	// Pire::Run(scanner1, scanner2, state1, state2, begin, end) of run.h:229-241
	const size_t n = 1000, len = 512;
	std::vector<char> recs(n * len);
	unsigned seed = 99;
	for (size_t i = 0; i < recs.size(); ++i) {
		seed = seed * 1103515245u + 12345u;
		recs[i] = "abcdefhelo w xyz"[(seed >> 16) & 15];
	}
	for (size_t i = 0; i < n; i += 7)
		memcpy(&recs[i * len + len - 12], "hello  world", 12);
	void* dText = nullptr;
	Pire::Hip::Check(pire_hip_device_alloc(recs.size(), &dText));
	Pire::Hip::Check(pire_hip_copy_to_device(dText, recs.data(), recs.size(), nullptr));
	Pire::Hip::Check(pire_hip_stream_synchronize(nullptr));
	Pire::Hip::PairBatchRunner<Pire::Scanner, Pire::SimpleScanner> dev(s1, s2);
	dev.Begin().End();
	const std::vector<Pair::State> dst = dev.RunDeviceStrided(dText, n, len, len).States();
	const std::vector<char> dfin = dev.Finals();
	CHECK(std::string(pire_hip_last_kernel()) == "generic" || std::string(pire_hip_last_kernel()) == "pair_tiled");
	size_t finals = 0;
	for (size_t i = 0; i < n; ++i) {
		Pire::Scanner::State a;
		Pire::SimpleScanner::State b;
		s1.Initialize(a);
		s2.Initialize(b);
		Pire::Step(s1, a, Pire::BeginMark);
		Pire::Step(s2, b, Pire::BeginMark);
		Pire::Run(s1, s2, a, b, &recs[i * len], &recs[i * len] + len);
		Pire::Step(s1, a, Pire::EndMark);
		Pire::Step(s2, b, Pire::EndMark);
		CHECK(dst[i].first == a && dst[i].second == b);
		CHECK((dfin[i] != 0) == (s1.Final(a) || s2.Final(b)));
		finals += dfin[i] != 0;
	}
	CHECK(finals >= n / 7);
	pire_hip_device_free(dText);
}

// Pire::CountingScanner / AdvancedCountingScanner (extra/count.h), built and driven as tests/count_ut.cpp:54-93 does.
template <class CountScanner>
static void TestCounting()
{
	const char* res[] = {"[a-z]+", "http", "abc"};
	const char* seps[] = {"\\s", ".*", ".*"};
	std::vector<Pire::ystring> strings;
	const char* fixed[] = {"abc def, abc def ghi, abc", "http://aaa, http://bbb, something in the middle, http://ccc, end",
	                       "abcabcabcabc", "", "x"};
	for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i)
		strings.push_back(fixed[i]);
	unsigned seed = 4242;
	for (int i = 0; i < 500; ++i) {
		Pire::ystring s;
		seed = seed * 1103515245u + 12345u;
		const size_t len = (seed >> 16) % 120;
		for (size_t k = 0; k < len; ++k) {
			seed = seed * 1103515245u + 12345u;
			s.push_back("abc def,http:/\n"[(seed >> 16) % 15]);
		}
		strings.push_back(s);
	}
	CountScanner glued;
	for (int k = 0; k < 3; ++k) {
		CountScanner one(Parse(res[k], false), Parse(seps[k], false));
		glued = k == 0 ? one : CountScanner::Glue(glued, one);
	}
	CHECK(glued.RegexpsCount() == 3);
	Pire::Hip::CountingBatchRunner<CountScanner> gpu(glued);
	gpu.Begin().Run(strings).End();
	for (size_t i = 0; i < strings.size(); ++i) {
		typename CountScanner::State st;
		glued.Initialize(st);
		Pire::Step(glued, st, Pire::BeginMark);
		Pire::Run(glued, st, strings[i].data(), strings[i].data() + strings[i].size());
		Pire::Step(glued, st, Pire::EndMark);
		for (int r = 0; r < 3; ++r)
			CHECK(gpu.Result(i, r) == st.Result(r));
		CHECK(gpu.StateIndices()[i] == glued.StateIndex(st));
	}
	CHECK(gpu.Result(0, 0) == 3 && gpu.Result(1, 1) == 3 && gpu.Result(2, 2) == 4);   // count_ut.cpp:97, 102, 103
}

// Pire::CapturingScanner (extra/capture.h), compiled and driven as tests/capture_ut.cpp:39-91 does.
static void TestCapture()
{
	const char* regexp = "google_id\\s*=\\s*[\'\"]([a-z0-9]+)[\'\"]\\s*;";
	Pire::Lexer lexer;
	lexer.Assign(regexp, regexp + strlen(regexp));
	lexer.AddFeature(Pire::Features::CaseInsensitive());
	lexer.AddFeature(Pire::Features::Capture(1));
	Pire::Fsm fsm = lexer.Parse();
	fsm.Surround();
	fsm.Determine();
	Pire::CapturingScanner sc = fsm.Compile<Pire::CapturingScanner>();
	std::vector<Pire::ystring> strings;
	const char* fixed[] = {"google_id = 'abcde';", "var google_id = 'abcde'; eval(google_id);", "google_id != 'abcde';",
	                       "google_id = 'abcde'; google_id = 'xyz';", "var google_id = 'abc de'; google_id = 'xyz';", ""};
	for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i)
		strings.push_back(fixed[i]);
	unsigned seed = 99;
	for (int i = 0; i < 400; ++i) {
		Pire::ystring s;
		for (int part = 0; part < 3; ++part) {
			seed = seed * 1103515245u + 12345u;
			if ((seed >> 16) & 1) {
				s += fixed[(seed >> 17) % 5];
			} else {
				const size_t len = (seed >> 17) % 20;
				for (size_t k = 0; k < len; ++k) {
					seed = seed * 1103515245u + 12345u;
					s.push_back("google_id ='x1;"[(seed >> 16) % 15]);
				}
			}
		}
		strings.push_back(s);
	}
	Pire::Hip::CaptureBatchRunner gpu(sc);
	gpu.Begin().Run(strings).End();
	size_t captured = 0;
	for (size_t i = 0; i < strings.size(); ++i) {
		Pire::CapturingScanner::State st;
		sc.Initialize(st);
		Pire::Step(sc, st, Pire::BeginMark);
		Pire::Run(sc, st, strings[i].data(), strings[i].data() + strings[i].size());
		Pire::Step(sc, st, Pire::EndMark);
		CHECK(gpu.Captured(i) == st.Captured());
		CHECK(gpu.Begin(i) == st.Begin() && gpu.End(i) == st.End());
		CHECK(gpu.Final(i) == sc.Final(st));
		CHECK(gpu.StateIndex(i) == sc.StateIndex(st));
		captured += st.Captured() ? 1 : 0;
	}
	CHECK(captured > 4);
	CHECK(gpu.Captured(0) && Pire::ystring(strings[0].data() + gpu.Begin(0) - 1, strings[0].data() + gpu.End(0) - 1) == "abcde");
	CHECK(gpu.Captured(4) && Pire::ystring(strings[4].data() + gpu.Begin(4) - 1, strings[4].data() + gpu.End(4) - 1) == "xyz");
	CHECK(!gpu.Captured(2));
}

// A few long strings: the library cuts them into segments scanned in parallel (segmented.hip); what comes back must
// still be the State the one-byte-at-a-time Runner of the reference ends in.
static void TestLongStrings()
{
	Pire::Scanner sc = Pire::Scanner::Glue(Parse("hello\\s+w.+d$").Compile<Pire::Scanner>(),
	                                       Parse("[0-9]{3}-[0-9]{4}").Compile<Pire::Scanner>());
	std::vector<Pire::ystring> strings;
	unsigned seed = 12345;
	for (int k = 0; k < 3; ++k) {
		Pire::ystring s;
		const size_t len = (size_t(1) << 20) + 777 * k;
		s.reserve(len + 32);
		while (s.size() < len) {
			seed = seed * 1664525u + 1013904223u;
			const unsigned r = seed >> 16;
			if (r % 5000 == 0)
				s += "hello   w";
			else if (r % 7000 == 1)
				s += " 555-1234 ";
			else
				s += char(' ' + r % 95);
		}
		if (k == 1)
			s += "orld";
		strings.push_back(s);
	}
	CompareAll(sc, strings);
	CHECK(std::string(pire_hip_last_kernel()).find("segmented") == 0);
}

// PrefixSuffix of pire_ut.cpp:278-306 through the shim: the suffix that ends where a prefix search stopped.
static void TestSuffix()
{
	static const char* text = "1234567890 --> middle --> end";
	Pire::Fsm fsm = Parse("-->", false);
	Pire::Scanner rsc = fsm.Reverse().Compile<Pire::Scanner>();
	Pire::Hip::Table<Pire::Scanner> table(rsc);
	const uint64_t offs[3] = {0, 14, 14 + 25};
	Pire::ystring flat = Pire::ystring(text, 14) + Pire::ystring(text, 25);
	for (int longest = 0; longest < 2; ++longest) {
		std::vector<const char*> got = longest ? Pire::Hip::BatchLongestSuffix(table, flat.data(), offs, 2)
		                                       : Pire::Hip::BatchShortestSuffix(table, flat.data(), offs, 2);
		for (int i = 0; i < 2; ++i) {
			const char* b = flat.data() + offs[i];
			const char* e = flat.data() + offs[i + 1];
			const char* want = longest ? Pire::LongestSuffix(rsc, e - 1, b - 1) : Pire::ShortestSuffix(rsc, e - 1, b - 1);
			CHECK(got[i] == want);
			CHECK(want != nullptr && want + 1 == e - 3);
		}
	}
}

// Device-resident text (RunDevice / RunDeviceStrided) and the pipelined host-pointer mode (batches of >= 64 MiB), both
// against the reference Runner on a sample and against each other on the whole batch; prints the rates of the three
// ways to hand a batch over (informational: the numbers quoted in DESIGN.md come from this line).
#include <chrono>
static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void TestDeviceMode()
{
	typedef Pire::Scanner Scanner;
	Scanner sc = Scanner::Glue(Parse("hello\\s+w.+d$").Compile<Scanner>(), Parse("Head(Inner)*Tail").Compile<Scanner>());
	const size_t n = size_t(1) << 16, len = 4096;              // 256 MiB: host mode takes the chunked pipeline
	void* pinnedRaw = nullptr;
	Pire::Hip::Check(pire_hip_host_alloc(n * len, &pinnedRaw));
	char* pinned = static_cast<char*>(pinnedRaw);
	std::vector<char> pageable(n * len);
	uint64_t h = 88172645463325252ull;
	for (size_t i = 0; i < n * len; i += 8) {                  // printable bytes, xorshift64
		h ^= h << 13; h ^= h >> 7; h ^= h << 17;
		for (int b = 0; b < 8; ++b)
			pageable[i + b] = char(0x20 + ((h >> (8 * b)) & 0xFF) % 95);
	}
	const char* w1 = "hello   world";
	const char* w2 = "HeadInnerInnerTail";
	for (size_t s = 0; s < n; s += 3) {
		memcpy(&pageable[s * len + len - strlen(w1)], w1, strlen(w1));          // '$'-anchored: at the tail
		if (s % 2 == 0)
			memcpy(&pageable[s * len + 1000 + s % 1000], w2, strlen(w2));       // unanchored: somewhere
	}
	memcpy(pinned, pageable.data(), n * len);
	std::vector<uint64_t> offs(n + 1);
	for (size_t i = 0; i <= n; ++i)
		offs[i] = i * len;

	Pire::Hip::Table<Scanner> table(sc);
	// host pointers, pageable and pinned
	Pire::Hip::BatchRunner<Scanner> host(table);
	host.Begin().Run(pageable.data(), offs.data(), n).End().MatchCounts();      // warm-up: uploads the table
	double t0 = Now();
	Pire::Hip::BatchRunner<Scanner> host2(table);
	const std::vector<Scanner::State> viaHost = host2.Begin().Run(pageable.data(), offs.data(), n).End().States();
	const double hostSec = Now() - t0;
	t0 = Now();
	Pire::Hip::BatchRunner<Scanner> host3(table);
	const std::vector<Scanner::State> viaPinned = host3.Begin().Run(pinned, offs.data(), n).End().States();
	const double pinnedSec = Now() - t0;
	CHECK(viaHost.size() == n && viaPinned == viaHost);
	CHECK(host2.MatchCounts() == host3.MatchCounts() && host2.MatchCounts()[1] == n);
	for (size_t i = 0; i < n; i += 997) {
		Scanner::State want = Pire::Runner(sc).Begin().Run(pageable.data() + i * len, len).End().State();
		CHECK(viaHost[i] == want);
	}
	CHECK(host2.MatchCounts()[0] >= n / 3);

	// device-resident text
	void* dText = nullptr;
	void* dOffs = nullptr;
	Pire::Hip::Check(pire_hip_device_alloc(n * len, &dText));
	Pire::Hip::Check(pire_hip_device_alloc((n + 1) * 8, &dOffs));
	Pire::Hip::Check(pire_hip_copy_to_device(dText, pinned, n * len, nullptr));
	Pire::Hip::Check(pire_hip_copy_to_device(dOffs, offs.data(), (n + 1) * 8, nullptr));
	Pire::Hip::Check(pire_hip_stream_synchronize(nullptr));
	Pire::Hip::BatchRunner<Scanner> dev(table);
	dev.Begin().End();
	CHECK(dev.RunDeviceStrided(dText, n, len, len).MatchCounts() == host2.MatchCounts());
	CHECK(dev.States() == viaHost);
	const int reps = 20;
	t0 = Now();
	for (int r = 0; r < reps; ++r)
		dev.RunDeviceStrided(dText, n, len, len).MatchCounts();
	const double devSec = (Now() - t0) / reps;
	CHECK(dev.RunDevice(dText, static_cast<const uint64_t*>(dOffs), n).MatchCounts() == host2.MatchCounts());   // ragged kernel
	CHECK(dev.States() == viaHost);
	CHECK(dev.DeviceStateIndices() != nullptr && dev.DeviceFinals() != nullptr);
	// resumed states on the device: first halves, then second halves from the returned states
	{
		Pire::Hip::BatchRunner<Scanner> a(table), b(table);
		const std::vector<Scanner::State> mid = a.Begin().RunDeviceStrided(dText, n, len / 2, len).States();
		const std::vector<Scanner::State> end =
		    b.From(mid).RunDeviceStrided(static_cast<const char*>(dText) + len / 2, n, len / 2, len).End().States();
		CHECK(end == viaHost);
	}
	const double gb = double(n * len) / 1e9;
	printf("shim rates, %zu x %zu B (%.0f MiB): host pointers pageable %.1f GB/s, pinned (pire_hip_host_alloc) %.1f GB/s, "
	       "device-resident (RunDeviceStrided + MatchCounts) %.1f GB/s\n",
	       n, len, double(n * len) / 1048576.0, gb / hostSec, gb / pinnedSec, gb / devSec);
	pire_hip_device_free(dText);
	pire_hip_device_free(dOffs);
	pire_hip_host_free(pinned);
}

int main()
{
	try {
		TestDeviceMode();
		TestSuffix();
		TestLongStrings();
		TestPrefixAndSlow();
		TestCapture();
		TestCounting<Pire::CountingScanner>();
		TestCounting<Pire::AdvancedCountingScanner>();
		TestCounting<Pire::NoGlueLimitCountingScanner>();
		TestScannerPair();
		TestHalfFinal();
		TestSimpleScanner();
		TestSuite<Pire::Scanner>();
		TestSuite<Pire::NonrelocScanner>();
		TestSuite<Pire::ScannerNoMask>();
	} catch (const std::exception& e) {
		fprintf(stderr, "exception: %s\n", e.what());
		return 2;
	}
	if (g_fail) {
		fprintf(stderr, "FAILED %d of %d checks\n", g_fail, g_checks);
		return 1;
	}
	printf("OK(shim %d checks)\n", g_checks);
	return 0;
}
