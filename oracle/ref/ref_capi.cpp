/*
 * TEST INFRASTRUCTURE (oracle/_ref build only) -- not part of the product path.
 *
 * A plain C ABI over the UNMODIFIED reference library, compiled from the sources
 * where they lie under /root/reference/pire (see oracle/Makefile).  It exists so
 * that python tests, the golden-vector generator and bench.py's cpu_baseline leg
 * can (a) compile patterns into Pire::Scanner tables exactly as the reference's
 * own benchmark does, (b) obtain the public Scanner::Save() blob the GPU table is
 * ingested from, and (c) run the reference's own
 * Runner(sc).Begin().Run(ptr,len).End() on the same bytes the GPU scans.
 *
 * Nothing here is algorithm code of ours: every function is a thin call into the
 * reference's public API (file:line cited per function).
 */

#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <pire/pire.h>
#include <pire/extra.h>
#include <pire/stub/memstreams.h>

namespace {

using Pire::Scanner;
using Pire::NonrelocScanner;

struct RefScanner {
	Scanner         reloc;
	NonrelocScanner nonreloc;     // deep copy, multi.h:213-220
	Pire::SlowScanner slow;       // only for kind == 2 handles
	bool            hasSlow = false;
};

thread_local std::string g_err;

void SetErr(const char* what) { g_err = what ? what : "unknown error"; }

/* tests/common.h:40-69 (ParseRegexp): option letters i,u,n,a. */
Pire::Fsm ParseOne(const char* pattern, const char* options)
{
	Pire::Lexer lexer;
	Pire::TVector<Pire::wchar32> ucs4;
	bool surround = true, reverse = false;
	for (const char* o = options ? options : ""; *o; ++o) {
		switch (*o) {
		case 'r': reverse = true; break;   // Fsm::Reverse(), for the suffix searches (pire_ut.cpp:283)
		case 'i': lexer.AddFeature(Pire::Features::CaseInsensitive()); break;
		case 'u': lexer.SetEncoding(Pire::Encodings::Utf8()); break;
		case 'n': surround = false; break;
		case 'a': lexer.AddFeature(Pire::Features::AndNotSupport()); break;
		default: throw Pire::Error(std::string("Unknown option: ") + *o);
		}
	}
	lexer.Encoding().FromLocal(pattern, pattern + strlen(pattern), std::back_inserter(ucs4));
	lexer.Assign(ucs4.begin(), ucs4.end());
	Pire::Fsm fsm = lexer.Parse();
	if (surround)
		fsm.Surround();           // fsm.cpp:1198-1203; bench.cpp:101-102,116-117
	if (reverse)
		fsm = fsm.Reverse();
	return fsm;
}

/* State <-> index using public API only (multi.h:161, 281-284, 99, 119, 347). */
template<class Sc>
struct Geometry {
	size_t base;
	size_t stride;
	explicit Geometry(const Sc& sc)
	{
		typedef typename Sc::Transition Tr;
		const size_t header = sizeof(typename Sc::ScannerRowHeader) / sizeof(Tr);
		const size_t align = sizeof(Pire::Impl::MaxSizeWord) / sizeof(Tr);
		const size_t row = (sc.LettersCount() + header + align - 1) / align * align;
		stride = row * sizeof(Tr);
		typename Sc::State init;
		sc.Initialize(init);
		base = init - sc.StateIndex(init) * stride;
	}
	size_t ToState(uint32_t idx) const { return base + size_t(idx) * stride; }
};

enum { FLAG_BEGIN = 1, FLAG_END = 2 };

template<class Sc>
void RunRange(const Sc& sc, const char* text, const uint64_t* offsets, uint64_t lo, uint64_t hi,
              uint32_t flags, const uint32_t* initIdx, uint32_t* outIdx, uint8_t* outFinal)
{
	Geometry<Sc> geo(sc);
	for (uint64_t i = lo; i < hi; ++i) {
		typename Sc::State st;
		if (initIdx)
			st = geo.ToState(initIdx[i]);
		else
			sc.Initialize(st);
		// run.h:365-392 (RunHelper) spelled out so that Begin/End are optional
		if (flags & FLAG_BEGIN)
			Pire::Step(sc, st, Pire::BeginMark);
		Pire::Run(sc, st, text + offsets[i], text + offsets[i + 1]);
		if (flags & FLAG_END)
			Pire::Step(sc, st, Pire::EndMark);
		if (outIdx)
			outIdx[i] = static_cast<uint32_t>(sc.StateIndex(st));
		if (outFinal)
			outFinal[i] = sc.Final(st) ? 1 : 0;
	}
}

template<class Sc>
void RunThreads(const Sc& sc, const char* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                const uint32_t* initIdx, uint32_t* outIdx, uint8_t* outFinal, int threads)
{
	if (threads <= 1) {
		RunRange(sc, text, offsets, 0, n, flags, initIdx, outIdx, outFinal);
		return;
	}
	// The reference has no threading; this index-range sharding driver is ours.
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; ++t) {
		uint64_t lo = n * t / threads, hi = n * (t + 1) / threads;
		pool.emplace_back([&, lo, hi] { RunRange(sc, text, offsets, lo, hi, flags, initIdx, outIdx, outFinal); });
	}
	for (auto& th : pool)
		th.join();
}

} // namespace

extern "C" {

const char* pire_ref_last_error(void) { return g_err.c_str(); }

/* Compile n patterns and glue them left to right -- tools/bench/bench.cpp:108-132. */
void* pire_ref_compile(const char* const* patterns, const char* const* options, int n, size_t glueMaxSize)
{
	try {
		std::unique_ptr<RefScanner> h(new RefScanner);
		for (int i = 0; i < n; ++i) {
			Pire::Fsm fsm = ParseOne(patterns[i], options ? options[i] : "");
			Scanner one = fsm.Compile<Scanner>();            // fsm.h:273-277
			if (i == 0) {
				one.Swap(h->reloc);
			} else {
				h->reloc = Scanner::Glue(h->reloc, one, glueMaxSize);   // multi.h:1092-1103
				if (h->reloc.Empty()) {
					SetErr("Scanner gluing failed - pattern too complicated");
					return nullptr;
				}
			}
		}
		h->nonreloc = NonrelocScanner(h->reloc);
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

/* A dictionary scanner, built the way samples/blacklist/blacklist.cpp:65-76 (Generate) builds one: the words joined
 * with Fsm::operator|= as fixed strings (Fsm().Append(word), no regexp syntax), then
 *   mode 0: wrapped as the sample does -- "^([a-z]+://)?([A-Za-z0-9\\-]+\\.)*" + words + "(/.*)?$" -- and compiled with
 *           Scanner(Fsm) (multi.h:123-131);
 *   mode 1: the same words Surround()ed (fsm.cpp:1198-1203) like every pattern of tools/bench (bench.cpp:101-102): the
 *           dictionary searched anywhere in a text;
 *   mode 2: as mode 1, every word a PATTERN (letters only) parsed by the lexer with the UTF-8 encoding -- pire_ut.cpp:181-209's
 *           way to a table of multi-byte letters. */
void* pire_ref_compile_dictionary(const char* const* words, int n, int mode)
{
	try {
		std::unique_ptr<RefScanner> h(new RefScanner);
		Pire::Fsm re = Pire::Fsm::MakeFalse();
		for (int i = 0; i < n; ++i) {
			if (mode == 2)
				re |= ParseOne(words[i], "un");   // through the lexer with Encodings::Utf8() (encoding.cpp:99-111, re_lexer.cpp), not Surround()ed yet
			else
				re |= Pire::Fsm().Append(words[i]);
		}
		if (mode == 0)
			re = Pire::Lexer("^([a-z]+://)?([A-Za-z0-9\\-]+\\.)*").Parse() + re + Pire::Lexer("(/.*)?$").Parse();
		else
			re.Surround();
		Scanner(re).Swap(h->reloc);
		h->nonreloc = NonrelocScanner(h->reloc);
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

/* Load a Scanner::Save() blob -- multi.h:575-599. */
void* pire_ref_load(const void* blob, size_t len)
{
	try {
		std::unique_ptr<RefScanner> h(new RefScanner);
		Pire::MemoryInput in(static_cast<const char*>(blob), len);
		h->reloc.Load(&in);
		h->nonreloc = NonrelocScanner(h->reloc);
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void pire_ref_free(void* h) { delete static_cast<RefScanner*>(h); }

/* Scanner::Save -- multi.h:620-624, 557-573.  Returns the blob size; copies if it fits. */
size_t pire_ref_save(void* h, void* buf, size_t cap)
{
	std::ostringstream out;
	static_cast<RefScanner*>(h)->reloc.Save(&out);
	const std::string s = out.str();
	if (buf && cap >= s.size())
		memcpy(buf, s.data(), s.size());
	return s.size();
}

size_t pire_ref_size(void* h)          { return static_cast<RefScanner*>(h)->reloc.Size(); }          // multi.h:134
size_t pire_ref_letters(void* h)       { return static_cast<RefScanner*>(h)->reloc.LettersCount(); }  // multi.h:140
size_t pire_ref_regexps(void* h)       { return static_cast<RefScanner*>(h)->reloc.RegexpsCount(); }  // multi.h:139
size_t pire_ref_bufsize(void* h)       { return static_cast<RefScanner*>(h)->reloc.BufSize(); }       // multi.h:297-305
int    pire_ref_empty(void* h)         { return static_cast<RefScanner*>(h)->reloc.Empty() ? 1 : 0; } // multi.h:135

uint32_t pire_ref_initial_index(void* h)
{
	const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
	Scanner::State st;
	sc.Initialize(st);
	return static_cast<uint32_t>(sc.StateIndex(st));
}

/* One Step() from a state index -- run.h:50-57, multi.h:189-192. */
uint32_t pire_ref_next(void* h, uint32_t idx, uint32_t ch)
{
	const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
	Scanner::State st = Geometry<Scanner>(sc).ToState(idx);
	Pire::Step(sc, st, static_cast<Pire::Char>(ch));
	return static_cast<uint32_t>(sc.StateIndex(st));
}

int pire_ref_final(void* h, uint32_t idx)   // multi.h:143
{
	const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
	return sc.Final(Geometry<Scanner>(sc).ToState(idx)) ? 1 : 0;
}

int pire_ref_dead(void* h, uint32_t idx)    // multi.h:147
{
	const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
	return sc.Dead(Geometry<Scanner>(sc).ToState(idx)) ? 1 : 0;
}

/* AcceptedRegexps -- multi.h:149-158.  Returns the count; writes up to cap ids. */
size_t pire_ref_accepted(void* h, uint32_t idx, uint64_t* out, size_t cap)
{
	const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
	auto range = sc.AcceptedRegexps(Geometry<Scanner>(sc).ToState(idx));
	size_t n = 0;
	for (const size_t* p = range.first; p != range.second; ++p, ++n)
		if (out && n < cap)
			out[n] = *p;
	return n;
}

/*
 * Runner(sc).Begin().Run(ptr,len).End() per string -- run.h:365-392, 271-275; call shape of
 * tools/bench/bench.cpp:244.  kind: 0 = Pire::Scanner, 1 = Pire::NonrelocScanner.
 * flags: bit0 = Begin(), bit1 = End().  threads > 1 shards strings by index (our driver).
 */
int pire_ref_run(void* h, int kind, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                 const uint32_t* initIdx, uint32_t* outIdx, uint8_t* outFinal, int threads)
{
	try {
		RefScanner* r = static_cast<RefScanner*>(h);
		const char* t = static_cast<const char*>(text);
		if (kind == 0)
			RunThreads(r->reloc, t, offsets, n, flags, initIdx, outIdx, outFinal, threads);
		else if (kind == 1)
			RunThreads(r->nonreloc, t, offsets, n, flags, initIdx, outIdx, outFinal, threads);
		else {
			SetErr("unknown scanner kind");
			return -1;
		}
		return 0;
	} catch (const std::exception& e) {
		SetErr(e.what());
		return -1;
	}
}

/*
 * LongestPrefix / ShortestPrefix -- run.h:277-311.  Writes the prefix length, or -1 for "no prefix"
 * (null return in the reference).
 */
int pire_ref_prefix(void* h, int longest, const void* text, const uint64_t* offsets, uint64_t n,
                    int throughBegin, int throughEnd, int64_t* outLen)
{
	try {
		const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
		// the reference reports "no prefix" as a null pointer, so a null base would make an empty match at
		// position 0 look like a miss: give empty batches a real address
		static const char kEmpty[1] = {0};
		const char* t = text ? static_cast<const char*>(text) : kEmpty;
		for (uint64_t i = 0; i < n; ++i) {
			const char* b = t + offsets[i];
			const char* e = t + offsets[i + 1];
			const char* p = longest ? Pire::LongestPrefix(sc, b, e, throughBegin != 0, throughEnd != 0)
			                        : Pire::ShortestPrefix(sc, b, e, throughBegin != 0, throughEnd != 0);
			outLen[i] = p ? static_cast<int64_t>(p - b) : -1;
		}
		return 0;
	} catch (const std::exception& e) {
		SetErr(e.what());
		return -1;
	}
}

/*
 * LongestSuffix / ShortestSuffix -- run.h:313-362.  Writes the suffix length ((last byte) - returned pointer), or -1
 * for the reference's null return.
 */
int pire_ref_suffix(void* h, int longest, const void* text, const uint64_t* offsets, uint64_t n,
                    int throughEnd, int throughBegin, int64_t* outLen)
{
	try {
		const Scanner& sc = static_cast<RefScanner*>(h)->reloc;
		static const char kEmpty[2] = {0, 0};
		const char* t = text ? static_cast<const char*>(text) : kEmpty + 1;
		for (uint64_t i = 0; i < n; ++i) {
			const char* rbegin = t + offsets[i + 1] - 1;
			const char* rend = t + offsets[i] - 1;
			const char* p = longest ? Pire::LongestSuffix(sc, rbegin, rend, throughEnd != 0, throughBegin != 0)
			                        : Pire::ShortestSuffix(sc, rbegin, rend, throughEnd != 0, throughBegin != 0);
			outLen[i] = p ? static_cast<int64_t>(rbegin - p) : -1;
		}
		return 0;
	} catch (const std::exception& e) {
		SetErr(e.what());
		return -1;
	}
}

/* ------------------------------------------------------------------ SlowScanner (scanners/slow.h) */

struct RefSlow {
	Pire::SlowScanner sc;
};

/* Fsm::Compile<SlowScanner>() -- slow.h:420-423; one pattern, options as ParseOne. */
void* pire_ref_slow_compile(const char* pattern, const char* options)
{
	try {
		std::unique_ptr<RefSlow> h(new RefSlow);
		Pire::Fsm fsm = ParseOne(pattern, options);
		h->sc = fsm.Compile<Pire::SlowScanner>();
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void* pire_ref_slow_load(const void* blob, size_t len)
{
	try {
		std::unique_ptr<RefSlow> h(new RefSlow);
		Pire::MemoryInput in(static_cast<const char*>(blob), len);
		h->sc.Load(&in);                                     // scanner_io.cpp:113-170
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void pire_ref_slow_free(void* h) { delete static_cast<RefSlow*>(h); }

size_t pire_ref_slow_save(void* h, void* buf, size_t cap)      // scanner_io.cpp:71-111
{
	std::ostringstream out;
	static_cast<RefSlow*>(h)->sc.Save(&out);
	const std::string s = out.str();
	if (buf && cap >= s.size())
		memcpy(buf, s.data(), s.size());
	return s.size();
}

size_t pire_ref_slow_size(void* h) { return static_cast<RefSlow*>(h)->sc.Size(); }
size_t pire_ref_slow_letters(void* h) { return static_cast<RefSlow*>(h)->sc.GetLettersCount(); }
int pire_ref_slow_empty(void* h) { return static_cast<RefSlow*>(h)->sc.Empty() ? 1 : 0; }

/*
 * Runner(sc).Begin().Run(ptr,len).End() with the SlowScanner specialisation of Run (slow.h:436-451).
 * outFinal[i] = Final(state) (slow.h:152-158).  outBits (nullable): the state SET as a bitset,
 * words = (Size()+31)/32 uint32 per string (the reference keeps a vector + bitset; the set is what is defined).
 */
int pire_ref_slow_run(void* h, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                      uint8_t* outFinal, uint32_t* outBits, int threads)
{
	try {
		const Pire::SlowScanner& sc = static_cast<RefSlow*>(h)->sc;
		const char* t = static_cast<const char*>(text);
		const size_t words = (sc.Size() + 31) / 32;
		auto range = [&](uint64_t lo, uint64_t hi) {
			for (uint64_t i = lo; i < hi; ++i) {
				Pire::SlowScanner::State st;
				sc.Initialize(st);
				if (flags & FLAG_BEGIN)
					Pire::Step(sc, st, Pire::BeginMark);
				Pire::Run(sc, st, t + offsets[i], t + offsets[i + 1]);
				if (flags & FLAG_END)
					Pire::Step(sc, st, Pire::EndMark);
				if (outFinal)
					outFinal[i] = sc.Final(st) ? 1 : 0;
				if (outBits) {
					uint32_t* w = outBits + i * words;
					memset(w, 0, words * 4);
					for (unsigned s : st.states)
						w[s / 32] |= 1u << (s % 32);
				}
			}
		};
		if (threads <= 1) {
			range(0, n);
		} else {
			std::vector<std::thread> pool;
			for (int k = 0; k < threads; ++k)
				pool.emplace_back(range, n * k / threads, n * (k + 1) / threads);
			for (auto& th : pool)
				th.join();
		}
		return 0;
	} catch (const std::exception& e) {
		SetErr(e.what());
		return -1;
	}
}

/* ---- Pire::SimpleScanner (scanners/simple.h): the dense-row, single-regexp scanner ------------------------- */

struct RefSimple {
	Pire::SimpleScanner sc;
	size_t base = 0, stride = 0;   // State <-> StateIndex through public API only (simple.h:73, 154-157)
	void Geometry()
	{
		stride = (Pire::MaxChar + 1) * sizeof(Pire::SimpleScanner::Transition);   // STATE_ROW_SIZE, simple.h:43
		Pire::SimpleScanner::State init;
		sc.Initialize(init);
		base = init - sc.StateIndex(init) * stride;   // == m_transitions + 1 slot (the tag precedes the row)
	}
};

void* pire_ref_simple_compile(const char* pattern, const char* options)
{
	try {
		std::unique_ptr<RefSimple> h(new RefSimple);
		Pire::Fsm fsm = ParseOne(pattern, options);
		h->sc = fsm.Compile<Pire::SimpleScanner>();     // simple.h:226-252
		h->Geometry();
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void* pire_ref_simple_empty()
{
	std::unique_ptr<RefSimple> h(new RefSimple);        // SimpleScanner() aliases Null(), simple.h:50
	h->Geometry();
	return h.release();
}

void* pire_ref_simple_load(const void* blob, size_t len)
{
	try {
		std::unique_ptr<RefSimple> h(new RefSimple);
		Pire::MemoryInput in(static_cast<const char*>(blob), len);
		h->sc.Load(&in);                                 // scanner_io.cpp:51-69
		h->Geometry();
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void pire_ref_simple_free(void* h) { delete static_cast<RefSimple*>(h); }

size_t pire_ref_simple_save(void* h, void* buf, size_t cap)       // scanner_io.cpp:35-49
{
	std::ostringstream out;
	static_cast<RefSimple*>(h)->sc.Save(&out);
	const std::string s = out.str();
	if (buf && cap >= s.size())
		memcpy(buf, s.data(), s.size());
	return s.size();
}

size_t pire_ref_simple_size(void* h) { return static_cast<RefSimple*>(h)->sc.Size(); }
int pire_ref_simple_empty_flag(void* h) { return static_cast<RefSimple*>(h)->sc.Empty() ? 1 : 0; }
size_t pire_ref_simple_regexps(void* h) { return static_cast<RefSimple*>(h)->sc.RegexpsCount(); }

uint32_t pire_ref_simple_initial(void* hh)
{
	RefSimple* h = static_cast<RefSimple*>(hh);
	Pire::SimpleScanner::State st;
	h->sc.Initialize(st);
	return uint32_t(h->sc.StateIndex(st));
}

uint32_t pire_ref_simple_next(void* hh, uint32_t idx, uint32_t ch)      // simple.h:76-81
{
	RefSimple* h = static_cast<RefSimple*>(hh);
	Pire::SimpleScanner::State st = h->base + size_t(idx) * h->stride;
	h->sc.Next(st, Pire::Char(ch));
	return uint32_t(h->sc.StateIndex(st));
}

int pire_ref_simple_final(void* hh, uint32_t idx)                        // simple.h:62
{
	RefSimple* h = static_cast<RefSimple*>(hh);
	return h->sc.Final(h->base + size_t(idx) * h->stride) ? 1 : 0;
}

/* Runner(sc).Begin().Run().End() per string (run.h:365-392); kind is ignored (one table form). */
int pire_ref_simple_run(void* hh, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                        const uint32_t* initIdx, uint32_t* outIdx, uint8_t* outFinal, int threads)
{
	RefSimple* h = static_cast<RefSimple*>(hh);
	const Pire::SimpleScanner& sc = h->sc;
	const char* t = static_cast<const char*>(text);
	auto range = [&](uint64_t lo, uint64_t hi) {
		for (uint64_t i = lo; i < hi; ++i) {
			Pire::SimpleScanner::State st;
			if (initIdx)
				st = h->base + size_t(initIdx[i]) * h->stride;
			else
				sc.Initialize(st);
			if (flags & FLAG_BEGIN)
				Pire::Step(sc, st, Pire::BeginMark);
			Pire::Run(sc, st, t + offsets[i], t + offsets[i + 1]);
			if (flags & FLAG_END)
				Pire::Step(sc, st, Pire::EndMark);
			if (outIdx)
				outIdx[i] = uint32_t(sc.StateIndex(st));
			if (outFinal)
				outFinal[i] = sc.Final(st) ? 1 : 0;
		}
	};
	if (threads <= 1) {
		range(0, n);
		return 0;
	}
	std::vector<std::thread> pool;     // sharding by string index is ours, the reference is single-threaded
	for (int k = 0; k < threads; ++k)
		pool.emplace_back(range, n * k / threads, n * (k + 1) / threads);
	for (auto& th : pool)
		th.join();
	return 0;
}

/* ---- Pire::HalfFinalScanner (scanners/half_final.h): a Scanner whose TakeAction counts, per regexp, the steps that
 * end in a state final for it.  Built as tests/count_ut.cpp:503-527 does. ------------------------------------- */

struct RefHalf {
	Pire::HalfFinalScanner sc;
};

/* tests/count_ut.cpp:36-44 (MkFsm): no Surround(); option 'u' = UTF-8 (the tests' default), else Latin1; 'i'. */
static Pire::Fsm MkCountFsm(const char* regexp, const char* options)
{
	Pire::Lexer lex;
	bool utf8 = false;
	for (const char* o = options ? options : ""; *o; ++o) {
		if (*o == 'u')
			utf8 = true;
		else if (*o == 'i')
			lex.AddFeature(Pire::Features::CaseInsensitive());
	}
	const Pire::Encoding& enc = utf8 ? Pire::Encodings::Utf8() : Pire::Encodings::Latin1();
	lex.SetEncoding(enc);
	Pire::TVector<Pire::wchar32> ucs4;
	enc.FromLocal(regexp, regexp + strlen(regexp), std::back_inserter(ucs4));
	lex.Assign(ucs4.begin(), ucs4.end());
	return lex.Parse();
}

/* mode: 0 MakeGreedyCounter(true), 1 MakeGreedyCounter(false), 2 MakeNonGreedyCounter(true,true),
 * 3 MakeNonGreedyCounter(true,false), 4 MakeNonGreedyCounter(false)   (count_ut.cpp:506-519),
 * 5 HalfFinalScanner(Fsm) = MakeScanner (half_final.h:38-46).  count > 1: glued left to right (count_ut.cpp:520-523). */
void* pire_ref_half_compile(const char* const* patterns, const int* modes, int count, const char* options)
{
	try {
		std::unique_ptr<RefHalf> h(new RefHalf);
		for (int i = 0; i < count; ++i) {
			const Pire::Fsm re = MkCountFsm(patterns[i], options);
			Pire::HalfFinalScanner one;
			if (modes[i] == 5) {
				one = Pire::HalfFinalScanner(re);
			} else {
				Pire::HalfFinalFsm fsm(re);
				switch (modes[i]) {
				case 0: fsm.MakeGreedyCounter(true); break;
				case 1: fsm.MakeGreedyCounter(false); break;
				case 2: fsm.MakeNonGreedyCounter(true, true); break;
				case 3: fsm.MakeNonGreedyCounter(true, false); break;
				case 4: fsm.MakeNonGreedyCounter(false); break;
				default: throw Pire::Error("unknown half-final mode");
				}
				one = Pire::HalfFinalScanner(fsm);
			}
			h->sc = i == 0 ? one : Pire::HalfFinalScanner::Glue(h->sc, one);      // half_final.h:196-198
			if (i && h->sc.Empty())
				throw Pire::Error("HalfFinalScanner::Glue failed (too many states)");
		}
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void* pire_ref_half_load(const void* blob, size_t len)
{
	try {
		std::unique_ptr<RefHalf> h(new RefHalf);
		Pire::MemoryInput in(static_cast<const char*>(blob), len);
		h->sc.Load(&in);                                  // Scanner::Load, multi.h:575-599 (inherited)
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void pire_ref_half_free(void* h) { delete static_cast<RefHalf*>(h); }

size_t pire_ref_half_save(void* h, void* buf, size_t cap)          // Scanner::Save, multi.h:557-573 (inherited)
{
	std::ostringstream out;
	static_cast<RefHalf*>(h)->sc.Save(&out);
	const std::string s = out.str();
	if (buf && cap >= s.size())
		memcpy(buf, s.data(), s.size());
	return s.size();
}

size_t pire_ref_half_size(void* h) { return static_cast<RefHalf*>(h)->sc.Size(); }
size_t pire_ref_half_regexps(void* h) { return static_cast<RefHalf*>(h)->sc.RegexpsCount(); }

/* tests/count_ut.cpp:54-63 (Run): Initialize, Step(BeginMark), Run, Step(EndMark); then Result(r) per regexp
 * (half_final.h:90-92), Final and StateIndex.  results: n * RegexpsCount() values. */
int pire_ref_half_run(void* hh, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                      uint32_t* outIdx, uint8_t* outFinal, uint64_t* results, int threads)
{
	const Pire::HalfFinalScanner& sc = static_cast<RefHalf*>(hh)->sc;
	const size_t R = sc.RegexpsCount();
	const char* t = static_cast<const char*>(text);
	auto range = [&](uint64_t lo, uint64_t hi) {
		for (uint64_t i = lo; i < hi; ++i) {
			Pire::HalfFinalScanner::State st;
			sc.Initialize(st);
			if (flags & FLAG_BEGIN)
				Pire::Step(sc, st, Pire::BeginMark);
			Pire::Run(sc, st, t + offsets[i], t + offsets[i + 1]);
			if (flags & FLAG_END)
				Pire::Step(sc, st, Pire::EndMark);
			if (outIdx)
				outIdx[i] = uint32_t(sc.StateIndex(st));
			if (outFinal)
				outFinal[i] = sc.Final(st) ? 1 : 0;
			if (results)
				for (size_t r = 0; r < R; ++r)
					results[i * R + r] = st.Result(r);
		}
	};
	if (threads <= 1) {
		range(0, n);
		return 0;
	}
	std::vector<std::thread> pool;     // sharding by string index is ours
	for (int k = 0; k < threads; ++k)
		pool.emplace_back(range, n * k / threads, n * (k + 1) / threads);
	for (auto& th : pool)
		th.join();
	return 0;
}

/* ---- Pire::CountingScanner / AdvancedCountingScanner (extra/count.h): count occurrences of `re` separated by `sep`.
 * LoadedScanner tables (scanners/loaded.h) whose transitions carry an action; TakeAction keeps per-regexp counters in
 * the state.  Built as tests/count_ut.cpp:64-93 builds them. ---------------------------------------------------- */

extern "C++" {
struct RefCount {
	int kind;                               // 0 CountingScanner, 1 AdvancedCountingScanner, 2 NoGlueLimitCountingScanner
	Pire::CountingScanner cs;
	Pire::AdvancedCountingScanner as;
	Pire::NoGlueLimitCountingScanner ns;
};

template <class Sc>
static void CountRun(const Sc& sc, const char* t, const uint64_t* offsets, uint64_t lo, uint64_t hi, uint32_t flags,
                     uint32_t* outIdx, uint64_t* results)
{
	const size_t R = sc.RegexpsCount();
	for (uint64_t i = lo; i < hi; ++i) {
		typename Sc::State st;
		sc.Initialize(st);                                           // count.h:127-133
		if (flags & FLAG_BEGIN)
			Pire::Step(sc, st, Pire::BeginMark);                     // tests/count_ut.cpp:54-63
		Pire::Run(sc, st, t + offsets[i], t + offsets[i + 1]);
		if (flags & FLAG_END)
			Pire::Step(sc, st, Pire::EndMark);
		if (outIdx)
			outIdx[i] = uint32_t(sc.StateIndex(st));
		if (results)
			for (size_t r = 0; r < R; ++r)
				results[i * R + r] = st.Result(int(r));              // count.h:206
	}
}
}  // extern "C++"

void* pire_ref_count_compile(int kind, const char* const* res, const char* const* seps, int count, const char* options)
{
	try {
		std::unique_ptr<RefCount> h(new RefCount);
		h->kind = kind;
		for (int i = 0; i < count; ++i) {
			const Pire::Fsm re = MkCountFsm(res[i], options), sep = MkCountFsm(seps[i], options);
			if (kind == 0) {
				Pire::CountingScanner one(re, sep);
				h->cs = i == 0 ? one : Pire::CountingScanner::Glue(h->cs, one);
				if (i && h->cs.Empty())
					throw Pire::Error("CountingScanner::Glue failed");
			} else if (kind == 1) {
				Pire::AdvancedCountingScanner one(re, sep);
				h->as = i == 0 ? one : Pire::AdvancedCountingScanner::Glue(h->as, one);
				if (i && h->as.Empty())
					throw Pire::Error("AdvancedCountingScanner::Glue failed");
			} else if (kind == 2) {
				Pire::NoGlueLimitCountingScanner one(re, sep);
				h->ns = i == 0 ? one : Pire::NoGlueLimitCountingScanner::Glue(h->ns, one);
				if (i && h->ns.Empty())
					throw Pire::Error("NoGlueLimitCountingScanner::Glue failed");
			} else {
				throw Pire::Error("unknown counting scanner kind");
			}
		}
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void* pire_ref_count_load(int kind, const void* blob, size_t len)
{
	try {
		std::unique_ptr<RefCount> h(new RefCount);
		h->kind = kind;
		Pire::MemoryInput in(static_cast<const char*>(blob), len);
		if (kind == 0)
			h->cs.Load(&in);                                         // LoadedScanner::Load, scanner_io.cpp:191-215
		else if (kind == 1)
			h->as.Load(&in);
		else
			h->ns.Load(&in);                                         // count.cpp:1020-1040
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void pire_ref_count_free(void* h) { delete static_cast<RefCount*>(h); }

size_t pire_ref_count_save(void* hh, void* buf, size_t cap)         // LoadedScanner::Save, scanner_io.cpp:172-189
{
	RefCount* h = static_cast<RefCount*>(hh);
	std::ostringstream out;
	if (h->kind == 0)
		h->cs.Save(&out);
	else if (h->kind == 1)
		h->as.Save(&out);
	else
		h->ns.Save(&out);                                            // count.cpp:1009-1018
	const std::string s = out.str();
	if (buf && cap >= s.size())
		memcpy(buf, s.data(), s.size());
	return s.size();
}

size_t pire_ref_count_size(void* hh)
{
	RefCount* h = static_cast<RefCount*>(hh);
	return h->kind == 0 ? h->cs.Size() : h->kind == 1 ? h->as.Size() : h->ns.Size();
}
size_t pire_ref_count_regexps(void* hh)
{
	RefCount* h = static_cast<RefCount*>(hh);
	return h->kind == 0 ? h->cs.RegexpsCount() : h->kind == 1 ? h->as.RegexpsCount() : h->ns.RegexpsCount();
}
size_t pire_ref_count_letters(void* hh)
{
	RefCount* h = static_cast<RefCount*>(hh);
	return h->kind == 0 ? h->cs.LettersCount() : h->kind == 1 ? h->as.LettersCount() : h->ns.LettersCount();
}

int pire_ref_count_run(void* hh, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                       uint32_t* outIdx, uint64_t* results, int threads)
{
	RefCount* h = static_cast<RefCount*>(hh);
	const char* t = static_cast<const char*>(text);
	auto range = [&](uint64_t lo, uint64_t hi) {
		if (h->kind == 0)
			CountRun(h->cs, t, offsets, lo, hi, flags, outIdx, results);
		else if (h->kind == 1)
			CountRun(h->as, t, offsets, lo, hi, flags, outIdx, results);
		else
			CountRun(h->ns, t, offsets, lo, hi, flags, outIdx, results);
	};
	if (threads <= 1) {
		range(0, n);
		return 0;
	}
	std::vector<std::thread> pool;     // sharding by string index is ours
	for (int k = 0; k < threads; ++k)
		pool.emplace_back(range, n * k / threads, n * (k + 1) / threads);
	for (auto& th : pool)
		th.join();
	return 0;
}

/* ---- Pire::CapturingScanner (extra/capture.h:49-162): one regexp, the substring matched by ONE pair of parentheses.
 * Built as tests/capture_ut.cpp:39-53 builds it. ---------------------------------------------------------------- */

extern "C++" {
struct RefCapture {
	Pire::CapturingScanner sc;
};
}

void* pire_ref_capture_compile(const char* pattern, int index, const char* options)
{
	try {
		std::unique_ptr<RefCapture> h(new RefCapture);
		Pire::Lexer lexer;
		lexer.Assign(pattern, pattern + strlen(pattern));
		for (const char* o = options ? options : ""; *o; ++o)
			if (*o == 'i')
				lexer.AddFeature(Pire::Features::CaseInsensitive());
		lexer.AddFeature(Pire::Features::Capture(size_t(index)));    // extra/capture.cpp:30-131
		Pire::Fsm fsm = lexer.Parse();
		fsm.Surround();
		fsm.Determine();
		h->sc = fsm.Compile<Pire::CapturingScanner>();
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void* pire_ref_capture_load(const void* blob, size_t len)
{
	try {
		std::unique_ptr<RefCapture> h(new RefCapture);
		Pire::MemoryInput in(static_cast<const char*>(blob), len);
		h->sc.Load(&in);                                             // LoadedScanner::Load, scanner_io.cpp:191-215
		return h.release();
	} catch (const std::exception& e) {
		SetErr(e.what());
		return nullptr;
	}
}

void pire_ref_capture_free(void* h) { delete static_cast<RefCapture*>(h); }

size_t pire_ref_capture_save(void* h, void* buf, size_t cap)
{
	std::ostringstream out;
	static_cast<RefCapture*>(h)->sc.Save(&out);
	const std::string s = out.str();
	if (buf && cap >= s.size())
		memcpy(buf, s.data(), s.size());
	return s.size();
}

size_t pire_ref_capture_size(void* h) { return static_cast<RefCapture*>(h)->sc.Size(); }

/* tests/capture_ut.cpp:75-83 (RunRegexp).  begin/end: State::Begin()/End() (1-based, counted from the BeginMark step,
 * capture.h:96-101, 108-115), -1 where npos; captured = State::Captured(). */
int pire_ref_capture_run(void* hh, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags, uint32_t* outIdx,
                         uint8_t* outFinal, uint8_t* outCaptured, int64_t* outBegin, int64_t* outEnd)
{
	const Pire::CapturingScanner& sc = static_cast<RefCapture*>(hh)->sc;
	const char* t = static_cast<const char*>(text);
	for (uint64_t i = 0; i < n; ++i) {
		Pire::CapturingScanner::State st;
		sc.Initialize(st);
		if (flags & FLAG_BEGIN)
			Pire::Step(sc, st, Pire::BeginMark);
		Pire::Run(sc, st, t + offsets[i], t + offsets[i + 1]);
		if (flags & FLAG_END)
			Pire::Step(sc, st, Pire::EndMark);
		if (outIdx)
			outIdx[i] = uint32_t(sc.StateIndex(st));
		if (outFinal)
			outFinal[i] = sc.Final(st) ? 1 : 0;
		if (outCaptured)
			outCaptured[i] = st.Captured() ? 1 : 0;
		if (outBegin)
			outBegin[i] = int64_t(st.Begin());     // npos == (size_t)-1 == -1
		if (outEnd)
			outEnd[i] = int64_t(st.End());
	}
	return 0;
}

} // extern "C"
