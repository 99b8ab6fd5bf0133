/*
 * TEST INFRASTRUCTURE -- the parity oracle.  NOT part of the product path.
 *
 * A plain-C restatement, on the CPU, of the one reference path this repository
 * accelerates: Pire::Runner(sc).Begin().Run(ptr,len).End() over a compiled
 * Pire::Scanner table (pire/run.h + pire/scanners/multi.h).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product library (pire_amd/csrc) never does.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *  (1) the known-answer vectors of the reference's own unit tests
 *      (tests/pire_ut.cpp, via the tests/golden fixtures), and
 *  (2) the unmodified reference library compiled from /root/reference
 *      (oracle/_ref/libpire_ref.so) on seeded random inputs: state index, Final
 *      flag and AcceptedRegexps lists, bit-exact.
 *
 * The scanner is ingested from the bytes of the reference's PUBLIC serialised
 * form, Scanner::Save() (multi.h:557-573), which is also what the GPU library
 * ingests.  A state is represented exactly as the reference stores it on disk:
 * the byte offset of its row from the start of the transitions array
 * (multi.h:564: `mc.initial -= m_transitions`).
 */
#ifndef PIRE_ORACLE_H
#define PIRE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scanner oracle_scanner;

enum { ORACLE_FLAG_BEGIN = 1, ORACLE_FLAG_END = 2 };

/* pire/defs.h:59-73 */
enum {
	ORACLE_EPSILON = 257,
	ORACLE_BEGIN_MARK = 258,
	ORACLE_END_MARK = 259,
	ORACLE_MAX_CHAR_UNALIGNED = 260,
	ORACLE_MAX_CHAR = 264
};

/* Scanner::Load (multi.h:575-599) + Header::Validate (scanners/common.h:65-77).
 * Returns 0 and a handle, or -1 with a message in err. The blob is copied. */
int oracle_scanner_load(const void* blob, size_t len, oracle_scanner** out, char* err, size_t errlen);
void oracle_scanner_free(oracle_scanner* sc);

uint32_t oracle_size(const oracle_scanner* sc);           /* Size()          multi.h:134 */
uint32_t oracle_letters_count(const oracle_scanner* sc);  /* LettersCount()  multi.h:140 */
uint32_t oracle_regexps_count(const oracle_scanner* sc);  /* RegexpsCount()  multi.h:139 */
int      oracle_empty(const oracle_scanner* sc);          /* Empty()         multi.h:135 */
uint32_t oracle_initial_index(const oracle_scanner* sc);  /* StateIndex(Initialize()) */
uint32_t oracle_row_stride(const oracle_scanner* sc);     /* RowSize()*sizeof(Transition), multi.h:347 */
uint32_t oracle_header_size(const oracle_scanner* sc);    /* HEADER_SIZE, multi.h:349 */

/* Translate() multi.h:163-166 minus HEADER_SIZE: the letter class of ch (0 <= ch < 264). */
uint32_t oracle_letter_class(const oracle_scanner* sc, uint32_t ch);

/* Step() run.h:50-57 on a state INDEX (StateIndex, multi.h:281-284). */
uint32_t oracle_next_index(const oracle_scanner* sc, uint32_t idx, uint32_t ch);

int oracle_final(const oracle_scanner* sc, uint32_t idx);  /* multi.h:143 */
int oracle_dead(const oracle_scanner* sc, uint32_t idx);   /* multi.h:147 */
/* AcceptedRegexps() multi.h:149-158: returns the count, writes up to cap ids. */
size_t oracle_accepted_regexps(const oracle_scanner* sc, uint32_t idx, uint64_t* out, size_t cap);

/*
 * For every string i in [0,n): Initialize (or init_idx[i]); Begin() if flags&1; Run over
 * text[offsets[i], offsets[i+1]); End() if flags&2; report StateIndex and Final.
 * threads > 1 shards the strings by index over pthreads (our driver; the reference is
 * single-threaded).  out arrays may be NULL.
 */
void oracle_run(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                uint32_t flags, const uint32_t* init_idx, uint32_t* out_idx, uint8_t* out_final,
                int threads);

/* counts[StateIndex] += 1 for the state each text byte's step ends in (working-set measurements; single-threaded) */
void oracle_visit_counts(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                         uint32_t flags, uint64_t* counts);

/*
 * The same walk, but following the reference's production control flow: DoRun's head / aligned
 * body / tail split (run.h:187-226) and the ExitMasks shortcut skipping of multi.h:938-1000 using
 * the portable (non-SSE) word compare of platform.h:216-253.  Exists to pin "the shortcut is
 * result-neutral" inside the oracle itself; must agree with oracle_run on every input.
 */
void oracle_run_shortcut(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                         uint32_t flags, const uint32_t* init_idx, uint32_t* out_idx, uint8_t* out_final);

/*
 * Pire::HalfFinalScanner (scanners/half_final.h): the same serialised table (it IS a Scanner, Save/Load are
 * inherited), but Initialize and every Step end with TakeAction, which counts, per regexp, the steps that end in a
 * state final for it (half_final.h:137-164).  results (nullable): n * RegexpsCount() values = State::Result(r).
 * Pinned by the reference's own vectors, tests/count_ut.cpp:541-550, 575.
 */
void oracle_run_half_final(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                           uint32_t flags, uint32_t* out_idx, uint8_t* out_final, uint64_t* results);

/* LongestSuffix / ShortestSuffix (run.h:313-362): suffix length or -1 (the text is walked backwards). */
void oracle_suffix(const oracle_scanner* sc, int longest, const void* text, const uint64_t* offsets,
                   uint64_t n, int through_end, int through_begin, int64_t* out_len);
/* LongestPrefix / ShortestPrefix (run.h:277-311): prefix length or -1. */
void oracle_prefix(const oracle_scanner* sc, int longest, const void* text, const uint64_t* offsets,
                   uint64_t n, int through_begin, int through_end, int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
