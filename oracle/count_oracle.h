/*
 * TEST INFRASTRUCTURE -- parity oracle for Pire::CountingScanner / Pire::AdvancedCountingScanner.  NOT part of the
 * product path.
 *
 * Plain-C restatement of Initialize + Begin() + Run() + End() over the counting scanners of pire/extra/count.h:
 * LoadedScanner tables (pire/scanners/loaded.h: u8 letter table, {u32 shift, u32 action} transitions, u8 tags) whose
 * TakeAction keeps, per regexp, a current and a total counter in the state.  Ingested from LoadedScanner::Save()
 * bytes (pire/scanner_io.cpp:172-189).
 *
 * Parity status: PINNED -- tests/test_counting.py checks it against the unmodified reference (oracle/_ref,
 * pire_ref_count_*) on the reference's own vectors (tests/count_ut.cpp:95-200) and on seeded random input.
 */
#ifndef COUNT_ORACLE_H
#define COUNT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_count oracle_count;

enum {
	ORACLE_COUNT_BASIC = 0,      /* CountingScanner */
	ORACLE_COUNT_ADVANCED = 1,   /* AdvancedCountingScanner */
	ORACLE_COUNT_NOGLUELIMIT = 2 /* NoGlueLimitCountingScanner: its own serialised form (type 5 + action lists), any
	                                number of regexps (count.h:330-504, count.cpp:1009-1040) */
};

/* LoadedScanner::Load, scanner_io.cpp:195-215 + Header::Validate, scanners/common.h:65-77. */
int oracle_count_load(const void* blob, size_t len, oracle_count** out, char* err, size_t errlen);
void oracle_count_free(oracle_count* sc);

uint32_t oracle_count_size(const oracle_count* sc);            /* Size(), loaded.h:112 */
uint32_t oracle_count_letters(const oracle_count* sc);         /* LettersCount(), loaded.h:118 */
uint32_t oracle_count_regexps(const oracle_count* sc);         /* m.regexpsCount, loaded.h:116 */
uint32_t oracle_count_initial_index(const oracle_count* sc);   /* StateIdx(m.initial), loaded.h:205-208 */
uint32_t oracle_count_letter(const oracle_count* sc, uint32_t ch);                    /* Translate, count.h:143-146 */
/* One transition: next state index and the action word (count.h:148-153). */
uint32_t oracle_count_next(const oracle_count* sc, uint32_t idx, uint32_t letter, uint32_t* action);

/* Per string: Initialize; Begin() if flags&1; Run; End() if flags&2; results[i*regexps + r] = State::Result(r). */
void oracle_count_run(const oracle_count* sc, int kind, const void* text, const uint64_t* offsets, uint64_t n,
                      uint32_t flags, uint32_t* out_idx, uint64_t* results);

/*
 * Pire::CapturingScanner (extra/capture.h:49-162), also a LoadedScanner table: actions BeginCapture = 1 /
 * EndCapture = 2 on transitions, Final from the tags.  Per string: Initialize; Begin() if flags&1; Run; End() if
 * flags&2 (tests/capture_ut.cpp:75-83); out_begin/out_end = State::Begin()/End() (-1 = npos), out_captured =
 * State::Captured().  Pinned by tests/capture_ut.cpp:93-153.
 */
void oracle_capture_run(const oracle_count* sc, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                        uint32_t* out_idx, uint8_t* out_final, uint8_t* out_captured, int64_t* out_begin,
                        int64_t* out_end);

#ifdef __cplusplus
}
#endif
#endif
