/*
 * TEST INFRASTRUCTURE -- parity oracle for Pire::SimpleScanner.  NOT part of the product path.
 *
 * Plain-C restatement of Runner(sc).Begin().Run().End() over a Pire::SimpleScanner
 * (pire/scanners/simple.h): one regexp, dense rows of MaxChar + 1 size_t slots (a tag slot followed by one
 * byte-shift per Char), no letter classes, no Dead states.  Ingested from SimpleScanner::Save() bytes
 * (pire/scanner_io.cpp:35-49).
 *
 * Parity status: PINNED -- tests/test_simple.py checks it against the unmodified reference
 * (oracle/_ref, pire_ref_simple_*) on the reference's own unit-test strings and on seeded random input:
 * StateIndex and Final, bit-exact.
 */
#ifndef SIMPLE_ORACLE_H
#define SIMPLE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_simple oracle_simple;

/* SimpleScanner::Load, scanner_io.cpp:51-69 + Header::Validate, scanners/common.h:65-77. */
int oracle_simple_load(const void* blob, size_t len, oracle_simple** out, char* err, size_t errlen);
void oracle_simple_free(oracle_simple* sc);

uint32_t oracle_simple_size(const oracle_simple* sc);           /* Size(), simple.h:54 */
int oracle_simple_empty(const oracle_simple* sc);               /* Empty(), simple.h:55 */
uint32_t oracle_simple_initial_index(const oracle_simple* sc);  /* StateIndex(Initialize()), simple.h:73, 154-157 */
uint32_t oracle_simple_next_index(const oracle_simple* sc, uint32_t idx, uint32_t ch);   /* Next, simple.h:76-81 */
int oracle_simple_final(const oracle_simple* sc, uint32_t idx);                         /* Final, simple.h:62 */

/* Per string: Initialize (or init_idx[i]); Begin() if flags&1; Run; End() if flags&2 (run.h:365-392). */
void oracle_simple_run(const oracle_simple* sc, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                       const uint32_t* init_idx, uint32_t* out_idx, uint8_t* out_final);

#ifdef __cplusplus
}
#endif
#endif
