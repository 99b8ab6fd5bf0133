/*
 * TEST INFRASTRUCTURE -- plain-C restatement of Pire::SlowScanner (pire/scanners/slow.h), the NFA-simulating
 * scanner of BASELINE config 5b.  NOT part of the product path.  Pinned by tests/test_slow.py against the
 * unmodified reference (oracle/_ref) and the golden vectors generated from it.
 *
 * Ingests SlowScanner::Save() bytes (scanner_io.cpp:71-111).  A state is the SET of active NFA states
 * (slow.h:63-74 keeps a vector plus a bitset; the set is what Next/Final are defined on), held here as a bitset.
 */
#ifndef PIRE_SLOW_ORACLE_H
#define PIRE_SLOW_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_slow oracle_slow;

int oracle_slow_load(const void* blob, size_t len, oracle_slow** out, char* err, size_t errlen);
void oracle_slow_free(oracle_slow* sc);
uint32_t oracle_slow_size(const oracle_slow* sc);      /* GetSize()         slow.h:84 */
uint32_t oracle_slow_letters(const oracle_slow* sc);   /* GetLettersCount() slow.h:81 */
int oracle_slow_empty(const oracle_slow* sc);          /* Empty()           slow.h:85 */

/* Initialize; Begin() if flags&1; Run (slow.h:436-451); End() if flags&2.  out_final = Final() (slow.h:152-158);
 * out_bits (nullable): (Size()+31)/32 uint32 words per string, bit s = NFA state s active. */
void oracle_slow_run(const oracle_slow* sc, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                     uint8_t* out_final, uint32_t* out_bits);

#ifdef __cplusplus
}
#endif
#endif
