/*
 * TEST / BENCH INFRASTRUCTURE -- host mirror of the synthetic corpus generator.
 *
 * The benchmark corpus is generated ON the GPU by
 * pire_amd/csrc/corpus_gen.hip so that 4 GiB .. 32 GiB of text never cross PCIe.
 * This file restates the same counter-based generator in plain C so that the CPU
 * side (oracle, reference baseline, parity tests) can regenerate ANY string of
 * the corpus bit-identically from (seed, string index).  Both sides are integer
 * only; tests/test_corpus.py pins device == host.
 *
 * Definition (all arithmetic modulo 2^64):
 *   mix(x)   = splitmix64 finaliser
 *   word(s,w)= mix(seed + s * 0x9E3779B97F4A7C15 + (w+1) * 0xD1B54A32D192ED03)
 *   byte k of 8-byte word w of string s = 0x20 + ((word >> 8k) & 0xFF) * 95 >> 8   (printable ASCII 0x20..0x7E)
 *   plant: string s carries plant p = s mod (nplants+1) - 1 (p = -1: none).  A plant is a
 *          literal byte string (a witness for one pattern) copied over the random bytes at
 *          offset len-|w| ("at_tail", for $-anchored patterns) or at
 *          mix(seed ^ s ^ 0xA5A5A5A5) mod (len-|w|+1).  Plants longer than the string are skipped.
 */
#ifndef PIRE_CORPUS_H
#define PIRE_CORPUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORPUS_MAX_PLANTS 16
#define CORPUS_PLANT_BYTES 64

typedef struct {
	uint32_t nplants;
	uint32_t len[CORPUS_MAX_PLANTS];
	uint32_t at_tail[CORPUS_MAX_PLANTS];
	uint8_t  bytes[CORPUS_MAX_PLANTS][CORPUS_PLANT_BYTES];
} corpus_plants;

/* Fill out[0..len) with string number `s` of the corpus. */
void corpus_fill_string(uint64_t seed, uint64_t s, uint64_t len, const corpus_plants* plants, uint8_t* out);

/* Fill strings [first, first+count), each `len` bytes, at out + i*stride. */
void corpus_fill(uint64_t seed, uint64_t first, uint64_t count, uint64_t len, uint64_t stride,
                 const corpus_plants* plants, uint8_t* out, int threads);

#ifdef __cplusplus
}
#endif
#endif
