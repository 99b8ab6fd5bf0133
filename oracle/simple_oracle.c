/*
 * TEST INFRASTRUCTURE -- see simple_oracle.h.  Every function cites the reference file:line it restates.
 */
#include "simple_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_CHAR 264u                 /* defs.h:73 */
#define BEGIN_MARK 258u               /* defs.h:64 */
#define END_MARK 259u                 /* defs.h:65 */
#define ROW_SLOTS (MAX_CHAR + 1)      /* STATE_ROW_SIZE: all characters + 1 slot for the final tag, simple.h:43 */

typedef struct {
	uint32_t magic, version, ptr_size, max_word_size, type, hdr_size;   /* scanners/common.h:44-63 */
} ref_header;

typedef struct {
	uint64_t states_count;   /* simple.h:171-174 */
	uint64_t initial;        /* on disk: byte offset from m_transitions (scanner_io.cpp:39-40) */
} simple_locals;

struct oracle_simple {
	simple_locals m;
	int empty;
	uint64_t* transitions;   /* states_count rows of ROW_SLOTS size_t: [tag, shift for Char 0, ..., Char 263] */
};

static int fail(char* err, size_t errlen, const char* msg)
{
	if (err && errlen)
		snprintf(err, errlen, "%s", msg);
	return -1;
}

int oracle_simple_load(const void* blob, size_t len, oracle_simple** out, char* err, size_t errlen)
{
	const uint8_t* p = (const uint8_t*)blob;
	ref_header h;
	oracle_simple* sc;
	size_t pos, bufsize;

	*out = NULL;
	if (len < sizeof(h))
		return fail(err, errlen, "EOF reached while loading scanner header");
	memcpy(&h, p, sizeof(h));
	/* Header::Validate, common.h:65-77; type SimpleScanner = 2, common.h:37; hdrsize = sizeof(Locals) */
	if (h.magic != 0x45524950u || h.ptr_size != 8 || h.max_word_size != 16 || h.type != 2 || h.hdr_size != sizeof(simple_locals))
		return fail(err, errlen, "Serialized regexp incompatible with your system");
	if (h.version != 7 && h.version != 6)
		return fail(err, errlen, "You are trying to used an incompatible version of a serialized regexp");
	pos = 24;   /* AlignSave(sizeof(Header)), scanner_io.cpp:38 */
	if (len < pos + sizeof(simple_locals) + 8)
		return fail(err, errlen, "EOF reached while loading scanner locals");
	sc = (oracle_simple*)calloc(1, sizeof(*sc));
	if (!sc)
		return fail(err, errlen, "out of memory");
	memcpy(&sc->m, p + pos, sizeof(simple_locals));   /* scanner_io.cpp:55 */
	pos += sizeof(simple_locals);
	sc->empty = p[pos] != 0;                          /* scanner_io.cpp:57-59 */
	pos += 8;
	if (sc->empty) {
		/* scanner_io.cpp:60-61: aliases Null() = Fsm::MakeFalse() compiled (simple.h:187-191): its rows were
		 * memset to zero (simple.h:234) and it has no final state, so every state loops and none accepts. */
		*out = sc;
		return 0;
	}
	bufsize = (size_t)ROW_SLOTS * sc->m.states_count * 8;   /* BufSize, simple.h:160-163 */
	if (sc->m.states_count == 0 || len < pos + bufsize) {
		free(sc);
		return fail(err, errlen, "EOF reached while loading scanner buffer");
	}
	sc->transitions = (uint64_t*)malloc(bufsize);
	if (!sc->transitions) {
		free(sc);
		return fail(err, errlen, "out of memory");
	}
	memcpy(sc->transitions, p + pos, bufsize);          /* Markup, simple.h:208-211 */
	*out = sc;
	return 0;
}

void oracle_simple_free(oracle_simple* sc)
{
	if (sc) {
		free(sc->transitions);
		free(sc);
	}
}

uint32_t oracle_simple_size(const oracle_simple* sc) { return (uint32_t)sc->m.states_count; }
int oracle_simple_empty(const oracle_simple* sc) { return sc->empty; }

/* A state is the byte offset, from m_transitions, of slot 1 of its row (SetInitial, simple.h:221-225). */
static inline uint32_t state_index(uint64_t st) { return (uint32_t)(st / (ROW_SLOTS * 8)); }      /* simple.h:154-157 */
static inline uint64_t index_to_state(uint32_t idx) { return ((uint64_t)idx * ROW_SLOTS + 1) * 8; }

uint32_t oracle_simple_initial_index(const oracle_simple* sc) { return state_index(sc->m.initial); }

static inline uint64_t next_state(const oracle_simple* sc, uint64_t st, uint32_t ch)
{
	if (sc->empty)
		return st;                                         /* zeroed rows: shift 0 */
	return st + sc->transitions[st / 8 + ch];              /* state += ((Transition*)state)[c], simple.h:78-79 */
}

uint32_t oracle_simple_next_index(const oracle_simple* sc, uint32_t idx, uint32_t ch)
{
	return state_index(next_state(sc, index_to_state(idx), ch));
}

int oracle_simple_final(const oracle_simple* sc, uint32_t idx)
{
	if (sc->empty)
		return 0;
	return sc->transitions[(size_t)idx * ROW_SLOTS] != 0;  /* *(((Transition*)state) - 1) != 0, simple.h:62 */
}

void oracle_simple_run(const oracle_simple* sc, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                       const uint32_t* init_idx, uint32_t* out_idx, uint8_t* out_final)
{
	const uint8_t* t = (const uint8_t*)text;
	uint64_t i, k;
	for (i = 0; i < n; ++i) {
		uint64_t st = init_idx ? index_to_state(init_idx[i]) : sc->m.initial;   /* Initialize, simple.h:73 */
		if (flags & 1)
			st = next_state(sc, st, BEGIN_MARK);                                /* Begin(), run.h:375 */
		for (k = offsets[i]; k < offsets[i + 1]; ++k)                            /* Run, run.h:248-266 form */
			st = next_state(sc, st, t[k]);
		if (flags & 2)
			st = next_state(sc, st, END_MARK);                                  /* End(), run.h:376 */
		if (out_idx)
			out_idx[i] = state_index(st);
		if (out_final)
			out_final[i] = (uint8_t)oracle_simple_final(sc, state_index(st));
	}
}
