/* TEST INFRASTRUCTURE -- see slow_oracle.h.  Every function cites the reference file:line it restates. */
#include "slow_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct oracle_slow {
	uint64_t states, letters, start;   /* SlowScanner::Locals, slow.h:340-344 */
	int empty;
	uint64_t* letter_of;               /* m_letters[MaxChar] (size_t), slow.h:349 */
	uint8_t* finals;                   /* m_finals[states] (bool) */
	uint64_t* jump_pos;                /* m_jumpPos[states*letters + 1] */
	uint32_t* jumps;                   /* m_jumps[] */
	uint32_t words;
};

static size_t up8(size_t v) { return (v + 7) & ~(size_t)7; }

static int fail(char* err, size_t n, const char* msg)
{
	if (err && n)
		snprintf(err, n, "%s", msg);
	return -1;
}

/* SlowScanner::Load -- scanner_io.cpp:113-170; layout written by Save, scanner_io.cpp:71-111. */
int oracle_slow_load(const void* blob, size_t len, oracle_slow** out, char* err, size_t errlen)
{
	const uint8_t* p = (const uint8_t*)blob;
	uint32_t hdr[6];
	size_t pos = 0, njumps;
	oracle_slow* sc;
	*out = NULL;
	if (len < 24 + 24 + 8)
		return fail(err, errlen, "EOF reached while loading SlowScanner");
	memcpy(hdr, p, 24);
	/* Header::Validate, scanners/common.h:65-77; Type 3 = ScannerIOTypes::SlowScanner (common.h:37) */
	if (hdr[0] != 0x45524950u || hdr[2] != 8 || hdr[3] != 16 || (hdr[1] != 7 && hdr[1] != 6) || hdr[4] != 3 || hdr[5] != 24)
		return fail(err, errlen, "Serialized regexp incompatible with your system");
	pos = 24;
	sc = (oracle_slow*)calloc(1, sizeof(*sc));
	memcpy(&sc->states, p + pos, 8);
	memcpy(&sc->letters, p + pos + 8, 8);
	memcpy(&sc->start, p + pos + 16, 8);
	pos += 24;
	sc->empty = p[pos] != 0;
	pos += 8;
	if (sc->empty) {
		/* Null() = Fsm::MakeFalse() compiled (slow.h:425-429): never final */
		sc->states = 1;
		sc->letters = 1;
		sc->start = 0;
		sc->words = 1;
		*out = sc;
		return 0;
	}
	sc->words = (uint32_t)((sc->states + 31) / 32);
	if (len < pos + 264 * 8)
		goto eof;
	sc->letter_of = (uint64_t*)malloc(264 * 8);
	memcpy(sc->letter_of, p + pos, 264 * 8);
	pos += 264 * 8;
	if (len < pos + up8(sc->states))
		goto eof;
	sc->finals = (uint8_t*)malloc(sc->states);
	memcpy(sc->finals, p + pos, sc->states);
	pos += up8(sc->states);
	{
		size_t npos = sc->states * sc->letters + 1;
		if (len < pos + npos * 8)
			goto eof;
		sc->jump_pos = (uint64_t*)malloc(npos * 8);
		memcpy(sc->jump_pos, p + pos, npos * 8);
		pos += npos * 8;
		njumps = sc->jump_pos[npos - 1];
	}
	if (len < pos + up8(njumps * 4))
		goto eof;
	sc->jumps = (uint32_t*)malloc(njumps * 4 + 4);
	memcpy(sc->jumps, p + pos, njumps * 4);
	*out = sc;
	return 0;
eof:
	oracle_slow_free(sc);
	return fail(err, errlen, "EOF reached while loading SlowScanner");
}

void oracle_slow_free(oracle_slow* sc)
{
	if (!sc)
		return;
	free(sc->letter_of);
	free(sc->finals);
	free(sc->jump_pos);
	free(sc->jumps);
	free(sc);
}

uint32_t oracle_slow_size(const oracle_slow* sc) { return (uint32_t)sc->states; }
uint32_t oracle_slow_letters(const oracle_slow* sc) { return (uint32_t)sc->letters; }
int oracle_slow_empty(const oracle_slow* sc) { return sc->empty; }

/* NextTranslated(current, next, l) -- slow.h:103-130: next = union of the jump lists of every active state. */
static void next_set(const oracle_slow* sc, const uint32_t* cur, uint32_t* next, uint32_t ch)
{
	const uint64_t l = sc->letter_of[ch];            /* Translate, slow.h:98-101 */
	uint32_t w;
	memset(next, 0, sc->words * 4);
	for (w = 0; w < sc->words; ++w) {
		uint32_t bits = cur[w];
		while (bits) {
			const uint32_t s = w * 32 + (uint32_t)__builtin_ctz(bits);
			const uint64_t* pos = sc->jump_pos + (uint64_t)s * sc->letters + l;
			uint64_t k;
			bits &= bits - 1;
			for (k = pos[0]; k < pos[1]; ++k)
				next[sc->jumps[k] / 32] |= 1u << (sc->jumps[k] % 32);
		}
	}
}

void oracle_slow_run(const oracle_slow* sc, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                     uint8_t* out_final, uint32_t* out_bits)
{
	const uint8_t* t = (const uint8_t*)text;
	uint32_t* a = (uint32_t*)malloc(sc->words * 4);
	uint32_t* b = (uint32_t*)malloc(sc->words * 4);
	uint64_t i;
	for (i = 0; i < n; ++i) {
		uint32_t *cur = a, *nxt = b, *tmp, w;
		const uint8_t* p;
		int fin = 0;
		memset(cur, 0, sc->words * 4);
		if (sc->empty) {
			/* the Null scanner: one non-final state, no way out */
			cur[0] = 1;
		} else {
			cur[sc->start / 32] |= 1u << (sc->start % 32);             /* Initialize, slow.h:89-95 */
			if (flags & 1) { next_set(sc, cur, nxt, 258); tmp = cur; cur = nxt; nxt = tmp; }   /* Begin() */
			for (p = t + offsets[i]; p != t + offsets[i + 1]; ++p) {   /* Run<SlowScanner>, slow.h:436-451 */
				next_set(sc, cur, nxt, *p);
				tmp = cur; cur = nxt; nxt = tmp;
			}
			if (flags & 2) { next_set(sc, cur, nxt, 259); tmp = cur; cur = nxt; nxt = tmp; }   /* End() */
			for (w = 0; w < sc->words && !fin; ++w) {                  /* Final, slow.h:152-158 */
				uint32_t bits = cur[w];
				while (bits) {
					if (sc->finals[w * 32 + (uint32_t)__builtin_ctz(bits)]) { fin = 1; break; }
					bits &= bits - 1;
				}
			}
		}
		if (out_final)
			out_final[i] = (uint8_t)fin;
		if (out_bits)
			memcpy(out_bits + i * sc->words, cur, sc->words * 4);
	}
	free(a);
	free(b);
}
