/*
 * TEST INFRASTRUCTURE -- see count_oracle.h.  Every function cites the reference file:line it restates.
 */
#include "count_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_CHAR 264u            /* defs.h:73 */
#define BEGIN_MARK 258u
#define END_MARK 259u
#define MAX_RE_COUNT 16u         /* loaded.h:74 */
#define INCREMENT_MASK ((1u << MAX_RE_COUNT) - 1u)          /* loaded.h:223 */
#define RESET_MASK (INCREMENT_MASK << MAX_RE_COUNT)         /* loaded.h:224 */

typedef struct {
	uint32_t magic, version, ptr_size, max_word_size, type, hdr_size;   /* scanners/common.h:44-63 */
} ref_header;

typedef struct {
	uint32_t states_count, letters_count, regexps_count, pad;   /* loaded.h:228-233 */
	uint64_t initial;                                           /* on disk: byte offset from m_jumps */
} loaded_locals;

typedef struct {
	uint32_t shift;    /* (newState - oldState) * StateSize(), truncated to 32 bits (loaded.h:188-193) */
	uint32_t action;
} transition;

struct oracle_count {
	loaded_locals m;
	uint8_t letters[MAX_CHAR];
	transition* jumps;
	uint8_t* tags;           /* m_tags[states] (loaded.h:242) */
	uint32_t type;          /* 4 LoadedScanner, 5 NoGlueLimitCountingScanner (common.h:39-40) */
	uint32_t* actions;      /* type 5: Actions[0] = length, then per action: resets count, ids, increments count, ids */
};

/* CountingState, count.h:204-234 */
typedef struct {
	uint64_t state;                 /* byte offset of the row from m_jumps */
	uint32_t current[MAX_RE_COUNT];
	uint32_t total[MAX_RE_COUNT];
	uint64_t updated_mask;
} count_state;

static int fail(char* err, size_t errlen, const char* msg)
{
	if (err && errlen)
		snprintf(err, errlen, "%s", msg);
	return -1;
}

static size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

int oracle_count_load(const void* blob, size_t len, oracle_count** out, char* err, size_t errlen)
{
	const uint8_t* p = (const uint8_t*)blob;
	ref_header h;
	oracle_count* sc;
	size_t pos, njumps;

	*out = NULL;
	if (len < sizeof(h))
		return fail(err, errlen, "EOF reached while loading scanner header");
	memcpy(&h, p, sizeof(h));
	/* type LoadedScanner = 4 (common.h:39); hdrsize = sizeof(Locals) */
	if (h.magic != 0x45524950u || h.ptr_size != 8 || h.max_word_size != 16 || (h.type != 4 && h.type != 5) ||
	    h.hdr_size != sizeof(loaded_locals))
		return fail(err, errlen, "Serialized regexp incompatible with your system");
	if (h.version != 7 && h.version != 6)
		return fail(err, errlen, "You are trying to used an incompatible version of a serialized regexp");
	pos = 24;
	if (len < pos + sizeof(loaded_locals))
		return fail(err, errlen, "EOF reached while loading scanner locals");
	sc = (oracle_count*)calloc(1, sizeof(*sc));
	if (!sc)
		return fail(err, errlen, "out of memory");
	memcpy(&sc->m, p + pos, sizeof(loaded_locals));           /* scanner_io.cpp:202 */
	pos += sizeof(loaded_locals);
	njumps = (size_t)sc->m.states_count * sc->m.letters_count;
	sc->type = h.type;
	if (sc->m.states_count == 0 || sc->m.letters_count == 0 || (h.type == 4 && sc->m.regexps_count > MAX_RE_COUNT) ||
	    len < pos + MAX_CHAR + njumps * 8 + sc->m.states_count) {
		free(sc);
		return fail(err, errlen, "EOF reached while loading scanner buffer");
	}
	memcpy(sc->letters, p + pos, MAX_CHAR);                   /* scanner_io.cpp:206 */
	pos += align8(MAX_CHAR);
	sc->jumps = (transition*)malloc(njumps * sizeof(transition));
	if (!sc->jumps) {
		free(sc);
		return fail(err, errlen, "out of memory");
	}
	memcpy(sc->jumps, p + pos, njumps * sizeof(transition));   /* scanner_io.cpp:207 */
	/* version 6 carries an extra, ignored action array; the tags follow (scanner_io.cpp:208-212): neither is used
	 * by the counting scanners' Next/TakeAction */
	pos += njumps * 8;
	if (h.version == 6)
		pos += align8(njumps * 4);
	sc->tags = (uint8_t*)malloc(sc->m.states_count);
	memcpy(sc->tags, p + pos, sc->m.states_count);            /* scanner_io.cpp:212 */
	pos += align8(sc->m.states_count);
	if (h.type == 5) {
		/* NoGlueLimitCountingScanner::Load, count.cpp:1020-1035: u32 size (0 = no table), then size-1 more words */
		uint32_t size;
		if (len < pos + 4) {
			oracle_count_free(sc);
			return fail(err, errlen, "EOF reached while loading the action table");
		}
		memcpy(&size, p + pos, 4);
		if (size) {
			if (len < pos + (size_t)size * 4) {
				oracle_count_free(sc);
				return fail(err, errlen, "EOF reached while loading the action table");
			}
			sc->actions = (uint32_t*)malloc((size_t)size * 4);
			memcpy(sc->actions, p + pos, (size_t)size * 4);
		}
	}
	*out = sc;
	return 0;
}

void oracle_count_free(oracle_count* sc)
{
	if (sc) {
		free(sc->jumps);
		free(sc->tags);
		free(sc->actions);
		free(sc);
	}
}

uint32_t oracle_count_size(const oracle_count* sc) { return sc->m.states_count; }
uint32_t oracle_count_letters(const oracle_count* sc) { return sc->m.letters_count; }
uint32_t oracle_count_regexps(const oracle_count* sc) { return sc->m.regexps_count; }

static inline uint64_t state_size(const oracle_count* sc) { return (uint64_t)sc->m.letters_count * 8; }   /* loaded.h:171-174 */
static inline uint32_t state_idx(const oracle_count* sc, uint64_t st) { return (uint32_t)(st / state_size(sc)); }

uint32_t oracle_count_initial_index(const oracle_count* sc) { return state_idx(sc, sc->m.initial); }
uint32_t oracle_count_letter(const oracle_count* sc, uint32_t ch) { return ch < MAX_CHAR ? sc->letters[ch] : 0; }

uint32_t oracle_count_next(const oracle_count* sc, uint32_t idx, uint32_t letter, uint32_t* action)
{
	const transition x = sc->jumps[(size_t)idx * sc->m.letters_count + letter];
	const uint64_t st = (uint64_t)idx * state_size(sc) + (uint64_t)(int64_t)(int32_t)x.shift;   /* SignExtend, loaded.h:210 */
	if (action)
		*action = x.action;
	return state_idx(sc, st);
}

/* PerformIncrement, count.h:175-182 with IncrementPerformer<MAX_RE_COUNT>, count.h:48-70 */
static inline void perform_increment(count_state* s, uint32_t mask)
{
	if (mask) {
		uint32_t i;
		for (i = MAX_RE_COUNT; i >= 1; --i)
			if (mask & (1u << (i - 1)))
				++s->current[i - 1];
		s->updated_mask |= ((uint64_t)mask) << MAX_RE_COUNT;
	}
}

/* PerformReset, count.h:184-192 with ResetPerformer<MAX_RE_COUNT>, count.h:82-101 */
static inline void perform_reset(count_state* s, uint32_t mask)
{
	mask &= (uint32_t)s->updated_mask;
	if (mask) {
		uint32_t i;
		for (i = MAX_RE_COUNT; i >= 1; --i)
			if ((mask & (1u << (MAX_RE_COUNT + (i - 1)))) && s->current[i - 1]) {
				if (s->current[i - 1] > s->total[i - 1])
					s->total[i - 1] = s->current[i - 1];
				s->current[i - 1] = 0;
			}
		s->updated_mask &= (uint64_t)(uint32_t)~mask;
	}
}

static inline void take_action(int kind, count_state* s, uint32_t a)
{
	if (kind == ORACLE_COUNT_BASIC) {      /* CountingScanner::TakeActionImpl, count.h:251-257 */
		if (a & INCREMENT_MASK)
			perform_increment(s, a);
		if (a & RESET_MASK)
			perform_reset(s, a);
	} else {                               /* AdvancedCountingScanner::TakeActionImpl, count.h:287-295 */
		if (a & RESET_MASK)
			perform_reset(s, a);
		if (a & INCREMENT_MASK)
			perform_increment(s, a);
	}
}

/* Step = Next + TakeAction (run.h:50-57; count.h:148-158) */
static inline void step(const oracle_count* sc, int kind, count_state* s, uint32_t ch)
{
	const transition x = sc->jumps[s->state / 8 + sc->letters[ch]];
	s->state += (uint64_t)(int64_t)(int32_t)x.shift;
	take_action(kind, s, x.action);
}

/* NoGlueLimitCountingScanner::TakeActionImpl, count.h:404-437, on NoGlueLimitCountingState (count.h:306-325):
 * Reset(id): current = 0;  Increment(id): ++current, total = max(total, current).  Resets before increments. */
static inline void noglue_take_action(const oracle_count* sc, uint32_t* current, uint32_t* total, uint32_t a)
{
	if (!a)
		return;
	if (sc->actions) {
		const uint32_t* act = sc->actions + a;
		uint32_t n;
		for (n = *act++; n--;)
			current[*act++] = 0;
		for (n = *act++; n--;) {
			const uint32_t id = *act++;
			if (++current[id] > total[id])
				total[id] = current[id];
		}
	} else {                       /* one regexp, no table: the raw Increment/Reset action bits (count.h:429-436) */
		if (a & 2u)
			current[0] = 0;
		if (a & 1u)
			if (++current[0] > total[0])
				total[0] = current[0];
	}
}

static inline void noglue_step(const oracle_count* sc, uint64_t* st, uint32_t* current, uint32_t* total, uint32_t ch)
{
	const transition x = sc->jumps[*st / 8 + sc->letters[ch]];   /* Next, count.h:148-153 */
	*st += (uint64_t)(int64_t)(int32_t)x.shift;
	noglue_take_action(sc, current, total, x.action);
}

void oracle_count_run(const oracle_count* sc, int kind, const void* text, const uint64_t* offsets, uint64_t n,
                      uint32_t flags, uint32_t* out_idx, uint64_t* results)
{
	const uint8_t* t = (const uint8_t*)text;
	const uint32_t R = sc->m.regexps_count;
	uint64_t i, k;
	uint32_t r;
	if (kind == ORACLE_COUNT_NOGLUELIMIT) {
		uint32_t* current = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
		uint32_t* total = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
		for (i = 0; i < n; ++i) {
			uint64_t st = sc->m.initial;                                  /* Initialize, count.h:398-401 */
			memset(current, 0, sizeof(uint32_t) * (R ? R : 1));
			memset(total, 0, sizeof(uint32_t) * (R ? R : 1));
			if (flags & 1)
				noglue_step(sc, &st, current, total, BEGIN_MARK);
			for (k = offsets[i]; k < offsets[i + 1]; ++k)
				noglue_step(sc, &st, current, total, t[k]);
			if (flags & 2)
				noglue_step(sc, &st, current, total, END_MARK);
			if (out_idx)
				out_idx[i] = state_idx(sc, st);
			if (results)
				for (r = 0; r < R; ++r)
					results[i * R + r] = current[r] > total[r] ? current[r] : total[r];
		}
		free(current);
		free(total);
		return;
	}
	for (i = 0; i < n; ++i) {
		count_state s;
		memset(&s, 0, sizeof(s));           /* Initialize, count.h:127-133 */
		s.state = sc->m.initial;
		if (flags & 1)
			step(sc, kind, &s, BEGIN_MARK);
		for (k = offsets[i]; k < offsets[i + 1]; ++k)
			step(sc, kind, &s, t[k]);
		if (flags & 2)
			step(sc, kind, &s, END_MARK);
		if (out_idx)
			out_idx[i] = state_idx(sc, s.state);
		if (results)
			for (r = 0; r < R; ++r)           /* Result(i) = max(current, total), count.h:206 */
				results[i * R + r] = s.current[r] > s.total[r] ? s.current[r] : s.total[r];
	}
}

/* ------------------------------------------------------------------ CapturingScanner */

typedef struct {
	uint64_t state, begin, end, counter;   /* State, capture.h:59-87 */
} capture_state;

#define CAPTURE_NPOS (~(uint64_t)0)

/* Step = Next + TakeAction (run.h:50-57) */
static inline void capture_step(const oracle_count* sc, capture_state* s, uint32_t ch)
{
	/* NextTranslated, capture.h:109-116: move, count the step */
	const transition x = sc->jumps[s->state / 8 + sc->letters[ch]];
	const int captured = s->begin != CAPTURE_NPOS && s->end != CAPTURE_NPOS;
	s->state += (uint64_t)(int64_t)(int32_t)x.shift;
	++s->counter;
	/* TakeAction, capture.h:96-102 */
	if ((x.action & 1u) && !captured)
		s->begin = s->counter - 1;
	else if ((x.action & 2u) && !captured)
		s->end = s->counter - 1;
}

void oracle_capture_run(const oracle_count* sc, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                        uint32_t* out_idx, uint8_t* out_final, uint8_t* out_captured, int64_t* out_begin,
                        int64_t* out_end)
{
	const uint8_t* t = (const uint8_t*)text;
	uint64_t i, k;
	for (i = 0; i < n; ++i) {
		capture_state s = { sc->m.initial, CAPTURE_NPOS, CAPTURE_NPOS, 0 };   /* Initialize, capture.h:89-94 */
		if (flags & 1)
			capture_step(sc, &s, BEGIN_MARK);
		for (k = offsets[i]; k < offsets[i + 1]; ++k)
			capture_step(sc, &s, t[k]);
		if (flags & 2)
			capture_step(sc, &s, END_MARK);
		if (out_idx)
			out_idx[i] = state_idx(sc, s.state);
		if (out_final)
			out_final[i] = (sc->tags[state_idx(sc, s.state)] & 1u) != 0;   /* Final, capture.h:134; FinalFlag = 1 */
		if (out_captured)
			out_captured[i] = s.begin != CAPTURE_NPOS && s.end != CAPTURE_NPOS;   /* Captured(), capture.h:61 */
		if (out_begin)
			out_begin[i] = (int64_t)s.begin;
		if (out_end)
			out_end[i] = (int64_t)s.end;
	}
}
