/*
 * TEST INFRASTRUCTURE -- the parity oracle.  NOT part of the product path.
 * See pire_oracle.h for scope and pinning status.  Every function cites the
 * reference file:line (under /root/reference) whose behaviour it restates.
 */
#include "pire_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* scanners/common.h:44-63 */
typedef struct {
	uint32_t magic, version, ptr_size, max_word_size, type, hdr_size;
} ref_header;

#define REF_MAGIC 0x45524950u        /* "PIRE", common.h:52 */
#define REF_VERSION 7u               /* common.h:53 */
#define REF_VERSION_MACTIONS 6u      /* common.h:54 */
#define REF_TYPE_SCANNER 1u          /* ScannerIOTypes::Scanner, common.h:35 */

/* multi.h:315-323 (struct Locals), x86-64 layout = 48 bytes */
typedef struct {
	uint32_t states_count;
	uint32_t letters_count;
	uint32_t regexps_count;
	uint32_t pad0;
	uint64_t initial;             /* on disk: byte offset from m_transitions (multi.h:564) */
	uint32_t final_table_size;
	uint32_t pad1;
	uint64_t relocation_signature;    /* Relocatable::Signature = 1 (multi.h:56) */
	uint64_t shortcutting_signature;  /* ExitMasks<N>: 0x2000+N (multi.h:701); NoShortcuts: 0x1000 (828) */
} ref_locals;

struct oracle_scanner {
	ref_locals m;
	int empty;
	uint8_t* buf;                 /* the BufSize() bytes, 8-aligned copy */
	const uint16_t* letters;      /* m_letters[MaxChar], values include HEADER_SIZE (multi.h:375) */
	const uint64_t* final_tab;    /* m_final */
	const uint64_t* final_index;  /* m_finalIndex */
	const uint32_t* transitions;  /* m_transitions, Relocatable::Transition = ui32 (multi.h:61) */
	uint32_t header_size;         /* HEADER_SIZE in transitions (multi.h:349) */
	uint32_t row_size;            /* RowSize() in transitions (multi.h:347) */
	uint32_t flags_off;           /* byte offset of Common.Flags inside the row header */
	uint32_t mask_count;          /* ExitMaskCount */
};

static size_t align_up(size_t v, size_t b) { return (v + b - 1) & ~(b - 1); }

static int fail(char* err, size_t errlen, const char* msg)
{
	if (err && errlen)
		snprintf(err, errlen, "%s", msg);
	return -1;
}

int oracle_scanner_load(const void* blob, size_t len, oracle_scanner** out, char* err, size_t errlen)
{
	const uint8_t* p = (const uint8_t*)blob;
	ref_header h;
	oracle_scanner* sc;
	size_t pos = 0, bufsize;

	*out = NULL;
	/* ValidateHeader, common.h:107-114; Header::Validate, common.h:65-77 */
	if (len < sizeof(h))
		return fail(err, errlen, "EOF reached while loading scanner header");
	memcpy(&h, p, sizeof(h));
	pos = align_up(sizeof(h), 8);
	if (h.magic != REF_MAGIC || h.ptr_size != 8 || h.max_word_size != 16)
		return fail(err, errlen, "Serialized regexp incompatible with your system");
	if (h.version != REF_VERSION && h.version != REF_VERSION_MACTIONS)
		return fail(err, errlen, "You are trying to used an incompatible version of a serialized regexp");
	if (h.type != REF_TYPE_SCANNER)
		return fail(err, errlen, "Serialized regexp incompatible with your system");
	if (h.hdr_size != sizeof(ref_locals))
		return fail(err, errlen, "Serialized regexp incompatible with your system");

	sc = (oracle_scanner*)calloc(1, sizeof(*sc));
	if (!sc)
		return fail(err, errlen, "out of memory");
	/* LoadPodType(s, sc.m); AlignLoad -- multi.h:582-583 */
	if (len < pos + sizeof(ref_locals)) {
		free(sc);
		return fail(err, errlen, "EOF reached while loading scanner locals");
	}
	memcpy(&sc->m, p + pos, sizeof(ref_locals));
	pos += align_up(sizeof(ref_locals), 8);
	if (sc->m.relocation_signature != 1) {
		free(sc);
		return fail(err, errlen, "Type mismatch while mmapping Pire::Scanner");
	}
	/* multi.h:584-585: shortcutting type must be one this loader knows the row header of */
	if (sc->m.shortcutting_signature == 0x1000) {
		sc->mask_count = 0;
	} else if ((sc->m.shortcutting_signature & ~(uint64_t)0xFF) == 0x2000 && (sc->m.shortcutting_signature & 0xFF) != 0) {
		sc->mask_count = (uint32_t)(sc->m.shortcutting_signature & 0xFF);
	} else {
		free(sc);
		return fail(err, errlen, "This scanner has different shortcutting type");
	}
	/* bool empty, padded to 8 -- multi.h:586-588 */
	if (len < pos + 1) {
		free(sc);
		return fail(err, errlen, "EOF reached while loading scanner");
	}
	sc->empty = p[pos] != 0;
	pos += 8;

	/* ExtendedRowHeader: ExitMasksArray[MaskCount * 2 * (16/8)] size_t, then Common.Flags
	 * (multi.h:704-767); NoShortcuts: Common only (multi.h:831-841). */
	sc->flags_off = sc->mask_count * 4 * 8;
	sc->header_size = (sc->flags_off + 8) / 4;
	/* RowSize = AlignUp(letters + HEADER_SIZE, sizeof(MaxSizeWord)/sizeof(Transition)) -- multi.h:347 */
	sc->row_size = (uint32_t)align_up(sc->m.letters_count + sc->header_size, 16 / 4);

	if (sc->empty) {
		/* multi.h:590-591: aliases the Null() scanner = Fsm::MakeFalse() compiled: one state,
		 * never final.  We keep no table; accessors below special-case it. */
		*out = sc;
		return 0;
	}

	/* BufSize -- multi.h:297-305 */
	bufsize = align_up((size_t)ORACLE_MAX_CHAR * 2
	                   + (size_t)sc->m.final_table_size * 8
	                   + (size_t)sc->m.states_count * 8
	                   + (size_t)sc->row_size * sc->m.states_count * 4, 8);
	if (len < pos + bufsize) {
		free(sc);
		return fail(err, errlen, "EOF reached while loading scanner buffer");
	}
	if (posix_memalign((void**)&sc->buf, 16, bufsize ? bufsize : 16)) {
		free(sc);
		return fail(err, errlen, "out of memory");
	}
	memcpy(sc->buf, p + pos, bufsize);
	/* Markup -- multi.h:381-388 */
	sc->letters = (const uint16_t*)sc->buf;
	sc->final_tab = (const uint64_t*)(sc->letters + ORACLE_MAX_CHAR);
	sc->final_index = sc->final_tab + sc->m.final_table_size;
	sc->transitions = (const uint32_t*)(sc->final_index + sc->m.states_count);
	*out = sc;
	return 0;
}

void oracle_scanner_free(oracle_scanner* sc)
{
	if (sc) {
		free(sc->buf);
		free(sc);
	}
}

uint32_t oracle_size(const oracle_scanner* sc) { return sc->m.states_count; }
uint32_t oracle_letters_count(const oracle_scanner* sc) { return sc->m.letters_count; }
uint32_t oracle_regexps_count(const oracle_scanner* sc) { return sc->empty ? 0 : sc->m.regexps_count; }
int oracle_empty(const oracle_scanner* sc) { return sc->empty; }
uint32_t oracle_row_stride(const oracle_scanner* sc) { return sc->row_size * 4; }
uint32_t oracle_header_size(const oracle_scanner* sc) { return sc->header_size; }

/* A state is the byte offset of its row from m_transitions. */
static inline uint64_t row_stride(const oracle_scanner* sc) { return (uint64_t)sc->row_size * 4; }

/* StateIndex -- multi.h:281-284 */
static inline uint32_t state_index(const oracle_scanner* sc, uint64_t st) { return (uint32_t)(st / row_stride(sc)); }
/* IndexToState -- multi.h:447-450 */
static inline uint64_t index_to_state(const oracle_scanner* sc, uint32_t idx) { return (uint64_t)idx * row_stride(sc); }

uint32_t oracle_initial_index(const oracle_scanner* sc)
{
	return sc->empty ? 0 : state_index(sc, sc->m.initial);   /* Initialize, multi.h:161 */
}

uint32_t oracle_letter_class(const oracle_scanner* sc, uint32_t ch)
{
	if (sc->empty || ch >= ORACLE_MAX_CHAR)
		return 0;
	return (uint32_t)sc->letters[ch] - sc->header_size;
}

/* Next = NextTranslated(state, Translate(c)) -- multi.h:163-192; Relocatable::Go -- multi.h:65:
 * state + SignExtend((i32) transition). */
static inline uint64_t next_state(const oracle_scanner* sc, uint64_t st, uint32_t ch)
{
	uint32_t letter = sc->letters[ch];
	int32_t shift = (int32_t)sc->transitions[st / 4 + letter];
	return st + (int64_t)shift;
}

static inline uint64_t row_flags(const oracle_scanner* sc, uint64_t st)
{
	uint64_t f;
	memcpy(&f, (const uint8_t*)sc->transitions + st + sc->flags_off, 8);
	return f;
}

/* multi.h:718-732: every copy of mask i holds the same value; read the first. */
static inline uint64_t row_mask(const oracle_scanner* sc, uint64_t st, uint32_t i)
{
	uint64_t m;
	memcpy(&m, (const uint8_t*)sc->transitions + st + (size_t)i * 4 * 8, 8);
	return m;
}

uint32_t oracle_next_index(const oracle_scanner* sc, uint32_t idx, uint32_t ch)
{
	if (sc->empty)
		return 0;
	return state_index(sc, next_state(sc, index_to_state(sc, idx), ch));
}

int oracle_final(const oracle_scanner* sc, uint32_t idx)
{
	if (sc->empty)
		return 0;
	return (row_flags(sc, index_to_state(sc, idx)) & 1) != 0;   /* FinalFlag, multi.h:91,143 */
}

int oracle_dead(const oracle_scanner* sc, uint32_t idx)
{
	if (sc->empty)
		return 1;   /* the Null scanner's only state can never reach a final one */
	return (row_flags(sc, index_to_state(sc, idx)) & 2) != 0;   /* DeadFlag, multi.h:92,147 */
}

size_t oracle_accepted_regexps(const oracle_scanner* sc, uint32_t idx, uint64_t* out, size_t cap)
{
	size_t n = 0;
	const uint64_t* p;
	if (sc->empty)
		return 0;
	/* multi.h:149-158: walk m_final from m_finalIndex[idx] to the End (= (size_t)-1) sentinel */
	for (p = sc->final_tab + sc->final_index[idx]; *p != (uint64_t)-1; ++p, ++n)
		if (out && n < cap)
			out[n] = *p;
	return n;
}

/* ------------------------------------------------------------------ the walk */

/* The reference's byte-wise formulation of DoRun (run.h:248-266, the PIRE_DEBUG build) with
 * RunPred (always Continue, run.h:63-67): Step() for every byte in order. */
static inline uint64_t run_bytes(const oracle_scanner* sc, uint64_t st, const uint8_t* b, const uint8_t* e)
{
	for (; b != e; ++b)
		st = next_state(sc, st, *b);
	return st;
}

typedef struct {
	const oracle_scanner* sc;
	const uint8_t* text;
	const uint64_t* offsets;
	uint64_t lo, hi;
	uint32_t flags;
	const uint32_t* init_idx;
	uint32_t* out_idx;
	uint8_t* out_final;
} run_job;

static void* run_range(void* arg)
{
	run_job* j = (run_job*)arg;
	const oracle_scanner* sc = j->sc;
	uint64_t i;
	for (i = j->lo; i < j->hi; ++i) {
		uint64_t st;
		if (sc->empty) {
			/* Null scanner: single non-final state, every transition a self loop */
			if (j->out_idx) j->out_idx[i] = 0;
			if (j->out_final) j->out_final[i] = 0;
			continue;
		}
		/* RunHelper -- run.h:365-392 */
		st = j->init_idx ? index_to_state(sc, j->init_idx[i]) : sc->m.initial;
		if (j->flags & ORACLE_FLAG_BEGIN)
			st = next_state(sc, st, ORACLE_BEGIN_MARK);          /* Begin(), run.h:375 */
		st = run_bytes(sc, st, j->text + j->offsets[i], j->text + j->offsets[i + 1]);
		if (j->flags & ORACLE_FLAG_END)
			st = next_state(sc, st, ORACLE_END_MARK);            /* End(), run.h:376 */
		if (j->out_idx) j->out_idx[i] = state_index(sc, st);
		if (j->out_final) j->out_final[i] = (row_flags(sc, st) & 1) != 0;
	}
	return NULL;
}

void oracle_run(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                uint32_t flags, const uint32_t* init_idx, uint32_t* out_idx, uint8_t* out_final,
                int threads)
{
	run_job base = { sc, (const uint8_t*)text, offsets, 0, n, flags, init_idx, out_idx, out_final };
	if (threads <= 1 || n < (uint64_t)threads) {
		run_range(&base);
		return;
	}
	{
		pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
		run_job* jobs = (run_job*)malloc(sizeof(run_job) * (size_t)threads);
		int t;
		for (t = 0; t < threads; ++t) {
			jobs[t] = base;
			jobs[t].lo = n * (uint64_t)t / (uint64_t)threads;
			jobs[t].hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
			pthread_create(&tids[t], NULL, run_range, &jobs[t]);
		}
		for (t = 0; t < threads; ++t)
			pthread_join(tids[t], NULL);
		free(tids);
		free(jobs);
	}
}

/* Where a batch's walks spend their steps: counts[StateIndex] += 1 for the state every text byte's step ENDS in
 * (Step of run.h:50-57 on each byte of run.h:248-266's loop; the marks are not counted).  Measurement aid for the
 * working-set figures of bench.py / DESIGN.md (how many distinct states a corpus visits); single-threaded. */
void oracle_visit_counts(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                         uint32_t flags, uint64_t* counts)
{
	const uint8_t* t = (const uint8_t*)text;
	uint64_t i;
	for (i = 0; i < n; ++i) {
		uint64_t st = sc->m.initial;
		const uint8_t* p = t + offsets[i];
		const uint8_t* e = t + offsets[i + 1];
		if (flags & ORACLE_FLAG_BEGIN)
			st = next_state(sc, st, ORACLE_BEGIN_MARK);
		for (; p != e; ++p) {
			st = next_state(sc, st, *p);
			counts[state_index(sc, st)]++;
		}
	}
}

/* ------------------------------------------- HalfFinalScanner: the same table, counting TakeAction */

/* HalfFinalScanner::TakeAction, half_final.h:156-164: if the state is Final, every entry of its final list
 * (multiplicities included, BuildFinals 202-213) bumps MatchedRegexps[id]. */
static inline void half_take_action(const oracle_scanner* sc, uint64_t st, uint64_t* matched)
{
	if (row_flags(sc, st) & 1) {
		const uint64_t* it = sc->final_tab + sc->final_index[state_index(sc, st)];
		for (; *it != (uint64_t)-1; ++it)
			matched[*it]++;
	}
}

void oracle_run_half_final(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                           uint32_t flags, uint32_t* out_idx, uint8_t* out_final, uint64_t* results)
{
	const uint8_t* t = (const uint8_t*)text;
	const uint32_t R = oracle_regexps_count(sc);
	uint64_t i, k;
	for (i = 0; i < n; ++i) {
		uint64_t* matched = results ? results + i * R : NULL;
		uint64_t scratch[1];
		uint64_t st;
		if (sc->empty) {
			if (out_idx) out_idx[i] = 0;
			if (out_final) out_final[i] = 0;
			continue;
		}
		if (!matched) {
			/* counts not wanted: still walk with a throw-away array of the right size */
			matched = (uint64_t*)calloc(R ? R : 1, sizeof(uint64_t));
		} else {
			memset(matched, 0, sizeof(uint64_t) * R);     /* Initialize, half_final.h:138-143 */
		}
		(void)scratch;
		st = sc->m.initial;
		half_take_action(sc, st, matched);                /* Initialize ends with TakeAction(state, 0) */
		if (flags & ORACLE_FLAG_BEGIN) {                  /* Step = Next + TakeAction, run.h:50-57 */
			st = next_state(sc, st, ORACLE_BEGIN_MARK);
			half_take_action(sc, st, matched);
		}
		/* Run: the generic AlignedRunner (run.h:153-184) -- Step() per byte in order; the Scanner-only
		 * shortcutting runner (multi.h:911) does not apply to the derived class */
		for (k = offsets[i]; k < offsets[i + 1]; ++k) {
			st = next_state(sc, st, t[k]);
			half_take_action(sc, st, matched);
		}
		if (flags & ORACLE_FLAG_END) {
			st = next_state(sc, st, ORACLE_END_MARK);
			half_take_action(sc, st, matched);
		}
		if (out_idx) out_idx[i] = state_index(sc, st);
		if (out_final) out_final[i] = (row_flags(sc, st) & 1) != 0;
		if (!results)
			free(matched);
	}
}

/* ------------------------------------------- the walk with the production control flow */

#define NO_SHORTCUT_MASK 1u   /* multi.h:640 */
#define NO_EXIT_MASK 2u       /* multi.h:641 */

static inline int no_exit(const oracle_scanner* sc, uint64_t st)      /* multi.h:800-805 / 855-861 */
{
	return sc->mask_count ? row_mask(sc, st, 0) == NO_EXIT_MASK : 0;
}
static inline int no_shortcut(const oracle_scanner* sc, uint64_t st)  /* multi.h:807-812 / 863-869 */
{
	return sc->mask_count ? row_mask(sc, st, 0) == NO_SHORTCUT_MASK : 1;
}

/* BasicInstructionSet::CheckBytes -- platform.h:120-124, applied to both size_t halves of a
 * 16-byte Word; a non-zero result means some byte of the chunk equals the mask's byte. */
static inline uint64_t check_bytes(uint64_t mask, uint64_t chunk)
{
	uint64_t mc = chunk ^ mask;
	return (mc - 0x0101010101010101ull) & ~mc & 0x8080808080808080ull;
}

/* Shortcutting::Run -- multi.h:814-819 via MaskChecker (644-688): advance over 16-byte words
 * while no byte of the word equals any exit byte of the state. */
static const uint8_t* shortcut_run(const oracle_scanner* sc, uint64_t st, const uint8_t* head, const uint8_t* tail)
{
	for (; head != tail; head += 16) {
		uint64_t lo, hi, any = 0;
		uint32_t i;
		memcpy(&lo, head, 8);
		memcpy(&hi, head + 8, 8);
		for (i = 0; i < sc->mask_count; ++i) {
			uint64_t m = row_mask(sc, st, i);
			any |= check_bytes(m, lo) | check_bytes(m, hi);
		}
		if (any)
			break;
	}
	return head;
}

/* AlignedRunner<Scanner>::RunAligned with RunPred -- multi.h:938-1000.  begin/end are 8-aligned. */
static uint64_t run_aligned(const oracle_scanner* sc, uint64_t st, const uint8_t* begin, const uint8_t* end)
{
	const uint8_t* head = (const uint8_t*)align_up((size_t)begin, 16);
	const uint8_t* tail = (const uint8_t*)((size_t)end & ~(size_t)15);
	int noshort;

	for (; begin != head && begin != end; begin += 8)
		st = run_bytes(sc, st, begin, begin + 8);
	if (begin == end)
		return st;
	if (no_exit(sc, st))
		return st;

	noshort = no_shortcut(sc, st);
	for (;;) {
		while (noshort && head != tail) {
			st = run_bytes(sc, st, head, head + 16);   /* RunMultiChunk, multi.h:918-923 */
			head += 16;
			noshort = no_shortcut(sc, st);
		}
		if (head == tail)
			break;
		if (no_exit(sc, st))
			return st;
		head = shortcut_run(sc, st, head, tail);
		noshort = 1;
	}
	for (; tail != end; tail += 8)
		st = run_bytes(sc, st, tail, tail + 8);
	return st;
}

/* Impl::DoRun -- run.h:187-226 */
static uint64_t do_run(const oracle_scanner* sc, uint64_t st, const uint8_t* begin, const uint8_t* end)
{
	const uint8_t* head = (const uint8_t*)((size_t)begin & ~(size_t)7);
	const uint8_t* tail = (const uint8_t*)((size_t)end & ~(size_t)7);
	size_t head_size = (size_t)(head + 8 - begin);
	size_t tail_size = (size_t)(end - tail);

	if (head == tail)
		return run_bytes(sc, st, begin, end);                 /* SafeRunChunk, run.h:199-202 */
	if (begin != head) {
		st = run_bytes(sc, st, begin, begin + head_size);     /* RunChunk on the head, run.h:209-215 */
		head += 8;
	}
	st = run_aligned(sc, st, head, tail);
	if (tail_size)
		st = run_bytes(sc, st, tail, tail + tail_size);       /* SafeRunChunk on the tail, run.h:222-223 */
	return st;
}

void oracle_run_shortcut(const oracle_scanner* sc, const void* text, const uint64_t* offsets, uint64_t n,
                         uint32_t flags, const uint32_t* init_idx, uint32_t* out_idx, uint8_t* out_final)
{
	const uint8_t* t = (const uint8_t*)text;
	uint64_t i;
	for (i = 0; i < n; ++i) {
		uint64_t st;
		if (sc->empty) {
			if (out_idx) out_idx[i] = 0;
			if (out_final) out_final[i] = 0;
			continue;
		}
		st = init_idx ? index_to_state(sc, init_idx[i]) : sc->m.initial;
		if (flags & ORACLE_FLAG_BEGIN)
			st = next_state(sc, st, ORACLE_BEGIN_MARK);
		st = do_run(sc, st, t + offsets[i], t + offsets[i + 1]);
		if (flags & ORACLE_FLAG_END)
			st = next_state(sc, st, ORACLE_END_MARK);
		if (out_idx) out_idx[i] = state_index(sc, st);
		if (out_final) out_final[i] = (row_flags(sc, st) & 1) != 0;
	}
}

/* ------------------------------------------------------------------ prefixes */

void oracle_prefix(const oracle_scanner* sc, int longest, const void* text, const uint64_t* offsets,
                   uint64_t n, int through_begin, int through_end, int64_t* out_len)
{
	static const uint8_t k_empty[1] = {0};
	/* the reference reports "no prefix" as a null pointer; with a null base an empty match at position 0 would be
	 * indistinguishable from a miss, so empty batches get a real address */
	const uint8_t* t = text ? (const uint8_t*)text : k_empty;
	uint64_t i;
	for (i = 0; i < n; ++i) {
		const uint8_t* begin = t + offsets[i];
		const uint8_t* end = t + offsets[i + 1];
		const uint8_t* p;
		const uint8_t* pos = NULL;
		uint64_t st;
		int stopped = 0;
		if (sc->empty) {
			out_len[i] = -1;
			continue;
		}
		st = sc->m.initial;
		if (through_begin)
			st = next_state(sc, st, ORACLE_BEGIN_MARK);
		if (longest) {
			/* LongestPrefix -- run.h:277-292 with LongestPrefixPred (87-100) */
			if (row_flags(sc, st) & 1)
				pos = begin;
			for (p = begin; p != end; ++p) {
				uint64_t f;
				st = next_state(sc, st, *p);
				f = row_flags(sc, st);
				if (f & 1)
					pos = p + 1;
				if (f & 2) {
					stopped = 1;
					break;
				}
			}
			(void)stopped;
			if (through_end) {
				st = next_state(sc, st, ORACLE_END_MARK);
				if (row_flags(sc, st) & 1)
					pos = end;
			}
		} else {
			/* ShortestPrefix -- run.h:294-311 with ShortestPrefixPred (69-85) */
			if (row_flags(sc, st) & 1) {
				out_len[i] = 0;
				continue;
			}
			for (p = begin; p != end; ++p) {
				uint64_t f;
				st = next_state(sc, st, *p);
				f = row_flags(sc, st);
				if (f & 1) {
					pos = p + 1;
					break;
				}
				if (f & 2)
					break;
			}
			if (through_end) {
				st = next_state(sc, st, ORACLE_END_MARK);
				if ((row_flags(sc, st) & 1) && pos == NULL)
					pos = end;
			}
		}
		out_len[i] = pos ? (int64_t)(pos - begin) : -1;
	}
}

/* ------------------------------------------------------------------ suffixes */

/* LongestSuffix / ShortestSuffix (run.h:313-362): the text is walked BACKWARDS from its last byte (the scanner is
 * normally compiled from Fsm::Reverse()).  out_len = length of the suffix, i.e. the reference's return pointer is
 * (last byte) - out_len; -1 where the reference returns null. */
void oracle_suffix(const oracle_scanner* sc, int longest, const void* text, const uint64_t* offsets,
                   uint64_t n, int through_end, int through_begin, int64_t* out_len)
{
	static const uint8_t k_empty[2] = {0, 0};
	const uint8_t* t = text ? (const uint8_t*)text : k_empty + 1;
	uint64_t i;
	for (i = 0; i < n; ++i) {
		const uint8_t* rbegin0 = t + offsets[i + 1] - 1;   /* last byte */
		const uint8_t* rend = t + offsets[i] - 1;          /* one before the first */
		const uint8_t* rbegin = rbegin0;
		uint64_t st;
		if (sc->empty) {
			out_len[i] = -1;
			continue;
		}
		st = sc->m.initial;
		if (through_end)
			st = next_state(sc, st, ORACLE_END_MARK);
		if (longest) {
			/* run.h:315-342 */
			int have = 0;
			int64_t pos = 0;
			while (rbegin != rend && !(row_flags(sc, st) & 2)) {
				if (row_flags(sc, st) & 1) {
					have = 1;
					pos = rbegin0 - rbegin;
				}
				st = next_state(sc, st, *rbegin);
				--rbegin;
			}
			if (row_flags(sc, st) & 1) {
				have = 1;
				pos = rbegin0 - rbegin;
			}
			if (through_begin) {
				st = next_state(sc, st, ORACLE_BEGIN_MARK);
				if (row_flags(sc, st) & 1) {
					have = 1;
					pos = rbegin0 - rbegin;
				}
			}
			out_len[i] = have ? pos : -1;
		} else {
			/* run.h:345-361 */
			for (; rbegin != rend && !(row_flags(sc, st) & 1) && !(row_flags(sc, st) & 2); --rbegin)
				st = next_state(sc, st, *rbegin);
			if (through_begin)
				st = next_state(sc, st, ORACLE_BEGIN_MARK);
			out_len[i] = (row_flags(sc, st) & 1) ? (int64_t)(rbegin0 - rbegin) : -1;
		}
	}
}

