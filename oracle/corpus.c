/* TEST / BENCH INFRASTRUCTURE -- see corpus.h. */
#include "corpus.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t z)
{
	z ^= z >> 30;
	z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27;
	z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return z;
}

static inline uint64_t corpus_word(uint64_t seed, uint64_t s, uint64_t w)
{
	return mix64(seed + s * 0x9E3779B97F4A7C15ull + (w + 1) * 0xD1B54A32D192ED03ull);
}

void corpus_fill_string(uint64_t seed, uint64_t s, uint64_t len, const corpus_plants* plants, uint8_t* out)
{
	uint64_t i;
	for (i = 0; i < len; i += 8) {
		uint64_t x = corpus_word(seed, s, i / 8);
		unsigned k;
		for (k = 0; k < 8 && i + k < len; ++k)
			out[i + k] = (uint8_t)(0x20 + ((((x >> (8 * k)) & 0xFF) * 95) >> 8));
	}
	if (plants && plants->nplants) {
		uint64_t slot = s % (plants->nplants + 1);
		if (slot != 0) {
			uint32_t p = (uint32_t)(slot - 1);
			uint64_t wl = plants->len[p];
			if (wl <= len) {
				uint64_t off = plants->at_tail[p]
					? len - wl
					: mix64(seed ^ s ^ 0xA5A5A5A5ull) % (len - wl + 1);
				memcpy(out + off, plants->bytes[p], wl);
			}
		}
	}
}

typedef struct {
	uint64_t seed, first, lo, hi, len, stride;
	const corpus_plants* plants;
	uint8_t* out;
} fill_job;

static void* fill_range(void* arg)
{
	fill_job* j = (fill_job*)arg;
	uint64_t i;
	for (i = j->lo; i < j->hi; ++i)
		corpus_fill_string(j->seed, j->first + i, j->len, j->plants, j->out + i * j->stride);
	return NULL;
}

void corpus_fill(uint64_t seed, uint64_t first, uint64_t count, uint64_t len, uint64_t stride,
                 const corpus_plants* plants, uint8_t* out, int threads)
{
	fill_job base = { seed, first, 0, count, len, stride, plants, out };
	if (threads <= 1 || count < (uint64_t)threads) {
		fill_range(&base);
		return;
	}
	{
		pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
		fill_job* jobs = (fill_job*)malloc(sizeof(fill_job) * (size_t)threads);
		int t;
		for (t = 0; t < threads; ++t) {
			jobs[t] = base;
			jobs[t].lo = count * (uint64_t)t / (uint64_t)threads;
			jobs[t].hi = count * (uint64_t)(t + 1) / (uint64_t)threads;
			pthread_create(&tids[t], NULL, fill_range, &jobs[t]);
		}
		for (t = 0; t < threads; ++t)
			pthread_join(tids[t], NULL);
		free(tids);
		free(jobs);
	}
}
