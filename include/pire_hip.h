/*
 * pire_hip.h -- C ABI of libpire_hip.so: the MI355X (gfx950) implementation of Pire's scan path.
 *
 * Scope: exactly one reference path --
 *     Pire::Runner(scanner).Begin().Run(ptr, len).End()            (pire/run.h:365-392)
 * i.e. the DFA walk  Step(BeginMark); Run(begin,end); Step(EndMark)  (run.h:50-57, 271-275) over a
 * compiled Pire::Scanner / Pire::NonrelocScanner table (pire/scanners/multi.h:87-554), batched over
 * N independent strings, one string per GPU lane.  Regex parsing, FSM construction, determinisation,
 * minimisation and Scanner::Glue stay in the reference library on the host; the hand-off between the
 * two is the reference's PUBLIC serialised form, Scanner::Save() (multi.h:557-573, 620-624).
 *
 * The reference has no FFI for this path: it is a header-only C++ template concept.  The binding a
 * Pire maintainer would add is therefore a C++ shim over this C ABI (include/pire_hip/batch_runner.hpp,
 * INTEGRATION.md).  All entry points take plain pointers and sizes; no C++ / torch types.
 *
 * Results are bit-exact with the reference: for every string, the StateIndex (multi.h:281-284) of the
 * state the reference would end in, and its Final flag (multi.h:143).
 *
 * Error model (the reference throws Pire::Error, pire/stub/stl.h:213-217): every call returns
 * PIRE_HIP_OK (0) or a negative code; pire_hip_last_error() returns the thread-local message.
 * There is NO CPU fallback: without a usable HIP device every run call fails with PIRE_HIP_ENODEVICE.
 */
#ifndef PIRE_HIP_H
#define PIRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIRE_HIP_ABI_VERSION 6

enum {
	PIRE_HIP_OK        =  0,
	PIRE_HIP_EINVAL    = -1,   /* bad argument                                                        */
	PIRE_HIP_EFORMAT   = -2,   /* blob rejected: same checks as Header::Validate, common.h:65-77,     */
	                           /* and Scanner::Load, multi.h:575-599                                  */
	PIRE_HIP_ENODEVICE = -3,   /* no HIP device / HIP runtime error (message has the hipError string) */
	PIRE_HIP_ENOMEM    = -4,
	PIRE_HIP_EUNSUPPORTED = -5,
	PIRE_HIP_ESELFTEST = -6    /* a kernel's first-use known-answer batch came back wrong (pire_hip_config.selftest):   */
	                           /* this build of the library must not be used on this device; no results were written   */
};

/* run flags */
enum {
	PIRE_HIP_RUN_BEGIN     = 1u << 0,   /* Step(BeginMark) before the text  -- RunHelper::Begin(), run.h:375 */
	PIRE_HIP_RUN_END       = 1u << 1,   /* Step(EndMark) after the text     -- RunHelper::End(),   run.h:376 */
	PIRE_HIP_RUN_ON_DEVICE = 1u << 2,   /* text/offsets/init/out pointers are DEVICE pointers; the call only */
	                                    /* enqueues work on `stream`. Without it they are HOST pointers and  */
	                                    /* the call copies in, runs, copies out and synchronises.            */
	                                    /* What can still make an ON_DEVICE call wait, and how to switch it   */
	                                    /* off (pire_hip_config): (1) pire_hip_run with DEVICE offsets and    */
	                                    /* n < 65 536 reads the first and last offset back (synchronises      */
	                                    /* `stream`) to spot few long strings -- no_offsets_peek = 1; (2) a   */
	                                    /* call that ends up in the segmented scan (few long strings of known */
	                                    /* length: strided records, HOST_OFFSETS, or the peek) synchronises   */
	                                    /* `stream` while it follows the chain -- no_segments = 1; (3) the    */
	                                    /* first call on a device uploads the table (hipMalloc + copies):     */
	                                    /* pire_hip_table_upload() beforehand.  With those three an ON_DEVICE */
	                                    /* call enqueues kernels and nothing else (it is then legal inside a  */
	                                    /* stream capture).  Automatic adaptation (auto_adapt) never runs     */
	                                    /* inside an ON_DEVICE call unless auto_adapt = 2.                    */
	                                    /* A graph captured from ON_DEVICE calls holds the pointers of the    */
	                                    /* table's device image: an AUTOMATIC adaptation keeps the images it  */
	                                    /* replaces alive until pire_hip_table_destroy() (a replay walks the  */
	                                    /* old ranking: same results); after the caller's OWN                 */
	                                    /* pire_hip_table_adapt() the old images are freed -- re-capture.     */
	                                    /* Device text is read in whole 128-byte aligned lines: the kernels   */
	                                    /* may load (never use) up to 127 bytes in front of the first and     */
	                                    /* behind the last byte of the text -- always inside the memory page  */
	                                    /* that holds that byte, so any device allocation will do.            */
	PIRE_HIP_RUN_GENERIC   = 1u << 3,   /* force the generic (offset-driven) kernel; testing/diagnostics     */
	PIRE_HIP_RUN_NO_PEEK   = 1u << 5,   /* with ON_DEVICE, pire_hip_run: this call never reads device offsets  */
	                                    /* back (what pire_hip_config.no_offsets_peek is for every call)       */
	PIRE_HIP_RUN_HOST_OFFSETS = 1u << 4 /* with ON_DEVICE, pire_hip_run only: `offsets` is a HOST pointer (the  */
	                                    /* text stays resident on the device, the caller knows where its        */
	                                    /* documents start).  The call copies the offsets and synchronises      */
	                                    /* `stream` before it returns; because the host then knows the lengths, */
	                                    /* few long strings get the segmented scan described at pire_hip_run.   */
};

typedef struct pire_hip_table pire_hip_table;

/* ---- library configuration ------------------------------------------------------------------------
 * The runtime knobs of the library, set through the ABI.  Process-wide; a call takes a snapshot
 * when it starts, so pire_hip_config_set() affects the calls that start after it.  0 = the default everywhere.
 * The environment variables of the same names in upper case with the prefix PIRE_HIP_ (PIRE_HIP_TILED_VARIANT,
 * PIRE_HIP_NO_SEGMENTS, ...) are read ONCE, when the library is loaded, and only seed this struct: no launch path
 * calls getenv().  `size` = sizeof(pire_hip_config) of the caller: the struct may grow at its end. */
typedef struct pire_hip_config {
	uint32_t size;
	/* fixed-length records (pire_hip_run_strided) */
	uint32_t tiled_variant;        /* 0 the shipped kernel; 23 dense rows with rotated columns (LDS bank = byte & 63; no  */
	                               /* gain measured, round 4); 2 without `nt`; 20 waves not kept in step; 22 the transpose */
	                               /* of the next tile hidden in the walk; 1 rows 260 bytes apart: A/B measurements, same  */
	                               /* results                                                                               */
	uint32_t checked;              /* 1: the checked kernel build, see pire_hip_table_check_failures()                 */
	/* tables (applies to tables created / glued afterwards) */
	uint32_t no_compact;           /* 1: no compact LDS tier behind the dense rows                                      */
	uint32_t prior_flat;           /* 1: an a-priori ranking that knows nothing (tests of adapt())                      */
	/* offset batches with actions (prefix searches, HalfFinal counting) */
	uint32_t ragged_act_always;    /* 1: never take the one-string-per-lane kernel for "quick" searches                 */
	uint32_t no_ragged_act;        /* 1: always take it                                                                 */
	/* few long strings: the segmented scan */
	uint32_t no_segments;          /* 1: never                                                                          */
	uint32_t segment_no_grid;      /* 1: every segment through the ragged kernel                                        */
	uint32_t segment_stats;        /* 1: print a host timeline per call (synchronises at every mark)                    */
	uint32_t segment_modes;        /* modes per call (default 6)                                                        */
	uint64_t segment_bytes;        /* bytes per segment; non-zero also FORCES the segmented scan for every eligible call */
	uint64_t segment_warmup;       /* warm-up bytes in front of a segment (default 256); SEGMENT_WARMUP_NONE = 0 bytes   */
	uint64_t segment_budget;       /* chain repair rounds (default 32); SEGMENT_BUDGET_NONE = 0 rounds                   */
	/* host-pointer mode */
	uint64_t host_chunk_bytes;     /* staging chunk size (default 256 MiB, minimum 4096)                                */
	uint32_t host_one_shot;        /* 1: one allocation + one copy per call instead of the pooled pipeline              */
	/* multi-device */
	uint32_t no_rccl;              /* 1: sum the per-device counters on the host                                        */
	/* SlowScanner */
	uint32_t slow_sets_in_memory;  /* 1: the wave-per-string form keeps its state sets in device memory                 */
	uint32_t slow_no_list;         /* 1: never the 16-slot list kernel: bitset kernel (<= 256 states) / wave per string   */
	/* adaptation of the dense-row ranking */
	uint32_t auto_adapt;           /* re-rank the LDS rows by itself when scans keep leaving them (see                   */
	                               /* pire_hip_table_adapt()): 0 default = in calls that synchronise anyway (host-pointer */
	                               /* forms, HOST_OFFSETS) right there -- the device is drained (hipDeviceSynchronize) --, */
	                               /* and in calls that only enqueue (ON_DEVICE) IN THE BACKGROUND: a worker thread copies */
	                               /* the counters on a stream of its own, re-ranks a copy of the table, uploads the new   */
	                               /* image, and a later launch boundary swaps it in -- such a call never waits for the    */
	                               /* device and stays legal inside a stream capture; 1 never; 2 at every launch boundary, */
	                               /* draining the device, ON_DEVICE calls included (such a call may then block and is not */
	                               /* capturable); 3 round 5's default: never inside a call that only enqueues             */
	uint32_t auto_adapt_min_traps; /* sampled trap count since the last ranking that triggers it (default 256)          */
	/* offset batches (pire_hip_run) */
	uint32_t ragged_variant;       /* 0 default: large offset batches (>= 160 MiB of text when the host knows the lengths, */
	                               /* >= 2^20 strings when the offsets are on the device) take the stream kernel (every    */
	                               /* lane walks a run of consecutive strings), smaller ones the ragged kernel (one string */
	                               /* per lane at a time); 1 always the ragged kernel; 2 the stream kernel from 256        */
	                               /* strings (tests) -- and, for tables on the class-indexed walk, the stream kernel on   */
	                               /* that walk, which 0 never takes (measured slower there; its image is  */
	                               /* built for tables uploaded while 2 is set).  Same results either way.                 */
	uint32_t host_staging;         /* device staging of the host-pointer forms of the prefix / suffix / half-final /   */
	                               /* counting / capture / slow entry points: 0 blocks cached per device between calls */
	                               /* (no allocation in steady state), 1 hipMalloc + hipFree per call (round 2),       */
	                               /* 2 the stream-ordered pool (hipMallocAsync): measurements                          */
	uint32_t no_offsets_peek;      /* 1: pire_hip_run with device offsets never reads offsets back (enqueue-only even for */
	                               /* small batches; few long strings then walk one per lane)                            */
	uint32_t segment_no_pair;      /* 1: the segmented scan never fuses two modes into one pass of the pair kernel       */
	uint32_t segment_no_product;   /* 1: two modes that are not functions of each other take the pair kernel, not a product */
	uint32_t segment_no_derive;    /* 1: the segmented scan walks every mode (none derived from mode 0's walk)           */
	uint32_t no_length_order;      /* 1: counting / SlowScanner kernels take strings in the caller's order, not by length */
	uint32_t capture_by_length;    /* 1: the one-string-per-lane capture kernels too take them by length (A/B: slower)   */
	uint32_t force_rccl;           /* 1: pire_hip_multi_create builds an RCCL communicator for ONE device as well (a       */
	                               /* one-rank all-reduce: exercises the RCCL path on a one-GPU box; default: host sum)   */
	uint32_t counting_variant;     /* counting / capturing scanners: 0 default = whole text lines per lane and entries     */
	                               /* that are LDS addresses (CountingRowKernel: rows indexed by the byte for tables of up */
	                               /* to 64 states, by the table's letters for any other whose rows fit the LDS; up to 8   */
	                               /* regexps) for batches that fill the GPU, else 16 bytes of text at a time and 16-bit   */
	                               /* entries / the 32-bit kernel; 1 always the latter; 2 the former whenever the table    */
	                               /* fits.  Same results either way.                                                      */
	uint32_t slow_stats;           /* 1: every SlowScanner call prints to stderr how many strings left the list kernel     */
	                               /* (synchronises the stream: measurements)                                              */
	uint32_t walk_variant;         /* fixed-length records of tables with more states than dense rows: 0 default = the     */
	                               /* class-indexed walk (every row of the first ~1 700 states of the ranking in LDS, the  */
	                               /* reference's two-lookup step, multi.h:169-192) once the share of the scans' steps     */
	                               /* outside the 255 dense rows passes 0.05 % (measured by adapt(); 5 % of the a-priori   */
	                               /* estimate before), else the dense rows; 1 always the dense rows; 2 always the         */
	                               /* class-indexed walk with one string per lane, 3 always with two strings per lane      */
	                               /* (twice the table loads on their way beyond the rows, half the waves); 0 takes two    */
	                               /* for batches that give every wave slot of the chip a task of 128 strings (2^19 on    */
	                               /* 256 CUs).  Same results either way.                                                  */
	uint32_t selftest;             /* the first time a table takes one of the kernels of pire_hip_run[_strided] (dense     */
	                               /* rows, class-indexed walk, one string per lane, stream) that kernel first scans a     */
	                               /* known-answer batch of 256 x 512 bytes -- text that walks this table's own states --  */
	                               /* and the library compares it with the host image's transitions (pire_hip_table_next): */
	                               /* a mismatch returns PIRE_HIP_ESELFTEST and nothing is written.  The kernels keep text */
	                               /* on its way in registers with hand-counted waits; this catches a build or a device    */
	                               /* on which that goes wrong where the build-time ISA audit (pire_hip_build_info) only   */
	                               /* argues that it cannot.  ~1 ms once per table and kernel, on a stream of its own      */
	                               /* (an ON_DEVICE call blocks for that long, once); skipped while `stream` is being      */
	                               /* captured.  0 default = on; 1 off; 2 on, with the expected answer of one string        */
	                               /* altered (tests of the failure path)                                                  */
	uint32_t zip_variant;          /* the class-indexed walk's LDS image: 0 default = ZIPPED (a row of their own only for   */
	                               /* <= 1 022 states; every other state of the tier 10 bytes: the row it is equal to       */
	                               /* except in <= 3 letters, and where those lead -- 8-10 000 states of a dictionary        */
	                               /* scanner in a CU's LDS instead of 2 200, two dependent LDS reads per byte instead of   */
	                               /* one) once adapt() has measured more than 0.4 % of the steps outside the plain rows     */
	                               /* and the zipped tier leaves less than 0.6 of that outside; 1 never; 2 whenever the     */
	                               /* table has more states than plain rows.  Read when a table is ranked (created,          */
	                               /* adapted).  Same results either way.                                                    */
} pire_hip_config;
#define PIRE_HIP_SEGMENT_WARMUP_NONE (~(uint64_t)0)
#define PIRE_HIP_SEGMENT_BUDGET_NONE (~(uint64_t)0)

/* What this library is: the compiler its kernels' ISA audits passed with and the units they looked at -- or that it is
 * not the audited product build.  The kernels that keep text on its way in registers rely on what hipcc emits for them
 * (no spill, no copy of a register a load still owes data to); `make` checks that for every build and does not link a
 * library that fails (pire_amd/csrc/Makefile, tools/audit/build_audit.py). */
const char* pire_hip_build_info(void);

/* Copies min(out->size, sizeof) bytes of the current configuration; out->size must be set by the caller. */
int pire_hip_config_get(pire_hip_config* out);
/* Replaces the configuration (fields beyond in->size keep their current values). */
int pire_hip_config_set(const pire_hip_config* in);

/* Geometry of an ingested scanner; mirrors the public getters of Pire::Scanner. */
typedef struct pire_hip_table_info {
	uint32_t abi_version;
	uint32_t states;          /* Scanner::Size()          multi.h:134 */
	uint32_t letters;         /* Scanner::LettersCount()  multi.h:140 */
	uint32_t regexps;         /* Scanner::RegexpsCount()  multi.h:139 */
	uint32_t initial;         /* StateIndex(Initialize()) multi.h:161, 281-284 */
	uint32_t empty;           /* Scanner::Empty()         multi.h:135 */
	uint32_t header_size;     /* HEADER_SIZE in transitions, multi.h:349 (18 for Scanner, 2 for ScannerNoMask) */
	uint32_t row_stride;      /* RowSize()*sizeof(Transition) of the RELOCATABLE form, multi.h:347 */
	uint32_t hot_states;      /* states resident in LDS as dense 256-column rows (device layout) */
	uint32_t lds_table_bytes; /* LDS bytes the table occupies per workgroup */
	uint64_t device_bytes;    /* HBM bytes of the device-side table */
	uint64_t ref_buf_size;    /* Scanner::BufSize()       multi.h:297-305 */
	uint64_t last_trap_samples; /* cold-state samples seen by the most recent pire_hip_table_adapt() */
	uint32_t adaptations;     /* how many times pire_hip_table_adapt() changed the LDS rows */
	uint32_t compact_states;  /* states (hot ones included) that also have a class-indexed u16 row in LDS: the
	                             exact re-walk of a chunk that left the dense rows stays in LDS for them */
	uint32_t scanner_type;    /* ScannerIOTypes of the ingested blob (scanners/common.h:34-40): 1 Scanner, 2 SimpleScanner */
	uint32_t zip_full_states; /* != 0: the wide walk's image is zipped (pire_hip_config.zip_variant) and this many of its   */
	                          /* wide_states have a row of their own                                                     */
	uint32_t wide_states;     /* states with a class-indexed row in the wide walk's LDS image (0: table fits the dense rows) */
	uint32_t wide_lds_bytes;  /* LDS bytes of that image per workgroup */
	float outside_dense_share;   /* share of the ranking's mass (scans seen by adapt(), else the a-priori byte model) on   */
	                             /* states without a dense row                                                             */
	float outside_wide_share;    /* ... without a row of the class-indexed walk.  Once that walk has run: the share of the */
	                             /* steps of the scans between the two most recent adapt() calls that it SAW outside its    */
	                             /* rows (of its visit samples, one lane per wave and 128-byte tile, those that found their */
	                             /* lane there) -- not a share of the ranking's mass, which only knows states some sample   */
	                             /* hit                                                                                     */
	uint32_t shares_measured;    /* 1: those shares come from visit counters                                               */
	float zip_outside_share;     /* the plan that chose between the two images (last ranking): share of the ranking's mass  */
	                             /* the zipped tier would leave outside (0: not planned) ...                               */
	uint64_t last_wide_trap_chunks; /* 16-byte wave-chunks (64 strings x 16 bytes) the class-indexed walk had to walk a    */
	                                /* second time because a lane left its rows, between the two most recent adapt() calls  */
	                                /* (exact, all devices)                                                                 */
	float wide_outside_chunk_share; /* ... as a share of the wave-chunks that walk was handed in that time (a wave that skips */
	                                /* the attempt on the rows alone counts the chunks in which a lane left them)           */
	float zip_plain_outside_share; /* ... and the plain rows                                                               */
} pire_hip_table_info;

/* ---- table life cycle -------------------------------------------------------------------------- */

/*
 * Ingest a scanner from the bytes written by Pire::Scanner::Save() / NonrelocScanner::Save()
 * (multi.h:557-573; a Nonreloc scanner serialises as Relocatable, multi.h:604-608) or by
 * Pire::SimpleScanner::Save() (scanner_io.cpp:35-49; dense rows without letter classes, one regexp, never Dead --
 * its columns are folded into letter classes here, results are identical).  Replaces
 * Scanner::Load (multi.h:575-599) / Scanner::Mmap (multi.h:244-279) for the GPU side.  Validates the
 * header like Header::Validate.  The blob is copied; the handle is immutable and may be shared between
 * host threads.  Works without a GPU (host-side parse only); the device image is uploaded on first run
 * or by pire_hip_table_upload().
 */
int pire_hip_table_create(const void* save_blob, size_t len, pire_hip_table** out);

/*
 * Scanner::Mmap / SimpleScanner::Mmap twin (multi.h:244-279, simple.h:120-151): ingest the scanner at the start
 * of a mapped image and report in *consumed (nullable) how many bytes it occupied -- the offset of the pointer
 * Mmap() returns -- so that images holding several scanners back to back can be walked.  The image is decoded,
 * not aliased: it may be unmapped as soon as the call returns.
 */
int pire_hip_table_mmap(const void* image, size_t size, pire_hip_table** out, size_t* consumed);

/* Map `path` read-only and ingest its first scanner: the deployment flow of samples/blacklist/blacklist.cpp:64-93
 * (compile once with Scanner::Save, mmap everywhere). */
int pire_hip_table_create_from_file(const char* path, pire_hip_table** out);

/*
 * Scanner::Glue(lhs, rhs, maxSize) (multi.h:1092-1103) on two ingested Pire::Scanner tables, host side, without the
 * reference library: the product automaton with the reference's own letter classes (glue.h:35-46, 123-127), the
 * reference's own breadth-first state numbering (determine.h:91-137) -- so StateIndex values are identical to those
 * of the scanner the reference would glue -- and its flags / AcceptedRegexps lists (rhs ids shifted by
 * lhs.RegexpsCount(), multi.h:1024-1043).  max_size 0 = the reference's default 80 000.  As in the reference an
 * empty lhs (rhs) returns a copy of rhs (lhs), and exceeding max_size yields an EMPTY scanner (info.empty = 1), not
 * an error.  The result is a table like any other: run it, adapt it, glue it further.
 */
int pire_hip_table_glue(const pire_hip_table* lhs, const pire_hip_table* rhs, size_t max_size, pire_hip_table** out);
/*
 * The same product, with the state discovery (Impl::Determine's loop, determine.h:100-122) done on the GPU: a
 * level-synchronous breadth-first search over state pairs whose new states are numbered by the position of their
 * first occurrence inside the level -- which is exactly the order in which the sequential reference loop meets them,
 * so the result is identical to pire_hip_table_glue() and to the scanner the reference would glue.  Needs a HIP
 * device (PIRE_HIP_ENODEVICE otherwise).  Worth it for large products (tens of thousands of states).
 */
int pire_hip_table_glue_gpu(const pire_hip_table* lhs, const pire_hip_table* rhs, size_t max_size, pire_hip_table** out);

/*
 * This table's own configuration: from now on every entry point that takes `t` runs under *cfg instead of the process-wide
 * configuration (pire_hip_config_set) -- kernel routing, walk / zip variants at its next ranking, the adaptation policy, the
 * first-use self-test --, on whatever thread it is called.  Two users of the library in one process need not agree on the
 * process-wide knobs (or touch them at all).  cfg == NULL: back to the process-wide configuration.  Fields beyond cfg->size take
 * the process-wide values of the moment of the call.  Waits for calls in flight on `t` (not for the device).
 * pire_hip_run_pair on two tables with configurations of their own: the first table's.
 */
int pire_hip_table_config_set(pire_hip_table* t, const pire_hip_config* cfg);
/* The configuration calls on `t` run under (its own, else the process-wide one); out->size as pire_hip_config_get. */
int pire_hip_table_config_get(const pire_hip_table* t, pire_hip_config* out);

/* Upload the device image to the CURRENT HIP device now (otherwise done lazily by the first run). */
int pire_hip_table_upload(pire_hip_table* t);

/*
 * Re-rank the LDS-resident dense rows from what the scans on this table actually visited (the kernels keep
 * sampled visit counters on the device).  A table is created with rows ranked by a byte model of "typical"
 * text; after a representative batch, adapt() promotes the states the data really spends its time in, so that
 * later batches stay on the one-LDS-gather-per-byte path.  Purely a performance call: results are bit-exact
 * with or without it.  Synchronises the device; must not run concurrently with scans on the same table.
 * *changed_rows (nullable) receives the number of rows that entered the dense set.
 */
int pire_hip_table_adapt(pire_hip_table* t, uint32_t* changed_rows);

/*
 * The checked build of the scan kernel (pire_hip_config.checked = 1; the analogue of the reference's
 * ValidateSkip, multi.h:925-934, which re-walks what the exit masks skipped): the wave-wide early-out on absorbing
 * states is only noted, the text is walked to its end all the same, and every lane whose state still moved after its
 * wave had been declared absorbing is counted.  *out receives the count since the last call (0 = the early-out was
 * result-neutral, as it must be) and the counter is cleared.  Synchronises the devices the table ran on.
 */
int pire_hip_table_check_failures(pire_hip_table* t, uint64_t* out);

void pire_hip_table_destroy(pire_hip_table* t);

int pire_hip_table_get_info(const pire_hip_table* t, pire_hip_table_info* out);
/* pire_hip_table_info has no size field and has grown at its end between ABI versions: a caller built against an older header
 * passes ITS sizeof and gets the fields it knows (min(size, sizeof) bytes are written) -- or checks pire_hip_abi_version(), what
 * the loaded library was built as, against its own PIRE_HIP_ABI_VERSION before calling pire_hip_table_get_info. */
int pire_hip_table_get_info_sized(const pire_hip_table* t, void* out, size_t size);
uint32_t pire_hip_abi_version(void);

/* ---- per-state queries on the host (no GPU involved) ------------------------------------------- */

/* Scanner::Final(state)  multi.h:143.   idx = StateIndex. Returns 0/1, or <0 on bad idx. */
int pire_hip_table_final(const pire_hip_table* t, uint32_t state_idx);
/* Scanner::Dead(state)   multi.h:147 */
int pire_hip_table_dead(const pire_hip_table* t, uint32_t state_idx);
/* Scanner::AcceptedRegexps(state) multi.h:149-158: (begin, count) describe an array owned by the table. */
int pire_hip_table_accepted_regexps(const pire_hip_table* t, uint32_t state_idx,
                                    const uint64_t** begin, size_t* count);
/* The letter class of ch (Translate(ch) - HEADER_SIZE, multi.h:163-166); ch < 264. */
int pire_hip_table_letter_class(const pire_hip_table* t, uint32_t ch);
/*
 * Scanner::Next (multi.h:189-192) as a TABLE ACCESSOR on state indices: the index reached from state_idx on
 * ch (ch < 260, ch != 257), or <0.  For inspecting an ingested table and for single Step()s on one host-side
 * state (RunHelper::Step, run.h:371); it is not, and must not be used as, a scan loop -- Run() is GPU only.
 */
int64_t pire_hip_table_next(const pire_hip_table* t, uint32_t state_idx, uint32_t ch);
/*
 * Device-layout introspection (tests): orig_of_perm[states] = reference state index of
 * each device ("perm") id; hot_rows[(hot_states+1)*256] = the dense LDS rows, entries are perm ids < hot_states
 * or hot_states (= trap: the lane leaves the LDS-resident set).  Either pointer may be NULL.
 */
int pire_hip_table_layout(const pire_hip_table* t, uint32_t* orig_of_perm, uint8_t* hot_rows);
/*
 * ... and of the class-indexed walk's LDS image (pire_hip_config.walk_variant; tables with more states than dense rows):
 * *wide_states device ids [0, wide_states) have a row, *pitch bytes apart, the first at LDS byte address *rows_offset;
 * rows[(wide_states + 1) * pitch / 2] (cap = its capacity in u16 entries; NULL: only the geometry): per row `letters`
 * entries = the device id of the target state (wide_states, the id of the last row -- the escape row, which leads to
 * itself -- for targets without a row), then the row's flags (1 Final, 2 Dead, 4 every transition a self loop).
 * wide_states == 0: no such image.
 */
int pire_hip_table_wide_layout(const pire_hip_table* t, uint16_t* rows, size_t cap, uint32_t* wide_states, uint32_t* pitch,
                               uint32_t* rows_offset);
/*
 * ... and of its ZIPPED form (pire_hip_config.zip_variant; pire_hip_table_wide_layout reports wide_states = 0 for such a table).
 * geometry[8] = { tier, full, pitch, rows_offset, headers_offset, exceptions_offset, image_end, 3 }: device ids [0, full) have a
 * row of their own (as above, `tier` = the escape state's id, row number `full` = the escape row), ids [full, tier) a header and
 * three exception targets; all zeros: the table's image is not zipped.  image[(image_end - rows_offset) / 2] (cap = its capacity
 * in u16 entries; NULL: only the geometry) = the bytes as they lie in LDS from rows_offset on:
 *     rows              (full + 1) x pitch bytes
 *     u32 header[tier + 1]  at headers_offset: bits 22..31 the row (device id < full, or `full`) this state's row is equal to except
 *                       in the letters at bits 1..7, 8..14, 15..21 (127 = none)
 *     u16 target[tier - full][3]  at exceptions_offset: where those letters lead (device id, `tier` = outside the tier)
 */
int pire_hip_table_zip_layout(const pire_hip_table* t, uint16_t* image, size_t cap, uint32_t geometry[8]);

/* ---- the hot path -------------------------------------------------------------------------------- */

/*
 * For each i in [0,n):   st = Initialize() or state #init_state_idx[i]       (RunHelper ctors, run.h:368-369)
 *                        if (flags & BEGIN) Step(st, BeginMark)
 *                        Run(st, text + offsets[i], text + offsets[i+1])    (Pire::Run, run.h:271-275)
 *                        if (flags & END)   Step(st, EndMark)
 *                        out_state_idx[i] = StateIndex(st);  out_final[i] = Final(st)
 * out_counts (nullable), uint64[regexps + 2], is ACCUMULATED into (caller zeroes it):
 *     [0] += number of strings whose end state is Final,  [1] += n,
 *     [2 + r] += number of strings whose end state lists regexp r in AcceptedRegexps().
 * init_state_idx, out_state_idx, out_final may each be NULL.  n == 0, zero-length strings and the empty
 * scanner are valid (tests/pire_ut.cpp:760-837).  Offsets are byte offsets into text, non-decreasing.
 * `stream` is a hipStream_t (NULL = default stream).
 *
 * Few long strings (strings of 8 KiB or more on average, too few of them to keep the lanes busy) are cut into
 * segments that are scanned in parallel from guessed start states; the chain of segments is then composed on the
 * device and only results computed from the true state are accepted, so the answers are the same
 * (pire_amd/csrc/segmented.hip: one 1 GiB string in 0.54 ms instead of 45 s).  The host has to know the lengths for
 * that: host pointers, pire_hip_run_strided, PIRE_HIP_RUN_HOST_OFFSETS -- and, for offsets that live on the device, a
 * batch of fewer than 65 536 strings is PEEKED at: its first and last offset are read back (which synchronises
 * `stream`), and all of them only if the segmented scan is then chosen (pire_hip_config.no_offsets_peek = 1 keeps such
 * calls enqueue-only).  A segmented call synchronises `stream` even with PIRE_HIP_RUN_ON_DEVICE.
 * PIRE_HIP_RUN_GENERIC (or pire_hip_config.no_segments) keeps one string per lane.
 */
int pire_hip_run(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n,
                 uint32_t flags, const uint32_t* init_state_idx,
                 uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts, void* stream);

/*
 * Same walk over fixed-length records: string i = text[i*stride, i*stride + len).  This is the layout the
 * tiled LDS-resident kernel is specialised for (len a multiple of 16, stride a multiple of 16, text
 * 16-byte aligned, device pointers); anything else is routed to the generic kernel with the same results.
 */
int pire_hip_run_strided(pire_hip_table* t, const void* text, uint64_t n, uint64_t len, uint64_t stride,
                         uint32_t flags, const uint32_t* init_state_idx,
                         uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts, void* stream);

/*
 * Pire::Step (run.h:50-57) on a device-resident array of state indices: state_idx[i] = Next(state_idx[i], ch),
 * ch < 260 (bytes, BeginMark 258, EndMark 259).  Device pointers only.
 */
int pire_hip_step(pire_hip_table* t, uint32_t* state_idx, uint64_t n, uint32_t ch, void* stream);

/*
 * Batched Runner over the table walked as a Pire::HalfFinalScanner (scanners/half_final.h:32-227).  A
 * HalfFinalScanner IS a Scanner (same Save() bytes, ingest it with pire_hip_table_create), but its Initialize and
 * every Step end with TakeAction, which counts, per regexp, the steps that end in a state final for it
 * (half_final.h:137-164): the number of -- possibly intersecting -- matches.  Per string i:
 *   out_results[i * RegexpsCount() + r] = State::Result(r)   (half_final.h:90-92; u32: a count is <= length + 3)
 *   out_state_idx / out_final (nullable)  = StateIndex / Final of the end state, as pire_hip_run.
 * flags: PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_ON_DEVICE (| PIRE_HIP_RUN_GENERIC: keep the
 * one-string-per-lane kernel; by default batches of >= 256 strings take the ragged kernel, which
 * re-walks exactly only the 16-byte chunks that touched a Final state).  PIRE_HIP_RUN_HOST_OFFSETS as for
 * pire_hip_run; few long strings of host-known length are counted segment-wise after the segmented scan has resolved
 * every segment's true start state.  Pinned by tests/count_ut.cpp:541-550, 575.
 * ON_DEVICE calls: what is said at PIRE_HIP_RUN_ON_DEVICE holds, with two additions.  Tables that count on the row kernel
 * (up to 8 regexps, rows that fit the LDS) keep a second image on the device: pire_hip_table_upload() uploads it with the
 * table; without that call the first pire_hip_run_half_final on a device allocates and copies it synchronously.  And a
 * call on the row kernel takes its per-call scratch (the length order of the strings, the list of strings whose counts
 * outgrow 16 bits) from the stream-ordered allocator (hipMallocAsync / hipFreeAsync on `stream`): enqueue-only, but
 * not something every capture mode accepts -- pire_hip_config.counting_variant = 1 keeps such calls off the row kernel.
 */
int pire_hip_run_half_final(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                            uint32_t* out_state_idx, uint8_t* out_final, uint32_t* out_results, void* stream);

/*
 * Pire::LongestPrefix / Pire::ShortestPrefix (run.h:277-311) for n strings: out_len[i] = length of the longest
 * (shortest) prefix of string i the scanner accepts, or -1 where the reference returns a null pointer.
 * through_begin / through_end as the reference's throughBeginMark / throughEndMark.  Scanning stops at the first
 * dead state (pire_ut.cpp:475-483).  flags: PIRE_HIP_RUN_ON_DEVICE, PIRE_HIP_RUN_GENERIC (as above).
 */
int pire_hip_prefix(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, int longest,
                    int through_begin, int through_end, uint32_t flags, int64_t* out_len, void* stream);

/*
 * Pire::Run(scanner1, scanner2, state1, state2, begin, end) (run.h:229-241) -- a Runner over
 * Pire::ScannerPair<Scanner1, Scanner2> (scanners/pair.h:33-94) -- for n strings: both scanners over the same text,
 *   out_state_idx1[i], out_state_idx2[i] = StateIndex of the two end states (pair.h:79-82),
 *   out_final[i] = Final(state1) || Final(state2)                            (pair.h:69-72).
 * Device pointers only (flags must carry PIRE_HIP_RUN_ON_DEVICE; BEGIN / END apply to both scanners).  Fixed-length
 * records (pire_hip_run_pair_strided; len a multiple of 256, stride of 16, text 16-byte aligned) take ONE pass with
 * both tables' dense rows in LDS: the text is read once and the two lookups of a byte overlap; anything else, and the
 * last n mod 64 records, takes two ordinary passes behind the same call.  Any output pointer may be NULL.
 */
int pire_hip_run_pair(pire_hip_table* t1, pire_hip_table* t2, const void* text, const uint64_t* offsets, uint64_t n,
                      uint32_t flags, uint32_t* out_state_idx1, uint32_t* out_state_idx2, uint8_t* out_final, void* stream);
int pire_hip_run_pair_strided(pire_hip_table* t1, pire_hip_table* t2, const void* text, uint64_t n, uint64_t len,
                              uint64_t stride, uint32_t flags, uint32_t* out_state_idx1, uint32_t* out_state_idx2,
                              uint8_t* out_final, void* stream);

/*
 * Pire::LongestSuffix / Pire::ShortestSuffix (run.h:313-362) for n strings: every string is walked BACKWARDS from its
 * last byte (the scanner is normally compiled from Fsm::Reverse(), pire_ut.cpp:283).  out_len[i] = length of the
 * longest (shortest) suffix accepted -- the reference returns the pointer (last byte) - out_len[i] -- or -1 where the
 * reference returns a null pointer.  through_end / through_begin as the reference's throughEndMark / throughBeginMark
 * (in that order: the walk starts at the end).  Scanning stops at the first dead state.  flags: PIRE_HIP_RUN_ON_DEVICE.
 * Pinned by tests/pire_ut.cpp:278-306 (PrefixSuffix) and 470-471 (ScanBoundaries on reversed texts).
 */
int pire_hip_suffix(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, int longest,
                    int through_end, int through_begin, uint32_t flags, int64_t* out_len, void* stream);

/* Name of the kernel the last run on this thread dispatched to ("tiled", "ragged", "generic", "ragged_prefix",
 * "prefix", "suffix", "pair_tiled", "ragged_half_final", "half_final", "segmented", "segmented+plain"); diagnostics. */
const char* pire_hip_last_kernel(void);
/* ",name,name,...": the kernels (names as pire_hip_last_kernel reports them) that have passed a first-use self-test in this
 * process so far (pire_hip_config.selftest); thread-local copy.  Diagnostics: tests/test_selftest.py holds every name the
 * library can emit against it. */
const char* pire_hip_selftested_kernels(void);
/* The instantiation behind it where there are several (e.g. "pirehip::ScanTiledKernel<16,2,nt,5>" for "tiled");
 * otherwise the same string as pire_hip_last_kernel(). */
const char* pire_hip_last_kernel_symbol(void);

/* Milliseconds the most recent kernel launched by this thread took, measured with hipEvents on the
 * launch stream when timing was enabled with pire_hip_set_timing(1).  Synchronises the stream. */
int   pire_hip_set_timing(int enabled);
float pire_hip_last_kernel_ms(void);

/* ---- SlowScanner (BASELINE config 5b) --------------------------------------------------------------- */
/*
 * Pire::SlowScanner (pire/scanners/slow.h:51-420): NFA simulation for patterns whose DFA would not fit anywhere
 * (e.g. /x.{40}$/).  Its state is the SET of active NFA states (slow.h:63-74), so results are the Final flag
 * (slow.h:152-158) and, optionally, the set itself as a bitset.  Ingests SlowScanner::Save() bytes
 * (pire/scanner_io.cpp:71-111).  Up to 256 NFA states the set of a string lives in the registers of ONE lane
 * (one string per lane); larger automata -- the reference has no limit, easy.h:155-161 falls back to this scanner
 * exactly when determinisation blows up -- keep the reference's sparse jump lists and give every string a whole wave
 * (the set is a bitset in LDS, or in device memory when 2 * states / 8 bytes do not fit there).
 */
typedef struct pire_hip_slow_table pire_hip_slow_table;

typedef struct pire_hip_slow_info {
	uint32_t states;      /* SlowScanner::Size()            slow.h:83-84 */
	uint32_t letters;     /* SlowScanner::GetLettersCount() slow.h:81    */
	uint32_t start;       /* m.start                        slow.h:343   */
	uint32_t words;       /* (states + 31) / 32: uint32 words of one state set */
	uint32_t empty;       /* SlowScanner::Empty()           slow.h:85    */
	uint32_t reserved;
	uint64_t mask_bytes;  /* size of the (state, letter) -> target-set matrix */
} pire_hip_slow_info;

int pire_hip_slow_table_create(const void* save_blob, size_t len, pire_hip_slow_table** out);
void pire_hip_slow_table_destroy(pire_hip_slow_table* t);
int pire_hip_slow_table_get_info(const pire_hip_slow_table* t, pire_hip_slow_info* out);

/*
 * For each string: Initialize (slow.h:89-95); Begin() if flags&BEGIN; Run (the SlowScanner specialisation,
 * slow.h:436-451); End() if flags&END.  out_final[i] = Final(state).  out_state_bits (nullable): `words` uint32 per
 * string, bit s set <=> NFA state s is in the final state set.  out_counts (nullable) uint64[2] accumulated:
 * [0] += strings ending Final, [1] += n.  Pointer/flag conventions as pire_hip_run.
 */
int pire_hip_slow_run(pire_hip_slow_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                      uint8_t* out_final, uint32_t* out_state_bits, uint64_t* out_counts, void* stream);
int pire_hip_slow_run_strided(pire_hip_slow_table* t, const void* text, uint64_t n, uint64_t len, uint64_t stride,
                              uint32_t flags, uint8_t* out_final, uint32_t* out_state_bits, uint64_t* out_counts,
                              void* stream);

/* ---- one process, several GPUs (BASELINE config C4) --------------------------- */
/*
 * The path shards by string: a string's walk depends on nothing but the table and its own bytes (run.h:271-275),
 * so GPU g scans its contiguous range of strings with no data-path exchange; the only collective is the sum of the
 * uint64[regexps+2] match counters -- one all-reduce over RCCL (xGMI) per batch, latency bound (80 bytes for 8
 * regexps).  The reference has no counterpart (a single-threaded CPU library); this is the C-ABI form of what
 * bench.py does with one process per GPU.  One table handle serves every device (it keeps an image per device).
 * RCCL is loaded at run time; without it (or when the communicator cannot be built) the per-device counters are
 * summed on the host -- pire_hip_multi_reduce_backend() says which ("rccl" or "host (...reason)").
 */
typedef struct pire_hip_multi pire_hip_multi;

/* One shard: fixed-length records resident on ITS device (pointers are device pointers of that device). */
typedef struct pire_hip_shard {
	const void*     text;
	uint64_t        n, len, stride;
	const uint32_t* init_state_idx;   /* nullable */
	uint32_t*       out_state_idx;    /* nullable */
	uint8_t*        out_final;        /* nullable */
} pire_hip_shard;

/* devices == NULL: HIP devices 0 .. ndev-1 (ndev <= 0: all of them).  Creates one stream and one counter buffer per
 * device and, for two or more distinct devices, the RCCL communicator (ncclCommInitAll). */
int  pire_hip_multi_create(const int* devices, int ndev, pire_hip_multi** out);
void pire_hip_multi_destroy(pire_hip_multi* m);
int  pire_hip_multi_device_count(const pire_hip_multi* m);
const char* pire_hip_multi_reduce_backend(const pire_hip_multi* m);

/*
 * Scan shards[g] on device g of `m` (all devices run concurrently, each on its own stream), then reduce the match
 * counters: out_counts (host memory, uint64[regexps+2], nullable) RECEIVES the totals over all shards (it is not
 * accumulated into).  flags: PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_GENERIC.  Returns when every
 * device has finished.  The caller's current device is restored.
 */
int pire_hip_multi_run_strided(pire_hip_multi* m, pire_hip_table* t, const pire_hip_shard* shards, uint32_t flags,
                               uint64_t* out_counts);
/*
 * Convenience form for a batch in HOST memory: strings [lo_g, hi_g) -- contiguous, balanced ranges -- are copied to
 * device g, scanned, and the per-string results copied back into the caller's arrays (PCIe-inclusive, like the
 * host-pointer mode of pire_hip_run).
 */
int pire_hip_multi_run_strided_host(pire_hip_multi* m, pire_hip_table* t, const void* text, uint64_t n, uint64_t len,
                                    uint64_t stride, uint32_t flags, const uint32_t* init_state_idx,
                                    uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts);

/* One shard of an OFFSET batch (what the reference's callers have: ragged lines, samples/pigrep/pigrep.cpp:38-45):
 * text and n + 1 byte offsets resident on ITS device, string i = text[offsets[i], offsets[i+1]). */
typedef struct pire_hip_shard_offsets {
	const void*     text;
	const uint64_t* offsets;
	uint64_t        n;
	const uint32_t* init_state_idx;   /* nullable */
	uint32_t*       out_state_idx;    /* nullable */
	uint8_t*        out_final;        /* nullable */
} pire_hip_shard_offsets;

/* pire_hip_multi_run_strided for offset batches: shards[g] lives on device g of the runner, all devices scan
 * concurrently (pire_hip_run on the runner's stream of each device), the counters are reduced as above. */
int pire_hip_multi_run(pire_hip_multi* m, pire_hip_table* t, const pire_hip_shard_offsets* shards, uint32_t flags,
                       uint64_t* out_counts);

/* Host-resident offset batch: cut into one contiguous run of whole strings per device, balanced by BYTES (a shard
 * boundary is the string boundary nearest to g / G of the text: ragged strings make string counts a poor measure of
 * work), staged through per-device buffers the runner keeps between calls (no allocation per call once they have
 * grown), scanned concurrently, results written back in string order.  offsets[0] need not be 0. */
int pire_hip_multi_run_host(pire_hip_multi* m, pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n,
                            uint32_t flags, const uint32_t* init_state_idx, uint32_t* out_state_idx, uint8_t* out_final,
                            uint64_t* out_counts);

/* First string of every shard the last pire_hip_multi_run_host / _run_strided_host call made: out[0 .. devices], the
 * last entry = n.  Diagnostics (tests, the per-device rates of a benchmark). */
int pire_hip_multi_last_split(const pire_hip_multi* m, uint64_t* out, int capacity);

/* ---- memory helpers ------------------------------------------------------------------------------- */
/*
 * Thin wrappers over the HIP runtime so that a caller of this ABI (and the header-only C++ shim) need not link HIP
 * itself.  Text placed in pire_hip_host_alloc() memory is pinned: the host-pointer mode of pire_hip_run then moves it
 * by plain DMA, overlapped chunk by chunk with the scan (pageable text is staged by the runtime; both are measured in
 * DESIGN.md).  Copies are asynchronous on `stream` (NULL = default stream); synchronise before reading the result.
 */
int  pire_hip_host_alloc(size_t bytes, void** out);
void pire_hip_host_free(void* p);
int  pire_hip_device_alloc(size_t bytes, void** out);
void pire_hip_device_free(void* p);
int  pire_hip_copy_to_device(void* dst_device, const void* src_host, size_t bytes, void* stream);
int  pire_hip_copy_to_host(void* dst_host, const void* src_device, size_t bytes, void* stream);
int  pire_hip_memset_device(void* dst_device, int value, size_t bytes, void* stream);
int  pire_hip_stream_synchronize(void* stream);

/* ---- errors ---------------------------------------------------------------------------------------- */
const char* pire_hip_last_error(void);
int pire_hip_device_count(void);

/* ---- benchmark utility (not part of the reference surface) ------------------------------------------ */
/* Witnesses planted into the synthetic corpus: string s carries plant (s mod (nplants+1)) - 1, none when that is
 * -1; at_tail: at the end of the string (for '$'-anchored patterns), otherwise at a generated offset. */
#define PIRE_HIP_CORPUS_MAX_PLANTS 16
#define PIRE_HIP_CORPUS_PLANT_BYTES 64
typedef struct pire_hip_corpus_plants {
	uint32_t nplants;
	uint32_t len[PIRE_HIP_CORPUS_MAX_PLANTS];
	uint32_t at_tail[PIRE_HIP_CORPUS_MAX_PLANTS];
	uint8_t  bytes[PIRE_HIP_CORPUS_MAX_PLANTS][PIRE_HIP_CORPUS_PLANT_BYTES];
} pire_hip_corpus_plants;

/*
 * Fill device memory with the synthetic corpus of SURVEY.md section 8d: string s occupies
 * out[(s-first)*stride, +len).  `plants` is a host pointer to a pire_hip_corpus_plants (the layout of
 * oracle/corpus.h, the generator's CPU twin) or NULL.  Bit-identical with oracle/corpus.c for the same (seed, s).
 */
int pire_hip_corpus_fill(void* device_out, uint64_t seed, uint64_t first, uint64_t count, uint64_t len,
                         uint64_t stride, const void* plants, void* stream);

/* ---- Pire::CountingScanner / Pire::AdvancedCountingScanner (extra/count.h) ------------------------- */

/*
 * The counting scanners of pire/extra/count.h: "count the occurrences of `re` separated by `sep`", up to 16 regexps
 * glued into one scanner.  They are LoadedScanner tables (scanners/loaded.h: transitions carry an Action) whose
 * TakeAction keeps a current and a total counter per regexp in the state (count.h:119-234).  Both classes serialise
 * through LoadedScanner::Save (scanner_io.cpp:172-189) with the same header, so the caller says which TakeAction
 * the table was built for: PIRE_HIP_COUNTING_BASIC = CountingScanner (increment, then reset; count.h:251-257),
 * PIRE_HIP_COUNTING_ADVANCED = AdvancedCountingScanner (reset, then increment; count.h:287-295).
 * PIRE_HIP_COUNTING_NOGLUELIMIT = NoGlueLimitCountingScanner, whose blob also carries the action lists.
 * Pinned by tests/count_ut.cpp:95-200.
 */
typedef struct pire_hip_counting_table pire_hip_counting_table;

#define PIRE_HIP_COUNTING_BASIC 0
#define PIRE_HIP_COUNTING_ADVANCED 1
#define PIRE_HIP_COUNTING_NOGLUELIMIT 2   /* NoGlueLimitCountingScanner: its own Save() form (count.cpp:1009-1018), any
                                            number of regexps; resets before increments (count.h:404-437) */

typedef struct pire_hip_counting_info {
	uint32_t states;    /* Size(), loaded.h:112 */
	uint32_t letters;   /* LettersCount(), loaded.h:118 */
	uint32_t regexps;   /* RegexpsCount(), loaded.h:116 */
	uint32_t initial;   /* StateIndex(Initialize()), count.h:127-133, 171 */
} pire_hip_counting_info;

int pire_hip_counting_table_create(const void* save_blob, size_t len, pire_hip_counting_table** out);
void pire_hip_counting_table_destroy(pire_hip_counting_table* t);
int pire_hip_counting_table_get_info(const pire_hip_counting_table* t, pire_hip_counting_info* out);
/* Which device forms the table has (performance only; every form gives the same results): out[0] = counter registers of
 * the 16-bit-entry form (0: none), out[1] = its LDS bytes; out[2] = 1 if the byte-indexed row form applies (<= 64 states),
 * out[3] = its LDS bytes; out[4] = counter registers of the letter-indexed row form (0: none), out[5] = its LDS bytes,
 * out[6] = its distinct actions; out[7] = 0. */
int pire_hip_counting_table_forms(const pire_hip_counting_table* t, uint32_t out[8]);

/*
 * Per string i: Initialize; Begin() if flags & BEGIN; Run; End() if flags & END (tests/count_ut.cpp:54-63), then
 *   out_results[i * regexps + r] = State::Result(r) = max(current[r], total[r])   (count.h:206)
 *   out_state_idx[i] (nullable)  = StateIndex of the end state                     (count.h:171)
 * flags: PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_ON_DEVICE.
 */
int pire_hip_counting_run(pire_hip_counting_table* t, int kind, const void* text, const uint64_t* offsets, uint64_t n,
                          uint32_t flags, uint32_t* out_state_idx, uint32_t* out_results, void* stream);

/* ---- Pire::CapturingScanner (extra/capture.h) ------------------------------------------------------ */

/*
 * The capturing scanner of pire/extra/capture.h:49-162: one regexp, the substring matched by ONE pair of parentheses
 * (built with Features::Capture(i), extra/capture.cpp).  It is a LoadedScanner table too (create it with
 * pire_hip_counting_table_create): transitions carry BeginCapture = 1 / EndCapture = 2, TakeAction records the step
 * counter (capture.h:96-116), Final comes from the state tags (capture.h:134).  Per string i, after
 * Initialize; Begin(); Run(); End() (tests/capture_ut.cpp:75-83):
 *   out_begin[i], out_end[i] = State::Begin(), State::End(): 1-based byte positions (the BeginMark step is counted;
 *                              the captured text is [begin - 1, end - 1), capture_ut.cpp:85-91), -1 where unset;
 *                              State::Captured() == (out_begin[i] >= 0 && out_end[i] >= 0)
 *   out_final[i], out_state_idx[i] (nullable) = Final / StateIndex of the end state.
 */
int pire_hip_capture_run(pire_hip_counting_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                         uint32_t* out_state_idx, uint8_t* out_final, int64_t* out_begin, int64_t* out_end, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIRE_HIP_H */
