// C ABI of libpire_hip.so (include/pire_hip.h).  Thin: argument checks, host<->device staging for the
// host-pointer convenience mode, kernel selection.  There is deliberately no CPU implementation of the walk
// here: without a HIP device the run entry points fail with PIRE_HIP_ENODEVICE.

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <new>
#include <stdexcept>

#include "internal.h"
#include "selftest.h"

namespace pirehip {

namespace {
thread_local std::string g_error;
thread_local const char* g_lastKernel = "none";
thread_local const char* g_lastSymbol = nullptr;
thread_local bool g_timing = false;
thread_local float g_lastMs = -1.0f;
}  // namespace

void SetError(const std::string& msg) { g_error = msg; }

// ---- device staging blocks cached between host-pointer calls (internal.h Staging) -------------------------------------
// Round 2 allocated and freed four or five device buffers in every host-pointer call of the prefix / suffix /
// half-final / counting / capture / slow entry points (77-110 us for 10 strings, most of it hipMalloc + hipFree, which
// also synchronise the device).  Blocks now come from per-device free lists by power-of-two size class and go back when
// the call returns (it has drained its stream by then); at most kStagingCacheBytes stay cached per device.
namespace {
constexpr size_t kStagingCacheBytes = size_t(1) << 30;
struct StagingCache {
	std::mutex mutex;
	std::vector<void*> free[48];   // by log2 of the block size
	size_t cached = 0;
};
StagingCache g_stagingCache[kMaxDevices];
int SizeClass(size_t bytes)
{
	int c = 12;   // 4 KiB at least
	while ((size_t(1) << c) < bytes)
		++c;
	return c;
}
}  // namespace

int StagingAcquire(size_t bytes, void** out, size_t* blockBytes)
{
	*out = nullptr;
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess || dev < 0 || dev >= kMaxDevices)
		return HipFail(e == hipSuccess ? hipErrorInvalidDevice : e, "hipGetDevice");
	const int c = SizeClass(bytes);
	*blockBytes = size_t(1) << c;
	{
		StagingCache& cache = g_stagingCache[dev];
		std::lock_guard<std::mutex> lock(cache.mutex);
		if (!cache.free[c].empty()) {
			*out = cache.free[c].back();
			cache.free[c].pop_back();
			cache.cached -= *blockBytes;
			return PIRE_HIP_OK;
		}
	}
	e = hipMalloc(out, *blockBytes);
	if (e != hipSuccess)
		return HipFail(e, "hipMalloc(staging)");
	return PIRE_HIP_OK;
}

namespace {
StagingCache g_pinnedCache;   // host blocks are not per device
}
int StagingAcquireHost(size_t bytes, void** out, size_t* blockBytes)
{
	*out = nullptr;
	const int c = SizeClass(bytes);
	*blockBytes = size_t(1) << c;
	{
		std::lock_guard<std::mutex> lock(g_pinnedCache.mutex);
		if (!g_pinnedCache.free[c].empty()) {
			*out = g_pinnedCache.free[c].back();
			g_pinnedCache.free[c].pop_back();
			g_pinnedCache.cached -= *blockBytes;
			return PIRE_HIP_OK;
		}
	}
	const hipError_t e = hipHostMalloc(out, *blockBytes, hipHostMallocDefault);
	if (e != hipSuccess)
		return HipFail(e, "hipHostMalloc(staging)");
	return PIRE_HIP_OK;
}

void StagingReleaseHost(void* p, size_t blockBytes)
{
	{
		std::lock_guard<std::mutex> lock(g_pinnedCache.mutex);
		if (g_pinnedCache.cached + blockBytes <= (size_t(64) << 20)) {
			g_pinnedCache.free[SizeClass(blockBytes)].push_back(p);
			g_pinnedCache.cached += blockBytes;
			return;
		}
	}
	(void)hipHostFree(p);
}

void StagingRelease(void* p, size_t blockBytes)
{
	int dev = -1;
	if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) {
		StagingCache& cache = g_stagingCache[dev];
		std::lock_guard<std::mutex> lock(cache.mutex);
		if (cache.cached + blockBytes <= kStagingCacheBytes) {
			cache.free[SizeClass(blockBytes)].push_back(p);
			cache.cached += blockBytes;
			return;
		}
	}
	(void)hipFree(p);
}

int HandleException() noexcept
{
	try {
		try {
			throw;
		} catch (const std::bad_alloc&) {
			g_error = "out of memory";
			return PIRE_HIP_ENOMEM;
		} catch (const std::length_error&) {
			g_error = "out of memory (a table of that size cannot be built)";
			return PIRE_HIP_ENOMEM;
		} catch (const std::exception& e) {
			g_error = std::string("internal error: ") + e.what();
			return PIRE_HIP_EINVAL;
		} catch (...) {
			g_error = "internal error";
			return PIRE_HIP_EINVAL;
		}
	} catch (...) {   // building the message failed too
		return PIRE_HIP_ENOMEM;
	}
}

int HipFail(hipError_t e, const char* what)
{
	g_error = std::string(what) + ": " + hipGetErrorString(e);
	(void)hipGetLastError();   // clear the sticky error
	return e == hipErrorOutOfMemory ? PIRE_HIP_ENOMEM : PIRE_HIP_ENODEVICE;
}

namespace {


// The caller holds the table's TableUse (internal.h): no adaptation between here and the caller's return.
// wantDist: also the image's per-state distance tables (EnsureActDist).
int FillParams(pire_hip_table* t, ScanParams* p, uint32_t flags, bool wantDist = false)
{
	DeviceTable d;   // a copy of the current device's image (pointers), taken under the table's lock
	if (int rc = UploadTable(t, &d))
		return rc;
	const HostTable& h = t->host;
	memset(p, 0, sizeof(*p));
	p->workBase = d.workCounter;
	p->workDevice = d.device >= 0 && d.device < kMaxDevices ? d.device : 0;
	p->hotRows = d.hotRows;
	p->hotRowsRot = d.hotRowsRot;
	p->topShare = h.topShare;
	p->hotFlags = d.hotFlags;
	p->cls = d.cls;
	p->nextPerm = d.nextPerm;
	p->flagsPerm = d.flagsPerm;
	p->origOfPerm = d.origOfPerm;
	p->permOfOrig = d.permOfOrig;
	p->acceptMaskPerm = d.acceptMaskPerm;
	p->acceptOffPerm = d.acceptOffPerm;
	p->acceptIds = d.acceptIds;
	p->finSelf = d.finSelf;
	p->finEnd = d.finEnd;
	p->visitHot = d.visitHot;
	p->visitCold = d.visitCold;
	p->trapSignal = d.trapSignalDev;
	p->compactRows = d.compactRows;
	p->compact = h.compact;
	p->wideRows = d.wideRows;
	p->next16 = d.next16;
	p->visitWide = d.visitWide;
	p->wide = d.wideRows ? h.wide : 0;
	p->zipFull = d.wideRows ? h.zipFull : 0;
	p->wideOutSlot = p->wide;
	p->wideRowsStream = d.wideRows ? d.wideRowsStream : nullptr;
	p->wideStream = d.wideRows ? d.wideStream : 0;
	p->outsideDense = h.outsideDense;
	p->outsideWide = h.outsideWide;
	p->massMeasured = h.massMeasured;
	if (!h.massMeasured && d.trapSignalHost) {
		// A table nobody has adapted yet: what its scans so far left in the trap signal (mapped host memory: a read) says more
		// about the share of the steps outside the dense rows than the a-priori byte model -- a sampled trap stands for 1 024
		// lane-steps there.  The choice of the walk (WideWanted) follows it from the launch after the first ones have run.
		const uint64_t bytes = t->bytesScanned.load(std::memory_order_relaxed);
		const uint64_t traps = *d.trapSignalHost;
		if (bytes >= (uint64_t(1) << 22) && traps >= 64) {
			const float live = float(std::min(1.0, double(traps) * 1024.0 / double(bytes)));
			if (live > p->outsideDense) {
				p->outsideDense = live;
				p->massMeasured = true;
			}
		}
	}
	p->wideLaunched = &t->wideLaunched;
	p->incPerm = d.incPerm;
	p->hotFinalLo = h.hotFinalLo;
	p->hotDeadLo = h.hotDeadLo;
	p->deadShare = h.deadShare;
	p->finalShare = h.finalShare;
	p->states = h.states;
	p->letters = h.letters;
	p->regexps = h.regexps;
	p->hot = h.hot;
	p->beginCls = h.cls[kBeginMark];
	p->endCls = h.cls[kEndMark];
	p->flags = flags;
	uint32_t start = h.initial;                                     // Initialize(), multi.h:161
	if (flags & PIRE_HIP_RUN_BEGIN)
		start = h.next[size_t(start) * h.letters + h.cls[kBeginMark]];  // Begin(), run.h:375
	p->startPerm = h.permOfOrig[start];
	p->hostPermOfOrig = h.permOfOrig.data();
	p->hostOrigOfPerm = h.origOfPerm.data();
	p->owner = t;
	if (wantDist)
		if (int rc = EnsureActDist(t, &p->distFinalPerm, &p->distFlaggedPerm))
			return rc;
	return PIRE_HIP_OK;
}

}  // namespace
// ---- library configuration (pire_hip_config) ----------------------------------------------------------------------------
namespace {
std::mutex g_cfgMutex;

uint64_t EnvU64(const char* name)
{
	const char* v = getenv(name);
	if (!v || !*v)
		return 0;
	char* end = nullptr;
	const unsigned long long x = strtoull(v, &end, 10);
	return end != v ? uint64_t(x) : 1;   // PIRE_HIP_NO_SEGMENTS=yes counts as set
}

// The environment seeds the defaults ONCE (static initialisation of the first GetConfig / config_set call).
pire_hip_config SeedFromEnvironment()
{
	pire_hip_config c;
	memset(&c, 0, sizeof(c));
	c.size = sizeof(c);
	c.tiled_variant = uint32_t(EnvU64("PIRE_HIP_TILED_VARIANT"));
	c.checked = EnvU64("PIRE_HIP_CHECKED") == 1;
	c.no_compact = EnvU64("PIRE_HIP_NO_COMPACT") != 0;
	c.prior_flat = EnvU64("PIRE_HIP_PRIOR_FLAT") != 0;
	c.ragged_act_always = EnvU64("PIRE_HIP_RAGGED_ACT_ALWAYS") != 0;
	c.no_ragged_act = EnvU64("PIRE_HIP_NO_RAGGED_ACT") != 0;
	c.no_segments = EnvU64("PIRE_HIP_NO_SEGMENTS") != 0;
	c.segment_no_grid = EnvU64("PIRE_HIP_SEGMENT_NO_GRID") != 0;
	c.segment_stats = EnvU64("PIRE_HIP_SEGMENT_STATS") != 0;
	c.segment_modes = uint32_t(EnvU64("PIRE_HIP_SEGMENT_MODES"));
	c.segment_bytes = EnvU64("PIRE_HIP_SEGMENT_BYTES");
	if (getenv("PIRE_HIP_SEGMENT_WARMUP"))
		c.segment_warmup = EnvU64("PIRE_HIP_SEGMENT_WARMUP") ? EnvU64("PIRE_HIP_SEGMENT_WARMUP") : PIRE_HIP_SEGMENT_WARMUP_NONE;
	if (getenv("PIRE_HIP_SEGMENT_BUDGET"))
		c.segment_budget = EnvU64("PIRE_HIP_SEGMENT_BUDGET") ? EnvU64("PIRE_HIP_SEGMENT_BUDGET") : PIRE_HIP_SEGMENT_BUDGET_NONE;
	c.host_chunk_bytes = EnvU64("PIRE_HIP_HOST_CHUNK_BYTES");
	c.host_one_shot = EnvU64("PIRE_HIP_HOST_ONE_SHOT") != 0;
	c.no_rccl = EnvU64("PIRE_HIP_NO_RCCL") != 0;
	c.slow_sets_in_memory = EnvU64("PIRE_HIP_SLOW_SETS_IN_MEMORY") == 1;
	c.slow_no_list = EnvU64("PIRE_HIP_SLOW_NO_LIST") != 0;
	c.auto_adapt = uint32_t(EnvU64("PIRE_HIP_AUTO_ADAPT"));
	c.auto_adapt_min_traps = uint32_t(EnvU64("PIRE_HIP_AUTO_ADAPT_MIN_TRAPS"));
	c.ragged_variant = uint32_t(EnvU64("PIRE_HIP_RAGGED_VARIANT"));
	c.host_staging = uint32_t(EnvU64("PIRE_HIP_HOST_STAGING"));
	c.no_offsets_peek = EnvU64("PIRE_HIP_NO_OFFSETS_PEEK") != 0;
	c.segment_no_pair = EnvU64("PIRE_HIP_SEGMENT_NO_PAIR") != 0;
	c.segment_no_product = EnvU64("PIRE_HIP_SEGMENT_NO_PRODUCT") != 0;
	c.segment_no_derive = EnvU64("PIRE_HIP_SEGMENT_NO_DERIVE") != 0;
	c.no_length_order = EnvU64("PIRE_HIP_NO_LENGTH_ORDER") != 0;
	c.capture_by_length = EnvU64("PIRE_HIP_CAPTURE_BY_LENGTH") != 0;
	c.force_rccl = EnvU64("PIRE_HIP_FORCE_RCCL") != 0;
	c.counting_variant = uint32_t(EnvU64("PIRE_HIP_COUNTING_VARIANT"));
	c.slow_stats = EnvU64("PIRE_HIP_SLOW_STATS") != 0;
	c.walk_variant = uint32_t(EnvU64("PIRE_HIP_WALK_VARIANT"));
	c.selftest = uint32_t(EnvU64("PIRE_HIP_SELFTEST"));
	c.zip_variant = uint32_t(EnvU64("PIRE_HIP_ZIP_VARIANT"));
	return c;
}

pire_hip_config& ConfigStorage()
{
	static pire_hip_config c = SeedFromEnvironment();
	return c;
}
}  // namespace

thread_local const pire_hip_config* g_cfgOverride = nullptr;
thread_local bool g_inEntrySelfTest = false;

pire_hip_config GetConfig()
{
	if (g_cfgOverride)
		return *g_cfgOverride;   // a first-use self-test on this thread (selftest.h): its routing knobs
	std::lock_guard<std::mutex> lock(g_cfgMutex);
	return ConfigStorage();
}

int RunSelfTestVariants(const std::vector<std::function<void(pire_hip_config&)>>& edits, const std::function<int()>& body)
{
	const pire_hip_config base = GetConfig();
	const pire_hip_config* const outer = g_cfgOverride;
	const bool was = g_inEntrySelfTest;
	int rc = PIRE_HIP_OK;
	for (size_t k = 0; k < edits.size() && rc == PIRE_HIP_OK; ++k) {
		pire_hip_config c = base;
		c.auto_adapt = 1;      // the table is not re-ranked under a self-test
		c.selftest = 1;        // ... and what the variant's call dispatches to is not tested again from inside
		c.no_segments = 1;
		edits[k](c);
		g_cfgOverride = &c;
		g_inEntrySelfTest = true;
		try {
			rc = body();
		} catch (...) {
			g_cfgOverride = outer;
			g_inEntrySelfTest = was;
			throw;
		}
		if (rc == PIRE_HIP_OK)
			NoteSelfTested(pire_hip_last_kernel());
		g_cfgOverride = outer;
		g_inEntrySelfTest = was;
	}
	return rc;
}

int SelfTestMismatch(const char* what, uint32_t string, const std::string& got, const std::string& want)
{
	SetError(std::string("self-test of ") + what + " (kernel '" + pire_hip_last_kernel() + "') failed: string " + std::to_string(string) +
	         " of the known-answer batch gave " + got + ", the table's transitions give " + want + " -- this build of libpire_hip.so (" +
	         pire_hip_build_info() + ") must not be used on this device");
	return PIRE_HIP_ESELFTEST;
}

bool EntrySelfTestDue(hipStream_t stream, uint32_t* mode)
{
	if (g_inEntrySelfTest)
		return false;
	*mode = GetConfig().selftest;
	if (*mode == 1)
		return false;
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	return hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing == hipStreamCaptureStatusNone;
}

namespace {
std::mutex g_testedMutex;
std::string g_tested;   // ",name,name,"
}  // namespace

void NoteSelfTested(const char* kernel)
{
	std::lock_guard<std::mutex> lock(g_testedMutex);
	const std::string key = std::string(",") + kernel + ",";
	if (g_tested.empty())
		g_tested = ",";
	if (g_tested.find(key) == std::string::npos)
		g_tested += std::string(kernel) + ",";
}

std::string SelfTestedKernels()
{
	std::lock_guard<std::mutex> lock(g_testedMutex);
	return g_tested;
}

void NoteKernel(const char* name, const char* symbol)
{
	g_lastKernel = name;
	g_lastSymbol = symbol;
}
namespace {

// One slot of the table's ring of ragged work counters per launch (launches of one table may overlap on streams).
unsigned long long* NextWorkSlot(pire_hip_table* t, const ScanParams& p)
{
	return WorkSlotOf(p.workBase, t->workSlot[p.workDevice].fetch_add(1));
}

}  // namespace
int PrepareScanParams(pire_hip_table* t, ScanParams* p, uint32_t flags, TableUse* use, bool wantDist, bool enqueueOnly)
{
	use->Acquire(t, enqueueOnly);
	return FillParams(t, p, flags, wantDist);
}
unsigned long long* TakeWorkSlot(pire_hip_table* t, const ScanParams& p) { return NextWorkSlot(t, p); }
namespace {

// ---- first-use self-test (pire_hip_config.selftest) ---------------------------------------------------------------------
// The kernels below keep text on its way in registers and count their own waits; the build argues from the ISA that this
// is sound (tools/audit), this checks it where it runs: the first time a table takes one of Dispatch's kernels, that
// kernel scans a batch whose answer the host image's transitions give (the accessor walk of pire_hip_table_next, 131 072
// steps), on a stream and with visit counters of its own.  The text is a walk through the table's own states that
// stays out of the dead ones where it can, so that the end state depends on every byte of the string.
enum KernelKind : int { kKindGeneric, kKindTiled, kKindWide, kKindRagged, kKindRaggedWide, kKindStream, kKindStreamWide };

int SelfTest(const ScanParams& p, int kind, const char* name, uint32_t mode)
{
	constexpr uint32_t kStrings = 256, kLen = 512;
	const HostTable& h = p.owner->host;
	std::vector<uint8_t> text(size_t(kStrings) * kLen);
	std::vector<uint32_t> want(kStrings);
	std::vector<uint64_t> offsets(kStrings + 1);
	uint64_t rng = 0x9E3779B97F4A7C15ull ^ (uint64_t(h.states) << 20) ^ h.letters;
	auto next = [&](uint32_t st, uint32_t ch) { return h.next[size_t(st) * h.letters + h.cls[ch]]; };
	const uint32_t start = p.hostOrigOfPerm[p.startPerm];
	for (uint32_t i = 0; i < kStrings; ++i) {
		uint32_t st = start;
		offsets[i] = uint64_t(i) * kLen;
		for (uint32_t j = 0; j < kLen; ++j) {
			uint32_t ch = 0, to = st;
			for (int attempt = 0; attempt < 4; ++attempt) {
				rng = rng * 6364136223846793005ull + 1442695040888963407ull;
				// (printable text twice out of three times: what the tables here are mostly about)
				ch = (rng >> 33) % 3 ? 32 + uint32_t(rng >> 40) % 95 : uint32_t(rng >> 40) & 255;
				to = next(st, ch);
				if (!(h.flags[to] & kDead))
					break;
			}
			text[size_t(i) * kLen + j] = uint8_t(ch);
			st = to;
		}
		if (p.flags & PIRE_HIP_RUN_END)
			st = next(st, kEndMark);
		want[i] = st;
	}
	offsets[kStrings] = uint64_t(kStrings) * kLen;
	if (mode == 2)
		want[kStrings / 2] ^= 1;   // tests of the failure path
	// one block: text | offsets | end states | work counter | visit counters (dense rows, wide rows, every state)
	const size_t offOffsets = text.size(), offOut = offOffsets + offsets.size() * 8, offWork = offOut + kStrings * 4;
	const size_t offHot = offWork + 256, offWide = offHot + kVisitHotSlots * 4;
	const size_t offCold = offWide + (size_t(p.wide) + 1) * 4, bytes = offCold + size_t(p.states) * 4;
	void* block = nullptr;
	size_t blockBytes = 0;
	if (int rc = StagingAcquire(bytes, &block, &blockBytes))
		return rc;
	uint8_t* base = static_cast<uint8_t*>(block);
	hipStream_t own = nullptr;
	hipError_t e = hipStreamCreateWithFlags(&own, hipStreamNonBlocking);
	std::vector<uint32_t> got(kStrings, ~0u);
	int rc = PIRE_HIP_OK;
	const int passes = kind == kKindWide ? 2 : 1;   // the class-indexed walk: with one and with two strings per lane
	for (int pass = 0; pass < passes && e == hipSuccess && rc == PIRE_HIP_OK; ++pass) {
		if ((e = hipMemsetAsync(base + offOut, 0xFF, bytes - offOut, own)) != hipSuccess ||
		    (e = hipMemsetAsync(base + offWork, 0, bytes - offWork, own)) != hipSuccess ||
		    (e = hipMemcpyAsync(base, text.data(), text.size(), hipMemcpyHostToDevice, own)) != hipSuccess ||
		    (e = hipMemcpyAsync(base + offOffsets, offsets.data(), offsets.size() * 8, hipMemcpyHostToDevice, own)) != hipSuccess)
			break;
		std::atomic<uint64_t> launched{0};
		ScanParams q = p;
		q.text = base;
		q.textEnd = text.size();
		q.ends = nullptr;
		q.initIdx = nullptr;
		q.outIdx = reinterpret_cast<uint32_t*>(base + offOut);
		q.outFinal = nullptr;
		q.outCounts = nullptr;
		q.flags = p.flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END);
		q.n = kStrings;
		q.len = kLen;
		q.stride = kLen;
		q.offsets = kind == kKindTiled || kind == kKindWide ? nullptr : reinterpret_cast<const uint64_t*>(base + offOffsets);
		q.visitHot = reinterpret_cast<uint32_t*>(base + offHot);
		q.visitWide = reinterpret_cast<uint32_t*>(base + offWide);
		q.visitCold = reinterpret_cast<uint32_t*>(base + offCold);
		q.trapSignal = nullptr;
		q.wideLaunched = &launched;
		q.forceLanes = pass ? 2 : 1;   // (the batch is too small for LaunchWide to choose two strings per lane by itself)
		unsigned long long* work = reinterpret_cast<unsigned long long*>(base + offWork);
		rc = kind == kKindWide ? LaunchWide(q, own) : kind == kKindTiled ? LaunchTiled(q, own)
		     : kind == kKindRaggedWide ? LaunchRaggedWide(q, work, own) : kind == kKindStream ? LaunchStream(q, own)
		     : kind == kKindStreamWide ? LaunchStreamWide(q, own) : kind == kKindRagged ? LaunchRagged(q, work, own) : LaunchGeneric(q, own);
		if (rc != PIRE_HIP_OK)
			break;
		if ((e = hipMemcpyAsync(got.data(), base + offOut, kStrings * 4, hipMemcpyDeviceToHost, own)) != hipSuccess ||
		    (e = hipStreamSynchronize(own)) != hipSuccess)
			break;
		for (uint32_t i = 0; i < kStrings; ++i)
			if (got[i] != want[i]) {
				SetError(std::string("self-test of the ") + name + " kernel failed: string " + std::to_string(i) + " of the known-answer batch ended in state " +
				         std::to_string(got[i]) + ", the table's transitions give " + std::to_string(want[i]) +
				         " -- this build of libpire_hip.so (" + pire_hip_build_info() + ") must not be used on this device");
				rc = PIRE_HIP_ESELFTEST;
				break;
			}
	}
	if (own) {
		// (an error path may leave the kernel or a copy on its way: the block goes back to the cache only behind them -- ADVICE r5)
		(void)hipStreamSynchronize(own);
		(void)hipStreamDestroy(own);
	}
	StagingRelease(block, blockBytes);
	if (rc == PIRE_HIP_OK && e != hipSuccess)
		rc = HipFail(e, "self-test");
	return rc;
}

int Dispatch(const ScanParams& p, hipStream_t stream, unsigned long long* workCounter = nullptr,
             uint64_t totalBytesHint = 0)
{
	if (p.n == 0)
		return PIRE_HIP_OK;
	const bool tiled = !(p.flags & PIRE_HIP_RUN_GENERIC) && TiledEligible(p);
	const bool ragged = !tiled && !(p.flags & PIRE_HIP_RUN_GENERIC) && workCounter && RaggedEligible(p, totalBytesHint);
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if (g_timing) {
		if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)
			return HipFail(hipGetLastError(), "hipEventCreate");
		(void)hipEventRecord(ev0, stream);
	}
	// a table whose scans keep leaving the dense rows: the class-indexed walk (wide.hip; offset batches: the ragged kernel on it)
	const bool wideTable = (tiled || ragged) && WideWanted(p, GetConfig());
	const bool wide = tiled && p.len >= 256 && wideTable;
	const bool streamWide = ragged && wideTable && StreamWideEligible(p, totalBytesHint);
	const bool raggedWide = ragged && wideTable && !streamWide;
	const bool streamed = ragged && !wideTable && StreamEligible(p, totalBytesHint);
	const int kind = wide ? kKindWide : tiled ? kKindTiled : streamWide ? kKindStreamWide : raggedWide ? kKindRaggedWide : streamed ? kKindStream
	                 : ragged ? kKindRagged : kKindGeneric;
	static const char* const kNames[] = {"generic", "tiled", "wide", "ragged", "ragged_wide", "stream", "stream_wide"};
	if (p.owner && !(p.owner->selfTested[p.workDevice].load(std::memory_order_relaxed) & (1u << kind))) {
		const uint32_t mode = GetConfig().selftest;
		hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
		if (mode != 1 && !(p.flags & (1u << 23)) &&   // (bit 23: device_common.h kPermIds, the segmented scan's internal passes)
		     (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing == hipStreamCaptureStatusNone)) {
			if (int rc = SelfTest(p, kind, kNames[kind], mode)) {
				if (ev0)
					(void)hipEventDestroy(ev0);
				if (ev1)
					(void)hipEventDestroy(ev1);
				return rc;
			}
			p.owner->selfTested[p.workDevice].fetch_or(1u << kind);
			NoteSelfTested(kNames[kind]);
		}
	}
	NoteKernel(kNames[kind]);
	if (p.owner && !(p.flags & (1u << 23)))   // (what the live estimate of FillParams divides the trap signal by)
	{
		// (offsets on the device: the host does not know the bytes -- the hint is ~0, which round 6's first form ADDED to the count:
		// it wrapped.  What the live estimate of FillParams divides by takes nothing for such a batch -- a guess there would turn
		// tables to the other walk on a guess --; the share of offset batches takes 48 bytes a string.)
		const bool known = !p.offsets || totalBytesHint != ~0ull;
		const uint64_t bytes = !p.offsets ? p.n * p.len : known ? totalBytesHint : p.n * 48;
		if (known)
			p.owner->bytesScanned.fetch_add(bytes, std::memory_order_relaxed);
		p.owner->bytesNominal.fetch_add(bytes, std::memory_order_relaxed);
		if (p.offsets)
			p.owner->bytesOffsetBatches.fetch_add(bytes, std::memory_order_relaxed);
	}
	int rc = wide ? LaunchWide(p, stream) : tiled ? LaunchTiled(p, stream) : streamWide ? LaunchStreamWide(p, stream)
	         : raggedWide ? LaunchRaggedWide(p, workCounter, stream) : streamed ? LaunchStream(p, stream)
	         : ragged ? LaunchRagged(p, workCounter, stream) : LaunchGeneric(p, stream);
	if (g_timing) {
		(void)hipEventRecord(ev1, stream);
		if (rc == PIRE_HIP_OK) {
			hipError_t e = hipEventSynchronize(ev1);
			if (e != hipSuccess)
				rc = HipFail(e, "hipEventSynchronize");
			else
				(void)hipEventElapsedTime(&g_lastMs, ev0, ev1);
		}
		(void)hipEventDestroy(ev0);
		(void)hipEventDestroy(ev1);
	}
	return rc;
}

// ---- host-pointer mode, chunked ---------------------------------------------------------------------------------------
// The batch is cut into chunks of whole strings (up to kHostChunkBytes of text each).  Chunk k goes
//     H2D text (+ its offsets / resume states)  ->  scan  ->  D2H results (into pinned staging)
// on stream k % 2 of a staging arena, so that the transfer of chunk k+1 overlaps the scan and the result copies of
// chunk k, the device staging is bounded whatever the batch size, and nothing is allocated per call: arenas (two
// streams, two sets of buffers grown on demand) are pooled per device for the life of the process.
// Measured (profiles/r02_pcie_probe.log, r02_host_mode.log): a scan is ~1 % of its own transfer, so the overlap buys
// little; what matters is (a) no hipMalloc/hipFree per call, (b) results never copied device -> PAGEABLE host inside
// the loop (that call blocks until the chunk's scan is over and serialises everything), (c) LARGE chunks -- the
// runtime stages pageable memory faster in few big copies (53 GB/s for 1 GiB at once, 47 in 32 MiB pieces).  Pinned
// text (pire_hip_host_alloc) is plain DMA at 57 GB/s.
constexpr size_t kHostChunkBytes = size_t(256) << 20;
constexpr size_t kHostChunkSlack = 4096;        // the kernels read whole 16-byte blocks around a string's ends
constexpr uint64_t kHostChunkStrings = 1u << 22;

struct HostArena {
	int device = -1;
	hipStream_t stream[2] = {nullptr, nullptr};
	hipEvent_t done[2] = {nullptr, nullptr};
	hipEvent_t ready = nullptr;
	// per slot, grown on demand
	uint8_t* text[2] = {nullptr, nullptr};
	size_t textCap[2] = {0, 0};
	uint64_t* offs[2] = {nullptr, nullptr};
	uint32_t* init[2] = {nullptr, nullptr};
	uint32_t* idx[2] = {nullptr, nullptr};
	uint8_t* fin[2] = {nullptr, nullptr};
	// results land in pinned host memory first: a device-to-host copy into the caller's (pageable) arrays would block
	// the host until the chunk's scan is over, and with it the transfer of the next chunk
	uint32_t* hostIdx[2] = {nullptr, nullptr};
	uint8_t* hostFin[2] = {nullptr, nullptr};
	size_t stringCap[2] = {0, 0};
	unsigned long long* counts = nullptr;
	size_t countBytes = 0;
};

std::mutex g_arenaMutex;
std::vector<HostArena*> g_arenas[kMaxDevices];

void FreeSlotStrings(HostArena* a, int k)
{
	(void)hipFree(a->offs[k]);
	(void)hipFree(a->init[k]);
	(void)hipFree(a->idx[k]);
	(void)hipFree(a->fin[k]);
	(void)hipHostFree(a->hostIdx[k]);
	(void)hipHostFree(a->hostFin[k]);
	a->offs[k] = nullptr;
	a->init[k] = a->idx[k] = a->hostIdx[k] = nullptr;
	a->fin[k] = a->hostFin[k] = nullptr;
	a->stringCap[k] = 0;
}

void DestroyArena(HostArena* a)
{
	for (int k = 0; k < 2; ++k) {
		if (a->stream[k])
			(void)hipStreamDestroy(a->stream[k]);
		if (a->done[k])
			(void)hipEventDestroy(a->done[k]);
		(void)hipFree(a->text[k]);
		FreeSlotStrings(a, k);
	}
	(void)hipFree(a->counts);
	if (a->ready)
		(void)hipEventDestroy(a->ready);
	delete a;
}

// Slot k of the arena can take `textBytes` of text and `strings` strings (the slot is idle when this is called).
int ReserveSlot(HostArena* a, int k, size_t textBytes, size_t strings)
{
	hipError_t e = hipSuccess;
	if (textBytes > a->textCap[k]) {
		(void)hipFree(a->text[k]);
		a->text[k] = nullptr;
		a->textCap[k] = 0;
		e = hipMalloc(reinterpret_cast<void**>(&a->text[k]), textBytes);
		if (e == hipSuccess)
			a->textCap[k] = textBytes;
	}
	if (e == hipSuccess && strings > a->stringCap[k]) {
		FreeSlotStrings(a, k);
		e = hipMalloc(reinterpret_cast<void**>(&a->offs[k]), (strings + 1) * 8);
		if (e == hipSuccess)
			e = hipMalloc(reinterpret_cast<void**>(&a->init[k]), strings * 4);
		if (e == hipSuccess)
			e = hipMalloc(reinterpret_cast<void**>(&a->idx[k]), strings * 4);
		if (e == hipSuccess)
			e = hipMalloc(reinterpret_cast<void**>(&a->fin[k]), strings);
		if (e == hipSuccess)
			e = hipHostMalloc(reinterpret_cast<void**>(&a->hostIdx[k]), strings * 4, hipHostMallocDefault);
		if (e == hipSuccess)
			e = hipHostMalloc(reinterpret_cast<void**>(&a->hostFin[k]), strings, hipHostMallocDefault);
		if (e == hipSuccess)
			a->stringCap[k] = strings;
	}
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "staging arena");
}

int AcquireArena(size_t countBytes, HostArena** out)
{
	int dev = -1;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess || dev < 0 || dev >= kMaxDevices)
		return HipFail(e == hipSuccess ? hipErrorInvalidDevice : e, "hipGetDevice");
	{
		std::lock_guard<std::mutex> lock(g_arenaMutex);
		auto& pool = g_arenas[dev];
		for (size_t i = 0; i < pool.size(); ++i)
			if (pool[i]->countBytes >= countBytes) {
				*out = pool[i];
				pool.erase(pool.begin() + i);
				return PIRE_HIP_OK;
			}
	}
	std::unique_ptr<HostArena, void (*)(HostArena*)> a(new HostArena, DestroyArena);
	a->device = dev;
	a->countBytes = std::max<size_t>(countBytes, 1024);
	for (int k = 0; k < 2 && e == hipSuccess; ++k) {
		e = hipStreamCreateWithFlags(&a->stream[k], hipStreamNonBlocking);
		if (e == hipSuccess)
			e = hipEventCreateWithFlags(&a->done[k], hipEventDisableTiming);
	}
	if (e == hipSuccess)
		e = hipMalloc(reinterpret_cast<void**>(&a->counts), a->countBytes);
	if (e == hipSuccess)
		e = hipEventCreateWithFlags(&a->ready, hipEventDisableTiming);
	if (e != hipSuccess)
		return HipFail(e, "staging arena");
	*out = a.release();
	return PIRE_HIP_OK;
}

void ReleaseArena(HostArena* a)
{
	std::lock_guard<std::mutex> lock(g_arenaMutex);
	g_arenas[a->device].push_back(a);
}

// Returns PIRE_HIP_OK and *done = true when it handled the batch; *done = false (nothing touched) when the batch does
// not fit the chunking (a string longer than a chunk) and the caller should take the one-shot path.
int RunHostPipelined(pire_hip_table* t, const ScanParams& base, const uint8_t* text, const uint64_t* offsets, uint64_t n,
                     uint64_t len, uint64_t stride, const uint32_t* init, uint32_t* outIdx, uint8_t* outFinal,
                     uint64_t* outCounts, bool* done)
{
	*done = false;
	size_t chunkBytes = kHostChunkBytes;
	if (const uint64_t knob = GetConfig().host_chunk_bytes)   // tests: many small chunks
		chunkBytes = std::max<size_t>(4096, std::min<size_t>(kHostChunkBytes, size_t(knob)));
	// chunk boundaries: [first[c], first[c+1]) strings, text bytes [lo, hi) of the caller's buffer
	std::vector<uint64_t> first{0};
	if (offsets) {
		while (first.back() < n) {
			const uint64_t f = first.back();
			const uint64_t limit = offsets[f] + chunkBytes;
			uint64_t l = uint64_t(std::upper_bound(offsets + f + 1, offsets + n + 1, limit) - offsets) - 1;   // last string that still fits
			l = std::min<uint64_t>(l, f + kHostChunkStrings);
			if (l == f)
				return PIRE_HIP_OK;   // a single string larger than a chunk: not this path
			first.push_back(l);
		}
	} else {
		if (stride > chunkBytes / 64)
			return PIRE_HIP_OK;
		// stride 0 = n empty records (len == 0, pire_ut.cpp:832-837 semantics per record): chunk by string count alone
		uint64_t per = std::min<uint64_t>(stride ? chunkBytes / stride : kHostChunkStrings, kHostChunkStrings) & ~uint64_t(1023);
		if (per == 0)
			per = stride ? chunkBytes / stride : kHostChunkStrings;
		for (uint64_t f = per; f < n; f += per)
			first.push_back(f);
		first.push_back(n);
	}
	const size_t cntBytes = (size_t(t->host.regexps) + 2) * 8;
	HostArena* arena = nullptr;
	if (int rc = AcquireArena(cntBytes, &arena))
		return rc;
	struct Return {
		HostArena* a;
		~Return()
		{
			// whatever happened, nothing of this call may still be in flight when the arena goes back to the pool
			(void)hipStreamSynchronize(a->stream[0]);
			(void)hipStreamSynchronize(a->stream[1]);
			ReleaseArena(a);
		}
	} giveBack{arena};
	*done = true;
	hipError_t e = hipSuccess;
	if (outCounts) {
		// the counters are accumulated into: start from the caller's values, both streams wait for them
		e = hipMemcpyAsync(arena->counts, outCounts, cntBytes, hipMemcpyHostToDevice, arena->stream[0]);
		if (e == hipSuccess)
			e = hipEventRecord(arena->ready, arena->stream[0]);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(arena->stream[1], arena->ready, 0);
		if (e != hipSuccess)
			return HipFail(e, "hipMemcpy(counts)");
	}
	// results of the chunk that used slot k last: out of the pinned staging into the caller's arrays
	auto drain = [&](size_t c) -> hipError_t {
		const int k = int(c & 1);
		hipError_t err = hipEventSynchronize(arena->done[k]);
		if (err != hipSuccess)
			return err;
		const uint64_t f = first[c], cnt = first[c + 1] - f;
		if (outIdx)
			memcpy(outIdx + f, arena->hostIdx[k], cnt * 4);
		if (outFinal)
			memcpy(outFinal + f, arena->hostFin[k], cnt);
		return hipSuccess;
	};
	const size_t chunks = first.size() - 1;
	for (size_t c = 0; c < chunks; ++c) {
		const int k = int(c & 1);
		hipStream_t s = arena->stream[k];
		const uint64_t f = first[c], cnt = first[c + 1] - f;
		if (c >= 2 && (e = drain(c - 2)) != hipSuccess)
			return HipFail(e, "hipEventSynchronize");
		{
			const uint64_t bytes = offsets ? offsets[f + cnt] - (offsets[f] & ~uint64_t(255)) : (cnt - 1) * stride + len;
			if (int rc = ReserveSlot(arena, k, size_t(bytes) + 2 * kHostChunkSlack, size_t(cnt)))
				return rc;
		}
		ScanParams p = base;
		p.n = cnt;
		uint64_t lo, hi;
		if (offsets) {
			lo = offsets[f] & ~uint64_t(255);   // keeps every string's alignment as it is in the caller's buffer
			hi = offsets[f + cnt];
			p.text = arena->text[k] - lo;       // so that the caller's offsets address the staged copy
			e = hipMemcpyAsync(arena->offs[k], offsets + f, (cnt + 1) * 8, hipMemcpyHostToDevice, s);
			p.offsets = arena->offs[k];
		} else {
			lo = f * stride;
			hi = (f + cnt - 1) * stride + len;
			p.text = arena->text[k];
			p.offsets = nullptr;
		}
		if (e == hipSuccess && hi > lo)
			e = hipMemcpyAsync(arena->text[k], text + lo, hi - lo, hipMemcpyHostToDevice, s);
		if (e == hipSuccess && init) {
			e = hipMemcpyAsync(arena->init[k], init + f, cnt * 4, hipMemcpyHostToDevice, s);
			p.initIdx = arena->init[k];
		}
		if (e != hipSuccess)
			return HipFail(e, "hipMemcpy(H2D)");
		p.outIdx = outIdx ? arena->idx[k] : nullptr;
		p.outFinal = outFinal ? arena->fin[k] : nullptr;
		p.outCounts = outCounts ? arena->counts : nullptr;
		if (int rc = Dispatch(p, s, NextWorkSlot(t, p), hi - lo))
			return rc;
		if (outIdx)
			e = hipMemcpyAsync(arena->hostIdx[k], arena->idx[k], cnt * 4, hipMemcpyDeviceToHost, s);
		if (e == hipSuccess && outFinal)
			e = hipMemcpyAsync(arena->hostFin[k], arena->fin[k], cnt, hipMemcpyDeviceToHost, s);
		if (e == hipSuccess)
			e = hipEventRecord(arena->done[k], s);
		if (e != hipSuccess)
			return HipFail(e, "hipMemcpy(D2H)");
	}
	for (size_t c = chunks >= 2 ? chunks - 2 : 0; c < chunks && e == hipSuccess; ++c)
		e = drain(c);
	if (e == hipSuccess && outCounts)
		e = hipMemcpy(outCounts, arena->counts, cntBytes, hipMemcpyDeviceToHost);
	if (e != hipSuccess)
		return HipFail(e, "copy back / synchronize");
	return PIRE_HIP_OK;
}

// Shared body of pire_hip_run / pire_hip_run_strided.
int RunImpl(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint64_t len, uint64_t stride,
            uint32_t flags, const uint32_t* init, uint32_t* outIdx, uint8_t* outFinal, uint64_t* outCounts,
            void* streamPtr)
{
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	// only the public bits: the upper ones are the kernels' internal switches (device_common.h)
	flags &= PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_ON_DEVICE | PIRE_HIP_RUN_GENERIC | PIRE_HIP_RUN_HOST_OFFSETS |
	         PIRE_HIP_RUN_NO_PEEK;
	ScanParams p;
	// (an automatic adaptation drains the device: only in calls that synchronise anyway, pire_hip_config.auto_adapt)
	TableUse use(t, (flags & PIRE_HIP_RUN_ON_DEVICE) && !(flags & PIRE_HIP_RUN_HOST_OFFSETS));
	if (int rc = FillParams(t, &p, flags))
		return rc;
	p.n = n;
	p.len = len;
	p.stride = stride;

	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		p.initIdx = init;
		p.outIdx = outIdx;
		p.outFinal = outFinal;
		p.outCounts = reinterpret_cast<unsigned long long*>(outCounts);
		if ((flags & PIRE_HIP_RUN_HOST_OFFSETS) && offsets && n) {
			// resident text, offsets known to the host: copy them into stream-ordered scratch
			for (uint64_t i = 0; i < n; ++i)
				if (offsets[i] > offsets[i + 1]) {
					SetError("offsets must be non-decreasing");
					return PIRE_HIP_EINVAL;
				}
			void* d = nullptr;
			hipError_t e = hipMallocAsync(&d, (n + 1) * 8, stream);
			if (e != hipSuccess)
				return HipFail(e, "hipMallocAsync(offsets)");
			e = hipMemcpyAsync(d, offsets, (n + 1) * 8, hipMemcpyHostToDevice, stream);
			int rc = e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemcpy(offsets)");
			p.offsets = static_cast<const uint64_t*>(d);
			const uint64_t textBytes = offsets[n];
			if (!rc) {
				if (!(flags & PIRE_HIP_RUN_GENERIC) && SegmentedEligible(n, textBytes - offsets[0]))
					rc = RunSegmented(t, p, offsets, stream);
				else
					rc = Dispatch(p, stream, NextWorkSlot(t, p), textBytes);
			}
			(void)hipFreeAsync(d, stream);
			if (!rc) {
				e = hipStreamSynchronize(stream);   // the caller's offsets array was the source of an async copy
				if (e != hipSuccess)
					rc = HipFail(e, "hipStreamSynchronize");
			}
			return rc;
		}
		// device offsets: the total text size is not known on the host; n >= 256 strings of unknown length still
		// need 128 readable bytes at `text` for the ragged kernel, which the caller guarantees by passing a batch
		// (a batch with less than 4 KiB of text is not worth a GPU launch; use PIRE_HIP_RUN_GENERIC to force the
		// offset-exact kernel)
		// Few long strings starve a one-string-per-lane kernel: cut them into segments (segmented.hip).  With
		// device offsets the host does not know the lengths, so only fixed-length records qualify here.
		if (!offsets && !(flags & PIRE_HIP_RUN_GENERIC) && n && SegmentedEligible(n, n * len))
			return RunSegmented(t, p, nullptr, stream);
		// Offsets on the device: the host does not know the lengths, and a handful of long strings would walk one per
		// lane at 25 MB/s each (a single 1 GiB string: 45 s against 0.5 ms segmented).  A batch small enough to leave
		// lanes idle (n < 65 536) is therefore PEEKED at: two words (first and last offset) come back -- which
		// synchronises `stream` -- and only if the cost model then asks for the segmented scan are all n + 1 offsets
		// fetched (<= 512 KB).  pire_hip_config.no_offsets_peek keeps such calls enqueue-only.
		if (offsets && n && n < 65536 && !(flags & (PIRE_HIP_RUN_GENERIC | PIRE_HIP_RUN_NO_PEEK)) && !GetConfig().no_offsets_peek) {
			uint64_t ends[2] = {0, 0};
			hipError_t e = hipMemcpyAsync(&ends[0], offsets, 8, hipMemcpyDeviceToHost, stream);
			if (e == hipSuccess)
				e = hipMemcpyAsync(&ends[1], offsets + n, 8, hipMemcpyDeviceToHost, stream);
			if (e == hipSuccess)
				e = hipStreamSynchronize(stream);
			if (e != hipSuccess)
				return HipFail(e, "reading the first and last offset back");
			const uint64_t total = ends[1] >= ends[0] ? ends[1] - ends[0] : 0;
			if (SegmentedEligible(n, total)) {
				std::vector<uint64_t> hostOffsets(size_t(n) + 1);
				e = hipMemcpyAsync(hostOffsets.data(), offsets, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, stream);
				if (e == hipSuccess)
					e = hipStreamSynchronize(stream);
				if (e != hipSuccess)
					return HipFail(e, "reading the offsets back");
				for (uint64_t i = 0; i < n; ++i)
					if (hostOffsets[i] > hostOffsets[i + 1]) {
						SetError("offsets must be non-decreasing");
						return PIRE_HIP_EINVAL;
					}
				return RunSegmented(t, p, hostOffsets.data(), stream);
			}
			return Dispatch(p, stream, NextWorkSlot(t, p), total);
		}
		return Dispatch(p, stream, NextWorkSlot(t, p), offsets ? ~0ull : 0);
	}

	// Host-pointer mode (PCIe-inclusive; the benchmark never times this mode): through the chunked, pooled staging of
	// RunHostPipelined; few long strings (segmented scan), timed calls and strings larger than a chunk are staged in one
	// piece.
	if (n == 0)
		return PIRE_HIP_OK;
	Staging st(stream);
	uint64_t textBytes;
	if (offsets) {
		for (uint64_t i = 0; i < n; ++i)
			if (offsets[i] > offsets[i + 1]) {
				SetError("offsets must be non-decreasing");
				return PIRE_HIP_EINVAL;
			}
		textBytes = offsets[n];
	} else {
		textBytes = (n - 1) * stride + len;
	}
	if (!text && textBytes) {
		// a null text pointer is fine only when every string is empty (tests/pire_ut.cpp:832-837)
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	if (init)
		for (uint64_t i = 0; i < n; ++i)
			if (init[i] >= t->host.states) {
				// host-side array: checking it costs nothing next to the transfer (device-side arrays are the caller's
				// responsibility, as a state is in the reference: Runner(sc, st) trusts st, run.h:368)
				SetError("init_state_idx out of range");
				return PIRE_HIP_EINVAL;
			}
	const bool segmented = !(flags & PIRE_HIP_RUN_GENERIC) && SegmentedEligible(n, textBytes);
	const size_t cntBytes = (size_t(t->host.regexps) + 2) * 8;
	// The pooled, chunked staging wins from a few hundred KiB on (3 MB 563 -> 308 us per call); a call whose inputs and
	// results all fit the one-piece staging's arena (internal.h, Staging: one copy in, one copy out) is faster there
	// (round 3: 10 strings 36 -> 30 us)
	const bool small = st.mode == 0 && textBytes + (n + 1) * 8 + n * 9 + cntBytes + 8 * 256 <= Staging::kArenaBytes;
	if (!segmented && !small && !g_timing && !GetConfig().host_one_shot) {
		bool done = false;
		const int rc = RunHostPipelined(t, p, static_cast<const uint8_t*>(text), offsets, n, len, stride, init, outIdx,
		                                outFinal, outCounts, &done);
		if (rc || done)
			return rc;
	}
	const uint8_t* dText = nullptr;
	if (int rc = st.In(static_cast<const uint8_t*>(text), size_t(textBytes), &dText, stream))
		return rc;
	p.text = dText;
	if (offsets)
		if (int rc = st.In(offsets, size_t(n + 1), &p.offsets, stream))
			return rc;
	if (init)
		if (int rc = st.In(init, size_t(n), &p.initIdx, stream))
			return rc;
	void* dIdx = nullptr;
	void* dFin = nullptr;
	const unsigned long long* dCnt = nullptr;
	if (outCounts) {   // accumulated into: an input as well as a result
		if (int rc = st.In(reinterpret_cast<const unsigned long long*>(outCounts), cntBytes / 8, &dCnt, stream))
			return rc;
		p.outCounts = const_cast<unsigned long long*>(dCnt);
	}
	if (outIdx) {
		if (int rc = st.Alloc(&dIdx, size_t(n) * 4))
			return rc;
		p.outIdx = static_cast<uint32_t*>(dIdx);
	}
	if (outFinal) {
		if (int rc = st.Alloc(&dFin, size_t(n)))
			return rc;
		p.outFinal = static_cast<uint8_t*>(dFin);
	}
	if (int rc = st.Flush())
		return rc;
	if (segmented) {
		if (int rc = RunSegmented(t, p, offsets, stream))
			return rc;
	} else if (int rc = Dispatch(p, stream, NextWorkSlot(t, p), textBytes)) {
		return rc;
	}
	if (int rc = st.Out(outIdx, dIdx, size_t(n) * 4))
		return rc;
	if (int rc = st.Out(outFinal, dFin, size_t(n)))
		return rc;
	if (int rc = st.Out(outCounts, dCnt, cntBytes))
		return rc;
	return st.Finish();
}

}  // namespace
}  // namespace pirehip

using namespace pirehip;

const char* pire_hip_selftested_kernels(void)
{
	static thread_local std::string copy;
	copy = SelfTestedKernels();
	return copy.c_str();
}

extern "C" {

int pire_hip_config_get(pire_hip_config* out)
{
	if (!out || out->size < sizeof(uint32_t)) {
		SetError("pire_hip_config_get: out->size must hold the caller's sizeof(pire_hip_config)");
		return PIRE_HIP_EINVAL;
	}
	const pire_hip_config c = GetConfig();
	const uint32_t n = std::min<uint32_t>(out->size, uint32_t(sizeof(c)));
	memcpy(out, &c, n);
	out->size = n;
	return PIRE_HIP_OK;
}

int pire_hip_config_set(const pire_hip_config* in)
{
	if (!in || in->size < sizeof(uint32_t)) {
		SetError("pire_hip_config_set: in->size must hold the caller's sizeof(pire_hip_config)");
		return PIRE_HIP_EINVAL;
	}
	std::lock_guard<std::mutex> lock(g_cfgMutex);
	pire_hip_config& c = ConfigStorage();
	memcpy(&c, in, std::min<size_t>(in->size, sizeof(c)));
	c.size = sizeof(c);
	return PIRE_HIP_OK;
}



const char* pire_hip_last_error(void) { return g_error.c_str(); }
const char* pire_hip_last_kernel(void) { return g_lastKernel; }
const char* pire_hip_last_kernel_symbol(void) { return g_lastSymbol ? g_lastSymbol : g_lastKernel; }

int pire_hip_set_timing(int enabled)
try {
	g_timing = enabled != 0;
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

float pire_hip_last_kernel_ms(void) { return g_lastMs; }

int pire_hip_device_count(void)
try {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return n;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_create(const void* save_blob, size_t len, pire_hip_table** out)
try {
	if (!out) {
		SetError("null out pointer");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	std::unique_ptr<pire_hip_table> t(new (std::nothrow) pire_hip_table);
	if (!t) {
		SetError("out of memory");
		return PIRE_HIP_ENOMEM;
	}
	if (int rc = BuildHostTable(save_blob, len, &t->host))
		return rc;
	*out = t.release();
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_mmap(const void* image, size_t size, pire_hip_table** out, size_t* consumed)
try {
	if (consumed)
		*consumed = 0;
	const int rc = pire_hip_table_create(image, size, out);
	if (rc == PIRE_HIP_OK && consumed)
		*consumed = size_t((*out)->host.blobBytes);
	return rc;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_create_from_file(const char* path, pire_hip_table** out)
try {
	if (!path || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) {
		SetError(std::string("cannot open ") + path + ": " + strerror(errno));
		return PIRE_HIP_EINVAL;
	}
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size <= 0) {
		SetError(std::string("cannot stat (or empty file) ") + path);
		close(fd);
		return PIRE_HIP_EFORMAT;
	}
	void* map = mmap(nullptr, size_t(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (map == MAP_FAILED) {
		SetError(std::string("cannot mmap ") + path + ": " + strerror(errno));
		return PIRE_HIP_ENOMEM;
	}
	const int rc = pire_hip_table_create(map, size_t(st.st_size), out);   // the image is decoded, nothing aliases it
	munmap(map, size_t(st.st_size));
	return rc;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

namespace {

int GlueImpl(const pire_hip_table* lhs, const pire_hip_table* rhs, size_t max_size, pire_hip_table** out, bool onDevice)
{
	if (!lhs || !rhs || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	std::unique_ptr<pire_hip_table> t(new (std::nothrow) pire_hip_table);
	if (!t) {
		SetError("out of memory");
		return PIRE_HIP_ENOMEM;
	}
	if (int rc = GlueHostTables(lhs->host, rhs->host, max_size, &t->host, onDevice))
		return rc;
	*out = t.release();
	return PIRE_HIP_OK;
}

}  // namespace

int pire_hip_table_glue_gpu(const pire_hip_table* lhs, const pire_hip_table* rhs, size_t max_size, pire_hip_table** out)
try {
	return GlueImpl(lhs, rhs, max_size, out, true);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_glue(const pire_hip_table* lhs, const pire_hip_table* rhs, size_t max_size, pire_hip_table** out)
try {
	return GlueImpl(lhs, rhs, max_size, out, false);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}


int pire_hip_table_upload(pire_hip_table* t)
try {
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	DeviceTable image;
	if (int rc = UploadTable(t, &image))
		return rc;
	// ... and what the counting walk of the same table (pire_hip_run_half_final) keeps on the device, when the table
	// qualifies for its row kernel: an ON_DEVICE call after this uploads nothing
	if (!t->host.empty && t->host.regexps && t->host.regexps <= 8)
		return UploadHalfRows(t);
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_config_set(pire_hip_table* t, const pire_hip_config* in)
try {
	if (!t || (in && in->size < sizeof(uint32_t))) {
		SetError("pire_hip_table_config_set: null table, or in->size does not hold the caller's sizeof(pire_hip_config)");
		return PIRE_HIP_EINVAL;
	}
	const pire_hip_config now = GetConfig();
	std::unique_lock<std::shared_mutex> exclusive(t->adaptMutex);   // no entry point is looking at it
	if (!in) {
		t->hasConfig = false;
		return PIRE_HIP_OK;
	}
	t->config = now;   // fields beyond in->size: the process-wide values of this moment
	memcpy(&t->config, in, std::min<size_t>(in->size, sizeof(t->config)));
	t->config.size = sizeof(t->config);
	t->hasConfig = true;
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_table_config_get(const pire_hip_table* t, pire_hip_config* out)
try {
	if (!t || !out || out->size < sizeof(uint32_t)) {
		SetError("pire_hip_table_config_get: null argument, or out->size does not hold the caller's sizeof(pire_hip_config)");
		return PIRE_HIP_EINVAL;
	}
	pire_hip_config c = GetConfig();
	{
		std::shared_lock<std::shared_mutex> stable(const_cast<pire_hip_table*>(t)->adaptMutex);
		if (t->hasConfig)
			c = t->config;
	}
	const uint32_t n = std::min<uint32_t>(out->size, uint32_t(sizeof(c)));
	memcpy(out, &c, n);
	out->size = n;
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_table_adapt(pire_hip_table* t, uint32_t* changed_rows)
try {
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	return AdaptTable(t, changed_rows);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_check_failures(pire_hip_table* t, uint64_t* out)
try {
	if (!t || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	return CheckFailures(t, out);
} catch (...) {
	return pirehip::HandleException();
}

void pire_hip_table_destroy(pire_hip_table* t)
{
	if (!t)
		return;
	JoinBackgroundAdapt(t);
	FreeAllDeviceTables(t);
	FreeHalfRows(t);
	if (t->segProduct)
		FreeAllDeviceTables(t->segProduct.get());   // the product automaton of the segmented scan's two modes
	delete t;
}

uint32_t pire_hip_abi_version(void) { return PIRE_HIP_ABI_VERSION; }

int pire_hip_table_get_info_sized(const pire_hip_table* t, void* out, size_t size)
try {
	if (!t || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	pire_hip_table_info full;
	if (int rc = pire_hip_table_get_info(t, &full))
		return rc;
	memcpy(out, &full, std::min(size, sizeof(full)));
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_table_get_info(const pire_hip_table* t, pire_hip_table_info* out)
try {
	if (!t || !out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	EnsureRanked(const_cast<pire_hip_table*>(t));   // hot_states / lds_table_bytes describe the ranked device layout
	std::shared_lock<std::shared_mutex> stable(const_cast<pire_hip_table*>(t)->adaptMutex);
	const HostTable& h = t->host;
	memset(out, 0, sizeof(*out));
	out->abi_version = PIRE_HIP_ABI_VERSION;
	out->states = h.states;
	out->letters = h.letters;
	out->regexps = h.regexps;
	out->initial = h.initial;
	out->empty = h.empty ? 1 : 0;
	out->header_size = h.headerSize;
	out->row_stride = h.rowStride;
	out->hot_states = h.hot;
	out->lds_table_bytes = MakeLayout(h.hot, 0, 256u, h.compact ? (h.compact + 1) * CompactPitch(h.letters) : 0).total;
	out->compact_states = h.compact;
	out->scanner_type = h.scannerType;
	out->zip_full_states = h.zipFull;
	out->zip_outside_share = h.zipOutside;
	out->zip_plain_outside_share = h.zipPlainOutside;
	out->wide_states = h.wide;
	out->wide_lds_bytes = h.wide ? MakeWideLayout(h.wide, h.letters, 0, h.zipFull).total : 0;
	out->outside_dense_share = h.outsideDense;
	out->outside_wide_share = h.outsideWide;
	out->shares_measured = h.massMeasured ? 1 : 0;
	out->last_wide_trap_chunks = h.lastWideTrapChunks;
	out->wide_outside_chunk_share = h.wideTwiceShare;
	out->device_bytes = 0;   // all images (one per device the table has run on)
	{
		// devs[] is written by the first run on a device, under uploadMutex (found by ThreadSanitizer, round 4: this loop
		// read it under the adaptation lock alone)
		std::lock_guard<std::mutex> images(const_cast<pire_hip_table*>(t)->uploadMutex);
		for (int k = 0; k < kMaxDevices; ++k)
			out->device_bytes += t->devs[k].device >= 0 ? t->devs[k].bytes : 0;
	}
	out->adaptations = h.adaptations;
	out->last_trap_samples = h.lastTrapSamples;
	out->ref_buf_size = h.refBufSize;
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

static int CheckIdx(const pire_hip_table* t, uint32_t idx)
{
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	if (idx >= t->host.states) {
		SetError("state index out of range");
		return PIRE_HIP_EINVAL;
	}
	return PIRE_HIP_OK;
}

int pire_hip_table_final(const pire_hip_table* t, uint32_t idx)
try {
	if (int rc = CheckIdx(t, idx))
		return rc;
	return (t->host.flags[idx] & kFinal) ? 1 : 0;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_dead(const pire_hip_table* t, uint32_t idx)
try {
	if (int rc = CheckIdx(t, idx))
		return rc;
	return (t->host.flags[idx] & kDead) ? 1 : 0;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_accepted_regexps(const pire_hip_table* t, uint32_t idx, const uint64_t** begin, size_t* count)
try {
	if (int rc = CheckIdx(t, idx))
		return rc;
	if (!begin || !count) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	const HostTable& h = t->host;
	*begin = h.acceptIds.data() + h.acceptOff[idx];
	*count = size_t(h.acceptOff[idx + 1] - h.acceptOff[idx]);
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_letter_class(const pire_hip_table* t, uint32_t ch)
try {
	if (!t || ch >= kMaxChar) {
		SetError("bad argument");
		return PIRE_HIP_EINVAL;
	}
	return t->host.cls[ch];
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int64_t pire_hip_table_next(const pire_hip_table* t, uint32_t idx, uint32_t ch)
{
	if (int rc = CheckIdx(t, idx))
		return rc;
	if (ch >= kMaxCharUnaligned || ch == kEpsilon) {
		SetError("character out of range");
		return PIRE_HIP_EINVAL;
	}
	const HostTable& h = t->host;
	return h.next[size_t(idx) * h.letters + h.cls[ch]];
}

int pire_hip_table_layout(const pire_hip_table* t, uint32_t* orig_of_perm, uint8_t* hot_rows)
try {
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	EnsureRanked(const_cast<pire_hip_table*>(t));
	const HostTable& h = t->host;
	if (orig_of_perm)
		memcpy(orig_of_perm, h.origOfPerm.data(), h.origOfPerm.size() * sizeof(uint32_t));
	if (hot_rows)
		memcpy(hot_rows, h.hotRows.data(), h.hotRows.size());
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_wide_layout(const pire_hip_table* t, uint16_t* rows, size_t cap, uint32_t* wide_states, uint32_t* pitch,
                               uint32_t* rows_offset)
try {
	if (!t) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	EnsureRanked(const_cast<pire_hip_table*>(t));
	std::shared_lock<std::shared_mutex> stable(const_cast<pire_hip_table*>(t)->adaptMutex);
	const HostTable& h = t->host;
	const uint32_t wide = h.zipFull ? 0 : h.wide;   // a zipped image: pire_hip_table_zip_layout
	const WideLayout wl = MakeWideLayout(wide, h.letters, 0);
	if (wide_states)
		*wide_states = wide;
	if (pitch)
		*pitch = wl.pitch;
	if (rows_offset)
		*rows_offset = wl.rowsOff;
	if (rows && wide) {
		const std::vector<uint16_t> img = BuildWideRows(h);
		const size_t want = size_t(wide + 1) * wl.pitch / 2;
		if (cap < want) {
			SetError("wide layout: buffer too small");
			return PIRE_HIP_EINVAL;
		}
		memcpy(rows, img.data(), want * sizeof(uint16_t));
	}
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_table_zip_layout(const pire_hip_table* t, uint16_t* image, size_t cap, uint32_t geometry[8])
try {
	if (!t || !geometry) {
		SetError("null table / geometry");
		return PIRE_HIP_EINVAL;
	}
	EnsureRanked(const_cast<pire_hip_table*>(t));
	std::shared_lock<std::shared_mutex> stable(const_cast<pire_hip_table*>(t)->adaptMutex);
	const HostTable& h = t->host;
	memset(geometry, 0, 8 * sizeof(uint32_t));
	if (!h.zipFull)
		return PIRE_HIP_OK;
	const WideLayout wl = MakeWideLayout(h.wide, h.letters, 0, h.zipFull);
	const uint32_t g[8] = {h.wide, h.zipFull, wl.pitch, wl.rowsOff, wl.hOff, wl.xOff, wl.imageEnd, kZipExceptions};
	memcpy(geometry, g, sizeof(g));
	if (image) {
		const std::vector<uint16_t> img = BuildWideRows(h);
		const size_t want = (wl.imageEnd - wl.rowsOff) / 2;
		if (cap < want) {
			SetError("zip layout: buffer too small");
			return PIRE_HIP_EINVAL;
		}
		memcpy(image, img.data(), want * sizeof(uint16_t));
	}
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_run(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                 const uint32_t* init_state_idx, uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts,
                 void* stream)
try {
	if (n && !offsets) {
		SetError("null offsets");
		return PIRE_HIP_EINVAL;
	}
	if (offsets && !(flags & PIRE_HIP_RUN_ON_DEVICE) && n) {
		// fixed-length, 16-byte-friendly batches handed over as offsets still get the tiled kernel
		const uint64_t len = offsets[1] - offsets[0];
		bool uniform = offsets[0] == 0 && len >= 128 && len % 16 == 0;
		for (uint64_t i = 1; uniform && i < n; ++i)
			uniform = offsets[i + 1] - offsets[i] == len;
		if (uniform)
			return RunImpl(t, text, nullptr, n, len, len, flags, init_state_idx, out_state_idx, out_final,
			               out_counts, stream);
	}
	return RunImpl(t, text, offsets, n, 0, 0, flags, init_state_idx, out_state_idx, out_final, out_counts, stream);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_run_strided(pire_hip_table* t, const void* text, uint64_t n, uint64_t len, uint64_t stride,
                         uint32_t flags, const uint32_t* init_state_idx, uint32_t* out_state_idx, uint8_t* out_final,
                         uint64_t* out_counts, void* stream)
try {
	if (stride < len) {
		SetError("stride smaller than len");
		return PIRE_HIP_EINVAL;
	}
	return RunImpl(t, text, nullptr, n, len, stride, flags, init_state_idx, out_state_idx, out_final, out_counts,
	               stream);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

// ---- first-use self-tests of the entry points with actions (selftest.h) -------------------------------------------------
namespace {

enum : uint32_t { kEntryPrefix = 8, kEntrySuffix = 9, kEntryHalfFinal = 10, kEntryPair = 11 };   // bits of pire_hip_table::selfTested

struct OwnStream {
	hipStream_t s = nullptr;
	OwnStream() { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
	~OwnStream()
	{
		if (s) {
			(void)hipStreamSynchronize(s);
			(void)hipStreamDestroy(s);
		}
	}
};

// the table's transitions, flags and final lists in the reference's numbering: they never change (an adaptation renumbers
// the DEVICE side only)
struct HostWalk {
	const HostTable& h;
	uint32_t Next(uint32_t st, uint32_t ch) const { return h.next[size_t(st) * h.letters + h.cls[ch]]; }
	bool Final(uint32_t st) const { return (h.flags[st] & kFinal) != 0; }
	bool Dead(uint32_t st) const { return (h.flags[st] & kDead) != 0; }
};

bool EntryTested(pire_hip_table* t, uint32_t bit, int* dev)
{
	*dev = -1;
	if (hipGetDevice(dev) != hipSuccess || *dev < 0 || *dev >= kMaxDevices)
		return true;   // (no device: the entry point says so itself)
	return (t->selfTested[*dev].load(std::memory_order_relaxed) & (1u << bit)) != 0;
}

// LongestPrefix / ShortestPrefix as the reference walks them (run.h:69-100, 277-311), on the host table
int64_t HostPrefix(const HostWalk& w, const uint8_t* b, const uint8_t* e, bool longest, bool throughBegin, bool throughEnd)
{
	uint32_t st = w.h.initial;
	if (throughBegin)
		st = w.Next(st, kBeginMark);
	int64_t pos = -1;
	if (longest) {
		if (w.Final(st))
			pos = 0;
		for (const uint8_t* p = b; p != e; ++p) {
			st = w.Next(st, *p);
			if (w.Final(st))
				pos = p + 1 - b;
			if (w.Dead(st))
				break;
		}
		if (throughEnd && w.Final(w.Next(st, kEndMark)))
			pos = e - b;
		return pos;
	}
	if (w.Final(st))
		return 0;
	for (const uint8_t* p = b; p != e; ++p) {
		st = w.Next(st, *p);
		if (w.Final(st))
			return p + 1 - b;
		if (w.Dead(st))
			break;
	}
	if (throughEnd && w.Final(w.Next(st, kEndMark)))
		return e - b;
	return -1;
}

// LongestSuffix / ShortestSuffix (run.h:313-362): backwards from the last byte
int64_t HostSuffix(const HostWalk& w, const uint8_t* first, const uint8_t* last1, bool longest, bool throughEnd, bool throughBegin)
{
	uint32_t st = w.h.initial;
	if (throughEnd)
		st = w.Next(st, kEndMark);
	const uint8_t* r = last1;   // one past the byte the walk takes next (going down)
	if (longest) {
		int64_t pos = -1;
		while (r != first && !w.Dead(st)) {
			if (w.Final(st))
				pos = last1 - r;
			st = w.Next(st, r[-1]);
			--r;
		}
		if (w.Final(st))
			pos = last1 - r;
		if (throughBegin && w.Final(w.Next(st, kBeginMark)))
			pos = last1 - r;
		return pos;
	}
	for (; r != first && !w.Final(st) && !w.Dead(st); --r)
		st = w.Next(st, r[-1]);
	if (throughBegin)
		st = w.Next(st, kBeginMark);
	return w.Final(st) ? int64_t(last1 - r) : -1;
}

KnownBatch TableBatch(const HostWalk& w, uint32_t n, uint32_t maxLen, uint32_t start, uint64_t seed)
{
	return MakeKnownBatch(n, maxLen, start, seed ^ (uint64_t(w.h.states) << 20) ^ w.h.letters,
	                      [&](uint32_t st, uint32_t ch) { return w.Next(st, ch); }, [&](uint32_t st) { return w.Dead(st); });
}

int SelfTestPrefix(pire_hip_table* t, bool suffix, int a, int b, hipStream_t stream)
{
	uint32_t mode = 0;
	int dev = -1;
	const uint32_t bit = suffix ? kEntrySuffix : kEntryPrefix;
	if (t->host.empty || EntryTested(t, bit, &dev) || !EntrySelfTestDue(stream, &mode))
		return PIRE_HIP_OK;
	const HostWalk w{t->host};
	// (prefix: a = throughBegin, b = throughEnd; suffix: a = throughEnd, b = throughBegin -- the mark the walk starts with first)
	const uint32_t start = a ? w.Next(t->host.initial, suffix ? kEndMark : kBeginMark) : t->host.initial;
	const KnownBatch kb = TableBatch(w, 320, 200, start, suffix ? 2 : 1);
	OwnStream own;
	for (int longest = 0; longest < 2; ++longest) {
		std::vector<int64_t> want(kb.n), got(kb.n);
		for (uint32_t i = 0; i < kb.n; ++i) {
			const uint8_t* s0 = kb.text.data() + kb.offsets[i];
			const uint8_t* s1 = kb.text.data() + kb.offsets[i + 1];
			want[i] = suffix ? HostSuffix(w, s0, s1, longest != 0, a != 0, b != 0) : HostPrefix(w, s0, s1, longest != 0, a != 0, b != 0);
		}
		if (mode == 2)
			want[kb.n / 2] += 1;
		std::vector<std::function<void(pire_hip_config&)>> variants;
		variants.push_back([](pire_hip_config& c) { c.ragged_act_always = 1; c.no_ragged_act = 0; });   // the ragged kernel with actions
		if (!suffix)
			variants.push_back([](pire_hip_config& c) { c.no_ragged_act = 1; c.ragged_act_always = 0; });   // one string per lane
		if (!suffix && t->host.wide && t->host.states <= 65536)   // ... and the ragged kernel with actions on the class-indexed walk
			variants.push_back([](pire_hip_config& c) { c.ragged_act_always = 1; c.no_ragged_act = 0; c.walk_variant = 2; c.tiled_variant = 0; c.checked = 0; });
		const int rc = RunSelfTestVariants(variants, [&]() -> int {
			std::fill(got.begin(), got.end(), int64_t(-77));
			const int r = suffix ? pire_hip_suffix(t, kb.text.data(), kb.offsets.data(), kb.n, longest, a, b, 0, got.data(), own.s)
			                     : pire_hip_prefix(t, kb.text.data(), kb.offsets.data(), kb.n, longest, a, b, 0, got.data(), own.s);
			if (r != PIRE_HIP_OK)
				return r;
			for (uint32_t i = 0; i < kb.n; ++i)
				if (got[i] != want[i])
					return SelfTestMismatch(suffix ? (longest ? "LongestSuffix" : "ShortestSuffix") : (longest ? "LongestPrefix" : "ShortestPrefix"),
					                        i, std::to_string(got[i]), std::to_string(want[i]));
			return PIRE_HIP_OK;
		});
		if (rc != PIRE_HIP_OK)
			return rc;
	}
	t->selfTested[dev].fetch_or(1u << bit);
	return PIRE_HIP_OK;
}

// HalfFinalScanner (half_final.h:137-164): Initialize and every Step end with TakeAction
int SelfTestHalfFinal(pire_hip_table* t, uint32_t flags, hipStream_t stream)
{
	uint32_t mode = 0;
	int dev = -1;
	if (t->host.empty || EntryTested(t, kEntryHalfFinal, &dev) || !EntrySelfTestDue(stream, &mode))
		return PIRE_HIP_OK;
	const HostTable& h = t->host;
	const HostWalk w{h};
	const uint32_t R = h.regexps;
	flags &= PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END;
	const uint32_t start = (flags & PIRE_HIP_RUN_BEGIN) ? w.Next(h.initial, kBeginMark) : h.initial;
	const KnownBatch kb = TableBatch(w, 320, 200, start, 3);
	std::vector<uint32_t> wantIdx(kb.n), wantRes(size_t(kb.n) * R, 0), gotIdx(kb.n), gotRes(size_t(kb.n) * R);
	std::vector<uint8_t> wantFin(kb.n), gotFin(kb.n);
	for (uint32_t i = 0; i < kb.n; ++i) {
		uint32_t* m = wantRes.data() + size_t(i) * R;
		auto take = [&](uint32_t st) {
			if (w.Final(st))
				for (uint64_t k = h.acceptOff[st]; k < h.acceptOff[st + 1]; ++k)
					m[h.acceptIds[k]]++;
		};
		uint32_t st = h.initial;
		take(st);
		if (flags & PIRE_HIP_RUN_BEGIN)
			take(st = w.Next(st, kBeginMark));
		for (uint64_t k = kb.offsets[i]; k < kb.offsets[i + 1]; ++k)
			take(st = w.Next(st, kb.text[k]));
		if (flags & PIRE_HIP_RUN_END)
			take(st = w.Next(st, kEndMark));
		wantIdx[i] = st;
		wantFin[i] = w.Final(st) ? 1 : 0;
	}
	if (mode == 2)
		wantIdx[kb.n / 2] ^= 1;
	std::vector<std::function<void(pire_hip_config&)>> variants;
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 2; });                        // the row kernel where the table has that image
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; c.no_ragged_act = 0; });   // the ragged kernel with actions
	variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; c.no_ragged_act = 1; });   // one string per lane
	if (h.wide && h.states <= 65536)   // the ragged kernel with actions on the class-indexed walk
		variants.push_back([](pire_hip_config& c) { c.counting_variant = 1; c.no_ragged_act = 0; c.walk_variant = 2; c.tiled_variant = 0; c.checked = 0; });
	OwnStream own;
	const int rc = RunSelfTestVariants(variants, [&]() -> int {
		std::fill(gotIdx.begin(), gotIdx.end(), ~0u);
		std::fill(gotRes.begin(), gotRes.end(), ~0u);
		const int r = pire_hip_run_half_final(t, kb.text.data(), kb.offsets.data(), kb.n, flags, gotIdx.data(), gotFin.data(), gotRes.data(), own.s);
		if (r != PIRE_HIP_OK)
			return r;
		for (uint32_t i = 0; i < kb.n; ++i) {
			if (gotIdx[i] != wantIdx[i] || gotFin[i] != wantFin[i])
				return SelfTestMismatch("HalfFinalScanner", i, "state " + std::to_string(gotIdx[i]), "state " + std::to_string(wantIdx[i]));
			for (uint32_t r2 = 0; r2 < R; ++r2)
				if (gotRes[size_t(i) * R + r2] != wantRes[size_t(i) * R + r2])
					return SelfTestMismatch("HalfFinalScanner", i, "count " + std::to_string(gotRes[size_t(i) * R + r2]) + " for regexp " + std::to_string(r2),
					                        std::to_string(wantRes[size_t(i) * R + r2]));
		}
		return PIRE_HIP_OK;
	});
	if (rc != PIRE_HIP_OK)
		return rc;
	t->selfTested[dev].fetch_or(1u << kEntryHalfFinal);
	return PIRE_HIP_OK;
}

}  // namespace

int pire_hip_run_half_final(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, uint32_t flags,
                            uint32_t* out_state_idx, uint8_t* out_final, uint32_t* out_results, void* streamPtr)
try {
	if (!t || (n && (!offsets || !out_results))) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	if (n)
		if (int rc = SelfTestHalfFinal(t, flags, stream))   // first use on this device: every kernel of this entry point, known answers
			return rc;
	ScanParams p;
	const bool enqueueOnly = (flags & PIRE_HIP_RUN_ON_DEVICE) && !(flags & PIRE_HIP_RUN_HOST_OFFSETS);
	TableUse use(t, enqueueOnly);
	if (int rc = FillParams(t, &p, 0, /*wantDist=*/true))   // startPerm = Initialize(); Begin() is a counted step in the kernel
		return rc;
	p.flags = flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END);
	p.n = n;
	if (n == 0)
		return PIRE_HIP_OK;
	const uint32_t R = t->host.regexps;
	p.actDist = p.distFinalPerm;
	{   // Initialize() and Begin() of a HalfFinalScanner end with TakeAction: walked once here (ScanParams::hfStart)
		const HostTable& h = t->host;
		uint32_t o = h.initial;
		auto take = [&](uint32_t st) {
			if ((h.flags[st] & kFinal) && h.incPacked)
				for (int r = 0; r < 8; ++r)
					p.hfStartC[r] += uint32_t(h.inc64[st] >> (8 * r)) & 0xFFu;
		};
		take(o);
		if (flags & PIRE_HIP_RUN_BEGIN) {
			o = h.next[size_t(o) * h.letters + h.cls[kBeginMark]];
			take(o);
		}
		p.hfStart = p.hostPermOfOrig[o];   // the numbering of the image in p (FillParams), not whatever t->host holds by now
	}
	// PIRE_HIP_RUN_GENERIC keeps the one-string-per-lane kernel (tests compare the two)
	const bool exactOnly = (flags & PIRE_HIP_RUN_GENERIC) != 0;
	// few long strings: the segmented scan resolves every segment's true start state, then the segments are counted
	// in parallel (segmented.hip); needs the lengths on the host and counters that pack.  false: not done.
	auto segmented = [&](const uint64_t* hostOffsets, uint32_t* dResults, bool* done) -> int {
		*done = false;
		if (exactOnly || !p.incPerm || !SegmentedEligible(n, hostOffsets[n] - hostOffsets[0]))
			return PIRE_HIP_OK;
		ScanParams ps;
		if (int rc = FillParams(t, &ps, flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END)))   // Begin() folded into the start; the same image as p (`use`)
			return rc;
		ps.n = n;
		ps.text = p.text;
		ps.offsets = p.offsets;
		ps.outIdx = p.outIdx;
		ps.outFinal = p.outFinal;
		bool incomplete = false;
		if (int rc = RunSegmented(t, ps, hostOffsets, stream, dResults, &incomplete))
			return rc;
		*done = !incomplete;
		return PIRE_HIP_OK;
	};
	// Scanners whose walk is in a Final state most of the time gain nothing from the ragged kernel's look for the chunks
	// that hold one: those take the row kernel of the counting scanners (counting.hip: whole text lines per lane, the
	// increments of the target state as the step's action), and what its 16-bit counters cannot hold -- strings of more
	// than 65 000 bytes -- the one-string-per-lane kernel, from the list the row kernel leaves.
	auto launchCounting = [&](uint32_t* dResults) -> int {
		// (the share of the byte model's visits that fall on Final states decides, as for the capturing scanners: at 0.2 %
		// of the steps a third of the ragged kernel's chunks are walked again)
		if (!exactOnly && (!RaggedActEligible(p) || p.finalShare > 0.002f)) {
			bool done = false;
			uint32_t* list = nullptr;
			if (int rc = LaunchHalfFinalRows(t, p.text, p.offsets, n, flags, p.outIdx, p.outFinal, dResults, stream, &done, &list))
				return rc;
			if (done) {
				const int rc = LaunchHalfFinal(p, dResults, stream, nullptr, list);
				(void)hipFreeAsync(list, stream);
				return rc;
			}
		}
		return LaunchHalfFinal(p, dResults, stream, exactOnly ? nullptr : NextWorkSlot(t, p));
	};
	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		p.outIdx = out_state_idx;
		p.outFinal = out_final;
		if (!(flags & PIRE_HIP_RUN_HOST_OFFSETS))
			return launchCounting(out_results);
		// resident text, offsets known to the host (see pire_hip_run)
		for (uint64_t i = 0; i < n; ++i)
			if (offsets[i] > offsets[i + 1]) {
				SetError("offsets must be non-decreasing");
				return PIRE_HIP_EINVAL;
			}
		void* d = nullptr;
		hipError_t e = hipMallocAsync(&d, (n + 1) * 8, stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(offsets)");
		e = hipMemcpyAsync(d, offsets, (n + 1) * 8, hipMemcpyHostToDevice, stream);
		int rc = e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemcpy(offsets)");
		p.offsets = static_cast<const uint64_t*>(d);
		bool counted = false;
		if (!rc)
			rc = segmented(offsets, out_results, &counted);
		if (!rc && !counted)
			rc = launchCounting(out_results);
		(void)hipFreeAsync(d, stream);
		if (!rc) {
			e = hipStreamSynchronize(stream);   // the caller's offsets array was the source of an async copy
			if (e != hipSuccess)
				rc = HipFail(e, "hipStreamSynchronize");
		}
		return rc;
	}
	Staging st(stream);
	for (uint64_t i = 0; i < n; ++i)
		if (offsets[i] > offsets[i + 1]) {
			SetError("offsets must be non-decreasing");
			return PIRE_HIP_EINVAL;
		}
	const uint64_t textBytes = offsets[n];
	if (!text && textBytes) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	const uint8_t* dText = nullptr;
	if (int rc = st.In(static_cast<const uint8_t*>(text), size_t(textBytes), &dText, stream))
		return rc;
	p.text = dText;
	if (int rc = st.In(offsets, size_t(n + 1), &p.offsets, stream))
		return rc;
	void *dIdx = nullptr, *dFin = nullptr, *dRes = nullptr;
	if (int rc = st.Alloc(&dIdx, size_t(n) * 4))
		return rc;
	if (int rc = st.Alloc(&dFin, size_t(n)))
		return rc;
	if (int rc = st.Alloc(&dRes, size_t(n) * std::max<uint32_t>(R, 1) * 4))
		return rc;
	p.outIdx = static_cast<uint32_t*>(dIdx);
	p.outFinal = static_cast<uint8_t*>(dFin);
	bool counted = false;
	if (int rc = segmented(offsets, static_cast<uint32_t*>(dRes), &counted))
		return rc;
	if (!counted)
		if (int rc = launchCounting(static_cast<uint32_t*>(dRes)))
			return rc;
	int rc = st.Out(out_state_idx, dIdx, size_t(n) * 4);
	if (!rc)
		rc = st.Out(out_final, dFin, size_t(n));
	if (!rc && R)
		rc = st.Out(out_results, dRes, size_t(n) * R * 4);
	return rc ? rc : st.Finish();
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_prefix(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, int longest,
                    int through_begin, int through_end, uint32_t flags, int64_t* out_len, void* streamPtr)
try {
	if (!t || (n && (!offsets || !out_len))) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	if (n)
		if (int rc = SelfTestPrefix(t, false, through_begin, through_end, stream))   // first use on this device (selftest.h)
			return rc;
	ScanParams p;
	TableUse use(t, (flags & PIRE_HIP_RUN_ON_DEVICE) != 0);
	if (int rc = FillParams(t, &p, through_begin ? PIRE_HIP_RUN_BEGIN : 0, /*wantDist=*/true))   // startPerm = Initialize [+ BeginMark]
		return rc;
	p.n = n;
	if (n == 0)
		return PIRE_HIP_OK;
	p.actDist = p.distFlaggedPerm;
	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		return LaunchPrefix(p, longest != 0, through_end != 0, reinterpret_cast<long long*>(out_len), stream,
		                    (flags & PIRE_HIP_RUN_GENERIC) ? nullptr : NextWorkSlot(t, p));
	}
	Staging st(stream);
	for (uint64_t i = 0; i < n; ++i)
		if (offsets[i] > offsets[i + 1]) {
			SetError("offsets must be non-decreasing");
			return PIRE_HIP_EINVAL;
		}
	const uint64_t textBytes = offsets[n];
	if (!text && textBytes) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	const uint8_t* dText = nullptr;
	if (int rc = st.In(static_cast<const uint8_t*>(text), size_t(textBytes), &dText, stream))
		return rc;
	p.text = dText;
	if (int rc = st.In(offsets, size_t(n + 1), &p.offsets, stream))
		return rc;
	void* dOut = nullptr;
	if (int rc = st.Alloc(&dOut, size_t(n) * 8))
		return rc;
	if (int rc = LaunchPrefix(p, longest != 0, through_end != 0, static_cast<long long*>(dOut), stream,
	                          (flags & PIRE_HIP_RUN_GENERIC) ? nullptr : NextWorkSlot(t, p)))
		return rc;
	if (int rc = st.Out(out_len, dOut, size_t(n) * 8))
		return rc;
	return st.Finish();
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

// Pire::Run(scanner1, scanner2, state1, state2, begin, end) (run.h:229-241) / Runner over ScannerPair (pair.h:33-94).
static int RunPairImpl(pire_hip_table* t1, pire_hip_table* t2, const void* text, const uint64_t* offsets, uint64_t n,
                       uint64_t len, uint64_t stride, uint32_t flags, uint32_t* outIdx1, uint32_t* outIdx2,
                       uint8_t* outFinal, void* streamPtr)
{
	if (!t1 || !t2) {
		SetError("null table");
		return PIRE_HIP_EINVAL;
	}
	if (!(flags & PIRE_HIP_RUN_ON_DEVICE)) {
		SetError("pire_hip_run_pair takes device pointers (PIRE_HIP_RUN_ON_DEVICE): stage the text once, scan it with both");
		return PIRE_HIP_EINVAL;
	}
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	flags &= PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_ON_DEVICE | PIRE_HIP_RUN_GENERIC;
	ScanParams a, b;
	TableUse use1, use2;   // in address order: two calls with the tables swapped must not wait for each other
	use1.Acquire(t1 < t2 ? t1 : t2, true);
	if (t1 != t2)
		use2.Acquire(t1 < t2 ? t2 : t1, true);
	if (int rc = FillParams(t1, &a, flags))
		return rc;
	if (int rc = FillParams(t2, &b, flags))
		return rc;
	if (n == 0)
		return PIRE_HIP_OK;
	a.n = b.n = n;
	a.len = b.len = len;
	a.stride = b.stride = stride;
	a.text = b.text = static_cast<const uint8_t*>(text);
	a.offsets = b.offsets = offsets;
	a.outIdx = outIdx1;
	a.outFinal = outFinal;
	uint64_t fused = 0;
	if (!offsets && !(flags & PIRE_HIP_RUN_GENERIC) && PairTiledEligible(a, b)) {
		if (int rc = LaunchPairTiled(a, b, outIdx2, stream))
			return rc;
		fused = n & ~uint64_t(63);
		if (fused == n)
			return PIRE_HIP_OK;
	}
	// what is left (everything, when the batch is not made of fixed-length records): two ordinary passes, Final = either
	const uint64_t rest = n - fused;
	void* fin2 = nullptr;
	if (outFinal) {
		hipError_t e = hipMallocAsync(&fin2, size_t(rest), stream);
		if (e != hipSuccess)
			return HipFail(e, "hipMallocAsync(pair finals)");
	}
	auto shift = [&](ScanParams& p, uint32_t* idx, uint8_t* fin) {
		p.n = rest;
		if (!offsets)
			p.text += fused * stride;
		else
			p.offsets += fused;
		p.outIdx = idx ? idx + fused : nullptr;
		p.outFinal = fin;
	};
	shift(a, outIdx1, outFinal ? outFinal + fused : nullptr);
	shift(b, outIdx2, static_cast<uint8_t*>(fin2));
	int rc = Dispatch(a, stream, NextWorkSlot(t1, a), offsets ? ~0ull : 0);
	if (!rc)
		rc = Dispatch(b, stream, NextWorkSlot(t2, b), offsets ? ~0ull : 0);
	if (!rc && outFinal)
		rc = LaunchOrFinal(outFinal + fused, static_cast<const uint8_t*>(fin2), rest, stream);
	if (fin2)
		(void)hipFreeAsync(fin2, stream);
	return rc;
}

int pire_hip_run_pair(pire_hip_table* t1, pire_hip_table* t2, const void* text, const uint64_t* offsets, uint64_t n,
                      uint32_t flags, uint32_t* out_state_idx1, uint32_t* out_state_idx2, uint8_t* out_final, void* stream)
try {
	if (n && !offsets) {
		SetError("null offsets");
		return PIRE_HIP_EINVAL;
	}
	return RunPairImpl(t1, t2, text, offsets, n, 0, 0, flags, out_state_idx1, out_state_idx2, out_final, stream);
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_run_pair_strided(pire_hip_table* t1, pire_hip_table* t2, const void* text, uint64_t n, uint64_t len,
                              uint64_t stride, uint32_t flags, uint32_t* out_state_idx1, uint32_t* out_state_idx2,
                              uint8_t* out_final, void* stream)
try {
	if (stride < len) {
		SetError("stride smaller than len");
		return PIRE_HIP_EINVAL;
	}
	return RunPairImpl(t1, t2, text, nullptr, n, len, stride, flags, out_state_idx1, out_state_idx2, out_final, stream);
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_suffix(pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n, int longest,
                    int through_end, int through_begin, uint32_t flags, int64_t* out_len, void* streamPtr)
try {
	if (!t || (n && (!offsets || !out_len))) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	hipStream_t stream = static_cast<hipStream_t>(streamPtr);
	if (n)
		if (int rc = SelfTestPrefix(t, true, through_end, through_begin, stream))   // first use on this device (selftest.h)
			return rc;
	ScanParams p;
	TableUse use(t, (flags & PIRE_HIP_RUN_ON_DEVICE) != 0);
	if (int rc = FillParams(t, &p, 0))
		return rc;
	if (through_end) {   // Initialize, then Step(EndMark): run.h:319-320 / 349-350
		const HostTable& h = t->host;
		p.startPerm = p.hostPermOfOrig[h.next[size_t(h.initial) * h.letters + h.cls[kEndMark]]];
	}
	p.n = n;
	if (n == 0)
		return PIRE_HIP_OK;
	if (flags & PIRE_HIP_RUN_ON_DEVICE) {
		p.text = static_cast<const uint8_t*>(text);
		p.offsets = offsets;
		return LaunchSuffix(p, longest != 0, through_begin != 0, reinterpret_cast<long long*>(out_len), stream);
	}
	Staging st(stream);
	for (uint64_t i = 0; i < n; ++i)
		if (offsets[i] > offsets[i + 1]) {
			SetError("offsets must be non-decreasing");
			return PIRE_HIP_EINVAL;
		}
	const uint64_t textBytes = offsets[n];
	if (!text && textBytes) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	const uint8_t* dText = nullptr;
	if (int rc = st.In(static_cast<const uint8_t*>(text), size_t(textBytes), &dText, stream))
		return rc;
	p.text = dText;
	if (int rc = st.In(offsets, size_t(n + 1), &p.offsets, stream))
		return rc;
	void* dOut = nullptr;
	if (int rc = st.Alloc(&dOut, size_t(n) * 8))
		return rc;
	if (int rc = LaunchSuffix(p, longest != 0, through_begin != 0, static_cast<long long*>(dOut), stream))
		return rc;
	if (int rc = st.Out(out_len, dOut, size_t(n) * 8))
		return rc;
	return st.Finish();
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_step(pire_hip_table* t, uint32_t* state_idx, uint64_t n, uint32_t ch, void* stream)
try {
	if (!t || (n && !state_idx) || ch >= kMaxCharUnaligned || ch == kEpsilon) {
		SetError("bad argument");
		return PIRE_HIP_EINVAL;
	}
	ScanParams p;
	TableUse use(t, /*enqueueOnly=*/true);   // device arrays: the call only enqueues
	if (int rc = FillParams(t, &p, 0))
		return rc;
	return LaunchStep(p, state_idx, n, t->host.cls[ch], static_cast<hipStream_t>(stream));
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_host_alloc(size_t bytes, void** out)
try {
	if (!out) {
		SetError("null out pointer");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	hipError_t e = hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault);
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipHostMalloc");
} catch (...) {
	return pirehip::HandleException();
}

void pire_hip_host_free(void* p)
{
	if (p)
		(void)hipHostFree(p);
}

int pire_hip_device_alloc(size_t bytes, void** out)
try {
	if (!out) {
		SetError("null out pointer");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	hipError_t e = hipMalloc(out, bytes ? bytes : 16);
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMalloc");
} catch (...) {
	return pirehip::HandleException();
}

void pire_hip_device_free(void* p)
{
	if (p)
		(void)hipFree(p);
}

int pire_hip_copy_to_device(void* dst_device, const void* src_host, size_t bytes, void* stream)
try {
	hipError_t e = bytes ? hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream))
	                     : hipSuccess;
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemcpyAsync(H2D)");
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_copy_to_host(void* dst_host, const void* src_device, size_t bytes, void* stream)
try {
	hipError_t e = bytes ? hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream))
	                     : hipSuccess;
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemcpyAsync(D2H)");
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_memset_device(void* dst_device, int value, size_t bytes, void* stream)
try {
	hipError_t e = bytes ? hipMemsetAsync(dst_device, value, bytes, static_cast<hipStream_t>(stream)) : hipSuccess;
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipMemsetAsync");
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_stream_synchronize(void* stream)
try {
	hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
	return e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipStreamSynchronize");
} catch (...) {
	return pirehip::HandleException();
}

int pire_hip_corpus_fill(void* device_out, uint64_t seed, uint64_t first, uint64_t count, uint64_t len,
                         uint64_t stride, const void* plants, void* stream)
try {
	if (!device_out && count && len) {
		SetError("null output");
		return PIRE_HIP_EINVAL;
	}
	return LaunchCorpusFill(static_cast<uint8_t*>(device_out), seed, first, count, len, stride, plants,
	                        static_cast<hipStream_t>(stream));
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

}  // extern "C"
