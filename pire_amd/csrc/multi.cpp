// One process, several GPUs: a batch of strings sharded by string index over the devices of a node, the
// uint64[regexps+2] match counters summed with ONE all-reduce over RCCL (xGMI).  SURVEY.md section 8(e) / BASELINE
// config C4.  The reference has nothing like it (a single-threaded CPU library): the walk of a string depends on
// nothing but the table and its own bytes (pire/run.h:271-275), so the shards never talk to each other while
// scanning; the only exchange is the counter vector (80 bytes for 8 regexps), which is latency bound.
//
// RCCL is loaded at run time (dlopen): the product library keeps libamdhip64 as its only link-time dependency, and
// a box without librccl, a communicator that cannot be built (e.g. the same device listed twice, used by the tests on
// a one-GPU box) or any RCCL error falls back to summing the per-device counters on the host.

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "internal.h"

namespace {

using namespace pirehip;

// the slice of rccl.h this file needs (types only; the functions come from dlsym)
typedef struct ncclComm* ncclComm_t;
enum { kNcclSuccess = 0, kNcclUint64 = 5, kNcclSum = 0 };   // ncclResult_t / ncclDataType_t / ncclRedOp_t values, rccl.h:448-464

struct Rccl {
	void* handle = nullptr;
	int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	int (*CommDestroy)(ncclComm_t) = nullptr;
	int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	int (*GetVersion)(int*) = nullptr;
	int (*CommCount)(ncclComm_t, int*) = nullptr;
	int (*CommCuDevice)(ncclComm_t, int*) = nullptr;
	int (*CommUserRank)(ncclComm_t, int*) = nullptr;
	bool ok = false;
};

const Rccl& LoadRccl()
{
	static const Rccl r = [] {
		Rccl x;
		if (pirehip::GetConfig().no_rccl)   // knob: force the host reduce (tests); looked at once, when RCCL is first wanted
			return x;
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (x.handle)
				break;
		}
		if (!x.handle)
			return x;
		x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.handle, "ncclCommInitAll"));
		x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
		x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
		x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
		x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
		x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
		x.GetVersion = reinterpret_cast<decltype(x.GetVersion)>(dlsym(x.handle, "ncclGetVersion"));
		x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(x.handle, "ncclCommCount"));
		x.CommCuDevice = reinterpret_cast<decltype(x.CommCuDevice)>(dlsym(x.handle, "ncclCommCuDevice"));
		x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(dlsym(x.handle, "ncclCommUserRank"));
		x.ok = x.CommInitAll && x.CommDestroy && x.AllReduce && x.GroupStart && x.GroupEnd;
		return x;
	}();
	return r;
}

constexpr uint32_t kMaxCounters = kMaxLdsCountRegexps + 2;

}  // namespace

// Device-side staging of the host-pointer entry points, kept between calls (grown on demand, freed with the runner):
// round 2 allocated and freed four buffers per device and call.
struct StagePool {
	void* buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // text, offsets, init, idx, fin
	size_t cap[5] = {0, 0, 0, 0, 0};
};
enum { kStageText = 0, kStageOffs, kStageInit, kStageIdx, kStageFin };

struct pire_hip_multi {
	std::vector<int> devices;
	std::vector<hipStream_t> streams;
	std::vector<unsigned long long*> counters;   // [kMaxCounters] u64 on each device
	std::vector<ncclComm_t> comms;               // empty: host reduce
	std::vector<unsigned long long> hostCounters;
	std::vector<StagePool> pools;                // one per device slot
	std::vector<uint64_t> lastSplit;             // first string of every shard of the last host-pointer call, then n
	std::string backend;
};

namespace {

// RAII: whatever device the caller had current is current again when the entry point returns
struct DeviceGuard {
	int saved = -1;
	DeviceGuard() { (void)hipGetDevice(&saved); }
	~DeviceGuard()
	{
		if (saved >= 0)
			(void)hipSetDevice(saved);
	}
};

void DestroyMulti(pire_hip_multi* m)
{
	const Rccl& r = LoadRccl();
	for (size_t g = 0; g < m->devices.size(); ++g) {
		(void)hipSetDevice(m->devices[g]);
		if (g < m->comms.size() && m->comms[g] && r.ok)
			(void)r.CommDestroy(m->comms[g]);
		if (g < m->streams.size() && m->streams[g])
			(void)hipStreamDestroy(m->streams[g]);
		if (g < m->counters.size() && m->counters[g])
			(void)hipFree(m->counters[g]);
		if (g < m->pools.size())
			for (void* q : m->pools[g].buf)
				if (q)
					(void)hipFree(q);
	}
	delete m;
}

// Sum of the per-device counters into out[0..count): RCCL all-reduce (every device ends up with the sum, device 0's
// copy is read back) or, without a communicator, one small copy per device and a host loop.
int ReduceCounters(pire_hip_multi* m, uint32_t count, uint64_t* out)
{
	const size_t G = m->devices.size();
	const Rccl& r = LoadRccl();
	bool reduced = false;
	if (!m->comms.empty() && r.ok) {
		int rc = r.GroupStart();
		for (size_t g = 0; g < G && rc == kNcclSuccess; ++g)
			rc = r.AllReduce(m->counters[g], m->counters[g], count, kNcclUint64, kNcclSum, m->comms[g], m->streams[g]);
		const int rcEnd = r.GroupEnd();
		if (rc == kNcclSuccess && rcEnd == kNcclSuccess) {
			reduced = true;
		} else {
			// a failed collective leaves the counters per-device: drop the communicator, sum on the host instead
			for (size_t g = 0; g < G; ++g) {
				(void)hipSetDevice(m->devices[g]);
				(void)hipStreamSynchronize(m->streams[g]);
			}
			m->comms.clear();
			m->backend = "host (RCCL all-reduce failed)";
		}
	}
	m->hostCounters.assign(size_t(G) * count, 0);
	const size_t readers = reduced ? 1 : G;
	for (size_t g = 0; g < readers; ++g) {
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e == hipSuccess)
			e = hipMemcpyAsync(&m->hostCounters[g * count], m->counters[g], size_t(count) * 8, hipMemcpyDeviceToHost,
			                   m->streams[g]);
		if (e != hipSuccess)
			return HipFail(e, "reading the match counters back");
	}
	for (size_t g = 0; g < G; ++g) {
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e == hipSuccess)
			e = hipStreamSynchronize(m->streams[g]);
		if (e != hipSuccess)
			return HipFail(e, "hipStreamSynchronize");
	}
	for (uint32_t i = 0; i < count; ++i) {
		uint64_t s = 0;
		for (size_t g = 0; g < readers; ++g)
			s += m->hostCounters[g * count + i];
		out[i] = s;
	}
	return PIRE_HIP_OK;
}


constexpr size_t kStageSlack = 4096;   // the kernels read whole 16-byte blocks around a string's ends (api.cpp kHostChunkSlack)

// The current device must be device slot g's.  Grows buffer `which` of slot g to at least `bytes`.
int ReserveStage(pire_hip_multi* m, size_t g, int which, size_t bytes, void** out)
{
	StagePool& p = m->pools[g];
	if (p.cap[which] < bytes) {
		if (p.buf[which]) {
			// nothing of an earlier call is still in flight: every entry point drains its streams before it returns
			(void)hipFree(p.buf[which]);
			p.buf[which] = nullptr;
			p.cap[which] = 0;
		}
		const size_t want = bytes + bytes / 4 + 4096;
		hipError_t e = hipMalloc(&p.buf[which], want);
		if (e != hipSuccess)
			return HipFail(e, "hipMalloc(multi-GPU staging)");
		p.cap[which] = want;
	}
	*out = p.buf[which];
	return PIRE_HIP_OK;
}

// After an error on one device: the devices launched before it keep writing the caller's shard outputs and the
// runner's counters -- wait for them before the error is returned.
void DrainAll(pire_hip_multi* m)
{
	for (size_t g = 0; g < m->devices.size(); ++g)
		if (hipSetDevice(m->devices[g]) == hipSuccess)
			(void)hipStreamSynchronize(m->streams[g]);
	(void)hipGetLastError();
}

// Launch one scan per device (launch(g) enqueues device g's shard on m->streams[g]), then reduce the counters or drain.
// `late(g)`: true for a shard whose launch BLOCKS (the segmented scan of few long strings follows its chain on the host
// and synchronises its stream): those run in a second round, after every other device has its shard enqueued (ADVICE r4:
// launched in device order, device g + 1 did not start before such a shard g was over).
struct NoLateShards {
	bool operator()(size_t) const { return false; }
};

template <class Launch, class Late = NoLateShards>
int RunOnAllDevices(pire_hip_multi* m, pire_hip_table* t, uint64_t* out_counts, Launch launch, Late late = Late())
{
	pire_hip_table_info info;
	if (int rc = pire_hip_table_get_info(t, &info))
		return rc;
	const uint32_t count = info.regexps + 2;
	if (out_counts && count > kMaxCounters) {
		SetError("out_counts is supported for scanners with at most 1024 regexps");
		return PIRE_HIP_EUNSUPPORTED;
	}
	const size_t G = m->devices.size();
	// every device starts its shard before any is waited for: the launches are asynchronous; the shards whose launch
	// blocks come last
	for (int round = 0; round < 2; ++round)
		for (size_t g = 0; g < G; ++g) {
			if (late(g) != (round == 1))
				continue;
			hipError_t e = hipSetDevice(m->devices[g]);
			if (e == hipSuccess && out_counts)
				e = hipMemsetAsync(m->counters[g], 0, size_t(count) * 8, m->streams[g]);
			int rc = e == hipSuccess ? PIRE_HIP_OK : HipFail(e, "hipSetDevice / counter reset");
			if (rc == PIRE_HIP_OK)
				rc = launch(g, out_counts ? reinterpret_cast<uint64_t*>(m->counters[g]) : nullptr);
			if (rc != PIRE_HIP_OK) {
				const std::string msg = pire_hip_last_error();   // DrainAll must not replace the message
				DrainAll(m);
				SetError(msg);
				return rc;
			}
		}
	if (out_counts)
		return ReduceCounters(m, count, out_counts);
	for (size_t g = 0; g < G; ++g) {
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e == hipSuccess)
			e = hipStreamSynchronize(m->streams[g]);
		if (e != hipSuccess)
			return HipFail(e, "hipStreamSynchronize");
	}
	return PIRE_HIP_OK;
}

}  // namespace

extern "C" {

int pire_hip_multi_create(const int* devices, int ndev, pire_hip_multi** out)
try {
	if (!out) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	*out = nullptr;
	int have = 0;
	hipError_t e = hipGetDeviceCount(&have);
	if (e != hipSuccess || have <= 0) {
		(void)hipGetLastError();
		SetError("no HIP device");
		return PIRE_HIP_ENODEVICE;
	}
	if (ndev <= 0 || !devices) {   // all devices of the node (or the first ndev of them)
		ndev = ndev > 0 && ndev < have ? ndev : have;
		devices = nullptr;
	}
	if (ndev > kMaxDevices) {
		SetError("too many devices");
		return PIRE_HIP_EINVAL;
	}
	DeviceGuard guard;
	std::unique_ptr<pire_hip_multi, void (*)(pire_hip_multi*)> m(new (std::nothrow) pire_hip_multi, DestroyMulti);
	if (!m) {
		SetError("out of memory");
		return PIRE_HIP_ENOMEM;
	}
	bool distinct = true;
	for (int g = 0; g < ndev; ++g) {
		const int d = devices ? devices[g] : g;
		if (d < 0 || d >= have) {
			SetError("device ordinal out of range");
			return PIRE_HIP_EINVAL;
		}
		for (int k : m->devices)
			distinct = distinct && k != d;
		m->devices.push_back(d);
	}
	m->streams.assign(ndev, nullptr);
	m->counters.assign(ndev, nullptr);
	m->pools.assign(ndev, StagePool());
	for (int g = 0; g < ndev; ++g) {
		if ((e = hipSetDevice(m->devices[g])) != hipSuccess ||
		    (e = hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking)) != hipSuccess ||
		    (e = hipMalloc(reinterpret_cast<void**>(&m->counters[g]), kMaxCounters * 8)) != hipSuccess)
			return HipFail(e, "setting up a device of the multi-GPU runner");
	}
	m->backend = "host";
	const Rccl& r = LoadRccl();
	// one device: the counters are the sum already; pire_hip_config.force_rccl builds the one-rank communicator all the
	// same, so that the RCCL path (the dlsym'ed entry points, the enum values copied from rccl.h) runs on a one-GPU box
	const bool wantRccl = ndev > 1 || pirehip::GetConfig().force_rccl;
	if (wantRccl && distinct && r.ok) {
		m->comms.assign(ndev, nullptr);
		const int rc = r.CommInitAll(m->comms.data(), ndev, m->devices.data());
		if (rc == kNcclSuccess) {
			m->backend = "rccl";
			int version = 0;
			if (r.GetVersion && r.GetVersion(&version) == kNcclSuccess)
				m->backend += " " + std::to_string(version);
			// what the communicator says it is (N > 1 has never run here -- whoever runs it first reads this): the
			// size every rank's handle reports and the device each one sits on
			if (r.CommCount && r.CommCuDevice && r.CommUserRank) {
				std::string ranks;
				int size0 = -1;
				bool consistent = true;
				for (int g = 0; g < ndev; ++g) {
					int size = -1, dev = -1, rank = -1;
					(void)r.CommCount(m->comms[g], &size);
					(void)r.CommCuDevice(m->comms[g], &dev);
					(void)r.CommUserRank(m->comms[g], &rank);
					if (g == 0)
						size0 = size;
					consistent = consistent && size == size0 && dev == m->devices[g] && rank == g;
					ranks += (g ? "," : "") + std::to_string(rank) + "@dev" + std::to_string(dev);
				}
				m->backend += ", communicator of " + std::to_string(size0) + " ranks [" + ranks + "]" + (consistent && size0 == ndev ? "" : " INCONSISTENT");
			}
		} else {
			m->comms.clear();
			m->backend = std::string("host (ncclCommInitAll: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error") + ")";
		}
	} else if (ndev > 1 && !r.ok) {
		m->backend = "host (librccl not available)";
	} else if (ndev > 1) {
		m->backend = "host (a device is listed twice)";
	}
	*out = m.release();
	return PIRE_HIP_OK;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

void pire_hip_multi_destroy(pire_hip_multi* m)
{
	if (!m)
		return;
	DeviceGuard guard;
	DestroyMulti(m);
}

int pire_hip_multi_device_count(const pire_hip_multi* m) { return m ? int(m->devices.size()) : 0; }

const char* pire_hip_multi_reduce_backend(const pire_hip_multi* m) { return m ? m->backend.c_str() : ""; }

int pire_hip_multi_run_strided(pire_hip_multi* m, pire_hip_table* t, const pire_hip_shard* shards, uint32_t flags,
                               uint64_t* out_counts)
try {
	if (!m || !t || !shards) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	DeviceGuard guard;
	flags = (flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_GENERIC)) | PIRE_HIP_RUN_ON_DEVICE;
	return RunOnAllDevices(m, t, out_counts, [&](size_t g, uint64_t* counters) {
		const pire_hip_shard& s = shards[g];
		if (s.n == 0)
			return int(PIRE_HIP_OK);
		return pire_hip_run_strided(t, s.text, s.n, s.len, s.stride, flags, s.init_state_idx, s.out_state_idx, s.out_final,
		                            counters, m->streams[g]);
	});
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_multi_run(pire_hip_multi* m, pire_hip_table* t, const pire_hip_shard_offsets* shards, uint32_t flags,
                       uint64_t* out_counts)
try {
	if (!m || !t || !shards) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	DeviceGuard guard;
	// no look at the offsets (pire_hip_run's peek reads two words back and synchronises the shard's stream): every device
	// has to start its shard before any is waited for.  A few long resident documents per device: pire_hip_run
	// with PIRE_HIP_RUN_HOST_OFFSETS on that device, or the host-pointer form below, which knows the lengths.
	flags = (flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_GENERIC)) | PIRE_HIP_RUN_ON_DEVICE | PIRE_HIP_RUN_NO_PEEK;
	return RunOnAllDevices(m, t, out_counts, [&](size_t g, uint64_t* counters) {
		const pire_hip_shard_offsets& s = shards[g];
		if (s.n == 0)
			return int(PIRE_HIP_OK);
		return pire_hip_run(t, s.text, s.offsets, s.n, flags, s.init_state_idx, s.out_state_idx, s.out_final, counters,
		                    m->streams[g]);
	});
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_multi_last_split(const pire_hip_multi* m, uint64_t* out, int capacity)
{
	if (!m || !out || capacity < int(m->lastSplit.size())) {
		SetError("pire_hip_multi_last_split: null argument or too small an array (devices + 1 entries)");
		return PIRE_HIP_EINVAL;
	}
	for (size_t i = 0; i < m->lastSplit.size(); ++i)
		out[i] = m->lastSplit[i];
	return int(m->lastSplit.size());
}

namespace {

// The host-pointer entry points: shard g = strings [lo[g], lo[g+1]), staged into device slot g's pooled buffers,
// scanned through the device-pointer entry point, results copied back in string order.  offsets == nullptr: records.
int RunHostSharded(pire_hip_multi* m, pire_hip_table* t, const uint8_t* base, const uint64_t* offsets, uint64_t n,
                   uint64_t len, uint64_t stride, const std::vector<uint64_t>& lo, uint32_t flags,
                   const uint32_t* init_state_idx, uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts)
{
	const size_t G = m->devices.size();
	m->lastSplit = lo;
	std::vector<pire_hip_shard> rec(G);
	std::vector<pire_hip_shard_offsets> rag(G);
	int rc = PIRE_HIP_OK;
	for (size_t g = 0; g < G && rc == PIRE_HIP_OK; ++g) {
		const uint64_t cnt = lo[g + 1] - lo[g];
		memset(&rec[g], 0, sizeof(rec[g]));
		memset(&rag[g], 0, sizeof(rag[g]));
		rec[g].len = len;
		rec[g].stride = stride;
		if (cnt == 0)
			continue;
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e != hipSuccess) {
			rc = HipFail(e, "hipSetDevice");
			break;
		}
		// the bytes of the shard; an offset shard keeps the caller's offsets and every string's alignment: the staged
		// copy starts at a 256-byte boundary of the caller's buffer and is addressed through a virtual base
		const uint64_t first = offsets ? offsets[lo[g]] & ~uint64_t(255) : lo[g] * stride;
		const uint64_t last = offsets ? offsets[lo[g + 1]] : (lo[g + 1] - 1) * stride + len;
		const size_t bytes = size_t(last - first);
		void *dText = nullptr, *dOffs = nullptr, *dIdx = nullptr, *dFin = nullptr, *dInit = nullptr;
		if ((rc = ReserveStage(m, g, kStageText, bytes + 2 * kStageSlack, &dText)) ||
		    (offsets && (rc = ReserveStage(m, g, kStageOffs, (cnt + 1) * 8, &dOffs))) ||
		    (out_state_idx && (rc = ReserveStage(m, g, kStageIdx, cnt * 4, &dIdx))) ||
		    (out_final && (rc = ReserveStage(m, g, kStageFin, cnt, &dFin))) ||
		    (init_state_idx && (rc = ReserveStage(m, g, kStageInit, cnt * 4, &dInit))))
			break;
		uint8_t* staged = static_cast<uint8_t*>(dText) + kStageSlack;
		if (bytes)
			e = hipMemcpyAsync(staged, base + first, bytes, hipMemcpyHostToDevice, m->streams[g]);
		if (e == hipSuccess && offsets)
			e = hipMemcpyAsync(dOffs, offsets + lo[g], (cnt + 1) * 8, hipMemcpyHostToDevice, m->streams[g]);
		if (e == hipSuccess && dInit)
			e = hipMemcpyAsync(dInit, init_state_idx + lo[g], cnt * 4, hipMemcpyHostToDevice, m->streams[g]);
		if (e != hipSuccess) {
			rc = HipFail(e, "hipMemcpy(H2D)");
			break;
		}
		rec[g].text = staged;
		rec[g].n = cnt;
		rec[g].init_state_idx = static_cast<const uint32_t*>(dInit);
		rec[g].out_state_idx = static_cast<uint32_t*>(dIdx);
		rec[g].out_final = static_cast<uint8_t*>(dFin);
		rag[g].text = staged - first;   // so that the caller's offsets address the staged copy
		rag[g].offsets = static_cast<const uint64_t*>(dOffs);
		rag[g].n = cnt;
		rag[g].init_state_idx = rec[g].init_state_idx;
		rag[g].out_state_idx = rec[g].out_state_idx;
		rag[g].out_final = rec[g].out_final;
	}
	if (rc == PIRE_HIP_OK && offsets) {
		// the host knows the lengths: a shard of few long strings takes the segmented scan (pire_hip_run with the host's
		// offsets; that call synchronises its stream), every other shard is enqueued without a look at its offsets
		const uint32_t f = (flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_GENERIC)) | PIRE_HIP_RUN_ON_DEVICE;
		auto segmented = [&](size_t g) {
			const pire_hip_shard_offsets& s = rag[g];
			return s.n != 0 && !(f & PIRE_HIP_RUN_GENERIC) && !s.init_state_idx &&
			       pirehip::SegmentedEligible(s.n, offsets[lo[g + 1]] - offsets[lo[g]]);
		};
		rc = RunOnAllDevices(m, t, out_counts, [&](size_t g, uint64_t* counters) {
			const pire_hip_shard_offsets& s = rag[g];
			if (s.n == 0)
				return int(PIRE_HIP_OK);
			if (segmented(g))
				return pire_hip_run(t, s.text, offsets + lo[g], s.n, f | PIRE_HIP_RUN_HOST_OFFSETS, nullptr, s.out_state_idx,
				                    s.out_final, counters, m->streams[g]);
			return pire_hip_run(t, s.text, s.offsets, s.n, f | PIRE_HIP_RUN_NO_PEEK, s.init_state_idx, s.out_state_idx, s.out_final,
			                    counters, m->streams[g]);
		}, segmented);
	} else if (rc == PIRE_HIP_OK) {
		rc = pire_hip_multi_run_strided(m, t, rec.data(), flags, out_counts);
	}
	for (size_t g = 0; g < G && rc == PIRE_HIP_OK; ++g) {
		const uint64_t cnt = lo[g + 1] - lo[g];
		if (cnt == 0)
			continue;
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e == hipSuccess && out_state_idx)
			e = hipMemcpyAsync(out_state_idx + lo[g], rec[g].out_state_idx, cnt * 4, hipMemcpyDeviceToHost, m->streams[g]);
		if (e == hipSuccess && out_final)
			e = hipMemcpyAsync(out_final + lo[g], rec[g].out_final, cnt, hipMemcpyDeviceToHost, m->streams[g]);
		if (e != hipSuccess)
			rc = HipFail(e, "hipMemcpy(D2H)");
	}
	// every stream drained before the call returns, error or not: the pooled buffers are free for the next call
	const std::string msg = rc != PIRE_HIP_OK ? pire_hip_last_error() : "";
	for (size_t g = 0; g < G; ++g) {
		(void)hipSetDevice(m->devices[g]);
		hipError_t e = hipStreamSynchronize(m->streams[g]);
		if (e != hipSuccess && rc == PIRE_HIP_OK)
			rc = HipFail(e, "hipStreamSynchronize");
	}
	if (!msg.empty())
		SetError(msg);
	return rc;
}

}  // namespace

int pire_hip_multi_run_strided_host(pire_hip_multi* m, pire_hip_table* t, const void* text, uint64_t n, uint64_t len,
                                    uint64_t stride, uint32_t flags, const uint32_t* init_state_idx,
                                    uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts)
try {
	if (!m || !t || (n && len && !text)) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	if (stride < len) {
		SetError("stride < len");
		return PIRE_HIP_EINVAL;
	}
	DeviceGuard guard;
	const size_t G = m->devices.size();
	// shard g owns strings [lo, hi): contiguous, balanced (records are equal work: sizes differ by at most one)
	std::vector<uint64_t> lo(G + 1);
	for (size_t g = 0; g <= G; ++g)
		lo[g] = n / G * g + std::min<uint64_t>(g, n % G);
	return RunHostSharded(m, t, static_cast<const uint8_t*>(text), nullptr, n, len, stride, lo, flags, init_state_idx,
	                      out_state_idx, out_final, out_counts);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_multi_run_host(pire_hip_multi* m, pire_hip_table* t, const void* text, const uint64_t* offsets, uint64_t n,
                            uint32_t flags, const uint32_t* init_state_idx, uint32_t* out_state_idx, uint8_t* out_final,
                            uint64_t* out_counts)
try {
	if (!m || !t || (n && !offsets)) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	for (uint64_t i = 0; i < n; ++i)
		if (offsets[i + 1] < offsets[i]) {
			SetError("offsets must be non-decreasing");
			return PIRE_HIP_EINVAL;
		}
	if (n && offsets[n] > offsets[0] && !text) {
		SetError("null text pointer with non-empty strings");
		return PIRE_HIP_EINVAL;
	}
	DeviceGuard guard;
	const size_t G = m->devices.size();
	// balanced by BYTES: the boundary of shard g is the string boundary nearest to g / G of the text (whole strings
	// only; empty strings at a boundary go with the shard in front).  Equal byte counts are equal work whatever the
	// lengths; when all the text is empty strings the split falls back to string counts.
	std::vector<uint64_t> lo(G + 1, 0);
	lo[G] = n;
	const uint64_t b0 = n ? offsets[0] : 0, total = n ? offsets[n] - b0 : 0;
	for (size_t g = 1; g < G; ++g) {
		if (total == 0) {
			lo[g] = n / G * g + std::min<uint64_t>(g, n % G);
			continue;
		}
		const uint64_t want = b0 + total / G * g + std::min<uint64_t>(g, total % G);
		const uint64_t* it = std::lower_bound(offsets, offsets + n + 1, want);   // first boundary at or behind `want`
		uint64_t k = uint64_t(it - offsets);
		if (k > 0 && want - offsets[k - 1] < offsets[k] - want)
			--k;                                                                    // the one in front is nearer
		lo[g] = std::max(lo[g - 1], std::min<uint64_t>(k, n));
	}
	return RunHostSharded(m, t, static_cast<const uint8_t*>(text), offsets, n, 0, 0, lo, flags, init_state_idx, out_state_idx,
	                      out_final, out_counts);
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

}  // extern "C"
