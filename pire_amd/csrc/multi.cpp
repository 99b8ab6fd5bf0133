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

#include <cstring>
#include <memory>
#include <new>
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
		x.ok = x.CommInitAll && x.CommDestroy && x.AllReduce && x.GroupStart && x.GroupEnd;
		return x;
	}();
	return r;
}

constexpr uint32_t kMaxCounters = kMaxLdsCountRegexps + 2;

}  // namespace

struct pire_hip_multi {
	std::vector<int> devices;
	std::vector<hipStream_t> streams;
	std::vector<unsigned long long*> counters;   // [kMaxCounters] u64 on each device
	std::vector<ncclComm_t> comms;               // empty: host reduce
	std::vector<unsigned long long> hostCounters;
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
	for (int g = 0; g < ndev; ++g) {
		if ((e = hipSetDevice(m->devices[g])) != hipSuccess ||
		    (e = hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking)) != hipSuccess ||
		    (e = hipMalloc(reinterpret_cast<void**>(&m->counters[g]), kMaxCounters * 8)) != hipSuccess)
			return HipFail(e, "setting up a device of the multi-GPU runner");
	}
	m->backend = "host";
	const Rccl& r = LoadRccl();
	if (ndev > 1 && distinct && r.ok) {
		m->comms.assign(ndev, nullptr);
		const int rc = r.CommInitAll(m->comms.data(), ndev, m->devices.data());
		if (rc == kNcclSuccess) {
			m->backend = "rccl";
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
	pire_hip_table_info info;
	if (int rc = pire_hip_table_get_info(t, &info))
		return rc;
	const uint32_t count = info.regexps + 2;
	if (out_counts && count > kMaxCounters) {
		SetError("out_counts is supported for scanners with at most 1024 regexps");
		return PIRE_HIP_EUNSUPPORTED;
	}
	DeviceGuard guard;
	const size_t G = m->devices.size();
	flags = (flags & (PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END | PIRE_HIP_RUN_GENERIC)) | PIRE_HIP_RUN_ON_DEVICE;
	// every device starts its shard before any is waited for: the launches are asynchronous
	for (size_t g = 0; g < G; ++g) {
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e == hipSuccess && out_counts)
			e = hipMemsetAsync(m->counters[g], 0, size_t(count) * 8, m->streams[g]);
		if (e != hipSuccess)
			return HipFail(e, "hipSetDevice / counter reset");
		const pire_hip_shard& s = shards[g];
		if (s.n == 0)
			continue;
		if (int rc = pire_hip_run_strided(t, s.text, s.n, s.len, s.stride, flags, s.init_state_idx, s.out_state_idx, s.out_final,
		                                  out_counts ? reinterpret_cast<uint64_t*>(m->counters[g]) : nullptr, m->streams[g]))
			return rc;
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
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

int pire_hip_multi_run_strided_host(pire_hip_multi* m, pire_hip_table* t, const void* text, uint64_t n, uint64_t len,
                                    uint64_t stride, uint32_t flags, const uint32_t* init_state_idx,
                                    uint32_t* out_state_idx, uint8_t* out_final, uint64_t* out_counts)
try {
	if (!m || !t || (n && !text)) {
		SetError("null argument");
		return PIRE_HIP_EINVAL;
	}
	if (stride < len) {
		SetError("stride < len");
		return PIRE_HIP_EINVAL;
	}
	DeviceGuard guard;
	const size_t G = m->devices.size();
	std::vector<pire_hip_shard> shards(G);
	std::vector<void*> owned;   // (device, pointer) pairs flattened: device buffers of this call
	std::vector<int> ownedDev;
	auto release = [&] {
		for (size_t i = 0; i < owned.size(); ++i) {
			(void)hipSetDevice(ownedDev[i]);
			(void)hipFree(owned[i]);
		}
	};
	auto alloc = [&](int dev, size_t bytes, void** p) -> int {
		hipError_t e = hipMalloc(p, bytes ? bytes : 16);
		if (e != hipSuccess)
			return HipFail(e, "hipMalloc(shard)");
		owned.push_back(*p);
		ownedDev.push_back(dev);
		return PIRE_HIP_OK;
	};
	const uint8_t* base = static_cast<const uint8_t*>(text);
	int rc = PIRE_HIP_OK;
	// shard g owns strings [lo, hi): contiguous, balanced (sizes differ by at most one)
	std::vector<uint64_t> lo(G + 1);
	for (size_t g = 0; g <= G; ++g)
		lo[g] = n / G * g + std::min<uint64_t>(g, n % G);
	for (size_t g = 0; g < G && rc == PIRE_HIP_OK; ++g) {
		const uint64_t cnt = lo[g + 1] - lo[g];
		const int dev = m->devices[g];
		hipError_t e = hipSetDevice(dev);
		if (e != hipSuccess) {
			rc = HipFail(e, "hipSetDevice");
			break;
		}
		pire_hip_shard& s = shards[g];
		memset(&s, 0, sizeof(s));
		s.n = cnt;
		s.len = len;
		s.stride = stride;
		if (cnt == 0)
			continue;
		const size_t bytes = size_t(cnt - 1) * stride + len;
		void *dText = nullptr, *dIdx = nullptr, *dFin = nullptr, *dInit = nullptr;
		if ((rc = alloc(dev, bytes, &dText)) || (out_state_idx && (rc = alloc(dev, cnt * 4, &dIdx))) ||
		    (out_final && (rc = alloc(dev, cnt, &dFin))) || (init_state_idx && (rc = alloc(dev, cnt * 4, &dInit))))
			break;
		e = hipMemcpyAsync(dText, base + lo[g] * stride, bytes, hipMemcpyHostToDevice, m->streams[g]);
		if (e == hipSuccess && dInit)
			e = hipMemcpyAsync(dInit, init_state_idx + lo[g], cnt * 4, hipMemcpyHostToDevice, m->streams[g]);
		if (e != hipSuccess) {
			rc = HipFail(e, "hipMemcpy(H2D)");
			break;
		}
		s.text = dText;
		s.init_state_idx = static_cast<const uint32_t*>(dInit);
		s.out_state_idx = static_cast<uint32_t*>(dIdx);
		s.out_final = static_cast<uint8_t*>(dFin);
	}
	if (rc == PIRE_HIP_OK)
		rc = pire_hip_multi_run_strided(m, t, shards.data(), flags, out_counts);
	for (size_t g = 0; g < G && rc == PIRE_HIP_OK; ++g) {
		const uint64_t cnt = lo[g + 1] - lo[g];
		if (cnt == 0)
			continue;
		hipError_t e = hipSetDevice(m->devices[g]);
		if (e == hipSuccess && out_state_idx)
			e = hipMemcpyAsync(out_state_idx + lo[g], shards[g].out_state_idx, cnt * 4, hipMemcpyDeviceToHost, m->streams[g]);
		if (e == hipSuccess && out_final)
			e = hipMemcpyAsync(out_final + lo[g], shards[g].out_final, cnt, hipMemcpyDeviceToHost, m->streams[g]);
		if (e != hipSuccess)
			rc = HipFail(e, "hipMemcpy(D2H)");
	}
	for (size_t g = 0; g < G; ++g) {
		(void)hipSetDevice(m->devices[g]);
		hipError_t e = hipStreamSynchronize(m->streams[g]);
		if (e != hipSuccess && rc == PIRE_HIP_OK)
			rc = HipFail(e, "hipStreamSynchronize");
	}
	release();
	return rc;
} catch (...) {
	return pirehip::HandleException();   // an exception must not unwind through the C ABI
}

}  // extern "C"
