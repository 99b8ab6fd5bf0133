/*
 * batch_runner.hpp -- header-only C++ shim that keeps the Pire::Scanner / Pire::Runner vocabulary on top of the
 * C ABI of libpire_hip.so (include/pire_hip.h).  This is the binding a Pire maintainer would add:
 * the reference has no FFI on this path, it is a compile-time template concept (pire/run.h:50-57, 271-275,
 * 365-392), so the drop-in is a class with the same fluent surface, batched over N strings:
 *
 *     Pire::Scanner sc = ...;                                  // built by the reference, on the host, unchanged
 *     Pire::Hip::BatchRunner<Pire::Scanner> run(sc);           // Save() -> pire_hip_table_create()
 *     run.Begin().Run(text, offsets, n).End();                 // == for each i: Runner(sc).Begin().Run(..).End()
 *     const auto& states = run.States();                       // std::vector<Pire::Scanner::State>
 *     bool ok = sc.Final(states[i]);                           // every Scanner accessor keeps working
 *     auto ids = sc.AcceptedRegexps(states[i]);
 *
 * It compiles against the unmodified reference headers (<pire/pire.h>); nothing in it re-implements the walk: the
 * states come back from the GPU as StateIndex values and are turned into Scanner::State (a row address inside the
 * host scanner) with public API only -- Initialize(), StateIndex(), LettersCount(), sizeof(ScannerRowHeader),
 * Transition (pire/scanners/multi.h:161, 281-284, 140, 119, 99, 347).
 *
 * Errors: the C ABI's negative codes are re-thrown as Pire::Error (pire/stub/stl.h:213-217), the reference's own
 * exception type for this library.
 */
#ifndef PIRE_HIP_BATCH_RUNNER_HPP
#define PIRE_HIP_BATCH_RUNNER_HPP

#include <cstdint>
#include <sstream>
#include <string>
#include <vector>

#include <pire/pire.h>

#include "../pire_hip.h"

namespace Pire {
namespace Hip {

inline void Check(int rc)
{
	if (rc < 0)
		throw Pire::Error(std::string("pire_hip: ") + pire_hip_last_error());
}

/* Bytes between the State values of consecutive StateIndex numbers, from public API only. */
template <class Scanner>
struct RowGeometry {
	static size_t Stride(const Scanner& sc)
	{
		typedef typename Scanner::Transition Tr;
		const size_t header = sizeof(typename Scanner::ScannerRowHeader) / sizeof(Tr);           // HEADER_SIZE, multi.h:349
		const size_t align = sizeof(Pire::Impl::MaxSizeWord) / sizeof(Tr);
		const size_t row = (sc.LettersCount() + header + align - 1) / align * align;             // RowSize(), multi.h:347
		return row * sizeof(Tr);
	}
};
template <>
struct RowGeometry<Pire::SimpleScanner> {
	static size_t Stride(const Pire::SimpleScanner&)
	{
		return (Pire::MaxChar + 1) * sizeof(Pire::SimpleScanner::Transition);                    // STATE_ROW_SIZE, simple.h:43
	}
};

/* A device-side copy of a compiled scanner.  Immutable, shareable between threads, like the scanner itself. */
template <class Scanner>
class Table {
public:
	explicit Table(const Scanner& sc)
	    : m_table(nullptr)
	{
		// the PUBLIC hand-off: Scanner::Save (multi.h:307, 557-573); NonrelocScanner saves as Relocatable (604-608);
		// SimpleScanner::Save (scanner_io.cpp:35-49)
		if (pire_hip_abi_version() != PIRE_HIP_ABI_VERSION)   // structs of this header vs. the loaded library's: refuse, do not guess
			throw Pire::Error("pire_hip: libpire_hip.so was built with another ABI version than this header");
		std::ostringstream out;
		sc.Save(&out);
		const std::string blob = out.str();
		Check(pire_hip_table_create(blob.data(), blob.size(), &m_table));

		// State <-> StateIndex geometry of THIS host scanner, public API only
		m_stride = RowGeometry<Scanner>::Stride(sc);
		typename Scanner::State init;
		sc.Initialize(init);
		m_base = init - sc.StateIndex(init) * m_stride;
	}
	~Table() { pire_hip_table_destroy(m_table); }

	pire_hip_table* Handle() const { return m_table; }

	/* Which states have rows in LDS is a performance choice the table makes from a byte model of text and then corrects
	 * from what the scans really visit -- by itself (pire_hip_config.auto_adapt = 0, the default: a worker thread re-ranks a
	 * copy of the table and a later launch swaps it in; no call is made to wait), or here, explicitly, after a representative
	 * batch.  Results never depend on it.  Returns the number of rows that entered the dense set.  May run while other
	 * threads scan with this table (it joins a running worker first). */
	unsigned Adapt()
	{
		uint32_t changed = 0;
		Check(pire_hip_table_adapt(m_table, &changed));
		return changed;
	}
	/* The automatic form off (true) or on (false) for the whole library; see pire_hip_config. */
	static void FreezeRanking(bool frozen)
	{
		pire_hip_config c;
		c.size = sizeof(c);
		Check(pire_hip_config_get(&c));
		c.auto_adapt = frozen ? 1 : 0;
		Check(pire_hip_config_set(&c));
	}
	typename Scanner::State ToState(uint32_t idx) const { return m_base + size_t(idx) * m_stride; }
	uint32_t ToIndex(typename Scanner::State st) const { return uint32_t((st - m_base) / m_stride); }

private:
	Table(const Table&);
	Table& operator=(const Table&);
	pire_hip_table* m_table;
	size_t m_base, m_stride;
};

/* A device buffer owned by the shim (grown on demand, freed with its owner). */
class DeviceBuffer {
public:
	DeviceBuffer() : m_ptr(nullptr), m_bytes(0) {}
	~DeviceBuffer() { pire_hip_device_free(m_ptr); }
	void* Reserve(size_t bytes)
	{
		if (bytes > m_bytes) {
			pire_hip_device_free(m_ptr);
			m_ptr = nullptr;
			m_bytes = 0;
			Check(pire_hip_device_alloc(bytes, &m_ptr));
			m_bytes = bytes;
		}
		return m_ptr;
	}
	void* Get() const { return m_ptr; }

private:
	DeviceBuffer(const DeviceBuffer&);
	DeviceBuffer& operator=(const DeviceBuffer&);
	void* m_ptr;
	size_t m_bytes;
};

/*
 * Batched twin of Pire::RunHelper (run.h:365-386).
 *
 * Two ways to hand the text over:
 *   Run(text, offsets, n)                     host pointers: the library moves the text over PCIe chunk by chunk,
 *                                             overlapped with the scan (PCIe bound, ~50 GB/s);
 *   RunDevice(text, offsets, n, stream)       DEVICE pointers: the text is already resident in HBM (a corpus that was
 *   RunDeviceStrided(text, n, len, stride, ..) loaded once, the output of another kernel); the scan runs at the speed
 *                                             of the C ABI (TB/s).  Results stay on the device until asked for:
 *                                             MatchCounts() moves 8 * (regexps + 2) bytes, States()/Finals() 5 bytes
 *                                             per string; DeviceStateIndices()/DeviceFinals() move nothing.
 */
template <class Scanner>
class BatchRunner {
public:
	typedef typename Scanner::State State;

	explicit BatchRunner(const Scanner& sc)
	    : m_own(new Table<Scanner>(sc)), m_table(m_own) { Reset(); }
	/* Re-use one device table for many batches. */
	explicit BatchRunner(const Table<Scanner>& table)
	    : m_own(nullptr), m_table(&table) { Reset(); }
	~BatchRunner() { delete m_own; }

	/* RunHelper(sc, st): resume every string from a previously returned state (run.h:368, 391-392). */
	BatchRunner& From(const std::vector<State>& states)
	{
		m_init.resize(states.size());
		for (size_t i = 0; i < states.size(); ++i)
			m_init[i] = m_table->ToIndex(states[i]);
		return *this;
	}

	BatchRunner& Begin() { m_flags |= PIRE_HIP_RUN_BEGIN; return *this; }     // run.h:375
	BatchRunner& End() { m_flags |= PIRE_HIP_RUN_END; return *this; }         // run.h:376

	/* Run(begin,end) for n strings: string i = text[offsets[i], offsets[i+1]).  Host pointers. */
	BatchRunner& Run(const char* text, const uint64_t* offsets, size_t n)
	{
		m_text = text;
		m_offsets = offsets;
		m_n = n;
		m_onDevice = false;
		m_ran = false;
		return *this;
	}

	/* The same with DEVICE pointers (text and offsets resident in HBM); `stream` is a hipStream_t or null. */
	BatchRunner& RunDevice(const void* deviceText, const uint64_t* deviceOffsets, size_t n, void* stream = nullptr)
	{
		m_text = static_cast<const char*>(deviceText);
		m_offsets = deviceOffsets;
		m_n = n;
		m_len = m_stride = 0;
		m_stream = stream;
		m_onDevice = true;
		m_ran = false;
		return *this;
	}

	/* Device-resident fixed-length records: string i = text[i * stride, i * stride + len). */
	BatchRunner& RunDeviceStrided(const void* deviceText, size_t n, size_t len, size_t stride, void* stream = nullptr)
	{
		m_text = static_cast<const char*>(deviceText);
		m_offsets = nullptr;
		m_n = n;
		m_len = len;
		m_stride = stride;
		m_stream = stream;
		m_onDevice = true;
		m_ran = false;
		return *this;
	}

	/* Convenience: a vector of strings (copied into one buffer). */
	BatchRunner& Run(const std::vector<ystring>& strings)
	{
		m_ownText.clear();
		m_ownOffsets.assign(1, 0);
		for (size_t i = 0; i < strings.size(); ++i) {
			m_ownText.append(strings[i]);
			m_ownOffsets.push_back(m_ownText.size());
		}
		return Run(m_ownText.data(), m_ownOffsets.data(), strings.size());
	}

	/* RunHelper::State() per string (run.h:378). */
	const std::vector<State>& States()
	{
		Fetch();
		return m_states;
	}

	/* operator bool of RunHelper per string (run.h:380): Final(State()). */
	const std::vector<char>& Finals()
	{
		Fetch();
		return m_final;
	}

	/* [0] strings ending in a final state, [1] strings scanned, [2+r] strings accepting regexp r. */
	const std::vector<uint64_t>& MatchCounts()
	{
		Execute();
		if (m_onDevice && !m_countsFetched) {
			Check(pire_hip_copy_to_host(m_counts.data(), m_devCounts.Get(), m_counts.size() * 8, m_stream));
			Check(pire_hip_stream_synchronize(m_stream));
			m_countsFetched = true;
		}
		return m_counts;
	}

	/* After RunDevice*: the results where the scan left them -- uint32 StateIndex and uint8 Final per string, device
	 * memory owned by this runner, valid until its next Run*; ordered after the scan on the stream given to RunDevice. */
	const uint32_t* DeviceStateIndices() { Execute(); return static_cast<const uint32_t*>(m_devIdx.Get()); }
	const uint8_t* DeviceFinals() { Execute(); return static_cast<const uint8_t*>(m_devFin.Get()); }

	const Table<Scanner>& GetTable() const { return *m_table; }
	uint32_t Flags() const { return m_flags; }

private:
	void Reset()
	{
		m_flags = 0;
		m_text = nullptr;
		m_offsets = nullptr;
		m_n = m_len = m_stride = 0;
		m_stream = nullptr;
		m_onDevice = m_ran = m_fetched = m_countsFetched = false;
	}

	void Execute()
	{
		if (m_ran)
			return;
		if (!m_init.empty() && m_init.size() != m_n)
			throw Pire::Error("pire_hip: From() and Run() disagree on the number of strings");
		pire_hip_table_info info;
		Check(pire_hip_table_get_info(m_table->Handle(), &info));
		m_counts.assign(size_t(info.regexps) + 2, 0);
		static const uint64_t kNoOffsets[1] = {0};
		if (m_onDevice) {
			uint32_t* idx = static_cast<uint32_t*>(m_devIdx.Reserve(m_n * 4));
			uint8_t* fin = static_cast<uint8_t*>(m_devFin.Reserve(m_n));
			uint64_t* cnt = static_cast<uint64_t*>(m_devCounts.Reserve(m_counts.size() * 8));
			Check(pire_hip_memset_device(cnt, 0, m_counts.size() * 8, m_stream));
			const uint32_t* init = nullptr;
			if (!m_init.empty()) {
				init = static_cast<const uint32_t*>(m_devInit.Reserve(m_n * 4));
				Check(pire_hip_copy_to_device(m_devInit.Get(), m_init.data(), m_n * 4, m_stream));
			}
			const uint32_t flags = m_flags | PIRE_HIP_RUN_ON_DEVICE;
			if (m_offsets || !m_n)
				Check(pire_hip_run(m_table->Handle(), m_text, m_offsets, m_n, flags, init, idx, fin, cnt, m_stream));
			else
				Check(pire_hip_run_strided(m_table->Handle(), m_text, m_n, m_len, m_stride, flags, init, idx, fin, cnt,
				                           m_stream));
			if (!m_init.empty())
				Check(pire_hip_stream_synchronize(m_stream));   // m_init was the source of an asynchronous copy
			m_fetched = m_countsFetched = false;
		} else {
			m_idx.resize(m_n);
			m_fin.resize(m_n);
			Check(pire_hip_run(m_table->Handle(), m_text, m_n ? m_offsets : kNoOffsets, m_n, m_flags,
			                   m_init.empty() ? nullptr : m_init.data(), m_idx.data(), m_fin.data(), m_counts.data(), nullptr));
			m_fetched = false;
			m_countsFetched = true;
		}
		m_ran = true;
	}

	/* Per-string results on the host, as Scanner::State values. */
	void Fetch()
	{
		Execute();
		if (m_fetched)
			return;
		if (m_onDevice) {
			m_idx.resize(m_n);
			m_fin.resize(m_n);
			Check(pire_hip_copy_to_host(m_idx.data(), m_devIdx.Get(), m_n * 4, m_stream));
			Check(pire_hip_copy_to_host(m_fin.data(), m_devFin.Get(), m_n, m_stream));
			Check(pire_hip_stream_synchronize(m_stream));
		}
		m_states.resize(m_n);
		m_final.resize(m_n);
		for (size_t i = 0; i < m_n; ++i) {
			m_states[i] = m_table->ToState(m_idx[i]);
			m_final[i] = char(m_fin[i]);
		}
		m_fetched = true;
	}

	BatchRunner(const BatchRunner&);
	BatchRunner& operator=(const BatchRunner&);

	Table<Scanner>* m_own;
	const Table<Scanner>* m_table;
	uint32_t m_flags;
	const char* m_text;
	const uint64_t* m_offsets;
	size_t m_n, m_len, m_stride;
	void* m_stream;
	bool m_onDevice, m_ran, m_fetched, m_countsFetched;
	std::vector<uint32_t> m_init, m_idx;
	std::vector<uint8_t> m_fin;
	std::vector<State> m_states;
	std::vector<char> m_final;
	std::vector<uint64_t> m_counts;
	DeviceBuffer m_devIdx, m_devFin, m_devCounts, m_devInit;
	ystring m_ownText;
	std::vector<uint64_t> m_ownOffsets;
};

template <class Scanner>
BatchRunner<Scanner>* NewBatchRunner(const Scanner& sc) { return new BatchRunner<Scanner>(sc); }

/*
 * Batched Pire::LongestPrefix / Pire::ShortestPrefix (run.h:277-311).  Returns, per string, the END pointer of the
 * prefix exactly as the reference does (null = no prefix), computed from the lengths the GPU returns.
 */
template <class Scanner>
std::vector<const char*> BatchPrefix(const Table<Scanner>& table, bool longest, const char* text, const uint64_t* offsets,
                                     size_t n, bool throughBeginMark = false, bool throughEndMark = false)
{
	std::vector<int64_t> len(n);
	static const uint64_t kNoOffsets[1] = {0};
	Check(pire_hip_prefix(table.Handle(), text, n ? offsets : kNoOffsets, n, longest ? 1 : 0, throughBeginMark ? 1 : 0,
	                      throughEndMark ? 1 : 0, 0, len.data(), nullptr));
	std::vector<const char*> out(n);
	for (size_t i = 0; i < n; ++i)
		out[i] = len[i] < 0 ? nullptr : text + offsets[i] + len[i];
	return out;
}

template <class Scanner>
std::vector<const char*> BatchLongestPrefix(const Table<Scanner>& t, const char* text, const uint64_t* offsets, size_t n,
                                            bool throughBeginMark = false, bool throughEndMark = false)
{
	return BatchPrefix(t, true, text, offsets, n, throughBeginMark, throughEndMark);
}

template <class Scanner>
std::vector<const char*> BatchShortestPrefix(const Table<Scanner>& t, const char* text, const uint64_t* offsets, size_t n,
                                             bool throughBeginMark = false, bool throughEndMark = false)
{
	return BatchPrefix(t, false, text, offsets, n, throughBeginMark, throughEndMark);
}

/*
 * Batched Pire::LongestSuffix / Pire::ShortestSuffix (run.h:313-362): every string is walked backwards from its last
 * byte.  Returns, per string, the pointer the reference returns -- one before the suffix's first byte, i.e.
 * (last byte) - length -- or null.
 */
template <class Scanner>
std::vector<const char*> BatchSuffix(const Table<Scanner>& table, bool longest, const char* text, const uint64_t* offsets,
                                     size_t n, bool throughEndMark = false, bool throughBeginMark = false)
{
	std::vector<int64_t> len(n);
	static const uint64_t kNoOffsets[1] = {0};
	Check(pire_hip_suffix(table.Handle(), text, n ? offsets : kNoOffsets, n, longest ? 1 : 0, throughEndMark ? 1 : 0,
	                      throughBeginMark ? 1 : 0, 0, len.data(), nullptr));
	std::vector<const char*> out(n);
	for (size_t i = 0; i < n; ++i)
		out[i] = len[i] < 0 ? nullptr : text + offsets[i + 1] - 1 - len[i];
	return out;
}

template <class Scanner>
std::vector<const char*> BatchLongestSuffix(const Table<Scanner>& t, const char* text, const uint64_t* offsets, size_t n,
                                            bool throughEndMark = false, bool throughBeginMark = false)
{
	return BatchSuffix(t, true, text, offsets, n, throughEndMark, throughBeginMark);
}

template <class Scanner>
std::vector<const char*> BatchShortestSuffix(const Table<Scanner>& t, const char* text, const uint64_t* offsets, size_t n,
                                             bool throughEndMark = false, bool throughBeginMark = false)
{
	return BatchSuffix(t, false, text, offsets, n, throughEndMark, throughBeginMark);
}

/*
 * BatchRunner over every GPU of the node: the batch is cut into one run of whole strings per device,
 * balanced by bytes, staged through buffers the runner keeps between calls, scanned on all devices at once; the match
 * counters are summed with one all-reduce over RCCL (or on the host, Backend() says which).  Same fluent surface:
 *     MultiBatchRunner<Pire::Scanner> gpus(sc);          // all devices; or (sc, {0, 1, 2, 3})
 *     gpus.Begin().Run(lines).End().States()
 */
template <class Scanner>
class MultiBatchRunner {
public:
	typedef typename Scanner::State State;

	explicit MultiBatchRunner(const Scanner& sc, const std::vector<int>& devices = std::vector<int>())
	    : m_table(sc), m_multi(nullptr), m_flags(0), m_text(nullptr), m_offsets(nullptr), m_n(0), m_ran(false)
	{
		Check(pire_hip_multi_create(devices.empty() ? nullptr : devices.data(), int(devices.size()), &m_multi));
	}
	~MultiBatchRunner() { pire_hip_multi_destroy(m_multi); }

	MultiBatchRunner& Begin() { m_flags |= PIRE_HIP_RUN_BEGIN; return *this; }
	MultiBatchRunner& End() { m_flags |= PIRE_HIP_RUN_END; return *this; }
	MultiBatchRunner& Run(const char* text, const uint64_t* offsets, size_t n)
	{
		m_text = text;
		m_offsets = offsets;
		m_n = n;
		m_ran = false;
		return *this;
	}
	MultiBatchRunner& Run(const std::vector<ystring>& strings)
	{
		m_ownText.clear();
		m_ownOffsets.assign(1, 0);
		for (size_t i = 0; i < strings.size(); ++i) {
			m_ownText.append(strings[i]);
			m_ownOffsets.push_back(m_ownText.size());
		}
		return Run(m_ownText.data(), m_ownOffsets.data(), strings.size());
	}
	const std::vector<State>& States() { Execute(); return m_states; }
	const std::vector<char>& Finals() { Execute(); return m_final; }
	const std::vector<uint64_t>& MatchCounts() { Execute(); return m_counts; }
	int Devices() const { return pire_hip_multi_device_count(m_multi); }
	const char* Backend() const { return pire_hip_multi_reduce_backend(m_multi); }
	Table<Scanner>& GetTable() { return m_table; }

private:
	MultiBatchRunner(const MultiBatchRunner&);
	MultiBatchRunner& operator=(const MultiBatchRunner&);
	void Execute()
	{
		if (m_ran)
			return;
		pire_hip_table_info info;
		Check(pire_hip_table_get_info(m_table.Handle(), &info));
		std::vector<uint32_t> idx(m_n);
		std::vector<uint8_t> fin(m_n);
		m_counts.assign(info.regexps + 2, 0);
		static const uint64_t none[1] = {0};
		Check(pire_hip_multi_run_host(m_multi, m_table.Handle(), m_text, m_n ? m_offsets : none, m_n, m_flags, nullptr,
		                              idx.data(), fin.data(), m_counts.data()));
		m_states.resize(m_n);
		m_final.resize(m_n);
		for (size_t i = 0; i < m_n; ++i) {
			m_states[i] = m_table.ToState(idx[i]);
			m_final[i] = char(fin[i]);
		}
		m_ran = true;
	}

	Table<Scanner> m_table;
	pire_hip_multi* m_multi;
	uint32_t m_flags;
	const char* m_text;
	const uint64_t* m_offsets;
	size_t m_n;
	bool m_ran;
	std::string m_ownText;
	std::vector<uint64_t> m_ownOffsets;
	std::vector<State> m_states;
	std::vector<char> m_final;
	std::vector<uint64_t> m_counts;
};

/*
 * Batched twin of Pire::ScannerPair<Scanner1, Scanner2> (scanners/pair.h:33-94) and of
 * Pire::Run(scanner1, scanner2, state1, state2, begin, end) (run.h:229-241): both scanners over the same strings, State =
 * pair of the two states (pair.h:35), Final = either (pair.h:69-72).  Host pointers: two passes (the walks are
 * independent, pair.h:52-66).  Device-resident fixed-length records (RunDeviceStrided): ONE fused pass with both tables
 * in LDS -- the text is read once (pire_hip_run_pair_strided).
 */
template <class Scanner1, class Scanner2>
class PairBatchRunner {
public:
	typedef ypair<typename Scanner1::State, typename Scanner2::State> State;

	PairBatchRunner(const Scanner1& s1, const Scanner2& s2)
	    : m_first(s1), m_second(s2), m_fused(false), m_ran(false), m_text(nullptr), m_n(0), m_len(0), m_stride(0), m_stream(nullptr) {}

	PairBatchRunner& Begin() { m_first.Begin(); m_second.Begin(); return *this; }
	PairBatchRunner& End() { m_first.End(); m_second.End(); return *this; }
	PairBatchRunner& Run(const char* text, const uint64_t* offsets, size_t n)
	{
		m_fused = false;
		m_first.Run(text, offsets, n);
		m_second.Run(text, offsets, n);
		return *this;
	}
	PairBatchRunner& Run(const std::vector<ystring>& strings)
	{
		m_fused = false;
		m_first.Run(strings);
		m_second.Run(strings);
		return *this;
	}
	/* Device-resident fixed-length records: one fused pass. */
	PairBatchRunner& RunDeviceStrided(const void* deviceText, size_t n, size_t len, size_t stride, void* stream = nullptr)
	{
		m_fused = true;
		m_ran = false;
		m_text = deviceText;
		m_n = n;
		m_len = len;
		m_stride = stride;
		m_stream = stream;
		return *this;
	}

	/* RunHelper<ScannerPair>::State() per string. */
	std::vector<State> States()
	{
		std::vector<State> out;
		if (m_fused) {
			ExecuteFused();
			out.resize(m_n);
			for (size_t i = 0; i < m_n; ++i)
				out[i] = ymake_pair(m_first.GetTable().ToState(m_idx1[i]), m_second.GetTable().ToState(m_idx2[i]));
			return out;
		}
		const std::vector<typename Scanner1::State>& a = m_first.States();
		const std::vector<typename Scanner2::State>& b = m_second.States();
		out.resize(a.size());
		for (size_t i = 0; i < a.size(); ++i)
			out[i] = ymake_pair(a[i], b[i]);
		return out;
	}
	/* ScannerPair::Final per string (pair.h:69-72). */
	std::vector<char> Finals()
	{
		if (m_fused) {
			ExecuteFused();
			return std::vector<char>(m_fin.begin(), m_fin.end());
		}
		const std::vector<char>& a = m_first.Finals();
		const std::vector<char>& b = m_second.Finals();
		std::vector<char> out(a.size());
		for (size_t i = 0; i < a.size(); ++i)
			out[i] = char(a[i] || b[i]);
		return out;
	}
	BatchRunner<Scanner1>& First() { return m_first; }
	BatchRunner<Scanner2>& Second() { return m_second; }

private:
	void ExecuteFused()
	{
		if (m_ran)
			return;
		uint32_t* d1 = static_cast<uint32_t*>(m_dev1.Reserve(m_n * 4));
		uint32_t* d2 = static_cast<uint32_t*>(m_dev2.Reserve(m_n * 4));
		uint8_t* df = static_cast<uint8_t*>(m_devFin.Reserve(m_n));
		Check(pire_hip_run_pair_strided(m_first.GetTable().Handle(), m_second.GetTable().Handle(), m_text, m_n, m_len, m_stride,
		                                m_first.Flags() | PIRE_HIP_RUN_ON_DEVICE, d1, d2, df, m_stream));
		m_idx1.resize(m_n);
		m_idx2.resize(m_n);
		m_fin.resize(m_n);
		Check(pire_hip_copy_to_host(m_idx1.data(), d1, m_n * 4, m_stream));
		Check(pire_hip_copy_to_host(m_idx2.data(), d2, m_n * 4, m_stream));
		Check(pire_hip_copy_to_host(m_fin.data(), df, m_n, m_stream));
		Check(pire_hip_stream_synchronize(m_stream));
		m_ran = true;
	}

	BatchRunner<Scanner1> m_first;
	BatchRunner<Scanner2> m_second;
	bool m_fused, m_ran;
	const void* m_text;
	size_t m_n, m_len, m_stride;
	void* m_stream;
	std::vector<uint32_t> m_idx1, m_idx2;
	std::vector<uint8_t> m_fin;
	DeviceBuffer m_dev1, m_dev2, m_devFin;
};

/*
 * Batched Runner over a Pire::HalfFinalScanner (scanners/half_final.h): per string the per-regexp match counts
 * State::Result(r) (half_final.h:90-92) that Initialize + Begin() + Run() + End() accumulate through TakeAction
 * (half_final.h:137-164), plus Final and StateIndex of the end state.  The scanner's State is an opaque class with
 * private members, so the results come back as plain numbers.
 */
template <class HalfScanner = Pire::HalfFinalScanner>
class HalfFinalBatchRunner {
public:
	explicit HalfFinalBatchRunner(const HalfScanner& sc)
	    : m_table(nullptr), m_regexps(sc.RegexpsCount()), m_flags(0), m_text(nullptr), m_offsets(nullptr), m_n(0), m_ran(false)
	{
		std::ostringstream out;
		sc.Save(&out);                                    // Scanner::Save, multi.h:557-573 (inherited)
		const std::string blob = out.str();
		Check(pire_hip_table_create(blob.data(), blob.size(), &m_table));
	}
	~HalfFinalBatchRunner() { pire_hip_table_destroy(m_table); }

	HalfFinalBatchRunner& Begin() { m_flags |= PIRE_HIP_RUN_BEGIN; return *this; }
	HalfFinalBatchRunner& End() { m_flags |= PIRE_HIP_RUN_END; return *this; }
	HalfFinalBatchRunner& Run(const char* text, const uint64_t* offsets, size_t n)
	{
		m_text = text;
		m_offsets = offsets;
		m_n = n;
		m_ran = false;
		return *this;
	}
	HalfFinalBatchRunner& Run(const std::vector<ystring>& strings)
	{
		m_ownText.clear();
		m_ownOffsets.assign(1, 0);
		for (size_t i = 0; i < strings.size(); ++i) {
			m_ownText.append(strings[i]);
			m_ownOffsets.push_back(m_ownText.size());
		}
		return Run(m_ownText.data(), m_ownOffsets.data(), strings.size());
	}

	/* State::Result(r) of string i. */
	size_t Result(size_t i, size_t r) { Execute(); return m_results[i * m_regexps + r]; }
	const std::vector<uint32_t>& Results() { Execute(); return m_results; }       // [n][RegexpsCount()]
	const std::vector<char>& Finals() { Execute(); return m_final; }              // Final(State()) per string
	const std::vector<uint32_t>& StateIndices() { Execute(); return m_idx; }      // StateIndex(State()) per string

private:
	void Execute()
	{
		if (m_ran)
			return;
		m_idx.assign(m_n, 0);
		std::vector<uint8_t> fin(m_n);
		m_results.assign(m_n * (m_regexps ? m_regexps : 1), 0);
		static const uint64_t kNoOffsets[1] = {0};
		Check(pire_hip_run_half_final(m_table, m_text, m_n ? m_offsets : kNoOffsets, m_n, m_flags, m_idx.data(), fin.data(),
		                              m_results.data(), nullptr));
		m_final.assign(fin.begin(), fin.end());
		m_ran = true;
	}

	HalfFinalBatchRunner(const HalfFinalBatchRunner&);
	HalfFinalBatchRunner& operator=(const HalfFinalBatchRunner&);
	pire_hip_table* m_table;
	size_t m_regexps;
	uint32_t m_flags;
	const char* m_text;
	const uint64_t* m_offsets;
	size_t m_n;
	bool m_ran;
	std::vector<uint32_t> m_idx, m_results;
	std::vector<char> m_final;
	ystring m_ownText;
	std::vector<uint64_t> m_ownOffsets;
};

/*
 * Batched Runner over a Pire::CountingScanner, AdvancedCountingScanner or NoGlueLimitCountingScanner (extra/count.h; include <pire/extra.h>
 * before this header): per string State::Result(r) for every glued regexp (count.h:206) after
 * Initialize + Begin() + Run() + End(), as tests/count_ut.cpp:54-63 drives them.
 */
#ifdef PIRE_EXTRA_COUNT_H
template <class CountScanner>
struct CountingKind;
template <>
struct CountingKind<Pire::CountingScanner> {
	enum { Value = PIRE_HIP_COUNTING_BASIC };
};
template <>
struct CountingKind<Pire::AdvancedCountingScanner> {
	enum { Value = PIRE_HIP_COUNTING_ADVANCED };
};
template <>
struct CountingKind<Pire::NoGlueLimitCountingScanner> {
	enum { Value = PIRE_HIP_COUNTING_NOGLUELIMIT };
};

template <class CountScanner>
class CountingBatchRunner {
public:
	explicit CountingBatchRunner(const CountScanner& sc)
	    : m_table(nullptr), m_regexps(sc.RegexpsCount()), m_flags(0), m_text(nullptr), m_offsets(nullptr), m_n(0), m_ran(false)
	{
		std::ostringstream out;
		sc.Save(&out);                                    // LoadedScanner::Save, scanner_io.cpp:172-189 / count.cpp:1009-1018
		const std::string blob = out.str();
		Check(pire_hip_counting_table_create(blob.data(), blob.size(), &m_table));
	}
	~CountingBatchRunner() { pire_hip_counting_table_destroy(m_table); }

	CountingBatchRunner& Begin() { m_flags |= PIRE_HIP_RUN_BEGIN; return *this; }
	CountingBatchRunner& End() { m_flags |= PIRE_HIP_RUN_END; return *this; }
	CountingBatchRunner& Run(const char* text, const uint64_t* offsets, size_t n)
	{
		m_text = text;
		m_offsets = offsets;
		m_n = n;
		m_ran = false;
		return *this;
	}
	CountingBatchRunner& Run(const std::vector<ystring>& strings)
	{
		m_ownText.clear();
		m_ownOffsets.assign(1, 0);
		for (size_t i = 0; i < strings.size(); ++i) {
			m_ownText.append(strings[i]);
			m_ownOffsets.push_back(m_ownText.size());
		}
		return Run(m_ownText.data(), m_ownOffsets.data(), strings.size());
	}

	/* State::Result(r) of string i. */
	size_t Result(size_t i, size_t r) { Execute(); return m_results[i * m_regexps + r]; }
	const std::vector<uint32_t>& Results() { Execute(); return m_results; }       // [n][RegexpsCount()]
	const std::vector<uint32_t>& StateIndices() { Execute(); return m_idx; }      // StateIndex(State()) per string

private:
	void Execute()
	{
		if (m_ran)
			return;
		m_idx.assign(m_n, 0);
		m_results.assign(m_n * (m_regexps ? m_regexps : 1), 0);
		static const uint64_t kNoOffsets[1] = {0};
		Check(pire_hip_counting_run(m_table, CountingKind<CountScanner>::Value, m_text, m_n ? m_offsets : kNoOffsets, m_n,
		                            m_flags, m_idx.data(), m_results.data(), nullptr));
		m_ran = true;
	}

	CountingBatchRunner(const CountingBatchRunner&);
	CountingBatchRunner& operator=(const CountingBatchRunner&);
	pire_hip_counting_table* m_table;
	size_t m_regexps;
	uint32_t m_flags;
	const char* m_text;
	const uint64_t* m_offsets;
	size_t m_n;
	bool m_ran;
	std::vector<uint32_t> m_idx, m_results;
	ystring m_ownText;
	std::vector<uint64_t> m_ownOffsets;
};
#endif  // PIRE_EXTRA_COUNT_H

/*
 * Batched Runner over a Pire::CapturingScanner (extra/capture.h:49-162; include <pire/extra.h> before this header):
 * per string State::Captured(), Begin(), End() and Final after Initialize + Begin() + Run() + End(), as
 * tests/capture_ut.cpp:75-83 drives it.  The captured text of string i is
 * [text + offsets[i] + Begin(i) - 1, text + offsets[i] + End(i) - 1) (capture_ut.cpp:85-91).
 */
#ifdef PIRE_EXTRA_CAPTURE_H
class CaptureBatchRunner {
public:
	explicit CaptureBatchRunner(const Pire::CapturingScanner& sc)
	    : m_table(nullptr), m_flags(0), m_text(nullptr), m_offsets(nullptr), m_n(0), m_ran(false)
	{
		std::ostringstream out;
		sc.Save(&out);                                    // LoadedScanner::Save, scanner_io.cpp:172-189
		const std::string blob = out.str();
		Check(pire_hip_counting_table_create(blob.data(), blob.size(), &m_table));
	}
	~CaptureBatchRunner() { pire_hip_counting_table_destroy(m_table); }

	CaptureBatchRunner& Begin() { m_flags |= PIRE_HIP_RUN_BEGIN; return *this; }
	CaptureBatchRunner& End() { m_flags |= PIRE_HIP_RUN_END; return *this; }
	CaptureBatchRunner& Run(const char* text, const uint64_t* offsets, size_t n)
	{
		m_text = text;
		m_offsets = offsets;
		m_n = n;
		m_ran = false;
		return *this;
	}
	CaptureBatchRunner& Run(const std::vector<ystring>& strings)
	{
		m_ownText.clear();
		m_ownOffsets.assign(1, 0);
		for (size_t i = 0; i < strings.size(); ++i) {
			m_ownText.append(strings[i]);
			m_ownOffsets.push_back(m_ownText.size());
		}
		return Run(m_ownText.data(), m_ownOffsets.data(), strings.size());
	}

	bool Captured(size_t i) { Execute(); return m_begin[i] >= 0 && m_end[i] >= 0; }     // State::Captured()
	size_t Begin(size_t i) { Execute(); return size_t(m_begin[i]); }                    // State::Begin() (npos if unset)
	size_t End(size_t i) { Execute(); return size_t(m_end[i]); }                        // State::End()
	bool Final(size_t i) { Execute(); return m_final[i] != 0; }
	size_t StateIndex(size_t i) { Execute(); return m_idx[i]; }

private:
	void Execute()
	{
		if (m_ran)
			return;
		m_idx.assign(m_n, 0);
		m_final.assign(m_n, 0);
		m_begin.assign(m_n, -1);
		m_end.assign(m_n, -1);
		static const uint64_t kNoOffsets[1] = {0};
		Check(pire_hip_capture_run(m_table, m_text, m_n ? m_offsets : kNoOffsets, m_n, m_flags, m_idx.data(), m_final.data(),
		                           m_begin.data(), m_end.data(), nullptr));
		m_ran = true;
	}

	CaptureBatchRunner(const CaptureBatchRunner&);
	CaptureBatchRunner& operator=(const CaptureBatchRunner&);
	pire_hip_counting_table* m_table;
	uint32_t m_flags;
	const char* m_text;
	const uint64_t* m_offsets;
	size_t m_n;
	bool m_ran;
	std::vector<uint32_t> m_idx;
	std::vector<uint8_t> m_final;
	std::vector<int64_t> m_begin, m_end;
	ystring m_ownText;
	std::vector<uint64_t> m_ownOffsets;
};
#endif  // PIRE_EXTRA_CAPTURE_H

/*
 * Batched Runner over a Pire::SlowScanner (scanners/slow.h): Matches(sc, str) per string, i.e.
 * Final(Runner(sc).Begin().Run(str).End().State()).
 */
class SlowBatchRunner {
public:
	explicit SlowBatchRunner(const Pire::SlowScanner& sc)
	    : m_table(nullptr)
	{
		std::ostringstream out;
		sc.Save(&out);                                    // scanner_io.cpp:71-111
		const std::string blob = out.str();
		Check(pire_hip_slow_table_create(blob.data(), blob.size(), &m_table));
	}
	~SlowBatchRunner() { pire_hip_slow_table_destroy(m_table); }

	/* Begin().Run().End() for n strings; returns operator bool of the reference's RunHelper per string. */
	std::vector<char> Matches(const char* text, const uint64_t* offsets, size_t n)
	{
		std::vector<uint8_t> fin(n);
		static const uint64_t kNoOffsets[1] = {0};
		Check(pire_hip_slow_run(m_table, text, n ? offsets : kNoOffsets, n, PIRE_HIP_RUN_BEGIN | PIRE_HIP_RUN_END,
		                        fin.data(), nullptr, nullptr, nullptr));
		return std::vector<char>(fin.begin(), fin.end());
	}

private:
	SlowBatchRunner(const SlowBatchRunner&);
	SlowBatchRunner& operator=(const SlowBatchRunner&);
	pire_hip_slow_table* m_table;
};

}  // namespace Hip
}  // namespace Pire

#endif
