// What a small host-pointer call can cost at best: launch + synchronise alone, with staged copies either side, and
// with the kernel reading its input from / writing its results to pinned host memory directly (no copies enqueued).
//   hipcc --offload-arch=gfx950 -O3 -o micro_smallcall tools/micro_smallcall.hip && ./micro_smallcall
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                  \
	do {                                                                                          \
		hipError_t e_ = (x);                                                                      \
		if (e_ != hipSuccess) {                                                                   \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                               \
			return 1;                                                                             \
		}                                                                                         \
	} while (0)

// a stand-in for a scan: every lane walks its string byte by byte through a 256-entry table in LDS
__global__ void Walk(const uint8_t* text, const uint64_t* offs, uint32_t n, uint32_t* out)
{
	__shared__ uint8_t tab[256];
	tab[threadIdx.x & 255] = uint8_t(threadIdx.x * 7 + 1);
	__syncthreads();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	uint32_t st = 0;
	for (uint64_t k = offs[i]; k < offs[i + 1]; ++k)
		st = tab[(st + text[k]) & 255];
	out[i] = st;
}

template <class F>
static double Median(F f, int reps = 200)
{
	std::vector<double> ts;
	for (int r = 0; r < reps; ++r) {
		auto t0 = std::chrono::steady_clock::now();
		f();
		ts.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
	}
	std::sort(ts.begin(), ts.end());
	return ts[ts.size() / 2];
}

int main()
{
	hipStream_t s;
	CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (uint32_t n : {10u, 1000u}) {
		for (uint32_t len : {100u, 1000u}) {
			const size_t bytes = size_t(n) * len;
			std::vector<uint8_t> text(bytes, 'a');
			std::vector<uint64_t> offs(n + 1);
			for (uint32_t i = 0; i <= n; ++i)
				offs[i] = uint64_t(i) * len;
			std::vector<uint32_t> res(n);
			uint8_t *dText, *pText;
			uint64_t *dOffs, *pOffs;
			uint32_t *dOut, *pOut;
			CHECK(hipMalloc(&dText, bytes));
			CHECK(hipMalloc(&dOffs, (n + 1) * 8));
			CHECK(hipMalloc(&dOut, n * 4));
			CHECK(hipHostMalloc(&pText, bytes, hipHostMallocDefault));
			CHECK(hipHostMalloc(&pOffs, (n + 1) * 8, hipHostMallocDefault));
			CHECK(hipHostMalloc(&pOut, n * 4, hipHostMallocDefault));
			const unsigned blocks = (n + 255) / 256;
			const double launchOnly = Median([&] {
				hipLaunchKernelGGL(Walk, dim3(blocks), dim3(256), 0, s, dText, dOffs, n, dOut);
				(void)hipStreamSynchronize(s);
			});
			const double staged = Median([&] {
				memcpy(pText, text.data(), bytes);
				memcpy(pOffs, offs.data(), (n + 1) * 8);
				(void)hipMemcpyAsync(dText, pText, bytes, hipMemcpyHostToDevice, s);
				(void)hipMemcpyAsync(dOffs, pOffs, (n + 1) * 8, hipMemcpyHostToDevice, s);
				hipLaunchKernelGGL(Walk, dim3(blocks), dim3(256), 0, s, dText, dOffs, n, dOut);
				(void)hipMemcpyAsync(pOut, dOut, n * 4, hipMemcpyDeviceToHost, s);
				(void)hipStreamSynchronize(s);
				memcpy(res.data(), pOut, n * 4);
			});
			const double stagedOneCopy = Median([&] {   // text and offsets in one block, one copy each way
				memcpy(pText, text.data(), bytes);
				(void)hipMemcpyAsync(dText, pText, bytes, hipMemcpyHostToDevice, s);
				hipLaunchKernelGGL(Walk, dim3(blocks), dim3(256), 0, s, dText, dOffs, n, dOut);
				(void)hipMemcpyAsync(pOut, dOut, n * 4, hipMemcpyDeviceToHost, s);
				(void)hipStreamSynchronize(s);
				memcpy(res.data(), pOut, n * 4);
			});
			const double outDirect = Median([&] {       // results written to pinned host memory by the kernel
				memcpy(pText, text.data(), bytes);
				memcpy(pOffs, offs.data(), (n + 1) * 8);
				(void)hipMemcpyAsync(dText, pText, bytes, hipMemcpyHostToDevice, s);
				(void)hipMemcpyAsync(dOffs, pOffs, (n + 1) * 8, hipMemcpyHostToDevice, s);
				hipLaunchKernelGGL(Walk, dim3(blocks), dim3(256), 0, s, dText, dOffs, n, pOut);
				(void)hipStreamSynchronize(s);
				memcpy(res.data(), pOut, n * 4);
			});
			const double allDirect = Median([&] {       // input read from pinned host memory too: one launch, nothing else
				memcpy(pText, text.data(), bytes);
				memcpy(pOffs, offs.data(), (n + 1) * 8);
				hipLaunchKernelGGL(Walk, dim3(blocks), dim3(256), 0, s, pText, pOffs, n, pOut);
				(void)hipStreamSynchronize(s);
				memcpy(res.data(), pOut, n * 4);
			});
			const double offsDirect = Median([&] {      // text staged (one copy), offsets and results direct
				memcpy(pText, text.data(), bytes);
				memcpy(pOffs, offs.data(), (n + 1) * 8);
				(void)hipMemcpyAsync(dText, pText, bytes, hipMemcpyHostToDevice, s);
				hipLaunchKernelGGL(Walk, dim3(blocks), dim3(256), 0, s, dText, pOffs, n, pOut);
				(void)hipStreamSynchronize(s);
				memcpy(res.data(), pOut, n * 4);
			});
			printf("%5u strings x %4u B: launch+sync %.1f us | 2 H2D + kernel + D2H %.1f | 1 H2D + kernel + D2H %.1f | 2 H2D + kernel, results "
			       "direct %.1f | 1 H2D, offsets + results direct %.1f | everything direct %.1f\n",
			       n, len, launchOnly, staged, stagedOneCopy, outDirect, offsDirect, allDirect);
			(void)hipFree(dText);
			(void)hipFree(dOffs);
			(void)hipFree(dOut);
			(void)hipHostFree(pText);
			(void)hipHostFree(pOffs);
			(void)hipHostFree(pOut);
		}
	}
	return 0;
}
