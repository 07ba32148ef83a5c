// tools/h2dbench.hip — how fast do pinned host blocks reach HBM on this box?  (round 6: the CSR upload moves 160-480 MB per query)
//   hipcc --offload-arch=gfx950 -O2 -o tools/h2dbench tools/h2dbench.hip && tools/h2dbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy_from_host(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_narrow_from_host(const long long *__restrict__ in, int *__restrict__ out, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (int)in[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	const size_t total = 256u << 20;
	void *h = nullptr, *d = nullptr;
	CK(hipHostMalloc(&h, total));
	CK(hipMalloc(&d, total));
	memset(h, 1, total);
	hipStream_t st[4];
	for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (size_t blk : { (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20 })
		for (int ns : { 1, 2, 4 }) {
			double best = 1e9;
			for (int rep = 0; rep < 3; rep++) {
				const double t0 = now();
				size_t k = 0;
				for (size_t lo = 0; lo < total; lo += blk, k++) CK(hipMemcpyAsync((char *)d + lo, (char *)h + lo, blk, hipMemcpyHostToDevice, st[k % ns]));
				const double t1 = now();
				for (int s = 0; s < ns; s++) CK(hipStreamSynchronize(st[s]));
				const double t2 = now();
				if (t2 - t0 < best) best = t2 - t0;
				if (rep == 2) printf("memcpyAsync block %4zu MB x %d streams: %6.2f ms = %5.1f GB/s (host time in the calls %.2f ms)\n", blk >> 20, ns, best * 1e3, total / best / 1e9, (t1 - t0) * 1e3);
			}
		}
	void *hd = nullptr;
	CK(hipHostGetDevicePointer(&hd, h, 0));
	for (int grid : { 64, 256, 1024 }) {
		double best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			const double t0 = now();
			hipLaunchKernelGGL(k_copy_from_host, dim3(grid), dim3(256), 0, st[0], (const uint4 *)hd, (uint4 *)d, total / 16);
			CK(hipStreamSynchronize(st[0]));
			best = std::min(best, now() - t0);
		}
		printf("kernel reading pinned host memory, %4d workgroups: %6.2f ms = %5.1f GB/s\n", grid, best * 1e3, total / best / 1e9);
		best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			const double t0 = now();
			hipLaunchKernelGGL(k_narrow_from_host, dim3(grid), dim3(256), 0, st[0], (const long long *)hd, (int *)d, total / 8);
			CK(hipStreamSynchronize(st[0]));
			best = std::min(best, now() - t0);
		}
		printf("kernel narrowing int64 -> int32 out of pinned host memory, %4d workgroups: %6.2f ms = %5.1f GB/s of host bytes\n", grid, best * 1e3, total / best / 1e9);
	}
	// host side: how fast do T threads fill pinned memory from pageable memory (memcpy)?
	std::vector<char> src(total, 2);
	for (int T : { 1, 2, 4, 8, 16, 32 }) {
		double best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			const double t0 = now();
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++) th.emplace_back([&, t] { memcpy((char *)h + total / T * t, src.data() + total / T * t, total / T); });
			for (auto &x : th) x.join();
			best = std::min(best, now() - t0);
		}
		printf("host memcpy pageable -> pinned, %2d threads: %6.2f ms = %5.1f GB/s\n", T, best * 1e3, total / best / 1e9);
	}
	// hipHostRegister of pageable memory
	{
		const double t0 = now();
		hipError_t e = hipHostRegister(src.data(), total, hipHostRegisterDefault);
		const double t1 = now();
		printf("hipHostRegister of 256 MB pageable: %s, %.2f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3);
		if (e == hipSuccess) {
			const double t2 = now();
			CK(hipMemcpyAsync(d, src.data(), total, hipMemcpyHostToDevice, st[0]));
			CK(hipStreamSynchronize(st[0]));
			printf("  memcpy out of the registered range: %.2f ms = %.1f GB/s\n", (now() - t2) * 1e3, total / (now() - t2) / 1e9);
			const double t3 = now();
			(void)hipHostUnregister(src.data());
			printf("  unregister %.2f ms\n", (now() - t3) * 1e3);
		}
	}
	return 0;
}
