// membench.hip — two measurements the roofline figures lean on (tools/, not part of the product):
//   copy    16-byte copy kernels in several launch shapes: the box's own HBM ceiling (read + written bytes / time)
//   segments  whole 1 KB / 4 KB segments at random starts, four requests per wavefront in flight: the ceiling for the
//           list walks of k_meet3 / k_meet4
//   gather  a 16- or 32-byte gather of known size with random indices: run under `rocprofv3 --pmc FETCH_SIZE` to
//           calibrate the counter for the access pattern of k_pull_lanes / k_meet3 (MI355X_MICROARCH.md §HBM: the 2x
//           correction is only calibrated for wide coalesced reads)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip      usage: tools/membench copy | gather | segments
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                                          \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
			exit(1);                                                                                                   \
		}                                                                                                              \
	} while (0)

template <int UN, bool NT>
__global__ __launch_bounds__(1024) void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, long n) {
	const long stride = (long)gridDim.x * blockDim.x;
	long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i + (UN - 1) * stride < n; i += UN * stride) {
		uint4 v[UN];
#pragma unroll
		for (int k = 0; k < UN; k++) {
			if (NT) {
				const unsigned long long *p = reinterpret_cast<const unsigned long long *>(in + i + k * stride);
				const unsigned long long lo = __builtin_nontemporal_load(p), hi = __builtin_nontemporal_load(p + 1);
				v[k] = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
			} else {
				v[k] = in[i + k * stride];
			}
		}
#pragma unroll
		for (int k = 0; k < UN; k++) out[i + k * stride] = v[k];
	}
	for (; i < n; i += stride) out[i] = in[i];
}

// every lane reads REC bytes at a random record of a table of `nrec` records (index list precomputed)
template <int REC>
__global__ void k_gather(const unsigned *__restrict__ idx, long n, const uint4 *__restrict__ table, uint4 *__restrict__ sink) {
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint4 *p = table + (size_t)idx[i] * (REC / 16);
	uint4 a = p[0];
	if (REC == 32) {
		const uint4 b = p[1];
		a.x ^= b.x;
		a.y ^= b.y;
	}
	if (a.x == 0x12345678u && a.y == 0x9abcdef0u) sink[0] = a; // never true: keeps the loads
}

// every wavefront reads whole segments (SEGW x 16 bytes per lane-row: 64 lanes x 16 B = 1 KB per request) at random
// 16-byte-aligned starts, four requests in flight: the access pattern of k_meet3 / k_meet4 (adjacency lists of a few
// hundred entries picked by a pair's one-hop list)
__global__ __launch_bounds__(256) void k_seggather(const unsigned *__restrict__ starts, long nseg, int seg_lines,
                                                   const uint4 *__restrict__ table, uint4 *__restrict__ sink) {
	const int lane = threadIdx.x & 63;
	const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
	uint4 acc = make_uint4(0, 0, 0, 0);
	for (long i = wave; i < nseg; i += nwaves) {
		const uint4 *p = table + starts[i] + lane;
		for (int l = 0; l < seg_lines; l += 4) {
			uint4 v[4];
#pragma unroll
			for (int k = 0; k < 4; k++) v[k] = l + k < seg_lines ? p[(l + k) * 64] : make_uint4(0, 0, 0, 0);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				acc.x ^= v[k].x;
				acc.y ^= v[k].y;
			}
		}
	}
	if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

template <int UN, bool NT> static void run_copy(const char *name, void *a, void *b, long n, int grid, int block) {
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_copy<UN, NT>), dim3(grid), dim3(block), 0, 0, (const uint4 *)a, (uint4 *)b, n);
	CK(hipEventRecord(e0, 0));
	const int iters = 10;
	for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_copy<UN, NT>), dim3(grid), dim3(block), 0, 0, (const uint4 *)a, (uint4 *)b, n);
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	printf("{\"kernel\": \"%s\", \"grid\": %d, \"block\": %d, \"GBps_read_plus_write\": %.0f}\n", name, grid, block,
	       2.0 * n * 16 * iters / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
	const char *mode = argc > 1 ? argv[1] : "copy";
	if (!strcmp(mode, "copy")) {
		const long bytes = 2l << 30, n = bytes / 16;
		void *a, *b;
		CK(hipMalloc(&a, bytes));
		CK(hipMalloc(&b, bytes));
		CK(hipMemset(a, 1, bytes));
		for (int block : { 256, 512, 1024 })
			for (int per_cu : { 4, 8, 16, 32 }) {
				const int grid = 256 * per_cu * 256 / block;
				run_copy<1, false>("copy16", a, b, n, grid, block);
				run_copy<4, false>("copy16x4", a, b, n, grid, block);
				run_copy<4, true>("copy16x4_nt", a, b, n, grid, block);
			}
		return 0;
	}
	if (!strcmp(mode, "segments")) {
		// random 1 KB / 4 KB segments out of a 160 MB table (the SF100 adjacency) and a 1 GB table
		const long nseg = 4l << 20;
		std::vector<unsigned> h(nseg);
		unsigned x = 777;
		void *idx, *table, *sink;
		CK(hipMalloc(&idx, nseg * 4));
		CK(hipMalloc(&table, 1l << 30));
		CK(hipMalloc(&sink, 64));
		CK(hipMemset(table, 0, 1l << 30));
		for (long table_bytes : { 160l << 20, 1l << 30 })
			for (int seg_lines : { 1, 4 })
				for (int waves_per_simd : { 4, 8 }) {
					const unsigned n16 = (unsigned)((table_bytes - (long)seg_lines * 1024) / 16);
					for (long i = 0; i < nseg; i++) {
						x = x * 1664525u + 1013904223u;
						h[i] = (x >> 3) % n16;
					}
					CK(hipMemcpy(idx, h.data(), nseg * 4, hipMemcpyHostToDevice));
					hipEvent_t e0, e1;
					CK(hipEventCreate(&e0));
					CK(hipEventCreate(&e1));
					const int grid = 256 * waves_per_simd; // 256 threads = one wave per SIMD per workgroup
					hipLaunchKernelGGL(k_seggather, dim3(grid), dim3(256), 0, 0, (const unsigned *)idx, nseg / 16, seg_lines, (const uint4 *)table, (uint4 *)sink);
					CK(hipEventRecord(e0, 0));
					hipLaunchKernelGGL(k_seggather, dim3(grid), dim3(256), 0, 0, (const unsigned *)idx, nseg, seg_lines, (const uint4 *)table, (uint4 *)sink);
					CK(hipEventRecord(e1, 0));
					CK(hipEventSynchronize(e1));
					float ms = 0;
					CK(hipEventElapsedTime(&ms, e0, e1));
					printf("{\"kernel\": \"seggather\", \"table_MB\": %ld, \"segment_bytes\": %d, \"waves_per_simd\": %d, "
					       "\"segments\": %ld, \"ms\": %.3f, \"GBps\": %.0f}\n",
					       table_bytes >> 20, seg_lines * 1024, waves_per_simd, nseg, ms, (double)nseg * seg_lines * 1024 / (ms * 1e-3) / 1e9);
				}
		return 0;
	}
	// gather: 64 M random reads of 16 / 32 bytes from tables of 4 MB (L2-resident) and 1 GB (HBM)
	const long n = 64l << 20;
	std::vector<unsigned> h(n);
	unsigned x = 12345;
	void *idx, *table, *sink;
	CK(hipMalloc(&idx, n * 4));
	CK(hipMalloc(&table, 1l << 30));
	CK(hipMalloc(&sink, 64));
	CK(hipMemset(table, 0, 1l << 30));
	for (long table_bytes : { 4l << 20, 1l << 30 })
		for (int rec : { 16, 32 }) {
			const unsigned nrec = (unsigned)(table_bytes / rec);
			for (long i = 0; i < n; i++) {
				x = x * 1664525u + 1013904223u;
				h[i] = (x >> 4) % nrec;
			}
			CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0, 0));
			if (rec == 16) hipLaunchKernelGGL(k_gather<16>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const unsigned *)idx, n, (const uint4 *)table, (uint4 *)sink);
			else hipLaunchKernelGGL(k_gather<32>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const unsigned *)idx, n, (const uint4 *)table, (uint4 *)sink);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			printf("{\"kernel\": \"gather%d\", \"table_MB\": %ld, \"reads\": %ld, \"index_bytes\": %ld, \"payload_bytes\": %ld, "
			       "\"ms\": %.3f, \"Greads_per_s\": %.1f}\n",
			       rec, table_bytes >> 20, n, n * 4, n * rec, ms, n / (ms * 1e-3) / 1e9);
		}
	return 0;
}
