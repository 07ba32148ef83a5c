// pgq_analytics.hip — the other consumers of the device CSR: local clustering coefficient,
// PageRank and weakly connected components.
//
//   local_clustering_coefficient   src/core/functions/scalar/local_clustering_coefficient.cpp:11-72
//       count = sum over the SLOTS of src's adjacency of |{slots of that neighbour's adjacency whose vertex is a
//       neighbour of src}|, result = float(count) / (deg_f * (deg_f - 1)), deg = number of slots.  Integer counting +
//       three IEEE float operations: bit-identical to the reference.  It is the two-hop walk of pgq_meet.hip with a
//       counter instead of a witness: one wavefront per row, the neighbour set in the LDS hash table + bit filter
//       (lists up to 512); longer lists take one 1024-thread workgroup per row and a vertex bit map (LDS when it fits,
//       a per-workgroup slice of global memory otherwise).
//   pagerank                       src/core/functions/scalar/pagerank.cpp:11-111
//       power iteration over v_size = V + 2 entries (the two trailing CSR offsets behave as dangling vertices), damping
//       0.85, stop when the largest change is below 1e-6.  Pull form: in-lists sorted by (source, slot) — a stable radix
//       sort of the forward slots by destination — and summed left to right by one thread per vertex, i.e. in exactly
//       the order the reference's `temp_rank[neighbor] += rank_contrib` accumulates, so every partial sum matches bit
//       for bit; only the dangling-rank total is a two-level sum (1024 ordered slices, then their ordered sum) instead
//       of one sequential chain, hence a tolerance of 1e-12 relative in the tests.  Computed once per CSR handle.
//
//   weakly_connected_component    src/core/functions/scalar/weakly_connected_component.cpp:14-104
//       the reference's component id is the root its sequential union-find schedule ends in (:14-34,83-90: goldens pin
//       e.g. id 2 for the cycle 0-1-2-3), not a canonical label.  The schedule is Kruskal's algorithm in CSR slot order, so
//       the O(E) part — which slots change the forest — is the minimum spanning forest under the weight "slot index":
//       Boruvka rounds on the device (k_wcc_*, below); the <= V - 1 chosen slots are then replayed in slot order with the
//       reference's Link on the host.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <mutex>

#include "pgq_search.h"
#include "pgq_walk.h"

namespace pgq {

// ---- local clustering coefficient ---------------------------------------------------------------------------------

constexpr float kLccBig = -2.0f; // d_out marker: the row's list is too long for the hash table

__device__ __forceinline__ float lcc_value(unsigned long long count, int deg) {
	const float df = static_cast<float>(deg);
	return static_cast<float>((long long)count) / (df * (df - 1.0f));
}

__global__ __launch_bounds__(64, 6) void k_lcc(int64_t n, const int64_t *__restrict__ src, int64_t V,
                                               const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                               float *__restrict__ out, u32 *__restrict__ counters) {
	__shared__ __attribute__((aligned(16))) u32 tab[kMeetSlots];
	__shared__ __attribute__((aligned(16))) u32 bm[kMeetFilterWords];
	const int lane = threadIdx.x & 63;
	const int64_t nwaves = (int64_t)gridDim.x;
	for (int64_t i = blockIdx.x; i < n; i += nwaves) {
		const int64_t s = src[i];
		if (s < 0) { // NULL row
			if (lane == 0) out[i] = 0.0f;
			continue;
		}
		if (s >= V) {
			if (lane == 0) {
				counters[1] = 1;
				out[i] = 0.0f;
			}
			continue;
		}
		const int b = (int)off[s], deg = (int)off[s + 1] - b;
		if (deg < 2) { // local_clustering_coefficient.cpp:44-47
			if (lane == 0) out[i] = 0.0f;
			continue;
		}
		if (deg > kMeetSetMax) {
			if (lane == 0) {
				out[i] = kLccBig;
				atomicAdd(&counters[0], 1u);
			}
			continue;
		}
#pragma unroll
		for (int k = 0; k < kMeetSlots / 256; k++)
			reinterpret_cast<uint4_alias *>(tab)[k * 64 + lane] = make_uint4(kMeetEmpty, kMeetEmpty, kMeetEmpty, kMeetEmpty);
#pragma unroll
		for (int k = 0; k < kMeetFilterWords / 256; k++) reinterpret_cast<uint4_alias *>(bm)[k * 64 + lane] = make_uint4(0, 0, 0, 0);
		__builtin_amdgcn_wave_barrier();
		for (int p = lane; p < deg; p += 64) {
			const u32 x = (u32)adj[b + p];
			const u32 fh = meet_fhash(x);
			atomicOr(&bm[fh >> 5], 1u << (fh & 31));
			u32 h = meet_hash(x);
			for (;;) {
				const u32 old = atomicCAS(&tab[h], kMeetEmpty, x);
				if (old == kMeetEmpty || old == x) break;
				h = (h + 1) & (kMeetSlots - 1);
			}
		}
		__builtin_amdgcn_wave_barrier();
		unsigned long long count = 0;
		(void)meet_walk(adj + b, deg, 0, 1, off, adj,
		                [&](u32 x, u32) {
			                const u32 fh = meet_fhash(x);
			                if (((bm[fh >> 5] >> (fh & 31)) & 1u) && meet_lookup(tab, x)) count++;
		                },
		                []() { return false; });
		for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
		if (lane == 0) out[i] = lcc_value(count, deg);
	}
}

__global__ void k_lcc_collect_big(int64_t n, const float *__restrict__ out, u32 *__restrict__ rows, u32 *__restrict__ count) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && out[i] == kLccBig) rows[atomicAdd(count, 1u)] = (u32)i;
}

// one 1024-thread workgroup per long-list row; `gmap` != null: the vertex bit map lives in this workgroup's slice of
// global memory (V too large for LDS)
__global__ __launch_bounds__(1024) void k_lcc_big(u32 nrows, const u32 *__restrict__ rows, const int64_t *__restrict__ src,
                                                  const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                  float *__restrict__ out, int bm_words, u32 *__restrict__ gmap) {
	extern __shared__ u32 s_lmap[];
	__shared__ unsigned long long s_count;
	u32 *map = gmap ? gmap + (size_t)blockIdx.x * (size_t)bm_words : s_lmap;
	const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
	for (u32 r = blockIdx.x; r < nrows; r += gridDim.x) {
		__syncthreads();
		const u32 i = rows[r];
		const int64_t s = src[i];
		const int b = (int)off[s], deg = (int)off[s + 1] - b;
		for (int k = tid; k < bm_words; k += 1024) map[k] = 0;
		if (tid == 0) s_count = 0;
		__syncthreads();
		for (int p = tid; p < deg; p += 1024) {
			const u32 x = (u32)adj[b + p];
			atomicOr(&map[x >> 5], 1u << (x & 31));
		}
		__syncthreads();
		unsigned long long count = 0;
		(void)meet_walk(adj + b, deg, wib, 16, off, adj, [&](u32 x, u32) { count += (map[x >> 5] >> (x & 31)) & 1u; },
		                []() { return false; });
		for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
		if (lane == 0 && count) atomicAdd(&s_count, count);
		__syncthreads();
		if (tid == 0) out[i] = lcc_value(s_count, deg);
	}
}

static int lcc_device(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, float *d_out) {
	hipStream_t st = ws->stream;
	if (n == 0) return PGQ_OK;
	if (n >= (1LL << 31)) return fail(PGQ_ERR_INVALID_ARG, "more than 2^31-1 rows in one call");
	PGQ_TRY(ws->counters.reserve(sizeof(Counters)));
	u32 *d_cnt = reinterpret_cast<u32 *>(ws->counters.p); // [0] long-list rows, [1] bad id, [2] compaction cursor
	PGQ_HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, st));
	hipLaunchKernelGGL(k_lcc, dim3((unsigned)std::min<int64_t>(n, (int64_t)device_cus() * 24 * 4)), dim3(64), 0, st, n, d_src, c->V, c->off, c->adj,
	                   d_out, d_cnt);
	u32 h[2] = { 0, 0 };
	PGQ_HIP_TRY(hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st));
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	if (h[1]) return fail(PGQ_ERR_INVALID_ARG, "vertex rowid out of range [0,V)");
	if (h[0] > 0) {
		const u32 nbig = h[0];
		PGQ_TRY(ws->def_idx.reserve((size_t)nbig * 4));
		hipLaunchKernelGGL(k_lcc_collect_big, dim3(blocks_for(n)), dim3(256), 0, st, n, d_out, ws->def_idx.as<u32>(), d_cnt + 2);
		const int bm_words = (int)((c->V + 31) / 32);
		const bool in_lds = (size_t)bm_words * 4 + 256 <= 150 * 1024;
		const unsigned grid = std::min<u32>(nbig, 256);
		u32 *gmap = nullptr;
		if (!in_lds) {
			PGQ_TRY(ws->lblk.reserve((size_t)grid * (size_t)bm_words * 4));
			gmap = ws->lblk.as<u32>();
		} else {
			static std::atomic<int> attr_set { 0 };
			if (!attr_set.load()) {
				(void)hipFuncSetAttribute((const void *)k_lcc_big, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
				attr_set.store(1);
			}
		}
		hipLaunchKernelGGL(k_lcc_big, dim3(grid), dim3(1024), in_lds ? (size_t)bm_words * 4 : 0, st, nbig, ws->def_idx.as<u32>(),
		                   d_src, c->off, c->adj, d_out, bm_words, gmap);
		PGQ_HIP_TRY(hipStreamSynchronize(st));
	}
	return PGQ_OK;
}

// ---- PageRank ---------------------------------------------------------------------------------------------------------

__global__ void k_pr_init(int64_t vs, double *__restrict__ rank) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < vs) rank[i] = 1.0 / static_cast<double>(vs);
}
// contrib[i] = rank[i] / out-degree (pagerank.cpp:55); dangling vertices (and the two trailing entries) contribute 0 here
__global__ void k_pr_contrib(int64_t V, int64_t vs, const int64_t *__restrict__ off, const double *__restrict__ rank,
                             double *__restrict__ contrib) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= vs) return;
	const int64_t deg = i < V ? off[i + 1] - off[i] : 0;
	contrib[i] = deg > 0 ? rank[i] / static_cast<double>(deg) : 0.0;
}
// total rank of the dangling entries: 1024 ordered slices, then their ordered sum (one workgroup)
__global__ __launch_bounds__(1024) void k_pr_dangling(int64_t V, int64_t vs, const int64_t *__restrict__ off,
                                                      const double *__restrict__ rank, double *__restrict__ total) {
	__shared__ double part[1024];
	const int64_t per = (vs + 1023) / 1024;
	const int64_t lo = (int64_t)threadIdx.x * per, hi = min(lo + per, vs);
	double s = 0.0;
	for (int64_t i = lo; i < hi; i++) {
		const bool dangling = i >= V || off[i + 1] == off[i];
		if (dangling) s += rank[i];
	}
	part[threadIdx.x] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		double t = 0.0;
		for (int k = 0; k < 1024; k++) t += part[k];
		*total = t;
	}
}
// new rank of vertex n: in-list summed left to right in (source, slot) order, then damping (pagerank.cpp:57-76)
__global__ void k_pr_pull(int64_t V, int64_t vs, const int64_t *__restrict__ roff, const int32_t *__restrict__ psrc,
                          const double *__restrict__ contrib, const double *__restrict__ rank,
                          const double *__restrict__ dangling, double *__restrict__ next, unsigned long long *__restrict__ max_delta) {
	const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	double delta = 0.0;
	if (n < vs) {
		double t = 0.0;
		if (n < V)
			for (int64_t k = roff[n]; k < roff[n + 1]; k++) t += contrib[psrc[k]];
		const double corr = *dangling / static_cast<double>(vs);
		const double r = (1 - 0.85) / static_cast<double>(vs) + 0.85 * (t + corr);
		next[n] = r;
		delta = fabs(r - rank[n]);
	}
	// non-negative doubles order like their bit patterns
	unsigned long long bits = (unsigned long long)__double_as_longlong(delta);
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned long long other = __shfl_xor(bits, o);
		bits = other > bits ? other : bits;
	}
	if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_delta, bits);
}

static int pagerank_compute(pgq_csr *c, Workspace *ws) {
	std::lock_guard<std::mutex> g(c->lazy_lock); // per handle (round 4: one process-wide mutex across the whole power iteration)
	if (c->pagerank) return PGQ_OK;
	hipStream_t st = ws->stream;
	const int64_t V = c->V, vs = V + 2;
	DevBuf rank[2], contrib, scal;
	auto cleanup = [&]() {
		for (DevBuf *b : { &rank[1], &contrib, &scal }) b->release();
	};
	// the reverse CSR lists the in-edges of a vertex by (source, forward slot): the order pagerank.cpp:60-66 adds in
	const int32_t *psrc = c->radj;
	auto body = [&]() -> int {
		for (DevBuf *b : { &rank[0], &rank[1], &contrib }) PGQ_TRY(b->reserve((size_t)vs * 8));
		PGQ_TRY(scal.reserve(64));
		hipLaunchKernelGGL(k_pr_init, dim3(blocks_for(vs)), dim3(256), 0, st, vs, rank[0].as<double>());
		double *d_dang = scal.as<double>();
		unsigned long long *d_max = reinterpret_cast<unsigned long long *>(scal.as<double>() + 1);
		int cur = 0, it = 0;
		for (;;) {
			PGQ_HIP_TRY(hipMemsetAsync(d_max, 0, 8, st));
			hipLaunchKernelGGL(k_pr_contrib, dim3(blocks_for(vs)), dim3(256), 0, st, V, vs, c->off, rank[cur].as<double>(),
			                   contrib.as<double>());
			hipLaunchKernelGGL(k_pr_dangling, dim3(1), dim3(1024), 0, st, V, vs, c->off, rank[cur].as<double>(), d_dang);
			hipLaunchKernelGGL(k_pr_pull, dim3(blocks_for(vs)), dim3(256), 0, st, V, vs, c->roff, psrc,
			                   contrib.as<double>(), rank[cur].as<double>(), d_dang, rank[cur ^ 1].as<double>(), d_max);
			unsigned long long bits = 0;
			PGQ_HIP_TRY(hipMemcpyAsync(&bits, d_max, 8, hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
			double max_delta;
			memcpy(&max_delta, &bits, 8);
			cur ^= 1;
			it++;
			if (max_delta < 1e-6 || it >= 10000) break;
		}
		c->pagerank_iterations = it;
		if (cur == 1) std::swap(rank[0], rank[1]);
		return PGQ_OK;
	};
	int rc = body();
	(void)hipStreamSynchronize(st);
	cleanup();
	if (rc != PGQ_OK) {
		rank[0].release();
		return rc;
	}
	c->pagerank = rank[0].as<double>(); // ownership moves to the CSR handle (freed with it)
	return PGQ_OK;
}

// ---- weakly connected components ---------------------------------------------------------------------------------------
// The reference's component id is NOT a canonical label: it is the root its sequential union-find ends in
// (weakly_connected_component.cpp:14-34,83-90: Link(i, neighbour) hangs i's root under the neighbour's root, vertices
// and slots in CSR order; the goldens pin e.g. id 2 for the cycle 0-1-2-3).  Two facts make that computable in parallel:
//   * an edge changes the forest iff its endpoints are in different trees when it is processed, i.e. iff it is an edge
//     of the MINIMUM SPANNING FOREST under the weights "CSR slot index" (the sequential pass IS Kruskal's algorithm in
//     slot order; the weights are distinct, so the forest is unique) — all other edges are no-ops;
//   * replaying only those <= V - 1 edges, in slot order, with the reference's Link gives the same forest roots.
// So the O(E) part runs on the device — Boruvka rounds: every component takes the smallest slot leaving it (atomicMin
// from both endpoints), hooks onto the other side (a mutual pair keeps the smaller root), pointer jumping — and the
// O(V alpha(V)) replay of the chosen edges runs on the host (it is the reference's own schedule, restricted to the edges
// that matter).  Computed once per CSR handle, like the reference's bind-data state.
__global__ void k_wcc_slot_src(int64_t V, const int64_t *__restrict__ off, u32 *__restrict__ slot_src) {
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	for (int64_t v = wave; v < V; v += nwaves)
		for (int64_t k = off[v] + lane; k < off[v + 1]; k += 64) slot_src[k] = (u32)v;
}
__global__ void k_wcc_init(int64_t V, u32 *__restrict__ comp) {
	const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v < V) comp[v] = (u32)v;
}
__global__ void k_wcc_min_edge(int64_t E, const u32 *__restrict__ slot_src, const int32_t *__restrict__ adj,
                               const u32 *__restrict__ comp, unsigned long long *__restrict__ best) {
	const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= E) return;
	const u32 cu = comp[slot_src[k]], cv = comp[(u32)adj[k]];
	if (cu == cv) return;
	if (best[cu] > (unsigned long long)k) atomicMin(&best[cu], (unsigned long long)k);
	if (best[cv] > (unsigned long long)k) atomicMin(&best[cv], (unsigned long long)k);
}
// roots hook onto the component on the other side of their smallest leaving slot; two roots that chose the same slot
// would point at each other: the larger one hooks, the smaller one stays a root (with distinct weights no longer cycle
// can form).  comp[] is only read here (every entry names a root of the round's start); the new parents go to hook[],
// which every root writes (itself when it stays a root).  Chosen slots are flagged for the host replay.
__global__ void k_wcc_hook(int64_t V, const u32 *__restrict__ slot_src, const int32_t *__restrict__ adj,
                           const u32 *__restrict__ comp, const unsigned long long *__restrict__ best, u32 *__restrict__ hook,
                           uint8_t *__restrict__ msf, u32 *__restrict__ hooks) {
	const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= V || comp[c] != (u32)c) return;
	const unsigned long long k = best[c];
	u32 target = (u32)c;
	if (k != ~0ull) {
		const u32 ra = comp[slot_src[k]], rb = comp[(u32)adj[k]];
		const u32 other = ra == (u32)c ? rb : ra;
		msf[k] = 1;
		if (!(best[other] == k && (u32)c < other)) { // mutual choice: the larger root hooks
			target = other;
			atomicAdd(hooks, 1u);
		}
	}
	hook[c] = target;
}
// pointer jumping over the hooks of the round's roots (a chain graph hooks every vertex onto its predecessor: without
// it the walk below would be quadratic)
__global__ void k_wcc_jump(int64_t V, const u32 *__restrict__ comp, u32 *__restrict__ hook, u32 *__restrict__ changed) {
	const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= V || comp[c] != (u32)c) return;
	const u32 p = __hip_atomic_load(&hook[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const u32 g = __hip_atomic_load(&hook[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (g != p) {
		__hip_atomic_store(&hook[c], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		*changed = 1;
	}
}
// every vertex follows the hooks from the root it had at the round's start to the new root
__global__ void k_wcc_flatten(int64_t V, const u32 *__restrict__ hook, const u32 *__restrict__ comp, u32 *__restrict__ comp_new) {
	const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= V) return;
	u32 x = comp[v];
	for (;;) {
		const u32 p = hook[x];
		if (p == x) break;
		x = p;
	}
	comp_new[v] = x;
}
__global__ void k_wcc_gather_edges(u32 n, const u32 *__restrict__ slots, const u32 *__restrict__ slot_src,
                                   const int32_t *__restrict__ adj, int32_t *__restrict__ out) {
	const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	out[2 * (size_t)i] = (int32_t)slot_src[slots[i]];
	out[2 * (size_t)i + 1] = adj[slots[i]];
}

// component id of every forest entry = the root of its tree (the replayed forest is read-only here)
__global__ void k_wcc_ids(int64_t vs, const int32_t *__restrict__ forest, int64_t *__restrict__ ids) {
	const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= vs) return;
	int32_t x = (int32_t)v;
	for (;;) {
		const int32_t p = forest[x];
		if (p == x) break;
		x = p;
	}
	ids[v] = (int64_t)x;
}

static int wcc_compute(pgq_csr *c, Workspace *ws) {
	std::lock_guard<std::mutex> g(c->lazy_lock); // per handle, like the reference's bind-data lock: unrelated CSRs and devices do not wait
	
	if (c->wcc) return PGQ_OK;
	hipStream_t st = ws->stream;
	const int64_t V = c->V, E = c->E, vs = V + 2;
	DevBuf slot_src, comp, comp2, hook, best, msf, sel, tmp, edges, cnt;
	std::vector<int32_t> forest((size_t)vs, 0); // (4-byte entries: V + 2 < 2^31, and the replay is a walk over this array in cache)
	static const bool wtrace = getenv("PGQ_WCC_TRACE") != nullptr; // where the once-per-handle computation's time goes (stderr)
	auto tnow = []() { return std::chrono::steady_clock::now(); };
	auto t_start = tnow();
	auto lap = [&](const char *what) {
		if (!wtrace) return;
		const auto t = tnow();
		fprintf(stderr, "[pgq] wcc %s: %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_start).count());
		t_start = t;
	};
	auto body = [&]() -> int {
		std::vector<int32_t> h_edges;
		u32 n_msf = 0;
		if (E > 0 && V > 0) {
			PGQ_TRY(slot_src.reserve((size_t)E * 4));
			PGQ_TRY(comp.reserve((size_t)V * 4));
			PGQ_TRY(comp2.reserve((size_t)V * 4));
			PGQ_TRY(hook.reserve((size_t)V * 4));
			PGQ_TRY(best.reserve((size_t)V * 8));
			PGQ_TRY(msf.reserve((size_t)E));
			PGQ_TRY(cnt.reserve(64));
			hipLaunchKernelGGL(k_wcc_slot_src, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, V, c->off, slot_src.as<u32>());
			hipLaunchKernelGGL(k_wcc_init, dim3(blocks_for(V)), dim3(256), 0, st, V, comp.as<u32>());
			PGQ_HIP_TRY(hipMemsetAsync(msf.p, 0, (size_t)E, st));
			u32 *cur = comp.as<u32>(), *nxt = comp2.as<u32>();
			for (int round = 0; round < 64; round++) { // the number of components at least halves per round
				PGQ_HIP_TRY(hipMemsetAsync(best.p, 0xFF, (size_t)V * 8, st));
				PGQ_HIP_TRY(hipMemsetAsync(cnt.p, 0, 4, st));
				hipLaunchKernelGGL(k_wcc_min_edge, dim3(blocks_for(E)), dim3(256), 0, st, E, slot_src.as<u32>(), c->adj, cur,
				                   best.as<unsigned long long>());
				hipLaunchKernelGGL(k_wcc_hook, dim3(blocks_for(V)), dim3(256), 0, st, V, slot_src.as<u32>(), c->adj, cur,
				                   best.as<unsigned long long>(), hook.as<u32>(), msf.as<uint8_t>(), cnt.as<u32>());
				u32 hooks = 0;
				PGQ_HIP_TRY(hipMemcpyAsync(&hooks, cnt.p, 4, hipMemcpyDeviceToHost, st));
				PGQ_HIP_TRY(hipStreamSynchronize(st));
				if (hooks == 0) break;
				for (int j = 0; j < 40; j++) { // halves the hook chains per pass
					u32 changed = 0;
					PGQ_HIP_TRY(hipMemsetAsync(cnt.as<u32>() + 1, 0, 4, st));
					hipLaunchKernelGGL(k_wcc_jump, dim3(blocks_for(V)), dim3(256), 0, st, V, cur, hook.as<u32>(), cnt.as<u32>() + 1);
					PGQ_HIP_TRY(hipMemcpyAsync(&changed, cnt.as<u32>() + 1, 4, hipMemcpyDeviceToHost, st));
					PGQ_HIP_TRY(hipStreamSynchronize(st));
					if (!changed) break;
				}
				hipLaunchKernelGGL(k_wcc_flatten, dim3(blocks_for(V)), dim3(256), 0, st, V, hook.as<u32>(), cur, nxt);
				std::swap(cur, nxt);
			}
			lap("Boruvka rounds");
			// the chosen slots in ascending order (= the order the reference processes them in)
			PGQ_TRY(sel.reserve((size_t)std::min<int64_t>(E, V) * 4 + 64));
			size_t sb = 0;
			hipcub::CountingInputIterator<u32> iota(0u);
			PGQ_HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, sb, iota, msf.as<uint8_t>(), sel.as<u32>(), cnt.as<u32>(), (int)E, st));
			PGQ_TRY(tmp.reserve(sb + 16));
			PGQ_HIP_TRY(hipcub::DeviceSelect::Flagged(tmp.p, sb, iota, msf.as<uint8_t>(), sel.as<u32>(), cnt.as<u32>(), (int)E, st));
			PGQ_HIP_TRY(hipMemcpyAsync(&n_msf, cnt.p, 4, hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
			if ((int64_t)n_msf >= V) return fail(PGQ_ERR_HIP, "internal error: spanning forest with more than V - 1 edges");
			if (n_msf > 0) {
				PGQ_TRY(edges.reserve((size_t)n_msf * 8));
				hipLaunchKernelGGL(k_wcc_gather_edges, dim3(blocks_for(n_msf)), dim3(256), 0, st, n_msf, sel.as<u32>(),
				                   slot_src.as<u32>(), c->adj, edges.as<int32_t>());
				h_edges.resize((size_t)n_msf * 2);
				PGQ_TRY(staged_download(h_edges.data(), edges.p, (size_t)n_msf * 8, st));
			}
		}
		lap("chosen slots to the host");
		// the reference's schedule on the edges that matter (weakly_connected_component.cpp:14-34,77-90); entry V + 1 is
		// left at 0 by the reference's resize and never linked
		int32_t *const f = forest.data();
		for (int64_t i = 0; i < vs - 1; i++) f[i] = (int32_t)i;
		auto root = [&](int32_t x) {
			for (;;) {
				const int32_t p = f[x];
				if (p == x) return x;
				f[x] = f[p];
				x = p;
			}
		};
		for (u32 i = 0; i < n_msf; i++) {
			const int32_t ra = root(h_edges[2 * (size_t)i]), rb = root(h_edges[2 * (size_t)i + 1]);
			if (ra != rb) f[ra] = rb;
		}
		lap("Link replay");
		// the ids (FindTreeRoot of every entry, :94-96) are read off on the device: the forest goes up as it is (1.8 MB instead
		// of 3.6 MB of ids after a host pass over every vertex); v = V + 1: forest entry 0 -> the root of vertex 0
		int64_t *d_ids = nullptr;
		PGQ_TRY(dev_alloc_as(&d_ids, (size_t)vs));
		hipError_t e0 = hook.reserve((size_t)vs * 4) == PGQ_OK ? hipSuccess : hipErrorOutOfMemory;
		hipError_t e1 = e0 == hipSuccess ? hipMemcpyAsync(hook.p, f, (size_t)vs * 4, hipMemcpyHostToDevice, st) : e0;
		if (e1 == hipSuccess) hipLaunchKernelGGL(k_wcc_ids, dim3(blocks_for(vs)), dim3(256), 0, st, vs, hook.as<int32_t>(), d_ids);
		hipError_t e2 = hipStreamSynchronize(st);
		if (e1 != hipSuccess || e2 != hipSuccess) {
			dev_free(d_ids);
			return fail(PGQ_ERR_HIP, "copying the component ids failed");
		}
		c->wcc = d_ids;
		lap("ids + upload");
		return PGQ_OK;
	};
	const int rc = body();
	(void)hipStreamSynchronize(st);
	for (DevBuf *b : { &slot_src, &comp, &comp2, &hook, &best, &msf, &sel, &tmp, &edges, &cnt }) b->release();
	return rc;
}

__global__ void k_wcc_gather(int64_t n, const int64_t *__restrict__ src, int64_t vs, const int64_t *__restrict__ ids,
                             int64_t *__restrict__ out, uint8_t *__restrict__ ok) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int64_t s = src[i];
	const bool valid = s >= 0 && s < vs; // weakly_connected_component.cpp:94-100
	ok[i] = valid ? 1 : 0;
	out[i] = valid ? ids[s] : 0;
}

__global__ void k_pr_gather(int64_t n, const int64_t *__restrict__ src, int64_t vs, const double *__restrict__ rank,
                            double *__restrict__ out, uint8_t *__restrict__ ok) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int64_t s = src[i];
	const bool valid = s >= 0 && s < vs; // pagerank.cpp:99-103
	ok[i] = valid ? 1 : 0;
	out[i] = valid ? rank[s] : 0.0;
}

} // namespace pgq

using namespace pgq;

extern "C" {

int pgq_local_clustering_coefficient_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, float *d_out) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!d_src || !d_out))) return fail(PGQ_ERR_INVALID_ARG, "NULL device array");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	return lcc_device(csr, lease.ws, n, d_src, d_out);
}

int pgq_local_clustering_coefficient(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, float *out, uint64_t *out_valid) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "Constraint Error: CSR not found. Is the graph populated?");
	if (V != csr->V) return fail(PGQ_ERR_INVALID_ARG, "V does not match the uploaded CSR");
	if (n < 0 || (n > 0 && (!out || !out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL output");
	if (n == 0) return PGQ_OK;
	FlatPairs fp;
	PGQ_TRY(flatten_pairs(V, n, src, src, fp, false)); // NULL rows come back as -1; out-of-range ids are rejected
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *ws = lease.ws;
	PGQ_TRY(ws->in_src.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_val.reserve((size_t)n * 4));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, fp.src.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	PGQ_TRY(lcc_device(csr, ws, n, ws->in_src.as<int64_t>(), ws->out_val.as<float>()));
	PGQ_TRY(staged_download(out, ws->out_val.p, (size_t)n * 4, ws->stream));
	mask_fill_valid(out_valid, n);
	for (int64_t i = 0; i < n; i++)
		if (fp.src[i] < 0) mask_set_invalid(out_valid, i); // local_clustering_coefficient.cpp:39-41
	return PGQ_OK;
}

int pgq_pagerank_device(pgq_csr_t *csr, double *d_rank, int *iterations) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "Constraint Error: CSR not found. Is the graph populated?");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	PGQ_TRY(pagerank_compute(csr, lease.ws));
	if (d_rank) PGQ_HIP_TRY(hipMemcpy(d_rank, csr->pagerank, (size_t)(csr->V + 2) * 8, hipMemcpyDeviceToDevice));
	if (iterations) *iterations = csr->pagerank_iterations;
	return PGQ_OK;
}

int pgq_pagerank(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, double *out, uint64_t *out_valid) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "Constraint Error: CSR not found. Is the graph populated?");
	if (V != csr->V) return fail(PGQ_ERR_INVALID_ARG, "V does not match the uploaded CSR");
	if (n < 0 || (n > 0 && (!out || !out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL output");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *ws = lease.ws;
	PGQ_TRY(pagerank_compute(csr, ws));
	if (n == 0) return PGQ_OK;
	// rows: NULL -> NULL; ids outside [0, V + 2) -> NULL (pagerank.cpp:93-104)
	std::vector<int64_t> ids((size_t)n);
	const int64_t *data = static_cast<const int64_t *>(src.data);
	for (int64_t r = 0; r < n; r++) {
		const int64_t p = src.sel ? (int64_t)src.sel[r] : r;
		const bool valid = !src.validity || ((src.validity[p >> 6] >> (p & 63)) & 1ULL);
		ids[(size_t)r] = valid ? data[p] : -1;
	}
	PGQ_TRY(ws->in_src.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_val.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_ok.reserve((size_t)n));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, ids.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	hipLaunchKernelGGL(k_pr_gather, dim3(blocks_for(n)), dim3(256), 0, ws->stream, n, ws->in_src.as<int64_t>(), V + 2,
	                   csr->pagerank, ws->out_val.as<double>(), ws->out_ok.as<uint8_t>());
	std::vector<uint8_t> ok((size_t)n);
	PGQ_HIP_TRY(hipStreamSynchronize(ws->stream));
	PGQ_TRY(staged_download(out, ws->out_val.p, (size_t)n * 8, ws->stream));
	PGQ_TRY(staged_download(ok.data(), ws->out_ok.p, (size_t)n, ws->stream));
	mask_fill_valid(out_valid, n);
	for (int64_t i = 0; i < n; i++)
		if (!ok[(size_t)i]) mask_set_invalid(out_valid, i);
	return PGQ_OK;
}

int pgq_weakly_connected_component_device(pgq_csr_t *csr, int64_t *d_ids) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "Constraint Error: CSR not found. Is the graph populated?");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	PGQ_TRY(wcc_compute(csr, lease.ws));
	if (d_ids) PGQ_HIP_TRY(hipMemcpy(d_ids, csr->wcc, (size_t)(csr->V + 2) * 8, hipMemcpyDeviceToDevice));
	return PGQ_OK;
}

int pgq_weakly_connected_component(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, int64_t *out, uint64_t *out_valid) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "Constraint Error: CSR not found. Is the graph populated?");
	if (V != csr->V) return fail(PGQ_ERR_INVALID_ARG, "V does not match the uploaded CSR");
	if (n < 0 || (n > 0 && (!out || !out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL output");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *ws = lease.ws;
	PGQ_TRY(wcc_compute(csr, ws));
	if (n == 0) return PGQ_OK;
	// rows: NULL -> NULL; ids outside [0, V + 2) -> NULL (weakly_connected_component.cpp:94-100)
	std::vector<int64_t> ids((size_t)n);
	const int64_t *data = static_cast<const int64_t *>(src.data);
	for (int64_t r = 0; r < n; r++) {
		const int64_t p = src.sel ? (int64_t)src.sel[r] : r;
		const bool valid = !src.validity || ((src.validity[p >> 6] >> (p & 63)) & 1ULL);
		ids[(size_t)r] = valid ? data[p] : -1;
	}
	PGQ_TRY(ws->in_src.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_val.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_ok.reserve((size_t)n));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, ids.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	hipLaunchKernelGGL(k_wcc_gather, dim3(blocks_for(n)), dim3(256), 0, ws->stream, n, ws->in_src.as<int64_t>(), V + 2,
	                   csr->wcc, ws->out_val.as<int64_t>(), ws->out_ok.as<uint8_t>());
	std::vector<uint8_t> ok((size_t)n);
	PGQ_HIP_TRY(hipStreamSynchronize(ws->stream));
	PGQ_TRY(staged_download(out, ws->out_val.p, (size_t)n * 8, ws->stream));
	PGQ_TRY(staged_download(ok.data(), ws->out_ok.p, (size_t)n, ws->stream));
	mask_fill_valid(out_valid, n);
	for (int64_t i = 0; i < n; i++)
		if (!ok[(size_t)i]) mask_set_invalid(out_valid, i);
	return PGQ_OK;
}

} // extern "C"
