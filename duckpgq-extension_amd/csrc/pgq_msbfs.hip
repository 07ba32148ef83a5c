// pgq_msbfs.hip — multi-source batched BFS on MI355X: iterativelength and shortestpath.
//
// What it replaces (reference, file:line relative to cwida/duckpgq-extension):
//   * the level kernel `IterativeLength`        src/core/functions/scalar/iterativelength.cpp:12-32
//   * its batch driver IterativeLengthFunction  src/core/functions/scalar/iterativelength.cpp:34-143
//   * the parent-tracking kernel + driver       src/core/functions/scalar/shortest_path.cpp:12-41, :43-207
//
// Design (DESIGN.md §3 has the long form and the measured numbers):
//   * a batch holds L = 64*WD searches ("lanes", WD <= 32); a lane is one *distinct source vertex*, every (src,dst)
//     pair with that source reads its answer from the lane (results are a pure function of (CSR,src,dst),
//     SURVEY.md §8e); pairs that cannot have a path (source without out-edges, destination without in-edges) are
//     answered NULL without a lane.  State per vertex is WD contiguous 64-bit lane-words: seen[V][WD],
//     frontier[V][WD], plus nz[V] = which of the WD words are non-empty.
//   * a level is expanded TOP-DOWN (k_push: one wavefront per 256-out-edge work item, one lane per out-neighbour,
//     scalar loop over the non-empty lane-words, atomicOr into seen/next), BOTTOM-UP DENSE (k_pull: one wavefront
//     per vertex, WD adjacent lanes gather the WD contiguous lane-words of one in-neighbour — 64/WD full
//     8*WD-byte segments per instruction, no atomics, the `next &= ~seen; seen |= next` sweep of
//     iterativelength.cpp:26-30 fused in) or BOTTOM-UP SPARSE (k_compact_frontier + k_pull_sparse: frontier packed
//     into a bit map + dense 32-byte records holding the first three words, recurrence organised by in-edge with
//     an LDS accumulator, long-tail words spread over the wavefront through an LDS queue, bit map and block bases
//     resident in LDS when they fit).  The host picks per level from the frontier's out-degree sum and
//     lane-word density.
//   * vertices whose in-degree exceeds `hub_chunk` are split into slices (k_pull_hub*).
//   * lengths come from destination probes (k_probe: is an in-neighbour of dst in the previous frontier?; k_probe2:
//     two hops, for the last few open pairs) or, for cross-product shaped calls, from the level's frontier at dst —
//     the event of iterativelength.cpp:119-129, "seen[dst] has the bit for the first time" (k_detect, behind the
//     non-empty-word mask).  Stragglers of a wide batch are re-run in a narrow one.
//   * independent batches are searched concurrently by worker threads on separate HIP streams.
//   * round 5: the rows of a call whose sources fit one batch are not sorted (k_pair_rows); top-down levels write into a
//     "sparse pool" of frontier buffers that is cleaned by walking nz, bottom-up levels into a "dense pool" that is never
//     zeroed (k_batch_reset); the per-level choice is ONE function of the counters (decide_level) and a batch's levels
//     are enqueued ahead of the host under the plan of the batch before, k_level_reset checking every level against
//     that function on the device (`done`: all level kernels return at once) — one wait per batch, not per level.
//   * shortestpath keeps every level's frontier bitmap instead of the reference's two 8 KiB/vertex parent
//     arrays (shortest_path.cpp:82-83) and rebuilds each path backwards with the reference's tie-break:
//     parent(x) = smallest frontier vertex of the previous level with an edge to x, edge = first CSR slot of
//     that parent holding x (shortest_path.cpp:21-31) — i.e. the in-edge of x with the smallest forward slot
//     whose source is in the previous frontier.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstring>
#include <memory>
#include <thread>

#include "pgq_search.h"

namespace pgq {

// ---- small device helpers -------------------------------------------------------------------------------------

__device__ __forceinline__ u64 wave_or_slots(u64 x, int wd) {
	for (int o = wd; o < 64; o <<= 1) x |= __shfl_xor(x, o);
	return x;
}

// Appends ceil(deg/chunk) work items for vertex n (lanes with `take`), wave-aggregated.
__device__ __forceinline__ void enqueue_items(bool take, int n, int64_t deg, int64_t chunk, u64 *__restrict__ q,
                                              u32 qcap, u32 *__restrict__ qcount) {
	const int lane = threadIdx.x & 63;
	// vertices without out-edges still get one (empty) item: the clear pass must see every frontier vertex,
	// otherwise stale words would survive in the 2-buffer ring
	u32 c = 0;
	if (take) c = deg > 0 ? (u32)((deg + chunk - 1) / chunk) : 1u;
	if (!__any(c != 0)) return;
	u32 incl = c;
	for (int o = 1; o < 64; o <<= 1) {
		u32 t = __shfl_up(incl, o);
		if (lane >= o) incl += t;
	}
	u32 total = __shfl(incl, 63);
	u32 base = 0;
	if (lane == 63) base = atomicAdd(qcount, total);
	base = __shfl(base, 63);
	u32 p = base + incl - c;
	for (u32 k = 0; k < c; k++)
		if (p + k < qcap) q[p + k] = (u64)(u32)n | ((u64)k << 32);
}

// ---- lane assignment -------------------------------------------------------------------------------------------

// A pair can only have a path if its source has an out-edge and (dst_rule) its destination an in-edge: everything
// else is unreachable without a search (NULL, like the exhausted lanes of iterativelength.cpp:133-139) and takes no
// lane — on directed R-MAT inputs that removes ~3/4 of the lanes.
__device__ __forceinline__ bool pair_needs_search(int64_t s, int64_t d, const int64_t *__restrict__ off,
                                                  const int64_t *__restrict__ roff, int dst_rule) {
	return off[s + 1] > off[s] && (!dst_rule || roff[d + 1] > roff[d]);
}

// sm.h_go != null: workgroup 0 also takes the pre-pass's sample of the distinct sources (a call the route memo sent
// straight to the lane batches keeps asking whether its rows still look like a cross product; the verdict lands in pinned
// memory and is read after the lane assignment's wait).  It rides in this launch: the other workgroups take as long anyway.
__global__ __launch_bounds__(256) void k_mark_sources(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                      u32 *__restrict__ flag, int64_t V, const int64_t *__restrict__ off,
                                                      const int64_t *__restrict__ roff, int dst_rule, int *__restrict__ bad,
                                                      SampleArgs sm) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		int64_t s = src[i], d = dst[i];
		if (s >= 0) { // not a NULL row
			if (s >= V || d < 0 || d >= V) *bad = 1;
			else if (s != d && pair_needs_search(s, d, off, roff, dst_rule)) flag[s] = 1;
		}
	}
	__shared__ u32 s_set[kSampleSlots];
	if (sm.h_go && blockIdx.x == 0) sample_distinct_sources(n, src, V, sm.meet_bytes, sm.edge_bytes, sm.out, sm.h_go, s_set);
}

// h_out (pinned host memory, device-addressable): {distinct sources, range-check flag} — what the host waits for next, written
// by the kernel itself instead of two copy commands behind it
__global__ void k_compact_sources(int64_t V, const u32 *__restrict__ flag, const u32 *__restrict__ rank,
                                  int32_t *__restrict__ usrc, const int *__restrict__ bad, u32 *__restrict__ h_out) {
	int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v < V && flag[v]) usrc[rank[v]] = (int32_t)v;
	if (v == 0) {
		h_out[0] = rank[V];
		h_out[1] = (u32)*bad;
	}
}
// the flag array and the counter block of the lane assignment in one launch (two memsets were four fill dispatches)
__global__ void k_prep_zero(u32 *__restrict__ flag, int64_t n_flag, u32 *__restrict__ cnt_words, int n_cnt) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_flag) flag[i] = 0;
	if (i < n_cnt) cnt_words[i] = 0;
}

__global__ void k_pair_keys(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                            const u32 *__restrict__ rank, const int64_t *__restrict__ off,
                            const int64_t *__restrict__ roff, int dst_rule, int64_t V, u32 key_trivial, u32 key_nolane,
                            u32 *__restrict__ key, u32 *__restrict__ idx) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int64_t s = src[i], d = dst[i];
	u32 k = key_nolane;
	if (s >= 0 && s < V && d >= 0 && d < V) {
		if (s == d) k = key_trivial;
		else if (pair_needs_search(s, d, off, roff, dst_rule)) k = rank[s];
	}
	key[i] = k;
	idx[i] = (u32)i;
}

// sorted-side copies: sdst, ssrc; sres = -1 (unresolved) / 0 (trivial)
__global__ void k_gather_sorted(int64_t n, const u32 *__restrict__ skey, const u32 *__restrict__ sidx,
                                const int64_t *__restrict__ src, const int64_t *__restrict__ dst, u32 key_trivial,
                                u32 key_nolane, int32_t *__restrict__ ssrc, int32_t *__restrict__ sdst,
                                int32_t *__restrict__ sres) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	u32 p = sidx[i];
	u32 k = skey[i];
	ssrc[i] = k == key_nolane ? -1 : (int32_t)src[p];
	sdst[i] = k == key_nolane ? -1 : (int32_t)dst[p];
	sres[i] = k == key_trivial ? 0 : -1;
}

// A call whose distinct sources fit ONE batch needs no grouping of its rows at all: the rows stay in the caller's order
// (row i = sorted position i), trivial rows are born answered (0) and rows without a lane born closed (-2: NULL like an
// exhausted lane) so that every per-row kernel of the batch [0, n) skips them by their result word.  One pass over the
// inputs instead of keys + a 32-bit radix sort of 2 M pairs (0.19 ms on the 2048 x 1024 cross product) + a gather.
__global__ void k_pair_rows(int64_t n, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                            const u32 *__restrict__ rank, const int64_t *__restrict__ off,
                            const int64_t *__restrict__ roff, int dst_rule, int64_t V, u32 *__restrict__ skey,
                            int32_t *__restrict__ ssrc, int32_t *__restrict__ sdst, int32_t *__restrict__ sres) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int64_t s = src[i], d = dst[i];
	u32 k = kNoLane;
	if (s >= 0 && s < V && d >= 0 && d < V) {
		if (s == d) k = kTrivial;
		else if (pair_needs_search(s, d, off, roff, dst_rule)) k = rank[s];
	}
	skey[i] = k;
	ssrc[i] = k == kNoLane ? -1 : (int32_t)s;
	sdst[i] = k == kNoLane ? -1 : (int32_t)d;
	sres[i] = k == kTrivial ? 0 : (k == kNoLane ? -2 : -1);
}

// bstart[b] = first sorted position with key >= b*L, b = 0..nb; bstart[nb+1] = first trivial key, [nb+2] = first no-lane key
__global__ void k_batch_bounds(const u32 *__restrict__ skey, int64_t n, u32 L, int nb, u32 key_trivial, u32 key_nolane,
                               int64_t *__restrict__ bstart) {
	int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b > nb + 2) return;
	// laned keys are < U <= nb*L < 2^31 <= the two sentinels
	u32 target = b <= nb ? (u32)b * L : (b == nb + 1 ? key_trivial : key_nolane);
	int64_t lo = 0, hi = n;
	while (lo < hi) {
		int64_t mid = (lo + hi) >> 1;
		if (skey[mid] < target) lo = mid + 1;
		else hi = mid;
	}
	bstart[b] = lo;
}

// sidx == nullptr: the rows were never permuted (k_pair_rows)
__global__ void k_scatter_results(int64_t n, const u32 *__restrict__ sidx, const int32_t *__restrict__ sres,
                                  const int64_t *__restrict__ soff, int64_t *__restrict__ out_len,
                                  int64_t *__restrict__ out_off) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int64_t p = sidx ? (int64_t)sidx[i] : i;
	out_len[p] = sres[i] < 0 ? -1 : (int64_t)sres[i]; // -1 open at exhaustion, -2 proven unreachable: both NULL
	if (out_off) out_off[p] = soff[i];
}

// ---- rows sorted by source for the source-centric kernel (search_device: run_sorted_ball) -------------------------------
// NULL and out-of-range sources sort behind every vertex (key V); the gathered rows carry the ORIGINAL ids, so that the
// kernel answers NULL rows with NULL and reports ids outside [0, V) like every other route.
__global__ void k_sort_keys(int64_t n, const int64_t *__restrict__ src, int64_t V, u32 *__restrict__ key, u32 *__restrict__ idx) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int64_t s = src[i];
	key[i] = (s < 0 || s >= V) ? (u32)V : (u32)s;
	idx[i] = (u32)i;
}
__global__ void k_sort_gather(int64_t n, const u32 *__restrict__ sidx, const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                              int64_t *__restrict__ ssrc, int64_t *__restrict__ sdst) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const u32 i = sidx[j];
	ssrc[j] = src[i];
	sdst[j] = dst[i];
}
__global__ void k_sort_scatter(int64_t n, const u32 *__restrict__ sidx, const int64_t *__restrict__ sout, int64_t *__restrict__ out) {
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) out[sidx[j]] = sout[j];
}

// ---- batch init ------------------------------------------------------------------------------------------------

template <int WD>
__global__ void k_init_batch(const int32_t *__restrict__ usrc, int64_t U, int64_t base, const int64_t *__restrict__ off,
                             u64 *__restrict__ front0, u32 *__restrict__ nz0, u64 *__restrict__ seen,
                             u64 *__restrict__ active, u64 *__restrict__ q, u32 qcap, int64_t chunk,
                             u32 rows, Counters *__restrict__ cnt) {
	if (blockIdx.x == 0 && threadIdx.x == 0) cnt->unresolved = rows; // every row of the batch is open (the device-side end test reads it)
	int64_t g = base + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const bool ok = g < U && g < base + 64 * WD;
	int v = 0;
	int64_t deg = 0;
	if (ok) {
		v = usrc[g];
		int l = (int)(g - base);
		u64 bit = 1ull << (l & 63);
		atomicOr(&front0[(size_t)v * WD + (l >> 6)], bit);
		atomicOr(&nz0[v], 1u << (l >> 6));
		atomicOr(&seen[(size_t)v * WD + (l >> 6)], bit);
		atomicOr(&active[l >> 6], bit);
		deg = off[v + 1] - off[v];
		atomicAdd(&cnt->front_edges, (u64)deg);
		atomicAdd(&cnt->front_vertices, 1u);
		atomicAdd(&cnt->front_words, 1u);
	}
	enqueue_items(ok, v, deg, chunk, q, qcap, &cnt->q_count[0]);
}

// ---- what a level leaves open: the active-lane mask and the count --------------------------------------------------------
// k_detect / k_probe collect the lane bits of the rows they leave open and their number per workgroup in LDS.  Round 4
// merged the words into the global mask with an atomic load + atomicOr per non-empty word and workgroup: 2048 workgroups x
// 32 words on the same four cache lines are served one at a time (~1.2 ns each: 80 of k_detect's 120 us on a 2.1 M-row
// cross product were this tail; with 256 workgroups the kernel took 55 us).  Tried first: a slot per workgroup, a ticket,
// the last workgroup ORs the slots — the agent-scope fence before every ticket is an L2 write-back on gfx950 and made
// both kernels slower.  Now the mask has kOpenRep COPIES: a workgroup ORs its non-empty words into copy blockIdx % kOpenRep
// (atomics that return nothing: nobody waits for them, 8 workgroups per line instead of 512), and whoever opens the
// next use of the mask folds the copies into the mask proper and zeroes them: k_level_reset of the next level after a
// k_detect, workgroup 0 of k_probe2 (or k_open_merge) after a k_probe.  The open-row count stays one atomicAdd per workgroup.
constexpr int kOpenGrid = 512; // workgroups (of 1024 threads: two per CU) k_detect / k_probe run at most
constexpr int kOpenRep = 64;   // copies of the mask
template <int WD>
__device__ __forceinline__ void publish_open_lanes(const u64 *s_act, const u32 *s_open, u64 *__restrict__ rep,
                                                   Counters *__restrict__ cnt) {
	if (threadIdx.x < WD) {
		const u64 m = s_act[threadIdx.x];
		if (m) (void)__hip_atomic_fetch_or(&rep[(size_t)(blockIdx.x % kOpenRep) * WD + threadIdx.x], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (threadIdx.x == 0 && *s_open) atomicAdd(&cnt->unresolved, *s_open);
}
// every thread of one workgroup (any size that is a multiple of 64): mask[w] = OR of the copies, copies zeroed (the kernels
// that wrote them are done).  Thread t takes word t % wd of the copies t / wd, t / wd + blockDim / wd, ...: all its loads
// are in flight together, the slices meet in LDS.  Returns nothing; ends with a barrier (the mask is written).
__device__ __forceinline__ void fold_open_lanes(u64 *__restrict__ rep, u64 *__restrict__ mask, int wd) {
	__shared__ u64 s_fold[32];
	if (threadIdx.x < 32) s_fold[threadIdx.x] = 0;
	__syncthreads();
	const int w = (int)threadIdx.x % wd, first = (int)threadIdx.x / wd, step = max(1, (int)blockDim.x / wd);
	u64 acc = 0;
	for (int r = first; r < kOpenRep; r += step) {
		acc |= rep[(size_t)r * wd + w];
		rep[(size_t)r * wd + w] = 0;
	}
	if (acc) atomicOr(&s_fold[w], acc);
	__syncthreads();
	if ((int)threadIdx.x < wd) mask[threadIdx.x] = s_fold[threadIdx.x];
	__syncthreads();
}
__global__ void k_open_merge(u64 *__restrict__ rep, u64 *__restrict__ mask, int wd, const Counters *__restrict__ cnt) {
	if (cnt->done) return;
	fold_open_lanes(rep, mask, wd);
}

// one launch at the start of every level instead of several tiny memsets: per-level counters, the next
// active-lane mask, and the frontier-queue counts a top-down level is about to fill.
// Levels enqueued ahead (sp.log != null): the host has not seen the previous level's counters, so this kernel
// (1) writes them into the pinned log, (2) ends the batch when the host loop would (nothing open, empty frontier, or the
// previous level's probe left <= its stop limit open), (3) checks that the level about to run is the one the counters call
// for (decide_level, the host loop's own rule) — else the enqueued levels are called off (`done`: every level kernel
// returns at once) and the host takes over from the logged counters.
struct SpecArgs {
	LevelLog *log;  // pinned; entry t - 1 = the counters after level t - 1
	u32 *status;    // pinned: {done code, level at which it was set}
	int t;          // the level this launch opens
	u32 planned;    // its enqueued kernels (kLv* bits); kLvNone: nothing is enqueued behind this launch, only log
	int prev_stop;  // stop limit of the previous level's probe (-1: none)
};
__global__ void k_level_reset(Counters *__restrict__ cnt, int act_zero, int zero_q0, int zero_q1, LevelRule rule, SpecArgs sp,
                              u64 *__restrict__ fold_rep) {
	const int t = threadIdx.x;
	__shared__ int s_off, s_nzw;
	if (cnt->done) return;
	if (fold_rep) fold_open_lanes(fold_rep, &cnt->act[act_zero ^ 1][0], rule.wd); // the level before ended with k_detect: its open lanes are still in the copies
	if (sp.log) {
		if (t < 64) { // the first wavefront counts the non-empty words of the mask
			const u64 m = __ballot(t < rule.wd && cnt->act[act_zero ^ 1][t < rule.wd ? t : 0] != 0);
			if (t == 0) s_nzw = __popcll(m);
		}
		__syncthreads();
		if (t == 0) {
			int off = 0;
			{
				const int nzw = s_nzw;
				LevelLog lg;
				lg.front_edges = cnt->front_edges;
				lg.edges_scanned = cnt->edges_scanned;
				lg.word_gathers = cnt->word_gathers;
				lg.front_vertices = cnt->front_vertices;
				lg.unresolved = cnt->unresolved;
				lg.front_words = cnt->front_words;
				lg.pad2 = cnt->pad2;
				lg.nzw = (u32)nzw;
				lg.r0 = 0;
				sp.log[sp.t - 1] = lg;
				u32 code = 0;
				if (lg.unresolved == 0 || lg.front_edges == 0) code = 1;
				else if (sp.prev_stop >= 0 && lg.unresolved <= (u32)sp.prev_stop) code = 1;
				else if (sp.planned == kLvNone) code = 3;
				else if (decide_level(rule, lg.front_edges, lg.front_words, lg.front_vertices, lg.unresolved,
				                      sp.t == 1 ? rule.wd : nzw) != sp.planned) code = 2;
				if (code) {
					cnt->done = code;
					sp.status[1] = (u32)sp.t;
					__threadfence_system();
					sp.status[0] = code;
					off = 1;
				}
			}
			s_off = off;
		}
		__syncthreads();
		if (s_off) return; // the counters stay as the last level left them
	}
	if (t == 0) {
		cnt->front_vertices = 0;
		cnt->unresolved = 0;
		cnt->front_edges = 0;
		cnt->edges_scanned = 0;
		cnt->word_gathers = 0;
		cnt->front_words = 0;
		cnt->pad = 0;
		cnt->pad2 = 0;
		if (zero_q0) cnt->q_count[0] = 0;
		if (zero_q1) cnt->q_count[1] = 0;
	}
	if (t < 32) cnt->act[act_zero][t] = 0;
}

// Frontier buffers that are only ever written through atomics (level 0, targets of top-down levels: the "sparse pool")
// keep the invariant "a non-zero lane-word has its bit in nz[v]", so they are zeroed by walking nz — 4 bytes per vertex
// read, only the flagged words written — instead of a memset of V x WD x 8 bytes (115 MB at SF100 / WD = 32) per use.
__device__ __forceinline__ void clean_row_by_nz(u64 *__restrict__ buf, u32 *__restrict__ nz, int64_t v, int wd) {
	u32 m = nz[v];
	if (!m) return;
	nz[v] = 0;
	while (m) {
		const int w = __ffs((int)m) - 1;
		m &= m - 1;
		buf[(size_t)v * wd + w] = 0;
	}
}
// inside a level (the target of a top-down level whose last use left it dirty); cnt != null: an enqueued-ahead level,
// which must leave the buffer alone when it does not run
__global__ void k_clean_by_nz(u64 *__restrict__ buf, u32 *__restrict__ nz, int64_t V, int wd, const Counters *__restrict__ cnt) {
	if (cnt && cnt->done) return;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) clean_row_by_nz(buf, nz, v, wd);
}
// One launch at the start of a batch instead of six fills: `seen` zeroed (streaming 16-byte stores), the counter block and
// the copies of the open-lane mask zeroed, the dirty buffers of the sparse pool cleaned by their nz.
__global__ __launch_bounds__(256) void k_batch_reset(uint4 *__restrict__ seen, size_t seen16, u32 *__restrict__ cnt_words, int n_cnt,
                                                     u64 *__restrict__ rep, int n_rep, u64 *__restrict__ buf_a, u32 *__restrict__ nz_a,
                                                     u64 *__restrict__ buf_b, u32 *__restrict__ nz_b, int64_t V, int wd) {
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = tid; i < seen16; i += stride) seen[i] = make_uint4(0, 0, 0, 0);
	if (blockIdx.x == 0) {
		for (int i = threadIdx.x; i < n_cnt; i += blockDim.x) cnt_words[i] = 0;
		for (int i = threadIdx.x; i < n_rep; i += blockDim.x) rep[i] = 0;
	}
	if (buf_a)
		for (int64_t v = (int64_t)tid; v < V; v += (int64_t)stride) clean_row_by_nz(buf_a, nz_a, v, wd);
	if (buf_b)
		for (int64_t v = (int64_t)tid; v < V; v += (int64_t)stride) clean_row_by_nz(buf_b, nz_b, v, wd);
}

// ---- top-down level ----------------------------------------------------------------------------------------------
// One wavefront per work item (frontier vertex, or a hub_chunk slice of one); lane = out-neighbour.
// For every non-empty lane-word of visit[v]: bits not yet in seen[n] are OR-ed into seen[n] and next[n]
// (iterativelength.cpp:18-25 + :26-30 restricted to the touched vertices; same unseen-masking as
// iterativelength2.cpp:24-26).  First toucher of n queues it for the next level.
template <int WD>
__global__ __launch_bounds__(256) void k_push(const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                              const u64 *__restrict__ visit, u64 *__restrict__ seen,
                                              u64 *__restrict__ next, u32 *__restrict__ nz_next,
                                              const u64 *__restrict__ active, const u64 *__restrict__ qcur, int par,
                                              u32 qcap, int64_t chunk, int stop_limit,
                                              Counters *__restrict__ cnt) {
	// the probe already answered (or the host will defer) what is left of the batch: skip the expansion
	if (level_is_off(cnt, stop_limit)) return;
	const int lane = threadIdx.x & 63;
	const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const u32 nwaves = (gridDim.x * blockDim.x) >> 6;
	const u32 nq = min(cnt->q_count[par], qcap);
	u64 scanned = 0, gathers = 0, mf = 0;
	u32 nf = 0, nw = 0;
	for (u32 i = wave; i < nq; i += nwaves) {
		const u64 item = qcur[i];
		const int v = (int)(u32)item;
		const int64_t b = off[v] + (int64_t)(item >> 32) * chunk;
		const int64_t e = min(off[v + 1], b + chunk);
		u64 val[WD];
#pragma unroll
		for (int w = 0; w < WD; w++) val[w] = visit[(size_t)v * WD + w] & active[w];
		for (int64_t base = b; base < e; base += 64) {
			const int64_t idx = base + lane;
			const bool has = idx < e;
			const int n = has ? adj[idx] : 0;
			bool enq = false;
#pragma unroll
			for (int w = 0; w < WD; w++) {
				if (val[w] == 0) continue; // wave-uniform
				if (has) {
					u64 *sp = &seen[(size_t)n * WD + w];
					u64 nb = val[w] & ~*sp;
					if (nb) {
						u64 old = atomicOr(sp, nb);
						u64 fr = nb & ~old;
						if (fr) {
							// the first lane that makes this lane-word non-empty flags it in nz; the first that flags
							// anything for n queues n (nz of the next level starts zeroed, so it doubles as "queued")
							if (atomicOr(&next[(size_t)n * WD + w], fr) == 0) {
								nw++;
								if (atomicOr(&nz_next[n], 1u << w) == 0) enq = true;
							}
						}
					}
				}
				gathers += (u64)min((int64_t)64, e - base);
			}
			int64_t deg = 0;
			if (enq) {
				deg = off[n + 1] - off[n];
				mf += (u64)deg;
				nf++;
			}
			// no queue append here: one shared counter serialises ~10^4 wave-level appends per launch; if the next
			// level is top-down again its queue is rebuilt from nz (k_queue_from_dense, a V-word scan)
			scanned += (u64)min((int64_t)64, e - base);
		}
	}
	// per-lane nf/mf, wave-uniform scanned/gathers; one atomic set per block
	for (int o = 32; o > 0; o >>= 1) {
		nf += __shfl_down(nf, o);
		nw += __shfl_down(nw, o);
		mf += __shfl_down(mf, o);
	}
	__shared__ u64 red[4][5];
	const int wib = threadIdx.x >> 6;
	if (lane == 0) {
		red[wib][0] = nf;
		red[wib][1] = nw;
		red[wib][2] = mf;
		red[wib][3] = scanned;
		red[wib][4] = gathers;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u64 t[5] = { 0, 0, 0, 0, 0 };
		for (int k = 0; k < (int)(blockDim.x >> 6); k++)
			for (int j = 0; j < 5; j++) t[j] += red[k][j];
		if (t[0]) atomicAdd(&cnt->front_vertices, (u32)t[0]);
		if (t[1]) atomicAdd(&cnt->front_words, (u32)t[1]);
		if (t[2]) atomicAdd(&cnt->front_edges, t[2]);
		if (t[3]) atomicAdd(&cnt->edges_scanned, t[3]);
		if (t[4]) atomicAdd(&cnt->word_gathers, t[4]);
	}
}

// zero the frontier words of the vertices just expanded (keeps the 2-buffer ring sparse-clean)
template <int WD>
__global__ void k_clear_items(const u64 *__restrict__ qcur, int par, u32 qcap, u64 *__restrict__ visit,
                              u32 *__restrict__ nz, const Counters *__restrict__ cnt) {
	if (cnt->done) return; // an enqueued-ahead level that does not run: the frontier it would clear is still needed
	const u32 nq = min(cnt->q_count[par], qcap);
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; t < (int64_t)nq * WD; t += stride) {
		u64 item = qcur[t / WD];
		if ((item >> 32) == 0) {
			visit[(size_t)(u32)item * WD + (t % WD)] = 0;
			if (t % WD == 0) nz[(u32)item] = 0;
		}
	}
}

// frontier queue from a dense frontier (bottom-up level followed by a top-down level)
template <int WD>
__global__ void k_queue_from_dense(const u32 *__restrict__ nz, int64_t V, const int64_t *__restrict__ off,
                                   int64_t chunk, u64 *__restrict__ q, u32 qcap, int par, Counters *__restrict__ cnt) {
	if (cnt->done) return;
	int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	const int64_t vmax = (V + 63) & ~63ll;
	for (int64_t v = v0; v < vmax; v += stride) {
		bool take = false;
		int64_t deg = 0;
		if (v < V) {
			take = nz[v] != 0;
			if (take) deg = off[v + 1] - off[v];
		}
		enqueue_items(take, (int)v, deg, chunk, q, qcap, &cnt->q_count[par]);
	}
}

// ---- bottom-up level ----------------------------------------------------------------------------------------------
// One wavefront per vertex n (persistent grid, interleaved assignment).  Lane = (slot, word): WD adjacent
// lanes read the WD contiguous lane-words of one in-neighbour, 64/WD neighbours per gather instruction.
//   next[n] = (OR over in-neighbours v of visit[v]) & active & ~seen[n];  seen[n] |= next[n]
// which is iterativelength.cpp:18-30 evaluated per destination instead of per source.  A vertex whose
// wanted lanes are already all seen is skipped; scanning stops early once every wanted lane is covered.
template <int WD>
__global__ __launch_bounds__(256) void k_pull(const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                              const int64_t *__restrict__ off, const int32_t *__restrict__ parts,
                                              int n_parts, const u64 *__restrict__ visit,
                                              const u32 *__restrict__ nz_cur, u64 *__restrict__ seen,
                                              u64 *__restrict__ next, u32 *__restrict__ nz_next,
                                              const u64 *__restrict__ active, int V, int64_t hub_threshold,
                                              int stop_limit, Counters *__restrict__ cnt) {
	constexpr int NS = 64 / WD;
	__shared__ u64 red[4][5];
	if (level_is_off(cnt, stop_limit)) return;
	const int lane = threadIdx.x & 63;
	const int word = lane & (WD - 1);
	const int slot = lane / WD;
	const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (gridDim.x * blockDim.x) >> 6;
	const u64 act = active[word];
	u64 nf = 0, mf = 0, scanned = 0, gath = 0, nwords = 0;
	// vertices are pre-cut into small contiguous ranges of bounded in-edge count (built at upload), dealt
	// round-robin to the waves: balanced for skewed degree distributions, consecutive adjacency per wave
	for (int p = wave; p < n_parts; p += nwaves)
	for (int n = parts[2 * p]; n < parts[2 * p + 1]; n++) {
		const int64_t b = roff[n], e = roff[n + 1];
		if (e - b > hub_threshold) continue; // handled by k_pull_hub
		const u64 s = seen[(size_t)n * WD + word];
		const u64 want = act & ~s;
		if (!__any(want != 0) || b == e) {
			if (lane < WD) next[(size_t)n * WD + lane] = 0;
			if (lane == 0) nz_next[n] = 0;
			continue;
		}
		u64 acc = 0;
		for (int64_t base = b; base < e; base += 64) {
			const int c64 = (int)min((int64_t)64, e - base);
			const int nb = lane < c64 ? radj[base + lane] : 0;
			// which lane-words of that in-neighbour are non-empty (4-byte lookup, L2 resident) ...
			const u32 nzb = lane < c64 ? nz_cur[nb] : 0u;
			// gather only words that are non-empty there and still wanted here.  All WD loads of a chunk are issued before
			// the first one is used, and they are unconditional (a load under a per-lane condition is waited for at the end
			// of its branch): lanes with nothing to fetch read word `word` of row 0 — one cached line — and drop it
			constexpr int GRP = WD < 8 ? WD : 8; // loads in flight per lane
#pragma unroll
			for (int r0 = 0; r0 < WD; r0 += GRP) {
				// a short last chunk (in-degree 89: 25 of 64 slots): its empty groups are not issued — every one of them is eight
				// 64-lane loads of a cached line through the texture path (1.50 -> 1.42 ms on the dense level of the SF100 cross
				// product).  Tried on top and dropped: 16 gathers in flight + the next chunk's entries / masks requested ahead
				// (no gain: the level moves ~10 GB of 128-byte lines, 40 M in-edges x a 256-byte row, in 1.4 ms — the fabric's rate)
				if (r0 * NS >= c64) break;
				u64 w[GRP];
				bool hot[GRP];
#pragma unroll
				for (int g = 0; g < GRP; g++) {
					const int j = (r0 + g) * NS + slot;
					const int nbj = __shfl(nb, j);
					const u32 nzj = __shfl(nzb, j);
					hot[g] = want != 0 && ((nzj >> word) & 1u);
					w[g] = visit[hot[g] ? (size_t)nbj * WD + word : (size_t)word];
				}
#pragma unroll
				for (int g = 0; g < GRP; g++) {
					acc |= hot[g] ? w[g] : 0ull;
					gath += hot[g] ? 1u : 0u;
				}
			}
			scanned += (u64)c64;
			if (base + 64 < e) {
				const u64 t = wave_or_slots(acc, WD);
				if (__all((t & want) == want)) {
					acc = t;
					break;
				}
			}
		}
		acc = wave_or_slots(acc, WD);
		const u64 fresh = acc & want;
		if (lane < WD) {
			next[(size_t)n * WD + lane] = fresh;
			if (fresh) seen[(size_t)n * WD + lane] = s | fresh;
		}
		const u64 fm = __ballot(lane < WD && fresh != 0);
		if (lane == 0) nz_next[n] = (u32)fm;
		if (fm) {
			nf += 1;
			nwords += (u64)__popcll(fm);
			mf += (u64)(off[n + 1] - off[n]);
		}
	}
	// per-lane gather counts -> wave total
	for (int o = 32; o > 0; o >>= 1) gath += __shfl_down(gath, o);
	// block reduction of the wave-uniform counters, one atomic set per block
	const int wib = threadIdx.x >> 6;
	if (lane == 0) {
		red[wib][0] = nf;
		red[wib][1] = mf;
		red[wib][2] = scanned;
		red[wib][3] = gath;
		red[wib][4] = nwords;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u64 a = 0, bsum = 0, c = 0, g = 0, wsum = 0;
		for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
			a += red[k][0];
			bsum += red[k][1];
			c += red[k][2];
			g += red[k][3];
			wsum += red[k][4];
		}
		if (wsum) atomicAdd(&cnt->front_words, (u32)wsum);
		if (a) atomicAdd(&cnt->front_vertices, (u32)a);
		if (bsum) atomicAdd(&cnt->front_edges, bsum);
		if (c) atomicAdd(&cnt->edges_scanned, c);
		if (g) atomicAdd(&cnt->word_gathers, g);
	}
}

// ---- bottom-up level, sparse variant ---------------------------------------------------------------------------------
// While the frontier holds few lane-words (early levels, straggler batches) gathering from the dense
// [V][WD] frontier array wastes a 128-byte fabric fetch per 8-byte word (measured: 2.8 GB fetched for 0.19 GB
// of words at SF100 level 2).  The frontier is therefore first packed:
//   bits[v/32]  1 bit per vertex "has any lane-word"      (V/8 bytes: L1/L2 resident)
//   meta[]      dense 32-byte records {non-empty-word mask, offset into cw, first three non-empty words}
//   cw[]        the 4th.. non-empty lane-words back to back (L2 resident)
struct __attribute__((aligned(32))) FrontMeta {
	u32 nz;   // non-empty-word mask
	u32 base; // offset of the 4th.. non-empty words in cw
	u64 w[3]; // first three non-empty words inline: 86 % of the hot in-edges at SF100 level 2 need no second fetch
};
constexpr int kInlineWords = 3;

template <int WD>
__global__ __launch_bounds__(256) void k_compact_frontier(const u32 *__restrict__ nz, const u64 *__restrict__ front,
                                                          int64_t V, FrontMeta *__restrict__ meta,
                                                          u64 *__restrict__ cw, u32 *__restrict__ bits,
                                                          u32 *__restrict__ bbase, u32 *__restrict__ totals, u32 cap,
                                                          int stop_limit, const Counters *__restrict__ cnt) {
	if (level_is_off(cnt, stop_limit)) return;
	const int lane = threadIdx.x & 63;
	// every wavefront owns one contiguous vertex range: pass 1 counts its frontier vertices and packed words, two
	// atomicAdds claim its slices of meta[] / cw[] (a per-64-vertices atomic serialised 65k times on R-MAT-22),
	// pass 2 assigns positions and copies.  meta[] is dense: vertex v's record sits at
	// bbase[v/64] + popcount(bits of the block below v), so the hot set of the gather is (frontier vertices) x 16 B
	const int64_t wave = (int64_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int64_t nwaves = (int64_t)((gridDim.x * blockDim.x) >> 6);
	const int64_t per = ((V + nwaves - 1) / nwaves + 63) & ~63ll;
	const int64_t v0 = wave * per, v1 = min(v0 + per, (V + 63) & ~63ll);
	u32 myw = 0, myv = 0;
	for (int64_t v = v0 + lane; v < v1; v += 64) {
		const u32 m = v < V ? nz[v] : 0u;
		if (m) {
			myw += (u32)max((int)__popc(m) - kInlineWords, 0);
			myv += 1u;
		}
	}
	for (int o = 32; o > 0; o >>= 1) {
		myw += __shfl_xor(myw, o);
		myv += __shfl_xor(myv, o);
	}
	u32 wrun = 0, vrun = 0;
	if (myv) {
		if (lane == 0) {
			vrun = atomicAdd(&totals[0], myv);
			wrun = myw ? atomicAdd(&totals[1], myw) : 0u;
		}
		vrun = __shfl(vrun, 0);
		wrun = __shfl(wrun, 0);
	}
	for (int64_t vb = v0; vb < v1; vb += 64) {
		const int64_t v = vb + lane;
		const u32 m = v < V ? nz[v] : 0u;
		const u64 any = __ballot(m != 0);
		if (lane == 0) {
			bits[vb >> 5] = (u32)any;
			bbase[vb >> 6] = vrun;
		}
		if (lane == 32) bits[(vb >> 5) + 1] = (u32)(any >> 32);
		if (!any) continue;
		const u32 c = (u32)max((int)__popc(m) - kInlineWords, 0); // words beyond the inline ones go to cw
		u32 incl = c;
		for (int o = 1; o < 64; o <<= 1) {
			const u32 t = __shfl_up(incl, o);
			if (lane >= o) incl += t;
		}
		const u32 base = wrun + incl - c;
		wrun += __shfl(incl, 63);
		if (m) {
			const u32 slot = vrun + (u32)__popcll(any & ((1ull << lane) - 1ull));
			FrontMeta rec;
			rec.nz = m;
			rec.base = base;
			u32 rest = m;
#pragma unroll
			for (int r = 0; r < kInlineWords; r++) {
				rec.w[r] = rest ? front[(size_t)v * WD + (__ffs((int)rest) - 1)] : 0ull;
				rest &= rest - 1;
			}
			meta[slot] = rec;
			u32 k = 0;
			while (rest) {
				const int w = __ffs((int)rest) - 1;
				rest &= rest - 1;
				if (base + k < cap) cw[base + k] = front[(size_t)v * WD + w];
				k++;
			}
		}
		vrun += (u32)__popcll(any);
	}
}

// The bottom-up recurrence organised by in-EDGE instead of by vertex: a wavefront owns a part (<= 32 consecutive
// vertices, no hubs) and walks the part's contiguous in-adjacency, one entry per lane, UN x 64 entries in
// flight.  A lane tests its in-neighbour's frontier bit, fetches the neighbour's {mask, offset}, and gathers only
// the packed words that are non-empty there AND still wanted by the entry's owner vertex, OR-ing them into the
// owner's row of an LDS accumulator (ds_or_b64).  seen/next rows of a part are contiguous -> coalesced
// prologue/epilogue.  Same results as k_pull (next = OR of in-neighbours' frontier words & active & ~seen).
template <int WD, int UN, int WPB, int PW>
__global__ __launch_bounds__(WPB * 64) void k_pull_sparse(const int64_t *__restrict__ roff, const int32_t *__restrict__ radj,
                                                          const uint8_t *__restrict__ rown,
                                                     const int64_t *__restrict__ off, const int32_t *__restrict__ parts,
                                                     int n_parts, const u32 *__restrict__ bits,
                                                     const u32 *__restrict__ bbase,
                                                     const FrontMeta *__restrict__ meta, const u64 *__restrict__ cw,
                                                     u64 *__restrict__ seen, u64 *__restrict__ next,
                                                     u32 *__restrict__ nz_next, const u64 *__restrict__ active,
                                                     int lds_bit_words, int spill_after, int stop_limit, Counters *__restrict__ cnt) {
	constexpr int NV = 16; // vertices per part (host-built parts never hold more)
	__shared__ u64 s_acc[WPB][NV * WD];
	__shared__ u32 s_want[WPB][NV];
	__shared__ u32 s_nzn[WPB][NV];
	__shared__ u64 red[WPB][5];
	constexpr int QCAP = 128; // spill queue entries per wavefront
	__shared__ u32 s_queue[WPB][QCAP];
	static_assert(NV * WD <= 512 && UN <= 4, "spill descriptor layout");
	// WPB == 16: the 1-bit frontier map (V/8 bytes) and the per-64-vertex record bases (V/16 bytes) live in LDS
	extern __shared__ u32 s_dyn[];
	u32 *s_bits = s_dyn;
	u32 *s_bbase = s_dyn + lds_bit_words;
	if (level_is_off(cnt, stop_limit)) return;
	if (WPB == 16) {
		for (int i = threadIdx.x; i < lds_bit_words; i += WPB * 64) s_bits[i] = bits[i];
		for (int i = threadIdx.x; i < lds_bit_words / 2; i += WPB * 64) s_bbase[i] = bbase[i];
		__syncthreads();
	}
	const u32 *bt = WPB == 16 ? s_bits : bits;
	const u32 *bb = WPB == 16 ? s_bbase : bbase;
	const int lane = threadIdx.x & 63;
	const int wib = threadIdx.x >> 6;
	u64 *acc = s_acc[wib];
	u32 *wantm = s_want[wib];
	u32 *nzn = s_nzn[wib];
	u32 *queue = s_queue[wib];
	const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (gridDim.x * blockDim.x) >> 6;
	u64 nf = 0, mf = 0, scanned = 0, gath = 0, nwords = 0;
	for (int p = wave; p < n_parts; p += nwaves) {
		const int v0 = parts[2 * p], v1 = parts[2 * p + 1];
		const int nv = v1 - v0;
		const int64_t e0 = roff[v0], e1 = roff[v1];
		// In-adjacency entries of a trip: UN == 4 reads them as one aligned 16-byte group (+ one 4-byte group of owner
		// rows) per lane — entry (lane, k) is slot base + 4*lane + k —, otherwise entry (lane, k) is base + 64*k + lane.
		// The first trip is requested ahead of the prologue's own loads.
		const int64_t eb0 = UN == 4 ? (e0 & ~3ll) : e0;
		int nbn[UN];
		u32 ownn[UN];
		auto fetch_adj = [&](int64_t base) {
			if constexpr (UN == 4) {
				const int64_t e = base + 4 * lane;
				int4 v = make_int4(-1, -1, -1, -1);
				u32 o = 0;
				if (e < e1) { // may run up to 3 entries past e1: both arrays are padded
					v = *reinterpret_cast<const int4 *>(radj + e);
					o = *reinterpret_cast<const u32 *>(rown + e);
				}
				nbn[0] = (e >= e0 && e < e1) ? v.x : -1;
				nbn[1] = (e + 1 >= e0 && e + 1 < e1) ? v.y : -1;
				nbn[2] = (e + 2 >= e0 && e + 2 < e1) ? v.z : -1;
				nbn[3] = (e + 3 >= e0 && e + 3 < e1) ? v.w : -1;
				ownn[0] = o & 255u;
				ownn[1] = (o >> 8) & 255u;
				ownn[2] = (o >> 16) & 255u;
				ownn[3] = o >> 24;
			} else {
#pragma unroll
				for (int k = 0; k < UN; k++) {
					const int64_t e = base + 64 * k + lane;
					nbn[k] = e < e1 ? radj[e] : -1;
					ownn[k] = e < e1 ? (u32)rown[e] : 0u; // owner row inside the part, precomputed at upload
				}
			}
		};
		fetch_adj(eb0);
		// -- prologue: row starts, wanted-word masks, zeroed accumulator
		if (lane < NV) {
			wantm[lane] = 0;
			nzn[lane] = 0;
		}
		__builtin_amdgcn_wave_barrier();
		for (int idx = lane; idx < nv * WD; idx += 64) {
			const int w = idx & (WD - 1);
			const u64 s = seen[(size_t)v0 * WD + idx];
			acc[idx] = 0;
			if (active[w] & ~s) atomicOr(&wantm[idx / WD], 1u << w);
		}
		__builtin_amdgcn_wave_barrier();
		// -- one in-edge per lane
		for (int64_t base = eb0; base < e1; base += 64 * UN) {
			int nb[UN], own[UN];
			bool hot[UN];
#pragma unroll
			for (int k = 0; k < UN; k++) {
				nb[k] = nbn[k];
				own[k] = (int)ownn[k];
			}
			// the next trip's adjacency entries are requested before this trip's dependent fetches are waited for
			if (base + 64 * UN < e1) fetch_adj(base + 64 * UN);
#pragma unroll
			for (int k = 0; k < UN; k++) hot[k] = nb[k] >= 0 && ((bt[nb[k] >> 5] >> (nb[k] & 31)) & 1u);
			// all UN record fetches are issued before any is consumed (independent 16-byte requests in flight)
			FrontMeta mt[UN];
#pragma unroll
			for (int k = 0; k < UN; k++) {
				mt[k].nz = 0;
				if (hot[k]) { // record index = base of the 64-vertex block + frontier vertices below nb in the block
					const int blk = nb[k] >> 6;
					const u64 bw = (u64)bt[2 * blk] | ((u64)bt[2 * blk + 1] << 32);
					mt[k] = meta[bb[blk] + (u32)__popcll(bw & ((1ull << (nb[k] & 63)) - 1ull))];
				}
			}
			// The first three non-empty words of a neighbour travel inside its record: they are OR-ed straight-line, no
			// second fetch (SF100 level 2: 86 % of the hot entries hold <= 3 words).
			u32 m[UN], anym = 0;
#pragma unroll
			for (int k = 0; k < UN; k++) {
				m[k] = mt[k].nz ? (mt[k].nz & wantm[own[k]]) : 0u;
				gath += (u64)__popc(m[k]);
				u32 rest = mt[k].nz;
#pragma unroll
				for (int r = 0; r < kInlineWords; r++) {
					const u32 low = rest & (0u - rest);
					if (m[k] & low) atomicOr(&acc[own[k] * WD + (__ffs((int)low) - 1)], mt[k].w[r]);
					rest ^= low;
				}
				m[k] &= rest; // what is left has word rank >= 3 and lives in cw
				anym |= m[k];
			}
			// Fallback for the words beyond the inline ones: one word of every chunk per trip, all cw fetches of a trip
			// in flight before the first is waited for.
			auto trip = [&]() {
				u64 val[UN * PW];
				int idx[UN * PW];
#pragma unroll
				for (int k = 0; k < UN; k++) {
#pragma unroll
					for (int j = 0; j < PW; j++) {
						idx[k * PW + j] = -1;
						val[k * PW + j] = 0;
						if (m[k]) {
							const int w = __ffs((int)m[k]) - 1;
							m[k] &= m[k] - 1;
							const int r = __popc(mt[k].nz & ((1u << w) - 1u));
							idx[k * PW + j] = own[k] * WD + w;
							val[k * PW + j] = cw[mt[k].base + r - kInlineWords];
						}
					}
				}
				anym = 0;
#pragma unroll
				for (int k = 0; k < UN; k++) {
#pragma unroll
					for (int j = 0; j < PW; j++)
						if (idx[k * PW + j] >= 0) atomicOr(&acc[idx[k * PW + j]], val[k * PW + j]);
					anym |= m[k];
				}
			};
			// The word counts have a long tail (SF100 level 2: 2.5 wanted words per hot entry on average, 10.7 for the
			// fullest of 256), so a per-lane loop over the remaining words leaves most lanes idle behind a few.  They
			// are spread over the whole wavefront instead: every lane writes a 4-byte descriptor per remaining word into
			// a per-wave LDS queue (position = exclusive prefix sum of the counts), then lane j serves descriptor j —
			// one wait for all of them.
			if (__any(anym != 0)) {
				u32 c = 0;
#pragma unroll
				for (int k = 0; k < UN; k++) c += (u32)__popc(m[k]);
				u32 incl = c;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) {
					const u32 t = __shfl_up(incl, o);
					if (lane >= o) incl += t;
				}
				const u32 total = __shfl(incl, 63);
				if (spill_after > 0 && total <= (u32)QCAP) {
					u32 pos = incl - c;
#pragma unroll
					for (int k = 0; k < UN; k++) {
						u32 mm = m[k];
						while (mm) { // descriptor: accumulator index | source lane << 9 | chunk << 15 | word rank << 17
							const int w = __ffs((int)mm) - 1;
							mm &= mm - 1;
							const u32 r = (u32)__popc(mt[k].nz & ((1u << w) - 1u)); // >= kInlineWords
							queue[pos++] = (u32)(own[k] * WD + w) | ((u32)lane << 9) | ((u32)k << 15) | (r << 17);
						}
						m[k] = 0;
					}
					__builtin_amdgcn_wave_barrier();
					for (u32 jb = 0; jb < total; jb += 64) {
						const bool on = jb + lane < total;
						const u32 d = on ? queue[jb + lane] : 0u;
						const int sl = (int)((d >> 9) & 63u), kk = (int)((d >> 15) & 3u);
						u32 cbase = 0;
#pragma unroll
						for (int k = 0; k < UN; k++) {
							const u32 t = (u32)__shfl((int)mt[k].base, sl);
							if (k == kk) cbase = t;
						}
						if (on) atomicOr(&acc[d & 511u], cw[cbase + (d >> 17) - (u32)kInlineWords]);
					}
					__builtin_amdgcn_wave_barrier();
				} else {
					while (__any(anym != 0)) trip();
				}
			}
			scanned += (u64)(min(base + 64 * UN, e1) - max(base, e0));
		}
		__builtin_amdgcn_wave_barrier();
		// -- epilogue: fold into seen/next (coalesced rows), non-empty-word masks, frontier stats
		for (int idx = lane; idx < nv * WD; idx += 64) {
			const int w = idx & (WD - 1);
			const u64 a = acc[idx];
			u64 fresh = 0;
			if (a) { // untouched rows need no second look at seen
				const u64 s = seen[(size_t)v0 * WD + idx];
				fresh = a & active[w] & ~s;
				if (fresh) {
					seen[(size_t)v0 * WD + idx] = s | fresh;
					atomicOr(&nzn[idx / WD], 1u << w);
				}
			}
			next[(size_t)v0 * WD + idx] = fresh;
		}
		__builtin_amdgcn_wave_barrier();
		if (lane < nv) {
			const u32 fm = nzn[lane];
			nz_next[v0 + lane] = fm;
			if (fm) {
				nf += 1;
				nwords += (u64)__popc(fm);
				mf += (u64)(off[v0 + lane + 1] - off[v0 + lane]);
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	for (int o = 32; o > 0; o >>= 1) {
		gath += __shfl_down(gath, o);
		nf += __shfl_down(nf, o);
		mf += __shfl_down(mf, o);
		nwords += __shfl_down(nwords, o);
	}
	if (lane == 0) {
		red[wib][0] = nf;
		red[wib][1] = mf;
		red[wib][2] = scanned;
		red[wib][3] = gath;
		red[wib][4] = nwords;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u64 a = 0, bsum = 0, c = 0, g = 0, wsum = 0;
		for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
			a += red[k][0];
			bsum += red[k][1];
			c += red[k][2];
			g += red[k][3];
			wsum += red[k][4];
		}
		if (wsum) atomicAdd(&cnt->front_words, (u32)wsum);
		if (a) atomicAdd(&cnt->front_vertices, (u32)a);
		if (bsum) atomicAdd(&cnt->front_edges, bsum);
		if (c) atomicAdd(&cnt->edges_scanned, c);
		if (g) atomicAdd(&cnt->word_gathers, g);
	}
}

// high in-degree vertices: zero next, OR partial results per slice, then fold into seen
template <int WD>
__global__ void k_pull_hub_zero(const int32_t *__restrict__ hubs, int64_t nh, u64 *__restrict__ next,
                                int stop_limit, const Counters *__restrict__ cnt) {
	if (level_is_off(cnt, stop_limit)) return;
	int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nh * WD) next[(size_t)hubs[t / WD] * WD + (t % WD)] = 0;
}

template <int WD>
__global__ __launch_bounds__(256) void k_pull_hub(const HubItem *__restrict__ items, int64_t n_items,
                                                  const int32_t *__restrict__ radj, const u64 *__restrict__ visit,
                                                  const u32 *__restrict__ nz_cur, const u64 *__restrict__ seen,
                                                  u64 *__restrict__ next, const u64 *__restrict__ active,
                                                  int stop_limit, Counters *__restrict__ cnt) {
	constexpr int NS = 64 / WD;
	if (level_is_off(cnt, stop_limit)) return;
	const int lane = threadIdx.x & 63;
	const int word = lane & (WD - 1);
	const int slot = lane / WD;
	const int64_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	__shared__ u64 red[4][2];
	u64 acc = 0, scanned = 0, gath = 0;
	const bool live = wave < n_items;
	const HubItem it = live ? items[wave] : HubItem{ 0, 0, 0, 0 };
	const int n = it.vertex;
	const u64 want = live ? (active[word] & ~seen[(size_t)n * WD + word]) : 0ull;
	const u64 have = live ? next[(size_t)n * WD + word] : 0ull; // other slices may already have covered it
	const bool skip = !live || __all(((have & want) == want));
	for (int64_t base = it.begin; !skip && base < it.end; base += 64) {
		const int c64 = (int)min((int64_t)64, it.end - base);
		const int nb = lane < c64 ? radj[base + lane] : 0;
		const u32 nzb = lane < c64 ? nz_cur[nb] : 0u;
		constexpr int GRP = WD < 8 ? WD : 8; // unconditional loads, GRP in flight per lane (see k_pull)
#pragma unroll
		for (int r0 = 0; r0 < WD; r0 += GRP) {
			u64 w[GRP];
			bool hot[GRP];
#pragma unroll
			for (int g = 0; g < GRP; g++) {
				const int j = (r0 + g) * NS + slot;
				const int nbj = __shfl(nb, j);
				const u32 nzj = __shfl(nzb, j);
				hot[g] = want != 0 && ((nzj >> word) & 1u);
				w[g] = visit[hot[g] ? (size_t)nbj * WD + word : (size_t)word];
			}
#pragma unroll
			for (int g = 0; g < GRP; g++) {
				acc |= hot[g] ? w[g] : 0ull;
				gath += hot[g] ? 1u : 0u;
			}
		}
		scanned += (u64)c64;
		if (base + 64 < it.end) {
			const u64 t = wave_or_slots(acc, WD);
			if (__all((t & want) == want)) {
				acc = t;
				break;
			}
		}
	}
	acc = wave_or_slots(acc, WD);
	const u64 fresh = acc & want;
	if (!skip && lane < WD && fresh) atomicOr(&next[(size_t)n * WD + lane], fresh);
	// one atomic pair per block, not per slice: ~25k slices on R-MAT-22 serialised on two counters otherwise
	for (int o = 32; o > 0; o >>= 1) gath += __shfl_down(gath, o);
	const int wib = threadIdx.x >> 6;
	if (lane == 0) {
		red[wib][0] = scanned;
		red[wib][1] = gath;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u64 a0 = 0, a1 = 0;
		for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
			a0 += red[k][0];
			a1 += red[k][1];
		}
		if (a0) atomicAdd(&cnt->edges_scanned, a0);
		if (a1) atomicAdd(&cnt->word_gathers, a1);
	}
}

template <int WD>
__global__ void k_pull_hub_fold(const int32_t *__restrict__ hubs, int64_t nh, const int64_t *__restrict__ off,
                                u64 *__restrict__ seen, const u64 *__restrict__ next, u32 *__restrict__ nz_next,
                                int stop_limit, Counters *__restrict__ cnt) {
	if (level_is_off(cnt, stop_limit)) return;
	int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= nh) return;
	const int n = hubs[h];
	u32 any = 0;
#pragma unroll
	for (int w = 0; w < WD; w++) {
		u64 f = next[(size_t)n * WD + w];
		if (f) {
			seen[(size_t)n * WD + w] |= f;
			any |= 1u << w;
		}
	}
	nz_next[n] = any;
	if (any) {
		atomicAdd(&cnt->front_words, (u32)__popc(any));
		atomicAdd(&cnt->front_vertices, 1u);
		atomicAdd(&cnt->front_edges, (u64)(off[n + 1] - off[n]));
	}
}

// ---- per-pair detection (iterativelength.cpp:119-129) ------------------------------------------------------------
// A row (lane l, destination d) is answered by the level whose FRONTIER holds bit l at d: seen[d] gets a bit at the level
// that puts it into next[d] (next = reached & ~seen), so "seen[dst] has the bit for the first time" (:123) and "the
// frontier of this level has the bit" are the same event, and every (vertex, lane) is in exactly one frontier.  Testing
// the frontier instead of `seen` lets the 4-byte non-empty-word mask nz[d] (a 1.8-MB table on the SF100 graph: L2
// resident) filter the rows first: after a top-down level almost no row goes on to the 8-byte gather out of the
// [V][WD] array.  UN rows per thread, every load of a stage issued for all of them before the first is used (round 4: one
// row per trip of a grid-stride loop, three dependent round trips each: 0.13 ms per 2.1 M rows, a third of what the box
// gathers).  The loads are unconditional (closed rows read entry 0): a load under a per-lane condition is waited for inside
// its branch.
// Grid-stride over the batch's rows.  A cross product has thousands of rows per lane: one atomic per open row on the
// same few words of the active-lane mask cost 3 ns each (14 M rows of 32 sources: 80 ms per level), and even a read of
// the mask per row through L2 is a hot spot on one channel (2 ms per level).  So the open rows' lane bits are collected
// in a workgroup-local mask in LDS and published once per workgroup (publish_open_lanes).
template <int WD, int UN>
__global__ __launch_bounds__(1024) void k_detect(int64_t lo, int64_t hi, const u32 *__restrict__ skey, const int32_t *__restrict__ sdst,
                                                 int32_t *__restrict__ sres, u32 base_lane, const u64 *__restrict__ front,
                                                 const u32 *__restrict__ nz, int level, u32 dense_words, u64 *__restrict__ rep,
                                                 Counters *__restrict__ cnt) {
	__shared__ u32 s_open;
	__shared__ u64 s_act[WD];
	if (level_is_off(cnt, -1)) return;
	// a frontier with more than `dense_words` non-empty words (half of all): the mask look-up would let nearly every row
	// through and only add a dependent round trip (level 3 of the SF100 cross product: 92 % of the rows are answered)
	const bool filter = cnt->front_words <= dense_words;
	if (threadIdx.x == 0) s_open = 0;
	if (threadIdx.x < WD) s_act[threadIdx.x] = 0;
	__syncthreads();
	u32 n_open = 0;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i0 = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < hi; i0 += stride * UN) {
		int64_t at[UN];
		bool open[UN];
		u32 l[UN];
		int d[UN];
		u32 m[UN];
		u64 w[UN];
#pragma unroll
		for (int u = 0; u < UN; u++) {
			const int64_t i = i0 + u * stride;
			at[u] = i < hi ? i : lo;
			open[u] = sres[at[u]] == -1 && i < hi;
		}
#pragma unroll
		for (int u = 0; u < UN; u++) {
			l[u] = skey[at[u]] - base_lane;
			d[u] = sdst[at[u]];
			if (!open[u]) l[u] = 0, d[u] = 0;
		}
#pragma unroll
		for (int u = 0; u < UN; u++) m[u] = filter ? nz[d[u]] : 0xFFFFFFFFu;
#pragma unroll
		for (int u = 0; u < UN; u++) {
			const int wq = (int)(l[u] >> 6);
			const bool hot = open[u] && ((m[u] >> wq) & 1u);
			w[u] = front[hot ? (size_t)d[u] * WD + wq : (size_t)0];
			if (!hot) w[u] = 0;
		}
#pragma unroll
		for (int u = 0; u < UN; u++) {
			if (!open[u]) continue;
			const u64 bit = 1ull << (l[u] & 63);
			if (w[u] & bit) {
				sres[at[u]] = level;
			} else {
				n_open++;
				if (!(s_act[l[u] >> 6] & bit)) atomicOr(&s_act[l[u] >> 6], bit); // a cross product's rows share few lanes: nearly always set already
			}
		}
	}
	for (int o = 32; o > 0; o >>= 1) n_open += __shfl_xor(n_open, o);
	if ((threadIdx.x & 63) == 0 && n_open) atomicAdd(&s_open, n_open);
	__syncthreads();
	publish_open_lanes<WD>(s_act, &s_open, rep, cnt);
}


// ---- destination probe ----------------------------------------------------------------------------------------------
// Evaluates the bottom-up step only where it matters for the answer: a still-open pair (lane l, dst d) has hop
// count `level` iff some in-neighbour of d carries lane l in the frontier of level-1.  One wavefront per pair.
// Run before a level is expanded, it answers pairs one full expansion earlier than iterativelength.cpp:119-129
// does (same values: the frontier of level-1 is exactly the set at distance level-1).
template <int WD>
__global__ __launch_bounds__(1024) void k_probe(int64_t lo, int64_t hi, const u32 *__restrict__ skey,
                                                const int32_t *__restrict__ sdst, int32_t *__restrict__ sres,
                                                u32 base_lane, const u64 *__restrict__ front,
                                                const u32 *__restrict__ nz, const int64_t *__restrict__ roff,
                                                const int32_t *__restrict__ radj, int level, u64 *__restrict__ rep,
                                                Counters *__restrict__ cnt, int max_in) {
	// Persistent wavefronts over 64-row chunks: a chunk's result words are read coalesced, and only the rows still open
	// (a ballot) get the wave-wide in-list scan — late levels of a cross product have millions of answered rows and a
	// few thousand open ones (a wavefront per ROW cost 2.4 ms of launches for 2 M rows).  Active-lane bits and the
	// open count are collected per workgroup like in k_detect.
	constexpr int kQueue = 1024;
	__shared__ u32 s_open;
	__shared__ u64 s_act[WD];
	__shared__ u32 q_row[kQueue], q_lane[kQueue];
	__shared__ int q_dst[kQueue];
	__shared__ u32 q_n, q_next, q_cut;
	constexpr int kHubQueue = 256;
	__shared__ u32 h_row[kHubQueue], h_lane[kHubQueue], h_n, h_found;
	__shared__ int h_dst[kHubQueue];
	if (cnt->done) return;
	if (threadIdx.x == 0) s_open = 0, q_n = 0, q_next = 0, q_cut = (u32)kQueue, h_n = 0;
	if (threadIdx.x < WD) s_act[threadIdx.x] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int64_t n = hi - lo;
	const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
	// few rows: G wavefronts share a chunk (each takes every G-th row of it), so that a wavefront scans the in-lists of a
	// handful of rows, not of 64
	const int64_t nchunks = (n + 63) >> 6;
	const int G = (int)max((int64_t)1, min((int64_t)64, nwaves / max(nchunks, (int64_t)1)));
	const int g = (int)(gw % G);
	u32 n_open = 0; // lane 0 counts
	// one open row: does an in-neighbour of its destination carry its lane in the frontier?
	auto probe_row = [&](int64_t row, u32 l, int d) {
		const int w = (int)(l >> 6);
		const u64 bit = 1ull << (l & 63);
		const int64_t b = roff[d], e = roff[d + 1];
		if (b == e) { // nothing points at dst: unreachable, no search needed (reported as NULL like :133-139)
			if (lane == 0) sres[row] = -2;
			return;
		}
		// a hub destination's in-list is not one wavefront's serial scan (64 entries per dependent round trip; R-MAT-22, 2048 x
		// 1024 rows: a few hundred destinations with 10^4..10^5 in-neighbours made every probe launch 1.4 ms, 12.9 of the call's
		// 23.8 ms): it is queued, and the workgroup's 16 wavefronts scan it together at the end
		if (e - b > (int64_t)max_in) {
			u32 p = kHubQueue;
			if (lane == 0) p = atomicAdd(&h_n, 1u);
			p = (u32)__builtin_amdgcn_readfirstlane((int)p);
			if (p < (u32)kHubQueue) {
				if (lane == 0) {
					h_row[p] = (u32)(row - lo);
					h_lane[p] = l;
					h_dst[p] = d;
				}
				return;
			} // (a full queue: scanned here after all)
		}
		bool found = false;
		for (int64_t base = b; base < e && !found; base += 64) {
			const int64_t j = base + lane;
			bool hit = false;
			if (j < e) {
				const int v = radj[j];
				if ((nz[v] >> w) & 1u) hit = (front[(size_t)v * WD + w] & bit) != 0;
			}
			found = __any(hit);
		}
		if (lane == 0) {
			if (found) {
				sres[row] = level;
			} else {
				n_open++;
				if (!(s_act[w] & bit)) atomicOr(&s_act[w], bit);
			}
		}
	};
	// More chunks than wavefronts (a cross product: 32,768 chunks holding 0..6 open rows each after level 3): with every
	// wavefront probing the open rows of ITS chunks, a wavefront's share was Poisson-distributed (mean 6, the longest of
	// 8192 wavefronts ~16) and the kernel as long as that one.  Now the 16 wavefronts of a workgroup first collect the open
	// rows of all their chunks in an LDS queue and then draw rows from it one by one: the spread is a workgroup's (~96 +-
	// 10 rows), not a wavefront's.  (Tried first: chunks handed out by one global counter — 16 K returning atomics on one
	// address made the kernel three times slower.)  A full queue: the chunk's rows are probed where they are found.
	const bool pooled = G == 1;
	for (int64_t chunk = gw / G; chunk * 64 < n; chunk += nwaves / G) {
		const int64_t i = lo + chunk * 64 + lane;
		const bool mine_open = i < hi && sres[i] == -1;
		u32 my_l = 0;
		int my_d = 0;
		if (mine_open) {
			my_l = skey[i] - base_lane;
			my_d = sdst[i];
		}
		u64 todo = __ballot(mine_open);
		if (pooled && todo) {
			const u32 c = (u32)__popcll(todo);
			u32 base = 0;
			if (lane == 0) base = atomicAdd(&q_n, c);
			base = (u32)__builtin_amdgcn_readfirstlane((int)base);
			if (base + c <= (u32)kQueue) {
				if (mine_open) {
					const u32 p = base + (u32)__popcll(todo & ((1ull << lane) - 1ull));
					q_row[p] = (u32)(i - lo);
					q_lane[p] = my_l;
					q_dst[p] = my_d;
				}
				continue;
			}
			// does not fit: its slots stay unwritten, so the queue ends where this (or an earlier such) chunk's slots begin —
			// every later chunk overflows too (the count only grows)
			if (lane == 0) atomicMin(&q_cut, base);
		}
		while (todo) {
			const int k = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			if (k % G != g) continue; // by row index: the split must not depend on what the chunk's other wavefronts have answered
			probe_row(lo + chunk * 64 + k, (u32)__builtin_amdgcn_readlane((int)my_l, k), __builtin_amdgcn_readlane(my_d, k));
		}
	}
	if (pooled) {
		__syncthreads();
		const u32 filled = min(q_n, q_cut);
		for (;;) {
			u32 p = 0;
			if (lane == 0) p = atomicAdd(&q_next, 1u);
			p = (u32)__builtin_amdgcn_readfirstlane((int)p);
			if (p >= filled) break;
			probe_row(lo + (int64_t)q_row[p], q_lane[p], q_dst[p]);
		}
	}
	// the hub destinations: one row after the other, 1024 in-neighbours per step
	__syncthreads();
	{
		const u32 hn = min(h_n, (u32)kHubQueue);
		for (u32 h = 0; h < hn; h++) {
			if (threadIdx.x == 0) h_found = 0;
			__syncthreads();
			const u32 l = h_lane[h];
			const int d = h_dst[h], w = (int)(l >> 6);
			const u64 bit = 1ull << (l & 63);
			const int64_t b = roff[d], e = roff[d + 1];
			for (int64_t base = b; base < e; base += 4 * 1024) {
				bool hit = false;
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int64_t j = base + u * 1024 + threadIdx.x;
					if (j < e) {
						const int v = radj[j];
						if ((nz[v] >> w) & 1u) hit |= (front[(size_t)v * WD + w] & bit) != 0;
					}
				}
				if (__any(hit) && lane == 0) h_found = 1;
				if (*(volatile u32 *)&h_found) break; // (a wavefront that has not seen the flag yet runs one more step: harmless)
			}
			__syncthreads();
			if (threadIdx.x == 0) {
				if (h_found) {
					sres[lo + (int64_t)h_row[h]] = level;
				} else {
					n_open++;
					if (!(s_act[w] & bit)) atomicOr(&s_act[w], bit);
				}
			}
			__syncthreads();
		}
	}
	if (lane == 0 && n_open) atomicAdd(&s_open, n_open);
	__syncthreads();
	publish_open_lanes<WD>(s_act, &s_open, rep, cnt);
}

// Two-hop destination probe: a pair still open after k_probe(level) has hop count level+1 iff some in-neighbour u of
// dst has an in-neighbour carrying the lane in the frontier of level-1.  One 1024-thread workgroup per open pair:
// it first sums the in-degrees of N_in(dst); if the walk N_in(dst) x N_in(u) would exceed `work_cap` in-edges (hub
// destinations) the pair is left open for the regular expansion / deferral, otherwise the 16 wavefronts split the
// u's and stop as soon as one finds a hit.  Runs only when few pairs are left (it replaces a full-width
// expansion that would serve only them).  Persistent workgroups: each scans 64-row chunks of the batch for the rows
// still open (a workgroup per ROW meant millions of empty launches on a cross product).
template <int WD>
__global__ __launch_bounds__(1024) void k_probe2(int64_t lo, int64_t hi, const u32 *__restrict__ skey,
                                                 const int32_t *__restrict__ sdst, int32_t *__restrict__ sres,
                                                 u32 base_lane, const u64 *__restrict__ front,
                                                 const u32 *__restrict__ nz, const int64_t *__restrict__ roff,
                                                 const int32_t *__restrict__ radj, int level, u32 run_below,
                                                 int64_t work_cap, u64 *__restrict__ rep, u64 *__restrict__ active_next,
                                                 Counters *__restrict__ cnt) {
	__shared__ unsigned long long s_work;
	__shared__ int s_found;
	__shared__ u32 s_list[64], s_n, s_answered;
	if (cnt->done) return;
	// the mask of the lanes k_probe left open, for the expansion launched behind this kernel (it stays a superset of
	// the lanes still open after the answers below)
	if (blockIdx.x == 0) fold_open_lanes(rep, active_next, WD);
	const u32 open_now = cnt->unresolved;
	if (open_now == 0 || open_now > run_below) return;
	const int lane = threadIdx.x & 63;
	const int wib = threadIdx.x >> 6, nw = blockDim.x >> 6;
	if (threadIdx.x == 0) s_answered = 0;
	// few rows: B workgroups share a 64-row chunk (each takes every B-th open row of it)
	const int64_t nchunks = (hi - lo + 63) >> 6;
	const int B = (int)max((int64_t)1, min((int64_t)64, (int64_t)gridDim.x / max(nchunks, (int64_t)1)));
	const int bsub = (int)(blockIdx.x % B);
	for (int64_t c0 = lo + (int64_t)(blockIdx.x / B) * 64; c0 < hi; c0 += (int64_t)(gridDim.x / B) * 64) {
		__syncthreads();
		if (threadIdx.x == 0) s_n = 0;
		__syncthreads();
		if (threadIdx.x < 64 && (int)(threadIdx.x % B) == bsub) { // this workgroup's share of the chunk: by row index, so that
			const int64_t i = c0 + threadIdx.x;                  // the split does not depend on what the others have answered
			if (i < hi && sres[i] == -1) s_list[atomicAdd(&s_n, 1u)] = (u32)(i - c0);
		}
		__syncthreads();
		const u32 cnt_open = s_n;
		for (u32 q = 0; q < cnt_open; q++) {
			__syncthreads(); // the previous pair's flags are no longer read
			const int64_t i = c0 + s_list[q];
			if (threadIdx.x == 0) {
				s_work = 0;
				s_found = 0;
			}
			__syncthreads();
			const u32 l = skey[i] - base_lane;
			const int w = (int)(l >> 6);
			const u64 bit = 1ull << (l & 63);
			const int d = sdst[i];
			const int64_t b = roff[d], e = roff[d + 1];
			unsigned long long mine = 0;
			for (int64_t k = b + threadIdx.x; k < e; k += blockDim.x) {
				const int u = radj[k];
				mine += (unsigned long long)(roff[u + 1] - roff[u]);
			}
			for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
			if (lane == 0 && mine) atomicAdd(&s_work, mine);
			__syncthreads();
			if ((int64_t)s_work > work_cap) continue; // too expensive here: the pair stays open (uniform: s_work is shared)
			for (int64_t k = b + wib; k < e; k += nw) {
				if (*(volatile int *)&s_found) break;
				const int u = radj[k]; // wave-uniform
				const int64_t ub = roff[u], ue = roff[u + 1];
				bool found = false;
				for (int64_t base = ub; base < ue && !found; base += 64) {
					const int64_t j = base + lane;
					bool hit = false;
					if (j < ue) {
						const int v = radj[j];
						if ((nz[v] >> w) & 1u) hit = (front[(size_t)v * WD + w] & bit) != 0;
					}
					found = __any(hit);
				}
				if (found) {
					if (lane == 0) s_found = 1;
					break;
				}
			}
			__syncthreads();
			if (threadIdx.x == 0 && s_found) {
				sres[i] = level + 1;
				s_answered++;
			}
		}
	}
	__syncthreads();
	if (threadIdx.x == 0 && s_answered) atomicSub(&cnt->unresolved, s_answered);
}

// ---- straggler deferral ------------------------------------------------------------------------------------------------
// When only a handful of pairs of a wide batch are still open, expanding another level for all 64*WD lanes is
// wasted bandwidth: the open pairs are marked (-3), collected after the batch loop and searched again in a
// narrow batch of their own (results are a pure function of (CSR, src, dst), so re-running them is exact).
__global__ void k_mark_deferred(int64_t lo, int64_t hi, int32_t *__restrict__ sres) {
	int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < hi && sres[i] == -1) sres[i] = -3;
}
__global__ void k_collect_deferred(int64_t n, const int32_t *__restrict__ sres, const int32_t *__restrict__ ssrc,
                                   const int32_t *__restrict__ sdst, int64_t *__restrict__ dsrc,
                                   int64_t *__restrict__ ddst, u32 *__restrict__ didx, u32 *__restrict__ count) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || sres[i] != -3) return;
	u32 p = atomicAdd(count, 1u);
	dsrc[p] = ssrc[i];
	ddst[p] = sdst[i];
	didx[p] = (u32)i;
}
__global__ void k_apply_deferred(int64_t nd, const u32 *__restrict__ didx, const int64_t *__restrict__ dlen,
                                 int32_t *__restrict__ sres, const int64_t *__restrict__ doff,
                                 int64_t *__restrict__ soff, int64_t child_base) {
	int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nd) return;
	sres[didx[j]] = dlen[j] < 0 ? -1 : (int32_t)dlen[j];
	if (soff && dlen[j] >= 0) soff[didx[j]] = child_base + doff[j]; // straggler paths were appended at child_base
}

// ---- traversed-edge accounting (measurement only; bench.py's MTEPS numerator) -----------------------------------
// S[l] += out-degree of every vertex whose frontier has lane l set.  LDS-privatised per block.
template <int WD>
__global__ __launch_bounds__(256) void k_lane_degree_sums(const u64 *__restrict__ front, const int64_t *__restrict__ off,
                                                          int64_t V, u64 *__restrict__ S) {
	__shared__ u64 acc[64 * WD];
	for (int i = threadIdx.x; i < 64 * WD; i += blockDim.x) acc[i] = 0;
	__syncthreads();
	int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; idx < V * WD; idx += stride) {
		u64 word = front[idx];
		if (!word) continue;
		const int64_t v = idx / WD;
		const int w = (int)(idx % WD);
		const u64 deg = (u64)(off[v + 1] - off[v]);
		while (word) {
			const int b = __ffsll((long long)word) - 1;
			atomicAdd(&acc[w * 64 + b], deg);
			word &= word - 1;
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < 64 * WD; i += blockDim.x)
		if (acc[i]) atomicAdd(&S[i], acc[i]);
}

// te[i] = sum over the levels this pair's own BFS expands (0..d-1, or all when unreachable)
__global__ void k_pair_te(int64_t lo, int64_t hi, const u32 *__restrict__ skey, const int32_t *__restrict__ sres,
                          u32 base_lane, const u64 *__restrict__ S, int L, int nlevels, int64_t *__restrict__ ste) {
	int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= hi) return;
	const u32 l = skey[i] - base_lane;
	const int d = sres[i];
	const int T = d < 0 ? nlevels : min(d, nlevels);
	u64 te = 0;
	for (int t = 0; t < T; t++) te += S[(size_t)t * L + l];
	ste[i] = (int64_t)te;
}

__global__ void k_scatter_te(int64_t n, const u32 *__restrict__ sidx, const int64_t *__restrict__ ste,
                             int64_t *__restrict__ out) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[sidx[i]] = ste[i];
}

// ---- path reconstruction (shortest_path.cpp:149-204 with the :21-31 tie-break) ---------------------------------
// One wavefront per resolved pair.  Walks back from dst: the in-edges of x are ordered by (source, forward slot), so
// the first in-slot whose source has the lane's bit in the previous level's frontier names the smallest such source;
// its first out-slot that points at x is the edge.
template <int WD>
__global__ __launch_bounds__(256) void k_reconstruct(int64_t lo, int64_t hi, const u32 *__restrict__ skey,
                                                     const int32_t *__restrict__ sdst,
                                                     const int32_t *__restrict__ sres,
                                                     const int64_t *__restrict__ soff, u32 base_lane,
                                                     const u64 *const *__restrict__ levels,
                                                     const int64_t *__restrict__ roff,
                                                     const int32_t *__restrict__ radj,
                                                     const int64_t *__restrict__ off,
                                                     const int32_t *__restrict__ adj,
                                                     const int64_t *__restrict__ edge_ids,
                                                     int64_t *__restrict__ child) {
	const int lane = threadIdx.x & 63;
	const int64_t i = lo + (int64_t)__builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	if (i >= hi) return;
	const int k = sres[i];
	if (k < 1) return;
	const u32 l = skey[i] - base_lane;
	const int w = (int)(l >> 6);
	const u64 bit = 1ull << (l & 63);
	int64_t *out = child + soff[i];
	int x = sdst[i];
	if (lane == 0) out[2 * k] = x;
	for (int t = k - 1; t >= 0; t--) {
		const u64 *F = levels[t];
		int pv = -1;
		const int64_t rb = roff[x], re = roff[x + 1];
		for (int64_t j0 = rb; j0 < re && pv < 0; j0 += 64) {
			const int64_t j = j0 + lane;
			const int v = j < re ? radj[j] : -1;
			const u64 hit = __ballot(v >= 0 && (F[(size_t)v * WD + w] & bit) != 0);
			if (hit) pv = __shfl(v, __ffsll((long long)hit) - 1);
		}
		int64_t slot = -1;
		const int64_t fb = off[pv], fe = off[pv + 1];
		for (int64_t e0 = fb; e0 < fe && slot < 0; e0 += 64) {
			const int64_t e = e0 + lane;
			const u64 hit = __ballot(e < fe && adj[e] == x);
			if (hit) slot = e0 + __ffsll((long long)hit) - 1;
		}
		if (lane == 0) {
			out[2 * t + 1] = edge_ids ? edge_ids[slot] : slot;
			out[2 * t] = pv;
		}
		x = pv;
	}
}

__global__ void k_trivial_paths(int64_t lo, int64_t hi, const int32_t *__restrict__ ssrc, int64_t base_off,
                                int64_t *__restrict__ soff, int64_t *__restrict__ child) {
	int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= hi) return;
	soff[i] = base_off + (i - lo);
	child[base_off + (i - lo)] = ssrc[i];
}

// ---- workspace ---------------------------------------------------------------------------------------------------

Workspace::~Workspace() {
	if (ev_block) (void)hipEventDestroy(ev_block);
	if (stream) (void)hipStreamDestroy(stream);
	if (h_cnt) (void)hipHostFree(h_cnt);
	if (h_bi) (void)hipHostFree(h_bi);
	if (h_log) (void)hipHostFree(h_log);
	if (h_meet) (void)hipHostFree(h_meet);
	if (h_io) (void)hipHostFree(h_io);
	if (h_bstart) (void)hipHostFree(h_bstart);
	for (DevBuf *b : { &seen, &qbuf[0], &qbuf[1], &qflag, &counters, &flag, &rank, &usrc, &key, &idx, &skey,
	                   &sidx, &ssrc, &sdst, &sres, &soff, &sort_tmp, &scan_tmp, &bstart, &levels_tab, &child, &in_src,
	                   &in_dst, &out_len, &out_off, &dist, &dirty[0], &dirty[1], &touched, &tflag, &out_val, &out_ok, &lane_sums, &ste, &def_src, &def_dst, &def_len,
	                   &def_idx, &def_off, &def_ent, &cbits, &cbbase, &cmeta, &cwords, &lblk, &lrec, &meet_cnt, &meet_rec, &meet_poff, &meet_maps, &meet_trace,
	                   &wb_scratch, &hv, &hmask, &hstart, &hmap, &route_dec, &ball_segs, &ball_trace, &sort_src, &sort_dst, &sort_out, &dist_b, &dirty_b[0], &dirty_b[1], &qbuf_b[0], &qbuf_b[1],
	                   &touched_b, &tflag_b, &bi_block, &dpart })
		b->release();
	for (auto *v : { &levels, &pool })
		for (auto &l : *v) {
			l->buf.release();
			l->nz.release();
		}
}

static std::mutex g_ws_lock;
static std::vector<Workspace *> g_ws_free; // every workspace remembers the device its buffers live on

void drop_idle_workspaces() { // of the calling thread's device: another device's pool does not help an allocation here
	std::vector<Workspace *> drop;
	{
		std::lock_guard<std::mutex> g(g_ws_lock);
		const int dev = current_device();
		for (size_t k = g_ws_free.size(); k-- > 0;)
			if (g_ws_free[k]->device == dev) {
				drop.push_back(g_ws_free[k]);
				g_ws_free.erase(g_ws_free.begin() + (long)k);
			}
	}
	for (Workspace *w : drop) delete w;
}

int WorkspaceLease::acquire() {
	{
		std::lock_guard<std::mutex> g(g_ws_lock);
		const int dev = current_device();
		for (size_t k = g_ws_free.size(); k-- > 0;)
			if (g_ws_free[k]->device == dev) {
				ws = g_ws_free[k];
				g_ws_free.erase(g_ws_free.begin() + (long)k);
				break;
			}
	}
	if (!ws) {
		ws = new Workspace();
		ws->device = current_device();
		// a half-built workspace never reaches the pool
		if (hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking) != hipSuccess ||
		    hipHostMalloc((void **)&ws->h_cnt, sizeof(Counters)) != hipSuccess ||
		    hipHostMalloc((void **)&ws->h_log, sizeof(LevelLog) * (kSpecLevels + 3)) != hipSuccess ||
		    hipHostMalloc(&ws->h_meet, 8192) != hipSuccess) {
			delete ws;
			ws = nullptr;
			return fail(PGQ_ERR_HIP, "cannot create a search workspace (stream / pinned counter block)");
		}
	}
	return PGQ_OK;
}
WorkspaceLease::~WorkspaceLease() {
	if (!ws) return;
	std::lock_guard<std::mutex> g(g_ws_lock);
	if (g_ws_free.size() < 8 * std::max<size_t>(1, enabled_devices().size())) g_ws_free.push_back(ws);
	else delete ws;
}

// every wait of the lane-batched search on its stream is counted (pgq_stats_t::host_waits)
#define PGQ_WAIT(stream)                                                                                               \
	do {                                                                                                               \
		PGQ_TRY(wait_stream(stream, thread_wait_event()));                                                          \
		tstats().s.host_waits++;                                                                                       \
	} while (0)

// ---- lane assignment (host side) ------------------------------------------------------------------------------------
// Stage 1: flag the distinct sources of the rows that need a search, rank them (= global lane ids), list them; the number
// of distinct sources and the range check come back with ONE wait.
static int lane_ranks(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, u32 *U_out,
                      bool dst_rule, const SampleArgs &sm = SampleArgs { 0.0, 0.0, nullptr, nullptr, 0, nullptr, 0 },
                      const std::function<int()> &pre_wait = nullptr) {
	hipStream_t st = ws->stream;
	const int64_t V = c->V;
	PGQ_TRY(ws->flag.reserve((size_t)(V + 1) * 4));
	PGQ_TRY(ws->rank.reserve((size_t)(V + 1) * 4));
	PGQ_TRY(ws->usrc.reserve((size_t)std::max<int64_t>(V, 1) * 4));
	for (DevBuf *b : { &ws->skey, &ws->ssrc, &ws->sdst, &ws->sres }) PGQ_TRY(b->reserve((size_t)n * 4));
	PGQ_TRY(ws->soff.reserve((size_t)n * 8));
	PGQ_TRY(ws->counters.reserve(sizeof(Counters)));
	int *d_bad = reinterpret_cast<int *>(ws->counters.p); // reused before the batch loop resets it
	KernelTimer kt(st, K_PREP);
	hipLaunchKernelGGL(k_prep_zero, dim3(blocks_for(V + 1)), dim3(256), 0, st, ws->flag.as<u32>(), V + 1, ws->counters.as<u32>(), (int)(sizeof(Counters) / 4));
	hipLaunchKernelGGL(k_mark_sources, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, d_dst, ws->flag.as<u32>(), V, c->off, c->roff, dst_rule ? 1 : 0, d_bad, sm);
	size_t tmp = 0;
	PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, ws->flag.as<u32>(), ws->rank.as<u32>(), (int)(V + 1), st));
	PGQ_TRY(ws->scan_tmp.reserve(tmp + 16));
	PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws->scan_tmp.p, tmp, ws->flag.as<u32>(), ws->rank.as<u32>(), (int)(V + 1), st));
	// the count and the range-check flag land in two words of the pinned counter block
	u32 *h2 = reinterpret_cast<u32 *>(ws->h_cnt);
	h2[0] = h2[1] = 0;
	hipLaunchKernelGGL(k_compact_sources, dim3(blocks_for(std::max<int64_t>(V, 1))), dim3(256), 0, st, V, ws->flag.as<u32>(), ws->rank.as<u32>(), ws->usrc.as<int32_t>(),
	                   d_bad, h2);
	kt.stop();
	if (pre_wait) PGQ_TRY(pre_wait()); // work the GPU can do while the host waits for the count (search_device: stage 2 ahead)
	// lane assignment, stage 1: the rows' ids read once (16 B per row), a flag written per row; flags zeroed, scanned
	// (read + rank written) and compacted: 20 B per vertex
	tstats().s.algo_bytes[K_PREP] += (double)n * 20.0 + (double)(V + 1) * 20.0;
	PGQ_WAIT(st);
	KernelTimer::flush();
	if (h2[1]) return fail(PGQ_ERR_INVALID_ARG, "src/dst rowid out of range [0,V)");
	*U_out = h2[0];
	return PGQ_OK;
}

static int reserve_bstart(Workspace *ws, int nb) {
	PGQ_TRY(ws->bstart.reserve((size_t)(nb + 3) * 8));
	if (ws->h_bstart_cap < (size_t)(nb + 3)) {
		if (ws->h_bstart) (void)hipHostFree(ws->h_bstart);
		ws->h_bstart = nullptr;
		ws->h_bstart_cap = (size_t)(nb + 3) * 2;
		PGQ_HIP_TRY(hipHostMalloc((void **)&ws->h_bstart, ws->h_bstart_cap * 8));
	}
	return PGQ_OK;
}

// Stage 2, rows grouped by lane: keys -> stable radix sort over `bits` key bits -> sorted copies.  The two sentinels
// (trivial rows, rows without a lane) sort behind every lane.
static int lane_rows_sorted(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, bool dst_rule,
                            u32 key_trivial, u32 key_nolane, int bits) {
	hipStream_t st = ws->stream;
	for (DevBuf *b : { &ws->key, &ws->idx, &ws->sidx }) PGQ_TRY(b->reserve((size_t)n * 4));
	KernelTimer kt(st, K_PREP);
	hipLaunchKernelGGL(k_pair_keys, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, d_dst, ws->rank.as<u32>(), c->off, c->roff, dst_rule ? 1 : 0, c->V,
	                   key_trivial, key_nolane, ws->key.as<u32>(), ws->idx.as<u32>());
	size_t stmp = 0;
	PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, stmp, ws->key.as<u32>(), ws->skey.as<u32>(), ws->idx.as<u32>(), ws->sidx.as<u32>(), (int)n, 0, bits, st));
	PGQ_TRY(ws->sort_tmp.reserve(stmp + 16));
	PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ws->sort_tmp.p, stmp, ws->key.as<u32>(), ws->skey.as<u32>(), ws->idx.as<u32>(), ws->sidx.as<u32>(), (int)n, 0, bits, st));
	hipLaunchKernelGGL(k_gather_sorted, dim3(blocks_for(n)), dim3(256), 0, st, n, ws->skey.as<u32>(), ws->sidx.as<u32>(), d_src, d_dst, key_trivial, key_nolane,
	                   ws->ssrc.as<int32_t>(), ws->sdst.as<int32_t>(), ws->sres.as<int32_t>());
	kt.stop();
	// stage 2, rows sorted: keys (28 B read, 8 written), a radix sort of 8-byte pairs (one histogram pass + a read and a
	// write per 8 key bits), the gather of the sorted copies (8 + 16 B read, 12 written)
	tstats().s.algo_bytes[K_PREP] += (double)n * (36.0 + 8.0 + 16.0 * ((bits + 7) / 8) + 36.0);
	return PGQ_OK;
}

// The cheapest-path driver's entry: lanes ranked, rows sorted by lane under the classic sentinels (its own batch width
// goes to batch_bounds afterwards).
int prepare_lanes(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, u32 *U_out,
                  bool dst_rule) {
	PGQ_TRY(lane_ranks(c, ws, n, d_src, d_dst, U_out, dst_rule));
	return lane_rows_sorted(c, ws, n, d_src, d_dst, dst_rule, kTrivial, kNoLane, 32);
}

int batch_bounds(Workspace *ws, int64_t n, int64_t L, int nb) {
	hipStream_t st = ws->stream;
	PGQ_TRY(reserve_bstart(ws, nb));
	{
		KernelTimer kt(st, K_PREP);
		hipLaunchKernelGGL(k_batch_bounds, dim3(blocks_for(nb + 3)), dim3(256), 0, st, ws->skey.as<u32>(), n, (u32)L,
		                   nb, kTrivial, kNoLane, ws->bstart.as<int64_t>());
		kt.stop();
	}
	PGQ_HIP_TRY(hipMemcpyAsync(ws->h_bstart, ws->bstart.p, (size_t)(nb + 3) * 8, hipMemcpyDeviceToHost, st));
	PGQ_WAIT(st);
	return PGQ_OK;
}

// The BFS driver's stage 2.  One batch (the distinct sources fit 64 x wd lanes) and nothing downstream that needs the
// trivial rows as a range (paths): the rows are NOT permuted (k_pair_rows; *identity = true, sidx is not written) and
// nothing is waited for.  Otherwise: sorted by lane over the bits the keys really have (the sentinels sit right behind
// the last batch: nb x L and nb x L + 1), bounds computed and fetched (one wait).
static int lane_rows_bfs(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, bool dst_rule,
                         int64_t L, int nb, bool may_skip_sort, bool *identity, bool rows_done_ahead = false) {
	hipStream_t st = ws->stream;
	PGQ_TRY(reserve_bstart(ws, nb));
	*identity = may_skip_sort && nb <= 1;
	if (*identity && rows_done_ahead) { // k_pair_rows ran in front of the lane assignment's wait (search_device)
		ws->h_bstart[0] = 0;
		for (int b = 1; b <= nb + 2; b++) ws->h_bstart[b] = n;
		return PGQ_OK;
	}
	if (*identity) {
		KernelTimer kt(st, K_PREP);
		hipLaunchKernelGGL(k_pair_rows, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, d_dst, ws->rank.as<u32>(), c->off, c->roff, dst_rule ? 1 : 0, c->V,
		                   ws->skey.as<u32>(), ws->ssrc.as<int32_t>(), ws->sdst.as<int32_t>(), ws->sres.as<int32_t>());
		kt.stop();
		// stage 2, rows in place: ids read again + the source's rank and degree (16 + 12 B), four 4-byte arrays written
		tstats().s.algo_bytes[K_PREP] += (double)n * 44.0;
		ws->h_bstart[0] = 0;
		for (int b = 1; b <= nb + 2; b++) ws->h_bstart[b] = n; // batch 0 = every row; no trivial / NULL ranges
		return PGQ_OK;
	}
	const u32 key_trivial = (u32)((int64_t)nb * L), key_nolane = key_trivial + 1; // nb x L <= U + L < 2^31 + 2^11
	int bits = 1;
	while (bits < 32 && (1ull << bits) <= (u64)key_nolane) bits++;
	PGQ_TRY(lane_rows_sorted(c, ws, n, d_src, d_dst, dst_rule, key_trivial, key_nolane, bits));
	{
		KernelTimer kt(st, K_PREP);
		hipLaunchKernelGGL(k_batch_bounds, dim3(blocks_for(nb + 3)), dim3(256), 0, st, ws->skey.as<u32>(), n, (u32)L, nb,
		                   key_trivial, key_nolane, ws->bstart.as<int64_t>());
		kt.stop();
	}
	PGQ_HIP_TRY(hipMemcpyAsync(ws->h_bstart, ws->bstart.p, (size_t)(nb + 3) * 8, hipMemcpyDeviceToHost, st));
	PGQ_WAIT(st);
	KernelTimer::flush();
	return PGQ_OK;
}

static int choose_words(int64_t unique_sources) {
	const Options &o = options();
	int wd = o.words;
	if (wd <= 0) {
		wd = 1;
		while (wd < o.max_words && (int64_t)wd * 64 < unique_sources) wd <<= 1;
	}
	if (wd != 1 && wd != 2 && wd != 4 && wd != 8 && wd != 16 && wd != 32)
		wd = wd > 32 ? 32 : (wd > 16 ? 16 : (wd > 8 ? 8 : (wd > 4 ? 4 : (wd > 2 ? 2 : 1))));
	return wd;
}

// A bigger buffer with the first `keep` bytes of the old one; the new block is given back if the copy fails (round 4
// returned through PGQ_HIP_TRY with it still allocated).
static int grow_keeping(DevBuf &buf, size_t bytes, size_t keep, hipStream_t st) {
	DevBuf bigger;
	PGQ_TRY(bigger.reserve(bytes));
	hipError_t e = hipSuccess;
	if (keep > 0) e = hipMemcpyAsync(bigger.p, buf.p, keep, hipMemcpyDeviceToDevice, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) {
		bigger.release();
		return fail(PGQ_ERR_HIP, std::string("growing a list buffer: ") + hipGetErrorString(e));
	}
	tstats().s.host_waits++;
	buf.release();
	buf = bigger;
	return PGQ_OK;
}

// The start of a lane batch that keeps no per-level frontiers: `seen`, the counter block and the open-lane copies zeroed,
// the sparse pool's buffers clean (k_batch_reset; a buffer that is new, was laid out for another (V, WD), or belongs to a
// batch that did not end normally has no nz to be trusted and is zeroed whole, once).  Called by run_batches at a batch's
// start — or EARLIER, by search_device, in front of the lane assignment's wait when the CSR remembers the width of the
// last one-batch call (`prereset_*` says so: the GPU then does this while the host waits for the source count).
static int batch_state_reset(Workspace *ws, int64_t V, int WD) {
	hipStream_t st = ws->stream;
	const size_t words = (size_t)std::max<int64_t>(V, 1) * (size_t)WD, nz_bytes = (size_t)std::max<int64_t>(V, 1) * 4;
	PGQ_TRY(ws->seen.reserve(words * 8));
	PGQ_TRY(ws->counters.reserve(sizeof(Counters)));
	PGQ_TRY(ws->dpart.reserve((size_t)kOpenRep * WD * 8));
	while (ws->pool.size() < 4) ws->pool.emplace_back(new LevelBuf());
	u64 *cb[2] = { nullptr, nullptr };
	u32 *cz[2] = { nullptr, nullptr };
	for (int k = 0; k < 2; k++) {
		LevelBuf *lb = ws->pool[(size_t)k].get();
		PGQ_TRY(lb->buf.reserve(words * 8));
		PGQ_TRY(lb->nz.reserve(nz_bytes));
		if (lb->init_buf != lb->buf.p || lb->init_nz != lb->nz.p || lb->lay_V != V || lb->lay_WD != WD || !ws->pool_trusted) {
			PGQ_HIP_TRY(hipMemsetAsync(lb->buf.p, 0, words * 8, st));
			PGQ_HIP_TRY(hipMemsetAsync(lb->nz.p, 0, nz_bytes, st));
			lb->init_buf = lb->buf.p;
			lb->init_nz = lb->nz.p;
			lb->lay_V = V;
			lb->lay_WD = WD;
			lb->dirty = false;
		}
		if (lb->dirty) {
			cb[k] = lb->buf.as<u64>();
			cz[k] = lb->nz.as<u32>();
			lb->dirty = false;
		}
	}
	ws->pool_trusted = false; // until the batch has ended normally
	KernelTimer kt(st, K_PREP);
	hipLaunchKernelGGL(k_batch_reset, dim3(8 * (unsigned)device_cus()), dim3(256), 0, st, ws->seen.as<uint4>(), (words * 8 + 15) / 16,
	                   ws->counters.as<u32>(), (int)(sizeof(Counters) / 4), ws->dpart.as<u64>(), kOpenRep * WD, cb[0], cz[0], cb[1],
	                   cz[1], V, WD);
	kt.stop();
	// `seen` zeroed (8 WD B per vertex) + the nz words of the sparse-pool buffers that are cleaned (4 B per vertex each)
	tstats().s.algo_bytes[K_PREP] += (double)words * 8.0 + (double)V * 4.0 * ((cb[0] != nullptr) + (cb[1] != nullptr));
	return PGQ_OK;
}

// ---- the batch driver ----------------------------------------------------------------------------------------------
// Runs the searches for rows already resident in device memory (d_src/d_dst, -1 src = NULL row).
// with_paths: also emit [src,e,v,...,dst] lists into ws->child and per-row offsets.
struct SearchOutput {
	int64_t child_used = 0;
	bool want_te = false; // fill ws->ste with per-row traversed-edge counts
	int depth = 0;        // nesting level of the straggler pass
	bool deferred = false;
	bool overflow = false; // the caller's child buffer was too small (lengths are still complete)
	bool bidir = false;    // iterativelengthbidirectional: every row through the per-row bidirectional search first
	bool from_meet = false; // these rows are what the pair-centric pre-pass left open: do not run it on them again
	bool no_ball = false;   // these rows are what the source-centric kernel left open: the pre-pass may take them, that kernel not again
	int ball_hint = -1;     // the caller has looked at the rows (chunk entry points: they sit in host memory): 0 = not grouped by source
	// chunk entry points: the rows sit in a staging buffer whose address is the same for every chunk, so what the route memo
	// remembers about "these buffers" says nothing about THESE rows (round-5 advisor finding: unrelated chunks hit the memo,
	// and every change of shape was routed one call late) — such calls neither read nor write it
	bool no_memo = false;
	bool prefer_lanes = false; // (in) large grouped call on a graph where the lane batches measured faster than the source-centric route, or their trial
	int route = 0;             // (out) 1: the source-centric kernel took the call (as it lay, or sorted by source)
	double source_runs = -1;   // (out) ... and counted this many source runs
};
static constexpr int kMaxTeLevels = 1024;

template <int WD>
static int run_batches(pgq_csr *c, Workspace *sh, Workspace *ws, int b0, int bstride, int64_t n, int64_t U,
                       bool with_paths, int64_t *d_child_ext, int64_t child_cap_ext, SearchOutput &outp) {
	// sh: the call's shared arrays (rows sorted by lane, per-row results, batch bounds); ws: this worker's private
	// search state (seen, frontiers, queues, counters, stream).  Workers take batches b0, b0+bstride, ...
	hipStream_t st = ws->stream;
	const int64_t V = c->V, E = c->E;
	const Options &opt = options();
	const int64_t L = 64 * WD;
	const int nb = (int)((U + L - 1) / L);
	const int64_t chunk = c->hub_threshold;                       // bottom-up: in-degree above this = hub
	const int64_t pchunk = std::max(64, options().push_chunk);    // top-down: out-edges per queue item
	const size_t words = (size_t)std::max<int64_t>(V, 1) * WD;
	const u32 qcap = (u32)std::min<int64_t>(V + E / pchunk + 128, 0xFFFFFF00ll);
	pgq_stats_t &S = tstats().s;

	PGQ_TRY(ws->seen.reserve(words * 8));
	PGQ_TRY(ws->qbuf[0].reserve((size_t)qcap * 8));
	PGQ_TRY(ws->qbuf[1].reserve((size_t)qcap * 8));
	PGQ_TRY(ws->counters.reserve(sizeof(Counters)));
	PGQ_TRY(ws->dpart.reserve((size_t)kOpenRep * WD * 8));
	if (outp.want_te) {
		PGQ_TRY(ws->lane_sums.reserve((size_t)kMaxTeLevels * L * 8));
		PGQ_TRY(sh->ste.reserve((size_t)n * 8));
		PGQ_HIP_TRY(hipMemsetAsync(sh->ste.p, 0, (size_t)n * 8, st));
	}
	Counters *d_cnt = ws->counters.as<Counters>();
	if (ws != sh) PGQ_WAIT(st); // (the caller's own stream orders its batches behind the lane assignment)
	const int64_t *bs = sh->h_bstart;
	// levels enqueued ahead of the host under the plan of the last batch of this width; paths keep every
	// level's frontier and the accounting pass its per-level sums: they stay on the round trip per level
	const bool spec_ok = opt.spec_levels && !with_paths && !outp.want_te;
	const int plan_slot = WD == 1 ? 0 : (WD == 2 ? 1 : (WD == 4 ? 2 : (WD == 8 ? 3 : (WD == 16 ? 4 : 5))));

	int64_t child_base = 0;
	int64_t *d_child = d_child_ext;
	bool child_overflow = false;
	std::vector<int32_t> h_res;
	std::vector<int64_t> h_off;

	const int ncu = device_cus();
	const unsigned pull_grid = (unsigned)std::max(1, opt.blocks_per_cu) * ncu;
	const unsigned push_grid = 8 * ncu;

	for (int b = b0; b < nb; b += bstride) {
		const int64_t lo = bs[b], hi = bs[b + 1];
		if (lo == hi) continue;
		const u32 base_lane = (u32)((int64_t)b * L);
		S.batches++;
		// -- reset per-batch state
		const size_t nz_bytes = (size_t)std::max<int64_t>(V, 1) * 4;
		// Frontier buffers.  shortestpath keeps every level's frontier (ws->levels[t]).  Otherwise two pools (ws->pool):
		// [0], [1] "sparse" — level 0 and the targets of top-down levels, written through atomics only, kept all-zero between
		// uses by walking their nz (k_batch_reset / k_clean_by_nz / k_clear_items); [2], [3] "dense" — targets of bottom-up
		// levels, which write every row, so they are never zeroed.  A level's target is the buffer of its pool that is not
		// the current frontier.  (Round 4: two buffers alternating, the top-down targets zeroed by a 115-MB memset per use.)
		auto level_buf = [&](int t, bool push_target, const LevelBuf *not_this) -> LevelBuf * {
			if (with_paths) {
				while (ws->levels.size() <= (size_t)t) ws->levels.emplace_back(new LevelBuf());
				return ws->levels[(size_t)t].get();
			}
			while (ws->pool.size() < 4) ws->pool.emplace_back(new LevelBuf());
			LevelBuf *a = ws->pool[push_target ? 0 : 2].get();
			return a != not_this ? a : ws->pool[push_target ? 1 : 3].get();
		};
		auto make_zero = [&](LevelBuf *lb) -> int { // shortestpath: a level's own buffer before a top-down level writes into it
			PGQ_TRY(lb->buf.reserve(words * 8));
			PGQ_TRY(lb->nz.reserve(nz_bytes));
			if (lb->dirty) {
				PGQ_HIP_TRY(hipMemsetAsync(lb->buf.p, 0, words * 8, st));
				PGQ_HIP_TRY(hipMemsetAsync(lb->nz.p, 0, nz_bytes, st));
				lb->dirty = false;
			}
			return PGQ_OK;
		};
		LevelBuf *cur = nullptr;
		const bool start_done = !with_paths && ws->prereset_V == V && ws->prereset_WD == WD; // in front of the lane assignment's wait
		ws->prereset_V = -1; // one batch's worth, whoever uses `seen` next
		if (with_paths) {
			PGQ_HIP_TRY(hipMemsetAsync(ws->seen.p, 0, words * 8, st));
			PGQ_HIP_TRY(hipMemsetAsync(ws->counters.p, 0, sizeof(Counters), st));
			PGQ_HIP_TRY(hipMemsetAsync(ws->dpart.p, 0, (size_t)kOpenRep * WD * 8, st)); // the copies of the open-lane mask (publish_open_lanes)
			for (auto &lb : ws->levels) lb->dirty = true; // previous batch / call left them in an unknown state
			cur = level_buf(0, true, nullptr);
			PGQ_TRY(make_zero(cur));
		} else {
			if (!start_done) PGQ_TRY(batch_state_reset(ws, V, WD));
			cur = level_buf(0, true, nullptr);
		}
		u64 *act_cur = &d_cnt->act[0][0]; // zeroed with the counter block above
		u32 open_before = (u32)(hi - lo); // rows open before the level whose counters are being looked at (byte model of detection)
		u32 sparse_front_words = 0;       // k_pull_sparse sizes its packed-word buffer from the frontier's words
		{
			KernelTimer kt(st, K_PREP);
			hipLaunchKernelGGL(k_init_batch<WD>, dim3(blocks_for(L)), dim3(256), 0, st, sh->usrc.as<int32_t>(), U,
			                   (int64_t)base_lane, c->off, cur->buf.as<u64>(), cur->nz.as<u32>(), ws->seen.as<u64>(),
			                   act_cur, ws->qbuf[0].as<u64>(), qcap, pchunk, (u32)(hi - lo), d_cnt);
			kt.stop();
		}
		cur->dirty = true;
		if (outp.want_te) {
			PGQ_HIP_TRY(hipMemsetAsync(ws->lane_sums.p, 0, (size_t)kMaxTeLevels * L * 8, st));
			hipLaunchKernelGGL(k_lane_degree_sums<WD>, dim3(4 * ncu), dim3(256), 0, st, cur->buf.as<u64>(), c->off, V,
			                   ws->lane_sums.as<u64>());
		}
		// The destination probe answers a pair one expansion early; it costs one in-neighbour scan per open pair,
		// so it is used while the batch has few pairs relative to the graph (not for cross products) and never in
		// the traversed-edge accounting pass (which needs every level of every lane).
		const bool use_probe = opt.probe && !outp.want_te; // whether a level is probed is decided per level (decide_level)
		// open pairs at or below this count stop the batch: 0 = everything answered, > 0 = defer the stragglers
		int stop = -1;
		if (use_probe) {
			stop = 0;
			// a narrow batch is scan-bound: re-running its stragglers costs as much as finishing them here
			if (opt.defer && outp.depth < 2 && WD >= 8) stop = (int)std::min<int64_t>(L / opt.defer, (hi - lo) / opt.defer);
		}
		LevelRule rule;
		rule.E = (double)E;
		rule.V = (double)V;
		rule.push_div = opt.push_div;
		rule.sparse_below = opt.sparse_below;
		rule.wd = WD;
		rule.force_mode = opt.force_mode;
		rule.force_pull = opt.force_pull;
		rule.probe_always = opt.probe_always;
		rule.use_probe = use_probe ? 1 : 0;
		const bool lanes_ok = opt.lanes && c->rpk != nullptr;

		// ---- host state of the level loop (what enqueueing a level changes; snapshotted per enqueued-ahead level) ----
		struct HostState {
			LevelBuf *cur;
			int par, act_sel;
			bool queue_valid, dirty[2];
			bool pending_fold; // the last level ended with k_detect and its open-lane words are still in the copies
		};
		HostState hs { cur, 0, 0, true, { false, false }, false }; // queue_valid: qbuf[par] describes `cur`
		auto save_dirty = [&](HostState &h) {
			for (size_t k = 0; k < 2 && k < ws->pool.size(); k++) h.dirty[k] = ws->pool[k]->dirty;
		};
		auto load_dirty = [&](const HostState &h) {
			for (size_t k = 0; k < 2 && k < ws->pool.size(); k++) ws->pool[k]->dirty = h.dirty[k];
		};
		u32 last_cw_cap = 0;
		int levels_run = 0;
		std::vector<uint8_t> ran_plan; // the levels this batch really ran: the next batch's plan

		// Launches the kernels of level t as `bits` says.  spec: the level runs ahead of the host's knowledge — its
		// k_level_reset logs and checks on the device (SpecArgs) and its memsets are kernels that honour `done`.
		auto enqueue_level = [&](int t, u32 bits, bool spec, int prev_stop) -> int {
			LevelBuf *cur = hs.cur;
			// The two-hop probe replaces an expansion for the rows still open — while it is the cheaper of the two: a row walks the
			// in-lists of its destination's in-neighbours (the graph's mean two-hop walk, at most probe2_cap entries), two
			// gathered 64-byte sectors per entry, one workgroup per row; a dense level moves E x (8 + 6 WD) bytes.  R-MAT-22
			// (mean two-hop walk over the cap): 4096 open rows made each probe 2.3 ms where a level takes 1.9, five times per
			// call, without saving a level — 11 of the cross product's 23.8 ms through the lanes.
			const int64_t probe2_rows_worth = std::max<int64_t>(
			    64, (int64_t)((double)E * (8.0 + 6.0 * WD) / (128.0 * std::max(64.0, std::min((double)opt.probe2_cap, c->two_hop_mean)))));
			const bool push = bits & kLvPush, sparse_level = bits & kLvSparse, probe_now = bits & kLvProbe;
			LevelBuf *nxt = level_buf(t, push, cur);
			const bool lanes_level = sparse_level && lanes_ok;
			u64 *act_cur = &d_cnt->act[hs.act_sel][0], *act_nxt = &d_cnt->act[hs.act_sel ^ 1][0];
			const int par = hs.par;
			auto zero_level = [&](LevelBuf *lb) -> int { // the target of a top-down level
				if (with_paths) return make_zero(lb);
				if (lb->dirty) { // (its buffers exist and are laid out for this batch: the batch's start saw to both pool members)
					hipLaunchKernelGGL(k_clean_by_nz, dim3(4 * ncu), dim3(256), 0, st, lb->buf.as<u64>(), lb->nz.as<u32>(), V, (int)WD,
					                   spec ? (const Counters *)d_cnt : (const Counters *)nullptr);
					lb->dirty = false;
				}
				return PGQ_OK;
			};
			// reset the per-level counters but keep the queue counts
			{
				const bool zq_cur = push && !hs.queue_valid; // queue rebuilt from the dense frontier below
				const int q_cur = par, q_nxt = par ^ 1;
				SpecArgs sp { nullptr, nullptr, t, bits, prev_stop };
				if (spec) {
					sp.log = ws->h_log;
					sp.status = reinterpret_cast<u32 *>(ws->h_log + kSpecLevels + 2);
				}
				KernelTimer kt(st, K_QUEUE); // bookkeeping between levels: no byte model, its time counts in the chain
				hipLaunchKernelGGL(k_level_reset, dim3(1), dim3(256), 0, st, d_cnt, hs.act_sel ^ 1,
				                   (int)((push && q_nxt == 0) || (zq_cur && q_cur == 0)),
				                   (int)((push && q_nxt == 1) || (zq_cur && q_cur == 1)), rule, sp,
				                   hs.pending_fold ? ws->dpart.as<u64>() : (u64 *)nullptr);
				kt.stop();
				hs.pending_fold = false;
			}
			if (probe_now) {
				KernelTimer kt(st, K_DETECT);
				// up to one wavefront per row (few rows: the wavefronts of a 64-row chunk share its open rows), at most 8192
				hipLaunchKernelGGL(k_probe<WD>, dim3(std::min(blocks_for((hi - lo) * 64, 1024), (unsigned)kOpenGrid)), dim3(1024), 0, st, lo, hi,
				                   sh->skey.as<u32>(), sh->sdst.as<int32_t>(), sh->sres.as<int32_t>(), base_lane,
				                   cur->buf.as<u64>(), cur->nz.as<u32>(), c->roff, c->radj, t, ws->dpart.as<u64>(), d_cnt, std::max(64, opt.probe_max_in));
				if (opt.probe2 && !with_paths)
					hipLaunchKernelGGL(k_probe2<WD>, dim3((unsigned)std::min<int64_t>(hi - lo, 2 * ncu)), dim3(1024), 0, st, lo, hi,
					                   sh->skey.as<u32>(), sh->sdst.as<int32_t>(), sh->sres.as<int32_t>(), base_lane,
					                   cur->buf.as<u64>(), cur->nz.as<u32>(), c->roff, c->radj, t,
					                   (u32)std::min<int64_t>(probe2_rows_worth,
					                                          std::max<int64_t>(std::min<int64_t>(L / std::max(1, opt.probe2_div), (hi - lo) / std::max(1, opt.probe2_div)),
					                                                            std::min<int64_t>(hi - lo, opt.probe2_abs))),
					                   (int64_t)opt.probe2_cap, ws->dpart.as<u64>(), act_nxt, d_cnt);
				else
					hipLaunchKernelGGL(k_open_merge, dim3(1), dim3(256), 0, st, ws->dpart.as<u64>(), act_nxt, WD, d_cnt);
				kt.stop();
				std::swap(act_cur, act_nxt); // the expansion below only serves lanes that still have open pairs
				hs.act_sel ^= 1;
			}
			const int stop_lvl = probe_now ? stop : -1; // the expansion returns at once when the probe left <= stop pairs open
			if (push) {
				if (!hs.queue_valid) {
					KernelTimer kt(st, K_QUEUE);
					hipLaunchKernelGGL(k_queue_from_dense<WD>, dim3(std::min(blocks_for(V), 16u * ncu)), dim3(256), 0, st,
					                   cur->nz.as<u32>(), V, c->off, pchunk, ws->qbuf[par].as<u64>(), qcap, par, d_cnt);
					kt.stop();
				}
				PGQ_TRY(zero_level(nxt));
				{
					KernelTimer kt(st, K_PUSH);
					hipLaunchKernelGGL(k_push<WD>, dim3(push_grid), dim3(256), 0, st, c->off, c->adj, cur->buf.as<u64>(),
					                   ws->seen.as<u64>(), nxt->buf.as<u64>(), nxt->nz.as<u32>(), act_cur,
					                   ws->qbuf[par].as<u64>(), par, qcap, pchunk, stop_lvl, d_cnt);
					kt.stop();
				}
				nxt->dirty = true;
				if (!with_paths) {
					KernelTimer kt(st, K_QUEUE);
					hipLaunchKernelGGL(k_clear_items<WD>, dim3(4 * ncu), dim3(256), 0, st, ws->qbuf[par].as<u64>(), par,
					                   qcap, cur->buf.as<u64>(), cur->nz.as<u32>(), d_cnt);
					kt.stop();
					cur->dirty = false;
				}
				hs.queue_valid = false; // rebuilt from nz if the next level is top-down too
			} else {
				PGQ_TRY(nxt->buf.reserve(words * 8));
				PGQ_TRY(nxt->nz.reserve(nz_bytes));
				if (c->n_pull_hub_vertices > 0) {
					KernelTimer kt(st, K_PULL_HUB);
					hipLaunchKernelGGL(k_pull_hub_zero<WD>, dim3(blocks_for(c->n_pull_hub_vertices * WD)), dim3(256), 0,
					                   st, c->pull_hub_vertices, c->n_pull_hub_vertices, nxt->buf.as<u64>(), stop_lvl, d_cnt);
					hipLaunchKernelGGL(k_pull_hub<WD>, dim3(blocks_for(c->n_pull_hub_items * 64)), dim3(256), 0, st,
					                   c->pull_hubs, c->n_pull_hub_items, c->radj, cur->buf.as<u64>(), cur->nz.as<u32>(),
					                   ws->seen.as<u64>(), nxt->buf.as<u64>(), act_cur, stop_lvl, d_cnt);
					hipLaunchKernelGGL(k_pull_hub_fold<WD>, dim3(blocks_for(c->n_pull_hub_vertices)), dim3(256), 0, st,
					                   c->pull_hub_vertices, c->n_pull_hub_vertices, c->off, ws->seen.as<u64>(),
					                   nxt->buf.as<u64>(), nxt->nz.as<u32>(), stop_lvl, d_cnt);
					kt.stop();
				}
				// frontier sparse in lane-words, or few lane-words still wanted -> edge-organised sparse kernel
				if (lanes_level) {
					KernelTimer kt(st, K_PULL_SPARSE);
					PGQ_TRY(pull_lanes_level(c, ws, WD, cur->buf.as<u64>(), cur->nz.as<u32>(), ws->seen.as<u64>(),
					                         nxt->buf.as<u64>(), nxt->nz.as<u32>(), act_cur, stop_lvl, d_cnt));
					kt.stop();
				} else if (sparse_level) {
					// only after a host round trip (never enqueued ahead): the packed-word buffer is sized from the frontier's words
					const u32 cw_cap = (u32)std::min<int64_t>((int64_t)sparse_front_words + 64, 0x7FFFFFF0ll);
					last_cw_cap = cw_cap;
					const int bit_words = (int)(((V + 63) / 64) * 2 + 2); // even: 64-vertex blocks
					PGQ_TRY(ws->cbits.reserve((size_t)bit_words * 4 + 64));
					PGQ_TRY(ws->cbbase.reserve((size_t)bit_words * 2 + 64));
					PGQ_TRY(ws->cmeta.reserve((size_t)std::max<int64_t>(V, 1) * sizeof(FrontMeta)));
					// sized generously once (2 words per vertex) so that growing frontiers do not reallocate per level
					PGQ_TRY(ws->cwords.reserve(std::max<size_t>((size_t)cw_cap, 2 * (size_t)std::max<int64_t>(V, 1)) * 8));
					u32 *d_total = reinterpret_cast<u32 *>(&d_cnt->pad); // {frontier vertices, packed words}
					KernelTimer kt(st, K_PULL_SPARSE);
					// >= 512 vertices per wavefront: two claims (atomicAdd) per wavefront, keep them few
					hipLaunchKernelGGL(k_compact_frontier<WD>, dim3(std::min(blocks_for(V / 8 + 1), 2u * ncu)), dim3(256), 0, st,
					                   cur->nz.as<u32>(), cur->buf.as<u64>(), V, ws->cmeta.as<FrontMeta>(),
					                   ws->cwords.as<u64>(), ws->cbits.as<u32>(), ws->cbbase.as<u32>(), d_total, cw_cap,
					                   stop_lvl, d_cnt);
					// graphs whose frontier bit map + block bases fit in LDS beside the accumulators run 1024-thread
					// workgroups that keep them there (SF100: 56 KB + 28 KB)
					const size_t dyn_bytes = (size_t)bit_words * 4 + (size_t)bit_words * 2;
#define PGQ_LAUNCH_SPARSE(UNR, VR)                                                                                       \
	do {                                                                                                               \
		auto kfn = k_pull_sparse<WD, UNR, 16, VR>;                                                                            \
		static std::atomic<size_t> static_lds { 0 };                                                                   \
		if (!static_lds) {                                                                                             \
			hipFuncAttributes fa;                                                                                      \
			static_lds = hipFuncGetAttributes(&fa, (const void *)kfn) == hipSuccess ? fa.sharedSizeBytes + 1 : 1;      \
		}                                                                                                              \
		const bool lds_map = opt.sparse_lds && static_lds > 1 && static_lds + dyn_bytes + 256 <= 160 * 1024;           \
		if (lds_map) {                                                                                                 \
			static std::atomic<size_t> attr_bytes { 0 };                                                               \
			if (attr_bytes < dyn_bytes) {                                                                              \
				(void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,               \
				                          (int)(160 * 1024 - static_lds));                                             \
				attr_bytes = 160 * 1024;                                                                               \
			}                                                                                                          \
			hipLaunchKernelGGL(kfn, dim3(ncu), dim3(1024), dyn_bytes, st, c->roff, c->radj, c->rown, c->off, c->pull_parts,      \
			                   c->n_pull_parts, ws->cbits.as<u32>(), ws->cbbase.as<u32>(), ws->cmeta.as<FrontMeta>(),  \
			                   ws->cwords.as<u64>(), ws->seen.as<u64>(), nxt->buf.as<u64>(), nxt->nz.as<u32>(),        \
			                   act_cur, bit_words, opt.sparse_spill, stop_lvl, d_cnt);                                     \
		} else {                                                                                                       \
			hipLaunchKernelGGL((k_pull_sparse<WD, UNR, 4, VR>), dim3(pull_grid), dim3(256), 0, st, c->roff, c->radj,       \
			                   c->rown, c->off, c->pull_parts, c->n_pull_parts, ws->cbits.as<u32>(), ws->cbbase.as<u32>(),      \
			                   ws->cmeta.as<FrontMeta>(), ws->cwords.as<u64>(), ws->seen.as<u64>(),                    \
			                   nxt->buf.as<u64>(), nxt->nz.as<u32>(), act_cur, bit_words, opt.sparse_spill, stop_lvl, d_cnt); \
		}                                                                                                              \
	} while (0)
					const int pw = std::max(1, std::min(3, opt.sparse_pw));
					if (opt.sparse_unroll >= 4) {
						if (pw == 1) PGQ_LAUNCH_SPARSE(4, 1);
						else if (pw == 2) PGQ_LAUNCH_SPARSE(4, 2);
						else PGQ_LAUNCH_SPARSE(4, 3);
					} else if (opt.sparse_unroll >= 2) {
						if (pw == 1) PGQ_LAUNCH_SPARSE(2, 1);
						else PGQ_LAUNCH_SPARSE(2, 2);
					} else {
						PGQ_LAUNCH_SPARSE(1, 2);
					}
#undef PGQ_LAUNCH_SPARSE
					kt.stop();
				} else {
					KernelTimer kt(st, K_PULL);
					hipLaunchKernelGGL(k_pull<WD>, dim3(pull_grid), dim3(256), 0, st, c->roff, c->radj, c->off,
					                   c->pull_parts, c->n_pull_parts, cur->buf.as<u64>(), cur->nz.as<u32>(),
					                   ws->seen.as<u64>(), nxt->buf.as<u64>(), nxt->nz.as<u32>(), act_cur, (int)V, chunk,
					                   stop_lvl, d_cnt);
					kt.stop();
				}
				nxt->dirty = true;
				hs.queue_valid = false;
			}
			if (outp.want_te) {
				if (t >= kMaxTeLevels) return fail(PGQ_ERR_UNSUPPORTED, "traversed-edge accounting supports at most 1023 levels");
				hipLaunchKernelGGL(k_lane_degree_sums<WD>, dim3(4 * ncu), dim3(256), 0, st, nxt->buf.as<u64>(), c->off, V,
				                   ws->lane_sums.as<u64>() + (size_t)t * L);
			}
			if (!probe_now) {
				// -- detect finished pairs (iterativelength.cpp:119-129) in the frontier just produced, rebuild the active-lane mask
				KernelTimer kt(st, K_DETECT);
				const int un = opt.detect_unroll >= 4 ? 4 : (opt.detect_unroll >= 2 ? 2 : 1);
				const dim3 grid(std::min(blocks_for((hi - lo + un - 1) / un, 1024), (unsigned)std::min(kOpenGrid, std::max(1, opt.detect_grid_mult) * ncu / 4)));
#define PGQ_DETECT(UNR)                                                                                                  \
	hipLaunchKernelGGL((k_detect<WD, UNR>), grid, dim3(1024), 0, st, lo, hi, sh->skey.as<u32>(), sh->sdst.as<int32_t>(),   \
	                   sh->sres.as<int32_t>(), base_lane, nxt->buf.as<u64>(), nxt->nz.as<u32>(), t,                       \
	                   (u32)std::min<size_t>(words / 2, 0xFFFFFFFFu), ws->dpart.as<u64>(), d_cnt)
				if (un == 4) PGQ_DETECT(4);
				else if (un == 2) PGQ_DETECT(2);
				else PGQ_DETECT(1);
#undef PGQ_DETECT
				// the words are folded into the mask by the next level's k_level_reset; the host loop reads the mask itself
				if (spec) hs.pending_fold = true;
				else hipLaunchKernelGGL(k_open_merge, dim3(1), dim3(256), 0, st, ws->dpart.as<u64>(), act_nxt, WD, d_cnt);
				kt.stop();
				hs.act_sel ^= 1;
			}
			hs.cur = nxt; // the caller takes it back when the level turns out not to have counted (deferral)
			return PGQ_OK;
		};

		// What the host does with a level's counters once it has them.  Returns 1 when the batch is over (the probe left at
		// most `stop` rows open: they are deferred), 0 to go on.
		auto post_level = [&](int t, u32 bits, const LevelLog &hc) -> int {
			const bool push = bits & kLvPush, sparse_level = bits & kLvSparse, probe_now = bits & kLvProbe;
			const bool lanes_level = sparse_level && lanes_ok;
			ran_plan.push_back((uint8_t)bits);
			if (sparse_level && !lanes_level && hc.pad2 > last_cw_cap)
				return fail(PGQ_ERR_HIP, "internal error: packed frontier holds " + std::to_string(hc.pad2) +
				                             " words, expected at most " + std::to_string(last_cw_cap));
			if (probe_now && hc.unresolved <= (u32)stop) { // the expansion kernels returned immediately
				if (hc.unresolved > 0) {
					hipLaunchKernelGGL(k_mark_deferred, dim3(blocks_for(hi - lo)), dim3(256), 0, st, lo, hi, sh->sres.as<int32_t>());
					outp.deferred = true;
					S.deferred_pairs += hc.unresolved;
				}
				return 1;
			}
			if (push) S.push_levels++;
			else S.pull_levels++;
			S.levels++;
			S.edges_scanned += (int64_t)hc.edges_scanned;
			S.word_gathers += (int64_t)hc.word_gathers;
			S.frontier_vertices += (int64_t)hc.front_vertices;
			// algorithmic bytes of this level's kernels
			if (push) {
				S.algo_bytes[K_PUSH] += (double)hc.edges_scanned * 4.0 + (double)hc.word_gathers * 16.0;
			} else {
				// per scanned in-edge: 4 B adjacency + 4 B non-empty-word mask; per gathered lane-word 8 B; per vertex:
				// offsets 16 B + seen/next words 16*WD B + mask 4 B (the sparse variant reads seen twice: 24*WD)
				if (lanes_level) // 4 B per in-slot, one 16-byte record per in-edge leaving a frontier vertex, seen read + next/seen written
					S.algo_bytes[K_PULL_SPARSE] += (double)hc.edges_scanned * 4.0 + (double)hc.word_gathers * 16.0 + (double)V * (4.0 + 24.0 * WD);
				else
					S.algo_bytes[sparse_level ? K_PULL_SPARSE : K_PULL] +=
					    (double)hc.edges_scanned * 8.0 + (double)hc.word_gathers * 8.0 +
					    (double)V * (20.0 + 16.0 * WD) + (sparse_level ? (double)hc.word_gathers * 8.0 + (double)V * 4.0 : 0.0);
			}
			// detection / probe: the rows' result words, lane ids and destinations (12 B per row of the batch) and, per row
			// still open, the 4-byte mask look-up and the 8-byte frontier word (detection), or its destination's in-list with
			// the two look-ups per entry (probe: mean in-degree x 16 B)
			S.algo_bytes[K_DETECT] += (double)(hi - lo) * 12.0 +
			                          (double)open_before * (probe_now ? (double)E / (double)std::max<int64_t>(V, 1) * 16.0 : 12.0);
			if (opt.trace)
				fprintf(stderr, "[pgq] batch %d level %d %s%s WD=%d front_v=%u front_e=%llu scanned=%llu gathers=%llu unresolved=%u push_ms=%.3f pull_ms=%.3f\n",
				        b, t, probe_now ? "probe+" : "", push ? "push" : (lanes_level ? "pull_lanes" : (sparse_level ? "pull_sparse" : "pull")), WD, hc.front_vertices, (unsigned long long)hc.front_edges,
				        (unsigned long long)hc.edges_scanned, (unsigned long long)hc.word_gathers, hc.unresolved,
				        S.kernel_ms[K_PUSH], S.kernel_ms[K_PULL] + S.kernel_ms[K_PULL_HUB] + S.kernel_ms[K_PULL_SPARSE]);
			open_before = hc.unresolved;
			levels_run = t;
			return 0;
		};

		LevelLog last {}; // the counters after the last level the host knows about
		bool batch_over = false;
		int t = 1;
		// ---- levels enqueued ahead under the plan of the batch before ----
		std::vector<uint8_t> plan;
		if (spec_ok) {
			std::lock_guard<std::mutex> g(c->plan_lock);
			plan = c->level_plan[plan_slot];
		}
		bool have_last = false;
		// (a plan whose FIRST level cannot be enqueued — a packed sparse level without the lane-list kernel — buys nothing: the
		// logging launch and its wait would only be added in front of the round-trip loop)
		if (!plan.empty() && (plan[0] & kLvSparse) && !lanes_ok) plan.clear();
		if (!plan.empty()) {
			std::vector<HostState> snaps;
			u32 *status = reinterpret_cast<u32 *>(ws->h_log + kSpecLevels + 2);
			status[0] = status[1] = 0;
			int prev_stop = -1, K = 0;
			for (; K < (int)plan.size() && K < kSpecLevels; K++) {
				const u32 bits = plan[(size_t)K];
				if ((bits & kLvSparse) && !lanes_ok) break; // k_pull_sparse sizes a buffer from the frontier's words: not ahead of the host
				save_dirty(hs);
				snaps.push_back(hs);
				PGQ_TRY(enqueue_level(K + 1, bits, true, prev_stop));
				prev_stop = (bits & kLvProbe) ? stop : -1;
			}
			save_dirty(hs);
			snaps.push_back(hs);
			{ // behind the last enqueued level: log its counters, say whether the batch is over
				SpecArgs sp { ws->h_log, status, K + 1, kLvNone, prev_stop };
				hipLaunchKernelGGL(k_level_reset, dim3(1), dim3(256), 0, st, d_cnt, hs.act_sel ^ 1, 0, 0, rule, sp,
				                   hs.pending_fold ? ws->dpart.as<u64>() : (u64 *)nullptr);
			}
			PGQ_WAIT(st);
			KernelTimer::flush();
			S.spec_batches++;
			const u32 code = status[0];
			const int t_stop = (int)status[1]; // levels 1 .. t_stop - 1 ran; log[k] = the counters after level k
			if (code == 0 || t_stop < 1 || t_stop > K + 1) return fail(PGQ_ERR_HIP, "internal error: enqueued-ahead levels left no status");
			open_before = (u32)(hi - lo);
			for (int k = 1; k < t_stop && !batch_over; k++) {
				const int r = post_level(k, plan[(size_t)k - 1], ws->h_log[k]);
				if (r < 0) return r;
				batch_over = r == 1;
			}
			// what the device really did: levels 1 .. t_stop - 1 (the kernels enqueued behind them returned at once)
			hs = snaps[(size_t)t_stop - 1];
			hs.pending_fold = false; // the k_level_reset that called the levels off had folded them already
			load_dirty(hs);
			if (!batch_over) {
				if (code == 1) {
					batch_over = true; // nothing open or an empty frontier: the loop below would not be entered
				} else { // the plan did not fit (2) or ran out (3): the host takes over at level t_stop
					last = ws->h_log[t_stop - 1];
					have_last = true;
					t = t_stop;
					PGQ_HIP_TRY(hipMemsetAsync(&d_cnt->done, 0, 4, st));
					S.spec_aborts++;
				}
			}
			if (t_stop >= 1) S.spec_levels += t_stop - 1;
		}
		if (!batch_over && !have_last) {
			PGQ_HIP_TRY(hipMemcpyAsync(ws->h_cnt, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st));
			PGQ_WAIT(st);
			last.front_edges = ws->h_cnt->front_edges;
			last.front_words = ws->h_cnt->front_words;
			last.front_vertices = ws->h_cnt->front_vertices;
			last.unresolved = (u32)(hi - lo);
			last.nzw = WD;
			open_before = last.unresolved;
		}
		// ---- one host round trip per level ----
		for (; !batch_over && last.unresolved > 0 && last.front_edges > 0; t++) {
			const u32 bits = decide_level(rule, last.front_edges, last.front_words, last.front_vertices, last.unresolved,
			                              t == 1 ? WD : (int)last.nzw);
			sparse_front_words = last.front_words;
			PGQ_TRY(enqueue_level(t, bits, false, -1));
			PGQ_HIP_TRY(hipMemcpyAsync(ws->h_cnt, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, st));
			PGQ_WAIT(st);
			KernelTimer::flush();
			const Counters &hc = *ws->h_cnt;
			last.front_edges = hc.front_edges;
			last.edges_scanned = hc.edges_scanned;
			last.word_gathers = hc.word_gathers;
			last.front_vertices = hc.front_vertices;
			last.unresolved = hc.unresolved;
			last.front_words = hc.front_words;
			last.pad2 = hc.pad2;
			last.nzw = 0;
			for (int w = 0; w < WD; w++) last.nzw += hc.act[hs.act_sel][w] != 0;
			const int r = post_level(t, bits, last);
			if (r < 0) return r;
			batch_over = r == 1;
		}
		if (spec_ok && !ran_plan.empty()) {
			// (the last batch of a call may hold a handful of lanes: its levels are not what the next FULL batch will run)
			const bool full = (U - (int64_t)base_lane) * 2 >= L;
			std::lock_guard<std::mutex> g(c->plan_lock);
			if (full || c->level_plan[plan_slot].empty()) c->level_plan[plan_slot] = ran_plan;
		}
		cur = hs.cur;
		ws->pool_trusted = true; // the dirty flags of the sparse pool say what the device did
		if (outp.want_te)
			hipLaunchKernelGGL(k_pair_te, dim3(blocks_for(hi - lo)), dim3(256), 0, st, lo, hi, sh->skey.as<u32>(),
			                   sh->sres.as<int32_t>(), base_lane, ws->lane_sums.as<u64>(), (int)L, levels_run + 1,
			                   sh->ste.as<int64_t>());
		// -- paths for this batch
		if (with_paths) {
			const int64_t cnt_pairs = hi - lo;
			h_res.resize(cnt_pairs);
			h_off.resize(cnt_pairs);
			PGQ_HIP_TRY(hipMemcpyAsync(h_res.data(), sh->sres.as<int32_t>() + lo, (size_t)cnt_pairs * 4,
			                           hipMemcpyDeviceToHost, st));
			PGQ_WAIT(st);
			int64_t need = child_base;
			for (int64_t i = 0; i < cnt_pairs; i++) {
				h_off[i] = need;
				if (h_res[i] > 0) need += 2 * (int64_t)h_res[i] + 1;
			}
			PGQ_HIP_TRY(hipMemcpyAsync(sh->soff.as<int64_t>() + lo, h_off.data(), (size_t)cnt_pairs * 8,
			                           hipMemcpyHostToDevice, st));
			bool fits = true;
			if (d_child_ext) {
				fits = need <= child_cap_ext;
			} else if ((size_t)need * 8 > sh->child.cap) {
				// grow, preserving what earlier batches wrote
				PGQ_TRY(grow_keeping(sh->child, (size_t)need * 8 * 2, (size_t)child_base * 8, st));
				d_child = sh->child.as<int64_t>();
			}
			if (!d_child_ext) d_child = sh->child.as<int64_t>();
			if (fits && need > child_base) {
				std::vector<const u64 *> tab((size_t)levels_run + 1);
				for (int t = 0; t <= levels_run; t++) tab[t] = ws->levels[t]->buf.as<u64>();
				PGQ_TRY(sh->levels_tab.reserve(tab.size() * sizeof(u64 *)));
				PGQ_HIP_TRY(hipMemcpyAsync(sh->levels_tab.p, tab.data(), tab.size() * sizeof(u64 *), hipMemcpyHostToDevice, st));
				KernelTimer kt(st, K_RECON);
				hipLaunchKernelGGL(k_reconstruct<WD>, dim3(blocks_for(cnt_pairs * 64)), dim3(256), 0, st, lo, hi,
				                   sh->skey.as<u32>(), sh->sdst.as<int32_t>(), sh->sres.as<int32_t>(),
				                   sh->soff.as<int64_t>(), base_lane, (const u64 *const *)sh->levels_tab.p, c->roff,
				                   c->radj, c->off, c->adj, c->edge_ids, d_child);
				kt.stop();
				PGQ_WAIT(st); // tab is a stack vector
				KernelTimer::flush();
			}
			if (!fits) child_overflow = true;
			child_base = need;
		}
	}
	if (with_paths && b0 == 0) {
		// src == dst rows: [src]
		const int64_t lo = bs[nb + 1], hi = bs[nb + 2];
		if (hi > lo) {
			int64_t need = child_base + (hi - lo);
			bool fits = true;
			if (d_child_ext) fits = need <= child_cap_ext;
			else if ((size_t)need * 8 > sh->child.cap) {
				PGQ_TRY(grow_keeping(sh->child, (size_t)need * 8, (size_t)child_base * 8, st));
			}
			if (!d_child_ext) d_child = sh->child.as<int64_t>();
			if (fits) {
				hipLaunchKernelGGL(k_trivial_paths, dim3(blocks_for(hi - lo)), dim3(256), 0, st, lo, hi,
				                   sh->ssrc.as<int32_t>(), child_base, sh->soff.as<int64_t>(), d_child);
			} else {
				child_overflow = true;
			}
			child_base = need;
		}
		outp.child_used = child_base;
		// not an early return: the straggler pass still has to run so that the lengths are complete and child_used
		// reports everything the caller must provide
		if (child_overflow) outp.overflow = true;
	}
	PGQ_WAIT(st); // other workers / the caller read sres next
	KernelTimer::flush();
	return PGQ_OK;
}

void merge_stats(pgq_stats_t &into, const pgq_stats_t &from) {
	into.batches += from.batches;
	into.levels += from.levels;
	into.push_levels += from.push_levels;
	into.pull_levels += from.pull_levels;
	into.edges_scanned += from.edges_scanned;
	into.word_gathers += from.word_gathers;
	into.frontier_vertices += from.frontier_vertices;
	into.deferred_pairs += from.deferred_pairs;
	into.spec_batches += from.spec_batches;
	into.spec_levels += from.spec_levels;
	into.spec_aborts += from.spec_aborts;
	into.host_waits += from.host_waits;
	into.ball_segments += from.ball_segments;
	into.ball_calls += from.ball_calls;
	for (int k = 0; k < PGQ_KCLASS_MAX; k++) {
		into.algo_bytes[k] += from.algo_bytes[k];
		into.kernel_ms[k] += from.kernel_ms[k];
		into.launches[k] += from.launches[k];
	}
}

// Whether the pair-centric pre-pass may run for this call at all, and whether the host-side cost model sends n rows
// (each taken as a distinct source) to it.  Shared by search_device and the chunk entry point (zero-copy staging).
constexpr int64_t kMeetDecideRows = 16384; // above: the distinct sources are sampled and the decision is taken on the device
static bool prepass_may(const pgq_csr *c, const SearchOutput &outp) {
	// depth 1 = the stragglers a lane batch deferred (a few far pairs of a cross product): the pre-pass answers them from
	// two-hop scans instead of another round of whole-graph levels
	return options().meet && !outp.want_te && outp.depth <= 1 && !outp.from_meet && c->E > 0 && c->fdesc != nullptr;
}
// Bytes the pre-pass moves per row.  Known once a pre-pass has run on this CSR (measured: its kernels count the entries
// they walk; calibrate_prepass runs 1024 pseudo-random pairs through it before the first large call is routed).  Before
// that: the cheaper endpoint's WHOLE two-hop neighbourhood, ~0.6 x E[in-degree x out-degree] entries — what a far pair
// costs.  Close pairs stop after a fraction of it: on the SF100-shaped graph the bound is 6x what 65,536 random pairs
// move (12 KB per row), and a 2048 x 32 cross product priced with it went through the lane batches at 0.73 ms where the
// pre-pass takes 0.22.
static double prepass_row_bytes(const pgq_csr *c) {
	const double measured = c->meet_bpr.load(std::memory_order_relaxed);
	return measured > 0 ? measured : c->two_hop_mean * 4.0 * 0.6;
}
__global__ void k_calibration_pairs(int64_t n, int64_t V, int64_t *__restrict__ src, int64_t *__restrict__ dst) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	auto mix = [](u64 x) { // splitmix64
		x += 0x9E3779B97F4A7C15ull;
		x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
		x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
		return x ^ (x >> 31);
	};
	src[i] = (int64_t)(mix(2 * (u64)i) % (u64)V);
	dst[i] = (int64_t)(mix(2 * (u64)i + 1) % (u64)V);
}
static int calibrate_prepass(pgq_csr *c) {
	std::lock_guard<std::mutex> g(c->lazy_lock);
	if (c->meet_bpr.load() > 0) return PGQ_OK;
	const int64_t n0 = 1024;
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *w = lease.ws;
	PGQ_TRY(w->in_src.reserve((size_t)n0 * 8));
	PGQ_TRY(w->in_dst.reserve((size_t)n0 * 8));
	PGQ_TRY(w->out_len.reserve((size_t)n0 * 8));
	hipLaunchKernelGGL(k_calibration_pairs, dim3(blocks_for(n0)), dim3(256), 0, w->stream, n0, c->V, w->in_src.as<int64_t>(), w->in_dst.as<int64_t>());
	pgq_stats_t &S = tstats().s;
	const pgq_stats_t saved = S; // the caller's statistics are about its own rows
	u32 nd = 0;
	bool ran = true;
	const int rc = meet_prepass(c, w, n0, w->in_src.as<int64_t>(), w->in_dst.as<int64_t>(), w->out_len.as<int64_t>(), &nd, nullptr, 0,
	                            0.0, 0.0, &ran, nullptr);
	const double bytes = (S.algo_bytes[K_MEET] - saved.algo_bytes[K_MEET]) + (S.algo_bytes[K_MEET4] - saved.algo_bytes[K_MEET4]) +
	                     (S.algo_bytes[K_BIBFS] - saved.algo_bytes[K_BIBFS]);
	S = saved;
	PGQ_TRY(rc);
	c->meet_bpr.store(std::max(64.0, bytes / (double)n0));
	calibration_store(c); // the next handle over a graph of this shape starts with it
	return PGQ_OK;
}
static bool prepass_takes(const pgq_csr *c, int64_t n, const SearchOutput &outp) {
	if (!prepass_may(c, outp)) return false;
	const double meet_bytes = (double)n * prepass_row_bytes(c);
	const double edge_bytes = options().meet_bias * (double)c->E;
	return meet_bytes <= lanes_cost_bytes(edge_bytes, (double)std::min<int64_t>(n, c->V), (double)n, (double)c->V);
}

// Lane assignment + sorting of the rows, then the templated batch loop; results scattered back to row order.
static int search_device_impl(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                              int64_t *d_out_len, bool with_paths, int64_t *d_out_off, int64_t *d_child_ext,
                              int64_t child_cap_ext, SearchOutput &outp);
// The byte models that pick a route price kernels at streaming rate; on a graph past the caches the source-centric route is
// nothing like that (R-MAT-22, 2048 x 1024 rows: global bit maps marked through DRAM atomics, 35,000 far rows searched one
// by one — 16.7 ms where the model says 0.1) and the lane batches are 4.6 x their model (12 ms).  So large grouped calls are
// TIMED, per graph shape: the best wall time per row of the source-centric route is kept (the best of at least two calls: a
// process's first call of a kind pays for allocations, kernel attributes and the calibration); when it is over
// `route_try_factor` x the lane batches' modelled time the next two such calls go through the lanes, and from then on through
// whichever measured faster.  Every route is exact, so this only moves time.  (The figures travel with the calibration cache.)
static int search_device(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                         int64_t *d_out_len, bool with_paths, int64_t *d_out_off, int64_t *d_child_ext,
                         int64_t child_cap_ext, SearchOutput &outp) {
	const Options &o = options();
	const bool timed = o.route_timing && o.ball == 1 && outp.depth == 0 && !with_paths && !outp.want_te && !outp.bidir && !outp.no_ball &&
	                   outp.ball_hint != 0 && n >= (int64_t)std::max(1, o.route_timing_rows);
	if (!timed) return search_device_impl(c, ws, n, d_src, d_dst, d_out_len, with_paths, d_out_off, d_child_ext, child_cap_ext, outp);
	const double tb = c->route_ball_ns.load(std::memory_order_relaxed), tl = c->route_lanes_ns.load(std::memory_order_relaxed);
	const int nb_s = c->route_ball_samples.load(std::memory_order_relaxed), nl_s = c->route_lanes_samples.load(std::memory_order_relaxed);
	// both figures are the best of at least two calls before they decide anything (a first call pays one-time costs)
	// ... and they speak for calls of their own size: a lane batch costs the same for 32 rows per source as for 1024, the
	// source-centric route does not — a call with under half the measured rows is left to the byte models
	const bool same_size = n * 2 >= c->route_rows.load(std::memory_order_relaxed);
	const bool trial = same_size && nb_s >= 2 && nl_s < 2 && c->route_try_lanes.load(std::memory_order_relaxed) != 0;
	outp.prefer_lanes = trial || (same_size && nb_s >= 2 && nl_s >= 2 && tl < tb);
	// (such a call neither follows nor feeds the route memo: what it would leave there — "these buffers go to the lanes" — must
	// not outlive the preference, and the decision kernel in front of the lanes is 40 us of a call that takes milliseconds)
	if (outp.prefer_lanes) outp.no_memo = true;
	const int64_t levels0 = tstats().s.levels;
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = search_device_impl(c, ws, n, d_src, d_dst, d_out_len, with_paths, d_out_off, d_child_ext, child_cap_ext, outp);
	if (rc != PGQ_OK) return rc;
	const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (double)n;
	if (outp.route == 1) {
		const double best = nb_s > 0 ? std::min(tb, ns) : ns;
		c->route_ball_ns.store(best, std::memory_order_relaxed);
		c->route_ball_samples.store(nb_s + 1, std::memory_order_relaxed);
		if (nb_s == 0 || n > c->route_rows.load(std::memory_order_relaxed)) c->route_rows.store(n, std::memory_order_relaxed);
		if (nb_s + 1 >= 2 && outp.source_runs > 0) { // the lane batches' modelled time per row, at 8 TB/s
			const double lanes_ns = lanes_cost_bytes(o.meet_bias * (double)c->E, std::min(outp.source_runs, (double)c->V), (double)n, (double)c->V) / 8000.0 / (double)n;
			if (best > o.route_try_factor * lanes_ns) c->route_try_lanes.store(1, std::memory_order_relaxed);
		}
	} else if (outp.prefer_lanes && tstats().s.levels > levels0) { // (the lane batches did run)
		c->route_lanes_ns.store(nl_s > 0 ? std::min(tl, ns) : ns, std::memory_order_relaxed);
		c->route_lanes_samples.store(nl_s + 1, std::memory_order_relaxed);
	}
	return PGQ_OK;
}

static int search_device_impl(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                              int64_t *d_out_len, bool with_paths, int64_t *d_out_off, int64_t *d_child_ext,
                              int64_t child_cap_ext, SearchOutput &outp) {
	hipStream_t st = ws->stream;
	pgq_stats_t &S = tstats().s;
	S.pairs += n;
	if (n == 0) return PGQ_OK;
	if (n >= (1LL << 31)) return fail(PGQ_ERR_INVALID_ARG, "more than 2^31-1 rows in one call");
	if (with_paths) PGQ_TRY(ensure_edge_ids(c)); // PGQ_UPLOAD_LAZY_EDGE_IDS: the first shortestpath call brings them over
	// Pair-centric pre-pass: rows at distance <= 3 are answered from two-hop neighbourhood scans (pgq_meet.hip); only
	// what it leaves open goes through the lane-batched search below.  It pays while the two-hop walks of all rows
	// cost less than the MS-BFS levels they replace: bytes ~ rows x E[in-degree x out-degree] x 4 (the cheaper
	// endpoint is expanded: ~0.6 of that) against ~16 B per edge per 2048-lane batch.
	const Options &mopt = options();
	const bool may_meet = prepass_may(c, outp);
	// a large call is about to be routed on the pre-pass's bytes per row: measured first if this CSR has none yet
	if (may_meet && n > kMeetDecideRows && mopt.meet_calibrate && c->meet_bpr.load(std::memory_order_relaxed) <= 0) PGQ_TRY(calibrate_prepass(c));
	// Cost model (bytes at streaming rate): the pre-pass walks, per row, the cheaper endpoint's two-hop neighbourhood
	// (~0.6 of E[in-degree x out-degree] entries when it has to walk all of it; it usually stops far earlier: the estimate
	// is on the safe side).  A lane batch of `wd` lane-words costs about one sparse and one dense bottom-up level
	// (or the probes that replace it): E x (12 + 3 wd) bytes — calibrated on the 2048-lane batch of the SF100-shaped
	// graph (0.94 ms ~ 4.3 GB at streaming rate; round 2 priced a batch at 16 B per edge and sent a 2048 x 32 cross product
	// through the lanes at three times the cost of the pre-pass).  `meet_bias` scales the lanes' side.
	const double meet_bytes = (double)n * prepass_row_bytes(c);
	const double edge_bytes = mopt.meet_bias * (double)c->E; // x (12 + 3 wd) per batch
	auto meet_pays = [&](int64_t distinct_sources) { return meet_bytes <= lanes_cost_bytes(edge_bytes, (double)distinct_sources, (double)n, (double)c->V); };
	// few rows: every row is taken as a distinct source (the pessimistic case for the pre-pass); many rows: a sampled
	// estimate of the distinct sources decides ON THE DEVICE, in the same launch chain (cross products share their lanes)
	const bool decide = n > kMeetDecideRows;
	int decide_mode = decide ? 1 : 0, observed_go = -1; // 2: the memo vouches for the pre-pass, the sample only observes (meet_prepass)
	// round 6: the source-centric kernels open the pre-pass's chain and decide on the device (pgq_ball.h); not for paths,
	// not for the rows that kernel itself left open
	int ball_mode = (with_paths || outp.no_ball || outp.bidir || outp.prefer_lanes || n < 2) ? 0 : std::max(0, std::min(2, mopt.ball));
	const bool ball_possible = ball_mode != 0 && c->ball_open_frac.load(std::memory_order_relaxed) <= 0.02; // before the memo's say on THESE rows as they lie
	if (ball_mode == 1) {
		if (outp.ball_hint == 0 || c->ball_open_frac.load(std::memory_order_relaxed) > 0.02) ball_mode = 0;
		else if (mopt.route_memo && outp.ball_hint < 0 && !outp.no_memo) { // (a caller that has looked at the rows knows better than the memo)
			std::lock_guard<std::mutex> g(c->plan_lock);
			const pgq_csr::RouteMemo &m = c->route_memo;
			if (m.ball_n == n && m.ball_src == (const void *)d_src && m.ball_dst == (const void *)d_dst) ball_mode = m.ball_yes ? 3 : 0;
		}
	}
	// the caller has counted the source runs on the host and found the rows grouped: the chain is the two kernels alone (if the
	// device's byte rule declines after all, run_meet falls back to the stage kernels)
	if (ball_mode == 1 && outp.ball_hint == 1) ball_mode = 3;
	const int ball_asked = ball_mode;
	bool ball_ran = false;
	double est_sources = -1.0; // the decision kernel's estimate of the distinct sources (large calls whose chain it opened)
	// shortestpath: the pre-pass also records each answered row's inner vertices (reference tie-break); their lists are
	// packed first, the lists of the rows left to the lane-batched search are appended behind them
	auto run_meet_paths = [&](bool *ran) -> int {
		u32 nd = 0;
		// the lists of the rows the pre-pass answers have at most 9 elements (distance <= 4): the buffer for them is sized
		// up front, so that they are written in the same launch chain as the search
		if (decide_mode == 1) { // asked first, on its own: the buffers below are only worth reserving when the pre-pass runs
			bool go = true;
			PGQ_TRY(meet_decide_alone(c, ws, n, d_src, meet_bytes, edge_bytes, &go));
			S.host_waits++;
			if (!go) {
				*ran = false;
				return PGQ_OK;
			}
			decide_mode = 0;
		}
		MeetPathsOut po;
		po.d_out_off = d_out_off;
		if (d_child_ext) {
			po.d_child = d_child_ext;
			po.child_cap = child_cap_ext;
		} else {
			// at most 9 elements per row (distance 4) — up to paths_reserve_mb; a call whose lists need more (over ~15 M rows at
			// the shipped 1 GB) has them written again into a buffer of the exact size (round-5 advisor finding: 72 bytes per row
			// reserved up front whatever the call)
			const size_t want = (size_t)std::max<int64_t>(n * 9, 1) * 8, lim = (size_t)std::max(0, mopt.paths_reserve_mb) << 20; // (0: 4 KB — the tests' way to the second emission)
			PGQ_TRY(ws->child.reserve(std::min(want, std::max<size_t>(lim, 4096))));
			po.d_child = ws->child.as<int64_t>();
			po.child_cap = (int64_t)(ws->child.cap / 8);
		}
		PGQ_HIP_TRY(hipMemsetAsync(d_out_off, 0, (size_t)n * 8, st));
		PGQ_TRY(meet_prepass(c, ws, n, d_src, d_dst, d_out_len, &nd, &po, decide_mode, meet_bytes, edge_bytes, ran, &observed_go));
		if (!*ran) return PGQ_OK;
		if (!d_child_ext && po.total > po.child_cap) { // (the kernel skipped the lists that did not fit)
			PGQ_TRY(ws->child.reserve((size_t)po.total * 8));
			po.d_child = ws->child.as<int64_t>();
			po.child_cap = (int64_t)(ws->child.cap / 8);
			PGQ_TRY(meet_reemit_paths(c, ws, n, d_src, d_dst, d_out_len, &po));
		}
		const int64_t total = po.total;
		WorkspaceLease inner;
		SearchOutput so2;
		if (nd > 0) {
			PGQ_TRY(ws->def_len.reserve((size_t)nd * 8));
			PGQ_TRY(ws->def_off.reserve((size_t)nd * 8));
			PGQ_TRY(inner.acquire());
			so2.depth = outp.depth + 1;
			so2.from_meet = true;
			S.pairs -= nd; // counted once
			PGQ_TRY(search_device(c, inner.ws, nd, ws->open_src, ws->open_dst,
			                      ws->def_len.as<int64_t>(), true, ws->def_off.as<int64_t>(), nullptr, 0, so2));
		}
		const int64_t need = total + so2.child_used;
		int64_t *d_child = po.d_child;
		if (d_child_ext) {
			if (need > child_cap_ext) outp.overflow = true;
		} else if ((size_t)need * 8 > ws->child.cap) { // the open rows' lists do not fit behind the others: a bigger buffer
			PGQ_TRY(grow_keeping(ws->child, (size_t)need * 8, (size_t)total * 8, st));
			d_child = ws->child.as<int64_t>();
		}
		if (nd > 0) {
			if (!outp.overflow && so2.child_used > 0)
				PGQ_HIP_TRY(hipMemcpyAsync(d_child + total, inner.ws->child.p, (size_t)so2.child_used * 8,
				                           hipMemcpyDeviceToDevice, st));
			PGQ_TRY(meet_apply_paths(ws, nd, ws->def_len.as<int64_t>(), ws->def_off.as<int64_t>(), total, d_out_len, d_out_off));
			PGQ_WAIT(st); // the inner workspace goes back to the pool after this
			KernelTimer::flush();
		}
		outp.child_used = need;
		if (outp.overflow)
			return fail(PGQ_ERR_INVALID_ARG, "child buffer too small: need " + std::to_string(need) + " elements");
		return PGQ_OK;
	};
	auto run_meet = [&](bool *ran) -> int {
		if (with_paths) return run_meet_paths(ran);
		u32 nd = 0;
		const double b0 = S.algo_bytes[K_MEET] + S.algo_bytes[K_MEET4] + S.algo_bytes[K_BIBFS];
		PGQ_TRY(meet_prepass(c, ws, n, d_src, d_dst, d_out_len, &nd, nullptr, decide_mode, meet_bytes, edge_bytes, ran, &observed_go,
		                     ball_mode, &ball_ran, &est_sources));
		if ((ball_asked == 1 || ball_asked == 3) && outp.ball_hint < 0 && !outp.no_memo) { // what the kernels said about these buffers
			std::lock_guard<std::mutex> g(c->plan_lock);
			c->route_memo.ball_n = n;
			c->route_memo.ball_src = d_src;
			c->route_memo.ball_dst = d_dst;
			c->route_memo.ball_yes = ball_ran;
		}
		if (ball_asked == 3 && !ball_ran) { // the kernels-alone chain declined these rows: the stage kernels after all
			*ran = true;
			PGQ_TRY(meet_prepass(c, ws, n, d_src, d_dst, d_out_len, &nd, nullptr, decide_mode, meet_bytes, edge_bytes, ran, &observed_go, 0, nullptr));
		}
		if (ball_ran && n >= 1024) {
			const double now = (double)nd / (double)n, old = c->ball_open_frac.load(std::memory_order_relaxed);
			c->ball_open_frac.store(0.5 * old + 0.5 * now, std::memory_order_relaxed);
		}
		if (ball_ran) {
			outp.route = 1;
			outp.source_runs = est_sources;
		}
		if (!*ran && !ball_ran) return PGQ_OK;
		*ran = true;
		if (n >= 1024 && !ball_ran) { // what these rows really moved refines the CSR's bytes per row (half the weight to the newest call)
			const double now = std::max(64.0, (S.algo_bytes[K_MEET] + S.algo_bytes[K_MEET4] + S.algo_bytes[K_BIBFS] - b0) / (double)n);
			const double old = c->meet_bpr.load(std::memory_order_relaxed);
			c->meet_bpr.store(old > 0 ? 0.5 * old + 0.5 * now : now, std::memory_order_relaxed);
		}
		if (nd > 0) {
			PGQ_TRY(ws->def_len.reserve((size_t)nd * 8));
			WorkspaceLease inner;
			PGQ_TRY(inner.acquire());
			SearchOutput so2;
			so2.depth = outp.depth + 1;
			// what the source-centric kernel left open (distance >= 5, unreachable, segments over its cap) is the pre-pass's kind
			// of row (k_meet4d / k_bibfs) before it is the lane batches'
			so2.from_meet = !ball_ran;
			so2.no_ball = true;
			S.pairs -= nd; // counted once
			PGQ_TRY(search_device(c, inner.ws, nd, ws->open_src, ws->open_dst,
			                      ws->def_len.as<int64_t>(), false, nullptr, nullptr, 0, so2));
			PGQ_TRY(meet_apply(ws, nd, ws->def_len.as<int64_t>(), d_out_len));
			PGQ_WAIT(st); // the inner workspace goes back to the pool after this
		}
		return PGQ_OK;
	};
	if (outp.bidir && !with_paths && !outp.want_te && outp.depth == 0 && c->E > 0) {
		u32 nd = 0;
		PGQ_TRY(meet_bidirectional(c, ws, n, d_src, d_dst, d_out_len, &nd));
		if (nd > 0) { // over k_bibfs's caps: the lane-batched search
			PGQ_TRY(ws->def_len.reserve((size_t)nd * 8));
			WorkspaceLease inner;
			PGQ_TRY(inner.acquire());
			SearchOutput so2;
			so2.depth = outp.depth + 1;
			so2.from_meet = true;
			S.pairs -= nd; // counted once
			PGQ_TRY(search_device(c, inner.ws, nd, ws->open_src, ws->open_dst,
			                      ws->def_len.as<int64_t>(), false, nullptr, nullptr, 0, so2));
			PGQ_TRY(meet_apply(ws, nd, ws->def_len.as<int64_t>(), d_out_len));
			PGQ_WAIT(st);
		}
		return PGQ_OK;
	}
	// Rows of few sources that are NOT grouped (a hash join's output order, a shuffled cross product): sorted by source — one
	// radix sort of (source, row) over log2 V bits, one gather — they are the source-centric kernel's input after all; its
	// answers (and those of the rows it leaves open) are scattered back by the sorted row index.  2.1 M rows: ~0.2 ms of
	// sorting + 0.3 ms of kernel against 2.2 ms through the lane batches.  *took = false: the device's byte rule declined.
	auto run_sorted_ball = [&](bool *took) -> int {
		*took = false;
		const int64_t V = c->V;
		for (DevBuf *b : { &ws->key, &ws->idx, &ws->skey, &ws->sidx }) PGQ_TRY(b->reserve((size_t)n * 4));
		for (DevBuf *b : { &ws->sort_src, &ws->sort_dst, &ws->sort_out }) PGQ_TRY(b->reserve((size_t)n * 8));
		int bits = 1;
		while (bits < 32 && (1ll << bits) <= V) bits++;
		{
			KernelTimer kt(st, K_PREP);
			hipLaunchKernelGGL(k_sort_keys, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, V, ws->key.as<u32>(), ws->idx.as<u32>());
			size_t stmp = 0;
			PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, stmp, ws->key.as<u32>(), ws->skey.as<u32>(), ws->idx.as<u32>(), ws->sidx.as<u32>(), (int)n, 0, bits, st));
			PGQ_TRY(ws->sort_tmp.reserve(stmp + 16));
			PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ws->sort_tmp.p, stmp, ws->key.as<u32>(), ws->skey.as<u32>(), ws->idx.as<u32>(), ws->sidx.as<u32>(), (int)n, 0, bits, st));
			hipLaunchKernelGGL(k_sort_gather, dim3(blocks_for(n)), dim3(256), 0, st, n, ws->sidx.as<u32>(), d_src, d_dst, ws->sort_src.as<int64_t>(),
			                   ws->sort_dst.as<int64_t>());
			kt.stop();
			// keys (8 B read, 8 written), the sort (a read and a write of 8-byte pairs per 8 key bits), the gather (4 + 16 read, 16 written)
			S.algo_bytes[K_PREP] += (double)n * (16.0 + 16.0 * ((bits + 7) / 8) + 36.0);
		}
		u32 nd = 0;
		bool ran = true, took_ball = false;
		double runs = -1.0;
		PGQ_TRY(meet_prepass(c, ws, n, ws->sort_src.as<int64_t>(), ws->sort_dst.as<int64_t>(), ws->sort_out.as<int64_t>(), &nd, nullptr, 0, meet_bytes,
		                     edge_bytes, &ran, nullptr, 3, &took_ball, &runs));
		if (!took_ball) return PGQ_OK;
		outp.route = 1;
		outp.source_runs = runs;
		if (nd > 0) { // what the kernel left open, in sorted positions: answered like run_meet's open rows, applied to the sorted output
			PGQ_TRY(ws->def_len.reserve((size_t)nd * 8));
			WorkspaceLease inner;
			PGQ_TRY(inner.acquire());
			SearchOutput so2;
			so2.depth = outp.depth + 1;
			so2.no_ball = true;
			so2.no_memo = true;
			S.pairs -= nd; // counted once
			PGQ_TRY(search_device(c, inner.ws, nd, ws->open_src, ws->open_dst, ws->def_len.as<int64_t>(), false, nullptr, nullptr, 0, so2));
			PGQ_TRY(meet_apply(ws, nd, ws->def_len.as<int64_t>(), ws->sort_out.as<int64_t>()));
		}
		{
			KernelTimer kt(st, K_PREP);
			hipLaunchKernelGGL(k_sort_scatter, dim3(blocks_for(n)), dim3(256), 0, st, n, ws->sidx.as<u32>(), ws->sort_out.as<int64_t>(), d_out_len);
			kt.stop();
			S.algo_bytes[K_PREP] += (double)n * 20.0;
		}
		PGQ_WAIT(st);
		KernelTimer::flush();
		*took = true;
		return PGQ_OK;
	};
	const bool sort_allowed = mopt.ball_sort && ball_possible && !outp.want_te && decide;
	bool sampled = false; // the sampled decision was asked for without the chain: read it after the next wait
	if (may_meet && meet_pays(std::min<int64_t>(n, c->V))) {
		if (sort_allowed && mopt.route_memo && !outp.no_memo) { // these buffers went through the sort last time: straight there
			bool again = false;
			{
				std::lock_guard<std::mutex> g(c->plan_lock);
				const pgq_csr::RouteMemo &m = c->route_memo;
				again = m.sorted_yes && m.n == n && m.src == (const void *)d_src && m.dst == (const void *)d_dst;
			}
			if (again) {
				bool took = false;
				PGQ_TRY(run_sorted_ball(&took));
				if (took) return PGQ_OK;
				std::lock_guard<std::mutex> g(c->plan_lock);
				c->route_memo.sorted_yes = false;
			}
		}
		bool skip = false;
		if (decide && mopt.route_memo && !outp.no_memo) {
			std::lock_guard<std::mutex> g(c->plan_lock);
			const pgq_csr::RouteMemo &m = c->route_memo;
			const bool same = m.n == n && m.src == (const void *)d_src && m.dst == (const void *)d_dst;
			skip = same && m.go == 0;
			if (same && m.go != 0) decide_mode = 2;
		}
		if (skip) {
			sampled = true; // taken inside the lane assignment's first launch (k_mark_sources)
		} else {
			bool ran = true;
			PGQ_TRY(run_meet(&ran));
			if (decide && !outp.no_memo) {
				std::lock_guard<std::mutex> g(c->plan_lock);
				c->route_memo.n = n;
				c->route_memo.src = d_src;
				c->route_memo.dst = d_dst;
				c->route_memo.go = ran ? 1 : 0;
				if (decide_mode == 2 && observed_go == 0 && !ball_ran) c->route_memo.go = 0; // these rows look like a cross product now: gated again next time
				c->route_memo.sorted_yes = false;
			}
			if (ran) return PGQ_OK;
			// neither the source-centric kernel (the rows are not grouped) nor the pre-pass (few distinct sources) took the call:
			// with at least 64 rows per source on average a sort by source makes it the former's
			if (sort_allowed && est_sources > 0 && est_sources * 64.0 <= (double)n) {
				bool took = false;
				PGQ_TRY(run_sorted_ball(&took));
				if (!outp.no_memo) {
					std::lock_guard<std::mutex> g(c->plan_lock);
					c->route_memo.sorted_yes = took;
				}
				if (took) return PGQ_OK;
			}
		}
	}
	u32 U = 0;
	// the accounting pass counts the full BFS of a pair even when dst has no in-edge, so it keeps those lanes
	SampleArgs sm { meet_bytes, edge_bytes, nullptr, nullptr, n, d_src, c->V };
	if (sampled) {
		PGQ_TRY(ws->route_dec.reserve(sizeof(MeetDecision)));
		sm.out = ws->route_dec.as<MeetDecision>();
		sm.h_go = reinterpret_cast<u32 *>(static_cast<char *>(ws->h_meet) + 4104);
		*sm.h_go = 0;
	}
	// Stage 2 ahead of the wait (round 5): when the last call with this row count on this CSR was a ONE-batch call of width
	// `ahead_wd`, its two kernels that need nothing from the host — the rows' lane ids (k_pair_rows reads the ranks on the
	// device) and the batch's start (k_batch_reset) — are enqueued BEFORE the host waits for the source count: the GPU used to
	// idle through that wait and its wake-up (~15 us) and then run them (48 us on the 2.1 M-row cross product).  If the count
	// says otherwise afterwards (more than one batch, another width) the normal path overwrites / redoes both: same answers.
	int ahead_wd = 0;
	const bool may_stay_in_place = !with_paths && !outp.want_te && mopt.sort_single_batch == 0;
	if (may_stay_in_place && mopt.stage2_ahead) {
		std::lock_guard<std::mutex> g(c->plan_lock);
		if (c->route_memo.id_n == n) ahead_wd = c->route_memo.id_wd;
	}
	bool rows_ahead = false;
	auto pre_wait = [&]() -> int {
		KernelTimer kt(st, K_PREP);
		hipLaunchKernelGGL(k_pair_rows, dim3(blocks_for(n)), dim3(256), 0, st, n, d_src, d_dst, ws->rank.as<u32>(), c->off, c->roff,
		                   outp.want_te ? 0 : 1, c->V, ws->skey.as<u32>(), ws->ssrc.as<int32_t>(), ws->sdst.as<int32_t>(), ws->sres.as<int32_t>());
		kt.stop();
		S.algo_bytes[K_PREP] += (double)n * 44.0;
		rows_ahead = true;
		PGQ_TRY(batch_state_reset(ws, c->V, ahead_wd));
		ws->prereset_V = c->V;
		ws->prereset_WD = ahead_wd;
		return PGQ_OK;
	};
	ws->prereset_V = -1;
	PGQ_TRY(lane_ranks(c, ws, n, d_src, d_dst, &U, !outp.want_te, sm, ahead_wd > 0 ? std::function<int()>(pre_wait) : std::function<int()>()));
	if (sampled) { // (lane_ranks has waited for the stream) what the sample says about these rows decides the next call's route
		const u32 v = *reinterpret_cast<const u32 *>(static_cast<const char *>(ws->h_meet) + 4104);
		if (v == 2) {
			std::lock_guard<std::mutex> g(c->plan_lock);
			c->route_memo.go = 1;
		}
	}
	S.unique_sources += U;
	if (with_paths) PGQ_HIP_TRY(hipMemsetAsync(ws->soff.p, 0, (size_t)n * 8, st));
	const int wd = choose_words(U);
	const int64_t Lb = 64 * (int64_t)wd;
	const int nb = (int)((U + Lb - 1) / Lb);
	bool identity = false; // the rows were left in the caller's order (one batch): no permutation to undo
	PGQ_TRY(lane_rows_bfs(c, ws, n, d_src, d_dst, !outp.want_te, Lb, nb, may_stay_in_place, &identity, rows_ahead));
	if (may_stay_in_place) { // what the next call with this row count may do ahead of its wait
		std::lock_guard<std::mutex> g(c->plan_lock);
		c->route_memo.id_n = identity ? n : -1;
		c->route_memo.id_wd = wd;
	}
	auto run = [&](Workspace *priv, int b0, int bstride, SearchOutput &o) -> int {
		switch (wd) {
		case 1: return run_batches<1>(c, ws, priv, b0, bstride, n, U, with_paths, d_child_ext, child_cap_ext, o);
		case 2: return run_batches<2>(c, ws, priv, b0, bstride, n, U, with_paths, d_child_ext, child_cap_ext, o);
		case 4: return run_batches<4>(c, ws, priv, b0, bstride, n, U, with_paths, d_child_ext, child_cap_ext, o);
		case 8: return run_batches<8>(c, ws, priv, b0, bstride, n, U, with_paths, d_child_ext, child_cap_ext, o);
		case 16: return run_batches<16>(c, ws, priv, b0, bstride, n, U, with_paths, d_child_ext, child_cap_ext, o);
		default: return run_batches<32>(c, ws, priv, b0, bstride, n, U, with_paths, d_child_ext, child_cap_ext, o);
		}
	};
	// Independent batches overlap on several streams (one host thread each): hides the per-level host round trip
	// and fills the GPU during the short levels.  Paths and the accounting pass keep a single ordered worker.
	int workers = std::max(1, std::min(options().streams, nb));
	if (with_paths || outp.want_te) workers = 1;
	int rc = PGQ_OK;
	if (workers == 1) {
		rc = run(ws, 0, 1, outp);
	} else {
		std::vector<WorkspaceLease> leases((size_t)workers - 1);
		for (auto &l : leases) PGQ_TRY(l.acquire());
		std::vector<int> rcs((size_t)workers, PGQ_OK);
		std::vector<std::string> errs((size_t)workers);
		std::vector<SearchOutput> outs((size_t)workers, outp);
		std::vector<pgq_stats_t> wstats((size_t)workers);
		std::vector<std::shared_ptr<WorkerTask>> pool;
		const int dev = current_device();
		Options *const parent_opt = options_override();
		for (int t = 1; t < workers; t++)
			pool.push_back(worker_submit(dev, [&, t]() {
				OptionScope opt_scope(parent_opt); // the handle's own options, if the call runs under them
				bind_thread_device(dev); // the caller's device (a multi-GPU shard may not be on the default one)
				int r = ensure_init(); // binds the device for this host thread
				if (r == PGQ_OK) {
					(void)pgq_reset_stats();
					r = run(leases[(size_t)t - 1].ws, t, workers, outs[(size_t)t]);
				}
				rcs[(size_t)t] = r;
				if (r != PGQ_OK) errs[(size_t)t] = pgq_last_error();
				wstats[(size_t)t] = tstats().s;
			}));
		rcs[0] = run(ws, 0, workers, outs[0]);
		for (size_t k = 0; k < pool.size(); k++) { // a job that threw never wrote its return code: take the pool's word for it
			const int wr = worker_wait(pool[k]);
			if (wr != PGQ_OK) {
				rcs[k + 1] = wr;
				errs[k + 1] = pgq_last_error();
			}
		}
		for (int t = 0; t < workers; t++) {
			if (rcs[(size_t)t] != PGQ_OK && rc == PGQ_OK) {
				rc = rcs[(size_t)t];
				if (t > 0) set_error(errs[(size_t)t]);
			}
			outp.deferred = outp.deferred || outs[(size_t)t].deferred;
			if (t > 0) merge_stats(S, wstats[(size_t)t]);
		}
	}
	if (rc == PGQ_OK && outp.deferred) {
		// second, narrow pass over the stragglers
		PGQ_TRY(ws->def_src.reserve((size_t)n * 8));
		PGQ_TRY(ws->def_dst.reserve((size_t)n * 8));
		PGQ_TRY(ws->def_len.reserve((size_t)n * 8));
		PGQ_TRY(ws->def_idx.reserve((size_t)n * 4));
		u32 *d_count = reinterpret_cast<u32 *>(ws->counters.p);
		PGQ_HIP_TRY(hipMemsetAsync(d_count, 0, 4, st));
		hipLaunchKernelGGL(k_collect_deferred, dim3(blocks_for(n)), dim3(256), 0, st, n, ws->sres.as<int32_t>(),
		                   ws->ssrc.as<int32_t>(), ws->sdst.as<int32_t>(), ws->def_src.as<int64_t>(),
		                   ws->def_dst.as<int64_t>(), ws->def_idx.as<u32>(), d_count);
		u32 nd = 0;
		PGQ_HIP_TRY(hipMemcpyAsync(&nd, d_count, 4, hipMemcpyDeviceToHost, st));
		PGQ_WAIT(st);
		if (nd > 0) {
			WorkspaceLease inner;
			PGQ_TRY(inner.acquire());
			SearchOutput so2;
			so2.depth = outp.depth + 1;
			S.pairs -= nd; // counted once
			if (!with_paths) {
				PGQ_TRY(search_device(c, inner.ws, nd, ws->def_src.as<int64_t>(), ws->def_dst.as<int64_t>(),
				                      ws->def_len.as<int64_t>(), false, nullptr, nullptr, 0, so2));
				hipLaunchKernelGGL(k_apply_deferred, dim3(blocks_for(nd)), dim3(256), 0, st, (int64_t)nd,
				                   ws->def_idx.as<u32>(), ws->def_len.as<int64_t>(), ws->sres.as<int32_t>(), nullptr,
				                   nullptr, (int64_t)0);
			} else {
				// the stragglers' paths land in the inner workspace's child buffer; append them to ours
				PGQ_TRY(ws->def_off.reserve((size_t)nd * 8));
				PGQ_TRY(search_device(c, inner.ws, nd, ws->def_src.as<int64_t>(), ws->def_dst.as<int64_t>(),
				                      ws->def_len.as<int64_t>(), true, ws->def_off.as<int64_t>(), nullptr, 0, so2));
				const int64_t base = outp.child_used, need = base + so2.child_used;
				int64_t *d_child = d_child_ext;
				if (d_child_ext) {
					if (need > child_cap_ext) {
						outp.child_used = need;
						outp.overflow = true;
					}
				} else {
					if ((size_t)need * 8 > ws->child.cap) {
						PGQ_TRY(grow_keeping(ws->child, (size_t)need * 8, (size_t)base * 8, st));
					}
					d_child = ws->child.as<int64_t>();
				}
				if (!outp.overflow) {
					if (so2.child_used > 0)
						PGQ_HIP_TRY(hipMemcpyAsync(d_child + base, inner.ws->child.p, (size_t)so2.child_used * 8,
						                           hipMemcpyDeviceToDevice, st));
					hipLaunchKernelGGL(k_apply_deferred, dim3(blocks_for(nd)), dim3(256), 0, st, (int64_t)nd,
					                   ws->def_idx.as<u32>(), ws->def_len.as<int64_t>(), ws->sres.as<int32_t>(),
					                   ws->def_off.as<int64_t>(), ws->soff.as<int64_t>(), base);
					PGQ_WAIT(st); // the inner workspace goes back to the pool after this
					outp.child_used = need;
				} else { // lengths of the stragglers are still reported
					hipLaunchKernelGGL(k_apply_deferred, dim3(blocks_for(nd)), dim3(256), 0, st, (int64_t)nd,
					                   ws->def_idx.as<u32>(), ws->def_len.as<int64_t>(), ws->sres.as<int32_t>(), nullptr,
					                   nullptr, (int64_t)0);
					outp.child_used = need;
				}
			}
		}
	}
	// results back to row order even when the child buffer overflowed (lengths are still right)
	KernelTimer kts(st, K_PREP);
	S.algo_bytes[K_PREP] += (double)n * (identity ? 12.0 : 16.0); // result word (+ permutation) read, 8-byte length written
	hipLaunchKernelGGL(k_scatter_results, dim3(blocks_for(n)), dim3(256), 0, st, n, identity ? nullptr : ws->sidx.as<u32>(), ws->sres.as<int32_t>(),
	                   ws->soff.as<int64_t>(), d_out_len, with_paths ? d_out_off : nullptr);
	kts.stop();
	PGQ_WAIT(st);
	KernelTimer::flush();
	if (rc == PGQ_OK && outp.overflow)
		rc = fail(PGQ_ERR_INVALID_ARG, "child buffer too small: need " + std::to_string(outp.child_used) + " elements");
	return rc;
}

static int check_csr(pgq_csr_t *csr, int64_t V) {
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "Constraint Error: Need to initialize CSR before doing shortest path");
	if (V != csr->V) return fail(PGQ_ERR_INVALID_ARG, "V does not match the uploaded CSR");
	return PGQ_OK;
}

} // namespace pgq

using namespace pgq;

// per-thread arena for list payloads returned by the chunk API
static thread_local std::vector<int64_t> t_child;

// body(k, lo, hi, replica, ws): shard k = rows [lo, hi) on device k's replica, called on a thread bound to that device
template <typename Body>
static int run_shards(pgq_csr_t *csr, int64_t n, Body body) {
	PGQ_TRY(pgq_csr_replicate(csr));
	std::vector<int> devs;
	std::vector<pgq_csr *> replicas;
	{ // a snapshot: another caller may rebuild the list for a new device set meanwhile (old replicas stay alive)
		std::lock_guard<std::mutex> g(csr->replica_lock);
		devs = csr->replica_devices;
		replicas = csr->replicas;
	}
	const int W = (int)devs.size();
	if (W == 0 || replicas.size() != (size_t)W) return fail(PGQ_ERR_INVALID_ARG, "CSR replicas do not match the enabled devices");
	for (int k = 0; k < W; k++)
		if (!replicas[(size_t)k] || replicas[(size_t)k]->device != devs[(size_t)k])
			return fail(PGQ_ERR_INVALID_ARG, "CSR replica on the wrong device");
	const int64_t per = (n + W - 1) / W;
	std::vector<int> rcs((size_t)W, PGQ_OK);
	std::vector<std::string> errs((size_t)W);
	std::vector<pgq_stats_t> wstats((size_t)W);
	auto shard = [&](int k) -> int {
		const int64_t lo = std::min<int64_t>((int64_t)k * per, n), hi = std::min<int64_t>(lo + per, n);
		if (hi == lo) return PGQ_OK;
		bind_thread_device(devs[(size_t)k]);
		PGQ_TRY(ensure_init());
		WorkspaceLease lease;
		PGQ_TRY(lease.acquire());
		return body(k, lo, hi, replicas[(size_t)k], lease.ws);
	};
	std::vector<std::shared_ptr<WorkerTask>> pool;
	Options *const parent_opt = options_override();
	for (int k = 1; k < W; k++)
		pool.push_back(worker_submit(devs[(size_t)k], [&, k]() {
			OptionScope opt_scope(parent_opt);
			(void)pgq_reset_stats();
			rcs[(size_t)k] = shard(k);
			if (rcs[(size_t)k] != PGQ_OK) errs[(size_t)k] = pgq_last_error();
			wstats[(size_t)k] = tstats().s;
		}));
	rcs[0] = shard(0);
	bind_thread_device(-1);
	(void)ensure_init();
	for (size_t k = 0; k < pool.size(); k++) { // a job that threw never wrote its return code: take the pool's word for it
		const int wr = worker_wait(pool[k]);
		if (wr != PGQ_OK) {
			rcs[k + 1] = wr;
			errs[k + 1] = pgq_last_error();
		}
	}
	int rc = PGQ_OK;
	for (int k = 0; k < W; k++) {
		if (rcs[(size_t)k] != PGQ_OK && rc == PGQ_OK) {
			rc = rcs[(size_t)k];
			if (k > 0) set_error(errs[(size_t)k]);
		}
		if (k > 0) merge_stats(tstats().s, wstats[(size_t)k]);
	}
	return rc;
}

extern "C" {

void pgq_thread_release(void) {
	t_child.clear();
	t_child.shrink_to_fit();
}

int pgq_release_cached_memory(void) {
	PGQ_TRY(ensure_init());
	drop_idle_workspaces();
	dev_cache_trim(); // and the freed CSR / upload blocks kept for the next upload
	return PGQ_OK;
}

static int iterativelength_bulk(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst, int64_t *d_out_len,
                                bool bidir) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!d_src || !d_dst || !d_out_len))) return fail(PGQ_ERR_INVALID_ARG, "NULL device array");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	SearchOutput so;
	so.bidir = bidir;
	return search_device(csr, lease.ws, n, d_src, d_dst, d_out_len, false, nullptr, nullptr, 0, so);
}
int pgq_iterativelength_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                    int64_t *d_out_len) {
	OptionScope opt_scope(csr);
	return iterativelength_bulk(csr, n, d_src, d_dst, d_out_len, false);
}
int pgq_iterativelength_bidirectional_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                                  int64_t *d_out_len) {
	OptionScope opt_scope(csr);
	return iterativelength_bulk(csr, n, d_src, d_dst, d_out_len, true);
}
int pgq_traversed_edges_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                    int64_t *d_out_len, int64_t *d_out_te) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!d_src || !d_dst || !d_out_len || !d_out_te))) return fail(PGQ_ERR_INVALID_ARG, "NULL device array");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	SearchOutput so;
	so.want_te = true;
	PGQ_TRY(search_device(csr, lease.ws, n, d_src, d_dst, d_out_len, false, nullptr, nullptr, 0, so));
	if (n > 0) {
		hipLaunchKernelGGL(k_scatter_te, dim3(blocks_for(n)), dim3(256), 0, lease.ws->stream, n, lease.ws->sidx.as<u32>(),
		                   lease.ws->ste.as<int64_t>(), d_out_te);
		PGQ_HIP_TRY(hipStreamSynchronize(lease.ws->stream));
	}
	return PGQ_OK;
}

int pgq_shortestpath_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                 int64_t *d_out_len, int64_t *d_out_offset, int64_t *d_child, int64_t child_cap,
                                 int64_t *child_used) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!d_src || !d_dst || !d_out_len || !d_out_offset || !d_child)))
		return fail(PGQ_ERR_INVALID_ARG, "NULL device array");
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	SearchOutput so;
	int rc = search_device(csr, lease.ws, n, d_src, d_dst, d_out_len, true, d_out_offset, d_child, child_cap, so);
	if (child_used) *child_used = so.child_used;
	return rc;
}

// Multi-GPU inside one process (the single DuckDB process the boundary targets): rows are cut into contiguous shards,
// one host thread per enabled device runs the identical single-GPU path on its shard against that device's replica of
// the CSR, and the results land in the caller's host array (the gather).  No collective inside the search
// (results are a pure function of (CSR, src, dst)).
int pgq_iterativelength_multi(pgq_csr_t *csr, int64_t n, const int64_t *src, const int64_t *dst, int64_t *out_len) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!src || !dst || !out_len))) return fail(PGQ_ERR_INVALID_ARG, "NULL array");
	if (n == 0) return PGQ_OK;
	return run_shards(csr, n, [&](int, int64_t lo, int64_t hi, pgq_csr_t *replica, Workspace *ws) -> int {
		const size_t bytes = (size_t)(hi - lo) * 8;
		PGQ_TRY(ws->in_src.reserve(bytes));
		PGQ_TRY(ws->in_dst.reserve(bytes));
		PGQ_TRY(ws->out_len.reserve(bytes));
		PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, src + lo, bytes, hipMemcpyHostToDevice, ws->stream));
		PGQ_HIP_TRY(hipMemcpyAsync(ws->in_dst.p, dst + lo, bytes, hipMemcpyHostToDevice, ws->stream));
		SearchOutput so;
		PGQ_TRY(search_device(replica, ws, hi - lo, ws->in_src.as<int64_t>(), ws->in_dst.as<int64_t>(),
		                      ws->out_len.as<int64_t>(), false, nullptr, nullptr, 0, so));
		return staged_download(out_len + lo, ws->out_len.p, bytes, ws->stream);
	});
}

// shortestpath on all enabled devices: every shard writes its lists into its own device buffer (grown once if the first
// guess was too small), the payloads are then concatenated in shard order into `child` and the list offsets shifted by
// the preceding shards' sizes — the gather of the ragged [v,e,v,...] lists.
int pgq_shortestpath_multi(pgq_csr_t *csr, int64_t n, const int64_t *src, const int64_t *dst, int64_t *out_len,
                           int64_t *out_offset, int64_t *child, int64_t child_cap, int64_t *child_used) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!src || !dst || !out_len || !out_offset)) || child_cap < 0 || (child_cap > 0 && !child))
		return fail(PGQ_ERR_INVALID_ARG, "NULL array");
	if (child_used) *child_used = 0;
	if (n == 0) return PGQ_OK;
	const size_t W = enabled_devices().size();
	std::vector<std::vector<int64_t>> payload(W);
	std::vector<int64_t> shard_lo(W, 0), shard_hi(W, 0);
	PGQ_TRY(run_shards(csr, n, [&](int k, int64_t lo, int64_t hi, pgq_csr_t *replica, Workspace *ws) -> int {
		const int64_t m = hi - lo;
		const size_t bytes = (size_t)m * 8;
		PGQ_TRY(ws->in_src.reserve(bytes));
		PGQ_TRY(ws->in_dst.reserve(bytes));
		PGQ_TRY(ws->out_len.reserve(bytes));
		PGQ_TRY(ws->out_off.reserve(bytes));
		PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, src + lo, bytes, hipMemcpyHostToDevice, ws->stream));
		PGQ_HIP_TRY(hipMemcpyAsync(ws->in_dst.p, dst + lo, bytes, hipMemcpyHostToDevice, ws->stream));
		DevBuf dchild; // not a workspace buffer: search_device uses ws->child for its own staging
		int64_t cap = std::max<int64_t>(16 * m, 1024), used = 0;
		int rc = PGQ_OK;
		for (int attempt = 0; attempt < 2; attempt++) {
			rc = dchild.reserve((size_t)cap * 8);
			if (rc != PGQ_OK) break;
			SearchOutput so;
			rc = search_device(replica, ws, m, ws->in_src.as<int64_t>(), ws->in_dst.as<int64_t>(), ws->out_len.as<int64_t>(),
			                   true, ws->out_off.as<int64_t>(), dchild.as<int64_t>(), cap, so);
			used = so.child_used;
			if (rc == PGQ_OK || used <= cap) break;
			cap = used; // too small: the search reported what it needs
		}
		if (rc == PGQ_OK) rc = staged_download(out_len + lo, ws->out_len.p, bytes, ws->stream);
		if (rc == PGQ_OK) rc = staged_download(out_offset + lo, ws->out_off.p, bytes, ws->stream);
		if (rc == PGQ_OK) {
			payload[(size_t)k].resize((size_t)used);
			if (used > 0) rc = staged_download(payload[(size_t)k].data(), dchild.p, (size_t)used * 8, ws->stream);
		}
		shard_lo[(size_t)k] = lo;
		shard_hi[(size_t)k] = hi;
		dchild.release();
		return rc;
	}));
	int64_t total = 0;
	for (size_t k = 0; k < W; k++) total += (int64_t)payload[k].size();
	if (child_used) *child_used = total;
	if (total > child_cap) return fail(PGQ_ERR_INVALID_ARG, "child buffer too small for the path lists (see *child_used)");
	int64_t base = 0;
	for (size_t k = 0; k < W; k++) {
		if (!payload[k].empty()) memcpy(child + base, payload[k].data(), payload[k].size() * 8);
		if (base)
			for (int64_t i = shard_lo[k]; i < shard_hi[k]; i++)
				if (out_len[i] >= 0) out_offset[i] += base;
		base += (int64_t)payload[k].size();
	}
	return PGQ_OK;
}

// cheapest_path_length on all enabled devices: out = n values (int64 or double by the CSR's weight type), out_valid = n
// bytes (1 = a path exists)
int pgq_cheapest_path_length_multi(pgq_csr_t *csr, int64_t n, const int64_t *src, const int64_t *dst, void *out,
                                   uint8_t *out_valid) {
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	if (n < 0 || (n > 0 && (!src || !dst || !out || !out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL array");
	if (n == 0) return PGQ_OK;
	return run_shards(csr, n, [&](int, int64_t lo, int64_t hi, pgq_csr_t *replica, Workspace *ws) -> int {
		const int64_t m = hi - lo;
		const size_t bytes = (size_t)m * 8;
		DevBuf d_src, d_dst, d_val, d_ok; // the bulk entry point leases its own workspace
		int rc = d_src.reserve(bytes);
		if (rc == PGQ_OK) rc = d_dst.reserve(bytes);
		if (rc == PGQ_OK) rc = d_val.reserve(bytes);
		if (rc == PGQ_OK) rc = d_ok.reserve((size_t)m);
		if (rc == PGQ_OK && (hipMemcpyAsync(d_src.p, src + lo, bytes, hipMemcpyHostToDevice, ws->stream) != hipSuccess ||
		                     hipMemcpyAsync(d_dst.p, dst + lo, bytes, hipMemcpyHostToDevice, ws->stream) != hipSuccess ||
		                     hipStreamSynchronize(ws->stream) != hipSuccess))
			rc = fail(PGQ_ERR_HIP, "copying a shard's rows to its device failed");
		if (rc == PGQ_OK)
			rc = pgq_cheapest_path_length_bulk_device(replica, m, d_src.as<int64_t>(), d_dst.as<int64_t>(), d_val.p,
			                                          d_ok.as<uint8_t>());
		if (rc == PGQ_OK) rc = staged_download(static_cast<char *>(out) + (size_t)lo * 8, d_val.p, bytes, ws->stream);
		if (rc == PGQ_OK) rc = staged_download(out_valid + lo, d_ok.p, (size_t)m, ws->stream);
		for (DevBuf *b : { &d_src, &d_dst, &d_val, &d_ok }) b->release();
		return rc;
	});
}

// pinned, device-addressable staging block of a workspace (grown on demand)
static int io_block(Workspace *ws, size_t bytes, void **host, void **dev) {
	if (ws->h_io_cap < bytes) {
		if (ws->h_io) (void)hipHostFree(ws->h_io);
		ws->h_io = nullptr;
		ws->h_io_cap = 0;
		const size_t want = std::max<size_t>(bytes * 2, 64 << 10);
		PGQ_HIP_TRY(hipHostMalloc(&ws->h_io, want));
		ws->h_io_cap = want;
	}
	*host = ws->h_io;
	PGQ_HIP_TRY(hipHostGetDevicePointer(dev, ws->h_io, 0));
	return PGQ_OK;
}

static int iterativelength_chunk(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, int64_t *out_len,
                                 uint64_t *out_valid, bool bidir) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	PGQ_TRY(check_csr(csr, V));
	if (n < 0 || (n > 0 && (!out_len || !out_valid))) return fail(PGQ_ERR_INVALID_ARG, "NULL output");
	if (n == 0) return PGQ_OK;
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *ws = lease.ws;
	if (!bidir && options().chunk_zero_copy && prepass_takes(csr, n, SearchOutput()) && n <= kMeetDecideRows) {
		// One DuckDB chunk through the pair-centric kernels: they read the rows straight out of a pinned staging block and
		// write the hop counts straight back into it (2048 rows = 32 KB in, 16 KB out over PCIe, one access per row), so the
		// call is two or three kernel launches and ONE wait — no copy commands (each is a stream operation of its own:
		// two in, two out cost more than the search of a chunk).
		void *hp = nullptr, *dp = nullptr;
		PGQ_TRY(io_block(ws, (size_t)n * 24, &hp, &dp));
		int64_t *h = static_cast<int64_t *>(hp), *d = static_cast<int64_t *>(dp);
		PGQ_TRY(flatten_pairs_into(V, n, src, dst, h, h + n));
		SearchOutput so;
		so.no_memo = true;
		{ // the rows are in host memory: whether they are grouped by source costs a pass over 2048 words here, two launches there
			int64_t runs = 1;
			for (int64_t i = 1; i < n; i++) runs += h[i] != h[i - 1];
			so.ball_hint = runs * 8 <= n ? 1 : 0;
		}
		PGQ_TRY(search_device(csr, ws, n, d, d + n, d + 2 * n, false, nullptr, nullptr, 0, so));
		const int64_t *res = h + 2 * n;
		for (int64_t w = 0; w < (n + 63) / 64; w++) { // payload of a NULL row stays -1 like iterativelength.cpp:100,137
			uint64_t m = 0;
			const int64_t lo = w * 64, cnt = std::min<int64_t>(64, n - lo);
			for (int64_t k = 0; k < cnt; k++) {
				const int64_t v = res[lo + k];
				out_len[lo + k] = v;
				m |= (uint64_t)(v >= 0) << k;
			}
			out_valid[w] = cnt == 64 ? m : (m | (~0ULL << cnt)); // bits past n stay set, as mask_fill_valid leaves them
		}
		return PGQ_OK;
	}
	FlatPairs fp;
	PGQ_TRY(flatten_pairs(V, n, src, dst, fp, false));
	PGQ_TRY(ws->in_src.reserve((size_t)n * 8));
	PGQ_TRY(ws->in_dst.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_len.reserve((size_t)n * 8));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, fp.src.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_dst.p, fp.dst.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	SearchOutput so;
	so.bidir = bidir;
	so.no_memo = true;
	PGQ_TRY(search_device(csr, ws, n, ws->in_src.as<int64_t>(), ws->in_dst.as<int64_t>(), ws->out_len.as<int64_t>(),
	                      false, nullptr, nullptr, 0, so));
	PGQ_TRY(staged_download(out_len, ws->out_len.p, (size_t)n * 8, ws->stream));
	mask_fill_valid(out_valid, n);
	for (int64_t i = 0; i < n; i++)
		if (out_len[i] < 0) mask_set_invalid(out_valid, i); // payload stays -1 like iterativelength.cpp:100,137
	return PGQ_OK;
}
int pgq_iterativelength(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, int64_t *out_len,
                        uint64_t *out_valid) {
	OptionScope opt_scope(csr);
	return iterativelength_chunk(csr, V, n, src, dst, out_len, out_valid, false);
}
int pgq_iterativelength_bidirectional(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, int64_t *out_len,
                                      uint64_t *out_valid) {
	OptionScope opt_scope(csr);
	return iterativelength_chunk(csr, V, n, src, dst, out_len, out_valid, true);
}

int pgq_shortestpath(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, uint64_t *out_offset,
                     uint64_t *out_length, uint64_t *out_valid, const int64_t **out_child, uint64_t *out_child_len) {
	CallScope in_flight;
	OptionScope opt_scope(csr);
	PGQ_TRY(ensure_init());
	PGQ_TRY(check_csr(csr, V));
	if (n < 0 || (n > 0 && (!out_offset || !out_length || !out_valid)) || !out_child || !out_child_len)
		return fail(PGQ_ERR_INVALID_ARG, "NULL output");
	*out_child = nullptr;
	*out_child_len = 0;
	if (n == 0) return PGQ_OK;
	FlatPairs fp;
	PGQ_TRY(flatten_pairs(V, n, src, dst, fp, false));
	WorkspaceLease lease;
	PGQ_TRY(lease.acquire());
	Workspace *ws = lease.ws;
	PGQ_TRY(ws->in_src.reserve((size_t)n * 8));
	PGQ_TRY(ws->in_dst.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_len.reserve((size_t)n * 8));
	PGQ_TRY(ws->out_off.reserve((size_t)n * 8));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_src.p, fp.src.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	PGQ_HIP_TRY(hipMemcpyAsync(ws->in_dst.p, fp.dst.data(), (size_t)n * 8, hipMemcpyHostToDevice, ws->stream));
	SearchOutput so;
	so.no_memo = true;
	PGQ_TRY(search_device(csr, ws, n, ws->in_src.as<int64_t>(), ws->in_dst.as<int64_t>(), ws->out_len.as<int64_t>(),
	                      true, ws->out_off.as<int64_t>(), nullptr, 0, so));
	std::vector<int64_t> len(n), off(n);
	PGQ_TRY(staged_download(len.data(), ws->out_len.p, (size_t)n * 8, ws->stream));
	PGQ_TRY(staged_download(off.data(), ws->out_off.p, (size_t)n * 8, ws->stream));
	t_child.resize((size_t)so.child_used);
	if (so.child_used > 0)
		PGQ_TRY(staged_download(t_child.data(), ws->child.p, (size_t)so.child_used * 8, ws->stream));
	mask_fill_valid(out_valid, n);
	for (int64_t i = 0; i < n; i++) {
		if (len[i] < 0) {
			mask_set_invalid(out_valid, i);
			out_offset[i] = 0;
			out_length[i] = 0;
		} else {
			out_offset[i] = (uint64_t)off[i];
			out_length[i] = (uint64_t)(2 * len[i] + 1);
		}
	}
	*out_child = t_child.data();
	*out_child_len = (uint64_t)so.child_used;
	return PGQ_OK;
}

} // extern "C"
