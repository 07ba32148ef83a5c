// pgq_search.h — per-call workspace and lane assignment shared by the BFS and cheapest-path drivers
#pragma once
#include <cmath>
#include <memory>

#include "pgq_internal.h"

namespace pgq {

static constexpr u32 kNoLane = 0xFFFFFFFFu;   // NULL rows
static constexpr u32 kTrivial = 0xFFFFFFFEu;  // src == dst rows (no search needed)
static constexpr int kSpecLevels = 61;        // levels of a batch enqueued ahead of the host at most (the pinned log holds 64 entries)

struct Counters {
	u32 q_count[2];      // work items in the frontier queues (by level parity)
	u32 front_vertices;  // vertices that received fresh bits this level
	u32 unresolved;      // pairs of the batch still without an answer
	u64 front_edges;     // sum of their out-degrees
	u64 edges_scanned;   // adjacency entries read this level
	u64 word_gathers;    // 8-byte lane-words gathered or RMW'd this level
	u32 front_words;     // non-empty lane-words of the frontier just produced
	u32 pad;             // packed frontier: records written ...
	u32 pad2;            // ... and overflow words written (k_compact_frontier totals)
	u32 done;            // levels enqueued ahead (spec_levels): set by k_level_reset when the batch is over (1), when the level's
	                     // plan is not what the counters call for (2) or when the plan has run out (3); every level kernel returns at once
	u32 r1[4];
	u64 act[2][32];      // active-lane masks (lanes that still have open pairs), double buffered
};

#ifdef __HIPCC__
// A level kernel has nothing to do when the batch is over on the device's own account (`done`) or when the probe before
// it left at most `stop_limit` rows open (the host defers them).
__device__ __forceinline__ bool level_is_off(const Counters *__restrict__ cnt, int stop_limit) {
	return cnt->done != 0 || (stop_limit >= 0 && cnt->unresolved <= (u32)stop_limit);
}
#endif

// bytes (at streaming rate) the lane-batched search is priced at for `distinct` sources: full 2048-lane batches of 32
// lane-words plus one narrower batch for the rest, E x (12 + 3 wd) each; `edge_bytes` = meet_bias x E.  Shared by the
// host-side decision (few rows) and k_meet_decide (sampled distinct sources).
// Round 5: `rows` / `V` add the DENSE bottom-up level, E x (8 + 6 wd), to every batch that holds more than V / 8 rows — with
// that many open rows the probes do not replace it (2048 sources x 128 .. 1024 destinations on the SF100-shaped graph: 2.1 -
// 2.3 ms per batch against 0.73 ms at x 32, where the probes do; priced without it a 2048 x 256 product went through the
// lanes at 2.05 ms where the pre-pass takes 1.16).
__host__ __device__ static inline double lanes_cost_bytes(double edge_bytes, double distinct, double rows, double V) {
	const double full = floor(distinct / 2048.0), rest = distinct - full * 2048.0;
	double wd = 0.0;
	if (rest > 0.0) {
		wd = 1.0;
		while (wd * 64.0 < rest) wd *= 2.0;
	}
	const double batches = full + (rest > 0.0 ? 1.0 : 0.0);
	const bool dense = batches > 0.0 && rows / batches > V / 8.0;
	return edge_bytes * (full * (12.0 + 3.0 * 32.0 + (dense ? 8.0 + 6.0 * 32.0 : 0.0)) +
	                     (rest > 0.0 ? 12.0 + 3.0 * wd + (dense ? 8.0 + 6.0 * wd : 0.0) : 0.0));
}

// ---- how many distinct sources? (the pre-pass's sampled decision; pgq_meet.hip and the lane assignment both run it) ----
struct MeetDecision {
	u32 go;           // 1: the pre-pass runs
	u32 sample_rows;  // non-NULL rows sampled
	u32 sample_fresh; // distinct sources among them
	u32 pad;
	double estimate;  // distinct sources of the whole input
};
#ifdef __HIPCC__
constexpr u32 kSampleEmpty = 0xFFFFFFFFu;
constexpr int kSampleRows = 2048, kSampleSlots = 4096;
// Every thread of ONE workgroup (any multiple of 64 threads).  s_set: kSampleSlots words of LDS the caller lends (its own
// array, or a map it is not using yet).  h_go (nullable): pinned host word that gets the verdict + 1.
__device__ __forceinline__ void sample_distinct_sources(int64_t n, const int64_t *__restrict__ src, int64_t V, double meet_bytes,
                                                        double edge_bytes, MeetDecision *__restrict__ out, u32 *__restrict__ h_go,
                                                        u32 *s_set) {
	const int nt = (int)blockDim.x;
	__shared__ u32 s_count[4];
	for (int k = threadIdx.x; k < kSampleSlots; k += nt) s_set[k] = kSampleEmpty;
	if (threadIdx.x < 4) s_count[threadIdx.x] = 0;
	__syncthreads();
	const int64_t sample = n < kSampleRows ? n : kSampleRows;
	// the sample = runs of 64 consecutive rows at evenly spaced offsets.  A join emits a cross product grouped by source:
	// single rows at a fixed stride can land on a different source every time (stride = group size) and make it look
	// like distinct pairs; inside a run a grouped input shows its repeats, and a shuffled one is sampled as well as by
	// single rows
	const double stride = (double)n / (double)((sample + 63) >> 6);
	// (round 5: every run is shifted by a pseudo-random part of its stride — offsets at exact multiples of n / 32 start every
	// run on a group boundary of a product whose groups divide that stride, and the boundary count below was off by 2x)
	auto run_start = [&](int64_t r) {
		const int64_t room = (int64_t)stride - 64;
		const int64_t jitter = room > 0 ? (int64_t)((((u64)r + 1) * 0x9E3779B97F4A7C15ull) >> 33) % room : 0;
		return (int64_t)((double)r * stride) + jitter;
	};
	// Round 5: the runs also show how the input is GROUPED.  `changes` counts the adjacent rows of a run whose sources
	// differ: a join's output (every source's rows in one stretch) changes source once per group, so n x changes / pairs
	// is its number of groups = distinct sources — which the hash-set estimate below cannot see (32 runs of a 2048 x 32
	// product hold 64 sources: "64 distinct", and the product went to the lane batches at three times the pre-pass's time).
	u32 fresh = 0, rows = 0, changes = 0, pairs = 0;
	const int64_t sample_up = (sample + 63) & ~(int64_t)63; // whole wavefronts stay in the loop for the shuffle
	for (int64_t k = threadIdx.x; k < sample_up; k += nt) {
		const bool in = k < sample;
		const int64_t at = min(n - 1, run_start(k >> 6) + (k & 63));
		const int64_t v = in ? src[at] : (int64_t)-1;
		int64_t pv = __shfl_up(v, 1);
		// a run's first row is compared with the row in front of the run: runs that start on a group boundary (evenly spaced
		// offsets and equal groups: a 2048 x 32 product sampled at stride 2048) would otherwise never see that boundary
		if ((k & 63) == 0) pv = in && at > 0 ? src[at - 1] : (int64_t)-1;
		if (in && v >= 0 && pv >= 0) {
			pairs++;
			changes += v != pv ? 1u : 0u;
		}
		if (v < 0) continue; // NULL row (or past the sample)
		rows++;
		const u32 x = (u32)v;
		u32 h = (x * 0x9E3779B1u) >> 20; // 12 bits
		for (;;) {
			const u32 old = atomicCAS(&s_set[h], kSampleEmpty, x);
			if (old == kSampleEmpty) fresh++;
			if (old == kSampleEmpty || old == x) break;
			h = (h + 1) & (kSampleSlots - 1);
		}
	}
	if (fresh) atomicAdd(&s_count[0], fresh);
	if (rows) atomicAdd(&s_count[1], rows);
	if (changes) atomicAdd(&s_count[2], changes);
	if (pairs) atomicAdd(&s_count[3], pairs);
	__syncthreads();
	// E[distinct](U) = U (1 - (1 - 1/U)^s) is increasing in U: every thread evaluates one point of a geometric grid between
	// the distinct sources seen and n ((1 - 1/U)^s as exp(s log1p(-1/U)), single precision) and the first point that
	// reaches the sampled count is the estimate — round 3 bisected on one thread (18 dependent steps: 4 of the 16 us this
	// kernel sits in front of every large call with)
	__shared__ u32 s_first;
	if (threadIdx.x == 0) s_first = (u32)nt - 1u;
	__syncthreads();
	const double d = s_count[0], sr = s_count[1];
	double est;
	if (sr < 1 || d < 1) {
		est = 1;
	} else if (d >= sr - 0.5) { // every sampled row had its own source
		est = (double)n;
	} else { // block-uniform branch
		const float fd = (float)d, fs = (float)sr, fn = (float)n;
		const float u = fd * __expf(__logf(fn / fd) * ((float)threadIdx.x / (float)(nt - 1)));
		const float e = u * (1.0f - __expf(fs * log1pf(-1.0f / u)));
		if (e >= fd) atomicMin(&s_first, threadIdx.x);
		__syncthreads();
		const float uf = fd * __expf(__logf(fn / fd) * ((float)s_first / (float)(nt - 1)));
		est = fmin((double)n, ceil((double)uf));
	}
	// grouped input (fewer than half of the adjacent sampled rows change their source): the groups counted through the
	// density of the changes.  A handful of changes in the whole sample (groups of hundreds of rows and more) is too few to
	// count by: then the first row of every run measures its own group — gallop + bisection to both ends, ~2 log2(g) loads —
	// and, a uniformly drawn row falling into a group in proportion to its length, n x mean(1 / g) is the number of groups.
	const double ch = s_count[2], pr = s_count[3];
	const bool grouped = pr >= 32.0 && ch * 2.0 < pr; // block-uniform
	__shared__ float s_inv[32];
	const int runs = (int)((sample + 63) >> 6);
	if (grouped && ch < 16.0) {
		if ((int)threadIdx.x < runs && threadIdx.x < 32) {
			const int64_t p = min(n - 1, run_start((int64_t)threadIdx.x));
			const int64_t v = src[p];
			auto extent = [&](int64_t dir) { // rows of v's stretch strictly beyond p in direction dir
				const int64_t room = dir > 0 ? n - 1 - p : p;
				int64_t step = 1;
				while (step <= room && src[p + dir * step] == v) step <<= 1;
				int64_t lo = step >> 1, hi = min(step, room + 1); // src[p + dir*lo] == v (or lo == 0); first mismatch in (lo, hi]
				while (hi - lo > 1) {
					const int64_t mid = (lo + hi) >> 1;
					if (src[p + dir * mid] == v) lo = mid;
					else hi = mid;
				}
				return lo;
			};
			s_inv[threadIdx.x] = 1.0f / (float)(1 + extent(1) + extent(-1));
		}
		__syncthreads();
	}
	if (threadIdx.x != 0) return;
	if (grouped) {
		double groups = (double)n * fmax(ch, 0.5) / pr;
		if (ch < 16.0) {
			double inv = 0;
			const int m = runs < 32 ? runs : 32;
			for (int r = 0; r < m; r++) inv += (double)s_inv[r];
			groups = (double)n * inv / (double)m;
		}
		est = fmin((double)n, fmax(d, groups));
	}
	const double distinct = fmin(est, (double)V);
	out->go = meet_bytes <= lanes_cost_bytes(edge_bytes, distinct, (double)n, (double)V) ? 1u : 0u;
	if (h_go) *h_go = out->go + 1u;
	out->sample_rows = s_count[1];
	out->sample_fresh = s_count[0];
	out->estimate = est;
}

#endif
// a sample riding in another kernel's launch (k_mark_sources: the lane assignment; k_meet4d: the pre-pass chain)
struct SampleArgs {
	double meet_bytes, edge_bytes;
	MeetDecision *out;
	u32 *h_go; // null: no sample
	int64_t n;
	const int64_t *src;
	int64_t V;
};

// What a level's kernels leave in the counter block, as the host needs it after the level (read back per level, or logged
// by the next level's k_level_reset into pinned memory when levels are enqueued ahead).
struct LevelLog {
	u64 front_edges, edges_scanned, word_gathers;
	u32 front_vertices, unresolved, front_words, pad2;
	u32 nzw; // non-empty words of the active-lane mask
	u32 r0;
};

// The per-level choices (top-down / bottom-up sparse / bottom-up dense, probe first or detect after) as one function of
// the counters, shared by the host loop and by k_level_reset, which checks an enqueued-ahead level against it.
struct LevelRule {
	double E, V, push_div, sparse_below;
	int wd, force_mode, force_pull, probe_always, use_probe;
};
enum : u32 { kLvPush = 1, kLvSparse = 2, kLvProbe = 4, kLvNone = 0x80 };
__host__ __device__ static inline u32 decide_level(const LevelRule &r, u64 front_edges, u32 front_words, u32 front_vertices,
                                                   u32 unresolved, int nzw) {
	bool push = r.force_mode == 1 || (r.force_mode == 0 && (double)front_edges * r.push_div < r.E);
	if (r.force_mode == 2) push = false;
	// expected wanted non-empty words per scanned in-edge: share of edges leaving frontier vertices x non-empty words per
	// frontier vertex x share of lane-words that still hold an active lane
	const double active_frac = (double)nzw / (double)r.wd;
	const double wpn = (double)front_edges / (r.E > 1.0 ? r.E : 1.0) *
	                   ((double)front_words / (double)(front_vertices > 1u ? front_vertices : 1u)) * active_frac;
	const bool sparse = !push && (r.force_pull == 1 || (r.force_pull == 0 && wpn < r.sparse_below));
	// The probe answers the pairs at distance t from frontier t-1 (one in-list scan per open pair) BEFORE level t is
	// expanded.  That only pays when it can spare an expensive expansion: before a top-down level (tiny) or a sparse
	// bottom-up level it costs more than it saves (cross product of 2048 sources x 32 destinations on the SF100-shaped
	// graph: the probes of levels 1 and 2 took 0.6 ms and spared nothing), before a DENSE bottom-up level (1.5 ms at WD = 32)
	// it answers what that level would have been run for.  Few open pairs: always.  In bytes at the rate the expansion
	// kernels stream: a probe is one latency-bound wavefront per open pair (measured 4.7 ns per pair at mean in-degree 89:
	// ~256 B per in-edge); a dense level moves ~E (8 + 6 WD), a sparse one ~4 E + 16 per frontier out-edge + V (4 + 24 WD), a
	// top-down one ~20 per frontier out-edge.
	bool probe = false;
	if (r.use_probe) {
		const double probe_bytes = (double)unresolved * (r.E / (r.V > 1.0 ? r.V : 1.0)) * 256.0;
		const double level_bytes = push ? (double)front_edges * 20.0
		                                : (!sparse ? r.E * (8.0 + 6.0 * r.wd)
		                                           : r.E * 4.0 + (double)front_edges * 16.0 + r.V * (4.0 + 24.0 * r.wd));
		probe = r.probe_always || probe_bytes <= level_bytes;
	}
	return (push ? kLvPush : 0u) | (sparse ? kLvSparse : 0u) | (probe ? kLvProbe : 0u);
}


struct LevelBuf {
	DevBuf buf;        // frontier lane-words [V][WD]
	DevBuf nz;         // per vertex: which of its WD words are non-empty (u32 bit mask)
	bool dirty = true; // may hold non-zero words
	// sparse pool (pgq_msbfs.hip): the allocation and the (V, WD) layout this buffer was zeroed whole for
	const void *init_buf = nullptr, *init_nz = nullptr;
	int64_t lay_V = -1;
	int lay_WD = 0;
};

struct Workspace {
	hipStream_t stream = nullptr;
	hipEvent_t ev_block = nullptr; // blocking event of wait_stream (created when the process first has many calls in flight)
	int device = 0; // where its buffers live (workspaces are pooled per device)
	DevBuf seen, qbuf[2], qflag, counters, flag, rank, usrc, key, idx, skey, sidx, ssrc, sdst, sres, soff,
	    sort_tmp, scan_tmp, bstart, levels_tab, child, in_src, in_dst, out_len, out_off, dist, dirty[2], touched,
	    tflag, out_val, out_ok, lane_sums, ste, def_src, def_dst, def_len, def_idx, def_off, def_ent, cbits, cbbase, cmeta, cwords, lblk, lrec, meet_cnt, meet_rec, meet_poff, meet_maps, meet_trace, wb_scratch, hv, hmask, hstart, hmap,
	    route_dec, ball_segs, ball_trace, sort_src, sort_dst, sort_out, dist_b, dirty_b[2], qbuf_b[2], touched_b, tflag_b, bi_block, dpart; // [kOpenGrid][WD + 1]: every workgroup's open-lane words + open-row count of the level's detection / probe
	std::vector<std::unique_ptr<LevelBuf>> levels; // shortestpath: one per level
	std::vector<std::unique_ptr<LevelBuf>> pool;   // otherwise: [0], [1] sparse pool, [2], [3] dense pool
	bool pool_trusted = false;                     // the last batch ended normally: the sparse pool's dirty flags are true
	int64_t prereset_V = -1;                       // batch_state_reset has run ahead for this (V, WD): the next batch's start is done
	int prereset_WD = 0;
	Counters *h_cnt = nullptr; // pinned
	LevelLog *h_log = nullptr; // pinned: [kSpecLevels + 2] counters per enqueued-ahead level, then two status words
	int64_t wb_V = -1;  // what wb_scratch's label arrays are initialised for
	int wb_grid = 0;
	void *h_meet = nullptr;    // pinned, 8 KB: the last workgroup of the pre-pass chain writes its statistics here
	bool meet_cnt_clean = false; // the device statistics block is all zero (the chain's last kernel leaves it so)
	// where the pre-pass left the rows it could not answer (one of the two queue regions inside def_src / def_dst / def_idx)
	int64_t *open_src = nullptr, *open_dst = nullptr;
	u32 *open_idx = nullptr;
	// chunk entry points: pinned staging block the kernels read the rows from / write the results to (device-addressable)
	void *h_io = nullptr;
	size_t h_io_cap = 0;
	int64_t *h_bstart = nullptr;
	size_t h_bstart_cap = 0;
	u32 epoch = 0;
	// cheapest path: which (V, lanes, type) the dist array is currently initialised for
	int64_t dist_V = -1;
	int dist_lanes = 0;
	int64_t dist_b_V = -1; // the same for the backward side's labels (relax_batches_bidir)
	int dist_b_lanes = 0;
	void *h_bi = nullptr;  // pinned: the bidirectional relaxation's counter block, copied back once per round
	u32 touch_epoch = 0;
	~Workspace();
};

struct WorkspaceLease {
	Workspace *ws = nullptr;
	int acquire();
	~WorkspaceLease();
};

static inline unsigned blocks_for(int64_t n, int block = 256) {
	return (unsigned)std::max<int64_t>(1, (n + block - 1) / block);
}

// One sparse bottom-up level in lane-list form (pgq_lanes.hip): packs the frontier (front/nz) and expands it into
// next/nz_next/seen on ws->stream; per-level statistics go to d_cnt like the other level kernels.
int pull_lanes_level(pgq_csr *c, Workspace *ws, int wd, const u64 *front, const u32 *nz, u64 *seen, u64 *next,
                     u32 *nz_next, const u64 *active, int stop, Counters *d_cnt);

// Pair-centric pre-pass (pgq_meet.hip): answers rows at distance <= 3 (and NULL / trivial / dead-end rows) into d_out,
// compacts the others into ws->def_src/def_dst/def_idx; meet_apply scatters their lengths back.  decide: whether the
// pre-pass pays (distinct sources, sampled) is settled on the device in the same launch chain; *ran = false: it did not run.
// shortestpath through the pre-pass: where the lists of the rows it answers go ([src, e, v, ..., dst], first-slot edges;
// entry i at the scanned offset, also stored in d_out_off[i]; a list that does not fit child_cap is not written) and how
// many elements they take
struct MeetPathsOut {
	int64_t *d_child = nullptr;
	int64_t child_cap = 0;
	int64_t *d_out_off = nullptr;
	int64_t total = 0;
};
// the lists of the rows the last meet_prepass(po) on this workspace answered, written again into po's (now larger) buffer
int meet_reemit_paths(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, const int64_t *d_out, MeetPathsOut *po);
int meet_prepass(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, int64_t *d_out,
                 u32 *n_open, MeetPathsOut *po, int decide_mode, double meet_bytes, double edge_bytes, bool *ran, int *observed_go,
                 int ball_mode = 0, bool *ball_ran = nullptr, double *est_sources = nullptr);
// the pre-pass's sampled decision alone, waited for (*go: the pre-pass pays)
int meet_decide_alone(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, double meet_bytes, double edge_bytes, bool *go);
// iterativelengthbidirectional: every row through k_bibfs (forward CSR from src, transposed CSR from dst); rows over its
// caps are compacted like the pre-pass's open rows
int meet_bidirectional(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, int64_t *d_out,
                       u32 *n_open);
// lengths + shifted offsets of the rows answered elsewhere, scattered back to row order
int meet_apply_paths(Workspace *ws, int64_t nd, const int64_t *d_len, const int64_t *d_off, int64_t base,
                     int64_t *d_out_len, int64_t *d_out_off);
int meet_apply(Workspace *ws, int64_t nd, const int64_t *d_len, int64_t *d_out);
// Assigns one lane per distinct source: fills ws->usrc (lane -> vertex), the row arrays sorted by lane
// (skey/sidx/ssrc/sdst/sres) and returns the number of distinct sources in *U.
int prepare_lanes(pgq_csr *c, Workspace *ws, int64_t n, const int64_t *d_src, const int64_t *d_dst, u32 *U,
                  bool dst_rule = true);
// bstart (pinned host) <- sorted-row boundaries of nb batches of L lanes (+ trivial / NULL tails)
int batch_bounds(Workspace *ws, int64_t n, int64_t L, int nb);
void merge_stats(pgq_stats_t &into, const pgq_stats_t &from); // a worker thread's counters into the caller's

} // namespace pgq
