// pgq_internal.h — shared internals of libpgq_hip (not installed; the public boundary is include/pgq_hip.h)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pgq_hip.h"

namespace pgq {

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- error plumbing -------------------------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define PGQ_HIP_TRY(expr)                                                                                          \
	do {                                                                                                           \
		hipError_t _e = (expr);                                                                                    \
		if (_e != hipSuccess) {                                                                                    \
			return ::pgq::fail(_e == hipErrorOutOfMemory ? PGQ_ERR_OOM : PGQ_ERR_HIP,                              \
			                   std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
		}                                                                                                          \
	} while (0)

#define PGQ_TRY(expr)                                                                                              \
	do {                                                                                                           \
		int _rc = (expr);                                                                                          \
		if (_rc != PGQ_OK) return _rc;                                                                             \
	} while (0)

int ensure_init();
// Waits for a stream the way the moment calls for (round 6).  hipStreamSynchronize spins: right for ONE caller (a chunk is
// ~50 us of kernels, a sleeping thread wakes up in 20-50), wrong for DuckDB's morsel-driven workers — 16 to 64 threads each in
// a chunk call (iterativelength.cpp:34 runs per DataChunk per thread) spin on as many cores, and a box that gives the
// process fewer cores than threads collapses (64 threads on a 16-CPU quota: 9.9 ms per chunk instead of 0.07).  With more than
// `block_above` calls in flight on this process the wait is an event created with hipEventBlockingSync: the thread sleeps.
struct CallScope { // one per C-ABI search call: counts the calls in flight
	CallScope();
	~CallScope();
};
int calls_in_flight();
hipEvent_t *thread_wait_event(); // this host thread's blocking event for the device it works on (never destroyed: a few per thread)
int wait_stream(hipStream_t st, hipEvent_t *blocking_event /* created on first use, owned by the caller */);
int ensure_edge_ids(pgq_csr *c); // copies a lazily uploaded edge-id array now (no-op otherwise); thread-safe
// What a CSR handle has learned about its graph — the pre-pass's measured bytes per row, the share of rows the
// source-centric kernel leaves open, the levels a full lane batch of each width runs — kept per graph SHAPE (V, E, largest
// degrees, mean two-hop walk) across handles: DuckPGQ builds a CSR per query (iterative_length_function_data.cpp:27,
// duckpgq_state.cpp:162-170), so a handle's first call is the common case, and a query over the same tables as the one
// before starts where that one ended instead of calibrating again.  Speed only: every route gives the same answers.
void calibration_load(pgq_csr *c);  // at the end of an upload
void calibration_store(pgq_csr *c); // when a handle dies, and whenever it has measured something new
// the device this host thread works on (pgq_init's device unless a multi-device call bound the thread to another)
int current_device();
// compute units of the device this host thread works on (cached per device; 256 on MI355X — the persistent grids are
// sized from it, not from the constant)
int device_cus();
void bind_thread_device(int device); // < 0: back to the default
const std::vector<int> &enabled_devices();

// ---- options --------------------------------------------------------------------------------------------
struct Options {
	int words = 0;          // lane-words per vertex per batch (0 = auto: 1,2,4,8,16,32 by unique sources)
	int max_words = 32;     // upper bound for auto
	double push_div = 24.0; // top-down while frontier out-degree sum * push_div < E
	int profile = 0;        // per-kernel-class HIP event timing
	int hub_chunk = 4096;   // in-degree above which a vertex is split into slices for the bottom-up kernels
	int push_chunk = 256;   // out-edges per top-down work item (short dependent chains per wavefront)
	int force_mode = 0;     // 0 adaptive, 1 always push, 2 always pull (tests)
	int force_pull = 0;     // 0 adaptive, 1 always k_pull_sparse, 2 always k_pull (tests)
	int blocks_per_cu = 8;  // persistent grid sizing for the pull kernel
	int chain = 1;          // cheapest_path_length: walk out-degree-1 chains per row before the batched relaxation
	int chain_cap = 4096;   // steps after which a chain is taken for a cycle and left to the batched relaxation
	int alloc_cache_mb = 8192; // freed CSR / upload blocks kept for the next upload, per process (0: straight hipFree)
	int relax_small_limit = 2048; // changed vertices at or below which relaxation rounds loop on the device
	int relax_light = 1;      // (1: where the mean out-degree makes it pay, 2: always, 0: never) batched relaxation over weight-sorted lists: edges above a cap that doubles phase by phase are not
	                          // scanned, and a vertex stops at the first edge that cannot beat its lanes' bounds (0: plain rounds)
	int relax_labels32 = 1;   // int64 weights whose path sums fit 31 bits keep 4-byte labels (rows of 256 bytes instead of 512)
	int relax_light_min_degree = 8; // relax_light = 1: only CSRs with at least this many edges per vertex (2: always)
	int relax_light_div = 4;  // first cap = mean weight / this
	int relax_bidir = 0;      // int64 weights, light-edges-first graphs, about one destination per source: every lane a (src, dst) pair
	                          // searched from both ends under a common distance cap (relax_batches_bidir).  Bit-exact in the tests; OFF as
	                          // shipped: on the weighted knows graph the two half-distance balls already hold the hubs — 30 M relaxed edges
	                          // per 64 pairs against 40 M one-sided, but 83 rounds instead of 35: 232 ms per 512 pairs against 89
	int relax_bidir_rows = 2; // ... when the call has at most this many rows per distinct source
	int relax_bidir_c0_div = 64;    // first distance cap = mean weight x 2 / this ...
	int relax_bidir_step_div = 128; // ... raised by mean weight x 2 / this per phase (and by a quarter, and past empty bands)
	int relax_streams = 6;    // batches of the relaxation side by side on their own label arrays (0: `streams`)
	int relax_split = 1;      // lists longer than 128 edges are relaxed 64 edges per wavefront by a second launch of the round
	int relax_delta_div = 0;  // batched relaxation: > 0 = a round only expands labels below a threshold that grows by mean weight / this
	                          // per round.  Off: measured on the weighted knows graph it does not pay (lanes reach a vertex in different
	                          // bands, so its adjacency is re-read per lane: 1.9 s per 512 pairs at 64, 1.24 s at 16, 1.16 s plain)
	int trace = 0;          // per-level line on stderr
	int probe = 1;          // destination probe before each expansion
	int probe2 = 1;         // two-hop destination probe when few pairs are left
	int probe2_div = 4;     // ... when open pairs <= lanes / probe2_div
	int probe_max_in = 4096; // destinations with more in-neighbours are not probed (one wavefront's serial scan); k_detect answers them a level later
	int probe2_abs = 4096;  // ... or when at most this many pairs are open, whatever the batch width
	int detect_grid_mult = 8;  // k_detect grid = this many 256-thread workgroups per CU at most (rows are taken grid-stride)
	int sort_single_batch = 0; // 1: rows sorted by lane even when the distinct sources fit one batch (round-4 behaviour; tests)
	int spec_levels = 1;    // levels of a batch enqueued ahead under the previous batch's plan, checked on the device: one wait per
	                        // batch instead of one per level (0: a host round trip after every level)
	int detect_unroll = 4;  // rows per thread of k_detect with their gathers in flight together (1, 2 or 4)
	int meet_calibrate = 1; // the pre-pass's bytes per row are measured (1024 pseudo-random pairs) before the first large call is routed
	int stage2_ahead = 1;   // one-batch calls that repeat a row count: lane ids + batch start enqueued in front of the lane assignment's wait
	int route_memo = 1;     // large calls on the buffers of the last one that the sample sent to the lane batches go there straight
	int probe_always = 0;   // 1: probe before every level of a batch that uses the probe (round-2 behaviour; tests)
	int probe2_cap = 1 << 16; // in-edges a two-hop probe may walk per pair
	int defer = 8;          // defer stragglers when open pairs <= lanes/defer (0 = never)
	int part_weight = 1024; // in-edges (+8 per vertex) per bottom-up work part
	int upload_threads = 8;  // host threads filling pinned blocks from a pageable CSR (one more thread issues the copies)
	int upload_narrow_host = 1; // 1: the staging threads narrow the adjacency to int32; 0: raw int64 over PCIe, narrowed on the device
	int streams = 3;        // batches searched concurrently (one host thread + HIP stream each)
	int sparse_lds = 1;     // keep the 1-bit frontier map in LDS when it fits (1024-thread workgroups)
	int sparse_spill = 3;   // > 0: words beyond a record's inline ones go through the per-wave LDS queue; 0: per-lane trips (tests)
	int sparse_pw = 1;      // packed words per chunk fetched per trip of k_pull_sparse's fallback loop (1..3)
	int sparse_unroll = 2;  // 64-entry chunks in flight per wave in k_pull_sparse (1, 2 or 4)
	double sparse_below = 1.5; // expected wanted non-empty words per in-neighbour below which k_pull_sparse runs
	int meet = 1;           // iterativelength: answer pairs at distance <= 3 by the pair-centric pre-pass (k_meet3) when cheaper
	int meet_cap = 1 << 14; // adjacency entries a pair's two-hop walk may scan in k_meet3 (one wavefront); beyond: k_meet4d (16 wavefronts)
	int chunk_zero_copy = 1;     // chunk entry points: the pre-pass kernels read / write the pinned staging block directly
	int meet_small_rows = 16384; // calls of at most this many rows are latency-bound: k_meet3 with more requests in flight and meet_cap_small
	int paths_reserve_mb = 1024; // shortestpath through the pre-pass: its lists' buffer is reserved up to this size up front (9 elements per row fit below ~15 M rows), written again at the exact size beyond
	int meet_spin_wait = 0;      // the chain's wait polls its pinned report (5-6 us per call) instead of synchronising the stream; see meet_wait
	int meet_wide_rows = 2048;   // calls of at most this many rows on graphs beyond the Infinity Cache: 2-4 wavefronts per row in the first stage (k_meet3w); 0: off
	int meet_wide_rows_always = 0; // ... on every graph (tests)
	int meet_cap_small = 1 << 14; // ... a lower walk cap (longer walks go to the 16-wavefront kernel sooner)
	int meet_cap_paths = 1 << 14; // the same for shortestpath rows (longer walks go to k_meet4: 16 wavefronts per row)
	int meet4 = 1;          // rows k_meet3 leaves open: LDS bit-map kernel for distance <= 4 (k_meet4) when V fits
	int meet4_test_cap = 1 << 15; // ... that the testing walk of the distance-4 step may scan after its probe found nothing (k_meet4d)
	int meet4_cap = 1 << 20; // adjacency entries either two-hop walk of a row may scan in k_meet4
	int wbibfs = 0;            // cheapest_path_length, int64 weights: a bidirectional delta-stepping search per row before the batched
	                           // relaxation (k_wbibfs).  Off by default: bit-exact in the tests, not yet measured at SF100 scale
	int wbibfs_rows = 1 << 20; // ... for at most this many rows per call
	int wbibfs_cap = 64 << 20; // adjacency entries a row may relax before it is left to the batched relaxation
	int wbibfs_queue = 1 << 17; // near-queue entries (vertices inside the current band, with duplicates)
	int wbibfs_prune = 0;       // skip relaxing a vertex whose label + the other side's radius already reaches `best` (model: -28 % work; untimed)
	int wbibfs_far = 1 << 21;   // far / touched entries (every labelled vertex once; at most V)
	int wbibfs_mem_mb = 2048;  // scratch budget (two label arrays of V entries per workgroup)
	int wbibfs_delta_div = 64; // band width = mean weight / this (a model run on the weighted knows graph: 3-10x fewer relaxations at 64 than at 8)
	int bibfs_rows = 256;      // k_bibfs (one bidirectional search per row) runs when at most this many rows are still open (0: off)
	int bibfs_rows_max = 16384; // ... on graphs with many edges it takes up to E / 4096 rows, at most this many (a lane batch there costs more than that many searches)
	int bibfs_grid = 64;       // workgroups of k_bibfs (one row at a time each; every one owns 4 x bibfs_queue words of scratch)
	int bibfs_cap = 8 << 20;   // adjacency entries one expansion of k_bibfs may read
	int bibfs_queue = 1 << 17; // frontier vertices per side k_bibfs keeps
	int meet4_lds_kb = 150;    // largest vertex bit map k_meet4 keeps in LDS (tests lower it to force the global-memory maps)
	int meet4_global_mb = 256; // vertex bit maps of k_meet4 in global memory when V does not fit in LDS: total budget (0: off)
	double meet_bias = 1.0; // pre-pass runs while its estimated bytes <= meet_bias x the MS-BFS estimate
	int lanes = 1;          // sparse bottom-up levels use the lane-list kernel (k_pull_lanes); 0: k_pull_sparse
	int lanes_unroll = 2;   // 64-entry chunks in flight per wave in k_pull_lanes (1, 2 or 4)
	int meet_trace = 0;      // debugging: per-workgroup timestamps of k_meet4d, summarised on stderr
	int meet4_grid_mult = 1; // k_meet4d grid = this many 1024-thread workgroups per CU (2 fit beside their LDS bit maps; rows are handed out
	                         // dynamically).  Round 5, after stage A / the probe shortened a row: one per CU 42.6 us, two 51.2, three 62.8 per 65,536 rows
	int meet_grid_mult = 8; // k_meet3 grid = this many times the 8192 one-wavefront workgroups the chip holds (rows per workgroup = n / grid)
	int meet_layout = 1;    // build the padded adjacency + slot descriptors at upload (the pre-pass needs them)
	int meet_align = 32;    // entries a padded list is aligned and padded to (4 = one 16-byte group; 16 / 32 = whole 64 / 128-byte lines: -4 % / -6 % on the pre-pass)
	// round 6: source-centric search for rows that arrive grouped by source (pgq_ball.h: k_ball_segments + k_src_ball)
	int ball = 1;               // 1: the device decides per call from the number of source runs; 2: always when allowed (tests); 0: never
	int ball_cap = 1 << 20;     // adjacency entries the two-hop ball of one source may hold; a segment over it leaves its far rows open
	int ball_test_cap = 1 << 15; // adjacency entries the backward two-hop walk of one row (distance 4) may scan
	int ball_seg_kb = 512;      // the least a segment costs in the decision, in KB at streaming rate (its ~15 dependent round trips on one of ~512 workgroup slots)
	int block_above = 3;        // more search calls than this in flight on the process: waits sleep on a blocking event instead of spinning
	int calibration_cache = 1;  // what a handle measures about its graph (bytes per row of the pre-pass, level plans, ...) is kept per graph
	                            // shape across handles (0: every handle starts from nothing; tests of the cold paths)
	int ball_head_mb = 512;     // the fixed-stride in-list heads (pgq_csr::rhead, V x 256 bytes) are built when they fit this many MB (0: never)
	int ball_seg_rows_small = 1024; // calls of at most meet_small_rows rows: a source run is cut into segments of this many rows (64..1024, a power
	                                // of two).  Measured on 1 source x 2048 destinations: 64, 128, 256 and 1024 all give 0.065 ms per chunk — the call is
	                                // its launches and its wait, not the scan — so the shorter segments stay an option
	int ball_grid = 0;          // > 0: at most this many workgroups of k_src_ball (debugging / sweeps)
	int route_timing = 1;     // large calls grouped by source: the route that measured faster on this graph shape is kept (search_device)
	int route_timing_rows = 65536; // ... calls of at least this many rows
	double route_try_factor = 4.0; // ... the lane batches are tried once when the source-centric route took this many times their modelled time
	int ball_sort = 1;          // rows with repeated sources that are NOT grouped are sorted by source first (0: such calls take the older routes)
	double ball_bias = 1.0;     // the ball runs while ball_bias x its estimated bytes <= the cheaper of the pre-pass and the lane batches
};
Options &options();
// per-handle options (pgq_csr_set_option): the override a host thread works under, and a scope that installs a
// handle's for the duration of a C-ABI call
// Host worker threads that outlive a call (one idle list per device: a worker keeps its HIP events and its device
// binding): `worker_submit` hands `fn` to an idle worker of `device` or starts a new one, `worker_wait` blocks until it
// has run.  A call used to start a std::thread per extra batch stream; nested use cannot deadlock (the pool grows).
struct WorkerTask;
std::shared_ptr<WorkerTask> worker_submit(int device, std::function<void()> fn);
int worker_wait(const std::shared_ptr<WorkerTask> &t); // PGQ_OK, or the error of a job that threw (its results are missing)
Options *options_override();
void set_options_override(Options *o);

// ---- kernel classes (stats) -----------------------------------------------------------------------------
enum KClass {
	K_PREP = 0,      // lane assignment, batch init
	K_PUSH = 1,      // top-down expansion
	K_PULL = 2,      // bottom-up expansion (+ fused sweep)
	K_PULL_HUB = 3,  // bottom-up for split high-in-degree vertices
	K_QUEUE = 4,     // frontier queue build / clear
	K_DETECT = 5,    // per-pair destination test + active-lane mask
	K_RECON = 6,     // path reconstruction
	K_RELAX = 7,     // cheapest path relaxation
	K_PULL_SPARSE = 8, // bottom-up expansion, edge-organised sparse variant
	K_MEET = 9,      // pair-centric two-hop pre-pass: k_meet3, one wavefront per row (pgq_meet.hip)
	K_MEET4 = 10,    // ... its bit-map kernels for the rows k_meet3 leaves open (k_meet4d / k_meet4)
	K_BIBFS = 11,    // ... one bidirectional search per row (k_bibfs)
	K_BALL = 12,     // source-centric search of rows grouped by source: k_ball_segments + k_src_ball (pgq_ball.h)
	K_COUNT = 13
};

struct ThreadStats {
	pgq_stats_t s;
	ThreadStats();
};
ThreadStats &tstats();

// ---- device CSR -----------------------------------------------------------------------------------------
struct HubItem {
	int32_t vertex;
	int32_t pad;
	int64_t begin, end; // slice of the (reverse) adjacency
};

} // namespace pgq

struct pgq_csr {
	int device = 0;
	int64_t V = 0, E = 0;
	int w_type = 0;
	// forward CSR (slot order == host CSR slot order, never re-sorted)
	int64_t *off = nullptr;      // V+1
	int32_t *adj = nullptr;      // E
	int64_t *edge_ids = nullptr; // E or null (slot index is the id)
	// PGQ_UPLOAD_LAZY_EDGE_IDS: the caller's host array, copied by ensure_edge_ids on the first call that reads edge ids
	const int64_t *lazy_edge_ids = nullptr;
	std::mutex edge_ids_lock;
	void *w = nullptr;           // E x 8 B or null
	// reverse CSR (in-neighbours), built on device at upload
	int64_t *roff = nullptr; // V+1
	int32_t *radj = nullptr; // E  source vertex of the in-edge
	// high in-degree vertices split into work items for the bottom-up kernel
	pgq::HubItem *pull_hubs = nullptr; // device
	int32_t *pull_hub_vertices = nullptr;
	int64_t n_pull_hub_items = 0, n_pull_hub_vertices = 0;
	int32_t *pull_parts = nullptr; // n_pull_parts (begin,end) vertex ranges, no hubs inside, <= 16 vertices each
	int n_pull_parts = 0;
	uint8_t *rown = nullptr; // E: owner vertex of every in-slot, as an index inside its part
	uint32_t *rpk = nullptr; // E (+ padding): radj | rown << 28, one word per in-slot for k_pull_lanes (null if V >= 2^28)
	// Layout of the pair-centric kernels (pgq_meet.hip; built at upload by build_meet_layout, null when meet_layout = 0):
	// padded adjacencies whose lists start on a 16-byte group boundary (aligned to `meet_align` entries) and are filled
	// up to whole groups with copies of their last entry; per vertex {first group, entries}; per adjacency slot a
	// 16-byte descriptor {neighbour, the neighbour's first group, its entries, 0} in the slot's own direction, so that a
	// two-hop walk needs no offset look-up
	int32_t *padj = nullptr, *rpadj = nullptr; // 4 x padj_groups / rpadj_groups entries
	uint2 *fseg = nullptr, *rseg = nullptr;    // V
	uint4 *fdesc = nullptr, *rdesc = nullptr;  // E (+ 1): slot order of adj / radj
	// entries of a vertex's two-hop walk in either direction (sum of its neighbours' list lengths, saturating): the
	// pair-centric kernels expand the endpoint whose walk is the shorter one (round 4; it was the shorter one-hop list)
	uint32_t *fwork = nullptr, *rwork = nullptr; // V
	// round 6, the source-centric kernel (pgq_ball.h): every vertex's first 62 in-neighbours at a FIXED stride of 256 bytes =
	// two 128-byte lines: entries 0..30 and the in-degree in the first (most destinations at distance 3 show a witness among
	// them: one line per row), entries 31..62 in the second; positions past the list's end repeat its last entry.  A row's
	// scan then starts from its destination id alone: no gather of the list's position (one more 128-byte line per row, a
	// sixth of the kernel's traffic) and no dependent round trip.  Built when V x 256 B fits `ball_head_mb`; null otherwise.
	uint4 *rhead = nullptr; // V x 16
	int64_t padj_groups = 0, rpadj_groups = 0;
	std::unique_ptr<pgq::Options> opt;   // this handle's own options (pgq_csr_set_option); null: the process-wide set
	std::atomic<int> meet_far_rows { 1 }; // the last pre-pass call left rows for k_bibfs (it is launched only then; pgq_meet.hip)
	int64_t hub_threshold = 0;
	int64_t max_out_degree = 0, max_in_degree = 0;
	double two_hop_mean = 0; // mean over vertices of in-degree x out-degree = expected two-hop walk of a random endpoint
	std::atomic<double> meet_bpr { 0.0 }; // bytes per row the pre-pass has been measured to move on this CSR (0: not yet; pgq_msbfs.hip)
	int64_t bytes = 0;
	bool has_negative_weight = false;
	// multi-GPU: copies of this CSR on the other enabled devices (pgq_csr_replicate), indexed like enabled_devices();
	// entry = this object for its own device.  Owned by the primary.
	// PageRank over this CSR (V + 2 doubles), computed once per handle like the reference's bind-data state
	void *rw = nullptr;          // E x 8 B: in-edge weights in reverse-CSR order (built on first use by the weighted pair search)
	double w_mean = 0;
	// cheapest_path_length on general graphs: the forward adjacency with every vertex's list sorted by weight, and the
	// weights in that order (built on first use); w_max_bits = the largest weight as its bit pattern
	int32_t *wadj = nullptr;
	void *wsorted = nullptr;
	int32_t *rwadj = nullptr;    // the same for the in-lists (radj / rw), for the backward side of the bidirectional relaxation
	void *rwsorted = nullptr;
	unsigned long long w_max_bits = 0;
	int64_t *wcc = nullptr;      // weakly_connected_component ids of the V + 2 forest entries (computed once per handle)
	double *pagerank = nullptr;
	int pagerank_iterations = 0;
	std::vector<pgq_csr *> replicas;
	std::vector<int> replica_devices;   // the device list `replicas` was built for (compared with enabled_devices())
	std::vector<pgq_csr *> retired;     // replicas of an earlier device list: calls in flight may still read them; freed with the CSR
	std::mutex replica_lock;            // guards the three vectors above
	std::mutex lazy_lock;               // guards the arrays built on first use (wcc)
	// the levels the last lane batch of each width (1, 2, ..., 32 lane-words) ran, as kLv* bits: the next batch of that width
	// enqueues them ahead of the host (pgq_msbfs.hip, spec_levels)
	std::mutex plan_lock;
	std::vector<uint8_t> level_plan[6];
	// where the sampled decision sent the last large call (more than 16,384 rows) on these buffers: a call that repeats it
	// (same row count, same device pointers: the binder evaluates iterativelength and shortestpath on the same pairs, a
	// benchmark repeats its step) does not launch the pre-pass chain just to have it called off.  Affects speed only: both
	// routes give the same answers, and the sample is taken again on every call (plan_lock guards it)
	struct RouteMemo {
		int64_t n = -1;
		const void *src = nullptr, *dst = nullptr;
		int go = 1;
		int64_t id_n = -1; // row count of the last call whose rows stayed in place (one batch) ...
		int id_wd = 0;     // ... and its batch width: the next call with that row count runs stage 2 ahead of its wait
		// round 6: the buffers of the last call the source-centric kernels looked at, and what they said.  Declined (scattered
		// pairs): the next call on them does not launch the two kernels again (6 us in front of a 0.2-ms call).  Taken: the next
		// chain is those two kernels alone, without the stage kernels that would only return at once behind them.  Speed only.
		int64_t ball_n = -1;
		const void *ball_src = nullptr, *ball_dst = nullptr;
		bool ball_yes = false;
		bool sorted_yes = false; // ... or took them after a sort by source (rows of a source scattered over the input)
	} route_memo;
	// share of a call's rows the source-centric kernel left open, last time it ran on this CSR (half the weight to the newest
	// call): above ~2 % those rows drag the lane batches along anyway (R-MAT: far and unreachable pairs), and the kernel
	// stays out of the chain until the CSR is uploaded again
	std::atomic<double> ball_open_frac { 0.0 };
	// large grouped calls, wall time per row in ns as measured on this graph shape (0: not yet): through the source-centric
	// kernel (everything it took: its own kernels and the search of the rows it left open) and through the lane batches
	// (the BEST time seen, not a mean: a process's first call of a kind also pays for allocations, kernel attributes, the
	// calibration — 26 ms where the call takes 0.3; the first version took that sample for the route's cost and left the SF100
	// cross product on the lane batches, 7 x slower, for good)
	std::atomic<double> route_ball_ns { 0.0 }, route_lanes_ns { 0.0 };
	std::atomic<int> route_ball_samples { 0 }, route_lanes_samples { 0 };
	std::atomic<int64_t> route_rows { 0 }; // rows of the calls the figures come from: they speak for calls of at least half that size
	std::atomic<int> route_try_lanes { 0 }; // the former cost far more than the byte model's price of the latter: time the latter (twice)
	bool is_replica = false;
};

namespace pgq {

struct OptionScope {
	Options *saved;
	explicit OptionScope(const pgq_csr *c) : saved(options_override()) {
		if (c && c->opt) set_options_override(c->opt.get());
	}
	explicit OptionScope(Options *o) : saved(options_override()) { set_options_override(o); }
	~OptionScope() { set_options_override(saved); }
};

// host mirror used by the chunk entry points: resolves UnifiedVectorFormat into flat arrays
struct FlatPairs {
	std::vector<int64_t> src, dst; // -1 src = NULL row
	std::vector<uint8_t> dst_valid;
};
int flatten_pairs(int64_t V, int64_t n, const pgq_vec_t &src, const pgq_vec_t &dst, FlatPairs &out,
                  bool check_dst_validity);

int flatten_pairs_into(int64_t V, int64_t n, const pgq_vec_t &src, const pgq_vec_t &dst, int64_t *out_src, int64_t *out_dst);

inline void mask_fill_valid(uint64_t *mask, int64_t n) {
	for (int64_t i = 0; i < (n + 63) / 64; i++) mask[i] = ~0ULL;
}
inline void mask_set_invalid(uint64_t *mask, int64_t row) { mask[row >> 6] &= ~(1ULL << (row & 63)); }

// event timing of one kernel class; no-op unless options().profile
struct KernelTimer {
	hipStream_t stream;
	int kclass;
	hipEvent_t a = nullptr, b = nullptr;
	KernelTimer(hipStream_t s, int k);
	void stop(); // records the stop event
	static void flush(); // resolves all pending pairs of this thread (call after a stream sync)
};

// scratch buffer that grows on demand, per workspace
// cached device blocks for CSRs and upload temporaries (pgq_runtime.hip)
int dev_alloc(void **out, size_t bytes);
void dev_free(void *p);
void dev_cache_trim();
void drop_idle_workspaces(); // frees the pooled (not leased) per-call workspaces (pgq_msbfs.hip)
template <typename T> inline int dev_alloc_as(T **out, size_t count) {
	void *p = nullptr;
	int rc = dev_alloc(&p, count * sizeof(T));
	*out = static_cast<T *>(p);
	return rc;
}

// device -> pageable host memory through a pinned block (a pageable hipMemcpy D2H is staged by the runtime at well under
// 1 GB/s on these boxes); waits for `st`
int staged_download(void *h_dst, const void *d_src, size_t bytes, hipStream_t st);

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	int reserve(size_t bytes);
	void release();
	template <typename T> T *as() { return static_cast<T *>(p); }
};

} // namespace pgq
