// pgq_runtime.hip — process state, error/option/stat plumbing and the device CSR (upload, reverse CSR, hubs).
//
// Replaces, on the device side, the CSR object of the reference
// (src/include/duckpgq/core/utils/compressed_sparse_row.hpp:25-47): the host CSR built by
// create_csr_vertex/create_csr_edge (src/core/functions/scalar/csr_creation.cpp:86-198) is copied once to HBM,
// the adjacency narrowed to int32, and an in-neighbour (reverse) CSR derived on the GPU for the bottom-up
// kernel and for path reconstruction.  Slot order of the forward CSR is preserved bit for bit — it decides
// which parallel edge shortestpath reports (shortest_path.cpp:23-30).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <immintrin.h>
#include <unordered_map>

#include "pgq_internal.h"

namespace pgq {

static thread_local std::string t_err;
void set_error(const std::string &msg) { t_err = msg; }
int fail(int code, const std::string &msg) {
	t_err = msg;
	return code;
}

static std::mutex g_init_lock;
static std::atomic<int> g_inited { 0 };
static int g_device = 0;
static std::vector<int> g_devices; // devices enabled for multi-GPU calls (pgq_init_devices); [g_device] by default
static thread_local int t_device = -1;
int current_device() { return t_device >= 0 ? t_device : g_device; }
void bind_thread_device(int device) { t_device = device; }
const std::vector<int> &enabled_devices() { return g_devices; }
static Options g_opt;
// a search call on a handle that carries its own options (pgq_csr_set_option) sees those, on every host thread that
// works for the call; everything else sees the process-wide set
static thread_local Options *t_opt = nullptr;
Options &options() { return t_opt ? *t_opt : g_opt; }
Options *options_override() { return t_opt; }
void set_options_override(Options *o) { t_opt = o; }

// ---- persistent host workers -------------------------------------------------------------------------------
struct WorkerTask {
	std::function<void()> fn;
	std::mutex m;
	std::condition_variable cv;
	bool done = false;
	int failed = PGQ_OK; // the job threw (std::bad_alloc in a worker's vectors, ...): its rows were never written
	std::string what;
};
namespace {
struct Worker {
	std::mutex m;
	std::condition_variable cv;
	std::shared_ptr<WorkerTask> job;
	int device = 0;
};
std::mutex g_worker_lock;
// leaked on purpose: detached workers may still be parked on their condition variables while statics are destroyed
std::map<int, std::vector<Worker *>> &idle_workers() {
	static auto *m = new std::map<int, std::vector<Worker *>>();
	return *m;
}
void worker_loop(Worker *w) {
	for (;;) {
		std::shared_ptr<WorkerTask> t;
		{
			std::unique_lock<std::mutex> lk(w->m);
			w->cv.wait(lk, [&] { return w->job != nullptr; });
			t.swap(w->job);
		}
		try {
			t->fn();
		} catch (const std::bad_alloc &) { // the jobs report through return codes; nothing may unwind into the pool, and a
			t->failed = PGQ_ERR_OOM;       // job that threw must not look like one that ran: worker_wait hands the failure on
			t->what = "a worker thread ran out of host memory";
		} catch (const std::exception &e) {
			t->failed = PGQ_ERR_HIP;
			t->what = std::string("a worker thread failed: ") + e.what();
		} catch (...) {
			t->failed = PGQ_ERR_HIP;
			t->what = "a worker thread failed with an unknown exception";
		}
		t->fn = nullptr; // drop the captures before the submitter is released
		{
			std::lock_guard<std::mutex> g(g_worker_lock);
			idle_workers()[w->device].push_back(w);
		}
		{
			std::lock_guard<std::mutex> g(t->m);
			t->done = true;
		}
		t->cv.notify_all();
	}
}
} // namespace

std::shared_ptr<WorkerTask> worker_submit(int device, std::function<void()> fn) {
	auto t = std::make_shared<WorkerTask>();
	t->fn = std::move(fn);
	Worker *w = nullptr;
	{
		std::lock_guard<std::mutex> g(g_worker_lock);
		auto &idle = idle_workers()[device];
		if (!idle.empty()) {
			w = idle.back();
			idle.pop_back();
		}
	}
	if (!w) {
		w = new Worker();
		w->device = device;
		std::thread(worker_loop, w).detach();
	}
	{
		std::lock_guard<std::mutex> g(w->m);
		w->job = t;
	}
	w->cv.notify_one();
	return t;
}

int worker_wait(const std::shared_ptr<WorkerTask> &t) {
	std::unique_lock<std::mutex> lk(t->m);
	t->cv.wait(lk, [&] { return t->done; });
	return t->failed == PGQ_OK ? PGQ_OK : fail(t->failed, t->what);
}

static std::atomic<int> g_calls { 0 };
CallScope::CallScope() { g_calls.fetch_add(1, std::memory_order_relaxed); }
CallScope::~CallScope() { g_calls.fetch_sub(1, std::memory_order_relaxed); }
int calls_in_flight() { return g_calls.load(std::memory_order_relaxed); }
hipEvent_t *thread_wait_event() {
	static thread_local hipEvent_t ev[64] = {};
	return &ev[current_device() & 63];
}
int wait_stream(hipStream_t st, hipEvent_t *ev) {
	if (ev && calls_in_flight() > options().block_above) {
		if (!*ev) PGQ_HIP_TRY(hipEventCreateWithFlags(ev, hipEventBlockingSync | hipEventDisableTiming));
		PGQ_HIP_TRY(hipEventRecord(*ev, st));
		PGQ_HIP_TRY(hipEventSynchronize(*ev));
		return PGQ_OK;
	}
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	return PGQ_OK;
}

int device_cus() {
	static std::atomic<int> cached[64] = {};
	const int dev = current_device() & 63;
	int v = cached[dev].load(std::memory_order_relaxed);
	if (v > 0) return v;
	int cus = 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || cus <= 0) cus = 256;
	cached[dev].store(cus, std::memory_order_relaxed);
	return cus;
}

ThreadStats::ThreadStats() { memset(&s, 0, sizeof(s)); }
ThreadStats &tstats() {
	static thread_local ThreadStats ts;
	return ts;
}

// one table for pgq_set_option / pgq_get_option
struct OptRef {
	const char *name;
	int *i;
	double *d;
};
static std::vector<OptRef> option_table(Options &o) {
	return {
		{ "words", &o.words, nullptr },
		{ "max_words", &o.max_words, nullptr },
		{ "push_div", nullptr, &o.push_div },
		{ "profile", &o.profile, nullptr },
		{ "hub_chunk", &o.hub_chunk, nullptr },
		{ "push_chunk", &o.push_chunk, nullptr },
		{ "force_mode", &o.force_mode, nullptr },
		{ "force_pull", &o.force_pull, nullptr },
		{ "blocks_per_cu", &o.blocks_per_cu, nullptr },
		{ "relax_small_limit", &o.relax_small_limit, nullptr },
		{ "relax_delta_div", &o.relax_delta_div, nullptr },
		{ "relax_light", &o.relax_light, nullptr },
		{ "relax_light_div", &o.relax_light_div, nullptr },
		{ "relax_light_min_degree", &o.relax_light_min_degree, nullptr },
		{ "relax_labels32", &o.relax_labels32, nullptr },
		{ "relax_split", &o.relax_split, nullptr },
		{ "relax_streams", &o.relax_streams, nullptr },
		{ "relax_bidir", &o.relax_bidir, nullptr },
		{ "relax_bidir_rows", &o.relax_bidir_rows, nullptr },
		{ "relax_bidir_c0_div", &o.relax_bidir_c0_div, nullptr },
		{ "relax_bidir_step_div", &o.relax_bidir_step_div, nullptr },
		{ "chain", &o.chain, nullptr },
		{ "chain_cap", &o.chain_cap, nullptr },
		{ "alloc_cache_mb", &o.alloc_cache_mb, nullptr },
		{ "trace", &o.trace, nullptr },
		{ "probe", &o.probe, nullptr },
		{ "defer", &o.defer, nullptr },
		{ "probe2", &o.probe2, nullptr },
		{ "probe2_cap", &o.probe2_cap, nullptr },
		{ "probe2_div", &o.probe2_div, nullptr },
		{ "probe2_abs", &o.probe2_abs, nullptr },
		{ "probe_max_in", &o.probe_max_in, nullptr },
		{ "probe_always", &o.probe_always, nullptr },
		{ "detect_grid_mult", &o.detect_grid_mult, nullptr },
		{ "sort_single_batch", &o.sort_single_batch, nullptr },
		{ "spec_levels", &o.spec_levels, nullptr },
		{ "route_memo", &o.route_memo, nullptr },
		{ "stage2_ahead", &o.stage2_ahead, nullptr },
		{ "meet_calibrate", &o.meet_calibrate, nullptr },
		{ "detect_unroll", &o.detect_unroll, nullptr },
		{ "part_weight", &o.part_weight, nullptr },
		{ "sparse_below", nullptr, &o.sparse_below },
		{ "sparse_unroll", &o.sparse_unroll, nullptr },
		{ "sparse_lds", &o.sparse_lds, nullptr },
		{ "sparse_pw", &o.sparse_pw, nullptr },
		{ "sparse_spill", &o.sparse_spill, nullptr },
		{ "streams", &o.streams, nullptr },
		{ "lanes", &o.lanes, nullptr },
		{ "meet", &o.meet, nullptr },
		{ "meet_cap", &o.meet_cap, nullptr },
		{ "meet_cap_small", &o.meet_cap_small, nullptr },
		{ "meet_wide_rows", &o.meet_wide_rows, nullptr },
		{ "meet_spin_wait", &o.meet_spin_wait, nullptr },
		{ "paths_reserve_mb", &o.paths_reserve_mb, nullptr },
		{ "meet_wide_rows_always", &o.meet_wide_rows_always, nullptr },
		{ "meet4_test_cap", &o.meet4_test_cap, nullptr },
		{ "chunk_zero_copy", &o.chunk_zero_copy, nullptr },
		{ "meet_small_rows", &o.meet_small_rows, nullptr },
		{ "meet_cap_paths", &o.meet_cap_paths, nullptr },
		{ "meet4", &o.meet4, nullptr },
		{ "meet4_cap", &o.meet4_cap, nullptr },
		{ "meet4_global_mb", &o.meet4_global_mb, nullptr },
		{ "meet4_lds_kb", &o.meet4_lds_kb, nullptr },
		{ "bibfs_rows", &o.bibfs_rows, nullptr },
		{ "bibfs_rows_max", &o.bibfs_rows_max, nullptr },
		{ "wbibfs", &o.wbibfs, nullptr },
		{ "wbibfs_rows", &o.wbibfs_rows, nullptr },
		{ "wbibfs_cap", &o.wbibfs_cap, nullptr },
		{ "wbibfs_queue", &o.wbibfs_queue, nullptr },
		{ "wbibfs_far", &o.wbibfs_far, nullptr },
		{ "wbibfs_prune", &o.wbibfs_prune, nullptr },
		{ "wbibfs_mem_mb", &o.wbibfs_mem_mb, nullptr },
		{ "wbibfs_delta_div", &o.wbibfs_delta_div, nullptr },
		{ "bibfs_cap", &o.bibfs_cap, nullptr },
		{ "bibfs_grid", &o.bibfs_grid, nullptr },
		{ "bibfs_queue", &o.bibfs_queue, nullptr },
		{ "meet_bias", nullptr, &o.meet_bias },
		{ "lanes_unroll", &o.lanes_unroll, nullptr },
		{ "upload_threads", &o.upload_threads, nullptr },
		{ "upload_narrow_host", &o.upload_narrow_host, nullptr },
		{ "meet_grid_mult", &o.meet_grid_mult, nullptr },
		{ "meet4_grid_mult", &o.meet4_grid_mult, nullptr },
		{ "meet_trace", &o.meet_trace, nullptr },
		{ "meet_layout", &o.meet_layout, nullptr },
		{ "meet_align", &o.meet_align, nullptr },
		{ "ball", &o.ball, nullptr },
		{ "ball_cap", &o.ball_cap, nullptr },
		{ "ball_test_cap", &o.ball_test_cap, nullptr },
		{ "ball_sort", &o.ball_sort, nullptr },
		{ "route_timing", &o.route_timing, nullptr },
		{ "route_timing_rows", &o.route_timing_rows, nullptr },
		{ "route_try_factor", nullptr, &o.route_try_factor },
		{ "ball_seg_kb", &o.ball_seg_kb, nullptr },
		{ "ball_grid", &o.ball_grid, nullptr },
		{ "ball_seg_rows_small", &o.ball_seg_rows_small, nullptr },
		{ "ball_head_mb", &o.ball_head_mb, nullptr },
		{ "calibration_cache", &o.calibration_cache, nullptr },
		{ "block_above", &o.block_above, nullptr },
		{ "ball_bias", nullptr, &o.ball_bias },
	};
}

static void env_int(const char *name, int &dst) {
	const char *v = getenv(name);
	if (v && *v) dst = atoi(v);
}
static void env_double(const char *name, double &dst) {
	const char *v = getenv(name);
	if (v && *v) dst = atof(v);
}

static int do_init(int device) {
	std::lock_guard<std::mutex> g(g_init_lock);
	if (g_inited.load()) return PGQ_OK;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		return fail(PGQ_ERR_NO_DEVICE, std::string("libpgq_hip needs a HIP device (gfx950); hipGetDeviceCount: ") +
		                                   (e == hipSuccess ? "0 devices" : hipGetErrorString(e)));
	}
	if (device < 0) {
		const char *v = getenv("PGQ_DEVICE");
		if (!v || !*v) v = getenv("LOCAL_RANK");
		device = (v && *v) ? atoi(v) : 0;
		if (device >= count) device = device % count;
	}
	if (device >= count) return fail(PGQ_ERR_INVALID_ARG, "pgq_init: device index out of range");
	e = hipSetDevice(device);
	if (e != hipSuccess) return fail(PGQ_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
	g_device = device;
	if (g_devices.empty()) g_devices.push_back(device);
	// every option of the table can be preset from the environment: PGQ_<NAME IN CAPITALS>
	for (const OptRef &r : option_table(g_opt)) {
		std::string name = "PGQ_";
		for (const char *q = r.name; *q; q++) name += (char)toupper((unsigned char)*q);
		if (r.i) env_int(name.c_str(), *r.i);
		else env_double(name.c_str(), *r.d);
	}
	g_inited.store(1);
	return PGQ_OK;
}

int ensure_init() {
	if (!g_inited.load()) PGQ_TRY(do_init(-1));
	// every host thread that calls in (DuckDB workers) must bind the device
	hipError_t e = hipSetDevice(current_device());
	if (e != hipSuccess) return fail(PGQ_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
	return PGQ_OK;
}

// ---- kernel timers ----------------------------------------------------------------------------------------
struct PendingTimer {
	hipEvent_t a, b;
	int kclass;
};
static thread_local std::vector<PendingTimer> t_pending;
static thread_local std::vector<hipEvent_t> t_event_pool;

static hipEvent_t get_event() {
	if (!t_event_pool.empty()) {
		hipEvent_t e = t_event_pool.back();
		t_event_pool.pop_back();
		return e;
	}
	hipEvent_t e = nullptr;
	(void)hipEventCreate(&e);
	return e;
}

KernelTimer::KernelTimer(hipStream_t s, int k) : stream(s), kclass(k) {
	tstats().s.launches[k]++;
	if (!options().profile) return;
	a = get_event();
	b = get_event();
	(void)hipEventRecord(a, stream);
}
void KernelTimer::stop() {
	if (!a) return;
	(void)hipEventRecord(b, stream);
	t_pending.push_back({ a, b, kclass });
	a = b = nullptr;
}
void KernelTimer::flush() {
	for (auto &p : t_pending) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) tstats().s.kernel_ms[p.kclass] += ms;
		t_event_pool.push_back(p.a);
		t_event_pool.push_back(p.b);
	}
	t_pending.clear();
}

// ---- device memory for CSRs and upload temporaries: freed blocks are kept for the next upload ------------------
// DuckDB builds a CSR per query (CreateCsr*), so the same dozen block sizes come back every few milliseconds;
// hipMalloc/hipFree of 100+ MB blocks cost more than the kernels that fill them.  Blocks are keyed by (device, rounded
// size); at most `alloc_cache_mb` of free blocks are kept (0 disables the cache).  Callers free only memory no kernel
// still uses (every upload and search ends with a stream synchronisation), which is what hipFree's implicit
// synchronisation used to guarantee.
namespace {
struct BlockCache {
	std::mutex lock;
	std::map<std::pair<int, size_t>, std::vector<void *>> free_blocks;
	std::unordered_map<void *, std::pair<int, size_t>> live;
	size_t cached_bytes = 0;
};
BlockCache &block_cache() {
	static BlockCache *c = new BlockCache(); // never destroyed: frees may run from static destructors
	return *c;
}
size_t round_block(size_t bytes) {
	const size_t g = bytes <= (1u << 20) ? 4096 : (2u << 20);
	return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
}
} // namespace

int dev_alloc(void **out, size_t bytes) {
	*out = nullptr;
	const size_t want = round_block(bytes);
	int dev = 0;
	(void)hipGetDevice(&dev);
	BlockCache &bc = block_cache();
	{
		std::lock_guard<std::mutex> g(bc.lock);
		auto it = bc.free_blocks.find({ dev, want });
		if (it != bc.free_blocks.end() && !it->second.empty()) {
			*out = it->second.back();
			it->second.pop_back();
			bc.cached_bytes -= want;
			bc.live[*out] = { dev, want };
			return PGQ_OK;
		}
	}
	hipError_t e = hipMalloc(out, want);
	if (e == hipErrorOutOfMemory) { // give the cached blocks back to the driver and try once more (like DevBuf::reserve)
		(void)hipGetLastError();
		dev_cache_trim();
		e = hipMalloc(out, want);
	}
	if (e == hipErrorOutOfMemory) { // ... and the buffers of this device's idle pooled workspaces (label arrays of V x 576 B per relaxation stream)
		(void)hipGetLastError();
		drop_idle_workspaces();
		e = hipMalloc(out, want);
	}
	if (e != hipSuccess) {
		*out = nullptr;
		return fail(PGQ_ERR_OOM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
	}
	std::lock_guard<std::mutex> g(bc.lock);
	bc.live[*out] = { dev, want };
	return PGQ_OK;
}

void dev_free(void *p) {
	if (!p) return;
	BlockCache &bc = block_cache();
	{
		std::lock_guard<std::mutex> g(bc.lock);
		auto it = bc.live.find(p);
		if (it != bc.live.end()) {
			const std::pair<int, size_t> key = it->second;
			bc.live.erase(it);
			const size_t cap = (size_t)std::max(0, options().alloc_cache_mb) << 20;
			if (bc.cached_bytes + key.second <= cap) {
				bc.free_blocks[key].push_back(p);
				bc.cached_bytes += key.second;
				return;
			}
		}
	}
	(void)hipFree(p);
}

void dev_cache_trim() {
	BlockCache &bc = block_cache();
	std::vector<std::pair<int, void *>> blocks;
	{
		std::lock_guard<std::mutex> g(bc.lock);
		for (auto &kv : bc.free_blocks)
			for (void *p : kv.second) blocks.push_back({ kv.first.first, p });
		bc.free_blocks.clear();
		bc.cached_bytes = 0;
	}
	int cur = 0;
	(void)hipGetDevice(&cur);
	for (auto &b : blocks) {
		if (b.first != cur) (void)hipSetDevice(b.first);
		(void)hipFree(b.second);
		if (b.first != cur) (void)hipSetDevice(cur);
	}
}

int DevBuf::reserve(size_t bytes) {
	if (bytes <= cap) return PGQ_OK;
	if (p) (void)hipFree(p);
	p = nullptr;
	cap = 0;
	size_t want = bytes + bytes / 8 + 256;
	hipError_t e = hipMalloc(&p, want);
	if (e == hipErrorOutOfMemory) { // gigabytes of freed CSR blocks may sit in the block cache: give them back and retry
		(void)hipGetLastError();
		dev_cache_trim();
		e = hipMalloc(&p, want);
	}
	if (e == hipErrorOutOfMemory) { // ... and the idle pooled workspaces (tens of GB on a large-V graph after a weighted search)
		(void)hipGetLastError();
		drop_idle_workspaces();
		e = hipMalloc(&p, want);
	}
	if (e != hipSuccess) return fail(PGQ_ERR_OOM, std::string("hipMalloc(") + std::to_string(want) + "): " +
	                                                  hipGetErrorString(e));
	cap = want;
	return PGQ_OK;
}
void DevBuf::release() {
	if (p) (void)hipFree(p);
	p = nullptr;
	cap = 0;
}

// ---- UnifiedVectorFormat -> flat arrays --------------------------------------------------------------------
// The same resolution straight into caller-provided arrays (the pinned staging block of the chunk entry points), dst
// validity not consulted (iterativelength.cpp:98,122); flat vectors without selection or validity take a branch-free loop.
int flatten_pairs_into(int64_t V, int64_t n, const pgq_vec_t &src, const pgq_vec_t &dst, int64_t *out_src, int64_t *out_dst) {
	const int64_t *sd = static_cast<const int64_t *>(src.data);
	const int64_t *dd = static_cast<const int64_t *>(dst.data);
	if (n > 0 && (!sd || !dd)) return fail(PGQ_ERR_INVALID_ARG, "src/dst data pointer is NULL");
	if (!src.sel && !dst.sel && !src.validity && !dst.validity) {
		uint64_t bad = 0;
		for (int64_t r = 0; r < n; r++) {
			const int64_t s = sd[r], d = dd[r];
			bad |= (uint64_t)(s < 0) | (uint64_t)(s >= V) | (uint64_t)(d < 0) | (uint64_t)(d >= V);
			out_src[r] = s;
			out_dst[r] = d;
		}
		if (!bad) return PGQ_OK; // else: the general loop below names the offender
	}
	for (int64_t r = 0; r < n; r++) {
		const int64_t sp = src.sel ? (int64_t)src.sel[r] : r;
		const int64_t dp = dst.sel ? (int64_t)dst.sel[r] : r;
		const bool s_ok = !src.validity || ((src.validity[sp >> 6] >> (sp & 63)) & 1ULL);
		const bool d_ok = !dst.validity || ((dst.validity[dp >> 6] >> (dp & 63)) & 1ULL);
		int64_t s = s_ok ? sd[sp] : -1;
		const int64_t d = dd[dp];
		if (s_ok && (s < 0 || s >= V)) return fail(PGQ_ERR_INVALID_ARG, "source rowid out of range [0,V)");
		if (s_ok && (d < 0 || d >= V)) {
			if (d_ok) return fail(PGQ_ERR_INVALID_ARG, "destination rowid out of range [0,V)");
			s = -1; // NULL dst with garbage payload: report NULL instead of reading out of bounds
		}
		out_src[r] = s;
		out_dst[r] = d;
	}
	return PGQ_OK;
}

int flatten_pairs(int64_t V, int64_t n, const pgq_vec_t &src, const pgq_vec_t &dst, FlatPairs &out,
                  bool check_dst_validity) {
	out.src.resize(n);
	out.dst.resize(n);
	out.dst_valid.assign(n, 1);
	const int64_t *sd = static_cast<const int64_t *>(src.data);
	const int64_t *dd = static_cast<const int64_t *>(dst.data);
	if (n > 0 && (!sd || !dd)) return fail(PGQ_ERR_INVALID_ARG, "src/dst data pointer is NULL");
	for (int64_t r = 0; r < n; r++) {
		int64_t sp = src.sel ? (int64_t)src.sel[r] : r;
		int64_t dp = dst.sel ? (int64_t)dst.sel[r] : r;
		bool s_ok = !src.validity || ((src.validity[sp >> 6] >> (sp & 63)) & 1ULL);
		bool d_ok = !dst.validity || ((dst.validity[dp >> 6] >> (dp & 63)) & 1ULL);
		int64_t s = s_ok ? sd[sp] : -1;
		int64_t d = dd[dp];
		if (check_dst_validity && !d_ok) {
			out.dst_valid[r] = 0;
			d = 0; // never dereferenced as a vertex for an invalid row
		}
		if (s_ok && (s < 0 || s >= V)) return fail(PGQ_ERR_INVALID_ARG, "source rowid out of range [0,V)");
		if (s_ok && out.dst_valid[r] && (d < 0 || d >= V)) {
			// The reference never looks at dst validity (iterativelength.cpp:98,122); a NULL dst carries an
			// arbitrary payload there.  Out-of-range payloads of *valid* sources are rejected here.
			if (d_ok) return fail(PGQ_ERR_INVALID_ARG, "destination rowid out of range [0,V)");
			s = -1; // NULL dst with garbage payload: report NULL instead of reading out of bounds
		}
		out.src[r] = s;
		out.dst[r] = d;
	}
	return PGQ_OK;
}

// ---- CSR upload kernels --------------------------------------------------------------------------------------

// adjacency int64 -> int32 with the range check
__global__ void k_narrow_adj(const int64_t *__restrict__ adj64, int32_t *__restrict__ adj32, int64_t E, int64_t V,
                             int *__restrict__ bad) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < E; i += stride) {
		int64_t d = adj64[i];
		if (d < 0 || d >= V) {
			*bad = 1;
			d = 0;
		}
		adj32[i] = (int32_t)d;
	}
}

// The reverse CSR is the forward slot list sorted (stably) by destination: in-edges of a vertex come out ordered by
// (source, forward slot).  That order is what the reference's tie-breaks are stated in (the parent of a path step is
// the smallest vertex of the previous level, shortest_path.cpp:21-31; PageRank adds in-edge contributions in edge
// order, pagerank.cpp:60-66), so nothing downstream needs the forward slot itself.
// slot_src[e] = source vertex of forward slot e: a workgroup takes 256 consecutive rows and expands them
__global__ __launch_bounds__(256) void k_slot_sources(int64_t V, const int64_t *__restrict__ off, u32 *__restrict__ slot_src) {
	__shared__ int64_t s_off[257];
	for (int64_t r0 = (int64_t)blockIdx.x * 256; r0 < V; r0 += (int64_t)gridDim.x * 256) {
		const int rows = (int)std::min<int64_t>(256, V - r0);
		__syncthreads();
		for (int k = threadIdx.x; k <= rows; k += 256) s_off[k] = off[r0 + k];
		__syncthreads();
		const int64_t e0 = s_off[0], e1 = s_off[rows];
		for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
			int lo = 0, hi = rows - 1; // largest k with s_off[k] <= e
			while (lo < hi) {
				const int mid = (lo + hi + 1) >> 1;
				if (s_off[mid] <= e) lo = mid;
				else hi = mid - 1;
			}
			slot_src[e] = (u32)(r0 + lo);
		}
	}
}
// roff from the sorted destination keys: entry p closes every row in (key[p-1], key[p]]
__global__ void k_row_bounds(const u32 *__restrict__ skey, int64_t E, int64_t V, int64_t *__restrict__ roff) {
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p > E) return;
	const int64_t a = p ? (int64_t)skey[p - 1] : -1;
	const int64_t b = p < E ? (int64_t)skey[p] : V;
	for (int64_t v = a + 1; v <= b; v++) roff[v] = p;
}

__global__ void k_check_offsets(const int64_t *__restrict__ off, int64_t V, int *__restrict__ bad) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < V && off[i] > off[i + 1]) *bad = 1;
	if (i == 0 && off[0] != 0) *bad = 1;
}

template <typename T> __global__ void k_any_negative(const T *__restrict__ w, int64_t E, int *__restrict__ flag) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < E; i += stride)
		if (w[i] < (T)0) *flag = 1;
}


// ---- CSR construction on the device (create_csr_vertex/create_csr_edge equivalent) -----------------------------
__global__ void k_check_rows(const int64_t *__restrict__ src, const int64_t *__restrict__ dst, int64_t n, int64_t V,
                             u32 *__restrict__ key, u32 *__restrict__ idx, int *__restrict__ bad) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		int64_t s = src[i], d = dst[i];
		if (s < 0 || s >= V || d < 0 || d >= V) {
			*bad = 1;
			s = 0;
		}
		key[i] = (u32)s;
		idx[i] = (u32)i;
	}
}
// slot i of the CSR holds row order[i]: gather dst / edge id / weight
__global__ void k_gather_rows(const u32 *__restrict__ order, int64_t n, const int64_t *__restrict__ dst,
                              const int64_t *__restrict__ eid, const int64_t *__restrict__ w,
                              int32_t *__restrict__ adj, int64_t *__restrict__ eid_out, int64_t *__restrict__ w_out) {
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		const u32 r = order[i];
		adj[i] = (int32_t)dst[r]; // range-checked by k_check_rows
		eid_out[i] = eid ? eid[r] : (int64_t)r;
		if (w_out) w_out[i] = w[r];
	}
}

// rown[e] = index of in-slot e's owner vertex inside its bottom-up work part (0..31): lets k_pull_sparse find the
// owner of an in-edge with one coalesced byte load instead of a binary search.  One wavefront per part.
__global__ void k_fill_rown(const int64_t *__restrict__ roff, const int32_t *__restrict__ parts, int n_parts,
                            const int32_t *__restrict__ radj, uint8_t *__restrict__ rown, uint32_t *__restrict__ rpk) {
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int nwaves = (gridDim.x * blockDim.x) >> 6;
	for (int p = wave; p < n_parts; p += nwaves) {
		const int v0 = parts[2 * p], v1 = parts[2 * p + 1];
		for (int v = v0; v < v1; v++)
			for (int64_t e = roff[v] + lane; e < roff[v + 1]; e += 64) {
				rown[e] = (uint8_t)(v - v0);
				if (rpk) rpk[e] = (uint32_t)radj[e] | ((uint32_t)(v - v0) << 28); // k_pull_lanes: one word per in-slot
			}
	}
}

struct UploadTrace {
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	void mark(const char *what) {
		if (!options().trace) return;
		(void)hipDeviceSynchronize();
		auto t1 = std::chrono::steady_clock::now();
		fprintf(stderr, "[pgq] upload %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
		t0 = t1;
	}
};

static int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
	int64_t g = (n + block - 1) / block;
	if (g < 1) g = 1;
	if (g > cap) g = cap;
	return (int)g;
}

// ---- pageable host memory -> HBM through pinned staging rings, several host threads ---------------------------
// A plain hipMemcpy from DuckDB's pageable vectors stages through one pinned buffer on one thread (~2.5 GB/s
// measured: 260 ms for the 0.64 GB SF100 CSR — far more than the search itself).  Here T threads each own a slice,
// a 2-slot pinned ring and a stream; the adjacency is narrowed to int32 while it is staged (half the PCIe bytes).
// Streams of the upload / build paths are kept (per device): creating and destroying one costs milliseconds on this runtime —
// three staged arrays with two streams each + the upload's own were ~15 ms of a 27-ms SF100 upload (round 6), a constant
// that neither more filler threads nor fewer bytes moved.
struct StreamPool {
	std::mutex lock;
	std::vector<std::pair<int, hipStream_t>> idle; // (device, stream)
	int get(hipStream_t *out) {
		const int dev = current_device();
		{
			std::lock_guard<std::mutex> g(lock);
			for (size_t k = 0; k < idle.size(); k++)
				if (idle[k].first == dev) {
					*out = idle[k].second;
					idle.erase(idle.begin() + (long)k);
					return PGQ_OK;
				}
		}
		PGQ_HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
		return PGQ_OK;
	}
	void put(hipStream_t st) {
		if (!st) return;
		std::lock_guard<std::mutex> g(lock);
		if (idle.size() < 32) idle.emplace_back(current_device(), st);
		else (void)hipStreamDestroy(st);
	}
};
static StreamPool g_streams;

struct PinnedPool {
	std::mutex lock;
	std::vector<void *> free_blocks;
	static constexpr size_t kBlock = 4u << 20;
	// Blocks are carved out of LARGE pinned allocations (64 blocks = 256 MB at a time; a small first one so that a process
	// that only ever downloads a result does not pin 256 MB): tools/h2dbench copies 129 GB/s into, and 57 GB/s out of, one
	// 256-MB hipHostMalloc, while the same traffic through 77 separate 4-MB allocations ran at 28 and 21 GB/s (round 6:
	// small pinned allocations are mapped with small pages).
	size_t chunks = 0;
	void *get() {
		std::lock_guard<std::mutex> g(lock);
		if (free_blocks.empty()) {
			const size_t nb = chunks == 0 ? 8 : 64;
			void *p = nullptr;
			if (hipHostMalloc(&p, nb * kBlock) != hipSuccess) {
				if (nb == 8 || hipHostMalloc(&p, 8 * kBlock) != hipSuccess) return nullptr;
				for (size_t k = 0; k < 8; k++) free_blocks.push_back(static_cast<char *>(p) + (7 - k) * kBlock);
			} else {
				for (size_t k = 0; k < nb; k++) free_blocks.push_back(static_cast<char *>(p) + (nb - 1 - k) * kBlock);
			}
			chunks++;
		}
		void *p = free_blocks.back();
		free_blocks.pop_back();
		return p;
	}
	void put(void *p) {
		std::lock_guard<std::mutex> g(lock);
		free_blocks.push_back(p);
	}
};
static PinnedPool g_pinned;

// int64 -> int32 with the range check [0, V), V < 2^31: AVX2 where the host has it (a scalar loop ran at ~7 GB/s of input per
// thread: eight of them were the upload's bottleneck at 56 GB/s; the vector loop is bound by memory).  Returns true when
// an element was out of range.
__attribute__((target("avx2"))) static bool narrow_block_avx2(const int64_t *__restrict__ src, int32_t *__restrict__ dst, size_t cnt, int64_t V) {
	const __m256i idx = _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7);
	__m256i acc_hi = _mm256_setzero_si256(), acc_max = _mm256_setzero_si256();
	size_t i = 0;
	for (; i + 8 <= cnt; i += 8) {
		const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
		const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 4));
		const __m256i pa = _mm256_permutevar8x32_epi32(a, idx), pb = _mm256_permutevar8x32_epi32(b, idx); // low dwords in the lower half, high dwords in the upper
		const __m256i lo = _mm256_permute2x128_si256(pa, pb, 0x20), hi = _mm256_permute2x128_si256(pa, pb, 0x31);
		_mm256_storeu_si256(reinterpret_cast<__m256i *>(dst + i), lo);
		acc_hi = _mm256_or_si256(acc_hi, hi);
		acc_max = _mm256_max_epu32(acc_max, lo);
	}
	alignas(32) uint32_t h[8], m[8];
	_mm256_store_si256(reinterpret_cast<__m256i *>(h), acc_hi);
	_mm256_store_si256(reinterpret_cast<__m256i *>(m), acc_max);
	bool oob = false;
	for (int k = 0; k < 8; k++) oob |= h[k] != 0 || (int64_t)m[k] >= V; // a non-zero high dword: negative, or >= 2^32
	for (; i < cnt; i++) {
		const int64_t x = src[i];
		oob |= x < 0 || x >= V;
		dst[i] = (int32_t)x;
	}
	return oob;
}
static bool narrow_block(const int64_t *__restrict__ src, int32_t *__restrict__ dst, size_t cnt, int64_t V) {
	static const bool avx2 = __builtin_cpu_supports("avx2");
	if (avx2) return narrow_block_avx2(src, dst, cnt, V);
	bool oob = false;
	for (size_t i = 0; i < cnt; i++) {
		const int64_t x = src[i];
		oob |= x < 0 || x >= V;
		dst[i] = (int32_t)x;
	}
	return oob;
}

// mode 0: raw bytes; mode 1: int64 -> int32 narrowing with range check [0, V) (elements counted in int64s)
// Round 6.  Rounds 1-5 let every staging thread issue its own hipMemcpyAsync on its own stream: the runtime serialises them,
// more than two threads were slower (8: 25 ms, 16: 55 ms per 320 MB array), and two threads' copy loops — not PCIe — set the
// pace: 20 ms for the SF100 adjacency.  tools/h2dbench on the same box: PCIe moves 57 GB/s out of pinned memory in 4-MB
// copies on two streams, eight threads memcpy 129 GB/s.  So: `upload_threads` fillers that make NO HIP call claim blocks from
// a counter and narrow / copy them into pinned slots — block b owns slot b mod R, with R up to 96 blocks (the whole SF100
// adjacency fits: nobody waits for a slot; a larger array wraps around and a filler waits for the copy that frees its
// slot) — and ONE thread, the caller, issues the copies in block order, spinning on a per-block ready flag (a condition
// variable per hand-over cost more than the 35 us a block's DMA takes).
static int staged_upload(void *d_dst, const void *h_src, size_t n_elems, size_t elem_bytes, int mode, int64_t V,
                         std::atomic<int> *bad) {
	if (n_elems == 0) return PGQ_OK;
	const size_t out_elem = mode == 1 ? 4 : elem_bytes;
	const size_t per_block = PinnedPool::kBlock / out_elem; // a pinned block is filled to the brim: 4 MB per copy
	const size_t nblocks = (n_elems + per_block - 1) / per_block;
	const int T = (int)std::min<size_t>(std::min<size_t>((size_t)std::max(1, options().upload_threads),
	                                                   std::max(1u, std::thread::hardware_concurrency())), nblocks);
	const size_t R = std::min<size_t>(nblocks, 96);
	static const int diag = getenv("PGQ_UPLOAD_DIAG") ? atoi(getenv("PGQ_UPLOAD_DIAG")) : 0; // 1: no copies, 2: no filling (timing experiments)
	PGQ_TRY(ensure_init());
	hipStream_t st[2] = { nullptr, nullptr };
	std::vector<void *> slots(R, nullptr);
	std::vector<hipEvent_t> evs(nblocks > R ? R : 0, nullptr);
	int rc = PGQ_OK;
	for (int k = 0; k < 2 && rc == PGQ_OK; k++)
		rc = g_streams.get(&st[k]);
	for (size_t k = 0; k < R && rc == PGQ_OK; k++) {
		slots[k] = g_pinned.get();
		if (!slots[k]) rc = fail(PGQ_ERR_OOM, "hipHostMalloc of a staging block failed");
	}
	for (size_t k = 0; k < evs.size() && rc == PGQ_OK; k++)
		if (hipEventCreateWithFlags(&evs[k], hipEventDisableTiming) != hipSuccess) rc = fail(PGQ_ERR_HIP, "hipEventCreate failed");
	std::unique_ptr<std::atomic<unsigned char>[]> ready(new std::atomic<unsigned char>[nblocks]);
	for (size_t b = 0; b < nblocks; b++) ready[b].store(0, std::memory_order_relaxed);
	size_t next_issue_bound = 0; // copies issued so far (the issuer's own)
	std::atomic<size_t> next_block { 0 }, copied { 0 }; // copied: blocks whose DMA has landed (their slots may be refilled)
	std::atomic<bool> stop { false };
	// Nobody spins: the boxes this runs on give a process a CPU quota, and a spinning issuer beside eight fillers burnt it
	// (fill rate 56 -> 17 GB/s).  The issuer sleeps on `cv` until the block it wants is ready (a filler notifies only when the
	// issuer has said it is waiting for that block); fillers that wait for a slot sleep on `cv_slot` until a copy has landed.
	std::mutex m;
	std::condition_variable cv, cv_slot;
	size_t waiting_for = (size_t)-1;
	auto filler = [&]() {
		for (;;) {
			const size_t b = next_block.fetch_add(1);
			if (b >= nblocks) return;
			if (b >= R && copied.load(std::memory_order_acquire) + R <= b) { // the slot still holds block b - R
				std::unique_lock<std::mutex> lk(m);
				cv_slot.wait(lk, [&] { return stop.load() || copied.load(std::memory_order_acquire) + R > b; });
				if (stop.load()) return;
			}
			const size_t lo = b * per_block, cnt = std::min(per_block, n_elems - lo);
			void *slot = slots[b % R];
			if (diag == 2) {
			} else if (mode == 1) {
				if (narrow_block(static_cast<const int64_t *>(h_src) + lo, static_cast<int32_t *>(slot), cnt, V)) bad->store(1);
			} else {
				memcpy(slot, static_cast<const char *>(h_src) + lo * elem_bytes, cnt * elem_bytes);
			}
			ready[b].store(1, std::memory_order_release);
			bool wake = false;
			{
				std::lock_guard<std::mutex> lk(m);
				wake = waiting_for == b;
			}
			if (wake) cv.notify_one();
		}
	};
	std::vector<std::thread> pool;
	if (rc == PGQ_OK)
		for (int t = 0; t < T; t++) pool.emplace_back(filler);
	size_t landed = 0; // blocks known to have landed (the issuer's view of `copied`)
	auto land = [&](bool block_on_oldest) { // returns the slots of copies that have landed
		const size_t before = landed;
		if (block_on_oldest) {
			(void)hipEventSynchronize(evs[landed % R]);
			landed++;
		}
		while (landed < next_issue_bound && hipEventQuery(evs[landed % R]) == hipSuccess) landed++;
		if (landed != before) {
			{
				std::lock_guard<std::mutex> lk(m);
				copied.store(landed, std::memory_order_release);
			}
			cv_slot.notify_all();
		}
	};
	for (size_t b = 0; b < nblocks && rc == PGQ_OK; b++) {
		next_issue_bound = b;
		while (!ready[b].load(std::memory_order_acquire)) {
			if (!evs.empty()) land(false);
			std::unique_lock<std::mutex> lk(m);
			waiting_for = b;
			cv.wait_for(lk, std::chrono::microseconds(evs.empty() ? 2000 : 100), [&] { return ready[b].load(std::memory_order_acquire) != 0; });
			waiting_for = (size_t)-1;
		}
		const size_t lo = b * per_block, cnt = std::min(per_block, n_elems - lo);
		hipStream_t s = st[b & 1];
		if (diag != 1 && hipMemcpyAsync(static_cast<char *>(d_dst) + lo * out_elem, slots[b % R], cnt * out_elem, hipMemcpyHostToDevice, s) != hipSuccess)
			rc = fail(PGQ_ERR_HIP, "hipMemcpyAsync (staged upload) failed");
		if (!evs.empty() && rc == PGQ_OK && hipEventRecord(evs[b % R], s) != hipSuccess) rc = fail(PGQ_ERR_HIP, "hipEventRecord failed");
		next_issue_bound = b + 1;
		if (!evs.empty()) land(b + 1 >= R + landed); // the ring is full of copies in flight: wait for the oldest
	}
	{
		std::lock_guard<std::mutex> lk(m);
		stop.store(true);
	}
	cv_slot.notify_all();
	for (auto &th : pool) th.join();
	for (int k = 0; k < 2; k++)
		if (st[k] && hipStreamSynchronize(st[k]) != hipSuccess && rc == PGQ_OK) rc = fail(PGQ_ERR_HIP, "staged upload: stream failed");
	for (auto e : evs)
		if (e) (void)hipEventDestroy(e);
	for (size_t k = 0; k < R; k++)
		if (slots[k]) g_pinned.put(slots[k]);
	for (int k = 0; k < 2; k++) g_streams.put(st[k]); // (synchronised above)
	return rc;
}

// device -> pageable host through one pinned block at a time (a pageable hipMemcpy D2H ran at ~0.6 GB/s here)
int staged_download(void *h_dst, const void *d_src, size_t bytes, hipStream_t st) {
	void *blk = g_pinned.get();
	if (!blk) return fail(PGQ_ERR_OOM, "hipHostMalloc of a staging block failed");
	int rc = PGQ_OK;
	for (size_t lo = 0; lo < bytes && rc == PGQ_OK; lo += PinnedPool::kBlock) {
		const size_t cnt = std::min(PinnedPool::kBlock, bytes - lo);
		if (hipMemcpyAsync(blk, static_cast<const char *>(d_src) + lo, cnt, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess)
			rc = fail(PGQ_ERR_HIP, "staged download failed");
		else memcpy(static_cast<char *>(h_dst) + lo, blk, cnt);
	}
	g_pinned.put(blk);
	return rc;
}

// ---- hub list, degree statistics and the bottom-up work partition, on the device -----------------------------------
struct UploadStats {
	unsigned long long two_hop_sum; // sum over v of in-degree x out-degree
	int max_in, max_out, n_hubs, n_parts, negative_weight, pad;
};
struct HubRow {
	int64_t vertex, begin, end;
};
__global__ __launch_bounds__(256) void k_degree_stats(int64_t V, const int64_t *__restrict__ off,
                                                      const int64_t *__restrict__ roff, int64_t chunk, int hub_cap,
                                                      UploadStats *__restrict__ st, HubRow *__restrict__ hubs) {
	__shared__ unsigned long long s_sum[4];
	__shared__ int s_in[4], s_out[4];
	unsigned long long sum = 0;
	int max_in = 0, max_out = 0;
	for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
		const int64_t rb = roff[v], re = roff[v + 1];
		const int64_t indeg = re - rb, outdeg = off[v + 1] - off[v];
		sum += (unsigned long long)indeg * (unsigned long long)outdeg;
		max_in = max(max_in, (int)indeg);
		max_out = max(max_out, (int)outdeg);
		if (indeg > chunk) { // rare: a few atomics per graph
			const int k = atomicAdd(&st->n_hubs, 1);
			if (k < hub_cap) hubs[k] = { v, rb, re };
		}
	}
	for (int sh = 32; sh; sh >>= 1) {
		sum += __shfl_xor(sum, sh);
		max_in = max(max_in, __shfl_xor(max_in, sh));
		max_out = max(max_out, __shfl_xor(max_out, sh));
	}
	const int w = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) {
		s_sum[w] = sum;
		s_in[w] = max_in;
		s_out[w] = max_out;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd(&st->two_hop_sum, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
		atomicMax(&st->max_in, max(max(s_in[0], s_in[1]), max(s_in[2], s_in[3])));
		atomicMax(&st->max_out, max(max(s_out[0], s_out[1]), max(s_out[2], s_out[3])));
	}
}
// Work partition for the bottom-up kernels: contiguous vertex ranges [begin,end) that never contain a hub, hold at most
// 16 vertices (LDS accumulator rows of k_pull_sparse) and at most `wmax` of (in-degree + 8 per vertex).  A thread packs
// one run of kPartRun consecutive vertices greedily; EMIT = false counts the parts of each run, EMIT = true writes
// them at the scanned base.
constexpr int kPartRun = 256;
template <bool EMIT>
__global__ void k_make_parts(int64_t V, const int64_t *__restrict__ roff, int64_t chunk, int64_t wmax,
                             int *__restrict__ run_count, const int *__restrict__ run_base, int32_t *__restrict__ parts,
                             UploadStats *__restrict__ st) {
	const int64_t run = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t v0 = run * kPartRun;
	if (v0 >= V) return;
	const int64_t v1 = min(V, v0 + kPartRun);
	int n = 0;
	const int base = EMIT ? run_base[run] : 0;
	int64_t begin = v0, acc = 0;
	auto close = [&](int64_t end) {
		if (end > begin) {
			if (EMIT) {
				parts[2 * (base + n)] = (int32_t)begin;
				parts[2 * (base + n) + 1] = (int32_t)end;
			}
			n++;
		}
		begin = end;
		acc = 0;
	};
	int64_t r = roff[v0];
	for (int64_t v = v0; v < v1; v++) {
		const int64_t rn = roff[v + 1], d = rn - r;
		r = rn;
		if (d > chunk) { // hub: handled by k_pull_hub, never inside a part
			close(v);
			begin = v + 1;
			continue;
		}
		const int64_t wv = d + 8;
		if (v > begin && (acc + wv > wmax || v - begin >= 16)) close(v);
		acc += wv;
	}
	close(v1);
	if (!EMIT) run_count[run] = n;
	else if (v1 == V) st->n_parts = base + n;
}

// ---- layout of the pair-centric kernels: padded adjacency + slot descriptors (pgq_walk.h) ------------------------------
// groups (16 bytes = 4 entries) a vertex's padded list occupies: its entries rounded up to `align` entries
__global__ void k_seg_groups(int64_t V, const int64_t *__restrict__ off, u32 align, u32 *__restrict__ ng) {
	const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= V) return;
	const u32 len = (u32)(off[v + 1] - off[v]);
	ng[v] = ((len + align - 1) / align) * (align >> 2);
}
__global__ void k_seg_fill(int64_t V, const int64_t *__restrict__ off, const u32 *__restrict__ gbeg, uint2 *__restrict__ seg) {
	const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (v < V) seg[v] = make_uint2(gbeg[v], (u32)(off[v + 1] - off[v]));
}
// One workgroup copies the lists of 256 consecutive vertices: their padded groups are one contiguous range, a thread
// writes whole groups (coalesced) and finds a group's vertex by binary search over the 257 group starts in LDS.
// Positions past a list's end repeat its last entry.
__global__ __launch_bounds__(256) void k_fill_padded(int64_t V, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                     const uint2 *__restrict__ seg, u32 total_groups,
                                                     int32_t *__restrict__ padj) {
	__shared__ u32 s_beg[257];
	__shared__ int64_t s_off[257];
	const int64_t v0 = (int64_t)blockIdx.x * 256;
	const int nv = (int)min((int64_t)256, V - v0);
	for (int t = threadIdx.x; t <= nv; t += 256) {
		s_beg[t] = (v0 + t < V) ? seg[v0 + t].x : total_groups;
		s_off[t] = off[v0 + t];
	}
	__syncthreads();
	const u32 g0 = s_beg[0], g1 = s_beg[nv];
	for (u32 g = g0 + threadIdx.x; g < g1; g += 256) {
		int lo = 0, hi = nv; // largest t with s_beg[t] <= g; lists without groups share their start with the next one
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (s_beg[mid] <= g) lo = mid;
			else hi = mid;
		}
		const int64_t b = s_off[lo], len = s_off[lo + 1] - b;
		const int64_t i0 = (int64_t)(g - s_beg[lo]) * 4;
		int4 o;
		o.x = adj[b + min(i0, len - 1)];
		o.y = adj[b + min(i0 + 1, len - 1)];
		o.z = adj[b + min(i0 + 2, len - 1)];
		o.w = adj[b + min(i0 + 3, len - 1)];
		reinterpret_cast<int4 *>(padj)[g] = o;
	}
}
__global__ void k_fill_desc(int64_t E, const int32_t *__restrict__ adj, const uint2 *__restrict__ seg, uint4 *__restrict__ desc) {
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= E) return;
	const u32 u = (u32)adj[e];
	const uint2 s = seg[u];
	desc[e] = make_uint4(u, s.x, s.y, 0u);
}

// entries of every vertex's two-hop walk in one direction = sum of its neighbours' list lengths (saturating u32).  One
// workgroup per 256 consecutive vertices: their slots are one contiguous range, read coalesced; a slot's vertex by
// binary search over the 257 offsets in LDS.  The 64 consecutive slots a wavefront holds belong to a few vertices (to ONE
// inside a hub's list): a segmented scan over the lanes adds them up first and only the last lane of a run touches the
// LDS accumulator — one atomic per slot serialised on a hub's single address (R-MAT-22: 9 ms per direction).
__global__ __launch_bounds__(256) void k_two_hop_work(int64_t V, const int64_t *__restrict__ off, const int32_t *__restrict__ adj,
                                                      const uint2 *__restrict__ seg, u32 *__restrict__ work) {
	__shared__ int64_t s_off[257];
	__shared__ unsigned long long s_w[256];
	const int64_t v0 = (int64_t)blockIdx.x * 256;
	const int nv = (int)min((int64_t)256, V - v0);
	for (int t = threadIdx.x; t <= nv; t += 256) s_off[t] = off[v0 + t];
	s_w[threadIdx.x] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int64_t e_end = s_off[nv];
	for (int64_t base = s_off[0] + (threadIdx.x & ~63); base < e_end; base += 256) { // wave-uniform bound: shuffles stay converged
		const int64_t e = base + lane;
		int lo = -1;
		unsigned long long val = 0;
		if (e < e_end) {
			lo = 0;
			int hi = nv; // largest t with s_off[t] <= e
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (s_off[mid] <= e) lo = mid;
				else hi = mid;
			}
			val = (unsigned long long)seg[(u32)adj[e]].y;
		}
		// inclusive segmented scan keyed by the vertex (runs are contiguous: slots ascend with the vertex)
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned long long up = __shfl_up(val, o);
			const int klo = __shfl_up(lo, o);
			if (lane >= o && klo == lo) val += up;
		}
		const int next = __shfl_down(lo, 1);
		if (lo >= 0 && (lane == 63 || next != lo)) atomicAdd(&s_w[lo], val);
	}
	__syncthreads();
	if ((int)threadIdx.x < nv) work[v0 + threadIdx.x] = (u32)min(s_w[threadIdx.x], 0xFFFFFFFFull);
}

static int build_meet_layout_dir(pgq_csr *c, const int64_t *off, const int32_t *adj, int32_t **padj, uint2 **seg,
                                 uint4 **desc, u32 **work, int64_t *groups_out, hipStream_t st) {
	const int64_t V = c->V, E = c->E;
	u32 align = (u32)std::max(4, options().meet_align) & ~3u;
	// group indices are 32-bit: E / 4 + V x align / 4 bounds the padded size; one group per list start always fits
	if ((double)E / 4.0 + (double)V * (align / 4.0) >= 4.0e9) align = 4;
	u32 *d_ng = nullptr, *d_gb = nullptr;
	void *d_tmp = nullptr;
	struct Temps {
		u32 *&a, *&b;
		void *&t;
		hipStream_t st;
		~Temps() {
			(void)hipStreamSynchronize(st); // the kernels below may still read them on an error return
			dev_free(a);
			dev_free(b);
			dev_free(t);
		}
	} temps { d_ng, d_gb, d_tmp, st };
	PGQ_TRY(dev_alloc_as(&d_ng, (size_t)V + 1));
	PGQ_TRY(dev_alloc_as(&d_gb, (size_t)V + 1));
	PGQ_HIP_TRY(hipMemsetAsync(d_ng + V, 0, sizeof(u32), st));
	hipLaunchKernelGGL(k_seg_groups, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, off, align, d_ng);
	size_t sb = 0;
	PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, d_ng, d_gb, (int)(V + 1), st));
	PGQ_TRY(dev_alloc(&d_tmp, sb + 16));
	PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, sb, d_ng, d_gb, (int)(V + 1), st));
	u32 total = 0;
	PGQ_HIP_TRY(hipMemcpyAsync(&total, d_gb + V, sizeof(u32), hipMemcpyDeviceToHost, st));
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	PGQ_TRY(dev_alloc_as(seg, (size_t)V));
	PGQ_TRY(dev_alloc_as(padj, (size_t)total * 4 + 4));
	PGQ_TRY(dev_alloc_as(desc, (size_t)E + 1));
	hipLaunchKernelGGL(k_seg_fill, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, off, d_gb, *seg);
	hipLaunchKernelGGL(k_fill_padded, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, off, adj, *seg, total, *padj);
	hipLaunchKernelGGL(k_fill_desc, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, E, adj, *seg, *desc);
	PGQ_TRY(dev_alloc_as(work, (size_t)V));
	hipLaunchKernelGGL(k_two_hop_work, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, V, off, adj, *seg, *work);
	*groups_out = (int64_t)total;
	return PGQ_OK;
}
// rhead (pgq_internal.h): 64 words per vertex = two 128-byte lines.  Word 31 — the last of the FIRST line — is the in-degree;
// words 0..30 are the list's entries 0..30, words 32..63 its entries 31..62; positions past the list's end repeat its last
// entry (an empty list: all ones, never tested because its count is 0).  One thread per 16-byte group.
__global__ __launch_bounds__(256) void k_fill_rhead(int64_t V, const uint2 *__restrict__ seg, const int32_t *__restrict__ padj,
                                                    uint4 *__restrict__ head) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int64_t v = t >> 4;
	const u32 j = (u32)(t & 15);
	if (v >= V) return;
	const uint2 s = seg[v];
	const u32 cnt = s.y;
	u32 x[4];
#pragma unroll
	for (u32 k = 0; k < 4; k++) {
		const u32 w = 4u * j + k;
		if (w == 31u) {
			x[k] = cnt;
		} else if (cnt == 0u) {
			x[k] = 0xFFFFFFFFu;
		} else {
			const u32 e = min(w < 31u ? w : w - 1u, cnt - 1u);
			x[k] = (u32)padj[(size_t)s.x * 4 + e];
		}
	}
	head[v * 16 + j] = make_uint4(x[0], x[1], x[2], x[3]);
}
static int build_meet_layout(pgq_csr *c, hipStream_t st) {
	if (!options().meet_layout || c->V <= 0 || c->E <= 0) return PGQ_OK;
	PGQ_TRY(build_meet_layout_dir(c, c->off, c->adj, &c->padj, &c->fseg, &c->fdesc, &c->fwork, &c->padj_groups, st));
	PGQ_TRY(build_meet_layout_dir(c, c->roff, c->radj, &c->rpadj, &c->rseg, &c->rdesc, &c->rwork, &c->rpadj_groups, st));
	if (options().ball && (size_t)c->V * 256 <= ((size_t)std::max(0, options().ball_head_mb) << 20)) {
		if (dev_alloc_as(&c->rhead, (size_t)c->V * 16) == PGQ_OK) // (no memory for it: the kernel gathers the list positions instead)
			hipLaunchKernelGGL(k_fill_rhead, dim3((unsigned)((c->V * 16 + 255) / 256)), dim3(256), 0, st, c->V, c->rseg, c->rpadj, c->rhead);
		else
			c->rhead = nullptr;
	}
	return PGQ_OK;
}

// Builds everything derived from (off, adj64) that already sit in device memory.
static int finish_upload(pgq_csr *c, const int64_t *d_adj64, hipStream_t st) { // d_adj64 == nullptr: c->adj is set
	const int64_t V = c->V, E = c->E;
	const size_t En = (size_t)std::max<int64_t>(E, 1);
	UploadTrace tr;
	int *d_flag = nullptr;
	u32 *d_slot_src = nullptr, *d_skey = nullptr;
	void *d_tmp = nullptr;
	struct Temps {
		int *&flag;
		u32 *&a, *&b;
		void *&t;
		hipStream_t st;
		~Temps() {
			(void)hipStreamSynchronize(st); // an error return may leave a sort writing them: not back to the cache before it is done
			dev_free(flag);
			dev_free(a);
			dev_free(b);
			dev_free(t);
		}
	} temps { d_flag, d_slot_src, d_skey, d_tmp, st };
	PGQ_TRY(dev_alloc_as(&d_flag, 2));
	PGQ_HIP_TRY(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), st));
	if (V > 0) {
		hipLaunchKernelGGL(k_check_offsets, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, c->off, V, d_flag);
	}
	// +4 entries: k_meet3 / k_pull_sparse read the adjacencies as aligned 16-byte groups
	if (!c->adj) PGQ_TRY(dev_alloc_as(&c->adj, En + 4));
	PGQ_TRY(dev_alloc_as(&c->radj, En + 4));
	PGQ_TRY(dev_alloc_as(&c->roff, (size_t)V + 1));
	if (E > 0 && d_adj64) hipLaunchKernelGGL(k_narrow_adj, dim3(grid_for(E)), dim3(256), 0, st, d_adj64, c->adj, E, V, d_flag);
	{ // the kernels below index by offsets and adjacency values: stop here if either is malformed
		int h_bad = 0;
		PGQ_HIP_TRY(hipMemcpyAsync(&h_bad, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
		PGQ_HIP_TRY(hipStreamSynchronize(st));
		if (h_bad) return fail(PGQ_ERR_INVALID_ARG, "CSR is malformed: offsets not monotone or adjacency out of [0,V)");
	}
	tr.mark("allocs + narrow");
	if (E > 0) {
		PGQ_TRY(dev_alloc_as(&d_slot_src, En));
		PGQ_TRY(dev_alloc_as(&d_skey, En));
		hipLaunchKernelGGL(k_slot_sources, dim3(grid_for(V, 256, 256 * 8)), dim3(256), 0, st, V, c->off, d_slot_src);
		int end_bit = 1;
		while ((1LL << end_bit) < V) end_bit++;
		size_t sb = 0;
		const u32 *keys = reinterpret_cast<const u32 *>(c->adj); // range-checked: every key is in [0,V)
		u32 *vals = reinterpret_cast<u32 *>(c->radj);
		PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, keys, d_skey, d_slot_src, vals, (int)E, 0, end_bit, st));
		PGQ_TRY(dev_alloc(&d_tmp, sb + 16));
		PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, sb, keys, d_skey, d_slot_src, vals, (int)E, 0, end_bit, st));
		hipLaunchKernelGGL(k_row_bounds, dim3((unsigned)((E + 1 + 255) / 256)), dim3(256), 0, st, d_skey, E, V, c->roff);
	} else {
		PGQ_HIP_TRY(hipMemsetAsync(c->roff, 0, (size_t)(V + 1) * sizeof(int64_t), st));
	}
	if (c->w && E > 0) {
		if (c->w_type == PGQ_W_INT64)
			hipLaunchKernelGGL(k_any_negative<int64_t>, dim3(grid_for(E)), dim3(256), 0, st, (const int64_t *)c->w, E,
			                   d_flag + 1);
		else
			hipLaunchKernelGGL(k_any_negative<double>, dim3(grid_for(E)), dim3(256), 0, st, (const double *)c->w, E,
			                   d_flag + 1);
	}
	tr.mark("reverse CSR (sort by destination)");
	// hubs, degree statistics and the work partition (device); only the counts and the hub rows come back
	const int64_t chunk = std::max(64, options().hub_chunk);
	c->hub_threshold = chunk;
	const int hub_cap = (int)(E / chunk + 1); // a hub has more than `chunk` in-edges
	const int64_t n_runs = (V + kPartRun - 1) / kPartRun;
	UploadStats *d_us = nullptr;
	HubRow *d_hub_rows = nullptr;
	int *d_run = nullptr; // [0,n_runs] counts, [n_runs+1, 2 n_runs+1] bases
	void *d_scan = nullptr;
	struct Temps2 {
		UploadStats *&a;
		HubRow *&b;
		int *&c;
		void *&d;
		hipStream_t st;
		~Temps2() {
			(void)hipStreamSynchronize(st);
			dev_free(a);
			dev_free(b);
			dev_free(c);
			dev_free(d);
		}
	} temps2 { d_us, d_hub_rows, d_run, d_scan, st };
	PGQ_TRY(dev_alloc_as(&d_us, 1));
	PGQ_TRY(dev_alloc_as(&d_hub_rows, (size_t)hub_cap));
	PGQ_TRY(dev_alloc_as(&d_run, (size_t)(2 * n_runs + 2)));
	PGQ_TRY(dev_alloc_as(&c->pull_parts, (size_t)std::max<int64_t>(2 * V, 2)));
	PGQ_TRY(dev_alloc((void **)&c->rown, En + 8)); // read as aligned 4-byte groups
	PGQ_HIP_TRY(hipMemsetAsync(d_us, 0, sizeof(UploadStats), st));
	PGQ_HIP_TRY(hipMemsetAsync(c->rown, 0, En + 8, st));
	if (V < (1ll << 28)) { // in-slots of hubs keep 0 (never read); padded for the 4 x 64-entry trips of k_pull_lanes
		PGQ_TRY(dev_alloc_as(&c->rpk, En + 2048));
		PGQ_HIP_TRY(hipMemsetAsync(c->rpk, 0, (En + 2048) * sizeof(uint32_t), st));
	}
	if (V > 0) {
		const int64_t wmax = std::max(64, options().part_weight);
		hipLaunchKernelGGL(k_degree_stats, dim3(grid_for(V, 256, 1024)), dim3(256), 0, st, V, c->off, c->roff, chunk, hub_cap,
		                   d_us, d_hub_rows);
		hipLaunchKernelGGL(k_make_parts<false>, dim3((unsigned)((n_runs + 63) / 64)), dim3(64), 0, st, V, c->roff, chunk,
		                   wmax, d_run, (const int *)nullptr, (int32_t *)nullptr, d_us);
		size_t sb = 0;
		PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, d_run, d_run + n_runs + 1, (int)n_runs, st));
		PGQ_TRY(dev_alloc(&d_scan, sb + 16));
		PGQ_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan, sb, d_run, d_run + n_runs + 1, (int)n_runs, st));
		hipLaunchKernelGGL(k_make_parts<true>, dim3((unsigned)((n_runs + 63) / 64)), dim3(64), 0, st, V, c->roff, chunk, wmax,
		                   d_run, d_run + n_runs + 1, c->pull_parts, d_us);
	}
	UploadStats us;
	int h_flag[2] = { 0, 0 };
	PGQ_HIP_TRY(hipMemcpyAsync(&us, d_us, sizeof(us), hipMemcpyDeviceToHost, st));
	PGQ_HIP_TRY(hipMemcpyAsync(h_flag, d_flag, sizeof(h_flag), hipMemcpyDeviceToHost, st));
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	if (h_flag[0]) return fail(PGQ_ERR_INVALID_ARG, "CSR is malformed: offsets not monotone or adjacency out of [0,V)");
	c->has_negative_weight = h_flag[1] != 0;
	c->max_in_degree = us.max_in;
	c->max_out_degree = us.max_out;
	c->two_hop_mean = (double)us.two_hop_sum / (double)std::max<int64_t>(V, 1);
	calibration_load(c); // (max degrees and E are set above)
	c->n_pull_parts = us.n_parts;
	if (us.n_parts > 0)
		hipLaunchKernelGGL(k_fill_rown, dim3((unsigned)device_cus() * 8), dim3(256), 0, st, c->roff, c->pull_parts, c->n_pull_parts, c->radj,
		                   c->rown, c->rpk);
	tr.mark("degree statistics, parts, owner bytes");
	int64_t n_items = 0;
	if (us.n_hubs > 0) { // few rows: slice them on the host, in vertex order
		std::vector<HubRow> rows((size_t)std::min(us.n_hubs, hub_cap));
		PGQ_HIP_TRY(hipMemcpyAsync(rows.data(), d_hub_rows, rows.size() * sizeof(HubRow), hipMemcpyDeviceToHost, st));
		PGQ_HIP_TRY(hipStreamSynchronize(st));
		std::sort(rows.begin(), rows.end(), [](const HubRow &a, const HubRow &b) { return a.vertex < b.vertex; });
		std::vector<HubItem> items;
		std::vector<int32_t> hubs;
		// slices of a quarter chunk: a wavefront walks its slice 64 entries at a time (latency-bound), so more, shorter
		// slices keep more of a hub's in-edges in flight
		const int64_t slice = std::max<int64_t>(64, chunk / 4);
		for (const HubRow &h : rows) {
			hubs.push_back((int32_t)h.vertex);
			for (int64_t b = h.begin; b < h.end; b += slice) items.push_back({ (int32_t)h.vertex, 0, b, std::min(b + slice, h.end) });
		}
		c->n_pull_hub_items = (int64_t)items.size();
		c->n_pull_hub_vertices = (int64_t)hubs.size();
		n_items = (int64_t)items.size();
		PGQ_TRY(dev_alloc((void **)&c->pull_hubs, items.size() * sizeof(HubItem)));
		PGQ_HIP_TRY(hipMemcpyAsync(c->pull_hubs, items.data(), items.size() * sizeof(HubItem), hipMemcpyHostToDevice, st));
		PGQ_TRY(dev_alloc((void **)&c->pull_hub_vertices, hubs.size() * sizeof(int32_t)));
		PGQ_HIP_TRY(hipMemcpyAsync(c->pull_hub_vertices, hubs.data(), hubs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
		PGQ_HIP_TRY(hipStreamSynchronize(st)); // the host vectors go out of scope
	}
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	tr.mark("hub slices");
	if (const int lrc = build_meet_layout(c, st); lrc != PGQ_OK) {
		// ~40 B per edge that only the pair-centric pre-pass needs: OUT OF MEMORY here is "no pre-pass" (the searches test
		// fdesc; pgq_csr_has_prepass_layout() says so), not a failed upload.  Anything else — a failed launch, a device
		// fault in the scans — is an error of the upload like any other step's
		(void)hipStreamSynchronize(st);
		if (lrc != PGQ_ERR_OOM) return lrc;
		(void)hipGetLastError();
		if (options().trace) fprintf(stderr, "[pgq] upload: no memory for the pair-centric layout (~40 B per edge): this CSR is searched without the pre-pass\n");
		for (void **q : { (void **)&c->padj, (void **)&c->rpadj, (void **)&c->fseg, (void **)&c->rseg, (void **)&c->fdesc,
		                  (void **)&c->rdesc, (void **)&c->fwork, (void **)&c->rwork, (void **)&c->rhead }) {
			dev_free(*q);
			*q = nullptr;
		}
		c->padj_groups = c->rpadj_groups = 0;
	}
	PGQ_HIP_TRY(hipStreamSynchronize(st));
	tr.mark("padded adjacency + slot descriptors");
	c->bytes = (V + 1) * 16 + 8 * V + E * (4 + 4 + 1 + (c->rpk ? 4 : 0)) + (c->edge_ids ? E * 8 : 0) + (c->w ? E * 8 : 0) +
	           n_items * (int64_t)sizeof(HubItem) +
	           (c->fdesc ? 2 * E * 16 + 2 * V * 12 + (c->padj_groups + c->rpadj_groups) * 16 : 0) + (c->rhead ? V * 256 : 0);
	return PGQ_OK;
}

// ---- calibration kept across handles (pgq_internal.h) -------------------------------------------------------------------
struct CalEntry {
	int64_t V, E, max_out, max_in;
	double two_hop_mean;
	double meet_bpr, ball_open_frac, route_ball_ns, route_lanes_ns;
	int route_try_lanes, route_ball_samples, route_lanes_samples;
	int64_t route_rows;
	std::vector<uint8_t> level_plan[6];
};
static std::mutex g_cal_lock;
static std::vector<CalEntry> g_cal; // a handful of graph shapes, the most recent last
static bool cal_same(const CalEntry &e, const pgq_csr *c) {
	return e.V == c->V && e.E == c->E && e.max_out == c->max_out_degree && e.max_in == c->max_in_degree && e.two_hop_mean == c->two_hop_mean;
}
void calibration_load(pgq_csr *c) {
	if (!c || !options().calibration_cache) return;
	std::lock_guard<std::mutex> g(g_cal_lock);
	for (const CalEntry &e : g_cal)
		if (cal_same(e, c)) {
			c->meet_bpr.store(e.meet_bpr, std::memory_order_relaxed);
			c->ball_open_frac.store(e.ball_open_frac, std::memory_order_relaxed);
			c->route_ball_ns.store(e.route_ball_ns, std::memory_order_relaxed);
			c->route_lanes_ns.store(e.route_lanes_ns, std::memory_order_relaxed);
			c->route_try_lanes.store(e.route_try_lanes, std::memory_order_relaxed);
			c->route_ball_samples.store(e.route_ball_samples, std::memory_order_relaxed);
			c->route_lanes_samples.store(e.route_lanes_samples, std::memory_order_relaxed);
			c->route_rows.store(e.route_rows, std::memory_order_relaxed);
			std::lock_guard<std::mutex> g2(c->plan_lock);
			for (int k = 0; k < 6; k++) c->level_plan[k] = e.level_plan[k];
			return;
		}
}
void calibration_store(pgq_csr *c) {
	if (!c || c->is_replica || !options().calibration_cache) return;
	CalEntry n { c->V, c->E, c->max_out_degree, c->max_in_degree, c->two_hop_mean, c->meet_bpr.load(std::memory_order_relaxed),
		         c->ball_open_frac.load(std::memory_order_relaxed), c->route_ball_ns.load(std::memory_order_relaxed),
		         c->route_lanes_ns.load(std::memory_order_relaxed), c->route_try_lanes.load(std::memory_order_relaxed),
		         c->route_ball_samples.load(std::memory_order_relaxed), c->route_lanes_samples.load(std::memory_order_relaxed),
		         c->route_rows.load(std::memory_order_relaxed), {} };
	bool any = n.meet_bpr > 0 || n.ball_open_frac > 0 || n.route_ball_ns > 0;
	{
		std::lock_guard<std::mutex> g2(c->plan_lock);
		for (int k = 0; k < 6; k++) {
			n.level_plan[k] = c->level_plan[k];
			any = any || !n.level_plan[k].empty();
		}
	}
	if (!any) return;
	std::lock_guard<std::mutex> g(g_cal_lock);
	for (size_t k = 0; k < g_cal.size(); k++)
		if (cal_same(g_cal[k], c)) {
			g_cal.erase(g_cal.begin() + (long)k);
			break;
		}
	if (g_cal.size() >= 16) g_cal.erase(g_cal.begin());
	g_cal.push_back(std::move(n));
}

static void destroy_csr(pgq_csr *c) {
	if (!c) return;
	calibration_store(c);
	if (!c->is_replica) {
		for (pgq_csr *r : c->replicas)
			if (r && r != c) destroy_csr(r);
		for (pgq_csr *r : c->retired)
			if (r && r != c) destroy_csr(r);
	}
	dev_free(c->off);
	dev_free(c->adj);
	dev_free(c->edge_ids);
	dev_free(c->w);
	dev_free(c->roff);
	dev_free(c->radj);
	dev_free(c->pull_hubs);
	dev_free(c->pull_hub_vertices);
	dev_free(c->pull_parts);
	dev_free(c->rown);
	dev_free(c->rpk);
	dev_free(c->padj);
	dev_free(c->rpadj);
	dev_free(c->fseg);
	dev_free(c->rseg);
	dev_free(c->rhead);
	dev_free(c->fwork);
	dev_free(c->rwork);
	dev_free(c->fdesc);
	dev_free(c->rdesc);
	dev_free(c->pagerank);
	dev_free(c->wcc);
	dev_free(c->rw);
	dev_free(c->wadj);
	dev_free(c->wsorted);
	dev_free(c->rwadj);
	dev_free(c->rwsorted);
	delete c;
}

static int staged_upload(void *d_dst, const void *h_src, size_t n_elems, size_t elem_bytes, int mode, int64_t V, std::atomic<int> *bad);
int ensure_edge_ids(pgq_csr *c) {
	if (!c || !c->lazy_edge_ids) return PGQ_OK; // (written once under the lock below, read racily here: a stale non-null only costs the lock)
	std::lock_guard<std::mutex> g(c->edge_ids_lock);
	if (!c->lazy_edge_ids) return PGQ_OK;
	int64_t *d = nullptr;
	PGQ_TRY(dev_alloc((void **)&d, (size_t)c->E * sizeof(int64_t)));
	std::atomic<int> oob { 0 };
	const int rc = staged_upload(d, c->lazy_edge_ids, (size_t)c->E, 8, 0, c->V, &oob);
	if (rc != PGQ_OK) {
		dev_free(d);
		return rc;
	}
	c->edge_ids = d;
	c->bytes += c->E * 8;
	c->lazy_edge_ids = nullptr;
	return PGQ_OK;
}

static int upload_impl(int64_t V, const int64_t *offsets, const int64_t *adj, const int64_t *edge_ids, const void *w,
                       int w_type, bool on_device, pgq_csr_t **out, bool lazy_ids = false) {
	PGQ_TRY(ensure_init());
	if (!out) return fail(PGQ_ERR_INVALID_ARG, "out handle pointer is NULL");
	*out = nullptr;
	if (V < 0) return fail(PGQ_ERR_INVALID_ARG, "V must not be negative");
	if (V >= (1LL << 31) - 1) return fail(PGQ_ERR_UNSUPPORTED, "V >= 2^31 - 1: the device CSR holds int32 vertex ids (include/pgq_hip.h, scale limits)");
	if (!offsets) return fail(PGQ_ERR_INVALID_ARG, "offsets is NULL");
	if (w_type < 0 || w_type > 2 || (w_type != 0 && !w)) return fail(PGQ_ERR_INVALID_ARG, "bad weight type / NULL weights");
	const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
	int64_t E = 0;
	if (on_device) PGQ_HIP_TRY(hipMemcpy(&E, offsets + V, sizeof(int64_t), hipMemcpyDeviceToHost));
	else E = offsets[V]; // E_used = v[V], not e.size()
	if (E < 0) return fail(PGQ_ERR_INVALID_ARG, "offsets[V] is negative");
	if (E >= (1LL << 31)) return fail(PGQ_ERR_UNSUPPORTED, "E >= 2^31: the device CSR addresses its adjacency with 32-bit positions (include/pgq_hip.h, scale limits)");
	if (E > 0 && !adj) return fail(PGQ_ERR_INVALID_ARG, "adj is NULL");
	pgq_csr *c = new pgq_csr();
	(void)hipGetDevice(&c->device);
	c->V = V;
	c->E = E;
	c->w_type = w_type;
	hipStream_t st = nullptr;
	int rc = PGQ_OK;
	int64_t *d_adj64 = nullptr;
	auto body = [&]() -> int {
		PGQ_TRY(g_streams.get(&st));
		PGQ_TRY(dev_alloc((void **)&c->off, (size_t)(V + 1) * sizeof(int64_t)));
		if (on_device) {
			PGQ_HIP_TRY(hipMemcpyAsync(c->off, offsets, (size_t)(V + 1) * sizeof(int64_t), kind, st));
		} else { // pageable: through the pinned blocks like the other arrays (a plain hipMemcpy of 3.6 MB of pageable memory took ~2 ms)
			std::atomic<int> unused { 0 };
			PGQ_TRY(staged_upload(c->off, offsets, (size_t)(V + 1), 8, 0, V, &unused));
		}
		if (E > 0 && on_device) {
			d_adj64 = const_cast<int64_t *>(adj);
			if (edge_ids) {
				PGQ_TRY(dev_alloc((void **)&c->edge_ids, (size_t)E * sizeof(int64_t)));
				PGQ_HIP_TRY(hipMemcpyAsync(c->edge_ids, edge_ids, (size_t)E * sizeof(int64_t), kind, st));
			}
			if (w_type != PGQ_W_NONE) {
				PGQ_TRY(dev_alloc((void **)&c->w, (size_t)E * 8));
				PGQ_HIP_TRY(hipMemcpyAsync(c->w, w, (size_t)E * 8, kind, st));
			}
		} else if (E > 0) {
			// pageable host arrays: staged through pinned rings by several threads, adjacency narrowed on the way
			std::atomic<int> oob { 0 };
			UploadTrace tr;
			if (options().upload_narrow_host) { // half the PCIe bytes, but the staging threads do the narrowing
				PGQ_TRY(dev_alloc((void **)&c->adj, (size_t)(E + 4) * sizeof(int32_t))); // +4: k_meet3 reads aligned 16-byte groups
				PGQ_TRY(staged_upload(c->adj, adj, (size_t)E, 8, 1, V, &oob));
				if (oob.load()) return fail(PGQ_ERR_INVALID_ARG, "CSR is malformed: adjacency out of [0,V)");
			} else { // raw int64 over PCIe, narrowed and range-checked by k_narrow_adj
				PGQ_TRY(dev_alloc((void **)&d_adj64, (size_t)E * sizeof(int64_t)));
				PGQ_TRY(staged_upload(d_adj64, adj, (size_t)E, 8, 0, V, &oob));
			}
			tr.mark("adjacency staged");
			if (edge_ids && lazy_ids) {
				c->lazy_edge_ids = edge_ids; // copied by ensure_edge_ids when a call reads edge ids
			} else if (edge_ids) {
				PGQ_TRY(dev_alloc((void **)&c->edge_ids, (size_t)E * sizeof(int64_t)));
				PGQ_TRY(staged_upload(c->edge_ids, edge_ids, (size_t)E, 8, 0, V, &oob));
				tr.mark("edge ids staged");
			}
			if (w_type != PGQ_W_NONE) {
				PGQ_TRY(dev_alloc((void **)&c->w, (size_t)E * 8));
				PGQ_TRY(staged_upload(c->w, w, (size_t)E, 8, 0, V, &oob));
			}
		}
		return finish_upload(c, d_adj64, st);
	};
	rc = body();
	if (st) {
		(void)hipStreamSynchronize(st);
		g_streams.put(st);
	}
	if (!on_device && d_adj64) dev_free(d_adj64);
	if (rc != PGQ_OK) {
		destroy_csr(c);
		return rc;
	}
	*out = c;
	return PGQ_OK;
}

// 16-byte copy kernel for the measured HBM ceiling.  One request per thread per round in a grid of one 1024-thread
// workgroup per CU (4 wavefronts per SIMD) was the fastest of the 36 shapes tools/membench tries on these boxes
// (5.6 TB/s read + written; wider grids and deeper unrolling lose 10-25 % to DRAM page conflicts).
__global__ __launch_bounds__(1024) void k_copy16(const uint4 *__restrict__ in, uint4 *__restrict__ out, int64_t n) {
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

} // namespace pgq

using namespace pgq;

extern "C" {

int pgq_init(int device) {
	if (g_inited.load()) return ensure_init();
	return do_init(device);
}

int pgq_device_count(void) {
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess) return 0;
	return count;
}

const char *pgq_last_error(void) { return t_err.c_str(); }
const char *pgq_version(void) { return "pgq_hip 0.6 (gfx950; pgq_stats_t: 65 words)"; }

int pgq_csr_upload_ex(int64_t V, const int64_t *offsets, const int64_t *adj, const int64_t *edge_ids, const void *w, int w_type,
                      unsigned flags, pgq_csr_t **out) {
	if (flags & ~PGQ_UPLOAD_LAZY_EDGE_IDS) return fail(PGQ_ERR_INVALID_ARG, "pgq_csr_upload_ex: unknown flag");
	return upload_impl(V, offsets, adj, edge_ids, w, w_type, false, out, (flags & PGQ_UPLOAD_LAZY_EDGE_IDS) != 0);
}
int pgq_csr_upload(int64_t V, const int64_t *offsets, const int64_t *adj, const int64_t *edge_ids, const void *w,
                   int w_type, pgq_csr_t **out) {
	return upload_impl(V, offsets, adj, edge_ids, w, w_type, false, out);
}
int pgq_csr_upload_device(int64_t V, const int64_t *d_offsets, const int64_t *d_adj, const int64_t *d_edge_ids,
                          const void *d_w, int w_type, pgq_csr_t **out) {
	return upload_impl(V, d_offsets, d_adj, d_edge_ids, d_w, w_type, true, out);
}

int pgq_csr_build_device(int64_t V, int64_t n_rows, const int64_t *d_src, const int64_t *d_dst,
                         const int64_t *d_edge_id, const void *d_w, int w_type, pgq_csr_t **out) {
	PGQ_TRY(ensure_init());
	if (!out) return fail(PGQ_ERR_INVALID_ARG, "out handle pointer is NULL");
	*out = nullptr;
	if (V < 0) return fail(PGQ_ERR_INVALID_ARG, "V must not be negative");
	if (V >= (1LL << 31) - 1) return fail(PGQ_ERR_UNSUPPORTED, "V >= 2^31 - 1: the device CSR holds int32 vertex ids (include/pgq_hip.h, scale limits)");
	if (n_rows < 0) return fail(PGQ_ERR_INVALID_ARG, "negative row count");
	if (n_rows >= (1LL << 31)) return fail(PGQ_ERR_UNSUPPORTED, "2^31 or more edge rows: the device CSR addresses its adjacency with 32-bit positions (include/pgq_hip.h, scale limits)");
	if (n_rows > 0 && (!d_src || !d_dst)) return fail(PGQ_ERR_INVALID_ARG, "NULL edge columns");
	if (w_type < 0 || w_type > 2 || (w_type != 0 && !d_w)) return fail(PGQ_ERR_INVALID_ARG, "bad weight type / NULL weights");
	const int64_t E = n_rows;
	pgq_csr *c = new pgq_csr();
	(void)hipGetDevice(&c->device);
	c->V = V;
	c->E = E;
	c->w_type = w_type;
	hipStream_t st = nullptr;
	u32 *d_key = nullptr, *d_idx = nullptr, *d_skey = nullptr, *d_order = nullptr;
	int *d_bad = nullptr;
	void *d_tmp = nullptr;
	auto body = [&]() -> int {
		PGQ_TRY(g_streams.get(&st));
		const size_t En = (size_t)std::max<int64_t>(E, 1);
		PGQ_TRY(dev_alloc_as(&d_key, En));
		PGQ_TRY(dev_alloc_as(&d_idx, En));
		PGQ_TRY(dev_alloc_as(&d_skey, En));
		PGQ_TRY(dev_alloc_as(&d_order, En));
		PGQ_TRY(dev_alloc_as(&d_bad, 1));
		PGQ_TRY(dev_alloc_as(&c->off, (size_t)V + 1));
		PGQ_TRY(dev_alloc_as(&c->adj, En + 4));
		PGQ_TRY(dev_alloc_as(&c->edge_ids, En));
		if (w_type != PGQ_W_NONE) PGQ_TRY(dev_alloc(&c->w, En * 8));
		PGQ_HIP_TRY(hipMemsetAsync(d_bad, 0, 4, st));
		if (E > 0) {
			hipLaunchKernelGGL(k_check_rows, dim3(grid_for(E)), dim3(256), 0, st, d_src, d_dst, E, V, d_key, d_idx, d_bad);
			int bad = 0;
			PGQ_HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
			PGQ_HIP_TRY(hipStreamSynchronize(st));
			if (bad) return fail(PGQ_ERR_INVALID_ARG, "edge endpoint out of range [0,V)");
			size_t sb = 0;
			int end_bit = 1;
			while ((1LL << end_bit) < V) end_bit++;
			PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, d_key, d_skey, d_idx, d_order, (int)E, 0, end_bit, st));
			PGQ_TRY(dev_alloc(&d_tmp, sb + 16));
			// stable LSD radix sort by source == arrival order per vertex of the single-threaded reference
			// (pos = ++v[src+1], csr_creation.cpp:132-138); the offsets (CsrInitializeEdge's prefix sum,
			// csr_creation.cpp:57-59) are the row boundaries of the sorted keys
			PGQ_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, sb, d_key, d_skey, d_idx, d_order, (int)E, 0, end_bit, st));
			hipLaunchKernelGGL(k_row_bounds, dim3((unsigned)((E + 1 + 255) / 256)), dim3(256), 0, st, d_skey, E, V, c->off);
			hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(E)), dim3(256), 0, st, d_order, E, d_dst, d_edge_id,
			                   (const int64_t *)d_w, c->adj, c->edge_ids, (int64_t *)c->w);
		} else {
			PGQ_HIP_TRY(hipMemsetAsync(c->off, 0, (size_t)(V + 1) * sizeof(int64_t), st));
		}
		// the temporaries go back to the block cache before finish_upload asks for its own (same sizes)
		PGQ_HIP_TRY(hipStreamSynchronize(st));
		for (void **p : { (void **)&d_key, (void **)&d_idx, (void **)&d_skey, (void **)&d_order, &d_tmp }) {
			dev_free(*p);
			*p = nullptr;
		}
		return finish_upload(c, nullptr, st);
	};
	int rc = body();
	if (st) {
		(void)hipStreamSynchronize(st);
		g_streams.put(st);
	}
	for (void *p : { (void *)d_key, (void *)d_idx, (void *)d_skey, (void *)d_order, (void *)d_bad, d_tmp }) dev_free(p);
	if (rc != PGQ_OK) {
		destroy_csr(c);
		return rc;
	}
	*out = c;
	return PGQ_OK;
}

int pgq_csr_download(const pgq_csr_t *c, int64_t *offsets, int64_t *adj, int64_t *edge_ids, void *w) {
	PGQ_TRY(ensure_init());
	if (!c) return fail(PGQ_ERR_INVALID_ARG, "NULL csr");
	hipStream_t st = nullptr; // the null stream: downloads are rare (tests, CSR spill) and may serialise
	if (offsets) PGQ_TRY(staged_download(offsets, c->off, (size_t)(c->V + 1) * 8, st));
	if (adj && c->E > 0) {
		std::vector<int32_t> a32((size_t)c->E);
		PGQ_TRY(staged_download(a32.data(), c->adj, (size_t)c->E * 4, st));
		for (int64_t i = 0; i < c->E; i++) adj[i] = a32[i];
	}
	if (edge_ids && c->E > 0) {
		PGQ_TRY(ensure_edge_ids(const_cast<pgq_csr_t *>(c)));
		if (c->edge_ids) PGQ_TRY(staged_download(edge_ids, c->edge_ids, (size_t)c->E * 8, st));
		else
			for (int64_t i = 0; i < c->E; i++) edge_ids[i] = i;
	}
	if (w && c->w && c->E > 0) PGQ_TRY(staged_download(w, c->w, (size_t)c->E * 8, st));
	return PGQ_OK;
}

int pgq_csr_free(pgq_csr_t *csr) {
	if (!csr) return PGQ_OK;
	PGQ_TRY(ensure_init());
	destroy_csr(csr);
	return PGQ_OK;
}
int pgq_init_devices(const int *devices, int n) {
	if (!devices || n < 1) return fail(PGQ_ERR_INVALID_ARG, "pgq_init_devices: empty device list");
	PGQ_TRY(pgq_init(devices[0]));
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PGQ_ERR_NO_DEVICE, "no HIP device");
	std::lock_guard<std::mutex> g(g_init_lock);
	std::vector<int> list;
	for (int k = 0; k < n; k++) {
		if (devices[k] < 0 || devices[k] >= count) return fail(PGQ_ERR_INVALID_ARG, "pgq_init_devices: device index out of range");
		list.push_back(devices[k]);
	}
	if (list[0] != g_device) return fail(PGQ_ERR_INVALID_ARG, "pgq_init_devices: the first device must be the one pgq_init bound");
	for (int a : list)
		for (int b : list)
			if (a != b) {
				int can = 0;
				(void)hipSetDevice(a);
				if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(b, 0); // xGMI peer copies
			}
	(void)hipSetDevice(g_device);
	g_devices = list;
	return PGQ_OK;
}
int pgq_init_mask(uint64_t device_mask) {
	std::vector<int> list;
	for (int d = 0; d < 64; d++)
		if ((device_mask >> d) & 1ull) list.push_back(d);
	return pgq_init_devices(list.data(), (int)list.size());
}
int pgq_num_enabled_devices(void) { return (int)g_devices.size(); }

// byte-for-byte copy of the device CSR (base arrays and everything derived at upload) to another device over xGMI
static int clone_csr(const pgq_csr *c, int dev, pgq_csr **out) {
	pgq_csr *r = new pgq_csr();
	r->device = dev;
	r->V = c->V;
	r->E = c->E;
	r->w_type = c->w_type;
	r->n_pull_hub_items = c->n_pull_hub_items;
	r->n_pull_hub_vertices = c->n_pull_hub_vertices;
	r->n_pull_parts = c->n_pull_parts;
	r->hub_threshold = c->hub_threshold;
	r->max_out_degree = c->max_out_degree;
	r->max_in_degree = c->max_in_degree;
	r->two_hop_mean = c->two_hop_mean;
	r->bytes = c->bytes;
	r->has_negative_weight = c->has_negative_weight;
	r->is_replica = true;
	*out = r;
	const size_t V1 = (size_t)c->V + 1, En = (size_t)std::max<int64_t>(c->E, 1);
	auto copy = [&](void **dst, const void *src, size_t bytes) -> int {
		*dst = nullptr;
		if (!src || bytes == 0) return PGQ_OK;
		PGQ_HIP_TRY(hipSetDevice(dev));
		PGQ_TRY(dev_alloc(dst, bytes));
		PGQ_HIP_TRY(hipMemcpyPeer(*dst, dev, src, c->device, bytes));
		return PGQ_OK;
	};
	PGQ_TRY(copy((void **)&r->off, c->off, V1 * 8));
	PGQ_TRY(copy((void **)&r->adj, c->adj, (En + 4) * 4));
	PGQ_TRY(copy((void **)&r->edge_ids, c->edge_ids, En * 8));
	PGQ_TRY(copy((void **)&r->w, c->w, En * 8));
	PGQ_TRY(copy((void **)&r->roff, c->roff, V1 * 8));
	PGQ_TRY(copy((void **)&r->radj, c->radj, (En + 4) * 4));
	PGQ_TRY(copy((void **)&r->pull_hubs, c->pull_hubs, (size_t)c->n_pull_hub_items * sizeof(HubItem)));
	PGQ_TRY(copy((void **)&r->pull_hub_vertices, c->pull_hub_vertices, (size_t)c->n_pull_hub_vertices * 4));
	PGQ_TRY(copy((void **)&r->pull_parts, c->pull_parts, (size_t)c->n_pull_parts * 2 * 4));
	PGQ_TRY(copy((void **)&r->rown, c->rown, En + 8));
	PGQ_TRY(copy((void **)&r->rpk, c->rpk, (En + 2048) * 4));
	r->padj_groups = c->padj_groups;
	r->rpadj_groups = c->rpadj_groups;
	PGQ_TRY(copy((void **)&r->padj, c->padj, (size_t)c->padj_groups * 16 + 16));
	PGQ_TRY(copy((void **)&r->rpadj, c->rpadj, (size_t)c->rpadj_groups * 16 + 16));
	PGQ_TRY(copy((void **)&r->fseg, c->fseg, (size_t)c->V * 8));
	PGQ_TRY(copy((void **)&r->rseg, c->rseg, (size_t)c->V * 8));
	PGQ_TRY(copy((void **)&r->rhead, c->rhead, (size_t)c->V * 256));
	PGQ_TRY(copy((void **)&r->fwork, c->fwork, (size_t)c->V * 4));
	PGQ_TRY(copy((void **)&r->rwork, c->rwork, (size_t)c->V * 4));
	PGQ_TRY(copy((void **)&r->fdesc, c->fdesc, (En + 1) * 16));
	PGQ_TRY(copy((void **)&r->rdesc, c->rdesc, (En + 1) * 16));
	PGQ_HIP_TRY(hipDeviceSynchronize());
	return PGQ_OK;
}

int pgq_csr_replicate(pgq_csr_t *c) {
	PGQ_TRY(ensure_edge_ids(c)); // replicas carry everything: a lazily uploaded array is copied now
	PGQ_TRY(ensure_init());
	if (!c || c->is_replica) return fail(PGQ_ERR_INVALID_ARG, "pgq_csr_replicate: NULL or replica handle");
	const std::vector<int> devs = enabled_devices();
	// one caller builds; a list built for another device set (a *_multi call before pgq_init_devices leaves a one-entry
	// list) is rebuilt, the old replicas stay alive until the CSR is freed because calls in flight may still read them
	std::lock_guard<std::mutex> g(c->replica_lock);
	if (!c->replicas.empty() && c->replica_devices == devs) return PGQ_OK;
	std::vector<pgq_csr *> reps(devs.size(), nullptr);
	int rc = PGQ_OK;
	bool self_used = false;
	for (size_t k = 0; k < devs.size() && rc == PGQ_OK; k++) {
		if (devs[k] == c->device && !self_used) {
			reps[k] = c;
			self_used = true;
			continue;
		}
		// reuse a replica of the previous list that sits on the right device
		for (pgq_csr *&old : c->replicas)
			if (old && old != c && old->device == devs[k]) {
				reps[k] = old;
				old = nullptr;
				break;
			}
		if (!reps[k]) rc = clone_csr(c, devs[k], &reps[k]);
	}
	(void)hipSetDevice(current_device());
	if (rc != PGQ_OK) {
		for (pgq_csr *r : reps)
			if (r && r != c) c->retired.push_back(r); // freed with the CSR (a reused one may still be in use)
		for (pgq_csr *old : c->replicas)
			if (old && old != c) c->retired.push_back(old);
		c->replicas.clear();
		c->replica_devices.clear();
		return rc;
	}
	for (pgq_csr *old : c->replicas)
		if (old && old != c) c->retired.push_back(old);
	c->replicas = reps;
	c->replica_devices = devs;
	return PGQ_OK;
}

int64_t pgq_csr_num_vertices(const pgq_csr_t *csr) { return csr ? csr->V : -1; }
int64_t pgq_csr_num_edges(const pgq_csr_t *csr) { return csr ? csr->E : -1; }
int pgq_csr_w_type(const pgq_csr_t *csr) { return csr ? csr->w_type : -1; }
int64_t pgq_csr_device_bytes(const pgq_csr_t *csr) { return csr ? csr->bytes : -1; }
int pgq_csr_has_prepass_layout(const pgq_csr_t *csr) { return csr && csr->fdesc != nullptr ? 1 : 0; }

} // extern "C"


extern "C" {

static int set_option_in(Options &o, const char *key, const char *value) {
	if (!key || !value) return fail(PGQ_ERR_INVALID_ARG, "NULL option");
	for (const OptRef &r : option_table(o)) {
		if (strcmp(r.name, key) != 0) continue;
		if (r.i) *r.i = atoi(value);
		else *r.d = atof(value);
		return PGQ_OK;
	}
	return fail(PGQ_ERR_INVALID_ARG, std::string("unknown option: ") + key);
}
static int get_option_in(Options &o, const char *key, double *value) {
	if (!key || !value) return fail(PGQ_ERR_INVALID_ARG, "NULL option");
	for (const OptRef &r : option_table(o)) {
		if (strcmp(r.name, key) != 0) continue;
		*value = r.i ? (double)*r.i : *r.d;
		return PGQ_OK;
	}
	return fail(PGQ_ERR_INVALID_ARG, std::string("unknown option: ") + key);
}

int pgq_set_option(const char *key, const char *value) { return set_option_in(g_opt, key, value); }
int pgq_get_option(const char *key, double *value) { return get_option_in(g_opt, key, value); }
// the value a key ships with (a default-constructed option set: what a process that sets nothing runs under)
int pgq_get_default_option(const char *key, double *value) {
	Options shipped;
	return get_option_in(shipped, key, value);
}

// Options of ONE handle: the first call copies the process-wide set, later searches on this handle use the copy (two
// DuckDB connections, or a test, can tune their own CSR without touching each other's).  Upload-time options
// (meet_align, hub_chunk, ...) were consumed when the handle was built and are not affected.
int pgq_csr_set_option(pgq_csr_t *csr, const char *key, const char *value) {
	if (!csr || csr->is_replica) return fail(PGQ_ERR_INVALID_ARG, "pgq_csr_set_option: NULL or replica handle");
	std::lock_guard<std::mutex> g(csr->replica_lock);
	if (!csr->opt) csr->opt.reset(new Options(g_opt));
	return set_option_in(*csr->opt, key, value);
}
int pgq_csr_get_option(pgq_csr_t *csr, const char *key, double *value) {
	if (!csr) return fail(PGQ_ERR_INVALID_ARG, "pgq_csr_get_option: NULL handle");
	return get_option_in(csr->opt ? *csr->opt : g_opt, key, value);
}

const char *pgq_kclass_name(int k) {
	static const char *names[K_COUNT] = { "prep",   "push",   "pull",  "pull_hub",
		                                  "queue",  "detect", "recon", "relax", "pull_sparse", "meet", "meet4", "bibfs", "ball" };
	return (k >= 0 && k < K_COUNT) ? names[k] : nullptr;
}
int pgq_get_stats_sized(pgq_stats_t *out, size_t struct_size) {
	if (!out) return fail(PGQ_ERR_INVALID_ARG, "NULL stats");
	memcpy(out, &tstats().s, std::min(struct_size, sizeof(pgq_stats_t)));
	return PGQ_OK;
}
int pgq_get_stats(pgq_stats_t *out) { return pgq_get_stats_sized(out, sizeof(pgq_stats_t)); }
int pgq_reset_stats(void) {
	memset(&tstats().s, 0, sizeof(pgq_stats_t));
	return PGQ_OK;
}

int pgq_measure_copy_bandwidth(int64_t bytes, int iters, double *out_gbps) {
	PGQ_TRY(ensure_init());
	if (bytes < 1024 || iters < 1 || !out_gbps) return fail(PGQ_ERR_INVALID_ARG, "bad bandwidth probe arguments");
	void *a = nullptr, *b = nullptr;
	PGQ_TRY(dev_alloc((void **)&a, (size_t)bytes));
	PGQ_TRY(dev_alloc((void **)&b, (size_t)bytes));
	PGQ_HIP_TRY(hipMemset(a, 1, (size_t)bytes));
	int64_t n = bytes / 16;
	hipEvent_t e0, e1;
	PGQ_HIP_TRY(hipEventCreate(&e0));
	PGQ_HIP_TRY(hipEventCreate(&e1));
	hipLaunchKernelGGL(k_copy16, dim3(256), dim3(1024), 0, 0, (const uint4 *)a, (uint4 *)b, n);
	PGQ_HIP_TRY(hipEventRecord(e0, 0));
	for (int i = 0; i < iters; i++)
		hipLaunchKernelGGL(k_copy16, dim3(256), dim3(1024), 0, 0, (const uint4 *)a, (uint4 *)b, n);
	PGQ_HIP_TRY(hipEventRecord(e1, 0));
	PGQ_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0.f;
	PGQ_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	*out_gbps = 2.0 * (double)(n * 16) * iters / (ms * 1e-3) / 1e9;
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	(void)hipFree(a);
	(void)hipFree(b);
	return PGQ_OK;
}

} // extern "C"
