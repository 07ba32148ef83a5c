// pgq_udf.cpp — host mirror of DuckPGQ's scalar-function layer above the device C ABI (include/pgq_udf.h).
//
// Mirrors, without DuckDB types:
//   * the per-connection state  DuckPGQState {csr_list, csr_lock, csr_to_delete}  src/include/duckpgq_state.hpp:36-38,
//     QueryEnd src/duckpgq_state.cpp:162-170, GetCSR :180-186
//   * the host CSR object `class CSR`  src/include/duckpgq/core/utils/compressed_sparse_row.hpp:25-47 and its
//     construction UDFs  src/core/functions/scalar/csr_creation.cpp:14-198 (these stay on the CPU: the joins that feed
//     them run inside DuckDB; the GPU takes over at the first search call)
//   * the search UDF wrappers: argument checks and exception texts of iterativelength.cpp:35-51,
//     shortest_path.cpp:48-58, cheapest_path_length_function_data.cpp:18-23, then one call into libpgq_hip.
// There is no CPU search path here: a search without a working device fails with the device library's error.
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "pgq_udf.h"

namespace {

thread_local std::string t_err;
int fail(const std::string &m) {
	t_err = m;
	return -1;
}
int device_fail() { return fail(std::string("Internal Error: ") + pgq_last_error()); }

struct View {
	const int64_t *data;
	const uint32_t *sel;
	const uint64_t *valid;
	explicit View(const pgq_vec_t &v) : data((const int64_t *)v.data), sel(v.sel), valid(v.validity) {}
	int64_t pos(int64_t r) const { return sel ? (int64_t)sel[r] : r; }
	bool ok(int64_t p) const { return !valid || ((valid[p >> 6] >> (p & 63)) & 1ULL); }
};

void fill_valid(uint64_t *m, int64_t n) {
	for (int64_t i = 0; i < (n + 63) / 64; i++) m[i] = ~0ULL;
}
void set_invalid(uint64_t *m, int64_t r) { m[r >> 6] &= ~(1ULL << (r & 63)); }

// class CSR (compressed_sparse_row.hpp:25-47) + the device mirror this project adds
struct HostCSR {
	std::unique_ptr<std::atomic<int64_t>[]> v;
	size_t vsize = 0;
	std::vector<int64_t> e, edge_ids, w;
	std::vector<double> w_double;
	// read outside csr_lock by the double-checked initialisation the reference uses (csr_creation.cpp:15,44,64)
	std::atomic<bool> initialized_v { false }, initialized_e { false }, initialized_w { false };
	std::atomic<pgq_csr_t *> device { nullptr };
	~HostCSR() {
		if (device.load()) pgq_csr_free(device.load());
	}
};

} // namespace

struct pgq_state {
	std::map<int32_t, std::shared_ptr<HostCSR>> csr_list;
	std::mutex csr_lock;
	std::set<int32_t> csr_to_delete;
};

namespace {

// A UDF call keeps its own reference for as long as it runs: delete_csr / QueryEnd / a replacing create_csr_vertex on
// another thread only drop the registry's reference, so the host arrays and the device CSR (kernels may still be
// reading it) outlive the call.
typedef std::shared_ptr<HostCSR> CsrRef;
CsrRef find_csr(pgq_state *s, int32_t id) {
	std::lock_guard<std::mutex> g(s->csr_lock);
	auto it = s->csr_list.find(id);
	return it == s->csr_list.end() ? nullptr : it->second;
}

// lazy, once-per-CSR upload under csr_lock (many DuckDB threads hit the first search chunk together)
pgq_csr_t *device_csr(pgq_state *s, const CsrRef &c, int64_t V) {
	std::lock_guard<std::mutex> g(s->csr_lock);
	if (c->device) return c->device;
	std::vector<int64_t> off((size_t)V + 1);
	for (int64_t i = 0; i <= V; i++) off[i] = c->v[i].load(std::memory_order_relaxed);
	const void *w = nullptr;
	int wt = PGQ_W_NONE;
	if (c->initialized_w) {
		if (!c->w.empty()) {
			w = c->w.data();
			wt = PGQ_W_INT64;
		} else if (!c->w_double.empty()) {
			w = c->w_double.data();
			wt = PGQ_W_DOUBLE;
		}
	}
	// an edgeless CSR never ran CsrInitializeEdge: its offsets are still the raw degrees (all zero)
	static const int64_t dummy = 0;
	const int64_t *adj = c->e.empty() ? &dummy : c->e.data();
	const int64_t *eids = c->edge_ids.empty() ? nullptr : c->edge_ids.data();
	if (!c->initialized_e) {
		for (auto &x : off) x = 0;
	}
	pgq_csr_t *h = nullptr;
	// the host CSR owns edge_ids until it is deleted, and the device handle dies first (~HostCSR frees it in its body): the ids cross PCIe only if
	// a shortestpath call asks for them (PGQ_UPLOAD_LAZY_EDGE_IDS)
	if (pgq_csr_upload_ex(V, off.data(), adj, eids, w, wt, PGQ_UPLOAD_LAZY_EDGE_IDS, &h) != PGQ_OK) return nullptr;
	c->device = h;
	return h;
}

int search_prologue(pgq_state *s, int32_t id, int64_t V, CsrRef *out, pgq_csr_t **dev, const char *what) {
	if (!s) return fail("Invalid Input Error: NULL state");
	CsrRef c = find_csr(s, id);
	if (!c || !c->initialized_v) // iterativelength.cpp:44-51, shortest_path.cpp:49-58
		return fail(std::string("Constraint Error: Need to initialize CSR before doing ") + what);
	if ((int64_t)c->vsize != V + 2) return fail("Invalid Input Error: vertex count does not match the CSR");
	pgq_csr_t *d = device_csr(s, c, V);
	if (!d) return device_fail();
	*out = c;
	*dev = d;
	return 0;
}

} // namespace

extern "C" {

const char *pgq_udf_last_error(void) { return t_err.c_str(); }
pgq_state_t *pgq_state_new(void) { return new pgq_state(); }
void pgq_state_free(pgq_state_t *s) { delete s; }

int pgq_state_query_end(pgq_state_t *s) { // duckpgq_state.cpp:162-170
	if (!s) return fail("Invalid Input Error: NULL state");
	std::lock_guard<std::mutex> g(s->csr_lock);
	for (int32_t id : s->csr_to_delete) s->csr_list.erase(id);
	s->csr_to_delete.clear();
	return 0;
}

int pgq_udf_create_csr_vertex(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t dense_id, pgq_vec_t cnt,
                              int64_t *out, uint64_t *out_valid) {
	if (!s || V < 0) return fail("Invalid Input Error: bad arguments");
	HostCSR *c;
	{ // CsrInitializeVertex (csr_creation.cpp:14-41)
		std::lock_guard<std::mutex> g(s->csr_lock);
		auto it = s->csr_list.find(id);
		if (it == s->csr_list.end() || !it->second->initialized_v) {
			auto csr = std::make_unique<HostCSR>();
			csr->v.reset(new (std::nothrow) std::atomic<int64_t>[(size_t)V + 2]);
			if (!csr->v) return fail("INTERNAL Error: Unable to initialize vector of size for csr vertex table representation");
			csr->vsize = (size_t)V + 2;
			for (size_t i = 0; i < csr->vsize; i++) csr->v[i] = 0;
			csr->initialized_v = true;
			s->csr_list[id] = std::move(csr);
		}
		c = s->csr_list[id].get();
	}
	View did(dense_id), cv(cnt);
	fill_valid(out_valid, n);
	for (int64_t r = 0; r < n; r++) { // BinaryExecutor: NULL in -> NULL out, lambda not run (:103-109)
		int64_t dp = did.pos(r), cp = cv.pos(r);
		if (!did.ok(dp) || !cv.ok(cp)) {
			set_invalid(out_valid, r);
			continue;
		}
		int64_t d = did.data[dp];
		if (d < 0 || d >= V) return fail("Invalid Input Error: dense_id out of range [0,V)");
		c->v[(size_t)d + 2] = cv.data[cp];
		out[r] = cv.data[cp];
	}
	return 0;
}

int pgq_udf_create_csr_edge(pgq_state_t *s, int32_t id, int64_t V, int64_t e_sum, int64_t e_count, int64_t n,
                            pgq_vec_t srcv, pgq_vec_t dstv, pgq_vec_t eidv, const pgq_vec_t *wv, int w_type,
                            int32_t *out, uint64_t *out_valid) {
	if (!s) return fail("Invalid Input Error: NULL state");
	if (e_sum != e_count) { // csr_creation.cpp:121-125
		std::lock_guard<std::mutex> g(s->csr_lock);
		s->csr_to_delete.insert(id);
		return fail("Constraint Error: Non-existent/non-unique vertices detected. Make sure all vertices referred by edge "
		            "tables exist and are unique for path-finding queries.");
	}
	if (e_sum < 0) return fail("Invalid Input Error: negative edge count");
	CsrRef c = find_csr(s, id);
	if (!c || !c->initialized_v) return fail("Constraint Error: CSR vertices must be created before the edges");
	if ((int64_t)c->vsize != V + 2) return fail("Invalid Input Error: vertex count does not match the CSR");
	if (!c->initialized_e) { // CsrInitializeEdge :43-61
		std::lock_guard<std::mutex> g(s->csr_lock);
		if (!c->initialized_e) {
			try { // no exception may cross the C ABI
				c->e.assign((size_t)e_sum, 0);
				c->edge_ids.assign((size_t)e_sum, 0);
			} catch (const std::exception &) { // bad_alloc, length_error
				c->e.clear();
				c->edge_ids.clear();
				return fail("Out of Memory Error: cannot allocate the CSR edge arrays");
			}
			for (int64_t i = 1; i < V + 2; i++) c->v[i] += c->v[i - 1];
			c->initialized_e = true;
		}
	}
	if (c->device) return fail("Constraint Error: CSR is already in use by a search; cannot add edges");
	View src(srcv), dst(dstv), eid(eidv);
	fill_valid(out_valid, n);
	const int64_t E = (int64_t)c->e.size();
	auto claim = [&](int64_t sv) -> int64_t { return ++c->v[(size_t)sv + 1]; };
	if (!wv) { // :126-139
		for (int64_t r = 0; r < n; r++) {
			int64_t sp = src.pos(r), dp = dst.pos(r), ep = eid.pos(r);
			if (!src.ok(sp) || !dst.ok(dp) || !eid.ok(ep)) {
				set_invalid(out_valid, r);
				continue;
			}
			int64_t sv = src.data[sp];
			if (sv < 0 || sv >= V) return fail("Invalid Input Error: edge source out of range [0,V)");
			int64_t slot = claim(sv);
			if (slot < 1 || slot > E) return fail("Invalid Input Error: more edges than counted for a vertex");
			c->e[(size_t)slot - 1] = dst.data[dp];
			c->edge_ids[(size_t)slot - 1] = eid.data[ep];
			out[r] = 1;
		}
		return 0;
	}
	if (!c->initialized_w) { // CsrInitializeWeight :63-84
		std::lock_guard<std::mutex> g(s->csr_lock);
		if (!c->initialized_w) {
			try {
				if (w_type == PGQ_W_INT64) c->w.assign((size_t)e_sum, 0);
				else if (w_type == PGQ_W_DOUBLE) c->w_double.assign((size_t)e_sum, 0.0);
				else return fail("Not implemented Error: Unrecognized weight type detected.");
			} catch (const std::exception &) { // bad_alloc, length_error
				return fail("Out of Memory Error: cannot allocate the CSR weight array");
			}
			c->initialized_w = true;
		}
	}
	// every chunk of one CSR carries the weight type the first chunk initialised (the other vector is empty)
	if ((w_type == PGQ_W_INT64 ? c->w.size() : w_type == PGQ_W_DOUBLE ? c->w_double.size() : 0) != c->e.size())
		return fail("Invalid Input Error: weight type differs from the one this CSR was initialised with");
	View wgt(*wv);
	for (int64_t r = 0; r < n; r++) { // :156-197
		int64_t sp = src.pos(r), dp = dst.pos(r), ep = eid.pos(r), wp = wgt.pos(r);
		if (!src.ok(sp) || !dst.ok(dp) || !eid.ok(ep) || !wgt.ok(wp)) {
			set_invalid(out_valid, r);
			continue;
		}
		int64_t sv = src.data[sp];
		if (sv < 0 || sv >= V) return fail("Invalid Input Error: edge source out of range [0,V)");
		int64_t slot = claim(sv);
		if (slot < 1 || slot > E) return fail("Invalid Input Error: more edges than counted for a vertex");
		c->e[(size_t)slot - 1] = dst.data[dp];
		c->edge_ids[(size_t)slot - 1] = eid.data[ep];
		if (w_type == PGQ_W_INT64) {
			int64_t x = wgt.data[wp];
			c->w[(size_t)slot - 1] = x;
			out[r] = (int32_t)x;
		} else {
			double x = reinterpret_cast<const double *>(wgt.data)[wp];
			c->w_double[(size_t)slot - 1] = x;
			out[r] = (int32_t)x;
		}
	}
	return 0;
}

int pgq_udf_bind_search(pgq_state_t *s, int32_t id) { // iterative_length_function_data.cpp:18-30
	if (!s) return fail("Invalid Input Error: NULL state");
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id);
	return 0;
}

int pgq_udf_iterativelength(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                            int64_t *out, uint64_t *out_valid) {
	CsrRef c;
	pgq_csr_t *d;
	if (search_prologue(s, id, V, &c, &d, "shortest path")) return -1;
	if (pgq_iterativelength(d, V, n, src, dst, out, out_valid) != PGQ_OK) return device_fail();
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id); // iterativelength.cpp:142
	return 0;
}

int pgq_udf_iterativelength2(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                             int64_t *out, uint64_t *out_valid) {
	return pgq_udf_iterativelength(s, id, V, n, src, dst, out, out_valid);
}

int pgq_udf_iterativelengthbidirectional(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                                         int64_t *out, uint64_t *out_valid) {
	// the bidirectional schedule (forward CSR from src, transposed CSR from dst): its own device entry point
	CsrRef c;
	pgq_csr_t *d;
	if (search_prologue(s, id, V, &c, &d, "shortest path")) return -1;
	if (pgq_iterativelength_bidirectional(d, V, n, src, dst, out, out_valid) != PGQ_OK) return device_fail();
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id); // iterativelength_bidirectional.cpp:152
	return 0;
}

int pgq_udf_shortestpath(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                         uint64_t *out_offset, uint64_t *out_length, uint64_t *out_valid, const int64_t **out_child,
                         uint64_t *out_child_len) {
	CsrRef c;
	pgq_csr_t *d;
	if (search_prologue(s, id, V, &c, &d, "shortest path")) return -1;
	if (pgq_shortestpath(d, V, n, src, dst, out_offset, out_length, out_valid, out_child, out_child_len) != PGQ_OK)
		return device_fail();
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id); // shortest_path.cpp:206
	return 0;
}

int pgq_udf_bind_cheapest(pgq_state_t *s, int32_t id, int *ret_type) { // cheapest_path_length_function_data.cpp:7-32
	if (!s) return fail("Invalid Input Error: NULL state");
	CsrRef c = find_csr(s, id);
	if (!c) return fail("Constraint Error: CSR not found with ID " + std::to_string(id)); // duckpgq_state.cpp:183
	{
		std::lock_guard<std::mutex> g(s->csr_lock);
		s->csr_to_delete.insert(id);
	}
	if (!(c->initialized_v && c->initialized_e && c->initialized_w))
		return fail("Constraint Error: Need to initialize CSR before doing cheapest path");
	if (ret_type) *ret_type = c->w.empty() ? PGQ_W_DOUBLE : PGQ_W_INT64;
	return 0;
}

int pgq_udf_cheapest_path_length(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                                 void *out, uint64_t *out_valid) {
	if (!s) return fail("Invalid Input Error: NULL state");
	CsrRef c = find_csr(s, id);
	if (!c) return fail("Constraint Error: CSR not found with ID " + std::to_string(id));
	if (!(c->initialized_v && c->initialized_e && c->initialized_w))
		return fail("Constraint Error: Need to initialize CSR before doing cheapest path");
	if ((int64_t)c->vsize != V + 2) return fail("Invalid Input Error: vertex count does not match the CSR");
	pgq_csr_t *d = device_csr(s, c, V);
	if (!d) return device_fail();
	if (pgq_cheapest_path_length(d, V, n, src, dst, out, out_valid) != PGQ_OK) return device_fail();
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id); // cheapest_path_length.cpp:162
	return 0;
}

int pgq_udf_reachability(pgq_state_t *s, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, uint8_t *out,
                         uint64_t *out_valid) {
	std::vector<int64_t> len((size_t)n);
	if (pgq_udf_iterativelength(s, id, V, n, src, dst, len.data(), out_valid)) return -1;
	View sv(src), dv(dst);
	for (int64_t r = 0; r < n; r++) {
		const bool any_null = !sv.ok(sv.pos(r)) || !dv.ok(dv.pos(r)); // a NULL endpoint gives NULL
		out[r] = len[r] >= 0 ? 1 : 0;
		if (!any_null) out_valid[r >> 6] |= 1ULL << (r & 63); // unreachable is FALSE, not NULL
		else out_valid[r >> 6] &= ~(1ULL << (r & 63));
	}
	return 0;
}

namespace {
int analytics_prologue(pgq_state *s, int32_t id, const char *what, CsrRef *out) {
	if (!s) return fail("Invalid Input Error: NULL state");
	CsrRef c = find_csr(s, id);
	if (!c) return fail("Constraint Error: CSR not found. Is the graph populated?");
	if (!(c->initialized_v && c->initialized_e)) return fail(std::string("Constraint Error: Need to initialize CSR before ") + what);
	*out = c;
	return 0;
}
} // namespace

int pgq_udf_local_clustering_coefficient(pgq_state_t *s, int32_t id, int64_t n, pgq_vec_t src, float *out, uint64_t *out_valid) {
	CsrRef c;
	if (analytics_prologue(s, id, "doing local clustering coefficient.", &c)) return -1;
	const int64_t V = (int64_t)c->vsize - 2;
	pgq_csr_t *d = device_csr(s, c, V);
	if (!d) return device_fail();
	if (pgq_local_clustering_coefficient(d, V, n, src, out, out_valid) != PGQ_OK) return device_fail();
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id); // local_clustering_coefficient.cpp:71
	return 0;
}

int pgq_udf_pagerank(pgq_state_t *s, int32_t id, int64_t n, pgq_vec_t src, double *out, uint64_t *out_valid) {
	CsrRef c;
	if (analytics_prologue(s, id, "running PageRank.", &c)) return -1;
	const int64_t V = (int64_t)c->vsize - 2;
	pgq_csr_t *d = device_csr(s, c, V);
	if (!d) return device_fail();
	if (pgq_pagerank(d, V, n, src, out, out_valid) != PGQ_OK) return device_fail();
	std::lock_guard<std::mutex> g(s->csr_lock);
	s->csr_to_delete.insert(id); // pagerank.cpp:109
	return 0;
}

// weakly_connected_component.cpp:36-104 on the device (pgq_weakly_connected_component: the spanning forest under the
// reference's processing order by Boruvka rounds, then the reference's own Link over its <= V - 1 edges); the ids are
// computed once per CSR like the reference's bind-data forest (info.state_converged).
int pgq_udf_weakly_connected_component(pgq_state_t *s, int32_t id, int64_t n, pgq_vec_t srcv, int64_t *out, uint64_t *out_valid) {
	CsrRef c;
	if (analytics_prologue(s, id, "doing weakly connected components.", &c)) return -1;
	const int64_t V = (int64_t)c->vsize - 2;
	pgq_csr_t *d = device_csr(s, c, V);
	if (!d) return device_fail();
	if (pgq_weakly_connected_component(d, V, n, srcv, out, out_valid) != PGQ_OK) return device_fail();
	std::lock_guard<std::mutex> g2(s->csr_lock);
	s->csr_to_delete.insert(id); // :103
	return 0;
}

int pgq_udf_delete_csr(pgq_state_t *s, int32_t id, int *out_flag) { // csr_deletion.cpp:10-20
	if (!s) return fail("Invalid Input Error: NULL state");
	std::lock_guard<std::mutex> g(s->csr_lock);
	*out_flag = s->csr_list.erase(id) == 1 ? 1 : 0;
	return 0;
}

int pgq_udf_csr_get_w_type(pgq_state_t *s, int32_t id, int32_t *out) { // csr_get_w_type.cpp:16-36
	if (!s) return fail("Invalid Input Error: NULL state");
	CsrRef c = find_csr(s, id);
	if (!c) return fail("Constraint Error: CSR not found with ID " + std::to_string(id));
	if (!c->initialized_w) *out = PGQ_W_NONE;
	else if (!c->w.empty()) *out = PGQ_W_INT64;
	else if (!c->w_double.empty()) *out = PGQ_W_DOUBLE;
	else return fail("INTERNAL Error: Corrupted weight vector");
	return 0;
}

int64_t pgq_udf_scan_csr_v(pgq_state_t *s, int32_t id, int64_t *out, int64_t cap) { // pgq_scan.cpp:84-111
	CsrRef c = s ? find_csr(s, id) : nullptr;
	if (!c) return fail("Constraint Error: CSR not found with ID " + std::to_string(id));
	int64_t k = std::min<int64_t>(cap, (int64_t)c->vsize);
	for (int64_t i = 0; i < k; i++) out[i] = c->v[i].load();
	return (int64_t)c->vsize;
}
int64_t pgq_udf_scan_csr_e(pgq_state_t *s, int32_t id, int64_t *out, int64_t cap) { // pgq_scan.cpp:15-42
	CsrRef c = s ? find_csr(s, id) : nullptr;
	if (!c) return fail("Constraint Error: CSR not found with ID " + std::to_string(id));
	int64_t k = std::min<int64_t>(cap, (int64_t)c->e.size());
	if (k > 0) memcpy(out, c->e.data(), (size_t)k * 8);
	return (int64_t)c->e.size();
}
int64_t pgq_udf_scan_csr_w(pgq_state_t *s, int32_t id, void *out, int64_t cap) { // pgq_scan.cpp:113-153
	CsrRef c = s ? find_csr(s, id) : nullptr;
	if (!c) return fail("Constraint Error: CSR not found with ID " + std::to_string(id));
	const void *p = c->w.empty() ? (const void *)c->w_double.data() : (const void *)c->w.data();
	int64_t sz = (int64_t)(c->w.empty() ? c->w_double.size() : c->w.size());
	int64_t k = std::min<int64_t>(cap, sz);
	if (k > 0) memcpy(out, p, (size_t)k * 8);
	return sz;
}

pgq_csr_t *pgq_udf_device_csr(pgq_state_t *s, int32_t id) {
	CsrRef c = s ? find_csr(s, id) : nullptr;
	if (!c || !c->initialized_v) {
		fail("Constraint Error: CSR not found with ID " + std::to_string(id));
		return nullptr;
	}
	pgq_csr_t *d = device_csr(s, c, (int64_t)c->vsize - 2);
	if (!d) device_fail();
	return d;
}

} // extern "C"
