// pgq_glue.cpp — the DuckDB side of the drop-in: replacement bodies of the reference's search UDFs that forward to
// libpgq_hip (include/pgq_hip.h).  Names, argument lists, exception texts and NULL rules are the reference's:
//   IterativeLengthFunction        src/core/functions/scalar/iterativelength.cpp:34-143   (also registered as iterativelength2)
//   ShortestPathFunction           src/core/functions/scalar/shortest_path.cpp:43-207
//   CheapestPathLengthFunction     src/core/functions/scalar/cheapest_path_length.cpp:138-163
//   LocalClusteringCoefficientFunction  src/core/functions/scalar/local_clustering_coefficient.cpp:11-72
//   PageRankFunction               src/core/functions/scalar/pagerank.cpp:11-111
// plus PathFindingRelation, the whole-relation entry point SURVEY.md §8f rank 2 asks for (length and path of every
// pair from ONE search, no 2048-row ceiling: what replaces the binder's iterativelength-filter-then-shortestpath
// double BFS, src/core/functions/table/match.cpp:467-495,658-707).
// Compiled here only against glue/duckdb_stub.hpp + glue/duckpgq_stub.hpp (`make -C duckpgq-extension_amd/csrc
// glue-check`): DuckDB is not vendored in this tree.  In the extension the two stub includes become
// "duckdb.hpp" / the extension's own headers and this file replaces the five function bodies.
#include <cstring>

#include "duckpgq_stub.hpp"

namespace duckdb {

namespace {

[[noreturn]] void ThrowDevice() { throw InternalException("libpgq_hip: %s", pgq_last_error()); }

// UnifiedVectorFormat -> the three pointers the C ABI takes (selection / validity may be null)
pgq_vec_t AsVec(const UnifiedVectorFormat &f) {
	pgq_vec_t v;
	v.data = f.data;
	v.sel = f.sel ? f.sel->data() : nullptr;
	v.validity = f.validity.GetData();
	return v;
}

// Once-per-CSR upload; many worker threads reach the first chunk of a query together.  Only the first v[V] entries of
// e / edge_ids / w are read (the undirected CTE over-allocates); std::atomic<int64_t> is read as int64_t exactly like
// iterativelength.cpp:53 does.
pgq_csr_t *DeviceCSR(DuckPGQState &state, CSR &csr, int64_t v_size) {
	lock_guard<mutex> guard(state.csr_lock);
	if (csr.device) return csr.device;
	const void *w = nullptr;
	int w_type = PGQ_W_NONE;
	if (csr.initialized_w) {
		if (!csr.w.empty()) {
			w = csr.w.data();
			w_type = PGQ_W_INT64;
		} else if (!csr.w_double.empty()) {
			w = csr.w_double.data();
			w_type = PGQ_W_DOUBLE;
		}
	}
	// the CSR owns edge_ids for as long as it lives and ~CSR() frees the device handle first: the ids cross PCIe only when a
	// shortestpath call asks for them (PGQ_UPLOAD_LAZY_EDGE_IDS), off the path of the binder's iterativelength filter
	if (pgq_csr_upload_ex(v_size, reinterpret_cast<const int64_t *>(csr.v), csr.e.data(), csr.edge_ids.data(), w, w_type,
	                   PGQ_UPLOAD_LAZY_EDGE_IDS, &csr.device) != PGQ_OK)
		ThrowDevice();
	return csr.device;
}

struct SearchInputs {
	std::shared_ptr<DuckPGQState> state;
	CSR *csr;
	int32_t csr_id;
	int64_t v_size;
	UnifiedVectorFormat src, dst;
};

// common front of the three search UDFs: state lookup, the reference's exceptions, argument vectors
SearchInputs Prepare(DataChunk &args, ExpressionState &state, const char *what) {
	auto &info = state.expr.Cast<BoundFunctionExpression>().BindInfo()->Cast<IterativeLengthFunctionData>();
	SearchInputs in;
	in.state = GetDuckPGQState(info.context);
	in.csr_id = info.csr_id;
	auto entry = in.state->csr_list.find(info.csr_id);
	if (entry == in.state->csr_list.end() || !entry->second->initialized_v)
		throw ConstraintException(string("Need to initialize CSR before doing ") + what);
	in.csr = entry->second.get();
	UnifiedVectorFormat vsize_fmt;
	args.data[1].ToUnifiedFormat(args.size(), vsize_fmt);
	in.v_size = reinterpret_cast<const int64_t *>(vsize_fmt.data)[0];
	args.data[2].ToUnifiedFormat(args.size(), in.src);
	args.data[3].ToUnifiedFormat(args.size(), in.dst);
	return in;
}

} // namespace

void IterativeLengthFunction(DataChunk &args, ExpressionState &state, Vector &result) {
	SearchInputs in = Prepare(args, state, "shortest path");
	result.SetVectorType(VectorType::FLAT_VECTOR);
	auto result_data = FlatVector::GetDataMutable<int64_t>(result);
	ValidityMask &validity = FlatVector::ValidityMutable(result);
	validity.Initialize(args.size()); // the library writes whole mask words
	// NULL src -> NULL (payload -1), src == dst -> 0, unreachable -> NULL (payload -1), dst validity ignored
	if (pgq_iterativelength(DeviceCSR(*in.state, *in.csr, in.v_size), in.v_size, (int64_t)args.size(), AsVec(in.src),
	                        AsVec(in.dst), result_data, validity.GetData()) != PGQ_OK)
		ThrowDevice();
	in.state->csr_to_delete.insert(in.csr_id);
}

void ShortestPathFunction(DataChunk &args, ExpressionState &state, Vector &result) {
	SearchInputs in = Prepare(args, state, "shortest path");
	result.SetVectorType(VectorType::FLAT_VECTOR);
	auto entries = FlatVector::GetDataMutable<list_entry_t>(result);
	ValidityMask &validity = FlatVector::ValidityMutable(result);
	validity.Initialize(args.size());
	const int64_t *child = nullptr; // owned by the library until this thread's next pgq_shortestpath
	uint64_t child_len = 0;
	vector<uint64_t> offsets(args.size()), lengths(args.size());
	if (pgq_shortestpath(DeviceCSR(*in.state, *in.csr, in.v_size), in.v_size, (int64_t)args.size(), AsVec(in.src),
	                     AsVec(in.dst), offsets.data(), lengths.data(), validity.GetData(), &child, &child_len) != PGQ_OK)
		ThrowDevice();
	ListVector::Reserve(result, child_len);
	if (child_len) memcpy(FlatVector::GetDataMutable<int64_t>(ListVector::GetEntry(result)), child, child_len * sizeof(int64_t));
	ListVector::SetListSize(result, child_len);
	for (idx_t i = 0; i < args.size(); i++) { // [src, e1, v1, ..., dst]; NULL rows keep {0, 0}
		entries[i].offset = offsets[i];
		entries[i].length = lengths[i];
	}
	in.state->csr_to_delete.insert(in.csr_id);
}

void CheapestPathLengthFunction(DataChunk &args, ExpressionState &state, Vector &result) {
	auto &info = state.expr.Cast<BoundFunctionExpression>().BindInfo()->Cast<CheapestPathLengthFunctionData>();
	auto duckpgq_state = GetDuckPGQState(info.context);
	auto entry = duckpgq_state->csr_list.find(info.csr_id);
	if (entry == duckpgq_state->csr_list.end()) throw ConstraintException("CSR not found with ID " + std::to_string(info.csr_id));
	CSR &csr = *entry->second;
	if (!(csr.initialized_v && csr.initialized_e && csr.initialized_w))
		throw ConstraintException("Need to initialize CSR before doing cheapest path");
	UnifiedVectorFormat vsize_fmt, src, dst;
	args.data[1].ToUnifiedFormat(args.size(), vsize_fmt);
	const int64_t v_size = reinterpret_cast<const int64_t *>(vsize_fmt.data)[0];
	args.data[2].ToUnifiedFormat(args.size(), src);
	args.data[3].ToUnifiedFormat(args.size(), dst);
	result.SetVectorType(VectorType::FLAT_VECTOR);
	ValidityMask &validity = FlatVector::ValidityMutable(result);
	validity.Initialize(args.size());
	// BIGINT result for int64 weights, DOUBLE otherwise: the bind fixed the type (cheapest_path_length_function_data.cpp:25-29)
	void *out = csr.w.empty() ? static_cast<void *>(FlatVector::GetDataMutable<double>(result))
	                          : static_cast<void *>(FlatVector::GetDataMutable<int64_t>(result));
	if (pgq_cheapest_path_length(DeviceCSR(*duckpgq_state, csr, v_size), v_size, (int64_t)args.size(), AsVec(src), AsVec(dst),
	                             out, validity.GetData()) != PGQ_OK)
		ThrowDevice();
	duckpgq_state->csr_to_delete.insert(info.csr_id);
}

void LocalClusteringCoefficientFunction(DataChunk &args, ExpressionState &state, Vector &result) {
	auto &info = state.expr.Cast<BoundFunctionExpression>().BindInfo()->Cast<LocalClusteringCoefficientFunctionData>();
	auto duckpgq_state = GetDuckPGQState(info.context);
	auto entry = duckpgq_state->csr_list.find(info.csr_id);
	if (entry == duckpgq_state->csr_list.end()) throw ConstraintException("CSR not found. Is the graph populated?");
	CSR &csr = *entry->second;
	if (!(csr.initialized_v && csr.initialized_e))
		throw ConstraintException("Need to initialize CSR before doing local clustering coefficient.");
	const int64_t v_size = (int64_t)csr.vsize - 2;
	UnifiedVectorFormat src;
	args.data[1].ToUnifiedFormat(args.size(), src);
	result.SetVectorType(VectorType::FLAT_VECTOR);
	ValidityMask &validity = FlatVector::ValidityMutable(result);
	validity.Initialize(args.size());
	if (pgq_local_clustering_coefficient(DeviceCSR(*duckpgq_state, csr, v_size), v_size, (int64_t)args.size(), AsVec(src),
	                                     FlatVector::GetDataMutable<float>(result), validity.GetData()) != PGQ_OK)
		ThrowDevice();
	duckpgq_state->csr_to_delete.insert(info.csr_id);
}

void PageRankFunction(DataChunk &args, ExpressionState &state, Vector &result) {
	auto &info = state.expr.Cast<BoundFunctionExpression>().BindInfo()->Cast<PageRankFunctionData>();
	auto duckpgq_state = GetDuckPGQState(info.context);
	auto entry = duckpgq_state->csr_list.find(info.csr_id);
	if (entry == duckpgq_state->csr_list.end()) throw ConstraintException("CSR not found. Is the graph populated?");
	CSR &csr = *entry->second;
	if (!(csr.initialized_v && csr.initialized_e)) throw ConstraintException("Need to initialize CSR before running PageRank.");
	const int64_t v_size = (int64_t)csr.vsize - 2;
	UnifiedVectorFormat src;
	args.data[1].ToUnifiedFormat(args.size(), src);
	result.SetVectorType(VectorType::FLAT_VECTOR);
	ValidityMask &validity = FlatVector::ValidityMutable(result);
	validity.Initialize(args.size());
	if (pgq_pagerank(DeviceCSR(*duckpgq_state, csr, v_size), v_size, (int64_t)args.size(), AsVec(src),
	                 FlatVector::GetDataMutable<double>(result), validity.GetData()) != PGQ_OK)
		ThrowDevice();
	duckpgq_state->csr_to_delete.insert(info.csr_id);
}

// Whole-relation entry point: every (src, dst) row of a relation resident in host memory, any
// row count, answered by ONE search that yields the hop count AND the path — the list of a reachable pair holds
// 2 * hops + 1 elements, so `lengths` needs no second BFS (the binder evaluates iterativelength as a filter and then
// shortestpath on the same pairs: match.cpp:473-474,477).  Rows with hops outside [lower, upper] get valid = false.
struct PathFindingResult {
	vector<int64_t> hops;        // -1 for NULL / unreachable
	vector<list_entry_t> lists;  // into `child`
	vector<int64_t> child;       // [src, e1, v1, ..., dst] back to back
	vector<bool> valid;
};

PathFindingResult PathFindingRelation(DuckPGQState &state, int32_t csr_id, const vector<int64_t> &src,
                                      const vector<int64_t> &dst, int64_t lower, int64_t upper) {
	auto entry = state.csr_list.find(csr_id);
	if (entry == state.csr_list.end() || !entry->second->initialized_v)
		throw ConstraintException("Need to initialize CSR before doing shortest path");
	CSR &csr = *entry->second;
	const int64_t v_size = (int64_t)csr.vsize - 2;
	const idx_t n = src.size();
	PathFindingResult r;
	r.hops.assign(n, -1);
	r.lists.assign(n, list_entry_t { 0, 0 });
	r.valid.assign(n, false);
	pgq_csr_t *device = DeviceCSR(state, csr, v_size);
	if (pgq_num_enabled_devices() > 1 && n >= 4096) {
		// several GPUs behind this process (pgq_init_devices at extension load): contiguous shards of the relation, one
		// replica of the CSR per device, the ragged lists gathered behind each other
		vector<int64_t> len(n), off(n);
		int64_t used = 0;
		// first guess: a list of h hops holds 2h + 1 elements, 16 per row covers paths of up to 7 hops on average (social
		// graphs: 3-4); a call that needs more says how much in `used` and is repeated once
		r.child.resize(16 * n);
		int rc = pgq_shortestpath_multi(device, (int64_t)n, src.data(), dst.data(), len.data(), off.data(), r.child.data(),
		                                (int64_t)r.child.size(), &used);
		if (rc != PGQ_OK && used > (int64_t)r.child.size()) { // the first guess was too small: the call said what it needs
			r.child.resize((size_t)used);
			rc = pgq_shortestpath_multi(device, (int64_t)n, src.data(), dst.data(), len.data(), off.data(), r.child.data(),
			                            (int64_t)r.child.size(), &used);
		}
		if (rc != PGQ_OK) ThrowDevice();
		r.child.resize((size_t)used);
		for (idx_t i = 0; i < n; i++) {
			if (len[i] < 0) continue; // NULL source or unreachable
			r.hops[i] = len[i];
			r.lists[i] = list_entry_t { (uint64_t)off[i], (uint64_t)(2 * len[i] + 1) };
			r.valid[i] = len[i] >= lower && len[i] <= upper;
		}
		state.csr_to_delete.insert(csr_id);
		return r;
	}
	vector<uint64_t> offsets(n), lengths(n), mask((n + 63) / 64 + 1);
	pgq_vec_t s { src.data(), nullptr, nullptr }, d { dst.data(), nullptr, nullptr };
	const int64_t *child = nullptr;
	uint64_t child_len = 0;
	if (pgq_shortestpath(device, v_size, (int64_t)n, s, d, offsets.data(), lengths.data(), mask.data(), &child, &child_len) !=
	    PGQ_OK)
		ThrowDevice();
	r.child.assign(child, child + child_len);
	for (idx_t i = 0; i < n; i++) {
		if (!((mask[i >> 6] >> (i & 63)) & 1)) continue;
		const int64_t hops = ((int64_t)lengths[i] - 1) / 2;
		r.hops[i] = hops;
		r.lists[i] = list_entry_t { offsets[i], lengths[i] };
		r.valid[i] = hops >= lower && hops <= upper;
	}
	state.csr_to_delete.insert(csr_id);
	return r;
}

} // namespace duckdb
