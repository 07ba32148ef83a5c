// duckpgq_stub.hpp — the reference's OWN types the search UDFs use, reduced to their data members, with the one change
// this project asks for: `class CSR` gains a device mirror that dies with it.
//   class CSR                      src/include/duckpgq/core/utils/compressed_sparse_row.hpp:25-47
//   DuckPGQState                   src/include/duckpgq_state.hpp:36-38, src/duckpgq_state.cpp:162-186
//   IterativeLengthFunctionData    src/include/duckpgq/core/functions/function_data/iterative_length_function_data.hpp
//   CheapestPathLengthFunctionData src/include/duckpgq/core/functions/function_data/cheapest_path_length_function_data.hpp
#pragma once
#include <atomic>

#include "duckdb_stub.hpp"
#include "pgq_hip.h"

namespace duckdb {

class CSR {
public:
	CSR() = default;
	~CSR() { // PATCH: QueryEnd / delete_csr erase the unique_ptr<CSR>, which now frees the HBM copy as well
		delete[] v;
		if (device) pgq_csr_free(device);
	}
	std::atomic<int64_t> *v = nullptr;
	vector<int64_t> e;
	vector<int64_t> edge_ids;
	vector<int64_t> w;
	vector<double> w_double;
	bool initialized_v = false, initialized_e = false, initialized_w = false;
	size_t vsize = 0;
	pgq_csr_t *device = nullptr; // PATCH: uploaded lazily by the first search UDF, under DuckPGQState::csr_lock
};

class DuckPGQState {
public:
	std::map<int32_t, unique_ptr<CSR>> csr_list;
	mutex csr_lock;
	std::unordered_set<int32_t> csr_to_delete;
};
std::shared_ptr<DuckPGQState> GetDuckPGQState(ClientContext &context);

struct IterativeLengthFunctionData : FunctionData {
	IterativeLengthFunctionData(ClientContext &context, int32_t csr_id) : context(context), csr_id(csr_id) {}
	ClientContext &context;
	int32_t csr_id;
};
struct ShortestPathFunctionData : IterativeLengthFunctionData { using IterativeLengthFunctionData::IterativeLengthFunctionData; };
struct CheapestPathLengthFunctionData : IterativeLengthFunctionData { using IterativeLengthFunctionData::IterativeLengthFunctionData; };
struct LocalClusteringCoefficientFunctionData : IterativeLengthFunctionData { using IterativeLengthFunctionData::IterativeLengthFunctionData; };
struct PageRankFunctionData : IterativeLengthFunctionData { using IterativeLengthFunctionData::IterativeLengthFunctionData; };
} // namespace duckdb
