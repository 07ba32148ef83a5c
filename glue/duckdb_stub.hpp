// duckdb_stub.hpp — the DuckDB declarations the path-finding call sites touch, reduced to what glue/pgq_glue.cpp needs
// to TYPE-CHECK in this tree (DuckDB itself is an empty, un-vendored submodule here: SURVEY.md Appendix C).  Shapes
// follow DuckDB's public API as the reference uses it (file:line = reference call site that needs the declaration):
//   UnifiedVectorFormat / SelectionVector / ValidityMask   iterativelength.cpp:57-64,98-101
//   FlatVector::GetDataMutable / ValidityMutable            iterativelength.cpp:66-70
//   ListVector::Reserve / GetEntry / SetListSize            shortest_path.cpp:70-76,149-204
//   list_entry_t                                            shortest_path.cpp:75,162-163
//   ExpressionState / BoundFunctionExpression::BindInfo     iterativelength.cpp:36-38
// Nothing here is linked or shipped; the real headers replace it when the glue is dropped into the extension.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace duckdb {
using idx_t = uint64_t;
using sel_t = uint32_t;
using data_ptr_t = uint8_t *;
using std::vector;
using std::string;
using std::mutex;
using std::lock_guard;
using std::unique_ptr;

struct list_entry_t {
	uint64_t offset;
	uint64_t length;
};

enum class VectorType : uint8_t { FLAT_VECTOR, CONSTANT_VECTOR, DICTIONARY_VECTOR };

struct SelectionVector {
	sel_t *sel_vector = nullptr;
	sel_t *data() const { return sel_vector; } // nullptr = identity selection
	idx_t get_index(idx_t i) const { return sel_vector ? sel_vector[i] : i; }
};

struct ValidityMask {
	uint64_t *mask = nullptr; // nullptr = all valid
	std::shared_ptr<std::vector<uint64_t>> owned;
	uint64_t *GetData() const { return mask; }
	void Initialize(idx_t count) {
		owned = std::make_shared<std::vector<uint64_t>>((count + 63) / 64, ~uint64_t(0));
		mask = owned->data();
	}
	bool RowIsValid(idx_t i) const { return !mask || ((mask[i >> 6] >> (i & 63)) & 1); }
	void SetInvalid(idx_t i) {
		if (!mask) Initialize(2048);
		mask[i >> 6] &= ~(uint64_t(1) << (i & 63));
	}
};

struct UnifiedVectorFormat {
	const SelectionVector *sel = nullptr;
	data_ptr_t data = nullptr;
	ValidityMask validity;
};

class Vector {
public:
	void ToUnifiedFormat(idx_t count, UnifiedVectorFormat &out);
	void SetVectorType(VectorType t);
};

class DataChunk {
public:
	vector<Vector> data;
	idx_t size() const;
};

struct FlatVector {
	template <class T> static T *GetDataMutable(Vector &v);
	static ValidityMask &ValidityMutable(Vector &v);
};
struct ListVector {
	static void Reserve(Vector &v, idx_t capacity);
	static Vector &GetEntry(Vector &v);
	static void SetListSize(Vector &v, idx_t size);
};

class ClientContext;
struct FunctionData {
	virtual ~FunctionData() = default;
	template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
};
struct Expression {
	template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
};
struct BoundFunctionExpression : Expression {
	FunctionData *bind_info = nullptr;
	FunctionData *BindInfo() const { return bind_info; }
};
struct ExpressionState {
	Expression &expr;
};

struct ConstraintException : std::runtime_error {
	explicit ConstraintException(const string &m) : std::runtime_error("Constraint Error: " + m) {}
};
struct InternalException : std::runtime_error {
	template <class... A> explicit InternalException(const char *fmt, A... a) : std::runtime_error(Format(fmt, a...)) {}
	template <class... A> static string Format(const char *fmt, A... a) {
		char buf[512];
		snprintf(buf, sizeof(buf), fmt, a...);
		return buf;
	}
};
} // namespace duckdb
