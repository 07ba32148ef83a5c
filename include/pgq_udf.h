/* pgq_udf.h — C ABI of libpgq_udf.so: DuckDB-free host mirror of DuckPGQ's scalar-function layer (L5/L6).
 *
 * The reference's UDF bodies take `(DataChunk &args, ExpressionState &state, Vector &result)` and reach their
 * CSR through `DuckPGQState` (per connection).  DuckDB is not available in this build environment, so the same
 * logic is exposed here with the chunk's columns passed as pgq_vec_t (UnifiedVectorFormat) and the result as a
 * FLAT vector + validity mask.  A DuckDB glue file only has to forward `ToUnifiedFormat` output to these
 * calls and rethrow pgq_udf_last_error() (INTEGRATION.md shows it).  One function per reference UDF:
 *
 *   pgq_udf_create_csr_vertex       CreateCsrVertexFunction      src/core/functions/scalar/csr_creation.cpp:86-110
 *   pgq_udf_create_csr_edge         CreateCsrEdgeFunction        src/core/functions/scalar/csr_creation.cpp:112-198
 *   pgq_udf_bind_search             IterativeLengthBind          src/core/functions/function_data/iterative_length_function_data.cpp:18-30
 *   pgq_udf_iterativelength         IterativeLengthFunction      src/core/functions/scalar/iterativelength.cpp:34-143
 *   pgq_udf_iterativelength2        IterativeLength2Function     src/core/functions/scalar/iterativelength2.cpp:33-130 (same results)
 *   pgq_udf_iterativelengthbidirectional  IterativeLengthBidirectionalFunction  src/core/functions/scalar/iterativelength_bidirectional.cpp:43-153 (intended semantics)
 *   pgq_udf_shortestpath            ShortestPathFunction         src/core/functions/scalar/shortest_path.cpp:43-207
 *   pgq_udf_bind_cheapest           CheapestPathLengthBind       src/core/functions/function_data/cheapest_path_length_function_data.cpp:7-32
 *   pgq_udf_cheapest_path_length    CheapestPathLengthFunction   src/core/functions/scalar/cheapest_path_length.cpp:138-163
 *   pgq_udf_delete_csr              DeleteCsrFunction            src/core/functions/scalar/csr_deletion.cpp:10-20
 *   pgq_udf_csr_get_w_type          GetCsrWTypeFunction          src/core/functions/scalar/csr_get_w_type.cpp:16-36
 *   pgq_udf_reachability            ReachabilityFunction         src/core/functions/scalar/reachability.cpp:165-254 (intended semantics)
 *   pgq_state_query_end             DuckPGQState::QueryEnd       src/duckpgq_state.cpp:162-170
 *   pgq_udf_scan_csr_v / _e / _w    get_csr_v / get_csr_e / get_csr_w   src/core/functions/table/pgq_scan.cpp:84-153
 *
 * Error convention: 0 on success; -1 with pgq_udf_last_error() holding the reference's exception text
 * ("Constraint Error: ...", "Invalid Input Error: ...") or the device library's message.
 */
#ifndef PGQ_UDF_H
#define PGQ_UDF_H

#include "pgq_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgq_state pgq_state_t; /* DuckPGQState: csr_list + csr_lock + csr_to_delete (duckpgq_state.hpp:36-38) */

pgq_state_t *pgq_state_new(void);
void pgq_state_free(pgq_state_t *);
int pgq_state_query_end(pgq_state_t *);
const char *pgq_udf_last_error(void);

int pgq_udf_create_csr_vertex(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t dense_id, pgq_vec_t cnt,
                              int64_t *out, uint64_t *out_valid);
/* w may be NULL (7-argument form); w_type PGQ_W_INT64 / PGQ_W_DOUBLE for the 8-argument form */
int pgq_udf_create_csr_edge(pgq_state_t *, int32_t id, int64_t V, int64_t e_sum, int64_t e_count, int64_t n,
                            pgq_vec_t src, pgq_vec_t dst, pgq_vec_t edge_id, const pgq_vec_t *w, int w_type,
                            int32_t *out, uint64_t *out_valid);

int pgq_udf_bind_search(pgq_state_t *, int32_t id);
int pgq_udf_iterativelength(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                            int64_t *out, uint64_t *out_valid);
int pgq_udf_iterativelength2(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                             int64_t *out, uint64_t *out_valid);
/* iterativelengthbidirectional(id, V, src, dst) -> BIGINT (src/core/functions/scalar/iterativelength_bidirectional.cpp:43-153).
 * The reference alternates a source-side and a destination-side BFS but indexes its inputs through a uint8_t* and
 * walks the forward CSR on the destination side: untested, parity unpinned.  What it intends to
 * return is the hop count, which is what this entry point returns (same values as pgq_udf_iterativelength). */
int pgq_udf_iterativelengthbidirectional(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                                         int64_t *out, uint64_t *out_valid);
int pgq_udf_shortestpath(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                         uint64_t *out_offset, uint64_t *out_length, uint64_t *out_valid, const int64_t **out_child,
                         uint64_t *out_child_len);
/* *ret_type: PGQ_W_INT64 -> BIGINT result, PGQ_W_DOUBLE -> DOUBLE result */
int pgq_udf_bind_cheapest(pgq_state_t *, int32_t id, int *ret_type);
int pgq_udf_cheapest_path_length(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                                 void *out, uint64_t *out_valid);
/* reachability(id, is_variant, V, src, dst) -> BOOL; the reference's BOOL argument only selects one of its three
 * internal BFS modes (reachability.cpp:154-163) and does not change the result, so it is not part of this call.
 * Boolean reachability with the semantics the reference intends (reachability.cpp:165-254 indexes its inputs
 * through a uint8_t* and has no test: parity unpinned): out[i] = 1 iff a path exists
 * (src == dst counts as reachable). */
int pgq_udf_reachability(pgq_state_t *, int32_t id, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                         uint8_t *out, uint64_t *out_valid);

/* The other CSR consumers.  Errors as in the reference: "Constraint Error: CSR not found. Is the
 * graph populated?" / "Need to initialize CSR before ..." (local_clustering_coefficient.cpp:16-23, pagerank.cpp:17-24,
 * weakly_connected_component.cpp:41-48).  All three run on the device (pgq_local_clustering_coefficient / pgq_pagerank /
 * pgq_weakly_connected_component).  The component id of weakly_connected_component is the root the reference's
 * sequential union-find schedule ends in (weakly_connected_component.cpp:14-34,83-90 — the goldens pin it): the device
 * finds the spanning forest under that schedule's edge order, the reference's Link replays its edges. */
int pgq_udf_local_clustering_coefficient(pgq_state_t *, int32_t id, int64_t n, pgq_vec_t src, float *out, uint64_t *out_valid);
int pgq_udf_pagerank(pgq_state_t *, int32_t id, int64_t n, pgq_vec_t src, double *out, uint64_t *out_valid);
int pgq_udf_weakly_connected_component(pgq_state_t *, int32_t id, int64_t n, pgq_vec_t src, int64_t *out, uint64_t *out_valid);

int pgq_udf_delete_csr(pgq_state_t *, int32_t id, int *out_flag);
int pgq_udf_csr_get_w_type(pgq_state_t *, int32_t id, int32_t *out);

/* table-function scans used by the reference's tests to observe the CSR: copy up to cap entries, return count */
int64_t pgq_udf_scan_csr_v(pgq_state_t *, int32_t id, int64_t *out, int64_t cap);
int64_t pgq_udf_scan_csr_e(pgq_state_t *, int32_t id, int64_t *out, int64_t cap);
int64_t pgq_udf_scan_csr_w(pgq_state_t *, int32_t id, void *out, int64_t cap);
/* device handle of a CSR (uploads it if needed); NULL on error */
pgq_csr_t *pgq_udf_device_csr(pgq_state_t *, int32_t id);

#ifdef __cplusplus
}
#endif
#endif
