/* pgq_hip.h — C ABI of libpgq_hip.so: DuckPGQ's path-finding hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary for the bodies of the reference's search UDFs.  Each entry point replaces
 * one reference function (file:line relative to the cwida/duckpgq-extension tree):
 *
 *   pgq_csr_upload            device mirror of `class CSR`                 src/include/duckpgq/core/utils/compressed_sparse_row.hpp:25-47
 *                             (called lazily, under csr_lock, by the first search UDF that sees the CSR;
 *                              built by create_csr_vertex/create_csr_edge  src/core/functions/scalar/csr_creation.cpp:86-198)
 *   pgq_csr_free              ~CSR() / DuckPGQState::QueryEnd / delete_csr   src/duckpgq_state.cpp:162-170, src/core/functions/scalar/csr_deletion.cpp:10-20
 *   pgq_iterativelength       IterativeLengthFunction                      src/core/functions/scalar/iterativelength.cpp:34-143
 *                             (also serves iterativelength2                src/core/functions/scalar/iterativelength2.cpp:33-130 — same results)
 *   pgq_shortestpath          ShortestPathFunction                         src/core/functions/scalar/shortest_path.cpp:43-207
 *   pgq_cheapest_path_length  CheapestPathLengthFunction                   src/core/functions/scalar/cheapest_path_length.cpp:138-163
 *   pgq_*_bulk_device         no reference counterpart: same semantics without the 2048-row chunk ceiling,
 *                             inputs/outputs resident in HBM (used by bench.py)
 *
 * Conventions
 *   - plain C, no exceptions, no C++ or torch types.  Every function returns PGQ_OK (0) or a negative
 *     pgq_status; pgq_last_error() gives the thread-local message of the last failure on this thread.
 *   - all entry points are thread-safe and re-entrant (DuckDB calls UDFs from many worker threads); a
 *     pgq_csr_t is immutable after upload and shared read-only.
 *   - pgq_vec_t is DuckDB's UnifiedVectorFormat as the UDFs consume it (iterativelength.cpp:57-64):
 *     row r reads data[sel ? sel[r] : r]; validity (uint64 words, bit set = valid, NULL = all valid) is
 *     indexed by that same selected position.
 *   - result validity masks are written for rows [0,n): bit set = valid.  The caller provides
 *     ceil(n/64) words.
 *   - ids are DuckDB rowids: dense 0..V-1.  Out-of-range src/dst (undefined behaviour in the reference,
 *     iterativelength.cpp:105,123) return PGQ_ERR_INVALID_ARG instead of corrupting memory.
 *   - the HIP extension is mandatory: without a usable gfx950 device every call fails with
 *     PGQ_ERR_NO_DEVICE.  There is no CPU fallback in this library.
 */
#ifndef PGQ_HIP_H
#define PGQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pgq_status {
	PGQ_OK = 0,
	PGQ_ERR_NO_DEVICE = -1,    /* no HIP device / runtime failure at init */
	PGQ_ERR_HIP = -2,          /* a HIP call failed; message has the hipError string */
	PGQ_ERR_OOM = -3,          /* device or host allocation failed */
	PGQ_ERR_INVALID_ARG = -4,  /* NULL handle, id out of range, bad weight type, ... */
	PGQ_ERR_NOT_WEIGHTED = -5, /* cheapest path on a CSR without weights ("Need to initialize CSR before doing cheapest path") */
	PGQ_ERR_UNSUPPORTED = -6   /* e.g. negative weights */
} pgq_status;

typedef struct pgq_csr pgq_csr_t; /* opaque: CSR resident in HBM */

typedef struct pgq_vec {
	const void *data;
	const uint32_t *sel;      /* nullable */
	const uint64_t *validity; /* nullable */
} pgq_vec_t;

/* weight types, same numbering as csr_get_w_type (src/core/functions/scalar/csr_get_w_type.cpp:16-36) */
#define PGQ_W_NONE 0
#define PGQ_W_INT64 1
#define PGQ_W_DOUBLE 2

/* ---- process ------------------------------------------------------------------------------------- */

/* Idempotent. device < 0: use env PGQ_DEVICE, else LOCAL_RANK, else 0. */
int pgq_init(int device);
int pgq_device_count(void);
/* Multi-GPU inside one process (SURVEY.md §8b `pgq_init(device_mask)`): enables several devices of this node for the
 * *_multi entry points; the first one is the device pgq_init binds (single-device calls keep using it).  Peer access
 * (xGMI) is enabled between them where the hardware allows.  pgq_init_devices accepts a repeated index (tests on a
 * one-GPU box). */
int pgq_init_devices(const int *devices, int n);
int pgq_init_mask(uint64_t device_mask);
int pgq_num_enabled_devices(void);
const char *pgq_last_error(void);
const char *pgq_version(void);

/* ---- CSR ------------------------------------------------------------------------------------------ */

/* Host arrays in the reference's layout: offsets = CSR::v (at least V+1 int64; the reference allocates
 * V+2), adj = CSR::e, edge_ids = CSR::edge_ids (nullable: slot index is used), w = CSR::w (int64) or
 * CSR::w_double (double) selected by w_type.  Only the first offsets[V] entries of adj/edge_ids/w are read
 * (the undirected CTE over-allocates).  The device copy stores int32 adjacency.
 * SCALE LIMITS of the device mirror (the reference's CSR holds int64 everywhere, compressed_sparse_row.hpp:34-35): a CSR
 * with V >= 2^31 - 1 vertices or E = offsets[V] >= 2^31 entries is REFUSED with PGQ_ERR_UNSUPPORTED by all three
 * upload / build entry points — vertex ids are int32 on the device, the padded lists of the pair-centric kernels are
 * addressed by 32-bit group indices, their queue entries hold 32-bit list positions.  Every BASELINE configuration is
 * inside (the largest, the SF100 reply forest at V = 2^28: E = 2.2 x 10^8).  One call takes at most 2^31 - 1 rows.
 * When the ~40 bytes per edge of the pair-centric layout do not fit in device memory the upload still succeeds and the
 * searches run without the pre-pass (same answers, slower on scattered pairs): pgq_csr_has_prepass_layout() tells. */
int pgq_csr_upload(int64_t V, const int64_t *offsets, const int64_t *adj, const int64_t *edge_ids, const void *w,
                   int w_type, pgq_csr_t **out);
/* pgq_csr_upload with flags.  PGQ_UPLOAD_LAZY_EDGE_IDS: the caller keeps `edge_ids` valid and unchanged until pgq_csr_free
 * (the reference's host CSR owns the vector for as long as the CSR exists: compressed_sparse_row.hpp:34-35, and the device
 * handle dies with it); the library then does NOT copy the E x 8 bytes at upload — iterativelength, cheapest_path_length and
 * the analytics never read edge ids — but on the first call that needs them (shortestpath, pgq_csr_download,
 * pgq_csr_replicate).  Half of a query's upload bytes, off the critical path of the binder's iterativelength filter
 * (match.cpp:658-671 runs it before any shortestpath call). */
#define PGQ_UPLOAD_LAZY_EDGE_IDS 1u
int pgq_csr_upload_ex(int64_t V, const int64_t *offsets, const int64_t *adj, const int64_t *edge_ids, const void *w,
                      int w_type, unsigned flags, pgq_csr_t **out);
/* Same, but the four arrays already live in device memory of the current device (they are copied). */
int pgq_csr_upload_device(int64_t V, const int64_t *d_offsets, const int64_t *d_adj, const int64_t *d_edge_ids,
                          const void *d_w, int w_type, pgq_csr_t **out);
/* CSR construction on the GPU from edge-table rows resident in HBM: what
 * create_csr_vertex + create_csr_edge (src/core/functions/scalar/csr_creation.cpp:86-198) compute, with the slot
 * order of the reference's single-threaded schedule (rows of one source keep their table order).  d_edge_id may be
 * NULL (row index is the edge id); d_w: int64 or double column per w_type. */
int pgq_csr_build_device(int64_t V, int64_t n_rows, const int64_t *d_src, const int64_t *d_dst,
                         const int64_t *d_edge_id, const void *d_w, int w_type, pgq_csr_t **out);
/* Copies the device CSR back in the reference's layout (int64); any pointer may be NULL.  get_csr_v/e/w analogue
 * (src/core/functions/table/pgq_scan.cpp:15-153) for the device-built CSR. */
int pgq_csr_download(const pgq_csr_t *csr, int64_t *offsets, int64_t *adj, int64_t *edge_ids, void *w);
/* Copies the device CSR (base arrays and everything derived at upload) to every other enabled device with peer copies
 * over xGMI: the CSR is replicated, the pairs are sharded (north_star).  Idempotent; replicas die with the handle. */
int pgq_csr_replicate(pgq_csr_t *csr);
int pgq_csr_free(pgq_csr_t *csr);
int64_t pgq_csr_num_vertices(const pgq_csr_t *csr);
int64_t pgq_csr_num_edges(const pgq_csr_t *csr);
int pgq_csr_w_type(const pgq_csr_t *csr);
int64_t pgq_csr_device_bytes(const pgq_csr_t *csr);
/* 1: the padded adjacency + slot descriptors of the pair-centric pre-pass were built at upload; 0: they were not (option
 * meet_layout = 0, or no memory for them): searches go through the lane batches only. */
int pgq_csr_has_prepass_layout(const pgq_csr_t *csr);

/* ---- searches, chunk form (host memory, UnifiedVectorFormat in, FLAT vector out) --------------------- */

/* iterativelength(csr_id, V, src, dst) -> BIGINT.  NULL src -> NULL (payload -1); src == dst -> 0; reachable
 * -> hop count; unreachable -> NULL (payload -1).  dst validity is ignored exactly like the reference.
 * Cost of one call (round 4): the columns are resolved into a pinned staging block of a pooled workspace that the
 * kernels read and write directly — two or three kernel launches and one wait, no copy commands (0.03 ms for one row,
 * 0.07 ms for 2048 rows on the SF100-shaped graph); re-entrant, every calling thread gets its own workspace and stream. */
int pgq_iterativelength(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, int64_t *out_len,
                        uint64_t *out_valid);

/* iterativelengthbidirectional(csr_id, V, src, dst) -> BIGINT (src/core/functions/scalar/iterativelength_bidirectional.cpp:43-153,
 * intended semantics: the reference's version is unreachable from the binder and indexes its backward CSR wrongly).
 * The same hop counts as pgq_iterativelength, found by one bidirectional BFS per row: forward over the CSR from src,
 * backward over the transposed CSR (built at upload) from dst, always expanding the cheaper side. */
int pgq_iterativelength_bidirectional(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst,
                                      int64_t *out_len, uint64_t *out_valid);

/* shortestpath(csr_id, V, src, dst) -> LIST(BIGINT) [src, e1, v1, ..., ek, dst].  list entries go to
 * out_offset/out_length (list_entry_t fields), the child payload to *out_child (owned by the library, valid
 * until the next pgq_shortestpath call on the same thread or pgq_thread_release).  NULL rows keep entry {0,0}. */
int pgq_shortestpath(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, uint64_t *out_offset,
                     uint64_t *out_length, uint64_t *out_valid, const int64_t **out_child, uint64_t *out_child_len);

/* cheapest_path_length(csr_id, V, src, dst) -> BIGINT | DOUBLE (by the CSR's weight type; out is int64_t* or
 * double*).  NULL src, NULL dst or unreachable -> NULL. */
int pgq_cheapest_path_length(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, pgq_vec_t dst, void *out,
                             uint64_t *out_valid);

void pgq_thread_release(void); /* drop this thread's result arena */
/* Frees the pooled per-call workspaces (search state sized by V x lanes is kept between calls for reuse) and the
 * freed CSR / upload blocks kept for the next upload (PGQ_ALLOC_CACHE_MB). */
int pgq_release_cached_memory(void);

/* ---- searches, bulk form (device memory, no chunk ceiling) ---------------------------------------------- */

/* d_src/d_dst/d_out_len: n int64 each in HBM.  d_out_len[i] = hop count, 0 for src==dst, -1 for NULL. */
int pgq_iterativelength_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                    int64_t *d_out_len);
/* the same rows through one bidirectional search each (pgq_iterativelength_bidirectional without the chunk ceiling) */
int pgq_iterativelength_bidirectional_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                                  int64_t *d_out_len);
/* Rows in host memory, answered by all enabled devices: contiguous shards, one host thread and one CSR replica per
 * device, results gathered into out_len (same values as pgq_iterativelength_bulk_device: hop count, 0, or -1). */
int pgq_iterativelength_multi(pgq_csr_t *csr, int64_t n, const int64_t *src, const int64_t *dst, int64_t *out_len);
/* shortestpath (ShortestPathFunction, src/core/functions/scalar/shortest_path.cpp:43-207) by all enabled devices:
 * out_len / out_offset as in the bulk form; the lists of all shards are gathered
 * into child (host memory, capacity child_cap int64) in row order of the shards.  *child_used = elements needed;
 * PGQ_ERR_INVALID_ARG when child_cap is too small (out_len is complete, the payload is not usable). */
int pgq_shortestpath_multi(pgq_csr_t *csr, int64_t n, const int64_t *src, const int64_t *dst, int64_t *out_len,
                           int64_t *out_offset, int64_t *child, int64_t child_cap, int64_t *child_used);
/* cheapest_path_length (CheapestPathLengthFunction, cheapest_path_length.cpp:138-163) by all enabled devices:
 * out = n int64 or double (by weight type), out_valid = n bytes. */
int pgq_cheapest_path_length_multi(pgq_csr_t *csr, int64_t n, const int64_t *src, const int64_t *dst, void *out,
                                   uint8_t *out_valid);
/* Measurement helper: same search, additionally d_out_te[i] = edges traversed by pair i's own level-synchronous BFS
 * up to the level that reaches dst (all levels if unreachable) — the numerator of bench.py's MTEPS. */
int pgq_traversed_edges_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                    int64_t *d_out_len, int64_t *d_out_te);
/* d_out_len as above; paths are written packed into d_child (capacity child_cap int64), entry i at
 * d_out_offset[i] with 2*len+1 elements (no entry for NULL rows).  *child_used returns the elements needed;
 * PGQ_ERR_INVALID_ARG if child_cap was too small: d_out_len is still complete and *child_used says what to provide,
 * the child payload is not usable. */
int pgq_shortestpath_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                 int64_t *d_out_len, int64_t *d_out_offset, int64_t *d_child, int64_t child_cap,
                                 int64_t *child_used);
/* out: int64 or double per weight type; NULL -> validity payload -1 (int64) / NaN is never used: d_out_valid
 * (n bytes, 1 = valid) carries validity. */
int pgq_cheapest_path_length_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, const int64_t *d_dst,
                                         void *d_out, uint8_t *d_out_valid);

/* ---- the other CSR consumers ------------------------------------------------------------- */

/* local_clustering_coefficient(csr_id, id) -> FLOAT   src/core/functions/scalar/local_clustering_coefficient.cpp:11-72
 * (bit-identical: integer counting + the reference's three float operations).  NULL rows -> NULL; ids outside [0,V)
 * (undefined behaviour in the reference) -> PGQ_ERR_INVALID_ARG.  Bulk form: d_src[i] < 0 = NULL row (value 0). */
int pgq_local_clustering_coefficient(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, float *out, uint64_t *out_valid);
int pgq_local_clustering_coefficient_bulk_device(pgq_csr_t *csr, int64_t n, const int64_t *d_src, float *d_out);
/* pagerank(csr_id, id) -> DOUBLE   src/core/functions/scalar/pagerank.cpp:11-111: power iteration over V + 2 entries,
 * damping 0.85, threshold 1e-6, computed once per handle; rows outside [0, V + 2) or NULL -> NULL.  Partial sums follow
 * the reference's accumulation order; the dangling total is a two-level sum (tests: 1e-12 relative).
 * pgq_pagerank_device copies all V + 2 ranks into d_rank (may be NULL) and reports the iteration count. */
int pgq_pagerank(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, double *out, uint64_t *out_valid);
int pgq_pagerank_device(pgq_csr_t *csr, double *d_rank, int *iterations);

/* weakly_connected_component(csr_id, id) -> BIGINT   src/core/functions/scalar/weakly_connected_component.cpp:36-104: the
 * component id is the root the reference's sequential union-find (Link(i, neighbour): i's root under the neighbour's,
 * vertices and slots in CSR order) ends in.  The edges that change that forest are the minimum spanning forest under
 * the weights "slot index": found on the device (Boruvka rounds), replayed in slot order by the reference's own Link
 * (V - 1 edges at most).  Computed once per handle.  NULL id -> NULL; ids outside [0, V + 2) -> NULL; the two trailing
 * forest entries behave like the reference's (V: its own root; V + 1: the root of vertex 0).
 * pgq_weakly_connected_component_device copies all V + 2 ids into d_ids (may be NULL: just compute). */
int pgq_weakly_connected_component(pgq_csr_t *csr, int64_t V, int64_t n, pgq_vec_t src, int64_t *out, uint64_t *out_valid);
int pgq_weakly_connected_component_device(pgq_csr_t *csr, int64_t *d_ids);

/* ---- tuning & measurement --------------------------------------------------------------------------------- */

/* Knobs (also readable from the environment at pgq_init: PGQ_WORDS, PGQ_PUSH_DIV, PGQ_PROFILE ...).
 * key/value strings; returns PGQ_ERR_INVALID_ARG for unknown keys. */
int pgq_set_option(const char *key, const char *value);
/* Current value of a knob (integers are returned as doubles). */
int pgq_get_option(const char *key, double *value);
/* the value the library ships with for this key, whatever the process or the environment has set since */
int pgq_get_default_option(const char *key, double *value);
/* Options of ONE handle: the first pgq_csr_set_option copies the process-wide set into the handle; searches on this
 * handle then run under the copy (on every host thread that works for the call), so that two connections, or a test,
 * can tune their own CSR without touching each other's.  Options consumed at upload (meet_align, hub_chunk, ...) are
 * not affected. */
int pgq_csr_set_option(pgq_csr_t *csr, const char *key, const char *value);
int pgq_csr_get_option(pgq_csr_t *csr, const char *key, double *value);

/* Per-thread counters of the searches run since the last reset, and per-kernel-class HIP-event time
 * (only accumulated while option "profile" = "1"). */
#define PGQ_KCLASS_MAX 16
typedef struct pgq_stats {
	int64_t batches;          /* lane batches started */
	int64_t levels;           /* BFS levels executed (all batches) */
	int64_t push_levels;      /* of which top-down */
	int64_t pull_levels;      /* of which bottom-up */
	int64_t edges_scanned;    /* physical adjacency entries read by push+pull kernels */
	int64_t word_gathers;     /* lane-words (8 B) gathered / RMW'd along those entries */
	int64_t frontier_vertices;/* sum over levels of vertices with a non-empty frontier word */
	int64_t unique_sources;   /* lanes used */
	int64_t pairs;            /* rows handled */
	int64_t deferred_pairs;   /* rows re-run in a narrow straggler batch */
	int64_t meet_pairs;       /* rows answered by the pair-centric pre-pass (distance <= 3, NULL, trivial) */
	double algo_bytes[PGQ_KCLASS_MAX]; /* algorithmic bytes per kernel class (DESIGN.md formulas) */
	double kernel_ms[PGQ_KCLASS_MAX];  /* HIP-event time per kernel class (profile=1) */
	int64_t launches[PGQ_KCLASS_MAX];
	int64_t spec_batches;     /* lane batches whose levels were enqueued ahead of the host (one wait per batch) */
	int64_t spec_levels;      /* levels that ran that way */
	int64_t spec_aborts;      /* batches whose enqueued levels were called off on the device (plan mismatch / too short) */
	int64_t host_waits;       /* stream synchronisations of the lane-batched search (lane assignment, levels, results) */
	int64_t ball_segments;    /* round 6: source runs (cut at 1024-row windows) the source-centric kernel answered, one ball each */
	int64_t ball_calls;       /* calls (or straggler passes) the source-centric kernel took */
} pgq_stats_t;
const char *pgq_kclass_name(int kclass); /* NULL past the last class */
/* pgq_stats_t grows at its END from release to release (pgq_version() names the release).  pgq_get_stats_sized writes at most
 * struct_size bytes — what a caller compiled against an older header passes as sizeof(pgq_stats_t) — so an old binary keeps
 * working against a newer library; pgq_get_stats(out) = pgq_get_stats_sized(out, sizeof(pgq_stats_t) of THIS header) and is
 * only safe for callers compiled against it. */
int pgq_get_stats_sized(pgq_stats_t *out, size_t struct_size);
int pgq_get_stats(pgq_stats_t *out);
int pgq_reset_stats(void);

/* Raw copy-bandwidth probe (bytes moved / second, read+write) used by bench.py for the measured HBM ceiling. */
int pgq_measure_copy_bandwidth(int64_t bytes, int iters, double *out_gbps);

#ifdef __cplusplus
}
#endif
#endif /* PGQ_HIP_H */
