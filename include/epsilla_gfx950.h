/* epsilla_gfx950.h — C ABI of libepsilla_gfx950.so, the MI355X (gfx950 / CDNA4) implementation of
 * Epsilla's ANN hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference/engine):
 *
 *   eps_index_create / destroy      VecSearchExecutor ctor + ExecutorPool slot
 *                                   (db/execution/vec_search_executor.cpp:29-73, executor_pool.hpp:10-31);
 *                                   metric selection = GetDistFunc (db/index/index.cpp:10-35)
 *   eps_index_attach_rows           the executor borrowing TableSegmentMVP::vector_tables_[f]
 *                                   (db/table_segment_mvp.hpp:85, alloc .cpp:106-111): row-major float[n][dim]
 *   eps_index_set_deleted           ConcurrentBitset bytes of table_segment->deleted_
 *                                   (utils/concurrent_bitset.cpp:9-18), read per Search() call (:839)
 *   eps_index_set_int_filter        the `ID < N`-class post-filter ExprEvaluator::LogicalEvaluate applies to
 *                                   result candidates (vec_search_executor.cpp:905-927, query/expr/expr_evaluator.cpp:170-258)
 *   eps_index_build                 ANNGraphSegment::BuildFromVectorTable (db/ann_graph_segment.cpp:201-242)
 *   eps_index_set_graph / get_graph the public CSR members offset_table_/neighbor_list_/navigation_point_
 *                                   (db/ann_graph_segment.hpp:44-49) the executor is constructed from
 *   eps_index_save_graph/load_graph ANNGraphSegment::SaveANNGraph (.cpp:156-199) / file ctor (.cpp:39-98),
 *                                   byte-identical `ann_graph_<field>.bin`
 *   eps_index_search                VecSearchExecutor::Search (.cpp:833-935) for a BATCH of queries: mode
 *                                   selection, SearchImpl (:518-715), BruteForceSearch (:717-768),
 *                                   PreFilterBruteForceSearch (:770-831), tail merge, post-filter; results are
 *                                   what the caller reads from search_result_ / distance_ (hpp:51-52)
 *   eps_normalize_rows              Normalize (db/vector.cpp:60-69) and the insert-time normalisation
 *                                   (db/table_segment_mvp.cpp:574-587)
 *   eps_merge_topk                  (new) merges per-shard top-k lists after the RCCL all-gather
 *
 * Pointer arguments documented "host or device" are classified with hipPointerGetAttributes; device
 * pointers are used in place (no copy) on the index's stream.  Every function returns a status code
 * from the reference's own table (utils/error.hpp:11-41): 0 = OK.  Calls on one handle must not overlap
 * (same contract as one VecSearchExecutor: one thread per executor instance); different handles are
 * independent.  The library never falls back to a CPU path: without a usable gfx950 device
 * eps_index_create fails with EPS_INFRA_UNEXPECTED_ERROR.
 */
#ifndef EPSILLA_GFX950_H_
#define EPSILLA_GFX950_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: utils/error.hpp:11-41 */
#define EPS_OK 0
#define EPS_USER_ERROR 30000
#define EPS_INFRA_UNEXPECTED_ERROR 40001
#define EPS_DB_UNEXPECTED_ERROR 50001
#define EPS_DB_UNSUPPORTED_ERROR 50002
#define EPS_NOT_IMPLEMENTED_ERROR 50009
#define EPS_INVALID_PAYLOAD 50400

/* meta::MetricType (db/catalog/meta_types.hpp:44-50) as used by GetDistFunc; "smaller = closer" for all:
 * EUCLIDEAN -> squared L2, COSINE -> 1 - dot (rows/queries normalised by the caller or eps_normalize_rows),
 * DOT_PRODUCT -> -dot. */
#define EPS_METRIC_EUCLIDEAN 0
#define EPS_METRIC_COSINE 1
#define EPS_METRIC_DOT_PRODUCT 2

/* search modes */
#define EPS_MODE_REFERENCE 0 /* VecSearchExecutor::Search's own mode selection (:855-935)          */
#define EPS_MODE_FLAT 1      /* exact flat scan of [0,n_total) (the BruteForceSearch answer for any n) */
#define EPS_MODE_GRAPH 2     /* graph traversal + tail even when n_indexed < BruteforceThreshold       */

/* flat-scan engines (EPS_MODE_FLAT, and the flat branches of EPS_MODE_REFERENCE) */
#define EPS_FLAT_AUTO 0
#define EPS_FLAT_STREAM 1 /* fp32 streaming scan, HBM-bound; any batch size                            */
#define EPS_FLAT_MFMA 2   /* fp16-MFMA lower-bound filter over a device-side half mirror + exact fp32
                             re-rank; returns the same exact answer; for large batches                 */
#define EPS_FLAT_MFMA_I8 3 /* the same with an int8 mirror as the filter's operand (twice the matrix rate,
                             half the mirror; the bound is computed from the stored residuals, so the
                             answer is again the exact one); tables an 8-bit grid cannot serve and batches
                             whose candidate lists overflow run the fp16 pass.  EPS_FLAT_AUTO picks this
                             form when it picks the matrix engine.                                     */

/* filter comparison operators for eps_index_set_int_filter */
#define EPS_OP_NONE 0
#define EPS_OP_LT 1
#define EPS_OP_LE 2
#define EPS_OP_EQ 3
#define EPS_OP_GE 4
#define EPS_OP_GT 5
#define EPS_OP_NE 6

/* Compiled filter: a postfix program over the packed attribute row of a candidate
 * (TableSegmentMVP::attribute_table_, rows of primitive_offset_ bytes), evaluated on a stack of doubles exactly as
 * ExprEvaluator::NumEvaluate / LogicalEvaluate do (query/expr/expr_evaluator.cpp:127-258): every number is a double,
 * comparisons and AND / OR / NOT produce 0 / 1, the row passes iff the final value is non-zero.  PUSH_* take `arg` = byte
 * offset of the attribute inside the row; PUSH_CONST takes `dval`; PUSH_DIST is the candidate's distance (`@distance`). */
#define EPS_FOP_PUSH_CONST 1
#define EPS_FOP_PUSH_DIST 2
#define EPS_FOP_PUSH_I8 3
#define EPS_FOP_PUSH_I16 4
#define EPS_FOP_PUSH_I32 5
#define EPS_FOP_PUSH_I64 6
#define EPS_FOP_PUSH_F32 7
#define EPS_FOP_PUSH_F64 8
#define EPS_FOP_PUSH_BOOL 9   /* true iff the byte is non-zero (expr_evaluator.cpp:56-59) */
#define EPS_FOP_ADD 10
#define EPS_FOP_SUB 11
#define EPS_FOP_MUL 12
#define EPS_FOP_DIV 13
#define EPS_FOP_MOD 14        /* fmod */
#define EPS_FOP_LT 15
#define EPS_FOP_LE 16
#define EPS_FOP_EQ 17
#define EPS_FOP_NE 18
#define EPS_FOP_GE 19
#define EPS_FOP_GT 20
#define EPS_FOP_AND 21
#define EPS_FOP_OR 22
#define EPS_FOP_NOT 23
#define EPS_FOP_EQ_BOOL 24    /* EQ / NE between boolean operands (:206-209) */
#define EPS_FOP_NE_BOOL 25
typedef struct eps_filter_op {
  int32_t op;
  int32_t arg;
  int64_t ival; /* reserved */
  double dval;
} eps_filter_op;

typedef struct eps_index eps_index; /* opaque */

/* Search knobs; mirror vectordb::Config (config/config.hpp:17-25) and the executor ctor arguments. */
typedef struct eps_search_params {
  int32_t mode;            /* EPS_MODE_*                                                       */
  int32_t flat_engine;     /* EPS_FLAT_*                                                       */
  int32_t prefilter;       /* Config::PreFilter: flat scan with the filter applied first        */
  int32_t intra_threads;   /* Config::IntraQueryThreads (T); the device runs T candidate
                              expansions of a query concurrently per round (1 = the reference's
                              deterministic single-thread order)                               */
  int64_t master_queue;    /* Config::MasterQueueSize (L)                                      */
  int64_t local_queue;     /* Config::LocalQueueSize (result cap, :872)                        */
  int64_t sync_interval;   /* Config::GlobalSyncInterval (expansions per worker per round)     */
  int32_t filter_in_traversal; /* 0 (default): the reference's semantics - deleted rows and the filter are judged on the
                              final top-L walk only (vec_search_executor.cpp:905-927), so a selective filter returns fewer than
                              `limit` rows.  1 (NOT the reference's answer): graph searches judge every row
                              they EVALUATE and return the closest visible ones; invisible rows still steer the walk.          */
  int32_t reserved;
} eps_search_params;

/* Build knobs; defaults = NSGConfig(45,50,300,100) (db/ann_graph_segment.cpp:29). */
typedef struct eps_build_params {
  int64_t search_length;
  int64_t out_degree;
  int64_t candidate_pool_size;
  int64_t knng;
  uint32_t seed;      /* rand_r state the NSG stage starts from (nsg.cpp:19 uses 100) */
  int32_t reserved;
} eps_build_params;

/* counters of the last eps_index_search call (the reference's commented-out
 * count_distance_computation_, vec_search_executor.hpp:162, revived) */
typedef struct eps_search_stats {
  int64_t dist_evals;       /* distance evaluations, summed over the batch                 */
  int64_t expansions;       /* graph nodes expanded, summed over the batch                 */
  int64_t rerank_rows;      /* rows re-ranked in exact fp32 by the MFMA engine             */
  int64_t overflow_queries; /* queries that fell back from the MFMA filter to the fp32 scan */
  double kernel_ms;         /* device time of the call measured with hipEvents on its stream */
  double main_kernel_ms;    /* device time of the dominant kernel only                      */
  int64_t main_kernel_launches;
  int64_t main_kernel_rows; /* rows covered by the launch timed in main_kernel_ms              */
  int64_t main_kernel_queries; /* queries covered by that launch (large batches run in slices)  */
  int64_t main_kernel_bits; /* operand width of that launch: 32 (fp32 stream / traversal), 16 or 8 (matrix engine) */
  double filter_ms_all;     /* device time of ALL filter-stage launches of the call (the dominant kernel runs once per stage; r4) */
  int64_t filter_rows_all;  /* rows those launches covered, summed (x main_kernel_queries = the call's matrix work)              */
  int64_t i8_folded;        /* 1: the 8-bit pass of this call ran with per-row margins folded into the rows' start values (a table whose rows differ: clamped / forced outlier rows; r4) */
  int64_t i8_declined;      /* 1: this call probed the 8-bit pass on this table, found its bound too loose for the data and ran the fp16 pass (r4) */
  int64_t one_pass;         /* 1: a handful of queries (<= 16, k <= 64; a deleted bitset, an int-column test or a compiled filter program) answered by ONE streaming pass over the 8-bit mirror + one re-rank (stream8_kernel.hpp) instead of the staged filter chain (r4) */
  int64_t i8_rotated;       /* 1: the 8-bit pass of this call ran in the table's ROTATED frame: rows and queries quantised as R x, R a fixed orthogonal map (signs, a permutation, 256-point Walsh-Hadamard blocks) - distances do not change, the grid's step and with it the bound's margin shrink on tables whose energy sits in a few columns (unit-norm embedding rows); chosen per table when its mirror is built (r6) */
} eps_search_stats;

void eps_default_search_params(eps_search_params* p);
void eps_default_build_params(eps_build_params* p);

int32_t eps_index_create(int64_t dim, int32_t metric, int32_t device, eps_index** out);
/* Hash-sharded index over `shards` GPUs of THIS process: row i of the table lives on shard i mod shards (device
 * devices[i mod shards]) as local row i / shards; every shard answers the whole batch on its rows, the per-shard top-k lists
 * are pushed peer to peer (xGMI) to the device that holds the caller's result buffers (devices[0] for host results) and merged
 * there.  The handle works with every eps_index_* entry point below.  Rows: a HOST table (eps_index_attach_rows: each shard reads
 * its rows with one strided copy) or, shard by shard, rows that already live on the shard's device (eps_index_attach_shard_rows).
 * Queries and results: host buffers, or (r4) device buffers on any device of the group - a shard on another device gets the
 * queries with one peer copy, the merge writes straight into the caller's ids / dist / counts (which must live together), and as
 * with a plain index the caller synchronises (eps_index_synchronize).  Bitsets, filter columns and attribute rows are host
 * buffers.  eps_index_build builds one graph per shard, eps_index_save/load_graph use <path>.shard<s> files; eps_index_set_graph
 * (one graph over the whole table), eps_index_set_stream and eps_index_search_walk are not available.
 * devices may repeat an ordinal (several shards on one GPU). */
int32_t eps_index_create_sharded(int64_t dim, int32_t metric, const int32_t* devices, int32_t shards, eps_index** out);
int32_t eps_index_destroy(eps_index* h);
const char* eps_index_last_error(const eps_index* h);
/* What kind of failure the last error was (r6; callers branch on this, never on the text).  EPS_ERRCLASS_DEVICE_RANGE: the request is valid for
 * the reference (config/config.hpp:28-44 accepts SearchQueueSize / LocalQueueSize up to 10^7 and IntraQueryThreads up to 128 at any
 * out-degree; VecSearchExecutor::Search, vec_search_executor.cpp:833-935, has no result cap) but lies outside what the device traversal runs
 * (queues <= 2^20 keys, IntraQueryThreads x out-degree <= 2048, <= 1024 rows per query under filter_in_traversal or when a graph result is
 * merged with an un-indexed tail): the same request with mode = EPS_MODE_FLAT is answered exactly, which is what the drop-in adapter does. */
#define EPS_ERRCLASS_OTHER 0
#define EPS_ERRCLASS_DEVICE_RANGE 1
int32_t eps_index_last_error_class(const eps_index* h);

/* use an existing HIP stream (hipStream_t passed as void*); NULL = the index's own non-blocking stream,
 * hipStreamLegacy ((void*)1) = the legacy default stream.  Device buffers handed to the index (rows, queries,
 * outputs, bitsets, columns) must be ready on the index's stream: share the producer's stream or synchronise
 * the producer before the call. */
int32_t eps_index_set_stream(eps_index* h, void* hip_stream);
int32_t eps_index_synchronize(eps_index* h);

/* rows: host or device, row-major float[n][dim].  Host rows are copied to HBM; device rows are
 * borrowed (caller keeps them alive and immutable for [0,n)).  Replaces any previous attachment. */
int32_t eps_index_attach_rows(eps_index* h, const float* rows, int64_t n);
/* rows beyond the current count (the un-indexed tail, :885-900); only for host-attached (owned) stores */
int32_t eps_index_append_rows(eps_index* h, const float* rows, int64_t n_new);
/* rows of ONE shard of a hash-sharded index: local row l is global row l * shards + shard; host memory (copied) or memory of the
 * shard's own device (borrowed).  Once every shard holds its rows the table has sum(n_local) rows; the counts must be the hash
 * split of that sum (shard s holds ceil((n - s) / shards) rows) or searches and builds fail with EPS_USER_ERROR.  On a plain index
 * shard 0 is the index itself.  (r4; replaces what TableSegmentMVP::vector_tables_ is for one GPU, table_segment_mvp.hpp:85) */
int32_t eps_index_attach_shard_rows(eps_index* h, int32_t shard, const float* rows, int64_t n_local);
/* dst gets its OWN device copy of the first n rows of src - an index of the same kind on the same device(s): plain / plain, or two
 * shard groups over the same device list - with one device-to-device copy per shard; replaces dst's rows like eps_index_attach_rows.
 * What TableMVP::Rebuild's snapshot is for the reference (table_mvp.cpp:133-156: the build works on rows [0, n) while inserts and
 * searches go on): the drop-in builds the new graph on such a copy and swaps it in, so a rebuild never blocks queries (r4). */
int32_t eps_index_clone_rows(eps_index* dst, eps_index* src, int64_t n);
int64_t eps_index_row_count(const eps_index* h);

/* On-disk table segment of the reference (`<db>/<table_id>/data_mvp.bin`, TableSegmentMVP::SaveTableSegment,
 * db/table_segment_mvp.cpp:939-1010) read straight into HBM: header (record count, first id), deleted
 * bitset, packed attribute rows (record_count x primitive_offset bytes), variable-length attributes (skipped), then one
 * float[record_count][dim] table per dense vector field.  The caller describes the schema-dependent sizes; the index takes
 * the rows of dense field `field` (its dim must equal the index's), the deleted bitset, and keeps the attribute rows on the
 * device for eps_index_set_filter_program(ops, nops, NULL, 0, 0).  *n_out = record count. */
typedef struct eps_table_layout {
  int64_t primitive_offset;   /* bytes per packed attribute row (TableSegmentMVP::primitive_offset_)            */
  int32_t var_len_attrs;      /* strings / sparse vectors per record (var_len_attr_num_)                       */
  int32_t dense_fields;       /* dense vector fields, in schema order (dense_vector_num_)                      */
  const int64_t* dense_dims;  /* their dimensions (vector_dims_)                                                */
  int32_t field;              /* which of them to load                                                          */
  int32_t reserved;
} eps_table_layout;
int32_t eps_index_load_table(eps_index* h, const char* data_mvp_path, const eps_table_layout* layout, int64_t* n_out);

/* global id = local row index * stride + base (hash sharding by row index: stride = #shards, base = rank) */
int32_t eps_index_set_id_map(eps_index* h, int64_t base, int64_t stride);

/* deleted: host or device bitset, bit (i&7) of byte (i>>3); nbytes >= ceil(n/8); NULL clears */
int32_t eps_index_set_deleted(eps_index* h, const uint8_t* bits, int64_t nbytes);
/* integer attribute column: value of row i at column + i*stride_bytes, width_bytes in {1,2,4,8}, signed;
 * host or device; rows failing `value <op> constant` are filtered. EPS_OP_NONE clears. */
int32_t eps_index_set_int_filter(eps_index* h, const void* column, int64_t stride_bytes, int32_t width_bytes,
                                 int32_t op, int64_t constant);

/* compiled filter program (see eps_filter_op): `rows` = packed attribute rows (host or device), row i at rows + i*stride_bytes,
 * n_rows of them.  Replaces eps_index_set_int_filter's filter; nops = 0 clears.  A row is visible iff it is not deleted
 * and the program leaves a non-zero value.  Programs are limited to 64 instructions and a stack depth of 16.  Host rows
 * are copied to the device, all of them, on every call. */
int32_t eps_index_set_filter_program(eps_index* h, const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride_bytes,
                                     int64_t n_rows);
/* The same with flags.  EPS_FILTER_ROWS_APPEND_ONLY: the caller promises that the host rows it handed over before FROM THIS
 * POINTER (same stride, same table attached) have not changed since - TableSegmentMVP::attribute_table_ is such a table: an
 * update is delete + insert (db/table_segment_mvp.cpp:476-587) - so only the rows beyond those already on the device are
 * uploaded.  Re-attaching rows (eps_index_attach_rows) forgets the cached copy. */
#define EPS_FILTER_ROWS_APPEND_ONLY 1
int32_t eps_index_set_filter_program_ex(eps_index* h, const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride_bytes,
                                        int64_t n_rows, int32_t flags);

/* graph over rows [0,n): built on the device, or supplied / exported as the reference's CSR */
int32_t eps_index_build(eps_index* h, int64_t n, const eps_build_params* p);
/* One stage of the build on its own: NsgIndex::SyncPrune's sort + SelectEdge (db/index/nsg/nsg.cpp:557-567, 655-685) for
 * caller-supplied candidate lists.  Node nodes[i] with candidates cands[i][0..cands_per_node) (-1 = none; the node itself is
 * skipped): candidates sorted by (L2 distance to the node, id), the closest kept, every further one of the first `depth`
 * (candidate_pool_size; <= 0 = all) kept iff no kept neighbour is closer to it than the node is (MRNG rule), at most out_degree.
 * out_ids [m][out_degree] (-1 padded), out_deg [m]; host arrays. */
int32_t eps_index_select_edges(eps_index* h, const int64_t* nodes, int64_t m, const int64_t* cands, int32_t cands_per_node,
                               int32_t depth, int32_t out_degree, int64_t* out_ids, int32_t* out_deg);
/* Another stage on its own: NsgIndex::InterInsert (db/index/nsg/nsg.cpp:583-653) over all n = row-count nodes, on caller-supplied
 * edge lists ids [n][out_degree] with deg[v] valid entries each (what Link leaves).  Every edge v->u offers v to u; u keeps its
 * own edges plus the offers if they fit into out_degree, otherwise SelectEdge(limit = false) over all of them sorted by (L2
 * distance, id) - applied once per node to the whole candidate set (the reference applies it incrementally in node order, which
 * differs where a list overflows: see DESIGN.md 3.4).  Every offer is considered (kept in a CSR by receiving node), so the result
 * does not depend on the order in which the device produces them.  out_ids [n][out_degree] (-1 padded), out_deg [n]; host arrays. */
int32_t eps_index_inter_insert(eps_index* h, const int64_t* ids, const int32_t* deg, int64_t n, int32_t out_degree, int64_t* out_ids,
                               int32_t* out_deg);
/* The first stage on its own: the K-nearest-neighbour graph of rows [0,n) that the build links (the reference's KNNGraph /
 * NN-Descent stage, db/index/knn/knn.hpp:90-135; K = p->knng).  out_ids [n][min(K, n-1)] host, closest first, -1 padded, the node
 * itself excluded.  Exact below 65 536 rows; above, the 128 closest by approximate key re-ranked in exact fp32. */
int32_t eps_index_knn_graph(eps_index* h, int64_t n, const eps_build_params* p, int64_t* out_ids);
/* The Link stage on its own (NsgIndex::Link without InterInsert, db/index/nsg/nsg.cpp:488-516: per node GetNeighbors :158-268 on the
 * kNN graph from the navigation node's first search_length neighbours, then SyncPrune :540-580 = pool + the node's own kNN row,
 * sorted, SelectEdge over the first candidate_pool_size).  knn: [n][min(p->knng, n-1)] host (NULL: the device's own kNN stage);
 * navigation_point < 0: the closest row to the centroid (reported in *nav_out).  out_ids [n][out_degree] (-1 padded), out_deg [n]. */
int32_t eps_index_link(eps_index* h, int64_t n, const int64_t* knn, int64_t navigation_point, const eps_build_params* p, int64_t* out_ids,
                       int32_t* out_deg, int64_t* nav_out);
int32_t eps_index_set_graph(eps_index* h, int64_t n, const int64_t* offsets, const int64_t* neighbors,
                            int64_t navigation_point);
int32_t eps_index_graph_info(const eps_index* h, int64_t* n, int64_t* edges, int64_t* navigation_point);
int32_t eps_index_get_graph(const eps_index* h, int64_t* offsets, int64_t* neighbors);
int32_t eps_index_save_graph(eps_index* h, const char* path);
int32_t eps_index_load_graph(eps_index* h, const char* path);

/* queries: host or device float[nq][dim]; ids_out int64[nq][k], dist_out float[nq][k], counts_out
 * int32[nq] (host or device, all three the same kind).  Unused slots are id -1 / +inf.
 * COSINE queries must already be normalised (TableMVP::Search does it, table_mvp.cpp:333-349). */
int32_t eps_index_search(eps_index* h, const float* queries, int64_t nq, int32_t k, const eps_search_params* p,
                         int64_t* ids_out, float* dist_out, int32_t* counts_out);
/* The candidates the reference's post-filter loop walks (vec_search_executor.cpp:905-927), for filters that only the host
 * DBMS can evaluate (strings, LIKE, IN, geo): same traversal / flat scan, same tail merge with searchLimit =
 * min(n_indexed, limit, L_local) (:872-900), deleted rows and device-side filters already removed, in walk order; at most
 * `cap` per query (cap >= limit).  ids_out [nq][cap], dist_out [nq][cap], counts_out [nq].  The caller applies its
 * predicate to these <= cap candidates and keeps the first `limit` that pass - O(L) host work as in the reference,
 * instead of O(N). */
int32_t eps_index_search_walk(eps_index* h, const float* queries, int64_t nq, int32_t limit, int32_t cap, const eps_search_params* p,
                              int64_t* ids_out, float* dist_out, int32_t* counts_out);
int32_t eps_index_last_stats(const eps_index* h, eps_search_stats* out);
/* main-kernel milliseconds (hipEvent pairs recorded on the index's stream) of the most recent search calls, oldest
 * first, at most min(cap, 64); synchronises the stream.  Returns the number written.  Lets a caller time a run of
 * searches without a host sync inside it. */
int32_t eps_index_kernel_times(eps_index* h, double* ms_out, int32_t cap);

/* in-place L2 normalisation of float[n][dim] (host or device). only_if_nonzero = 1 reproduces the
 * insert path (sum > 1e-10), 0 the query path (unconditional). */
int32_t eps_normalize_rows(float* rows, int64_t n, int64_t dim, int32_t only_if_nonzero, int32_t device,
                           void* hip_stream);

/* merges `shards` sorted top-k lists per query by (dist,id): dist/ids are [shards][nq][k] device or host
 * arrays (e.g. the output of an all-gather); out_* are [nq][k]. */
int32_t eps_merge_topk(const float* dist, const int64_t* ids, int32_t shards, int64_t nq, int32_t k,
                       float* out_dist, int64_t* out_ids, int32_t device, void* hip_stream);

/* the same merge over ONE gathered device buffer ("one ncclAllGather over a packed buffer"): shard s
 * contributed shard_stride_bytes bytes holding int64 ids[nq][k] at offset 0 and float dist[nq][k] at dist_offset_bytes. */
int32_t eps_merge_topk_packed(const void* gathered, int64_t shard_stride_bytes, int64_t dist_offset_bytes, int32_t shards,
                              int64_t nq, int32_t k, float* out_dist, int64_t* out_ids, int32_t device, void* hip_stream);

/* ---- The exchange step of the sharded path, owned by the library (r6).  One process per GPU: every rank holds, for the whole
 * batch, the top-k lists over ITS shard (global ids); eps_exchange_allgather_merge packs them as [ids int64[nq][k] | dist f32[nq][k]]
 * (12 k nq bytes per rank), runs ONE ncclAllGather on the RCCL communicator the handle owns (xGMI inside a node) and merges the `world`
 * sorted lists of every query by (dist, id) - every rank ends with the same global top-k in out_ids / out_dist.  Lists, results and the
 * stream belong to the rank's device; the call is asynchronous on that stream.  RCCL is resolved at run time (dlopen; a copy the process
 * already holds - PyTorch ships one - is the one used).  Bootstrap, the caller's only part: rank 0 fills 128 bytes with
 * eps_exchange_unique_id and sends them to every rank (bench.py: a gloo broadcast), every rank calls eps_exchange_create (collective:
 * returns when all `world` ranks have joined).  *out is set even on failure (its message: eps_exchange_last_error).  world <= 16.
 * The reference has no counterpart (no sharding, SURVEY 8e "Semantics vs reference"): sharded exact top-k == unsharded exact top-k. */
#define EPS_EXCHANGE_ID_BYTES 128
typedef struct eps_exchange eps_exchange;
int32_t eps_exchange_unique_id(void* id128);
int32_t eps_exchange_create(int32_t rank, int32_t world, const void* id128, int32_t device, eps_exchange** out);
int32_t eps_exchange_allgather_merge(eps_exchange* x, const int64_t* ids, const float* dist, int64_t nq, int32_t k, int64_t* out_ids, float* out_dist,
                                     void* hip_stream);
/* device time of the two parts - all-gather, merge - of the last calls (hipEvents on their stream, kept for 64 calls; waits for them):
 * us_pairs[2 i], us_pairs[2 i + 1] for the oldest .. newest of min(calls, 64, max_calls) calls; returns how many (-1: failure) */
int32_t eps_exchange_times(eps_exchange* x, double* us_pairs, int32_t max_calls);
/* what the handle runs on: rank, world, RCCL's version number and the path of the library that answers */
int32_t eps_exchange_info(eps_exchange* x, int32_t* rank, int32_t* world, int32_t* rccl_version, char* rccl_path, int64_t cap);
/* The b = 1 form of the same step ("for b=1 prefer direct P2P stores into a peer-mapped buffer + flag"; r6): for a handful of queries -
 * nq * k * 12 <= 16 KB per rank - a collective costs more than the bytes it moves.  Every rank owns a MAILBOX in device memory that every peer maps
 * (hipIpc); a call stores this rank's packed lists into every peer's mailbox, raises a flag there (release at system scope), waits on the device until
 * all `world` flags of the OWN mailbox show the call's number, and merges - three small launches on the caller's stream, no collective, no host round
 * trip.  Setup: eps_exchange_create_direct (no communicator; or an exchange made by eps_exchange_create), eps_exchange_mailbox_export on every rank
 * (64 bytes), the host gathers the `world` handles by whatever it has, eps_exchange_mailbox_connect(x, handles[world][64]).  Ranks may share a device.
 * A peer that does not deliver within ~10 s is reported by a later call (EPS_INFRA_UNEXPECTED_ERROR) instead of hanging the stream. */
#define EPS_EXCHANGE_HANDLE_BYTES 64
int32_t eps_exchange_create_direct(int32_t rank, int32_t world, int32_t device, eps_exchange** out);
int32_t eps_exchange_mailbox_export(eps_exchange* x, void* handle64);
int32_t eps_exchange_mailbox_connect(eps_exchange* x, const void* handles);
int32_t eps_exchange_direct_merge(eps_exchange* x, const int64_t* ids, const float* dist, int64_t nq, int32_t k, int64_t* out_ids, float* out_dist,
                                  void* hip_stream);
const char* eps_exchange_last_error(eps_exchange* x);
void eps_exchange_destroy(eps_exchange* x);

/* Engine-selection switches.  The library reads NO environment variable: what used to be lab switches are entries of one
 * process-wide table that only this call writes (name, value as text; value == NULL removes the entry; name == NULL empties
 * the table).  No entry changes a result - they pick between engines that return the same bits (tests/ run every answer
 * through both sides of each) or turn diagnostics on:
 *   EPS_DEBUG (stage log on stderr), EPS_TRV_PROF (traversal phase profile on stderr),
 *   EPS_TRV_PREFILTER 0|1 (8-bit lower-bound test of the traversal), EPS_TRV_VISITED bitmap|stamps, EPS_TRV_STAMP_START, EPS_TRV_WAVES 4|8|16, EPS_TRV_PER_CU, EPS_TRV_LDS_KB,
 *   EPS_FLAT_ONE_PASS 0|1, EPS_ONE_PASS_TIMED, EPS_S8_WG_PER_CU, EPS_S8_HOST_WORDS 0|1, EPS_S8_TWO_LAUNCHES 0|1, EPS_S8_MAX_Q 1..16, EPS_S8_MAX_K 1..64, EPS_S8_FILTER_PROGRAMS 0|1, EPS_S8_RERANK 0|1, EPS_HOST_STAGING 0|1, EPS_RERANK_SPLIT, EPS_MFMA_BITS 8|16, EPS_MFMA_MAX_BATCH,
 *   EPS_MFMA_PROBE, EPS_MFMA_SEED, EPS_MFMA_GROUPSYNC, EPS_MFMA_SYNC_SHIFT, EPS_MFMA_STAGES, EPS_MFMA_KERNEL, EPS_MFMA_NARROW,
 *   EPS_MFMA_TWO_PER_CU, EPS_MFMA_FOLD, EPS_MFMA_MANTISSA, EPS_BUILD_BLOCK, EPS_BUILD_VISITED, EPS_BUILD_PREFILTER,
 *   EPS_MIRROR_ROTATE 0|1 (frame of the 8-bit grid: identity | rotated; unset = chosen per table when its mirror is first built; read at that moment only),
 *   EPS_MIRROR_CLIP e (the grid cuts 10^-e of the sampled values off each tail, 1 <= e <= 9, in either frame; unset = 10^-7, rotated frame 10^-6;
 *   read when the mirror is first built), EPS_S8_FOLD 0|1 (the one-pass search on tables whose margins are folded per batch).
 * Switches that make answers WRONG on purpose (kernel ablations for profiling) exist only in a lab build (-DEPS_LAB), which
 * also falls back to the environment for names the table does not hold.  Returns EPS_OK. */
int32_t eps_set_tuning(const char* name, const char* value);

#ifdef __cplusplus
}
#endif
#endif /* EPSILLA_GFX950_H_ */
