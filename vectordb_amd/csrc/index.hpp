// Host side of libepsilla_gfx950: the per-field index object behind the C ABI (include/epsilla_gfx950.h).
// It plays the role of one VecSearchExecutor + its ANNGraphSegment (reference:
// engine/db/execution/vec_search_executor.hpp:30-74, engine/db/ann_graph_segment.hpp:22-55) with the
// vector table, graph and scratch mirrored in HBM.
#pragma once
#include <atomic>
#include <thread>
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/epsilla_gfx950.h"
#include "kernels.hpp"

namespace eps {

struct DevBuf {  // growable device allocation
  void* p = nullptr;
  size_t cap = 0;
  ~DevBuf();
  // returns false on allocation failure
  bool reserve(size_t bytes);
  void release();
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

// Reclaimable scratch (r6): a device allocation that only makes a path faster (the traversal's visited stamps: 4 bytes x nodes x slots,
// tens of GB) is registered here with a flag its owner holds while host code is about to launch kernels on it.  When ANY DevBuf::reserve of the
// process fails, every registered buffer whose owner is not in that window is released (hipFree waits for the kernels already launched) and
// the allocation is tried again; the owner finds its buffer gone at its next search and takes the form that needs no such table.
struct ScratchClaim {
  DevBuf* buf = nullptr;
  ScratchClaim();
  ~ScratchClaim();
  ScratchClaim(const ScratchClaim&) = delete;
  ScratchClaim& operator=(const ScratchClaim&) = delete;
  // The owner's window (from "is the table there?" to its last launch on it) and the reclaimer's release exclude each other.  enter(): the owner -
  // WAITS for a release in progress on another thread; false = this thread is already inside (a nested call: nothing to take, nothing to give back).
  // try_enter(): the reclaimer - never waits, never takes what its own thread is using.
  bool enter() {
    if (holder_.load() == std::this_thread::get_id()) return false;
    mu_.lock();
    holder_.store(std::this_thread::get_id());
    return true;
  }
  bool try_enter() {
    if (holder_.load() == std::this_thread::get_id() || !mu_.try_lock()) return false;
    holder_.store(std::this_thread::get_id());
    return true;
  }
  void leave() {
    holder_.store(std::thread::id());
    mu_.unlock();
  }

 private:
  std::mutex mu_;
  std::atomic<std::thread::id> holder_{};
};
size_t scratch_reclaim();   // bytes released

// Engine-selection switches (eps_set_tuning, include/epsilla_gfx950.h): the value of `name` in the process-wide table, or null.  The
// product library never reads the environment; a lab build (-DEPS_LAB) falls back to getenv for names the table does not hold.
const char* tune_env(const char* name);
// the integer value of a switch, read ONCE (a second lookup may find the entry gone: eps_set_tuning is a process-wide runtime call), or `dflt`
int tune_int(const char* name, int dflt);

struct BuildStage;   // stage-level entry of the graph build (below)
struct Quant8View {   // the table's 8-bit mirror as other kernels see it (mfma_filter.hip)
  const signed char* x8 = nullptr;
  const int* acc0 = nullptr;
  const float* scal8 = nullptr;
  const float* mu = nullptr;   // [d_pad8] the grid's centre (one value per column)
  int d_pad8 = 0;
  int cols8 = 0;          // columns of a mirror row that carry values: dim (identity frame), dim rounded up to 256 (rotated frame, r6) - what a kernel that reads
                          // a row's leading bytes only (the traversal's prefilter) must cover
  float step = 1.f, u = 1.f;
  int64_t epoch8 = 0;     // counts the mirror's (re)builds: row constants copied elsewhere (the graph's edge constants) are stale when it moves
  bool per_batch = false; // acc0 carries per-batch margins (fold8): it changes with every batch of queries
};
struct HalfMirror;   // fp16 mirror + per-row bounds for the MFMA filter engine (mfma_filter.hip)
struct GraphDev;     // device CSR + traversal scratch (traverse.hip)

// What the C ABI dispatches to: one device index, or a group of them over a hash-sharded table (shard_group.cpp).
class IndexBase {
 public:
  virtual ~IndexBase() {}
  virtual int32_t set_stream(void* s) = 0;
  virtual int32_t synchronize() = 0;
  virtual int32_t attach_rows(const float* rows, int64_t n) = 0;
  virtual int32_t append_rows(const float* rows, int64_t n_new) = 0;
  // rows of one shard of a hash-sharded index (local row l = global row l * shards + shard); a plain index is its own shard 0
  virtual int32_t attach_shard_rows(int32_t shard, const float* rows, int64_t n_local) = 0;
  // an OWNED device copy of the first n rows of `src` (an index of the same kind on the same device(s)): what a build on the side
  // works on while `src` keeps serving and growing
  virtual int32_t clone_rows(IndexBase& src, int64_t n) = 0;
  virtual int32_t set_id_map(int64_t base, int64_t stride) = 0;
  virtual int32_t set_deleted(const uint8_t* bits, int64_t nbytes) = 0;
  virtual int32_t set_int_filter(const void* column, int64_t stride, int32_t width, int32_t op, int64_t constant) = 0;
  virtual int32_t set_filter_program(const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride, int64_t n_rows, int32_t flags = 0) = 0;
  virtual int32_t build(int64_t n, const eps_build_params* p) = 0;
  virtual int32_t set_graph(int64_t n, const int64_t* off, const int64_t* nbr, int64_t nav) = 0;
  virtual int32_t graph_info(int64_t* n, int64_t* edges, int64_t* nav) const = 0;
  virtual int32_t get_graph(int64_t* off, int64_t* nbr) const = 0;
  virtual int32_t save_graph(const char* path) = 0;
  virtual int32_t load_graph(const char* path) = 0;
  virtual int32_t search(const float* queries, int64_t nq, int32_t k, const eps_search_params* p, int64_t* ids, float* dist,
                         int32_t* counts, int32_t walk_limit = 0) = 0;
  virtual int64_t row_count() const = 0;
  virtual int32_t last_stats(eps_search_stats* out) = 0;
  virtual int kernel_times(double* ms_out, int cap) = 0;
  const char* last_error() const { return err_.c_str(); }
  int32_t last_error_class() const { return err_class_; }
  int32_t fail(int32_t code, const std::string& msg, int32_t err_class = 0) {
    err_ = msg;
    err_class_ = err_class;
    return code;
  }
  std::string err_;
  int32_t err_class_ = 0;   // EPS_ERRCLASS_* of the last failure (eps_index_last_error_class)
};

class Index : public IndexBase {
 public:
  Index(int64_t dim, int metric, int device);
  ~Index();

  int32_t init();
  int32_t set_stream(void* s) override;
  int32_t synchronize() override;
  int32_t attach_rows(const float* rows, int64_t n) override;
  int32_t attach_shard_rows(int32_t shard, const float* rows, int64_t n_local) override {
    if (shard != 0) return fail(EPS_USER_ERROR, "attach_shard_rows: not a sharded index (only shard 0 exists)");
    return attach_rows(rows, n_local);
  }
  int32_t clone_rows(IndexBase& src, int64_t n) override;
  const float* device_rows() const { return d_rows_; }
  // rows of a strided host table: row i at rows + i*pitch_floats (hash-sharded tables: pitch = shards*dim)
  int32_t attach_rows_strided(const float* rows, int64_t n, int64_t pitch_floats, bool copy_device_rows = false);
  int32_t append_rows_strided(const float* rows, int64_t n_new, int64_t pitch_floats);
  int32_t append_rows(const float* rows, int64_t n_new) override;
  int32_t set_id_map(int64_t base, int64_t stride) override;
  int32_t set_deleted(const uint8_t* bits, int64_t nbytes) override;
  int32_t set_int_filter(const void* column, int64_t stride, int32_t width, int32_t op, int64_t constant) override;
  int32_t set_filter_program(const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride, int64_t n_rows, int32_t flags = 0) override;
  int32_t set_filter_program_pitched(const eps_filter_op* ops, int32_t nops, const void* rows, int64_t src_pitch, int64_t row_bytes, int64_t n_rows,
                                     int32_t flags);
  int32_t build(int64_t n, const eps_build_params* p) override;
  int32_t set_graph(int64_t n, const int64_t* off, const int64_t* nbr, int64_t nav) override;
  int32_t graph_info(int64_t* n, int64_t* edges, int64_t* nav) const override;
  int32_t get_graph(int64_t* off, int64_t* nbr) const override;
  int32_t save_graph(const char* path) override;
  int32_t load_graph(const char* path) override;
  int32_t last_stats(eps_search_stats* out) override;
  int32_t load_table(const char* path, const eps_table_layout* layout, int64_t* n_out);
  // walk_limit > 0: candidate-walk form (eps_index_search_walk): k = cap, the tail merge uses searchLimit(walk_limit)
  int32_t search(const float* queries, int64_t nq, int32_t k, const eps_search_params* p, int64_t* ids, float* dist,
                 int32_t* counts, int32_t walk_limit = 0) override;

  int64_t row_count() const override { return n_rows_; }
  const eps_search_stats& stats() const { return stats_; }

  // ---- used by the engine translation units
  int32_t hip_fail(hipError_t e, const char* what);
  FilterSpec filter_spec() const;
  hipStream_t stream() const { return stream_; }

  int64_t dim_;
  int metric_;
  int device_;
  hipStream_t stream_ = nullptr;
  bool own_stream_ = false;

  // vector table mirror: row-major float[n_rows_][dim_] (engine/db/table_segment_mvp.cpp:106-111)
  const float* d_rows_ = nullptr;
  DevBuf rows_buf_;          // owns the storage when rows were attached from host memory
  bool rows_owned_ = false;
  int64_t n_rows_ = 0;
  int64_t rows_version_ = 0;  // bumped on attach/append (invalidates the fp16 mirror)
  // FLAT_AUTO, single-query traffic on a table without a mirror: calls seen on this rows version (flat_mfma_profitable: after a few of
  // them the 8-bit mirror is worth its HBM - the one-pass search answers such a call in a third of the stream scan's time)
  mutable int64_t small_calls_version_ = -1;
  mutable int small_calls_ = 0;
  int64_t scan_limit_ = -1;   // >= 0: flat engines scan rows [0, scan_limit_) only (graph build over a prefix)

  int64_t id_base_ = 0, id_stride_ = 1;

  const uint8_t* d_deleted_ = nullptr;
  DevBuf deleted_buf_;
  const uint8_t* d_fcol_ = nullptr;
  DevBuf fcol_buf_;
  int64_t f_stride_ = 0;
  int32_t f_width_ = 0, f_op_ = 0;
  int64_t f_value_ = 0;
  DevBuf prog_buf_, prog_rows_buf_;   // compiled filter program + (when handed over from the host) the attribute rows
  const uint8_t* d_prog_rows_ = nullptr;
  const void* prog_rows_host_ = nullptr;   // host table the cached device copy mirrors (append-only)
  int64_t prog_rows_stride_ = 0, prog_rows_pitch_ = 0, prog_rows_uploaded_ = 0;
  int64_t loaded_attr_rows_ = 0, loaded_attr_stride_ = 0;   // attribute rows kept by load_table
  int64_t prog_stride_ = 0, prog_rows_n_ = 0;
  int32_t prog_len_ = 0;
  bool prog_uses_dist_ = false;
  int32_t walk_limit_ = 0;            // of the search call in progress
  bool prefilter_call_ = false;

  // graph (reference layout kept on the host for get/save; device form in GraphDev)
  int64_t n_indexed_ = 0;
  int64_t nav_ = 0;
  std::vector<int64_t> h_off_, h_nbr_;
  GraphDev* graph_ = nullptr;

  HalfMirror* mirror_ = nullptr;

  // scratch
  DevBuf q_buf_, partial_buf_, run_buf_, out_buf_, tmp_buf_, page_buf_;   // out_buf_: [ids | distances | counts] of a call with host result pointers
  // r5: that block goes to the host in ONE copy into page-locked memory and is split there (three pageable device-to-host copies staged separately
  // cost a batch of 1024 ~0.1 ms and a single-vector call three trips through the DMA queue)
  struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~HostBuf();
    bool reserve(size_t bytes);
  } h_out_, h_q_;   // (h_q_: host queries are copied to page-locked memory first, then DMA'd: the runtime's own path for pageable sources is slower)
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  // main-kernel event pairs of the last KRING search calls (read back after a run without a sync inside it);
  // evk0_/evk1_ alias the pair of the call in progress
  static constexpr int KRING = 64;
  hipEvent_t kring_[KRING][2] = {};
  bool kring_valid_[KRING] = {};
  int64_t kring_seq_ = 0;
  hipEvent_t evk0_ = nullptr, evk1_ = nullptr;
  // event pairs around EVERY filter-stage launch of the call in progress (eps_search_stats::filter_ms_all: the dominant kernel runs
  // once per stage, the ring above times the largest launch only)
  static constexpr int STAGE_EV = 32;
  hipEvent_t stage_ev_[STAGE_EV][2] = {};
  int stage_n_ = 0;
  int kernel_times(double* ms_out, int cap) override;
  int64_t deleted_bytes_ = 0;   // length of the bitset behind d_deleted_
  int64_t fcol_rows_ = 0;       // rows the attribute column behind d_fcol_ covers

  eps_search_stats stats_{};
  // set by search() around a matrix-engine call: what the engine launches right before its final host sync (the result conversion),
  // so that the device works through the round trip; only called when the batch is one slice
  std::function<void()> pre_sync_;
  bool result_finalized_ = false;   // the conversion has run on the CURRENT contents of the result keys (an engine that rewrites them clears it)
  int64_t pre_sync_nq_ = -1;   // queries of the call that set it (a batch run in slices converts after the last slice instead)
  // ... and where that conversion writes: the matrix engine's last re-rank does it itself (RerankArgs::fin_*) instead of a launch
  int64_t* fin_ids_ = nullptr;
  float* fin_dist_ = nullptr;
  int32_t* fin_cnt_ = nullptr;

 private:
  int32_t flat_stream(const float* dq, int64_t nq, int k, int64_t row_begin, int64_t row_end, u64* run_keys,
                      bool merge_run, int metric = -1, bool filtered = true);
  int32_t flat_stream_page(const float* dq, int64_t nq, int k, int64_t row_begin, int64_t row_end, u64* run_keys, bool merge_run, int metric,
                           bool filtered, const u64* lo, int64_t lo_stride);
  friend int32_t quant8_view(Index&, Quant8View*);
  friend void quant8_queries(Index&, const Quant8View&, const float*, int64_t, signed char*, float*);
  friend int32_t flat_mfma_search(Index&, const float*, int64_t, int, u64*, bool, int);
  friend int32_t flat_mfma_search_slice(Index&, const float*, int64_t, int, u64*, bool, int, int, bool);
  friend int32_t graph_build(Index&, int64_t, const eps_build_params&, const BuildStage*);
  friend int32_t select_edges(Index&, const int64_t*, int64_t, const int64_t*, int32_t, int32_t, int32_t, int64_t*, int32_t*);
  friend int32_t inter_insert(Index&, const int64_t*, const int32_t*, int64_t, int32_t, int64_t*, int32_t*);
  friend int32_t graph_search(Index&, const float*, int64_t, int, const eps_search_params&, u64*, int64_t*, int);
  friend int32_t graph_search_impl(Index&, const float*, int64_t, int, const eps_search_params&, u64*, int64_t*, int, int64_t);
};

// engines implemented in their own translation units
// approx = true: no exact re-rank, the top-k is selected on the fp16 keys (kNN-graph construction), filters ignored
// bits: operand width of the filter pass - 8 (int8 mirror; falls back to 16 where the table does not fit an 8-bit grid or the
// candidate lists overflow), 16 (fp16 mirror), 0 = the library's choice
int32_t flat_mfma_search(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx = false, int bits = 0);
int32_t flat_mfma_search_slice(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx, int cap_scale = 1, int bits = 16,
                               bool auto_bits = false);  // <= 2048 queries
bool flat_mfma_profitable(const Index& ix, int64_t nq, int k);  // AUTO heuristic
int32_t quant8_view(Index& ix, Quant8View* v);
void quant8_queries(Index& ix, const Quant8View& v, const float* dq, int64_t nq, signed char* q8, float* qstat);
void half_mirror_free(HalfMirror* m);
int32_t graph_upload(Index& ix);
void graph_free(GraphDev* g);
// writes result keys [nq][k] and counts
int32_t graph_search(Index& ix, const float* dq, int64_t nq, int k, const eps_search_params& p, u64* run_keys,
                     int64_t* evals, int walk_limit = 0);
int32_t graph_search_impl(Index& ix, const float* dq, int64_t nq, int k, const eps_search_params& p, u64* run_keys, int64_t* evals, int walk_limit,
                          int64_t ecap_min);   // (ecap_min: slots of the filtered traversal's evaluation log, grown on overflow)
// stage-level entry of the build (eps_index_knn_graph / eps_index_link): run a prefix of the stages, on caller-supplied inputs
struct BuildStage {
  const int64_t* knn_in = nullptr;   // [n][K] kNN graph to link (-1 padded) instead of computing one
  int64_t nav_in = -1;               // navigation node to use (-1: the closest row to the centroid)
  int stop_after = 0;                // 1: after the kNN graph (out_ids [n][K]); 2: after Link (out_ids [n][R], out_deg [n]); 0: whole build
  int64_t* out_ids = nullptr;
  int32_t* out_deg = nullptr;
  int64_t* nav_out = nullptr;
};
int32_t graph_build(Index& ix, int64_t n, const eps_build_params& p, const BuildStage* stage = nullptr);
int32_t select_edges(Index& ix, const int64_t* nodes, int64_t m, const int64_t* cands, int32_t cpn, int32_t depth, int32_t R, int64_t* out_ids,
                     int32_t* out_deg);
int32_t inter_insert(Index& ix, const int64_t* ids, const int32_t* deg, int64_t n, int32_t R, int64_t* out_ids, int32_t* out_deg);

bool is_device_ptr(const void* p);

// hash-sharded table over several devices of one process (shard_group.cpp)
IndexBase* make_shard_group(int64_t dim, int metric, const int32_t* devices, int32_t shards, std::string* err);

}  // namespace eps
