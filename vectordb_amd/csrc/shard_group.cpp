// Hash-sharded index over several GPUs of ONE process (north_star: "the corpus is hash-sharded across the
// 8 GPUs of one node with per-shard top-k merged").  The reference DBMS is a single process, so this is the form its
// drop-in executor uses: G per-device indices (one per shard, each with its own stream and scratch), row i of the table
// lives on shard i mod G as local row i / G (global id = local*G + shard via the id map), every shard answers the
// whole query batch on its rows concurrently, the per-shard [nq][k] (id, dist) lists are pushed to shard 0's device over
// xGMI with hipMemcpyPeerAsync (peer-to-peer, no host hop) and merged there by the same k-way merge kernel the
// one-process-per-GPU form runs after its RCCL all-gather (bench.py).  There is no other exchange step.
//
// All eps_index_* entry points work on the handle eps_index_create_sharded returns.  A sharded table is ingested from the DBMS's
// host column (each shard reads its rows with one strided copy, nothing is re-packed) or shard by shard from rows that already
// live on the shard's device (eps_index_attach_shard_rows, r4).  Queries and results are host buffers, or (r4) device buffers on
// any device of the group: a shard on another device gets the queries with one peer copy, and the per-shard lists are gathered
// and merged on the device that holds the caller's result buffers - no host hop, no host sync.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "index.hpp"

namespace eps {

namespace {

// device ordinal of a device / managed pointer, -1 for host memory
int device_of(const void* p) {
  if (!p) return -1;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? a.device : -1;
}

class ShardGroup : public IndexBase {
 public:
  ShardGroup(int64_t dim, int metric) : dim_(dim), metric_(metric) {}
  ~ShardGroup() override {
    {
      std::lock_guard<std::mutex> lk(pool_mu_);
      pool_stop_ = true;
    }
    pool_job_.notify_all();
    for (auto& t : pool_) t.join();
    const int dev0 = merge_dev_ >= 0 ? merge_dev_ : (shard_.empty() ? 0 : shard_[0]->device_);
    free_on(dev0, gathered_);
    free_on(dev0, m_ids_);
    free_on(dev0, m_dist_);
    if (merge_done_) (void)hipEventDestroy(merge_done_);
  }

  int32_t init(const int32_t* devices, int32_t shards, std::string* err) {
    for (int s = 0; s < shards; ++s) {
      std::unique_ptr<Index> ix(new Index(dim_, metric_, devices[s]));
      const int32_t rc = ix->init();
      if (rc != EPS_OK) {
        *err = std::string("shard ") + std::to_string(s) + ": " + ix->last_error();
        return rc;
      }
      ix->set_id_map(s, shards);
      shard_.push_back(std::move(ix));
    }
    // peer access between every pair of the group's devices (the merge runs where the caller's result buffers live); failure is
    // not fatal (hipMemcpyPeerAsync then stages through the host)
    for (int a = 0; a < shards; ++a)
      for (int b = 0; b < shards; ++b)
        if (devices[a] != devices[b]) {
          (void)hipSetDevice(devices[a]);
          (void)hipDeviceEnablePeerAccess(devices[b], 0);
          (void)hipGetLastError();
        }
    shard_rows_.assign((size_t)shards, 0);
    // one worker thread per shard for the lifetime of the group (r2 spawned and joined G threads per call, search included)
    for (int s = 0; s < shards; ++s) pool_.emplace_back([this, s]() { worker(s); });
    return EPS_OK;
  }

  int G() const { return (int)shard_.size(); }
  int64_t rows_of(int s, int64_t n) const { return n > s ? (n - s + G() - 1) / G() : 0; }

  void worker(int s) {
    (void)hipSetDevice(shard_[s]->device_);
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(pool_mu_);
    for (;;) {
      pool_job_.wait(lk, [&] { return pool_stop_ || pool_gen_ != seen; });
      if (pool_stop_) return;
      seen = pool_gen_;
      lk.unlock();
      int32_t rc;
      try {
        rc = pool_fn_(s, *shard_[s]);
      } catch (const std::exception& e) {
        rc = shard_[s]->fail(EPS_DB_UNEXPECTED_ERROR, e.what());
      } catch (...) {
        rc = shard_[s]->fail(EPS_DB_UNEXPECTED_ERROR, "unexpected exception");
      }
      lk.lock();
      pool_rc_[s] = rc;
      if (--pool_pending_ == 0) pool_done_.notify_all();
    }
  }

  template <class F>
  int32_t each(F&& f) {   // f(s, Index&) on every shard, concurrently, on the group's worker threads; first error wins
    {
      std::unique_lock<std::mutex> lk(pool_mu_);
      pool_fn_ = std::function<int32_t(int, Index&)>(std::forward<F>(f));
      pool_rc_.assign((size_t)G(), EPS_OK);
      pool_pending_ = G();
      ++pool_gen_;
      pool_job_.notify_all();
      pool_done_.wait(lk, [&] { return pool_pending_ == 0; });
      pool_fn_ = nullptr;
    }
    for (int s = 0; s < G(); ++s)
      if (pool_rc_[s] != EPS_OK) return fail(pool_rc_[s], "shard " + std::to_string(s) + ": " + shard_[s]->last_error(), shard_[s]->last_error_class());
    return EPS_OK;
  }

  int32_t set_stream(void*) override { return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: every shard runs on its own stream"); }
  int32_t synchronize() override {
    return each([](int, Index& ix) { return ix.synchronize(); });
  }
  int32_t attach_rows(const float* rows, int64_t n) override {
    if (n < 0 || (n > 0 && !rows)) return fail(EPS_USER_ERROR, "attach_rows: bad arguments");
    if (is_device_ptr(rows)) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: rows are ingested from host memory");
    const int32_t rc = each([&](int s, Index& ix) { return ix.attach_rows_strided(rows + (int64_t)s * dim_, rows_of(s, n), (int64_t)G() * dim_); });
    if (rc == EPS_OK) {
      n_rows_ = n;
      for (int s = 0; s < G(); ++s) shard_rows_[(size_t)s] = rows_of(s, n);
    }
    return rc;
  }
  // rows of ONE shard (local row l = global row l * G + shard), host memory or memory of the shard's own device (used in place)
  int32_t attach_shard_rows(int32_t shard, const float* rows, int64_t n_local) override {
    if (shard < 0 || shard >= G() || n_local < 0 || (n_local > 0 && !rows)) return fail(EPS_USER_ERROR, "attach_shard_rows: bad arguments");
    const int dv = device_of(rows);
    if (dv >= 0 && dv != shard_[(size_t)shard]->device_)
      return fail(EPS_USER_ERROR, "attach_shard_rows: the rows live on device " + std::to_string(dv) + ", shard " + std::to_string(shard) + " on device " +
                                      std::to_string(shard_[(size_t)shard]->device_));
    if (hipSetDevice(shard_[(size_t)shard]->device_) != hipSuccess) return fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    const int32_t rc = shard_[(size_t)shard]->attach_rows(rows, n_local);
    if (rc != EPS_OK) return fail(rc, "shard " + std::to_string(shard) + ": " + shard_[(size_t)shard]->last_error(), shard_[(size_t)shard]->last_error_class());
    shard_rows_[(size_t)shard] = n_local;
    n_rows_ = 0;
    for (int64_t v : shard_rows_) n_rows_ += v;
    return EPS_OK;
  }
  int32_t clone_rows(IndexBase& src_base, int64_t n) override {
    ShardGroup* src = dynamic_cast<ShardGroup*>(&src_base);
    if (!src || src == this || src->G() != G() || src->dim_ != dim_) return fail(EPS_USER_ERROR, "clone_rows: the source must be another shard group of the same shape");
    for (int s = 0; s < G(); ++s)
      if (src->shard_[(size_t)s]->device_ != shard_[(size_t)s]->device_) return fail(EPS_USER_ERROR, "clone_rows: the groups' shards live on different devices");
    if (n < 0 || n > src->n_rows_ || !src->split_ok()) return fail(EPS_USER_ERROR, "clone_rows: n exceeds the source's rows");
    const int32_t rc = each([&](int s, Index& ix) { return ix.clone_rows(*src->shard_[(size_t)s], rows_of(s, n)); });
    if (rc == EPS_OK) {
      n_rows_ = n;
      for (int s = 0; s < G(); ++s) shard_rows_[(size_t)s] = rows_of(s, n);
    }
    return rc;
  }
  bool split_ok() const {   // the shards hold a hash split of n_rows_ rows (attach_shard_rows fills them one at a time)
    for (int s = 0; s < G(); ++s)
      if (shard_rows_[(size_t)s] != rows_of(s, n_rows_)) return false;
    return true;
  }
  int32_t append_rows(const float* rows, int64_t n_new) override {
    if (n_new < 0 || (n_new > 0 && !rows)) return fail(EPS_USER_ERROR, "append_rows: bad arguments");
    if (is_device_ptr(rows)) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: rows are ingested from host memory");
    // global row n_rows_ + j goes to shard (n_rows_ + j) mod G
    const int64_t n0 = n_rows_;
    const int32_t rc = each([&](int s, Index& ix) {
      const int64_t first = ((s - n0) % G() + G()) % G();   // first j with (n0 + j) mod G == s
      const int64_t cnt = n_new > first ? (n_new - first + G() - 1) / G() : 0;
      return cnt ? ix.append_rows_strided(rows + first * dim_, cnt, (int64_t)G() * dim_) : EPS_OK;
    });
    if (rc == EPS_OK) {
      n_rows_ += n_new;
      for (int s = 0; s < G(); ++s) shard_rows_[(size_t)s] = rows_of(s, n_rows_);
    }
    return rc;
  }
  int32_t set_id_map(int64_t, int64_t) override { return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: the id map is the sharding itself"); }
  int32_t set_deleted(const uint8_t* bits, int64_t nbytes) override {
    if (!bits || nbytes <= 0) return each([](int, Index& ix) { return ix.set_deleted(nullptr, 0); });
    if (is_device_ptr(bits)) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: the deleted bitset comes from host memory");
    if (nbytes < (n_rows_ + 7) / 8) return fail(EPS_USER_ERROR, "set_deleted: bitset shorter than ceil(rows/8) bytes");
    return each([&](int s, Index& ix) {   // bit i of the table = bit i / G of shard i mod G
      const int64_t ns = rows_of(s, n_rows_);
      std::vector<uint8_t> local((size_t)(ns + 7) / 8, 0);
      for (int64_t l = 0; l < ns; ++l) {
        const int64_t i = l * G() + s;
        if ((bits[i >> 3] >> (i & 7)) & 1) local[l >> 3] |= uint8_t(1u << (l & 7));
      }
      return ix.set_deleted(local.empty() ? nullptr : local.data(), (int64_t)local.size());
    });
  }
  int32_t set_int_filter(const void* column, int64_t stride, int32_t width, int32_t op, int64_t constant) override {
    if (op == EPS_OP_NONE || !column) return each([](int, Index& ix) { return ix.set_int_filter(nullptr, 0, 0, EPS_OP_NONE, 0); });
    if (is_device_ptr(column)) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: filter columns come from host memory");
    return each([&](int s, Index& ix) {   // the shard's rows are every G-th row of the column: same memory, G x the stride
      return ix.set_int_filter(static_cast<const char*>(column) + (int64_t)s * stride, stride * G(), width, op, constant);
    });
  }
  int32_t set_filter_program(const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride, int64_t n_rows, int32_t flags) override {
    if (nops <= 0 || !ops) return each([](int, Index& ix) { return ix.set_filter_program(nullptr, 0, nullptr, 0, 0); });
    if (!rows || is_device_ptr(rows)) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: attribute rows come from host memory");
    if (stride <= 0 || n_rows < n_rows_) return fail(EPS_USER_ERROR, "set_filter_program: attribute rows missing or shorter than the table");
    return each([&](int s, Index& ix) {   // the shard's rows are every G-th row: ONLY those are uploaded (packed at the original stride)
      return ix.set_filter_program_pitched(ops, nops, static_cast<const char*>(rows) + (int64_t)s * stride, stride * G(), stride, rows_of(s, n_rows), flags);
    });
  }
  int32_t build(int64_t n, const eps_build_params* p) override {   // every shard builds the graph of its own rows
    if (n < 0 || n > n_rows_) return fail(EPS_USER_ERROR, "build: n exceeds the attached rows");
    if (!split_ok()) return fail(EPS_USER_ERROR, "build: the shards' row counts are not a hash split of the table (attach_shard_rows on every shard first)");
    return each([&](int s, Index& ix) { return ix.build(rows_of(s, n), p); });
  }
  int32_t set_graph(int64_t, const int64_t*, const int64_t*, int64_t) override {
    return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: a graph over the whole table cannot be split; build per shard (eps_index_build) or load per-shard files");
  }
  int32_t graph_info(int64_t* n, int64_t* edges, int64_t* nav) const override {
    int64_t tn = 0, te = 0;
    for (auto& ix : shard_) {
      int64_t a = 0, b = 0, c = 0;
      ix->graph_info(&a, &b, &c);
      tn += a;
      te += b;
    }
    if (n) *n = tn;
    if (edges) *edges = te;
    if (nav) *nav = -1;
    return EPS_OK;
  }
  int32_t get_graph(int64_t*, int64_t*) const override {
    return const_cast<ShardGroup*>(this)->fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: there is one graph per shard; save them with eps_index_save_graph (<path>.shard<s>)");
  }
  int32_t save_graph(const char* path) override {   // <path>.shard<s>, each in the reference's ann_graph file format
    if (!path) return fail(EPS_USER_ERROR, "save_graph: null path");
    return each([&](int s, Index& ix) { return ix.save_graph((std::string(path) + ".shard" + std::to_string(s)).c_str()); });
  }
  int32_t load_graph(const char* path) override {
    if (!path) return fail(EPS_USER_ERROR, "load_graph: null path");
    return each([&](int s, Index& ix) { return ix.load_graph((std::string(path) + ".shard" + std::to_string(s)).c_str()); });
  }

  int32_t search(const float* queries, int64_t nq, int32_t k, const eps_search_params* pp, int64_t* ids, float* dist, int32_t* counts,
                 int32_t walk_limit) override {
    if (nq < 0 || k <= 0) return fail(EPS_USER_ERROR, "search: nq must be >= 0 and k > 0");
    if (nq == 0) return EPS_OK;
    if (!queries || !ids || !dist) return fail(EPS_USER_ERROR, "search: null buffer");
    if (walk_limit) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: candidate walks are per index");
    if (!split_ok()) return fail(EPS_USER_ERROR, "search: the shards' row counts are not a hash split of the table (attach_shard_rows on every shard first)");
    // where the buffers live: host, or a device of the group (r4).  The merge runs on the device that holds the result buffers
    // (shard 0's for host results); ids / dist / counts must live together.
    const int q_dev = device_of(queries), r_dev = device_of(ids);
    if (device_of(dist) != r_dev || (counts && device_of(counts) != r_dev)) return fail(EPS_USER_ERROR, "search: ids, dist and counts must live in the same memory");
    int ms = 0;   // the shard whose device (and stream) merges
    if (r_dev >= 0) {
      ms = -1;
      for (int s = 0; s < G(); ++s)
        if (shard_[(size_t)s]->device_ == r_dev) { ms = s; break; }
      if (ms < 0) return fail(EPS_DB_UNSUPPORTED_ERROR, "sharded index: device result buffers must live on one of the group's devices");
    }
    const int mdev = shard_[(size_t)ms]->device_;
    const size_t nk = (size_t)nq * k;
    // per-shard results stay on the shard's device; they are pushed to the merging device and merged there
    if (nk > cap_ || mdev != merge_dev_) {
      const int old = merge_dev_ >= 0 ? merge_dev_ : mdev;
      free_on(old, m_ids_);
      free_on(old, m_dist_);
      free_on(old, gathered_);
      cap_ = 0;
      if (hipSetDevice(mdev) != hipSuccess) return fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
      // one gathered buffer: [G][ids int64[nk] | dist f32[nk]] (the layout eps_merge_topk_packed takes)
      stride_ = (nk * 12 + 7) / 8 * 8;
      if (hipMalloc(&gathered_, stride_ * G()) != hipSuccess || hipMalloc(&m_ids_, nk * 8) != hipSuccess || hipMalloc(&m_dist_, nk * 4) != hipSuccess)
        return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (shard merge)");
      cap_ = nk;
      merge_dev_ = mdev;
      merge_pending_ = false;   // (hipFree waited for everything that read the old buffer)
      if (merge_done_) {        // (an event belongs to the device it was created on)
        (void)hipEventDestroy(merge_done_);
        merge_done_ = nullptr;
      }
    }
    stride_ = (nk * 12 + 7) / 8 * 8;   // (the gathered layout of THIS call; the buffer holds at least cap_ entries per shard)
    local_ids_.resize((size_t)G());
    local_dist_.resize((size_t)G());
    local_cnt_.resize((size_t)G());
    local_q_.resize((size_t)G());
    const size_t qbytes = (size_t)nq * dim_ * sizeof(float);
    int32_t rc = each([&](int s, Index& ix) -> int32_t {
      if (hipSetDevice(ix.device_) != hipSuccess) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
      if (!local_ids_[s].reserve(nk * 8) || !local_dist_[s].reserve(nk * 4) || !local_cnt_[s].reserve((size_t)nq * 4))
        return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
      const float* q = queries;
      if (q_dev >= 0 && q_dev != ix.device_) {   // device queries on another GPU: one peer copy (3 MB at batch 1024 x 768)
        if (!local_q_[s].reserve(qbytes)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
        const hipError_t eq = hipMemcpyPeerAsync(local_q_[s].p, ix.device_, queries, q_dev, qbytes, ix.stream_);
        if (eq != hipSuccess) return ix.hip_fail(eq, "peer copy of the queries");
        q = local_q_[s].as<float>();
      }
      int32_t r = ix.search(q, nq, k, pp, local_ids_[s].as<int64_t>(), local_dist_[s].as<float>(), local_cnt_[s].as<int32_t>());
      if (r != EPS_OK) return r;
      // the one exchange step: this shard's [nq][k] lists -> the merging device, peer to peer over xGMI
      char* dst = static_cast<char*>(gathered_) + (size_t)s * stride_;
      // (a merge with device results returns without a host sync: the previous call's merge may still be READING gathered_ on the merging
      // shard's stream - this shard's copy into it waits for that merge's event)
      if (merge_pending_) {
        const hipError_t ew = hipStreamWaitEvent(ix.stream_, merge_done_, 0);
        if (ew != hipSuccess) return ix.hip_fail(ew, "wait for the previous shard merge");
      }
      hipError_t e = hipMemcpyPeerAsync(dst, mdev, local_ids_[s].p, ix.device_, nk * 8, ix.stream_);
      if (e == hipSuccess) e = hipMemcpyPeerAsync(dst + nk * 8, mdev, local_dist_[s].p, ix.device_, nk * 4, ix.stream_);
      if (e == hipSuccess) e = hipStreamSynchronize(ix.stream_);
      return e == hipSuccess ? EPS_OK : ix.hip_fail(e, "peer copy of the shard's top-k");
    });
    if (rc != EPS_OK) return rc;
    if (hipSetDevice(mdev) != hipSuccess) return fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    hipStream_t s0 = shard_[(size_t)ms]->stream_;
    const float* gd = reinterpret_cast<const float*>(static_cast<char*>(gathered_) + nk * 8);
    if (r_dev >= 0) {   // device results: merged straight into the caller's buffers; the caller synchronises (eps_index_synchronize)
      launch_merge_shards(gd, static_cast<const int64_t*>(gathered_), G(), nq, k, dist, ids, s0, (int64_t)stride_, counts);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess && !merge_done_) e = hipEventCreateWithFlags(&merge_done_, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(merge_done_, s0);
      if (e != hipSuccess) return fail(EPS_INFRA_UNEXPECTED_ERROR, std::string("shard merge: ") + hipGetErrorString(e));
      merge_pending_ = true;
      return EPS_OK;
    }
    launch_merge_shards(gd, static_cast<const int64_t*>(gathered_), G(), nq, k, static_cast<float*>(m_dist_), static_cast<int64_t*>(m_ids_), s0, (int64_t)stride_);
    hipError_t e = hipMemcpyAsync(ids, m_ids_, nk * 8, hipMemcpyDeviceToHost, s0);
    if (e == hipSuccess) e = hipMemcpyAsync(dist, m_dist_, nk * 4, hipMemcpyDeviceToHost, s0);
    if (e == hipSuccess) e = hipStreamSynchronize(s0);
    if (e != hipSuccess) return fail(EPS_INFRA_UNEXPECTED_ERROR, std::string("shard merge: ") + hipGetErrorString(e));
    merge_pending_ = false;
    if (counts)
      for (int64_t q = 0; q < nq; ++q) {
        int32_t c = 0;
        while (c < k && ids[q * k + c] >= 0) ++c;
        counts[q] = c;
      }
    return EPS_OK;
  }

  int64_t row_count() const override { return n_rows_; }
  int32_t last_stats(eps_search_stats* out) override {   // counters summed over the shards, times = the slowest shard
    eps_search_stats t;
    std::memset(&t, 0, sizeof(t));
    for (auto& ix : shard_) {
      eps_search_stats s;
      ix->last_stats(&s);
      t.dist_evals += s.dist_evals;
      t.expansions += s.expansions;
      t.rerank_rows += s.rerank_rows;
      t.overflow_queries += s.overflow_queries;
      t.kernel_ms = std::max(t.kernel_ms, s.kernel_ms);
      t.main_kernel_ms = std::max(t.main_kernel_ms, s.main_kernel_ms);
      t.main_kernel_launches += s.main_kernel_launches;
      t.main_kernel_rows += s.main_kernel_rows;
      t.main_kernel_queries = std::max(t.main_kernel_queries, s.main_kernel_queries);
      t.main_kernel_bits = std::max(t.main_kernel_bits, s.main_kernel_bits);
      t.i8_declined += s.i8_declined;
      t.i8_folded += s.i8_folded;
      t.one_pass += s.one_pass;
      t.filter_ms_all = std::max(t.filter_ms_all, s.filter_ms_all);
      t.filter_rows_all += s.filter_rows_all;
    }
    *out = t;
    return EPS_OK;
  }
  int kernel_times(double* ms_out, int cap) override {   // per call the slowest shard, as last_stats reports
    if (!ms_out || cap <= 0) return 0;
    int n = -1;
    std::vector<double> t((size_t)cap);
    for (auto& ix : shard_) {
      const int c = ix->kernel_times(t.data(), cap);
      if (n < 0) {
        n = c;
        std::copy(t.begin(), t.begin() + c, ms_out);
      } else {
        n = std::min(n, c);   // (aligned at the most recent call)
        for (int i = 0; i < n; ++i) ms_out[i] = std::max(ms_out[i], t[(size_t)i]);
      }
    }
    return n < 0 ? 0 : n;
  }

 private:
  static void free_on(int dev, void*& p) {
    if (!p) return;
    (void)hipSetDevice(dev);
    (void)hipFree(p);
    p = nullptr;
  }
  int64_t dim_;
  int metric_;
  int64_t n_rows_ = 0;
  std::vector<std::unique_ptr<Index>> shard_;
  std::vector<int64_t> shard_rows_;   // rows every shard holds (attach_rows: the hash split; attach_shard_rows: as handed over)
  std::vector<DevBuf> local_ids_, local_dist_, local_cnt_, local_q_;
  int merge_dev_ = -1;                // device gathered_ / m_ids_ / m_dist_ live on
  void* gathered_ = nullptr;
  void* m_ids_ = nullptr;
  void* m_dist_ = nullptr;
  size_t cap_ = 0, stride_ = 0;
  hipEvent_t merge_done_ = nullptr;   // recorded behind a merge that returned without a host sync (device result buffers)
  bool merge_pending_ = false;
  // worker pool
  std::vector<std::thread> pool_;
  std::mutex pool_mu_;
  std::condition_variable pool_job_, pool_done_;
  std::function<int32_t(int, Index&)> pool_fn_;
  std::vector<int32_t> pool_rc_;
  uint64_t pool_gen_ = 0;
  int pool_pending_ = 0;
  bool pool_stop_ = false;
};

}  // namespace

IndexBase* make_shard_group(int64_t dim, int metric, const int32_t* devices, int32_t shards, std::string* err) {
  std::unique_ptr<ShardGroup> g(new ShardGroup(dim, metric));
  if (g->init(devices, shards, err) != EPS_OK) return nullptr;
  return g.release();
}

}  // namespace eps
