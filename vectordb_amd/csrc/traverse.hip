// Graph traversal path of libepsilla_gfx950: the device form of VecSearchExecutor::SearchImpl and the
// graph branch of VecSearchExecutor::Search (reference: engine/db/execution/vec_search_executor.cpp:518-715,
// 873-927; algorithm spec SURVEY.md Appendix A.2/A.4).
//
// One workgroup (4 wavefronts) per query, the whole batch in flight at once:
//   * master queue of L candidates sorted by (dist,id) in LDS, `checked` flag in bit 0 of the key
//     (Candidate, candidate.hpp:7-23; set_L_, vec_search_executor.hpp:55);
//   * visited set = per-query bitmap in HBM, test-and-set with one atomicOr per neighbour
//     (is_visited_, :403-406);
//   * a round expands the first M unchecked candidates at once: CSR rows gathered, visited-filtered and
//     compacted through LDS, surviving rows streamed with 16 B/lane loads by all four wavefronts
//     (ExpandOneCandidate, :384-444), the survivors rank-sorted and merged into the queue in place
//     (AddIntoQueue :75-117 / MergeTwoQueues :150-217 become one parallel merge).
// With M = 1 (IntraQueryThreads = 1) the sequence of expansions and the final queue are exactly the
// reference's single-thread result (see DESIGN.md "traversal equivalence"); M > 1 plays the role of the
// reference's T workers, whose interleaving is not deterministic in the reference either.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "index.hpp"
#include "traverse_kernel.hpp"

namespace eps {

struct GraphDev {
  DevBuf off;        // int64 [n+1]            (CSR form, used when max_degree > 64)
  DevBuf nbr;        // u32 [E]                (CSR) or u32 [n][fixed_deg] padded with 0xFFFFFFFF
  int fixed_deg = 0; // > 0: fixed-stride adjacency — one dependent load less per expansion than CSR
  DevBuf init_ids;   // u32 [L]
  int64_t init_L = -1;
  int64_t max_degree = 0;
  DevBuf visited;    // u32 [nq_chunk][words]
  DevBuf queue;      // u64 [nq][L]
  DevBuf counters;   // unsigned long long [2]
  DevBuf tail;       // u64 [nq][k] brute-force tail lists
};

void graph_free(GraphDev* g) { delete g; }

int32_t graph_upload(Index& ix) {
  if (!ix.graph_) ix.graph_ = new GraphDev();
  GraphDev& g = *ix.graph_;
  g.init_L = -1;
  const int64_t n = ix.n_indexed_;
  if (n <= 0) return EPS_OK;
  const int64_t e = ix.h_off_[n];
  int64_t maxdeg = 0;
  for (int64_t i = 0; i < n; ++i) maxdeg = std::max(maxdeg, ix.h_off_[i + 1] - ix.h_off_[i]);
  g.max_degree = maxdeg;
  hipError_t er = hipSuccess;
  if (maxdeg > 0 && maxdeg <= 64) {
    // NSG out-degree is <= 50 (+ connectivity repair): store every list at a fixed stride so the traversal
    // reads neighbours of node v at nbr[v*deg ..] without first fetching offsets ("one coalesced
    // CSR row read each")
    const int deg = (int)((maxdeg + 3) / 4 * 4);
    std::vector<u32> pad((size_t)n * deg, 0xFFFFFFFFu);
    for (int64_t i = 0; i < n; ++i) {
      const int64_t b = ix.h_off_[i], c = ix.h_off_[i + 1] - b;
      for (int64_t j = 0; j < c; ++j) pad[(size_t)i * deg + j] = (u32)ix.h_nbr_[b + j];
    }
    if (!g.nbr.reserve(pad.size() * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "set_graph: out of device memory");
    er = hipMemcpyAsync(g.nbr.p, pad.data(), pad.size() * 4, hipMemcpyHostToDevice, ix.stream_);
    if (er == hipSuccess) er = hipStreamSynchronize(ix.stream_);
    g.fixed_deg = deg;
  } else {
    std::vector<u32> nb32((size_t)e);
    for (int64_t i = 0; i < e; ++i) nb32[i] = (u32)ix.h_nbr_[i];
    if (!g.off.reserve((size_t)(n + 1) * 8) || !g.nbr.reserve((size_t)(e > 0 ? e : 1) * 4))
      return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "set_graph: out of device memory");
    er = hipMemcpyAsync(g.off.p, ix.h_off_.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ix.stream_);
    if (er == hipSuccess && e > 0) er = hipMemcpyAsync(g.nbr.p, nb32.data(), (size_t)e * 4, hipMemcpyHostToDevice, ix.stream_);
    if (er == hipSuccess) er = hipStreamSynchronize(ix.stream_);
    g.fixed_deg = 0;
  }
  if (er != hipSuccess) return ix.hip_fail(er, "set_graph upload");
  return EPS_OK;
}

// PrepareInitIds (vec_search_executor.cpp:487-516): distinct CSR neighbours of nav, then nav+1, nav+2, ...
// wrapping.  The reference requires n >= L (its fill loop never ends otherwise); callers clamp L to n.
static void prepare_init_ids(const Index& ix, int64_t L, std::vector<u32>& out) {
  const int64_t n = ix.n_indexed_;
  std::vector<uint8_t> sel((size_t)n, 0);
  out.clear();
  out.reserve((size_t)L);
  for (int64_t e = ix.h_off_[ix.nav_]; e < ix.h_off_[ix.nav_ + 1] && (int64_t)out.size() < L; ++e) {
    const int64_t v = ix.h_nbr_[e];
    if (sel[v]) continue;
    sel[v] = 1;
    out.push_back((u32)v);
  }
  int64_t tmp = ix.nav_ + 1;
  while ((int64_t)out.size() < L) {
    if (tmp == n) tmp = 0;
    const int64_t v = tmp++;
    if (sel[v]) continue;
    sel[v] = 1;
    out.push_back((u32)v);
  }
}

// ------------------------------------------------------------------------------------------------
// Search() epilogue for one query (vec_search_executor.cpp:873-927): optional merge of the brute-force tail
// into the first K slots (MergeTwoQueuesInto1stQueueSeqFixed, :150-217 — note only K slots take part),
// then the post-filter walk over the first cand_num queue entries until K results.
struct PostArgs {
  const u64* queue;   // [nq][L] traversal keys (id<<1|checked)
  int L;
  const u64* tail;    // [nq][k] flat-scan keys of the un-indexed tail (plain id), or null
  int k;              // output width (= limit)
  int K;              // searchLimit = min(n_indexed, limit, L_local)
  int cand_num_tail;  // min(L_master, n_total)
  int cand_num;       // min(L_master, n_indexed)
  FilterSpec f;
  u64* run_keys;      // [nq][k] output keys (plain id)
};

__device__ __forceinline__ bool ckey_less(u64 a, u64 b) { return a < b; }  // plain (dist,id) keys

__global__ __launch_bounds__(64) void post_kernel(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 m[];  // [L] master as plain keys
  const int64_t q = blockIdx.x;
  const int L = a.L;
  for (int i = threadIdx.x; i < L; i += 64) {
    const u64 v = a.queue[q * L + i];
    m[i] = v == KEY_EMPTY ? KEY_EMPTY : ((v & 0xFFFFFFFF00000000ull) | ((v & 0xFFFFFFFFull) >> 1));
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  int cand_num = a.cand_num;
  if (a.tail) {
    const u64* bf = a.tail + q * a.k;
    int n2 = 0;
    while (n2 < a.k && bf[n2] != KEY_EMPTY) ++n2;
    if (n2 > 0) {
      const int n1 = a.K;
      // lower_bound of bf[0] in m[0..n1)
      int lo = 0, hi = n1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ckey_less(m[mid], bf[0])) lo = mid + 1; else hi = mid;
      }
      const int idx = lo;
      if (idx == n1) {
        // nothing merges
      } else if (idx == n1 - 1) {
        m[idx] = bf[0];
      } else {
        if (key_id(bf[0]) != key_id(m[idx])) {
          for (int j = n1 - 1; j > idx; --j) m[j] = m[j - 1];
          m[idx] = bf[0];
        }
        int i1 = idx + 1, i2 = 1;
        for (int ins = idx + 1; ins < n1; ++ins) {
          if (i1 >= n1 || i2 >= n2) break;
          if (ckey_less(m[i1], bf[i2])) {
            ++i1;
          } else if (ckey_less(bf[i2], m[i1])) {
            for (int j = n1 - 1; j > ins; --j) m[j] = m[j - 1];
            m[ins] = bf[i2++];
            ++i1;
          } else {
            ++i2;
            ++i1;
          }
        }
      }
      cand_num = a.cand_num_tail;
    }
  }
  int res = 0;
  for (int i = 0; i < cand_num && res < a.K; ++i) {
    const u64 v = m[i];
    if (v == KEY_EMPTY) continue;
    if (!row_visible(a.f, key_id(v))) continue;
    a.run_keys[q * a.k + res] = v;
    ++res;
  }
  for (; res < a.k; ++res) a.run_keys[q * a.k + res] = KEY_EMPTY;
}

// ------------------------------------------------------------------------------------------------ host
int32_t graph_search(Index& ix, const float* dq, int64_t nq, int k, const eps_search_params& p, u64* run_keys,
                     int64_t* evals_out) {
  GraphDev& g = *ix.graph_;
  const int64_t n = ix.n_indexed_;
  int64_t L = p.master_queue;
  if (L > n) L = n;  // the reference would spin forever in PrepareInitIds when L > n (see prepare_init_ids)
  if (L > 4096) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: SearchQueueSize > 4096 is not supported by the LDS-resident queue (use the flat engines)");
  int M = p.intra_threads;
  if (M > TRV_MAXM) M = TRV_MAXM;
  int Lp2 = 1;
  while (Lp2 < L) Lp2 <<= 1;
  hipStream_t s = ix.stream_;
  hipError_t er;
  if (g.init_L != L) {
    std::vector<u32> init;
    prepare_init_ids(ix, L, init);
    if (!g.init_ids.reserve((size_t)L * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
    er = hipMemcpyAsync(g.init_ids.p, init.data(), (size_t)L * 4, hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (er != hipSuccess) return ix.hip_fail(er, "init ids upload");
    g.init_L = L;
  }
  const int64_t words = (n + 31) / 32;
  // visited bitmaps: process the batch in slices so the scratch stays below ~2 GiB
  int64_t slice = nq;
  const int64_t max_bytes = (int64_t)2 << 30;
  if (slice * words * 4 > max_bytes) slice = std::max<int64_t>(1, max_bytes / (words * 4));
  if (!g.visited.reserve((size_t)slice * words * 4) || !g.queue.reserve((size_t)nq * L * 8) || !g.counters.reserve(16))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (traversal scratch)");
  er = hipMemsetAsync(g.counters.p, 0, 16, s);
  if (er != hipSuccess) return ix.hip_fail(er, "memset");

  const bool vec4 = (ix.dim_ % 4 == 0) && ((reinterpret_cast<uintptr_t>(ix.d_rows_) & 15) == 0);
  const int qstride = ((int)ix.dim_ + 3) & ~3;
  (void)qstride;
  const size_t shm = traverse_lds_bytes((int)ix.dim_, Lp2, false);
  TraverseArgs a;
  a.rows = ix.d_rows_;
  a.dim = (int)ix.dim_;
  a.metric = ix.metric_;
  a.off = g.fixed_deg > 0 ? nullptr : g.off.as<int64_t>();
  a.nbr = g.nbr.as<u32>();
  a.fixed_deg = g.fixed_deg;
  a.log = nullptr;
  a.log_cnt = nullptr;
  a.log_cap = 0;
  a.init_ids = g.init_ids.as<u32>();
  a.L = (int)L;
  a.Lp2 = Lp2;
  a.M = M;
  a.visited = g.visited.as<u32>();
  a.words = words;
  a.counters = g.counters.as<unsigned long long>();
  (void)hipEventRecord(ix.evk0_, s);
  for (int64_t q0 = 0; q0 < nq; q0 += slice) {
    const int64_t cnt = std::min(slice, nq - q0);
    er = hipMemsetAsync(g.visited.p, 0, (size_t)cnt * words * 4, s);  // is_visited.clear()/resize(n), :711-714
    if (er != hipSuccess) return ix.hip_fail(er, "memset visited");
    a.queries = dq + q0 * ix.dim_;
    a.out_queue = g.queue.as<u64>() + q0 * L;
    // few queries: 16 wavefronts per query (latency); many queries: 4 per query (throughput, more queries per CU)
    static const int wide_env = getenv("EPS_TRV_WIDE") ? atoi(getenv("EPS_TRV_WIDE")) : -1;
    const bool wide = wide_env >= 0 ? wide_env != 0 : nq <= 256;
    if (vec4 && wide)
      hipLaunchKernelGGL((traverse_kernel<true, false, false, 16>), dim3((unsigned)cnt), dim3(1024), shm, s, a);
    else if (vec4)
      hipLaunchKernelGGL((traverse_kernel<true, false, false, 4>), dim3((unsigned)cnt), dim3(256), shm, s, a);
    else if (wide)
      hipLaunchKernelGGL((traverse_kernel<false, false, false, 16>), dim3((unsigned)cnt), dim3(1024), shm, s, a);
    else
      hipLaunchKernelGGL((traverse_kernel<false, false, false, 4>), dim3((unsigned)cnt), dim3(256), shm, s, a);
    ix.stats_.main_kernel_launches += 1;
  }
  (void)hipEventRecord(ix.evk1_, s);

  // brute-force tail over the rows the graph does not cover yet (:885-900)
  const int64_t n_total = ix.n_rows_;
  const u64* tail = nullptr;
  if (n_total > n) {
    if (!g.tail.reserve((size_t)nq * k * 8)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
    int32_t rc = ix.flat_stream(dq, nq, k, n, n_total, g.tail.as<u64>(), false);
    if (rc != EPS_OK) return rc;
    tail = g.tail.as<u64>();
  }
  PostArgs pa;
  pa.queue = g.queue.as<u64>();
  pa.L = (int)L;
  pa.tail = tail;
  pa.k = k;
  int64_t K = n;
  if (k < K) K = k;
  if (p.local_queue < K) K = p.local_queue;
  if (L < K) K = L;
  pa.K = (int)K;
  pa.cand_num_tail = (int)std::min<int64_t>(L, n_total);
  pa.cand_num = (int)std::min<int64_t>(L, n);
  pa.f = ix.filter_spec();
  pa.run_keys = run_keys;
  hipLaunchKernelGGL(post_kernel, dim3((unsigned)nq), dim3(64), (size_t)L * 8, s, pa);
  er = hipGetLastError();
  if (er != hipSuccess) return ix.hip_fail(er, "traversal launch");
  unsigned long long h[2] = {0, 0};
  er = hipMemcpyAsync(h, g.counters.p, 16, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return ix.hip_fail(er, "traversal");
  ix.stats_.dist_evals += (int64_t)h[0];
  ix.stats_.expansions += (int64_t)h[1];
  if (evals_out) *evals_out = (int64_t)h[0];
  return EPS_OK;
}

}  // namespace eps
