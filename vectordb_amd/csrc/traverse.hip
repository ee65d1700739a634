// Graph traversal path of libepsilla_gfx950: the device form of VecSearchExecutor::SearchImpl and the
// graph branch of VecSearchExecutor::Search (reference: engine/db/execution/vec_search_executor.cpp:518-715,
// 873-927; algorithm spec SURVEY.md Appendix A.2/A.4).
//
// The kernel is traverse2_kernel.hpp (one workgroup per visited-set slot walking over the queries of the batch;
// T workers with local queues in lockstep; queues in LDS or, for large SearchQueueSize, in HBM; visited bitmap
// with an undo log instead of an O(N) reset per query).  This file holds the host side: device CSR, scratch,
// PrepareInitIds, the launch, and post_kernel = the epilogue of Search() (tail merge + post-filter walk).
// With IntraQueryThreads = 1 the sequence of expansions and the final queue are exactly the reference's
// single-thread result; with T > 1 they are the reference's result under the lockstep interleaving of its
// workers (deterministic here, racy in the reference), restated by oracle/epsilla_oracle.c for parity.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "index.hpp"
#include "traverse2_kernel.hpp"

namespace eps {

struct GraphDev {
  DevBuf off;        // int64 [n+1]            (CSR form, used when max_degree > 64)
  DevBuf nbr;        // u32 [E]                (CSR) or u32 [n][fixed_deg] padded with 0xFFFFFFFF
  int fixed_deg = 0; // > 0: fixed-stride adjacency — one dependent load less per expansion than CSR
  DevBuf init_ids;   // u32 [L]
  int64_t init_L = -1;
  int64_t max_degree = 0;
  DevBuf visited;    // u32 [slots][words]: all-zero between searches (the kernel undoes what it sets)
  int64_t vis_slots = 0, vis_words = 0;
  bool vis_dirty = true;   // a failed launch may have left bits behind: re-zero before the next search
  DevBuf vlog;       // u32 [slots][vcap] undo log of the visited set
  // r5: visited set by generation stamp (Trv2Args::gens) where HBM allows - u32 [slots][n], zero-initialised once; gen_last = the last stamp handed out
  DevBuf gens;
  int64_t gens_slots = 0, gens_n = 0;
  uint32_t gen_last = 0;
  ScratchClaim gens_claim;   // r6: the table is reclaimable scratch (index.hpp) - any failed allocation of the process may take it back between searches
  GraphDev() { gens_claim.buf = &gens; }
  DevBuf qglobal;    // u64 [slots][qtot] queues of large-L searches
  DevBuf auxglobal;  // int [slots][2*Lq]
  DevBuf queue;      // u64 [nq][L]
  DevBuf counters;   // unsigned long long [2]
  DevBuf tail;       // u64 [nq][k] brute-force tail lists
  DevBuf q8, qstat8;       // prefilter: the batch on the mirror's grid, signed char [nq][d_pad8], float [nq][4]
  // r5, edge constants: the mirror's row constant (acc0) of every neighbour, stored in the adjacency list's layout ([n][fixed_deg] int32), so that
  // an expansion gets the constants of its neighbours in the access that fetches the list and the prefilter gathers nothing but the mirror rows.
  // Valid for one build of the mirror (acc0_epoch); not used while the mirror folds per-batch margins into the constants.
  DevBuf nbr_acc0;
  int64_t acc0_epoch = -1;
  // the prefilter pays only where the 8-bit bound settles most neighbours (uniform-like value distributions: 82 % at 10M x 768);
  // where distances are small against the table's value range (clustered / low intrinsic dimension) most rows pass it and it is
  // pure overhead.  Judged on what the kernel counts: two consecutive searches in which more than 60 % of the neighbour evaluations
  // still read the fp32 row switch it off for this graph (until the graph is replaced).
  int pf_strikes = 0;
  bool pf_off = false;
  DevBuf elog, elog_cnt;   // filtered traversal: u64 [slice][ecap] evaluated (dist, id) keys, u32 [slice] counts
};

void graph_free(GraphDev* g) { delete g; }

int32_t graph_upload(Index& ix) {
  if (!ix.graph_) ix.graph_ = new GraphDev();
  GraphDev& g = *ix.graph_;
  g.init_L = -1;
  g.pf_strikes = 0;
  g.pf_off = false;
  g.acc0_epoch = -1;
  if (g.gens_claim.try_enter()) {   // a new graph: the stamp table of the old one is given back (it is sized by the node count; the next search decides anew)
    g.gens.release();
    g.gens_slots = g.gens_n = 0;
    g.gens_claim.leave();
  }
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
  const u64* tail;    // [nq][tail_k] flat-scan keys of the un-indexed tail (plain id), or null
  int tail_k;
  int k;              // output width
  int K;              // searchLimit = min(n_indexed, limit, L_local): slots that take part in the tail merge
  int Klds;           // slots staged in LDS: K when there is a tail to merge, else 0 (the walk then reads the queue from HBM)
  int Kout;           // results wanted per query (= K for Search(); the whole walk for eps_index_search_walk)
  int cand_num_tail;  // min(L_master, n_total)
  int cand_num;       // min(L_master, n_indexed)
  FilterSpec f;
  u64* run_keys;      // [nq][k] output keys (plain id)
};

__device__ __forceinline__ bool ckey_less(u64 a, u64 b) { return a < b; }  // plain (dist,id) keys
__device__ __forceinline__ u64 plain_key(u64 v) {  // traversal key -> (dist,id) key
  return v == KEY_EMPTY ? KEY_EMPTY : ((v & 0xFFFFFFFF00000000ull) | ((v & 0xFFFFFFFFull) >> 1));
}

// One wavefront per query.  Only the first K slots of the master queue take part in the tail merge (they are staged in
// LDS); the post-filter walk reads slots >= K straight from the queue in HBM, 64 candidates per step (ballot + prefix).
__global__ __launch_bounds__(64) void post_kernel(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 m[];  // [K] head of the master queue as plain keys
  __shared__ int s_cand;
  const int64_t q = blockIdx.x;
  const int L = a.L;
  const int lane = threadIdx.x;
  const u64* qu = a.queue + q * L;
  for (int i = lane; i < a.Klds; i += 64) m[i] = plain_key(qu[i]);
  if (lane == 0) s_cand = a.cand_num;
  __syncthreads();
  if (a.tail && lane == 0) {
    const u64* bf = a.tail + q * a.tail_k;
    int n2 = 0;
    while (n2 < a.tail_k && bf[n2] != KEY_EMPTY) ++n2;
    if (n2 > 0) {
      const int n1 = a.K;
      // MergeTwoQueuesInto1stQueueSeqFixed (:150-217) on m[0..n1): lower_bound of bf[0]
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
      s_cand = a.cand_num_tail;
    }
  }
  __syncthreads();
  const int cand_num = s_cand;
  int res = 0;
  for (int base = 0; base < cand_num && res < a.Kout; base += 64) {
    const int i = base + lane;
    u64 v = KEY_EMPTY;
    if (i < cand_num) v = i < a.Klds ? m[i] : plain_key(qu[i]);
    const bool ok = v != KEY_EMPTY && row_visible(a.f, key_id(v), key_dist(v));
    const u64 mask = __ballot(ok);
    const int rank = res + __popcll(mask & ((1ull << lane) - 1ull));
    if (ok && rank < a.Kout) a.run_keys[q * a.k + rank] = v;
    res += __popcll(mask);
  }
  if (res > a.Kout) res = a.Kout;
  for (int i = res + lane; i < a.k; i += 64) a.run_keys[q * a.k + i] = KEY_EMPTY;
}

// edge constants: out[e] = acc0[nbr[e]] (0 on the padding of a list)
__global__ void edge_acc0_kernel(const u32* nbr, const int* acc0, int* out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const u32 v = nbr[i];
    out[i] = v == 0xFFFFFFFFu ? 0 : acc0[v];
  }
}

// ------------------------------------------------------------------------------------------------ host
template <bool VEC4, int NW, bool QG, bool PF>
static void launch_trv2(const Trv2Args& a, int slots, size_t shm, hipStream_t s) {
  // (idempotent and cheap; per call so that no process-wide state is needed)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(traverse2_kernel<VEC4, NW, QG, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipLaunchKernelGGL((traverse2_kernel<VEC4, NW, QG, PF>), dim3((unsigned)slots), dim3(NW * 64), shm, s, a);
}

// result keys beyond the first K of every query -> empty (the reference's result-count cap)
__global__ void cap_keys_kernel(u64* run_keys, int64_t nq, int k, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq * k && (int)(i % k) >= K) run_keys[i] = KEY_EMPTY;
}

int32_t graph_search_impl(Index& ix, const float* dq, int64_t nq, int k, const eps_search_params& p, u64* run_keys,
                                 int64_t* evals_out, int walk_limit, int64_t ecap_min) {
  GraphDev& g = *ix.graph_;
  const int64_t n = ix.n_indexed_;
  int64_t L = p.master_queue;
  if (L > n) L = n;  // the reference would spin forever in PrepareInitIds when L > n (see prepare_init_ids)
  // The reference accepts SearchQueueSize up to 10^7 and IntraQueryThreads up to 128 (config/config.hpp:28-44).  What the device
  // does not run it refuses - it never runs a different configuration than the one asked for.
  if (L > ((int64_t)1 << 20)) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: SearchQueueSize > 1048576 is not supported (use the flat engines)", EPS_ERRCLASS_DEVICE_RANGE);
  int64_t Lq = p.local_queue;
  if (Lq > n) Lq = n;
  if (Lq > ((int64_t)1 << 20)) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: LocalQueueSize > 1048576 is not supported (use the flat engines)", EPS_ERRCLASS_DEVICE_RANGE);
  const int T = p.intra_threads;
  if (T > TRV2_MAXT) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: IntraQueryThreads > 128 is not supported (the reference's own limit, config/config.hpp:29)", EPS_ERRCLASS_DEVICE_RANGE);
  const int I = (int)std::min<int64_t>(p.sync_interval, 1 << 20);
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
  const int dp = (int)(((g.fixed_deg > 0 ? (int64_t)g.fixed_deg : std::max<int64_t>(g.max_degree, 1)) + 7) / 8 * 8);   // edge slots per worker and step
  if ((int64_t)T * dp > 2048)   // the T adjacency lists of one lockstep step share the workgroup's LDS
    return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: IntraQueryThreads x maximum out-degree (rounded up to 8) > 2048 is not supported: " + std::to_string(T) + " x " +
                                                 std::to_string(dp) + " (at the build's out-degree cap of 64: IntraQueryThreads <= 32)", EPS_ERRCLASS_DEVICE_RANGE);
  // r6: the capacity a worker's queue is LAID OUT with.  Between two MergeAllQueuesToMaster calls (which empty it, :297-326) a worker's queue holds its
  // share of the master's unchecked candidates - PickTopMToWorkers deals them round-robin, every T-th stays with the master: <= ceil(L / T) (:328-356) -
  // plus what its <= I expansions of the round insert, <= out-degree each (:384-444).  A capacity above that bound never clamps an insert and never
  // stops the dealing early, so the walk with LocalQueueSize = min(LocalQueueSize, bound) is the walk with LocalQueueSize, key for key
  // (test_lockstep_workers_match_oracle holds whole queues against the oracle, which keeps the caller's value).  What it buys: T = 4, L = 2000
  // needs 3 x 1460 + 2048 keys instead of 3 x 2000 + 2048 - the queues stay in LDS where the larger layout sent them to HBM.
  if (T > 1) Lq = std::min<int64_t>(Lq, (L + T - 1) / T + (int64_t)I * dp);
  const int64_t qtot = (int64_t)(T - 1) * Lq + Lp2;
  const bool vec4 = (ix.dim_ % 4 == 0) && ((reinterpret_cast<uintptr_t>(ix.d_rows_) & 15) == 0);
  // queues in LDS while a workgroup's working set leaves room for at least two workgroups per CU (the row gathers of
  // the other one cover this one's queue maintenance); larger SearchQueueSize: queues in HBM
  // Between 80 and 150 KB the queues still fit the CU's LDS once: one workgroup of 16 wavefronts per CU then beats queues in
  // HBM with 4 x 4 wavefronts (T = 4, L = 2000 at 10M x 768, batch 1024: 80 ms vs 108 ms); beyond that: queues in HBM.
  // Filtered traversal (eps_search_params::filter_in_traversal): the reference judges deleted rows and the filter on
  // the final top-L walk only (:905-927), so a filter that lets 1 % of the rows through leaves ~L/100 results.  Here every row
  // the search EVALUATES (an order of magnitude more than L) is logged with its distance and the k closest VISIBLE ones are the
  // answer; invisible rows keep their place in the queues, i.e. the walk itself is the reference's.
  const FilterSpec fspec = ix.filter_spec();
  const bool filtered = p.filter_in_traversal != 0 && walk_limit == 0 && (fspec.deleted || fspec.column || fspec.prog);
  // 8-bit lower-bound prefilter of the distance phase (traverse2_kernel.hpp, step d0): pays when a mirror row is much shorter than
  // the fp32 row and the table is beyond the caches; the filtered traversal logs EVERY evaluated distance, so it cannot skip any.
  // EPS_TRV_PREFILTER=0/1 overrides (A/B, small-table tests).
  bool prefilter = !filtered && ix.dim_ >= 128 && n >= 65536 && !g.pf_off;
  const char* pf_env = tune_env("EPS_TRV_PREFILTER");
  if (pf_env) prefilter = !filtered && atoi(pf_env) != 0;
  Quant8View q8v;
  if (prefilter) {
    // the prefilter is an optimisation: if its mirror (n x d bytes, a quarter of the table) cannot be had - HBM-tight tables - the
    // walk runs on the fp32 rows as it did before the prefilter existed, and stops asking for this graph
    const int32_t rc = quant8_view(ix, &q8v);
    if (rc != EPS_OK) {
      q8v = Quant8View();
      g.pf_off = true;
    }
    prefilter = q8v.x8 != nullptr;
  }
  // edge constants (see GraphDev): fixed-stride lists, constants that do not change per batch, and room for one more copy of the lists
  const int* nbr_acc0 = nullptr;
  if (prefilter && g.fixed_deg > 0 && !q8v.per_batch) {
    const int64_t total = n * g.fixed_deg;
    if (g.acc0_epoch != q8v.epoch8) {
      if (g.nbr_acc0.reserve((size_t)total * 4)) {
        hipLaunchKernelGGL(edge_acc0_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, ix.stream_, g.nbr.as<u32>(), q8v.acc0,
                           g.nbr_acc0.as<int>(), total);
        g.acc0_epoch = q8v.epoch8;
      } else {
        (void)hipGetLastError();   // (an optimisation: without it the kernel gathers acc0[id] as before)
      }
    }
    if (g.acc0_epoch == q8v.epoch8) nbr_acc0 = g.nbr_acc0.as<int>();
  }
  const size_t lds_need = traverse2_lds_bytes((int)ix.dim_, T, (int)Lq, qtot, dp, false, prefilter, q8v.cols8);
  // r3 (scripts/lab/trv_large_l2.sh, bench_random_graph.py): for BATCHES that band is better served with the queues in HBM and 8
  // wavefronts per query, two queries per CU (T = 4, L = 2000, batch 1024: 1M x 768 46.6 -> 39.4 ms, 10M-row proxy 56.8 -> 49.3 ms;
  // 4 wavefronts 64.6, 16 wavefronts 66.1), and 8 wavefronts also beat the 4 that queues in HBM used to get (L = 4000: 126.6 ->
  // 93.6 ms, L = 8000: 350.8 -> 246.7 ms).  A handful of queries (<= 256, latency) keeps the one-workgroup-per-CU form.
  const int lds_kb = tune_int("EPS_TRV_LDS_KB", 0);   // (A/B knob)
  const size_t lds_limit = lds_kb > 0 ? (size_t)std::max(16, lds_kb) * 1024 : (nq <= 256 ? (size_t)150 * 1024 : (size_t)80 * 1024);
  const bool qglobal = lds_need > lds_limit;
  const bool one_per_cu = !qglobal && lds_need > (size_t)80 * 1024;
  const size_t shm = traverse2_lds_bytes((int)ix.dim_, T, (int)Lq, qtot, dp, qglobal, prefilter, q8v.cols8);
  // few queries: 16 wavefronts per query (latency); many queries: fewer per query (throughput, more queries per CU)
  const char* waves_s = tune_env("EPS_TRV_WAVES");
  // r6: LDS queues that leave room for only TWO workgroups per CU (40-80 KB: T = 4 from L ~ 600) get 8 wavefronts each - 16 per CU as with four small
  // workgroups, and the step's T x degree rows spread over twice the lanes (10M x 768 proxy, batch 1024, T = 4: L = 700 14.3 -> 12.5 ms, L = 1000
  // 18.8 -> 17.1, L = 2000 37.1 -> 33.3 where HBM queues took 42.5; 16 wavefronts: 17.1 / 23.4 / 45.7 - profiles/r6_traverse_large_queues_proxy.txt)
  int nw = waves_s ? atoi(waves_s) : ((nq <= 256 || one_per_cu) ? 16 : ((qglobal || shm > (size_t)40 * 1024) ? 8 : 4));
  if (const char* wide_s = tune_env("EPS_TRV_WIDE")) nw = atoi(wide_s) != 0 ? 16 : 4;   // (older switch)
  if (nw != 4 && nw != 8 && nw != 16) nw = 4;
  hipDeviceProp_t prop;
  er = hipGetDeviceProperties(&prop, ix.device_);
  if (er != hipSuccess) return ix.hip_fail(er, "device properties");
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  int per_cu = (int)std::min<size_t>((size_t)(16 / nw), (size_t)(160 * 1024) / shm);   // 4 wavefronts per SIMD at this register use (~104 VGPRs)
  if (const char* pc = tune_env("EPS_TRV_PER_CU")) per_cu = std::max(1, atoi(pc));
  if (per_cu < 1) per_cu = 1;
  const int64_t words = (n + 31) / 32;
  int64_t slots = std::min<int64_t>(nq, (int64_t)cus * per_cu);
  slots = std::min<int64_t>(slots, std::max<int64_t>(1, ((int64_t)4 << 30) / (words * 4)));
  if (qglobal) slots = std::min<int64_t>(slots, std::max<int64_t>(1, ((int64_t)8 << 30) / (qtot * 8)));
  const int vcap = (int)std::min<int64_t>((int64_t)1 << 20, std::max<int64_t>(1024, words / 4));
  // results of the traversal are consumed per slice of queries so that the [slice][L] queue copy stays below 2 GiB
  // filtered traversal: slots of a query's evaluation log - every distance the walk evaluates is logged; a walk that evaluates more
  // (large T x I, high degree) reports it and the search is repeated with a log that holds them all (ecap_min, below)
  const int64_t ecap_plan = std::min<int64_t>(n, std::max<int64_t>(std::max<int64_t>(16384, 64 * L), ecap_min));
  const int64_t slice = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)2 << 30) / (std::max<int64_t>(L, p.filter_in_traversal ? ecap_plan : 0) * 8)));
  // visited set: generation stamps (one atomicMax per edge, no reset) when 4 bytes per node and slot fit comfortably - a quarter of the free HBM
  // and at most 64 GB - else the bitmap with its undo log.  EPS_TRV_VISITED=bitmap|stamps overrides (A/B, tests).
  bool stamps = false;
  struct ClaimHold {   // from here to the return of this call the table is not reclaimable (the launches below use it)
    ScratchClaim& c;
    bool mine;
    explicit ClaimHold(ScratchClaim& cl) : c(cl), mine(cl.enter()) {}
    ~ClaimHold() { if (mine) c.leave(); }
  } hold(g.gens_claim);
  {
    const size_t need = (size_t)slots * (size_t)n * 4;
    size_t free_b = 0, total_b = 0;
    if (!g.gens.p) g.gens_slots = g.gens_n = 0;   // (taken back since the last search)
    const bool have = (g.gens_slots >= slots && g.gens_n == n && g.gens.p);
    if (have) stamps = true;
    else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 4 && need <= ((size_t)64 << 30)) stamps = true;
    if (const char* ve = tune_env("EPS_TRV_VISITED")) stamps = std::strcmp(ve, "stamps") == 0 ? true : (std::strcmp(ve, "bitmap") == 0 ? false : stamps);
    if (stamps && !have) {
      g.gens.release();
      if (g.gens.reserve(need)) {
        er = hipMemsetAsync(g.gens.p, 0, need, s);
        if (er != hipSuccess) return ix.hip_fail(er, "memset visited stamps");
        g.gens_slots = slots;
        g.gens_n = n;
        g.gen_last = 0;
        if (const char* st0 = tune_env("EPS_TRV_STAMP_START")) g.gen_last = (uint32_t)strtoul(st0, nullptr, 0);   // (tests: the 32-bit counter's wrap-around)
      } else {
        (void)hipGetLastError();
        g.gens_slots = g.gens_n = 0;
        stamps = false;   // (no room: the bitmap)
      }
    }
    if (!stamps && g.gens.p) {   // the bitmap serves this search: a stale stamp table is not kept beside it
      g.gens.release();
      g.gens_slots = g.gens_n = 0;
    }
  }
  if ((!stamps && (!g.visited.reserve((size_t)slots * words * 4) || !g.vlog.reserve((size_t)slots * vcap * 4))) ||
      !g.queue.reserve((size_t)slice * L * 8) || !g.counters.reserve(256) ||
      (qglobal && (!g.qglobal.reserve((size_t)slots * qtot * 8) || !g.auxglobal.reserve((size_t)slots * 2 * Lq * 4))))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (traversal scratch)");
  if (!stamps && (g.vis_dirty || g.vis_slots < slots || g.vis_words != words)) {
    // (re)establish the invariant the kernel maintains: every slot's bitmap is all-zero between searches
    er = hipMemsetAsync(g.visited.p, 0, g.visited.cap, s);
    if (er != hipSuccess) return ix.hip_fail(er, "memset visited");
    g.vis_slots = (int64_t)(g.visited.cap / ((size_t)words * 4));
    g.vis_words = words;
  }
  if (!stamps) g.vis_dirty = true;   // until this search has completed (stamps need no such care: a half-finished walk leaves only old stamps behind)
  er = hipMemsetAsync(g.counters.p, 0, 256, s);
  if (er != hipSuccess) return ix.hip_fail(er, "memset");
  const bool prof = tune_env("EPS_TRV_PROF") != nullptr;

  Trv2Args a;
  a.rows = ix.d_rows_;
  a.dim = (int)ix.dim_;
  a.metric = ix.metric_;
  a.off = g.fixed_deg > 0 ? nullptr : g.off.as<int64_t>();
  a.nbr = g.nbr.as<u32>();
  a.fixed_deg = g.fixed_deg;
  a.dp = dp;
  a.hslots = traverse2_hash_slots(T, dp);
  a.init_ids = g.init_ids.as<u32>();
  a.L = (int)L;
  a.Lq = (int)Lq;
  a.Lp2 = Lp2;
  a.T = T;
  a.I = I;
  a.qtot = qtot;
  a.qglobal = qglobal ? g.qglobal.as<u64>() : nullptr;
  a.auxglobal = qglobal ? g.auxglobal.as<int>() : nullptr;
  a.visited = g.visited.as<u32>();
  a.words = words;
  a.vlog = g.vlog.as<u32>();
  a.vcap = vcap;
  a.gens = stamps ? g.gens.as<u32>() : nullptr;
  a.gens_n = n;
  a.gen_base = 0;
  a.counters = g.counters.as<unsigned long long>();
  a.prof = prof ? g.counters.as<unsigned long long>() + 8 : nullptr;
  if (filtered && k > 1024) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: filter_in_traversal returns at most 1024 rows per query", EPS_ERRCLASS_DEVICE_RANGE);
  const int64_t ecap = filtered ? ecap_plan : 0;
  a.elog = nullptr;
  a.elog_cnt = nullptr;
  a.elog_cap = (int)ecap;
  a.x8 = prefilter ? q8v.x8 : nullptr;
  a.acc0 = q8v.acc0;
  a.nbr_acc0 = nbr_acc0;
  a.scal8 = q8v.scal8;
  a.d_pad8 = q8v.d_pad8;
  a.cols8 = q8v.cols8;
  a.u8 = q8v.u;
  a.q8 = nullptr;
  a.qstat8 = nullptr;
  if (prefilter) {
    if (!g.q8.reserve((size_t)nq * q8v.d_pad8) || !g.qstat8.reserve((size_t)nq * 16)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
    quant8_queries(ix, q8v, dq, nq, g.q8.as<signed char>(), g.qstat8.as<float>());
  }
  {   // fp32 rounding of the distances step d compares with the bound: G lanes x (d / G) sequential fmas + a log2(G)-level tree
    const int G = group_lanes(ix.dim_, vec4);
    const float terms = (float)((ix.dim_ + G - 1) / G) + 6.f;
    a.slack8 = std::max(8e-6f, 2.f * (3.f * terms + 6.f) * 5.9604645e-8f);
  }

  // brute-force tail over the rows the graph does not cover yet (:885-900)
  const int64_t n_total = ix.n_rows_;
  const u64* tail = nullptr;
  const int limit = walk_limit > 0 ? walk_limit : k;   // the reference's `limit`
  const int tail_k = std::min(limit, 8192);            // only min(|tail|, limit) tail entries are ever merged (:890-899)
  if (n_total > n) {
    if (!g.tail.reserve((size_t)nq * tail_k * 8)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
    int32_t rc = ix.flat_stream(dq, nq, tail_k, n, n_total, g.tail.as<u64>(), false);
    if (rc != EPS_OK) return rc;
    tail = g.tail.as<u64>();
  }
  PostArgs pa;
  pa.queue = g.queue.as<u64>();
  pa.L = (int)L;
  pa.k = k;
  pa.tail_k = tail_k;
  int64_t K = n;
  if (limit < K) K = limit;
  if (p.local_queue < K) K = p.local_queue;
  if (L < K) K = L;
  if (tail && K > 8192) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "search: limit > 8192 with an un-indexed tail is not supported (rebuild the graph over the new rows)");
  pa.K = (int)K;
  pa.Klds = tail ? (int)K : 0;   // <= 64 KB of LDS; without a tail nothing is staged, whatever K
  pa.Kout = walk_limit > 0 ? k : (int)K;
  pa.cand_num_tail = (int)std::min<int64_t>(L, n_total);
  pa.cand_num = (int)std::min<int64_t>(L, n);
  pa.f = ix.filter_spec();

  (void)hipEventRecord(ix.evk0_, s);
  for (int64_t q0 = 0; q0 < nq; q0 += slice) {
    const int64_t cnt = std::min(slice, nq - q0);
    a.queries = dq + q0 * ix.dim_;
    a.q8 = prefilter ? g.q8.as<signed char>() + q0 * q8v.d_pad8 : nullptr;
    a.qstat8 = prefilter ? g.qstat8.as<float>() + q0 * 4 : nullptr;
    a.nq = cnt;
    a.out_queue = g.queue.as<u64>();
    if (filtered) {
      if (!g.elog.reserve((size_t)slice * ecap * 8) || !g.elog_cnt.reserve((size_t)slice * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (evaluation log)");
      a.elog = g.elog.as<u64>();
      a.elog_cnt = g.elog_cnt.as<u32>();
    }
    const int sl = (int)std::min<int64_t>(slots, cnt);
    if (stamps) {   // the stamps this launch hands out: one per query of a slot; a wrap-around of the 32-bit counter starts over from a zeroed table
      const uint32_t per_slot = (uint32_t)((cnt + sl - 1) / sl);
      if (g.gen_last > 0xFFFFFFF0u - per_slot) {
        er = hipMemsetAsync(g.gens.p, 0, (size_t)g.gens_slots * (size_t)g.gens_n * 4, s);
        if (er != hipSuccess) return ix.hip_fail(er, "memset visited stamps");
        g.gen_last = 0;
      }
      a.gen_base = g.gen_last;
      g.gen_last += per_slot;
    }
#define EPS_TRV_LAUNCH(V4, NW_)                                        \
  do {                                                                 \
    if (qglobal) {                                                     \
      if (prefilter) launch_trv2<V4, NW_, true, true>(a, sl, shm, s);  \
      else launch_trv2<V4, NW_, true, false>(a, sl, shm, s);           \
    } else {                                                           \
      if (prefilter) launch_trv2<V4, NW_, false, true>(a, sl, shm, s); \
      else launch_trv2<V4, NW_, false, false>(a, sl, shm, s);          \
    }                                                                  \
  } while (0)
    if (vec4) {
      if (nw == 16) EPS_TRV_LAUNCH(true, 16); else if (nw == 8) EPS_TRV_LAUNCH(true, 8); else EPS_TRV_LAUNCH(true, 4);
    } else {
      if (nw == 16) EPS_TRV_LAUNCH(false, 16); else if (nw == 8) EPS_TRV_LAUNCH(false, 8); else EPS_TRV_LAUNCH(false, 4);
    }
#undef EPS_TRV_LAUNCH
    ix.stats_.main_kernel_launches += 1;
    ix.stats_.main_kernel_bits = 32;
    if (q0 + cnt >= nq) (void)hipEventRecord(ix.evk1_, s);   // (with several slices the pair spans all traversal launches and the post kernels between them)
    pa.tail = tail ? tail + q0 * tail_k : nullptr;
    pa.run_keys = run_keys + q0 * k;
    if (filtered) {
      // the k closest visible evaluated rows (+ the visible rows of the un-indexed tail, already filtered by its flat scan)
      launch_merge_lists(a.elog, (int)ecap, k, cnt, run_keys + q0 * k, false, s, a.elog_cnt, &fspec);
      if (tail) launch_merge_lists(tail + q0 * tail_k, tail_k, k, cnt, run_keys + q0 * k, true, s);
      // the same result-count cap as the unfiltered path: min(n_indexed, limit, LocalQueueSize) (:872)
      const int64_t Kf = std::min<int64_t>(std::min<int64_t>(n, limit), p.local_queue);
      if (Kf < k) hipLaunchKernelGGL(cap_keys_kernel, dim3((unsigned)((cnt * k + 255) / 256)), dim3(256), 0, s, run_keys + q0 * k, cnt, k, (int)Kf);
    } else {
      hipLaunchKernelGGL(post_kernel, dim3((unsigned)cnt), dim3(64), (size_t)pa.Klds * 8, s, pa);
    }
  }
  er = hipGetLastError();
  if (er != hipSuccess) return ix.hip_fail(er, "traversal launch");
  unsigned long long h[24];
  std::memset(h, 0, sizeof(h));
  er = hipMemcpyAsync(h, g.counters.p, sizeof(h), hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return ix.hip_fail(er, "traversal");
  if (!stamps) g.vis_dirty = false;
  if (filtered && h[5] > 0) {
    // some query evaluated more rows than its log holds; the evaluations that were dropped are the LATE ones - the closest.  Repeat
    // with a log sized for the longest walk seen (bounded by n: a walk evaluates a row at most once)
    const int64_t need = std::min<int64_t>(n, (int64_t)h[6] + (int64_t)h[6] / 4 + 1024);
    if (need > ecap) return graph_search_impl(ix, dq, nq, k, p, run_keys, evals_out, walk_limit, need);
  }
  if (prof) {
    static const char* names[10] = {"seeds+sort", "scatter", "select", "gather+visited", "dedupe", "distances", "rank sort", "queue insert", "merge-all", "results+reset"};
    unsigned long long tot = 0;
    for (int i = 0; i < 10; ++i) tot += h[8 + i];
    fprintf(stderr, "[eps trv] nq=%lld T=%d L=%lld I=%d %s queues, %d wavefronts/query, %lld slots, LDS %zu B: steps/query %.1f rounds/query %.1f expansions/query %.1f evals/query %.1f\n",
            (long long)nq, T, (long long)L, I, qglobal ? "HBM" : "LDS", nw, (long long)slots, shm, (double)h[2] / nq, (double)h[3] / nq, (double)h[1] / nq, (double)h[0] / nq);
    for (int i = 0; i < 10; ++i) fprintf(stderr, "[eps trv]   %-16s %5.1f %%\n", names[i], tot ? 100.0 * h[8 + i] / tot : 0.0);
  }
  ix.stats_.dist_evals += (int64_t)h[0];
  ix.stats_.expansions += (int64_t)h[1];
  if (prefilter) ix.stats_.rerank_rows += (int64_t)h[4];   // fp32 rows step d still read (seeds not counted)
  if (prefilter && !pf_env) {
    const unsigned long long nbr_evals = h[0] > (unsigned long long)(L * nq) ? h[0] - (unsigned long long)(L * nq) : 0;
    if (nbr_evals >= 4096) {   // (enough evaluations to judge)
      g.pf_strikes = (double)h[4] > 0.6 * (double)nbr_evals ? g.pf_strikes + 1 : 0;
      if (g.pf_strikes >= 2) g.pf_off = true;
    }
  }
  if (prof && prefilter) fprintf(stderr, "[eps trv]   8-bit prefilter: %.1f of %.1f neighbour evaluations per query read the fp32 row\n", (double)h[4] / nq, (double)(h[0] - (unsigned long long)(L * nq)) / nq);
  if (evals_out) *evals_out = (int64_t)h[0];
  return EPS_OK;
}

int32_t graph_search(Index& ix, const float* dq, int64_t nq, int k, const eps_search_params& p, u64* run_keys, int64_t* evals_out, int walk_limit) {
  return graph_search_impl(ix, dq, nq, k, p, run_keys, evals_out, walk_limit, 0);
}

}  // namespace eps
