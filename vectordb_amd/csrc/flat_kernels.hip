// Flat-scan path of libepsilla_gfx950: the device form of VecSearchExecutor::BruteForceSearch /
// PreFilterBruteForceSearch (reference: engine/db/execution/vec_search_executor.cpp:717-831).
//
// The reference computes every distance into a scratch Candidate[N], compacts by deleted/filter and
// std::sorts all N survivors.  Here one pass streams the contiguous fp32 row store from HBM exactly once
// per group of NQ queries (16 B/lane coalesced loads, G lanes per row, shuffle reduction) and keeps a
// k-entry sorted list per wavefront in registers; per-wave lists are merged by a second small kernel.
// Only candidates that would enter the top-k touch the deleted bitset / filter column, so the
// algorithmic HBM traffic is N*dim*4 bytes per query group.
#include "kernels.hpp"
#include "stream8_kernel.hpp"

namespace eps {

// ------------------------------------------------------------------------------------------------
template <int NQ, int KPL>
__device__ __forceinline__ void offer(WaveTopK<KPL> (&list)[NQ], u64 (&thr)[NQ], int q, u64 key, bool valid,
                                      const FilterSpec& f, int k, bool unique) {
  u64 m = __ballot(valid && key < thr[q]);
  while (m) {
    const int l = __ffsll((long long)m) - 1;
    m &= m - 1;
    const u64 x = shfl64(key, l);
    if (x < thr[q] && row_visible(f, key_id(x), key_dist(x))) {
      if (unique) list[q].insert_unique(x); else list[q].insert(x);
      const u64 kth = list[q].entry(k - 1);
      thr[q] = kth < thr[q] ? kth : thr[q];
    }
  }
}

template <int NQ, int KPL, bool VEC4, int U>
__global__ __launch_bounds__(256) void flat_scan_kernel(FlatScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  const int64_t q0 = (int64_t)blockIdx.y * NQ;
  for (int i = threadIdx.x; i < NQ * qstride; i += 256) {
    const int q = i / qstride, c = i - q * qstride;
    const int64_t qq = (q0 + q < a.nq) ? q0 + q : a.nq - 1;
    smem[i] = c < dim ? a.queries[qq * dim + c] : 0.f;
  }
  __syncthreads();

  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  const int64_t W = (int64_t)gridDim.x * 4;
  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nrows = a.row_end - a.row_begin;
  const int64_t chunk = (nrows + W - 1) / W;
  const int64_t begin = a.row_begin + w * chunk;
  const int64_t end = begin + chunk < a.row_end ? begin + chunk : a.row_end;

  WaveTopK<KPL> list[NQ];
  u64 thr[NQ], lo[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    list[q].init();
    const int64_t qq = (q0 + q < a.nq) ? q0 + q : a.nq - 1;
    thr[q] = a.thr_in ? a.thr_in[qq] : KEY_EMPTY;
    lo[q] = a.lo_in ? a.lo_in[qq * a.lo_stride] : 0;   // result pages beyond the first: only keys ordered after the previous page's last
  }

  for (int64_t r0 = begin; r0 < end; r0 += RPW * U) {
    const float* rp[U];
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = r0 + u * RPW + g;
      const int64_t rc = row[u] < end ? row[u] : end - 1;
      rp[u] = a.rows + rc * dim;
    }
    float acc[U][NQ];
    row_dists<U, NQ, VEC4>(rp, smem, qstride, dim, a.metric, G, acc);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool valid = (t == 0) && row[u] < end;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const u64 key = make_key(finish_dist(a.metric, acc[u][q]), (u32)row[u]);
        offer<NQ, KPL>(list, thr, q, key, valid && (!a.lo_in || key > lo[q]), a.f, a.k, false);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (q0 + q < a.nq) list[q].store(a.partial + ((q0 + q) * W + w) * a.k, a.k);
}

static int pick_kpl(int k) { return k <= 64 ? 1 : k <= 128 ? 2 : k <= 256 ? 4 : k <= 512 ? 8 : 16; }
static int pick_nq(int64_t nq, int k, int dim) {
  if (k > 128) return 1;
  int want = nq >= 3 ? 4 : (nq == 2 ? 2 : 1);
  // the NQ queries are staged in LDS: keep NQ * dim floats within 64 KB
  while (want > 1 && (size_t)want * ((dim + 3) & ~3) * sizeof(float) > 65536) want >>= 1;
  return want;
}

int flat_scan_waves(int64_t nrows, int64_t nq, int dim) {
  (void)dim;
  // callers size `partial` with the same k they launch with; NQ depends on k only through k > 128,
  // which can only increase the number of groups -> fewer blocks in x. Use the smallest NQ for the bound.
  const int64_t groups_min = (nq + 3) / 4;
  int64_t gx_rows = (nrows + 127) / 128;  // >= 32 rows per wavefront
  if (gx_rows < 1) gx_rows = 1;
  int64_t gx_cap = 2048 / (groups_min > 0 ? groups_min : 1);
  if (gx_cap < 1) gx_cap = 1;
  int64_t gx = gx_rows < gx_cap ? gx_rows : gx_cap;
  return (int)(gx * 4);
}

template <int NQ, int KPL>
static void launch_flat_scan_t(const FlatScanArgs& a, bool vec4, hipStream_t s) {
  const int gx = a.W / 4;
  const int64_t groups = (a.nq + NQ - 1) / NQ;
  dim3 grid(gx, (unsigned)groups);
  const size_t shm = (size_t)NQ * ((a.dim + 3) & ~3) * sizeof(float);
  if (vec4)
    hipLaunchKernelGGL((flat_scan_kernel<NQ, KPL, true, 4>), grid, dim3(256), shm, s, a);
  else
    hipLaunchKernelGGL((flat_scan_kernel<NQ, KPL, false, 4>), grid, dim3(256), shm, s, a);
}

void launch_flat_scan(const FlatScanArgs& a, hipStream_t s) {
  if (a.nq <= 0 || a.row_end <= a.row_begin) return;
  const bool vec4 = (a.dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.rows) & 15) == 0);
  const int kpl = pick_kpl(a.k);
  const int nq = pick_nq(a.nq, a.k, a.dim);
#define EPS_CASE(NQ_, KPL_) \
  if (nq == NQ_ && kpl == KPL_) return launch_flat_scan_t<NQ_, KPL_>(a, vec4, s);
  EPS_CASE(1, 1) EPS_CASE(2, 1) EPS_CASE(4, 1)
  EPS_CASE(1, 2) EPS_CASE(2, 2) EPS_CASE(4, 2)
  EPS_CASE(1, 4) EPS_CASE(1, 8) EPS_CASE(1, 16)
#undef EPS_CASE
}

// ------------------------------------------------------------------------------------------------
// merge the `lists` key slots of each query (per-wave sorted lists, or an unsorted candidate list) into its top-k.
// One block per query.
// NW wavefronts per query: 4 for batches, 16 for a handful of queries (a single query's 4096 seed keys: 4 rounds instead of 16)
template <int KPL, int NW>
__global__ __launch_bounds__(NW * 64) void merge_lists_kernel(const u64* partial, int lists, int k, u64* run_keys,
                                                          int merge_run, const u32* counts, FilterSpec vis, u64 id_stride, u32 id_head,
                                                          u32* seed_cand, int seed_cap, u32* seed_cnt) {
  __shared__ u64 sh[NW][KPL * 64];
  constexpr int NT = NW * 64;
  const int64_t q = blockIdx.x;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const u64* src = partial + q * (int64_t)lists;   // `lists` = key slots per query
  int total = lists;
  if (counts) total = counts[q] < (u32)total ? (int)counts[q] : total;  // variable-length candidate list
  WaveTopK<KPL> L[1];
  u64 thr[1];
  L[0].init();
  thr[0] = KEY_EMPTY;
  FilterSpec nof = {nullptr, nullptr, 0, 0, 0, 0};
  if (merge_run) {
    thr[0] = run_keys[q * k + (k - 1)];
    if (wave == 0) L[0].load(run_keys + q * k, k);
  }
  const int rounded = (total + NT - 1) / NT * NT;
  for (int i = threadIdx.x; i < rounded; i += NT) {
    const u64 key = i < total ? src[i] : KEY_EMPTY;
    if (id_stride) {   // seed selection over a SAMPLE: entry ids are sample indices (seed_row maps them to rows)
      const bool ok = key != KEY_EMPTY && row_visible(vis, seed_row(key_id(key), id_head, id_stride), key_dist(key));
      offer<1, KPL>(L, thr, 0, key, ok, nof, k, false);
    } else {
      offer<1, KPL>(L, thr, 0, key, key != KEY_EMPTY, vis, k, false);   // vis: only rows the filter lets through
    }
  }
  L[0].store(&sh[wave][0], k);
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < NW; ++w) {
      for (int e0 = 0; e0 < k; e0 += 64) {
        const int e = e0 + lane;
        const u64 key = e < k ? sh[w][e] : KEY_EMPTY;
        offer<1, KPL>(L, thr, 0, key, key != KEY_EMPTY, nof, k, false);
      }
    }
    if (seed_cand) {   // the k best seeds' rows -> the query's candidate list; run_keys stays empty (the re-rank fills it with exact keys)
      int c = 0;
#pragma unroll
      for (int r = 0; r < KPL; ++r) {
        const int e = r * 64 + lane;
        const u64 key = L[0].key[r];
        const bool ok = e < k && key != KEY_EMPTY;
        const unsigned long long m = __ballot(ok);
        if (ok) seed_cand[q * (int64_t)seed_cap + c + __popcll(m & ((1ull << lane) - 1ull))] = id_stride ? seed_row(key_id(key), id_head, id_stride) : key_id(key);
        c += __popcll(m);
        if (e < k) run_keys[q * k + e] = KEY_EMPTY;
      }
      if (lane == 0) seed_cnt[q] = (u32)c;
    } else {
      L[0].store(run_keys + q * k, k);
    }
  }
}

void launch_merge_lists(const u64* partial, int lists, int k, int64_t nq, u64* run_keys, bool merge_run, hipStream_t s,
                        const u32* counts, const FilterSpec* visible, u64 id_stride, u32 id_head, u32* seed_cand, int seed_cap, u32* seed_cnt) {
  const FilterSpec vis = visible ? *visible : no_filter();
  if (nq <= 0) return;
  const int kpl = pick_kpl(k);
  const bool wide = nq <= 64 && kpl <= 2;
#define EPS_CASE(KPL_) \
  if (kpl == KPL_) {   \
    if (wide && KPL_ <= 2)                                                                                                                                       \
      hipLaunchKernelGGL((merge_lists_kernel<(KPL_ <= 2 ? KPL_ : 1), 16>), dim3((unsigned)nq), dim3(1024), 0, s, partial, lists, k, run_keys, merge_run ? 1 : 0, counts, vis, \
                         id_stride, id_head, seed_cand, seed_cap, seed_cnt);                                                                                      \
    else                                                                                                                                                         \
      hipLaunchKernelGGL((merge_lists_kernel<KPL_, 4>), dim3((unsigned)nq), dim3(256), 0, s, partial, lists, k, run_keys, merge_run ? 1 : 0, counts, vis, id_stride, id_head, \
                         seed_cand, seed_cap, seed_cnt);                                                                                                          \
    return;            \
  }
  EPS_CASE(1) EPS_CASE(2) EPS_CASE(4) EPS_CASE(8) EPS_CASE(16)
#undef EPS_CASE
}

// ------------------------------------------------------------------------------------------------
// exact re-rank of candidate rows gathered by id (fp32, direct form) into the running top-k.
// NW wavefronts per query: 4 for batches (one workgroup per query fills the chip), 16 for a handful of queries (latency: a stage's
// ~150 candidate rows are 3 gather rounds instead of 10).
template <int KPL, bool VEC4, int NW>
__global__ __launch_bounds__(NW * 64) void rerank_kernel(RerankArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [qstride] query, then NW lists
  constexpr int NT = NW * 64;
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  u64* sh = reinterpret_cast<u64*>(smem + qstride);  // qstride*4 bytes is a multiple of 16
  const int64_t q = blockIdx.x;
  for (int i = threadIdx.x; i < qstride; i += NT) smem[i] = i < dim ? a.queries[q * dim + i] : 0.f;
  if (a.fuse && a.gsync && q == 0)
    for (int i = threadIdx.x; i < 256; i += NT) a.gsync[i] = 0;
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  constexpr int U = 4;
  __shared__ u32 s8_kept, s8_lost;
  __shared__ int s8_T;
  if (a.s8_G) {   // one-pass search: the selection step in front of the re-rank (stream8_kernel.hpp)
    if (threadIdx.x == 0) s8_kept = s8_lost = 0;
    if (wave == 0) {
      int gk;
      const int T = stream8_threshold_of(a.s8_G + q * (a.s8_slots * S8_SLOT_STRIDE), a.k, a.qstat + q * 4, a.scal, a.metric, a.u, a.slack, lane, gk, a.s8_slots);
      if (lane == 0) s8_T = T;
    }
    __syncthreads();
    stream8_select<NT>(s8_T, a.s8_counts + q * (int64_t)a.s8_waves, a.s8_lists + q * (int64_t)a.s8_waves * S8_WAVE_CAP, a.s8_waves, a.s8_cand + q * (int64_t)a.cap,
                       a.cap, &s8_kept, &s8_lost);
    __syncthreads();
    if (threadIdx.x == 0 && s8_lost) atomicAdd(a.overflow, 1u);   // a wavefront's list lost entries: the caller repeats the batch on the staged chain
    if (a.s8_reset && (int)threadIdx.x < a.s8_slots)   // (the table has been read - threshold and selection above; the next call finds it empty)
      const_cast<int*>(a.s8_G)[(q * a.s8_slots + threadIdx.x) * S8_SLOT_STRIDE] = S8_EMPTY;
  }
  const u32 cnt_raw = a.s8_G ? s8_kept : a.cand_count[q];
  u32 cnt = cnt_raw;
  if (cnt > (u32)a.cap) cnt = (u32)a.cap;
  const u32* cand = a.cand + q * (int64_t)a.cap;

  WaveTopK<KPL> L[1];
  u64 thr[1];
  L[0].init();
  thr[0] = a.s8_G ? KEY_EMPTY : a.run_keys[q * a.k + (a.k - 1)];
  if (wave == 0 && !a.s8_G) L[0].load(a.run_keys + q * a.k, a.k);

  for (u32 c0 = wave * RPW * U; c0 < cnt; c0 += NW * RPW * U) {
    const float* rp[U];
    u32 id[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32 ci = c0 + u * RPW + g;
      ok[u] = ci < cnt;
      id[u] = cand[ok[u] ? ci : cnt - 1];
      rp[u] = a.rows + (int64_t)id[u] * dim;
    }
    float acc[U][1];
    row_dists<U, 1, VEC4>(rp, smem, qstride, dim, a.metric, G, acc);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u64 key = make_key(finish_dist(a.metric, acc[u][0]), id[u]);
      offer<1, KPL>(L, thr, 0, key, ok[u] && t == 0, a.f, a.k, true);
    }
  }
  L[0].store(sh + wave * (KPL * 64), a.k);
  __syncthreads();   // (every wavefront has read the query's candidate count by now)
  if (wave == 0) {
    FilterSpec nof = {nullptr, nullptr, 0, 0, 0, 0};
    for (int w = 1; w < NW; ++w) {
      for (int e0 = 0; e0 < a.k; e0 += 64) {
        const int e = e0 + lane;
        const u64 key = e < a.k ? sh[w * (KPL * 64) + e] : KEY_EMPTY;
        offer<1, KPL>(L, thr, 0, key, key != KEY_EMPTY, nof, a.k, true);
      }
    }
    L[0].store(a.run_keys + q * a.k, a.k);
    if (a.fin_ids) {   // the call's last re-rank: the caller-visible result of this query (finalize_kernel's arithmetic)
      int c = 0;
#pragma unroll
      for (int r = 0; r < KPL; ++r) {
        const int e = r * 64 + lane;
        const u64 key = L[0].key[r];
        const bool valid = e < a.k && key != KEY_EMPTY;
        if (e < a.k) {
          a.fin_ids[q * a.k + e] = valid ? (int64_t)key_id(key) * a.fin_stride + a.fin_base : -1;
          a.fin_dist[q * a.k + e] = valid ? key_dist(key) : __builtin_inff();
        }
        c += __popcll(__ballot(valid));
      }
      if (a.fin_counts && lane == 0) a.fin_counts[q] = c;
    }
    if (a.fuse) {
      const u64 kth = L[0].entry(a.k - 1);   // (wave-uniform)
      if (lane == 0) {
        if (a.fuse & 1) {
          if (cnt_raw > (u32)a.cap) atomicAdd(a.overflow, 1u);
          atomicAdd(a.total, (unsigned long long)cnt);
        }
        a.cand_count[q] = 0;
        if (a.T_next) {
          const float* qs = a.qstat + q * 4;
          if (a.bits == 8) {
            static_cast<int*>(a.T_next)[q] = kth == KEY_EMPTY ? -(1 << 30) : stage_threshold8(key_dist(kth), qs, a.scal, a.metric, a.u, a.slack, 0);
          } else {
            static_cast<float*>(a.T_next)[q] = kth == KEY_EMPTY ? 3.0e38f : stage_threshold16(key_dist(kth), qs, a.scal, a.metric, a.slack, 0);
          }
        }
        if (a.pub) {   // (lane 0 of wavefront 0 issued every one of this block's counter updates above)
          __threadfence();
          if (atomicAdd(a.pub_ticket, 1u) == gridDim.x - 1) {
            const u32 ov = atomicAdd(a.overflow, 0u);
            const unsigned long long tot = atomicAdd(a.total, 0ull);
            a.pub[0] = ov;
            a.pub[2] = (u32)tot;
            a.pub[3] = (u32)(tot >> 32);
            __threadfence_system();
            if (a.s8_reset) {   // (every block has finished with the counters: each took its ticket after its last update)
              *a.overflow = 0u;
              *a.total = 0ull;
              *a.pub_ticket = 0u;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The re-rank of a ONE-PASS call (stream8_kernel.hpp; r5, late): the same result as rerank_kernel with s8_G set - selection against the final
// table, exact fp32 distances of the survivors, the k best by (distance, id), the call's bookkeeping - laid out for what this launch is: the tail
// of a 0.18 ms call, one workgroup per query, 50-500 candidates.  rerank_kernel spent ~25 us there (83 us at k = 64) of which little was
// memory time: a chain of waits (query to LDS | table -> threshold | list counts | list entries | candidates through global memory | a
// row's three 16-byte pieces per lane one round trip after the other | 15 per-wavefront lists merged one entry at a time).  Here
//  * everything that does not depend on the threshold is in flight before it is computed (the query, the lists' counts);
//  * the survivors' ids stay in LDS;
//  * a row's pieces are all issued before the first is used (row_dists<.., NL = 3>: the fmas run in the same order, the sums are the
//    stream engine's bit for bit);
//  * the k best are found by RANK: every key is compared with every other (LDS broadcast reads), a key of rank r < k goes to slot r.  Keys are
//    unique ((distance, row) with every row in exactly one list), so the ranks are a permutation.
// Up to S8R_CAP survivors per query (more: the overflow counter, i.e. the staged chain repeats the call - as with `cap` before).
constexpr int S8R_CAP = 4096, S8R_SRC = 2 * S8R_CAP;   // (survivors per query; raw list entries per query: more than that is a call for the staged chain)
template <bool VEC4>
__global__ __launch_bounds__(1024) void s8_rerank_kernel(RerankArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [qstride] query, then u64 keys[S8R_CAP]
  constexpr int NT = 1024, NW = 16, U = 4;   // (U x NL = 12 pieces of 16 bytes per lane in flight; 98 registers of the 128 that 1024 threads leave each)
  __shared__ u32 ids_s[S8R_CAP];
  __shared__ u64 best_s[64];
  __shared__ u32 kept_s, lost_s, wtot_s[16];
  __shared__ int T_s;
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  u64* keys_s = reinterpret_cast<u64*>(smem + qstride);
  const int64_t q = blockIdx.x;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  // ---- in flight before the threshold exists: the query, this thread's share of the lists' counts (<= S8_MAX_WAVES / NT = 8 lists)
  const float q0v = (int)threadIdx.x < dim ? a.queries[q * dim + threadIdx.x] : 0.f;
  const u32* counts = a.s8_counts + q * (int64_t)a.s8_waves;
  const u64* lists = a.s8_lists + q * (int64_t)a.s8_waves * S8_WAVE_CAP;
  u32 have[S8_MAX_WAVES / NT];
#pragma unroll
  for (int j = 0; j < S8_MAX_WAVES / NT; ++j) {
    const int w = (int)threadIdx.x + j * NT;
    have[j] = w < a.s8_waves ? counts[w] : 0u;
  }
  if (threadIdx.x == 0) kept_s = lost_s = 0;
  if (threadIdx.x < 64) best_s[threadIdx.x] = KEY_EMPTY;
  // this thread's lists laid end to end: where its entries start in the flat order of all raw entries (block-wide exclusive scan)
  u32 mine_n = 0;
  bool lost = false;
#pragma unroll
  for (int j = 0; j < S8_MAX_WAVES / NT; ++j) {
    lost |= have[j] > (u32)S8_WAVE_CAP;
    have[j] = have[j] < (u32)S8_WAVE_CAP ? have[j] : (u32)S8_WAVE_CAP;
    mine_n += have[j];
  }
  u32 incl = mine_n;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 v = __shfl_up(incl, o);
    incl += lane >= o ? v : 0u;
  }
  if (lane == 63) wtot_s[wave] = incl;
  if (wave == 0) {
    int gk;
    const int T = stream8_threshold_of(a.s8_G + q * (a.s8_slots * S8_SLOT_STRIDE), a.k, a.qstat + q * 4, a.scal, a.metric, a.u, a.slack, lane, gk, a.s8_slots);
    if (lane == 0) T_s = T;
  }
  if ((int)threadIdx.x < qstride) smem[threadIdx.x] = q0v;
  for (int i = threadIdx.x + NT; i < qstride; i += NT) smem[i] = i < dim ? a.queries[q * dim + i] : 0.f;
  __syncthreads();
  // ---- selection, flat: one LDS word per raw entry says where it lives, then every entry is ONE independent load (a thread walking its own list
  // pays a memory round trip per entry - the lists of a k = 64 call hold up to ~20)
  u32* src_s = reinterpret_cast<u32*>(keys_s);   // (S8R_SRC words: the keys' space, not yet in use)
  u32 total_raw = 0, off = incl - mine_n;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const u32 tw = wtot_s[w];
    total_raw += tw;
    off += w < wave ? tw : 0u;
  }
  if (lost || total_raw > (u32)S8R_SRC) lost_s = 1;
#pragma unroll
  for (int j = 0; j < S8_MAX_WAVES / NT; ++j) {
    const u32 w = threadIdx.x + (u32)j * NT;
    for (u32 i = 0; i < have[j]; ++i, ++off)
      if (off < (u32)S8R_SRC) src_s[off] = w * (u32)S8_WAVE_CAP + i;
  }
  __syncthreads();
  const int T = T_s;
  total_raw = total_raw < (u32)S8R_SRC ? total_raw : (u32)S8R_SRC;
  for (u32 e0 = threadIdx.x; e0 < total_raw; e0 += NT) {
    const u64 e = lists[src_s[e0]];
    if ((int)(u32)(e >> 32) >= T) {
      const u32 slot = atomicAdd(&kept_s, 1u);
      if (slot < (u32)S8R_CAP) ids_s[slot] = (u32)e;
    }
  }
  __syncthreads();
  if (a.s8_reset && (int)threadIdx.x < a.s8_slots)   // (the table has been read: the next call finds it empty)
    const_cast<int*>(a.s8_G)[(q * a.s8_slots + threadIdx.x) * S8_SLOT_STRIDE] = S8_EMPTY;
  const u32 cnt_raw = kept_s;
  const u32 cnt = cnt_raw < (u32)S8R_CAP ? cnt_raw : (u32)S8R_CAP;
  // ---- exact fp32 distances: G lanes per row, U rows per wavefront in flight, every piece of a row issued at once
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  for (u32 c0 = wave * RPW * U; c0 < cnt; c0 += NW * RPW * U) {
    const float* rp[U];
    u32 id[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32 ci = c0 + u * RPW + g;
      ok[u] = ci < cnt;
      id[u] = ids_s[ok[u] ? ci : cnt - 1];
      rp[u] = a.rows + (int64_t)id[u] * dim;
    }
    float acc[U][1];
    row_dists<U, 1, VEC4, 3>(rp, smem, qstride, dim, a.metric, G, acc);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u] && t == 0) {
        const float dist = finish_dist(a.metric, acc[u][0]);
        keys_s[c0 + u * RPW + g] = row_visible(a.f, id[u], dist) ? make_key(dist, id[u]) : KEY_EMPTY;   // (rows the filter hides sort last)
      }
    }
  }
  if (threadIdx.x < 8 && cnt + threadIdx.x < ((cnt + 7u) & ~7u)) keys_s[cnt + threadIdx.x] = KEY_EMPTY;
  __syncthreads();
  // ---- the k best by rank (8 keys per step, no early exit: the LDS reads of a step are independent - one latency per 8 keys, not per key)
  for (u32 ci = threadIdx.x; ci < cnt; ci += NT) {
    const u64 mine = keys_s[ci];
    if (mine == KEY_EMPTY) continue;
    int rank = 0;
    for (u32 j0 = 0; j0 < cnt; j0 += 8) {
#pragma unroll
      for (u32 jj = 0; jj < 8; ++jj) {
        const u64 o = keys_s[j0 + jj];   // (padded with KEY_EMPTY up to a multiple of 8)
        rank += (o < mine || (o == mine && j0 + jj < ci)) ? 1 : 0;
      }
    }
    if (rank < a.k) best_s[rank] = mine;
  }
  __syncthreads();
  if (wave != 0) return;
  // ---- the query's result and the call's bookkeeping (rerank_kernel's, KPL = 1)
  const u64 key = lane < a.k ? best_s[lane] : KEY_EMPTY;
  if (lane < a.k) a.run_keys[q * a.k + lane] = key;
  const bool valid = lane < a.k && key != KEY_EMPTY;
  if (a.fin_ids) {
    if (lane < a.k) {
      a.fin_ids[q * a.k + lane] = valid ? (int64_t)key_id(key) * a.fin_stride + a.fin_base : -1;
      a.fin_dist[q * a.k + lane] = valid ? key_dist(key) : __builtin_inff();
    }
    const int c = __popcll(__ballot(valid));
    if (a.fin_counts && lane == 0) a.fin_counts[q] = c;
  }
  if (lane == 0) {
    if (lost_s) atomicAdd(a.overflow, 1u);   // a wavefront's list lost entries: the caller repeats the batch on the staged chain
    if (a.fuse & 1) {
      if (cnt_raw > (u32)S8R_CAP || cnt_raw > (u32)a.cap) atomicAdd(a.overflow, 1u);
      atomicAdd(a.total, (unsigned long long)cnt);
    }
    a.cand_count[q] = 0;
    if (a.pub) {   // (lane 0 of wavefront 0 issued every one of this block's counter updates above)
      __threadfence();
      if (atomicAdd(a.pub_ticket, 1u) == gridDim.x - 1) {
        const u32 ov = atomicAdd(a.overflow, 0u);
        const unsigned long long tot = atomicAdd(a.total, 0ull);
        a.pub[0] = ov;
        a.pub[2] = (u32)tot;
        a.pub[3] = (u32)(tot >> 32);
        __threadfence_system();
        if (a.s8_reset) {   // (every block has finished with the counters: each took its ticket after its last update)
          *a.overflow = 0u;
          *a.total = 0ull;
          *a.pub_ticket = 0u;
        }
      }
    }
  }
}

// the same over several workgroups per query (RerankArgs::parts): blockIdx.x = q * parts + part
template <int KPL, bool VEC4>
__global__ __launch_bounds__(256) void rerank_split_kernel(RerankArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [qstride] query, then 4 lists
  constexpr int NW = 4, NT = 256;
  __shared__ u32 last_flag;
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  u64* sh = reinterpret_cast<u64*>(smem + qstride);
  const int64_t q = blockIdx.x / a.parts;
  const int part = (int)(blockIdx.x % a.parts);
  for (int i = threadIdx.x; i < qstride; i += NT) smem[i] = i < dim ? a.queries[q * dim + i] : 0.f;
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  constexpr int U = 4;
  const u32 cnt_raw = a.cand_count[q];
  u32 cnt = cnt_raw;
  if (cnt > (u32)a.cap) cnt = (u32)a.cap;
  const u32* cand = a.cand + q * (int64_t)a.cap;
  // this workgroup's share: candidates [c_lo, c_hi)
  const u32 share = (cnt + (u32)a.parts - 1) / (u32)a.parts;
  const u32 c_lo = (u32)part * share;
  const u32 c_hi = c_lo + share < cnt ? c_lo + share : cnt;

  WaveTopK<KPL> L[1];
  u64 thr[1];
  L[0].init();
  thr[0] = a.run_keys[q * a.k + (a.k - 1)];   // (the running k-th best bounds every share)
  for (u32 c0 = c_lo + wave * RPW * U; c0 < c_hi; c0 += NW * RPW * U) {
    const float* rp[U];
    u32 id[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32 ci = c0 + u * RPW + g;
      ok[u] = ci < c_hi;
      id[u] = cand[ok[u] ? ci : c_hi - 1];
      rp[u] = a.rows + (int64_t)id[u] * dim;
    }
    float acc[U][1];
    row_dists<U, 1, VEC4>(rp, smem, qstride, dim, a.metric, G, acc);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u64 key = make_key(finish_dist(a.metric, acc[u][0]), id[u]);
      offer<1, KPL>(L, thr, 0, key, ok[u] && t == 0, a.f, a.k, true);
    }
  }
  L[0].store(sh + wave * (KPL * 64), a.k);
  __syncthreads();
  FilterSpec nof = {nullptr, nullptr, 0, 0, 0, 0};
  if (wave == 0) {   // this workgroup's k best -> its slot
    for (int w = 1; w < NW; ++w)
      for (int e0 = 0; e0 < a.k; e0 += 64) {
        const int e = e0 + lane;
        const u64 key = e < a.k ? sh[w * (KPL * 64) + e] : KEY_EMPTY;
        offer<1, KPL>(L, thr, 0, key, key != KEY_EMPTY, nof, a.k, true);
      }
    L[0].store(a.part_keys + (q * a.parts + part) * (int64_t)a.k, a.k);
  }
  // publish, and find out whether this workgroup is the last of its query (agent-scope release on the stores above, acquire for the
  // other workgroups' slots: they may have been written through another XCD's L2)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const u32 ticket = __hip_atomic_fetch_add(&a.part_done[q], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = ticket == (u32)a.parts - 1 ? 1u : 0u;
    __threadfence();
  }
  __syncthreads();
  if (!last_flag) return;
  if (wave == 0) {
    L[0].load(a.run_keys + q * a.k, a.k);
    thr[0] = L[0].entry(a.k - 1);
    for (int p = 0; p < a.parts; ++p)
      for (int e0 = 0; e0 < a.k; e0 += 64) {
        const int e = e0 + lane;
        const u64 key = e < a.k ? __builtin_nontemporal_load(a.part_keys + (q * a.parts + p) * (int64_t)a.k + e) : KEY_EMPTY;
        offer<1, KPL>(L, thr, 0, key, key != KEY_EMPTY, nof, a.k, true);
      }
    L[0].store(a.run_keys + q * a.k, a.k);
    if (lane == 0) a.part_done[q] = 0;
    if (a.fin_ids) {
      int c = 0;
#pragma unroll
      for (int r = 0; r < KPL; ++r) {
        const int e = r * 64 + lane;
        const u64 key = L[0].key[r];
        const bool valid = e < a.k && key != KEY_EMPTY;
        if (e < a.k) {
          a.fin_ids[q * a.k + e] = valid ? (int64_t)key_id(key) * a.fin_stride + a.fin_base : -1;
          a.fin_dist[q * a.k + e] = valid ? key_dist(key) : __builtin_inff();
        }
        c += __popcll(__ballot(valid));
      }
      if (a.fin_counts && lane == 0) a.fin_counts[q] = c;
    }
    if (a.fuse) {
      const u64 kth = L[0].entry(a.k - 1);
      if (a.gsync && q == 0)
        for (int i = lane; i < 256; i += 64) a.gsync[i] = 0;
      if (lane == 0) {
        if (a.fuse & 1) {
          if (cnt_raw > (u32)a.cap) atomicAdd(a.overflow, 1u);
          atomicAdd(a.total, (unsigned long long)cnt);
        }
        a.cand_count[q] = 0;
        if (a.T_next) {
          const float* qs = a.qstat + q * 4;
          if (a.bits == 8) {
            static_cast<int*>(a.T_next)[q] = kth == KEY_EMPTY ? -(1 << 30) : stage_threshold8(key_dist(kth), qs, a.scal, a.metric, a.u, a.slack, 0);
          } else {
            static_cast<float*>(a.T_next)[q] = kth == KEY_EMPTY ? 3.0e38f : stage_threshold16(key_dist(kth), qs, a.scal, a.metric, a.slack, 0);
          }
        }
      }
    }
  }
}

void launch_rerank(const RerankArgs& a, hipStream_t s) {
  if (a.nq <= 0) return;
  if (a.s8_G && a.s8_fast && pick_kpl(a.k) == 1 && a.s8_waves <= S8_MAX_WAVES && !a.T_next && !a.gsync) {   // a one-pass call's tail
    const bool v4 = (a.dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.rows) & 15) == 0);
    const size_t sm = (size_t)((a.dim + 3) & ~3) * sizeof(float) + (size_t)S8R_CAP * sizeof(u64);
    if (v4) hipLaunchKernelGGL((s8_rerank_kernel<true>), dim3((unsigned)a.nq), dim3(1024), sm, s, a);
    else hipLaunchKernelGGL((s8_rerank_kernel<false>), dim3((unsigned)a.nq), dim3(1024), sm, s, a);
    return;
  }
  if (a.parts > 1 && a.part_keys && a.part_done && pick_kpl(a.k) <= 2) {
    const bool v4 = (a.dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.rows) & 15) == 0);
    const int kp = pick_kpl(a.k);
    const size_t sm = (size_t)((a.dim + 3) & ~3) * sizeof(float) + (size_t)4 * kp * 64 * sizeof(u64);
    const dim3 grid((unsigned)(a.nq * a.parts));
    if (kp == 1) {
      if (v4) hipLaunchKernelGGL((rerank_split_kernel<1, true>), grid, dim3(256), sm, s, a);
      else hipLaunchKernelGGL((rerank_split_kernel<1, false>), grid, dim3(256), sm, s, a);
    } else {
      if (v4) hipLaunchKernelGGL((rerank_split_kernel<2, true>), grid, dim3(256), sm, s, a);
      else hipLaunchKernelGGL((rerank_split_kernel<2, false>), grid, dim3(256), sm, s, a);
    }
    return;
  }
  const bool vec4 = (a.dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.rows) & 15) == 0);
  const int kpl = pick_kpl(a.k);
  const int nw = (a.nq <= 64 && kpl <= 4) ? 16 : 4;
  const size_t shm = (size_t)((a.dim + 3) & ~3) * sizeof(float) + (size_t)nw * kpl * 64 * sizeof(u64);
#define EPS_CASE(KPL_)                                                                                               \
  if (kpl == KPL_) {                                                                                                 \
    if (nw == 16) {                                                                                                  \
      if (vec4) hipLaunchKernelGGL((rerank_kernel<KPL_, true, 16>), dim3((unsigned)a.nq), dim3(1024), shm, s, a);    \
      else hipLaunchKernelGGL((rerank_kernel<KPL_, false, 16>), dim3((unsigned)a.nq), dim3(1024), shm, s, a);        \
    } else {                                                                                                         \
      if (vec4) hipLaunchKernelGGL((rerank_kernel<KPL_, true, 4>), dim3((unsigned)a.nq), dim3(256), shm, s, a);      \
      else hipLaunchKernelGGL((rerank_kernel<KPL_, false, 4>), dim3((unsigned)a.nq), dim3(256), shm, s, a);          \
    }                                                                                                                \
    return;                                                                                                          \
  }
  EPS_CASE(1) EPS_CASE(2) EPS_CASE(4) EPS_CASE(8) EPS_CASE(16)
#undef EPS_CASE
}

// ------------------------------------------------------------------------------------------------
__global__ void finalize_kernel(const u64* run_keys, int64_t nq, int k, int64_t id_base, int64_t id_stride,
                                int64_t* ids, float* dist, int32_t* counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * k) return;
  const u64 key = run_keys[i];
  const bool valid = key != KEY_EMPTY;
  ids[i] = valid ? (int64_t)key_id(key) * id_stride + id_base : -1;
  dist[i] = valid ? key_dist(key) : __builtin_inff();
  if (counts && (i % k) == 0) {
    int c = 0;
    for (int e = 0; e < k; ++e) c += run_keys[i + e] != KEY_EMPTY;
    counts[i / k] = c;
  }
}
void launch_finalize(const u64* run_keys, int64_t nq, int k, int64_t id_base, int64_t id_stride, int64_t* ids,
                     float* dist, int32_t* counts, hipStream_t s) {
  const int64_t n = nq * k;
  if (n <= 0) return;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, run_keys, nq, k, id_base,
                     id_stride, ids, dist, counts);
}

__global__ void fill_u64_kernel(u64* p, int64_t n, u64 v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill_u64(u64* p, int64_t n, u64 v, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

// ------------------------------------------------------------------------------------------------
// Normalize (db/vector.cpp:60-69; insert path table_segment_mvp.cpp:574-587). One wavefront per row.
__global__ __launch_bounds__(256) void normalize_kernel(float* rows, int64_t n, int dim, int only_if_nonzero) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = lane_id();
  float* p = rows + r * dim;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) s = fmaf(p[c], p[c], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (only_if_nonzero && !(s > 1e-10f)) return;
  const float nrm = sqrtf(s);
  for (int c = lane; c < dim; c += 64) p[c] = p[c] / nrm;
}
void launch_normalize(float* rows, int64_t n, int dim, bool only_if_nonzero, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, rows, n, dim,
                     only_if_nonzero ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// k-way merge of per-shard sorted lists by (dist, id) — what every rank does after the all-gather.
// shard s's lists start `stride_bytes` after shard s-1's (0 = densely packed [shards][nq][k] arrays)
__global__ void merge_shards_kernel(const float* dist, const int64_t* ids, int shards, int64_t nq, int k,
                                    float* out_dist, int64_t* out_ids, int64_t stride_bytes, int32_t* out_counts) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  int head[16];
  int found = 0;
  for (int s = 0; s < shards; ++s) head[s] = 0;
  for (int e = 0; e < k; ++e) {
    int best = -1;
    float bd = 0.f;
    int64_t bi = 0;
    for (int s = 0; s < shards; ++s) {
      if (head[s] >= k) continue;
      const int64_t o = q * k + head[s];
      const int64_t id = *reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(ids + o) + (stride_bytes ? stride_bytes : nq * k * 8) * s);
      if (id < 0) { head[s] = k; continue; }
      const float d = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dist + o) + (stride_bytes ? stride_bytes : nq * k * 4) * s);
      if (best < 0 || d < bd || (d == bd && id < bi)) {
        best = s; bd = d; bi = id;
      }
    }
    if (best < 0) {
      out_dist[q * k + e] = __builtin_inff();
      out_ids[q * k + e] = -1;
    } else {
      out_dist[q * k + e] = bd;
      out_ids[q * k + e] = bi;
      head[best]++;
      ++found;
    }
  }
  if (out_counts) out_counts[q] = found;
}
void launch_merge_shards(const float* dist, const int64_t* ids, int shards, int64_t nq, int k, float* out_dist,
                         int64_t* out_ids, hipStream_t s, int64_t stride_bytes, int32_t* out_counts) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(merge_shards_kernel, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, s, dist, ids, shards, nq, k,
                     out_dist, out_ids, stride_bytes, out_counts);
}

}  // namespace eps
