// Search-time traversal kernel of libepsilla_gfx950: VecSearchExecutor::SearchImpl (reference:
// engine/db/execution/vec_search_executor.cpp:518-715) with its T workers, local queues, PickTopMToWorkers
// (:328-356), ExpandOneCandidate (:384-444), AddIntoQueue (:75-117), the GlobalSyncInterval and
// MergeAllQueuesToMaster / MergeTwoQueuesInto1stQueueSeqFixed (:297-326, :150-217), for any SearchQueueSize.
//
// One workgroup owns one "slot" (visited bitmap + undo log + queue scratch) and walks over the queries
// q = slot, slot + #slots, ...   Per query:
//   * queue storage is the reference's set_L_ layout: T-1 worker queues of Lq keys, then the master queue of L
//     keys (vec_search_executor.cpp:59-68), in LDS when it fits next to the work arrays, otherwise in HBM
//     (QGLOBAL; L up to 2^20);  a key is ord(dist) << 32 | id << 1 | checked, so (key >> 1) order is
//     Candidate::operator< (candidate.hpp:16-22);
//   * the T workers of a round run in LOCKSTEP: step i of every worker before step i+1 of any worker; inside one
//     step all T expansions are in flight together (T adjacency lists gathered, visited-tested with one atomicOr
//     each, the surviving rows streamed by every wavefront of the workgroup) and conflicts on the visited set
//     are resolved in worker order, master last.  This is one of the interleavings the reference's racing
//     OpenMP workers may produce, it is deterministic, and oracle/epsilla_oracle.c restates it
//     (eo_set_schedule(1)) - so parity is bit-exact for every T, not only T = 1;
//   * a worker's <= deg new candidates of a step are inserted into its queue AT ONCE (rank sort + in-place merge,
//     processed from the tail in register-sized chunks).  Inserting the survivors of one expansion together
//     yields the same queue, and the same lowest insert position, as the reference's one-by-one AddIntoQueue
//     (DESIGN.md "traversal equivalence");
//   * visited set = N-bit bitmap per slot that is kept all-zero BETWEEN queries: the ids whose bit a query set
//     are appended to an undo log and exactly those words are cleared at the end (the reference's
//     is_visited.clear()/resize(n), :711-714, is O(N) per query; here it is O(evaluations)).  If the log
//     overflows the slot's bitmap is cleared wholesale by the workgroup.
#pragma once
#include "kernels.hpp"

namespace eps {

struct Trv2Args {
  const float* rows;
  int dim;
  int metric;
  const int64_t* off;   // CSR offsets, or null when fixed_deg > 0
  const u32* nbr;       // CSR neighbours, or [n][fixed_deg] lists padded with 0xFFFFFFFF
  int fixed_deg;
  int dp;               // edge slots per worker and step (>= the longest adjacency list)
  int hslots;           // slots of the per-step ownership table: power of two >= 2*T*dp, 0 when T == 1
  const u32* init_ids;  // [L]
  const float* queries;
  int64_t nq;
  int L, Lq, Lp2;       // master queue length, worker queue capacity, next power of two >= L
  int T, I;             // workers (IntraQueryThreads), expansions per worker and round (GlobalSyncInterval)
  int64_t qtot;         // keys of queue storage per slot = (T-1)*Lq + Lp2
  u64* qglobal;         // QGLOBAL: [slots][qtot]
  int* auxglobal;       // QGLOBAL: [slots][2*Lq] merge scratch
  u32* visited;         // [slots][words], all-zero on entry and on exit
  int64_t words;
  u32* vlog;            // [slots][vcap] undo log
  int vcap;
  // r5, visited set by GENERATION STAMP instead of bitmap + undo log (where HBM allows: 4 bytes per node and slot): gens[slot][node] holds the stamp of
  // the last query of this slot that reached the node; a query's stamp is gen_base + its ordinal in the slot, so "visited" is `old stamp == mine`,
  // test-and-set is ONE atomicMax, and nothing is reset at the end of a walk (the reset's ~evaluations scattered writes were 5 % of the launch at
  // 10M x 768: profiles/r5_traverse_lab_10M_proxy.txt session 4).  null: the bitmap.
  u32* gens;
  int64_t gens_n;       // nodes per slot
  u32 gen_base;         // stamps of this launch: gen_base + 1 + (q - slot) / slots
  u64* out_queue;       // [nq][L] final master queues
  unsigned long long* counters;  // [0] distance evaluations, [1] expansions, [2] steps, [3] rounds, [4] fp32 rows read in step d
  unsigned long long* prof;      // optional [16]: shader-clock ticks per phase summed over the workgroups (EPS_TRV_PROF)
  // filtered traversal (eps_search_params::filter_in_traversal): every distance the search evaluates is also logged as a plain
  // (dist, id) key, so that the caller can pick the k best VISIBLE rows among ALL evaluated nodes instead of among the final queue
  u64* elog;                     // [nq][elog_cap] or null
  u32* elog_cnt;                 // [nq] evaluations of the query (may exceed elog_cap: only the first elog_cap are stored)
  int elog_cap;
  // exact lower-bound prefilter of step d on the table's 8-bit mirror (mfma_filter.hip: one grid per table, integer dot product,
  // the row constant folded into acc0): a neighbour whose 768-byte mirror row PROVES `dist > bound` is dropped without its fp32
  // row ever being read; every other neighbour goes through the unchanged fp32 evaluation.  Results are bit-identical with or
  // without it (the proof is the flat engine's: device_common.hpp stage_threshold8).  x8 == null: off.
  const signed char* x8;         // [n_pad][d_pad8]
  const int* acc0;               // [n_pad]
  // r5: the row constants of a node's neighbours stored WITH its adjacency list ([n][fixed_deg], same layout as nbr): an expansion reads them
  // in the same access as the list, and the prefilter finds them in LDS instead of gathering 4 scattered bytes per evaluation (what bounds
  // the kernel is the number of scattered locations per evaluation, profiles/r5_traverse_lab_10M_proxy.txt).  null: gather acc0[id].
  const int* nbr_acc0;
  const float* scal8;            // the mirror's table-wide bounds (residual, |xh|, |x|^2, -, |R|)
  const signed char* q8;         // [nq][d_pad8] the queries on the same grid
  const float* qstat8;           // [nq][4] |q|^2, |q|, |q - qh|, C + c
  int d_pad8;
  int cols8;                     // leading bytes of a mirror row that carry values (Quant8View::cols8: dim, or dim rounded up to 256 in the rotated frame)
  float u8, slack8;
};

#ifndef EPS_TRV_U
#define EPS_TRV_U 4   // rows in flight per lane group in the distance phases
#endif
#ifndef EPS_TRV_UPF
#define EPS_TRV_UPF 3  // prefilter form of the kernel: fp32 rows in flight per lane group (seeds, survivors of step d0) ...
#endif
#ifndef EPS_TRV_NL
#define EPS_TRV_NL 3   // ... and 16-byte pieces of each (row_dists, device_common.hpp).  r5: 3 x 3 (4 x 1 until r4; profiles/r5_traverse_lab_10M_proxy.txt)
#endif
#ifndef EPS_TRV_U8
#define EPS_TRV_U8 3   // prefilter: mirror rows in flight per lane group (2 until r4) ...
#endif
#ifndef EPS_TRV_NL8
#define EPS_TRV_NL8 3  // ... and 16-byte pieces of each per lane
#endif
constexpr int TRV2_SB = 4096;       // keys of the LDS staging block of the QGLOBAL bitonic sort
constexpr int TRV2_MAXT = 128;      // the reference's limit for IntraQueryThreads (config/config.hpp:29)
// ints of scalar scratch: 16 scalars, nine [TS] per-worker arrays, [16] per-wave counts, [TS + 16] edge offsets; TS = T rounded up to 16
__host__ __device__ inline int trv2_tstride(int T) { return T <= 16 ? 16 : (T + 15) & ~15; }
__host__ __device__ inline int trv2_sh_ints(int T) { return 48 + 10 * trv2_tstride(T); }
// bytes of the prefilter's LDS block: the query's four statistics, the query on the 8-bit grid, alignment spare
__host__ __device__ inline int trv2_q8len(int dim) { return (dim + 15) & ~15; }
__host__ __device__ inline int trv2_pf_bytes(int dim) { return 16 + 16 + trv2_q8len(dim); }
constexpr u32 TRV2_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ u64 q2key(float d, u32 id) { return ((u64)f2ord(d + 0.0f) << 32) | ((u64)id << 1); }

// number of keys in q[0..n) ordered before x (both compared without the checked bit)
__device__ __forceinline__ int lower_bound_q(const u64* q, int n, u64 x) {
  int lo = 0, hi = n;
  const u64 xs = x >> 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((q[mid] >> 1) < xs) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// compare-exchange network passes of a bitonic sort over buf[0..n) (n a power of two) for strides
// stride_hi, stride_hi/2, ..., 1; `gbase` is the global index of buf[0], `size` the bitonic block size.
template <int NT>
__device__ __forceinline__ void bitonic_passes(u64* buf, int n, int64_t gbase, int64_t size, int stride_hi) {
  for (int stride = stride_hi; stride > 0; stride >>= 1) {
    for (int i = threadIdx.x; i < (n >> 1); i += NT) {
      const int lo = ((i / stride) * (stride << 1)) + (i % stride);
      const int hi = lo + stride;
      const bool up = (((gbase + lo) & size) == 0);
      const u64 x = buf[lo], y = buf[hi];
      if ((x > y) == up) {
        buf[lo] = y;
        buf[hi] = x;
      }
    }
    __syncthreads();
  }
}

// PF: with the 8-bit prefilter (step d0).  Compiled for 4 wavefronts per SIMD (<= 128 VGPRs, the occupancy the host side plans
// with; a handful of dwords spill outside the loops); the form without it needs no such cap (109-117 VGPRs, nothing spilled).
template <bool VEC4, int NW, bool QGLOBAL, bool PF>
__global__ __launch_bounds__(NW * 64, PF ? 4 : 1) void traverse2_kernel(Trv2Args a) {
  constexpr int NT = NW * 64;
  constexpr int R = 4;           // queue elements per thread and merge chunk
  constexpr int C = NT * R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  const int T = a.T, L = a.L, Lq = a.Lq, DP = a.dp;
  const int ecap = T * DP;
  float* sq = reinterpret_cast<float*>(smem_raw);                          // [qstride]
  u64* qlds = reinterpret_cast<u64*>(sq + qstride);                         // queues (or the sort staging block)
  u64* newk = qlds + (QGLOBAL ? (int64_t)TRV2_SB : a.qtot);                 // [ecap] keys of this step, per worker segment
  u64* sorted = newk + ecap;                                                // [ecap]
  u32* eid = reinterpret_cast<u32*>(sorted + ecap);                         // [ecap] neighbour id per edge slot
  u32* eraw = eid + ecap;                                                   // [ecap] 1 = this slot's atomicOr set the bit
  u32* work = eraw + ecap;                                                  // [ecap] ids to evaluate, per worker segment
  int* npos = reinterpret_cast<int*>(work + ecap);                          // [ecap] insert positions
  int* eacc = reinterpret_cast<int*>(newk);                                 // PF, nbr_acc0: [ecap] row constant per edge slot (steps b-c; `newk` is written from step d on)
  int* wacc = eacc + ecap;                                                  //               [ecap] ... per entry of `work` (steps c-d0)
  u32* wwk = reinterpret_cast<u32*>(sorted);                                // PF: [ecap] worker of every entry of `work` (steps c-d0; `sorted` is written in step e)
  const int H = a.hslots;                                                   // ownership table slots (power of two; 0 when T == 1)
  u32* hid = reinterpret_cast<u32*>(npos + ecap);                           // [H] node id, TRV2_NONE = empty
  int* hmin = reinterpret_cast<int*>(hid + H);                              // [H] lowest edge slot that met the node this step
  int* auxl = hmin + H;                                                     // !QGLOBAL: [2*Lq] MergeAll scratch (T > 1)
  int* sh = auxl + ((QGLOBAL || T == 1) ? 0 : 2 * Lq);                      // [trv2_sh_ints(T)]
  // sh[0] scratch (first found / pmin), sh[1] unchecked count, sh[2] undo-log fill, sh[3] any worker selected,
  // sh[4] running prefix, sh[5] non-duplicate count, sh[6] r of MergeAll, sh[7] log overflow, sh[8] evaluations logged (elog),
  // sh[9] prefilter threshold of this step
  const int TS = trv2_tstride(T);
  const int SH = trv2_sh_ints(T);
  int* s_kuc = sh + 16;            // [T] first possibly-unchecked position per queue
  int* s_size = s_kuc + TS;        // [T] queue sizes
  int* s_its = s_size + TS;        // [T] expansions done this round
  int* s_sel = s_its + TS;         // [T] node selected this step, or -1
  int* s_deg = s_sel + TS;         // [T] its degree
  int* s_wcnt = s_deg + TS;        // [T] ids to evaluate
  int* s_nnew = s_wcnt + TS;       // [T] keys that passed the bound
  int* s_wave = s_nnew + TS;       // [NW <= 16] per-wave counts
  int* s_selpos = s_wave + 16;     // [T] queue position of the selected candidate
  int* s_pcnt = s_selpos + TS;     // [T] ids that passed the 8-bit prefilter
  int* s_eoff = s_pcnt + TS;       // [T+1] first edge slot of every worker in this step
  // prefilter block (only present in the launch's LDS size when a.x8): statistics of the query on the table's grid, then the query
  constexpr bool pf = PF;   // (a.x8 != null)
  const size_t pf_off = ((size_t)(reinterpret_cast<unsigned char*>(sh + SH) - smem_raw) + 15) & ~(size_t)15;
  float* qst = reinterpret_cast<float*>(smem_raw + pf_off);                // [4] |q|^2, |q|, |q - qh|, C + c
  signed char* sq8 = reinterpret_cast<signed char*>(qst + 4);               // [q8len]
  const int q8len = trv2_q8len(a.cols8 > dim ? a.cols8 : dim);
  constexpr int U8 = EPS_TRV_U8, NL8 = EPS_TRV_NL8;
  int G8 = 4;
  while (G8 < 64 && G8 * 16 * NL8 < q8len) G8 <<= 1;                        // lanes per mirror row (NL8 pieces of 16 bytes each)
  const int RPW8 = 64 / G8;

  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = tid >> 6;
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  constexpr int U = PF ? EPS_TRV_UPF : EPS_TRV_U;
  const int64_t slot = blockIdx.x;
  u32* vis = a.gens ? nullptr : a.visited + slot * a.words;
  u32* vlog = a.gens ? nullptr : a.vlog + slot * (int64_t)a.vcap;
  u32* gen = a.gens ? a.gens + slot * a.gens_n : nullptr;
  u64* qbase = QGLOBAL ? a.qglobal + slot * a.qtot : qlds;
  int* aux = QGLOBAL ? a.auxglobal + slot * (int64_t)(2 * Lq) : auxl;
  u64* master = qbase + (int64_t)(T - 1) * Lq;
  unsigned long long evals = 0, expansions = 0, steps = 0, rounds = 0, fetched = 0;
  // phase clocks (wave-uniform scalar reads; thread 0 adds every lap to the launch's totals - only with EPS_TRV_PROF, no
  // registers held otherwise): 0 seeds+sort, 1 scatter, 2 select, 3 gather+visited, 4 dedupe, 5 distances, 6 rank sort,
  // 7 queue insert, 8 merge-all, 9 results+reset
  const bool prof = a.prof != nullptr;
  long long tprev = prof ? clock64() : 0;
#define TRV2_LAP(i)                 \
  if (prof) {                       \
    const long long tn = clock64(); \
    if (tid == 0) atomicAdd(&a.prof[i], (unsigned long long)(tn - tprev)); \
    tprev = tn;                     \
  }

  for (int64_t q = slot; q < a.nq; q += gridDim.x) {
    // ------------------------------------------------------------------ InitializeSetLPara (:446-485)
    for (int i = tid; i < qstride; i += NT) sq[i] = i < dim ? a.queries[q * dim + i] : 0.f;
    if (pf) {   // the query on the table's grid and its statistics (query_prep8_kernel, launched ahead of this kernel)
      for (int i = tid; i < (q8len >> 2); i += NT) reinterpret_cast<u32*>(sq8)[i] = reinterpret_cast<const u32*>(a.q8 + q * a.d_pad8)[i];
      if (tid < 4) qst[tid] = a.qstat8[q * 4 + tid];
    }
    for (int i = tid; i < SH; i += NT) sh[i] = 0;
    for (int i = tid; i < H; i += NT) {
      hid[i] = TRV2_NONE;
      hmin[i] = 0x7FFFFFFF;
    }
    for (int i = L + tid; i < a.Lp2; i += NT) master[i] = KEY_EMPTY;
    const u32 stamp = a.gen_base + 1u + (u32)((q - slot) / gridDim.x);
    for (int i = tid; i < L; i += NT) {
      const u32 id = a.init_ids[i];
      if (gen) atomicMax(&gen[id], stamp); else atomicOr(&vis[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    for (int c0 = wave * RPW * U; c0 < L; c0 += NW * RPW * U) {
      const float* rp[U];
      u32 id[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ci = c0 + u * RPW + g;
        ok[u] = ci < L;
        id[u] = a.init_ids[ok[u] ? ci : L - 1];
        rp[u] = a.rows + (int64_t)id[u] * dim;
      }
      float acc[U][1];
      row_dists<U, 1, VEC4, (PF ? EPS_TRV_NL : 1)>(rp, sq, qstride, dim, a.metric, G, acc);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[u] && t == 0) {
          const float dseed = finish_dist(a.metric, acc[u][0]);
          master[c0 + u * RPW + g] = q2key(dseed, id[u]);
          if (a.elog) {
            const int lp = atomicAdd(&sh[8], 1);
            if (lp < a.elog_cap) a.elog[q * (int64_t)a.elog_cap + lp] = make_key(dseed, id[u]);
          }
        }
    }
    evals += L;
    __syncthreads();
    // std::sort(master) (:478): bitonic over Lp2 keys (padding sorts last)
    if (!QGLOBAL) {
      for (int64_t size = 2; size <= a.Lp2; size <<= 1) bitonic_passes<NT>(master, a.Lp2, 0, size, (int)(size >> 1));
    } else {
      const int nb = a.Lp2 < TRV2_SB ? a.Lp2 : TRV2_SB;   // keys per staging block
      for (int64_t b0 = 0; b0 < a.Lp2; b0 += nb) {        // every block sorted on its own, directions as in the full network
        for (int i = tid; i < nb; i += NT) qlds[i] = master[b0 + i];
        __syncthreads();
        for (int64_t size = 2; size <= nb; size <<= 1) bitonic_passes<NT>(qlds, nb, b0, size, (int)(size >> 1));
        for (int i = tid; i < nb; i += NT) master[b0 + i] = qlds[i];
        __syncthreads();
      }
      for (int64_t size = (int64_t)nb << 1; size <= a.Lp2; size <<= 1) {
        for (int64_t stride = size >> 1; stride >= nb; stride >>= 1) {   // strides that span blocks: in HBM
          for (int64_t i = tid; i < (a.Lp2 >> 1); i += NT) {
            const int64_t lo = ((i / stride) * (stride << 1)) + (i % stride);
            const int64_t hi = lo + stride;
            const bool up = ((lo & size) == 0);
            const u64 x = master[lo], y = master[hi];
            if ((x > y) == up) {
              master[lo] = y;
              master[hi] = x;
            }
          }
          __syncthreads();
        }
        for (int64_t b0 = 0; b0 < a.Lp2; b0 += nb) {                     // the rest of this size inside each block: in LDS
          for (int i = tid; i < nb; i += NT) qlds[i] = master[b0 + i];
          __syncthreads();
          bitonic_passes<NT>(qlds, nb, b0, size, nb >> 1);
          for (int i = tid; i < nb; i += NT) master[b0 + i] = qlds[i];
          __syncthreads();
        }
      }
    }
    if (tid == 0) s_size[T - 1] = L;
    __syncthreads();
    TRV2_LAP(0)

    // ------------------------------------------------------------------ rounds
    // round 0 = the reference's one sequential expansion on the master queue (:556-593); every later round =
    // PickTopMToWorkers, <= I lockstep steps, MergeAllQueuesToMaster (:601-698)
    for (int round = 0;; ++round) {
      if (round > 0) {
        // ---- PickTopMToWorkers (:328-356): the i-th unchecked master candidate at or after k goes to worker i mod T
        // (copied unchecked, the master copy marked checked); every T-th stays with the master.  The reference stops
        // when worker 0's queue is full, i.e. after unchecked candidate number (Lq-1)*T.
        const int kmaster = s_kuc[T - 1];
        const int imax = (Lq - 1) * T;
        if (tid == 0) sh[4] = 0;
        __syncthreads();
        for (int base = kmaster; base < L; base += NT) {
          const int p = base + tid;
          const bool un = p < L && !(master[p] & 1ull);
          const u64 m = __ballot(un);
          if (lane == 0) s_wave[wave] = __popcll(m);
          __syncthreads();
          int i = sh[4];
          for (int w2 = 0; w2 < wave; ++w2) i += s_wave[w2];
          i += __popcll(m & ((1ull << lane) - 1ull));
          if (un && i <= imax) {
            const int dest = i % T;
            if (dest != T - 1) {
              qbase[(int64_t)dest * Lq + i / T] = master[p];
              master[p] |= 1ull;
            }
          }
          __syncthreads();
          if (tid == 0) {
            int tot = sh[4];
            for (int w2 = 0; w2 < NW; ++w2) tot += s_wave[w2];
            sh[4] = tot;
          }
          __syncthreads();
        }
        const int U_ = sh[4];   // unchecked candidates seen
        if (U_ == 0) break;     // uniform: no unchecked candidate left -> the search is over (:609-611)
        if (tid < T) {
          const int last = (U_ - 1 < imax) ? U_ - 1 : imax;   // highest unchecked index that was distributed
          if (tid < T - 1) s_size[tid] = last >= tid ? (last - tid) / T + 1 : 0;
          s_kuc[tid] = tid == T - 1 ? kmaster : 0;
          s_its[tid] = 0;
        }
        __syncthreads();
      }
      const int I_round = round == 0 ? 1 : a.I;
      ++rounds;
      TRV2_LAP(1)

      // ---- lockstep steps
      while (true) {
        // a. every worker with iterations left takes its first unchecked candidate at or after k_uc (:632-672).  One
        //    wavefront per worker (ballot over 64 queue positions at a time), the workers side by side.
        // (sh[3] "somebody selected" and sh[10] "nodes to evaluate" are zero here: zeroed with the query's scratch, and again at the end of every step -
        // r5: a barrier of this workgroup costs ~0.4 us among 16 wavefronts per CU, a step had 17 of them, a T = 1 step lasts 17 us)
        for (int w = wave; w < T; w += NW) {
          u64* qw = qbase + (int64_t)w * Lq;
          const int size = s_size[w];
          const bool can = s_its[w] < I_round;
          int found = -1;
          if (can) {
            for (int base = s_kuc[w]; base < size; base += 64) {   // wave-uniform
              const int p = base + lane;
              const u64 m = __ballot(p < size && !(qw[p] & 1ull));
              if (m) {
                found = base + __ffsll((long long)m) - 1;
                break;
              }
            }
          }
          if (lane == 0) {
            if (found >= 0) {
              const u64 key = qw[found];
              qw[found] = key | 1ull;
              const u32 node = (u32)((key >> 1) & 0x7FFFFFFFu);
              s_sel[w] = (int)node;
              s_selpos[w] = found;
              s_kuc[w] = found;
              s_its[w] += 1;
              s_deg[w] = a.fixed_deg > 0 ? a.fixed_deg : (int)(a.off[node + 1] - a.off[node]);
              sh[3] = 1;
            } else {
              s_sel[w] = -1;
              s_deg[w] = 0;
              if (can) s_kuc[w] = size;   // ran off the end of its queue
            }
            s_wcnt[w] = 0;
            s_nnew[w] = 0;
            if (T == 1) {   // (one worker: its edge offsets need no prefix over the others)
              s_eoff[0] = 0;
              s_eoff[1] = (s_deg[0] + 7) & ~7;
            }
          }
        }
        __syncthreads();
        if (T > 1) {
          if (tid == 0) {
            int acc = 0;
            for (int w = 0; w < T; ++w) {   // segments start on multiples of 8 edge slots (the rank sort reads 8 keys at a time)
              s_eoff[w] = acc;
              acc += (s_deg[w] + 7) & ~7;
            }
            s_eoff[T] = acc;
          }
          __syncthreads();
        }
        TRV2_LAP(2)
        if (!sh[3]) break;   // uniform: nobody expanded -> the round is over
        ++steps;
        const float bound = key_dist(master[L - 1]);   // last_dist (:546): worst of the master queue before this step

        // b. gather the T adjacency lists (laid out one after another: worker w's edges at [eoff[w], eoff[w+1])),
        //    test-and-set visited
        int nsel = 0;
        for (int w = 0; w < T; ++w) nsel += s_sel[w] >= 0;
        expansions += nsel;
        const int etot = s_eoff[T];
        for (int e = tid; e < etot; e += NT) {
          int w = 0;
          while (w < T - 1 && e >= s_eoff[w + 1]) ++w;
          const int j = e - s_eoff[w];
          const int node = s_sel[w];
          u32 nb = TRV2_NONE;
          if (j < s_deg[w]) {
            const int64_t rowbase = a.fixed_deg > 0 ? (int64_t)node * a.fixed_deg : a.off[node];
            nb = a.nbr[rowbase + j];
            if (PF && a.nbr_acc0) eacc[e] = a.nbr_acc0[rowbase + j];
          }
          u32 raw = 0;
          if (nb != TRV2_NONE) {
            const u32 bit = 1u << (nb & 31);
            raw = gen ? (atomicMax(&gen[nb], stamp) < stamp ? 1u : 0u) : ((atomicOr(&vis[nb >> 5], bit) & bit) ? 0u : 1u);
            if (T > 1) {
              // ownership (step c) in the same pass (r5: it was a publish pass, a bid pass and a resolve pass): EVERY edge slot enters its node
              // into the step's table and bids for it with its index; the slot whose test-and-set made the node visited also sets the entry's top
              // bit ("became visited in THIS step").  After one barrier: the lowest bidder of a node with that bit owns it.
              u32 hs = (nb * 2654435761u) & (u32)(H - 1);
              while (true) {
                const u32 old = atomicCAS(&hid[hs], TRV2_NONE, nb);
                if (old == TRV2_NONE || (old & 0x7FFFFFFFu) == nb) break;
                hs = (hs + 1) & (u32)(H - 1);
              }
              atomicMin(&hmin[hs], e);
              if (raw) atomicOr(&hid[hs], 0x80000000u);
              npos[e] = (int)hs;   // (insert positions are not live before step f)
            }
          }
          eid[e] = nb;
          eraw[e] = raw;
        }
        __syncthreads();
        TRV2_LAP(3)
        // c. a node met by several workers in the same step belongs to the first of them in worker order (the
        //    reference's sequential `if (is_visited[nb]) continue; is_visited[nb] = true;`, :403-406, under the
        //    lockstep schedule); which edge slot's atomicOr happened to win is irrelevant.  Every edge slot whose node
        //    became visited in this step bids with its index, the lowest index owns the node.
        for (int e = tid; e < etot; e += NT) {
          const u32 nb = eid[e];
          bool mine = false;
          if (nb != TRV2_NONE) {
            if (T == 1) {
              mine = eraw[e] != 0;
            } else {
              const int hs = npos[e];
              mine = hmin[hs] == e && (hid[hs] >> 31) != 0;
            }
          }
          if (mine) {
            int w = 0;
            while (w < T - 1 && e >= s_eoff[w + 1]) ++w;
            if (PF) {
              // prefilter form (r5): ONE list of the step's nodes to evaluate, whatever worker they belong to - the mirror-row passes index it
              // directly (they used to map every item to (worker, offset) through the per-worker counts: T - 1 dependent LDS reads per row)
              const int pos = atomicAdd(&sh[10], 1);
              work[pos] = nb;
              wwk[pos] = (u32)w;
              if (a.nbr_acc0) wacc[pos] = eacc[e];
            } else {
              const int pos = atomicAdd(&s_wcnt[w], 1);
              work[s_eoff[w] + pos] = nb;
            }
            if (!gen) {
              const int ls = atomicAdd(&sh[2], 1);
              if (ls < a.vcap) vlog[ls] = nb; else sh[7] = 1;
            }
          }
        }
        if (pf) {   // (step d0's threshold and survivor counters: laid down here, under this step's barrier)
          if (tid == 0) sh[9] = stage_threshold8(bound, qst, a.scal8, a.metric, a.u8, a.slack8, 0);
          if (tid < T) s_pcnt[tid] = 0;
        }
        __syncthreads();
        if (T > 1)   // leave the ownership table empty for the next step
          for (int i = tid; i < H; i += NT) {
            hid[i] = TRV2_NONE;
            hmin[i] = 0x7FFFFFFF;
          }
        TRV2_LAP(4)
        // d. distances of all surviving neighbours (every wavefront, 16 B/lane row loads); candidates beyond the
        //    bound are dropped (`dist > dist_bound`, :427)
        int nwork = 0;
        if (PF) nwork = sh[10];
        else
          for (int w = 0; w < T; ++w) nwork += s_wcnt[w];
        evals += nwork;
        const u32* wk = work;
        const int* cntp = s_wcnt;
        if (pf) {
          // d0. 8-bit lower bound first: dot(xi, qi) + acc0[x] >= Tq(bound) is NECESSARY for dist <= bound (the integer dot product is
          //     exact, the row constant and the quantisation residuals are on the safe side of the threshold), so only the rows that
          //     pass are worth their 4 d bytes.  Survivors are compacted per worker; everything downstream sees the smaller lists.
          const int Tq = sh[9];
          u32* surv = reinterpret_cast<u32*>(sorted) + ecap;   // (the second half of `sorted`: written in step e only)
          const int g8 = lane / G8, t8 = lane & (G8 - 1);
          // G8 lanes per row, every lane NL8 16-byte pieces of it 16*G8 bytes apart (one instruction covers 16*G8 contiguous bytes
          // of each of its 64/G8 rows); U8 rows per lane group: U8*NL8 loads in flight per lane, 64/G8*U8 rows per wavefront and pass
          for (int c0 = wave * RPW8 * U8; c0 < nwork; c0 += NW * RPW8 * U8) {
            const signed char* rp8[U8];
            int wslot[U8], dot[U8], a0[U8];   // wslot: (worker << 16) | slot of the id in `work`, -1 = past the end
#pragma unroll
            for (int u = 0; u < U8; ++u) {
              int ci = c0 + u * RPW8 + g8;
              const bool ok = ci < nwork;
              if (!ok) ci = nwork - 1;
              const int sl = ci;
              wslot[u] = ok ? sl : -1;
              const u32 id = work[sl];
              rp8[u] = a.x8 + (int64_t)id * a.d_pad8;
              a0[u] = t8 == 0 ? (a.nbr_acc0 ? wacc[sl] : a.acc0[id]) : 0;    // (gathered: in flight with the row pieces)
              dot[u] = 0;
            }
#pragma unroll 1
            for (int c = t8 * 16; c < q8len; c += G8 * 16 * NL8) {
              i32x4 xv[U8][NL8];
#pragma unroll
              for (int j = 0; j < NL8; ++j) {
                const int cj = c + j * G8 * 16;
                const bool in = cj < q8len;
#pragma unroll
                for (int u = 0; u < U8; ++u) xv[u][j] = in ? *reinterpret_cast<const i32x4*>(rp8[u] + cj) : i32x4{0, 0, 0, 0};
              }
#pragma unroll
              for (int j = 0; j < NL8; ++j) {
                const int cj = c + j * G8 * 16;
                const i32x4 qv = cj < q8len ? *reinterpret_cast<const i32x4*>(sq8 + cj) : i32x4{0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < U8; ++u) {
                  dot[u] = __builtin_amdgcn_sdot4(xv[u][j][0], qv[0], dot[u], false);
                  dot[u] = __builtin_amdgcn_sdot4(xv[u][j][1], qv[1], dot[u], false);
                  dot[u] = __builtin_amdgcn_sdot4(xv[u][j][2], qv[2], dot[u], false);
                  dot[u] = __builtin_amdgcn_sdot4(xv[u][j][3], qv[3], dot[u], false);
                }
              }
            }
            if (G8 == 16) {   // (one DPP row per mirror row - d in (512, 768]: 4 VALU instructions per sum instead of 4 LDS-crossbar round trips; integer sums, any order)
#pragma unroll
              for (int u = 0; u < U8; ++u) dot[u] = row16_sum(dot[u]);
            } else {
              for (int o = G8 >> 1; o > 0; o >>= 1) {
#pragma unroll
                for (int u = 0; u < U8; ++u) dot[u] += __shfl_xor(dot[u], o);
              }
            }
#pragma unroll
            for (int u = 0; u < U8; ++u)
              if (wslot[u] >= 0 && t8 == 0 && dot[u] + a0[u] >= Tq) {
                const int sl = wslot[u], w = (int)wwk[sl];
                const int pp = atomicAdd(&s_pcnt[w], 1);
                surv[s_eoff[w] + pp] = work[sl];
              }
          }
          __syncthreads();
          wk = surv;
          cntp = s_pcnt;   // (what every later phase of the step counts a worker's keys by)
          nwork = 0;
          for (int w = 0; w < T; ++w) nwork += cntp[w];
        }
        fetched += nwork;
        if (tid < T) {   // pad every segment's keys to a multiple of 8 with EMPTY (sorts last, never counted)
          const int c = cntp[tid];
          for (int p = c; p < ((c + 7) & ~7); ++p) newk[s_eoff[tid] + p] = KEY_EMPTY;
        }
        for (int c0 = wave * RPW * U; c0 < nwork; c0 += NW * RPW * U) {
          const float* rp[U];
          u32 id[U];
          int slotk[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            int ci = c0 + u * RPW + g;
            ok[u] = ci < nwork;
            if (!ok[u]) ci = nwork - 1;
            int w = 0, base = 0;   // item ci -> (worker, index in its segment)
            for (; w < T - 1; ++w) {
              const int c = cntp[w];
              if (ci < base + c) break;
              base += c;
            }
            slotk[u] = s_eoff[w] + (ci - base);
            id[u] = wk[slotk[u]];
            rp[u] = a.rows + (int64_t)id[u] * dim;
          }
          float acc[U][1];
          row_dists<U, 1, VEC4, (PF ? EPS_TRV_NL : 1)>(rp, sq, qstride, dim, a.metric, G, acc);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (ok[u] && t == 0) {
              const float d = finish_dist(a.metric, acc[u][0]);
              newk[slotk[u]] = (d > bound) ? KEY_EMPTY : q2key(d, id[u]);
              if (a.elog) {   // (also the candidates the bound drops: a visible row beyond the queue's worst may still be an answer)
                const int lp = atomicAdd(&sh[8], 1);
                if (lp < a.elog_cap) a.elog[q * (int64_t)a.elog_cap + lp] = make_key(d, id[u]);
              }
            }
          }
        }
        __syncthreads();
        TRV2_LAP(5)
        // e. rank-sort every worker's survivors inside its segment: one thread per key, the segment read 8 keys at a time
        //    (independent LDS reads in flight).  Valid keys are distinct (a node is evaluated once), dropped ones are EMPTY:
        //    rank = number of smaller keys.
        for (int e0 = 0; e0 < etot; e0 += NT) {   // uniform trip count (ballots below)
          const int e = e0 + tid;
          int w = 0;
          bool placed = false;
          if (e < etot) {
            while (w < T - 1 && e >= s_eoff[w + 1]) ++w;
            const int j = e - s_eoff[w];
            const int cnt = cntp[w];
            if (j < cnt) {
              const u64* seg = newk + s_eoff[w];
              const u64 mine = seg[j];
              if (mine != KEY_EMPTY) {
                int rank = 0;
                for (int c0 = 0; c0 < cnt; c0 += 8) {
                  u64 o[8];
#pragma unroll
                  for (int u = 0; u < 8; ++u) o[u] = seg[c0 + u];
#pragma unroll
                  for (int u = 0; u < 8; ++u) rank += o[u] < mine ? 1 : 0;
                }
                sorted[s_eoff[w] + rank] = mine;
                placed = true;
              }
            }
          }
          for (int ww = 0; ww < T; ++ww) {        // survivors per worker: one LDS atomic per wavefront and worker
            const u64 m = __ballot(placed && w == ww);
            if (m && lane == __ffsll((long long)m) - 1) atomicAdd(&s_nnew[ww], __popcll(m));
          }
        }
        __syncthreads();
        TRV2_LAP(6)
        // f. AddIntoQueue for every worker: merge its sorted survivors into its queue in place.  The workers are handled side
        //    by side, TPW threads each, in the same barrier schedule (the chunk loop runs as long as the longest needs).
        {
          int Tp = 1;
          while (Tp < T) Tp <<= 1;                 // (T <= 128 <= NT)
          const int TPW = NT / Tp;                 // threads per worker
          const int CW = TPW * R;                  // queue entries per worker and chunk
          const int w = tid / TPW;                 // this thread's worker (may be >= T: idle group)
          const int tg = tid - w * TPW;
          const bool mine_w = w < T && s_sel[w] >= 0;
          const int nnew = mine_w ? s_nnew[w] : 0;
          const int cap = w == T - 1 ? L : Lq;
          const int size = mine_w ? s_size[w] : 0;
          u64* qw = qbase + (int64_t)(w < T ? w : 0) * Lq;
          const u64* sw = sorted + (w < T ? s_eoff[w] : 0);
          int* np = npos + (w < T ? s_eoff[w] : 0);
          for (int j = tg; j < nnew; j += TPW) np[j] = j + lower_bound_q(qw, size, sw[j]);
          if (tid == 0) sh[0] = 0;
          __syncthreads();
          const int pmin = nnew > 0 ? np[0] : cap;
          const bool moves = nnew > 0 && pmin < cap && size > 0;
          const int top = moves ? ((size - 1) / CW) * CW : 0;
          const int nchunks = moves ? (top / CW - pmin / CW + 1) : 0;
          int maxchunks = 1;
          if (L > CW || Lq > CW) {   // (queues longer than one chunk per worker: the workers agree on the trip count; else one predicated trip)
            if (tg == 0 && nchunks > 0) atomicMax(&sh[0], nchunks);
            __syncthreads();
            maxchunks = sh[0];
          }
          // old entries at positions >= pmin move right by the number of new keys ordered before them; chunks from the tail so
          // that nothing unread is overwritten
          for (int it = 0; it < maxchunks; ++it) {
            u64 v[R];
            int dst[R];
            const int cb = top - it * CW;
#pragma unroll
            for (int i = 0; i < R; ++i) {
              const int p = cb + i * TPW + tg;
              dst[i] = -1;
              v[i] = KEY_EMPTY;
              if (it < nchunks && p >= pmin && p < size) {
                v[i] = qw[p];
                dst[i] = p + lower_bound_q(sw, nnew, v[i]);
              }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < R; ++i)
              if (dst[i] >= 0 && dst[i] < cap) qw[dst[i]] = v[i];
          }
          __syncthreads();
          if (nnew > 0 && pmin < cap)
            for (int j = tg; j < nnew; j += TPW)
              if (np[j] < cap) qw[np[j]] = sw[j];
          if (mine_w && tg == 0) {
            if (nnew > 0) s_size[w] = size + nnew < cap ? size + nnew : cap;
            const int r = (nnew > 0 && pmin < cap) ? pmin : cap;
            const int kuc = s_selpos[w];
            s_kuc[w] = r <= kuc ? r : kuc + 1;   // (:661-665)
          }
          if (tid == 0) {   // (for the next step's select, see there)
            sh[3] = 0;
            sh[10] = 0;
          }
          __syncthreads();
        }
        TRV2_LAP(7)
      }  // steps

      if (round == 0 || T == 1) continue;
      // ---- MergeAllQueuesToMaster (:297-326): worker queues into the fixed-size master, in worker order;
      // equal (dist,id) = duplicate: the master's copy stays and turns unchecked if the worker's copy is unchecked (:182-213)
      for (int w = 0; w < T - 1; ++w) {
        const int nb = s_size[w];
        if (nb == 0) continue;   // uniform
        const u64* B = qbase + (int64_t)w * Lq;
        int* bpos = aux;
        int* ndp = aux + Lq;
        // pass 1: where every worker key falls in the master; duplicates
        for (int j = tid; j < nb; j += NT) {
          const u64 kb = B[j];
          const int cnt = lower_bound_q(master, L, kb);
          bool dup = cnt < L && (master[cnt] >> 1) == (kb >> 1);
          if (dup && !(kb & 1ull) && (master[cnt] & 1ull)) master[cnt] &= ~1ull;
          bpos[j] = cnt | (dup ? (int)0x80000000 : 0);
        }
        __syncthreads();
        const int r = bpos[0] & 0x7FFFFFFF;   // lowest insert position (lower_bound of the worker's best key, :158-163)
        if (r < L) {
          // pass 2: ndp[j] = non-duplicate worker keys before j
          if (tid == 0) sh[4] = 0;
          __syncthreads();
          for (int base = 0; base < nb; base += NT) {
            const int j = base + tid;
            const bool nd = j < nb && bpos[j] >= 0;
            const u64 m = __ballot(nd);
            if (lane == 0) s_wave[wave] = __popcll(m);
            __syncthreads();
            int i = sh[4];
            for (int w2 = 0; w2 < wave; ++w2) i += s_wave[w2];
            i += __popcll(m & ((1ull << lane) - 1ull));
            if (j < nb) ndp[j] = i;
            __syncthreads();
            if (tid == 0) {
              int tot = sh[4];
              for (int w2 = 0; w2 < NW; ++w2) tot += s_wave[w2];
              sh[4] = tot;
            }
            __syncthreads();
          }
          const int nnd = sh[4];
          // pass 3: master entries at positions >= r move right by the non-duplicate worker keys ordered before them
          for (int cb = ((L - 1) / C) * C; cb + C > r && cb >= 0; cb -= C) {
            u64 v[R];
            int dst[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
              const int p = cb + i * NT + tid;
              dst[i] = -1;
              v[i] = KEY_EMPTY;
              if (p >= r && p < L) {
                v[i] = master[p];
                const int jb = lower_bound_q(B, nb, v[i]);
                dst[i] = p + (jb == nb ? nnd : ndp[jb]);
              }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < R; ++i)
              if (dst[i] >= 0 && dst[i] < L) master[dst[i]] = v[i];
          }
          __syncthreads();
          // pass 4: the non-duplicate worker keys
          for (int j = tid; j < nb; j += NT) {
            const int bp = bpos[j];
            if (bp >= 0) {
              const int dst = bp + ndp[j];
              if (dst < L) master[dst] = B[j];
            }
          }
        }
        __syncthreads();
        if (tid == 0) {
          if (r <= s_kuc[T - 1]) s_kuc[T - 1] = r;   // (:686-689)
          s_size[w] = 0;
        }
        __syncthreads();
      }
      TRV2_LAP(8)
    }  // rounds

    // ------------------------------------------------------------------ results + visited reset
    if (a.out_queue)
      for (int i = tid; i < L; i += NT) a.out_queue[q * L + i] = master[i];
    if (a.elog_cnt && tid == 0) {
      a.elog_cnt[q] = (u32)sh[8];
      if ((int)sh[8] > a.elog_cap) {   // the log was too short for this walk: the host repeats the search with a longer one
        atomicAdd(&a.counters[5], 1ull);
        atomicMax(&a.counters[6], (unsigned long long)sh[8]);
      }
    }
    const int nlog = sh[2];
    if (gen) {
      // (nothing to undo: the next query of this slot carries a larger stamp)
    } else if (sh[7] || nlog > a.vcap) {
      for (int64_t i = tid; i < a.words; i += NT) vis[i] = 0;
    } else {
      for (int i = tid; i < L; i += NT) vis[a.init_ids[i] >> 5] = 0;
      for (int i = tid; i < nlog; i += NT) vis[vlog[i] >> 5] = 0;
    }
    __threadfence();
    __syncthreads();
    TRV2_LAP(9)
  }
#undef TRV2_LAP
  if (tid == 0) {
    atomicAdd(&a.counters[0], evals);
    atomicAdd(&a.counters[1], expansions);
    atomicAdd(&a.counters[2], steps);
    atomicAdd(&a.counters[3], rounds);
    atomicAdd(&a.counters[4], fetched);
  }
}

// LDS bytes of one workgroup; qglobal = queues in HBM
inline int traverse2_hash_slots(int T, int dp) {
  if (T <= 1) return 0;
  int h = 64;
  while (h < 2 * T * dp) h <<= 1;
  return h;
}
inline size_t traverse2_lds_bytes(int dim, int T, int Lq, int64_t qtot, int dp, bool qglobal, bool prefilter = false, int cols8 = 0) {
  const int qstride = (dim + 3) & ~3;
  const size_t ecap = (size_t)T * dp;
  return (size_t)qstride * 4 + (qglobal ? (size_t)TRV2_SB : (size_t)qtot) * 8 + ecap * (8 + 8 + 4 + 4 + 4 + 4) + (size_t)traverse2_hash_slots(T, dp) * 8 +
         ((qglobal || T == 1) ? 0 : (size_t)2 * Lq * 4) + (size_t)trv2_sh_ints(T) * 4 + (prefilter ? (size_t)trv2_pf_bytes(cols8 > dim ? cols8 : dim) : 0);
}

}  // namespace eps
