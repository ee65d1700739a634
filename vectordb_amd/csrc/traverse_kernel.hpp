// Device traversal kernel shared by the search path (traverse.hip) and the graph build (graph_build.hip).
// See traverse.hip for the mapping to VecSearchExecutor::SearchImpl (vec_search_executor.cpp:518-715) and
// graph_build.hip for its use as NsgIndex::GetNeighbors (nsg.cpp:158-268).
#pragma once
#include "kernels.hpp"

namespace eps {

// ------------------------------------------------------------------------------------------------ kernel
struct TraverseArgs {
  const float* rows;
  int dim;
  int metric;
  const int64_t* off;   // CSR offsets, or null when fixed_deg > 0
  const u32* nbr;       // CSR neighbours, or [n][fixed_deg] lists padded with 0xFFFFFFFF
  int fixed_deg;
  const u32* init_ids;
  const float* queries;
  int L;       // queue length
  int Lp2;     // next power of two >= L
  int nseeds;  // number of init_ids (0: L)
  int Lsel;    // candidates are selected for expansion among the first Lsel queue positions only (0: L)
  int M;       // expansions per round
  u32* visited;      // BITMAP mode: [gridDim.x][words] in HBM
  int64_t words;
  u64* out_queue;    // [nq][L] (may be null)
  unsigned long long* counters;  // [0] distance evals, [1] expansions, [2] (counters_n > 2) searches that filled the visited hash
  int counters_n;
  u32* ghash;        // HASHVIS: non-null = the visited hash lives in HBM, [gridDim.x][hslots] u32, pre-set to 0xFF bytes by the host
  int hslots;        //          (power of two); null = TRV_HASH slots in LDS
  u64* log;          // LOG mode: [nq][log_cap] plain (dist,id) keys: the final queue = the L closest evaluated nodes, ascending
  u32* log_cnt;      // [nq]
  int log_cap;       // >= L
  // exact 8-bit lower-bound prefilter of step 3 (same test as traverse2_kernel's step d0); x8 == null: off
  const signed char* x8;   // [n_pad][d_pad8] the table's 8-bit mirror
  const int* acc0;         // [n_pad]
  const float* scal8;
  const signed char* q8;   // [gridDim.x][d_pad8] this launch's queries on the mirror's grid
  const float* qstat8;     // [gridDim.x][4]
  int d_pad8;
  int cols8;               // leading bytes of a mirror row that carry values (Quant8View::cols8)
  float u8, slack8;
};

constexpr int TRV_HASH = 8192;   // LDS visited hash slots (HASHVIS mode), power of two
constexpr u32 TRV_NONE = 0xFFFFFFFFu;

constexpr int TRV_CHUNK = 256;   // neighbours handled per sub-round
constexpr int TRV_MAXM = 16;

// queue key: ord(dist) << 32 | id << 1 | checked
__device__ __forceinline__ u64 qkey(float d, u32 id, u32 checked) { return ((u64)f2ord(d + 0.0f) << 32) | ((u64)id << 1) | checked; }

// test-and-set of the visited set; returns true when `id` was not visited before
// (HASHVIS: `open` = the table had room for a whole chunk of insertions when the chunk started - decided once per chunk from
// a count that is stable there, so that WHICH nodes a nearly full table still admits does not depend on thread timing)
template <bool HASHVIS>
__device__ __forceinline__ bool visit(u32* vis, u32* hash, int* hcount, u32 id, bool open = true, int hbits = 13) {
  if (!HASHVIS) {
    const u32 bit = 1u << (id & 31);
    const u32 old = atomicOr(&vis[id >> 5], bit);
    return !(old & bit);
  } else {
    if (!open) return false;  // table (nearly) full: stop discovering (build-time searches only)
    const u32 mask = (1u << hbits) - 1u;
    u32 h = (id * 2654435761u) >> (32 - hbits);
    for (u32 probe = 0; probe <= mask; ++probe) {
      const u32 old = atomicCAS(&hash[h], TRV_NONE, id);
      if (old == TRV_NONE) {
        atomicAdd(hcount, 1);
        return true;
      }
      if (old == id) return false;
      h = (h + 1) & mask;
    }
    return false;
  }
}

// NW = wavefronts per workgroup (4: many queries in flight; 16: one query spread over a whole CU's worth of waves,
// so the ~50 candidate rows of an expansion are all in flight at once — single-query latency)
template <bool VEC4, bool HASHVIS, bool LOG, int NW>
__global__ __launch_bounds__(NW * 64) void traverse_kernel(TraverseArgs a) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  float* sq = reinterpret_cast<float*>(smem_raw);                        // [qstride]
  u64* queue = reinterpret_cast<u64*>(sq + qstride);                     // [Lp2]
  u64* newk = queue + a.Lp2;                                              // [TRV_CHUNK] unsorted keys of this sub-round
  u64* sorted = newk + TRV_CHUNK;                                         // [TRV_CHUNK]
  u32* work = reinterpret_cast<u32*>(sorted + TRV_CHUNK);                 // [TRV_CHUNK] surviving neighbour ids
  int* sh = reinterpret_cast<int*>(work + TRV_CHUNK);                     // small scalars [64]
  u32* hash = (HASHVIS && a.ghash) ? a.ghash + (int64_t)blockIdx.x * a.hslots : reinterpret_cast<u32*>(sh + 64);   // [TRV_HASH] in LDS (HASHVIS only) or [hslots] in HBM
  const int hbits = (HASHVIS && a.ghash) ? 31 - __clz(a.hslots) : 13;
  const int hlimit = (HASHVIS && a.ghash) ? (a.hslots / 4) * 3 : (TRV_HASH * 3) / 4;
  // sh[0]=work count, sh[1]=selected count, sh[2]=k (first possibly-unchecked position), sh[3]=valid new count,
  // sh[4]=r_min, sh[5]=position of the first selected candidate, sh[6]=hash fill, sh[7]=log fill, sh[8..8+M) selected node ids, sh[24..24+M+1) edge prefix, sh[41]=prefilter
  // threshold, sh[42]=prefilter survivors of this chunk, sh[43]=fp32 rows read in step 3, sh[48..48+NW) per-wave counts
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = tid >> 6;
  const int64_t q = blockIdx.x;
  const int L = a.L;
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  constexpr int U = 4;
  // prefilter block at the end of the launch's LDS (traverse_lds_bytes(..., prefilter = true))
  const bool pf = a.x8 != nullptr;
  const size_t pf_off = ((size_t)qstride * 4 + (size_t)a.Lp2 * 8 + TRV_CHUNK * 8 * 2 + TRV_CHUNK * 4 + 64 * 4 + ((HASHVIS && !a.ghash) ? TRV_HASH * 4 : 0) + 15) & ~(size_t)15;
  float* qst = reinterpret_cast<float*>(smem_raw + pf_off);      // [4]
  signed char* sq8 = reinterpret_cast<signed char*>(qst + 4);     // [q8len]
  const int q8len = ((a.cols8 > dim ? a.cols8 : dim) + 15) & ~15;
  constexpr int U8 = 2, NL8 = 3;
  int G8 = 4;
  while (G8 < 64 && G8 * 16 * NL8 < q8len) G8 <<= 1;
  const int RPW8 = 64 / G8;
  u32* vis = HASHVIS ? nullptr : a.visited + q * a.words;
  unsigned long long evals = 0, expansions = 0;
  u64* qlog = LOG ? a.log + q * (int64_t)a.log_cap : nullptr;
  if (HASHVIS) {
    if (!a.ghash)
      for (int i = tid; i < TRV_HASH; i += NT) hash[i] = TRV_NONE;
    if (tid == 0) sh[6] = 0;
  }
  if (LOG && tid == 0) sh[7] = 0;
  if (tid == 0) sh[43] = 0;   // (read again by thread 0 only)
  if (HASHVIS || LOG) __syncthreads();

  for (int i = tid; i < qstride; i += NT) sq[i] = i < dim ? a.queries[q * dim + i] : 0.f;
  if (pf) {
    for (int i = tid; i < (q8len >> 2); i += NT) reinterpret_cast<u32*>(sq8)[i] = reinterpret_cast<const u32*>(a.q8 + q * a.d_pad8)[i];
    if (tid < 4) qst[tid] = a.qstat8[q * 4 + tid];
  }
  for (int i = tid; i < a.Lp2; i += NT) queue[i] = KEY_EMPTY;
  // InitializeSetLPara (:446-485): mark seeds visited, L seed distances, sort
  const int NS = a.nseeds > 0 ? a.nseeds : L;     // seeds (<= L)
  const int LS = a.Lsel > 0 ? a.Lsel : L;         // expansion window
  for (int i = tid; i < NS; i += NT) visit<HASHVIS>(vis, hash, &sh[6], a.init_ids[i], true, hbits);
  __syncthreads();
  for (int c0 = wave * RPW * U; c0 < NS; c0 += NW * RPW * U) {
    const float* rp[U];
    u32 id[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ci = c0 + u * RPW + g;
      ok[u] = ci < NS;
      id[u] = a.init_ids[ok[u] ? ci : NS - 1];
      rp[u] = a.rows + (int64_t)id[u] * dim;
    }
    float acc[U][1];
    row_dists<U, 1, VEC4>(rp, sq, qstride, dim, a.metric, G, acc);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ok[u] && t == 0) {
        const float d = finish_dist(a.metric, acc[u][0]);
        queue[c0 + u * RPW + g] = qkey(d, id[u], 0);
      }
  }
  evals += NS;
  __syncthreads();
  // bitonic sort of queue[0..Lp2)
  for (int size = 2; size <= a.Lp2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (a.Lp2 >> 1); i += NT) {
        const int lo = ((i / stride) * (stride << 1)) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const u64 x = queue[lo], y = queue[hi];
        if ((x > y) == up) {
          queue[lo] = y;
          queue[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) sh[2] = 0;
  __syncthreads();

  // ---- rounds
  while (true) {
    // 1. select the first M unchecked candidates at positions >= k, mark them checked
    if (tid == 0) sh[1] = 0;
    __syncthreads();
    for (int base = sh[2]; base < LS; base += NT) {
      const int p = base + tid;
      const bool un = p < LS && !(queue[p] & 1ull);
      const u64 m = __ballot(un);
      if (lane == 0) sh[48 + wave] = __popcll(m);
      __syncthreads();
      int before = sh[1];
      for (int w2 = 0; w2 < wave; ++w2) before += sh[48 + w2];
      const int rank = before + __popcll(m & ((1ull << lane) - 1ull));
      if (un && rank < a.M) {
        sh[8 + rank] = (int)((queue[p] >> 1) & 0x7FFFFFFFu);
        queue[p] |= 1ull;
        if (rank == 0) sh[5] = p;  // everything before the first selected candidate is checked
      }
      __syncthreads();
      if (tid == 0) {
        int tot = sh[1];
        for (int w2 = 0; w2 < NW; ++w2) tot += sh[48 + w2];
        sh[1] = tot < a.M ? tot : a.M;
      }
      __syncthreads();
      if (sh[1] >= a.M) break;
    }
    const int nsel = sh[1];
    if (nsel == 0) break;
    expansions += nsel;
    if (tid == 0) {
      sh[2] = sh[5];
      int acc = 0;
      for (int i = 0; i < nsel; ++i) {
        sh[24 + i] = acc;
        acc += a.fixed_deg > 0 ? a.fixed_deg : (int)(a.off[sh[8 + i] + 1] - a.off[sh[8 + i]]);
      }
      sh[24 + nsel] = acc;
    }
    __syncthreads();
    const int total_edges = sh[24 + nsel];

    for (int e0 = 0; e0 < total_edges; e0 += TRV_CHUNK) {
      // 2. gather neighbour ids, test-and-set visited, compact survivors
      if (tid == 0) {
        sh[0] = 0;
        sh[3] = 0;
      }
      __syncthreads();
      const bool hash_open = !HASHVIS || sh[6] + TRV_CHUNK <= hlimit;   // (sh[6] is stable here: see visit())
      {
        const int e = e0 + tid;
        bool fresh = false;
        u32 nb = 0;
        if (tid < TRV_CHUNK && e < total_edges) {
          int i = 0;
          while (i + 1 < nsel && sh[24 + i + 1] <= e) ++i;
          const int64_t rowbase = a.fixed_deg > 0 ? (int64_t)sh[8 + i] * a.fixed_deg : a.off[sh[8 + i]];
          nb = a.nbr[rowbase + (e - sh[24 + i])];
          fresh = nb != TRV_NONE && visit<HASHVIS>(vis, hash, &sh[6], nb, hash_open, hbits);
        }
        const u64 m = __ballot(fresh);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&sh[0], __popcll(m));
        wbase = __shfl(wbase, 0);
        if (fresh) work[wbase + __popcll(m & ((1ull << lane) - 1ull))] = nb;
      }
      __syncthreads();
      int nwork = sh[0];
      if (nwork == 0) continue;
      evals += nwork;

      // 3. distances; drop candidates beyond the current worst-of-queue (dist > bound, :427)
      const float bound = key_dist(queue[L - 1]);
      const u32* wk = work;
      if (pf) {
        // 3a. the 8-bit mirror row first: dot(xi, qi) + acc0[x] >= Tq(bound) is necessary for dist <= bound (device_common.hpp,
        //     stage_threshold8), so only the rows that pass are read in fp32.  While the queue is not full everything passes.
        if (tid == 0) {
          sh[41] = queue[L - 1] == KEY_EMPTY ? (int)0x80000000 : stage_threshold8(bound, qst, a.scal8, a.metric, a.u8, a.slack8, 0);
          sh[42] = 0;
        }
        __syncthreads();
        const int Tq = sh[41];
        u32* surv = reinterpret_cast<u32*>(sorted);   // (free until step 4)
        const int g8 = lane / G8, t8 = lane & (G8 - 1);
        for (int c0 = wave * RPW8 * U8; c0 < nwork; c0 += NW * RPW8 * U8) {
          const signed char* rp8[U8];
          int slot[U8], dot[U8], a0[U8];
#pragma unroll
          for (int u = 0; u < U8; ++u) {
            const int ci = c0 + u * RPW8 + g8;
            slot[u] = ci < nwork ? ci : -1;
            const u32 id = work[ci < nwork ? ci : nwork - 1];
            rp8[u] = a.x8 + (int64_t)id * a.d_pad8;
            a0[u] = t8 == 0 ? a.acc0[id] : 0;
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
          if (G8 == 16) {   // (one DPP row per mirror row: device_common.hpp row16_sum; integer sums, any order)
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
            if (slot[u] >= 0 && t8 == 0 && dot[u] + a0[u] >= Tq) surv[atomicAdd(&sh[42], 1)] = work[slot[u]];
        }
        __syncthreads();
        nwork = sh[42];
        wk = surv;
        if (nwork == 0) continue;
        if (tid == 0) sh[43] += nwork;
      }
      for (int c0 = wave * RPW * U; c0 < nwork; c0 += NW * RPW * U) {
        const float* rp[U];
        u32 id[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ci = c0 + u * RPW + g;
          ok[u] = ci < nwork;
          id[u] = wk[ok[u] ? ci : nwork - 1];
          rp[u] = a.rows + (int64_t)id[u] * dim;
        }
        float acc[U][1];
        row_dists<U, 1, VEC4>(rp, sq, qstride, dim, a.metric, G, acc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (ok[u] && t == 0) {
            const float d = finish_dist(a.metric, acc[u][0]);
            newk[c0 + u * RPW + g] = (d > bound) ? KEY_EMPTY : qkey(d, id[u], 0);
          }
        }
      }
      __syncthreads();
      // 4. rank-sort the survivors (nwork <= 256: one thread per key, broadcast LDS reads)
      if (tid < nwork) {
        const u64 mine = newk[tid];
        if (mine != KEY_EMPTY) {
          int rank = 0;
          for (int j = 0; j < nwork; ++j) {
            const u64 o = newk[j];
            rank += (o < mine) || (o == mine && j < tid);
          }
          sorted[rank] = mine;
          atomicAdd(&sh[3], 1);
        }
      }
      __syncthreads();
      const int nnew = sh[3];
      if (nnew == 0) continue;
      // 5. in-place parallel merge of sorted[0..nnew) into queue[0..L): read phase, barrier, write phase
      u64 oldv[4096 / NT];
      int oldp[4096 / NT];
#pragma unroll
      for (int i = 0; i < 4096 / NT; ++i) {  // L <= 4096; constant trip count keeps oldv/oldp in registers
        const int p = tid + i * NT;
        oldp[i] = L;
        oldv[i] = KEY_EMPTY;
        if (p < L) {
          const u64 v = queue[p];
          int lo = 0, hi = nnew;  // number of new keys ordered before v
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((sorted[mid] >> 1) < (v >> 1)) lo = mid + 1; else hi = mid;
          }
          oldv[i] = v;
          oldp[i] = p + lo;
        }
      }
      u64 nv = KEY_EMPTY;
      int np = L;
      if (tid < nnew) {
        nv = sorted[tid];
        int lo = 0, hi = L;  // number of old keys ordered before nv
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((queue[mid] >> 1) < (nv >> 1)) lo = mid + 1; else hi = mid;
        }
        np = tid + lo;
        if (tid == 0) sh[4] = np;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4096 / NT; ++i)
        if (oldp[i] < L) queue[oldp[i]] = oldv[i];
      if (np < L) queue[np] = nv;
      __syncthreads();
      if (tid == 0 && sh[4] < sh[2]) sh[2] = sh[4];
      __syncthreads();
    }
    // first possibly-unchecked position: everything before the old k was checked and stays so unless a
    // new candidate landed there (handled via r_min above)
    __syncthreads();
  }
  if (a.out_queue)
    for (int i = tid; i < L; i += NT) a.out_queue[q * L + i] = queue[i];
  if (LOG) {
    // the log = the final queue: the L closest evaluated nodes in ascending (dist, id) order.  (Until r2 every evaluation was
    // appended to a global list of 3840 entries, which EVERY Link search of a 768-d table overran - what got logged were the
    // nodes met first, i.e. the far ones.)  With L >= candidate_pool_size this is all SyncPrune's depth-limited SelectEdge can see.
    int cnt = 0;
    for (int i = tid; i < L; i += NT) {
      const u64 k2 = queue[i];
      if (k2 != KEY_EMPTY) {
        qlog[i] = ((k2 >> 32) << 32) | ((k2 >> 1) & 0x7FFFFFFFull);
        ++cnt;
      }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) atomicAdd(&sh[7], cnt);
    __syncthreads();
    if (tid == 0) a.log_cnt[q] = (u32)sh[7];
  }
  if (tid == 0) {
    atomicAdd(&a.counters[0], evals);
    atomicAdd(&a.counters[1], expansions);
    if (pf && a.counters_n > 3) atomicAdd(&a.counters[3], (unsigned long long)sh[43]);   // fp32 rows step 3 still read
    if (HASHVIS && a.counters_n > 2 && sh[6] + TRV_CHUNK > hlimit) atomicAdd(&a.counters[2], 1ull);   // searches that filled the visited hash
  }
}


inline size_t traverse_lds_bytes(int dim, int Lp2, bool hashvis, bool prefilter = false, int cols8 = 0) {
  const int qstride = (dim + 3) & ~3;
  return (size_t)qstride * 4 + (size_t)Lp2 * 8 + TRV_CHUNK * 8 * 2 + TRV_CHUNK * 4 + 64 * 4 + (hashvis ? TRV_HASH * 4 : 0) +
         (prefilter ? (size_t)(16 + 16 + (((cols8 > dim ? cols8 : dim) + 15) & ~15)) : 0);
}

}  // namespace eps

