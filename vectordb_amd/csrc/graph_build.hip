// Device graph build: ANNGraphSegment::BuildFromVectorTable (reference: engine/db/ann_graph_segment.cpp:201-242)
//   KNNGraph (NN-Descent, K = 100)  -> NsgIndex::Build (InitNavigationPoint, Link = GetNeighbors + SyncPrune,
//   InterInsert, CheckConnectivity)  -> CSR  (engine/db/index/knn/knn.hpp:90-135, engine/db/index/nsg/nsg.cpp:45-775)
//
// MI355X form:
//   1. kNN graph = the K nearest rows of every row by one batched flat scan per 2048-row block of "queries"
//      on the matrix cores (mfma_filter.hip in approx mode: fp16 keys, no re-rank) — the exact object NN-Descent
//      approximates iteratively under per-node spinlocks.  O(n^2 d) flops, all MFMA.
//   2. navigation node = row closest to the centroid (exact flat scan with k = 1; the reference searches the
//      kNN graph from a random start for the same thing, nsg.cpp:101-155).
//   3. Link: per node, best-first search on the kNN graph from the navigation node with a queue of
//      search_length (GetNeighbors, nsg.cpp:158-268) = traverse_kernel with an LDS hash as visited set and a
//      log of every evaluated (dist,id); then SyncPrune/SelectEdge (nsg.cpp:540-580, 655-685): pool = log +
//      kNN(v), sorted, MRNG rule over the first candidate_pool_size candidates, out_degree edges kept.
//   4. InterInsert (nsg.cpp:583-653): reverse edges are scattered into per-node candidate lists with atomics and
//      every node re-runs SelectEdge over (its edges + reverse candidates) when they exceed out_degree.  The
//      reference does this serially in node order (its omp-for is orphaned, nsg.cpp:531-536); the batched form
//      applies the same rule to the same candidate sets, so edge sets differ only by processing order.
//   5. CheckConnectivity (nsg.cpp:687-775): reachability from the navigation node on the host (BFS over the
//      CSR), every unreached node gets an in-edge from the closest reached node found by a device search.
// NSG distances are always L2 (ann_graph_segment.cpp:216); the kNN stage uses the field metric (knn.hpp:41-53).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "index.hpp"
#include "traverse_kernel.hpp"

namespace eps {

constexpr int PRUNE_POOL = 4096;   // sorted candidate pool per node (LDS)

// ------------------------------------------------------------------------------------------------ kNN extraction
__global__ void knn_extract_kernel(const u64* run_keys, int k1, int64_t q0, int64_t nq, int K, u32* knn_ids) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const u64* src = run_keys + q * k1;
  u32* dst = knn_ids + (q0 + q) * K;
  int o = 0;
  for (int e = 0; e < k1 && o < K; ++e) {
    const u64 key = src[e];
    if (key == KEY_EMPTY) break;
    const u32 id = key_id(key);
    if ((int64_t)id == q0 + q) continue;  // self
    dst[o++] = id;
  }
  for (; o < K; ++o) dst[o] = TRV_NONE;
}

// the approximate top-kA of a kNN pass as candidate lists for the exact re-rank (run_keys is reset to EMPTY for it)
__global__ void knn_keys_to_cand_kernel(const u64* approx_keys, int kA, int64_t nq, u32* cand, u32* cnt) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  u32 c = 0;
  for (int e = 0; e < kA; ++e) {
    const u64 key = approx_keys[q * kA + e];
    if (key != KEY_EMPTY) cand[q * (int64_t)kA + c++] = key_id(key);
  }
  cnt[q] = c;
}
__global__ void knn_export_kernel(const u32* knn, int64_t count, int64_t* out) {   // u32 lists (TRV_NONE padded) -> int64 (-1 padded)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = knn[i] == TRV_NONE ? -1 : (int64_t)knn[i];
}
__global__ void knn_import_kernel(const int64_t* in, int64_t count, int64_t n, u32* knn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) knn[i] = (in[i] < 0 || in[i] >= n) ? TRV_NONE : (u32)in[i];
}

// ------------------------------------------------------------------------------------------------ centroid
__global__ __launch_bounds__(256) void centroid_kernel(const float* rows, int64_t n, int dim, int64_t rows_per_block,
                                                       float* acc) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int c = threadIdx.x; c < dim; c += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += rows[r * dim + c];
    atomicAdd(&acc[c], s);
  }
}
__global__ void scale_kernel(float* v, int dim, float f) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < dim) v[c] *= f;
}
__global__ void gather_rows_kernel(const float* rows, const u32* ids, int64_t m, int dim, float* out) {
  const int64_t i = blockIdx.x;
  if (i >= m) return;
  const float* src = rows + (int64_t)ids[i] * dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) out[i * dim + c] = src[c];
}

// ------------------------------------------------------------------------------------------------ prune (SelectEdge)
struct PruneArgs {
  const float* rows;
  int dim;
  int64_t v0;             // first node of this launch (node = v0 + blockIdx.x)
  const u32* node_ids;    // optional: node = node_ids[blockIdx.x]; listB and the outputs are then indexed by blockIdx.x
  const u64* log;         // A: [nb][log_cap] plain (dist,id) keys, indexed by blockIdx.x; may be null
  const u32* log_cnt;
  int log_cap;
  const u32* listB;       // B: [n][degB] ids whose distance to v is computed here (kNN lists), TRV_NONE padded; may be null
  int degB;
  const u32* idsC;        // C/D: [n][cap] (id, dist) pairs with counts; may be null
  const float* distC;
  const u32* cntC;
  int capC;
  const u32* idsD;
  const float* distD;
  const u32* cntD;
  int capD;
  const unsigned long long* offD;   // non-null: D is a CSR - node v's pairs are idsD / distD [offD[v], offD[v+1])
  int depth;              // candidates scanned (candidate_pool_size), <= 0: unlimited
  int R;                  // out_degree
  u32* out_ids;           // [n][R], TRV_NONE padded
  float* out_dist;
  u32* out_deg;
};

template <bool VEC4>
__global__ __launch_bounds__(256) void prune_kernel(PruneArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int dim = a.dim;
  const int qstride = (dim + 3) & ~3;
  float* sq = reinterpret_cast<float*>(smem_raw);            // [qstride] V[v], later V[p]
  u64* pool = reinterpret_cast<u64*>(sq + qstride);          // [PRUNE_POOL]
  u32* kept = reinterpret_cast<u32*>(pool + PRUNE_POOL);     // [R]
  float* keptd = reinterpret_cast<float*>(kept + a.R);       // [R]
  int* sh = reinterpret_cast<int*>(keptd + a.R);             // [64] scalars, then the staged candidate vectors of the SelectEdge walk
  const int tid = threadIdx.x;
  const int lane = lane_id();
  const int wave = tid >> 6;
  const int64_t v = a.node_ids ? (int64_t)a.node_ids[blockIdx.x] : a.v0 + blockIdx.x;
  const int64_t vrow = a.node_ids ? (int64_t)blockIdx.x : v;   // row of listB / the outputs
  const int G = group_lanes(dim, VEC4);
  const int RPW = 64 / G;
  const int g = lane / G;
  const int t = lane & (G - 1);
  constexpr int U = 4;

  if (tid == 0) sh[0] = 0;
  for (int i = tid; i < PRUNE_POOL; i += 256) pool[i] = KEY_EMPTY;
  for (int i = tid; i < qstride; i += 256) sq[i] = i < dim ? a.rows[v * dim + i] : 0.f;
  __syncthreads();
  // ---- gather the pool
  if (a.log) {
    const u32 cnt = a.log_cnt[blockIdx.x];
    const u64* src = a.log + (int64_t)blockIdx.x * a.log_cap;
    for (u32 i = tid; i < cnt; i += 256) {
      const int slot = atomicAdd(&sh[0], 1);
      if (slot < PRUNE_POOL) pool[slot] = src[i];
    }
  }
  if (a.idsC) {
    const u32 cnt = a.cntC[v] < (u32)a.capC ? a.cntC[v] : (u32)a.capC;
    for (u32 i = tid; i < cnt; i += 256) {
      const int slot = atomicAdd(&sh[0], 1);
      if (slot < PRUNE_POOL) pool[slot] = make_key(a.distC[v * a.capC + i], a.idsC[v * a.capC + i]);
    }
  }
  if (a.idsD) {
    const int64_t baseD = a.offD ? (int64_t)a.offD[v] : v * a.capD;
    const u32 cnt = a.offD ? (u32)(a.offD[v + 1] - a.offD[v]) : (a.cntD[v] < (u32)a.capD ? a.cntD[v] : (u32)a.capD);
    for (u32 i = tid; i < cnt; i += 256) {
      const int slot = atomicAdd(&sh[0], 1);
      if (slot < PRUNE_POOL) pool[slot] = make_key(a.distD[baseD + i], a.idsD[baseD + i]);
    }
  }
  __syncthreads();
  if (a.listB) {  // "avoid lose nearest neighbor in knng" (nsg.cpp:542-557): distances to the kNN list of v
    const u32* lst = a.listB + vrow * a.degB;
    const int base = sh[0];
    for (int c0 = wave * RPW * U; c0 < a.degB; c0 += 4 * RPW * U) {
      const float* rp[U];
      u32 id[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ci = c0 + u * RPW + g;
        ok[u] = ci < a.degB;
        id[u] = ok[u] ? lst[ci] : TRV_NONE;
        ok[u] = ok[u] && id[u] != TRV_NONE;
        rp[u] = a.rows + (int64_t)(ok[u] ? id[u] : (u32)v) * dim;
      }
      float acc[U][1];
      row_dists<U, 1, VEC4>(rp, sq, qstride, dim, 0, G, acc);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int slot = base + c0 + u * RPW + g;
        if (t == 0 && c0 + u * RPW + g < a.degB && slot < PRUNE_POOL) pool[slot] = ok[u] ? make_key(acc[u][0], id[u]) : KEY_EMPTY;
      }
    }
    __syncthreads();
  }
  // ---- sort by (dist,id)   (std::sort(pool), nsg.cpp:560 — ties by id here)
  // only the occupied prefix of the pool, rounded up to a power of two (everything behind it is KEY_EMPTY): Link's pool is the
  // 512-entry queue + the kNN list, InterInsert's a few dozen entries - a quarter / a sixtieth of the 4096 slots
  int P2 = 64;
  {
    const int filled = sh[0] + (a.listB ? a.degB : 0);
    while (P2 < filled && P2 < PRUNE_POOL) P2 <<= 1;
  }
  for (int size = 2; size <= P2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (P2 >> 1); i += 256) {
        const int lo = ((i / stride) * (stride << 1)) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const u64 x = pool[lo], y = pool[hi];
        if ((x > y) == up) {
          pool[lo] = y;
          pool[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  // ---- SelectEdge (nsg.cpp:655-685): walk unique candidates != v in order; keep p iff no kept r has dist(r,p) < dist(v,p)
  // count unique candidates first (needed for the "fits without pruning" case of InterInsert, nsg.cpp:640-650)
  if (tid == 0) {
    sh[1] = 0;  // nk
    sh[2] = 0;  // cursor
    sh[3] = 0;  // scanned
  }
  __syncthreads();
  const bool unlimited = a.depth <= 0;
  int uniq = 0;
  if (unlimited) {
    int local = 0;
    for (int i = tid; i < P2; i += 256) {
      const u64 key = pool[i];
      if (key != KEY_EMPTY && key_id(key) != (u32)v && (i == 0 || pool[i - 1] != key)) ++local;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if (lane == 0) sh[8 + wave] = local;
    __syncthreads();
    uniq = sh[8] + sh[9] + sh[10] + sh[11];
  }
  const bool keep_all = unlimited && uniq <= a.R;
  // The walk is sequential by definition (a candidate is judged against everything kept BEFORE it), but PB candidates are
  // evaluated per round: their vectors are staged in LDS, every kept row is streamed once against all of them (and the batch's
  // own rows against the later members of the batch), then one thread resolves the batch in order.  A quarter of the barriers
  // and of the row traffic of one candidate per round.
  constexpr int PB = 4;
  float* sqb = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(sh + 64) + 15) & ~(uintptr_t)15);   // [PB][qstride] candidate vectors (16-B aligned)
  // sh[16+b] pool position of batch member b, sh[24+b] "a row kept before the batch is closer", sh[32 + b*PB + b2] "batch member b2 (< b)
  // is closer to member b than v is"
  while (true) {
    if (tid == 0) {   // the next <= PB unique candidates
      int c = sh[2];
      int nb = 0;
      while (nb < PB && c < P2 && (unlimited || sh[3] < a.depth)) {
        const u64 key = pool[c];
        if (key == KEY_EMPTY) break;
        const bool dup = (c > 0 && pool[c - 1] == key) || key_id(key) == (u32)v;
        if (!dup) {
          sh[16 + nb] = c;
          sh[24 + nb] = 0;
          for (int b2 = 0; b2 < PB; ++b2) sh[32 + nb * PB + b2] = 0;
          ++nb;
          sh[3] += 1;
        }
        ++c;
      }
      sh[2] = c;
      sh[6] = nb;
    }
    __syncthreads();
    const int nb = sh[6];
    const int nk = sh[1];
    if (nb == 0 || nk >= a.R) break;
    if (!keep_all && (nk > 0 || nb > 1)) {
      for (int i = tid; i < PB * qstride; i += 256) {
        const int b2 = i / qstride, c = i - b2 * qstride;
        const u32 pid = key_id(pool[sh[16 + (b2 < nb ? b2 : 0)]]);
        sqb[i] = c < dim ? a.rows[(int64_t)pid * dim + c] : 0.f;
      }
      __syncthreads();
      float pd[PB];
#pragma unroll
      for (int b2 = 0; b2 < PB; ++b2) pd[b2] = b2 < nb ? key_dist(pool[sh[16 + b2]]) : -1.f;
      const int nrows = nk + nb - 1;   // kept rows, then batch members 0 .. nb-2 (member b is judged against members < b)
      for (int c0 = wave * RPW * U; c0 < nrows; c0 += 4 * RPW * U) {
        const float* rp[U];
        int ci[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ci[u] = c0 + u * RPW + g;
          ok[u] = ci[u] < nrows;
          const int cc = ok[u] ? ci[u] : 0;
          const u32 rid = cc < nk ? kept[cc] : key_id(pool[sh[16 + (cc - nk)]]);
          rp[u] = a.rows + (int64_t)rid * dim;
        }
        float acc[U][PB];
        row_dists<U, PB, VEC4>(rp, sqb, qstride, dim, 0, G, acc);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u] || t != 0) continue;
#pragma unroll
          for (int b2 = 0; b2 < PB; ++b2) {
            if (b2 >= nb || !(acc[u][b2] < pd[b2])) continue;
            if (ci[u] < nk) sh[24 + b2] = 1;
            else if (ci[u] - nk < b2) sh[32 + b2 * PB + (ci[u] - nk)] = 1;
          }
        }
      }
      __syncthreads();
    }
    if (tid == 0) {   // resolve the batch in candidate order
      int k2 = nk;
      int accepted[PB];
      for (int b2 = 0; b2 < nb && k2 < a.R; ++b2) {
        bool bad = sh[24 + b2] != 0;
        for (int b3 = 0; b3 < b2 && !bad; ++b3) bad = accepted[b3] && sh[32 + b2 * PB + b3] != 0;
        accepted[b2] = (keep_all || !bad) ? 1 : 0;
        if (accepted[b2]) {
          const u64 pkey = pool[sh[16 + b2]];
          kept[k2] = key_id(pkey);
          keptd[k2] = key_dist(pkey);
          ++k2;
        }
      }
      sh[1] = k2;
    }
    __syncthreads();
  }
  const int nk = sh[1];
  for (int i = tid; i < a.R; i += 256) {
    a.out_ids[vrow * a.R + i] = i < nk ? kept[i] : TRV_NONE;
    a.out_dist[vrow * a.R + i] = i < nk ? keptd[i] : 0.f;
  }
  if (tid == 0) a.out_deg[vrow] = (u32)nk;
}

// reverse edges: for every edge v->u offer v to u (InterInsert, nsg.cpp:583-653).  All offers are kept, in a CSR by receiving
// node (count, host prefix sum, fill): r2 first kept the first 64 arrivals per node, which is neither deterministic nor what the
// reference does for hub nodes - it considers every offer.  The order inside a node's segment is arbitrary; the prune kernel
// sorts its pool by (dist, id), so the result is not.
__global__ void rev_count_kernel(const u32* ids, const u32* deg, int64_t n, int R, u32* rev_cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * R) return;
  const int64_t v = i / R;
  if ((u32)(i - v * R) >= deg[v]) return;
  atomicAdd(&rev_cnt[ids[i]], 1u);
}
__global__ void rev_fill_kernel(const u32* ids, const float* dist, const u32* deg, int64_t n, int R, const unsigned long long* rev_off, u32* cursor,
                                u32* rev_ids, float* rev_dist) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * R) return;
  const int64_t v = i / R;
  if ((u32)(i - v * R) >= deg[v]) return;
  const u32 u = ids[i];
  const unsigned long long slot = rev_off[u] + atomicAdd(&cursor[u], 1u);
  rev_ids[slot] = (u32)v;
  rev_dist[slot] = dist[i];
}

// ------------------------------------------------------------------------------------------------ host
#define HIPCHK(expr)                                         \
  do {                                                       \
    hipError_t e__ = (expr);                                 \
    if (e__ != hipSuccess) return ix.hip_fail(e__, #expr);   \
  } while (0)

static size_t prune_lds_bytes(int dim, int R) {   // v's vector, pool, kept ids + distances, 64 scalars, 4 staged candidate vectors
  return (size_t)((dim + 3) & ~3) * 4 + (size_t)PRUNE_POOL * 8 + (size_t)R * 8 + 64 * 4 + 16 + (size_t)4 * ((dim + 3) & ~3) * 4;
}

// InterInsert (nsg.cpp:583-653), batched: every edge v->u offers v (with dist(v,u)) to u; every node then runs SelectEdge(limit =
// false) over (its own edges + the offers), which keeps everything when the unique candidates fit into out_degree.  ids / dist /
// deg: [n][R] device lists as Link leaves them.
static int32_t inter_insert_device(Index& ix, const u32* nsg_ids, const float* nsg_dist, const u32* nsg_deg, int64_t n, int R, DevBuf& out_ids,
                                   DevBuf& out_dist, DevBuf& out_deg) {
  hipStream_t s = ix.stream_;
  const int dim = (int)ix.dim_;
  const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(ix.d_rows_) & 15) == 0);
  const size_t prn_shm = prune_lds_bytes(dim, R);
  DevBuf rev_ids, rev_dist, rev_cnt, rev_off;
  if (!rev_cnt.reserve((size_t)n * 4) || !rev_off.reserve((size_t)(n + 1) * 8) || !out_ids.reserve((size_t)n * R * 4) ||
      !out_dist.reserve((size_t)n * R * 4) || !out_deg.reserve((size_t)n * 4))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (InterInsert)");
  const unsigned blocks = (unsigned)((n * R + 255) / 256);
  HIPCHK(hipMemsetAsync(rev_cnt.p, 0, (size_t)n * 4, s));
  hipLaunchKernelGGL(rev_count_kernel, dim3(blocks), dim3(256), 0, s, nsg_ids, nsg_deg, n, R, rev_cnt.as<u32>());
  std::vector<u32> h_cnt((size_t)n);
  std::vector<unsigned long long> h_off((size_t)n + 1);
  HIPCHK(hipMemcpyAsync(h_cnt.data(), rev_cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  h_off[0] = 0;
  for (int64_t v = 0; v < n; ++v) h_off[(size_t)v + 1] = h_off[(size_t)v] + h_cnt[(size_t)v];
  const size_t total = (size_t)h_off[(size_t)n];
  if (!rev_ids.reserve(std::max<size_t>(total, 1) * 4) || !rev_dist.reserve(std::max<size_t>(total, 1) * 4))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (InterInsert)");
  HIPCHK(hipMemcpyAsync(rev_off.p, h_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(rev_cnt.p, 0, (size_t)n * 4, s));   // now the fill cursors
  hipLaunchKernelGGL(rev_fill_kernel, dim3(blocks), dim3(256), 0, s, nsg_ids, nsg_dist, nsg_deg, n, R, rev_off.as<unsigned long long>(), rev_cnt.as<u32>(),
                     rev_ids.as<u32>(), rev_dist.as<float>());
  PruneArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.rows = ix.d_rows_;
  pa.dim = dim;
  pa.idsC = nsg_ids;
  pa.distC = nsg_dist;
  pa.cntC = nsg_deg;
  pa.capC = R;
  pa.idsD = rev_ids.as<u32>();
  pa.distD = rev_dist.as<float>();
  pa.offD = rev_off.as<unsigned long long>();
  pa.depth = 0;  // SelectEdge(limit = false)
  pa.R = R;
  pa.out_ids = out_ids.as<u32>();
  pa.out_dist = out_dist.as<float>();
  pa.out_deg = out_deg.as<u32>();
  for (int64_t v0 = 0; v0 < n; v0 += 1 << 20) {
    const int64_t nb = std::min<int64_t>(1 << 20, n - v0);
    pa.v0 = v0;
    if (vec4)
      hipLaunchKernelGGL((prune_kernel<true>), dim3((unsigned)nb), dim3(256), prn_shm, s, pa);
    else
      hipLaunchKernelGGL((prune_kernel<false>), dim3((unsigned)nb), dim3(256), prn_shm, s, pa);
  }
  HIPCHK(hipGetLastError());
  return EPS_OK;
}

int32_t graph_build(Index& ix, int64_t n, const eps_build_params& bp, const BuildStage* hook) {
  hipStream_t s = ix.stream_;
  const int dim = (int)ix.dim_;
  if (n <= 1) {
    if (hook && hook->stop_after) return ix.fail(EPS_USER_ERROR, "build stage: need at least two rows");
    std::vector<int64_t> off((size_t)n + 1, 0), nbr;
    return ix.set_graph(n, off.data(), nbr.data(), 0);
  }
  const int K = (int)std::min<int64_t>(bp.knng, n - 1);
  const int R = (int)bp.out_degree;
  const int Ls = (int)std::min<int64_t>(bp.search_length, n);
  if (K <= 0 || R <= 0 || Ls <= 0 || K + 1 > 1024 || R > 512)
    return ix.fail(EPS_USER_ERROR, "build: unsupported parameters (need 0 < knng < 1024, 0 < out_degree <= 512)");
  const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(ix.d_rows_) & 15) == 0);
  const bool debug = tune_env("EPS_DEBUG") != nullptr;
  struct EventPair {   // released on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evp;
  HIPCHK(hipEventCreate(&evp.a));
  HIPCHK(hipEventCreate(&evp.b));
  hipEvent_t e0 = evp.a, e1 = evp.b;
  auto lap = [&](const char* what) {
    if (!debug) return;
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "[eps build] %-28s %10.2f ms\n", what, ms);
    (void)hipEventRecord(e0, s);
  };
  (void)hipEventRecord(e0, s);

  // ---- 1. kNN graph (KNNGraph / NN-Descent in the reference, knn.hpp:90-135, nndescent.hpp:96-192: approximate there).
  // Below 65 536 rows: exact stream scan.  Above: every block of B rows goes through the matrix filter in its approximate-key
  // mode (8-bit operands, r3), which selects the kA = 128 closest rows by approximate key, and those 128 are re-ranked in exact
  // fp32 - the lists are the exact K nearest unless a true neighbour's approximate rank falls beyond 128 (the approximate key's
  // error is a small fraction of the gap between the K-th and the 128-th neighbour; tests/test_gpu_build.py measures the recall).
  DevBuf knn, run, runA, candA, cntA;
  const int k1 = K + 1;
  const int64_t B = std::max(256, tune_int("EPS_BUILD_BLOCK", 2048));   // queries per kNN pass
  const bool use_mfma = n >= 65536;
  const int kA = (int)std::min<int64_t>(n, std::max(k1, 128));
  if (!knn.reserve((size_t)n * K * 4) || !run.reserve((size_t)B * k1 * 8)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (kNN graph)");
  if (hook && hook->knn_in) {   // stage entry: the caller's kNN graph instead
    DevBuf tmp;
    if (!tmp.reserve((size_t)n * K * 8)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (kNN graph)");
    HIPCHK(hipMemcpyAsync(tmp.p, hook->knn_in, (size_t)n * K * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(knn_import_kernel, dim3((unsigned)(((int64_t)n * K + 255) / 256)), dim3(256), 0, s, tmp.as<int64_t>(), (int64_t)n * K, n, knn.as<u32>());
    HIPCHK(hipStreamSynchronize(s));
  } else {
    if (use_mfma && (!runA.reserve((size_t)B * kA * 8) || !candA.reserve((size_t)B * kA * 4) || !cntA.reserve((size_t)B * 4)))
      return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (kNN graph)");
    ix.scan_limit_ = n;  // the graph covers rows [0,n) only
    for (int64_t q0 = 0; q0 < n; q0 += B) {
      const int64_t nq = std::min(B, n - q0);
      const float* dq = ix.d_rows_ + q0 * dim;
      int32_t rc;
      if (use_mfma) {
        rc = flat_mfma_search(ix, dq, nq, kA, runA.as<u64>(), true);
        if (rc == EPS_OK) {
          hipLaunchKernelGGL(knn_keys_to_cand_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, runA.as<u64>(), kA, nq, candA.as<u32>(), cntA.as<u32>());
          launch_fill_u64(run.as<u64>(), nq * k1, KEY_EMPTY, s);
          RerankArgs ra;
          ra.rows = ix.d_rows_;
          ra.dim = dim;
          ra.metric = ix.metric_;
          ra.queries = dq;
          ra.nq = nq;
          ra.k = k1;
          ra.f = no_filter();
          ra.cand = candA.as<u32>();
          ra.cand_count = cntA.as<u32>();
          ra.cap = kA;
          ra.run_keys = run.as<u64>();
          ra.fuse = 0;
          launch_rerank(ra, s);
        }
      } else {
        rc = ix.flat_stream(dq, nq, k1, 0, n, run.as<u64>(), false, -1, false);
      }
      if (rc != EPS_OK) {
        ix.scan_limit_ = -1;
        return rc;
      }
      hipLaunchKernelGGL(knn_extract_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, run.as<u64>(), k1, q0, nq, K,
                         knn.as<u32>());
    }
    ix.scan_limit_ = -1;
  }
  lap("kNN graph");
  if (hook && hook->stop_after == 1) {
    DevBuf tmp;
    if (!tmp.reserve((size_t)n * K * 8)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (kNN graph)");
    hipLaunchKernelGGL(knn_export_kernel, dim3((unsigned)(((int64_t)n * K + 255) / 256)), dim3(256), 0, s, knn.as<u32>(), (int64_t)n * K, tmp.as<int64_t>());
    HIPCHK(hipMemcpyAsync(hook->out_ids, tmp.p, (size_t)n * K * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return EPS_OK;
  }

  // ---- 2. navigation node: closest row to the centroid (always L2)
  DevBuf cen;
  if (!cen.reserve((size_t)dim * 4 + 64)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory");
  HIPCHK(hipMemsetAsync(cen.p, 0, (size_t)dim * 4, s));
  {
    const int64_t rpb = std::max<int64_t>(64, (n + 2047) / 2048);
    hipLaunchKernelGGL(centroid_kernel, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, s, ix.d_rows_, n, dim, rpb, cen.as<float>());
    hipLaunchKernelGGL(scale_kernel, dim3((dim + 255) / 256), dim3(256), 0, s, cen.as<float>(), dim, 1.0f / (float)n);
  }
  int32_t rc = ix.flat_stream(cen.as<float>(), 1, 1, 0, n, run.as<u64>(), false, 0, false);
  if (rc != EPS_OK) return rc;
  u64 navkey = 0;
  HIPCHK(hipMemcpyAsync(&navkey, run.p, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const int64_t nav = (hook && hook->nav_in >= 0 && hook->nav_in < n) ? hook->nav_in : (int64_t)key_id(navkey);
  lap("navigation node");

  // seeds of every Link search: the first Ls kNN entries of nav (GetNeighbors, nsg.cpp:174-195), then nav+1, ...
  std::vector<u32> seeds;
  {
    std::vector<u32> row((size_t)K);
    HIPCHK(hipMemcpyAsync(row.data(), knn.as<u32>() + nav * K, (size_t)K * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<uint8_t> sel((size_t)n, 0);
    for (int i = 0; i < K && (int)seeds.size() < Ls; ++i)
      if (row[i] != TRV_NONE && !sel[row[i]]) {
        sel[row[i]] = 1;
        seeds.push_back(row[i]);
      }
    int64_t tmp = nav + 1;
    while ((int)seeds.size() < Ls) {
      if (tmp >= n) tmp = 0;
      const int64_t v = tmp++;
      if (sel[v]) continue;
      sel[v] = 1;
      seeds.push_back((u32)v);
    }
  }
  DevBuf d_seeds, logb, logc, counters, nsg_ids, nsg_dist, nsg_deg, visb;
  // visited set of the Link / connectivity searches (the reference's has_calculated bitset, n bits per search):
  //   default        a hash table of 32768 slots per search in HBM (128 KB: the ~1000 searches in flight keep theirs in L2 /
  //                  Infinity Cache), reset per batch of 16384 searches; a search that fills 3/4 of it stops discovering -
  //                  four times the evaluations any search of the 10M x 768 build makes (5.9 k on average)
  //   EPS_BUILD_VISITED=bitmap   an exact n-bit bitmap per search, zeroed per batch (batches sized to 2 GiB: 1717 searches at
  //                  10M rows, where the memsets and the HBM atomics cost +20 s of Link: 56 s vs 36 s)
  //   EPS_BUILD_VISITED=lds      the r1 table of 8192 slots in LDS: a third of the searches at 1M x 768 fill it (5.4 k
  //                  evaluations instead of 5.8 k); same recall
  const char* vis_env = tune_env("EPS_BUILD_VISITED");
  const bool bitmap_vis = vis_env && std::strcmp(vis_env, "bitmap") == 0;
  const bool ghash_vis = !bitmap_vis && !(vis_env && std::strcmp(vis_env, "lds") == 0);
  constexpr int GHASH_SLOTS = 32768;
  const int64_t vis_words = (n + 31) / 32;
  int64_t NB = std::min<int64_t>(n, 16384);
  if (bitmap_vis) NB = std::max<int64_t>(256, std::min<int64_t>(NB, ((int64_t)2 << 30) / (vis_words * 4)));
  if (bitmap_vis && !visb.reserve((size_t)NB * vis_words * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (visited bitmaps)");
  if (ghash_vis && !visb.reserve((size_t)NB * GHASH_SLOTS * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (visited tables)");
  if (!d_seeds.reserve((size_t)Ls * 4) || !logb.reserve((size_t)NB * 2048 * 8) || !logc.reserve((size_t)NB * 4) ||
      !counters.reserve(32) || !nsg_ids.reserve((size_t)n * R * 4) || !nsg_dist.reserve((size_t)n * R * 4) ||
      !nsg_deg.reserve((size_t)n * 4))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (NSG link)");
  HIPCHK(hipMemcpyAsync(d_seeds.p, seeds.data(), (size_t)Ls * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(counters.p, 0, 32, s));
  // The search keeps the Lcap closest evaluated nodes (its queue) but expands only among the first Ls = search_length of them,
  // which is the reference's search; the queue is what SyncPrune gets as `fullset`: its SelectEdge looks at the first
  // candidate_pool_size entries of the sorted pool only, so Lcap >= candidate_pool_size loses nothing.
  int Lp2 = 512;
  while (Lp2 < Ls || (bp.candidate_pool_size > 0 && Lp2 < bp.candidate_pool_size)) Lp2 <<= 1;
  if (bp.candidate_pool_size <= 0 || Lp2 > 2048) Lp2 = 2048;   // (unlimited depth: the 2048 closest)
  if (Lp2 < Ls) return ix.fail(EPS_DB_UNSUPPORTED_ERROR, "build: search_length > 2048 is not supported");
  const int Lcap = Lp2;
  // 8-bit lower-bound prefilter of the searches' distance step (traverse_kernel.hpp, step 3a): the mirror is the one the kNN stage
  // just scanned; EPS_BUILD_PREFILTER=0 turns it off (A/B - the graph is the same either way)
  Quant8View q8v;
  bool prefilter = dim >= 128;
  if (const char* e = tune_env("EPS_BUILD_PREFILTER")) prefilter = atoi(e) != 0;
  DevBuf q8b, qstat8b;
  if (prefilter) {
    const int32_t rc = quant8_view(ix, &q8v);
    if (rc != EPS_OK) q8v = Quant8View();   // (optional: without the mirror the searches read the fp32 rows, the graph is the same)
    prefilter = q8v.x8 != nullptr;
  }
  const size_t trv_shm = traverse_lds_bytes(dim, Lp2, true, prefilter, q8v.cols8);
  const size_t trv_shm_bm = traverse_lds_bytes(dim, Lp2, false, prefilter, q8v.cols8);
  const size_t prn_shm = prune_lds_bytes(dim, R);
  TraverseArgs ta;
  ta.rows = ix.d_rows_;
  ta.dim = dim;
  ta.metric = 0;
  ta.off = nullptr;
  ta.nbr = knn.as<u32>();
  ta.fixed_deg = K;
  ta.init_ids = d_seeds.as<u32>();
  ta.L = Lcap;
  ta.Lp2 = Lp2;
  ta.nseeds = Ls;
  ta.Lsel = Ls;
  ta.M = 1;
  ta.visited = bitmap_vis ? visb.as<u32>() : nullptr;
  ta.words = bitmap_vis ? vis_words : 0;
  ta.ghash = ghash_vis ? visb.as<u32>() : nullptr;
  ta.hslots = ghash_vis ? GHASH_SLOTS : 0;
  ta.out_queue = nullptr;
  ta.counters = counters.as<unsigned long long>();
  ta.counters_n = 4;
  ta.log = logb.as<u64>();
  ta.log_cnt = logc.as<u32>();
  ta.log_cap = Lcap;
  ta.x8 = nullptr;
  ta.acc0 = q8v.acc0;
  ta.scal8 = q8v.scal8;
  ta.q8 = nullptr;
  ta.qstat8 = nullptr;
  ta.d_pad8 = q8v.d_pad8;
  ta.cols8 = q8v.cols8;
  ta.u8 = q8v.u;
  {
    const int G = group_lanes(dim, vec4);
    ta.slack8 = std::max(8e-6f, 2.f * (3.f * ((float)((dim + G - 1) / G) + 6.f) + 6.f) * 5.9604645e-8f);
  }
  if (prefilter) {
    if (!q8b.reserve((size_t)NB * q8v.d_pad8) || !qstat8b.reserve((size_t)NB * 16)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory");
    ta.x8 = q8v.x8;
    ta.q8 = q8b.as<signed char>();
    ta.qstat8 = qstat8b.as<float>();
  }
  PruneArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.rows = ix.d_rows_;
  pa.dim = dim;
  pa.log = logb.as<u64>();
  pa.log_cnt = logc.as<u32>();
  pa.log_cap = Lcap;
  pa.listB = knn.as<u32>();
  pa.degB = K;
  pa.depth = (int)bp.candidate_pool_size;
  pa.R = R;
  pa.out_ids = nsg_ids.as<u32>();
  pa.out_dist = nsg_dist.as<float>();
  pa.out_deg = nsg_deg.as<u32>();
  // ---- 3. Link
  auto launch_link_search = [&](int64_t nb) -> hipError_t {
    if (bitmap_vis) {   // exact visited set: one n-bit bitmap per search of the batch, zeroed per batch
      hipError_t e = hipMemsetAsync(visb.p, 0, (size_t)nb * vis_words * 4, s);
      if (e != hipSuccess) return e;
      if (vec4)
        hipLaunchKernelGGL((traverse_kernel<true, false, true, 4>), dim3((unsigned)nb), dim3(256), trv_shm_bm, s, ta);
      else
        hipLaunchKernelGGL((traverse_kernel<false, false, true, 4>), dim3((unsigned)nb), dim3(256), trv_shm_bm, s, ta);
    } else {
      const size_t shm = ghash_vis ? trv_shm_bm : trv_shm;   // (the LDS layout without the table)
      if (ghash_vis) {
        hipError_t e = hipMemsetAsync(visb.p, 0xFF, (size_t)nb * GHASH_SLOTS * 4, s);   // TRV_NONE in every slot
        if (e != hipSuccess) return e;
      }
      if (vec4)
        hipLaunchKernelGGL((traverse_kernel<true, true, true, 4>), dim3((unsigned)nb), dim3(256), shm, s, ta);
      else
        hipLaunchKernelGGL((traverse_kernel<false, true, true, 4>), dim3((unsigned)nb), dim3(256), shm, s, ta);
    }
    return hipSuccess;
  };
  for (int64_t v0 = 0; v0 < n; v0 += NB) {
    const int64_t nb = std::min(NB, n - v0);
    ta.queries = ix.d_rows_ + v0 * dim;
    if (prefilter) quant8_queries(ix, q8v, ta.queries, nb, q8b.as<signed char>(), qstat8b.as<float>());
    HIPCHK(launch_link_search(nb));
    if (v0 == 0 && prefilter && !tune_env("EPS_BUILD_PREFILTER") && n > NB) {
      // judged on the first batch: where more than 60 % of the neighbour evaluations still read the fp32 row (distances small
      // against the table's value range: clustered / low intrinsic dimension) the 8-bit test is pure overhead - off for the rest
      unsigned long long hc[4] = {0, 0, 0, 0};
      HIPCHK(hipMemcpyAsync(hc, counters.p, 32, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      const double nbr_evals = (double)hc[0] - (double)nb * Ls;
      if (nbr_evals > 0 && (double)hc[3] > 0.6 * nbr_evals) {
        prefilter = false;
        ta.x8 = nullptr;
        if (debug) fprintf(stderr, "[eps build] Link: 8-bit prefilter switched off after the first batch (%.0f %% of the neighbour evaluations passed it)\n", 100.0 * (double)hc[3] / nbr_evals);
      }
    }
    pa.v0 = v0;
    if (vec4)
      hipLaunchKernelGGL((prune_kernel<true>), dim3((unsigned)nb), dim3(256), prn_shm, s, pa);
    else
      hipLaunchKernelGGL((prune_kernel<false>), dim3((unsigned)nb), dim3(256), prn_shm, s, pa);
  }
  HIPCHK(hipGetLastError());
  lap("Link (search + SelectEdge)");
  if (hook && hook->stop_after == 2) {   // what Link leaves: per node <= R edges, before InterInsert
    std::vector<u32> li((size_t)n * R), ld((size_t)n);
    HIPCHK(hipMemcpyAsync(li.data(), nsg_ids.p, (size_t)n * R * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ld.data(), nsg_deg.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int64_t v = 0; v < n; ++v) {
      hook->out_deg[v] = (int32_t)ld[v];
      for (int j = 0; j < R; ++j) hook->out_ids[v * R + j] = j < (int)ld[v] ? (int64_t)li[(size_t)v * R + j] : -1;
    }
    if (hook->nav_out) *hook->nav_out = nav;
    return EPS_OK;
  }
  if (debug) {
    unsigned long long hcnt[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(hcnt, counters.p, 32, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    fprintf(stderr, "[eps build] Link: %.0f evaluations, %.1f expansions per search; queue of %d kept as the pool; visited set: %s, %llu searches filled it\n",
            (double)hcnt[0] / (double)n, (double)hcnt[1] / (double)n, Lcap, bitmap_vis ? "bitmap" : (ghash_vis ? "hash of 32768 in HBM" : "hash of 8192 in LDS"), hcnt[2]);
    if (prefilter) fprintf(stderr, "[eps build] Link: 8-bit prefilter on, %.0f neighbour evaluations per search read the fp32 row\n", (double)hcnt[3] / (double)n);
  }

  // ---- 4. InterInsert
  DevBuf out_ids, out_dist, out_deg;
  {
    const int32_t rc = inter_insert_device(ix, nsg_ids.as<u32>(), nsg_dist.as<float>(), nsg_deg.as<u32>(), n, R, out_ids, out_dist, out_deg);
    if (rc != EPS_OK) return rc;
  }
  lap("InterInsert");

  // ---- 5. connectivity on the host
  std::vector<u32> h_ids((size_t)n * R), h_deg((size_t)n);
  HIPCHK(hipMemcpyAsync(h_ids.data(), out_ids.p, (size_t)n * R * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h_deg.data(), out_deg.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  std::vector<uint8_t> reached((size_t)n, 0);
  {
    std::vector<u32> stack;
    stack.push_back((u32)nav);
    reached[nav] = 1;
    while (!stack.empty()) {
      const u32 x = stack.back();
      stack.pop_back();
      for (u32 j = 0; j < h_deg[x]; ++j) {
        const u32 y = h_ids[(size_t)x * R + j];
        if (!reached[y]) {
          reached[y] = 1;
          stack.push_back(y);
        }
      }
    }
  }
  std::vector<u32> orphans;
  for (int64_t i = 0; i < n; ++i)
    if (!reached[i]) orphans.push_back((u32)i);
  // FindUnconnectedNode (nsg.cpp:734-775): the lowest-id unreached node x gets an in-edge from the closest REACHED node
  // of a search for V[x]; the DFS then continues through x, so only one node per unreached component receives an
  // extra edge.  Here the searches of all initially unreached nodes run as one device batch (their 32 closest
  // evaluated nodes are kept), and the attach + DFS-continue loop runs on the host in id order against the live
  // `reached` set.
  constexpr int TRACE_K = 32;
  std::vector<std::pair<u32, u32>> extra;   // (root, orphan) edges
  const size_t n_orphans = orphans.size();
  if (!orphans.empty()) {
    std::vector<u32> nseeds;
    {
      std::vector<uint8_t> sel((size_t)n, 0);
      for (u32 j = 0; j < h_deg[nav] && (int)nseeds.size() < Ls; ++j) {
        const u32 y = h_ids[(size_t)nav * R + j];
        if (!sel[y]) {
          sel[y] = 1;
          nseeds.push_back(y);
        }
      }
      int64_t tmp = nav + 1;
      while ((int)nseeds.size() < Ls) {
        if (tmp >= n) tmp = 0;
        const int64_t v = tmp++;
        if (sel[v]) continue;
        sel[v] = 1;
        nseeds.push_back((u32)v);
      }
    }
    DevBuf d_orph, d_q, d_top;
    const int64_t m = (int64_t)orphans.size();
    const int64_t OB = std::min<int64_t>(m, NB);
    if (!d_orph.reserve((size_t)m * 4) || !d_q.reserve((size_t)OB * dim * 4) || !d_top.reserve((size_t)OB * TRACE_K * 8))
      return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "build: out of device memory (connectivity)");
    HIPCHK(hipMemcpyAsync(d_orph.p, orphans.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_seeds.p, nseeds.data(), (size_t)Ls * 4, hipMemcpyHostToDevice, s));
    ta.nbr = out_ids.as<u32>();
    ta.fixed_deg = R;
    std::vector<u64> top((size_t)m * TRACE_K);
    for (int64_t o0 = 0; o0 < m; o0 += OB) {
      const int64_t nb = std::min(OB, m - o0);
      hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nb), dim3(256), 0, s, ix.d_rows_, d_orph.as<u32>() + o0, nb, dim, d_q.as<float>());
      ta.queries = d_q.as<float>();
      if (prefilter) quant8_queries(ix, q8v, ta.queries, nb, q8b.as<signed char>(), qstat8b.as<float>());
      HIPCHK(launch_link_search(nb));
      // the TRACE_K closest evaluated nodes of every search, ascending
      launch_merge_lists(logb.as<u64>(), Lcap, TRACE_K, nb, d_top.as<u64>(), false, s, logc.as<u32>());
      HIPCHK(hipMemcpyAsync(top.data() + (size_t)o0 * TRACE_K, d_top.p, (size_t)nb * TRACE_K * 8, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
    }
    // Out-degree stays <= 64 (the reference has no cap: on hub-prone data its repair edges pile up on a few nodes -
    // 291 edges on one node at 1M x 768): a saturated node passes the orphan on to the next closest reached node of the
    // trace, so every adjacency list keeps the fixed 256-byte stride the traversal kernel reads with one row fetch.
    const u32 deg_cap = (u32)std::max(64, R + 1);
    std::vector<u32> live_extra((size_t)n, 0);
    std::vector<u32> stack;
    uint64_t lcg = bp.seed ? bp.seed : 100;
    for (int64_t i = 0; i < m; ++i) {
      const u32 x = orphans[i];
      if (reached[x]) continue;   // linked through an earlier orphan's subtree
      int64_t root = -1;
      for (int e = 0; e < TRACE_K; ++e) {
        const u64 key = top[(size_t)i * TRACE_K + e];
        if (key == KEY_EMPTY) break;
        const u32 c = key_id(key);
        if (reached[c] && h_deg[c] + live_extra[c] < deg_cap) {
          root = c;
          break;
        }
      }
      while (root < 0) {          // a random linked node (nsg.cpp:763-771), here one that still has room
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        const u32 c = (u32)((lcg >> 33) % (uint64_t)n);
        if (reached[c] && h_deg[c] + live_extra[c] < deg_cap) root = c;
      }
      live_extra[root]++;
      extra.emplace_back((u32)root, x);
      reached[x] = 1;
      stack.push_back(x);
      while (!stack.empty()) {
        const u32 v = stack.back();
        stack.pop_back();
        for (u32 j = 0; j < h_deg[v]; ++j) {
          const u32 y = h_ids[(size_t)v * R + j];
          if (!reached[y]) {
            reached[y] = 1;
            stack.push_back(y);
          }
        }
      }
    }
  }
  lap("connectivity");

  // ---- CSR in the reference's layout (ann_graph_segment.cpp:222-241)
  std::vector<int64_t> off((size_t)n + 1);
  std::vector<u32> extra_cnt((size_t)n, 0);
  for (auto& pr : extra) extra_cnt[pr.first]++;
  int64_t e = 0;
  for (int64_t i = 0; i < n; ++i) {
    off[i] = e;
    e += h_deg[i] + extra_cnt[i];
  }
  off[n] = e;
  std::vector<int64_t> nbr((size_t)e);
  std::vector<int64_t> fill((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    for (u32 j = 0; j < h_deg[i]; ++j) nbr[off[i] + j] = h_ids[(size_t)i * R + j];
    fill[i] = off[i] + h_deg[i];
  }
  for (auto& pr : extra) nbr[fill[pr.first]++] = pr.second;
  if (debug) {
    u32 maxdeg = 0;
    for (int64_t i = 0; i < n; ++i) maxdeg = std::max(maxdeg, h_deg[i] + extra_cnt[i]);
    fprintf(stderr, "[eps build] n=%lld edges=%lld avg degree %.1f max degree %u, %zu nodes unreached after InterInsert, %zu repair edges, nav %lld\n",
            (long long)n, (long long)e, (double)e / n, maxdeg, n_orphans, extra.size(), (long long)nav);
  }
  return ix.set_graph(n, off.data(), nbr.data(), nav);
}

// SyncPrune's sort + SelectEdge (nsg.cpp:557-567, 655-685) for caller-supplied candidate lists: node nodes[i] with
// candidates cands[i][0..cands_per_node) (-1 = none; the node itself is skipped) -> its <= out_degree pruned out-edges.
// The stage of the build that is deterministic given its input, exposed so that it can be pinned against the reference's
// SelectEdge on identical pools (tests/test_gpu_build.py).
int32_t select_edges(Index& ix, const int64_t* nodes, int64_t m, const int64_t* cands, int32_t cpn, int32_t depth, int32_t R,
                     int64_t* out_ids, int32_t* out_deg) {
  if (m <= 0) return EPS_OK;
  if (!nodes || !cands || !out_ids || !out_deg || cpn <= 0 || cpn > PRUNE_POOL || R <= 0 || R > 512)
    return ix.fail(EPS_USER_ERROR, "select_edges: bad arguments");
  hipStream_t s = ix.stream_;
  const int dim = (int)ix.dim_;
  std::vector<u32> h_nodes((size_t)m), h_c((size_t)m * cpn);
  for (int64_t i = 0; i < m; ++i) {
    if (nodes[i] < 0 || nodes[i] >= ix.n_rows_) return ix.fail(EPS_USER_ERROR, "select_edges: node out of range");
    h_nodes[i] = (u32)nodes[i];
    for (int j = 0; j < cpn; ++j) {
      const int64_t c = cands[i * cpn + j];
      if (c >= ix.n_rows_) return ix.fail(EPS_USER_ERROR, "select_edges: candidate out of range");
      h_c[(size_t)i * cpn + j] = c < 0 ? TRV_NONE : (u32)c;
    }
  }
  DevBuf d_nodes, d_c, d_oi, d_od, d_deg;
  if (!d_nodes.reserve((size_t)m * 4) || !d_c.reserve((size_t)m * cpn * 4) || !d_oi.reserve((size_t)m * R * 4) || !d_od.reserve((size_t)m * R * 4) ||
      !d_deg.reserve((size_t)m * 4))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "select_edges: out of device memory");
  HIPCHK(hipMemcpyAsync(d_nodes.p, h_nodes.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d_c.p, h_c.data(), (size_t)m * cpn * 4, hipMemcpyHostToDevice, s));
  PruneArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.rows = ix.d_rows_;
  pa.dim = dim;
  pa.node_ids = d_nodes.as<u32>();
  pa.listB = d_c.as<u32>();
  pa.degB = cpn;
  pa.depth = depth;
  pa.R = R;
  pa.out_ids = d_oi.as<u32>();
  pa.out_dist = d_od.as<float>();
  pa.out_deg = d_deg.as<u32>();
  const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(ix.d_rows_) & 15) == 0);
  const size_t shm = prune_lds_bytes(dim, R);
  if (vec4)
    hipLaunchKernelGGL((prune_kernel<true>), dim3((unsigned)m), dim3(256), shm, s, pa);
  else
    hipLaunchKernelGGL((prune_kernel<false>), dim3((unsigned)m), dim3(256), shm, s, pa);
  HIPCHK(hipGetLastError());
  std::vector<u32> h_oi((size_t)m * R), h_deg((size_t)m);
  HIPCHK(hipMemcpyAsync(h_oi.data(), d_oi.p, (size_t)m * R * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h_deg.data(), d_deg.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int64_t i = 0; i < m; ++i) {
    out_deg[i] = (int32_t)h_deg[i];
    for (int j = 0; j < R; ++j) out_ids[i * R + j] = (u32)j < h_deg[i] ? (int64_t)h_oi[(size_t)i * R + j] : -1;
  }
  return EPS_OK;
}

// edge distances dist(v, ids[v][j]) for the lists handed to inter_insert() (one wavefront per edge)
__global__ __launch_bounds__(256) void edge_dist_kernel(const float* rows, int dim, const u32* ids, const u32* deg, int64_t n, int R, float* dist) {
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n * R) return;
  const int64_t v = e / R;
  const int j = (int)(e - v * R);
  if ((u32)j >= deg[v]) return;
  const float* a = rows + v * dim;
  const float* b = rows + (int64_t)ids[e] * dim;
  float acc = 0.f;
  for (int c = lane_id(); c < dim; c += 64) {
    const float t = a[c] - b[c];
    acc = fmaf(t, t, acc);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) dist[e] = acc;
}

// The InterInsert stage alone on caller-supplied edge lists (like-for-like build parity, tests/test_gpu_build.py):
// ids [n][R] (deg[v] valid entries each) -> out_ids [n][R] (-1 padded), out_deg.  n must equal the attached row count.
int32_t inter_insert(Index& ix, const int64_t* ids, const int32_t* deg, int64_t n, int32_t R, int64_t* out_ids, int32_t* out_deg) {
  if (n <= 0) return EPS_OK;
  if (!ids || !deg || !out_ids || !out_deg || R <= 0 || R > 512 || n != ix.n_rows_) return ix.fail(EPS_USER_ERROR, "inter_insert: bad arguments");
  hipStream_t s = ix.stream_;
  std::vector<u32> h_ids((size_t)n * R, TRV_NONE), h_deg((size_t)n);
  for (int64_t v = 0; v < n; ++v) {
    if (deg[v] < 0 || deg[v] > R) return ix.fail(EPS_USER_ERROR, "inter_insert: degree out of range");
    h_deg[v] = (u32)deg[v];
    for (int j = 0; j < deg[v]; ++j) {
      const int64_t u = ids[v * R + j];
      if (u < 0 || u >= n) return ix.fail(EPS_USER_ERROR, "inter_insert: neighbour out of range");
      h_ids[(size_t)v * R + j] = (u32)u;
    }
  }
  DevBuf d_ids, d_dist, d_deg, o_ids, o_dist, o_deg;
  if (!d_ids.reserve((size_t)n * R * 4) || !d_dist.reserve((size_t)n * R * 4) || !d_deg.reserve((size_t)n * 4))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "inter_insert: out of device memory");
  HIPCHK(hipMemcpyAsync(d_ids.p, h_ids.data(), (size_t)n * R * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d_deg.p, h_deg.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(edge_dist_kernel, dim3((unsigned)((n * R + 3) / 4)), dim3(256), 0, s, ix.d_rows_, (int)ix.dim_, d_ids.as<u32>(), d_deg.as<u32>(), n, (int)R,
                     d_dist.as<float>());
  const int32_t rc = inter_insert_device(ix, d_ids.as<u32>(), d_dist.as<float>(), d_deg.as<u32>(), n, R, o_ids, o_dist, o_deg);
  if (rc != EPS_OK) return rc;
  std::vector<u32> r_ids((size_t)n * R), r_deg((size_t)n);
  HIPCHK(hipMemcpyAsync(r_ids.data(), o_ids.p, (size_t)n * R * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(r_deg.data(), o_deg.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int64_t v = 0; v < n; ++v) {
    out_deg[v] = (int32_t)r_deg[v];
    for (int j = 0; j < R; ++j) out_ids[v * R + j] = (u32)j < r_deg[v] ? (int64_t)r_ids[(size_t)v * R + j] : -1;
  }
  return EPS_OK;
}

}  // namespace eps
