// Batched flat scan on the matrix cores: the large-batch form of VecSearchExecutor::BruteForceSearch
// (reference: engine/db/execution/vec_search_executor.cpp:717-768), returning the SAME exact answer as the
// fp32 streaming scan.
//
// Idea: at batch b the flat scan is a GEMM, 2*b*N*d flops over N*d row elements.
// gfx950 has no fast fp32/xf32 MFMA (fp32-in MFMA runs at the vector rate), so the GEMM runs on an fp16 mirror
// of the row store with fp32 accumulation and is used only as a LOWER-BOUND FILTER:
//     key(q,x) = |x|^2 - 2 q.x   (L2; -q.x for IP/COSINE)        exact key, dist = key + const(q)
//     approx   = base[x] + s*acc,  acc = sum_k qh_k xh_k         what the MFMA tile produces
//     |key - approx| <= |s| * (|q| * E1[x] + |q - qh| * |xh|)    E1[x] = |x - xh| + gamma*|xh|   (Cauchy-Schwarz +
//                                                                 fp32 accumulation slack), no distributional assumption
// A row can only be in the exact top-k if approx - bound <= T, where T is ANY valid upper bound of the k-th best
// exact key — we use the k-th best exact key found so far.  Rows that pass are re-ranked in exact fp32 by the
// gather kernel (flat_kernels.hip: rerank_kernel).  The scan is staged so T tightens:
//     stage 0: exact fp32 stream scan of the first S0 rows            -> running top-k
//     stage i: MFMA filter over the next, geometrically larger chunk  -> candidates -> exact re-rank -> top-k
// Expected candidates per query ~ k * sum_i (chunk_i / rows_before_i): a few hundred at N = 10M, k = 10.
// If a query's candidate buffer overflows (adversarial order/duplicates) the batch falls back to the fp32 scan,
// so the result is exact in every case.
//
// Kernel: 128 rows x 128 queries per workgroup, 4 wavefronts (2x2) x 64x64 outputs each = 2x2 tiles of
// v_mfma_f32_32x32x16_f16; K-step 64; both operands K-contiguous, staged HBM->LDS by global_load_lds (16 B/lane,
// double-buffered, next tile in flight under the MFMAs); LDS rows XOR-swizzled on the source address so the
// ds_read_b128 fragment reads are conflict-free; workgroup->tile map keeps the 8 query tiles of one row tile on
// one XCD back-to-back so the row tile is fetched from HBM once and re-read from that XCD's L2.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "index.hpp"

namespace eps {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 64;   // v1 tile
constexpr int BM2 = 256;                      // v2 row tile (256 rows x 128 queries, 8 wavefronts)
constexpr int ROWPAD = 256;                   // mirror rows are padded to this

struct HalfMirror {
  DevBuf xh;       // _Float16 [n_pad][d_pad]
  DevBuf xn;       // float [n_pad]  |x|^2 (+inf on padding rows)
  DevBuf zeros;    // float [n_pad]  base for IP / COSINE (+inf on padding rows)
  DevBuf scal;     // float [4]: E1max, nxh_max, xn_max, overflow flag (as float bits)
  DevBuf qh;       // _Float16 [b_pad][d_pad]
  DevBuf qstat;    // float [b_pad][4]: |q|^2, |q|, |q-qh|, unused
  DevBuf T;        // float [b_pad]
  DevBuf cand;     // u32 [b][cap]
  DevBuf cnt;      // u32 [b] + overflow counter at [b]
  int64_t version = -1;
  int64_t n = 0, n_pad = 0;
  int d_pad = 0;
  bool fp16_range_ok = true;
  float h_scal[4] = {0, 0, 0, 0};
};

void half_mirror_free(HalfMirror* m) { delete m; }

// ------------------------------------------------------------------------------------------------ mirror build
__device__ __forceinline__ void atomic_max_pos(float* addr, float v) {  // v >= 0
  atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void half_mirror_kernel(const float* rows, int64_t n, int64_t n_pad, int dim, int d_pad,
                                                          _Float16* xh, float* xn, float* zeros, float* scal, float gamma) {
  // one wavefront per row, grid-stride over rows; the four per-index maxima are reduced in registers and
  // published with ONE atomic per wavefront (an atomic per row serialises 10M rows on four addresses)
  const int lane = lane_id();
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float m_e1 = 0.f, m_nxh = 0.f, m_xn = 0.f, m_bad = 0.f;
  const bool vec = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_pad; r += nwaves) {
    _Float16* dst = xh + r * d_pad;
    if (r >= n) {
      for (int c = lane; c < d_pad; c += 64) dst[c] = (_Float16)0.f;
      if (lane == 0) {
        xn[r] = __builtin_inff();
        zeros[r] = __builtin_inff();
      }
      continue;
    }
    const float* src = rows + r * dim;
    float s2 = 0.f, e2 = 0.f, h2 = 0.f, mx = 0.f;
    if (vec) {  // 16 B/lane loads, 8 B/lane stores (d_pad is a multiple of 64, so c + 3 < d_pad)
      for (int c = lane * 4; c < d_pad; c += 256) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < dim) x = *reinterpret_cast<const float4*>(src + c);
        half4 h;
        h[0] = (_Float16)x.x; h[1] = (_Float16)x.y; h[2] = (_Float16)x.z; h[3] = (_Float16)x.w;
        *reinterpret_cast<half4*>(dst + c) = h;
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float hf = (float)h[e];
          s2 = fmaf(xs[e], xs[e], s2);
          const float er = xs[e] - hf;
          e2 = fmaf(er, er, e2);
          h2 = fmaf(hf, hf, h2);
          mx = fmaxf(mx, fabsf(xs[e]));
        }
      }
    } else {
      for (int c = lane; c < d_pad; c += 64) {
        const float x = c < dim ? src[c] : 0.f;
        const _Float16 h = (_Float16)x;
        const float hf = (float)h;
        dst[c] = h;
        s2 = fmaf(x, x, s2);
        const float e = x - hf;
        e2 = fmaf(e, e, e2);
        h2 = fmaf(hf, hf, h2);
        mx = fmaxf(mx, fabsf(x));
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      s2 += __shfl_xor(s2, o);
      e2 += __shfl_xor(e2, o);
      h2 += __shfl_xor(h2, o);
      mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) {
      xn[r] = s2;
      zeros[r] = 0.f;
    }
    const float nxh = sqrtf(h2) * 1.000001f;
    m_e1 = fmaxf(m_e1, sqrtf(e2) * 1.000001f + gamma * nxh);
    m_nxh = fmaxf(m_nxh, nxh);
    m_xn = fmaxf(m_xn, s2);
    if (!(mx <= 65504.f) || s2 != s2) m_bad = 1.f;  // beyond the fp16 range, or NaN
  }
  if (lane == 0) {
    atomic_max_pos(&scal[0], m_e1);
    atomic_max_pos(&scal[1], m_nxh);
    atomic_max_pos(&scal[2], m_xn);
    if (m_bad != 0.f) atomic_max_pos(&scal[3], 1.f);
  }
}

__global__ __launch_bounds__(256) void query_prep_kernel(const float* q, int64_t nq, int64_t b_pad, int dim, int d_pad,
                                                         _Float16* qh, float* qstat) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= b_pad) return;
  const int lane = lane_id();
  _Float16* dst = qh + r * d_pad;
  if (r >= nq) {
    for (int c = lane; c < d_pad; c += 64) dst[c] = (_Float16)0.f;
    if (lane == 0) qstat[r * 4 + 0] = qstat[r * 4 + 1] = qstat[r * 4 + 2] = qstat[r * 4 + 3] = 0.f;
    return;
  }
  const float* src = q + r * dim;
  float s2 = 0.f, e2 = 0.f;
  for (int c = lane; c < d_pad; c += 64) {
    const float x = c < dim ? src[c] : 0.f;
    const _Float16 h = (_Float16)x;
    dst[c] = h;
    s2 = fmaf(x, x, s2);
    const float e = x - (float)h;
    e2 = fmaf(e, e, e2);
  }
  for (int o = 32; o > 0; o >>= 1) {
    s2 += __shfl_xor(s2, o);
    e2 += __shfl_xor(e2, o);
  }
  if (lane == 0) {
    qstat[r * 4 + 0] = s2;
    qstat[r * 4 + 1] = sqrtf(s2) * 1.000001f;
    qstat[r * 4 + 2] = sqrtf(e2) * 1.000001f;
    qstat[r * 4 + 3] = 0.f;
  }
}

// T[j]: pass threshold in approx-key space for query j, from the current k-th best exact distance.
__global__ void threshold_kernel(const u64* run_keys, int k, int64_t nq, int64_t b_pad, const float* qstat,
                                 const float* scal, int metric, float* T) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b_pad) return;
  if (j >= nq) {
    T[j] = -__builtin_inff();
    return;
  }
  const u64 kth = run_keys[j * k + (k - 1)];
  const float FMAX = 3.0e38f;
  if (kth == KEY_EMPTY) {
    T[j] = FMAX;  // fewer than k visible rows so far: everything passes (bounded by the candidate cap)
    return;
  }
  const float thr = key_dist(kth);
  const float qn2 = qstat[j * 4 + 0], nq_ = qstat[j * 4 + 1], eq = qstat[j * 4 + 2];
  const float e1max = scal[0], nxhmax = scal[1], xnmax = scal[2];
  const float s = metric == 0 ? 2.f : 1.f;
  const float c = metric == 0 ? qn2 : (metric == 1 ? 1.f : 0.f);
  const float margin = s * (nq_ * e1max + eq * nxhmax);
  const float scale = metric == 0 ? (fabsf(thr) + qn2 + xnmax) : (fabsf(thr) + 1.f + nq_ * nxhmax);
  float t = (thr - c) + margin + 8e-6f * scale;
  T[j] = fminf(t, FMAX);
}

// ------------------------------------------------------------------------------------------------ filter kernel
struct FilterArgs {
  const _Float16* xh;   // [n_pad][d_pad]
  const _Float16* qh;   // [b_pad][d_pad]
  const float* base;    // [n_pad]
  const float* T;       // [b_pad]
  int d_pad;
  int tiles_q;          // b_pad / BN
  int64_t tile0;        // first row tile of this stage
  int64_t ntiles;       // row tiles in this stage
  int64_t row_hi;       // rows >= row_hi are not reported
  int64_t nq;
  float s;              // -2 (L2) or -1
  u32* cand;
  u64* cand_keys;       // approx mode: (approx dist, row) keys instead of row ids
  const float* qstat;   // [b_pad][4] (approx mode: |q|^2 to turn keys into distances)
  int metric;
  u32* cnt;
  int cap;
  int ablate;           // profiling only (EPS_MFMA_ABLATE): v1: bit0 skip staging loads, bit1 skip MFMAs, bit2 skip LDS
                        // fragment reads; v3: bit3 skip the query-operand DMA, bit4 skip the row-operand DMA
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 3) + (chunk ^ ((row >> 1) & 7)); }  // 16-B granule index

__global__ __launch_bounds__(256, 2) void mfma_filter_kernel(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // layout: [2 stages][A 16 KB | B 16 KB] then base[128] floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware map: block b runs on XCD b%8; the tiles_q query tiles of a row tile are consecutive on one XCD
  const int64_t bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const int64_t local = bid >> 3;
  const int qt = (int)(local % a.tiles_q);
  const int64_t rt = (local / a.tiles_q) * 8 + xcd;
  if (rt >= a.ntiles) return;
  const int64_t row0 = (a.tile0 + rt) * BM;
  const int64_t q0 = (int64_t)qt * BN;
  const int ldk = a.d_pad;
  const int KT = ldk / BK;

  float* base_lds = reinterpret_cast<float*>(lds + 2 * 32768);
  if (tid < BM) base_lds[tid] = a.base[row0 + tid];

  const _Float16* gA = a.xh + row0 * ldk;
  const _Float16* gB = a.qh + q0 * ldk;

  // per-thread staging coordinates: 4 granules of A and 4 of B per K-tile
  int g_row[4], g_chunk[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 256 + tid;
    g_row[it] = s >> 3;
    g_chunk[it] = (s & 7) ^ ((g_row[it] >> 1) & 7);
  }
  auto stage = [&](int kt, int buf) {
    unsigned char* dA = lds + buf * 32768;
    unsigned char* dB = dA + 16384;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const _Float16* sa = gA + (int64_t)g_row[it] * ldk + kt * BK + g_chunk[it] * 8;
      const _Float16* sb = gB + (int64_t)g_row[it] * ldk + kt * BK + g_chunk[it] * 8;
      const int wbase = (it * 256 + wave * 64) * 16;  // wave-uniform LDS base; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)(dA + wbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(dB + wbase), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow0 = wm * 64 + (lane & 31);
  const int brow0 = wn * 64 + (lane & 31);
  const int khalf = lane >> 5;

  const bool ab_noload = a.ablate & 1, ab_nomfma = a.ablate & 2, ab_nolds = a.ablate & 4;
  if (!ab_noload) stage(0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT && !ab_noload) stage(kt + 1, (kt + 1) & 1);
    const unsigned char* sA = lds + (kt & 1) * 32768;
    const unsigned char* sB = sA + 16384;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int chunk = kk * 2 + khalf;
      half8 fa[2], fb[2];
      if (!ab_nolds) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          fa[f] = *reinterpret_cast<const half8*>(sA + swz(arow0 + f * 32, chunk) * 16);
          fb[f] = *reinterpret_cast<const half8*>(sB + swz(brow0 + f * 32, chunk) * 16);
        }
      } else {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            fa[f][e] = (_Float16)(float)(kk + e);
            fb[f][e] = (_Float16)(float)(lane + e);
          }
      }
      if (!ab_nomfma) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          asm volatile("" ::"v"(fa[f]));
          asm volatile("" ::"v"(fb[f]));
        }
      }
    }
  }

  // epilogue: approx lower-bound key vs per-query threshold; survivors are appended to the candidate lists
  float Tj[2], cj[2];
  int64_t qj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    qj[j] = q0 + wn * 64 + j * 32 + (lane & 31);
    Tj[j] = a.T[qj[j]];
    cj[j] = a.cand_keys ? (a.metric == 0 ? a.qstat[qj[j] * 4] : (a.metric == 1 ? 1.f : 0.f)) : 0.f;
  }
  __syncthreads();  // base_lds visible (first barrier of the K loop already ordered it; kept for KT == 0 safety)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = wm * 64 + i * 32 + 4 * khalf;
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(&base_lds[rbase + 8 * g]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bool any = false;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = (r & 3) == 0 ? bv[r >> 2].x : (r & 3) == 1 ? bv[r >> 2].y : (r & 3) == 2 ? bv[r >> 2].z : bv[r >> 2].w;
        v[r] = fmaf(acc[i][j][r], a.s, b);
        any |= (v[r] <= Tj[j]);
      }
      if (__any(any)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (v[r] <= Tj[j]) {
            const int64_t row = row0 + rbase + (r & 3) + 8 * (r >> 2);
            if (row < a.row_hi && qj[j] < a.nq) {
              const u32 slot = atomicAdd(&a.cnt[qj[j]], 1u);
              if (slot < (u32)a.cap) {
                if (a.cand_keys) {
                  float dapx = v[r] + cj[j];
                  if (a.metric == 0) dapx = fmaxf(dapx, 0.f);
                  a.cand_keys[qj[j] * (int64_t)a.cap + slot] = make_key(dapx, (u32)row);
                } else {
                  a.cand[qj[j] * (int64_t)a.cap + slot] = (u32)row;
                }
              }
            }
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ v2 kernel
// 256 rows x 128 queries per workgroup, 8 wavefronts (4 x 2) x 64x64 outputs, K-step 64, THREE LDS slots of 48 KB:
// two K-tiles are always in flight (counted s_waitcnt vmcnt(6), raw s_barrier — a __syncthreads() would drain the
// LDS-DMA queue to zero), so twice the bytes are outstanding per CU compared with v1 while the row tile is twice as
// tall (170 flop per L2 byte instead of 128).
__global__ __launch_bounds__(512, 2) void mfma_filter_kernel_v2(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int SLOT = 49152;  // A 256x128 B + B 128x128 B
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int64_t bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const int64_t local = bid >> 3;
  const int qt = (int)(local % a.tiles_q);
  const int64_t rt = (local / a.tiles_q) * 8 + xcd;
  if (rt >= a.ntiles) return;
  const int64_t row0 = (a.tile0 + rt) * BM2;
  const int64_t q0 = (int64_t)qt * BN;
  const int ldk = a.d_pad;
  const int KT = ldk / BK;

  float* base_lds = reinterpret_cast<float*>(lds + 3 * SLOT);
  if (tid < BM2) base_lds[tid] = a.base[row0 + tid];

  const _Float16* gA = a.xh + row0 * ldk;
  const _Float16* gB = a.qh + q0 * ldk;
  // staging: A = 2048 granules (4 per thread), B = 1024 granules (2 per thread)
  int a_row[4], a_chunk[4], b_row[2], b_chunk[2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 512 + tid;
    a_row[it] = s >> 3;
    a_chunk[it] = (s & 7) ^ ((a_row[it] >> 1) & 7);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int s = it * 512 + tid;
    b_row[it] = s >> 3;
    b_chunk[it] = (s & 7) ^ ((b_row[it] >> 1) & 7);
  }
  auto stage = [&](int kt, int slot) {
    unsigned char* dA = lds + slot * SLOT;
    unsigned char* dB = dA + 32768;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const _Float16* sa = gA + (int64_t)a_row[it] * ldk + kt * BK + a_chunk[it] * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)(dA + (it * 512 + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const _Float16* sb = gB + (int64_t)b_row[it] * ldk + kt * BK + b_chunk[it] * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(dB + (it * 512 + wave * 64) * 16), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow0 = wm * 64 + (lane & 31);
  const int brow0 = wn * 64 + (lane & 31);
  const int khalf = lane >> 5;

  stage(0, 0);
  if (KT > 1) stage(1, 1);
  int slot = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // tile kt has landed once at most the 6 loads of tile kt+1 are still outstanding
    if (kt + 1 < KT)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < KT) {
      int ns = slot + 2;
      if (ns >= 3) ns -= 3;
      stage(kt + 2, ns);  // slot (kt+2)%3 == (kt-1)%3: every wave finished reading it before this barrier
    }
    const unsigned char* sA = lds + slot * SLOT;
    const unsigned char* sB = sA + 32768;
    // software pipeline over the four K=16 sub-steps: the fragments of sub-step kk+1 are read from LDS while the
    // MFMAs of sub-step kk run (one wave per SIMD per block: nothing else hides the ds_read latency)
    half8 fa[2][2], fb[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      fa[0][f] = *reinterpret_cast<const half8*>(sA + swz(arow0 + f * 32, khalf) * 16);
      fb[0][f] = *reinterpret_cast<const half8*>(sB + swz(brow0 + f * 32, khalf) * 16);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk < 3) {
        const int chunk = (kk + 1) * 2 + khalf;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          fa[nxt][f] = *reinterpret_cast<const half8*>(sA + swz(arow0 + f * 32, chunk) * 16);
          fb[nxt][f] = *reinterpret_cast<const half8*>(sB + swz(brow0 + f * 32, chunk) * 16);
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    slot = slot + 1 == 3 ? 0 : slot + 1;
  }

  float Tj[2], cj[2];
  int64_t qj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    qj[j] = q0 + wn * 64 + j * 32 + (lane & 31);
    Tj[j] = a.T[qj[j]];
    cj[j] = a.cand_keys ? (a.metric == 0 ? a.qstat[qj[j] * 4] : (a.metric == 1 ? 1.f : 0.f)) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = wm * 64 + i * 32 + 4 * khalf;
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(&base_lds[rbase + 8 * g]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bool any = false;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = (r & 3) == 0 ? bv[r >> 2].x : (r & 3) == 1 ? bv[r >> 2].y : (r & 3) == 2 ? bv[r >> 2].z : bv[r >> 2].w;
        v[r] = fmaf(acc[i][j][r], a.s, b);
        any |= (v[r] <= Tj[j]);
      }
      if (__any(any)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (v[r] <= Tj[j]) {
            const int64_t row = row0 + rbase + (r & 3) + 8 * (r >> 2);
            if (row < a.row_hi && qj[j] < a.nq) {
              const u32 slot_c = atomicAdd(&a.cnt[qj[j]], 1u);
              if (slot_c < (u32)a.cap) {
                if (a.cand_keys) {
                  float dapx = v[r] + cj[j];
                  if (a.metric == 0) dapx = fmaxf(dapx, 0.f);
                  a.cand_keys[qj[j] * (int64_t)a.cap + slot_c] = make_key(dapx, (u32)row);
                } else {
                  a.cand[qj[j] * (int64_t)a.cap + slot_c] = (u32)row;
                }
              }
            }
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ v3 kernel
// Persistent form.  Ablation of v1 (profiles/r1_mfma_ablation.txt) showed the filter was bound by per-workgroup
// latency, not by the matrix cores: a workgroup that lives for one 128x128 tile pays its launch + first-load latency
// (~6 us) for 12 K-steps of work, and each K-step exposes one L2->LDS round trip.  v3 launches ONE workgroup per CU
// (8 wavefronts, 2 x 4, each 128 rows x 64 queries = 4 x 2 tiles of v_mfma_f32_32x32x16_f16) that walks a list of
// 256 x 256 tiles; the (tile, K-step) sequence is one software pipeline — the loads of step s+1 (possibly the next
// tile's first K-step, plus its |x|^2 column) are issued right after the barrier of step s and land under the 32
// MFMAs per wavefront of step s; the epilogue of a tile runs under the first loads of the next.  256 flop per L2
// byte (v1: 128).  Tile order keeps the query tiles of one row tile on one XCD at the same time.
constexpr int BM3 = 256, BN3 = 256;
template <bool ABL>  // ABL: profiling build with the EPS_MFMA_ABLATE switches compiled in
__global__ __launch_bounds__(512, 2) void mfma_filter_kernel_v3(FilterArgs a) {
  const int ablate = ABL ? a.ablate : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int SLOT = 65536;  // A 256 x 128 B | B 256 x 128 B
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int khalf = lane >> 5;
  float* base_lds = reinterpret_cast<float*>(lds + 2 * SLOT);  // [2][256]

  // work list of this workgroup
  const int xcd = blockIdx.x & 7;
  const int local = blockIdx.x >> 3;              // 0 .. gridDim/8-1 workgroups on this XCD
  const int per_xcd = gridDim.x >> 3;
  const int QTB = a.tiles_q < per_xcd ? a.tiles_q : per_xcd;
  const int G = per_xcd / QTB;                    // row tiles in flight per XCD
  const int qslot = local % QTB;
  const int rg = local / QTB;
  if (rg >= G) return;
  // row tiles of this XCD: rt = xcd + 8*j; this workgroup takes j = rg, rg+G, ...; query tiles qt = qslot, qslot+QTB, ...
  const int64_t nj = (a.ntiles - xcd + 7) / 8;    // row tiles on this XCD (may be <= 0)
  const int nqt = (a.tiles_q - qslot + QTB - 1) / QTB;
  const int64_t my_rows = nj > rg ? (nj - rg + G - 1) / G : 0;
  const int64_t ntile = my_rows * nqt;
  if (ntile <= 0) return;
  const int ldk = a.d_pad;
  const int KT = ldk / BK;

  int g_off[4];  // element offset of this thread's granule `it` inside a K-step of a 256-row operand tile
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 512 + tid;
    const int row = s >> 3;
    g_off[it] = row * ldk + ((s & 7) ^ ((row >> 1) & 7)) * 8;
  }
  auto tile_rt = [&](int64_t t) { return (int64_t)xcd + 8 * (rg + (t / nqt) * G); };
  auto tile_qt = [&](int64_t t) { return qslot + (int)(t % nqt) * QTB; };
  // operand bases of the tile being computed and of the tile whose first K-step is prefetched (one division per tile)
  const _Float16 *gA_cur, *gB_cur, *gA_nx, *gB_nx;
  const float* gbase_nx;
  auto set_next = [&](int64_t t) {
    const int64_t rt = tile_rt(t);
    gA_nx = a.xh + (a.tile0 + rt) * BM3 * (int64_t)ldk;
    gB_nx = a.qh + (int64_t)tile_qt(t) * BN3 * ldk;
    gbase_nx = a.base + (a.tile0 + rt) * BM3;
  };
  // one quarter of a K-step's staging: piece `it` of A and of B
  auto stage_piece = [&](const _Float16* gA, const _Float16* gB, int kt, int slot, int it) {
    unsigned char* dA = lds + slot * SLOT;
    unsigned char* dB = dA + 32768;
    const int off = g_off[it] + kt * BK;
    const int wbase = (it * 512 + wave * 64) * 16;
    if (!(ablate & 16))
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + off),
                                       (__attribute__((address_space(3))) void*)(dA + wbase), 16, 0, 0);
    if (!(ablate & 8))
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + off),
                                       (__attribute__((address_space(3))) void*)(dB + wbase), 16, 0, 0);
  };
  auto stage_base = [&](const float* gb, int64_t t) {  // |x|^2 (or 0) column of the tile's 256 rows, wavefronts 0-3
    if (wave < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + wave * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(base_lds + (t & 1) * 256 + wave * 64), 4, 0, 0);
  };

  f32x16 acc[4][2];
  const int arow0 = wm * 128 + (lane & 31);
  const int brow0 = wn * 64 + (lane & 31);

  // thresholds of this workgroup's query tile, loaded before any LDS-DMA is in flight (ordinary loads make the
  // compiler wait vmcnt(0), which would drain the pipeline if done per tile)
  const float inv_s = 1.0f / a.s;  // s = -2 (L2) or -1: exact
  float Tq[2], cj[2];   // Tq = T/s: threshold in accumulator space (a row passes iff acc >= Tq)
  int64_t qj[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    qj[j] = (int64_t)qslot * BN3 + wn * 64 + j * 32 + (lane & 31);
    Tq[j] = a.T[qj[j]] * inv_s;
    cj[j] = a.cand_keys ? (a.metric == 0 ? a.qstat[qj[j] * 4] : (a.metric == 1 ? 1.f : 0.f)) : 0.f;
  }
  set_next(0);
#pragma unroll
  for (int it = 0; it < 4; ++it) stage_piece(gA_nx, gB_nx, 0, 0, it);
  stage_base(gbase_nx, 0);
  int slot = 0;
  for (int64_t t = 0; t < ntile; ++t) {
    gA_cur = gA_nx;
    gB_cur = gB_nx;
    const int64_t row0 = (a.tile0 + tile_rt(t)) * BM3;
    const int64_t q0 = (int64_t)tile_qt(t) * BN3;
    if (t + 1 < ntile) set_next(t + 1);
    // accumulators start at base/s (= -|x|^2/2 for L2, 0 otherwise; -inf on padding rows), so that the finished
    // accumulator is (approx key)/s and the epilogue is one max + one compare per 16 outputs.  The |x|^2 column of
    // this tile was staged with its first K-step; that step has not been waited for yet when t == 0 / a tile starts,
    // so the init happens after the first barrier of the tile (kt == 0 below).
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt == 0) {
        const float* bl0 = base_lds + (t & 1) * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rbase = wm * 128 + i * 32 + 4 * khalf;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(&bl0[rbase + 8 * g]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              acc[i][j][4 * g + 0] = bv.x * inv_s;
              acc[i][j][4 * g + 1] = bv.y * inv_s;
              acc[i][j][4 * g + 2] = bv.z * inv_s;
              acc[i][j][4 * g + 3] = bv.w * inv_s;
            }
          }
        }
      }
      // next step of the (tile, K-step) stream; its staging is spread over the four K=16 sub-steps below so that the
      // DMA issue cost of one wavefront overlaps the MFMAs of the wavefront sharing its SIMD
      const bool same = kt + 1 < KT;
      const bool more = same || (t + 1 < ntile);
      const _Float16* pA = same ? gA_cur : gA_nx;
      const _Float16* pB = same ? gB_cur : gB_nx;
      const int nk_ = same ? kt + 1 : 0;
      const unsigned char* sA = lds + slot * SLOT;
      const unsigned char* sB = sA + 32768;
      // fragments of sub-step kk+1 are read from LDS while the MFMAs of sub-step kk issue (register double buffer)
      half8 fa[2][4], fb[2][2];
#pragma unroll
      for (int f = 0; f < 4; ++f) fa[0][f] = *reinterpret_cast<const half8*>(sA + swz(arow0 + f * 32, khalf) * 16);
#pragma unroll
      for (int f = 0; f < 2; ++f) fb[0][f] = *reinterpret_cast<const half8*>(sB + swz(brow0 + f * 32, khalf) * 16);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk < 3 && !(ablate & 4)) {
          const int chunk = (kk + 1) * 2 + khalf;
#pragma unroll
          for (int f = 0; f < 4; ++f) fa[nxt][f] = *reinterpret_cast<const half8*>(sA + swz(arow0 + f * 32, chunk) * 16);
#pragma unroll
          for (int f = 0; f < 2; ++f) fb[nxt][f] = *reinterpret_cast<const half8*>(sB + swz(brow0 + f * 32, chunk) * 16);
        }
        if (more) {
          stage_piece(pA, pB, nk_, slot ^ 1, kk);
          if (kk == 0 && !same) stage_base(gbase_nx, t + 1);
        }
        if (!(ablate & 2)) {
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
          for (int f = 0; f < 4; ++f) asm volatile("" ::"v"(fa[cur][f]));
#pragma unroll
          for (int f = 0; f < 2; ++f) asm volatile("" ::"v"(fb[cur][f]));
        }
      }
      slot ^= 1;
    }
    // ---- epilogue of tile t (the first K-step of tile t+1 is already in flight)
    if (nqt > 1) {  // the query tile changes between tiles: reload its thresholds (ordinary loads: drains the DMA queue once)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        qj[j] = q0 + wn * 64 + j * 32 + (lane & 31);
        Tq[j] = a.T[qj[j]] * inv_s;
        cj[j] = a.cand_keys ? (a.metric == 0 ? a.qstat[qj[j] * 4] : (a.metric == 1 ? 1.f : 0.f)) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rbase = wm * 128 + i * 32 + 4 * khalf;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // pass  <=>  s*acc <= T  <=>  acc >= T/s  (s < 0): one running max over the 16 outputs of this lane
        float mx = acc[i][j][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
        if (__any(mx >= Tq[j])) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (acc[i][j][r] >= Tq[j]) {
              const int64_t row = row0 + rbase + (r & 3) + 8 * (r >> 2);
              if (row < a.row_hi && qj[j] < a.nq && !ablate) {
                const u32 slot_c = atomicAdd(&a.cnt[qj[j]], 1u);
                if (slot_c < (u32)a.cap) {
                  if (a.cand_keys) {
                    float dapx = acc[i][j][r] * a.s + cj[j];
                    if (a.metric == 0) dapx = fmaxf(dapx, 0.f);
                    a.cand_keys[qj[j] * (int64_t)a.cap + slot_c] = make_key(dapx, (u32)row);
                  } else {
                    a.cand[qj[j] * (int64_t)a.cap + slot_c] = (u32)row;
                  }
                }
              }
            }
          }
        }
      }
    }
  }
}

__global__ void count_overflow_kernel(const u32* cnt, int64_t nq, int cap, u32* overflow) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nq && cnt[j] > (u32)cap) atomicAdd(overflow, 1u);
}
__global__ void sum_counts_kernel(const u32* cnt, int64_t nq, int cap, unsigned long long* total) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nq) atomicAdd(total, (unsigned long long)(cnt[j] < (u32)cap ? cnt[j] : (u32)cap));
}

// ------------------------------------------------------------------------------------------------ host
static int32_t ensure_mirror(Index& ix) {
  if (!ix.mirror_) ix.mirror_ = new HalfMirror();
  HalfMirror& m = *ix.mirror_;
  if (m.version == ix.rows_version_) return EPS_OK;
  const int64_t n = ix.n_rows_;
  const int64_t n_pad = (n + ROWPAD - 1) / ROWPAD * ROWPAD;
  const int d_pad = (int)((ix.dim_ + BK - 1) / BK * BK);
  if (!m.xh.reserve((size_t)n_pad * d_pad * 2) || !m.xn.reserve((size_t)n_pad * 4) || !m.zeros.reserve((size_t)n_pad * 4) ||
      !m.scal.reserve(64))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the fp16 mirror");
  hipStream_t s = ix.stream_;
  hipError_t er = hipMemsetAsync(m.scal.p, 0, 64, s);
  if (er != hipSuccess) return ix.hip_fail(er, "memset");
  // fp32 accumulation slack of the MFMA dot product: <= 4 * d * 2^-24 * |qh||xh| (generous: covers any
  // internal summation order / truncating adder)
  const float gamma = 4.0f * (float)d_pad * 5.9604645e-8f;
  hipLaunchKernelGGL(half_mirror_kernel, dim3((unsigned)std::min<int64_t>((n_pad + 3) / 4, 8192)), dim3(256), 0, s, ix.d_rows_, n, n_pad,
                     (int)ix.dim_, d_pad, m.xh.as<_Float16>(), m.xn.as<float>(), m.zeros.as<float>(), m.scal.as<float>(), gamma);
  er = hipMemcpyAsync(m.h_scal, m.scal.p, 16, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return ix.hip_fail(er, "fp16 mirror build");
  m.fp16_range_ok = (m.h_scal[3] == 0.f);
  m.n = n;
  m.n_pad = n_pad;
  m.d_pad = d_pad;
  m.version = ix.rows_version_;
  return EPS_OK;
}

bool flat_mfma_profitable(const Index& ix, int64_t nq, int k) {
  // worth it only when the GEMM is big enough to beat the HBM-bound stream scan
  if (nq < 32 || ix.n_rows_ < 65536 || k > 128) return false;
  if (ix.mirror_ && ix.mirror_->version == ix.rows_version_ && !ix.mirror_->fp16_range_ok) return false;
  return true;
}

int32_t flat_mfma_search(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx) {
  int32_t rc = ensure_mirror(ix);
  if (rc != EPS_OK) return rc;
  HalfMirror& m = *ix.mirror_;
  const int64_t n = ix.scan_limit_ >= 0 ? std::min(ix.scan_limit_, ix.n_rows_) : ix.n_rows_;
  if (!m.fp16_range_ok) {
    // values beyond the fp16 range: the filter bound would be vacuous; the exact stream engine takes over
    return ix.flat_stream(dq, nq, k, 0, n, run_keys, false, -1, !approx);
  }
  hipStream_t s = ix.stream_;
  const int64_t b_pad = (nq + BN3 - 1) / BN3 * BN3;
  const int cap = std::max(4096, 64 * k);
  if (!m.qh.reserve((size_t)b_pad * m.d_pad * 2) || !m.qstat.reserve((size_t)b_pad * 16) || !m.T.reserve((size_t)b_pad * 4) ||
      !m.cand.reserve((size_t)nq * cap * (approx ? 8 : 4)) || !m.cnt.reserve((size_t)(nq + 4) * 4 + 16))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)((b_pad + 3) / 4)), dim3(256), 0, s, dq, nq, b_pad, (int)ix.dim_,
                     m.d_pad, m.qh.as<_Float16>(), m.qstat.as<float>());

  // stage boundaries (multiples of BM): S0, 32*S0, 256*S0, n
  int64_t S0 = std::max<int64_t>(4096, (int64_t)(32 * k + ROWPAD - 1) / ROWPAD * ROWPAD);
  std::vector<int64_t> bounds;
  bounds.push_back(std::min(S0, n));
  for (int64_t bnd : {S0 * 32, S0 * 256}) {
    if (bnd < n && bnd > bounds.back()) bounds.push_back(bnd);
  }
  if (bounds.back() < n) bounds.push_back(n);

  // stage 0: exact scan of the head
  rc = ix.flat_stream(dq, nq, k, 0, bounds[0], run_keys, false, -1, !approx);
  if (rc != EPS_OK) return rc;
  ix.stats_.main_kernel_launches = 0;

  u32* cnt = m.cnt.as<u32>();
  u32* overflow = cnt + nq;                                                    // [1]
  unsigned long long* total = reinterpret_cast<unsigned long long*>(cnt + nq + 2);  // 8-byte aligned? ensured below
  if ((reinterpret_cast<uintptr_t>(total) & 7) != 0) total = reinterpret_cast<unsigned long long*>(cnt + nq + 3);
  hipError_t er = hipMemsetAsync(cnt + nq, 0, 32, s);
  if (er != hipSuccess) return ix.hip_fail(er, "memset");

  FilterArgs fa;
  fa.xh = m.xh.as<_Float16>();
  fa.qh = m.qh.as<_Float16>();
  fa.base = ix.metric_ == 0 ? m.xn.as<float>() : m.zeros.as<float>();
  fa.T = m.T.as<float>();
  fa.d_pad = m.d_pad;
  fa.tiles_q = (int)(b_pad / BN);
  fa.nq = nq;
  fa.s = ix.metric_ == 0 ? -2.f : -1.f;
  fa.cand = m.cand.as<u32>();
  fa.cand_keys = approx ? m.cand.as<u64>() : nullptr;
  fa.qstat = m.qstat.as<float>();
  fa.metric = ix.metric_;
  fa.cnt = cnt;
  fa.cap = cap;
  fa.ablate = getenv("EPS_MFMA_ABLATE") ? atoi(getenv("EPS_MFMA_ABLATE")) : 0;

  RerankArgs ra;
  ra.rows = ix.d_rows_;
  ra.dim = (int)ix.dim_;
  ra.metric = ix.metric_;
  ra.queries = dq;
  ra.nq = nq;
  ra.k = k;
  ra.f = ix.filter_spec();
  ra.cand = fa.cand;
  ra.cand_count = cnt;
  ra.cap = cap;
  ra.run_keys = run_keys;

  static const int version = getenv("EPS_MFMA_KERNEL") ? atoi(getenv("EPS_MFMA_KERNEL")) : 3;
  const int bm = version == 1 ? BM : BM2;  // v2 and v3 use 256-row tiles
  const size_t shm = version == 1 ? 2 * 32768 + BM * sizeof(float) : version == 2 ? 3 * 49152 + BM2 * sizeof(float) : 2 * 65536 + 2 * 256 * sizeof(float);
  static int num_cus = 0;
  if (!num_cus) {
    hipDeviceProp_t prop;
    num_cus = hipGetDeviceProperties(&prop, ix.device_) == hipSuccess ? prop.multiProcessorCount : 256;
    num_cus = num_cus / 8 * 8;
    if (num_cus < 8) num_cus = 8;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 32768 + BM * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_filter_kernel_v2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * 49152 + BM2 * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_filter_kernel_v3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 65536 + 2 * 256 * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_filter_kernel_v3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 65536 + 2 * 256 * sizeof(float)));
    attr_set = true;
  }
  bool first = true;
  for (size_t st = 0; st + 1 < bounds.size(); ++st) {
    const int64_t lo = bounds[st], hi = bounds[st + 1];
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, s, run_keys, k, nq, b_pad,
                       m.qstat.as<float>(), m.scal.as<float>(), ix.metric_, m.T.as<float>());
    er = hipMemsetAsync(cnt, 0, (size_t)nq * 4, s);
    if (er != hipSuccess) return ix.hip_fail(er, "memset");
    fa.tile0 = lo / bm;
    fa.ntiles = (hi + bm - 1) / bm - fa.tile0;
    fa.row_hi = hi;
    const int64_t blocks = (fa.ntiles + 7) / 8 * 8 * fa.tiles_q;
    const bool biggest = (st + 2 == bounds.size());
    if (biggest) (void)hipEventRecord(ix.evk0_, s);
    if (biggest) ix.stats_.main_kernel_rows = hi - lo;
    if (version == 1) {
      hipLaunchKernelGGL(mfma_filter_kernel, dim3((unsigned)blocks), dim3(256), shm, s, fa);
    } else if (version == 2) {
      hipLaunchKernelGGL(mfma_filter_kernel_v2, dim3((unsigned)blocks), dim3(512), shm, s, fa);
    } else {
      FilterArgs f3 = fa;
      f3.tiles_q = (int)(b_pad / BN3);
      if (f3.ablate)
        hipLaunchKernelGGL(mfma_filter_kernel_v3<true>, dim3((unsigned)num_cus), dim3(512), shm, s, f3);
      else
        hipLaunchKernelGGL(mfma_filter_kernel_v3<false>, dim3((unsigned)num_cus), dim3(512), shm, s, f3);
    }
    if (biggest) (void)hipEventRecord(ix.evk1_, s);
    hipLaunchKernelGGL(count_overflow_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, cnt, nq, cap, overflow);
    hipLaunchKernelGGL(sum_counts_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, cnt, nq, cap, total);
    if (getenv("EPS_DEBUG")) {
      std::vector<u32> hc((size_t)nq);
      std::vector<float> hT((size_t)nq);
      std::vector<u64> hk((size_t)nq * k);
      (void)hipMemcpyAsync(hc.data(), cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, s);
      (void)hipMemcpyAsync(hT.data(), m.T.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s);
      (void)hipMemcpyAsync(hk.data(), run_keys, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s);
      (void)hipStreamSynchronize(s);
      u32 mn = ~0u, mx = 0; double sum = 0; float tmin = 3e38f, tmax = -3e38f; int64_t empt = 0, big = 0;
      for (int64_t j = 0; j < nq; ++j) { mn = std::min(mn, hc[j]); mx = std::max(mx, hc[j]); sum += hc[j]; tmin = std::min(tmin, hT[j]); tmax = std::max(tmax, hT[j]); empt += hk[j * k + k - 1] == KEY_EMPTY; big += hc[j] > (u32)cap; }
      fprintf(stderr, "[eps] stage %zu rows [%lld,%lld) tiles %lld blocks %lld: cnt min %u mean %.1f max %u (>cap: %lld), T min %g max %g, empty kth %lld, scal %g %g %g\n", st, (long long)lo, (long long)hi, (long long)fa.ntiles, (long long)blocks, mn, sum / nq, mx, (long long)big, tmin, tmax, (long long)empt, m.h_scal[0], m.h_scal[1], m.h_scal[2]);
    }
    if (approx)
      launch_merge_lists(fa.cand_keys, cap, k, nq, run_keys, true, s, cnt);  // select on the fp16 keys
    else
      launch_rerank(ra, s);
    first = false;
  }
  (void)first;
  er = hipGetLastError();
  if (er != hipSuccess) return ix.hip_fail(er, "MFMA filter launch");
  struct {
    u32 overflow, pad;
    unsigned long long total;
  } h = {0, 0, 0};
  er = hipMemcpyAsync(&h.overflow, overflow, 4, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipMemcpyAsync(&h.total, total, 8, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return ix.hip_fail(er, "MFMA filter");
  ix.stats_.rerank_rows += (int64_t)h.total;
  ix.stats_.dist_evals += nq * (n - bounds[0]);
  ix.stats_.main_kernel_launches = 1;
  if (h.overflow) {
    ix.stats_.overflow_queries += h.overflow;
    if (!approx && !fa.ablate) return ix.flat_stream(dq, nq, k, 0, n, run_keys, false);  // exact fallback for the (rare) overflow case
  }
  return EPS_OK;
}

}  // namespace eps
