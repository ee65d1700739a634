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
//     stage 0: the first S0 rows: their k best approximate keys (same kernel, every row a candidate), re-ranked
//              exactly, give the first T (with a deleted bitset / attribute filter: exact fp32 stream scan instead)
//     stage i: MFMA filter over a geometrically larger chunk          -> candidates -> exact re-rank -> top-k
// Expected candidates per query ~ k * sum_i (chunk_i / rows_before_i): a few hundred at N = 10M, k = 10.
// If a query's candidate buffer overflows (adversarial order/duplicates) the batch falls back to the fp32 scan.
// The bound is a worst-case one (no distributional assumption) for the fp16 rounding and the accumulation order of the
// GEMM; the fp32 rounding of the re-ranked keys it is compared with is covered by a slack that grows with d.
//
// Kernels: mfma_kernels.hpp - v7 (default: persistent, 4 wavefronts x 256 rows x 64 queries per 256 x 256 tile, query
// fragments straight to VGPRs, 4-slot LDS-DMA ring for the row operand) and v3, the fallback for d_pad % 128 != 0 or
// d_pad < 256 (tests/test_gpu_parity.py::test_mfma_engine_is_exact covers d = 33 and d = 100).  Staging, seeds, re-rank and the overflow fallback are in flat_mfma_search_slice below.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "index.hpp"
#include "mfma_kernels.hpp"

namespace eps {

constexpr int EPS_MFMA_MANTISSA_DEFAULT = 10;

struct HalfMirror {
  DevBuf xh;       // _Float16 [n_pad][d_pad]
  DevBuf xn;       // float [n_pad]  |x|^2 (+inf on padding rows)
  DevBuf zeros;    // float [n_pad]  base for IP / COSINE (+inf on padding rows)
  DevBuf xn_s;     // float [n_pad]  -|x|^2/2 (= xn / s for L2; -inf on padding rows): v5 accumulator init
  DevBuf zeros_s;  // float [n_pad]  0 (-inf on padding rows)
  DevBuf qf;       // _Float16 fragment-major copy of qh (v5)
  DevBuf gsync;    // u32 [64]: v7 group arrival counters
  DevBuf sxh, sbase, sbase_u;   // seed sample: S0 rows spread evenly over [0, n) (fp16 rows, their base / s, their base)
  int64_t sample_version = -1, sample_n = 0, sample_rows = 0;
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
  int num_cus = 0;             // CUs of this index's device (persistent grid size)
  int drop = -1;               // low mantissa bits the mirror and the query operand leave at zero (to_half_drop)
  int64_t extended_rows = 0;   // rows converted by incremental extensions (test hook)
  float h_scal[4] = {0, 0, 0, 0};
};

void half_mirror_free(HalfMirror* m) { delete m; }

// ------------------------------------------------------------------------------------------------ mirror build
// fp32 -> fp16 keeping only the top `10 - drop` mantissa bits (round to nearest even at that width).  The MI355X clocks to its
// power budget and the matrix pipe draws less on operands that toggle fewer bits (scripts/lab/mfma_power.hip: the same MFMA
// stream runs 7.6 % / 12 % faster at 7 / 5 mantissa bits); the filter only needs a lower bound, and every bound below is
// computed from the residual |x - xh| of the value actually stored, so the result stays exact for any `drop` - coarser
// operands just let a few more candidates through to the fp32 re-rank.
__device__ __forceinline__ _Float16 to_half_drop(float x, int drop) {
  const _Float16 h = (_Float16)x;
  if (drop <= 0) return h;
  unsigned short u = __builtin_bit_cast(unsigned short, h);
  const unsigned short keep = (unsigned short)~((1u << drop) - 1u);
  const unsigned short r = (unsigned short)(u + ((1u << (drop - 1)) - 1u) + ((u >> drop) & 1u));   // RNE (carries into the exponent)
  u = ((r & 0x7C00u) == 0x7C00u && (u & 0x7C00u) != 0x7C00u) ? (unsigned short)(u & keep) : (unsigned short)(r & keep);   // never round up to inf
  return __builtin_bit_cast(_Float16, u);
}

__device__ __forceinline__ void atomic_max_pos(float* addr, float v) {  // v >= 0
  atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// rows [row0, n_pad) are (re)written: row0 = 0 builds the mirror, row0 = rows mirrored so far extends it after an append
// (the per-index maxima in `scal` only ever grow, so they are accumulated across calls)
__global__ __launch_bounds__(256) void half_mirror_kernel(const float* rows, int64_t row0, int64_t n, int64_t n_pad, int dim, int d_pad,
                                                          _Float16* xh, float* xn, float* zeros, float* xn_s, float* zeros_s, float* scal,
                                                          float gamma, int drop) {
  // one wavefront per row, grid-stride over rows; the four per-index maxima are reduced in registers and
  // published with ONE atomic per wavefront (an atomic per row serialises 10M rows on four addresses)
  const int lane = lane_id();
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float m_e1 = 0.f, m_nxh = 0.f, m_xn = 0.f, m_bad = 0.f;
  const bool vec = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  for (int64_t r = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_pad; r += nwaves) {
    _Float16* dst = xh + r * d_pad;
    if (r >= n) {
      for (int c = lane; c < d_pad; c += 64) dst[c] = (_Float16)0.f;
      if (lane == 0) {
        xn[r] = __builtin_inff();
        zeros[r] = __builtin_inff();
        xn_s[r] = -__builtin_inff();
        zeros_s[r] = -__builtin_inff();
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
        h[0] = to_half_drop(x.x, drop); h[1] = to_half_drop(x.y, drop); h[2] = to_half_drop(x.z, drop); h[3] = to_half_drop(x.w, drop);
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
        const _Float16 h = to_half_drop(x, drop);
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
      xn_s[r] = -0.5f * s2;
      zeros_s[r] = 0.f;
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
                                                         _Float16* qh, float* qstat, int drop) {
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
    const _Float16 h = to_half_drop(x, drop);
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
// Also resets what the stage's filter launch accumulates into (candidate counts, group arrival counters), so a stage is
// threshold -> filter -> counts -> re-rank without separate memsets.
__global__ void threshold_kernel(const u64* run_keys, int k, int64_t nq, int64_t b_pad, const float* qstat,
                                 const float* scal, int metric, float* T, u32* cnt, u32* gsync, float slack) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b_pad) return;
  if (j < nq) cnt[j] = 0;
  if (gsync && j < 256) gsync[j] = 0;
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
  float t = (thr - c) + margin + slack * scale;
  T[j] = fminf(t, FMAX);
}


// seed sample: the first S/2 rows of the table plus S/2 rows spread evenly over the rest, copied next to each other (a
// positional filter - "only the newest rows" or "only the oldest" - leaves at least half of the seeds' share visible)
__global__ __launch_bounds__(256) void seed_sample_kernel(const _Float16* xh, const float* base_s, const float* base, unsigned long long stride,
                                                          u32 head, int d_pad, _Float16* sxh, float* sbase, float* sbase_u) {
  const int64_t i = blockIdx.x;
  const int64_t r = seed_row((u32)i, head, stride);
  const half8* src = reinterpret_cast<const half8*>(xh + r * d_pad);
  half8* dst = reinterpret_cast<half8*>(sxh + i * d_pad);
  for (int c = threadIdx.x; c < d_pad / 8; c += 256) dst[c] = src[c];
  if (threadIdx.x == 0) {
    sbase[i] = base_s[r];
    sbase_u[i] = base[r];
  }
}

// seeds: the k best APPROXIMATE keys of the head rows (in run_keys) -> candidate ids for the exact re-rank; run_keys is
// reset so that the re-rank leaves exactly the seeds' exact keys in it
__global__ void seed_to_cand_kernel(u64* run_keys, int k, int64_t nq, u32* cand, int cap, u32* cnt, unsigned long long id_stride, u32 id_head) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  u32 c = 0;
  for (int e = 0; e < k; ++e) {
    const u64 key = run_keys[q * k + e];
    if (key != KEY_EMPTY) cand[q * (int64_t)cap + c++] = id_stride ? seed_row(key_id(key), id_head, id_stride) : key_id(key);
    run_keys[q * k + e] = KEY_EMPTY;
  }
  cnt[q] = c;
}

// per stage: queries whose candidate list overflowed, and the number of rows that will be re-ranked
__global__ void stage_counts_kernel(const u32* cnt, int64_t nq, int cap, u32* overflow, unsigned long long* total) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nq) return;
  const u32 c = cnt[j];
  if (c > (u32)cap) atomicAdd(overflow, 1u);
  atomicAdd(total, (unsigned long long)(c < (u32)cap ? c : (u32)cap));
}

// ------------------------------------------------------------------------------------------------ host
static bool grow_keep(DevBuf& b, size_t bytes, size_t keep, hipStream_t s) {
  if (bytes <= b.cap) return true;
  DevBuf bigger;
  if (!bigger.reserve(bytes + bytes / 4)) return false;
  if (keep && b.p && hipMemcpyAsync(bigger.p, b.p, keep, hipMemcpyDeviceToDevice, s) != hipSuccess) return false;
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  b.release();
  b.p = bigger.p;
  b.cap = bigger.cap;
  bigger.p = nullptr;
  bigger.cap = 0;
  return true;
}

static int32_t ensure_mirror(Index& ix) {
  if (!ix.mirror_) ix.mirror_ = new HalfMirror();
  HalfMirror& m = *ix.mirror_;
  const int64_t n = ix.n_rows_;
  if (m.drop < 0) {   // EPS_MFMA_MANTISSA = mantissa bits the fp16 operands keep (10 = plain fp16); fixed per mirror
    const char* e = getenv("EPS_MFMA_MANTISSA");
    const int keep = e ? std::min(10, std::max(2, atoi(e))) : EPS_MFMA_MANTISSA_DEFAULT;
    m.drop = 10 - keep;
  }
  if (m.version == ix.rows_version_ && m.n == n) return EPS_OK;
  // appended rows: only the new rows are converted; the 15 GB mirror of a 10M-row table is not rebuilt
  const bool extend = m.version == ix.rows_version_ && m.n > 0 && m.n < n;
  const int64_t n_pad = (n + ROWPAD - 1) / ROWPAD * ROWPAD;
  const int d_pad = (int)((ix.dim_ + BK - 1) / BK * BK);
  hipStream_t s = ix.stream_;
  const size_t keep_rows = extend ? (size_t)m.n : 0;
  if (!grow_keep(m.xh, (size_t)n_pad * d_pad * 2, keep_rows * d_pad * 2, s) || !grow_keep(m.xn, (size_t)n_pad * 4, keep_rows * 4, s) ||
      !grow_keep(m.zeros, (size_t)n_pad * 4, keep_rows * 4, s) || !grow_keep(m.xn_s, (size_t)n_pad * 4, keep_rows * 4, s) ||
      !grow_keep(m.zeros_s, (size_t)n_pad * 4, keep_rows * 4, s) || !m.scal.reserve(64))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the fp16 mirror");
  hipError_t er = hipSuccess;
  if (!extend) er = hipMemsetAsync(m.scal.p, 0, 64, s);
  if (er != hipSuccess) return ix.hip_fail(er, "memset");
  // fp32 accumulation slack of the MFMA dot product: <= 4 * d * 2^-24 * |qh||xh| (generous: covers any
  // internal summation order / truncating adder)
  const float gamma = 4.0f * (float)d_pad * 5.9604645e-8f;
  const int64_t row0 = extend ? m.n : 0;
  hipLaunchKernelGGL(half_mirror_kernel, dim3((unsigned)std::min<int64_t>((n_pad - row0 + 3) / 4, 8192)), dim3(256), 0, s, ix.d_rows_, row0, n, n_pad,
                     (int)ix.dim_, d_pad, m.xh.as<_Float16>(), m.xn.as<float>(), m.zeros.as<float>(), m.xn_s.as<float>(), m.zeros_s.as<float>(),
                     m.scal.as<float>(), gamma, m.drop);
  er = hipMemcpyAsync(m.h_scal, m.scal.p, 16, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return ix.hip_fail(er, "fp16 mirror build");
  m.fp16_range_ok = (m.h_scal[3] == 0.f);
  m.n = n;
  m.n_pad = n_pad;
  m.d_pad = d_pad;
  m.version = ix.rows_version_;
  m.extended_rows += extend ? n - row0 : 0;
  return EPS_OK;
}

bool flat_mfma_profitable(const Index& ix, int64_t nq, int k) {
  // FLAT_AUTO: both engines return the same bits, so this is purely a cost decision (scripts/bench_midbatch.py:
  // 10M x 768: stream 5.1 / 11.3 / 43.7 ms vs filter 3.5 / 3.6 / 3.8 ms at 1 / 8 / 32 queries - the filter reads the
  // half-size fp16 mirror once per <= 2048 queries, the stream scan reads the fp32 rows once per 4 queries;
  // 200k x 128: break-even near 32 queries).
  if (ix.n_rows_ < 65536 || k > 128) return false;
  const bool have_mirror = ix.mirror_ && ix.mirror_->version == ix.rows_version_;
  if (have_mirror && !ix.mirror_->fp16_range_ok) return false;
  if (nq < 8 && !have_mirror) return false;   // do not spend 50 % more HBM on a mirror for single-query traffic alone
  const double rows = (double)ix.n_rows_, d = (double)ix.dim_, dp = (double)((ix.dim_ + 127) / 128 * 128);
  const double stream_s = std::ceil((double)nq / 4.0) * rows * d * 4.0 / 6.0e12 + 0.05e-3;
  const double filter_s = 0.25e-3 + std::max(rows * dp * 2.0 / 5.0e12, 2.0 * 256.0 * std::ceil((double)nq / 256.0) * rows * dp / 1.2e15);
  return filter_s < stream_s;
}

int32_t flat_mfma_search_slice(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx, int cap_scale) {
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
  const int cap = std::max(4096, 64 * k) * cap_scale;   // candidate slots per query and stage
  if (!m.qh.reserve((size_t)b_pad * m.d_pad * 2) || !m.qstat.reserve((size_t)b_pad * 16) || !m.T.reserve((size_t)b_pad * 4) ||
      !m.cand.reserve((size_t)nq * cap * 8) || !m.cnt.reserve((size_t)(nq + 4) * 4 + 16))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)((b_pad + 3) / 4)), dim3(256), 0, s, dq, nq, b_pad, (int)ix.dim_,
                     m.d_pad, m.qh.as<_Float16>(), m.qstat.as<float>(), m.drop);
  // kernel choice: v5 / v7 want K-steps in pairs (d_pad % 128 == 0, >= 256); other shapes stay on v3
  const char* ver_s = getenv("EPS_MFMA_KERNEL");   // 3 | 7 (A/B); v7 needs K-steps in pairs, other shapes stay on v3
  const int version_env = ver_s && atoi(ver_s) == 3 ? 3 : 7;
  const int version = (version_env == 7 && (m.d_pad % 128 != 0 || m.d_pad < 256)) ? 3 : version_env;
  if (version >= 7) {
    if (!m.qf.reserve((size_t)b_pad * m.d_pad * 2)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
    hipLaunchKernelGGL(pack_qf_kernel, dim3((unsigned)((b_pad / 32) * (m.d_pad / 16))), dim3(64), 0, s, m.qh.as<_Float16>(),
                       m.qf.as<_Float16>(), b_pad, m.d_pad);
  }

  // Staging.  Every MFMA stage needs a valid upper bound T of the final k-th best exact key; it tightens stage by stage.
  //  * seeded (exact mode, no deleted bitset / attribute filter): the head [0, S0) goes through the SAME MFMA kernel in
  //    its approx-key mode with T = +inf, the k best approximate keys are re-ranked exactly, and their k-th exact key is
  //    the first T (any k exact keys bound the k-th best).  The stages then start at row 0; the exact re-rank dedups rows
  //    it meets twice.  Stage sizes grow by the cube root of n / S0, which minimises the re-ranked rows ~ k * sum(ratios).
  //  * otherwise: the head is scanned exactly (with the filter) by the stream kernel, stages 32 x and 256 x S0.
  int64_t S0 = std::max<int64_t>(4096, (int64_t)(32 * k + ROWPAD - 1) / ROWPAD * ROWPAD);
  const FilterSpec fs = ix.filter_spec();
  const bool seed_env = !(getenv("EPS_MFMA_SEED") && atoi(getenv("EPS_MFMA_SEED")) == 0);
  const bool seeded = seed_env && n > 4 * S0;   // with a filter the seeds are the k best VISIBLE head rows
  std::vector<int64_t> bounds;
  if (seeded) {
    bounds.push_back(approx ? S0 : 0);   // approx mode keeps the head's approximate keys themselves: no second visit
    const double r = std::max(4.0, std::cbrt((double)n / (double)S0));
    // stage boundaries on multiples of the rows one "round" of the persistent grid covers (256 workgroups x 256 rows /
    // query tiles), so the small stages do not end on a mostly idle round
    const int64_t qt = std::max<int64_t>(1, b_pad / 256);
    const int64_t unit = 256 * std::max<int64_t>(1, 256 / std::gcd<int64_t>(256, qt));
    for (double f : {r, r * r}) {
      int64_t bnd = (int64_t)((double)S0 * f) / ROWPAD * ROWPAD;
      if (bnd >= 2 * unit) bnd = bnd / unit * unit;
      if (bnd < n && bnd > bounds.back()) bounds.push_back(bnd);
    }
  } else {
    bounds.push_back(std::min(S0, n));
    for (int64_t bnd : {S0 * 32, S0 * 256}) {
      if (bnd < n && bnd > bounds.back()) bounds.push_back(bnd);
    }
  }
  if (bounds.back() < n) bounds.push_back(n);

  if (!seeded) {
    // stage 0: exact scan of the head
    rc = ix.flat_stream(dq, nq, k, 0, bounds[0], run_keys, false, -1, !approx);
    if (rc != EPS_OK) return rc;
  }
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
  fa.qf = m.qf.as<_Float16>();
  fa.base = ix.metric_ == 0 ? m.xn.as<float>() : m.zeros.as<float>();
  fa.base_s = ix.metric_ == 0 ? m.xn_s.as<float>() : m.zeros_s.as<float>();
  fa.T = m.T.as<float>();
  fa.d_pad = m.d_pad;
  fa.tiles_q = (int)(b_pad / BN3);
  fa.nq = nq;
  fa.s = ix.metric_ == 0 ? -2.f : -1.f;
  fa.inv_s = 1.f / fa.s;
  fa.cand = m.cand.as<u32>();
  fa.cand_keys = approx ? m.cand.as<u64>() : nullptr;
  fa.qstat = m.qstat.as<float>();
  fa.metric = ix.metric_;
  fa.cnt = cnt;
  fa.cap = cap;
  fa.group_sync = nullptr;
  fa.sync_shift = getenv("EPS_MFMA_SYNC_SHIFT") ? std::min(8, std::max(0, atoi(getenv("EPS_MFMA_SYNC_SHIFT")))) : 2;
  fa.dense = 0;
  fa.ablate = 0;

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

  const int bm = BM3;   // every kernel generation works on 256-row tiles
  const size_t shm = version >= 7 ? V7_LDS_BYTES : 2 * 65536 + 2 * 256 * sizeof(float);
  if (!m.num_cus) {   // per index (= per device): no process-wide state
    hipDeviceProp_t prop;
    m.num_cus = hipGetDeviceProperties(&prop, ix.device_) == hipSuccess ? prop.multiProcessorCount : 256;
    m.num_cus = m.num_cus / 8 * 8;
    if (m.num_cus < 8) m.num_cus = 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_filter_kernel_v3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 65536 + 2 * 256 * sizeof(float)));
    for (const void* fn : {reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_IDS>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_KEYS>),
                           reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_DENSE>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_IDS>),
                           reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_KEYS>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_DENSE>)})
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)V7_LDS_BYTES);
  }
  const int num_cus = m.num_cus;
  const bool narrow_env = !(getenv("EPS_MFMA_NARROW") && atoi(getenv("EPS_MFMA_NARROW")) == 0);
  const bool gsync_env = !(getenv("EPS_MFMA_GROUPSYNC") && atoi(getenv("EPS_MFMA_GROUPSYNC")) == 0);
  if (!m.gsync.reserve(1024)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  auto launch_filter = [&](const FilterArgs& f) {
    {
      FilterArgs f3 = f;
      f3.tiles_q = (int)(b_pad / BN3);
      if (version >= 7) {
        f3.group_sync = gsync_env ? m.gsync.as<u32>() : nullptr;
        if (f3.group_sync && f3.dense) (void)hipMemsetAsync(f3.group_sync, 0, 1024, s);   // (stages: reset by threshold_kernel)
        const int mode = f3.dense ? FM_DENSE : (f3.cand_keys ? FM_KEYS : FM_IDS);
        const dim3 grid((unsigned)num_cus), block(256);
        if (nq <= 128 && narrow_env) {   // one 128-query tile: half the padded MFMA work, the pass streams the mirror
          f3.tiles_q = 1;
          if (mode == FM_DENSE) hipLaunchKernelGGL((mfma_filter_kernel_v7<1, FM_DENSE>), grid, block, shm, s, f3);
          else if (mode == FM_KEYS) hipLaunchKernelGGL((mfma_filter_kernel_v7<1, FM_KEYS>), grid, block, shm, s, f3);
          else hipLaunchKernelGGL((mfma_filter_kernel_v7<1, FM_IDS>), grid, block, shm, s, f3);
        } else {
          if (mode == FM_DENSE) hipLaunchKernelGGL((mfma_filter_kernel_v7<2, FM_DENSE>), grid, block, shm, s, f3);
          else if (mode == FM_KEYS) hipLaunchKernelGGL((mfma_filter_kernel_v7<2, FM_KEYS>), grid, block, shm, s, f3);
          else hipLaunchKernelGGL((mfma_filter_kernel_v7<2, FM_IDS>), grid, block, shm, s, f3);
        }
      }
      else hipLaunchKernelGGL(mfma_filter_kernel_v3, dim3((unsigned)num_cus), dim3(512), shm, s, f3);
    }
  };
  if (seeded) {
    launch_fill_u64(reinterpret_cast<u64*>(m.T.p), b_pad / 2, 0x7F8000007F800000ull, s);   // T = +inf: every head row is a candidate
    const bool dense = version >= 7;   // v7 writes the head's keys densely (slot = row); older kernels append with atomics
    er = dense ? hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cnt), (int)S0, (size_t)nq, s) : hipMemsetAsync(cnt, 0, (size_t)nq * 4, s);
    if (er != hipSuccess) return ix.hip_fail(er, "memset");
    unsigned long long seed_stride = 0;   // != 0: the seed pass ran over the sample, ids are sample indices
    u32 seed_head = 0;
    FilterArgs f0 = fa;
    f0.dense = dense ? 1 : 0;
    f0.cand_keys = m.cand.as<u64>();
    if (!approx) {   // exact mode: seeds from a sample spread over the whole table (approx mode keeps the head's keys)
      const u32 sample_head = (u32)(S0 / 2);
      const unsigned long long sample_stride = (unsigned long long)(((unsigned __int128)(n - sample_head) << 32) / (unsigned __int128)(S0 - sample_head));
      if (m.sample_version != ix.rows_version_ || m.sample_n != n || m.sample_rows != S0) {   // (also after an append: n changed)
        if (!m.sxh.reserve((size_t)S0 * m.d_pad * 2) || !m.sbase.reserve((size_t)S0 * 4) || !m.sbase_u.reserve((size_t)S0 * 4))
          return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (seed sample)");
        hipLaunchKernelGGL(seed_sample_kernel, dim3((unsigned)S0), dim3(256), 0, s, m.xh.as<_Float16>(), fa.base_s, fa.base, sample_stride, sample_head, m.d_pad,
                           m.sxh.as<_Float16>(), m.sbase.as<float>(), m.sbase_u.as<float>());
        m.sample_version = ix.rows_version_;
        m.sample_n = n;
        m.sample_rows = S0;
      }
      f0.xh = m.sxh.as<_Float16>();
      f0.base_s = m.sbase.as<float>();
      f0.base = m.sbase_u.as<float>();
      seed_stride = sample_stride;
      seed_head = sample_head;
    }
    f0.tile0 = 0;
    f0.ntiles = (S0 + bm - 1) / bm;
    f0.row_hi = S0;
    launch_filter(f0);
    launch_merge_lists(f0.cand_keys, cap, k, nq, run_keys, false, s, cnt, approx ? nullptr : &fs, seed_stride, seed_head);   // k best approximate keys of the visible seeds
    if (!approx) {
      hipLaunchKernelGGL(seed_to_cand_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, run_keys, k, nq, fa.cand, cap, cnt, seed_stride, seed_head);
      launch_rerank(ra, s);                                                    // -> their exact keys
    }
  }
  // fp32 rounding of the keys the threshold compares: |x|^2, |q|^2 and the re-ranked distance are each a 64-lane sum of
  // d_pad/64 sequential fmas per lane plus a 6-level shuffle tree, i.e. <= (d_pad/64 + 6) * 2^-24 relative to their
  // magnitude each; doubled for safety.  (A fixed 8e-6 was only enough up to d ~ 1000.)
  const float rerank_slack = std::max(8e-6f, 2.f * (3.f * ((float)m.d_pad / 64.f + 6.f) + 2.f) * 5.9604645e-8f);
  bool first = true;
  for (size_t st = 0; st + 1 < bounds.size(); ++st) {
    const int64_t lo = bounds[st], hi = bounds[st + 1];
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, s, run_keys, k, nq, b_pad,
                       m.qstat.as<float>(), m.scal.as<float>(), ix.metric_, m.T.as<float>(), cnt, m.gsync.as<u32>(), rerank_slack);
    fa.tile0 = lo / bm;
    fa.ntiles = (hi + bm - 1) / bm - fa.tile0;
    fa.row_hi = hi;
    const bool biggest = (st + 2 == bounds.size());
    if (biggest) (void)hipEventRecord(ix.evk0_, s);
    if (biggest) {
      ix.stats_.main_kernel_rows = hi - lo;
      ix.stats_.main_kernel_queries = nq;
    }
    launch_filter(fa);
    if (biggest) (void)hipEventRecord(ix.evk1_, s);
    hipLaunchKernelGGL(stage_counts_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, cnt, nq, cap, overflow, total);
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
      fprintf(stderr, "[eps] stage %zu rows [%lld,%lld) tiles %lld: cnt min %u mean %.1f max %u (>cap: %lld), T min %g max %g, empty kth %lld, scal %g %g %g\n", st, (long long)lo, (long long)hi, (long long)fa.ntiles, mn, sum / nq, mx, (long long)big, tmin, tmax, (long long)empt, m.h_scal[0], m.h_scal[1], m.h_scal[2]);
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
  ix.stats_.dist_evals += nq * (n - bounds[0]) + (seeded ? nq * S0 : 0);   // (exact mode visits the head twice)
  ix.stats_.main_kernel_launches = 1;
  if (h.overflow) {
    ix.stats_.overflow_queries += h.overflow;
    if (!approx && !fa.ablate) {
      // (selective filters inflate the lists by 1 / pass fraction, adversarial row orders by more): first retry with 16 x
      // the candidate slots - re-ranking tens of thousands of rows per query is still ~50 x cheaper than the stream scan
      // of a large batch - then the exact stream engine
      if (cap_scale == 1 && (size_t)nq * cap * 16 * 8 <= ((size_t)4 << 30)) return flat_mfma_search_slice(ix, dq, nq, k, run_keys, approx, 16);
      return ix.flat_stream(dq, nq, k, 0, n, run_keys, false);
    }
  }
  return EPS_OK;
}

// Batches beyond 2048 queries run as slices of 2048: the kernel keeps one slice's fp16 query tile set (3 MB) resident in
// each XCD's 4 MB L2 while the row operand streams past; at 4096 / 8192 queries per pass the query fragments thrash L2
// and the filter drops to 0.37 / 0.27 of the MFMA peak (0.46 in slices; bench.py --rows 1250000 --batch 8192).
int32_t flat_mfma_search(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx) {
  const int64_t slice = getenv("EPS_MFMA_MAX_BATCH") ? std::max(256, atoi(getenv("EPS_MFMA_MAX_BATCH"))) : 2048;
  if (nq <= slice) return flat_mfma_search_slice(ix, dq, nq, k, run_keys, approx);
  for (int64_t q0 = 0; q0 < nq; q0 += slice) {   // the counters in ix.stats_ accumulate over the slices
    const int32_t rc = flat_mfma_search_slice(ix, dq + q0 * ix.dim_, std::min(slice, nq - q0), k, run_keys + q0 * k, approx);
    if (rc != EPS_OK) return rc;
  }
  return EPS_OK;
}

}  // namespace eps
