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
// 8-bit first pass (r3).  The matrix pipe multiplies int8 operands at twice the fp16 rate (v_mfma_i32_32x32x32_i8: measured
// 3.42 POP/s against 1.75 PFLOP/s for the fp16 instruction in the same loop, profiles/r3_mfma_peak_i8_vs_fp16.txt), and a
// lower-bound filter may use any operand whose error it can bound.  Rows and queries are quantised on ONE grid per index,
//     xh = z + sx * xi,  xi = clamp(rint((x - z) / sx), -127, 127),   z = (min + max) / 2,  sx = (max - min) / 254
// so that the dot product of the quantised vectors is exact integer arithmetic plus per-row and per-query constants:
//     qh.xh = d z^2 + z sx SX[x] + z sx SQ[q] + sx^2 * sum_k qi_k xi_k          (SX, SQ: sums of the int8 values)
//     approx(q,x) = R[x] + C[q] - u * dot,   u = |s| sx^2,   R, C: the row / query constants (kernels below)
//     |key - approx| <= |s| * (|q| * |x - xh| + |q - qh| * |xh|)                 (Cauchy-Schwarz on the stored residuals; no
//                                                                                 accumulation slack: the int32 sum is exact)
// A row passes iff  dot + acc0[x] >= Tq[q]  with acc0 = ceil(-R/u) folded into the accumulator's start value and
// Tq = floor((C - T)/u): one integer max + compare per 16 outputs, as in the fp16 kernel.  On U[0,1) rows at d = 768 the margin is
// ~2.0 key units (fp16: 0.03) against a spread of 5.5 per sigma of the distance distribution: a few dozen candidates per query
// in the last stage instead of ~10 - noise for the fp32 re-rank.  The kernel is the v7 kernel with the MFMA instruction and the
// accumulator type exchanged (mfma_kernels.hpp, V7Op): a K-step moves the same 128 bytes per row, but covers 128 dimensions
// instead of 64.  Tables the grid does not fit (all values equal, non-finite values, constants beyond int32) and batches whose
// candidate lists overflow fall back to the fp16 engine.
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
#include "stream8_kernel.hpp"

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
  DevBuf seedc;    // u32 [b][k]: rows of the k best seeds of every query (the seed stage's candidate lists)
  DevBuf partk, partd;   // a handful of queries: per-workgroup partial top-k lists of the split re-rank [b][8][k], arrival counters [b] (zero between launches)
  // 8-bit mirror (first-pass operand of the filter, see above)
  DevBuf x8;       // int8 [n_pad8][d_pad8]
  DevBuf acc0;     // int32 [n_pad8]: accumulator start of every row = ceil(-R/u) + 1 (-2^30 on padding rows)
  DevBuf sx8, sacc0;            // seed sample of the 8-bit mirror
  int64_t sample8_version = -1, sample8_n = 0, sample8_rows = 0;
  DevBuf scal8;    // float [8]: max |x' - xh'|, max |xh'|, max |x|^2, bad flag, max |R|, |mu|, max |x'| (during the build: min / max of x - mean as ordered u32 in [6], [7])
  // r4, per-row margins: the two norms of every row the Cauchy-Schwarz margin multiplies the query's with (erow = +inf: a row whose
  // constant leaves the accumulator's range - it is not tested, it always passes), the batch's folded start values, the maxima with
  // the two margin entries zeroed (what thresholds of folded launches read), the batch's largest query norms, the range histogram
  DevBuf erow, hrow;   // float [n_pad8]
  DevBuf acc0b;        // int32 [n_pad8]: acc0 + the row's margin for the CURRENT batch (fold8_kernel)
  DevBuf scal8f;       // float [8]
  DevBuf qmax;         // u32 [2]: float bits of the batch's max |q'| and max |q' - qh'| (query_prep8_kernel, atomicMax)
  DevBuf hist;         // u32 [4096 + 8]: histogram of x - mean over the sample; [4096]: forced rows
  DevBuf q8;       // int8 [b_pad][d_pad8]
  float h_scal8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  DevBuf mu8;      // float [d_pad8]: the grid's centre, one value per column (zeros beyond dim)
  float step8 = 0.f;            // the grid: xh' = step8 * xi around mu8
  // r6: the grid's frame (device_common.hpp, rot256_load): rot8 = rows and queries are quantised as R x; sp8 = int32 [d_pad8], R's
  // permutation and signs.  Chosen on the first build from the steps the two frames need on the same sample; kept when rows are appended.
  bool rot8 = false;
  int rot_w8 = 0;               // columns the rotation covers: dim rounded up to 256 (<= d_pad8; the columns beyond stay zero)
  DevBuf sp8;
  float step8_identity = 0.f, step8_rotated = 0.f;   // what the choice saw (stats; 0 = that frame was not measured)
  bool i8_trusted = false;      // the library's own choice has seen a batch through the 8-bit pass on this mirror (no probe needed)
  int64_t version8 = -1, n8 = 0, n_pad8 = 0, forced_rows8 = 0;
  int64_t epoch8 = 0;           // full (re)builds of the 8-bit mirror (an extension keeps the grid and every existing row's constant)
  bool fold8 = false;           // exact-mode users fold per-row margins per batch (rows differ); else table-wide margin in the thresholds
  int d_pad8 = 0;
  bool i8_ok = false;
  int i8_overflows = 0;         // consecutive batches whose 8-bit pass overflowed its candidate lists (the fp16 pass then answered)
  // r4, a handful of queries in one pass (stream8_kernel.hpp): the shared best-accumulator tables + raw candidate counters, the raw lists
  DevBuf s8g, s8raw;            // table slots (S8_TABLE_WORDS) + per-wavefront candidate counts;  u64 [nq][waves][S8_WAVE_CAP]
  DevBuf s8mask;                // r5: u8 [(n + 7) / 8] - a call's compiled filter PROGRAM (and bitset, and column test) evaluated once per row into
                                // one bitset (bit set = row invisible), which the pass and its re-rank then read as a deleted bitset
  // rows version on which the one-pass form overflowed twice in a row (the staged chain serves it); [0]: k <= 16, [1]: k = 17..64 - a larger k
  // passes more rows against the same lists, and must not talk the table out of the form for the small-k traffic
  int64_t s8_declined_version[6] = {-1, -1, -1, -1, -1, -1};
  int s8_overflows[6] = {0, 0, 0, 0, 0, 0};
  // under a deleted bitset / an int-column filter an overflow usually means "fewer than k rows visible": the rows' version says nothing
  // about it, so two such overflows in a row make the next 32 filtered calls go straight to the staged chain, then the one-pass form is tried again
  int s8_filt_overflows[6] = {0, 0, 0, 0, 0, 0}, s8_filt_skip[6] = {0, 0, 0, 0, 0, 0};   // (per class, as above)
  int s8_cus = 0;               // CUs of the device (grid of the one-pass kernel)
  // r5: the call's two result counters land in host-mapped memory (written by the last block of the re-rank launch), read after the stream
  // sync: no device-to-host copy at the end of a 0.2 ms call
  struct HostWords {
    u32* p = nullptr;
    ~HostWords() { if (p) (void)hipHostFree(p); }
    bool get() {
      if (!p && hipHostMalloc(reinterpret_cast<void**>(&p), 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
      return p != nullptr;
    }
  } s8_pub;
  // r5: the re-rank launch of a one-pass call leaves the table slots empty and the counters zero (RerankArgs::s8_reset), and the pass quantises its
  // queries itself - the next call is TWO launches, no prep launch.  s8_clean_* = the buffers that state lives in; anything else that writes
  // them (the staged chain's counters, a reallocation, a failed call) clears it and the next call starts with the prep launch again
  const void* s8_clean_cnt = nullptr;
  const void* s8_clean_g = nullptr;
  int64_t extended_rows8 = 0;
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
    // A component beyond the fp16 range is stored as +-65504, not +-inf: inf * 0 would make NaN accumulators that fail every
    // `acc >= T` and silently drop rows from an exact answer.  Clamped, the component's residual enters |q - qh| like any
    // other rounding error: the bound stays valid (and becomes so loose that the batch ends on the stream engine).
    const _Float16 h = to_half_drop(fminf(fmaxf(x, -65504.f), 65504.f), drop);
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

// ------------------------------------------------------------------------------------------------ 8-bit mirror
// r4: the grid is CENTRED.  Rows and queries are quantised as x' = x - mu on a symmetric grid, xh' = step * xi, with one mu per
// COLUMN (column means of a strided sample of the table + the mid-range of what is left, so that the grid is symmetric).  Distances
// do not care: |x - q|^2 = |x' - q'|^2, q.x = q'.x' + mu.x' + q.mu - the cross terms are a per-row and a per-query constant, exact in
// fp32, folded into R[x] and C[q] - and the Cauchy-Schwarz margin now scales with |q - mu| and |x - mu| instead of |q| and |x|:
// half the margin on U[0,1) rows (every candidate the filter passes for nothing costs a 3 KB gather in the re-rank, an fp32 row in
// the traversal), a third on tables whose columns have their own means, and tables far from the origin lose nothing.  ANY mu keeps
// the bound valid (tests/test_bound_math.py::test_any_centre_keeps_the_bound_valid): it only has to be the same vector for the rows
// and the queries, so it is fixed when the mirror is first built and kept when rows are appended.
constexpr int CENTRE_SEG = 32;   // segments of the sample, summed in a fixed order: mu is bit-reproducible (the build's approximate
                                 // kNN keys depend on it, and two builds of one table must give the same graph)
// partial column sums of sample rows r = (seg * per_seg + i) * stride, i < per_seg:  part[seg][col]
// (clamp_mean != null: every value is clamped into [clamp_mean[col] + clo, clamp_mean[col] + chi] first - the second, ROBUST estimate of the
// column means: an outlier of 30 000 in a 65 536-row sample would otherwise move its column's centre by half the grid)
__global__ __launch_bounds__(256) void colsum_kernel(const float* rows, int64_t n, int dim, int64_t stride, int64_t per_seg, float* part, const float* clamp_mean,
                                                     float clo, float chi) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int64_t seg = blockIdx.y;
  float s = 0.f;
  if (col < dim) {
    for (int64_t i = sub; i < per_seg; i += 4) {
      const int64_t r = (seg * per_seg + i) * stride;
      if (r < n) {
        float v = rows[r * dim + col];
        if (clamp_mean) v = fminf(fmaxf(v, clamp_mean[col] + clo), clamp_mean[col] + chi);
        s += v;
      }
    }
  }
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && col < dim) part[seg * dim + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// mean[col] = sum of the segments (fixed order) / sampled rows; columns beyond dim: 0
__global__ void colmean_kernel(const float* part, int dim, int d_pad8, float inv_count, float* mu) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= d_pad8) return;
  float s = 0.f;
  if (col < dim)
    for (int g = 0; g < CENTRE_SEG; ++g) s += part[g * dim + col];
  mu[col] = col < dim ? s * inv_count : 0.f;
}
// value range of x - mean over the whole table (ordered-u32 images, so atomicMin / atomicMax work on them): scal8[6] = min,
// scal8[7] = max; one wavefront per row, grid-stride
template <bool ROT>
__global__ __launch_bounds__(256) void minmax_kernel(const float* rows, int64_t n, int dim, const float* mean, u32* scal8, const int* sp, int d_pad8) {
  float lo = __builtin_inff(), hi = -__builtin_inff();
  bool bad = false;
  const int lane = lane_id();
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const bool vec = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += nwaves) {
    const float* src = rows + r * dim;
    if (ROT) {   // the rotated frame (device_common.hpp, rot256_load): d_pad8 here = the rotation's width
      for (int c = lane * 4; c < d_pad8; c += 256) {
        double xd[4];
        rot256_load(src, dim, sp, c, lane, xd);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = (float)(xd[e] - (double)mean[c + e]);
          lo = fminf(lo, a);
          hi = fmaxf(hi, a);
          bad |= !(fabsf(a) < 3.0e38f);
        }
      }
    } else if (vec) {
      for (int c = lane * 4; c < dim; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        const float4 m = *reinterpret_cast<const float4*>(mean + c);
        const float a = v.x - m.x, b = v.y - m.y, c2 = v.z - m.z, d2 = v.w - m.w;
        lo = fminf(fminf(lo, a), fminf(fminf(b, c2), d2));
        hi = fmaxf(fmaxf(hi, a), fmaxf(fmaxf(b, c2), d2));
        bad |= !(fabsf(v.x) < 3.0e38f) || !(fabsf(v.y) < 3.0e38f) || !(fabsf(v.z) < 3.0e38f) || !(fabsf(v.w) < 3.0e38f);
      }
    } else {
      for (int c = lane; c < dim; c += 64) {
        const float a = src[c] - mean[c];
        lo = fminf(lo, a);
        hi = fmaxf(hi, a);
        bad |= !(fabsf(src[c]) < 3.0e38f);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o));
    hi = fmaxf(hi, __shfl_xor(hi, o));
  }
  if (lane == 0) {
    if (lo <= hi) {
      atomicMin(&scal8[6], f2ord(lo + 0.0f));
      atomicMax(&scal8[7], f2ord(hi + 0.0f));
    }
  }
  if (__any(bad) && lane == 0) atomicMax(&scal8[3], __float_as_uint(1.f));   // inf / NaN somewhere: the grid would be meaningless
}
// mu = mean + z0 (z0: mid-range of x - mean, so that the grid is symmetric around 0); scal8[5] = |mu| (enters the fp32 slack of IP / COSINE)
__global__ __launch_bounds__(64) void mu_finish_kernel(float* mu, int dim, float z0, float* scal8) {
  float s2 = 0.f;
  for (int c = lane_id(); c < dim; c += 64) {
    const float v = mu[c] + z0;
    mu[c] = v;
    s2 = fmaf(v, v, s2);
  }
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
  if (lane_id() == 0) scal8[5] = sqrtf(s2) * 1.00001f;
}

// r4: the grid's range is set by the BULK of the values: 4096-bin histogram of x - mean over the sample rows (integer atomics: the
// result does not depend on the order), the host cuts both tails at max(2, 1e-7 x values) sample values.  One outlier value used to
// stretch the grid - and with it every row's margin - by its distance; now its row is clamped and pays with ITS OWN residual.
__global__ __launch_bounds__(256) void centre_hist_kernel(const float* rows, int64_t n, int dim, int64_t stride, int64_t sampled, const float* mean, float lo,
                                                          float inv_binw, u32* hist) {
  __shared__ u32 h[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
  __syncthreads();
  for (int64_t j = blockIdx.x; j < sampled; j += gridDim.x) {
    const int64_t r = j * stride;
    if (r >= n) break;
    const float* src = rows + r * dim;
    for (int c = threadIdx.x; c < dim; c += 256) {
      int b = (int)((src[c] - mean[c] - lo) * inv_binw);
      b = b < 0 ? 0 : (b > 4095 ? 4095 : b);
      atomicAdd(&h[b], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}
// the rotated frame's column statistics (means, histogram) come from the same kernels, run over the rotated images of the sample rows:
// out[j][0 .. d_pad8) = R rows[j * stride], fp32 (statistics only: any centre and any step keep the bound valid)
__global__ __launch_bounds__(256) void rot_sample_kernel(const float* rows, int64_t n, int dim, int64_t stride, int64_t sampled, int d_pad8, const int* sp, float* out) {
  const int lane = lane_id();
  for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < sampled; j += (int64_t)gridDim.x * 4) {
    const int64_t r = j * stride;
    if (r >= n) break;
    for (int c = lane * 4; c < d_pad8; c += 256) {
      double xd[4];
      rot256_load(rows + r * dim, dim, sp, c, lane, xd);
      *reinterpret_cast<float4*>(out + j * d_pad8 + c) = make_float4((float)xd[0], (float)xd[1], (float)xd[2], (float)xd[3]);
    }
  }
}
constexpr int ACC_FORCE = 0x38000000;   // start value of a row that must pass whatever the threshold (thresholds <= TQ_MAX8, |dot| < 2^27)
// the batch's margins folded into the rows' start values: acc0b[x] = acc0[x] + ceil(|s| (Qn E[x] + Eq H[x]) / u) + 1, Qn / Eq = the batch's
// largest |q'| / |q' - qh'| (>= every query's own margin for row x); forced rows and rows whose margin leaves the range: ACC_FORCE
__global__ __launch_bounds__(256) void fold8_kernel(const int* acc0, const float* erow, const float* hrow, int64_t n, int64_t n_pad, const u32* qmax, float s_abs,
                                                    float inv_u, int* acc0b) {
  const float qn = __uint_as_float(qmax[0]), eq = __uint_as_float(qmax[1]);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_pad; r += (int64_t)gridDim.x * 256) {
    int v = acc0[r];
    if (r < n) {
      const float add = ceilf(s_abs * (qn * erow[r] + eq * hrow[r]) * inv_u) + 1.f;
      v = add < 536870912.f ? v + (int)add : ACC_FORCE;   // (erow = +inf, or a NaN: forced)
    }
    acc0b[r] = v;
  }
}
// after (re)quantising: scal8[6] = max |x'| bound (for the thresholds' fp32 slack); scal8f = scal8 with the two margin entries zeroed
__global__ void scal_finish_kernel(float* scal8, float* scal8f) {
  if (threadIdx.x == 0) scal8[6] = scal8[0] + scal8[1];
  __syncthreads();
  if (threadIdx.x < 8) scal8f[threadIdx.x] = threadIdx.x < 2 ? 0.f : scal8[threadIdx.x];
}

// rows [row0, n_pad) are (re)written, as in half_mirror_kernel.  x' = x - mu;  metric 0: R = |x'|^2; otherwise R = -mu.x'.  u = |s| step^2.
// ROT: the table's rotated frame (device_common.hpp, rot256_load) - x' = fl32(R x - mu); the first rot_w = ceil(dim / 256) * 256 columns carry values
template <bool ROT>
__global__ __launch_bounds__(256) void quant_mirror_kernel(const float* rows, int64_t row0, int64_t n, int64_t n_pad, int dim, int d_pad8, const float* mu, float step,
                                                           float inv_step, float inv_u, int metric, signed char* x8, int* acc0, float* scal8, float* erow,
                                                           float* hrow, u32* forced_count, const int* sp, int rot_w) {
  const int lane = lane_id();
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float m_e1 = 0.f, m_nxh = 0.f, m_xn = 0.f, m_bad = 0.f, m_r = 0.f, m_emin = __builtin_inff();
  const bool vec = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
  for (int64_t r = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_pad; r += nwaves) {
    signed char* dst = x8 + r * d_pad8;
    if (r >= n) {
      for (int c = lane * 4; c < d_pad8; c += 256) *reinterpret_cast<u32*>(dst + c) = 0u;
      if (lane == 0) {
        acc0[r] = -(1 << 30);
        erow[r] = 0.f;
        hrow[r] = 0.f;
      }
      continue;
    }
    const float* src = rows + r * dim;
    float s2 = 0.f, e2 = 0.f, h2 = 0.f, c2 = 0.f, mx = 0.f;
    for (int c = lane * 4; c < d_pad8; c += 256) {   // d_pad8 is a multiple of 256
      float xs[4] = {0.f, 0.f, 0.f, 0.f};
      double xd[4] = {0.0, 0.0, 0.0, 0.0};
      if (ROT) {
        if (c < rot_w) rot256_load(src, dim, sp, c, lane, xd);   // (columns [rot_w, d_pad8): padding, zero codes)
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[e] = (float)xd[e];   // (enters |x|^2 only: a slack scale)
      } else if (vec) {
        if (c < dim) {
          const float4 v = *reinterpret_cast<const float4*>(src + c);
          xs[0] = v.x; xs[1] = v.y; xs[2] = v.z; xs[3] = v.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[e] = c + e < dim ? src[c + e] : 0.f;
      }
      const float4 mv = *reinterpret_cast<const float4*>(mu + c);   // (mu has d_pad8 entries, zeros beyond dim)
      const float ms[4] = {mv.x, mv.y, mv.z, mv.w};
      u32 packed = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ROT ? c < rot_w : c + e < dim) {
          const float dx = ROT ? (float)(xd[e] - (double)ms[e]) : xs[e] - ms[e];   // x' (one rounding in either frame)
          const int xi = quant8(dx, 0.f, inv_step);
          const float res = fmaf(-step, (float)xi, dx);     // x' - xh'
          const float xh = step * (float)xi;
          packed |= (u32)(xi & 255) << (8 * e);
          s2 = fmaf(xs[e], xs[e], s2);
          e2 = fmaf(res, res, e2);
          h2 = fmaf(xh, xh, h2);
          c2 = fmaf(dx, dx, c2);
          mx = fmaf(ms[e], dx, mx);
        }
      }
      *reinterpret_cast<u32*>(dst + c) = packed;
    }
    for (int o = 32; o > 0; o >>= 1) {
      s2 += __shfl_xor(s2, o);
      e2 += __shfl_xor(e2, o);
      h2 += __shfl_xor(h2, o);
      c2 += __shfl_xor(c2, o);
      mx += __shfl_xor(mx, o);
    }
    const float R = metric == 0 ? c2 : -mx;
    const float a0 = ceilf(-R * inv_u) + 1.f;
    const float e1 = sqrtf(e2) * 1.00001f + 1.2e-7f * sqrtf(c2) + (ROT ? 1e-12f * sqrtf(s2) : 0.f);   // (+ the rounding of x - mu itself; ROT: + the fp64 transform's)
    const float nxh = sqrtf(h2) * 1.00001f;
    if (s2 != s2 || !(s2 < 3.0e38f)) m_bad = 1.f;
    // |acc0| must stay below 2^29 (the dot product adds < 2^27).  A row beyond that - an outlier far outside the clipped grid - is FORCED:
    // never selected on approximate keys (acc0 = -2^30, as on padding rows), always passed by the exact filter (erow = +inf -> fold8_kernel
    // gives it ACC_FORCE); it does not enter the table's maxima (they only serve rows that are tested)
    const bool forced = !(fabsf(a0) < 536870912.f);
    if (lane == 0) {
      acc0[r] = forced ? -(1 << 30) : (int)a0;
      erow[r] = forced ? __builtin_inff() : e1;
      hrow[r] = nxh;
      if (forced) atomicAdd(forced_count, 1u);
    }
    if (forced) continue;
    m_emin = fminf(m_emin, e1);
    m_e1 = fmaxf(m_e1, e1);
    m_nxh = fmaxf(m_nxh, nxh);
    m_xn = fmaxf(m_xn, s2);
    m_r = fmaxf(m_r, fabsf(R));
  }
  if (lane == 0) {
    atomic_max_pos(&scal8[0], m_e1);
    atomic_max_pos(&scal8[1], m_nxh);
    atomic_max_pos(&scal8[2], m_xn);
    if (m_bad != 0.f) atomic_max_pos(&scal8[3], 1.f);
    atomic_max_pos(&scal8[4], m_r);
    atomicMin(reinterpret_cast<unsigned int*>(&scal8[7]), __float_as_uint(m_emin));   // (smallest residual norm of a tested row: non-negative floats order like their bits)
  }
}

// queries on the table's grid.  qstat[r] = |q|^2, |q'|, |q' - qh'|, C[q] (the constant that turns u-scaled accumulators into
// approximate distances: dist ~ a.s * acc + qstat[3]);  q' = q - mu;  C = |q'|^2 (L2), 1 - q.mu (COSINE), -q.mu (DOT)
// (r4) what the flat engine used to do in two more launches of its own, for calls that are a chain of short dependent launches (one
// query: every launch is ~5 us of latency): the fragment-major copy of the query operand the v7 kernel reads (pack_qf_kernel) and the
// start state of a seeded call (seed_prologue_kernel).  All-null: plain query_prep8 (the traversal's prefilter).
struct Prep8Extra {
  signed char* qf = nullptr;   // fragment-major copy: [b_pad/32][d_pad8/32][64 lanes][16 bytes]
  u64* T2 = nullptr;           // prologue: thresholds (pairs), n2 entries, value Tv
  int64_t n2 = 0;
  u64 Tv = 0;
  u32* cnt = nullptr;          // prologue: cnt[0 .. nq) = cntv, cnt[nq .. nq + 8) = 0
  u32 cntv = 0;
  u32* gsync = nullptr;        // prologue: 256 group counters = 0
  u32* qmax = nullptr;         // [2] (zeroed by the caller): atomicMax of the float bits of |q'| and |q' - qh'| over the batch (fold8_kernel reads them)
  int* s8g = nullptr;          // one-pass form (stream8_kernel.hpp): table slots = empty, raw candidate counters = 0
  int s8_slots = S8_SLOTS;     // slots per query of that call (64 | 128)
};
template <bool ROT>
__global__ __launch_bounds__(256) void query_prep8_kernel(const float* q, int64_t nq, int64_t b_pad, int dim, int d_pad8, const float* mu, float step, float inv_step,
                                                          int metric, signed char* q8, float* qstat, Prep8Extra x, const int* sp, int rot_w) {
  if (blockIdx.x == 0) {   // the seeded call's start state (nothing in this launch reads it)
    for (int64_t i = threadIdx.x; x.T2 && i < x.n2; i += 256) x.T2[i] = x.Tv;
    for (int64_t i = threadIdx.x; x.cnt && i < (x.s8g ? S8_MAX_Q + 8 : nq + 8); i += 256) x.cnt[i] = i < nq ? x.cntv : 0u;
    if (x.gsync) x.gsync[threadIdx.x] = 0;
    if (x.s8g) {   // (the slots of the call's queries - at least of the first four: a later call of 1-2 queries that skips this launch relies on its own
                   // slots being empty, and the re-rank of every call restores exactly the slots it used - and S8_MAX_Q + 8 counter words)
      const int qinit = nq > 4 ? (int)nq : 4;
      for (int i = threadIdx.x; i < qinit * x.s8_slots; i += 256) x.s8g[i * S8_SLOT_STRIDE] = S8_EMPTY;
    }
  }
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= b_pad) return;
  const int lane = lane_id();
  signed char* dst = q8 + r * d_pad8;
  // where byte column c of row r lives in the fragment-major copy
  const int64_t fbase = x.qf ? ((r >> 5) * (int64_t)(d_pad8 >> 5)) * 1024 + (r & 31) * 16 : 0;
  auto fput = [&](int c, u32 v) {
    if (x.qf) *reinterpret_cast<u32*>(x.qf + fbase + (int64_t)(c >> 5) * 1024 + ((c >> 4) & 1) * 512 + (c & 15)) = v;
  };
  if (r >= nq) {
    for (int c = lane * 4; c < d_pad8; c += 256) {
      *reinterpret_cast<u32*>(dst + c) = 0u;
      fput(c, 0u);
    }
    if (lane == 0) qstat[r * 4 + 0] = qstat[r * 4 + 1] = qstat[r * 4 + 2] = qstat[r * 4 + 3] = 0.f;
    return;
  }
  const float* src = q + r * dim;
  float s2 = 0.f, e2 = 0.f, c2 = 0.f, qm = 0.f;
  for (int c = lane * 4; c < d_pad8; c += 256) {
    u32 packed = 0;
    double xd[4] = {0.0, 0.0, 0.0, 0.0};
    if (ROT && c < rot_w) rot256_load(src, dim, sp, c, lane, xd);   // the query in the table's rotated frame
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ROT ? c < rot_w : c + e < dim) {
        const float xv = ROT ? (float)xd[e] : src[c + e];
        const float m = mu[c + e];
        const float dx = ROT ? (float)(xd[e] - (double)m) : xv - m;
        const int qi = quant8(dx, 0.f, inv_step);
        const float res = fmaf(-step, (float)qi, dx);
        packed |= (u32)(qi & 255) << (8 * e);
        s2 = fmaf(xv, xv, s2);
        c2 = fmaf(dx, dx, c2);
        e2 = fmaf(res, res, e2);
        qm = fmaf(xv, m, qm);
      }
    }
    *reinterpret_cast<u32*>(dst + c) = packed;
    fput(c, packed);
  }
  for (int o = 32; o > 0; o >>= 1) {
    s2 += __shfl_xor(s2, o);
    e2 += __shfl_xor(e2, o);
    c2 += __shfl_xor(c2, o);
    qm += __shfl_xor(qm, o);
  }
  if (lane == 0) {
    const float nqc = sqrtf(c2) * 1.000001f, eqc = sqrtf(e2) * 1.00001f + 1.2e-7f * sqrtf(c2) + (ROT ? 1e-12f * sqrtf(s2) : 0.f);
    qstat[r * 4 + 0] = s2;
    qstat[r * 4 + 1] = nqc;
    qstat[r * 4 + 2] = eqc;
    qstat[r * 4 + 3] = metric == 0 ? c2 : (metric == 1 ? 1.f - qm : -qm);
    if (x.qmax) {   // (non-negative floats order like their bit patterns; a NaN's pattern is above every number: its batch forces every row)
      atomicMax(&x.qmax[0], __float_as_uint(nqc));
      atomicMax(&x.qmax[1], __float_as_uint(eqc));
    }
  }
}

// (the frame is the mirror's: HalfMirror::rot8 / sp8)
static void launch_query_prep8(dim3 grid, hipStream_t s, const int* sp, int rot_w, const float* q, int64_t nq, int64_t b_pad, int dim, int d_pad8, const float* mu, float step,
                               int metric, signed char* q8, float* qstat, const Prep8Extra& x) {
  if (sp) hipLaunchKernelGGL(query_prep8_kernel<true>, grid, dim3(256), 0, s, q, nq, b_pad, dim, d_pad8, mu, step, 1.f / step, metric, q8, qstat, x, sp, rot_w);
  else hipLaunchKernelGGL(query_prep8_kernel<false>, grid, dim3(256), 0, s, q, nq, b_pad, dim, d_pad8, mu, step, 1.f / step, metric, q8, qstat, x, (const int*)nullptr, 0);
}

// T[j]: pass threshold of query j for the next filter launch, from the current k-th best key (formulas: device_common.hpp,
// stage_threshold8 / stage_threshold16).  Also resets what the launch accumulates into (candidate counts, group arrival counters).
// In exact mode a stage's re-rank computes the next stage's thresholds itself (RerankArgs::fuse); these kernels serve the
// approx mode (kNN build), the unseeded staging, and the padding entries.
// One-pass calls under a compiled filter PROGRAM (r5): the pass itself cannot call the evaluator (a function call = scratch memory for every
// wavefront of an HBM-bound stream), and it does not have to - whether a row is visible does not depend on the query, so the whole predicate
// (deleted bitset, int-column test, program; @distance = 0 as PreFilterBruteForceSearch evaluates it, vec_search_executor.cpp:795) is
// evaluated ONCE per row into a bitset in the deleted bitset's layout (bit i of byte i >> 3 set = row i is NOT visible) by this launch,
// and pass + re-rank read that.  One thread per ROW, the wavefront's 64 verdicts gathered by a ballot, 8 lanes write the 8 bytes (a thread per
// byte evaluated its 8 rows one after the other - 8 dependent attribute loads - and the launch took 21 us at 1M rows under rocprofv3).
__global__ __launch_bounds__(256) void filter_mask_kernel(FilterSpec f, int64_t n, uint8_t* mask) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool visible = r < n && row_visible(f, (u32)r, 0.f);
  const unsigned long long hidden = ~__ballot(visible);   // bit l: row (first row of the wavefront + l) is NOT visible (rows >= n: hidden)
  const int lane = lane_id();
  const int64_t wbase = r - lane;                          // (a multiple of 64)
  if (lane < 8 && wbase + lane * 8 < n) mask[(wbase >> 3) + lane] = (uint8_t)(hidden >> (lane * 8));
}

__global__ void threshold8_kernel(const u64* run_keys, int k, int64_t nq, int64_t b_pad, const float* qstat, const float* scal8, int metric,
                                  float u, int* T, u32* cnt, u32* gsync, float slack, int approx, int pad_only) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b_pad) return;
  if (!pad_only && gsync && j < 256) gsync[j] = 0;   // (all 256 counters, whatever nq: b_pad >= 256)
  if (j >= nq) {
    T[j] = 0x7FFFFFFF;   // padding queries never pass
    return;
  }
  if (pad_only) return;
  cnt[j] = 0;
  const u64 kth = run_keys[j * k + (k - 1)];
  // fewer than k visible rows so far: everything passes (bounded by the candidate cap)
  T[j] = kth == KEY_EMPTY ? -(1 << 30) : stage_threshold8(key_dist(kth), qstat + j * 4, scal8, metric, u, slack, approx);
}

__global__ void threshold_kernel(const u64* run_keys, int k, int64_t nq, int64_t b_pad, const float* qstat,
                                 const float* scal, int metric, float* T, u32* cnt, u32* gsync, float slack, int approx, int pad_only) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b_pad) return;
  if (!pad_only && gsync && j < 256) gsync[j] = 0;
  if (j >= nq) {
    T[j] = -__builtin_inff();
    return;
  }
  if (pad_only) return;
  cnt[j] = 0;
  const u64 kth = run_keys[j * k + (k - 1)];
  T[j] = kth == KEY_EMPTY ? 3.0e38f : stage_threshold16(key_dist(kth), qstat + j * 4, scal, metric, slack, approx);
}


// Start of a seeded call in one launch (three memset nodes cost a single-query call ~50 us): thresholds = +inf, every query's list
// length = the seed count (the dense seed pass writes slot = row), overflow flag and candidate total = 0, group counters = 0.
__global__ void seed_prologue_kernel(u64* T2, int64_t n2, u64 v, u32* cnt, int64_t nq, u32 cntv, u32* gsync) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) T2[i] = v;
  if (i < nq) cnt[i] = cntv;
  if (i < 8) cnt[nq + i] = 0;
  if (gsync && i < 256) gsync[i] = 0;
}

// seed sample: the first S/2 rows of the table plus S/2 rows spread evenly over the rest, copied next to each other (a
// positional filter - "only the newest rows" or "only the oldest" - leaves at least half of the seeds' share visible)
__global__ __launch_bounds__(256) void seed_sample_kernel(const _Float16* xh, const u32* base_s, const u32* base, unsigned long long stride,
                                                          u32 head, int d_pad, _Float16* sxh, u32* sbase, u32* sbase_u) {   // (base columns: raw words - fp32 or int32)
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
    const char* e = tune_env("EPS_MFMA_MANTISSA");
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

static float host_ord2f(u32 o) {
  const u32 u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// the 8-bit mirror: centre and grid from the table's values on the first build, kept when rows are appended
static int32_t ensure_mirror8(Index& ix) {
  if (!ix.mirror_) ix.mirror_ = new HalfMirror();
  HalfMirror& m = *ix.mirror_;
  const int64_t n = ix.n_rows_;
  if (m.version8 == ix.rows_version_ && m.n8 == n) return EPS_OK;
  const bool extend = m.version8 == ix.rows_version_ && m.n8 > 0 && m.n8 < n && m.i8_ok;
  const int64_t n_pad = (n + ROWPAD - 1) / ROWPAD * ROWPAD;
  const int d_pad8 = std::max(512, (int)((ix.dim_ + 255) / 256 * 256));   // K-steps of 128 bytes, in pairs, at least four
  const int dim = (int)ix.dim_;
  hipStream_t s = ix.stream_;
  const size_t keep_rows = extend ? (size_t)m.n8 : 0;
  if (!grow_keep(m.x8, (size_t)n_pad * d_pad8, keep_rows * d_pad8, s) || !grow_keep(m.acc0, (size_t)n_pad * 4, keep_rows * 4, s) || !m.scal8.reserve(64) ||
      !m.mu8.reserve((size_t)d_pad8 * 4) || !grow_keep(m.erow, (size_t)n_pad * 4, keep_rows * 4, s) || !grow_keep(m.hrow, (size_t)n_pad * 4, keep_rows * 4, s) ||
      !m.acc0b.reserve((size_t)n_pad * 4) || !m.scal8f.reserve(64) || !m.qmax.reserve(16) || !m.hist.reserve((4096 + 8) * 4)) {
    (void)hipGetLastError();
    if (!extend) {   // (nothing half-built stays behind: the callers for whom the mirror is optional carry on without it)
      m.x8.release();
      m.acc0.release();
      m.erow.release();
      m.hrow.release();
      m.acc0b.release();
      m.version8 = -1;
    }
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the 8-bit mirror");
  }
  hipError_t er = hipSuccess;
  if (!extend) {
    // centre: column means of up to 65 536 rows spread evenly over the table (any centre is valid, see above), summed in a fixed order
    const int64_t sample = std::min<int64_t>(n, 65536);
    const int64_t stride = std::max<int64_t>(1, n / sample);
    const int64_t per_seg = (sample + CENTRE_SEG - 1) / CENTRE_SEG;
    const int64_t sampled = std::min<int64_t>((n + stride - 1) / stride, per_seg * CENTRE_SEG);
    // r6: the frame.  0 = identity, 1 = rotated, otherwise the library's choice (both measured, below)
    const char* rot_e = tune_env("EPS_MIRROR_ROTATE");
    const int rot_mode = rot_e ? atoi(rot_e) : -1;
    const int W = (dim + 255) / 256 * 256;   // the rotation's width (the mirror pads rows to at least 512 bytes: those columns stay zero)
    m.rot_w8 = W;
    if (rot_mode != 0) {   // R's permutation and signs: a fixed sequence (splitmix64), the same for every table of this width
      std::vector<int32_t> sp((size_t)d_pad8, 0);
      for (int i = 0; i < W; ++i) sp[(size_t)i] = i;
      uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)W;
      auto next = [&st]() {
        uint64_t z = (st += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
      };
      for (int i = W - 1; i > 0; --i) std::swap(sp[(size_t)i], sp[(size_t)(next() % (uint64_t)(i + 1))]);
      for (int i = 0; i < W; ++i)
        if (next() & 1ull) sp[(size_t)i] |= (int32_t)0x80000000u;
      if (!m.sp8.reserve((size_t)d_pad8 * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the 8-bit mirror");
      er = hipMemcpyAsync(m.sp8.p, sp.data(), (size_t)d_pad8 * 4, hipMemcpyHostToDevice, s);
      if (er == hipSuccess) er = hipStreamSynchronize(s);
      if (er != hipSuccess) return ix.hip_fail(er, "8-bit mirror: rotation table");
    }
    // One frame's grid: centre into `mu_out`, the clipped value range, whether the table fits one grid at all.  `rot`: measured on the
    // rotated images of the sample rows (all d_pad8 columns carry values), the value range over the whole table through the transform.
    struct Grid { bool ok = false; float z0 = 0.f, half = 127.f, step = 0.f; };
    auto measure = [&](bool rot, DevBuf& mu_out, Grid* g) -> int32_t {
      DevBuf part, rsample;
      const float* srows = ix.d_rows_;
      int64_t sn = n, sstride = stride;
      int sdim = dim;
      if (rot) {
        if (!rsample.reserve((size_t)sampled * W * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the 8-bit mirror");
        hipLaunchKernelGGL(rot_sample_kernel, dim3((unsigned)std::min<int64_t>((sampled + 3) / 4, 8192)), dim3(256), 0, s, ix.d_rows_, n, dim, stride, sampled, W,
                           m.sp8.as<int>(), rsample.as<float>());
        srows = rsample.as<float>();
        sn = sampled;
        sstride = 1;
        sdim = W;
      }
      hipError_t e2 = hipMemsetAsync(m.scal8.p, 0, 32, s);
      if (e2 == hipSuccess) e2 = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m.scal8.as<u32>() + 6), (int)0xFFFFFFFFu, 1, s);
      if (e2 != hipSuccess) return ix.hip_fail(e2, "memset");
      if (!part.reserve((size_t)CENTRE_SEG * sdim * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the 8-bit mirror");
      hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((sdim + 63) / 64), CENTRE_SEG), dim3(256), 0, s, srows, sn, sdim, sstride, per_seg, part.as<float>(),
                         (const float*)nullptr, 0.f, 0.f);
      hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)((d_pad8 + 255) / 256)), dim3(256), 0, s, part.as<float>(), sdim, d_pad8, 1.f / (float)sampled, mu_out.as<float>());
      const unsigned mm_grid = (unsigned)std::min<int64_t>((n + 3) / 4, 8192);
      if (rot) hipLaunchKernelGGL(minmax_kernel<true>, dim3(mm_grid), dim3(256), 0, s, ix.d_rows_, n, dim, mu_out.as<float>(), m.scal8.as<u32>(), m.sp8.as<int>(), W);
      else hipLaunchKernelGGL(minmax_kernel<false>, dim3(mm_grid), dim3(256), 0, s, ix.d_rows_, n, dim, mu_out.as<float>(), m.scal8.as<u32>(), (const int*)nullptr, d_pad8);
      e2 = hipMemcpyAsync(m.h_scal8, m.scal8.p, 32, hipMemcpyDeviceToHost, s);
      if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);   // (also keeps `part` alive until its readers are done)
      if (e2 != hipSuccess) return ix.hip_fail(e2, "8-bit mirror: value range");
      u32 omin, omax;
      std::memcpy(&omin, &m.h_scal8[6], 4);
      std::memcpy(&omax, &m.h_scal8[7], 4);
      const float lo = host_ord2f(omin), hi = host_ord2f(omax);
      g->ok = m.h_scal8[3] == 0.f && omin <= omax && hi > lo && std::isfinite(lo) && std::isfinite(hi) && std::isfinite(hi - lo);
      float clo = lo, chi = hi;
      // (EPS_MIRROR_CLIP = e: cut 10^-e of the sample's values off each tail instead of 10^-7, and keep the cut in the rotated frame too - lab knob, r6)
      const char* clip_e = tune_env("EPS_MIRROR_CLIP");
      const bool clip_set = clip_e != nullptr;
      // r6, rotated frame: the cut is kept, at 10^-6 per tail.  Near-Gaussian columns put the whole table's range at ~6.3 sigma (10M x 768 values)
      // while 10^-6 of them lie beyond 4.9 sigma: the step - and with it both terms of the margin - shrinks by a quarter, the 0.15 % of rows with
      // a clamped value carry their own residual (folded per batch: 40 us at 10M rows).  10M x 768 embedding-like rows, batch 1024: 958 instead of
      // 2186 re-ranked rows per query, 8.80 -> 7.64 ms per step (10^-7: 7.87, 10^-5: 8.08 - clamped residuals start to dominate;
      // profiles/r6_embedding_like_grid_cut.txt).  1M x 768, 1 / 3 / 8 / 16 queries per call (the one-pass search, which serves tables with
      // folded margins since r6): 0.232 / 0.281 / 0.299 / 0.356 ms against 0.286 / 0.380 / 0.497 / 0.464 with the whole range
      // (profiles/r6_rotated_frame_one_pass_1M.txt).
      const double clip_frac = clip_set ? std::pow(10.0, -std::max(1.0, std::min(9.0, atof(clip_e)))) : (rot ? 1e-6 : 1e-7);
      auto clip_range = [&]() -> int32_t {
      // clip both tails of the SAMPLE's x - mean at max(2, 1e-7 x values) values (centre_hist_kernel); where the cut removes most of the
      // range - an outlier thousands of grid widths away leaves the bulk in ONE bin - the histogram is taken again inside the cut (values
      // outside fall into the edge bins), up to three times
      for (int round = 0; g->ok && round < 3; ++round) {
        const float binw = (chi - clo) / 4096.f;
        if (!(binw > 0.f) || !std::isfinite(1.f / binw)) break;
        std::vector<u32> hh(4096);
        hipError_t e3 = hipMemsetAsync(m.hist.p, 0, (4096 + 8) * 4, s);
        if (e3 != hipSuccess) return ix.hip_fail(e3, "memset");
        hipLaunchKernelGGL(centre_hist_kernel, dim3((unsigned)std::min<int64_t>(sampled, 4096)), dim3(256), 0, s, srows, sn, sdim, sstride, sampled, mu_out.as<float>(),
                           clo, 1.f / binw, m.hist.as<u32>());
        e3 = hipMemcpyAsync(hh.data(), m.hist.p, 4096 * 4, hipMemcpyDeviceToHost, s);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(s);
        if (e3 != hipSuccess) return ix.hip_fail(e3, "8-bit mirror: value histogram");
        const unsigned long long tol = std::max<unsigned long long>(2ull, (unsigned long long)(clip_frac * (double)sampled * (double)sdim));
        unsigned long long cum = 0;
        int blo = 0, bhi = 4095;
        for (blo = 0; blo < 4096; ++blo) {
          cum += hh[(size_t)blo];
          if (cum > tol) break;
        }
        cum = 0;
        for (bhi = 4095; bhi >= 0; --bhi) {
          cum += hh[(size_t)bhi];
          if (cum > tol) break;
        }
        const float a = clo + (float)blo * binw, b = clo + (float)(bhi + 1) * binw;
        if (!(blo < 4096 && bhi >= 0 && b > a)) break;
        const bool cut_most = (b - a) < 0.25f * (chi - clo);
        clo = a;
        chi = b;
        if (!cut_most) break;
      }
      return EPS_OK;
      };
      int32_t rc = clip_range();
      if (rc != EPS_OK) return rc;
      if (g->ok && (clo > lo || chi < hi)) {
        // something was cut: the column means it polluted are estimated again from values clamped into the cut (same fixed summation
        // order), and the range once more around the new means (the first range, widened by the largest move of a mean, bounds it)
        DevBuf mean1;
        std::vector<float> h1((size_t)sdim), h2((size_t)sdim);
        if (!mean1.reserve((size_t)d_pad8 * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the 8-bit mirror");
        hipError_t e3 = hipMemcpyAsync(mean1.p, mu_out.p, (size_t)d_pad8 * 4, hipMemcpyDeviceToDevice, s);
        if (e3 != hipSuccess) return ix.hip_fail(e3, "memcpy");
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((sdim + 63) / 64), CENTRE_SEG), dim3(256), 0, s, srows, sn, sdim, sstride, per_seg, part.as<float>(),
                           mean1.as<float>(), clo, chi);
        hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)((d_pad8 + 255) / 256)), dim3(256), 0, s, part.as<float>(), sdim, d_pad8, 1.f / (float)sampled, mu_out.as<float>());
        e3 = hipMemcpyAsync(h1.data(), mean1.p, (size_t)sdim * 4, hipMemcpyDeviceToHost, s);
        if (e3 == hipSuccess) e3 = hipMemcpyAsync(h2.data(), mu_out.p, (size_t)sdim * 4, hipMemcpyDeviceToHost, s);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(s);
        if (e3 != hipSuccess) return ix.hip_fail(e3, "8-bit mirror: column means");
        float delta = 0.f;
        for (int c = 0; c < sdim; ++c) delta = std::max(delta, std::fabs(h2[(size_t)c] - h1[(size_t)c]));
        clo = lo - delta;
        chi = hi + delta;
        rc = clip_range();
        if (rc != EPS_OK) return rc;
      }
      g->z0 = g->ok ? 0.5f * clo + 0.5f * chi : 0.f;
      g->half = g->ok ? std::max(chi - g->z0, g->z0 - clo) : 127.f;
      g->step = g->half / 127.f;
      if (g->ok && !(g->step > 0.f && std::isfinite(1.f / (g->step * g->step)))) g->ok = false;
      return EPS_OK;
    };
    // The choice: a row's residual norm is ~ step x sqrt(columns that carry values / 12) in either frame - the identity frame quantises
    // `dim` columns, the rotated one W = dim rounded up to 256 - so the frame with the smaller product gives the tighter margin.  The rotated frame must
    // win clearly (0.75): at equal margins the identity frame's query preparation is cheaper, and it is the frame every earlier round measured.
    Grid gi, gr;
    DevBuf mu_rot;
    m.step8_identity = m.step8_rotated = 0.f;
    if (rot_mode != 1) {
      const int32_t rc = measure(false, m.mu8, &gi);
      if (rc != EPS_OK) return rc;
      m.step8_identity = gi.ok ? gi.step : 0.f;
    }
    if (rot_mode != 0 && (rot_mode == 1 || gi.ok)) {
      if (!mu_rot.reserve((size_t)d_pad8 * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory for the 8-bit mirror");
      const int32_t rc = measure(true, mu_rot, &gr);
      if (rc != EPS_OK) return rc;
      m.step8_rotated = gr.ok ? gr.step : 0.f;
    }
    m.rot8 = rot_mode == 1 || (rot_mode != 0 && gi.ok && gr.ok &&
                               (double)gr.step * std::sqrt((double)W) < 0.75 * (double)gi.step * std::sqrt((double)dim));
    const Grid& g = m.rot8 ? gr : gi;
    if (m.rot8) {
      er = hipMemcpyAsync(m.mu8.p, mu_rot.p, (size_t)d_pad8 * 4, hipMemcpyDeviceToDevice, s);
      if (er == hipSuccess) er = hipStreamSynchronize(s);
      if (er != hipSuccess) return ix.hip_fail(er, "memcpy");
    }
    m.i8_ok = g.ok;
    er = hipMemsetAsync(m.scal8.p, 0, 32, s);
    if (er != hipSuccess) return ix.hip_fail(er, "memset");
    if (m.i8_ok) {
      er = hipMemsetAsync(m.hist.p, 0, (4096 + 8) * 4, s);   // ([4096]: the forced-row counter of the quantising pass)
      if (er != hipSuccess) return ix.hip_fail(er, "memset");
    }
    const float z0 = g.z0;
    m.step8 = g.step;
    if (m.i8_ok) {
      hipLaunchKernelGGL(mu_finish_kernel, dim3(1), dim3(64), 0, s, m.mu8.as<float>(), m.rot8 ? W : dim, z0, m.scal8.as<float>());
      er = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m.scal8.as<u32>() + 7), 0x7F800000, 1, s);   // min residual norm: +inf
      if (er != hipSuccess) return ix.hip_fail(er, "memset");
    }
  }
  if (m.i8_ok) {
    const float u = (ix.metric_ == 0 ? 2.f : 1.f) * m.step8 * m.step8;
    const int64_t row0 = extend ? m.n8 : 0;
    const dim3 qgrid((unsigned)std::min<int64_t>((n_pad - row0 + 3) / 4, 8192));
    if (m.rot8)
      hipLaunchKernelGGL(quant_mirror_kernel<true>, qgrid, dim3(256), 0, s, ix.d_rows_, row0, n, n_pad, dim, d_pad8, m.mu8.as<float>(), m.step8, 1.f / m.step8, 1.f / u,
                         ix.metric_, m.x8.as<signed char>(), m.acc0.as<int>(), m.scal8.as<float>(), m.erow.as<float>(), m.hrow.as<float>(), m.hist.as<u32>() + 4096,
                         m.sp8.as<int>(), m.rot_w8);
    else
      hipLaunchKernelGGL(quant_mirror_kernel<false>, qgrid, dim3(256), 0, s, ix.d_rows_, row0, n, n_pad, dim, d_pad8, m.mu8.as<float>(), m.step8, 1.f / m.step8, 1.f / u,
                         ix.metric_, m.x8.as<signed char>(), m.acc0.as<int>(), m.scal8.as<float>(), m.erow.as<float>(), m.hrow.as<float>(), m.hist.as<u32>() + 4096,
                         (const int*)nullptr, 0);
    hipLaunchKernelGGL(scal_finish_kernel, dim3(1), dim3(64), 0, s, m.scal8.as<float>(), m.scal8f.as<float>());
    u32 forced = 0;
    er = hipMemcpyAsync(m.h_scal8, m.scal8.p, 32, hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipMemcpyAsync(&forced, m.hist.as<u32>() + 4096, 4, hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (er != hipSuccess) return ix.hip_fail(er, "8-bit mirror build");
    m.forced_rows8 = (int64_t)forced;   // (accumulated over extensions: the counter is only zeroed with the histogram)
    // a non-finite value, or a table most of whose row constants leave the accumulator's range (IP / COSINE far from the origin: |mu . x'| / step^2):
    // the fp16 engine serves this table
    if (m.h_scal8[3] != 0.f || (double)forced > 0.01 * (double)n) m.i8_ok = false;
    // Per-row margins are folded per batch only where rows DIFFER: a forced row, or a residual norm beyond 1.5 x the smallest (a clamped
    // value somewhere).  On homogeneous tables (every row a plain rounding residual: within a few per cent of each other) the table-wide
    // margin in the thresholds is as tight, keeps every query's own norms, and costs no pass over the rows (10M rows: 40 us per batch).
    m.fold8 = forced > 0 || m.h_scal8[0] > 1.5f * m.h_scal8[7] || (tune_int("EPS_MFMA_FOLD", 0) != 0);
    if (tune_int("EPS_MFMA_FOLD", 1) == 0 && forced == 0) m.fold8 = false;
    m.extended_rows8 += extend ? n - row0 : 0;
  }
  if (!m.i8_ok) {   // nothing of it is used: give the memory back
    m.x8.release();
    m.acc0.release();
    m.erow.release();
    m.hrow.release();
    m.acc0b.release();
  }
  if (!extend) {
    m.i8_overflows = 0;
    m.i8_trusted = false;
    m.epoch8 += 1;
  }
  m.n8 = n;
  m.n_pad8 = n_pad;
  m.d_pad8 = d_pad8;
  m.version8 = ix.rows_version_;
  return EPS_OK;
}

// The 8-bit mirror for kernels outside this file (the traversal's lower-bound prefilter): built or extended on demand;
// v->x8 stays null when the table cannot be put on one grid (non-finite values, constants beyond int32).
int32_t quant8_view(Index& ix, Quant8View* v) {
  *v = Quant8View();
  const int32_t rc = ensure_mirror8(ix);
  if (rc != EPS_OK) return rc;
  const HalfMirror& m = *ix.mirror_;
  if (!m.i8_ok) return EPS_OK;
  v->x8 = m.x8.as<signed char>();
  // (what the traversal kernels read: the start values with the CURRENT batch's per-row margins folded in - quant8_queries below folds
  // them after every query preparation - and the maxima whose two margin entries are zero: their thresholds carry no margin)
  v->acc0 = m.fold8 ? m.acc0b.as<int>() : m.acc0.as<int>();
  v->scal8 = m.fold8 ? m.scal8f.as<float>() : m.scal8.as<float>();
  v->mu = m.mu8.as<float>();
  v->d_pad8 = m.d_pad8;
  v->cols8 = m.rot8 ? m.rot_w8 : (int)ix.dim_;
  v->step = m.step8;
  v->u = (ix.metric_ == 0 ? 2.f : 1.f) * m.step8 * m.step8;
  v->epoch8 = m.epoch8;
  v->per_batch = m.fold8;
  return EPS_OK;
}

// nq queries on the mirror's grid: q8 [nq][d_pad8], qstat [nq][4] (device buffers of the caller)
void quant8_queries(Index& ix, const Quant8View& v, const float* dq, int64_t nq, signed char* q8, float* qstat) {
  HalfMirror& m = *ix.mirror_;
  Prep8Extra px;
  if (m.fold8) {
    px.qmax = m.qmax.as<u32>();
    (void)hipMemsetAsync(m.qmax.p, 0, 8, ix.stream_);
  }
  launch_query_prep8(dim3((unsigned)((nq + 3) / 4)), ix.stream_, m.rot8 ? m.sp8.as<int>() : nullptr, m.rot_w8, dq, nq, nq, (int)ix.dim_, v.d_pad8, v.mu, v.step, ix.metric_, q8,
                     qstat, px);
  if (m.fold8)
  hipLaunchKernelGGL(fold8_kernel, dim3((unsigned)std::min<int64_t>((m.n_pad8 + 255) / 256, 8192)), dim3(256), 0, ix.stream_, m.acc0.as<int>(), m.erow.as<float>(),
                     m.hrow.as<float>(), m.n8, m.n_pad8, m.qmax.as<u32>(), ix.metric_ == 0 ? 2.f : 1.f, 1.f / v.u, m.acc0b.as<int>());
}

bool flat_mfma_profitable(const Index& ix, int64_t nq, int k) {
  // FLAT_AUTO: both engines return the same bits, so this is purely a cost decision.  The stream scan reads the fp32 rows once
  // per 4 queries; the filter reads its mirror once per <= 2048 queries - the 8-bit mirror is a quarter of the rows' bytes, the
  // fp16 mirror half - and pays ~20 small launches and one host sync per call (bench.py configs c2, 1M x 768, one query: stream
  // 0.72 ms, 8-bit filter 0.48 ms end to end; scripts/bench_midbatch.py, 10M x 768: stream 5.1 / 11.3 / 43.7 ms vs fp16 filter
  // 3.5 / 3.6 / 3.8 ms at 1 / 8 / 32 queries).
  if (ix.n_rows_ < 65536 || k > 128) return false;
  const HalfMirror* m = ix.mirror_;
  const bool have16 = m && m->version == ix.rows_version_;
  const bool known8 = m && m->version8 == ix.rows_version_;
  const bool have8 = known8 && m->i8_ok;
  const bool can16 = !have16 || m->fp16_range_ok;
  if (known8 && !have8 && !can16) return false;                 // neither mirror can serve this table
  // up to 16 queries, k <= 64, rows of <= 1024 bytes: the one-pass search (stream8_kernel.hpp) - one pass over d_pad8 + 4 bytes per row
  const bool one_pass_shape = nq <= S8_MAX_Q && k <= S8_MAX_K && ix.dim_ <= 1024 && !(tune_int("EPS_FLAT_ONE_PASS", 1) == 0);
  if (nq < 8 && !have8 && !have16) {
    // single-query traffic alone does not get a mirror (n x d bytes of HBM + a pass over the table to build it) at once: r4, after 16 such
    // calls on the same rows it does, where the one-pass search can use it (0.20 ms instead of 0.62 ms per call at 1M x 768)
    if (ix.small_calls_version_ != ix.rows_version_) {
      ix.small_calls_version_ = ix.rows_version_;
      ix.small_calls_ = 0;
    }
    if (!one_pass_shape || known8 || ++ix.small_calls_ <= 16) return false;
  }
  const bool use8 = have8 || !known8;                            // (an 8-bit mirror would be built first)
  const double rows = (double)ix.n_rows_, d = (double)ix.dim_;
  const double op_bytes = use8 ? std::max(512.0, std::ceil(d / 256.0) * 256.0) : std::ceil(d / 128.0) * 256.0;   // operand bytes per row
  const double rate = use8 ? 2.0e15 : 1.2e15;                    // matrix rate the filter kernel reaches
  const double dp = use8 ? op_bytes : op_bytes / 2.0;
  const double stream_s = std::ceil((double)nq / 4.0) * rows * d * 4.0 / 6.0e12 + 0.2e-3;
  const double filter_s = (use8 && one_pass_shape)   // (r5: filter programs take the one-pass form too - behind one mask launch; r6: so do tables with folded margins)
                              ? 0.07e-3 + rows * (std::ceil(d / 256.0) * 256.0 + 4.0) / 5.7e12
                              : 0.35e-3 + std::max(rows * op_bytes / 5.0e12, 2.0 * 128.0 * std::ceil((double)nq / 128.0) * rows * dp / rate);
  return filter_s < stream_s;
}

// A handful of queries (<= 16, k <= 64) in ONE pass over the 8-bit mirror: stream8_kernel.hpp.  *done = false: not applicable to this call,
// or a list overflowed - the staged chain below answers it (results are bit-identical either way: both end in the same exact re-rank).
static int32_t flat_stream8_slice(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool* done) {
  *done = false;
  HalfMirror& m = *ix.mirror_;
  const int64_t n = ix.scan_limit_ >= 0 ? std::min(ix.scan_limit_, ix.n_rows_) : ix.n_rows_;
  const int pieces = m.d_pad8 / 256;
  if (tune_int("EPS_FLAT_ONE_PASS", 1) == 0) return EPS_OK;
  const int max_q = std::min(S8_MAX_Q, std::max(1, tune_int("EPS_S8_MAX_Q", S8_MAX_Q)));   // (A/B switch: 4 = the r4 form, 5+ queries on the staged chain)
  const int max_k = std::min(S8_MAX_K, std::max(1, tune_int("EPS_S8_MAX_K", S8_MAX_K)));   // (A/B switch: 16 = the r4 range, larger k on the staged chain)
  if (nq < 1 || nq > max_q || k < 1 || k > max_k || n < 65536 || n > m.n8 || m.d_pad8 % 256 != 0 || pieces < 2 || pieces > 4) return EPS_OK;
  // r6: tables whose margins are folded per batch (m.fold8: rows with a clamped value) take the form too - behind the prep launch and the fold
  // launch (the margins depend on the call's queries), with margin-free thresholds and offers of `accumulator - 2 x the row's margin` (stream8_offer_value)
  const bool fold = m.fold8 && tune_int("EPS_S8_FOLD", 1) != 0;   // (A/B switch: 0 = such tables on the staged chain, as until r6)
  if (m.fold8 && !fold) return EPS_OK;
  // (17..32 queries, r6: the two-column-block pass beats the staged chain where a query leaves a few hundred candidates - 1M x 768 U[0,1): 17 / 24 / 32
  // queries 0.235 / 0.254 / 0.281 ms against 0.38 - and loses where it leaves several times that: the same table of embedding-like rows 0.433 /
  // 0.465 / 0.505 against 0.432 / 0.443 / 0.443 (profiles/r6_one_pass_17_to_32_queries.txt).  Tables with folded margins are the looser ones.)
  if (nq > 16 && m.fold8 && tune_int("EPS_S8_MAX_Q", 0) == 0) return EPS_OK;
  // (... and on LARGE tables: with two column blocks the pass is no longer purely HBM-bound - 24 MFMAs and 12 LDS operand reads per 16-row block - and its
  // time grows with the queries, while the chain's 128-query tiles stream the mirror at the HBM rate whatever nq: 10M x 768, 17 / 24 / 32 queries: 1.46 /
  // 1.54 / 1.73 ms against the chain's 1.53; 4M: 0.66 / 0.70 / 0.79 against 0.78; 2M: 0.38 / 0.40 / 0.45 against 0.51)
  if (nq > 16 && (double)(nq - 16) * (double)n > 64e6 && tune_int("EPS_S8_MAX_Q", 0) == 0) return EPS_OK;
  // (r6: ... and per kernel form - five or more queries share one list budget per query and overflow on tables where one query does not:
  // an 8-query batch must not talk the table out of the form for single-query traffic)
  const int kclass = (k <= 16 ? 0 : 1) + (nq <= 4 ? 0 : (nq <= 16 ? 2 : 4));   // (r6: 17..32 queries - two column blocks - are a form of their own)
  if (m.s8_declined_version[kclass] == ix.rows_version_) return EPS_OK;
  FilterSpec fs = ix.filter_spec();
  // a compiled filter program: evaluated once per row into a bitset (filter_mask_kernel) that pass and re-rank read as a deleted bitset.  Only
  // where the predicate does not depend on the candidate's distance (Index::search sends programs that read @distance outside a pre-filter
  // call to the stream engine before they get here; checked again, the staged chain evaluates per candidate)
  const bool masked = fs.prog != nullptr;
  if (masked && ix.prog_uses_dist_ && !ix.prefilter_call_) return EPS_OK;
  if (masked && tune_int("EPS_S8_FILTER_PROGRAMS", 1) == 0) return EPS_OK;   // (A/B switch: programs on the staged chain, as until r4)
  const bool filtered = fs.deleted || fs.column || masked;
  if (filtered && m.s8_filt_skip[kclass] > 0) {   // (a mask that starves the pass used to cost a wasted pass on EVERY call)
    --m.s8_filt_skip[kclass];
    return EPS_OK;
  }
  hipStream_t s = ix.stream_;
  const int cap = std::max(4096, 64 * k);
  if (!m.qstat.reserve((size_t)S8_MAX_Q * 16) || !m.q8.reserve((size_t)S8_MAX_Q * m.d_pad8) || !m.cand.reserve((size_t)nq * cap * 8) || !m.cnt.reserve((size_t)(S8_MAX_Q + 16) * 4) ||
      !m.s8g.reserve((size_t)S8_TABLE_WORDS * 4 + (size_t)S8_MAX_Q * S8_MAX_WAVES * 4) || !m.s8raw.reserve((size_t)nq * S8_MAX_WAVES * S8_WAVE_CAP * 8) ||
      (masked && !m.s8mask.reserve((size_t)(n + 7) / 8 + 16)))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  if (masked) {   // (every call: the caller's bitset and attribute rows are used in place and may have changed since the last one)
    fs.prog_use_dist = 0;
    hipLaunchKernelGGL(filter_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fs, n, m.s8mask.as<uint8_t>());
    fs = no_filter();
    fs.deleted = m.s8mask.as<uint8_t>();
  }
  // (cnt: nq + 8 <= 12 words in use here; DevBuf::reserve never hands out less than 256 bytes)
  u32* cnt = m.cnt.as<u32>();
  u32* overflow = cnt + nq;
  unsigned long long* total = reinterpret_cast<unsigned long long*>(cnt + nq + 2);
  if ((reinterpret_cast<uintptr_t>(total) & 7) != 0) total = reinterpret_cast<unsigned long long*>(cnt + nq + 3);
  const float u8 = (ix.metric_ == 0 ? 2.f : 1.f) * m.step8 * m.step8;
  const float rerank_slack = std::max(8e-6f, 2.f * (3.f * ((float)((ix.dim_ + 63) / 64 * 64) / 64.f + 6.f) + 6.f) * 5.9604645e-8f);
  const bool host_words = !(tune_int("EPS_S8_HOST_WORDS", 1) == 0) && m.s8_pub.get();   // (A/B switch)
  const bool two_launches = host_words && !(tune_int("EPS_S8_TWO_LAUNCHES", 1) == 0) && !tune_env("EPS_DEBUG");  // (A/B switch; the debug log reads the table after the call)
  // (3-4 queries keep the prep launch: next to four queries' slices and two chunks in flight the in-kernel form does not fit 256 registers)
  // (k = 17..64: 128 slots per query in the same table - a layout of its own, so such a call always starts with the prep launch, which empties
  // the slots it uses, and never leaves the "clean" state behind)
  const int slots = k <= 16 ? S8_SLOTS : S8_SLOTS_WIDE;
  const bool mfma_form = nq > 4 || (slots != S8_SLOTS && nq > 2);   // stream8m_kernel (16 query columns: the prep launch lays down 16 rows, zeros beyond nq)
  // (a table in the rotated frame keeps the prep launch: the transform is its work, not the pass's)
  const bool clean = two_launches && nq <= 2 && slots == S8_SLOTS && m.s8_clean_cnt == m.cnt.p && m.s8_clean_g == m.s8g.p && !m.rot8 && !fold;
  m.s8_clean_cnt = m.s8_clean_g = nullptr;   // (set again when this call has come back)
  if (!clean) {
    Prep8Extra px;
    px.cnt = cnt;     // candidate counts, overflow and total counters = 0
    px.cntv = 0;
    px.s8g = m.s8g.as<int>();
    px.s8_slots = slots;
    if (fold) {
      px.qmax = m.qmax.as<u32>();
      if (hipMemsetAsync(m.qmax.p, 0, 8, s) != hipSuccess) return ix.hip_fail(hipGetLastError(), "memset");
    }
    const int prep_rows = mfma_form ? (nq > 16 ? 32 : 16) : 4;   // (the matrix form reads 16 / 32 query rows: zeros beyond nq)
    launch_query_prep8(dim3((unsigned)(prep_rows / 4)), s, m.rot8 ? m.sp8.as<int>() : nullptr, m.rot_w8, dq, nq, (int64_t)prep_rows, (int)ix.dim_, m.d_pad8, m.mu8.as<float>(),
                       m.step8, ix.metric_, m.q8.as<signed char>(), m.qstat.as<float>(), px);
    if (fold)   // (16 bytes per row: 5 us at 1M rows, 40 us at 10M - where the pass itself takes 1.4 ms)
      hipLaunchKernelGGL(fold8_kernel, dim3((unsigned)std::min<int64_t>((m.n_pad8 + 255) / 256, 8192)), dim3(256), 0, s, m.acc0.as<int>(), m.erow.as<float>(),
                         m.hrow.as<float>(), m.n8, m.n_pad8, m.qmax.as<u32>(), ix.metric_ == 0 ? 2.f : 1.f, 1.f / u8, m.acc0b.as<int>());
  }
  ix.stats_.i8_folded = fold ? 1 : 0;
  Stream8Args a;
  a.x8 = m.x8.as<signed char>();
  a.acc0 = fold ? m.acc0b.as<int>() : m.acc0.as<int>();
  a.acc0_raw = fold ? m.acc0.as<int>() : nullptr;
  a.n = n;
  a.d_pad8 = m.d_pad8;
  a.q8 = m.q8.as<signed char>();
  a.qstat = m.qstat.as<float>();
  a.scal = fold ? m.scal8f.as<float>() : m.scal8.as<float>();   // (folded: the margin entries are zero - the margins are in the start values)
  a.nq = (int)nq;
  a.k = k;
  a.metric = ix.metric_;
  a.u = u8;
  a.slack = rerank_slack;
  a.G = m.s8g.as<int>();
  a.slots = slots;
  a.raw_cnt = m.s8g.as<u32>() + S8_TABLE_WORDS;
  a.raw = m.s8raw.as<u64>();
  a.f = fs;
  a.qf32 = dq;
  a.mu = m.mu8.as<float>();
  a.qstat_out = m.qstat.as<float>();
  a.dim = (int)ix.dim_;
  a.step = m.step8;
  a.inv_step = 1.f / m.step8;
#ifdef EPS_LAB   // (kernel ablations make answers wrong on purpose: lab builds only)
  a.ablate = tune_int("EPS_S8_ABLATE", 0);
#else
  a.ablate = 0;
#endif
  if (!m.s8_cus) {   // (once per mirror: the query costs more than the search)
    hipDeviceProp_t prop;
    m.s8_cus = hipGetDeviceProperties(&prop, ix.device_) == hipSuccess ? std::max(8, prop.multiProcessorCount) : 256;
  }
  const int cus = m.s8_cus;
  const int wg_per_cu = std::max(1, tune_int("EPS_S8_WG_PER_CU", 2));
  const dim3 grid((unsigned)std::min<int64_t>(std::min<int64_t>((int64_t)cus * wg_per_cu, S8_MAX_WAVES / 4), (n + 63) / 64)), block(256);
  a.waves = (int)grid.x * 4;
  // (no event pair around the pass by default: a record between two dependent launches costs this chain 5-10 us each; kernel_ms covers
  // the call.  EPS_ONE_PASS_TIMED=1 - bench.py's roofline leg - records the pair: main_kernel_ms = the pass)
  const bool timed = tune_int("EPS_ONE_PASS_TIMED", 0) != 0;
  if (timed) (void)hipEventRecord(ix.evk0_, s);
#define EPS_S8_LAUNCH_(P_, PREP_)                                                                    \
  do {                                                                                               \
    if (nq == 1) hipLaunchKernelGGL((stream8_kernel<P_, 1, PREP_>), grid, block, 0, s, a);           \
    else if (nq == 2) hipLaunchKernelGGL((stream8_kernel<P_, 2, PREP_>), grid, block, 0, s, a);      \
    else hipLaunchKernelGGL((stream8_kernel<P_, 4, false>), grid, block, 0, s, a);                   \
  } while (0)
#define EPS_S8_LAUNCH(P_)                    \
  do {                                       \
    if (clean) EPS_S8_LAUNCH_(P_, true);     \
    else EPS_S8_LAUNCH_(P_, false);          \
  } while (0)
  if (slots != S8_SLOTS && nq <= 2) {   // k = 17..64, 1-2 queries: the v_dot4 pass with two slots per lane
#define EPS_S8_WIDE(P_)                                                                                   \
  do {                                                                                                    \
    if (nq == 1) hipLaunchKernelGGL((stream8_kernel<P_, 1, false, true>), grid, block, 0, s, a);          \
    else hipLaunchKernelGGL((stream8_kernel<P_, 2, false, true>), grid, block, 0, s, a);                  \
  } while (0)
    if (pieces == 2) EPS_S8_WIDE(2);
    else if (pieces == 3) EPS_S8_WIDE(3);
    else EPS_S8_WIDE(4);
#undef EPS_S8_WIDE
  } else if (mfma_form && nq > 16) {   // r6, 17..32 queries: two 16-query column blocks per row block
    if (pieces == 2) hipLaunchKernelGGL((stream8m_kernel<2, 2>), grid, block, 0, s, a);
    else if (pieces == 3) hipLaunchKernelGGL((stream8m_kernel<3, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((stream8m_kernel<4, 2>), grid, block, 0, s, a);
  } else if (mfma_form) {   // 5..16 queries (and 3-4 with k > 16): the same pass on the matrix cores
    if (pieces == 2) hipLaunchKernelGGL((stream8m_kernel<2, 1>), grid, block, 0, s, a);
    else if (pieces == 3) hipLaunchKernelGGL((stream8m_kernel<3, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((stream8m_kernel<4, 1>), grid, block, 0, s, a);
  } else if (pieces == 2) EPS_S8_LAUNCH(2);
  else if (pieces == 3) EPS_S8_LAUNCH(3);
  else EPS_S8_LAUNCH(4);
#undef EPS_S8_LAUNCH
#undef EPS_S8_LAUNCH_
  if (timed) (void)hipEventRecord(ix.evk1_, s);
  RerankArgs ra;
  ra.rows = ix.d_rows_;
  ra.dim = (int)ix.dim_;
  ra.metric = ix.metric_;
  ra.queries = dq;
  ra.nq = nq;
  ra.k = k;
  ra.f = fs;
  ra.cand = m.cand.as<u32>();
  ra.cand_count = cnt;
  ra.cap = cap;
  ra.run_keys = run_keys;
  ra.fuse = 1;                 // (counts only: overflow / total)
  ra.overflow = overflow;
  ra.total = total;
  ra.T_next = nullptr;
  ra.qstat = m.qstat.as<float>();
  ra.scal = a.scal;
  ra.bits = 8;
  ra.u = u8;
  ra.slack = rerank_slack;
  ra.gsync = nullptr;
  ra.s8_G = a.G;                 // (the launch selects its candidates from the pass's lists first)
  ra.s8_slots = slots;
  ra.s8_fast = !(tune_int("EPS_S8_RERANK", 1) == 0);   // (A/B switch: 0 = rerank_kernel with the selection prologue, as until r5)
  ra.s8_counts = a.raw_cnt;
  ra.s8_lists = a.raw;
  ra.s8_waves = a.waves;
  ra.s8_cand = m.cand.as<u32>();
  if (host_words) {
    ra.pub = m.s8_pub.p;
    ra.pub_ticket = cnt + nq + 6;   // (zeroed by the prep launch with the other counters: cnt[nq .. nq + 8))
    ra.s8_reset = two_launches && slots == S8_SLOTS ? 1 : 0;
  }
  const bool fin_here = ix.pre_sync_ && nq == ix.pre_sync_nq_ && ix.fin_ids_ != nullptr;
  if (fin_here) {
    ra.fin_ids = ix.fin_ids_;
    ra.fin_dist = ix.fin_dist_;
    ra.fin_counts = ix.fin_cnt_;
    ra.fin_base = ix.id_base_;
    ra.fin_stride = ix.id_stride_;
  }
  launch_rerank(ra, s);
  if (fin_here) (void)hipEventRecord(ix.ev1_, s);
  hipError_t er = hipGetLastError();
  if (er != hipSuccess) return ix.hip_fail(er, "one-pass flat search launch");
  struct {
    u32 overflow, pad;
    unsigned long long total;
  } h = {0, 0, 0};
  if (!fin_here && ix.pre_sync_ && nq == ix.pre_sync_nq_) ix.pre_sync_();
  // (both counters in ONE small copy: every copy is a trip through the DMA queue at the end of a 0.2 ms call)
  u32 hraw[6] = {0, 0, 0, 0, 0, 0};
  const size_t span = (size_t)(reinterpret_cast<const char*>(total) + 8 - reinterpret_cast<const char*>(overflow));
  if (host_words) {
    er = hipStreamSynchronize(s);
    if (er != hipSuccess) return ix.hip_fail(er, "one-pass flat search");
    const volatile u32* hp = m.s8_pub.p;
    h.overflow = hp[0];
    h.total = (unsigned long long)hp[2] | ((unsigned long long)hp[3] << 32);
    if (two_launches && slots == S8_SLOTS) {   // (the launch left slots and counters as the next call needs them)
      m.s8_clean_cnt = m.cnt.p;
      m.s8_clean_g = m.s8g.p;
    }
  } else {
    er = hipMemcpyAsync(hraw, overflow, span, hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (er != hipSuccess) return ix.hip_fail(er, "one-pass flat search");
    h.overflow = hraw[0];
    memcpy(&h.total, reinterpret_cast<const char*>(hraw) + (span - 8), 8);
  }
  if (tune_env("EPS_DEBUG") || (h.overflow && tune_env("EPS_DEBUG_ONE_PASS_OVERFLOW"))) {
    std::vector<u32> hc((size_t)S8_TABLE_WORDS + (size_t)S8_MAX_Q * S8_MAX_WAVES);
    (void)hipMemcpy(hc.data(), m.s8g.p, hc.size() * 4, hipMemcpyDeviceToHost);
    for (int64_t q = 0; q < nq; ++q) {
      unsigned long long raw = 0;
      u32 most = 0;
      int filled = 0;
      for (int w = 0; w < a.waves; ++w) {
        raw += hc[S8_TABLE_WORDS + q * a.waves + w];
        most = std::max(most, hc[S8_TABLE_WORDS + q * a.waves + w]);
      }
      for (int i = 0; i < slots; ++i) filled += (int)hc[(q * slots + i) * S8_SLOT_STRIDE] != S8_EMPTY;
      fprintf(stderr, "[eps one pass] query %lld: %llu raw candidates (at most %u in one wavefront's list of %d), %d of the slots filled, re-ranked (all queries) %llu, overflow %u\n",
              (long long)q, raw, most, S8_WAVE_CAP, filled, h.total, h.overflow);
    }
  }
  if (h.overflow) {   // (too loose a bound for this table, or a filter that leaves fewer than k rows visible: the staged chain answers)
    ix.result_finalized_ = false;
    if (filtered) {
      if (++m.s8_filt_overflows[kclass] >= 2) {
        m.s8_filt_overflows[kclass] = 0;
        m.s8_filt_skip[kclass] = 32;
      }
    } else if (++m.s8_overflows[kclass] >= 2) {
      m.s8_declined_version[kclass] = ix.rows_version_;
    }
    return EPS_OK;
  }
  if (filtered) m.s8_filt_overflows[kclass] = 0; else m.s8_overflows[kclass] = 0;
  if (fin_here) ix.result_finalized_ = true;
  ix.stats_.rerank_rows += (int64_t)h.total;
  ix.stats_.dist_evals += nq * n;
  ix.stats_.main_kernel_launches = timed ? 1 : 0;   // (0: not timed on its own)
  ix.stats_.main_kernel_rows = n;
  ix.stats_.main_kernel_queries = nq;
  ix.stats_.main_kernel_bits = 8;
  ix.stats_.one_pass = 1;
  ix.stats_.i8_rotated = m.rot8 ? 1 : 0;
  *done = true;
  return EPS_OK;
}

int32_t flat_mfma_search_slice(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx, int cap_scale, int bits, bool auto_bits) {
  // operand width of the filter pass: 8 = int8 mirror (when the table fits its grid), 16 = fp16 mirror
  bool i8 = bits == 8;
  int32_t rc = EPS_OK;
  if (i8) {
    rc = ensure_mirror8(ix);
    if (rc != EPS_OK) return rc;
    // a table whose 8-bit bound is too loose to filter (an outlier stretches the grid, rows far outside it) overflows on every
    // batch: after three in a row the library's own choice stops paying for the 8-bit pass first (an explicit EPS_FLAT_MFMA_I8
    // request still gets it; re-attaching rows re-arms it)
    if (!ix.mirror_->i8_ok || (auto_bits && ix.mirror_->i8_overflows >= 3)) i8 = false;
  }
  if (i8 && !approx && cap_scale == 1 && nq <= S8_MAX_Q) {
    bool done = false;
    rc = flat_stream8_slice(ix, dq, nq, k, run_keys, &done);
    if (rc != EPS_OK || done) return rc;
  }
  if (!i8) {
    rc = ensure_mirror(ix);
    if (rc != EPS_OK) return rc;
  }
  HalfMirror& m = *ix.mirror_;
  const int64_t n = ix.scan_limit_ >= 0 ? std::min(ix.scan_limit_, ix.n_rows_) : ix.n_rows_;
  if (!i8 && !m.fp16_range_ok) {
    // values beyond the fp16 range: the filter bound would be vacuous; the exact stream engine takes over
    return ix.flat_stream(dq, nq, k, 0, n, run_keys, false, -1, !approx);
  }
  hipStream_t s = ix.stream_;
  const int64_t b_pad = (nq + BN3 - 1) / BN3 * BN3;
  const int cap = std::max(4096, 64 * k) * cap_scale;   // candidate slots per query and stage
  const int d_pad_h = i8 ? m.d_pad8 / 2 : m.d_pad;      // row pitch of the operand in 2-byte units (what the kernels count in)
  const float u8 = (ix.metric_ == 0 ? 2.f : 1.f) * m.step8 * m.step8;   // key units per accumulator unit of the 8-bit pass
  if (!m.qstat.reserve((size_t)b_pad * 16) || !m.T.reserve((size_t)b_pad * 4) || !m.cand.reserve((size_t)nq * cap * 8) ||
      !m.cnt.reserve((size_t)(nq + 4) * 4 + 16) || !m.seedc.reserve((size_t)nq * k * 4) || !(i8 ? m.q8.reserve((size_t)b_pad * m.d_pad8) : m.qh.reserve((size_t)b_pad * m.d_pad * 2)))
    return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  // kernel choice: v5 / v7 want K-steps in pairs (d_pad % 128 == 0, >= 256); other shapes stay on v3
  const char* ver_s = tune_env("EPS_MFMA_KERNEL");   // 3 | 7 (A/B); v7 needs K-steps in pairs, other shapes stay on v3
  const int version_env = (ver_s && atoi(ver_s) == 3 && !i8) ? 3 : 7;
  const int version = (version_env == 7 && (d_pad_h % 128 != 0 || d_pad_h < 256)) ? 3 : version_env;   // (the 8-bit mirror is padded for v7)
  if (version >= 7 && !m.qf.reserve((size_t)b_pad * d_pad_h * 2)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  if (!m.gsync.reserve(1024)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
  // (what the staging below decides, needed here already: the 8-bit query preparation also lays down a seeded call's start state)
  const int64_t S0 = std::max<int64_t>(4096, (int64_t)(32 * k + ROWPAD - 1) / ROWPAD * ROWPAD);
  const bool seed_env = !(tune_int("EPS_MFMA_SEED", 1) == 0);
  const bool seeded = seed_env && n > 4 * S0;   // with a filter the seeds are the k best VISIBLE head rows
  const bool prologue = seeded && version >= 7;   // one launch resets everything a seeded call starts from
  const bool gsync_env = !(tune_int("EPS_MFMA_GROUPSYNC", 1) == 0);
  const bool prep_does_it_all = i8 && version >= 7;   // fragment-major copy + prologue inside query_prep8_kernel: two launches less per call
  hipError_t er_ = hipSuccess;
  const bool fold = i8 && !approx && m.fold8;   // exact mode on a table whose rows differ: per-row margins folded into the start values, thresholds without margin
  if (i8) {
    Prep8Extra px;
    if (fold) {
      px.qmax = m.qmax.as<u32>();
      er_ = hipMemsetAsync(m.qmax.p, 0, 8, s);
      if (er_ != hipSuccess) return ix.hip_fail(er_, "memset");
    }
    if (prep_does_it_all) {
      px.qf = m.qf.as<signed char>();
      if (prologue) {   // thresholds = 0x7F800000 pairs (+inf as fp32; as the 8-bit pass's int32 thresholds: never passes - the padding entries keep it)
        px.T2 = reinterpret_cast<u64*>(m.T.p);
        px.n2 = b_pad / 2;
        px.Tv = 0x7F8000007F800000ull;
        px.cnt = m.cnt.as<u32>();
        px.cntv = (u32)S0;
        px.gsync = gsync_env ? m.gsync.as<u32>() : nullptr;
      }
    }
    launch_query_prep8(dim3((unsigned)((b_pad + 3) / 4)), s, m.rot8 ? m.sp8.as<int>() : nullptr, m.rot_w8, dq, nq, b_pad, (int)ix.dim_, m.d_pad8, m.mu8.as<float>(), m.step8,
                       ix.metric_, m.q8.as<signed char>(), m.qstat.as<float>(), px);
    if (fold) ix.stats_.i8_folded = 1;
    ix.stats_.i8_rotated = m.rot8 ? 1 : 0;
    if (fold)
      hipLaunchKernelGGL(fold8_kernel, dim3((unsigned)std::min<int64_t>((m.n_pad8 + 255) / 256, 8192)), dim3(256), 0, s, m.acc0.as<int>(), m.erow.as<float>(),
                         m.hrow.as<float>(), m.n8, m.n_pad8, m.qmax.as<u32>(), ix.metric_ == 0 ? 2.f : 1.f, 1.f / u8, m.acc0b.as<int>());
  } else {
    hipLaunchKernelGGL(query_prep_kernel, dim3((unsigned)((b_pad + 3) / 4)), dim3(256), 0, s, dq, nq, b_pad, (int)ix.dim_,
                       m.d_pad, m.qh.as<_Float16>(), m.qstat.as<float>(), m.drop);
  }
  const _Float16* q_op = i8 ? reinterpret_cast<const _Float16*>(m.q8.p) : m.qh.as<_Float16>();   // the query operand, row-major
  if (version >= 7 && !prep_does_it_all)
    hipLaunchKernelGGL(pack_qf_kernel, dim3((unsigned)((b_pad / 32) * (d_pad_h / 16))), dim3(64), 0, s, q_op, m.qf.as<_Float16>(), b_pad, d_pad_h);

  // Staging.  Every MFMA stage needs a valid upper bound T of the final k-th best exact key; it tightens stage by stage.
  //  * seeded (exact mode, no deleted bitset / attribute filter): the head [0, S0) goes through the SAME MFMA kernel in
  //    its approx-key mode with T = +inf, the k best approximate keys are re-ranked exactly, and their k-th exact key is
  //    the first T (any k exact keys bound the k-th best).  The stages then start at row 0; the exact re-rank dedups rows
  //    it meets twice.  Stage sizes grow by the cube root of n / S0, which minimises the re-ranked rows ~ k * sum(ratios).
  //  * otherwise: the head is scanned exactly (with the filter) by the stream kernel, stages 32 x and 256 x S0.
  const FilterSpec fs = ix.filter_spec();
  std::vector<int64_t> bounds;
  if (seeded) {
    bounds.push_back(approx ? S0 : 0);   // approx mode keeps the head's approximate keys themselves: no second visit
    // Stage count.  A stage whose rows outnumber the rows before it by a factor f passes ~ k * c * f candidates per query, c = how
    // many times more rows lie within the bound's margin of the threshold than below it (fp16: ~1; int8: ~4-5 on U[0,1) rows at
    // d = 768); S stages with equal ratios (n / S0)^(1/S) cost S * k * c * (n / S0)^(1/S) candidates and S times the per-stage
    // overhead (launch tails + one re-rank launch, ~0.1 ms at 10M rows).  A candidate costs its wavefront ~900 cycles in the
    // filter's epilogue and 3 KB of gather in the re-rank, so the looser 8-bit bound wants more, smaller steps (measured at
    // 10M x 768, batch 1024: EPS_MFMA_STAGES sweep in profiles/r3_stage_sweep.txt).
    const char* st_env = tune_env("EPS_MFMA_STAGES");
    // Few queries (<= 64): 4 stages.  A call is then a chain of short dependent launches (profiles/r3_single_query_timeline.txt: one
    // query on 1M x 768 = 400 us of back-to-back kernels, 185 us of them the filter stages streaming the mirror once, 110 us seven
    // one-workgroup re-ranks), and two re-ranks less beat the longer lists: scripts/lab/stages_by_batch.py, 1M x 768, p50 ms at
    // 1 / 16 / 64 queries: 0.392 / 0.440 / 0.504 (3 stages), 0.400 / 0.438 / 0.494 (4), 0.431 / 0.466 / 0.515 (6); from 128 queries
    // on 6 stages win (0.648 vs 0.707 with 3), at 10M rows as well.
    // r4, a handful of queries (<= 4): 3 stages - with the centred grid a stage passes half the candidates it used to, and every stage
    // less is one filter launch tail and one one-workgroup re-rank off a chain of dependent launches (scripts/lab/single_query_stages.sh)
    int nstages = i8 ? (nq <= 4 ? 3 : (nq <= 64 ? 4 : 6)) : 3;
    if (i8)   // ... but never so few that a stage's expected k * c * ratio candidates come near the list capacity
      while (nstages < 8 && (double)k * 5.0 * std::pow((double)n / (double)S0, 1.0 / (double)nstages) > 0.5 * (double)cap) ++nstages;
    if (st_env) nstages = std::min(8, std::max(1, atoi(st_env)));
    const double r = std::max(i8 ? 3.0 : 4.0, std::pow((double)n / (double)S0, 1.0 / (double)nstages));
    // stage boundaries on multiples of the rows one "round" of the persistent grid covers (256 workgroups x 256 rows /
    // query tiles), so the small stages do not end on a mostly idle round
    const int64_t qt = std::max<int64_t>(1, b_pad / 256);
    const int64_t unit = 256 * std::max<int64_t>(1, 256 / std::gcd<int64_t>(256, qt));
    double f = 1.0;
    for (int st = 1; st < nstages; ++st) {
      f *= r;
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
  m.s8_clean_cnt = nullptr;   // (the chain's counters live where the one-pass form keeps its own)
  u32* seed_cand_buf = m.seedc.as<u32>();
  u32* seed_cnt_buf = cnt;   // (merge_lists reads a query's seed count before it writes the candidate count there; the re-rank zeroes it)
  u32* overflow = cnt + nq;                                                    // [1]
  unsigned long long* total = reinterpret_cast<unsigned long long*>(cnt + nq + 2);  // 8-byte aligned? ensured below
  if ((reinterpret_cast<uintptr_t>(total) & 7) != 0) total = reinterpret_cast<unsigned long long*>(cnt + nq + 3);
  hipError_t er = prologue ? hipSuccess : hipMemsetAsync(cnt + nq, 0, 32, s);
  if (er != hipSuccess) return ix.hip_fail(er, "memset");

  FilterArgs fa;
  fa.xh = i8 ? reinterpret_cast<const _Float16*>(m.x8.p) : m.xh.as<_Float16>();
  fa.qh = q_op;
  fa.qf = m.qf.as<_Float16>();
  // (8-bit: int32 words, only ever moved; exact mode: the batch's folded start values)
  fa.base = i8 ? (fold ? m.acc0b.as<float>() : m.acc0.as<float>()) : (ix.metric_ == 0 ? m.xn.as<float>() : m.zeros.as<float>());
  fa.base_s = i8 ? (fold ? m.acc0b.as<float>() : m.acc0.as<float>()) : (ix.metric_ == 0 ? m.xn_s.as<float>() : m.zeros_s.as<float>());
  fa.T = m.T.as<float>();
  fa.d_pad = d_pad_h;
  fa.tiles_q = (int)(b_pad / BN3);
  fa.nq = nq;
  fa.s = i8 ? -u8 : (ix.metric_ == 0 ? -2.f : -1.f);
  fa.inv_s = 1.f / fa.s;
  fa.cand = m.cand.as<u32>();
  fa.cand_keys = approx ? m.cand.as<u64>() : nullptr;
  fa.qstat = m.qstat.as<float>();
  fa.metric = ix.metric_;
  fa.cnt = cnt;
  fa.cap = cap;
  fa.group_sync = nullptr;
  fa.sync_shift = std::min(8, std::max(0, tune_int("EPS_MFMA_SYNC_SHIFT", 2)));
  fa.dense = 0;
  fa.ablate = 0;
  fa.prof = nullptr;
#ifdef EPS_V7_PROF
  static DevBuf prof_buf;   // (lab builds only)
  if (prof_buf.reserve(64)) {
    (void)hipMemsetAsync(prof_buf.p, 0, 64, s);
    fa.prof = prof_buf.as<unsigned long long>();
  }
#endif

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
  // stage bookkeeping folded into the re-rank (exact mode): counts of the stage it follows + thresholds of the stage that follows it
  ra.fuse = approx ? 0 : 1;
  ra.overflow = overflow;
  ra.total = total;
  ra.T_next = nullptr;
  ra.qstat = m.qstat.as<float>();
  ra.scal = i8 ? (fold ? m.scal8f.as<float>() : m.scal8.as<float>()) : m.scal.as<float>();
  ra.bits = i8 ? 8 : 16;
  ra.u = u8;
  ra.slack = 0.f;   // (set below, with the stages)
  ra.gsync = m.gsync.as<u32>();
  if (!approx && nq <= 16 && k <= 128 && tune_int("EPS_RERANK_SPLIT", 0) != 0) {
    // a handful of queries: every re-rank spread over 8 workgroups per query (RerankArgs::parts).  Opt-in: measured, it takes 5 us off a
    // 340 us single-query call (profiles/r4_single_query_latency.txt) - a re-rank of ~150 rows is a chain of dependent latencies, not a
    // bandwidth problem - and is not worth a cross-workgroup hand-off on the default path
    const bool fresh = m.partd.cap < (size_t)nq * 4;
    if (!m.partk.reserve((size_t)nq * 8 * k * 8) || !m.partd.reserve((size_t)64 * 4)) return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (scratch)");
    if (fresh) {
      const hipError_t e0 = hipMemsetAsync(m.partd.p, 0, m.partd.cap, s);
      if (e0 != hipSuccess) return ix.hip_fail(e0, "memset");
    }
    ra.parts = 8;
    ra.part_keys = m.partk.as<u64>();
    ra.part_done = m.partd.as<u32>();
  }

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
                           reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_KEYS>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_DENSE>),
                           reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_IDS, true>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_KEYS, true>),
                           reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_DENSE, true>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_IDS, true>),
                           reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_KEYS, true>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<1, FM_DENSE, true>)})
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)V7_LDS_BYTES);
    for (const void* fn : {reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_IDS, true, 4>), reinterpret_cast<const void*>(mfma_filter_kernel_v7<2, FM_KEYS, true, 4>)})
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v7_lds_bytes(4));
  }
  const int num_cus = m.num_cus;
  const bool narrow_env = !(tune_int("EPS_MFMA_NARROW", 1) == 0);
  const bool two_per_cu = tune_int("EPS_MFMA_TWO_PER_CU", 0) != 0;   // (lab until measured)
  auto launch_filter = [&](const FilterArgs& f) {
    {
      FilterArgs f3 = f;
      f3.tiles_q = (int)(b_pad / BN3);
      if (version >= 7) {
        f3.group_sync = gsync_env ? m.gsync.as<u32>() : nullptr;
        if (f3.group_sync && f3.dense && !prologue) (void)hipMemsetAsync(f3.group_sync, 0, 1024, s);   // (stages: reset by threshold_kernel / the re-rank)
        const int mode = f3.dense ? FM_DENSE : (f3.cand_keys ? FM_KEYS : FM_IDS);
        const dim3 grid((unsigned)num_cus), block(256);
#define EPS_V7_LAUNCH(JQ_, I8_)                                                                                             \
  do {                                                                                                                      \
    if (mode == FM_DENSE) hipLaunchKernelGGL((mfma_filter_kernel_v7<JQ_, FM_DENSE, I8_>), grid, block, shm, s, f3);         \
    else if (mode == FM_KEYS) hipLaunchKernelGGL((mfma_filter_kernel_v7<JQ_, FM_KEYS, I8_>), grid, block, shm, s, f3);      \
    else hipLaunchKernelGGL((mfma_filter_kernel_v7<JQ_, FM_IDS, I8_>), grid, block, shm, s, f3);                            \
  } while (0)
        if (nq <= 128 && narrow_env) {   // one 128-query tile: half the padded MFMA work, the pass streams the mirror
          f3.tiles_q = 1;
          if (i8) EPS_V7_LAUNCH(1, true); else EPS_V7_LAUNCH(1, false);
        } else if (i8 && mode != FM_DENSE && two_per_cu) {
          // r4: 128-row tiles, two workgroups per CU (see the kernel: NRB = 4)
          f3.tile0 *= 2;
          f3.ntiles *= 2;
          const dim3 grid2((unsigned)num_cus * 2);
          if (mode == FM_KEYS) hipLaunchKernelGGL((mfma_filter_kernel_v7<2, FM_KEYS, true, 4>), grid2, block, v7_lds_bytes(4), s, f3);
          else hipLaunchKernelGGL((mfma_filter_kernel_v7<2, FM_IDS, true, 4>), grid2, block, v7_lds_bytes(4), s, f3);
        } else {
          if (i8) EPS_V7_LAUNCH(2, true); else EPS_V7_LAUNCH(2, false);
        }
#undef EPS_V7_LAUNCH
      }
      else hipLaunchKernelGGL(mfma_filter_kernel_v3, dim3((unsigned)num_cus), dim3(512), shm, s, f3);
    }
  };
  // fp32 rounding of the keys the threshold compares: |x|^2, |q|^2 and the re-ranked distance are each a 64-lane sum of
  // d_pad/64 sequential fmas per lane plus a 6-level shuffle tree, i.e. <= (d_pad/64 + 6) * 2^-24 relative to their
  // magnitude each; doubled for safety.  (A fixed 8e-6 was only enough up to d ~ 1000.)
  const float rerank_slack = std::max(8e-6f, 2.f * (3.f * ((float)((ix.dim_ + 63) / 64 * 64) / 64.f + 6.f) + 6.f) * 5.9604645e-8f);
  ra.slack = rerank_slack;
  const bool fused = !approx;   // exact mode: every re-rank also does its stage's counts and the next stage's thresholds
  if (seeded) {
    const bool dense = version >= 7;   // v7 writes the head's keys densely (slot = row); older kernels append with atomics
    if (prologue && prep_does_it_all) {
      // (query_prep8_kernel laid the start state down)
    } else if (prologue) {
      const int64_t cells = std::max<int64_t>(std::max<int64_t>(b_pad / 2, nq), 256);
      hipLaunchKernelGGL(seed_prologue_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, reinterpret_cast<u64*>(m.T.p), b_pad / 2, 0x7F8000007F800000ull,
                         cnt, nq, (u32)S0, gsync_env ? m.gsync.as<u32>() : nullptr);
    } else {
      launch_fill_u64(reinterpret_cast<u64*>(m.T.p), b_pad / 2, 0x7F8000007F800000ull, s);   // T = +inf: every head row is a candidate
      er = dense ? hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cnt), (int)S0, (size_t)nq, s) : hipMemsetAsync(cnt, 0, (size_t)nq * 4, s);
      if (er != hipSuccess) return ix.hip_fail(er, "memset");
    }
    unsigned long long seed_stride = 0;   // != 0: the seed pass ran over the sample, ids are sample indices
    u32 seed_head = 0;
    FilterArgs f0 = fa;
    f0.dense = dense ? 1 : 0;
    f0.cand_keys = m.cand.as<u64>();
    if (!approx) {   // exact mode: seeds from a sample spread over the whole table (approx mode keeps the head's keys)
      const u32 sample_head = (u32)(S0 / 2);
      const unsigned long long sample_stride = (unsigned long long)(((unsigned __int128)(n - sample_head) << 32) / (unsigned __int128)(S0 - sample_head));
      // (one sample per operand width)
      int64_t& smp_version = i8 ? m.sample8_version : m.sample_version;
      int64_t& smp_n = i8 ? m.sample8_n : m.sample_n;
      int64_t& smp_rows = i8 ? m.sample8_rows : m.sample_rows;
      DevBuf& smp_x = i8 ? m.sx8 : m.sxh;
      DevBuf& smp_base = i8 ? m.sacc0 : m.sbase;
      DevBuf& smp_base_u = i8 ? m.sacc0 : m.sbase_u;
      if (smp_version != ix.rows_version_ || smp_n != n || smp_rows != S0) {   // (also after an append: n changed)
        if (!smp_x.reserve((size_t)S0 * d_pad_h * 2) || !smp_base.reserve((size_t)S0 * 4) || !smp_base_u.reserve((size_t)S0 * 4))
          return ix.fail(EPS_INFRA_UNEXPECTED_ERROR, "MFMA engine: out of device memory (seed sample)");
        // (8-bit: the sample keeps the UNFOLDED start values - the dense seed pass ranks approximate keys and tests nothing - so it stays
        // valid from batch to batch)
        hipLaunchKernelGGL(seed_sample_kernel, dim3((unsigned)S0), dim3(256), 0, s, fa.xh, i8 ? m.acc0.as<u32>() : reinterpret_cast<const u32*>(fa.base_s),
                           i8 ? m.acc0.as<u32>() : reinterpret_cast<const u32*>(fa.base),
                           sample_stride, sample_head, d_pad_h, smp_x.as<_Float16>(), smp_base.as<u32>(), smp_base_u.as<u32>());
        smp_version = ix.rows_version_;
        smp_n = n;
        smp_rows = S0;
      }
      f0.xh = smp_x.as<_Float16>();
      f0.base_s = smp_base.as<float>();
      f0.base = smp_base_u.as<float>();
      seed_stride = sample_stride;
      seed_head = sample_head;
    }
    f0.tile0 = 0;
    f0.ntiles = (S0 + bm - 1) / bm;
    f0.row_hi = S0;
    launch_filter(f0);
    // k best approximate keys of the visible seeds; exact mode: straight to the candidate lists of the re-rank that follows (the
    // dense seed keys live in the upper half of the candidate buffer's u64 view, the lists in its u32 view: disjoint for cap >= 2 k)
    if (approx) launch_merge_lists(f0.cand_keys, cap, k, nq, run_keys, false, s, cnt, nullptr, seed_stride, seed_head);
    else launch_merge_lists(f0.cand_keys, cap, k, nq, run_keys, false, s, cnt, &fs, seed_stride, seed_head, seed_cand_buf, k, seed_cnt_buf);
    if (!approx) {
      ra.cand = seed_cand_buf;
      ra.cap = k;
      ra.cand_count = seed_cnt_buf;
      ra.fuse = 2;                                                              // (thresholds of the first stage; the seeds are not a stage's candidates)
      ra.T_next = m.T.p;
      launch_rerank(ra, s);                                                    // -> their exact keys
      ra.cand = fa.cand;
      ra.cap = cap;
      ra.cand_count = cnt;
    }
  }
  bool first = true;
  bool fin_done = false;
  const bool probe = i8 && auto_bits && !approx && seeded && !m.i8_trusted && bounds.size() > 3 && !(tune_int("EPS_MFMA_PROBE", 1) == 0);
  for (size_t st = 0; st + 1 < bounds.size(); ++st) {
    const int64_t lo = bounds[st], hi = bounds[st + 1];
    {
      // thresholds of this stage: from the previous re-rank (fused), except for the padding entries (once), the approx mode and
      // the unseeded staging (whose stage 0 was a stream scan)
      const bool have_T = fused && (st > 0 || seeded);
      const int pad_only = have_T ? 1 : 0;
      // (8-bit, seeded: the prologue left 0x7F800000 in the padding entries - as an int32 threshold "never passes" - and the dense
      // seed pass does not read thresholds, so the pad-only launch is not needed)
      if ((!have_T || st == 0) && !(have_T && i8 && prologue)) {
        if (i8)
          hipLaunchKernelGGL(threshold8_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, s, run_keys, k, nq, b_pad, m.qstat.as<float>(),
                             fold ? m.scal8f.as<float>() : m.scal8.as<float>(), ix.metric_, u8, m.T.as<int>(), cnt, m.gsync.as<u32>(), rerank_slack, approx ? 1 : 0, pad_only);
        else
          hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, s, run_keys, k, nq, b_pad,
                             m.qstat.as<float>(), m.scal.as<float>(), ix.metric_, m.T.as<float>(), cnt, m.gsync.as<u32>(), rerank_slack, approx ? 1 : 0, pad_only);
      }
    }
    fa.tile0 = lo / bm;
    fa.ntiles = (hi + bm - 1) / bm - fa.tile0;
    fa.row_hi = hi;
    const bool biggest = (st + 2 == bounds.size());
    if (biggest) (void)hipEventRecord(ix.evk0_, s);
    if (biggest) {
      ix.stats_.main_kernel_rows = hi - lo;
      ix.stats_.main_kernel_queries = nq;
      ix.stats_.main_kernel_bits = i8 ? 8 : 16;
    }
    // (timed for throughput-sized batches only: an event record between two dependent launches costs a single-query chain ~4 us per
    // launch boundary; the build's kNN stage runs thousands of calls: untimed)
    const int sev = (!approx && nq >= 256 && ix.stage_n_ < Index::STAGE_EV) ? ix.stage_n_++ : -1;
    if (sev >= 0) (void)hipEventRecord(ix.stage_ev_[sev][0], s);
    launch_filter(fa);
    if (sev >= 0) {
      (void)hipEventRecord(ix.stage_ev_[sev][1], s);
      ix.stats_.filter_rows_all += hi - lo;
    }
    if (biggest) (void)hipEventRecord(ix.evk1_, s);
    if (!fused) hipLaunchKernelGGL(stage_counts_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, cnt, nq, cap, overflow, total);
    if (tune_env("EPS_DEBUG")) {
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
    if (approx) {
      launch_merge_lists(fa.cand_keys, cap, k, nq, run_keys, true, s, cnt);  // select on the approximate keys
    } else {
      ra.fuse = 3;                                                            // this stage's counts + the next stage's thresholds
      ra.T_next = (st + 2 < bounds.size()) ? m.T.p : nullptr;
      const bool fin_here = biggest && ix.pre_sync_ && nq == ix.pre_sync_nq_ && ix.fin_ids_ != nullptr;
      if (fin_here) {   // the last re-rank writes the caller-visible result itself (rewritten by a fall-back pass if the lists overflowed)
        ra.fin_ids = ix.fin_ids_;
        ra.fin_dist = ix.fin_dist_;
        ra.fin_counts = ix.fin_cnt_;
        ra.fin_base = ix.id_base_;
        ra.fin_stride = ix.id_stride_;
      }
      launch_rerank(ra, s);
      if (fin_here) {
        (void)hipEventRecord(ix.ev1_, s);
        ix.result_finalized_ = true;
        fin_done = true;
      }
      if (probe && st == 0) {
        // The library's own choice of the 8-bit pass is PROBED once per mirror: every stage passes ~ k * c * ratio candidates per query
        // (c = how many times more rows lie within the bound's margin of the threshold than below it), so the first, smallest stage
        // predicts the others.  Where c is large - rows whose neighbours are close against the value range: low intrinsic dimension,
        // tight clusters - the lists of the big stages would overflow and the batch would pay the 8-bit attempt AND the fp16 pass
        // (r3: 10M x 768 manifold set 73 k -> 39 k q/s); here it pays one small stage and one sync, once.
        struct { u32 overflow, pad; unsigned long long total; } hp = {0, 0, 0};
        er = hipMemcpyAsync(&hp.overflow, overflow, 4, hipMemcpyDeviceToHost, s);
        if (er == hipSuccess) er = hipMemcpyAsync(&hp.total, total, 8, hipMemcpyDeviceToHost, s);
        if (er == hipSuccess) er = hipStreamSynchronize(s);
        if (er != hipSuccess) return ix.hip_fail(er, "MFMA filter (probe)");
        if (hp.overflow || hp.total > (unsigned long long)nq * (unsigned long long)cap / 6) {
          // declined for this mirror (re-attaching rows re-arms it; EPS_FLAT_MFMA_I8 still forces it) - unless a deleted bitset or a filter is
          // active: a selective filter inflates the lists by 1 / (its pass fraction) whatever the bound is worth, so such a batch only
          // decides for itself and the next one probes again
          if (!(fs.deleted || fs.column || fs.prog)) m.i8_overflows = 3;
          ix.stats_.i8_declined += 1;
          return flat_mfma_search_slice(ix, dq, nq, k, run_keys, approx, 1, 16, false);
        }
        m.i8_trusted = true;
      }
    }
    first = false;
  }
  (void)first;
  er = hipGetLastError();
  if (er != hipSuccess) return ix.hip_fail(er, "MFMA filter launch");
  struct {
    u32 overflow, pad;
    unsigned long long total;
  } h = {0, 0, 0};
  if (!fin_done && ix.pre_sync_ && !approx && nq == ix.pre_sync_nq_) ix.pre_sync_();   // (speculative: a fall-back pass below converts again)
  er = hipMemcpyAsync(&h.overflow, overflow, 4, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipMemcpyAsync(&h.total, total, 8, hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return ix.hip_fail(er, "MFMA filter");
#ifdef EPS_V7_PROF
  if (fa.prof) {
    unsigned long long pr[4] = {0, 0, 0, 0};
    (void)hipMemcpy(pr, fa.prof, 32, hipMemcpyDeviceToHost);
    if (pr[3]) fprintf(stderr, "[eps v7 prof] wave-tiles %llu: head %.0f  K loop %.0f  epilogue %.0f cycles per tile (all stages of this call)\n", pr[3],
                       (double)pr[0] / pr[3], (double)pr[1] / pr[3], (double)pr[2] / pr[3]);
  }
#endif
  ix.stats_.rerank_rows += (int64_t)h.total;
  ix.stats_.dist_evals += nq * (n - bounds[0]) + (seeded ? nq * S0 : 0);   // (exact mode visits the head twice)
  ix.stats_.main_kernel_launches = 1;
  if (h.overflow) {
    ix.result_finalized_ = false;   // (whatever was converted before the sync is stale: a pass below rewrites the result keys)
    ix.stats_.overflow_queries += h.overflow;
    if (!approx && !fa.ablate) {
      // (selective filters inflate the lists by 1 / pass fraction, adversarial row orders by more): first retry with 16 x
      // the candidate slots - re-ranking tens of thousands of rows per query is still ~50 x cheaper than the stream scan
      // of a large batch - then the exact stream engine
      if (i8) {   // the looser 8-bit bound let too much through: fp16 pass
        m.i8_overflows += 1;
        m.i8_trusted = false;
        return flat_mfma_search_slice(ix, dq, nq, k, run_keys, approx, 1, 16, false);
      }
      if (cap_scale == 1 && (size_t)nq * cap * 16 * 8 <= ((size_t)4 << 30)) return flat_mfma_search_slice(ix, dq, nq, k, run_keys, approx, 16, 16, false);
      return ix.flat_stream(dq, nq, k, 0, n, run_keys, false);
    }
  }
  if (i8) m.i8_overflows = 0;
  return EPS_OK;
}

// Batches beyond 2048 queries run as slices of 2048: the kernel keeps one slice's fp16 query tile set (3 MB) resident in
// each XCD's 4 MB L2 while the row operand streams past; at 4096 / 8192 queries per pass the query fragments thrash L2
// and the filter drops to 0.37 / 0.27 of the MFMA peak (0.46 in slices; bench.py --rows 1250000 --batch 8192).
int32_t flat_mfma_search(Index& ix, const float* dq, int64_t nq, int k, u64* run_keys, bool approx, int bits) {
  if (ix.n_rows_ <= 0) return ix.flat_stream(dq, nq, k, 0, 0, run_keys, false);   // (nothing to mirror)
  const bool auto_bits = bits != 8 && bits != 16;
  if (auto_bits) {   // the library's choice: 8-bit first pass unless switched off (EPS_MFMA_BITS=16, A/B) - tables it cannot serve fall back by themselves
    const char* e = tune_env("EPS_MFMA_BITS");
    bits = (e && atoi(e) == 16) ? 16 : 8;
  }
  const int64_t slice = std::max(256, tune_int("EPS_MFMA_MAX_BATCH", 2048));
  if (nq <= slice) return flat_mfma_search_slice(ix, dq, nq, k, run_keys, approx, 1, bits, auto_bits);
  for (int64_t q0 = 0; q0 < nq; q0 += slice) {   // the counters in ix.stats_ accumulate over the slices
    const int32_t rc = flat_mfma_search_slice(ix, dq + q0 * ix.dim_, std::min(slice, nq - q0), k, run_keys + q0 * k, approx, 1, bits, auto_bits);
    if (rc != EPS_OK) return rc;
  }
  return EPS_OK;
}

}  // namespace eps
