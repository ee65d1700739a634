// Device-side building blocks shared by every kernel of libepsilla_gfx950 (gfx950 / CDNA4 only).
//
//  * Candidate keys.  The reference orders candidates by (distance, id) (db/execution/candidate.hpp:16-22).
//    On the device a candidate is one u64: order-preserving image of the fp32 distance in the high word,
//    32-bit local row id in the low word, so one v_cmp_lt_u64 is the reference's operator<.
//  * WaveTopK: a sorted k-list held in the registers of ONE 64-lane wavefront (entry e lives in register
//    e/64 of lane e%64).  Insert = ballot/popcount rank + one wave_shr; replaces the reference's
//    lower_bound + memmove (AddIntoQueue, vec_search_executor.cpp:75-117).
//  * row_dists: G-lane-per-row fp32 distance evaluation with 16 B/lane coalesced loads and
//    __shfl_xor reductions; replaces fvec_L2sqr / fvec_inner_product (db/index/distance_simd.cpp:180-214).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eps {

using u64 = unsigned long long;
using u32 = unsigned int;

constexpr u64 KEY_EMPTY = ~0ull;

// One instruction of a compiled filter (eps_filter_op in include/epsilla_gfx950.h): a postfix program over the packed
// attribute row of a candidate, evaluated on a small stack of doubles exactly as ExprEvaluator::NumEvaluate /
// LogicalEvaluate do (query/expr/expr_evaluator.cpp:127-258: every number is a double, booleans are 0 / 1).
struct FilterOp {
  int32_t op;     // EPS_FOP_*
  int32_t arg;    // byte offset inside the attribute row (attribute loads)
  int64_t ival;
  double dval;    // constant
};

struct FilterSpec {            // deleted bitset + `int column <op> constant` or a compiled program (device pointers)
  const uint8_t* deleted;      // may be null
  const uint8_t* column;       // may be null
  int64_t stride;
  int32_t width;
  int32_t op;                  // EPS_OP_*
  int64_t value;
  const FilterOp* prog;        // may be null: compiled filter program over rows of `prog_stride` bytes at `prog_rows`
  const uint8_t* prog_rows;
  int64_t prog_stride;
  int32_t prog_len;
  int32_t prog_use_dist;       // 0: @distance evaluates to 0 (PreFilterBruteForceSearch, :795)
};
__host__ __device__ inline FilterSpec no_filter() { return FilterSpec{nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, 0, 0, 0}; }

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// fp32 -> u32 preserving order (NaN sorts last, -0 is folded into +0 by the callers' `+ 0.0f`)
__device__ __forceinline__ u32 f2ord(float f) {
  u32 u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(u32 o) {
  u32 u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float dist, u32 id) { return ((u64)f2ord(dist + 0.0f) << 32) | (u64)id; }
__device__ __forceinline__ float key_dist(u64 k) { return ord2f((u32)(k >> 32)); }
__host__ __device__ __forceinline__ u32 key_id(u64 k) { return (u32)k; }

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  lo = __shfl(lo, src);
  hi = __shfl(hi, src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up64(u64 v, int d) {
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  lo = __shfl_up(lo, d);
  hi = __shfl_up(hi, d);
  return ((u64)hi << 32) | lo;
}

// EPS_FOP_* opcodes (mirrored in include/epsilla_gfx950.h)
enum : int32_t {
  FOP_PUSH_CONST = 1, FOP_PUSH_DIST = 2, FOP_PUSH_I8 = 3, FOP_PUSH_I16 = 4, FOP_PUSH_I32 = 5, FOP_PUSH_I64 = 6, FOP_PUSH_F32 = 7,
  FOP_PUSH_F64 = 8, FOP_PUSH_BOOL = 9, FOP_ADD = 10, FOP_SUB = 11, FOP_MUL = 12, FOP_DIV = 13, FOP_MOD = 14, FOP_LT = 15, FOP_LE = 16,
  FOP_EQ = 17, FOP_NE = 18, FOP_GE = 19, FOP_GT = 20, FOP_AND = 21, FOP_OR = 22, FOP_NOT = 23, FOP_EQ_BOOL = 24, FOP_NE_BOOL = 25
};

__device__ inline bool eval_filter_program(const FilterSpec& f, u32 id, float dist) {
  double st[16];
  int sp = 0;
  const uint8_t* row = f.prog_rows + (int64_t)id * f.prog_stride;
  for (int i = 0; i < f.prog_len; ++i) {
    const FilterOp o = f.prog[i];
    if (o.op <= FOP_PUSH_BOOL) {
      double v = 0.0;
      switch (o.op) {
        case FOP_PUSH_CONST: v = o.dval; break;
        case FOP_PUSH_DIST: v = f.prog_use_dist ? (double)dist : 0.0; break;
        case FOP_PUSH_I8: v = (double)*(const int8_t*)(row + o.arg); break;
        case FOP_PUSH_I16: v = (double)*(const int16_t*)(row + o.arg); break;
        case FOP_PUSH_I32: v = (double)*(const int32_t*)(row + o.arg); break;
        case FOP_PUSH_I64: v = (double)*(const int64_t*)(row + o.arg); break;
        case FOP_PUSH_F32: v = (double)*(const float*)(row + o.arg); break;
        case FOP_PUSH_F64: v = *(const double*)(row + o.arg); break;
        case FOP_PUSH_BOOL: v = row[o.arg] != 0 ? 1.0 : 0.0; break;   // "true iff byte != 0" (expr_evaluator.cpp:56-59)
      }
      if (sp < 16) st[sp++] = v;
    } else if (o.op == FOP_NOT) {
      if (sp >= 1) st[sp - 1] = st[sp - 1] != 0.0 ? 0.0 : 1.0;
    } else if (sp >= 2) {
      const double b = st[--sp], a = st[sp - 1];
      double r = 0.0;
      switch (o.op) {
        case FOP_ADD: r = a + b; break;
        case FOP_SUB: r = a - b; break;
        case FOP_MUL: r = a * b; break;
        case FOP_DIV: r = a / b; break;
        case FOP_MOD: r = fmod(a, b); break;
        case FOP_LT: r = a < b; break;
        case FOP_LE: r = a <= b; break;
        case FOP_EQ: r = a == b; break;
        case FOP_NE: r = a != b; break;
        case FOP_GE: r = a >= b; break;
        case FOP_GT: r = a > b; break;
        case FOP_AND: r = (a != 0.0) && (b != 0.0); break;
        case FOP_OR: r = (a != 0.0) || (b != 0.0); break;
        case FOP_EQ_BOOL: r = (a != 0.0) == (b != 0.0); break;
        case FOP_NE_BOOL: r = (a != 0.0) != (b != 0.0); break;
      }
      st[sp - 1] = r;
    }
  }
  return sp >= 1 && st[0] != 0.0;
}

// deleted_.test(id) || !LogicalEvaluate(root, id, dist)  (vec_search_executor.cpp:751, :911): dist is the candidate's distance
__device__ __forceinline__ bool row_visible(const FilterSpec& f, u32 id, float dist = 0.f) {
  if (f.deleted && ((f.deleted[id >> 3] >> (id & 7)) & 1)) return false;
  if (f.op && f.column) {
    const uint8_t* p = f.column + (int64_t)id * f.stride;
    int64_t v;
    switch (f.width) {
      case 1: v = *(const int8_t*)p; break;
      case 2: v = *(const int16_t*)p; break;
      case 8: v = *(const int64_t*)p; break;
      default: v = *(const int32_t*)p; break;
    }
    switch (f.op) {
      case 1: return v < f.value;
      case 2: return v <= f.value;
      case 3: return v == f.value;
      case 4: return v >= f.value;
      case 5: return v > f.value;
      case 6: return v != f.value;
    }
  }
  if (f.prog) return eval_filter_program(f, id, dist);
  return true;
}

// ---- pass thresholds of the MFMA filter stages (mfma_filter.hip explains the bounds).  kth_dist: the k-th best exact (or, in approx
// mode, approximate) distance so far; qs: the query's |q|^2, |q|, |q - qh|, (8-bit: C + const) ; sc: the mirror's maxima.
// fp16 operands: T in key space (float)
__device__ __forceinline__ float stage_threshold16(float thr, const float* qs, const float* sc, int metric, float slack, int approx) {
  const float qn2 = qs[0], nq_ = qs[1], eq = qs[2];
  const float e1max = sc[0], nxhmax = sc[1], xnmax = sc[2];
  const float s = metric == 0 ? 2.f : 1.f;
  const float c = metric == 0 ? qn2 : (metric == 1 ? 1.f : 0.f);
  const float margin = s * (nq_ * e1max + eq * nxhmax);
  const float scale = metric == 0 ? (fabsf(thr) + qn2 + xnmax) : (fabsf(thr) + 1.f + nq_ * nxhmax);
  const float t = (thr - c) + (approx ? 0.f : margin) + slack * scale;   // (approx mode ranks on the approximate keys: no margin)
  return fminf(t, 3.0e38f);
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
// value on the table's 8-bit grid (xh = z + step * xi)
__device__ __forceinline__ int quant8(float x, float z, float inv_step) {
  float t = rintf((x - z) * inv_step);
  t = fminf(fmaxf(t, -127.f), 127.f);   // (rows appended after the grid was fixed may lie outside it: clamped, the residual grows, the bound stays valid)
  return (int)t;
}
// ---- r6: the 8-bit grid in a ROTATED frame.  The Cauchy-Schwarz margin of the 8-bit bound is (step x sqrt(columns / 12)) x the query's norm:
// one step for the whole table, so a few columns that carry most of a row's energy (unit-norm embedding rows: 8 dominant dimensions of 768)
// stretch the grid for all the others and the margin reaches the spread of the distances themselves - the 8-bit pass cannot filter and the
// fp16 pass serves the table at half the matrix rate.  L2, dot and cosine do not change under an orthogonal map of rows AND queries, so
// such a table is quantised as y = R x, R = blockdiag(H_256 / 16) . S . P: a fixed permutation P of the d_pad8 (zero-padded) columns,
// signs S, and a 256-point Walsh-Hadamard transform per 256-column block.  R's entries are +-1/16: R is EXACTLY orthogonal as a real
// matrix, so q.x = (Rq).(Rx) is an identity, not an approximation.  Every rotated column is a signed mean of 256 of the row's values: the
// columns all have the same spread, nothing dominates, and the step shrinks to what the row's norm needs (embedding-like rows: 0.36 x,
// 14 x fewer survivors per query; U[0,1) rows, which fill the grid evenly as they are: 2.9 x WORSE - so the frame is chosen per table,
// when the mirror is first built, from the two steps measured on the same sample: ensure_mirror8).
// Arithmetic: the transform runs in fp64 (8 add levels: error <= 8 x 2^-53 x 16 |x_block| per column, i.e. <= 1.5e-14 |x| for the row) and
// y - mu is rounded to fp32 ONCE - the rounding the identity frame's x - mu has, and that its residual norm already carries
// (+ 1.2e-7 sqrt(c2)); the fp64 remainder is added as + 1e-12 |x|.  Restated in numpy in tests/test_bound_math.py (rotate_rows).
// sp[p]: source column of rotated-input position p, bit 31 = negative sign; positions whose source is >= dim read 0.
// One wavefront per row; lane l holds positions c .. c + 3, c = block * 256 + 4 l: two butterfly levels in the lane, six across lanes.
__device__ __forceinline__ void rot256_load(const float* src, int dim, const int* sp, int c, int lane, double xd[4]) {
  const int4 s4 = *reinterpret_cast<const int4*>(sp + c);
  const int ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = ss[e] & 0x7FFFFFFF;
    const float v = col < dim ? src[col] : 0.f;
    xd[e] = (double)(ss[e] < 0 ? -v : v);
  }
  {
    const double a = xd[0] + xd[1], b = xd[0] - xd[1], cc = xd[2] + xd[3], d = xd[2] - xd[3];
    xd[0] = a + cc;
    xd[1] = b + d;
    xd[2] = a - cc;
    xd[3] = b - d;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double other = __shfl_xor(xd[e], o);
      xd[e] = up ? other - xd[e] : xd[e] + other;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) xd[e] *= 0.0625;
}

// int8 operands: T in accumulator units (int32): a row passes iff dot + acc0 >= T.  thr: a distance; qs: |q|^2, |q - mu|, |q' - qh'|, C[q]
// (query_prep8_kernel); sc: max |x' - xh'|, max |xh'|, max |x|^2, -, max |R|, |mu| (quant_mirror_kernel).  Restated in numpy, with the
// implication it must guarantee, in tests/test_bound_math.py.
__device__ __forceinline__ int stage_threshold8(float thr, const float* qs, const float* sc, int metric, float u, float slack, int approx) {
  const float qn2 = qs[0], nqc = qs[1], eq = qs[2], Cq = qs[3];
  const float e1max = sc[0], nxhmax = sc[1], xnmax = sc[2], rmax = sc[4], mun = sc[5], xcmax = sc[6];   // (folded launches: sc[0] = sc[1] = 0, the margin is in the rows' start values)
  const float s = metric == 0 ? 2.f : 1.f;
  const float margin = s * (nqc * e1max + eq * nxhmax);   // Cauchy-Schwarz on the two stored residuals, in the centred frame
  // fp32 evaluation of the re-ranked distance, of R and of C (each a d-term sum whose terms' magnitudes sum to at most the scale
  // below), of x - mu, and of the two divisions by u
  const float scale = metric == 0 ? fabsf(thr) + 2.f * fabsf(Cq) + 2.f * rmax
                                  : fabsf(thr) + 1.f + sqrtf(qn2) * (sqrtf(xnmax) + mun) + mun * xcmax + fabsf(Cq) + rmax;
  // approx mode (the build's kNN stage) ranks on the approximate keys themselves: a row is wanted iff its APPROXIMATE key beats the
  // k-th best approximate key so far - no margin (with it several times the rows pass, and every one costs an append)
  const float t = approx ? thr + slack * scale + 4.f * u : thr + margin + slack * scale + 4.f * u;
  float v = floorf((Cq - t) / u) - 2.f;
  // (never above TQ_MAX8: a forced row - start value ACC_FORCE = 0x38000000, |dot| < 2^27 - passes every threshold a query can have)
  v = fminf(fmaxf(v, -1073741824.f), 805306368.f);
  return (int)v;
}

// sum over the 16 lanes of a DPP row (every lane gets it): two quad permutes, then half-row and row mirrors - 4 VALU instructions, no LDS
// (__shfl_xor compiles to ds_bpermute: 16 LDS round trips per step of this kernel)
__device__ __forceinline__ int row16_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror: the other quad of the same 8 lanes
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: the other half of the row
  return v;
}

// Seed sample of the MFMA engine: sample entry i is table row i for the first `head` entries (the head of the table), then
// rows spread evenly over the rest: head + ((i - head) * stride >> 32), stride = (rest rows / rest entries) in 32.32.
__host__ __device__ __forceinline__ u32 seed_row(u32 i, u32 head, u64 stride) {
  return i < head ? i : head + (u32)(((u64)(i - head) * stride) >> 32);
}

// metric epilogue on the raw accumulator (sum of squared differences for L2, dot otherwise)
__device__ __forceinline__ float finish_dist(int metric, float acc) {
  return metric == 0 ? acc : (metric == 1 ? 1.0f - acc : -acc);
}

// ------------------------------------------------------------------------------------------------
template <int KPL>
struct WaveTopK {
  u64 key[KPL];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int r = 0; r < KPL; ++r) key[r] = KEY_EMPTY;
  }
  // e-th smallest entry (0-based), wave-uniform
  __device__ __forceinline__ u64 entry(int e) const {
    u64 v = KEY_EMPTY;
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      u64 t = shfl64(key[r], e & 63);
      if ((e >> 6) == r) v = t;
    }
    return v;
  }
  // x must be wave-uniform.  Entries beyond KPL*64 fall off the end.
  __device__ __forceinline__ void insert(u64 x) {
    const int lane = lane_id();
    int pos = 0;
#pragma unroll
    for (int r = 0; r < KPL; ++r) pos += __popcll(__ballot(key[r] < x));
#pragma unroll
    for (int r = KPL - 1; r >= 0; --r) {
      u64 up = shfl_up64(key[r], 1);
      u64 carry = r > 0 ? shfl64(key[r - 1], 63) : 0ull;
      u64 prev = lane == 0 ? carry : up;
      const int e = r * 64 + lane;
      key[r] = e > pos ? prev : (e == pos ? x : key[r]);
    }
  }
  // insert unless an identical key (same row, same distance bits) is already present
  __device__ __forceinline__ void insert_unique(u64 x) {
    bool dup = false;
#pragma unroll
    for (int r = 0; r < KPL; ++r) dup |= (__ballot(key[r] == x) != 0ull);
    if (!dup) insert(x);
  }
  __device__ __forceinline__ void store(u64* dst, int k) const {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      const int e = r * 64 + lane;
      if (e < k) dst[e] = key[r];
    }
  }
  __device__ __forceinline__ void load(const u64* src, int k) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
      const int e = r * 64 + lane;
      key[r] = e < k ? src[e] : KEY_EMPTY;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Geometry of "G lanes per row": G = smallest power of two with G*4 >= dim (VEC4) or G >= dim (scalar),
// capped at 64.  RPW = 64 / G rows are evaluated by one wavefront at once.
__host__ __device__ inline int group_lanes(int64_t dim, bool vec4) {
  int64_t need = vec4 ? (dim + 3) / 4 : dim;
  int g = 1;
  while (g < 64 && g < need) g <<= 1;
  return g;
}

// Accumulate U rows x NQ queries.  rowp[u] points at the row for this lane's group (already clamped to
// a valid row); qs = LDS copy of the NQ queries, each dim floats (16-B aligned when VEC4).
// Returns raw accumulators reduced over the G lanes of the group (valid in every lane of the group).
// NL = 16-byte pieces of every row a lane has in flight at once (VEC4 only).  The loop over a row's pieces has a run-time trip
// count, so hipcc issues the U loads of one piece, waits for them, multiplies, and only then issues the next piece: at d = 768
// (three pieces per lane) a GATHER of U rows costs three dependent memory round trips.  NL = 3 issues all of them first; the
// fmas run in the same order either way, so the sums are bit-identical.  It costs 32 more VGPRs at U = 4 and every kernel that
// gathers rows is at its occupancy edge, so it is a lab knob, not the default (profiles/r3_row_pieces_in_flight_ab.txt): traversal
// with the 8-bit prefilter (-DEPS_TRV_NL=3) 1M x 768: T = 1 8.85 -> 8.03 ms, T = 4 7.26 -> 6.92 ms, but 10M x 768 T = 4 10.2 ->
// 10.4 ms (T = 1 11.3 -> 10.6); without the prefilter 13.0 -> 16.6 ms (4 -> 3 wavefronts per SIMD); the build's searches
// (Link 2.74 -> 2.88 s); the re-rank: no change.
template <int U, int NQ, bool VEC4, int NL = 1>
__device__ __forceinline__ void row_dists(const float* const (&rowp)[U], const float* qs, int64_t qstride, int dim,
                                          int metric, int G, float (&acc)[U][NQ]) {
  const int t = lane_id() & (G - 1);
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[u][q] = 0.f;
  if (VEC4 && NL > 1) {
    for (int c = t * 4; c < dim; c += G * 4 * NL) {
      float4 x[NL][U];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int cj = c + j * G * 4;
        if (cj < dim) {
#pragma unroll
          for (int u = 0; u < U; ++u) x[j][u] = *reinterpret_cast<const float4*>(rowp[u] + cj);
        }
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int cj = c + j * G * 4;
        if (cj < dim) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const float4 qv = *reinterpret_cast<const float4*>(qs + q * qstride + cj);
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (metric == 0) {
                const float a = x[j][u].x - qv.x, b = x[j][u].y - qv.y, c2 = x[j][u].z - qv.z, d2 = x[j][u].w - qv.w;
                acc[u][q] = fmaf(a, a, acc[u][q]);
                acc[u][q] = fmaf(b, b, acc[u][q]);
                acc[u][q] = fmaf(c2, c2, acc[u][q]);
                acc[u][q] = fmaf(d2, d2, acc[u][q]);
              } else {
                acc[u][q] = fmaf(x[j][u].x, qv.x, acc[u][q]);
                acc[u][q] = fmaf(x[j][u].y, qv.y, acc[u][q]);
                acc[u][q] = fmaf(x[j][u].z, qv.z, acc[u][q]);
                acc[u][q] = fmaf(x[j][u].w, qv.w, acc[u][q]);
              }
            }
          }
        }
      }
    }
  } else
  if (VEC4) {
    for (int c = t * 4; c < dim; c += G * 4) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const float4*>(rowp[u] + c);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 qv = *reinterpret_cast<const float4*>(qs + q * qstride + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (metric == 0) {
            const float a = x[u].x - qv.x, b = x[u].y - qv.y, c2 = x[u].z - qv.z, d2 = x[u].w - qv.w;
            acc[u][q] = fmaf(a, a, acc[u][q]);
            acc[u][q] = fmaf(b, b, acc[u][q]);
            acc[u][q] = fmaf(c2, c2, acc[u][q]);
            acc[u][q] = fmaf(d2, d2, acc[u][q]);
          } else {
            acc[u][q] = fmaf(x[u].x, qv.x, acc[u][q]);
            acc[u][q] = fmaf(x[u].y, qv.y, acc[u][q]);
            acc[u][q] = fmaf(x[u].z, qv.z, acc[u][q]);
            acc[u][q] = fmaf(x[u].w, qv.w, acc[u][q]);
          }
        }
      }
    }
  } else {
    for (int c = t; c < dim; c += G) {
      float x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = rowp[u][c];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float qv = qs[q * qstride + c];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (metric == 0) {
            const float a = x[u] - qv;
            acc[u][q] = fmaf(a, a, acc[u][q]);
          } else {
            acc[u][q] = fmaf(x[u], qv, acc[u][q]);
          }
        }
      }
    }
  }
  for (int o = G >> 1; o > 0; o >>= 1) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[u][q] += __shfl_xor(acc[u][q], o);
  }
}

}  // namespace eps
