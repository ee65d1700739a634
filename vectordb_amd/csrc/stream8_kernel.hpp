// Exact flat search of a HANDFUL of queries (<= 4 on v_dot4; r5: 5..16 on the matrix cores, k <= 64) in ONE pass over the 8-bit mirror (r4; BASELINE
// configs[1]: one query per call).
//
// What it replaces: for a few queries the staged MFMA chain of mfma_filter.hip is a GEMM-shaped answer to a GEMV-shaped problem - a
// single-query call was 9 dependent launches (query prep, seed pass, seed selection, seed re-rank, 3 x (filter stage + re-rank)), 0.33 ms
// of which only the last stage's 0.12 ms was the unavoidable pass over the mirror (profiles/r4_single_query_latency.txt).  Here the pass
// IS the call: every wavefront streams its share of the mirror (HBM-bound: d_pad8 bytes + 4 per row, v_dot4_i32_i8, no matrix cores),
// and what the stages were for - a pass threshold that tightens as the scan proceeds - lives in a 64-slot table per query in device
// memory that all wavefronts share:
//
//   G[q][0..64)  slot j = the largest accumulator (= approximate key, larger = closer) any visible row with hash(row) mod 64 = j has
//                shown so far (INT_MIN = empty slot), kept by one fire-and-forget atomicMax per offer - no lock, no compare-and-swap
//                loop.  The k largest slots are k DISTINCT rows, and whatever k rows they are, the k-th best EXACT distance of the
//                result is at most the largest upper bound among them: ub(acc) = C[q] - u acc + margin (the Cauchy-Schwarz margin of
//                stage_threshold8, which bounds |exact - approximate| in BOTH directions;
//                tests/test_bound_math.py::test_upper_bound_of_the_approximate_key).  With 64 slots for k <= 16 the k-th largest slot
//                ends within a few ranks of the k-th best row; r5: k = 17..64 use 128 slots (Stream8Args::slots; the k-th largest of S
//                hash buckets' maxima is the row of rank ~ S ln(S / (S - k)): rank 89 for k = 64 of 128, where 64 slots would give the
//                worst bucket's best row, rank ~300, and four times the candidates).  Every slot has its own 256 bytes: device-scope atomics on ONE address
//                are served one after the other, ~50 ns each on this machine (measured: the first versions of this kernel - a 16-entry
//                table in one cache line, offered to by 4096 wavefronts at start - spent 0.2 ms there), on 64 lines they overlap.
//   pass test    a row is a candidate iff acc >= stage_threshold8(ub(k-th largest slot)) - the same arithmetic every filter stage
//                uses, with the k-th best exact key replaced by that upper bound.  Slots only grow, a stale read only loosens the test.
//
// Candidates are appended as (acc, row) pairs to a PRIVATE list per wavefront (32 entries; an append through a shared counter is a
// returning device-scope atomic - microseconds during which that wavefront issues no loads - and the kernel ends with its slowest
// wavefront: the shared-counter version lost ~20 us of a 160 us pass to some 500 of them).  Early in the pass the table is loose and lets
// junk through (every wavefront first offers the BEST row of its first chunk to the empty table and only then starts testing, which bounds the
// junk to a few hundred entries); the selection step (below) drops it against the FINAL table without touching a
// row, and the ordinary re-rank kernel (flat_kernels.hip) computes the survivors' exact fp32 distances, applies the deleted bitset /
// int-column filter (calls with a compiled filter PROGRAM take the staged chain), and writes the caller-visible result.  Overflow of either list is reported through the re-rank's overflow counter and the
// caller repeats the batch on the staged chain.  Rows with a FORCED start value (mfma_filter.hip: ACC_FORCE) are always candidates and
// never enter the table.
#pragma once
#include "device_common.hpp"
#include "kernels.hpp"

namespace eps {

// kernel ablations (profiling; the answers are wrong) exist in lab builds only: in the product they compile away
#ifdef EPS_LAB
#define S8_ABLATE (a.ablate)
#else
#define S8_ABLATE 0
#endif


struct Stream8Args {
  const signed char* x8;   // [n_pad8][d_pad8]
  const int* acc0;         // [n_pad8] the rows' start values as the pass adds them: the table's own, or (r6) those with the call's per-row margins folded in
  const int* acc0_raw = nullptr;   // r6, folded calls only: the start values WITHOUT margins (see stream8_offer_value); null: acc0 carries none
  int64_t n;               // rows to scan
  int d_pad8;
  const signed char* q8;   // [>= nq][d_pad8] row-major
  const float* qstat;      // [>= nq][4]
  const float* scal;       // the mirror's maxima (HalfMirror::scal8)
  int nq, k, metric;
  float u, slack;
  int* G;                  // [nq][slots] slots, S8_SLOT_STRIDE ints apart
  int slots = 64;          // S8_SLOTS (k <= 16) | S8_SLOTS_WIDE (k = 17..64)
  u32* raw_cnt;            // [nq][waves]: entries every wavefront of the grid found (written once, when it ends)
  u64* raw;                // [nq][waves][S8_WAVE_CAP]: (acc << 32) | row - a private list per wavefront: no atomic, nothing to wait for
  int waves;               // wavefronts of the grid (<= S8_MAX_WAVES)
  // r5, PREP instantiations: the launch quantises its queries itself (one wavefront per query, every workgroup for itself: 3 KB from L2) -
  // no query-prep launch in front of the pass.  Workgroup 0 leaves the queries' constants in qstat_out for the re-rank launch.
  const float* qf32 = nullptr;   // [nq][dim] the queries as given
  const float* mu = nullptr;     // [d_pad8] the grid's centre
  float* qstat_out = nullptr;    // = qstat, writable
  int dim = 0;
  float step = 1.f, inv_step = 1.f;
  int ablate;              // lab (EPS_S8_ABLATE; results are wrong): 1 = no table, no test - the bare stream + dot products; 2 = no periodic
                           // re-read of the table; 4 = no start-up (offer, barriers, first read), tests against "nothing passes"; 8 = the
                           // start-up as it is, then tests against "nothing passes"
  FilterSpec f;
};

constexpr int S8_FORCE_LIMIT = 0x30000000;   // (= TQ_MAX8: accumulators at or above it belong to forced rows)
constexpr int S8_EMPTY = -2147483647 - 1;
constexpr int S8_SLOTS = 64, S8_SLOTS_WIDE = 128, S8_SLOT_STRIDE = 64;   // (stride in 4-byte words)
constexpr int S8_MAX_K = 64;   // k of a one-pass call: <= 16 with 64 slots per query, 17..64 with 128 (r5)
constexpr int S8_MAX_Q = 32;   // queries of a one-pass call: <= 4 on v_dot4 (stream8_kernel), 5..16 on the matrix cores (stream8m_kernel, r5), 17..32 on two column blocks (r6)
constexpr int S8_TABLE_WORDS = S8_MAX_Q * S8_SLOTS_WIDE * S8_SLOT_STRIDE;
constexpr int S8_WAVE_CAP = 32, S8_MAX_WAVES = 8192;   // (32: a run of identical rows - 16 of them in one chunk - must not fill a list by itself)

// upper bound of the exact fp32 distance of a row whose accumulator is `acc` (see the header; mirrors stage_threshold8 term by term)
__device__ __forceinline__ float stream8_ub(int acc, const float* qs, const float* sc, int metric, float u, float slack) {
  const float qn2 = qs[0], nqc = qs[1], eq = qs[2], Cq = qs[3];
  const float e1max = sc[0], nxhmax = sc[1], xnmax = sc[2], rmax = sc[4], mun = sc[5], xcmax = sc[6];
  const float s = metric == 0 ? 2.f : 1.f;
  const float margin = s * (nqc * e1max + eq * nxhmax);
  const float dapx = Cq - u * (float)acc;
  const float scale = metric == 0 ? fabsf(dapx) + margin + 2.f * fabsf(Cq) + 2.f * rmax
                                  : fabsf(dapx) + margin + 1.f + sqrtf(qn2) * (sqrtf(xnmax) + mun) + mun * xcmax + fabsf(Cq) + rmax;
  // (the start value is ceil(-R/u) + 1: up to two units above the real quotient; (float)acc and the product round once each)
  return dapx + margin + 2.f * slack * scale + 4.f * u;
}

// pass threshold of a query from its table G: the k-th largest slot (one wavefront, one slot per lane - two with 128 slots; every lane gets
// the result).  Slots are ordered by (value descending, slot index ascending): exactly one of them has rank k - 1.
__device__ __forceinline__ int stream8_threshold_of(const int* G, int k, const float* qs, const float* sc, int metric, float u, float slack, int lane, int& gkth,
                                                    int slots = S8_SLOTS, int* filled = nullptr) {
  const int v = __hip_atomic_load(G + lane * S8_SLOT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int kth;
  if (filled) *filled = __popcll(__ballot(v != S8_EMPTY));   // (the start-up wait only; folds away elsewhere)
  if (slots == S8_SLOTS) {
    int rank = 0;   // slots that order before this lane's (larger value, or equal and lower lane)
    for (int i = 0; i < 64; ++i) {
      const int w = __builtin_amdgcn_readlane(v, i);
      rank += (w > v || (w == v && i < lane)) ? 1 : 0;
    }
    const unsigned long long m = __ballot(rank == k - 1);
    kth = __builtin_amdgcn_readlane(v, __ffsll((long long)m) - 1);
  } else {   // 128 slots: lane l holds slots l and l + 64
    const int v1 = __hip_atomic_load(G + (lane + 64) * S8_SLOT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (filled) *filled += __popcll(__ballot(v1 != S8_EMPTY));
    int r0 = 0, r1 = 0;
    for (int i = 0; i < 64; ++i) {
      const int w0 = __builtin_amdgcn_readlane(v, i), w1 = __builtin_amdgcn_readlane(v1, i);
      r0 += ((w0 > v || (w0 == v && i < lane)) ? 1 : 0) + (w1 > v ? 1 : 0);                        // (slot i + 64 never precedes slot `lane` on a tie)
      r1 += (w0 >= v1 ? 1 : 0) + ((w1 > v1 || (w1 == v1 && i < lane)) ? 1 : 0);                    // (slot i always precedes slot lane + 64 on a tie)
    }
    const unsigned long long m0 = __ballot(r0 == k - 1), m1 = __ballot(r1 == k - 1);
    kth = m0 ? __builtin_amdgcn_readlane(v, __ffsll((long long)m0) - 1) : __builtin_amdgcn_readlane(v1, m1 ? __ffsll((long long)m1) - 1 : 63);   // (1 <= k <= 128: one of the two is set)
  }
  gkth = kth;
  if (kth == S8_EMPTY) return -(1 << 30);   // fewer than k slots filled so far: everything passes
  return stage_threshold8(stream8_ub(kth, qs, sc, metric, u, slack), qs, sc, metric, u, slack, 0);
}

// The start-up wait (both kernels below).  A workgroup reads its first thresholds once the table holds enough of the wavefronts' first
// offers.  "Enough" was k filled slots until r5 - but the k-th largest of EXACTLY k filled slots is the WORST row offered so far, and once in
// a few thousand calls that is a below-median row: a workgroup that read the table at that moment let most of its rows through, and on a
// small table (90 000 rows = 3 chunks per wavefront, no refresh in time) a 32-entry list overflowed and the staged chain had to repeat the
// call (found by scripts/lab/one_pass_small_table_stress.py: 1 call in 5040, "37 in one wavefront's list of 32").  Now: max(2 k, k + 8) filled slots (capped at
// 7/8 of the table: a third of the rows visible leaves a slot or two of 128 empty for good) - the k-th largest of twice as many offers is a
// median offer, not the worst - and when k are there but the rest does not come (a filter that leaves few rows visible) three more looks,
// then the wavefront takes what there is.
constexpr int S8_STARTUP_GRACE = 3;
__device__ __forceinline__ int stream8_startup_need(int k, int slots) {
  const int want = 2 * k > k + 8 ? 2 * k : k + 8;
  const int most = slots - slots / 8;
  return want < most ? want : most;
}

// one query, one wavefront: the arithmetic of query_prep8_kernel (mfma_filter.hip) term by term, the bytes to `dst`, the four constants to `qs`
__device__ __forceinline__ void stream8_prep_query(const float* src, int dim, int d_pad8, const float* mu, float step, float inv_step, int metric, int lane,
                                                   signed char* dst, float* qs, float* qs2) {
  float s2 = 0.f, e2 = 0.f, c2 = 0.f, qm = 0.f;
  for (int c = lane * 4; c < d_pad8; c += 256) {
    u32 packed = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e < dim) {
        const float xv = src[c + e];
        const float m = mu[c + e];
        const float dx = xv - m;
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
  }
  for (int o = 32; o > 0; o >>= 1) {
    s2 += __shfl_xor(s2, o);
    e2 += __shfl_xor(e2, o);
    c2 += __shfl_xor(c2, o);
    qm += __shfl_xor(qm, o);
  }
  if (lane == 0) {
    const float nqc = sqrtf(c2) * 1.000001f, eqc = sqrtf(e2) * 1.00001f + 1.2e-7f * sqrtf(c2);
    const float v3 = metric == 0 ? c2 : (metric == 1 ? 1.f - qm : -qm);
    qs[0] = s2;
    qs[1] = nqc;
    qs[2] = eqc;
    qs[3] = v3;
    if (qs2) {
      qs2[0] = s2;
      qs2[1] = nqc;
      qs2[2] = eqc;
      qs2[3] = v3;
    }
  }
}

// r6: one-pass calls on tables whose margins are folded per batch (rows with a clamped value carry their own residual: large rotated tables,
// tables with outliers).  The pass then adds acc0 = raw + m (m = the row's margin for this call's queries, fold8_kernel) and tests
// `dot + raw + m >= T` against MARGIN-FREE thresholds - the staged chain's folded test.  What a slot must hold is a number whose upper bound
// ub(.) = C[q] - u (.) holds for the ROW it came from: exact <= C - u (dot + raw) + u m = C - u (dot + raw - m), i.e. the accumulator minus
// TWICE the row's margin.  Offers are rare (a lane that beat the k-th slot): the two start values are read again here, not kept per chunk.
__device__ __forceinline__ int stream8_offer_value(const Stream8Args& a, int acc, int64_t row) {
  return a.acc0_raw ? acc - 2 * (a.acc0[row] - a.acc0_raw[row]) : acc;
}
__device__ __forceinline__ void stream8_offer(const Stream8Args& a, int q, int acc, u32 row) {   // (one lane; rare)
  // only rows the result may contain bound it.  The deleted bitset and the `int column <op> constant` filter are tested here (straight
  // code); calls with a filter PROGRAM take the staged chain (its evaluator in this kernel means a function call, i.e. scratch memory
  // for every wavefront of the pass)
  FilterSpec f = a.f;
  f.prog = nullptr;
  if (!row_visible(f, row)) return;
  const u32 slot = ((row * 2654435761u) >> 12) & (u32)(a.slots - 1);
  (void)__hip_atomic_fetch_max(a.G + (q * a.slots + (int)slot) * S8_SLOT_STRIDE, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// PIECES = d_pad8 / 256: a row is PIECES x 256 bytes, 16 lanes x 16 bytes each; four rows per wavefront and step, U steps in flight.
// The table is READ by one wavefront per workgroup, every fourth iteration, and handed to the other three through LDS: read by every
// wavefront in every iteration (4096 cache-bypassing loads of one 64-byte line per round) the loads queued up at that line's memory
// channel for ~20 us per iteration - the first version of this kernel ran at 2.1 TB/s because of it.
// WIDE: the 128-slot table of k = 17..64 (a compile-time choice here: the second slot per lane costs the refresh ~10 registers, which the
// 3-4-query forms do not have - those calls run stream8m_kernel, where the slot count is a run-time value)
template <int PIECES, int NQ, bool PREP, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void stream8_kernel(Stream8Args a) {
  static_assert(!WIDE || (NQ <= 2 && !PREP), "128 slots: 1-2 queries behind the prep launch");
  constexpr int SLOTS = WIDE ? S8_SLOTS_WIDE : S8_SLOTS;
  constexpr int U = PIECES <= 3 ? 4 : 2;
  constexpr int CH = 4 * U;   // rows per wavefront and iteration
  __shared__ int T_s[4], gkth_s[4];   // (this kernel: <= 4 queries)
  __shared__ __attribute__((aligned(16))) signed char q8_s[PREP ? NQ * PIECES * 256 : 16];
  __shared__ float qstat_s[16];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, t = lane & 15;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t W = (int64_t)gridDim.x * 4;

  int4 qv[NQ][PIECES];
  if (!PREP) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int p = 0; p < PIECES; ++p)
        qv[q][p] = q < a.nq ? *reinterpret_cast<const int4*>(a.q8 + (int64_t)q * a.d_pad8 + p * 256 + t * 16) : make_int4(0, 0, 0, 0);
  }

  // (PREP) the queries' bytes and constants, computed by the workgroup for itself; where registers allow (<= 2 queries) AFTER the first loads
  // have been issued, so that the quantisation runs under their latency
  static_assert(!PREP || NQ <= 2, "3-4 queries: the prep launch (registers)");
  constexpr bool PREP_LATE = true;
  auto prep = [&]() __attribute__((always_inline)) {
    if (wave < NQ && wave < a.nq)
      stream8_prep_query(a.qf32 + (int64_t)wave * a.dim, a.dim, PIECES * 256, a.mu, a.step, a.inv_step, a.metric, lane, q8_s + wave * (PIECES * 256), qstat_s + wave * 4,
                         blockIdx.x == 0 ? a.qstat_out + wave * 4 : nullptr);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int p = 0; p < PIECES; ++p)
        qv[q][p] = q < a.nq ? *reinterpret_cast<const int4*>(q8_s + q * (PIECES * 256) + p * 256 + t * 16) : make_int4(0, 0, 0, 0);
  };
  if (PREP && !PREP_LATE) prep();

  auto dots = [&](const int4 (&xv)[PIECES], int a0v, int (&out)[NQ]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      int d = 0;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        d = __builtin_amdgcn_sdot4(xv[p].x, qv[q][p].x, d, false);
        d = __builtin_amdgcn_sdot4(xv[p].y, qv[q][p].y, d, false);
        d = __builtin_amdgcn_sdot4(xv[p].z, qv[q][p].z, d, false);
        d = __builtin_amdgcn_sdot4(xv[p].w, qv[q][p].w, d, false);
      }
      out[q] = row16_sum(d) + a0v;
    }
  };
  auto refresh = [&]() __attribute__((always_inline)) {   // (one wavefront: the table -> this workgroup's thresholds)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q < a.nq) {
        int gm;
        float qs[4];   // (by value: a pointer that may be LDS or global is a flat pointer)
#pragma unroll
        for (int i = 0; i < 4; ++i) qs[i] = PREP ? qstat_s[q * 4 + i] : a.qstat[q * 4 + i];
        const int Tq = stream8_threshold_of(a.G + q * SLOTS * S8_SLOT_STRIDE, a.k, qs, a.scal, a.metric, a.u, a.slack, lane, gm, SLOTS);
        if (lane == 0) {
          T_s[q] = Tq;
          gkth_s[q] = gm;
        }
      }
    }
  };

  auto load_chunk = [&](int64_t base, int4 (&xv)[U][PIECES], int (&a0)[U]) __attribute__((always_inline)) {
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
      int64_t r = base + uu * 4 + g;
      r = r < a.n ? r : a.n - 1;
      const signed char* p0 = a.x8 + r * a.d_pad8 + t * 16;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) xv[uu][p] = *reinterpret_cast<const int4*>(p0 + p * 256);
      a0[uu] = a.acc0[r];
    }
  };
  u32 mine[NQ];   // entries of this wavefront's private candidate lists (wave-uniform)
#pragma unroll
  for (int q = 0; q < NQ; ++q) mine[q] = 0;
  // candidates of one chunk: appended to the wavefront's own list, offered to the table where they beat its k-th slot
  auto test_chunk = [&](int64_t base, const int (&acc)[U][NQ]) __attribute__((always_inline)) {
    if (S8_ABLATE & 1) {   // (keeps the loads and the arithmetic alive)
      int x = 0;
#pragma unroll
      for (int uu = 0; uu < U; ++uu)
#pragma unroll
        for (int q = 0; q < NQ; ++q) x ^= acc[uu][q];
      if (x == 0x7fffffff && base == -5) a.raw[0] = 1;
      return;
    }
    int T[NQ], gkth[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      T[q] = q < a.nq ? *reinterpret_cast<volatile int*>(&T_s[q]) : 2147483647;
      gkth[q] = q < a.nq ? *reinterpret_cast<volatile int*>(&gkth_s[q]) : 2147483647;
    }
    bool any = false;
#pragma unroll
    for (int uu = 0; uu < U; ++uu)
#pragma unroll
      for (int q = 0; q < NQ; ++q) any |= (acc[uu][q] >= T[q] || acc[uu][q] > gkth[q]) && base + uu * 4 + g < a.n;
    if (!__ballot(any && t == 0)) return;   // (the common step ends here)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q >= a.nq) continue;
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        const int64_t row = base + uu * 4 + g;
        const bool live = t == 0 && row < a.n;
        const int v = acc[uu][q];
        const bool pass = live && v >= T[q];
        const unsigned long long mask = __ballot(pass);
        if (mask) {
          if (pass) {
            const u32 slot = mine[q] + (u32)__popcll(mask & ((1ull << lane) - 1ull));
            if (slot < (u32)S8_WAVE_CAP) a.raw[((int64_t)q * a.waves + wid) * S8_WAVE_CAP + slot] = ((u64)(u32)v << 32) | (u32)row;
          }
          mine[q] += (u32)__popcll(mask);
        }
        // the table (a row offered twice lands in the same slot: still distinct rows)
        if (live && v > gkth[q] && v < S8_FORCE_LIMIT) stream8_offer(a, q, stream8_offer_value(a, v, row), (u32)row);
      }
    }
  };

  const int64_t first = wid * CH, stride = W * CH;
  int4 xa[U][PIECES], xb[U][PIECES];
  int a0a[U], a0b[U];
  int acc[U][NQ];
  if (first < a.n) load_chunk(first, xa, a0a);
  if (first + stride < a.n) load_chunk(first + stride, xb, a0b);
  if (PREP && PREP_LATE) prep();
  // ---- the first chunk feeds the empty table before anything is tested: ONE row per wavefront (the chunks are spread over the whole
  // mirror) is offered, the workgroup's first thresholds are read from what has arrived, and only then the chunk is tested - so the table
  // holds the best of a few thousand rows before the first candidate is appended
  if (first < a.n) {
#pragma unroll
    for (int uu = 0; uu < U; ++uu) dots(xa[uu], a0a[uu], acc[uu]);
    // the best of the wavefront's 4 U rows (per query): one offer per wavefront, of a row that beat 15 others
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q >= a.nq || (S8_ABLATE & 5)) continue;
      int bv = S8_EMPTY, br = 0;
#pragma unroll
      for (int uu = 0; uu < U; ++uu) {
        const int v = acc[uu][q];
        const bool better = first + uu * 4 + g < a.n && v < S8_FORCE_LIMIT && v > bv;
        br = better ? uu * 4 + g : br;
        bv = better ? v : bv;
      }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const int ov = __shfl_xor(bv, o), orow = __shfl_xor(br, o);
        const bool better = ov > bv;
        br = better ? orow : br;
        bv = better ? ov : bv;
      }
      if (lane == 0 && bv != S8_EMPTY) stream8_offer(a, q, stream8_offer_value(a, bv, first + br), (u32)(first + br));
    }
  }
  if (!(S8_ABLATE & 5)) {
    __syncthreads();
    // (the offers are fire-and-forget atomics: a workgroup that gets here before k slots have been filled by anyone would test its
    // first chunk against "everything passes"; it waits for them, a few microseconds at most - bounded, then it takes what there is)
    if (wave == 0) {
      const int need = stream8_startup_need(a.k, SLOTS);
      int grace = 0;
      for (int spin = 0; spin < 48; ++spin) {
        int state = 2;   // 0: a query has fewer than k slots filled, 1: k but not `need`, 2: every query has what it needs
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q < a.nq) {
            int gm, filled;
            float qs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qs[i] = PREP ? qstat_s[q * 4 + i] : a.qstat[q * 4 + i];
            const int Tq = stream8_threshold_of(a.G + q * SLOTS * S8_SLOT_STRIDE, a.k, qs, a.scal, a.metric, a.u, a.slack, lane, gm, SLOTS, &filled);
            if (lane == 0) {
              T_s[q] = Tq;
              gkth_s[q] = gm;
            }
            const int sq = gm == S8_EMPTY ? 0 : (filled >= need ? 2 : 1);
            state = sq < state ? sq : state;
          }
        }
        if (state == 2 || (state == 1 && ++grace > S8_STARTUP_GRACE)) break;
        __builtin_amdgcn_s_sleep(16);
      }
    }
    __syncthreads();
  }
  if (S8_ABLATE & 12) {
    __syncthreads();
    if (threadIdx.x < 4) T_s[threadIdx.x] = gkth_s[threadIdx.x] = 2147483647;
    __syncthreads();
  }
  if (first < a.n) test_chunk(first, acc);
  // ---- the stream, two chunks in flight per wavefront: the loads of chunk i + 2 are issued before chunk i + 1 is multiplied
  int it = 1;
  for (int64_t base = first + stride; base < a.n; base += 2 * stride, it += 2) {
    if (base + stride < a.n) load_chunk(base + stride, xa, a0a);
    // the thresholds are read again after 2, 4, 8 chunks (the table tightens fastest at the start: a stale threshold there is what lets
    // junk into the lists) and then every 8, the four wavefronts in turn (the refresher waits for 64 cache-bypassing loads)
    if ((it == 1 || it == 3 || (it & 7) == 7) && wave == (((it >> 1) + (it >> 3)) & 3) && !(S8_ABLATE & 15)) refresh();   // (it = 1, 3, 7, 15, 23, 31, ... -> wavefront 0, 1, 3, 0, 1, 2, ...: in turn - until r4 the steady state always picked wavefront 3)
#pragma unroll
    for (int uu = 0; uu < U; ++uu) dots(xb[uu], a0b[uu], acc[uu]);
    test_chunk(base, acc);
    if (base + stride >= a.n) break;
    if (base + 2 * stride < a.n) load_chunk(base + 2 * stride, xb, a0b);
#pragma unroll
    for (int uu = 0; uu < U; ++uu) dots(xa[uu], a0a[uu], acc[uu]);
    test_chunk(base + stride, acc);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (q < a.nq && lane == 0) a.raw_cnt[(int64_t)q * a.waves + wid] = mine[q];
}

// ---- 5..16 queries in one pass (r5).  Four queries' slices are what a lane's registers hold next to two chunks in flight; beyond that the
// v_dot4 form falls off (r4: slices in LDS 0.64 ms at 8 queries, two passes side by side 0.38; r5: eight slices in registers at one
// wavefront per SIMD 0.39-0.46 - the staged chain's 0.36 ms was the best of them).  A handful of queries x 16 rows IS a small matrix
// product, and v_mfma_i32_16x16x64_i8 computes 16 rows x 16 query columns x 64 bytes of K in one instruction - 12 per 16-row block
// at 768 bytes per row, ~11 us of matrix pipe for a 1M-row table - so the pass stays what the single-query pass is: HBM-bound.
//   operands  lane l = (column c = l % 16, K group g = l / 16) supplies bytes [64 j + 16 g, +16) of ROW base + c (A) and of QUERY c (B, resident:
//             12 x 4 registers; queries >= nq are zero rows of q8) for MFMA j; the K order is the same on both sides, so the sum is the plain
//             dot product.  A block's 12 loads cover 16 consecutive rows = 12 KB of the mirror completely (64 bytes per row and instruction).
//   results   lane l holds rows base + 4 g + {0, 1, 2, 3} of column c, started from acc0[row]: one compare per value against the column's
//             threshold (T_s[c]), the common block ends at one ballot.
// Table, thresholds, private candidate lists and the selection + re-rank behind the pass are stream8_kernel's (above), per query.
// NB (r6): 16-query column blocks per call - 1: 5..16 queries; 2: 17..32 queries, the row operand of a block feeds both column blocks (24 MFMAs per
// 16 rows at d = 768: ~25 us of matrix pipe per 1M rows, the pass stays HBM-bound; the queries' operands 96 registers, two row blocks in flight 96).
// (NB = 2 with both column blocks' operands in registers spills a few dwords at 256 - and ANY scratch costs every wavefront of the launch its
// set-up; with ONE workgroup per CU - 512 registers - the pass ran at half its rate, 17 queries 0.300 ms against 0.217 for 16.  So the second column
// block's operands live in LDS, [K-step][lane] x 16 bytes: conflict-free ds_read_b128, 12 per row block and wavefront, issued ahead of the MFMAs that use them)
template <int PIECES, int NB = 1>
__global__ __launch_bounds__(256, 2) void stream8m_kernel(Stream8Args a) {
  constexpr int KS = PIECES * 4;   // MFMAs per 16-row block and column block
  __shared__ int T_s[S8_MAX_Q], gkth_s[S8_MAX_Q];
  __shared__ u32 mine_s[4][S8_MAX_Q];   // entries of each wavefront's private list, per query
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t W = (int64_t)gridDim.x * 4;
  if (lane < S8_MAX_Q) mine_s[wave][lane] = 0;

  constexpr bool Q0_LDS = NB == 2 && PIECES == 4;   // (1024-byte rows: 16 + 16 + 16 operand registers per K-step leave no room for either column block's queries)
  i32x4 qv[Q0_LDS ? 1 : KS];
  __shared__ i32x4 q2_s[NB == 2 ? KS : 1][NB == 2 ? 64 : 1];   // (NB = 2: the second column block's operands, the same in every wavefront)
  __shared__ i32x4 q1_s[Q0_LDS ? KS : 1][Q0_LDS ? 64 : 1];
  if (!Q0_LDS) {
#pragma unroll
    for (int j = 0; j < KS; ++j) qv[j] = *reinterpret_cast<const i32x4*>(a.q8 + (int64_t)col * a.d_pad8 + j * 64 + kg * 16);
  }
  if (NB == 2) {
    for (int j = wave; j < KS; j += 4) {
      q2_s[j][lane] = *reinterpret_cast<const i32x4*>(a.q8 + (int64_t)(col + 16) * a.d_pad8 + j * 64 + kg * 16);
      if (Q0_LDS) q1_s[j][lane] = *reinterpret_cast<const i32x4*>(a.q8 + (int64_t)col * a.d_pad8 + j * 64 + kg * 16);
    }
    __syncthreads();
  }

  auto load_block = [&](int64_t base, i32x4 (&xv)[KS], i32x4& c0) __attribute__((always_inline)) {
    int64_t r = base + col;
    r = r < a.n ? r : a.n - 1;
    const signed char* p0 = a.x8 + r * a.d_pad8 + kg * 16;
#pragma unroll
    for (int j = 0; j < KS; ++j) xv[j] = *reinterpret_cast<const i32x4*>(p0 + j * 64);
    c0 = *reinterpret_cast<const i32x4*>(a.acc0 + base + 4 * kg);   // (acc0 has n_pad8 >= n rounded up to 256 entries; base is a multiple of 16)
  };
  auto dots = [&](const i32x4 (&xv)[KS], const i32x4& c0, i32x4 (&acc)[NB]) __attribute__((always_inline)) {
    acc[0] = c0;
#pragma unroll
    for (int j = 0; j < KS; ++j) acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv[j], Q0_LDS ? q1_s[j][lane] : qv[Q0_LDS ? 0 : j], acc[0], 0, 0, 0);
    if (NB == 2) {
      acc[NB - 1] = c0;
#pragma unroll
      for (int j = 0; j < KS; ++j) acc[NB - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xv[j], q2_s[j][lane], acc[NB - 1], 0, 0, 0);
    }
  };
  // the table -> thresholds: wavefront w serves the queries q = w, w + 4, ... (a refresh of 16 queries by one wavefront would cost it ~2 k instructions)
  auto refresh = [&]() __attribute__((always_inline)) {
    for (int q = wave; q < a.nq; q += 4) {
      int gm;
      const int Tq = stream8_threshold_of(a.G + q * a.slots * S8_SLOT_STRIDE, a.k, a.qstat + q * 4, a.scal, a.metric, a.u, a.slack, lane, gm, a.slots);
      if (lane == 0) {
        *reinterpret_cast<volatile int*>(&T_s[q]) = Tq;
        *reinterpret_cast<volatile int*>(&gkth_s[q]) = gm;
      }
    }
  };
  auto test_block = [&](int64_t base, const i32x4 (&accs)[NB]) __attribute__((always_inline)) {
    const int64_t row0 = base + 4 * kg;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int qc = col + 16 * b;
      const i32x4 acc = accs[b];
      const int T = qc < a.nq ? *reinterpret_cast<volatile int*>(&T_s[qc]) : 2147483647;
      const int gk = qc < a.nq ? *reinterpret_cast<volatile int*>(&gkth_s[qc]) : 2147483647;
      bool any = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) any |= (acc[r] >= T || acc[r] > gk) && row0 + r < a.n;
      if (!__ballot(any)) continue;   // (the common block ends here)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + r;
        const int v = acc[r];
        const bool live = row < a.n && qc < a.nq;
        unsigned long long m = __ballot(live && v >= T);
        while (m) {   // (rare) one entry at a time: the lane it belongs to appends it to ITS query's list of this wavefront
          const int i = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int q = (i & 15) + 16 * b;
          const u32 slot = *reinterpret_cast<volatile u32*>(&mine_s[wave][q]);
          if (lane == i) {
            if (slot < (u32)S8_WAVE_CAP) a.raw[((int64_t)q * a.waves + wid) * S8_WAVE_CAP + slot] = ((u64)(u32)v << 32) | (u32)row;
            *reinterpret_cast<volatile u32*>(&mine_s[wave][q]) = slot + 1;
          }
        }
        if (live && v > gk && v < S8_FORCE_LIMIT) stream8_offer(a, qc, stream8_offer_value(a, v, row), (u32)row);
      }
    }
  };

  constexpr int CH = 16;
  const int64_t first = wid * CH, stride = W * CH;
  i32x4 xa[KS], xb[KS], ca, cb;
  if (first < a.n) load_block(first, xa, ca);
  if (first + stride < a.n) load_block(first + stride, xb, cb);
  i32x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = i32x4{S8_EMPTY, S8_EMPTY, S8_EMPTY, S8_EMPTY};
  // ---- the first block feeds the empty table before anything is tested: per query the best of the wavefront's 16 rows is offered
  if (first < a.n) {
    dots(xa, ca, acc);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      int bv = S8_EMPTY, br = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool better = first + 4 * kg + r < a.n && acc[b][r] < S8_FORCE_LIMIT && acc[b][r] > bv;
        br = better ? 4 * kg + r : br;
        bv = better ? acc[b][r] : bv;
      }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const int ov = __shfl_xor(bv, o), orow = __shfl_xor(br, o);
        const bool better = ov > bv;
        br = better ? orow : br;
        bv = better ? ov : bv;
      }
      const int qc = col + 16 * b;
      if (kg == 0 && qc < a.nq && bv != S8_EMPTY) stream8_offer(a, qc, stream8_offer_value(a, bv, first + br), (u32)(first + br));
    }
  }
  __syncthreads();
  // (as in stream8_kernel: a workgroup that gets here before k slots of a query have been filled by anyone waits for them - bounded)
  {
    const int need = stream8_startup_need(a.k, a.slots);
    int grace = 0;
    for (int spin = 0; spin < 48; ++spin) {
      int state = 2;   // (of this wavefront's queries: see stream8_kernel)
      for (int q = wave; q < a.nq; q += 4) {
        int gm, filled;
        const int Tq = stream8_threshold_of(a.G + q * a.slots * S8_SLOT_STRIDE, a.k, a.qstat + q * 4, a.scal, a.metric, a.u, a.slack, lane, gm, a.slots, &filled);
        if (lane == 0) {
          *reinterpret_cast<volatile int*>(&T_s[q]) = Tq;
          *reinterpret_cast<volatile int*>(&gkth_s[q]) = gm;
        }
        const int sq = gm == S8_EMPTY ? 0 : (filled >= need ? 2 : 1);
        state = sq < state ? sq : state;
      }
      if (state == 2 || (state == 1 && ++grace > S8_STARTUP_GRACE)) break;
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  if (first < a.n) test_block(first, acc);
  // ---- the stream, two blocks in flight per wavefront
  int it = 1;
  for (int64_t base = first + stride; base < a.n; base += 2 * stride, it += 2) {
    if (base + stride < a.n) load_block(base + stride, xa, ca);
    if (it == 1 || it == 3 || (it & 7) == 7) refresh();   // (every wavefront its own queries; a stale threshold only loosens the test)
    dots(xb, cb, acc);
    test_block(base, acc);
    if (base + stride >= a.n) break;
    if (base + 2 * stride < a.n) load_block(base + 2 * stride, xb, cb);
    dots(xa, ca, acc);
    test_block(base + stride, acc);
  }
  if (lane < a.nq) a.raw_cnt[(int64_t)lane * a.waves + wid] = *reinterpret_cast<volatile u32*>(&mine_s[wave][lane]);
}

// The selection step - candidates that still pass against the FINAL table go to the re-rank's list, the rest (the junk of the first
// microseconds) is dropped without touching a row - runs as the prologue of the re-rank kernel (flat_kernels.hip, RerankArgs::s8_G): one
// launch less in a chain of four.  `lists` = the wavefronts' private lists of this query, `counts` their lengths.
template <int NT>
__device__ __forceinline__ void stream8_select(int T, const u32* counts, const u64* lists, int waves, u32* cand, int cap, u32* kept, u32* lost) {
  for (int w0 = 0; w0 < waves; w0 += 4 * NT) {   // (four wavefronts' counts per thread first: independent loads)
    u32 have[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = w0 + (int)threadIdx.x + j * NT;
      have[j] = w < waves ? counts[w] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!have[j]) continue;
      const int w = w0 + (int)threadIdx.x + j * NT;
      if (have[j] > (u32)S8_WAVE_CAP) *lost = 1;
      const u32 n_raw = have[j] < (u32)S8_WAVE_CAP ? have[j] : (u32)S8_WAVE_CAP;
      for (u32 i = 0; i < n_raw; ++i) {
        const u64 e = lists[(int64_t)w * S8_WAVE_CAP + i];
        if ((int)(u32)(e >> 32) >= T) {
          const u32 slot = atomicAdd(kept, 1u);
          if (slot < (u32)cap) cand[slot] = (u32)e;
        }
      }
    }
  }
}

}  // namespace eps
