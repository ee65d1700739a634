// The batched-flat-scan filter kernels (v3, v5, v7; v1 and v2 live on only in profiles/r1_mfma_ablation.txt) of mfma_filter.hip, in a header so that the kernel lab
// (scripts/lab/mfma_lab.hip) can time experimental variants beside them.  See mfma_filter.hip for the method.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "device_common.hpp"

namespace eps {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 64;                        // K-step of every kernel generation
constexpr int ROWPAD = 256;                   // mirror rows are padded to this

// ------------------------------------------------------------------------------------------------ filter kernel
struct FilterArgs {
  const _Float16* xh;   // [n_pad][d_pad]
  const _Float16* qh;   // [b_pad][d_pad]
  const _Float16* qf;   // fragment-major copy of qh: [b_pad/32][d_pad/16][64 lanes][8] (v5: query operand straight to VGPRs)
  const float* base;    // [n_pad]
  const float* base_s;  // [n_pad] base / s (v5: accumulators are initialised straight from it)
  int dense;            // v7 seed pass: every row is a candidate - keys go to slot (row - first row), no test, no atomics
  u32* group_sync;      // v7: one arrival counter per group of workgroups that share row tiles (zeroed per launch), or null
  int sync_shift;       // v7: the group meets before every 2^sync_shift-th tile
  const float* T;       // [b_pad]
  int d_pad;
  int tiles_q;          // b_pad / BN
  int64_t tile0;        // first row tile of this stage
  int64_t ntiles;       // row tiles in this stage
  int64_t row_hi;       // rows >= row_hi are not reported
  int64_t nq;
  float s;              // -2 (L2) or -1
  float inv_s;          // 1 / s (v7 reads it as a kernel argument: as a value computed in the kernel it is one more VGPR live across the tile loop)
  u32* cand;
  u64* cand_keys;       // approx mode: (approx dist, row) keys instead of row ids
  const float* qstat;   // [b_pad][4] (approx mode: |q|^2 to turn keys into distances)
  int metric;
  u32* cnt;
  int cap;
  unsigned long long* prof;   // lab builds (-DEPS_V7_PROF) only: [4] shader-clock sums over all wavefronts: tile head, K loop, epilogue, tiles
  int ablate;           // profiling only (EPS_MFMA_ABLATE): v1: bit0 skip staging loads, bit1 skip MFMAs, bit2 skip LDS
                        // fragment reads; v3: bit3 skip the query-operand DMA, bit4 skip the row-operand DMA
};

// max of the 16 accumulators a lane holds of one 32 x 32 block: 8 x v_max3_f32 in ONE asm statement (chained fmaxf costs 10
// instructions - hipcc canonicalises operands it cannot prove quiet - and separate asm statements get an s_nop each)
__device__ __forceinline__ float max16f(const f32x16& v) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3\n\t"
      "v_max3_f32 %0, %0, %4, %5\n\t"
      "v_max3_f32 %0, %0, %6, %7\n\t"
      "v_max3_f32 %0, %0, %8, %9\n\t"
      "v_max3_f32 %0, %0, %10, %11\n\t"
      "v_max3_f32 %0, %0, %12, %13\n\t"
      "v_max3_f32 %0, %0, %14, %15\n\t"
      "v_max_f32 %0, %0, %16"
      : "=&v"(d)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
        "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
  return d;
}

__device__ __forceinline__ int max16i(const i32x16& v) {   // the same for the 8-bit kernel's integer accumulators
  int d;
  asm("v_max3_i32 %0, %1, %2, %3\n\t"
      "v_max3_i32 %0, %0, %4, %5\n\t"
      "v_max3_i32 %0, %0, %6, %7\n\t"
      "v_max3_i32 %0, %0, %8, %9\n\t"
      "v_max3_i32 %0, %0, %10, %11\n\t"
      "v_max3_i32 %0, %0, %12, %13\n\t"
      "v_max3_i32 %0, %0, %14, %15\n\t"
      "v_max_i32 %0, %0, %16"
      : "=&v"(d)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
        "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
  return d;
}

// operand / accumulator types and the MFMA of the two operand widths of the v7 kernel.  The 8-bit form multiplies int8 rows by
// int8 queries into int32 (v_mfma_i32_32x32x32_i8: the same 16 bytes per lane and the same 32 cycles as the fp16 instruction, twice
// the K) - its instruction stream, LDS layout and DMA pieces are byte for byte those of the fp16 kernel, a K-step covers 128 bytes of
// a row either way.
template <bool I8> struct V7Op;
template <> struct V7Op<false> {
  typedef half8 frag;
  typedef f32x16 accv;
  typedef float scalar;
  static __device__ __forceinline__ accv mfma(const frag& a, const frag& b, const accv& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ scalar max16(const accv& v) { return max16f(v); }
};
template <> struct V7Op<true> {
  typedef i32x4 frag;
  typedef i32x16 accv;
  typedef int scalar;
  static __device__ __forceinline__ accv mfma(const frag& a, const frag& b, const accv& c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ scalar max16(const accv& v) { return max16i(v); }
};

// VGPR-form accumulators (r4).  An MFMA's C / D operands may live in arch VGPRs as well as in the accumulator file (one ACC_CD bit
// for both), and its A / B operands may live in AGPRs.  v7 kept all 256 accumulators in AGPRs and the 128 fragment registers in
// VGPRs - so the epilogue paid 256 v_accvgpr_read_b32 per tile (VALU cannot read AGPRs), half of its ~4.5 k cycles.  With
// EPS_V7_VI = n the first n of the 8 row blocks (32 n accumulators... x JQ) are kept in arch VGPRs and the operand fragments move to
// the accumulator file (the LDS / global loads that fill them are hand-issued asm already and can target AGPRs directly): the
// epilogue's max / compare chains read those blocks in place.  The MFMAs become inline asm, which hipcc neither schedules nor pads:
// every hazard is handled where it arises (see the kernel).
#ifndef EPS_V7_VI
#define EPS_V7_VI 7
#endif
// (tried and dropped, profiles/r4_flat_ab_epilogue_late.txt: the next tile's start values loaded at the end of a row block's own epilogue
// iteration; the pending-list flush check only after a tile that appended - no measurable difference either way)
#ifndef EPS_V7_GROUP
#define EPS_V7_GROUP 2       // row blocks per epilogue test: the maxima of GROUP x JQ blocks share one compare + branch (see the kernel)
#endif
#ifndef EPS_V7_HITMASK
#define EPS_V7_HITMASK 1   // FM_IDS hit blocks: per-lane bit mask of the passing values instead of 16 exec-masked branches (see the kernel)
#endif
#ifndef EPS_V7_TILE
#define EPS_V7_TILE 0   // tile-level epilogue test (see the kernel; measured 1.5 % slower than the per-block form, profiles/r4_epilogue_ablation.txt): lab switch, needs EPS_V7_VI > 0
#endif
#if EPS_V7_VI > 0
#define EPS_FRAG_C "=a"
#define EPS_FRAG_RW "+a"
template <bool I8> struct V7Asm;
template <> struct V7Asm<true> {
  template <class A, class F> static __device__ __forceinline__ void v(A& acc, const F& a, const F& b) { asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b)); }
  template <class A, class F> static __device__ __forceinline__ void a_(A& acc, const F& a, const F& b) { asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b)); }
  template <class A, class F> static __device__ __forceinline__ void v2(A& d, const F& a, const F& b, const A& c) { asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c)); }
  template <class A, class F> static __device__ __forceinline__ void a2(A& d, const F& a, const F& b, const A& c) { asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&a"(d) : "a"(a), "a"(b), "a"(c)); }
};
template <> struct V7Asm<false> {
  template <class A, class F> static __device__ __forceinline__ void v(A& acc, const F& a, const F& b) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b)); }
  template <class A, class F> static __device__ __forceinline__ void a_(A& acc, const F& a, const F& b) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b)); }
  template <class A, class F> static __device__ __forceinline__ void v2(A& d, const F& a, const F& b, const A& c) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c)); }
  template <class A, class F> static __device__ __forceinline__ void a2(A& d, const F& a, const F& b, const A& c) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&a"(d) : "a"(a), "a"(b), "a"(c)); }
};
#else
#define EPS_FRAG_C "=v"
#define EPS_FRAG_RW "+v"
#endif

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 3) + (chunk ^ ((row >> 1) & 7)); }  // 16-B granule index

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
__global__ __launch_bounds__(512, 2) void mfma_filter_kernel_v3(FilterArgs a) {
  constexpr int ablate = 0;   // (the ablation switches of the kernel lab compile away)
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

// hand-issued LDS fragment reads and global fragment loads (waited for by count; hipcc's own waitcnt insertion would
// drain the queues at every use)
#define EPS_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : EPS_FRAG_C(dst) : "v"(addr), "n"(off))
#define EPS_GLOAD_B128(dst, voff, sbase, off) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : EPS_FRAG_C(dst) : "v"(voff), "s"(sbase), "n"(off))
// r6 lab ablations of the operand transport (-DEPS_LAB -DEPS_V7_ABL=..., answers wrong by construction; scripts/lab/r6_headline_cap.sh):
//   32  no LDS fragment reads (the row fragments keep the random bytes they are given at the start of the kernel)
//   64  no global traffic in the K loop: no LDS-DMA pieces, no query-fragment loads, no start-value column
// with 1 (no epilogue): 1 | 32 | 64 = the kernel's own MFMA stream and nothing else - the rate the matrix pipe sustains on this board under this
// kernel's barriers and scalar code, i.e. what no operand schedule of this tile shape can exceed.
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & 32)
#undef EPS_DS_READ_B128
#define EPS_DS_READ_B128(dst, addr, off) asm volatile("" : EPS_FRAG_RW(dst) : "v"(addr))   /* (read-write: the fragment keeps the random bytes it was given) */
#endif
//  128  no query-fragment loads only (the row operand's LDS-DMA ring stays): what a QUERY-STATIONARY tile could reach at best - a workgroup's 256
//       queries never change, their fragments are the half of the global operand traffic that residency would remove
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & (64 | 128))
#undef EPS_GLOAD_B128
#define EPS_GLOAD_B128(dst, voff, sbase, off) asm volatile("" : EPS_FRAG_RW(dst) : "v"(voff), "s"(sbase))
#endif

// ------------------------------------------------------------------------------------------------ v7 kernel
// (v5, the 8-wavefront predecessor of this kernel, lives in scripts/lab/lab_v5.hpp.)  Its operand transport - query
// fragments straight to VGPRs, 4-slot LDS-DMA ring for the row operand, counted waits - with FOUR wavefronts per workgroup (one per SIMD, 256 VGPRs + 256 AGPRs each): wavefront w =
// all 256 rows x queries [64w, 64w+64) = 8 x 2 tiles of v_mfma_f32_32x32x16_f16, so every row fragment read from LDS
// feeds two MFMAs (v5: one).  Measured (scripts/lab, profiles/r1_mfma_lab.txt): the LDS read volume, not the schedule,
// is what costs v5 its clock (GRBM cycles/us drop from 2.07 GHz with MFMAs alone to 1.72 GHz with the fragment reads);
// halving it is worth more than the second wavefront per SIMD.  With one wavefront per SIMD the schedule is explicit:
// after every pair of MFMAs (64 cycles of matrix pipe) exactly one LDS read is issued in their shadow - the fragment the
// same pair needs in the NEXT K=16 sub-step, 512 cycles ahead, waited for by count (lgkmcnt(7)) - and on odd pairs one
// LDS-DMA piece (row operand, three steps ahead) or one query-fragment load (two steps ahead).
// JQ = 32-query blocks per wavefront: 2 -> 256-query tiles (the throughput shape), 1 -> 128-query tiles for batches of
// <= 128 queries, which halves the (padded) MFMA work and leaves the pass bound by streaming the fp16 mirror.
// MODE (what a tile's epilogue does with the accumulators; compile-time so that the tile loop carries one path only):
//   FM_IDS   rows with acc >= T_q are appended to the query's candidate list as row ids           (exact mode, every stage)
//   FM_KEYS  the same, as (approximate distance, row) keys                                         (approx mode)
//   FM_DENSE seed pass: the approximate key of EVERY row goes to slot (row - first row): no test, no atomics
// Appending (r2): a hit used to cost the wavefront one returning global atomic + the vmcnt(0) that waits for it - which also
// drains the LDS-DMA ring - i.e. ~1 us of a ~21 us tile, and the other three wavefronts wait for it at the next barrier; the
// main stage of the 10M-row scan sees ~0.3 hits per wavefront and tile.  Hits now go to a per-wavefront LDS list (slot by
// an LDS atomic: lgkmcnt only) that is flushed to the global lists, all entries at once, when it is half full (and at the
// end of the kernel); only a hit that finds the list full takes the direct path.
enum { FM_IDS = 0, FM_KEYS = 1, FM_DENSE = 2 };
constexpr int V7_CAPW = 128;   // entries of a wavefront's pending-candidate list
constexpr size_t v7_lds_bytes(int nrb) { return (size_t)4 * nrb * 4096 + 2 * 256 * sizeof(float) + 4096 + 64 + 4 * V7_CAPW * (8 + 4); }
constexpr size_t V7_LDS_BYTES = v7_lds_bytes(8);

// I8 (8-bit operands): a.xh / a.qf hold int8 [..][2 * d_pad] (d_pad counts 2-byte units in both forms), a.base_s the int32 accumulator
// start of every row, a.T the int32 pass thresholds (a row passes iff its accumulator >= T), a.s the (negative) key units per
// accumulator unit, a.qstat[q][3] the query's constant of the approximate distance = a.s * accumulator + constant.
// NRB (r4) = 32-row blocks per wavefront = rows per tile / 32.  8: the form described above, one workgroup per CU (a wavefront owns all 512
// registers of its SIMD lane).  4: 128-row tiles, half the accumulators (all of them in arch VGPRs), query fragments single-buffered, 16-KB
// ring slots - a wavefront fits 256 registers and a workgroup 78 KB of LDS, so TWO workgroups share a CU: while one is in its tile head /
// epilogue (18 % of a tile, matrix pipe idle) or waits at a barrier, the other one's MFMAs run.  The LDS read volume per MFMA is v7's (a row
// fragment still feeds two MFMAs); the query fragments are loaded twice as often per MFMA (L2 traffic x 1.5).  a.tile0 / a.ntiles count
// tiles of 32 NRB rows.
template <int JQ, int MODE, bool I8 = false, int NRB = 8>
__global__ __launch_bounds__(256, NRB == 8 ? 1 : 2) void mfma_filter_kernel_v7(FilterArgs a) {
  static_assert(NRB == 8 || (NRB == 4 && JQ == 2 && I8 && MODE != FM_DENSE), "row blocks per wavefront");
  constexpr int TR = 32 * NRB;            // rows per tile
  constexpr int FBUF = NRB == 8 ? 2 : 1;  // K-steps of query fragments held in registers
  constexpr int NPIECE = NRB;             // LDS-DMA pieces (32 rows x 128 B) per K-step
#ifndef EPS_V7_VI4
#define EPS_V7_VI4 2
#endif
  // row blocks whose accumulators live in arch VGPRs.  With AGPRs in use hipcc splits a wavefront's register budget evenly (NRB = 4: 128 arch
  // + 128 accumulator-file registers): two blocks (64) + everything else in the arch half, two blocks + the 64 fragment registers in the other
  constexpr int VI = NRB == 8 ? EPS_V7_VI : (EPS_V7_VI > 0 ? EPS_V7_VI4 : 0);
  typedef V7Op<I8> OP;
  typedef typename OP::frag frag_t;
  typedef typename OP::accv acc_t;
  typedef typename OP::scalar thr_t;
  constexpr int QT = 128 * JQ;   // queries per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int ASLOT = TR * 128;  // TR rows x 128 B
  constexpr int RING = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  float* base_lds = reinterpret_cast<float*>(lds + RING * ASLOT);  // [2][256]

  const int xcd = blockIdx.x & 7;
  const int local = blockIdx.x >> 3;
  const int per_xcd = gridDim.x >> 3;
  const int QTB = a.tiles_q < per_xcd ? a.tiles_q : per_xcd;
  const int G = per_xcd / QTB;
  const int qslot = local % QTB;
  const int rg = local / QTB;
  if (rg >= G) return;
  const int64_t nj = (a.ntiles - xcd + 7) / 8;
  const int nqt = (a.tiles_q - qslot + QTB - 1) / QTB;
  const int64_t my_rows = nj > rg ? (nj - rg + G - 1) / G : 0;
  const int64_t ntile = my_rows * nqt;
  if (ntile <= 0) return;
  const int ldk = a.d_pad;
  const int KT = ldk / 64;      // even and >= 4

  // LDS-DMA piece `it` (0..7) of a K-step covers rows [32 it, 32 it + 32) of the tile: lane offsets differ from piece 0's
  // only by it * 32 rows, which goes into the scalar base - one offset register for all pieces
  u32 g_off0;
  // tile t of this workgroup = (row index ri = t / nqt, query index qi = t % nqt), kept as two counters that are stepped
  // (a 64-bit division per tile is ~150 scalar instructions on this machine, and the tile loop had two)
  auto tile_rt = [&](int ri) { return (int64_t)xcd + 8 * (rg + (int64_t)ri * G); };
  auto tile_qt = [&](int qi) { return qslot + qi * QTB; };
  auto rows_of = [&](int ri) { return a.xh + (a.tile0 + tile_rt(ri)) * TR * (int64_t)ldk; };
  // fragment stream of this wavefront's first 32-query block; the second block follows at + (ldk/16)*512 halfs
  auto frags_of = [&](int qi) { return a.qf + ((int64_t)(tile_qt(qi) * (4 * JQ) + wave * JQ) * (ldk / 16)) * 512; };
  auto advance = [&](int& ri, int& qi) { if (++qi == nqt) { qi = 0; ++ri; } };
  const int64_t jstride = (int64_t)(ldk / 16) * 512;
  u32 lane16, lane4;
  const u32 lds_base = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue_base = [&](int ri, int par) {  // pre-scaled |x|^2 column of row tile ri -> base_lds[par] (64 rows per wavefront)
    if (wave * 64 >= TR) return;   // (NRB = 4: the first two wavefronts; the others' VMEM counts run one behind, which only makes their waits stricter)
    const float* pb = a.base_s + (a.tile0 + tile_rt(ri)) * TR + wave * 64;
    const u32 m0v = lds_base + RING * ASLOT + (u32)((par * 256 + wave * 64) * 4);
#if !(defined(EPS_V7_ABL) && (EPS_V7_ABL & 64))
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(lane4), "s"(pb), "s"(m0v) : "memory");
#endif
  };
  // LDS-DMA, saddr form: 32-bit lane offset + scalar base, M0 = LDS address of lane 0's 16 bytes.  Hand-issued so the
  // compiler neither forms 64-bit VGPR addresses nor tracks these in its waitcnt model (see v5).
  auto issue_piece = [&](const _Float16* pA, u32 slot_off, int it) {   // pA: first row of the tile, at the K-step to fetch
    const _Float16* sb = pA + (int64_t)it * 32 * ldk;
    const u32 m0v = lds_base + slot_off + (it * 256 + wave * 64) * 16;
#if !(defined(EPS_V7_ABL) && (EPS_V7_ABL & 64))
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(g_off0), "s"(sb), "s"(m0v) : "memory");
#endif
  };

  // the same in two halves for the K loop: address + M0 behind one MFMA, the DMA instruction alone behind the next (the MFMA
  // between them also provides the wait state M0 needs; nothing else in the loop touches M0 - checked in the ISA)
  auto prep_piece = [&](const _Float16* pA, u32 slot_off, int it) __attribute__((always_inline)) {
    const _Float16* sb = pA + (int64_t)it * 32 * ldk;
    const u32 m0v = lds_base + slot_off + (it * 256 + wave * 64) * 16;
    asm volatile("s_mov_b32 m0, %1" : "+s"(sb) : "s"(m0v) : "memory");
    return sb;
  };
  auto fire_piece = [&](const _Float16* sb) __attribute__((always_inline)) {
#if !(defined(EPS_V7_ABL) && (EPS_V7_ABL & 64))
    asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(g_off0), "s"(sb) : "memory");
#endif
  };

  acc_t acc[NRB][JQ];
  frag_t fb[FBUF][4][JQ];
  frag_t fa[2][NRB];
  int64_t qj[JQ];
  // Tq = T/s: a row passes iff acc >= Tq (s < 0); cj: approx-mode constant of the query.  Per-lane constants of the
  // query tile: parked in LDS and read back at each epilogue - as registers they would be live across the K loop, get
  // spilled, and their scratch reload would again drain the VMEM queue (vmcnt(0)) once per tile.
  float* tq_lds = base_lds + 512 + wave * 256;   // [4][64] per wavefront: Tq0, Tq1, cj0, cj1
#pragma unroll
  for (int j = 0; j < JQ; ++j) {
    qj[j] = (int64_t)qslot * QT + wave * (32 * JQ) + j * 32 + l31;
    tq_lds[j * 64 + lane] = I8 ? a.T[qj[j]] : a.T[qj[j]] * a.inv_s;   // (I8: int32 bits, moved as they are)
    tq_lds[(2 + j) * 64 + lane] = MODE != FM_IDS ? (I8 ? a.qstat[qj[j] * 4 + 3] : (a.metric == 0 ? a.qstat[qj[j] * 4] : (a.metric == 1 ? 1.f : 0.f))) : 0.f;
  }
  // Everything derived from the lane id that the K loop keeps in registers is RE-DERIVED at the top of every tile from
  // v_mbcnt (a dozen VALU instructions): values that live across the tile loop get spilled around the epilogue's
  // register peak, and a scratch reload makes hipcc wait vmcnt(0) - which drains the LDS-DMA ring once per tile.
  u32 faddr[4];
  auto lane_values = [&]() __attribute__((always_inline)) {
    u32 ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    lane16 = ln * 16;
    lane4 = ln * 4;
    const u32 td = (u32)wave * 64 + ln, row = td >> 3;
    g_off0 = (row * (u32)ldk + ((td & 7) ^ ((row >> 1) & 7)) * 8) * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) faddr[kk] = lds_base + (u32)swz((int)(ln & 31), kk * 2 + (int)(ln >> 5)) * 16;
  };
  lane_values();
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & (32 | 64 | 128))
  {   // the ablated transports leave operands where they are: give fragments and ring bytes that toggle like data (a zero operand runs the pipe at 2.4 GHz)
    u32 h = (u32)tid * 2654435761u + (u32)blockIdx.x * 40503u + 12345u;
    auto nx = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
    for (int i = tid; i < RING * ASLOT / 4; i += 256) reinterpret_cast<u32*>(lds)[i] = nx();
    typedef u32 __attribute__((ext_vector_type(4))) u32x4;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < NRB; ++i) { u32x4 v = {nx(), nx(), nx(), nx()}; fa[c][i] = __builtin_bit_cast(frag_t, v); }
#pragma unroll
    for (int c = 0; c < FBUF; ++c)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < JQ; ++j) { u32x4 v = {nx(), nx(), nx(), nx()}; fb[c][kk][j] = __builtin_bit_cast(frag_t, v); }
    __syncthreads();
  }
#endif

  const int64_t a_stride = (int64_t)8 * G * TR * ldk;   // halfs between consecutive row tiles of this workgroup
  int ri_c = 0, qi_c = 0;        // tile t
  int ri_n = 0, qi_n = 0;        // tile t + 1
  advance(ri_n, qi_n);
  const _Float16* A_t = rows_of(0);
  const _Float16* A_n = ntile > 1 ? rows_of(ri_n) : A_t;
  const _Float16* B_t = frags_of(0);
  const _Float16* B_n = (nqt > 1 && ntile > 1) ? frags_of(qi_n) : B_t;
  // pending-candidate list of this wavefront (see MODE above): counter, (query << 32 | row) entries, approximate keys
  u32* wcnt = reinterpret_cast<u32*>(lds + RING * ASLOT + 2048 + 4096) + wave * 4;
  u64* wbuf = reinterpret_cast<u64*>(lds + RING * ASLOT + 2048 + 4096 + 64) + wave * V7_CAPW;
  float* wkey = reinterpret_cast<float*>(lds + RING * ASLOT + 2048 + 4096 + 64 + 4 * V7_CAPW * 8) + wave * V7_CAPW;
  // (the same counter as an LDS-address-space volatile: through the generic pointer the per-tile look at it was a FLAT load + vmcnt(0), which
  // drained the LDS-DMA ring once per tile)
  volatile __attribute__((address_space(3))) u32* wcnt_lds = (volatile __attribute__((address_space(3))) u32*)wcnt;
  if (MODE != FM_DENSE && lane == 0) *wcnt = 0;
  auto append = [&](int64_t qq, u32 row, float dapx) __attribute__((always_inline)) {   // straight to the global list
    const u32 slot_c = atomicAdd(&a.cnt[qq], 1u);
    if (slot_c < (u32)a.cap) {
      if (MODE == FM_KEYS) a.cand_keys[qq * (int64_t)a.cap + slot_c] = make_key(dapx, row);
      else a.cand[qq * (int64_t)a.cap + slot_c] = row;
    }
  };
  auto flush = [&]() __attribute__((always_inline)) {
    u32 ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    u32 n = *wcnt_lds;
    n = n < (u32)V7_CAPW ? n : (u32)V7_CAPW;
    for (u32 e = ln; e < n; e += 64) {
      const u64 v = wbuf[e];
      append((int64_t)(v >> 32), (u32)v, MODE == FM_KEYS ? wkey[e] : 0.f);
    }
    if (ln == 0) *wcnt_lds = 0;
  };
  const bool rendezvous = a.group_sync && a.tiles_q <= per_xcd;   // (then nqt == 1 for every member of the group)
  u32* gs_ctr = a.group_sync + (xcd * G + rg);
  const int64_t sync_mask = ((int64_t)1 << a.sync_shift) - 1;   // the group meets before every 2^sync_shift-th tile
  if (rendezvous && wave == 0 && lane == 0) __hip_atomic_fetch_add(gs_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue_base(0, 0);
  // prologue = the issue groups of the imaginary steps -3, -2, -1 (16 operations each from -2 on)
#pragma unroll
  for (int it = 0; it < NPIECE; ++it) issue_piece(A_t, 0, it);
#pragma unroll
  for (int it = 0; it < NPIECE; ++it) issue_piece(A_t + 64, ASLOT, it);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    EPS_GLOAD_B128(fb[0][kk][0], lane16, B_t + kk * 512, 0);
    if (JQ == 2) EPS_GLOAD_B128(fb[0][kk][JQ - 1], lane16, B_t + jstride + kk * 512, 0);
  }
#pragma unroll
  for (int it = 0; it < NPIECE; ++it) issue_piece(A_t + 128, 2 * ASLOT, it);
  if (FBUF == 2) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      EPS_GLOAD_B128(fb[FBUF - 1][kk][0], lane16, B_t + 2048 + kk * 512, 0);
      if (JQ == 2) EPS_GLOAD_B128(fb[FBUF - 1][kk][JQ - 1], lane16, B_t + jstride + 2048 + kk * 512, 0);
    }
    if (JQ == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // slots 0 and 1 + fragments of step 0
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (slot 2's pieces may stay in flight)
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  EPS_DS_READ_B128(fa[0][0], faddr[0], 0);
  EPS_DS_READ_B128(fa[0][1], faddr[0], 4096);
  EPS_DS_READ_B128(fa[0][2], faddr[0], 8192);
  EPS_DS_READ_B128(fa[0][3], faddr[0], 12288);
  if (NRB == 8) {
    EPS_DS_READ_B128(fa[0][NRB - 4], faddr[0], 16384);
    EPS_DS_READ_B128(fa[0][NRB - 3], faddr[0], 20480);
    EPS_DS_READ_B128(fa[0][NRB - 2], faddr[0], 24576);
    EPS_DS_READ_B128(fa[0][NRB - 1], faddr[0], 28672);
  }

  // One wavefront per SIMD: after every pair of MFMAs (64 cycles of matrix pipe) exactly one other instruction is
  // issued in its shadow - the LDS read of the row fragment that the same pair will need in the NEXT sub-step
  // (8 pairs = 512 cycles ahead, waited for by count: lgkmcnt(7)), and on odd pairs one LDS-DMA piece / fragment load.
  // Prefetch cursors, stepped at the END of every K-step (in the shadow of its last MFMAs): where the LDS-DMA of this
  // step reads (the K-step three ahead, possibly in the next tile), where its query-fragment loads read (two ahead),
  // and the ring slots.  Computed at the head of the step (~30 scalar instructions: selects between this tile and the
  // next, 64-bit address arithmetic) they sat between the barrier and the first MFMA of every step.
  const _Float16* pA_run = A_t + 3 * 64;          // KT >= 4
  int akt_run = 3;
  const _Float16* pB_run = B_t + (int64_t)FBUF * 2048;   // (NRB = 4: one step ahead, reloaded in place one sub-step after use)
  int bkt_run = FBUF;
  u32 sA_run = 0, sN_run = ASLOT, sD_run = 3 * ASLOT;   // slot being multiplied, the next one, the one being filled (byte offsets)
  u32 ad_run = faddr[1];                                // LDS address of the first sub-step's fragment reads (slot 0)
  auto step = [&](auto U, auto FIRST) __attribute__((always_inline)) {
    constexpr int rb = decltype(U)::value % FBUF;
    constexpr bool first = decltype(FIRST)::value;   // first K-step of a tile: only acc[.][0] holds the base column
    const u32 sA = sA_run, sN = sN_run, sD = sD_run;
    const _Float16* pA = pA_run;
    const _Float16* pB = pB_run;
    const _Float16* pA_nx = pA;
    const _Float16* pB_nx = pB;
    int akt_nx = akt_run, bkt_nx = bkt_run;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      const u32 ad = ad_run;      // LDS address of this sub-step's fragment reads (the NEXT sub-step's operands)
      if (FBUF == 1 && kk > 0) {
        // single-buffered query fragments: this sub-step's pair was reloaded one K-step ago, in the sub-step after its use.  VMEM
        // operations of a step, in issue order: p0 | F00 p1 F01 | F10 p2 F11 | F20 p3 F21 | F30 F31 (p = LDS-DMA piece, Fkj = fragment
        // of sub-step k, query block j): what may still be in flight when F(kk, 1) of the previous step must have landed
        if (kk == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        // An MFMA occupies the pipe for 32 cycles and the next one cannot issue before that, so EACH of the two leaves
        // ~28 cycles (about five issue slots) in which the wavefront can issue something else for free: the LDS read and
        // the scalar preparation of the pair's VMEM instruction go behind the first, the VMEM instruction itself (every
        // other pair: one query-fragment load or one LDS-DMA piece) alone behind the second.
        const bool has_frag = (i == 1 || i == 3) && kk > 0 && (i >> 1) < JQ;   // fragments of the PREVIOUS sub-step's slot, for two steps from now
        const bool has_dma = NRB == 8 ? (i == 5 || i == 7) : i == 2;
        const _Float16* vsrc = nullptr;
        if (NRB == 8) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#if EPS_V7_VI > 0
        if (JQ == 2 && first && kk == 0) {   // D != C: the second block's accumulator is born from the first block's initial value
          if (i < VI) V7Asm<I8>::v2(acc[i][JQ - 1], fa[cur][i], fb[rb][kk][JQ - 1], acc[i][0]);
          else V7Asm<I8>::a2(acc[i][JQ - 1], fa[cur][i], fb[rb][kk][JQ - 1], acc[i][0]);
        } else {
          if (i < VI) V7Asm<I8>::v(acc[i][0], fa[cur][i], fb[rb][kk][0]);
          else V7Asm<I8>::a_(acc[i][0], fa[cur][i], fb[rb][kk][0]);
        }
#else
        if (JQ == 2 && first && kk == 0) {   // D != C: the second block's accumulator is born from the first block's initial value
          acc[i][JQ - 1] = OP::mfma(fa[cur][i], fb[rb][kk][JQ - 1], acc[i][0]);
        } else {
          acc[i][0] = OP::mfma(fa[cur][i], fb[rb][kk][0], acc[i][0]);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        {
          switch (i) {
            case 0: EPS_DS_READ_B128(fa[nxt][0], ad, 0); break;
            case 1: EPS_DS_READ_B128(fa[nxt][1], ad, 4096); break;
            case 2: EPS_DS_READ_B128(fa[nxt][2], ad, 8192); break;
            case 3: EPS_DS_READ_B128(fa[nxt][3], ad, 12288); break;
            case 4: EPS_DS_READ_B128(fa[nxt][NRB - 4], ad, 16384); break;
            case 5: EPS_DS_READ_B128(fa[nxt][NRB - 3], ad, 20480); break;
            case 6: EPS_DS_READ_B128(fa[nxt][NRB - 2], ad, 24576); break;
            default: EPS_DS_READ_B128(fa[nxt][NRB - 1], ad, 28672); break;
          }
        }
        if (has_frag) {
          vsrc = pB + (i >> 1) * jstride + (kk - 1) * 512;
          asm volatile("" : "+s"(vsrc));
        } else if (has_dma) {
          vsrc = prep_piece(pA, sD, NRB == 8 ? kk * 2 + (i >> 1) - 2 : kk);
        }
        if (i == NRB - 1) {   // the next sub-step's read address (kk = 3: the next K-step's first sub-step, in the slot after this one)
          u32 adn = faddr[(kk + 2) & 3] + (kk < 2 ? sA : sN);
          asm volatile("" : "+v"(adn));
          ad_run = adn;
        }
        if (kk == 3 && i == NRB - 2) {   // the next step's cursors
          akt_nx = akt_run + 1;
          pA_nx = pA + 64;
          if (akt_nx == KT) {
            akt_nx = 0;
            pA_nx = A_n;
          }
          bkt_nx = bkt_run + 1;
          pB_nx = pB + 2048;
          if (bkt_nx == KT) {
            bkt_nx = 0;
            pB_nx = B_n;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#if EPS_V7_VI > 0
        if (JQ == 2) {
          if (first && kk == 0) {
            if (i < VI) V7Asm<I8>::v(acc[i][0], fa[cur][i], fb[rb][kk][0]);
            else V7Asm<I8>::a_(acc[i][0], fa[cur][i], fb[rb][kk][0]);
          } else {
            if (i < VI) V7Asm<I8>::v(acc[i][JQ - 1], fa[cur][i], fb[rb][kk][JQ - 1]);
            else V7Asm<I8>::a_(acc[i][JQ - 1], fa[cur][i], fb[rb][kk][JQ - 1]);
          }
        }
#else
        if (JQ == 2) {
          if (first && kk == 0) acc[i][0] = OP::mfma(fa[cur][i], fb[rb][kk][0], acc[i][0]);
          else acc[i][JQ - 1] = OP::mfma(fa[cur][i], fb[rb][kk][JQ - 1], acc[i][JQ - 1]);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // (all loads of the loop stay in straight-line code: around a branch hipcc gives an asm load's destination a fresh
        // register and copies it - possibly before the data has landed; r2 tried to stagger the wavefronts' DMA issue that way)
        if (has_frag) EPS_GLOAD_B128(fb[rb][kk - 1][(i >> 1) % JQ], lane16, vsrc, 0);
        else if (has_dma) fire_piece(vsrc);
      }
    }
    {
      EPS_GLOAD_B128(fb[rb][3][0], lane16, pB + 3 * 512, 0);
      if (JQ == 2) EPS_GLOAD_B128(fb[rb][3][JQ - 1], lane16, pB + jstride + 3 * 512, 0);
    }
    pA_run = pA_nx;
    akt_run = akt_nx;
    pB_run = pB_nx;
    bkt_run = bkt_nx;
    sA_run = sN;
    sN_run = (sN + ASLOT) & (RING * ASLOT - 1);
    sD_run = sA;
    if (FBUF == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (F01 of this step has landed: the next step's first sub-step reads it)
    else if (JQ == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // this step's 8 DMA pieces + 4 JQ fragment loads may stay in flight
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

#ifdef EPS_V7_PROF
  unsigned long long pf_head = 0, pf_k = 0, pf_epi = 0;
#endif
  for (int64_t t = 0; t < ntile; ++t) {
#ifdef EPS_V7_PROF
    const unsigned long long pf_t0 = __builtin_readcyclecounter();
#endif
    const int64_t row0 = (a.tile0 + tile_rt(ri_c)) * TR;
    const int64_t qbase = (int64_t)tile_qt(qi_c) * QT + wave * (32 * JQ);   // scalar
    lane_values();
    ad_run = faddr[1] + sA_run;   // (re-derived with the lane values: nothing lane-dependent lives across the epilogue)
    // The QTB workgroups of a group stream the SAME row tiles (each against its own query tile) and only the first to
    // ask pays the HBM fetch - if the others ask within the few microseconds the lines survive in this XCD's L2.  With
    // the operands prefetched three steps ahead nothing self-synchronises them any more (measured: FETCH_SIZE 1.9 x
    // the algorithmic bytes), so they rendezvous every few tiles: each member ARRIVES when its K loop of the previous
    // tile ends (before that tile's epilogue, so the round trip of the atomic hides under it) and here only polls
    // (scalar loads: no VMEM counter involved) until the whole group has arrived - bounded, so a missing member can
    // only cost time.
    if (rendezvous && wave == 0 && (t & sync_mask) == 0) {
      const u32 want = (u32)QTB * (u32)((t >> a.sync_shift) + 1);
      for (int spin = 0; spin < 1024; ++spin) {
        u32 seen;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(gs_ctr) : "memory");
        if (seen >= want) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (nqt > 1 && t > 0) {   // (everything lane-dependent re-derived here as well: nothing of it may live across the K loop)
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        const int64_t qjt = (int64_t)tile_qt(qi_c) * QT + wave * (32 * JQ) + j * 32 + ((lane16 >> 4) & 31);
        tq_lds[j * 64 + (lane16 >> 4)] = I8 ? a.T[qjt] : a.T[qjt] * a.inv_s;
        float cm;   // (materialised here from a scalar: as an ordinary value hipcc keeps it in a VGPR across the tile loop and spills it)
        {
          const int sv = a.metric == 1 ? 0x3f800000 : 0;   // 1.0f : 0.0f
          asm volatile("v_mov_b32 %0, %1" : "=v"(cm) : "s"(sv));
        }
        tq_lds[(2 + j) * 64 + (lane16 >> 4)] = MODE != FM_IDS ? (I8 ? a.qstat[qjt * 4 + 3] : (a.metric == 0 ? a.qstat[qjt * 4] : cm)) : 0.f;
      }
    }
    if (t + 1 < ntile) issue_base(ri_n, (int)((t + 1) & 1));
    // acc[i][0] <- the pre-scaled |x|^2 column (8-bit: the rows' start values) of the tile
    auto init_block = [&](int i, int par, int kh4) __attribute__((always_inline)) {   // (i: a constant after unrolling)
      const float* bl0 = base_lds + par * 256;
      const int rbase = i * 32 + kh4;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        if (I8) {
          const int4 bv = *reinterpret_cast<const int4*>(&bl0[rbase + 8 * gq]);
          acc[i][0][4 * gq + 0] = bv.x;
          acc[i][0][4 * gq + 1] = bv.y;
          acc[i][0][4 * gq + 2] = bv.z;
          acc[i][0][4 * gq + 3] = bv.w;
        } else {
          const float4 bv = *reinterpret_cast<const float4*>(&bl0[rbase + 8 * gq]);
          acc[i][0][4 * gq + 0] = bv.x;
          acc[i][0][4 * gq + 1] = bv.y;
          acc[i][0][4 * gq + 2] = bv.z;
          acc[i][0][4 * gq + 3] = bv.w;
        }
      }
    };
    {
      const int kh4 = (int)(lane16 >> 7) & 4;     // = 4 * khalf
#define EPS_INIT_AT_HEAD(I_)                                                                        \
  {                                                                                                 \
    init_block((I_), (int)(t & 1), kh4);                                                            \
    __builtin_amdgcn_sched_barrier(0); /* one row block at a time: hoisting all 64 reads costs spills */ \
  }
      EPS_INIT_AT_HEAD(0) EPS_INIT_AT_HEAD(1) EPS_INIT_AT_HEAD(2) EPS_INIT_AT_HEAD(3)
      if (NRB == 8) { EPS_INIT_AT_HEAD(NRB - 4) EPS_INIT_AT_HEAD(NRB - 3) EPS_INIT_AT_HEAD(NRB - 2) EPS_INIT_AT_HEAD(NRB - 1) }
#undef EPS_INIT_AT_HEAD
    }
#ifdef EPS_V7_PROF
    const unsigned long long pf_t1 = __builtin_readcyclecounter();
#endif
    step(std::integral_constant<int, 0>{}, std::true_type{});
    step(std::integral_constant<int, 1>{}, std::false_type{});
    for (int kt = 2; kt < KT; kt += 2) {
      step(std::integral_constant<int, 0>{}, std::false_type{});
      step(std::integral_constant<int, 1>{}, std::false_type{});
    }
#if EPS_V7_VI > 0
    // the asm MFMAs' results are read by VALU code below (max / compare on the VGPR blocks, v_accvgpr_read on the others): hipcc pads
    // nothing after an asm statement, an 8-pass MFMA's D needs 11 wait states before a VALU reader, a 16-pass one 19
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#endif
#ifdef EPS_V7_PROF
    const unsigned long long pf_t2 = __builtin_readcyclecounter();
#endif
    if (rendezvous && wave == 0 && t + 1 < ntile && ((t + 1) & sync_mask) == 0 && lane16 == 0) __hip_atomic_fetch_add(gs_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    A_t = A_n;
    B_t = B_n;
    ri_c = ri_n;
    qi_c = qi_n;
    {
      const int ri_before = ri_n;
      advance(ri_n, qi_n);
      if (t + 2 < ntile) {
        if (ri_n != ri_before) A_n += a_stride;   // the next row tile of this workgroup: 8 * G tiles further on
        if (nqt > 1) B_n = frags_of(qi_n);
      }
    }
    // the epilogue derives its lane constants afresh too (nothing lane-dependent is live across the K loop but the
    // operand offsets the loop itself uses)
    u32 lne;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lne));
    const int l31e = (int)(lne & 31), kh4e = (int)(lne >> 5) * 4;
    thr_t Tq[JQ];
    float cj[JQ];
#if EPS_V7_TILE
    // the tile's thresholds: the LDS reads are issued here and waited for after the max chains below (hipcc sinks an ordinary load to its
    // first use and waits for it there, ~100 exposed cycles per tile)
    u32 tqraw[2] = {0u, 0u};
    if (I8 && MODE != FM_DENSE) {
      const u32 tq_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) float*)tq_lds + lne * 4u;
      asm volatile("ds_read_b32 %0, %1" : "=v"(tqraw[0]) : "v"(tq_addr));
      if (JQ == 2) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(tqraw[1]) : "v"(tq_addr));
    }
#endif
#pragma unroll
    for (int j = 0; j < JQ; ++j) {
#if EPS_V7_TILE
      if (!(I8 && MODE != FM_DENSE))
#endif
      Tq[j] = __builtin_bit_cast(thr_t, tq_lds[j * 64 + lne]);
      cj[j] = MODE == FM_DENSE ? tq_lds[(2 + j) * 64 + lne] : 0.f;   // (FM_KEYS reads it where a row passes: one live register less)
    }
    // Tile-level test (r4, 8-bit kernel).  The per-block form below costs a wavefront ~1800 cycles per tile although a block passes
    // once in ~100 (lab ablations, profiles/r4_epilogue_ablation.txt: the 16 maxima themselves 2 % of the kernel, the 16 compare +
    // TAKEN-branch pairs around the hit code and the hit code 9 %): with one wavefront per SIMD nothing hides a taken branch's
    // refetch.  With the accumulators in arch VGPRs a second look at them is free, so: the maximum of ALL 128 values a lane holds for
    // each of its query columns (four interleaved v_max3 chains, no branch), ONE compare per column, ONE branch per tile that is NOT
    // taken on the common path; only a tile in which something passed (one in six at the last stage of a 10M-row scan) runs the
    // per-block code, and only it can have filled the pending list, so the flush check moves there too.
    bool tile_hit = true;
#if EPS_V7_TILE
    if (I8 && MODE != FM_DENSE) {
      bool h = false;
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        int p0 = (int)acc[0][j][0], p1 = (int)acc[0][j][1], p2 = (int)acc[0][j][2], p3 = (int)acc[0][j][3];
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
#pragma unroll
          for (int r = (i == 0 ? 4 : 0); r < 16; r += 4) {
            if (i < VI) {
              const int a0 = (int)acc[i][j][r], a1 = (int)acc[i][j][r + 1], a2 = (int)acc[i][j][r + 2], a3 = (int)acc[i][j][r + 3];
              p0 = p0 > a0 ? p0 : a0;
              p1 = p1 > a1 ? p1 : a1;
              p2 = p2 > a2 ? p2 : a2;
              p3 = p3 > a3 ? p3 : a3;
            } else {
              // a block that lives in the accumulator file: read through two scratch registers, four values at a time (as plain C++
              // hipcc copies the whole block out, and back in again when it runs out of arch VGPRs)
              int t0, t1;
              asm("v_accvgpr_read_b32 %2, %4\n\tv_accvgpr_read_b32 %3, %5\n\tv_max_i32 %0, %0, %2\n\tv_max_i32 %1, %1, %3\n\t"
                  "v_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7\n\tv_max_i32 %0, %0, %2\n\tv_max_i32 %1, %1, %3"
                  : "+v"(p0), "+v"(p1), "=&v"(t0), "=&v"(t1)
                  : "a"((int)acc[i][j][r]), "a"((int)acc[i][j][r + 1]), "a"((int)acc[i][j][r + 2]), "a"((int)acc[i][j][r + 3]));
            }
          }
        }
        const int m01 = p0 > p1 ? p0 : p1, m23 = p2 > p3 ? p2 : p3;
        if (j == 0) {   // (after the first column's chains: the thresholds have long landed)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tqraw[0]), "+v"(tqraw[1]));
          __builtin_amdgcn_sched_barrier(0);
        }
        Tq[j] = (thr_t)(int)tqraw[j];
        h |= (m01 > m23 ? m01 : m23) >= (int)Tq[j];
      }
      tile_hit = __any(h);
    }
#endif
    if (__builtin_expect(tile_hit, 0)) {
#if EPS_V7_TILE && EPS_V7_VI > 0 && EPS_V7_VI < 8
    if (I8 && MODE != FM_DENSE) {   // (the blocks in the accumulator file become "new" values here, or hipcc copies them out on the common path)
#pragma unroll
      for (int i = VI; i < NRB; ++i)
#pragma unroll
        for (int j = 0; j < JQ; ++j) asm volatile("" : "+a"(acc[i][j]));
    }
#endif
#pragma unroll
    for (int i0 = 0; i0 < NRB; i0 += EPS_V7_GROUP) {
    // r4 (EPS_V7_GROUP > 1): the maxima of GROUP row blocks x JQ query blocks are combined per query column and tested ONCE - one compare
    // pair + one untaken branch per group instead of one per block (lab ablation: the 16 compare + branch pairs cost the launch 3 %);
    // unlike the tile-level test nothing is computed twice: a group that passes re-uses its blocks' maxima for the per-block tests
    thr_t mxg[EPS_V7_GROUP][JQ];
    bool group_hit = true;
    if (EPS_V7_GROUP > 1 && MODE != FM_DENSE) {
      bool h = false;
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        thr_t gm = mxg[0][j] = OP::max16(acc[i0][j]);
#pragma unroll
        for (int ii = 1; ii < EPS_V7_GROUP; ++ii) {
          __builtin_amdgcn_sched_barrier(0);
          mxg[ii][j] = OP::max16(acc[i0 + ii][j]);
          gm = gm > mxg[ii][j] ? gm : mxg[ii][j];
        }
        h |= gm >= Tq[j];
      }
      group_hit = __any(h);
    }
    if (EPS_V7_GROUP == 1 || MODE == FM_DENSE || __builtin_expect(group_hit, 0)) {
#pragma unroll
    for (int ii = 0; ii < EPS_V7_GROUP; ++ii) {
      const int i = i0 + ii;
      const int rbase = i * 32 + kh4e;
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        __builtin_amdgcn_sched_barrier(0);   // one 32 x 32 block at a time (bounded register pressure)
        if (MODE == FM_DENSE) {   // seed pass (approx keys of ALL head rows): slot = row index, no compare, no atomic
          const int64_t qq = qbase + j * 32 + l31e;
          if (qq < a.nq) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int64_t row = row0 + rbase + (r & 3) + 8 * (r >> 2);
              float dapx = (float)acc[i][j][r] * a.s + cj[j];
              const bool nan = dapx != dapx;
              if (a.metric == 0) dapx = fmaxf(dapx, 0.f);
              if (row < a.row_hi)
                a.cand_keys[qq * (int64_t)a.cap + (row - a.tile0 * TR)] =
                    nan ? KEY_EMPTY : make_key(dapx, (u32)row);
            }
          }
          continue;
        }
        // the block's running max: 8 x v_max3_f32 (fmaxf chains cost 10: hipcc canonicalises the first two operands)
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & 1)
        continue;   // lab ablation: no epilogue work at all (results are wrong; what the whole epilogue costs)
#endif
        const thr_t mx = EPS_V7_GROUP > 1 ? mxg[ii][j] : OP::max16(acc[i][j]);
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & 2)
        asm volatile("" ::"v"(mx));   // lab ablation: maxima computed, never compared (what compare + branch + hit path cost)
        continue;
#endif
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & 8)
        if (__any(mx >= Tq[j])) asm volatile("s_nop 0");   // lab ablation: compare + branch, empty hit path (what the 16 branch pairs cost)
        continue;
#endif
#if defined(EPS_V7_ABL) && (EPS_V7_ABL & 16)
        if (__any(mx >= Tq[j]) && a.ablate) {   // lab ablation: the hit code is all there (code size, branches) and never runs (a.ablate == 0)
#else
        if (__any(mx >= Tq[j])) {
#endif
          // (rare) everything the hit path needs is derived behind this opaque copy of the lane id, or hipcc hoists the
          // address arithmetic of all 16 blocks into the common path
          int l31h = l31e, rbh = rbase;
          asm volatile("" : "+v"(l31h), "+v"(rbh));
          const int64_t qq = qbase + j * 32 + l31h;
#if EPS_V7_HITMASK
          // r4: the hit block without 16 exec-masked branches (lab ablation, profiles/r4_epilogue_ablation.txt: the hit code running costs
          // the launch 3 %, its being there - 35 KB of unrolled per-value branches - another 2 %).  Row-id lists need no accumulator
          // VALUE, only WHICH of a lane's 16 values passed: a 16-bit mask per lane (branch-free compares), then only the lanes with a
          // bit set walk their bits - almost always one lane, one bit.  (Approximate-key lists pick the value by a 16-way select.)
          {
            u32 hm = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) hm |= acc[i][j][r] >= Tq[j] ? (1u << r) : 0u;
            if (qq >= a.nq) hm = 0;
            while (hm) {
              const int r = __builtin_ctz(hm);
              hm &= hm - 1;
              const int64_t row64 = row0 + rbh + (r & 3) + 8 * (r >> 2);
              if (row64 >= a.row_hi) continue;       // (rows beyond the stage's last row: the tile that crosses it)
              const u32 row = (u32)row64;
              float dapx = 0.f;
              if (MODE == FM_KEYS) {   // the value itself: picked out of the lane's 16 by a select chain (rare path)
                thr_t v = acc[i][j][0];
#pragma unroll
                for (int rr = 1; rr < 16; ++rr) v = r == rr ? acc[i][j][rr] : v;
                dapx = (float)v * a.s + tq_lds[(2 + j) * 64 + lne];
                if (a.metric == 0) dapx = fmaxf(dapx, 0.f);
              }
              const u32 e = atomicAdd(wcnt, 1u);   // LDS: no VMEM counter involved
              if (e < (u32)V7_CAPW) {
                wbuf[e] = ((u64)qq << 32) | row;
                if (MODE == FM_KEYS) wkey[e] = dapx;
              } else {
                append(qq, row, dapx);
              }
            }
            continue;
          }
#endif
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (acc[i][j][r] >= Tq[j]) {
              const int64_t row = row0 + rbh + (r & 3) + 8 * (r >> 2);
              if (row < a.row_hi && qq < a.nq) {
                float dapx = 0.f;
                if (MODE == FM_KEYS) {
                  dapx = (float)acc[i][j][r] * a.s + tq_lds[(2 + j) * 64 + lne];
                  if (a.metric == 0) dapx = fmaxf(dapx, 0.f);
                }
                const u32 e = atomicAdd(wcnt, 1u);   // LDS: no VMEM counter involved
                if (e < (u32)V7_CAPW) {
                  wbuf[e] = ((u64)qq << 32) | (u32)row;
                  if (MODE == FM_KEYS) wkey[e] = dapx;
                } else {
                  append(qq, (u32)row, dapx);
                }
              }
            }
          }
        }
      }
    }
    }   // group_hit
    }
#if !(defined(EPS_V7_ABL) && (EPS_V7_ABL & 4))   // (lab ablation: no flush check)
    if (MODE != FM_DENSE) {
      if (*wcnt_lds >= (u32)(V7_CAPW / 2)) flush();
    }
#endif
    }   // tile_hit
#ifdef EPS_V7_PROF
    const unsigned long long pf_t3 = __builtin_readcyclecounter();
    pf_head += pf_t1 - pf_t0;
    pf_k += pf_t2 - pf_t1;
    pf_epi += pf_t3 - pf_t2;
#endif
  }
#ifdef EPS_V7_PROF
  if (a.prof && (threadIdx.x & 63) == 0) {
    atomicAdd(&a.prof[0], pf_head);
    atomicAdd(&a.prof[1], pf_k);
    atomicAdd(&a.prof[2], pf_epi);
    atomicAdd(&a.prof[3], (unsigned long long)ntile);
  }
#endif
  if (MODE != FM_DENSE) flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// row-major fp16 queries -> fragment-major
__global__ void pack_qf_kernel(const _Float16* qh, _Float16* qf, int64_t b_pad, int d_pad) {
  const int64_t frag = blockIdx.x;               // (qblock, kc)
  const int KC = d_pad / 16;
  const int64_t qb = frag / KC;
  const int kc = (int)(frag % KC);
  const int lane = threadIdx.x;
  const half8 v = *reinterpret_cast<const half8*>(qh + (qb * 32 + (lane & 31)) * d_pad + kc * 16 + (lane >> 5) * 8);
  *reinterpret_cast<half8*>(qf + (frag * 64 + lane) * 8) = v;
}


}  // namespace eps
