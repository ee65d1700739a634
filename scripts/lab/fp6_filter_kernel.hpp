// FP6 (E2M3) coarse filter pass - lab form (r5).  A derivative of mfma_filter_kernel_v7 (mfma_kernels.hpp: tile order, LDS-DMA ring, counted waits,
// rendezvous, pending-candidate lists and the float epilogue are that kernel's), with the operand transport of the 6-bit format:
//   v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2: E2M3 x E2M3, scales 2^0) takes 32 codes = 24 bytes per lane and operand, in SIX consecutive
//   registers.  A K-step is still 128 bytes per row (the ring, the DMA pieces and the XOR swizzle move 16-byte granules and do not change), now
//   holding 128 codes = two MFMAs (m = 0, 1) x two lane halves (h = 0, 1):
//     granule 2m + h     bytes  0..15 of fragment (m, h)      -> ds_read_b128 into registers 0..3 of the operand
//     granule 4 + m      bytes 16..23 of fragments (m, 0), (m, 1) -> ds_read_b64  (+ 8h) into registers 4..5
//     granules 6, 7      padding
//   (the order of a row's codes along K is free as long as the query operand uses the same: both mirrors are packed by the same rule).
//   Query fragments: [b_pad/32][2 * KT][2048 bytes]: bytes 0..15 of the 64 lanes' fragments (1024 bytes), then their bytes 16..23 (512 bytes), then padding -
//   dwordx4 + dwordx2 straight into registers 0..3 / 4..5, both fully coalesced.
// Codes are multiples of 1/8 in [-7.5, 7.5]: every product is a multiple of 1/64, a 768-term sum stays below 2^24 / 64, so the fp32 accumulators
// hold EXACT integers (in units of 1/64) as long as the rows' start values are such integers too - the pass is integer arithmetic on a
// non-uniform 6-bit grid, and its pass test acc >= T is as exact as the int8 kernel's.
#pragma once
#include "mfma_kernels.hpp"

namespace eps {

typedef int i32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));
struct F6Frag {   // an operand of the 6-bit MFMA: two loads, one register tuple (the coalescer joins them: checked in the ISA)
  i32x4 lo;
  i32x2_t hi;
};
__device__ __forceinline__ i32x8_t f6_cat(const F6Frag& f) {
  const i32x4 h4 = __builtin_shufflevector(f.hi, f.hi, 0, 1, -1, -1);
  return __builtin_shufflevector(f.lo, h4, 0, 1, 2, 3, 4, 5, -1, -1);
}
__device__ __forceinline__ f32x16 f6_mfma(const F6Frag& a, const F6Frag& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f6_cat(a), f6_cat(b), c, 2, 2, 0, 127, 0, 127);
}
#define EPS_F6_DS_LO(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"((dst).lo) : "v"(addr), "n"(off))
#define EPS_F6_DS_HI(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"((dst).hi) : "v"(addr), "n"(off))
#define EPS_F6_GL_LO(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"((dst).lo) : "v"(voff), "s"(sbase))
#define EPS_F6_GL_HI(dst, voff, sbase) asm volatile("global_load_dwordx2 %0, %1, %2 offset:1024" : "=v"((dst).hi) : "v"(voff), "s"(sbase))

// JQ = 2 (256-query tiles), row-id lists (FM_IDS); a.xh = the 6-bit mirror [n_pad][2 * d_pad bytes] (d_pad / 64 K-steps of 128 bytes), a.qf the
// query fragments, a.base_s the rows' start values, a.T the pass thresholds (a row passes iff its accumulator >= T), a.inv_s is not used.
__global__ __launch_bounds__(256, 1) void mfma_filter_kernel_f6(FilterArgs a) {
  constexpr int JQ = 2, MODE = FM_IDS, NRB = 8;
  constexpr bool I8 = false;
  constexpr int TR = 32 * NRB;            // rows per tile
  constexpr int NPIECE = NRB;             // LDS-DMA pieces (32 rows x 128 B) per K-step
  constexpr int GROUP = 2;                // row blocks per epilogue test (GROUP)
  typedef f32x16 acc_t;
  typedef float thr_t;
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
  const int KT = ldk / 64;      // a multiple of 3 and >= 6 (768 or 1536 codes per row)

  // LDS-DMA piece `it` (0..7) of a K-step covers rows [32 it, 32 it + 32) of the tile: lane offsets differ from piece 0's
  // only by it * 32 rows, which goes into the scalar base - one offset register for all pieces
  u32 g_off0;
  // the lanes of an LDS-DMA piece that carry codes: granules 6 and 7 of a row's K-step are padding and are not fetched (a quarter of the piece's L2 -> LDS traffic)
  unsigned long long dma_mask;
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
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(lane4), "s"(pb), "s"(m0v) : "memory");
  };
  // LDS-DMA, saddr form: 32-bit lane offset + scalar base, M0 = LDS address of lane 0's 16 bytes.  Hand-issued so the
  // compiler neither forms 64-bit VGPR addresses nor tracks these in its waitcnt model (see v5).
  auto issue_piece = [&](const _Float16* pA, u32 slot_off, int it) {   // pA: first row of the tile, at the K-step to fetch
    const _Float16* sb = pA + (int64_t)it * 32 * ldk;
    const u32 m0v = lds_base + slot_off + (it * 256 + wave * 64) * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" : : "v"(g_off0), "s"(sb), "s"(m0v), "s"(dma_mask) : "memory");
  };
  // K loop form.  A single wavefront issues one instruction every ~4 cycles, i.e. 16 per slot of two MFMAs, and every slot already carries a wait, two
  // LDS reads and a VMEM operation: the piece's row offset lives in the LANE offset (8 registers, set per tile), so that all 8 pieces of a K-step share
  // one scalar base, and M0 is written by ONE s_add from the slot's base (v7: 64-bit pointer add + 32-bit add + s_mov per piece)
  u32 g_offp[NPIECE];

  // the same in two halves for the K loop: address + M0 behind one MFMA, the DMA instruction alone behind the next (the MFMA
  // between them also provides the wait state M0 needs; nothing else in the loop touches M0 - checked in the ISA)
#ifndef EPS_F6_DMA_MASK
#define EPS_F6_DMA_MASK 0   // (lab) 1: the padding granules are not fetched (exec-masked DMA: two more scalar instructions per piece)
#endif
#define EPS_F6_PREP_PIECE(M0BASE_, IT_) asm volatile("s_add_u32 m0, %0, %1" : : "s"(M0BASE_), "n"((IT_) * 4096) : "memory", "scc")
  auto fire_piece = [&](const _Float16* sb, u32 goff) __attribute__((always_inline)) {
    if (EPS_F6_DMA_MASK) asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" : : "v"(goff), "s"(sb), "s"(dma_mask) : "memory");
    else asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(goff), "s"(sb) : "memory");
  };

  acc_t acc[NRB][JQ];
  // THREE K-steps of query fragments in registers: the second pair of a step's buffer is only free when the step ends, so it is reloaded during the
  // NEXT step (no burst of loads between the last MFMA of a step and its barrier); everything a step issues is needed two or three K-steps on.
  // (Measured: neither this, nor vmcnt(32) instead of (16), nor 7 instead of 12 instructions per slot, nor leaving the padding granules out of the
  // DMA moved the K loop by more than 3 %: profiles/r5_fp6_filter_kernel_lab.txt.)
  F6Frag fb[3][2][JQ];   // [K-step mod 3][MFMA of the K-step][query block]
  F6Frag fa[2][NRB];
  int64_t qj[JQ];
  // Tq = T/s: a row passes iff acc >= Tq (s < 0); cj: approx-mode constant of the query.  Per-lane constants of the
  // query tile: parked in LDS and read back at each epilogue - as registers they would be live across the K loop, get
  // spilled, and their scratch reload would again drain the VMEM queue (vmcnt(0)) once per tile.
  float* tq_lds = base_lds + 512 + wave * 256;   // [4][64] per wavefront: Tq0, Tq1, cj0, cj1
#pragma unroll
  for (int j = 0; j < JQ; ++j) {
    qj[j] = (int64_t)qslot * QT + wave * (32 * JQ) + j * 32 + l31;
    tq_lds[j * 64 + lane] = a.T[qj[j]];
    tq_lds[(2 + j) * 64 + lane] = 0.f;
  }
  // Everything derived from the lane id that the K loop keeps in registers is RE-DERIVED at the top of every tile from
  // v_mbcnt (a dozen VALU instructions): values that live across the tile loop get spilled around the epilogue's
  // register peak, and a scratch reload makes hipcc wait vmcnt(0) - which drains the LDS-DMA ring once per tile.
  u32 fal[2], fah[2], lane8;   // LDS addresses of a lane's fragment reads (MFMA m of a K-step: bytes 0..15, bytes 16..23), its offset in the second half of a query-fragment block
  auto lane_values = [&]() __attribute__((always_inline)) {
    u32 ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    lane16 = ln * 16;
    lane8 = ln * 8;
    lane4 = ln * 4;
    const u32 td = (u32)wave * 64 + ln, row = td >> 3;
    g_off0 = (row * (u32)ldk + ((td & 7) ^ ((row >> 1) & 7)) * 8) * 2;
    dma_mask = __ballot(((td & 7) ^ ((row >> 1) & 7)) < 6);
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) g_offp[it] = g_off0 + (u32)it * 32u * (u32)ldk * 2u;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      fal[m] = lds_base + (u32)swz((int)(ln & 31), m * 2 + (int)(ln >> 5)) * 16;
      fah[m] = lds_base + (u32)swz((int)(ln & 31), 4 + m) * 16 + (ln >> 5) * 8;
    }
  };
  lane_values();

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
  for (int st = 0; st < 3; ++st) {   // the issue groups of the imaginary steps -3, -2, -1: 8 pieces + 8 fragment loads each
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) issue_piece(A_t + st * 64, st * ASLOT, it);
#pragma unroll
    for (int m = 0; m < (st == 2 ? 1 : 2); ++m)   // (step 2's second pair is loaded by step 0, as every step loads the previous step's buffer's)
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        EPS_F6_GL_LO(fb[st][m][j], lane16, B_t + st * 2048 + j * jstride + m * 1024);
        EPS_F6_GL_HI(fb[st][m][j], lane8, B_t + st * 2048 + j * jstride + m * 1024);
      }
  }
  asm volatile("s_waitcnt vmcnt(28)" ::: "memory");   // slot 0 + the fragments of step 0 (slots 1, 2 and the fragments of steps 1, 2 may stay in flight: 8 + 8 + 8 + 4)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifndef EPS_F6_ABL
#define EPS_F6_ABL 0   // lab ablations of the K loop (answers are wrong): 1 no LDS-DMA, 2 no 8-byte fragment reads, 4 no fragment reads at all, 8 no query-fragment loads
#endif
#define EPS_F6_READ_FRAG(DST_, AL_, AH_, I_)                          \
  if (!(EPS_F6_ABL & 4)) EPS_F6_DS_LO(DST_, AL_, (I_) * 4096);        \
  if (!(EPS_F6_ABL & 6)) EPS_F6_DS_HI(DST_, AH_, (I_) * 4096);
  EPS_F6_READ_FRAG(fa[0][0], fal[0], fah[0], 0)
  EPS_F6_READ_FRAG(fa[0][1], fal[0], fah[0], 1)
  EPS_F6_READ_FRAG(fa[0][2], fal[0], fah[0], 2)
  EPS_F6_READ_FRAG(fa[0][3], fal[0], fah[0], 3)
  EPS_F6_READ_FRAG(fa[0][NRB - 4], fal[0], fah[0], 4)
  EPS_F6_READ_FRAG(fa[0][NRB - 3], fal[0], fah[0], 5)
  EPS_F6_READ_FRAG(fa[0][NRB - 2], fal[0], fah[0], 6)
  EPS_F6_READ_FRAG(fa[0][NRB - 1], fal[0], fah[0], 7)

  // One wavefront per SIMD.  A K-step is 2 sub-steps (m = 0, 1: one K = 64 MFMA per row block and query block each) x 8 row blocks = 16 slots of
  // two MFMAs (64 cycles of matrix pipe); in the shadow of a slot's MFMAs the wavefront issues the two LDS reads of the row fragment the same
  // slot needs in the NEXT sub-step (16 reads = 8 slots ahead, waited for by count: lgkmcnt(14)) and one VMEM operation:
  //   m = 0: slot i fires LDS-DMA piece i of the K-step three ahead;  m = 1: slots 0..3 reload the first fragment pair of THIS step's buffer (for the
  //   K-step three ahead), slots 4..7 the second pair of the PREVIOUS step's buffer (for the K-step two ahead) - 16 VMEM operations per K-step, as in
  //   v7, one per slot; a slot is 7 instructions besides its MFMAs (wait, 2 LDS reads, s_add m0 / pointer select, VMEM).
  const _Float16* pA_run = A_t + 3 * 64;          // KT >= 4
  int akt_run = 3;
  const _Float16* pB_run = B_t + (int64_t)3 * 2048;   // (the fragments a step loads are those of the K-step three ahead)
  int bkt_run = 3;
  const _Float16* pBp_run = B_t + (int64_t)2 * 2048;  // the block before it: the second fragment pair of the PREVIOUS step's buffer is reloaded from there
  u32 sA_run = 0, sN_run = ASLOT, sD_run = 3 * ASLOT;   // slot being multiplied, the next one, the one being filled (byte offsets)
  u32 adl_run = fal[1], adh_run = fah[1];               // LDS addresses of the first sub-step's fragment reads (sub-step 1 of slot 0)
  auto step = [&](auto U, auto FIRST) __attribute__((always_inline)) {
    constexpr int rb = decltype(U)::value % 3;
    constexpr bool first = decltype(FIRST)::value;   // first K-step of a tile: only acc[.][0] holds the rows' start values
    const u32 sA = sA_run, sN = sN_run, sD = sD_run;
    (void)sA;
    const _Float16* pA = pA_run;
    const _Float16* pB = pB_run;
    const _Float16* pBp = pBp_run + 1024;          // (second fragment pair of the K-step before pB's)
    const _Float16* pB1 = pB + jstride;
    const _Float16* pBp1 = pBp + jstride;
    const u32 m0base = lds_base + sD + (u32)wave * 1024u;   // LDS address of this wavefront's lanes in piece 0 of the slot being filled
    const _Float16* pA_nx = pA;
    const _Float16* pB_nx = pB;
    int akt_nx = akt_run, bkt_nx = bkt_run;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int cur = m, nxt = m ^ 1;
      const u32 adl = adl_run, adh = adh_run;   // LDS addresses of this sub-step's fragment reads (the NEXT sub-step's operands)
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        const bool has_dma = m == 0;
        const bool has_frag = m == 1 && i < 2 * JQ;   // (fragment (0, i >> 1) of the K-step three ahead: low 16 bytes at even i, high 8 at odd i)
        const bool has_prev = m == 1 && i >= NRB - 2 * JQ;   // (fragment (1, .) of the PREVIOUS step's buffer - free during all of this step - for the K-step two ahead)
        const _Float16* vsrc = nullptr;
        asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (first && m == 0) acc[i][JQ - 1] = f6_mfma(fa[cur][i], fb[rb][m][JQ - 1], acc[i][0]);   // D != C: the second block's accumulator is born from the first block's start value
        else acc[i][0] = f6_mfma(fa[cur][i], fb[rb][m][0], acc[i][0]);
        __builtin_amdgcn_sched_barrier(0);
        switch (i) {
          case 0: EPS_F6_READ_FRAG(fa[nxt][0], adl, adh, 0) break;
          case 1: EPS_F6_READ_FRAG(fa[nxt][1], adl, adh, 1) break;
          case 2: EPS_F6_READ_FRAG(fa[nxt][2], adl, adh, 2) break;
          case 3: EPS_F6_READ_FRAG(fa[nxt][3], adl, adh, 3) break;
          case 4: EPS_F6_READ_FRAG(fa[nxt][NRB - 4], adl, adh, 4) break;
          case 5: EPS_F6_READ_FRAG(fa[nxt][NRB - 3], adl, adh, 5) break;
          case 6: EPS_F6_READ_FRAG(fa[nxt][NRB - 2], adl, adh, 6) break;
          default: EPS_F6_READ_FRAG(fa[nxt][NRB - 1], adl, adh, 7) break;
        }
        if (has_frag) {
          vsrc = (i >> 1) ? pB1 : pB;
          asm volatile("" : "+s"(vsrc));
        } else if (has_prev) {
          vsrc = ((i - (NRB - 2 * JQ)) >> 1) ? pBp1 : pBp;
          asm volatile("" : "+s"(vsrc));
        } else if (has_dma) {
          switch (i) {
            case 0: EPS_F6_PREP_PIECE(m0base, 0); break;
            case 1: EPS_F6_PREP_PIECE(m0base, 1); break;
            case 2: EPS_F6_PREP_PIECE(m0base, 2); break;
            case 3: EPS_F6_PREP_PIECE(m0base, 3); break;
            case 4: EPS_F6_PREP_PIECE(m0base, 4); break;
            case 5: EPS_F6_PREP_PIECE(m0base, 5); break;
            case 6: EPS_F6_PREP_PIECE(m0base, 6); break;
            default: EPS_F6_PREP_PIECE(m0base, 7); break;
          }
        }
        if (i == NRB - 1) {   // the next sub-step's read addresses: it fetches sub-step m of the NEXT slot either way (m = 0: (s + 1, 0); m = 1: (s + 1, 1))
          u32 anl = fal[m] + sN, anh = fah[m] + sN;
          asm volatile("" : "+v"(anl), "+v"(anh));
          adl_run = anl;
          adh_run = anh;
        }
        if (m == 1 && i == NRB - 2) {   // the next step's cursors
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
        if (first && m == 0) acc[i][0] = f6_mfma(fa[cur][i], fb[rb][m][0], acc[i][0]);
        else acc[i][JQ - 1] = f6_mfma(fa[cur][i], fb[rb][m][JQ - 1], acc[i][JQ - 1]);
        __builtin_amdgcn_sched_barrier(0);
        // (all loads of the loop stay in straight-line code, see v7)
        if (has_frag) {
          if (!(EPS_F6_ABL & 8)) {
            if ((i & 1) == 0) EPS_F6_GL_LO(fb[rb][0][(i >> 1) % JQ], lane16, vsrc);
            else EPS_F6_GL_HI(fb[rb][0][(i >> 1) % JQ], lane8, vsrc);
          }
        } else if (has_prev) {
          if (!(EPS_F6_ABL & 8)) {
            if ((i & 1) == 0) EPS_F6_GL_LO(fb[(rb + 2) % 3][1][((i - (NRB - 2 * JQ)) >> 1) % JQ], lane16, vsrc);
            else EPS_F6_GL_HI(fb[(rb + 2) % 3][1][((i - (NRB - 2 * JQ)) >> 1) % JQ], lane8, vsrc);
          }
        } else if (has_dma) {
          if (!(EPS_F6_ABL & 1)) fire_piece(pA, g_offp[i]);
        }
      }
    }
    pBp_run = pB;
    pA_run = pA_nx;
    akt_run = akt_nx;
    pB_run = pB_nx;
    bkt_run = bkt_nx;
    sA_run = sN;
    sN_run = (sN + ASLOT) & (RING * ASLOT - 1);
    sD_run = sA;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // this step's 8 DMA pieces + 8 fragment loads may stay in flight; what it issued is needed two (second pairs) or three K-steps on
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

#ifdef EPS_F6_PROF
  unsigned long long pf_head = 0, pf_k = 0, pf_epi = 0;
#endif
  for (int64_t t = 0; t < ntile; ++t) {
#ifdef EPS_F6_PROF
    const unsigned long long pf_t0 = __builtin_readcyclecounter();
#endif
    const int64_t row0 = (a.tile0 + tile_rt(ri_c)) * TR;
    const int64_t qbase = (int64_t)tile_qt(qi_c) * QT + wave * (32 * JQ);   // scalar
    lane_values();
    adl_run = fal[1] + sA_run;   // (re-derived with the lane values: nothing lane-dependent lives across the epilogue)
    adh_run = fah[1] + sA_run;
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
        tq_lds[j * 64 + (lane16 >> 4)] = a.T[qjt];
        float cm;   // (materialised here from a scalar: as an ordinary value hipcc keeps it in a VGPR across the tile loop and spills it)
        {
          const int sv = a.metric == 1 ? 0x3f800000 : 0;   // 1.0f : 0.0f
          asm volatile("v_mov_b32 %0, %1" : "=v"(cm) : "s"(sv));
        }
        tq_lds[(2 + j) * 64 + (lane16 >> 4)] = MODE != FM_IDS ? cm : 0.f;
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
#ifdef EPS_F6_PROF
    const unsigned long long pf_t1 = __builtin_readcyclecounter();
#endif
    step(std::integral_constant<int, 0>{}, std::true_type{});   // (KT is a multiple of 3: the fragment buffer of a K-step is a compile-time index)
    step(std::integral_constant<int, 1>{}, std::false_type{});
    step(std::integral_constant<int, 2>{}, std::false_type{});
    for (int kt = 3; kt < KT; kt += 3) {
      step(std::integral_constant<int, 0>{}, std::false_type{});
      step(std::integral_constant<int, 1>{}, std::false_type{});
      step(std::integral_constant<int, 2>{}, std::false_type{});
    }
#ifdef EPS_F6_PROF
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
#pragma unroll
    for (int j = 0; j < JQ; ++j) {
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
    if (__builtin_expect(tile_hit, 0)) {
#pragma unroll
    for (int i0 = 0; i0 < NRB; i0 += GROUP) {
    // r4 (GROUP > 1): the maxima of GROUP row blocks x JQ query blocks are combined per query column and tested ONCE - one compare
    // pair + one untaken branch per group instead of one per block (lab ablation: the 16 compare + branch pairs cost the launch 3 %);
    // unlike the tile-level test nothing is computed twice: a group that passes re-uses its blocks' maxima for the per-block tests
    thr_t mxg[GROUP][JQ];
    bool group_hit = true;
    if (GROUP > 1 && MODE != FM_DENSE) {
      bool h = false;
#pragma unroll
      for (int j = 0; j < JQ; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        thr_t gm = mxg[0][j] = max16f(acc[i0][j]);
#pragma unroll
        for (int ii = 1; ii < GROUP; ++ii) {
          __builtin_amdgcn_sched_barrier(0);
          mxg[ii][j] = max16f(acc[i0 + ii][j]);
          gm = gm > mxg[ii][j] ? gm : mxg[ii][j];
        }
        h |= gm >= Tq[j];
      }
      group_hit = __any(h);
    }
    if (GROUP == 1 || MODE == FM_DENSE || __builtin_expect(group_hit, 0)) {
#pragma unroll
    for (int ii = 0; ii < GROUP; ++ii) {
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
        const thr_t mx = GROUP > 1 ? mxg[ii][j] : max16f(acc[i][j]);
        if (__any(mx >= Tq[j])) {
          // (rare) everything the hit path needs is derived behind this opaque copy of the lane id, or hipcc hoists the
          // address arithmetic of all 16 blocks into the common path
          int l31h = l31e, rbh = rbase;
          asm volatile("" : "+v"(l31h), "+v"(rbh));
          const int64_t qq = qbase + j * 32 + l31h;
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
    if (MODE != FM_DENSE) {
      if (*wcnt_lds >= (u32)(V7_CAPW / 2)) flush();
    }
    }   // tile_hit
#ifdef EPS_F6_PROF
    const unsigned long long pf_t3 = __builtin_readcyclecounter();
    pf_head += pf_t1 - pf_t0;
    pf_k += pf_t2 - pf_t1;
    pf_epi += pf_t3 - pf_t2;
#endif
  }
#ifdef EPS_F6_PROF
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



}  // namespace eps
