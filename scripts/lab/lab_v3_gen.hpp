// GENERATED from vectordb_amd/csrc/mfma_kernels.hpp (v3) by scripts/lab/gen_lab_v3.py: the shipped kernel with compile-time ablation knobs
#pragma once
namespace eps {
template <int KNOB>  // 2 no MFMA, 4 no LDS fragment reads, 8 no B DMA, 16 no A DMA, 32 no barrier, 64 no epilogue, 128 no init
__global__ __launch_bounds__(512, 2) void lab_v3(FilterArgs a) {
  constexpr int ablate = KNOB;
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
      if (!(ablate & 32)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt == 0 && !(ablate & 128)) {
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
if (!(ablate & 64))
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
              if (row < a.row_hi && qj[j] < a.nq && !(ablate & 1)) {
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

}  // namespace eps
