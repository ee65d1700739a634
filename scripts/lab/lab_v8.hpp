// v8 candidate (v7 + the last K-step of every tile block-major with the epilogue interleaved): 
// v7 candidate: 256 x 256 tile, FOUR wavefronts (one per SIMD, up to 512 registers each), wavefront w = all 256 rows x
// queries [64w, 64w+64) = 8 x 2 tiles of v_mfma_f32_32x32x16_f16.  Same operand transport as v5 (row operand: 4-slot
// LDS-DMA ring with counted waits; query operand: fragment-major, straight to VGPRs, two steps ahead) but every row
// fragment read from LDS now feeds two MFMAs instead of one: half the LDS read traffic per flop.
#pragma once
namespace eps {

template <int KNOB>
__global__ __launch_bounds__(256, 1) void lab_v8(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int ASLOT = 32768;  // 256 rows x 128 B
  constexpr int RING = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5;
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

  int g_off[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int s = it * 256 + tid;
    const int row = s >> 3;
    g_off[it] = row * ldk + ((s & 7) ^ ((row >> 1) & 7)) * 8;
  }
  auto tile_rt = [&](int64_t t) { return (int64_t)xcd + 8 * (rg + (t / nqt) * G); };
  auto tile_qt = [&](int64_t t) { return qslot + (int)(t % nqt) * QTB; };
  auto rows_of = [&](int64_t t) { return a.xh + (a.tile0 + tile_rt(t)) * 256 * (int64_t)ldk; };
  // fragment stream of this wavefront's first 32-query block; the second block follows at + (ldk/16)*512 halfs
  auto frags_of = [&](int64_t t) { return a.qf + ((int64_t)(tile_qt(t) * 8 + wave * 2) * (ldk / 16)) * 512; };
  const int64_t jstride = (int64_t)(ldk / 16) * 512;
  const u32 lane16 = lane * 16;
  auto issue_base = [&](int64_t t) {
    const float* pb = a.base_s + (a.tile0 + tile_rt(t)) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + wave * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(base_lds + (t & 1) * 256 + wave * 64), 4, 0, 0);
  };
  u32 g_off2[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) g_off2[it] = (u32)g_off[it] * 2;
  const u32 lds_base = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue_piece = [&](const _Float16* pA, int kt, int slot, int it) {
    if (KNOB & 2048) {
      // saddr form: 32-bit lane offset + scalar base, M0 = LDS destination of lane 0
      const _Float16* sb = pA + kt * 64;
      const u32 m0v = lds_base + slot * ASLOT + (it * 256 + wave * 64) * 16;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(g_off2[it]), "s"(sb), "s"(m0v) : "memory");
    } else if (!(KNOB & 16))
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pA + g_off[it] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(lds + slot * ASLOT + (it * 256 + wave * 64) * 16), 16, 0, 0);
  };

  f32x16 acc[8][2];
  half8 fb[2][4][2];
  half8 F[16];   // row fragments: normal steps use F[(kk & 1) * 8 + pair]; the block-major last step F[(pair & 3) * 4 + kk]
  const float inv_s = 1.0f / a.s;
  int64_t qj[2];
  float Tq[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    qj[j] = (int64_t)qslot * 256 + wave * 64 + j * 32 + l31;
    Tq[j] = a.T[qj[j]] * inv_s;
  }
  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = swz(l31, kk * 2 + khalf) * 16;

  const _Float16* A_t = rows_of(0);
  const _Float16* A_n = ntile > 1 ? rows_of(1) : A_t;
  const _Float16* B_t = frags_of(0);
  const _Float16* B_n = (nqt > 1 && ntile > 1) ? frags_of(1) : B_t;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue_base(0);
  // prologue = the issue groups of the imaginary steps -3, -2, -1 (16 operations each from -2 on)
#pragma unroll
  for (int it = 0; it < 8; ++it) issue_piece(A_t, 0, 0, it);
#pragma unroll
  for (int it = 0; it < 8; ++it) issue_piece(A_t, 1, 1, it);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    EPS_GLOAD_B128(fb[0][kk][0], lane16, B_t + kk * 512, 0);
    EPS_GLOAD_B128(fb[0][kk][1], lane16, B_t + jstride + kk * 512, 0);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) issue_piece(A_t, 2, 2, it);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    EPS_GLOAD_B128(fb[1][kk][0], lane16, B_t + 2048 + kk * 512, 0);
    EPS_GLOAD_B128(fb[1][kk][1], lane16, B_t + jstride + 2048 + kk * 512, 0);
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // slots 0 and 1 + fragments of step 0
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  u32 faddr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) faddr[kk] = lds0 + foff[kk];
  EPS_DS_READ_B128(F[0], faddr[0], 0);
  EPS_DS_READ_B128(F[1], faddr[0], 4096);
  EPS_DS_READ_B128(F[2], faddr[0], 8192);
  EPS_DS_READ_B128(F[3], faddr[0], 12288);
  EPS_DS_READ_B128(F[4], faddr[0], 16384);
  EPS_DS_READ_B128(F[5], faddr[0], 20480);
  EPS_DS_READ_B128(F[6], faddr[0], 24576);
  EPS_DS_READ_B128(F[7], faddr[0], 28672);

  int slot = 0;
  // One wavefront per SIMD: after every pair of MFMAs (64 cycles of matrix pipe) exactly one other instruction is
  // issued in its shadow - the LDS read of the row fragment that the same pair will need in the NEXT sub-step
  // (8 pairs = 512 cycles ahead, waited for by count: lgkmcnt(7)), and on odd pairs one LDS-DMA piece / fragment load.
  auto step = [&](int kt, auto U, auto PRELAST) __attribute__((always_inline)) {
    constexpr int rb = decltype(U)::value;
    constexpr bool pre_last = decltype(PRELAST)::value;   // the next step is the tile's block-major last step
    const int nslot = (slot + 1) & 3;
    const int dslot = (slot + 3) & 3;
    const u32 sA = slot * ASLOT, sN = nslot * ASLOT;
    const _Float16* pA = kt + 3 < KT ? A_t : A_n;
    const int akt = kt + 3 < KT ? kt + 3 : kt + 3 - KT;
    const _Float16* pB = (kt + 2 < KT ? B_t : B_n) + (int64_t)((kt + 2 < KT ? kt + 2 : kt + 2 - KT) * 4) * 512;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = (kk & 1) * 8, nxt = ((kk + 1) & 1) * 8;
      u32 ad[4];
      if (kk < 3) {
        ad[0] = faddr[kk + 1] + sA;
      } else if (!pre_last) {
        ad[0] = faddr[0] + sN;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) ad[q] = faddr[q] + sN;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[cur + i], fb[rb][kk][0], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[cur + i], fb[rb][kk][1], acc[i][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 3 && pre_last) {   // F[i] <- (pair i >> 2, sub-step i & 3) of the next slot
          if (i < 4) EPS_DS_READ_B128(F[nxt + i], ad[i & 3], 0);
          else EPS_DS_READ_B128(F[nxt + i], ad[i & 3], 4096);
        } else {
          switch (i) {
            case 0: EPS_DS_READ_B128(F[nxt + 0], ad[0], 0); break;
            case 1: EPS_DS_READ_B128(F[nxt + 1], ad[0], 4096); break;
            case 2: EPS_DS_READ_B128(F[nxt + 2], ad[0], 8192); break;
            case 3: EPS_DS_READ_B128(F[nxt + 3], ad[0], 12288); break;
            case 4: EPS_DS_READ_B128(F[nxt + 4], ad[0], 16384); break;
            case 5: EPS_DS_READ_B128(F[nxt + 5], ad[0], 20480); break;
            case 6: EPS_DS_READ_B128(F[nxt + 6], ad[0], 24576); break;
            default: EPS_DS_READ_B128(F[nxt + 7], ad[0], 28672); break;
          }
        }
        if (i == 1 || i == 3) {
          if (kk > 0) EPS_GLOAD_B128(fb[rb][kk - 1][i >> 1], lane16, pB + (i >> 1) * jstride + (kk - 1) * 512, 0);
        } else if (i == 5 || i == 7) {
          issue_piece(pA, akt, dslot, kk * 2 + (i >> 1) - 2);
        }
      }
    }
    EPS_GLOAD_B128(fb[rb][3][0], lane16, pB + 3 * 512, 0);
    EPS_GLOAD_B128(fb[rb][3][1], lane16, pB + jstride + 3 * 512, 0);
    slot = nslot;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // The threshold test of one 32 x 32 output block, in two halves so that each half fits the shadow of two MFMAs.
  float mxs = 0.f;
  auto epi = [&](int i, int j, int part, int64_t row0) __attribute__((always_inline)) {
    if (part == 0) {
      mxs = acc[i][j][0];
#pragma unroll
      for (int r = 1; r < 8; ++r) mxs = fmaxf(mxs, acc[i][j][r]);
    } else {
      float mx = mxs;
#pragma unroll
      for (int r = 8; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
      if (__any(mx >= Tq[j])) {
        const int rbase = i * 32 + 4 * khalf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc[i][j][r] >= Tq[j]) {
            const int64_t row = row0 + rbase + (r & 3) + 8 * (r >> 2);
            if (row < a.row_hi && qj[j] < a.nq) {
              const u32 slot_c = atomicAdd(&a.cnt[qj[j]], 1u);
              if (slot_c < (u32)a.cap) a.cand[qj[j] * (int64_t)a.cap + slot_c] = (u32)row;
            }
          }
        }
      }
    }
  };

  // Last K-step of a tile (odd parity): block-major - pair i runs its four K=16 sub-steps back to back, so its 32 x 64
  // outputs are final 256 matrix-pipe cycles before pair i+1's, and their threshold test issues in the shadow of pair
  // i+1's MFMAs instead of after the tile with the pipe idle.  Fragment ring: pair i reads F[(i & 3) * 4 + kk], loaded two
  // pairs ahead; pairs 6 and 7 fetch the next step's (pair n, sub-step 0) fragments into F[0..7].
  auto last_step = [&](int kt, int64_t row0) __attribute__((always_inline)) {
    constexpr int rb = 1;
    const int nslot = (slot + 1) & 3;
    const int dslot = (slot + 3) & 3;
    const u32 sA = slot * ASLOT, sN = nslot * ASLOT;
    const _Float16* pA = kt + 3 < KT ? A_t : A_n;
    const int akt = kt + 3 < KT ? kt + 3 : kt + 3 - KT;
    const _Float16* pB = (kt + 2 < KT ? B_t : B_n) + (int64_t)((kt + 2 < KT ? kt + 2 : kt + 2 - KT) * 4) * 512;
    u32 adL[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) adL[q] = faddr[q] + sA;
    const u32 adN0 = faddr[0] + sN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[(i & 3) * 4 + kk], fb[rb][kk][0], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[(i & 3) * 4 + kk], fb[rb][kk][1], acc[i][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 6) {
          switch (i + 2) {
            case 2: EPS_DS_READ_B128(F[((i + 2) & 3) * 4 + kk], adL[kk], 8192); break;
            case 3: EPS_DS_READ_B128(F[((i + 2) & 3) * 4 + kk], adL[kk], 12288); break;
            case 4: EPS_DS_READ_B128(F[((i + 2) & 3) * 4 + kk], adL[kk], 16384); break;
            case 5: EPS_DS_READ_B128(F[((i + 2) & 3) * 4 + kk], adL[kk], 20480); break;
            case 6: EPS_DS_READ_B128(F[((i + 2) & 3) * 4 + kk], adL[kk], 24576); break;
            default: EPS_DS_READ_B128(F[((i + 2) & 3) * 4 + kk], adL[kk], 28672); break;
          }
        } else {
          switch ((i - 6) * 4 + kk) {
            case 0: EPS_DS_READ_B128(F[0], adN0, 0); break;
            case 1: EPS_DS_READ_B128(F[1], adN0, 4096); break;
            case 2: EPS_DS_READ_B128(F[2], adN0, 8192); break;
            case 3: EPS_DS_READ_B128(F[3], adN0, 12288); break;
            case 4: EPS_DS_READ_B128(F[4], adN0, 16384); break;
            case 5: EPS_DS_READ_B128(F[5], adN0, 20480); break;
            case 6: EPS_DS_READ_B128(F[6], adN0, 24576); break;
            default: EPS_DS_READ_B128(F[7], adN0, 28672); break;
          }
        }
        if (kk == 1) issue_piece(pA, akt, dslot, i);
        if (i > 0 && !(KNOB & 8192)) epi(i - 1, kk >> 1, kk & 1, row0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      EPS_GLOAD_B128(fb[rb][kk][0], lane16, pB + kk * 512, 0);
      EPS_GLOAD_B128(fb[rb][kk][1], lane16, pB + jstride + kk * 512, 0);
    }
    if (KNOB & 8192) {
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) epi(i, q >> 1, q & 1, row0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) epi(7, q >> 1, q & 1, row0);
    slot = nslot;
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  for (int64_t t = 0; t < ntile; ++t) {
    const int64_t row0 = (a.tile0 + tile_rt(t)) * 256;
    if (nqt > 1 && t > 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        qj[j] = (int64_t)tile_qt(t) * 256 + wave * 64 + j * 32 + l31;
        Tq[j] = a.T[qj[j]] * inv_s;
      }
    }
    if (t + 1 < ntile) issue_base(t + 1);
    {
      const float* bl0 = base_lds + (t & 1) * 256;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rbase = i * 32 + 4 * khalf;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 bv = *reinterpret_cast<const float4*>(&bl0[rbase + 8 * gq]);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j][4 * gq + 0] = bv.x;
            acc[i][j][4 * gq + 1] = bv.y;
            acc[i][j][4 * gq + 2] = bv.z;
            acc[i][j][4 * gq + 3] = bv.w;
          }
        }
      }
    }
    for (int kt = 0; kt < KT - 2; kt += 2) {
      step(kt, std::integral_constant<int, 0>{}, std::false_type{});
      step(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    step(KT - 2, std::integral_constant<int, 0>{}, std::true_type{});
    last_step(KT - 1, row0);
    A_t = A_n;
    B_t = B_n;
    if (t + 2 < ntile) {
      A_n = rows_of(t + 2);
      if (nqt > 1) B_n = frags_of(t + 2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace eps
