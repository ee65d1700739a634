// v6 candidate: v5 (query operand in registers, 4-slot LDS-DMA ring for the row operand) with an explicit ping-pong
// schedule between the two wavefronts of each SIMD.
#pragma once
namespace eps {

template <int KNOB>
__global__ __launch_bounds__(512, 2) void lab_v6(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int ASLOT = 32768;  // 256 rows x 128 B
  constexpr int RING = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: lets the fragment stream base live in SGPRs
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
  const int KT = ldk / 64;      // even and >= 4 (d_pad is a multiple of 128, >= 256)

  int g_off[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int s = it * 512 + tid;
    const int row = s >> 3;
    g_off[it] = row * ldk + ((s & 7) ^ ((row >> 1) & 7)) * 8;
  }
  auto tile_rt = [&](int64_t t) { return (int64_t)xcd + 8 * (rg + (t / nqt) * G); };
  auto tile_qt = [&](int64_t t) { return qslot + (int)(t % nqt) * QTB; };
  auto rows_of = [&](int64_t t) { return a.xh + (a.tile0 + tile_rt(t)) * 256 * (int64_t)ldk; };
  auto frags_of = [&](int64_t t) { return a.qf + ((int64_t)(tile_qt(t) * 8 + wave) * (ldk / 16)) * 512; };  // + lane * 8
  const u32 lane16 = lane * 16;
  auto issue_base = [&](int64_t t) {  // |x|^2 column of tile t -> base_lds[t & 1]
    const float* pb = a.base + (a.tile0 + tile_rt(t)) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + (wave & 3) * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(base_lds + (t & 1) * 256 + (wave & 3) * 64), 4, 0, 0);
  };
  auto issue_piece = [&](const _Float16* pA, int kt, int slot, int it) {
    if (!(KNOB & 16))
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pA + g_off[it] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(lds + slot * ASLOT + (it * 512 + wave * 64) * 16), 16, 0, 0);
  };

  f32x16 acc[8];
  half8 fb[2][4];
  half8 fa[8];
  const float inv_s = 1.0f / a.s;
  int64_t qj = (int64_t)qslot * 256 + wave * 32 + l31;
  float Tq = a.T[qj] * inv_s;

  int foff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) foff[kk] = swz(l31, kk * 2 + khalf) * 16;

  // The (tile, K-step) sequence is one pipeline.  During step g every wavefront issues, in this order interleaved with
  // its MFMAs: 4 LDS-DMA pieces of step g+3 (ring slot (g+3)%4) and the 4 query fragments of step g+2 (into the register
  // buffer step g is freeing).  Steps past the end re-read the last tile (harmless) so the loop body has no branches.
  const _Float16* A_t = rows_of(0);
  const _Float16* A_n = ntile > 1 ? rows_of(1) : A_t;
  const _Float16* B_t = frags_of(0);
  const _Float16* B_n = B_t;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the threshold load) keep the counted waits below exact
  issue_base(0);
  // prologue = the issue groups of the imaginary steps -3, -2, -1
#pragma unroll
  for (int it = 0; it < 4; ++it) issue_piece(A_t, 0, 0, it);
#pragma unroll
  for (int it = 0; it < 4; ++it) issue_piece(A_t, 1, 1, it);
  EPS_GLOAD_B128(fb[0][0], lane16, B_t, 0);
  EPS_GLOAD_B128(fb[0][1], lane16, B_t, 1024);
  EPS_GLOAD_B128(fb[0][2], lane16, B_t, 2048);
  EPS_GLOAD_B128(fb[0][3], lane16, B_t, 3072);
#pragma unroll
  for (int it = 0; it < 4; ++it) issue_piece(A_t, 2, 2, it);
  EPS_GLOAD_B128(fb[1][0], lane16, B_t + 2048, 0);
  EPS_GLOAD_B128(fb[1][1], lane16, B_t + 2048, 1024);
  EPS_GLOAD_B128(fb[1][2], lane16, B_t + 2048, 2048);
  EPS_GLOAD_B128(fb[1][3], lane16, B_t + 2048, 3072);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // slots 0 and 1 + fragments of step 0
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  u32 faddr[4];   // LDS address of this lane's granule of row l31, per K=16 sub-step, in the slot being computed
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) faddr[kk] = lds0 + foff[kk];
  const int grp = wave >> 2;   // wavefronts w and w+4 share a SIMD: group 0 computes in even phases, group 1 in odd ones
  auto frag_loads = [&](u32 ad) __attribute__((always_inline)) {
    EPS_DS_READ_B128(fa[0], ad, 0);
    EPS_DS_READ_B128(fa[1], ad, 4096);
    EPS_DS_READ_B128(fa[2], ad, 8192);
    EPS_DS_READ_B128(fa[3], ad, 12288);
    EPS_DS_READ_B128(fa[4], ad, 16384);
    EPS_DS_READ_B128(fa[5], ad, 20480);
    EPS_DS_READ_B128(fa[6], ad, 24576);
    EPS_DS_READ_B128(fa[7], ad, 28672);
  };

  int slot = 0;     // LDS ring slot of the step being computed
  // Ping-pong: a K-step is 8 phases separated by s_barrier.  In every phase one wavefront of each SIMD issues the 8
  // MFMAs of one K=16 sub-step while its partner reads the 8 fragments of its next sub-step from LDS and issues its
  // share of the LDS-DMA (row operand, three steps ahead) and of the query-fragment loads (two steps ahead).
  auto step = [&](int kt, auto U, auto GRP) __attribute__((always_inline)) {
    constexpr int rb = decltype(U)::value;
    constexpr int grp_c = decltype(GRP)::value;
    const int nslot = (slot + 1) & 3;
    const int dslot = (slot + 3) & 3;
    const u32 sA = slot * ASLOT, sN = nslot * ASLOT;
    const _Float16* pA = kt + 3 < KT ? A_t : A_n;
    const int akt = kt + 3 < KT ? kt + 3 : kt + 3 - KT;
    const _Float16* pB = (kt + 2 < KT ? B_t : B_n) + (int64_t)((kt + 2 < KT ? kt + 2 : kt + 2 - KT) * 4) * 512;
    auto compute = [&](int kk) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!(KNOB & 2)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[rb][kk], acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    if (grp_c == 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        compute(kk);
        bar();
        if (!(KNOB & 64)) frag_loads(faddr[(kk + 1) & 3] + (kk < 3 ? sA : sN));
        issue_piece(pA, akt, dslot, kk);
        if (!(KNOB & 32)) EPS_GLOAD_B128(fb[rb][kk], lane16, pB + kk * 512, 0);
        if (kk == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        bar();
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!(KNOB & 64)) frag_loads(faddr[kk] + sA);
        issue_piece(pA, akt, dslot, kk);
        if (kk > 0 && !(KNOB & 32)) EPS_GLOAD_B128(fb[rb][kk - 1], lane16, pB + (kk - 1) * 512, 0);
        bar();
        compute(kk);
        if (kk == 3) {
          if (!(KNOB & 32)) EPS_GLOAD_B128(fb[rb][3], lane16, pB + 3 * 512, 0);
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        bar();
      }
    }
    slot = nslot;
  };

  auto run = [&](auto GRP) __attribute__((always_inline)) {
    if (decltype(GRP)::value == 0) frag_loads(faddr[0]);
  for (int64_t t = 0; t < ntile; ++t) {
    const int64_t row0 = (a.tile0 + tile_rt(t)) * 256;
    if (nqt > 1 && t > 0) {
      qj = (int64_t)tile_qt(t) * 256 + wave * 32 + l31;
      Tq = a.T[qj] * inv_s;
    }
    if (t + 1 < ntile) issue_base(t + 1);   // a whole tile ahead; counted out by the next step's vmcnt(8)
    {
      const float* bl0 = base_lds + (t & 1) * 256;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rbase = i * 32 + 4 * khalf;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 bv = *reinterpret_cast<const float4*>(&bl0[rbase + 8 * gq]);
          acc[i][4 * gq + 0] = bv.x * inv_s;
          acc[i][4 * gq + 1] = bv.y * inv_s;
          acc[i][4 * gq + 2] = bv.z * inv_s;
          acc[i][4 * gq + 3] = bv.w * inv_s;
        }
      }
    }
    for (int kt = 0; kt < KT; kt += 2) {
      step(kt, std::integral_constant<int, 0>{}, GRP);
      step(kt + 1, std::integral_constant<int, 1>{}, GRP);
    }
    A_t = A_n;
    B_t = B_n;
    if (t + 2 < ntile) {
      A_n = rows_of(t + 2);
      if (nqt > 1) B_n = frags_of(t + 2);
    }
    // ---- epilogue of tile t (the next tile's first steps are already in flight)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rbase = i * 32 + 4 * khalf;
      float mx = acc[i][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[i][r]);
      if (__any(mx >= Tq)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc[i][r] >= Tq) {
            const int64_t row = row0 + rbase + (r & 3) + 8 * (r >> 2);
            if (row < a.row_hi && qj < a.nq) {
              const u32 slot_c = atomicAdd(&a.cnt[qj], 1u);
              if (slot_c < (u32)a.cap) a.cand[qj * (int64_t)a.cap + slot_c] = (u32)row;
            }
          }
        }
      }
    }
  }
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace eps
