// Random row-gather ceiling of one MI355X (lab; not part of the product): what HBM delivers to a kernel that does NOTHING but what the
// traversal's distance phase does to memory - every lane group reads whole rows of S bytes at random offsets of a table far larger than
// the caches, 16 bytes per lane, R rows in flight per group - with no queues, no visited set, no arithmetic beyond keeping the loads
// alive.  traverse2_kernel's achieved rate is judged against THIS number, not against the 8 TB/s streaming peak (MI355X_MICROARCH.md
// lists 6.29 TB/s for a float4 copy).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gather_peak.hip -o scripts/lab/gather_peak
//   scripts/lab/gather_peak [table_GB=30] [rows_per_launch=33554432]
// Prints, for row sizes 768 B (an 8-bit mirror row at d = 768), 3072 B (an fp32 row) and the traversal's mix (per fp32 row ~6 mirror
// rows), GB/s at 2 / 4 / 8 rows in flight per lane group and 4 / 8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                                               \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {   // splitmix64
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// G lanes per row (G * 16 * PIECES = S bytes), R rows in flight per group; each group walks `per_group` random rows
template <int G, int PIECES, int R>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* table, uint64_t table_rows, int64_t per_group, uint32_t seed, uint4* sink) {
  const int lane = threadIdx.x & 63;
  const int t = lane & (G - 1);
  const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  constexpr int ROW16 = G * PIECES;   // 16-byte units per row
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int64_t i = 0; i < per_group; i += R) {
    uint4 v[R][PIECES];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t row = mix(((uint64_t)group << 24) ^ (uint64_t)(i + r) ^ ((uint64_t)seed << 48)) % table_rows;
      const uint4* p = table + row * ROW16 + t;
#pragma unroll
      for (int c = 0; c < PIECES; ++c) v[r][c] = p[c * G];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < PIECES; ++c) {
        acc.x ^= v[r][c].x;
        acc.y += v[r][c].y;
        acc.z ^= v[r][c].z;
        acc.w += v[r][c].w;
      }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // (never true: keeps the loads)
}

template <int G, int PIECES, int R>
static double run(const uint4* table, uint64_t bytes, int64_t rows_total, int waves_per_simd, uint4* sink, const char* what) {
  const uint64_t S = (uint64_t)G * PIECES * 16;
  const uint64_t table_rows = bytes / S;
  const int blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs x waves / 4 waves per block
  const int64_t groups = (int64_t)blocks * 256 / G;
  int64_t per_group = (rows_total + groups - 1) / groups;
  per_group = (per_group + R - 1) / R * R;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_kernel<G, PIECES, R>), dim3(blocks), dim3(256), 0, 0, table, table_rows, per_group, (uint32_t)rep, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = (double)groups * per_group * S / (ms * 1e-3) / 1e9;
    if (rep > 0 && gbs > best) best = gbs;
  }
  printf("%-34s S=%5llu B  %d lanes/row  %d rows in flight  %d waves/SIMD: %8.0f GB/s\n", what, (unsigned long long)S, G, R, waves_per_simd, best);
  return best;
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 30.0;
  const int64_t rows = argc > 2 ? atoll(argv[2]) : (int64_t)1 << 25;
  const uint64_t bytes = (uint64_t)(gb * 1e9) / 49152 * 49152;
  uint4* table = nullptr;
  uint4* sink = nullptr;
  CHECK(hipMalloc(&table, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(table, 1, bytes));
  CHECK(hipDeviceSynchronize());
  printf("table %.1f GB, %lld rows gathered per launch\n", bytes / 1e9, (long long)rows);
  for (int w : {4, 8}) {
    // an 8-bit mirror row of d = 768: 768 B = 16 lanes x 3 pieces (the traversal's prefilter shape), or 48 lanes x 1 piece ~ 64 lanes
    if (w == 4) {
      run<16, 3, 2>(table, bytes, rows * 4, 4, sink, "mirror row, prefilter shape");
      run<16, 3, 4>(table, bytes, rows * 4, 4, sink, "mirror row, prefilter shape");
      run<16, 3, 8>(table, bytes, rows * 4, 4, sink, "mirror row");
      run<64, 3, 2>(table, bytes, rows, 4, sink, "fp32 row, one wavefront per row");
      run<64, 3, 4>(table, bytes, rows, 4, sink, "fp32 row, one wavefront per row");
    } else {
      run<16, 3, 2>(table, bytes, rows * 4, 8, sink, "mirror row, prefilter shape");
      run<16, 3, 4>(table, bytes, rows * 4, 8, sink, "mirror row, prefilter shape");
      run<64, 3, 2>(table, bytes, rows, 8, sink, "fp32 row, one wavefront per row");
      run<64, 3, 4>(table, bytes, rows, 8, sink, "fp32 row, one wavefront per row");
    }
  }
  CHECK(hipFree(table));
  CHECK(hipFree(sink));
  return 0;
}
