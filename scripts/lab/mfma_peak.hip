// How fast can ONE wavefront per SIMD issue v_mfma_f32_32x32x16_f16 (vs two per SIMD)?  hipcc --offload-arch=gfx950 -O3 scripts/lab/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// same loop, but consecutive MFMAs read DIFFERENT operand registers holding random fp16 data (8 A and 8 B fragments,
// generated once): the matrix pipe's inputs toggle as they do on real data, with no VALU work in the loop
template <int NACC, int MODE>   // MODE 0: random sign/mantissa, |v| in [1,2);  1: v ~ U[0,1) like the benchmark's rows and queries
__global__ __launch_bounds__(1024) void peak_rand(float* out, int iters, long long* clk) {
  unsigned x = threadIdx.x * 2654435761u + 12345u + blockIdx.x * 977u;
  half8 A[8], B[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 ua, ub;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      ua[w] = (x & 0x83FF83FFu) | 0x3C003C00u;
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      ub[w] = (x & 0x83FF83FFu) | 0x3C003C00u;
    }
    A[f] = __builtin_bit_cast(half8, ua);
    B[f] = __builtin_bit_cast(half8, ub);
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        A[f][e] = (_Float16)((float)(A[f][e] < (_Float16)0 ? -A[f][e] : A[f][e]) - 1.0f);
        B[f][e] = (_Float16)((float)(B[f][e] < (_Float16)0 ? -B[f][e] : B[f][e]) - 1.0f);
      }
    }
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(u + i) & 7], B[(u * 3 + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - w0;
  }
}
template <int NACC, int MODE>
static void run_rand(const char* name, int threads, int iters) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  long long* clk; hipMalloc(&clk, 16);
  hipLaunchKernelGGL((peak_rand<NACC, MODE>), dim3(256), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((peak_rand<NACC, MODE>), dim3(256), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flop = 256.0 * (threads / 64) * (double)iters * NACC * 32768.0;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-52s %8.3f ms  %7.1f TFLOP/s   shader clock %.2f GHz\n", name, ms, flop / ms * 1e-9, (double)h[0] / ((double)h[1] * 10.0));
}
template <int NACC>
__global__ __launch_bounds__(1024) void peak(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
static void run(const char* name, int threads, int iters) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(peak<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(peak<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flop = 256.0 * (threads / 64) * (double)iters * NACC * 32768.0;
  printf("%-40s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms * 1e-9);
}
int main() {
  run<4>("1 wave/SIMD, 4 accumulators", 256, 160000);
  run<4>("2 waves/SIMD, 4 accumulators each", 512, 80000);
  run<4>("4 waves/SIMD, 4 accumulators each", 1024, 40000);
  run_rand<4, 0>("random sign/mantissa in [1,2), 1 wave/SIMD", 256, 160000);
  run_rand<4, 0>("random sign/mantissa in [1,2), 4 waves/SIMD", 1024, 40000);
  run_rand<4, 1>("U[0,1) operands, 1 wave/SIMD", 256, 160000);
  run_rand<4, 1>("U[0,1) operands, 2 waves/SIMD", 512, 80000);
  run_rand<4, 1>("U[0,1) operands, 4 waves/SIMD", 1024, 40000);
  return 0;
}
