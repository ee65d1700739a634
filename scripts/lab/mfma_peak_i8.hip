// What does the matrix pipe sustain on 8-bit integer operands (v_mfma_i32_32x32x32_i8), one wavefront per SIMD, on operands that
// toggle like a quantised U[0,1) table?  Beside it the fp16 loop of mfma_peak.hip, same launch shape, so the two rates and the
// two power-limited clocks can be read off one run.
//   hipcc --offload-arch=gfx950 -O3 scripts/lab/mfma_peak_i8.hip -o scripts/lab/mfma_peak_i8 && scripts/lab/mfma_peak_i8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned xs(unsigned& x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

// MODE 0: bytes uniform in [-127, 127] (a centred, fully used 8-bit grid); 1: bytes uniform in [0, 127] (uncentred grid: sign bit
// idle); 2: all zero (the clock ceiling)
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void peak_i8(int* out, int iters, long long* clk) {
  unsigned x = threadIdx.x * 2654435761u + 12345u + blockIdx.x * 977u;
  i32x4 A[8], B[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned a = 0, b = 0;
      for (int e = 0; e < 4; ++e) {
        int va = (int)(xs(x) % 255u) - 127, vb = (int)(xs(x) % 255u) - 127;
        if (MODE == 1) { va = va < 0 ? -va : va; vb = vb < 0 ? -vb : vb; }
        if (MODE == 2) { va = 0; vb = 0; }
        a |= (unsigned)(va & 255) << (8 * e);
        b |= (unsigned)(vb & 255) << (8 * e);
      }
      A[f][w] = (int)a;
      B[f][w] = (int)b;
    }
  i32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[(u + i) & 7], B[(u * 3 + i) & 7], acc[i], 0, 0, 0);
  }
  int s = 0;
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

template <int NACC>
__global__ __launch_bounds__(256) void peak_f16(float* out, int iters, long long* clk) {
  unsigned x = threadIdx.x * 2654435761u + 12345u + blockIdx.x * 977u;
  half8 A[8], B[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      A[f][e] = (_Float16)((xs(x) >> 8) * (1.0f / 16777216.0f));
      B[f][e] = (_Float16)((xs(x) >> 8) * (1.0f / 16777216.0f));
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
static void run_i8(const char* name, int iters) {
  int* out; hipMalloc(&out, 256 * 256 * 4);
  long long* clk; hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((peak_i8<NACC, MODE>), dim3(256), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((peak_i8<NACC, MODE>), dim3(256), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double ops = 256.0 * 4 * (double)iters * NACC * 65536.0;   // 32 x 32 x 32 x 2
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-58s %8.3f ms  %7.1f TOP/s   shader clock %.2f GHz\n", name, ms, ops / ms * 1e-9, (double)h[0] / ((double)h[1] * 10.0));
}
template <int NACC>
static void run_f16(const char* name, int iters) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  long long* clk; hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((peak_f16<NACC>), dim3(256), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((peak_f16<NACC>), dim3(256), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double ops = 256.0 * 4 * (double)iters * NACC * 32768.0;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-58s %8.3f ms  %7.1f TFLOP/s shader clock %.2f GHz\n", name, ms, ops / ms * 1e-9, (double)h[0] / ((double)h[1] * 10.0));
}

int main() {
  run_f16<4>("fp16 32x32x16, U[0,1) operands, 1 wave/SIMD", 160000);
  run_i8<4, 0>("i8 32x32x32, bytes in [-127,127], 1 wave/SIMD, 4 acc", 160000);
  run_i8<16, 0>("i8 32x32x32, bytes in [-127,127], 1 wave/SIMD, 16 acc", 40000);
  run_i8<4, 1>("i8 32x32x32, bytes in [0,127], 1 wave/SIMD, 4 acc", 160000);
  run_i8<4, 2>("i8 32x32x32, zeros, 1 wave/SIMD, 4 acc", 160000);
  run_f16<4>("fp16 again", 160000);
  return 0;
}
