// hipcc (ROCm 7.2) miscompiles __builtin_bit_cast(unsigned, v[r]) on an ext_vector ELEMENT: every index reads element 0 (the four
// ds_write_b128 below store the same quad); (unsigned)v[r] is correct.  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only
#include <hip/hip_runtime.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const i32x4* a, const i32x4* b, int* out, int T) {
  __shared__ unsigned ent[64 * 20];
  i32x16 acc[2];
  for (int r = 0; r < 16; ++r) { acc[0][r] = a[threadIdx.x][r & 3] + r; acc[1][r] = r; }
  for (int it = 0; it < 8; ++it) {
    acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x + it], b[threadIdx.x + it], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x + it], b[threadIdx.x], acc[1], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int mx = acc[j][0];
    for (int r = 1; r < 16; ++r) mx = max(mx, acc[j][r]);
    if (__any(mx >= T)) {
      if (mx >= T) {
        unsigned* e = ent + threadIdx.x * 20;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[4 + r] = __builtin_bit_cast(unsigned, acc[j][r]);
      }
    }
  }
  __syncthreads();
  out[threadIdx.x] = ent[(threadIdx.x * 7) % (64 * 20)];
}
