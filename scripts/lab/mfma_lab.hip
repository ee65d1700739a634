// Kernel lab for the batched-flat-scan filter (not part of the product): times experimental variants of the MFMA
// filter kernel beside the shipped ones on a synthetic fp16 mirror, and checks that every variant reports the same
// candidate set.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vectordb_amd/csrc scripts/lab/mfma_lab.hip -o scripts/lab/mfma_lab
// Run:   scripts/lab/mfma_lab [rows=2097152] [variants=all]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mfma_kernels.hpp"
#include "lab_kernels.hpp"

using namespace eps;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ float u01(uint64_t i) {  // splitmix64 -> U[0,1)
  uint64_t z = i + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ void fill_kernel(_Float16* x, float* base, float* base_s, int64_t n, int d, uint64_t seed) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) {
    const _Float16 h = (_Float16)u01(seed + (uint64_t)row * d + c);
    x[row * d + c] = h;
    s += (float)h * (float)h;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0 && base) {
    base[row] = s;
    base_s[row] = -0.5f * s;
  }
}
__global__ void checksum_kernel(const u32* cand, const u32* cnt, int64_t nq, int cap, unsigned long long* out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nq) return;
  const u32 c = cnt[j] < (u32)cap ? cnt[j] : (u32)cap;
  unsigned long long s = 0;
  for (u32 i = 0; i < c; ++i) s += (unsigned long long)cand[j * (int64_t)cap + i] * 2654435761ull + 1;
  atomicAdd(&out[0], s);
  atomicAdd(&out[1], (unsigned long long)cnt[j]);
}

struct Variant {
  const char* name;
  void (*launch)(const FilterArgs&, int num_cus, hipStream_t);
  bool exact;  // reports the true candidate set (ablated variants do not)
};

template <typename K>
static void set_shm(K k, int bytes) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); }

static void launch_v3(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(mfma_filter_kernel_v3, 2 * 65536 + 2048), true);
  (void)once;
  hipLaunchKernelGGL(mfma_filter_kernel_v3, dim3(cus), dim3(512), 2 * 65536 + 2048, s, a);
}
template <int KNOB>
static void launch_v3k(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(lab_v3<KNOB>, 2 * 65536 + 2048), true);
  (void)once;
  hipLaunchKernelGGL(lab_v3<KNOB>, dim3(cus), dim3(512), 2 * 65536 + 2048, s, a);
}

template <int KNOB>
static void launch_v5(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(lab_v5<KNOB>, 4 * 32768 + 2048), true);
  (void)once;
  hipLaunchKernelGGL(lab_v5<KNOB>, dim3(cus), dim3(512), 4 * 32768 + 2048, s, a);
}

static void launch_v5p(const FilterArgs& a, int cus, hipStream_t s) { launch_v5<0>(a, cus, s); }   // (the shipped v5 moved here: lab_v5<0>)
static void launch_v7p(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(mfma_filter_kernel_v7<2, FM_IDS>, (int)V7_LDS_BYTES), true);
  (void)once;
  hipLaunchKernelGGL((mfma_filter_kernel_v7<2, FM_IDS>), dim3(cus), dim3(256), V7_LDS_BYTES, s, a);
}
template <int KNOB>
static void launch_v6(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(lab_v6<KNOB>, 4 * 32768 + 2048), true);
  (void)once;
  hipLaunchKernelGGL(lab_v6<KNOB>, dim3(cus), dim3(512), 4 * 32768 + 2048, s, a);
}

template <int KNOB>
static void launch_v7(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(lab_v7<KNOB>, 4 * 32768 + 2048), true);
  (void)once;
  hipLaunchKernelGGL(lab_v7<KNOB>, dim3(cus), dim3(256), 4 * 32768 + 2048, s, a);
}

template <int KNOB>
static void launch_v8(const FilterArgs& a, int cus, hipStream_t s) {
  static bool once = (set_shm(lab_v8<KNOB>, 4 * 32768 + 2048), true);
  (void)once;
  hipLaunchKernelGGL(lab_v8<KNOB>, dim3(cus), dim3(256), 4 * 32768 + 2048, s, a);
}

#include "lab_variants.inc"

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2097152;
  const std::string which = argc > 2 ? argv[2] : "all";
  const int d = 768, nq = 1024, cap = 4096;
  const int64_t n_pad = (n + 255) / 256 * 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount / 8 * 8;
  printf("# %s, %d CUs, rows %lld x %d fp16, %d queries\n", prop.name, cus, (long long)n, d, nq);
  _Float16 *xh, *qh, *qf;
  float *base, *base_s, *T;
  u32 *cand, *cnt;
  unsigned long long* sums;
  CK(hipMalloc(&xh, (size_t)n_pad * d * 2));
  CK(hipMalloc(&qh, (size_t)nq * d * 2));
  CK(hipMalloc(&qf, (size_t)nq * d * 2));
  CK(hipMalloc(&base, (size_t)n_pad * 4));
  CK(hipMalloc(&base_s, (size_t)n_pad * 4));
  CK(hipMalloc(&T, (size_t)nq * 4));
  CK(hipMalloc(&cand, (size_t)nq * cap * 4));
  CK(hipMalloc(&cnt, (size_t)(nq + 1) * 4));
  CK(hipMalloc(&sums, 16));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n_pad + 3) / 4)), dim3(256), 0, 0, xh, base, base_s, n_pad, d, 1ull);
  hipLaunchKernelGGL(fill_kernel, dim3((nq + 3) / 4), dim3(256), 0, 0, qh, (float*)nullptr, (float*)nullptr, (int64_t)nq, d, 0x5555555555ull);
  hipLaunchKernelGGL(pack_qf_kernel, dim3((nq / 32) * (d / 16)), dim3(64), 0, 0, qh, qf, (int64_t)nq, d);
  // key = |x|^2 - 2 q.x ~ N(-128, ~11^2) for U[0,1) data: T = -170 passes ~1e-4 of the pairs
  std::vector<float> hT(nq, argc > 3 ? (float)atof(argv[3]) : -170.f);
  CK(hipMemcpy(T, hT.data(), nq * 4, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  FilterArgs a{};
  a.xh = xh; a.qh = qh; a.qf = qf; a.base = base; a.base_s = base_s; a.T = T; a.d_pad = d; a.tiles_q = nq / 256; a.tile0 = 0; a.ntiles = n_pad / 256;
  a.row_hi = n; a.nq = nq; a.s = -2.f; a.inv_s = -0.5f; a.sync_shift = 0; a.cand = cand; a.cand_keys = nullptr; a.qstat = nullptr; a.metric = 0; a.cnt = cnt;
  a.cap = cap; a.ablate = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long ref[2] = {0, 0};
  struct Checked { const Variant* v; unsigned long long hash, cnt; const char* verdict; };
  std::vector<Checked> checked;
  auto verdict_of = [&](const Variant& v, const unsigned long long* h) -> const char* {
    if (!v.exact) return "";
    if (!ref[1] && !ref[0]) { ref[0] = h[0]; ref[1] = h[1]; return "(reference set)"; }
    return (h[0] == ref[0] && h[1] == ref[1]) ? "OK" : "MISMATCH";
  };
  const double flop = 2.0 * (double)n_pad * nq * d;
  for (const Variant& v : kVariants) {
    if (which != "all" && which.find(v.name) == std::string::npos) continue;
    for (int w = 0; w < 2; ++w) v.launch(a, cus, 0);
    CK(hipMemsetAsync(cnt, 0, (nq + 1) * 4, 0));
    CK(hipMemsetAsync(sums, 0, 16, 0));
    v.launch(a, cus, 0);
    std::vector<u32> hc((size_t)nq * cap), hn(nq);
    CK(hipMemcpy(hc.data(), cand, hc.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hn.data(), cnt, nq * 4, hipMemcpyDeviceToHost));
    unsigned long long h[2] = {0, 0};
    for (int j = 0; j < nq; ++j) {
      if (hn[j] > (u32)cap) continue;   // overflowed: an arbitrary subset was kept
      const u32 c = hn[j];
      std::sort(hc.begin() + (size_t)j * cap, hc.begin() + (size_t)j * cap + c);
      for (u32 i = 0; i < c; ++i) h[0] = h[0] * 1099511628211ull + hc[(size_t)j * cap + i] + 1;
      h[0] = h[0] * 1099511628211ull + 0xABCDu;
      h[1] += hn[j];
    }
    static std::vector<u32> ref_c, ref_n;
    if (v.exact) {
      if (ref_c.empty()) { ref_c = hc; ref_n = hn; }
      else {
        int bad = 0;
        for (int j = 0; j < nq; ++j) {
          const u32 c = std::min<u32>(hn[j], cap), c0 = std::min<u32>(ref_n[j], cap);
          if (hn[j] > (u32)cap && ref_n[j] > (u32)cap) continue;
          bool same = c == c0;
          for (u32 i = 0; same && i < c; ++i) same = hc[(size_t)j * cap + i] == ref_c[(size_t)j * cap + i];
          if (!same && bad++ < 3) {
            printf("  query %d: ref %u cands, this %u:", j, c0, c);
            for (u32 i = 0; i < std::max(c, c0) && i < 12; ++i)
              printf(" [%u|%u]", i < c0 ? ref_c[(size_t)j * cap + i] : 0u, i < c ? hc[(size_t)j * cap + i] : 0u);
            printf("\n");
          }
        }
        printf("  %d queries differ\n", bad);
      }
    }
    checked.push_back({&v, h[0], h[1], verdict_of(v, h)});
  }
  // timing: several rounds over all variants (clock / thermal state drifts within a process), median per variant
  const int rounds = 5, reps = 3;
  std::vector<std::vector<float>> times(checked.size());
  for (int r = 0; r < rounds; ++r) {
    for (size_t i = 0; i < checked.size(); ++i) {
      const Variant& v = *checked[i].v;
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < reps; ++q) v.launch(a, cus, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      times[i].push_back(ms / reps);
    }
  }
  for (size_t i = 0; i < checked.size(); ++i) {
    std::sort(times[i].begin(), times[i].end());
    const float med = times[i][rounds / 2];
    printf("%-28s median %7.3f ms (min %7.3f max %7.3f)  %7.1f TFLOP/s  cand %llu %s\n", checked[i].v->name, med, times[i][0],
           times[i][rounds - 1], flop / med * 1e-9, checked[i].cnt, checked[i].verdict);
  }
  return 0;
}
