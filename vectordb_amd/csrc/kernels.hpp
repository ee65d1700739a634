// Host-callable launchers for the HIP kernels of libepsilla_gfx950 (definitions in *.hip).
// All pointers are device pointers unless stated otherwise; all launches are asynchronous on `s`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace eps {

// ---------------------------------------------------------------- flat scan (BruteForceSearch, :717-768)
struct FlatScanArgs {
  const float* rows;
  int64_t row_begin, row_end;   // local row range scanned
  int dim;
  int metric;
  const float* queries;         // [nq][dim]
  int64_t nq;
  int k;
  FilterSpec f;
  u64* partial;                 // [nq][W][k] per-wave sorted lists
  int W;                        // wavefronts per query = gridDim.x * 4
  const u64* thr_in;            // optional [nq] initial thresholds (only keys < thr can enter), or null
  const u64* lo_in;             // optional: only keys > lo_in[q * lo_stride] can enter (paging through results beyond 1024 per query), or null
  int64_t lo_stride;
};
// returns W (partial lists per query). `partial` must hold nq * flat_scan_waves(...) * k keys.
int flat_scan_waves(int64_t nrows, int64_t nq, int dim);
void launch_flat_scan(const FlatScanArgs& a, hipStream_t s);

// merge the `slots` keys of each query (plus, if merge_run, the existing run_keys[q][k]) into run_keys[q][k].
// counts (optional): per-query number of valid keys at the front of its slots (candidate lists).
// seed_cand != null (the MFMA engine's seed stage, r4): instead of the k best keys, their ROWS go to seed_cand[q][0 .. count) (sample
// indices mapped by seed_row), seed_cnt[q] = count and run_keys[q] is left EMPTY - what seed_to_cand_kernel did in a launch of its own
void launch_merge_lists(const u64* keys, int slots, int k, int64_t nq, u64* run_keys, bool merge_run, hipStream_t s,
                        const u32* counts = nullptr, const FilterSpec* visible = nullptr, u64 id_stride = 0, u32 id_head = 0,
                        u32* seed_cand = nullptr, int seed_cap = 0, u32* seed_cnt = nullptr);

// exact fp32 re-rank of gathered candidate rows into run_keys (sorted, unique)
struct RerankArgs {
  const float* rows;
  int dim;
  int metric;
  const float* queries;     // [nq][dim]
  int64_t nq;
  int k;
  FilterSpec f;
  const u32* cand;          // [nq][cap] local row ids
  u32* cand_count;          // [nq] (may exceed cap: only min(count,cap) are present); zeroed per query when fuse is set
  int cap;
  u64* run_keys;            // [nq][k] in/out
  // Stage bookkeeping of the MFMA engine folded into the re-rank (r3; it was two tiny launches per stage): after query q's re-rank
  // the block (i) adds the stage's candidate count to the overflow / total counters, (ii) computes the NEXT stage's pass threshold
  // from the query's new k-th best exact key, (iii) zeroes the query's candidate count and (block 0) the group arrival counters
  // for the next filter launch.  fuse = 0: none of it.
  int fuse;
  u32* overflow;            // [1]
  unsigned long long* total;   // [1]
  void* T_next;             // float [b_pad] (fp16 operands) / int [b_pad] (int8 operands), or null after the last stage
  const float* qstat;       // [b_pad][4]
  const float* scal;        // the mirror's maxima
  int bits;                 // 8 | 16
  float u, slack;
  u32* gsync;               // [256] or null
  // the call's LAST re-rank also converts the result keys (ids = local * stride + base, distances, counts): finalize_kernel's work
  // without its launch (r4: a single-query call is a chain of short dependent launches); fin_ids == null: not this launch
  int64_t* fin_ids = nullptr;
  float* fin_dist = nullptr;
  int32_t* fin_counts = nullptr;
  int64_t fin_base = 0, fin_stride = 1;
  // a handful of queries (r4): a query's candidates are spread over `parts` workgroups (one workgroup gathering a stage's ~150 rows of
  // 3 KB is ~20 us of a single-query call); each writes its k best to part_keys[q][part][k], the last to arrive (part_done[q], agent-scope
  // release / acquire) merges them with the running list and does the bookkeeping above.  parts <= 1: one workgroup per query.
  int parts = 0;
  u64* part_keys = nullptr;
  u32* part_done = nullptr;   // [nq], zero between launches (the last workgroup zeroes it again)
  // one-pass search of a handful of queries (stream8_kernel.hpp): the launch first SELECTS its candidates - the entries of the pass's
  // per-wavefront lists that still pass against the final table of best accumulators - into `cand` (written through s8_cand), starts
  // from an empty running list, and counts a lost list entry as an overflow.  s8_G == null: an ordinary re-rank
  const int* s8_G = nullptr;        // [nq][s8_slots]
  int s8_slots = 64;                // 64 (k <= 16) | 128 (k = 17..64): Stream8Args::slots
  const u32* s8_counts = nullptr;   // [nq][s8_waves]
  const u64* s8_lists = nullptr;    // [nq][s8_waves][S8_WAVE_CAP]
  int s8_waves = 0;
  u32* s8_cand = nullptr;           // = cand
  // one-pass form: the block that finishes last (ticket) copies overflow / total to host-mapped words [0] = overflow, [2..3] = total, so the
  // caller needs no device-to-host copy after the launch (a copy is a trip through the DMA queue at the end of a 0.2 ms call)
  u32* pub = nullptr;               // host-mapped [4], or null
  u32* pub_ticket = nullptr;        // device word, zero at launch
  int s8_fast = 0;                  // 1: s8_rerank_kernel (r5, late: the one-pass call's own re-rank); 0: rerank_kernel's s8 prologue (A/B)
  int s8_reset = 0;                 // 1: the launch leaves the one-pass state as the next call needs it (table slots empty, counters zero): no prep launch then
};
void launch_rerank(const RerankArgs& a, hipStream_t s);

// run_keys -> caller-visible results: ids = local*stride+base (int64), dist fp32, counts int32
void launch_finalize(const u64* run_keys, int64_t nq, int k, int64_t id_base, int64_t id_stride, int64_t* ids,
                     float* dist, int32_t* counts, hipStream_t s);

void launch_normalize(float* rows, int64_t n, int dim, bool only_if_nonzero, hipStream_t s);
void launch_merge_shards(const float* dist, const int64_t* ids, int shards, int64_t nq, int k, float* out_dist,
                         int64_t* out_ids, hipStream_t s, int64_t stride_bytes = 0, int32_t* out_counts = nullptr);
void launch_fill_u64(u64* p, int64_t n, u64 v, hipStream_t s);

}  // namespace eps
