/* ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's ANN hot path (epsilla-cloud/vectordb 0.3.18), written
 * from the algorithm spec in SURVEY.md Appendix A and the cited reference lines.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
 * (vectordb_amd/csrc, libepsilla_gfx950.so) never links, imports or falls back to anything here.
 *
 * PARITY STATUS: PINNED.  tests/test_oracle_vs_ref.py checks every function below against the
 * reference's own sources compiled verbatim (oracle/_ref/libepsilla_ref.so, recipe in
 * oracle/Makefile) on seeded inputs, and tests/golden/ holds fixtures generated from that build
 * (scripts/gen_golden.py) plus the reference's own known-answer gtest cases
 * (engine/test/engine/db/db_server.cpp:92-319, 514-751, 1085-1245, 1407-1630).
 *
 * Deviations, all deliberate and documented in DESIGN.md:
 *   - kNN graph: exact brute force here; the reference uses randomised NN-Descent
 *     (engine/db/index/knn/nndescent.hpp:96-192), which only approximates this result.
 *   - std::sort on NSG `Neighbor` (distance-only order, neighbor.hpp:25-28) is unstable on ties in
 *     the reference; here a stable merge sort is used, so results agree on tie-free inputs.
 *   - multi-thread SearchImpl (T > 1) is racy in the reference; here the T workers of a round run
 *     either one after another in worker order, or in lockstep (eo_set_schedule(1): step i of every
 *     worker before step i+1 of any, workers in order within a step — the schedule the device kernel
 *     implements); both are interleavings the reference allows.  T == 1 is bit-deterministic on both sides.
 *   - filters: only `<int column> <op> <int const>` (what the C-ABI lowers to the device); the
 *     reference's general expression engine (engine/query/expr) stays on the host DBMS.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * a1/a2  distance kernels
 * fvec_L2sqr / fvec_inner_product: engine/db/index/distance_simd.cpp:205-214, 180-187.
 * The source loops are scalar, but they are compiled under
 * `#pragma GCC optimize("unroll-loops,associative-math,no-signed-zeros")` (platform_macros.hpp:135-140)
 * and the release build (-O3, no -march: engine/build.sh:24) vectorises them to 4 SSE lanes with
 * separate mulps/addps (no FMA).  Verified by disassembling distance_simd.o built with those flags:
 *   acc[l] += t*t for l = i mod 4 over the first d - d%4 elements; res = (acc0+acc2) + (acc1+acc3)
 *   (movhlps/addps, shufps/addps); then the d%4 tail (L2: pair-sum then single; IP: one by one).  d <= 3: scalar.
 * This file is compiled without -ffast-math / contraction so the order below is what runs.
 */
float eo_fvec_l2sqr(const float* x, const float* y, int64_t d) {
  if (d <= 3) {
    float res = 0.f;
    for (int64_t i = 0; i < d; ++i) {
      const float t = x[i] - y[i];
      res += t * t;
    }
    return res;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int64_t d4 = d & ~(int64_t)3;
  for (int64_t i = 0; i < d4; i += 4) {
    const float t0 = x[i] - y[i], t1 = x[i + 1] - y[i + 1], t2 = x[i + 2] - y[i + 2], t3 = x[i + 3] - y[i + 3];
    const float m0 = t0 * t0, m1 = t1 * t1, m2 = t2 * t2, m3 = t3 * t3;
    a0 = a0 + m0;
    a1 = a1 + m1;
    a2 = a2 + m2;
    a3 = a3 + m3;
  }
  const float h0 = a0 + a2, h1 = a1 + a3;
  float res = h0 + h1;
  /* tail (d%4 = 1..3): GCC vectorises two tail elements as a 2-lane op and adds their SUM to res
   * (mulps; shufps; addss; addss), then a last odd element on its own. */
  int64_t i = d4;
  if (d - i >= 2) {
    const float t0 = x[i] - y[i], t1 = x[i + 1] - y[i + 1];
    const float m0 = t0 * t0, m1 = t1 * t1;
    const float s = m1 + m0;
    res = res + s;
    i += 2;
  }
  if (i < d) {
    const float t = x[i] - y[i];
    res += t * t;
  }
  return res;
}

float eo_fvec_ip(const float* x, const float* y, int64_t d) {
  if (d <= 3) {
    float res = 0.f;
    for (int64_t i = 0; i < d; ++i) res += x[i] * y[i];
    return res;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int64_t d4 = d & ~(int64_t)3;
  for (int64_t i = 0; i < d4; i += 4) {
    const float m0 = x[i] * y[i], m1 = x[i + 1] * y[i + 1], m2 = x[i + 2] * y[i + 2], m3 = x[i + 3] * y[i + 3];
    a0 = a0 + m0;
    a1 = a1 + m1;
    a2 = a2 + m2;
    a3 = a3 + m3;
  }
  const float h0 = a0 + a2, h1 = a1 + a3;
  float res = h0 + h1;
  for (int64_t i = d4; i < d; ++i) res += x[i] * y[i];
  return res;
}

/* a3  GetDistFunc (engine/db/index/index.cpp:10-35): metric 0 = EUCLIDEAN -> L2Sqr (space_l2.hpp:24),
 * 1 = COSINE -> 1 - 1.0f*dot (space_cosine.hpp:13-16), 2 = DOT_PRODUCT -> -dot (space_ip.hpp:18).
 * Argument order is (row, query). */
float eo_dist(int metric, const float* row, const float* q, int64_t d) {
  switch (metric) {
    case 1: return 1 - 1.0f * eo_fvec_ip(row, q, d);
    case 2: return -eo_fvec_ip(row, q, d);
    default: return eo_fvec_l2sqr(row, q, d);
  }
}

void eo_dist_batch(int metric, const float* rows, int64_t n, const float* q, int64_t d, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = eo_dist(metric, rows + i * d, q, d);
}

/* a18  query-side Normalize (engine/db/vector.cpp:60-69): unconditional; zero vector -> NaN. */
void eo_normalize_query(float* v, int64_t d) {
  float sum = 0;
  for (int64_t i = 0; i < d; i++) sum += v[i] * v[i];
  sum = sqrtf(sum);
  for (int64_t i = 0; i < d; i++) v[i] /= sum;
}
/* a18  insert-side normalisation (engine/db/table_segment_mvp.cpp:574-587): only when sum > 1e-10. */
int eo_normalize_insert(float* v, int64_t d) {
  float sum = 0;
  for (int64_t i = 0; i < d; i++) sum += v[i] * v[i];
  if (sum > 1e-10) {
    sum = sqrtf(sum);
    for (int64_t i = 0; i < d; i++) v[i] /= sum;
    return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a4  Candidate (engine/db/execution/candidate.hpp:7-23): strict weak order (distance, id).
 */
typedef struct {
  int64_t id;
  float dist;
  uint8_t checked;
} eo_cand;

static int cand_less(const eo_cand* a, const eo_cand* b) {
  if (a->dist != b->dist) return a->dist < b->dist;
  return a->id < b->id;
}
static int cand_cmp_qsort(const void* a, const void* b) {
  const eo_cand *x = (const eo_cand*)a, *y = (const eo_cand*)b;
  if (cand_less(x, y)) return -1;
  if (cand_less(y, x)) return 1;
  return 0;
}
/* std::lower_bound over q[0..n) */
static int64_t cand_lower_bound(const eo_cand* q, int64_t n, const eo_cand* c) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (cand_less(&q[mid], c)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* row filter restricted to `int column <op> const` + the deleted bitset
 * (ConcurrentBitset: bit id&7 of byte id>>3, engine/utils/concurrent_bitset.cpp:9-18). */
typedef struct {
  const uint8_t* deleted; /* may be NULL */
  const uint8_t* attr;    /* base of the packed attribute rows, or NULL */
  int64_t stride;         /* bytes between rows (primitive_offset_) */
  int32_t width;          /* 1,2,4,8 byte signed int */
  int32_t op;             /* 0 none, 1 <, 2 <=, 3 ==, 4 >=, 5 >, 6 != */
  int64_t value;
} eo_filter;

static int row_deleted(const eo_filter* f, int64_t id) {
  return f && f->deleted && ((f->deleted[id >> 3] >> (id & 7)) & 1);
}
static int row_passes(const eo_filter* f, int64_t id) {
  if (!f || f->op == 0 || !f->attr) return 1;
  const uint8_t* p = f->attr + id * f->stride;
  int64_t v;
  switch (f->width) {
    case 1: v = *(const int8_t*)p; break;
    case 2: { int16_t t; memcpy(&t, p, 2); v = t; break; }
    case 8: { int64_t t; memcpy(&t, p, 8); v = t; break; }
    default: { int32_t t; memcpy(&t, p, 4); v = t; break; }
  }
  /* the reference evaluates numbers as double (expr_evaluator.cpp:127-164); exact for |v| < 2^53 */
  const double a = (double)v, b = (double)f->value;
  switch (f->op) {
    case 1: return a < b;
    case 2: return a <= b;
    case 3: return a == b;
    case 4: return a >= b;
    case 5: return a > b;
    case 6: return a != b;
    default: return 1;
  }
}

/* ------------------------------------------------------------------------------------------------
 * a12  BruteForceSearch / PreFilterBruteForceSearch (engine/db/execution/vec_search_executor.cpp:717-831)
 * distances for [start,end); drop deleted / filtered rows; full sort by (dist,id).  The two variants
 * differ only in whether the filter runs before or after the distance (no @distance support here),
 * so they return the same list.  `out` must hold end-start entries; returns the survivor count.
 */
int64_t eo_bruteforce(int metric, const float* rows, int64_t d, int64_t start, int64_t end, const float* q,
                      const eo_filter* f, eo_cand* out) {
  int64_t m = 0;
  for (int64_t id = start; id < end; ++id) {
    if (row_deleted(f, id) || !row_passes(f, id)) continue;
    out[m].id = id;
    out[m].dist = eo_dist(metric, rows + id * d, q, d);
    out[m].checked = 0;
    ++m;
  }
  qsort(out, (size_t)m, sizeof(eo_cand), cand_cmp_qsort);
  return m;
}

/* ------------------------------------------------------------------------------------------------
 * a8  AddIntoQueue (vec_search_executor.cpp:75-117): bounded sorted insert with id de-dup at the
 * lower_bound slot; returns the insert position, or `cap` when rejected.
 */
static int64_t add_into_queue(eo_cand* q, int64_t* size, int64_t cap, const eo_cand* c) {
  if (*size == 0) {
    q[(*size)++] = *c;
    return 0;
  }
  int64_t end = *size;
  int64_t loc = cand_lower_bound(q, end, c);
  if (loc != end) {
    if (c->id == q[loc].id) return cap; /* duplicate */
    if (*size >= cap) {
      --(*size);
      --end;
    }
  } else {
    if (*size < cap) {
      q[loc] = *c;
      ++(*size);
      return *size - 1;
    }
    return cap;
  }
  memmove(q + loc + 1, q + loc, (size_t)(end - loc) * sizeof(eo_cand));
  q[loc] = *c;
  ++(*size);
  return loc;
}

/* InsertOneElementAt (:135-148): shift [idx, n-1) right by one (last falls off), q[idx] = c. */
static void insert_one_at(const eo_cand* c, eo_cand* q, int64_t idx, int64_t n) {
  memmove(q + idx + 1, q + idx, (size_t)(n - idx - 1) * sizeof(eo_cand));
  q[idx] = *c;
}

/* a10  MergeTwoQueuesInto1stQueueSeqFixed (:150-217).  q1 has fixed size n1; q2 is sorted, size n2. */
int64_t eo_merge_fixed(eo_cand* q1, int64_t n1, const eo_cand* q2, int64_t n2) {
  int64_t idx = cand_lower_bound(q1, n1, &q2[0]);
  if (idx == n1) return idx;
  if (idx == n1 - 1) {
    q1[idx] = q2[0];
    return idx;
  }
  if (q2[0].id != q1[idx].id) {
    insert_one_at(&q2[0], q1, idx, n1);
  } else if (!q2[0].checked && q1[idx].checked) {
    q1[idx].checked = 0;
  }
  if (n2 == 1) return idx;
  int64_t i1 = idx + 1, i2 = 1;
  for (int64_t ins = idx + 1; ins < n1; ++ins) {
    if (i1 >= n1 || i2 >= n2) break;
    if (cand_less(&q1[i1], &q2[i2])) {
      ++i1;
    } else if (cand_less(&q2[i2], &q1[i1])) {
      insert_one_at(&q2[i2++], q1, ins, n1);
      ++i1;
    } else {
      if (!q2[i2].checked && q1[i1].checked) q1[i1].checked = 0;
      ++i2;
      ++i1;
    }
  }
  return idx;
}

/* a5  PrepareInitIds (:487-516): distinct CSR neighbours of nav (in order, up to L), then
 * nav+1, nav+2, ... wrapping, skipping already selected ids.  Requires n >= L. */
void eo_prepare_init_ids(int64_t n, const int64_t* off, const int64_t* nbr, int64_t nav, int64_t L, int64_t* init) {
  uint8_t* sel = (uint8_t*)calloc((size_t)n, 1);
  int64_t cnt = 0;
  for (int64_t e = off[nav]; e < off[nav + 1] && cnt < L; ++e) {
    int64_t v = nbr[e];
    if (sel[v]) continue;
    sel[v] = 1;
    init[cnt++] = v;
  }
  int64_t tmp = nav + 1;
  while (cnt < L) {
    if (tmp == n) tmp = 0;
    int64_t v = tmp++;
    if (sel[v]) continue;
    sel[v] = 1;
    init[cnt++] = v;
  }
  free(sel);
}

typedef struct {
  int metric;
  const float* rows;
  int64_t d;
  const int64_t* off;
  const int64_t* nbr;
  const float* q;
  uint8_t* visited;
  int64_t evals;
} eo_ctx;

/* a7  ExpandOneCandidate (:384-444).  `bound` points at the live worst-of-master slot (:546). */
static int64_t expand_one(eo_ctx* c, int64_t cand_id, const float* bound, eo_cand* queue, int64_t* qsize, int64_t cap) {
  int64_t nk = cap;
  for (int64_t e = c->off[cand_id]; e < c->off[cand_id + 1]; ++e) {
    int64_t nb = c->nbr[e];
    if (c->visited[nb]) continue;
    c->visited[nb] = 1;
    ++c->evals;
    float dist = eo_dist(c->metric, c->rows + nb * c->d, c->q, c->d);
    if (dist > *bound) continue;
    eo_cand cand = {nb, dist, 0};
    int64_t r = add_into_queue(queue, qsize, cap, &cand);
    if (r < nk) nk = r;
  }
  return nk;
}

/* a11  SearchImpl (:518-715) with a6 InitializeSetLPara (:446-485), a9 PickTopMToWorkers (:328-356),
 * a10 MergeAllQueuesToMaster (:297-326).
 * set_L holds (T-1)*Lq + L candidates; worker w < T-1 owns [w*Lq, (w+1)*Lq), the master is queue T-1
 * (start (T-1)*Lq, size L).  `visited` is an n-byte scratch that must be zero on entry; it is zeroed
 * again on exit (the reference's per-query clear()+resize(n), :711-714).
 * On return the master queue set_L[(T-1)*Lq .. +L) is the sorted result.  Returns #distance evals. */
static int g_lockstep = 0;
/* selects the interleaving eo_search_impl uses for T > 1 (see the parallel region below); returns the previous value */
int eo_set_schedule(int lockstep) {
  const int old = g_lockstep;
  g_lockstep = lockstep;
  return old;
}

int64_t eo_search_impl(int metric, const float* rows, int64_t d, int64_t n, const int64_t* off, const int64_t* nbr,
                       const int64_t* init_ids, const float* q, int T, int64_t L, int64_t Lq, int64_t I, eo_cand* set_L,
                       uint8_t* visited) {
  eo_ctx c = {metric, rows, d, off, nbr, q, visited, 0};
  int64_t* starts = (int64_t*)malloc(sizeof(int64_t) * (size_t)T);
  int64_t* sizes = (int64_t*)calloc((size_t)T, sizeof(int64_t));
  for (int w = 0; w < T; ++w) starts[w] = w * Lq;
  const int64_t ms = starts[T - 1];
  eo_cand* master = set_L + ms;

  /* InitializeSetLPara */
  for (int64_t i = 0; i < L; ++i) visited[init_ids[i]] = 1;
  for (int64_t i = 0; i < L; ++i) {
    int64_t v = init_ids[i];
    ++c.evals;
    master[i].id = v;
    master[i].dist = eo_dist(metric, rows + v * d, q, d);
    master[i].checked = 0;
  }
  qsort(master, (size_t)L, sizeof(eo_cand), cand_cmp_qsort);
  sizes[T - 1] = L;
  const float* last_dist = &master[L - 1].dist; /* live reference, :546 */

  int64_t k_master = 0;
  int no_need = 0;
  { /* one sequential expansion (:556-593) */
    if (k_master == L) {
      no_need = 1;
    } else {
      int64_t r;
      eo_cand* cand = &master[k_master];
      if (!cand->checked) {
        cand->checked = 1;
        r = expand_one(&c, cand->id, last_dist, master, &sizes[T - 1], L);
      } else {
        r = L;
      }
      if (r <= k_master) k_master = r; else ++k_master;
    }
  }
  while (!no_need) {
    /* PickTopMToWorkers */
    int64_t count = 0;
    {
      int dest = 0;
      const int64_t bound_i = sizes[T - 1];
      for (int64_t ci = k_master; ci < bound_i; ++ci) {
        if (master[ci].checked) continue;
        ++count;
        if (T - 1 != dest) {
          set_L[starts[dest] + sizes[dest]++] = master[ci];
          master[ci].checked = 1;
          if (sizes[dest] == Lq) break;
          ++dest;
        } else {
          dest = 0;
        }
      }
    }
    if (!count) break;
    /* the parallel region (:616-679).  The reference's T OpenMP workers race on is_visited and on the live bound;
     * two deterministic interleavings of it are restated here (both are executions the reference allows):
     *   schedule 0: the workers run one after another in worker order, each doing all of its <= I expansions;
     *   schedule 1 ("lockstep"): expansion i of every worker happens before expansion i+1 of any worker, and
     *     within one step the workers go in worker order (master last).  This is the schedule the device kernel
     *     implements (all T expansions of a step are in flight at once; conflicts on the visited set are
     *     resolved in worker order), so GPU-vs-oracle parity at T > 1 is bit-exact under it. */
    if (!g_lockstep) {
      for (int w = 0; w < T; ++w) {
        eo_cand* queue = set_L + starts[w];
        int64_t* qsize = &sizes[w];
        int64_t k_uc = (T - 1 != w) ? 0 : k_master;
        int64_t it = 0;
        while (it < I && k_uc < *qsize) {
          eo_cand* cand = &queue[k_uc];
          if (!cand->checked) {
            cand->checked = 1;
            ++it;
            int64_t r = expand_one(&c, cand->id, last_dist, queue, qsize, Lq);
            if (r <= k_uc) k_uc = r; else ++k_uc;
          } else {
            ++k_uc;
          }
          if (T - 1 == w) k_master = k_uc;
        }
      }
    } else {
      int64_t* k_uc = (int64_t*)malloc(sizeof(int64_t) * (size_t)T);
      int64_t* its = (int64_t*)calloc((size_t)T, sizeof(int64_t));
      for (int w = 0; w < T; ++w) k_uc[w] = (T - 1 != w) ? 0 : k_master;
      for (;;) {
        int any = 0;
        for (int w = 0; w < T; ++w) {
          eo_cand* queue = set_L + starts[w];
          int64_t* qsize = &sizes[w];
          /* the `else ++k_uc` branch (:671) costs no iteration: skip checked entries */
          while (its[w] < I && k_uc[w] < *qsize && queue[k_uc[w]].checked) ++k_uc[w];
          if (T - 1 == w) k_master = k_uc[w];
          if (!(its[w] < I && k_uc[w] < *qsize)) continue;
          eo_cand* cand = &queue[k_uc[w]];
          cand->checked = 1;
          ++its[w];
          int64_t r = expand_one(&c, cand->id, last_dist, queue, qsize, Lq);
          if (r <= k_uc[w]) k_uc[w] = r; else ++k_uc[w];
          if (T - 1 == w) k_master = k_uc[w];
          any = 1;
        }
        if (!any) break;
      }
      free(k_uc);
      free(its);
    }
    /* MergeAllQueuesToMaster */
    {
      int64_t nk = L;
      for (int w = 0; w < T - 1; ++w) {
        if (sizes[w] == 0) continue;
        int64_t r = eo_merge_fixed(master, L, set_L + starts[w], sizes[w]);
        if (r < nk) nk = r;
        sizes[w] = 0;
      }
      if (nk <= k_master) k_master = nk;
    }
  }
  memset(visited, 0, (size_t)n);
  free(starts);
  free(sizes);
  return c.evals;
}

/* ------------------------------------------------------------------------------------------------
 * a13  Search (:833-935): mode selection, brute-force tail merge, post-filter walk.
 *   n_indexed = graph record_number_, n_total = table record_number_, limit = requested k.
 *   prefilter -> flat over [0,n_total); n_indexed < 512 -> flat, count additionally <= Lq;
 *   else graph search + (if n_total > n_indexed) flat over the tail merged into the first K slots.
 * ids_out/dist_out need max(L, limit) entries.  Returns result_size.
 */
int64_t eo_search(int metric, const float* rows, int64_t d, int64_t n_indexed, int64_t n_total, const int64_t* off,
                  const int64_t* nbr, int64_t nav, const float* q, int64_t limit, int T, int64_t L, int64_t Lq, int64_t I,
                  int prefilter, const eo_filter* f, int64_t* ids_out, double* dist_out, int64_t* evals_out) {
  int64_t result = 0;
  if (evals_out) *evals_out = 0;
  if (prefilter || n_indexed < 512) {
    eo_cand* bf = (eo_cand*)malloc(sizeof(eo_cand) * (size_t)(n_total > 0 ? n_total : 1));
    int64_t m = eo_bruteforce(metric, rows, d, 0, n_total, q, f, bf);
    result = m < limit ? m : limit;
    if (!prefilter && result > Lq) result = Lq;
    for (int64_t i = 0; i < result; ++i) {
      ids_out[i] = bf[i].id;
      dist_out[i] = bf[i].dist;
    }
    if (evals_out) *evals_out = n_total;
    free(bf);
    return result;
  }
  int64_t K = n_indexed;
  if (limit < K) K = limit;
  if (Lq < K) K = Lq;
  int64_t* init = (int64_t*)malloc(sizeof(int64_t) * (size_t)L);
  eo_prepare_init_ids(n_indexed, off, nbr, nav, L, init);
  eo_cand* set_L = (eo_cand*)calloc((size_t)((T - 1) * Lq + L), sizeof(eo_cand));
  uint8_t* visited = (uint8_t*)calloc((size_t)n_indexed, 1);
  int64_t ev = eo_search_impl(metric, rows, d, n_indexed, off, nbr, init, q, T, L, Lq, I, set_L, visited);
  eo_cand* master = set_L + (int64_t)(T - 1) * Lq;
  int64_t cand_num;
  if (n_total > n_indexed) {
    eo_cand* bf = (eo_cand*)malloc(sizeof(eo_cand) * (size_t)(n_total - n_indexed));
    int64_t m = eo_bruteforce(metric, rows, d, n_indexed, n_total, q, f, bf);
    ev += n_total - n_indexed;
    int64_t bfn = m < limit ? m : limit;
    if (bfn > 0) {
      eo_merge_fixed(master, K, bf, bfn); /* only the first K slots take part (:894-900) */
      cand_num = L < n_total ? L : n_total;
    } else {
      cand_num = L < n_indexed ? L : n_indexed;
    }
    free(bf);
  } else {
    cand_num = L < n_indexed ? L : n_indexed;
  }
  for (int64_t i = 0; i < cand_num && result < K; ++i) {
    int64_t id = master[i].id;
    if (row_deleted(f, id) || !row_passes(f, id)) continue;
    ids_out[result] = id;
    dist_out[result] = master[i].dist;
    ++result;
  }
  if (evals_out) *evals_out = ev;
  free(init);
  free(set_L);
  free(visited);
  return result;
}

/* copy helpers for ctypes callers */
void eo_cands_unpack(const eo_cand* c, int64_t n, int64_t* ids, float* dists, uint8_t* checked) {
  for (int64_t i = 0; i < n; ++i) {
    ids[i] = c[i].id;
    dists[i] = c[i].dist;
    if (checked) checked[i] = c[i].checked;
  }
}
void eo_cands_pack(eo_cand* c, int64_t n, const int64_t* ids, const float* dists, const uint8_t* checked) {
  for (int64_t i = 0; i < n; ++i) {
    c[i].id = ids[i];
    c[i].dist = dists[i];
    c[i].checked = checked ? checked[i] : 0;
  }
}
int64_t eo_sizeof_cand(void) { return (int64_t)sizeof(eo_cand); }

/* ------------------------------------------------------------------------------------------------
 * a16 (replacement)  exact K-nearest-neighbour graph: for each row the K closest other rows under the
 * field metric, ascending (dist,id).  Stands in for NN-Descent (knn.hpp:90-135), whose output is an
 * approximation of exactly this list.  O(n^2 d): small n only.
 */
void eo_knn_exact(int metric, const float* rows, int64_t n, int64_t d, int64_t K, int64_t* out) {
  eo_cand* tmp = (eo_cand*)malloc(sizeof(eo_cand) * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    int64_t m = 0;
    for (int64_t j = 0; j < n; ++j) {
      if (j == i) continue;
      tmp[m].id = j;
      /* OracleL2::operator()(p,q) = dist(row p, row q) (knn.hpp:76-85) */
      tmp[m].dist = eo_dist(metric, rows + i * d, rows + j * d, d);
      tmp[m].checked = 0;
      ++m;
    }
    qsort(tmp, (size_t)m, sizeof(eo_cand), cand_cmp_qsort);
    for (int64_t k = 0; k < K; ++k) out[i * K + k] = k < m ? tmp[k].id : -1;
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * a17  NSG build from a kNN graph (engine/db/index/nsg/nsg.cpp:45-775; always L2, ann_graph_segment.cpp:216)
 */
typedef struct {
  int64_t id;
  float dist;
  uint8_t flag;
} eo_nb; /* Neighbor (nsg/neighbor.hpp:12-29), ordered by distance only */

typedef struct {
  int64_t* v;
  int64_t n, cap;
} ivec;
static void ivec_push(ivec* a, int64_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 8;
    a->v = (int64_t*)realloc(a->v, sizeof(int64_t) * (size_t)a->cap);
  }
  a->v[a->n++] = x;
}
typedef struct {
  eo_nb* v;
  int64_t n, cap;
} nbvec;
static void nbvec_push(nbvec* a, eo_nb x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 64;
    a->v = (eo_nb*)realloc(a->v, sizeof(eo_nb) * (size_t)a->cap);
  }
  a->v[a->n++] = x;
}
/* stable merge sort by distance (see header note on std::sort ties) */
static void nb_sort(eo_nb* a, int64_t n) {
  if (n < 2) return;
  eo_nb* tmp = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)n);
  for (int64_t w = 1; w < n; w *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * w) {
      int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int64_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (a[j].dist < a[i].dist) ? a[j++] : a[i++];
      while (i < mid) tmp[k++] = a[i++];
      while (j < hi) tmp[k++] = a[j++];
    }
    memcpy(a, tmp, sizeof(eo_nb) * (size_t)n);
  }
  free(tmp);
}

/* InsertIntoPool (nsg_helper.cpp:10-51) */
static int64_t insert_into_pool(eo_nb* addr, int64_t K, eo_nb nn) {
  int64_t left = 0, right = K - 1;
  if (addr[left].dist > nn.dist) {
    memmove(&addr[left + 1], &addr[left], (size_t)(K - 1) * sizeof(eo_nb));
    addr[left] = nn;
    return left;
  }
  if (addr[right].dist < nn.dist) {
    addr[K] = nn;
    return K;
  }
  while (left < right - 1) {
    int64_t mid = (left + right) / 2;
    if (addr[mid].dist > nn.dist) right = mid; else left = mid;
  }
  while (left > 0) {
    if (addr[left].dist < nn.dist) break;
    if (addr[left].id == nn.id) return K + 1;
    left--;
  }
  if (addr[left].id == nn.id || addr[right].id == nn.id) return K + 1;
  memmove(&addr[right + 1], &addr[right], (size_t)(K - 1 - right) * sizeof(eo_nb));
  addr[right] = nn;
  return right;
}

typedef struct {
  const float* rows;
  int64_t n, d;
  int64_t search_length, out_degree, cand_pool;
  ivec* knng; /* n lists */
  ivec* nsg;  /* n lists */
  int64_t nav;
  unsigned seed;
} nsg_t;

/* The three GetNeighbors overloads (nsg.cpp:158-268, 271-378, 380-486) share one body; they differ in
 * the graph walked, whether the seed entries / the expanded entries are recorded in `fullset`, and
 * whether the has_calculated bitset is the caller's.  rand_r(&seed) fills the seed list when the
 * navigation node has fewer than search_length neighbours. */
static void get_neighbors(nsg_t* s, const float* query, eo_nb* resset /* search_length+1 */, ivec* graph, uint8_t* flags,
                          nbvec* fullset, int record_seeds, int record_expanded) {
  int64_t buffer_size = s->search_length;
  int64_t* init = (int64_t*)malloc(sizeof(int64_t) * (size_t)buffer_size);
  int64_t count = 0;
  for (int64_t i = 0; i < buffer_size && i < graph[s->nav].n; ++i) {
    init[i] = graph[s->nav].v[i];
    flags[init[i]] = 1;
    ++count;
  }
  while (count < buffer_size) {
    int64_t id = (int64_t)(rand_r(&s->seed) % (uint64_t)s->n);
    if (flags[id]) continue;
    init[count++] = id;
    flags[id] = 1;
  }
  for (int64_t i = 0; i < buffer_size; ++i) {
    int64_t id = init[i];
    eo_nb nb = {id, eo_fvec_l2sqr(s->rows + id * s->d, query, s->d), 0};
    resset[i] = nb;
    if (record_seeds) nbvec_push(fullset, nb);
  }
  nb_sort(resset, buffer_size);
  int64_t cursor = 0;
  while (cursor < buffer_size) {
    int64_t nearest_updated_pos = buffer_size;
    if (!resset[cursor].flag) {
      resset[cursor].flag = 1;
      int64_t start_pos = resset[cursor].id;
      ivec* lst = &graph[start_pos];
      for (int64_t i = 0; i < lst->n; ++i) {
        int64_t id = lst->v[i];
        if (flags[id]) continue;
        flags[id] = 1;
        eo_nb nn = {id, eo_fvec_l2sqr(s->rows + id * s->d, query, s->d), 0};
        if (record_expanded) nbvec_push(fullset, nn);
        if (nn.dist >= resset[buffer_size - 1].dist) continue;
        int64_t pos = insert_into_pool(resset, buffer_size, nn);
        if (pos < nearest_updated_pos) nearest_updated_pos = pos;
        /* "trick": resset.size() == search_length == buffer_size, so buffer_size never grows */
      }
    }
    if (cursor >= nearest_updated_pos) cursor = nearest_updated_pos; else ++cursor;
  }
  free(init);
}

/* SelectEdge (nsg.cpp:655-685): MRNG rule — keep p iff every kept r has dist(r,p) >= dist(v,p). */
static void select_edge(nsg_t* s, int64_t* cursor, eo_nb* pool, int64_t pool_n, eo_nb* result, int64_t* rn, int limit) {
  int64_t depth = limit ? s->cand_pool : pool_n;
  while (*rn < s->out_degree && *cursor < depth && (++(*cursor)) < pool_n) {
    eo_nb* p = &pool[*cursor];
    int ok = 1;
    for (int64_t t = 0; t < *rn; ++t) {
      float dist = eo_fvec_l2sqr(s->rows + result[t].id * s->d, s->rows + p->id * s->d, s->d);
      if (dist < p->dist) {
        ok = 0;
        break;
      }
    }
    if (ok) result[(*rn)++] = *p;
  }
}

/* SyncPrune's sort + SelectEdge on a caller-supplied candidate list (nsg.cpp:557-567, 655-685): distances node->candidate,
 * stable sort by distance, the node itself skipped if it sorts first, result = [closest] + SelectEdge(limit = depth > 0).
 * Returns the number of ids written to out (<= out_degree). */
int64_t eo_select_edge(const float* rows, int64_t d, int64_t node, const int64_t* cands, int64_t m, int64_t depth, int64_t out_degree,
                       int64_t* out) {
  nsg_t s;
  memset(&s, 0, sizeof(s));
  s.rows = rows;
  s.d = d;
  s.out_degree = out_degree;
  s.cand_pool = depth;
  eo_nb* pool = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(m > 0 ? m : 1));
  int64_t pn = 0;
  for (int64_t i = 0; i < m; ++i) {
    if (cands[i] < 0) continue;
    pool[pn].id = cands[i];
    pool[pn].dist = eo_fvec_l2sqr(rows + node * d, rows + cands[i] * d, d);
    pool[pn].flag = 1;
    ++pn;
  }
  int64_t rn = 0;
  if (pn > 0) {
    nb_sort(pool, pn);
    int64_t cursor = 0;
    if (pool[cursor].id == node) ++cursor;
    if (cursor < pn) {
      eo_nb* result = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(out_degree + 1));
      result[rn++] = pool[cursor];
      select_edge(&s, &cursor, pool, pn, result, &rn, depth > 0);
      for (int64_t i = 0; i < rn; ++i) out[i] = result[i].id;
      free(result);
    }
  }
  free(pool);
  return rn;
}

static int64_t g_nsg_n = 0;
static ivec* g_nsg = NULL;
static int64_t g_nsg_nav = 0;

static void free_lists(ivec* l, int64_t n) {
  if (!l) return;
  for (int64_t i = 0; i < n; ++i) free(l[i].v);
  free(l);
}

/* InterInsert for every node (nsg.cpp:531-536, 583-653), serial over v (the omp-for is orphaned: one thread runs the loop).
 * cut: n * out_degree distances, -1 terminated like cut_graph_dist; result: scratch of out_degree + 1 entries. */
static void inter_insert_all(nsg_t* s, float* cut, eo_nb* result) {
  const int64_t n = s->n, out_degree = s->out_degree;
  nbvec wait = {0, 0, 0};
  for (int64_t v = 0; v < n; ++v) {
    float* ndp = cut + v * out_degree;
    for (int64_t i = 0; i < out_degree; ++i) {
      if (ndp[i] == -1) break;
      int64_t cn = s->nsg[v].v[i];
      ivec* nsn = &s->nsg[cn];
      float* nsd = cut + cn * out_degree;
      wait.n = 0;
      int dup = 0;
      for (int64_t j = 0; j < out_degree; ++j) {
        if (nsd[j] == -1) break;
        if (v == nsn->v[j]) {
          dup = 1;
          break;
        }
        eo_nb e = {nsn->v[j], nsd[j], 0};
        nbvec_push(&wait, e);
      }
      if (dup) continue;
      eo_nb cur = {v, ndp[i], 0};
      nbvec_push(&wait, cur);
      if (wait.n > out_degree) {
        int64_t start = 0, rn = 0;
        nb_sort(wait.v, wait.n);
        result[rn++] = wait.v[start];
        select_edge(s, &start, wait.v, wait.n, result, &rn, 0);
        /* overwrites the first rn slots only: neither the id vector nor the -1 terminator is
         * shortened, so stale tail entries survive (nsg.cpp:632-639) */
        for (int64_t j = 0; j < rn; ++j) {
          nsn->v[j] = result[j].id;
          nsd[j] = result[j].dist;
        }
      } else {
        for (int64_t j = 0; j < out_degree; ++j) {
          if (nsd[j] == -1) {
            ivec_push(nsn, cur.id);
            nsd[j] = cur.dist;
            if (j + 1 < out_degree) nsd[j + 1] = -1;
            break;
          }
        }
      }
    }
  }
  free(wait.v);
}

/* The InterInsert stage alone, on caller-supplied per-node edge lists (ids: n * out_degree, deg[v] valid entries each, as Link
 * leaves them; distances are computed here as SyncPrune stores them).  out_ids: n * out_degree, -1 padded; out_deg[v]. */
void eo_inter_insert(const float* rows, int64_t n, int64_t d, const int64_t* ids, const int64_t* deg, int64_t out_degree,
                     int64_t* out_ids, int64_t* out_deg) {
  nsg_t s;
  memset(&s, 0, sizeof(s));
  s.rows = rows; s.n = n; s.d = d; s.out_degree = out_degree;
  s.nsg = (ivec*)calloc((size_t)n, sizeof(ivec));
  float* cut = (float*)malloc(sizeof(float) * (size_t)(n * out_degree > 0 ? n * out_degree : 1));
  for (int64_t v = 0; v < n; ++v) {
    for (int64_t i = 0; i < deg[v]; ++i) {
      const int64_t u = ids[v * out_degree + i];
      ivec_push(&s.nsg[v], u);
      cut[v * out_degree + i] = eo_fvec_l2sqr(rows + v * d, rows + u * d, d);
    }
    if (deg[v] < out_degree) cut[v * out_degree + deg[v]] = -1;
  }
  eo_nb* result = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(out_degree + 1));
  inter_insert_all(&s, cut, result);
  free(result);
  free(cut);
  for (int64_t v = 0; v < n; ++v) {
    out_deg[v] = s.nsg[v].n;
    for (int64_t i = 0; i < out_degree; ++i) out_ids[v * out_degree + i] = i < s.nsg[v].n ? s.nsg[v].v[i] : -1;
  }
  free_lists(s.nsg, n);
}

/* The per-node half of NsgIndex::Link (nsg.cpp:488-516): GetNeighbors on the kNN graph + SyncPrune (:540-580) for every node, in
 * node order.  Shared by eo_nsg_build (whose whole result is pinned bit-exactly to the reference) and eo_nsg_link. */
static void link_stage(nsg_t* s, uint8_t* flags, eo_nb* resset, float* cut, eo_nb* result) {
  const int64_t n = s->n, d = s->d, out_degree = s->out_degree;
  const float* rows = s->rows;
  nbvec full = {0, 0, 0};
  for (int64_t v = 0; v < n; ++v) {
    full.n = 0;
    memset(flags, 0, (size_t)n);
    get_neighbors(s, rows + v * d, resset, s->knng, flags, &full, 1, 1);
    /* SyncPrune (nsg.cpp:540-580) */
    for (int64_t i = 0; i < s->knng[v].n; ++i) {
      int64_t id = s->knng[v].v[i];
      if (flags[id]) continue;
      eo_nb nb = {id, eo_fvec_l2sqr(rows + v * d, rows + id * d, d), 1};
      nbvec_push(&full, nb);
    }
    nb_sort(full.v, full.n);
    int64_t cursor = 0, rn = 0;
    if (full.v[cursor].id == v) cursor++;
    result[rn++] = full.v[cursor];
    select_edge(s, &cursor, full.v, full.n, result, &rn, 1);
    float* dp = cut + v * out_degree;
    for (int64_t i = 0; i < rn; ++i) {
      ivec_push(&s->nsg[v], result[i].id);
      dp[i] = result[i].dist;
    }
    if (rn < out_degree) dp[rn] = -1;
  }
  free(full.v);
}

/* The Link stage alone on a given kNN graph and navigation node: what every node's edge list is BEFORE InterInsert.
 * out_ids [n][out_degree] (-1 padded), out_deg [n].  (With K >= search_length - the defaults, 100 >= 45 - GetNeighbors never
 * draws random start nodes, nsg.cpp:187, so the stage has no hidden state besides its inputs.) */
void eo_nsg_link(const float* rows, int64_t n, int64_t d, const int64_t* knn, int64_t K, int64_t search_length, int64_t out_degree,
                 int64_t cand_pool, int64_t nav, unsigned seed0, int64_t* out_ids, int64_t* out_deg) {
  nsg_t s;
  s.rows = rows; s.n = n; s.d = d;
  s.search_length = search_length; s.out_degree = out_degree; s.cand_pool = cand_pool;
  s.seed = seed0;
  s.nav = nav;
  s.knng = (ivec*)calloc((size_t)n, sizeof(ivec));
  s.nsg = (ivec*)calloc((size_t)n, sizeof(ivec));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < K; ++j)
      if (knn[i * K + j] >= 0) ivec_push(&s.knng[i], knn[i * K + j]);
  uint8_t* flags = (uint8_t*)calloc((size_t)n, 1);
  eo_nb* resset = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(search_length + 2));
  float* cut = (float*)malloc(sizeof(float) * (size_t)(n * out_degree));
  eo_nb* result = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(out_degree + 1));
  link_stage(&s, flags, resset, cut, result);
  for (int64_t v = 0; v < n; ++v) {
    out_deg[v] = s.nsg[v].n;
    for (int64_t j = 0; j < out_degree; ++j) out_ids[v * out_degree + j] = j < s.nsg[v].n ? s.nsg[v].v[j] : -1;
  }
  free(result);
  free(cut);
  free(resset);
  free(flags);
  free_lists(s.knng, n);
  free_lists(s.nsg, n);
}

/* NsgIndex::Build (nsg.cpp:45-99).  knn: n*K ids, -1 padded.  Result is kept in a static and fetched
 * with eo_nsg_fetch().  Returns the total number of edges.  seed0 = 100 is the reference's initial
 * value of the global rand_r state (nsg.cpp:19). */
int64_t eo_nsg_build(const float* rows, int64_t n, int64_t d, const int64_t* knn, int64_t K, int64_t search_length,
                     int64_t out_degree, int64_t cand_pool, unsigned seed0) {
  nsg_t s;
  s.rows = rows; s.n = n; s.d = d;
  s.search_length = search_length; s.out_degree = out_degree; s.cand_pool = cand_pool;
  s.seed = seed0;
  s.knng = (ivec*)calloc((size_t)n, sizeof(ivec));
  s.nsg = (ivec*)calloc((size_t)n, sizeof(ivec));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < K; ++j)
      if (knn[i * K + j] >= 0) ivec_push(&s.knng[i], knn[i * K + j]);
  uint8_t* flags = (uint8_t*)calloc((size_t)n, 1);
  eo_nb* resset = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(search_length + 2));

  /* InitNavigationPoint (nsg.cpp:101-155): float mean in row order, random start, search on knng */
  {
    float* center = (float*)calloc((size_t)d, sizeof(float));
    for (int64_t i = 0; i < n; i++)
      for (int64_t j = 0; j < d; j++) center[j] += rows[i * d + j];
    for (int64_t j = 0; j < d; j++) center[j] /= n;
    s.nav = (int64_t)(rand_r(&s.seed) % (uint64_t)n);
    get_neighbors(&s, center, resset, s.knng, flags, NULL, 0, 0);
    s.nav = resset[0].id;
    memset(flags, 0, (size_t)n);
    free(center);
  }

  /* Link (nsg.cpp:488-538) */
  float* cut = (float*)malloc(sizeof(float) * (size_t)(n * out_degree));
  {
    eo_nb* result = (eo_nb*)malloc(sizeof(eo_nb) * (size_t)(out_degree + 1));
    link_stage(&s, flags, resset, cut, result);
    inter_insert_all(&s, cut, result);
    free(result);
  }
  free(cut);

  /* CheckConnectivity / DFS / FindUnconnectedNode (nsg.cpp:687-775) */
  {
    uint8_t* linked = (uint8_t*)calloc((size_t)n, 1);
    int64_t linked_count = 0;
    int64_t root = s.nav;
    int64_t* stack = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
    nbvec pool = {0, 0, 0};
    while (linked_count < n) {
      /* DFS */
      int64_t sp = 0, start = root;
      stack[sp++] = root;
      if (!linked[root]) linked_count++;
      linked[root] = 1;
      while (sp > 0) {
        int64_t next = n + 1;
        for (int64_t i = 0; i < s.nsg[start].n; i++) {
          if (!linked[s.nsg[start].v[i]]) {
            next = s.nsg[start].v[i];
            break;
          }
        }
        if (next == n + 1) {
          --sp;
          if (sp == 0) break;
          start = stack[sp - 1];
          continue;
        }
        start = next;
        linked[start] = 1;
        stack[sp++] = start;
        ++linked_count;
      }
      if (linked_count >= n) break;
      /* FindUnconnectedNode */
      int64_t id = n;
      for (int64_t i = 0; i < n; i++)
        if (!linked[i]) {
          id = i;
          break;
        }
      if (id == n) break;
      pool.n = 0;
      memset(flags, 0, (size_t)n);
      get_neighbors(&s, rows + id * d, resset, s.nsg, flags, &pool, 0, 1);
      nb_sort(pool.v, pool.n);
      int found = 0;
      for (int64_t i = 0; i < pool.n; i++) {
        if (linked[pool.v[i].id]) {
          root = pool.v[i].id;
          found = 1;
          break;
        }
      }
      if (!found) {
        while (1) {
          int64_t rid = (int64_t)(rand_r(&s.seed) % (uint64_t)n);
          if (linked[rid]) {
            root = rid;
            break;
          }
        }
      }
      ivec_push(&s.nsg[root], id);
    }
    free(linked);
    free(stack);
    free(pool.v);
  }
  free(flags);
  free(resset);
  free_lists(s.knng, n);
  free_lists(g_nsg, g_nsg_n);
  g_nsg = s.nsg;
  g_nsg_n = n;
  g_nsg_nav = s.nav;
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) total += s.nsg[i].n;
  return total;
}

/* flatten to CSR exactly as BuildFromVectorTable does (ann_graph_segment.cpp:222-241) */
int64_t eo_nsg_fetch(int64_t* off, int64_t* nbr) {
  int64_t o = 0;
  for (int64_t i = 0; i < g_nsg_n; ++i) {
    off[i] = o;
    for (int64_t j = 0; j < g_nsg[i].n; ++j) nbr[o + j] = g_nsg[i].v[j];
    o += g_nsg[i].n;
  }
  off[g_nsg_n] = o;
  return g_nsg_nav;
}

/* ------------------------------------------------------------------------------------------------
 * a15  ann_graph_<field>.bin (ann_graph_segment.cpp:156-199 write, :39-98 read):
 *   i64 n, i64 first_record_id, i64 offsets[n+1], i64 neighbors[offsets[n]], i64 navigation_point
 */
int eo_graph_file_write(const char* path, int64_t n, int64_t first_id, const int64_t* off, const int64_t* nbr, int64_t nav) {
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  fwrite(&n, 8, 1, f);
  fwrite(&first_id, 8, 1, f);
  fwrite(off, 8, (size_t)(n + 1), f);
  fwrite(nbr, 8, (size_t)off[n], f);
  fwrite(&nav, 8, 1, f);
  fclose(f);
  return 0;
}
/* two-phase read: call with off == NULL to get n and the edge count, then with buffers. */
int eo_graph_file_read(const char* path, int64_t* n, int64_t* edges, int64_t* first_id, int64_t* off, int64_t* nbr, int64_t* nav) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  int64_t hdr[2];
  if (fread(hdr, 8, 2, f) != 2) { fclose(f); return -2; }
  *n = hdr[0];
  *first_id = hdr[1];
  if (!off) {
    if (fseek(f, 8 * hdr[0], SEEK_CUR) != 0) { fclose(f); return -2; }
    int64_t e;
    if (fread(&e, 8, 1, f) != 1) { fclose(f); return -2; }
    *edges = e;
    fclose(f);
    return 0;
  }
  if (fread(off, 8, (size_t)(hdr[0] + 1), f) != (size_t)(hdr[0] + 1)) { fclose(f); return -2; }
  *edges = off[hdr[0]];
  if (fread(nbr, 8, (size_t)*edges, f) != (size_t)*edges) { fclose(f); return -2; }
  if (fread(nav, 8, 1, f) != 1) { fclose(f); return -2; }
  fclose(f);
  return 0;
}
