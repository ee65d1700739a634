#!/usr/bin/env python
"""Headline benchmark: QPS at recall@10 >= 0.999 on 10M x 768 L2, k=10, batch=1024 (BASELINE.json configs[2]),
synthetic i.i.d. U[0,1) fp32 rows generated on the device, inputs resident in HBM when the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]       (N > 1: under torch.distributed.run one rank per GPU, as the driver launches it;
                                                            started plain, bench.py launches the N ranks itself; --inproc: one process,
                                                            eps_index_create_sharded over N devices)

One "step" = one batch of queries through the hot path (eps_index_search: flat scan or graph traversal -> top-k).
With N > 1 the corpus is hash-sharded by row index (row i lives on rank i mod N), every rank answers the same query
batch on its shard, and the per-shard top-k lists are merged after ONE RCCL all-gather of a packed per-rank buffer
[ids int64[batch][k] | dist f32[batch][k]].  Weak scaling, per-GPU work fixed, in one of two forms:
  --scale rows (default):  every rank holds --rows rows (BASELINE configs[4]: 10M per GPU = 80M rows at N = 8) and the
                           batch stays --batch; whole-job queries/s should stay flat while the corpus grows,
                           `work_rate` (query*rows/s) is the number that scales;
  --scale queries:         the corpus stays --rows in total, each rank holds rows/N of it and the batch grows to
                           N x --batch.
`--mode graph` measures the traversal kernel as the primary (builds the graph over --rows first; minutes at 10M).  The
default flat run adds a SECONDARY traversal measurement on the first --graph-rows rows (reference defaults T=4, L=500).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HBM traffic of the dominant kernel is a PMC measurement (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, corrected as
# MI355X_MICROARCH.md prescribes: gfx950 FETCH_SIZE counts 128-B requests at 64 B -> x2; KiB units) and cannot be taken inside this
# run: `roofline.traffic` is null here and `roofline.traffic_reference` names the committed profile of the same launch shape (file,
# sha256 of the file as it lies in this tree, the bytes it shows, and which round's kernel it was taken on).
TRAFFIC_REF = {
    "mfma8": {"file": "profiles/r5_pmc_10Mx768_b1024.csv", "bytes_per_launch": (2 * 2753192 + 3823) * 1024.0, "algorithmic_bytes": 7280256 * 768.0,
              "launch": "mfma_filter_kernel_v7<2, FM_IDS, int8>, 7,280,256 rows x 1024 queries (last of 6 stages), occurrence 11 of the PMC passes",
              "taken_on": "r5 library (scripts/run_flat_profile_r5.sh; the filter kernel itself is r4's)"},
    "mfma": {"file": "profiles/r2_pmc_10Mx768_b1024.csv", "bytes_per_launch": (2 * 7340245 + 7460) * 1024.0, "algorithmic_bytes": 9262720 * 1536.0,
             "launch": "mfma_filter_kernel_v7<2, FM_IDS> (fp16), 9,262,720 rows x 1024 queries", "taken_on": "r2 kernel"},
    "graph_T4_L500": {"file": "profiles/r5_traverse_10Mx768_pmc.csv", "bytes_per_launch": (2 * 19176716 + 1055271) * 1024.0, "algorithmic_bytes": 41.1e9,
                      "launch": "traverse2_kernel T=4 L=500 batch 1024, 10M-node device-built graph, 8-bit prefilter, edge constants, visited stamps (occurrences 0, 1)",
                      "taken_on": "r5 kernel (scripts/run_10m_graph_r5.sh)"},
}


def power_leg(torch, step, queries, seconds, device_index):
    """The same step, back to back for `seconds`, while a thread samples the GPU's shader clock and socket power (amdsmi, in-process):
    what the dominant kernel's roofline fraction has to be read against on a board that runs it at its power limit.  Outside the timed
    region of the contract line; None where the SMI library cannot be used."""
    import threading
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[device_index]
    except Exception as e:   # noqa: BLE001
        return {"failed": repr(e)[:200]}
    clk, pw, stop = [], [], threading.Event()

    def num(v):
        try:
            return float(v)
        except Exception:   # noqa: BLE001
            return None

    def sample():
        while not stop.is_set():
            try:
                c = num(amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX).get("clk"))
                pi = amdsmi.amdsmi_get_power_info(h)
                w = num(pi.get("current_socket_power"))
                if w is None or w <= 0:
                    w = num(pi.get("average_socket_power"))
                if c:
                    clk.append(c)
                if w:
                    pw.append(w)
            except Exception:   # noqa: BLE001
                pass
            stop.wait(0.02)
    out = {}
    try:
        try:
            lim = num(amdsmi.amdsmi_get_power_info(h).get("power_limit"))
            out["power_limit_w"] = (lim / 1e6 if lim and lim > 1e5 else lim)   # (reported in microwatts by this library version)
        except Exception:   # noqa: BLE001
            pass
        th = threading.Thread(target=sample, daemon=True)
        torch.cuda.synchronize()
        th.start()
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                step(queries[steps % len(queries)])
                steps += 1
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        stop.set()
        th.join(1.0)
        half = len(clk) // 2     # (the second half of the run: the clock has settled under the cap by then)
        out.update({"steps": steps, "seconds": el, "ms_per_step_sustained": 1e3 * el / steps, "samples": len(clk),
                    "sclk_mhz_under_load": float(np.median(clk[half:])) if clk else None,
                    "socket_power_w_under_load": float(np.median(pw[len(pw) // 2:])) if pw else None})
    except Exception as e:   # noqa: BLE001
        out["failed"] = repr(e)[:200]
    finally:
        stop.set()
        try:
            amdsmi.amdsmi_shut_down()
        except Exception:   # noqa: BLE001
            pass
    return out


def traffic_ref(key):
    import hashlib
    r = dict(TRAFFIC_REF[key])
    path = os.path.join(ROOT, r["file"])
    r["sha256"] = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
    return r


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_GATHER_CEILING_GBS = 6050.0   # measured here: random whole-row gathers (768 B and 3 KB rows, 30 GB table, 2-8 rows in flight per lane group,
                                  # 4-8 wavefronts per SIMD) with nothing else in the kernel: 5.99-6.10 TB/s (scripts/lab/gather_peak.hip, profiles/r4_gather_peak.txt)
MFMA_F16_PEAK_TF = 2500.0  # dense bf16/f16 MFMA peak (nominal, 2.4 GHz)
MFMA_F16_SUSTAINED_TF = 1814.0  # measured: v_mfma_f32_32x32x16_f16 alone, operands toggling like data, 1.82 GHz (scripts/lab/mfma_peak.hip)
MFMA_I8_PEAK_TOPS = 5000.0      # dense 8-bit MFMA peak: twice the fp16 rate (MI355X_MICROARCH.md lists the FP8 dense peak ~5 P and I8 at ~2x bf16)
MFMA_I8_SUSTAINED_TOPS = 3424.0  # measured: v_mfma_i32_32x32x32_i8 alone on bytes in [-127, 127], 1.74 GHz (profiles/r3_mfma_peak_i8_vs_fp16.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (--scale rows) or in total (--scale queries)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="EUCLIDEAN")
    ap.add_argument("--mode", default="flat", choices=["flat", "graph"])
    ap.add_argument("--engine", default="auto", choices=["auto", "stream", "mfma", "mfma8"],
                    help="flat scan engine: auto (the library's choice: int8 first pass), stream (fp32), mfma (fp16 filter), mfma8 (int8 filter)")
    ap.add_argument("--data", default="uniform", choices=["uniform", "clustered", "manifold"],
                    help="clustered: SURVEY 8d secondary set, 1000 Gaussian clusters sigma=0.1; manifold: a 16-dimensional uniform latent embedded "
                         "linearly in --dim dimensions + 1 %% noise (low intrinsic dimension, as learned embeddings have; tertiary, not in BASELINE)")
    ap.add_argument("--T", type=int, default=4, help="graph: IntraQueryThreads")
    ap.add_argument("--L", type=int, default=500, help="graph: SearchQueueSize")
    ap.add_argument("--load-graph", default=None)
    ap.add_argument("--save-graph", default=None)
    ap.add_argument("--recall-queries", type=int, default=1024)
    ap.add_argument("--graph-rows", type=int, default=1_000_000, help="rows of the secondary traversal measurement (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline legs (0 = skip)")
    ap.add_argument("--power-seconds", type=float, default=2.0, help="N = 1: after the timed region the same step runs back to back for this long while the shader clock and "
                    "socket power are sampled (roofline.under_load; 0 = skip)")
    ap.add_argument("--no-e2e", dest="e2e", action="store_false", help="skip the end-to-end (host -> host, pipelined) repetition of the timed steps")
    ap.add_argument("--scale", default="rows", choices=["queries", "rows"])
    ap.add_argument("--inproc", action="store_true", help="N > 1 in ONE process: eps_index_create_sharded over devices 0..N-1 (the form the single-process "
                                                          "reference DBMS uses), per-shard lists merged on the caller's device; no torch.distributed")
    ap.add_argument("--configs", default="c1,c2,c4,secondary,embedding", help="further BASELINE configs measured into the line's `configs` object at N = 1: c1 (configs[0]: 100k x 128 through both "
                                                     "`epsilla` modules), embedding (unit-norm Gaussian rows with dominant dimensions, COSINE, on the --rows table), c2 (1M x 768, batch 1, latency), "
                                                     "c4 (COSINE + ID < N filter on the --rows table, batch --batch), secondary (SURVEY 8d clustered set + the manifold set at --graph-rows: "
                                                     "flat and graph legs); 'none' skips them")
    return ap.parse_args()


def gen_rows(torch, n, d, seed, device, kind="uniform", centres=None):
    """synthetic fp32 rows generated on the device in slabs (seeded per rank): i.i.d. U[0,1), or the clustered mixture"""
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1 << 19
    for s in range(0, n, step):
        e = min(n, s + step)
        if kind == "uniform":
            X[s:e] = torch.rand((e - s, d), generator=g, device=device, dtype=torch.float32)
        elif kind == "manifold":
            z = torch.rand((e - s, centres.shape[0]), generator=g, device=device, dtype=torch.float32)
            X[s:e] = z @ centres + 0.01 * torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        else:
            a = torch.randint(0, centres.shape[0], (e - s,), generator=g, device=device)
            X[s:e] = centres[a] + 0.1 * torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
    return X


def exact_topk_torch(torch, X, q, k, id_base, id_stride):
    """fp32 direct-form exact scan of ONE query in torch (independent ground truth; not timed)."""
    best_d, best_i = None, None
    step = 1 << 20
    for s in range(0, X.shape[0], step):
        e = min(X.shape[0], s + step)
        dd = ((X[s:e] - q) ** 2).sum(1)
        kk = min(k, e - s)
        v, i = torch.topk(dd, kk, largest=False)
        i = (i + s) * id_stride + id_base
        if best_d is None:
            best_d, best_i = v, i
        else:
            v = torch.cat([best_d, v])
            i = torch.cat([best_i, i])
            o = torch.argsort(v, stable=True)[:k]
            best_d, best_i = v[o], i[o]
    return best_d, best_i


def recall_of(got, want):
    got, want = np.asarray(got), np.asarray(want)
    return float(np.mean([len(set(got[i].tolist()) & set(want[i].tolist())) / float(want.shape[1]) for i in range(len(want))]))


class CpuBaseline:
    """The reference's own CPU paths (oracle/_ref = the reference's sources compiled verbatim) timed on this box's host cores on
    the SAME rows, bounded by sampling queries, not rows.  Reported baselines only - never part of the product
    path; everything under oracle/ that bench.py touches is touched in this class."""

    def __init__(self, torch, X):
        from oracle import pyoracle
        self.py = pyoracle
        self.torch = torch
        self.n, self.d = X.shape
        self.cores = os.cpu_count() or 1
        self.ref = pyoracle.Ref() if pyoracle.ref_available() else None
        self.ptr = None
        if self.ref is not None:
            self.threads = int(self.ref.L.ref_omp_max_threads())
            self.arr, self.ptr = self.ref.alloc_rows(self.n, self.d, self.threads)   # page-aligned, first-touched by the scan's own OpenMP schedule
            self.copy_s = self.load(X)

    def load(self, X):
        t0 = time.time()
        step = 1 << 19
        rows = min(self.n, X.shape[0])   # (a smaller table overwrites the head of the buffer: the legs that follow scan only those rows)
        for s in range(0, rows, step):
            e = min(rows, s + step)
            self.arr[s:e] = X[s:e].cpu().numpy()
        return time.time() - t0

    def close(self):
        if self.ptr is not None:
            self.ref.free_rows(self.ptr)
            self.ptr = None

    def port(self, X, Q, budget_s):
        """oracle/_ref absent: the plain-C restatement, scalar - a far weaker baseline, labelled "port" """
        orc = self.py.Oracle()
        rows = X[:200_000].cpu().numpy()
        q = Q[0].cpu().numpy()
        t0 = time.time()
        done = 0
        while done < 4 and time.time() - t0 < budget_s:
            orc.dist_batch(0, rows, q)
            done += 1
        sec = time.time() - t0
        return {"value": done * rows.shape[0] / sec / self.n, "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": "%d scalar scans of a %d-row sample, scaled to %d rows (oracle/_ref absent)" % (done, rows.shape[0], self.n)}

    def bruteforce(self, Qh, k, gt_ids, budget_s, rows=None, metric=0, gpu_ids=None):
        """leg "bruteforce": VecSearchExecutor::BruteForceSearch (:717-768) over the first `rows` rows, OpenMP over all cores -
        exact, so it is the reference's answer at recall >= 0.999 whenever its traversal needs a queue so long that it evaluates
        most of the table (uniform data: profiles/r2_graph_*.jsonl)"""
        ref, n, d, threads = self.ref, rows or self.n, self.d, self.threads
        nb = 2
        ids, ds, sec = ref.bruteforce_many(self.ptr, n, d, Qh[:nb], k, metric=metric, threads=threads)
        per = float(np.mean(sec[1:])) if nb > 1 else float(sec[0])
        more = int(max(0, min(len(Qh) - nb, (budget_s - float(np.sum(sec))) / max(per, 1e-3))))
        if more > 0:
            ids2, ds2, sec2 = ref.bruteforce_many(self.ptr, n, d, Qh[nb:nb + more], k, metric=metric, threads=threads)
            ids, sec = np.concatenate([ids, ids2]), np.concatenate([sec, sec2])
        nbq = len(sec)
        qps = (nbq - 1) / float(np.sum(sec[1:])) if nbq > 1 else 1.0 / float(sec[0])   # first query pays the scratch allocation
        return {"leg": "bruteforce", "what": "reference VecSearchExecutor::BruteForceSearch over %d x %d rows, %d OpenMP threads" % (n, d, threads),
                "qps": qps, "queries": nbq, "p50_ms": 1e3 * float(np.median(sec[1:] if nbq > 1 else sec)), "p99_ms": 1e3 * float(np.max(sec[1:] if nbq > 1 else sec)),
                "recall_at_10": recall_of(ids, gt_ids[:nbq]) if gt_ids is not None else None, "evals_per_query": n, "effective_GBps": qps * n * d * 4 / 1e9,
                # the GPU's answers for the same queries (the timed path's last step), position by position against the reference's own
                "gpu_headline_answers_equal": (int(sum(bool(np.array_equal(ids[i], gpu_ids[i])) for i in range(nbq))) if gpu_ids is not None else None)}

    def distance_scan(self, Qh, k, gt_ids, budget_s):
        """the distance phase of that brute force on its own (GetDistFunc under `omp parallel for`, :729-735) + an O(n) top-k
        selection instead of the reference's serial compaction and std::sort of all n candidates: what the host's memory system
        delivers to the reference's distance kernel"""
        try:
            t0 = time.time()
            nscan = 0
            sel_ids = []
            while nscan < min(4, len(Qh)) and (nscan == 0 or time.time() - t0 < budget_s):
                dist = self.ref.dist_batch(0, self.arr, Qh[nscan])
                idx = np.argpartition(dist, k)[:k]
                sel_ids.append(idx[np.lexsort((idx, dist[idx]))])
                nscan += 1
            sec_scan = (time.time() - t0) / nscan
            return {"leg": "distance_scan_only",
                    "what": "reference fvec_L2sqr via GetDistFunc over %d x %d rows under omp parallel for (%d threads) + numpy argpartition top-%d; "
                            "not a path the reference has (its BruteForceSearch adds a serial compaction and a std::sort of all candidates)" % (self.n, self.d, self.threads, k),
                    "qps": 1.0 / sec_scan, "queries": nscan, "recall_at_10": recall_of(np.stack(sel_ids), gt_ids[:nscan]), "effective_GBps": self.n * self.d * 4 / sec_scan / 1e9}
        except Exception as e:   # a report only
            return {"leg": "distance_scan_only", "what": "failed: %r" % (e,), "qps": 0.0, "queries": 0, "recall_at_10": 0.0}

    def graph(self, graph, Qh, k, Lcpu, budget_s, E=None, T=4):
        """leg "graph": SearchImpl under the reference's concurrency model, E executors x T OpenMP workers (E x T = cores by default;
        E = 1: single-query latency), at SearchQueueSize Lcpu, on the device-built graph of the first rows"""
        ref = self.ref
        off, nbr, nav, gn, ggt = graph
        g = ref.graph_from_arrays(off, nbr, nav)
        E = E or max(1, self.threads // T)
        nqg = min(len(Qh), 4 * E if E > 1 else 32)
        ids_g, ds_g, lat, wall = ref.pool_search(g, self.ptr, self.d, Qh[:nqg], k, E=E, T=T, L=Lcpu)
        reps = int(max(0, min(16, budget_s / max(wall, 1e-3) - 1)))
        if reps > 0 and E > 1:
            nqg2 = min(len(Qh), nqg * (reps + 1))
            ids_g, ds_g, lat, wall = ref.pool_search(g, self.ptr, self.d, Qh[:nqg2], k, E=E, T=T, L=Lcpu)
            nqg = nqg2
        ref.L.ref_graph_free(g)
        return {"leg": "graph", "what": "reference SearchImpl on the device-built graph of the first %d rows, %d executor(s) x %d OpenMP workers, SearchQueueSize %d" % (gn, E, T, Lcpu),
                "qps": nqg / wall, "queries": nqg, "p50_ms": 1e3 * float(np.median(lat)), "p99_ms": 1e3 * float(np.percentile(lat, 99)),
                "recall_at_10": recall_of(ids_g, ggt[:nqg]), "rows": gn}

    def prefilter(self, idc_host, flt, Qh, k, metric, gt_ids=None, gpu_ids=None):
        """the reference's PreFilterBruteForceSearch (:770-831) with its own filter parser / ExprEvaluator (BASELINE configs[3])"""
        ids, ds, cnt, sec = self.ref.prefilter_many(self.ptr, self.n, self.d, idc_host, flt, Qh, k, metric=metric, threads=self.threads)
        return {"leg": "prefilter_bruteforce", "what": "reference PreFilterBruteForceSearch, filter %r, %d x %d rows, %d OpenMP threads" % (flt, self.n, self.d, self.threads),
                "qps": len(sec) / float(np.sum(sec)), "queries": len(sec), "p50_ms": 1e3 * float(np.median(sec)), "p99_ms": 1e3 * float(np.max(sec)),
                "visible_rows": int(cnt[0]), "recall_at_10": recall_of(ids, gt_ids[:len(sec)]) if gt_ids is not None else None,
                "gpu_headline_answers_equal": (int(sum(bool(np.array_equal(ids[i], gpu_ids[i])) for i in range(len(sec)))) if gpu_ids is not None else None)}


def cpu_baseline(cpu, args, X, Q, gt_ids, graph, budget_s, gpu_ids=None):
    """the `cpu_baseline` object of the headline line (BASELINE configs[2]): legs bruteforce / distance_scan_only / graph"""
    n, d, k = cpu.n, cpu.d, args.k
    if cpu.ref is None:
        return cpu.port(X, Q, budget_s)
    Qh = Q.cpu().numpy()
    legs = [cpu.bruteforce(Qh, k, gt_ids, budget_s * 0.6, gpu_ids=gpu_ids), cpu.distance_scan(Qh, k, gt_ids, budget_s * 0.1)]
    if graph is not None:
        legs.append(cpu.graph(graph, Qh, k, args.L if args.mode == "graph" else 500, budget_s * 0.3))
    # the baseline of record is the best path the REFERENCE itself offers at recall >= 0.999 on the full table
    ok = [l for l in legs if l["leg"] in ("bruteforce", "graph") and l["recall_at_10"] >= 0.999 and l.get("rows", n) == n]
    best = max(ok, key=lambda l: l["qps"]) if ok else legs[0]
    return {"value": best["qps"], "unit": "queries/s", "cores": cpu.threads, "kind": "reference", "best_leg": best["leg"],
            "sample": "%s: %d queries on the full %d x %d table (rows copied from the GPU in %.1f s, parallel first touch); host has %d logical cores"
                      % (best["what"], best["queries"], n, d, cpu.copy_s, cpu.cores),
            "legs": legs}


def config_c2(amd, torch, args, X, qlast, dev, stream, local_rank, cpu, graph_index, graph_for_cpu):
    """BASELINE configs[1]: 1M x 768 L2, k = 10, batch = 1 - single-query latency on one MI355X, inputs resident in HBM; every call
    is timed from issue to torch.cuda.synchronize().  Exact engines (fp32 stream scan, int8 matrix filter) and the traversal at the
    reference's defaults; beside them the reference's BruteForceSearch and SearchImpl (one executor, T = 4) on this box's cores."""
    n1, d, k = args.graph_rows, args.dim, args.k
    out = {"workload": "%d x %d L2, k=%d, batch=1: one query per call, sequential" % (n1, d, k)}
    ix = amd.GpuIndex(d, args.metric, device=local_rank)
    ix.set_stream(stream)
    ix.attach_rows(X[:n1])
    o = (torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))
    nq1 = min(200, qlast.shape[0])

    def latency(index, **kw):
        for i in range(3):
            index.search(qlast[i:i + 1], k, out=o, **kw)
        torch.cuda.synchronize()
        lat, res, one = [], [], 0
        for i in range(nq1):
            t0 = time.perf_counter()
            index.search(qlast[i:i + 1], k, out=o, **kw)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            res.append(o[0][0].cpu().numpy().copy())
            one += int(index.stats().get("one_pass", 0))
        km = index.kernel_times(64)
        return {"p50_ms": 1e3 * float(np.median(lat)), "p99_ms": 1e3 * float(np.percentile(lat, 99)), "qps": nq1 / float(np.sum(lat)), "queries": nq1,
                "main_kernel_ms": float(np.median(km)) if km else None, "one_pass_calls": one}, np.stack(res)
    gpu = {}
    gpu["stream"], gt1 = latency(ix, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    if gpu["stream"]["main_kernel_ms"]:
        ach = n1 * d * 4 / (gpu["stream"]["main_kernel_ms"] * 1e-3) / 1e9
        gpu["stream"]["roofline"] = {"bound": "hbm", "kernel": "flat_scan_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                     "note": "a %.1f GB table: partly served by L2 / Infinity Cache on repeated scans" % (n1 * d * 4 / 1e9)}
    gpu["stream"]["recall_at_10"] = 1.0
    ix.search(qlast[:64], k, out=(torch.empty((64, k), dtype=torch.int64, device=dev), torch.empty((64, k), dtype=torch.float32, device=dev),
                                  torch.empty((64,), dtype=torch.int32, device=dev)), mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)   # builds the 8-bit mirror
    gpu["mfma_i8"], r8 = latency(ix, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    gpu["mfma_i8"]["recall_at_10"] = recall_of(r8, gt1)
    gpu["mfma_i8"]["one_pass"] = int(ix.stats().get("one_pass", 0))   # r4: 1 = ONE streaming pass over the 8-bit mirror + one re-rank (stream8_kernel)
    # r5: how many of the leg's calls the one-pass form answered; the others overflowed a wavefront's list and were answered by the staged chain
    gpu["mfma_i8"]["one_pass_fallbacks"] = gpu["mfma_i8"]["queries"] - gpu["mfma_i8"]["one_pass_calls"]
    if gpu["mfma_i8"]["one_pass"]:
        # the pass itself against the HBM roofline: timed in a second run (an event pair around it costs the untimed call ~10 us)
        amd.set_tuning("EPS_ONE_PASS_TIMED", "1")
        try:
            timed, _ = latency(ix, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        finally:
            amd.set_tuning("EPS_ONE_PASS_TIMED", None)
        if timed["main_kernel_ms"]:
            row_bytes = (d + 255) // 256 * 256 + 4          # the mirror's row pitch + the row's int32 start value
            ach = n1 * row_bytes / (timed["main_kernel_ms"] * 1e-3) / 1e9
            gpu["mfma_i8"]["roofline"] = {"bound": "hbm", "kernel": "stream8_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                          "kernel_ms": timed["main_kernel_ms"], "algorithmic_bytes": n1 * row_bytes,
                                          "note": "one pass over the 8-bit mirror (%d bytes per row) + 4 bytes of start value per row; timed run p50 %.3f ms" % (row_bytes - 4, timed["p50_ms"])}
    gpu["auto"], ra = latency(ix, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    gpu["auto"]["recall_at_10"] = recall_of(ra, gt1)
    if graph_index is not None:
        gpu["graph_T4_L500"], rg = latency(graph_index, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=500, local_queue=500)
        gpu["graph_T4_L500"]["recall_at_10"] = recall_of(rg, gt1)

    # r5 (late): the one-pass form beyond configs[1]'s own shape - k = 64 (128 table slots per query) and a compiled filter program (evaluated once per
    # row into a bitset by one launch in front of the pass) - one query per call, each answer compared with the fp32 stream engine's under the same setting
    def probe(kk, calls=100, check=10):
        oo = (torch.empty((1, kk), dtype=torch.int64, device=dev), torch.empty((1, kk), dtype=torch.float32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))
        rr = (torch.empty_like(oo[0]), torch.empty_like(oo[1]), torch.empty_like(oo[2]))
        for i in range(3):
            ix.search(qlast[i:i + 1], kk, out=oo, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        torch.cuda.synchronize()
        lat, one, same = [], 0, 0
        for i in range(calls):
            t0 = time.perf_counter()
            ix.search(qlast[i:i + 1], kk, out=oo, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            one += int(ix.stats().get("one_pass", 0))
            if i < check:
                ix.search(qlast[i:i + 1], kk, out=rr, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
                torch.cuda.synchronize()
                same += int(torch.equal(oo[0], rr[0]) and torch.equal(oo[1], rr[1]))
        return {"k": kk, "p50_ms": 1e3 * float(np.median(lat)), "p99_ms": 1e3 * float(np.percentile(lat, 99)), "queries": calls, "one_pass_calls": one,
                "answers_equal_to_the_stream_engine": "%d of %d" % (same, check)}
    try:
        wide = {"what": "one query per call on the one-pass form beyond k <= 16 / no filter program (r5); p50 issue -> sync, inputs in HBM"}
        wide["k64"] = probe(64)
        attr = torch.arange(n1, dtype=torch.int32, device=dev).view(torch.uint8).reshape(n1, 4)
        ix.set_filter_program([("i32", 0), ("const", 3), ("%",), ("const", 1), ("=",)], attr, stride=4)
        wide["filter_program_id_mod_3_eq_1_k%d" % k] = probe(k)
        amd.set_tuning("EPS_S8_FILTER_PROGRAMS", "0")
        try:
            wide["filter_program_id_mod_3_eq_1_k%d_staged_chain" % k] = probe(k, calls=40, check=0)
        finally:
            amd.set_tuning("EPS_S8_FILTER_PROGRAMS", None)
        ix.set_filter_program(None)
        out["one_pass_widened"] = wide
    except Exception as e:
        out["one_pass_widened"] = {"failed": repr(e)}
    ix.close()
    exact = [v for v in gpu.values() if v["recall_at_10"] >= 0.999]
    out["gpu"] = gpu
    out["value"] = {"p50_ms": min(v["p50_ms"] for v in exact), "what": "best exact engine, end to end per call (host issue + device + sync), inputs in HBM"}
    if cpu is not None and cpu.ref is not None:
        try:
            Qh = qlast[:64].cpu().numpy()
            legs = [cpu.bruteforce(Qh, k, gt1, 4.0, rows=n1, gpu_ids=r8)]   # (r8 = the one-pass engine's answers for the same queries: the leg's `value`)
            if graph_for_cpu is not None:
                gg = (graph_for_cpu[0], graph_for_cpu[1], graph_for_cpu[2], graph_for_cpu[3], gt1)
                legs.append(cpu.graph(gg, Qh, k, 500, 3.0, E=1, T=4))
            out["cpu_reference"] = {"cores": cpu.threads, "legs": legs}
        except Exception as e:
            out["cpu_reference"] = {"failed": repr(e)}
    return out


def config_c4(amd, torch, args, X, qlast, dev, stream, local_rank, cpu):
    """BASELINE configs[3]: 10M x 768 COSINE + `ID < N` metadata filter, k = 10, batch 1024.  Rows normalised as at insert
    (table_segment_mvp.cpp:574-587), queries as TableMVP::Search does (table_mvp.cpp:333-343); the filter is evaluated inside the
    exact scan (Config::PreFilter semantics: the reference's post-filter over the top-L walk starves).  Beside it the
    reference's PreFilterBruteForceSearch with its own expression evaluator."""
    n, d, k, b = X.shape[0], args.dim, args.k, qlast.shape[0]
    out = {"workload": "%dM x %d COSINE + ID < N, k=%d, batch=%d, filter evaluated inside the exact scan" % (n // 1_000_000, d, k, b)}
    Xn = torch.empty_like(X)
    for s in range(0, n, 1 << 19):
        e = min(n, s + (1 << 19))
        Xn[s:e] = X[s:e]
    amd.normalize_rows(Xn, only_if_nonzero=True, device=local_rank, stream=stream)
    Qn = qlast.clone()
    amd.normalize_rows(Qn, only_if_nonzero=False, device=local_rank, stream=stream)
    torch.cuda.synchronize()
    idc = torch.arange(n, dtype=torch.int32, device=dev)
    ix = amd.GpuIndex(d, "COSINE", device=local_rank)
    ix.set_stream(stream)
    ix.attach_rows(Xn)
    o = (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), dtype=torch.float32, device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
    g64 = (torch.empty((64, k), dtype=torch.int64, device=dev), torch.empty((64, k), dtype=torch.float32, device=dev), torch.empty((64,), dtype=torch.int32, device=dev))
    out["gpu"] = {}
    gts, gots = {}, {}
    for sel in (0.5, 0.1, 0.9):
        bound = int(n * sel)
        ix.set_int_filter(idc, "<", bound)
        kw = dict(mode=amd.MODE_REFERENCE, prefilter=1)
        ix.search(Qn, k, out=o, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ix.search(Qn, k, out=o, **kw)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / 3
        st = ix.stats()
        got = o[0].cpu().numpy().copy()
        ix.search(Qn[:64], k, out=g64, flat_engine=amd.FLAT_STREAM, **kw)
        torch.cuda.synchronize()
        gts[sel] = g64[0].cpu().numpy().copy()
        gots[sel] = got[:64]
        out["gpu"]["ID < %d (%d %%)" % (bound, int(sel * 100))] = {
            "qps": b / sec, "ms_per_step": 1e3 * sec, "recall_at_10": recall_of(got[:64], gts[sel]), "recall_check": "64 queries vs the fp32 stream engine with the same filter",
            "all_results_pass_the_filter": bool((got < bound).all()), "operand_bits": int(st.get("main_kernel_bits", 0)), "rerank_rows_per_query": st["rerank_rows"] / float(b),
            "main_kernel_ms": float(np.median(ix.kernel_times(3)))}
    ix.close()
    if cpu is not None and cpu.ref is not None:
        try:
            copy_s = cpu.load(Xn)
            idc_host = np.arange(n, dtype=np.int32)
            Qh = Qn[:2].cpu().numpy()
            legs = [cpu.prefilter(idc_host, "ID < %d" % int(n * 0.5), Qh, k, 1, gts[0.5], gpu_ids=gots[0.5]),
                    cpu.prefilter(idc_host, "ID < %d" % int(n * 0.1), Qh[:1], k, 1, gts[0.1], gpu_ids=gots[0.1])]
            out["cpu_reference"] = {"cores": cpu.threads, "legs": legs, "normalised_rows_copied_in_s": copy_s}
        except Exception as e:
            out["cpu_reference"] = {"failed": repr(e)}
    del Xn
    return out


def config_c1(args):
    """BASELINE configs[0]: 100k x 128 VECTOR_FLOAT EUCLIDEAN through engine/bindings - the `epsilla` CPython module
    (bindings/python/interface.cpp:260-331, unmodified in both builds): insert() in 1000-row JSON batches, 1000 query() calls, k = 10.
    Two modules, each in a process of its own (scripts/epsilla_module_driver.py): the reference's own build (oracle/_ref/pymod: the CPU
    engine) and the drop-in build (dropin/_build: the same binding over libepsilla_gfx950).  The binding never rebuilds, so both answer
    with the exact scan; the answers are compared id by id."""
    import subprocess
    import tempfile
    drv = os.path.join(ROOT, "scripts", "epsilla_module_driver.py")
    mods = (("reference_cpu", os.path.join(ROOT, "oracle", "_ref", "pymod")), ("gfx950_dropin", os.path.join(ROOT, "dropin", "_build")))
    out = {"workload": "100k x 128 EUCLIDEAN through the `epsilla` CPython module: insert() in 1000-row JSON batches, 1000 sequential query() calls (one host vector in, "
                       "a list of dicts out), k=10; per call everything included (JSON, GIL, H2D / D2H on the GPU side)"}
    answers = {}
    for name, mdir in mods:
        if not os.path.exists(os.path.join(mdir, "epsilla.so")):
            out[name] = {"skipped": "%s/epsilla.so is not built" % os.path.relpath(mdir, ROOT)}
            continue
        with tempfile.TemporaryDirectory() as td:
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, drv, mdir, os.path.join(td, "db"), "c1", "100000", "128", "1000"], capture_output=True, text=True, timeout=420, cwd=ROOT)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("EPSILLA_JSON ")]
            if r.returncode != 0 or not line:
                out[name] = {"failed": (r.stderr or r.stdout)[-400:]}
                continue
            j = json.loads(line[-1][len("EPSILLA_JSON "):])
            answers[name] = j["flat"].pop("results")
            leg = {"module": j.get("module"), "insert_s": j["insert_s"], "query": j["flat"], "process_s": time.perf_counter() - t0}
            if "graph" in j:   # (drop-in only: its additive rebuild() + query_batch(); the reference binding has neither, interface.h:22-32)
                j["graph"].pop("results", None)
                j["query_batch"].pop("results", None)
                leg["after_rebuild"] = {"rebuild_s": j.get("rebuild_s"), "query": j["graph"], "query_batch": j["query_batch"]}
            out[name] = leg
    if len(answers) == 2:
        a, g = answers["reference_cpu"], answers["gfx950_dropin"]
        out["same_ids"] = int(sum(x[0] == y[0] for x, y in zip(a, g)))
        out["queries"] = len(a)
        out["max_rel_distance_error"] = float(max(max((abs(u - v) / max(abs(u), 1e-12) for u, v in zip(x[1], y[1])), default=0.0) for x, y in zip(a, g)))
        if out["reference_cpu"]["query"]["qps"]:
            out["gpu_over_cpu"] = out["gfx950_dropin"]["query"]["qps"] / out["reference_cpu"]["query"]["qps"]
    return out


def config_embedding_like(amd, torch, args, X, dev, stream, local_rank):
    """The shape learned embeddings have and the U[0,1) recipe does not: unit-norm rows, Gaussian coordinates, a few
    dominant dimensions (the first 8 coordinates carry 4 x the scale of the rest), COSINE.  Written IN PLACE over the headline table (this leg
    runs last).  What it asks of the 8-bit first pass: the grid must cover coordinates of very different spread; the bench line says which
    operand width served the batch, how many rows reached the fp32 re-rank, and the recall against the fp32 stream scan."""
    n, d, k, b = X.shape[0], args.dim, args.k, args.batch
    g = torch.Generator(device=dev).manual_seed(77)
    scale = torch.ones((d,), dtype=torch.float32, device=dev)
    scale[:8] = 4.0
    for s in range(0, n, 1 << 19):
        e = min(n, s + (1 << 19))
        X[s:e] = torch.randn((e - s, d), generator=g, device=dev, dtype=torch.float32) * scale
    amd.normalize_rows(X, only_if_nonzero=True, device=local_rank, stream=stream)
    Q = torch.randn((b, d), generator=torch.Generator(device=dev).manual_seed(78), device=dev, dtype=torch.float32) * scale
    amd.normalize_rows(Q, only_if_nonzero=False, device=local_rank, stream=stream)
    torch.cuda.synchronize()
    ix = amd.GpuIndex(d, "COSINE", device=local_rank)
    ix.set_stream(stream)
    ix.attach_rows(X)
    o = (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), dtype=torch.float32, device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
    for _ in range(3):
        ix.search(Q, k, out=o, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search(Q, k, out=o, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / 5
    st = ix.stats()
    km = ix.kernel_times(5)
    got = o[0].cpu().numpy().copy()
    nrec = min(128, b)
    g2 = (torch.empty((nrec, k), dtype=torch.int64, device=dev), torch.empty((nrec, k), dtype=torch.float32, device=dev), torch.empty((nrec,), dtype=torch.int32, device=dev))
    ix.search(Q[:nrec], k, out=g2, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    torch.cuda.synchronize()
    out = {"workload": "%dM x %d COSINE, unit-norm rows with Gaussian coordinates, 8 dominant dimensions (4 x scale), k=%d, batch=%d, exact flat scan (the library's engine choice)"
                       % (n // 1_000_000, d, k, b),
           "qps": b / sec, "ms_per_step": 1e3 * sec, "recall_at_10": recall_of(got[:nrec], g2[0].cpu().numpy()), "recall_check": "%d queries vs the fp32 stream scan" % nrec,
           "operand_bits": int(st.get("main_kernel_bits", 0)), "rerank_rows_per_query": st["rerank_rows"] / float(b), "overflow_queries": st["overflow_queries"],
           "main_kernel_ms": float(np.median(km)) if km else None}
    ix.close()
    return out


def config_secondary(amd, torch, args, dev, stream, local_rank, cpu, kind):
    """The sets the BASELINE recipe is NOT ("additionally report one clustered synthetic set clearly labelled as secondary"):
    `clustered` = 1000 Gaussian clusters (centres U[0,1)^d, sigma 0.1), `manifold` = a 16-dimensional uniform latent embedded linearly in d
    dimensions + 1 % noise (what learned embeddings look like; tertiary).  --graph-rows rows, batch --batch: the exact flat scan (which
    operand width the library chose, whether it probed the 8-bit pass and declined it) and the traversal on a device-built NSG at the
    reference's T = 4 for two queue sizes, beside the reference's own SearchImpl on the same graph.  Shows where each path wins."""
    n1, d, k, b = args.graph_rows, args.dim, args.k, args.batch
    gc = torch.Generator(device=dev).manual_seed(41)
    centres = torch.rand((1000, d), generator=gc, device=dev) if kind == "clustered" else 0.25 * torch.randn((16, d), generator=gc, device=dev)
    X1 = gen_rows(torch, n1, d, 142, dev, kind, centres)
    Q1 = gen_rows(torch, b, d, 143, dev, kind, centres)
    out = {"workload": "%d x %d L2, k=%d, batch=%d, synthetic %s set (NOT the BASELINE recipe)" % (n1, d, k, b, kind)}
    ix = amd.GpuIndex(d, "EUCLIDEAN", device=local_rank)
    ix.set_stream(stream)
    ix.attach_rows(X1)
    o = (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), dtype=torch.float32, device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
    nrec = min(256, b)
    g = (torch.empty((nrec, k), dtype=torch.int64, device=dev), torch.empty((nrec, k), dtype=torch.float32, device=dev), torch.empty((nrec,), dtype=torch.int32, device=dev))
    ix.search(Q1[:nrec], k, out=g, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    ix.synchronize()
    gt = g[0].cpu().numpy().copy()

    def timed(index, reps=3, **kw):
        index.search(Q1, k, out=o, **kw)
        index.synchronize()
        first = index.stats()
        index.search(Q1, k, out=o, **kw)
        index.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            index.search(Q1, k, out=o, **kw)
        index.synchronize()
        return (time.perf_counter() - t0) / reps, index.stats(), first
    sec, st, first = timed(ix, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    flat_ids = o[0].cpu().numpy().copy()
    out["flat"] = {"qps": b / sec, "ms_per_step": 1e3 * sec, "recall_at_10": recall_of(flat_ids[:nrec], gt), "recall_check": "%d queries vs the fp32 stream engine" % nrec,
                   "operand_bits": int(st.get("main_kernel_bits", 0)), "rerank_rows_per_query": st["rerank_rows"] / float(b), "overflow_queries": int(st["overflow_queries"]),
                   "first_call_probed_and_declined_the_8bit_pass": bool(first.get("i8_declined", 0))}
    t0 = time.perf_counter()
    ix.build(n1)
    ix.synchronize()
    out["graph_build_s"] = time.perf_counter() - t0
    gn_, ge_, _ = ix.graph_info()
    out["graph"] = {}
    for L in (100, 500):
        sec, st, _ = timed(ix, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=L, local_queue=L)
        out["graph"]["T4_L%d" % L] = {"qps": b / sec, "ms_per_step": 1e3 * sec, "recall_at_10": recall_of(o[0].cpu().numpy(), flat_ids),
                                      "evals_per_query": st["dist_evals"] / float(b), "fp32_rows_per_query": st["rerank_rows"] / float(b)}
    out["graph"]["avg_degree"] = ge_ / float(gn_)
    if cpu is not None and cpu.ref is not None:
        try:
            cpu.load(X1)
            Qh = Q1.cpu().numpy()
            off, nbr, nav = ix.get_graph()
            legs = [cpu.bruteforce(Qh[:8], k, flat_ids, 2.0, rows=n1, gpu_ids=flat_ids)]
            for L in (100, 500):
                legs.append(cpu.graph((off, nbr, nav, n1, flat_ids), Qh, k, L, 2.0))
            out["cpu_reference"] = {"cores": cpu.threads, "legs": legs}
        except Exception as e:
            out["cpu_reference"] = {"failed": repr(e)}
    ix.close()
    del X1
    return out


def main_inproc(args):
    """--inproc: the OTHER multi-GPU form of SURVEY 8e - one process, eps_index_create_sharded over devices 0..N-1 (what the
    single-process reference DBMS would hold): every shard's rows generated on its own device and handed over in place
    (eps_index_attach_shard_rows), queries and results on device 0, per-shard lists pushed peer to peer and merged there.  Same
    workload and JSON contract as the one-process-per-GPU form; no torch.distributed, no RCCL."""
    import torch
    import vectordb_amd as amd
    from vectordb_amd.build import build
    build()
    G = args.gpus
    ndev = torch.cuda.device_count()
    backend_note = ""
    if ndev < G:
        if os.environ.get("EPS_BENCH_BACKEND", "nccl") == "nccl":
            raise SystemExit("bench.py --inproc --gpus %d: only %d device(s) visible" % (G, ndev))
        backend_note = " (EPS_BENCH_BACKEND != nccl: %d shards on %d device(s) - plumbing check, not a measurement)" % (G, ndev)
    devices = [s % max(1, ndev) for s in range(G)]
    n, d, b, k = args.rows, args.dim, args.batch, args.k
    if args.scale == "queries":
        n, b = args.rows // G, args.batch * G
    grp = amd.GpuIndex(d, args.metric, devices=devices)
    parts = []
    for s in range(G):
        dev = torch.device("cuda", devices[s])
        torch.cuda.set_device(dev)
        parts.append(gen_rows(torch, n, d, 42 + s, dev, "uniform"))
        torch.cuda.synchronize(dev)
        grp.attach_shard_rows(s, parts[-1])
    dev0 = torch.device("cuda", devices[0])
    torch.cuda.set_device(dev0)
    gq = torch.Generator(device=dev0).manual_seed(43)
    queries = [torch.rand((b, d), generator=gq, device=dev0, dtype=torch.float32) for _ in range(args.steps + args.warmup)]
    ids = torch.empty((b, k), dtype=torch.int64, device=dev0)
    dd = torch.empty((b, k), dtype=torch.float32, device=dev0)
    cnt = torch.empty((b,), dtype=torch.int32, device=dev0)
    torch.cuda.synchronize()
    engine = {"auto": amd.FLAT_AUTO, "stream": amd.FLAT_STREAM, "mfma": amd.FLAT_MFMA, "mfma8": amd.FLAT_MFMA_I8}[args.engine]
    skw = dict(mode=amd.MODE_FLAT, flat_engine=engine)
    for w in range(args.warmup):
        grp.search(queries[w], k, out=(ids, dd, cnt), **skw)
    grp.synchronize()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        grp.search(queries[args.warmup + s_], k, out=(ids, dd, cnt), **skw)
    grp.synchronize()
    elapsed = time.perf_counter() - t0
    st = grp.stats()
    main_ms = grp.kernel_times(64)[-args.steps:]
    got = ids.cpu().numpy().copy()
    nrec = min(args.recall_queries, b)
    g = (torch.empty((nrec, k), dtype=torch.int64, device=dev0), torch.empty((nrec, k), dtype=torch.float32, device=dev0), torch.empty((nrec,), dtype=torch.int32, device=dev0))
    torch.cuda.synchronize()
    grp.search(queries[-1][:nrec], k, out=g, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    grp.synchronize()
    recall = recall_of(got[:nrec], g[0].cpu().numpy())
    qps = b * args.steps / elapsed
    kernel_ms = float(np.mean(main_ms)) if main_ms else 0.0
    bits = int(st.get("main_kernel_bits", 0))
    # st sums the shards' counters: rows of the timed launch over ALL shards, time = the slowest shard
    flops = 2.0 * float(st.get("main_kernel_queries", b) or b) * float(st["main_kernel_rows"]) * d
    peak = (MFMA_I8_PEAK_TOPS if bits == 8 else MFMA_F16_PEAK_TF) * G
    roof = {"bound": "mfma", "kernel": "mfma_filter_kernel_v7<%s>, largest stage, summed over the %d shards (time = the slowest shard)" % ("int8" if bits == 8 else "fp16", G),
            "achieved": flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms else None, "peak": peak, "unit": "TOP/s" if bits == 8 else "TFLOP/s", "traffic": None,
            "kernel_ms_per_launch": kernel_ms}
    roof["frac"] = roof["achieved"] / peak if roof["achieved"] else None
    print(json.dumps({
        "metric": "QPS @ recall@10>=0.999, 10Mx768 L2", "value": qps, "unit": "queries/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (exact fp32 distances; %d-bit matrix pass as a lower-bound filter, survivors re-ranked in fp32)" % bits, "data": "synthetic",
        "recall_at_10": recall, "recall_check": {"queries": nrec, "ground_truth": "exact fp32 stream scan through the same shard group"},
        "config": {"workload": "%dM x %d L2 exact flat scan, k=%d, batch=%d per step, %d rows per GPU, %d GPU(s), rows_total=%d, ONE process: eps_index_create_sharded%s"
                               % ((n * G) // 1_000_000, d, k, b, n, G, n * G, backend_note),
                   "mode": "flat", "engine": args.engine, "parallelism": "in-process shard group x%d: row-hash shards, peer-to-peer push of packed top-k, merge on device 0" % G},
        "roofline": roof, "stats": {"rerank_rows_per_query": st["rerank_rows"] / float(b), "overflow_queries": st["overflow_queries"]},
        "work_rate": {"value": qps * n * G, "unit": "query*rows/s"}}))
    grp.close()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: re-exec under torch.distributed.run, one rank per GPU (the same
    line the driver uses), and pass its exit code on.  Never silently measures fewer GPUs than asked for."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and not args.inproc and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.inproc:
        return main_inproc(args)
    import torch
    import torch.distributed as dist
    import vectordb_amd as amd
    from vectordb_amd.build import build

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or let bench.py launch the ranks itself: "
                         "unset WORLD_SIZE)" % (args.gpus, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # EPS_BENCH_BACKEND=gloo lets the N > 1 path be exercised with several ranks on ONE GPU (exchange staged through the
    # host); the real multi-GPU run uses nccl (= RCCL over xGMI)
    backend = os.environ.get("EPS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)   # before the process group: the nccl barrier below runs on the current device
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def all_gather(dst, src):
        if backend == "nccl":
            dist.all_gather_into_tensor(dst, src)
        else:
            hs = src.cpu()
            parts = [torch.empty_like(hs) for _ in range(world)]
            dist.all_gather(parts, hs)
            dst.copy_(torch.stack(parts).reshape(dst.shape))
    if rank == 0:
        build()
    if world > 1:
        dist.barrier()
    # r6: the exchange step of the measured path belongs to the LIBRARY (eps_exchange: its own RCCL communicator, ncclAllGather + k-way merge in one
    # call).  torch.distributed only bootstraps it - the 128-byte unique id travels in one broadcast - and keeps the control plane (barriers, the
    # max over ranks, the ranks' identity).  Ranks that share a device (EPS_BENCH_BACKEND=gloo: the one-GPU plumbing runs) cannot form an RCCL
    # communicator; they keep the host-staged all-gather + eps_merge_topk_packed, and the line says so (exchange.collective).
    xchg = None
    xchg_note = None
    if world > 1 and backend == "nccl" and os.environ.get("EPS_BENCH_EXCHANGE", "eps") == "eps":
        box = [None]
        if rank == 0:
            try:
                box[0] = amd.Exchange.unique_id()
            except Exception as e:  # noqa: BLE001
                box[0] = "failed: %r" % (e,)
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], (bytes, bytearray)):
            ok = 1
            try:
                xchg = amd.Exchange(rank, world, box[0], device=local_rank)
            except Exception as e:  # noqa: BLE001
                ok, xchg_note = 0, "eps_exchange_create failed on rank %d: %r" % (rank, e)
            okt = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", local_rank))
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) == 0:   # (all ranks or none)
                if xchg is not None:
                    xchg.close()
                xchg = None
                xchg_note = xchg_note or "eps_exchange_create failed on another rank"
        else:
            xchg_note = "eps_exchange_unique_id %s" % (box[0],)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n, d, b, k = args.rows, args.dim, args.batch, args.k
    if world > 1 and args.scale == "queries":
        n = args.rows // world          # this rank's shard of the fixed corpus (row i lives on rank i mod world)
        b = args.batch * world          # every rank answers the whole (larger) batch on its shard

    centres = None
    if args.data == "clustered":
        centres = torch.rand((1000, d), generator=torch.Generator(device=dev).manual_seed(41), device=dev)
    if args.data == "manifold":
        centres = 0.25 * torch.randn((16, d), generator=torch.Generator(device=dev).manual_seed(41), device=dev)
    X = gen_rows(torch, n, d, 42 + rank, dev, args.data, centres)   # this rank's shard: global row id = local*world + rank
    gq = torch.Generator(device=dev).manual_seed(43)                # same queries on every rank
    if args.data == "uniform":
        queries = [torch.rand((b, d), generator=gq, device=dev, dtype=torch.float32) for _ in range(args.steps + args.warmup)]
    else:
        queries = [gen_rows(torch, b, d, 43 + 1000 * i, dev, args.data, centres) for i in range(args.steps + args.warmup)]

    stream = torch.cuda.current_stream().cuda_stream
    ix = amd.GpuIndex(d, args.metric, device=local_rank)
    ix.set_stream(stream)
    ix.attach_rows(X)
    ix.set_id_map(rank, world)
    engine = {"auto": amd.FLAT_AUTO, "stream": amd.FLAT_STREAM, "mfma": amd.FLAT_MFMA, "mfma8": amd.FLAT_MFMA_I8}[args.engine]
    mode = amd.MODE_FLAT if args.mode == "flat" else amd.MODE_GRAPH
    skw = dict(mode=mode, flat_engine=engine)
    build_s = None
    if args.mode == "graph":
        t0 = time.perf_counter()
        if args.load_graph:
            ix.load_graph(args.load_graph)
        else:
            ix.build(n)
        ix.synchronize()
        build_s = time.perf_counter() - t0
        if args.save_graph:
            ix.save_graph(args.save_graph)
        skw.update(intra_threads=args.T, master_queue=args.L, local_queue=args.L)

    # one packed result buffer per rank: ids int64[b][k] then dist f32[b][k]; the search writes straight into it.  Three of everything
    # (slot = step % 3): the end-to-end run copies step i's results to the host while steps i + 1, i + 2 compute
    pack_n = (b * k * 12 + 7) // 8 * 8
    packs = [torch.empty((pack_n,), dtype=torch.uint8, device=dev) for _ in range(3)]
    idss = [p_[: b * k * 8].view(torch.int64).view(b, k) for p_ in packs]
    dds = [p_[b * k * 8: b * k * 12].view(torch.float32).view(b, k) for p_ in packs]
    cnt = torch.empty((b,), dtype=torch.int32, device=dev)
    pack, ids, dd = packs[0], idss[0], dds[0]
    xev = []   # N > 1: (before all-gather, after all-gather, after merge) event triples of the timed steps
    if world > 1:
        gathered = torch.empty((world, pack_n), dtype=torch.uint8, device=dev)
        m_ds = [torch.empty((b, k), dtype=torch.float32, device=dev) for _ in range(3)]
        m_is = [torch.empty((b, k), dtype=torch.int64, device=dev) for _ in range(3)]

    def step(q, slot=0, timed=False):
        ix.search(q, k, out=(idss[slot], dds[slot], cnt), **skw)
        if world > 1:
            # the one exchange step of the path: ONE all-gather of the packed per-shard top-k, then a k-way merge
            if timed:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
            if xchg is not None:   # one library call: ncclAllGather of the packed lists + the merge (its own hipEvents split the two)
                xchg.allgather_merge(idss[slot], dds[slot], m_is[slot], m_ds[slot], stream=stream)
                if timed:
                    ev[1].record()
            else:
                all_gather(gathered, packs[slot])
                if timed:
                    ev[1].record()
                amd.merge_topk_packed(gathered, pack_n, b * k * 8, world, b, k, m_ds[slot], m_is[slot], device=local_rank, stream=stream)
            if timed:
                ev[2].record()
                xev.append(ev)
            return m_ds[slot], m_is[slot]
        return dds[slot], idss[slot]

    for w in range(args.warmup):
        step(queries[w])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        out_d, out_i = step(queries[args.warmup + s], 0, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    got_i = out_i.clone()
    # device time of the dominant kernel of every timed step: hipEvent pairs the library recorded on this stream, read
    # back only now (no host sync inside the timed region)
    main_ms = ix.kernel_times(64)[-args.steps:]
    st = ix.stats()
    xchg_us = [(e[0].elapsed_time(e[1]) * 1e3, e[1].elapsed_time(e[2]) * 1e3) for e in xev]
    if xchg is not None:   # (the library's own event triples of the timed steps: all-gather | merge)
        xchg_us = xchg.times_us(64)[-args.steps:]

    # ---- the same steps END TO END ("QPS end-to-end including H2D of queries and D2H of results"; the reference's entry takes a
    # host vector per call, table_mvp.cpp:359-380): every batch starts in host memory and its ids / distances end in host memory,
    # inside the timed region.  `value` stays the device-resident rate (the contract: inputs resident in HBM); this one is reported beside it.
    e2e = None
    if args.e2e:
        # Through the C ABI's own host path: eps_index_search with HOST query and result pointers - the library copies the batch up, searches, copies
        # the ids / distances down, inside the call (csrc/index.cpp) - exactly what the drop-in's VecSearchExecutor hands it.  Staging by the caller
        # on a second stream was measured and dropped (scripts/lab/e2e_dbg.py; a two- and a three-slot form of this loop at 10M x 768):
        # a copy that runs CONCURRENTLY with the traversal's one long launch - 1024 workgroups that fill every CU exactly once - takes a workgroup
        # slot from it and the launch runs a second round (0.71-0.82 of the device-resident rate; 0.93-0.97 through the library's path), and for the
        # flat scan's chain of short launches both forms measure the same (0.97).
        nst = args.steps
        qn = [queries[args.warmup + s].cpu().numpy() for s in range(nst)]
        rh_i = [torch.empty((b, k), dtype=torch.int64).pin_memory() for _ in range(nst)] if world > 1 else None
        rh_d = [torch.empty((b, k), dtype=torch.float32).pin_memory() for _ in range(nst)] if world > 1 else None
        last = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(nst):
            if world == 1:
                last = ix.search(qn[s], k, **skw)                # host in, host out
            else:
                o_d, o_i = step(qn[s], s % 3)                    # host queries in; the per-shard lists meet on the devices (all-gather, merge) ...
                rh_i[s].copy_(o_i, non_blocking=True)            # ... and the merged answer comes down
                rh_d[s].copy_(o_d, non_blocking=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        t = torch.tensor([el2], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el2 = float(t.item())
        last_ids = last[0] if world == 1 else rh_i[nst - 1].numpy()
        same = bool((last_ids == got_i.cpu().numpy()).all())   # (the same queries as the device-resident run's last step: the same answer, now in host memory)
        e2e = {"value": b * nst / el2, "unit": "queries/s", "ms_per_step": 1e3 * el2 / nst, "frac_of_device_resident": elapsed / el2,
               "last_step_equals_device_resident_run": same,
               "what": "eps_index_search with HOST query%s pointers (pageable numpy): H2D of the batch, the search%s, D2H of ids + distances, all inside the call and "
                       "inside the timed region; no caller-side staging (see the comment in bench.py for the staged form that measured worse)"
                       % ((" and result", "") if world == 1 else ("", ", all-gather, merge"))}
    # clock and power under this workload (N = 1, after the timed region): the filter kernel runs against the board's power limit
    power = power_leg(torch, step, queries, args.power_seconds, local_rank) if (world == 1 and rank == 0 and args.power_seconds > 0) else None

    # ---- N > 1: what proves that N ranks on N devices took part (the driver's SCALE run cannot be watched from here)
    ranks_info = None
    if world > 1:
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "device_ordinal": int(torch.cuda.current_device()), "device_name": torch.cuda.get_device_name(local_rank),
                "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": int(getattr(props, "pci_bus_id", -1)), "pid": os.getpid(),
                "main_kernel_ms": float(np.mean(main_ms)) if main_ms else None, "search_rows": int(n),
                "all_gather_us_mean": float(np.mean([x[0] for x in xchg_us])) if xchg_us else None,
                "merge_us_mean": float(np.mean([x[1] for x in xchg_us])) if xchg_us else None}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        if backend == "nccl":
            seen = set((r_["device_ordinal"], r_["device_uuid"], r_["pci_bus_id"]) for r_ in ranks_info)
            if len(seen) != world:
                raise SystemExit("bench.py: %d ranks under nccl but only %d distinct devices: %r" % (world, len(seen), ranks_info))

    # ---- recall@10 of the last batch: exact ground truth from the fp32 direct-form stream scan (an independent code path
    # of the library, itself pinned to the oracle by the tests) for --recall-queries queries, and a torch fp32 scan for 16
    nrec = min(args.recall_queries, b)
    qlast = queries[-1]
    g_ids = torch.empty((nrec, k), dtype=torch.int64, device=dev)
    g_dd = torch.empty((nrec, k), dtype=torch.float32, device=dev)
    g_cnt = torch.empty((nrec,), dtype=torch.int32, device=dev)
    ix.search(qlast[:nrec], k, out=(g_ids, g_dd, g_cnt), mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    ix.synchronize()
    if world > 1:
        gp = torch.empty(((nrec * k * 12 + 7) // 8 * 8,), dtype=torch.uint8, device=dev)
        gp[: nrec * k * 8] = g_ids.view(torch.uint8).flatten()
        gp[nrec * k * 8: nrec * k * 12] = g_dd.view(torch.uint8).flatten()
        gg = torch.empty((world, gp.numel()), dtype=torch.uint8, device=dev)
        all_gather(gg, gp)
        t_d = torch.empty((nrec, k), dtype=torch.float32, device=dev)
        t_i = torch.empty((nrec, k), dtype=torch.int64, device=dev)
        amd.merge_topk_packed(gg, gp.numel(), nrec * k * 8, world, nrec, k, t_d, t_i, device=local_rank, stream=stream)
        torch.cuda.synchronize()
        g_ids = t_i
    gt = g_ids.cpu().numpy()
    recall = recall_of(got_i[:nrec].cpu().numpy(), gt)
    ntorch = min(16, nrec)
    torch_hits = 0
    for qi in range(ntorch):
        td, ti = exact_topk_torch(torch, X, qlast[qi], k, rank, world)
        if world > 1:
            ad = torch.empty((world, k), dtype=torch.float32, device=dev)
            ai = torch.empty((world, k), dtype=torch.int64, device=dev)
            all_gather(ad, td.contiguous())
            all_gather(ai, ti.contiguous())
            o = torch.argsort(ad.flatten(), stable=True)[:k]
            ti = ai.flatten()[o]
        torch_hits += len(set(ti.tolist()) & set(gt[qi].tolist()))
    gt_vs_torch = torch_hits / float(ntorch * k)

    # ---- secondary: the traversal kernel at the reference's defaults on a device-built graph of the first rows
    secondary = None
    graph_for_cpu = None
    ix2 = None
    if args.mode == "flat" and world == 1 and args.graph_rows and args.graph_rows <= n:
        gn = args.graph_rows
        ix2 = amd.GpuIndex(d, args.metric, device=local_rank)
        ix2.set_stream(stream)
        ix2.attach_rows(X[:gn])
        t1 = time.perf_counter()
        ix2.build(gn)
        ix2.synchronize()
        gbuild = time.perf_counter() - t1
        gn_, ge_, gnav = ix2.graph_info()
        o2 = (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), dtype=torch.float32, device=dev),
              torch.empty((b,), dtype=torch.int32, device=dev))
        ix2.search(qlast, k, out=o2, mode=amd.MODE_FLAT)
        ix2.synchronize()
        ggt = o2[0].cpu().numpy().copy()
        gkw = dict(mode=amd.MODE_GRAPH, intra_threads=4, master_queue=500, local_queue=500)
        ix2.search(qlast, k, out=o2, **gkw)
        ix2.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            ix2.search(qlast, k, out=o2, **gkw)
        ix2.synchronize()
        gel = (time.perf_counter() - t1) / 3
        gms = float(np.mean(ix2.kernel_times(3)))
        gst = ix2.stats()
        alg = amd.traversal_gather_bytes(gst, d, ge_ / float(gn_), 500 * b)
        secondary = {"what": "traverse2_kernel (SearchImpl, IntraQueryThreads=4, SearchQueueSize=500) on the device-built NSG of the first %d rows, batch %d" % (gn, b),
                     "build_s": gbuild, "avg_degree": ge_ / float(gn_), "qps": b / gel, "recall_at_10": recall_of(o2[0].cpu().numpy(), ggt),
                     "evals_per_query": gst["dist_evals"] / float(b), "expansions_per_query": gst["expansions"] / float(b),
                     "roofline": {"bound": "hbm", "achieved": alg / (gms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms_per_launch": gms,
                                  "note": "algorithmic gather bytes (vectordb_amd.traversal_gather_bytes); a %d-row table (%.1f GB) is partly served by L2/Infinity Cache - the HBM-only figure is the 10M-row run under profiles/" % (gn, gn * d * 4 / 1e9)}}
        if args.cpu_seconds > 0:
            off, nbr, nav = ix2.get_graph()
            graph_for_cpu = (off, nbr, nav, gn, ggt)

    if args.mode == "graph" and world == 1 and args.cpu_seconds > 0:
        off, nbr, nav = ix.get_graph()
        graph_for_cpu = (off, nbr, nav, n, gt)     # the CPU graph leg searches the same device-built graph

    if rank == 0:
        qps = b * args.steps / elapsed
        kernel_ms = float(np.mean(main_ms)) if main_ms else 0.0   # hipEvent pair around the dominant launch of every timed step
        krows = float(st["main_kernel_rows"])
        kq = float(st.get("main_kernel_queries", b) or b)
        used_mfma = st.get("rerank_rows", 0) > 0
        if args.mode == "graph":
            n_, e_, nav_ = ix.graph_info()
            alg_bytes = amd.traversal_gather_bytes(st, d, e_ / float(n_), min(args.L, n) * b)
            roof = {"bound": "hbm", "kernel": "traverse2_kernel (gather: X*(8+4*deg) adjacency bytes + per evaluation the 8-bit mirror row (d+4 bytes), the fp32 row (4d) only for "
                                              "seeds and for neighbours the 8-bit bound cannot rule out; E, X and the fp32 reads counted by the kernel)",
                    "fp32_rows_per_query": st["rerank_rows"] / float(b),
                    "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "traffic": None,
                    "traffic_reference": traffic_ref("graph_T4_L500") if (n == 10_000_000 and b == 1024 and d == 768 and args.T == 4 and args.L == 500 and args.data == "uniform" and st["rerank_rows"] > 0) else None}
        elif used_mfma:
            # algorithmic flops of the timed launch: 2 * batch * rows * d, on the fp16 dense MFMA roof
            flops = 2.0 * kq * krows * d   # queries x rows of the timed launch (batches > 2048 run in slices)
            bits = int(st.get("main_kernel_bits", 16))
            roof = {"bound": "mfma", "kernel": "mfma_filter_kernel_v7<%s> (largest of the filter stages: %d of %d rows x %d of %d queries)" % ("int8" if bits == 8 else "fp16", krows, n, kq, b),
                    "achieved": flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms else None,
                    "peak": MFMA_I8_PEAK_TOPS if bits == 8 else MFMA_F16_PEAK_TF, "unit": "TOP/s" if bits == 8 else "TFLOP/s", "operand_bits": bits,
                    # PMC bytes are not measurable inside this run: null, and the committed profile of exactly this launch shape beside it
                    "traffic": None,
                    "traffic_reference": (traffic_ref("mfma") if (bits == 16 and krows == 9262720) else traffic_ref("mfma8") if (bits == 8 and krows == 7280256) else None)
                                         if (n == 10_000_000 and b == 1024 and d == 768 and kq == 1024) else None,
                    "algorithmic_bytes": krows * ((d + 255) // 256 * 256 if bits == 8 else (d + 63) // 64 * 64 * 2)}
            # the same kernel runs once per filter stage: all stage launches of the LAST timed step (hipEvent pairs around each)
            if st.get("filter_ms_all", 0) > 0 and st.get("filter_rows_all", 0) > 0:
                fl_all = 2.0 * kq * float(st["filter_rows_all"]) * d
                roof["all_stage_launches"] = {"kernel_ms": st["filter_ms_all"], "rows": int(st["filter_rows_all"]),
                                              "achieved": fl_all / (st["filter_ms_all"] * 1e-3) / 1e12, "frac": fl_all / (st["filter_ms_all"] * 1e-3) / 1e12 / roof["peak"],
                                              "what": "every mfma_filter_kernel_v7 stage launch of the last timed step (seed pass excluded), same peak"}
        else:
            # a flat scan needs rows*4*d bytes ONCE per batch; the stream engine re-reads the store once
            # per group of 4 queries, which this figure deliberately does not credit.
            alg_bytes = krows * 4 * d
            roof = {"bound": "hbm", "kernel": "flat_scan_kernel", "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        roof["frac"] = (roof["achieved"] / roof["peak"]) if roof["achieved"] else None
        if args.mode == "graph" and roof["achieved"]:
            roof["gather_ceiling_measured"] = HBM_GATHER_CEILING_GBS
            roof["frac_of_gather_ceiling"] = roof["achieved"] / HBM_GATHER_CEILING_GBS
        if used_mfma and args.mode == "flat":
            # whole step against the same roof: the step's algorithmic work (2 * batch * rows * d) over ms_per_step - re-ranks, seeds, launches included
            roof["whole_step_frac"] = 2.0 * b * n * d / (elapsed / args.steps) / 1e12 / roof["peak"]
        if used_mfma and args.mode == "flat" and roof["achieved"]:
            sus = MFMA_I8_SUSTAINED_TOPS if roof.get("operand_bits") == 8 else MFMA_F16_SUSTAINED_TF
            roof["sustained_peak_measured"] = sus
            roof["frac_of_sustained"] = roof["achieved"] / sus
            roof["fp16_equivalent"] = {"what": "the same algorithmic flops against the fp16 dense MFMA peak (the r2 line's roof)", "frac": roof["achieved"] / MFMA_F16_PEAK_TF}
        if power is not None:
            roof["under_load"] = power
            if used_mfma and args.mode == "flat" and power.get("sclk_mhz_under_load") and roof.get("operand_bits") == 8:
                # the dense peak assumes the 2.4 GHz boost clock; what the matrix pipe could deliver at the clock the board sustains
                # under THIS kernel's power draw (nothing else about the kernel changed), and the step's work against it
                at_clk = MFMA_I8_PEAK_TOPS * power["sclk_mhz_under_load"] / 2400.0
                roof["under_load"]["peak_at_that_clock"] = at_clk
                roof["under_load"]["whole_step_frac_of_peak_at_that_clock"] = 2.0 * b * n * d / (power["ms_per_step_sustained"] * 1e-3) / 1e12 / at_clk
        roof["kernel_ms_per_step"] = kernel_ms * (b / kq if used_mfma and args.mode == "flat" else 1.0)   # all slices of a step
        roof["kernel_ms_per_launch"] = kernel_ms
        roof["timed_launches"] = len(main_ms)
        res = {
            "metric": "QPS @ recall@10>=0.999, 10Mx768 L2",
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": ("weak" if world > 1 else None),
            "vs_baseline": None,
            "value_device_resident": qps,   # (= value: queries generated on the device, results left there - the contract's timed region)
            "end_to_end": e2e,              # the same steps host -> host, copies overlapped; None with --no-e2e
            "dtype": ("f32 (exact fp32 distances; the batched scan runs %s as a lower-bound filter, survivors re-ranked in fp32)"
                      % ("an int8 MFMA pass (int32 accumulation) over an 8-bit mirror of the rows" if int(st.get("main_kernel_bits", 0)) == 8 else
                         "an fp16 MFMA pass (fp32 accumulation) over a half mirror of the rows" if int(st.get("main_kernel_bits", 0)) == 16 else "no matrix pass"))
                     if args.mode == "flat" else "f32",
            "data": "synthetic" if args.data == "uniform" else ("synthetic (clustered: 1000 Gaussian clusters, sigma 0.1 - SURVEY 8d secondary set)" if args.data == "clustered"
                                                                else "synthetic (manifold: 16-dimensional uniform latent embedded in %d dimensions + 1%% noise - tertiary set, not the BASELINE recipe)" % d),
            "recall_at_10": recall,
            "recall_check": {"queries": nrec, "ground_truth": "exact fp32 direct-form stream scan of all rows (EPS_FLAT_STREAM)",
                             "ground_truth_vs_torch_fp32_scan": gt_vs_torch, "torch_queries": ntorch},
            "config": {"workload": "%dM x %d L2 %s, k=%d, batch=%d per step, %d rows per GPU, %d GPU(s), rows_total=%d%s"
                                   % ((n * world) // 1_000_000, d, "exact flat scan" if args.mode == "flat" else "graph traversal T=%d L=%d" % (args.T, args.L),
                                      k, b, n, world, n * world,
                                      "" if world == 1 else (" (rows per GPU fixed = BASELINE configs[4] shape, same batch on every shard)" if args.scale == "rows"
                                                             else " (corpus hash-sharded over the GPUs, batch = %d x %d)" % (world, args.batch))),
                       "mode": args.mode, "engine": args.engine, "parallelism": "row-hash-shard x%d + one RCCL all-gather of packed top-k" % world},
            "roofline": roof,
            "stats": {"dist_evals_per_query": st["dist_evals"] / float(b), "rerank_rows_per_query": st["rerank_rows"] / float(b),
                      "expansions_per_query": st["expansions"] / float(b), "overflow_queries": st["overflow_queries"]},
            "work_rate": {"value": qps * n * world, "unit": "query*rows/s"},
        }
        if world > 1:
            # one all-gather of pack_n bytes per rank and step (12 B x k x batch), then a k-way merge on every rank
            res["exchange"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective": ("eps_exchange (rccl)" if xchg is not None else ("all_gather_into_tensor" if backend == "nccl" else "all_gather (host staged)")),
                               "library": xchg.info() if xchg is not None else None, "fallback_reason": xchg_note,
                               "bytes_per_rank_per_step": int(pack_n), "bytes_gathered_per_rank_per_step": int(pack_n * world),
                               "all_gather_us_per_step": [round(x[0], 1) for x in xchg_us], "merge_us_per_step": [round(x[1], 1) for x in xchg_us],
                               "all_gather_us_mean": float(np.mean([x[0] for x in xchg_us])) if xchg_us else None,
                               "merge_us_mean": float(np.mean([x[1] for x in xchg_us])) if xchg_us else None,
                               "ranks": ranks_info,
                               "distinct_devices": len(set((r_["device_ordinal"], r_["device_uuid"], r_["pci_bus_id"]) for r_ in ranks_info)),
                               "merged_answer_check": "recall_at_10 above = the merged top-k of the last step against the MERGED exact fp32 stream scans of every shard (and, for 16 queries, "
                                                      "against torch fp32 scans of every shard merged the same way): recall_check"}
        if build_s is not None:
            res["graph_build_s"] = build_s
        if secondary:
            res["secondary_traversal"] = secondary
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            try:
                cpu = CpuBaseline(torch, X)
                res["cpu_baseline"] = cpu_baseline(cpu, args, X, qlast, gt, graph_for_cpu, args.cpu_seconds,
                                                   gpu_ids=(got_i.cpu().numpy() if args.mode == "flat" else None))
                res["gpu_over_cpu"] = qps / res["cpu_baseline"]["value"] if res["cpu_baseline"].get("value") else None
            except Exception as e:  # the baseline is a report, never a dependency of the product path
                res["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": "failed: %r" % (e,)}
        # ---- the other single-GPU configurations of BASELINE.json, each with the reference's CPU path beside it
        want = [] if (args.configs == "none" or world > 1 or args.mode != "flat" or args.data != "uniform") else args.configs.split(",")
        if want:
            res["configs"] = {}
            if "c2" in want and args.graph_rows and args.graph_rows <= n:
                try:
                    res["configs"]["c2_1Mx768_b1_latency"] = config_c2(amd, torch, args, X, qlast, dev, stream, local_rank, cpu, ix2, graph_for_cpu)
                except Exception as e:
                    res["configs"]["c2_1Mx768_b1_latency"] = {"failed": repr(e)}
            if "c4" in want and args.metric == "EUCLIDEAN":
                try:
                    res["configs"]["c4_cosine_id_filter_b1024"] = config_c4(amd, torch, args, X, qlast, dev, stream, local_rank, cpu)
                except Exception as e:
                    res["configs"]["c4_cosine_id_filter_b1024"] = {"failed": repr(e)}
            if "c1" in want:
                try:
                    res["configs"]["c1_100kx128_bindings"] = config_c1(args)
                except Exception as e:
                    res["configs"]["c1_100kx128_bindings"] = {"failed": repr(e)}
            if "secondary" in want and args.graph_rows and args.graph_rows <= n:
                for kind in ("clustered", "manifold"):   # (last: the CPU legs overwrite the head of the host copy of the table)
                    try:
                        res["configs"]["secondary_%s_%dx%d" % (kind, args.graph_rows, d)] = config_secondary(amd, torch, args, dev, stream, local_rank, cpu, kind)
                    except Exception as e:
                        res["configs"]["secondary_%s_%dx%d" % (kind, args.graph_rows, d)] = {"failed": repr(e)}
        if cpu is not None:
            cpu.close()
        if "embedding" in want and world == 1:   # (last: it overwrites the table in place)
            try:
                ix.close()
                res["configs"]["embedding_like_%dMx%d" % (n // 1_000_000, d)] = config_embedding_like(amd, torch, args, X, dev, stream, local_rank)
            except Exception as e:
                res["configs"]["embedding_like_%dMx%d" % (n // 1_000_000, d)] = {"failed": repr(e)}
        print(json.dumps(res))
    if ix2 is not None:
        ix2.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
