"""The legs bench.py's line carries BESIDE the headline step (r6: split out of bench.py, which keeps the contract - arguments, ranks, the timed
region, the roofline and the JSON line): synthetic row generators, the recall helpers, the CPU baseline (the reference's own engine compiled from its
sources, oracle/_ref - or the scalar restatement, labelled "port") and one function per further BASELINE config / labelled secondary set:

  config_c1               BASELINE configs[0]  100k x 128 through engine/bindings - both `epsilla` modules, each in its own process
  config_c2               BASELINE configs[1]  1M x 768 L2, one query per call: exact engines + the traversal, the reference beside them
  config_c4               BASELINE configs[3]  10M x 768 COSINE + `ID < N`, batch 1024
  config_embedding_like   unit-norm Gaussian rows with 8 dominant columns, COSINE (the shape learned embeddings have and U[0,1) has not)
  config_secondary        clustered / manifold sets: where the flat scan and where the traversal wins (NOT the BASELINE recipe)

Nothing here is product code: the product is vectordb_amd/ (csrc -> libepsilla_gfx950.so) and dropin/.  Only `cpu_baseline` / `CpuBaseline` touch oracle/."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_GATHER_CEILING_GBS = 6050.0   # measured here: random whole-row gathers (768 B and 3 KB rows, 30 GB table, 2-8 rows in flight per lane group,
                                  # 4-8 wavefronts per SIMD) with nothing else in the kernel: 5.99-6.10 TB/s (scripts/lab/gather_peak.hip, profiles/r4_gather_peak.txt)
MFMA_F16_PEAK_TF = 2500.0  # dense bf16/f16 MFMA peak (nominal, 2.4 GHz)
MFMA_F16_SUSTAINED_TF = 1814.0  # measured: v_mfma_f32_32x32x16_f16 alone, operands toggling like data, 1.82 GHz (scripts/lab/mfma_peak.hip)
MFMA_I8_PEAK_TOPS = 5000.0      # dense 8-bit MFMA peak: twice the fp16 rate (MI355X_MICROARCH.md lists the FP8 dense peak ~5 P and I8 at ~2x bf16)
MFMA_I8_SUSTAINED_TOPS = 3424.0  # measured: v_mfma_i32_32x32x32_i8 alone on bytes in [-127, 127], 1.74 GHz (profiles/r3_mfma_peak_i8_vs_fp16.txt)


def pmc_bytes(child_argv, kernel_substr):
    """HBM bytes of ONE launch measured live: `child_argv` (a python command that runs the launch a few times) is run twice under `rocprofv3 --pmc` - FETCH_SIZE,
    then WRITE_SIZE: separate passes, as MI355X_MICROARCH.md prescribes - and the LONGEST dispatch whose kernel name contains `kernel_substr` is read out of the
    profiler's database.  gfx950: FETCH_SIZE counts 128-byte requests at 64 -> x 2; both counters are in KiB.  Returns {"bytes": ..} or {"failed": reason}."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {"failed": "rocprofv3 not found"}
    out = {"how": "two child runs under rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE); the longest dispatch of the kernel; FETCH_SIZE x 2 (gfx950), KiB"}
    vals = {}
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="eps_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k_, None)
        try:
            r = subprocess.run([exe, "--pmc", counter, "-d", td, "-o", "pmc", "--"] + list(child_argv), cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            dbs = glob.glob(os.path.join(td, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return {"failed": "%s pass: rc %d, %d database(s): %s" % (counter, r.returncode, len(dbs), (r.stderr or r.stdout)[-300:])}
            c = sqlite3.connect(dbs[0])
            best = None
            for did, kn, val, dur in c.execute("select dispatch_id, kernel_name, sum(value), max(duration) from counters_collection where counter_name = ? group by dispatch_id", (counter,)):
                if kernel_substr in kn and (best is None or dur > best[1]):
                    best = (val, dur, kn)
            c.close()
            if best is None:
                return {"failed": "%s pass: no dispatch of %s in the profile" % (counter, kernel_substr)}
            vals[counter] = best
        except Exception as e:  # noqa: BLE001
            return {"failed": "%s pass: %r" % (counter, e)}
        finally:
            shutil.rmtree(td, ignore_errors=True)
    out["FETCH_SIZE_KiB"], out["WRITE_SIZE_KiB"] = float(vals["FETCH_SIZE"][0]), float(vals["WRITE_SIZE"][0])
    out["bytes"] = (2.0 * out["FETCH_SIZE_KiB"] + out["WRITE_SIZE_KiB"]) * 1024.0
    out["dispatch_us_under_the_profiler"] = float(vals["FETCH_SIZE"][1]) / 1e3
    out["kernel"] = vals["FETCH_SIZE"][2][:120]
    out["seconds"] = time.time() - t0
    return out


def pmc_counters(child_argv, kernel_substr, counters):
    """`counters` (one rocprofv3 --pmc pass) of the LONGEST dispatch whose kernel name contains `kernel_substr`: {"values": {name: sum over the device}} or {"failed": ..}"""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {"failed": "rocprofv3 not found"}
    td = tempfile.mkdtemp(prefix="eps_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    try:
        r = subprocess.run([exe, "--pmc"] + list(counters) + ["-d", td, "-o", "pmc", "--"] + list(child_argv), cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
        dbs = glob.glob(os.path.join(td, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return {"failed": "rc %d, %d database(s): %s" % (r.returncode, len(dbs), (r.stderr or r.stdout)[-300:])}
        c = sqlite3.connect(dbs[0])
        per = {}
        for did, kn, cn, val, dur in c.execute("select dispatch_id, kernel_name, counter_name, sum(value), max(duration) from counters_collection group by dispatch_id, counter_name"):
            if kernel_substr in kn:
                per.setdefault(did, {"_dur": dur})[cn] = val
        c.close()
        if not per:
            return {"failed": "no dispatch of %s in the profile" % kernel_substr}
        best = max(per.values(), key=lambda v: v["_dur"])
        return {"values": {k_: float(v) for k_, v in best.items() if k_ != "_dur"}, "dispatch_us_under_the_profiler": float(best["_dur"]) / 1e3}
    except Exception as e:  # noqa: BLE001
        return {"failed": repr(e)}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def profiler_kernel_us(child_argv, kernel_substr, top_fraction=0.85):
    """The dominant kernel's launch duration as `rocprofv3 --kernel-trace --stats` sees it, live: one child run; the median duration of the dispatches of the kernel
    within `top_fraction` of its longest one (= the largest stage's launches) - what the hipEvent figure of the roofline must agree with."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {"failed": "rocprofv3 not found"}
    td = tempfile.mkdtemp(prefix="eps_kt_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    t0 = time.time()
    try:
        r = subprocess.run([exe, "--kernel-trace", "--stats", "-d", td, "-o", "kt", "--"] + list(child_argv), cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
        dbs = glob.glob(os.path.join(td, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return {"failed": "rc %d, %d database(s): %s" % (r.returncode, len(dbs), (r.stderr or r.stdout)[-300:])}
        c = sqlite3.connect(dbs[0])
        dur = sorted(float(d) / 1e3 for (d,) in c.execute("select duration from kernels where name like ?", ("%" + kernel_substr + "%",)))
        c.close()
        if not dur:
            return {"failed": "no dispatch of %s in the trace" % kernel_substr}
        big = [d for d in dur if d >= top_fraction * dur[-1]]
        return {"how": "one child run under rocprofv3 --kernel-trace --stats; median of the kernel's dispatches within %.0f %% of its longest" % (100 * top_fraction),
                "median_us": float(np.median(big)), "dispatches": len(big), "of": len(dur), "seconds": time.time() - t0}
    except Exception as e:  # noqa: BLE001
        return {"failed": repr(e)}
    finally:
        shutil.rmtree(td, ignore_errors=True)


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
                                          "note": "one pass over the 8-bit mirror (%d bytes per row) + 4 bytes of start value per row; timed run p50 %.3f ms" % (row_bytes - 4, timed["p50_ms"]),
                                          "traffic": None}
            if getattr(args, "pmc", False) and n1 == 1_000_000 and d == 768:
                # r6: the pass's HBM bytes measured live (40 single-query calls of scripts/prof_single_query.py under rocprofv3 --pmc, the longest stream8 dispatch)
                live = pmc_bytes([sys.executable, os.path.join(ROOT, "scripts", "prof_single_query.py"), str(n1), str(d)], "stream8_kernel")
                gpu["mfma_i8"]["roofline"]["traffic_measurement"] = live
                if "bytes" in live:
                    gpu["mfma_i8"]["roofline"]["traffic"] = live["bytes"]
                    gpu["mfma_i8"]["roofline"]["traffic_over_algorithmic"] = live["bytes"] / float(n1 * row_bytes)
    gpu["auto"], ra = latency(ix, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    gpu["auto"]["recall_at_10"] = recall_of(ra, gt1)
    if graph_index is not None:
        gpu["graph_T4_L500"], rg = latency(graph_index, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=500, local_queue=500)
        gpu["graph_T4_L500"]["recall_at_10"] = recall_of(rg, gt1)

    # r5 (late): the one-pass form beyond configs[1]'s own shape - k = 64 (128 table slots per query) and a compiled filter program (evaluated once per
    # row into a bitset by one launch in front of the pass) - one query per call, each answer compared with the fp32 stream engine's under the same setting
    def probe(kk, calls=100, check=10, per_call=1):
        oo = (torch.empty((per_call, kk), dtype=torch.int64, device=dev), torch.empty((per_call, kk), dtype=torch.float32, device=dev), torch.empty((per_call,), dtype=torch.int32, device=dev))
        rr = (torch.empty_like(oo[0]), torch.empty_like(oo[1]), torch.empty_like(oo[2]))
        span = max(1, qlast.shape[0] - per_call)
        for i in range(3):
            ix.search(qlast[i:i + per_call], kk, out=oo, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        torch.cuda.synchronize()
        lat, one, same = [], 0, 0
        for i in range(calls):
            qs = qlast[i % span:i % span + per_call]
            t0 = time.perf_counter()
            ix.search(qs, kk, out=oo, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            one += int(ix.stats().get("one_pass", 0))
            if i < check:
                ix.search(qs, kk, out=rr, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
                torch.cuda.synchronize()
                same += int(torch.equal(oo[0], rr[0]) and torch.equal(oo[1], rr[1]))
        return {"k": kk, "queries_per_call": per_call, "p50_ms": 1e3 * float(np.median(lat)), "p99_ms": 1e3 * float(np.percentile(lat, 99)), "queries": calls, "one_pass_calls": one,
                "answers_equal_to_the_stream_engine": "%d of %d" % (same, check)}
    try:
        wide = {"what": "one query per call on the one-pass form beyond k <= 16 / no filter program (r5); p50 issue -> sync, inputs in HBM"}
        wide["k64"] = probe(64)
        for per_call in (8, 16, 32):   # (r5: up to 16 queries per call on the matrix cores; r6: 17..32 on two column blocks)
            wide["%d_queries_per_call_k%d" % (per_call, k)] = probe(k, calls=60, check=6, per_call=per_call)
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
    # r6: ONE query per call on the same table (the one-pass search under folded margins: the pass is 7.7 GB of mirror at this size)
    single = None
    try:
        o1 = (torch.empty((1, k), dtype=torch.int64, device=dev), torch.empty((1, k), dtype=torch.float32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))
        lat, one = [], 0
        for i in range(24):
            t0 = time.perf_counter()
            ix.search(Q[i:i + 1], k, out=o1, mode=amd.MODE_FLAT)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            one += int(ix.stats().get("one_pass", 0))
        single = {"p50_ms": 1e3 * float(np.median(lat[4:])), "calls": 20, "one_pass_calls_of_24": one, "last_answer_equals_the_batch": bool(torch.equal(o1[0][0].cpu(), torch.from_numpy(got[23])))}
    except Exception as e:  # noqa: BLE001
        single = {"failed": repr(e)}
    out = {"workload": "%dM x %d COSINE, unit-norm rows with Gaussian coordinates, 8 dominant dimensions (4 x scale), k=%d, batch=%d, exact flat scan (the library's engine choice)"
                       % (n // 1_000_000, d, k, b),
           "qps": b / sec, "ms_per_step": 1e3 * sec, "recall_at_10": recall_of(got[:nrec], g2[0].cpu().numpy()), "recall_check": "%d queries vs the fp32 stream scan" % nrec,
           "operand_bits": int(st.get("main_kernel_bits", 0)), "rerank_rows_per_query": st["rerank_rows"] / float(b), "overflow_queries": st["overflow_queries"],
           "main_kernel_ms": float(np.median(km)) if km else None, "one_query_per_call": single}
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
