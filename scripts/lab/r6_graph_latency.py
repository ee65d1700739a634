"""r6: where a single-query traversal call's time goes and where its p99 comes from.  1M x 768 uniform, device-built NSG, T = 4 / 1,
L = 500, one query per call, 200 calls: sorted latencies, the walk's counters on the slowest calls, the kernel's phase profile of one call."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")
import vectordb_amd as amd
n, d = int(os.environ.get("ROWS", 1_000_000)), 768
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19)); X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
Q = torch.rand((256, d), generator=torch.Generator(device="cuda").manual_seed(43), device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream(); ix.attach_rows(X)
t0 = time.perf_counter(); ix.build(); ix.synchronize(); print("build s", time.perf_counter() - t0, flush=True)
o = (torch.empty((1, 10), dtype=torch.int64, device="cuda"), torch.empty((1, 10), device="cuda"), torch.empty((1,), dtype=torch.int32, device="cuda"))
for T in (4, 1):
    kw = dict(mode=amd.MODE_GRAPH, intra_threads=T, master_queue=500, local_queue=500)
    for i in range(3):
        ix.search(Q[i:i + 1], 10, out=o, **kw); torch.cuda.synchronize()
    lat, st = [], []
    for i in range(200):
        t = time.perf_counter(); ix.search(Q[i:i + 1], 10, out=o, **kw); torch.cuda.synchronize(); lat.append(time.perf_counter() - t)
        s_ = ix.stats(); st.append((s_["dist_evals"], s_["expansions"], s_["rerank_rows"], s_.get("main_kernel_ms", 0)))
    lat = np.array(lat) * 1e3
    order = np.argsort(lat)
    print(json.dumps({"T": T, "p50": float(np.median(lat)), "p90": float(np.percentile(lat, 90)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max()), "min": float(lat.min())}))
    print(" slowest:", [(int(i), round(float(lat[i]), 3), st[i]) for i in order[-6:]])
    print(" fastest:", [(int(i), round(float(lat[i]), 3), st[i]) for i in order[:3]])
    km = ix.kernel_times(64)
    print(" kernel ms (last 64 calls): median %.3f max %.3f" % (float(np.median(km)), float(np.max(km))), flush=True)
    os.environ["EPS_TRV_PROF"] = "1"
    ix.search(Q[5:6], 10, out=o, **kw); torch.cuda.synchronize()
    del os.environ["EPS_TRV_PROF"]
