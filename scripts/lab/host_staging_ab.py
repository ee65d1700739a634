"""r5: the C ABI's host-pointer path (queries and results in pageable host memory) with and without the library's page-locked staging (EPS_HOST_STAGING),
next to the device-resident call: flat scan, batch 1024, and a single vector per call.   python scripts/lab/host_staging_ab.py [rows=10000000]"""
import os
import sys
import time

os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vectordb_amd as amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d, k = 768, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
X = torch.empty((n, d), device=dev)
for s in range(0, n, 1 << 20):
    e = min(n, s + (1 << 20))
    X[s:e] = torch.rand((e - s, d), generator=g, device=dev)
ix = amd.GpuIndex(d, 0, device=0).use_torch_stream()
ix.attach_rows(X)
for b, steps in ((1024, 12), (1, 100)):
    Qd = [torch.rand((b, d), generator=g, device=dev) for _ in range(steps)]
    Qh = [q.cpu().numpy() for q in Qd]
    out = (torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), device=dev), torch.empty((b,), dtype=torch.int32, device=dev))
    for i in range(3):
        ix.search(Qd[i], k, out=out, mode=amd.MODE_FLAT)
    torch.cuda.synchronize()
    lat = []
    for i in range(steps):
        t0 = time.perf_counter()
        ix.search(Qd[i], k, out=out, mode=amd.MODE_FLAT)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    res = float(np.median(lat))
    ref = out[0].cpu().numpy().copy()
    line = "batch %4d: device-resident %.3f ms" % (b, 1e3 * res)
    for rep in range(2):
        for staging in ("0", "1"):
            os.environ["EPS_HOST_STAGING"] = staging
            for i in range(3):
                ix.search(Qh[i], k, mode=amd.MODE_FLAT)
            lat = []
            for i in range(steps):
                t0 = time.perf_counter()
                ids, dist, cnt = ix.search(Qh[i], k, mode=amd.MODE_FLAT)
                lat.append(time.perf_counter() - t0)
            assert (ids == ref).all()
            line += " | staging %s: %.3f ms (%.3f of resident)" % (staging, 1e3 * float(np.median(lat)), res / float(np.median(lat)))
    print(line, flush=True)
