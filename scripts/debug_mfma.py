import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd
def run(n, d, nq, k=10):
    g = torch.Generator(device="cuda").manual_seed(42)
    X = torch.empty((n, d), device="cuda")
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20)); X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
    Q = torch.rand((nq, d), generator=g, device="cuda")
    ix = amd.GpuIndex(d, 0); ix.set_stream(torch.cuda.current_stream().cuda_stream); ix.attach_rows(X)
    outs = []
    for eng in (amd.FLAT_MFMA, amd.FLAT_STREAM):
        ids = torch.empty((nq, k), dtype=torch.int64, device="cuda"); dist = torch.empty((nq, k), device="cuda"); cnt = torch.empty((nq,), dtype=torch.int32, device="cuda")
        ix.search(Q, k, out=(ids, dist, cnt), mode=amd.MODE_FLAT, flat_engine=eng); torch.cuda.synchronize()
        outs.append((ids.cpu().numpy(), dist.cpu().numpy(), ix.stats()))
    a, b = outs
    miss = [sorted(set(b[0][q]) - set(a[0][q])) for q in range(nq)]
    nm = sum(len(m) for m in miss)
    allm = np.array([i for m in miss for i in m])
    print("n=%d d=%d nq=%d: missing %d of %d; rerank/query=%.1f overflow=%d" % (n, d, nq, nm, nq * k, a[2]["rerank_rows"] / nq, a[2]["overflow_queries"]))
    if nm:
        print("  missing ids: min %d max %d; histogram by 1M:" % (allm.min(), allm.max()), np.bincount(allm >> 20))
        print("  tile (id>>7) mod 8 histogram:", np.bincount((allm >> 7) & 7, minlength=8))
        qs = [q for q in range(nq) if miss[q]]
        print("  queries with misses: %d, query-tile histogram:" % len(qs), np.bincount(np.array(qs) >> 7))
        print("  within-tile row (id & 127) hist/32:", np.bincount((allm & 127) >> 5, minlength=4), " query lane-block (q&127)>>5:", np.bincount((np.array([q for q in range(nq) for _ in miss[q]]) & 127) >> 5, minlength=4))
    ix.close(); del X
for n, d, nq in [(2_000_000, 768, 1024), (10_000_000, 768, 1024)]:
    run(n, d, nq)
