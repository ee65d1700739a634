import os, sys, time
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import vectordb_amd as amd
n, d, b, k = 1_000_000, 768, 1024, 10
dev = torch.device("cuda", 0)
X = torch.rand((n, d), generator=torch.Generator(device=dev).manual_seed(42), device=dev)
ix = amd.GpuIndex(d, 0, device=0); ix.set_stream(torch.cuda.current_stream().cuda_stream); ix.attach_rows(X); ix.build(n); ix.synchronize()
nst = 12  # (see bench.py for the three-slot form that came out of this)
queries = [torch.rand((b, d), generator=torch.Generator(device=dev).manual_seed(50 + i), device=dev) for i in range(nst)]
packs = [(torch.empty((b, k), dtype=torch.int64, device=dev), torch.empty((b, k), dtype=torch.float32, device=dev)) for _ in range(2)]
cnt = torch.empty((b,), dtype=torch.int32, device=dev)
for mode, kw in (("flat", dict(mode=amd.MODE_FLAT)), ("graph", dict(mode=amd.MODE_GRAPH, intra_threads=4, master_queue=500, local_queue=500))):
    def step(q, slot): ix.search(q, k, out=(packs[slot][0], packs[slot][1], cnt), **kw); return packs[slot][1], packs[slot][0]
    for i in range(3): step(queries[i], 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(nst): step(queries[s], 0)
    torch.cuda.synchronize(); res = (time.perf_counter() - t0) / nst
    for variant in ("full", "no_d2h", "no_h2d", "same_stream"):
        cs = torch.cuda.Stream(device=dev) if variant != "same_stream" else torch.cuda.current_stream()
        main_s = torch.cuda.current_stream()
        qh = [torch.empty((b, d), dtype=torch.float32).pin_memory() for _ in range(nst)]
        for s in range(nst): qh[s].copy_(queries[s])
        rh_i = [torch.empty((b, k), dtype=torch.int64).pin_memory() for _ in range(nst)]
        rh_d = [torch.empty((b, k), dtype=torch.float32).pin_memory() for _ in range(nst)]
        dq = [torch.empty((b, d), dtype=torch.float32, device=dev) for _ in range(2)]
        ev_up = [torch.cuda.Event() for _ in range(2)]; ev_done = [torch.cuda.Event() for _ in range(2)]; ev_down = [torch.cuda.Event() for _ in range(2)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(cs):
            dq[0].copy_(qh[0], non_blocking=True); ev_up[0].record(cs)
        tsearch = 0.0
        for s in range(nst):
            cur = s & 1
            if s + 1 < nst and variant != "no_h2d":
                with torch.cuda.stream(cs):
                    cs.wait_event(ev_done[1 - cur]); dq[1 - cur].copy_(qh[s + 1], non_blocking=True); ev_up[1 - cur].record(cs)
            main_s.wait_event(ev_up[cur]); main_s.wait_event(ev_down[cur])
            t1 = time.perf_counter()
            o_d, o_i = step(dq[cur] if variant != "no_h2d" else queries[s], cur)
            tsearch += time.perf_counter() - t1
            ev_done[cur].record(main_s)
            if variant != "no_d2h":
                with torch.cuda.stream(cs):
                    cs.wait_event(ev_done[cur]); rh_i[s].copy_(o_i, non_blocking=True); rh_d[s].copy_(o_d, non_blocking=True); ev_down[cur].record(cs)
        cs.synchronize(); torch.cuda.synchronize(); el = (time.perf_counter() - t0) / nst
        print(mode, variant, "resident %.3f ms  e2e %.3f ms  in-search %.3f ms" % (1e3 * res, 1e3 * el, 1e3 * tsearch / nst), flush=True)
