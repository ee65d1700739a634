#!/usr/bin/env python
"""r5 (late): p50 latency of small calls at rows x dim (default 1M x 768) on the one-pass form against the staged chain, in ONE process on ONE table:
k = 10 / 32 / 64 (EPS_S8_MAX_K=16 sends k > 16 to the chain, as until r4) and a compiled filter program (EPS_S8_FILTER_PROGRAMS=0 sends it to the
chain), 1 / 4 / 8 / 16 queries per call; every answer compared with the stream engine's.   python one_pass_wide_k_and_programs.py [rows dim]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vectordb_amd as amd  # noqa: E402

n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 768)
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.empty((n, d), device="cuda")
for s in range(0, n, 1 << 19):
    e = min(n, s + (1 << 19))
    X[s:e] = torch.rand((e - s, d), generator=g, device="cuda")
Q = torch.rand((4096, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
attr = torch.arange(n, dtype=torch.int32, device="cuda").view(torch.uint8).reshape(n, 4)


def p50(nq, k, reps=40):
    o = (torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
    for _ in range(3):
        ix.search(Q[:nq], k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    torch.cuda.synchronize()
    lat, one = [], 0
    for i in range(reps):
        q = Q[(i * nq) % 2048:(i * nq) % 2048 + nq]
        t0 = time.perf_counter()
        ix.search(q, k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
        one += ix.stats()["one_pass"]
    r = (torch.empty_like(o[0]), torch.empty_like(o[1]), torch.empty_like(o[2]))
    ix.search(q, k, out=r, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    torch.cuda.synchronize()
    assert torch.equal(o[0], r[0]) and torch.equal(o[1], r[1]), (nq, k)
    return 1e3 * float(np.median(lat)), one, int(ix.stats()["rerank_rows"])


print("rows %d dim %d; p50 ms per call (calls on the one-pass form of %d, rows re-ranked by the last call)" % (n, d, 40))
for what, prog in (("no filter", None), ("filter program  id %% 3 = 1", [("i32", 0), ("const", 3), ("%",), ("const", 1), ("=",)])):
    ix.set_filter_program(prog, attr if prog else None, stride=4)
    for k in ((10, 32, 64) if prog is None else (10, 64)):
        for nq in (1, 4, 8, 16):
            line = []
            for name, sw in (("one pass", {}), ("chain", {"EPS_S8_MAX_K": "16"} if prog is None else {"EPS_S8_FILTER_PROGRAMS": "0"})):
                if name == "chain" and prog is None and k <= 16:
                    continue
                for a, b in sw.items():
                    amd.set_tuning(a, b)
                ms, one, rr = p50(nq, k)
                for a in sw:
                    amd.set_tuning(a, None)
                line.append("%s %.3f (%d, %d)" % (name, ms, one, rr))
            print("%-28s k %2d  queries %2d:  %s" % (what, k, nq, "   ".join(line)), flush=True)
ix.set_filter_program(None)
print("the call's tail: s8_rerank_kernel (default) against rerank_kernel with the selection prologue (EPS_S8_RERANK=0), p50 ms per call")
for k in (10, 64):
    for nq in (1, 2, 4, 8, 16):
        line = []
        for name, v in (("s8_rerank", None), ("rerank", "0"), ("s8_rerank", None), ("rerank", "0")):
            amd.set_tuning("EPS_S8_RERANK", v)
            ms, one, rr = p50(nq, k)
            line.append("%s %.3f" % (name, ms))
        amd.set_tuning("EPS_S8_RERANK", None)
        print("k %2d  queries %2d:  %s" % (k, nq, "   ".join(line)), flush=True)
ix.close()
