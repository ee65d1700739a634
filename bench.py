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
# process.  r6: at N = 1 on the headline shape the run measures it LIVE by re-running its headline step in two child processes under rocprofv3
# (pmc_traffic below; --no-pmc skips it): `roofline.traffic` carries the bytes of the largest dispatch, `roofline.traffic_measurement` how they were
# taken.  Where that is not possible (no rocprofv3, the graph mode, N > 1) `traffic` stays null; `roofline.traffic_reference` names the committed
# profile of the same launch shape either way (file, sha256 of the file as it lies in this tree, the bytes it shows, the round's kernel it was taken on).
TRAFFIC_REF = {
    "mfma8": {"file": "profiles/r6_pmc_10Mx768_b1024.csv", "bytes_per_launch": (2 * 2752675 + 3866) * 1024.0, "algorithmic_bytes": 7280256 * 768.0,
              "launch": "mfma_filter_kernel_v7<2, FM_IDS, int8, 8>, 7,280,256 rows x 1024 queries (last of 6 stages), occurrence 5 of the PMC passes (library kernels' rows only)",
              "taken_on": "r6 library (scripts/run_flat_profile_r6.sh: PMC passes, then the line; the filter kernel itself is r4's)"},
    "mfma": {"file": "profiles/r2_pmc_10Mx768_b1024.csv", "bytes_per_launch": (2 * 7340245 + 7460) * 1024.0, "algorithmic_bytes": 9262720 * 1536.0,
             "launch": "mfma_filter_kernel_v7<2, FM_IDS> (fp16), 9,262,720 rows x 1024 queries", "taken_on": "r2 kernel"},
    "graph_T4_L500": {"file": "profiles/r6_traverse_10Mx768_pmc.csv", "bytes_per_launch": (2 * 19184815 + 1055271) * 1024.0, "algorithmic_bytes": 41.1e9,
                      "launch": "traverse2_kernel T=4 L=500 batch 1024, 10M-node device-built graph, 8-bit prefilter, edge constants, visited stamps (occurrences 0, 1)",
                      "taken_on": "r6 library (scripts/run_10m_graph_r6.sh: PMC passes, then the line; the kernel is r5's)"},
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


def pmc_traffic(args, kernel_substr, want_rows_hint):
    """HBM traffic of the dominant launch, measured LIVE: this script re-runs its headline step in two child processes under `rocprofv3 --pmc`
    (bench_legs.pmc_bytes).  A failure here never fails the bench (roofline.traffic stays null and the committed reference stands)."""
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--graph-rows", "0", "--configs", "none", "--recall-queries", "64",
             "--power-seconds", "0", "--no-e2e", "--no-pmc", "--rows", str(args.rows), "--dim", str(args.dim), "--batch", str(args.batch), "--k", str(args.k), "--metric", args.metric,
             "--engine", args.engine]
    out = pmc_bytes(child, kernel_substr)
    out["matrix_pipe"] = pmc_counters(child, kernel_substr, ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    child[child.index("--steps") + 1] = "6"   # (more launches for the trace's median)
    out["kernel_trace"] = profiler_kernel_us(child, kernel_substr + "<2, 0")
    return out


def traffic_ref(key):
    import hashlib
    r = dict(TRAFFIC_REF[key])
    path = os.path.join(ROOT, r["file"])
    r["sha256"] = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
    return r


from bench_legs import (HBM_PEAK_GBS, HBM_GATHER_CEILING_GBS, MFMA_F16_PEAK_TF, MFMA_F16_SUSTAINED_TF, MFMA_I8_PEAK_TOPS, MFMA_I8_SUSTAINED_TOPS,  # noqa: E402
                        CpuBaseline, config_c1, config_c2, config_c4, config_embedding_like, config_secondary, cpu_baseline, exact_topk_torch, gen_rows, pmc_bytes, pmc_counters, profiler_kernel_us, recall_of)


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
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", help="N = 1, flat mode, >= 5M rows: skip the two child runs under rocprofv3 --pmc that measure the dominant launch's HBM "
                                                                         "traffic live (roofline.traffic)")
    ap.add_argument("--no-e2e", dest="e2e", action="store_false", help="skip the end-to-end (host -> host, pipelined) repetition of the timed steps")
    ap.add_argument("--scale", default="rows", choices=["queries", "rows"])
    ap.add_argument("--inproc", action="store_true", help="N > 1 in ONE process: eps_index_create_sharded over devices 0..N-1 (the form the single-process "
                                                          "reference DBMS uses), per-shard lists merged on the caller's device; no torch.distributed")
    ap.add_argument("--configs", default="c1,c2,c4,secondary,embedding", help="further BASELINE configs measured into the line's `configs` object at N = 1: c1 (configs[0]: 100k x 128 through both "
                                                     "`epsilla` modules), embedding (unit-norm Gaussian rows with dominant dimensions, COSINE, on the --rows table), c2 (1M x 768, batch 1, latency), "
                                                     "c4 (COSINE + ID < N filter on the --rows table, batch --batch), secondary (SURVEY 8d clustered set + the manifold set at --graph-rows: "
                                                     "flat and graph legs); 'none' skips them")
    return ap.parse_args()


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
    # ... and for a handful of queries (<= 16 KB of lists per rank: --batch up to 136 at k = 10) its b = 1 form: no collective at all - peer stores into
    # hipIpc-mapped mailboxes + flags (eps_exchange_direct_merge).  Works between ranks that share a device too, so the one-GPU gloo runs use it.
    mbox = None
    b_step = args.batch * (world if args.scale == "queries" else 1)
    if world > 1 and b_step * args.k * 12 <= 16384 and os.environ.get("EPS_BENCH_EXCHANGE", "eps") in ("eps", "mailbox"):
        ok = 1
        try:
            mbox = amd.Exchange.direct(rank, world, device=local_rank)
            handles = [None] * world
            dist.all_gather_object(handles, mbox.mailbox_export())
            mbox.mailbox_connect(handles)
        except Exception as e:  # noqa: BLE001
            ok, xchg_note = 0, "eps_exchange mailbox setup failed on rank %d: %r" % (rank, e)
        okt = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:   # (all ranks or none)
            if mbox is not None and ok:
                mbox.close()
            mbox = None
            xchg_note = xchg_note or "eps_exchange mailbox setup failed on another rank"
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
            if mbox is not None:   # b = 1 form: peer stores + flag, device-side wait, merge - no collective
                mbox.direct_merge(idss[slot], dds[slot], m_is[slot], m_ds[slot], stream=stream)
                if timed:
                    ev[1].record()
            elif xchg is not None:   # one library call: ncclAllGather of the packed lists + the merge (its own hipEvents split the two)
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
    if mbox is not None:   # (the library's own event triples of the timed steps: push + wait | merge)
        xchg_us = mbox.times_us(64)[-args.steps:]
    elif xchg is not None:   # (all-gather | merge)
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
        if used_mfma and args.mode == "flat" and world == 1 and args.pmc and n >= 5_000_000 and not args.inproc:
            # r6: the HBM traffic of the dominant launch measured by this very run (two child processes under rocprofv3 --pmc), per launch like `achieved`
            live = pmc_traffic(args, "mfma_filter_kernel_v7", krows)
            roof["traffic_measurement"] = live
            if "bytes" in live:
                roof["traffic"] = live["bytes"]
                roof["traffic_over_algorithmic"] = live["bytes"] / roof["algorithmic_bytes"] if roof.get("algorithmic_bytes") else None
            mp = (live.get("matrix_pipe") or {}).get("values") or {}
            if mp.get("GRBM_GUI_ACTIVE") and mp.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                # busy cycles summed over the chip's SIMDs against the active cycles summed over its XCDs: the fraction of a clock the matrix pipe of a SIMD works
                prop = torch.cuda.get_device_properties(local_rank)
                simds, xcds = 4 * int(prop.multi_processor_count), 8
                roof["matrix_pipe_busy"] = (mp["SQ_VALU_MFMA_BUSY_CYCLES"] / simds) / (mp["GRBM_GUI_ACTIVE"] / xcds)
            kt = live.get("kernel_trace") or {}
            if "median_us" in kt and kernel_ms:   # (the profiler's view of the same launch: serialised dispatches, a few per cent above the hipEvent time of the timed region)
                roof["rocprofv3_kernel_ms"] = kt["median_us"] / 1e3
                roof["rocprofv3_over_hipevents"] = kt["median_us"] / 1e3 / kernel_ms
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
            res["exchange"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective": ("eps_exchange mailbox (peer stores + flag, no collective)" if mbox is not None else "eps_exchange (rccl)" if xchg is not None else ("all_gather_into_tensor" if backend == "nccl" else "all_gather (host staged)")),
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
