"""Parity at BASELINE.json's full sizes (configs[2] and configs[3]: 10M x 768, k = 10, batch 1024, one MI355X) against the
COMPILED REFERENCE (oracle/_ref = the reference's own sources): position-wise ids, distances to 1e-4 - not set recall.

  configs[2]  L2: eps_index_search (EPS_FLAT_AUTO -> int8 matrix filter + fp32 re-rank) vs the reference's BruteForceSearch
              (engine/db/execution/vec_search_executor.cpp:717-768) on 32 of the 1024 queries, EVERY returned distance (1024 x 10) against the
              reference's own distance function on that (row, query) pair (r6), the fp32 stream engine on 64, and size-independent
              properties on all 1024 (sorted by (dist, id), unique ids, k results);
  configs[3]  COSINE on rows normalised as at insert (db/table_segment_mvp.cpp:574-587) + `ID < N` at 10 / 50 / 90 % selectivity
              (Config::PreFilter semantics) vs the reference's PreFilterBruteForceSearch (:770-831) driven by the reference's own
              filter parser and ExprEvaluator, 2-3 queries per selectivity; every returned distance against the reference's function;
  embedding-like (r6)  the same 10M rows overwritten with unit-norm Gaussian rows, 8 dominant columns (bench.py's `embedding_like` leg), COSINE:
              the table the library moves to the ROTATED 8-bit frame; vs the reference's BruteForceSearch on 16 queries + all distances.

The 10M-row table is generated on the device (seeded), copied once to page-aligned host memory for the reference (30.7 GB).
Needs ~31 GB of host memory and ~45 GB of HBM; takes about a minute, most of it the reference's scans."""

import os

import numpy as np
import pytest

from helpers import assert_topk_match

pytestmark = pytest.mark.gpu
N, D, B, K = 10_000_000, 768, 1024, 10


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd
    from vectordb_amd.build import build
    build()
    return vectordb_amd


@pytest.fixture(scope="module")
def table(ref):
    import torch
    if torch.cuda.mem_get_info(0)[0] < 60 << 30:
        pytest.skip("needs ~45 GB of free HBM")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(42)
    X = torch.empty((N, D), dtype=torch.float32, device=dev)
    for s in range(0, N, 1 << 19):
        e = min(N, s + (1 << 19))
        X[s:e] = torch.rand((e - s, D), generator=g, device=dev)
    Q = torch.rand((B, D), generator=torch.Generator(device=dev).manual_seed(43), device=dev)
    threads = int(ref.L.ref_omp_max_threads())
    arr, ptr = ref.alloc_rows(N, D, threads)
    state = {"X": X, "Q": Q, "arr": arr, "ptr": ptr, "threads": threads, "torch": torch, "dev": dev}
    yield state
    ref.free_rows(ptr)


def _to_host(t, tag):
    """the reference's copy of the rows as they are NOW on the device (tag: which form - the COSINE test normalises them in place)"""
    if t.get("host") == tag:
        return
    X, arr = t["X"], t["arr"]
    for s in range(0, N, 1 << 19):
        e = min(N, s + (1 << 19))
        arr[s:e] = X[s:e].cpu().numpy()
    t["host"] = tag


def _all_distances_are_the_references(ref, t, ids, dd, metric, what):
    """every (row, query) pair the batch returned, through the reference's own distance function (GetDistFunc's choice for the
    metric: L2Sqr / InnerProduct / CosineDistance, db/index/space_*.hpp) on the host copy of the rows - 1e-4 relative, as assert_topk_match"""
    Qh = t["Q"][:ids.shape[0]].cpu().numpy()
    arr = t["arr"]
    worst = 0.0
    for q in range(ids.shape[0]):
        for j in range(ids.shape[1]):
            want = ref.dist(metric, arr[int(ids[q, j])], Qh[q])
            err = abs(float(dd[q, j]) - want) / max(abs(want), 1e-6)
            worst = max(worst, err)
            assert err <= 1e-4, (what, q, j, int(ids[q, j]), float(dd[q, j]), want)
    return worst


def _search(amd, t, ix, n_queries, **kw):
    torch = t["torch"]
    ids = torch.empty((n_queries, K), dtype=torch.int64, device=t["dev"])
    dd = torch.empty((n_queries, K), dtype=torch.float32, device=t["dev"])
    cnt = torch.empty((n_queries,), dtype=torch.int32, device=t["dev"])
    ix.search(t["Q"][:n_queries], K, out=(ids, dd, cnt), **kw)
    ix.synchronize()
    return ids.cpu().numpy(), dd.cpu().numpy(), cnt.cpu().numpy(), ix.stats()


def test_configs2_10M_x_768_L2_batch_1024_matches_the_reference_bruteforce(amd, ref, table):
    t = table
    _to_host(t, "l2")
    ix = amd.GpuIndex(D, "EUCLIDEAN", device=0).use_torch_stream()
    ix.attach_rows(t["X"])
    ids, dd, cnt, st = _search(amd, t, ix, B, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    assert st["main_kernel_bits"] == 8 and st["overflow_queries"] == 0 and st["rerank_rows"] > 0, st   # the headline path itself
    # the reference, same rows, same queries
    nref = 32
    rid, rd, sec = ref.bruteforce_many(t["ptr"], N, D, t["Q"][:nref].cpu().numpy(), K, metric=0, threads=t["threads"])
    for q in range(nref):
        assert_topk_match(ids[q], dd[q], rid[q], rd[q], what="configs[2] q%d" % q)
    _all_distances_are_the_references(ref, t, ids, dd, 0, "configs[2]")
    # the fp32 stream engine (the library's other exact path), bit for bit on 64 queries
    sid, sd, scnt, _ = _search(amd, t, ix, 64, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert np.array_equal(ids[:64], sid) and np.array_equal(dd[:64], sd)
    # size-independent properties on the whole batch
    assert (cnt == K).all() and (ids >= 0).all() and (ids < N).all()
    assert (np.diff(dd, axis=1) >= 0).all()
    assert all(len(set(r.tolist())) == K for r in ids)
    tie = np.diff(dd, axis=1) == 0
    assert (np.diff(ids, axis=1)[tie] > 0).all()          # equal distances are ordered by id (Candidate::operator<)
    # r4: the same table asked ONE vector at a time (what TableMVP::Search does; the one-pass search of stream8_kernel.hpp at full size, or
    # the staged chain where it hands over): the batch's answers, bit for bit, hence the reference's
    for q in range(3):
        t1 = {**t, "Q": t["Q"][q:q + 1]}
        i1, d1, c1, _ = _search(amd, t1, ix, 1, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
        assert np.array_equal(i1[0], ids[q]) and np.array_equal(d1[0], dd[q]) and c1[0] == K, q
    ix.close()


def test_configs2_10M_x_768_graph_path_matches_the_reference_searchimpl(amd, ref, oracle, table, tmp_path):
    """configs[2] on the path the north_star NAMES: the NSG of all 10M rows built on the device (~2 min), written in
    the reference's ann_graph file format, loaded by the reference's own ANNGraphSegment file constructor, walked by the reference's
    SearchImpl (vec_search_executor.cpp:518-715; IntraQueryThreads = 1 - the deterministic form - SearchQueueSize = 500) next to the
    device traversal of the same graph on the same queries: ids position by position, distances to 1e-4, distance evaluations within
    0.5 %.  (Recall at this queue size on uniform rows is ~0.12 on BOTH sides: the flat scan above is the path that meets the
    metric's recall bar, this test pins the traversal itself at full size.)"""
    t = table
    torch = t["torch"]
    _to_host(t, "l2")
    nq, L = 12, 500
    ix = amd.GpuIndex(D, "EUCLIDEAN", device=0).use_torch_stream()
    ix.attach_rows(t["X"])
    ix.build(N)
    ix.synchronize()
    gn, ge, nav = ix.graph_info()
    assert gn == N and ge > 30 * N
    os.makedirs(str(tmp_path / "7"), exist_ok=True)
    ix.save_graph(str(tmp_path / "7" / "ann_graph_1.bin"))
    gref = ref.L.ref_graph_load(str(tmp_path).encode(), 7, 1)
    assert gref, "the reference could not load the device-written 10M-node graph file"
    assert ref.L.ref_graph_n(gref) == N and ref.L.ref_graph_nav(gref) == nav
    Qh = t["Q"][:nq].cpu().numpy()
    ex = ref.executor(gref, t["arr"], T=1, L=L, count=True)
    ref.L.ref_dist_calls_reset()
    rid, rd, _ = ref.search_many(ex, Qh, K)
    ev_ref = ref.L.ref_dist_calls_reset()
    ref.L.ref_executor_free(ex)
    ids, dd, cnt, st = _search(amd, t, ix, nq, mode=amd.MODE_GRAPH, intra_threads=1, master_queue=L, local_queue=L)
    for q in range(nq):
        assert_topk_match(ids[q], dd[q], rid[q], rd[q], what="configs[2] graph q%d" % q)
    assert abs(st["dist_evals"] - ev_ref) <= 0.005 * ev_ref, (st["dist_evals"], ev_ref)
    assert st["rerank_rows"] > 0     # the traversal's 8-bit prefilter was on (the walk is the reference's either way)
    # the whole batch at the reference's default T = 4: size-independent properties
    ids4, dd4, cnt4, st4 = _search(amd, t, ix, B, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=L, local_queue=L)
    assert (cnt4 == K).all() and (ids4 >= 0).all() and (ids4 < N).all() and (np.diff(dd4, axis=1) >= 0).all()
    assert all(len(set(r.tolist())) == K for r in ids4)
    # r6: IntraQueryThreads = 4 at SearchQueueSize 2000 - the worker queues laid out with their occupancy bound (ceil(L / T) + I x out-degree keys: LDS
    # where the caller's capacity sent them to HBM), 8 wavefronts per query - against the oracle's lockstep schedule ON THIS GRAPH, which keeps the
    # caller's LocalQueueSize: the same walk key for key (ids position-wise, distances to 1e-4) at the size the layout change is for
    off, nbr, nav_ = ref.graph_arrays(gref)
    ref.L.ref_graph_free(gref)
    L2 = 2000
    init2 = oracle.prepare_init_ids(off, nbr, nav, L2)
    ids2, dd2, cnt2, st2 = _search(amd, t, ix, 3, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=L2, local_queue=L2)
    for q in range(3):
        oid, od, _ = oracle.search_impl(0, t["arr"], off, nbr, init2, Qh[q], T=4, L=L2, I=15, lockstep=True)
        assert_topk_match(ids2[q], dd2[q], oid[:K], od[:K], what="configs[2] graph T=4 L=2000 q%d" % q)
    del off, nbr
    ix.close()
    try:
        os.remove(str(tmp_path / "7" / "ann_graph_1.bin"))   # 4 GB
    except OSError:
        pass


def test_configs3_10M_x_768_cosine_with_id_filter_matches_the_reference_prefilter(amd, ref, table):
    t = table
    torch = t["torch"]
    amd.normalize_rows(t["X"], only_if_nonzero=True, device=0, stream=torch.cuda.current_stream().cuda_stream)   # as at insert
    amd.normalize_rows(t["Q"], only_if_nonzero=False, device=0, stream=torch.cuda.current_stream().cuda_stream)  # as TableMVP::Search
    torch.cuda.synchronize()
    _to_host(t, "cosine")
    idc = torch.arange(N, dtype=torch.int32, device=t["dev"])
    idc_host = np.arange(N, dtype=np.int32)
    ix = amd.GpuIndex(D, "COSINE", device=0).use_torch_stream()
    ix.attach_rows(t["X"])
    for sel in (0.5, 0.1, 0.9):
        qh = t["Q"][:3 if sel == 0.5 else 2].cpu().numpy()   # (the reference evaluates its expression tree on all 10M rows per query: ~5 s each)
        bound = int(N * sel)
        ix.set_int_filter(idc, "<", bound)
        ids, dd, cnt, st = _search(amd, t, ix, B, mode=amd.MODE_REFERENCE, prefilter=1)
        assert st["main_kernel_bits"] == 8, st
        assert (cnt == K).all() and (ids < bound).all() and (ids >= 0).all()
        assert (np.diff(dd, axis=1) >= 0).all()
        rid, rd, rcnt, sec = ref.prefilter_many(t["ptr"], N, D, idc_host, "ID < %d" % bound, qh, K, metric=1, threads=t["threads"])
        assert (rcnt == bound).all()
        for q in range(len(qh)):
            assert_topk_match(ids[q], dd[q], rid[q], rd[q], what="configs[3] %d%% q%d" % (int(sel * 100), q))
        if sel == 0.1:
            _all_distances_are_the_references(ref, t, ids, dd, 1, "configs[3] 10 %")
    ix.close()


def test_embedding_like_10M_x_768_cosine_stays_on_the_8_bit_pass_and_matches_the_reference_bruteforce(amd, ref, table):
    """unit-norm Gaussian rows with 8 dominant columns - the table whose 8-bit bound was too loose in the row frame (the fp16
    pass served it at half the rate) - at the headline's size.  The library's own choice (EPS_FLAT_AUTO) must be the 8-bit pass in the rotated
    frame with no overflowing query, and the answers the reference's BruteForceSearch answers (vec_search_executor.cpp:717-768, COSINE on rows
    normalised as at insert): 16 queries position-wise, every returned distance through the reference's CosineDistance, the fp32 stream
    engine bit for bit on 64, properties on all 1024.  Runs after the configs[3] test: it overwrites the shared table."""
    t = table
    torch = t["torch"]
    X, dev = t["X"], t["dev"]
    g = torch.Generator(device=dev).manual_seed(77)
    scale = torch.ones((D,), dtype=torch.float32, device=dev)
    scale[:8] = 4.0
    for s in range(0, N, 1 << 19):
        e = min(N, s + (1 << 19))
        X[s:e] = torch.randn((e - s, D), generator=g, device=dev, dtype=torch.float32) * scale
    Q = torch.randn((B, D), generator=torch.Generator(device=dev).manual_seed(78), device=dev, dtype=torch.float32) * scale
    amd.normalize_rows(X, only_if_nonzero=True, device=0, stream=torch.cuda.current_stream().cuda_stream)
    amd.normalize_rows(Q, only_if_nonzero=False, device=0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = {**t, "Q": Q, "host": None}
    _to_host(t, "embedding")
    table["host"] = "embedding"
    ix = amd.GpuIndex(D, "COSINE", device=0).use_torch_stream()
    ix.attach_rows(X)
    ids, dd, cnt, st = _search(amd, t, ix, B, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    ids, dd, cnt, st = _search(amd, t, ix, B, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)   # (the second call: past the table's first-call probe)
    assert (st["main_kernel_bits"], st["i8_rotated"], st["overflow_queries"]) == (8, 1, 0), st
    nref = 16
    rid, rd, sec = ref.bruteforce_many(t["ptr"], N, D, Q[:nref].cpu().numpy(), K, metric=1, threads=t["threads"])
    for q in range(nref):
        assert_topk_match(ids[q], dd[q], rid[q], rd[q], what="embedding-like q%d" % q)
    _all_distances_are_the_references(ref, t, ids, dd, 1, "embedding-like")
    sid, sd, scnt, _ = _search(amd, t, ix, 64, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert np.array_equal(ids[:64], sid) and np.array_equal(dd[:64], sd)
    assert (cnt == K).all() and (ids >= 0).all() and (ids < N).all() and (np.diff(dd, axis=1) >= 0).all()
    assert all(len(set(r.tolist())) == K for r in ids)
    ix.close()


def test_configs1_1M_x_768_one_query_per_call_matches_the_reference_bruteforce(amd, ref, oracle):
    """BASELINE configs[1] itself: 1M x 768 L2, k = 10, ONE query per eps_index_search call, 200 calls in a row - the one-pass
    search of csrc/stream8_kernel.hpp (one streaming pass over the 8-bit mirror, a racy shared table of the best accumulators, one exact
    re-rank) at the size the config names, pinned DIRECTLY to the compiled reference: every answer position-wise against the reference's
    BruteForceSearch (vec_search_executor.cpp:717-768) on the same rows; the same with `ID < N` at 50 % and 1 % against the reference's
    PreFilterBruteForceSearch (:770-831, its own parser and ExprEvaluator); and with a deleted bitset against the C oracle's restatement
    (bit-exact against the reference, tests/test_oracle_vs_ref.py).  `one_pass` must be 1 on every plain call - a call that overflowed a
    wavefront's list and was answered by the staged chain is counted and must not happen on this data."""
    import torch
    from helpers import bitset
    from oracle.pyoracle import make_filter
    n, nq = 1_000_000, 200
    dev = torch.device("cuda", 0)
    X = torch.rand((n, D), generator=torch.Generator(device=dev).manual_seed(42), device=dev)
    Q = torch.rand((nq, D), generator=torch.Generator(device=dev).manual_seed(43), device=dev)
    threads = int(ref.L.ref_omp_max_threads())
    arr, ptr = ref.alloc_rows(n, D, threads)
    try:
        arr[:] = X.cpu().numpy()
        Qh = Q.cpu().numpy()
        ix = amd.GpuIndex(D, "EUCLIDEAN", device=0).use_torch_stream()
        ix.attach_rows(X)
        o = (torch.empty((1, K), dtype=torch.int64, device=dev), torch.empty((1, K), dtype=torch.float32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev))

        def one_by_one(queries):
            ids, dd, one = [], [], 0
            for i in queries:
                ix.search(Q[i:i + 1], K, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
                ix.synchronize()
                st = ix.stats()
                assert st["main_kernel_bits"] == 8, st
                one += st["one_pass"]
                ids.append(o[0][0].cpu().numpy().copy())
                dd.append(o[1][0].cpu().numpy().copy())
                assert int(o[2][0]) == K
            return ids, dd, one
        # ---- plain: 200 sequential single-query calls
        ids, dd, one = one_by_one(range(nq))
        assert one == nq, "%d of %d calls fell back to the staged chain" % (nq - one, nq)
        rid, rd, sec = ref.bruteforce_many(ptr, n, D, Qh, K, metric=0, threads=threads)
        for q in range(nq):
            assert_topk_match(ids[q], dd[q], rid[q], rd[q], what="configs[1] q%d" % q)
        # ---- a deleted bitset (every 7th row and the 40 best answers of query 0): the oracle's BruteForceSearch with the same bitset
        gone = sorted(set(range(5, n, 7)) | set(int(v) for v in ref.bruteforce_many(ptr, n, D, Qh[:1], 40, metric=0, threads=threads)[0][0]))
        gone_set = set(gone)
        bits = bitset(n, gone)
        ix.set_deleted(bits)
        ids, dd, one = one_by_one(range(3))
        assert one == 3
        flt, keep = make_filter(deleted=bits)
        Xh = arr[:n]
        for q in range(3):
            oid, od = oracle.topk_flat(0, Xh, Qh[q], K, flt=flt)
            assert not (set(ids[q].tolist()) & gone_set)
            assert_topk_match(ids[q], dd[q], oid, od, what="configs[1] deleted q%d" % q)
        ix.set_deleted(None)
        # ---- `ID < N`: the int-column filter is evaluated inside the pass; the reference evaluates its expression tree before the distances
        idc = torch.arange(n, dtype=torch.int32, device=dev)
        idc_host = np.arange(n, dtype=np.int32)
        for bound, nf in ((n // 2, 8), (n // 100, 4)):
            ix.set_int_filter(idc, "<", bound)
            ids, dd, one = one_by_one(range(nf))
            # (50 %: the pass answers every call; 1 %: visible rows are too rare for the table of best accumulators to tighten in time - the
            # pass overflows, the staged chain answers, and after two such calls the filtered calls skip the pass: the same exact answer)
            assert one == nf or bound < n // 10, (bound, one)
            rid, rd, rcnt, sec = ref.prefilter_many(ptr, n, D, idc_host, "ID < %d" % bound, Qh[:nf], K, metric=0, threads=threads)
            assert (rcnt == bound).all()
            for q in range(nf):
                assert (ids[q] < bound).all()
                assert_topk_match(ids[q], dd[q], rid[q], rd[q], what="configs[1] ID < %d q%d" % (bound, q))
        ix.set_int_filter(None, "<", 0)
        ix.close()
    finally:
        ref.free_rows(ptr)
