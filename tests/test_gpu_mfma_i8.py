"""The 8-bit first pass of the batched flat scan (EPS_FLAT_MFMA_I8: int8 mirror on ONE grid per index, v_mfma_i32_32x32x32_i8,
integer thresholds, exact fp32 re-rank).  The contract is the fp16 engine's: the SAME exact answer as the fp32 stream scan
(`BruteForceSearch`, reference engine/db/execution/vec_search_executor.cpp:717-768), bit for bit, for any data - the filter's
bound is computed from the residuals of the stored bytes, so coarse operands may only cost re-ranked rows, never results."""
import os

import numpy as np
import pytest

from helpers import assert_topk_match, bitset, data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd
    from vectordb_amd.build import build
    build()
    return vectordb_amd


def same(a, b, what=""):
    assert np.array_equal(a[0], b[0]), "%s: %d ids differ" % (what, (a[0] != b[0]).sum())
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), what


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq", [(70_000, 768, 40), (100_000, 128, 130), (66_000, 100, 64), (80_000, 33, 33), (150_000, 384, 1100),
                                    (70_000, 1024, 300)])
def test_i8_engine_is_exact(amd, oracle, metric, n, d, nq):
    """int8 filter + fp32 re-rank == fp32 stream scan (same rows, same distance bits) == fp16 engine, and the oracle on a sample.
    Shapes: K depths of 4 (the minimum: d <= 512 pads to 512 bytes), 6 and 8 K-steps, padded last query tiles, 128-query
    tiles (nq <= 128) and several 256-query tiles."""
    X = data(n, d, 7 + d)
    Q = data(nq, d, 8 + d)
    if metric == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for k in (1, 10, 100):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        st = ix.stats()
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        assert st["main_kernel_bits"] == 8, st           # the 8-bit pass did run (no silent fall-back to fp16)
        assert st["overflow_queries"] == 0 and st["rerank_rows"] > 0
        # the looser bound may let more rows through than the fp16 pass, but it has to stay a FILTER
        assert st["rerank_rows"] < nq * (8.0 * k * n / 4096 + 512), "filter is not selective: %d" % st["rerank_rows"]
        same(a, b, "k=%d" % k)
    c = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    assert ix.stats()["main_kernel_bits"] == 16
    same(c, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), "fp16 vs int8")
    for qi in range(0, nq, 13):
        rid, rd = oracle.topk_flat(metric, X, Q[qi], 10)
        ids, dist, cnt = ix.search(Q[qi:qi + 1].repeat(32, 0), 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        assert_topk_match(ids[5], dist[5], rid, rd, what="int8 vs oracle q%d" % qi)
    ix.close()


@pytest.mark.parametrize("kind", ["gauss", "offset", "heavy_tail", "signed_ip", "tiny", "query_outliers"])
def test_i8_engine_on_other_distributions(amd, kind):
    """The grid is one (zero, step) pair per index, so what the values look like decides how tight the bound is - never the
    answer.  Gaussian rows (grid spans ~10 sigma), rows far from the origin (|x|^2 ~ 1e4 per dimension: the row constant is
    summed without cancellation), one huge outlier (the grid is stretched 100x: nearly everything passes, lists overflow, the
    fp16 pass takes over), signed rows under DOT_PRODUCT, values ~1e-3, and queries far outside the table's range (clamped
    on the grid: their own residual enters the bound)."""
    rng = np.random.default_rng(5)
    n, d, nq, metric = 90_000, 256, 200, 0
    if kind == "gauss":
        X = rng.standard_normal((n, d), dtype=np.float32)
        Q = rng.standard_normal((nq, d), dtype=np.float32)
    elif kind == "offset":
        X = 100.0 + rng.random((n, d), dtype=np.float32)
        Q = 100.0 + rng.random((nq, d), dtype=np.float32)
    elif kind == "heavy_tail":
        X = rng.random((n, d), dtype=np.float32)
        X[12345, 7] = 100.0
        Q = rng.random((nq, d), dtype=np.float32)
    elif kind == "signed_ip":
        metric = 2
        X = rng.standard_normal((n, d), dtype=np.float32)
        Q = rng.standard_normal((nq, d), dtype=np.float32) * 3.0
    elif kind == "tiny":
        X = rng.random((n, d), dtype=np.float32) * 1e-3
        Q = rng.random((nq, d), dtype=np.float32) * 1e-3
    else:
        X = rng.random((n, d), dtype=np.float32)
        Q = rng.random((nq, d), dtype=np.float32) * 4.0 - 1.5
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    st = ix.stats()
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    same(a, b, kind)
    if kind in ("gauss", "tiny"):
        assert st["main_kernel_bits"] == 8 and st["overflow_queries"] == 0, st
    auto = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    same(auto, b, kind + " (auto)")
    ix.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [192, 768, 1000])
def test_one_to_sixteen_queries_are_one_pass_over_the_mirror(amd, oracle, monkeypatch, metric, d):
    """r4 (stream8_kernel.hpp; r5: 5..16 queries by the same pass on the matrix cores, stream8m_kernel): up to 16 queries with k <= 64 (r4: 16) are answered by ONE streaming pass over the 8-bit mirror (shared table of the
    best accumulators seen -> pass threshold from their UPPER bounds), one selection against the final table and one exact re-rank: the
    same bits as the stream scan - rows of 2 / 3 / 4 x 256 bytes, ties ordered by id, queries that ARE rows, with a deleted bitset,
    an int-column filter, and (r5) a compiled filter PROGRAM (evaluated once per row into a bitset by filter_mask_kernel in front of the pass),
    repeated calls (the table is reset per call) - and `one_pass` in the stats says which form ran.  k > 64, more queries,
    EPS_FLAT_ONE_PASS=0, EPS_S8_MAX_K or (programs) EPS_S8_FILTER_PROGRAMS=0 take the staged chain."""
    n = 200_003 if d < 700 else 90_000     # (a last chunk that is not full)
    X, Q = data(n, d, 171 + d), data(32, d, 172 + d)   # (r6: up to 32 queries per call - two 16-query column blocks on the matrix cores)
    X[5000:5040] = X[4999]
    if metric == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    Q[1] = X[5010]
    idc = np.arange(n, dtype=np.int32)
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    from oracle.pyoracle import make_filter
    for setup in ("plain", "deleted", "deleted + filter", "deleted + filter program"):
        if setup == "deleted":
            ix.set_deleted(bitset(n, range(3, n, 11)))
        if setup == "deleted + filter":
            ix.set_int_filter(idc, ">=", 1000)
        # r5: the one-pass form against the ORACLE directly (the C restatement of BruteForceSearch, itself bit-exact against the compiled
        # reference: tests/test_oracle_vs_ref.py), not only against the library's own stream engine
        if "program" not in setup:
            flt, keep = make_filter(deleted=bitset(n, range(3, n, 11)) if setup != "plain" else None, attr=idc if "filter" in setup else None,
                                    op=">=" if "filter" in setup else None, value=1000)
            got = ix.search(Q[:2], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            assert ix.stats()["one_pass"] == 1
            for qi in range(2):
                rid, rd = oracle.topk_flat(metric, X, Q[qi], 10, flt=flt if setup != "plain" else None)
                assert_topk_match(got[0][qi], got[1][qi], rid, rd, what="one pass vs oracle, %s q%d" % (setup, qi))
        if setup == "deleted + filter program":     # (r5: the one-pass form behind one mask launch; the same predicate as the setup before it)
            ix.set_int_filter(None, ">=", 0)
            ix.set_filter_program([("i32", 0), ("const", 1000), (">=",)], rows=idc.view(np.uint8).reshape(n, 4), stride=4)
            flt, keep = make_filter(deleted=bitset(n, range(3, n, 11)), attr=idc, op=">=", value=1000)
            got = ix.search(Q[:2], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            assert ix.stats()["one_pass"] == 1
            for qi in range(2):
                rid, rd = oracle.topk_flat(metric, X, Q[qi], 10, flt=flt)
                assert_topk_match(got[0][qi], got[1][qi], rid, rd, what="one pass vs oracle, %s q%d" % (setup, qi))
            monkeypatch.setenv("EPS_S8_FILTER_PROGRAMS", "0")     # (A/B switch: programs on the staged chain, as until r4 - same bits)
            chain = ix.search(Q[:2], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            assert ix.stats()["one_pass"] == 0
            monkeypatch.delenv("EPS_S8_FILTER_PROGRAMS")
            same(chain, got, "filter program: chain vs one pass")
        for nq in (1, 2, 3, 4, 5, 8, 13, 16, 17, 24, 32, 1):     # (.. and back to one: whatever state a call leaves behind serves the next)
            for k in (1, 10, 16, 17, 40, 64, 10):    # (r5: k = 17..64 on 128 table slots per query - a table layout of its own - and back)
                for rep in range(2):
                    a = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
                    st = ix.stats()
                    # (k <= 16: always the one-pass form here.  Larger k passes more rows against the same candidate lists: on the BASELINE's shape -
                    # L2 - it must be the one-pass form as well; where the bound is loose against the spread - normalised rows - a call may hand over
                    # to the staged chain (whose own lists may overflow into the fp16 pass: exact either way, checked below), which must not cost
                    # the small-k calls after it their one-pass form)
                    assert st["one_pass"] == 1 or (k > 16 and metric != 0), (setup, nq, k, st)
                    if st["one_pass"]:
                        assert (st["main_kernel_bits"], st["overflow_queries"]) == (8, 0), (setup, nq, k, st)
                        assert st["rerank_rows"] < nq * (2000 if k <= 16 else 4096), st     # the junk of the first microseconds is dropped before any row is read
                    same(a, ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "%s nq %d k %d" % (setup, nq, k))
        a = ix.search(Q[:2], 65, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        assert ix.stats()["one_pass"] == 0
        same(a, ix.search(Q[:2], 65, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "k 65")
        monkeypatch.setenv("EPS_S8_MAX_K", "16")     # (A/B switch: k > 16 on the staged chain, as until r4 - same bits)
        a = ix.search(Q[:2], 40, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        assert ix.stats()["one_pass"] == 0
        monkeypatch.delenv("EPS_S8_MAX_K")
        b = ix.search(Q[:2], 40, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        assert ix.stats()["one_pass"] == 1 or metric != 0
        same(a, b, "k 40: chain vs one pass")
    ix.set_filter_program(None)
    monkeypatch.setenv("EPS_FLAT_ONE_PASS", "0")
    a = ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["one_pass"] == 0
    monkeypatch.delenv("EPS_FLAT_ONE_PASS")
    same(a, ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), "chain vs one pass")
    assert ix.stats()["one_pass"] == 1
    ix.close()


@pytest.mark.parametrize("metric", [0, 2])
def test_one_pass_tail_on_rows_that_are_not_a_multiple_of_four_floats(amd, metric):
    """r5 (late): the one-pass call's own re-rank (s8_rerank_kernel: flat selection through LDS, every piece of a row in flight at once, the k
    best by rank) in its scalar-load form - d = 333: rows are not 16-byte multiples - against the stream engine; k up
    to 64, a run of identical rows (equal distances: ordered by id), EPS_S8_RERANK=0 (rerank_kernel with the selection prologue) the same bits."""
    n, d = 70_001, 333
    X, Q = data(n, d, 77), data(16, d, 78)
    X[3000:3030] = X[2999]
    Q[0] = X[3010]
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for nq in (1, 2, 5, 16):
        for k in (1, 10, 40, 64):
            want = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
            a = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            assert ix.stats()["one_pass"] == 1 or metric != 0, (nq, k)
            same(a, want, "d 333 nq %d k %d" % (nq, k))
            os.environ["EPS_S8_RERANK"] = "0"
            try:
                b = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            finally:
                del os.environ["EPS_S8_RERANK"]
            same(b, want, "d 333 nq %d k %d (rerank_kernel)" % (nq, k))
    ix.close()


def test_filter_programs_take_the_one_pass_form_behind_a_mask(amd):
    """r5: a call under a compiled filter program (packed attribute rows {i32 id; f32 price; u8 flag; pad; f64 w}) is the one-pass search too:
    filter_mask_kernel evaluates the predicate once per row (with the deleted bitset) into a bitset, pass and re-rank read that.  Against numpy
    (fp64 distances, stable order) for 1, 3, 7 and 16 queries; the attribute rows and the bitset are re-read on EVERY call (they are the
    caller's memory, used in place: a change between two calls must show); a program that reads @distance outside a pre-filter call stays on
    the engine that evaluates it per candidate."""
    n, d = 110_003, 256
    X, Q = data(n, d, 41), data(16, d, 42)
    rng = np.random.default_rng(43)
    rows = np.zeros(n, dtype=np.dtype([("id", "<i4"), ("price", "<f4"), ("flag", "u1"), ("pad", "u1", 7), ("w", "<f8")]))
    rows["id"], rows["price"], rows["flag"], rows["w"] = np.arange(n), rng.random(n), rng.integers(0, 2, n), rng.random(n)
    dist = np.stack([((X.astype(np.float64) - q) ** 2).sum(1) for q in Q])
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    dele = np.zeros(n, dtype=bool)
    dele[5::13] = True
    ix.set_deleted(bitset(n, np.flatnonzero(dele)))
    progs = [
        ([("i32", 0), ("const", 35000), ("<",), ("f32", 4), ("const", 0.5), ("<",), ("and",)], lambda: (rows["id"] < 35000) & (rows["price"] < 0.5)),
        ([("bool", 8), ("not",), ("f64", 16), ("const", 0.25), (">=",), ("or",)], lambda: (rows["flag"] == 0) | (rows["w"] >= 0.25)),
        ([("i32", 0), ("const", 7), ("%",), ("const", 3), ("=",)], lambda: rows["id"] % 7 == 3),
    ]
    for prog, ref_mask in progs:
        ix.set_filter_program(prog, rows)
        m = ref_mask() & ~dele
        for nq in (1, 3, 7, 16):
            ids, dd, cnt = ix.search(Q[:nq], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            st = ix.stats()
            assert (st["one_pass"], st["overflow_queries"]) == (1, 0), (prog, nq, st)
            for qi in range(nq):
                want = np.argsort(np.where(m, dist[qi], np.inf), kind="stable")[:10]
                assert list(ids[qi]) == list(want), (prog, nq, qi)
            same((ids, dd, cnt), ix.search(Q[:nq], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "program, nq %d" % nq)
    # the caller's memory changes between two calls (device-resident attribute rows, used in place): the next call's mask follows
    import torch
    rows_d = torch.from_numpy(rows.view(np.uint8).reshape(n, rows.dtype.itemsize).copy()).cuda()
    ix.set_filter_program([("f32", 4), ("const", 0.5), ("<",)], rows_d, stride=rows.dtype.itemsize)
    a = ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["one_pass"] == 1
    m = (rows["price"] < 0.5) & ~dele
    assert list(a[0][0]) == list(np.argsort(np.where(m, dist[0], np.inf), kind="stable")[:10])
    price = rows_d.view(torch.float32).reshape(n, rows.dtype.itemsize // 4)[:, 1]
    price[torch.from_numpy(a[0][0]).cuda()] = 0.75        # the ten answers leave the filter
    torch.cuda.synchronize()
    b = ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["one_pass"] == 1
    m[a[0][0]] = False
    assert list(b[0][0]) == list(np.argsort(np.where(m, dist[0], np.inf), kind="stable")[:10])
    # @distance outside a pre-filter call: evaluated per candidate, not by a mask
    sd = np.sort(dist[0])
    thr = float(0.5 * (sd[199] + sd[200]))
    ix.set_deleted(None)
    ix.set_filter_program([("dist",), ("const", thr), (">",)], rows)
    ids, dd, cnt = ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["one_pass"] == 0
    assert list(ids[0]) == list(np.argsort(dist[0], kind="stable")[200:210])
    ix.close()


@pytest.mark.parametrize("switches", [{}, {"EPS_S8_TWO_LAUNCHES": "0"}, {"EPS_S8_HOST_WORDS": "0"}, {"EPS_HOST_STAGING": "0"}, {"EPS_S8_RERANK": "0"},
                                      {"EPS_S8_RERANK": "0", "EPS_S8_TWO_LAUNCHES": "0"}])
def test_one_pass_call_forms_return_the_same_bits(amd, monkeypatch, switches):
    """r5: a one-pass call is two launches (the pass quantises its queries itself, the re-rank leaves table and counters clean for the next call) and
    its two result counters reach the host through host-mapped words; host-pointer calls get their results in one page-locked copy; the call's tail is
    s8_rerank_kernel (r5, late; EPS_S8_RERANK=0: rerank_kernel with the selection prologue).  Every form
    against the stream engine, over calls that alternate query counts (1, 2 take the two-launch form, 3, 4 the prep launch), k, a staged-chain call
    in between (it uses the same counter block) and host / device pointers."""
    import torch
    for k_, v in switches.items():
        monkeypatch.setenv(k_, v)
    n, d = 120_011, 768
    X, Q = data(n, d, 901), data(8, d, 902)
    Q[2] = X[777]
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    want = {}
    for nq in (1, 2, 3, 4, 8):
        for k in (1, 10):
            want[nq, k] = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    Qd = torch.from_numpy(Q).cuda()
    order = [(1, 10), (1, 10), (2, 1), (4, 10), (1, 1), (8, 10), (1, 10), (3, 10), (2, 10), (1, 10)]
    for rep in range(3):
        for nq, k in order:
            a = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            st = ix.stats()
            assert st["one_pass"] == 1 and st["overflow_queries"] == 0, (switches, nq, k, st)
            same(a, want[nq, k], "%s host nq %d k %d" % (switches, nq, k))
            o = (torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), device="cuda"), torch.empty((nq,), dtype=torch.int32, device="cuda"))
            ix.search(Qd[:nq], k, out=o, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            torch.cuda.synchronize()
            same((o[0].cpu().numpy(), o[1].cpu().numpy(), o[2].cpu().numpy()), want[nq, k], "%s device nq %d k %d" % (switches, nq, k))
    ix.close()


def test_auto_builds_the_mirror_for_single_query_traffic_after_a_few_calls(amd):
    """FLAT_AUTO on a table without a mirror: single queries run the fp32 stream scan (no HBM spent on a mirror for a caller that may ask
    once), and from the 17th call on the same rows the 8-bit mirror is built and the one-pass search answers - same bits either way;
    re-attaching rows starts the count again."""
    n, d = 120_000, 256
    X, Q = data(n, d, 310), data(40, d, 311)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for attach in range(2):
        seen = []
        for i in range(24):
            r = ix.search(Q[i:i + 1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
            st = ix.stats()
            seen.append((st["main_kernel_bits"], st["one_pass"]))
            assert np.array_equal(r[0][0], ref[0][i]) and np.array_equal(r[1][0], ref[1][i]), (attach, i)
        assert seen[:16] == [(32, 0)] * 16 and seen[17:] == [(8, 1)] * 7, seen
        ix.attach_rows(X)
    ix.close()


def test_one_pass_search_survives_adversarial_order_and_selective_deletion(amd):
    """rows sorted from far to near (the table of best accumulators is always behind: nearly every row passes -> the raw list overflows -> the
    staged chain answers), 99.9 % of the rows deleted (only visible rows may enter the table), fewer visible rows than k: the answer is
    the stream scan's whatever ran."""
    rng = np.random.default_rng(5)
    n, d = 200_000, 256
    q0 = rng.random((1, d), dtype=np.float32)
    X = rng.random((n, d), dtype=np.float32)
    X = np.ascontiguousarray(X[np.argsort(-((X - q0) ** 2).sum(1))])
    Q = (q0 + 0.01 * rng.random((2, d), dtype=np.float32)).astype(np.float32)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    for _ in range(3):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "adversarial order")
    ix.close()
    X, Q = data(n, d, 220), data(3, d, 221)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    for keep_every in (1000, 40_000):    # 200 visible rows; 5 visible rows (< k)
        dele = np.ones(n, dtype=bool)
        dele[::keep_every] = False
        ix.set_deleted(bitset(n, np.flatnonzero(dele)))
        a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "keep every %d" % keep_every)
        assert (a[2] == min(10, len(range(0, n, keep_every)))).all(), a[2]
    ix.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_a_handful_of_queries_takes_the_short_chain_and_the_split_rerank(amd, monkeypatch, metric):
    """r4: 1 ... 16 queries per call run the fused launch chain (query preparation + fragment copy + start state in one launch, seed
    selection straight into the candidate lists by a 16-wavefront workgroup, finalisation in the last re-rank, 3 stages up to 4
    queries) and every re-rank spread over 8 workgroups per query with a last-arrival merge: the same bits as the stream scan, with
    deletions and a filter, for k = 1 / 10 / 100, and the same with the split re-rank switched off."""
    n, d = 150_000, 192
    X, Q = data(n, d, 71), data(16, d, 72)
    X[5000:5040] = X[4999]                      # ties: equal distances, ordered by id
    if metric == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    Q[1] = X[5010]
    idc = np.arange(n, dtype=np.int32)
    for split in ("1", "0"):
        monkeypatch.setenv("EPS_RERANK_SPLIT", split)
        ix = amd.GpuIndex(d, metric)
        ix.attach_rows(X)
        for setup in ("plain", "deleted + filter"):
            if setup != "plain":
                ix.set_deleted(bitset(n, range(3, n, 11)))
                ix.set_int_filter(idc, ">=", 1000)
            for nq in (1, 2, 3, 4, 5, 16):
                for k in (1, 10, 100):
                    a = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
                    st = ix.stats()
                    assert st["main_kernel_bits"] == 8 and st["overflow_queries"] == 0, (split, setup, nq, k, st)
                    same(a, ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "split %s %s nq %d k %d" % (split, setup, nq, k))
            # the same call many times: the arrival counters are back at zero after every launch
            first = ix.search(Q[:2], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            for _ in range(20):
                same(ix.search(Q[:2], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), first, "repeat")
        ix.close()


def test_auto_stops_paying_for_an_8_bit_pass_that_never_filters(amd):
    """A table the 8-bit bound cannot filter (heavy-tailed values in every row).  r4: the library's own choice PROBES the
    8-bit pass on the first batch it sends through a mirror - the candidate count of the first, smallest stage predicts the others -
    finds it too loose and answers with the fp16 pass at once (r3 paid three whole overflowing 8-bit attempts first); later batches
    go to the fp16 pass directly; an explicit EPS_FLAT_MFMA_I8 request still runs the 8-bit pass (and overflows into the fp16 pass);
    every answer is the stream scan's.  EPS_MFMA_PROBE=0 restores the r3 behaviour (three overflows, then fp16)."""
    rng = np.random.default_rng(6)
    n, d, nq = 90_000, 256, 96
    # (r4's clipped grid + per-row margins serve a table with a FEW outliers - next test; what defeats ONE grid is a heavy tail in every
    # row: Cauchy values - whatever the grid, most rows are clamped somewhere and carry residuals as large as the distances)
    X = np.clip(rng.standard_cauchy((n, d)), -1e4, 1e4).astype(np.float32)     # (inside the fp16 range: the fp16 pass can serve it)
    Q = np.clip(rng.standard_cauchy((nq, d)), -1e4, 1e4).astype(np.float32)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    seen = []
    for it in range(4):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO), ref, "auto %d" % it)
        st = ix.stats()
        seen.append((st["overflow_queries"] > 0, st["main_kernel_bits"], st["i8_declined"]))
    assert seen[0] == (False, 16, 1) and all(s == (False, 16, 0) for s in seen[1:]), seen
    same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ref, "explicit int8")
    assert ix.stats()["overflow_queries"] > 0
    ix.close()
    os.environ["EPS_MFMA_PROBE"] = "0"
    try:
        ix = amd.GpuIndex(d, 0)
        ix.attach_rows(X)
        seen = []
        for it in range(5):
            same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO), ref, "auto, no probe %d" % it)
            st = ix.stats()
            seen.append((st["overflow_queries"] > 0, st["main_kernel_bits"]))
        assert all(o for o, b in seen[:3]) and seen[3] == (False, 16) and seen[4] == (False, 16), seen
        ix.close()
    finally:
        del os.environ["EPS_MFMA_PROBE"]


def test_a_selective_filter_does_not_talk_the_library_out_of_the_8_bit_pass(amd):
    """The probe judges the BOUND by the first stage's candidate count - but a filter that lets 0.3 % of the rows through inflates every
    list by 1 / 0.003 whatever the bound is worth.  Such a first batch may be answered by another engine (exactly), yet it must not decide
    for the mirror: the next unfiltered batch runs the 8-bit pass."""
    n, d, nq = 120_000, 128, 70
    X, Q = data(n, d, 51), data(nq, d, 52)
    idc = np.arange(n, dtype=np.int32)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_int_filter(idc, "<", n // 300)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)       # the mirror's FIRST batch: probed under the filter
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "filtered first batch")
    assert (a[0] < n // 300).all()
    ix.set_int_filter(None, None, 0)
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    st = ix.stats()
    assert (st["main_kernel_bits"], st["i8_declined"], st["overflow_queries"]) == (8, 0, 0), st
    same(b, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "unfiltered second batch")
    ix.close()


def test_a_few_outlier_values_cost_their_rows_not_the_table(amd):
    """r4: the grid is clipped to the bulk of the values and every row carries its own residual norms, folded per batch into its
    accumulator start value.  One value of 100 in a U[0,1) table (r3 / early r4: grid stretched 100 x, 8-bit pass useless, fp16 pass
    served the table) now only clamps its own row; a value so far out that the row's constant leaves the accumulator's range makes the
    row FORCED (always a candidate, re-ranked exactly).  The 8-bit pass serves the table, nothing is declined or overflows, and the
    outlier rows are still found exactly when they ARE the answer (queries = those rows; a query that is itself far outside the grid)."""
    rng = np.random.default_rng(26)
    n, d, nq = 100_000, 256, 80
    X = rng.random((n, d), dtype=np.float32)
    X[4321, 3] = 100.0            # clamped, large residual, still tested
    X[777, 200] = -40.0
    X[60_000, 17] = 30_000.0      # forced: |R| / u beyond 2^29
    Q = rng.random((nq, d), dtype=np.float32)
    clean = amd.GpuIndex(d, 0)          # the same table without the three values: what the filter passes there
    Xc = X.copy()
    Xc[4321, 3], Xc[777, 200], Xc[60_000, 17] = 0.5, 0.5, 0.5
    clean.attach_rows(Xc)
    clean.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    stc = clean.stats()
    assert (stc["main_kernel_bits"], stc["i8_folded"]) == (8, 0), stc      # homogeneous rows: table-wide margin, no fold pass
    clean.close()
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for it in range(3):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO), ref, "auto %d" % it)
        st = ix.stats()
        assert (st["main_kernel_bits"], st["i8_declined"], st["overflow_queries"], st["i8_folded"]) == (8, 0, 0, 1), (it, st)
        # ... and it FILTERS as if the outliers were not there (the forced row adds one candidate per query and stage)
        # (the forced row is a candidate of every query in every stage, the two clamped rows of most; the batch's largest query norms
        # instead of every query's own)
        assert st["rerank_rows"] < 1.6 * stc["rerank_rows"] + 24 * nq, (st["rerank_rows"], stc["rerank_rows"])
    # queries that ARE the outlier rows (far outside the grid themselves: their batch's margins are as large as their residuals, the
    # lists overflow and the fp16 / stream engines answer - exactly): the rows are found
    Qo = np.stack([X[4321], X[777], X[60_000], X[60_000] + np.float32(0.01)] + [Q[i] for i in range(8)])
    refo = ix.search(Qo, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert refo[0][0, 0] == 4321 and refo[0][1, 0] == 777 and refo[0][2, 0] == 60_000 and refo[0][3, 0] == 60_000
    same(ix.search(Qo, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), refo, "outlier queries")
    same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO), ref, "regular batch again")
    assert ix.stats()["main_kernel_bits"] == 8
    # rows appended after the grid was fixed, one of them far outside: same machinery
    Xa = rng.random((500, d), dtype=np.float32)
    Xa[7, 9] = 55.0
    ix.append_rows(Xa)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ref, "appended outlier")
    st = ix.stats()
    assert st["main_kernel_bits"] == 8 and st["overflow_queries"] == 0, st
    ix.close()


def test_auto_keeps_the_8_bit_pass_where_it_filters_and_probes_only_once(amd):
    """... and on a table the centred grid serves (a 16-dimensional manifold in 256 dimensions - r3's grid overflowed on such rows at
    scale) the probe passes: every AUTO batch runs the 8-bit pass, none is declined, all equal the stream scan."""
    rng = np.random.default_rng(16)
    n, d, nq = 120_000, 256, 200
    A = (0.25 * rng.standard_normal((16, d))).astype(np.float32)
    X = (rng.random((n, 16), dtype=np.float32) @ A + 0.01 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    Q = (rng.random((nq, 16), dtype=np.float32) @ A + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for it in range(3):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO), ref, "auto %d" % it)
        st = ix.stats()
        assert (st["main_kernel_bits"], st["i8_declined"], st["overflow_queries"]) == (8, 0, 0), (it, st)
    ix.close()


def test_i8_engine_declines_tables_it_cannot_serve(amd):
    """All values equal (no grid), or a non-finite value: the request for the 8-bit pass is served by the fp16 / stream engine,
    same answer."""
    n, d, nq = 70_000, 128, 64
    Q = data(nq, d, 3)
    X = np.full((n, d), 0.25, np.float32)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["main_kernel_bits"] != 8
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "constant table")
    ix.close()
    X = data(n, d, 4)
    X[777, 5] = np.inf
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["main_kernel_bits"] != 8
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    assert np.array_equal(a[0], b[0])
    ix.close()


def test_i8_engine_with_deleted_filter_and_ties(amd, oracle):
    from oracle.pyoracle import make_filter
    n, d, nq = 90_000, 64, 48
    rng = np.random.default_rng(3)
    base = rng.random((300, d), dtype=np.float32)
    X = base[rng.integers(0, 300, n)]                     # every row has ~300 exact duplicates: massive ties
    Q = rng.random((nq, d), dtype=np.float32)
    bits = bitset(n, range(0, n, 3))
    idc = np.arange(n, dtype=np.int64)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_deleted(bits)
    ix.set_int_filter(idc, ">=", 1000)
    for k in (10, 64):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        same(a, b)
    flt, keep = make_filter(deleted=bits, attr=idc, stride=8, width=8, op=">=", value=1000)
    rid, rd = oracle.topk_flat(0, X, Q[0], 10, flt=flt)
    assert np.array_equal(a[0][0][:10], rid)               # ties resolve by id exactly as Candidate::operator<
    ix.close()


def test_i8_engine_adversarial_order_and_selective_filter(amd):
    """Rows sorted from far to near (every stage's threshold is too optimistic -> lists overflow -> fp16 pass -> stream scan), and
    99.5 % of the rows deleted (thresholds from visible rows only).  Whatever chain of fall-backs runs, the answer is the scan's."""
    rng = np.random.default_rng(1)
    n, d, nq = 300_000, 32, 64
    q0 = rng.random((1, d), dtype=np.float32)
    X = rng.random((n, d), dtype=np.float32)
    X = np.ascontiguousarray(X[np.argsort(-((X - q0) ** 2).sum(1))])
    Q = (q0 + 0.01 * rng.random((nq, d), dtype=np.float32)).astype(np.float32)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "adversarial order")
    ix.close()
    n, d, nq = 400_000, 64, 96
    X, Q = data(n, d, 120), data(nq, d, 121)
    dele = rng.random(n) < 0.995
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    ix.set_deleted(np.packbits(dele, bitorder="little"))
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "selective filter")
    assert not dele[a[0][a[0] >= 0]].any()
    ix.close()


def test_i8_mirror_is_extended_by_appended_rows(amd):
    """Appended rows are quantised on the grid the table already has - the mirror is extended, not rebuilt - and the answer over
    old + new rows equals the scan's.  Rows beyond the grid are clamped: their residual enters the (per-index) bound, which may
    then be too loose for the 8-bit pass to pay (the fp16 pass takes over); the answer does not change."""
    n0, n1, d, nq = 80_000, 30_000, 256, 128
    X0, X1, Q = data(n0, d, 1), data(n1, d, 2) * 0.98 + 0.01, data(nq, d, 3)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X0)
    ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    ix.append_rows(X1)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["main_kernel_bits"] == 8
    b = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    same(a, b, "after append")
    assert (a[0] >= n0).any()                              # some of the new rows are among the answers
    X2 = data(5_000, d, 4) * 1.5 - 0.25                    # far outside [0, 1): clamped on the grid
    ix.append_rows(X2)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "after an out-of-grid append")
    ix.close()


def test_query_beyond_the_fp16_range_is_not_dropped(amd):
    """a query component beyond 65504 became +inf in the fp16 operand, and inf * 0 = NaN accumulators
    silently failed `acc >= T` for rows holding an exact 0 in that column.  Such a batch now runs on an engine that can
    represent it; the answer is the stream scan's."""
    n, d, nq = 70_000, 128, 32
    X = data(n, d, 9)
    X[::2, 5] = 0.0
    Q = data(nq, d, 10)
    Q[3, 5] = 1.0e6
    Q[7, 0] = -3.0e5
    ix = amd.GpuIndex(d, 2)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for eng in (amd.FLAT_MFMA, amd.FLAT_MFMA_I8, amd.FLAT_AUTO):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=eng), ref, "engine %d" % eng)
    ix.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_one_pass_search_on_a_table_with_folded_margins(amd, metric):
    """r6: ONE-PASS calls (1-16 queries per call, k up to 64) on a table whose rows differ - a clamped value, a forced row - so that its margins are
    folded per batch: the per-row margins folded for the call's queries (fold8_kernel behind the prep launch), margin-free thresholds, offers of
    `accumulator - 2 x the row's margin` (stream8_offer_value: the number whose upper bound holds for the row).  The scan's answer bit for bit, outlier
    rows and outlier queries included; the forced row is a candidate of every call and never enters the slot table."""
    rng = np.random.default_rng(27 + metric)
    n, d = 120_000, 512
    X = rng.random((n, d), dtype=np.float32)
    X[4321, 3] = 100.0
    X[777, 200] = -40.0
    if metric == 0:
        X[60_000, 17] = 30_000.0      # forced
    Q = rng.random((16, d), dtype=np.float32)
    Qo = np.stack([X[4321], X[777], X[60_000], X[5] + np.float32(0.01)] + [Q[i] for i in range(12)])
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["i8_folded"] == 1, ix.stats()
    served = 0
    for nq1, k1 in ((1, 10), (2, 10), (4, 16), (8, 10), (16, 10), (1, 64), (3, 40)):
        for qset in (Q, Qo):
            got = ix.search(qset[:nq1], k1, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
            st1 = ix.stats()
            same(got, ix.search(qset[:nq1], k1, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "one-pass, folded margins: %d queries k=%d" % (nq1, k1))
            served += st1["one_pass"]
            assert not st1["one_pass"] or (st1["i8_folded"] == 1 and st1["main_kernel_bits"] == 8), st1
    assert served >= 5, served      # (a call of outlier queries may overflow its lists and take the staged chain; regular queries must not)
    ix.close()
