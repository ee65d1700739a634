"""r6: the 8-bit mirror in a ROTATED frame (vectordb_amd/csrc/device_common.hpp rot256_load, mfma_filter.hip ensure_mirror8; arithmetic restated
and proven in tests/test_bound_math.py).  The frame may only change how many rows the 8-bit test lets through - never an answer: every user
of the mirror (the staged matrix filter, the one-pass search, the traversal's prefilter, appended rows) must return the fp32 stream scan's
bits (`BruteForceSearch`, reference engine/db/execution/vec_search_executor.cpp:717-768) in either frame, and the library's own choice must
put embedding-like tables on the rotated frame (and keep them on the 8-bit pass) and U[0,1) tables on the identity frame."""
import numpy as np
import pytest

from helpers import assert_topk_match, data

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd
    from vectordb_amd.build import build
    build()
    return vectordb_amd


def same(a, b, what=""):
    assert np.array_equal(a[0], b[0]), "%s: %d ids differ" % (what, (a[0] != b[0]).sum())
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2]), what


def embedding_like(n, d, seed, dominant=8, weight=4.0):
    """bench.py's embedding-like rows: Gaussian, `dominant` columns `weight` x the others, unit norm"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X[:, :dominant] *= weight
    return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,nq", [(70_000, 768, 40), (100_000, 128, 130), (66_000, 100, 64), (80_000, 33, 33), (70_000, 1024, 300)])
def test_forced_rotated_frame_is_exact_on_any_table(amd, oracle, monkeypatch, metric, n, d, nq):
    """EPS_MIRROR_ROTATE=1 on U[0,1) rows (the frame they do NOT want: the bound is ~3 x looser, so the lists are longer - and every answer is
    still the scan's): widths that pad to one, two, three and four 256-column blocks"""
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "1")
    X = data(n, d, 17 + d)
    Q = data(nq, d, 18 + d)
    if metric == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    if metric == 2:
        X = X - 0.5
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for k in (1, 10, 100):
        a = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        st = ix.stats()
        b = ix.search(Q, k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        same(a, b, "k=%d" % k)
        if st["main_kernel_bits"] == 8:
            assert st["i8_rotated"] == 1, st
    for qi in range(0, nq, 17):
        rid, rd = oracle.topk_flat(metric, X, Q[qi], 10)
        ids, dist, cnt = ix.search(Q[qi:qi + 1].repeat(32, 0), 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        assert_topk_match(ids[5], dist[5], rid, rd, what="rotated int8 vs oracle q%d" % qi)
    ix.close()


@pytest.mark.parametrize("metric", [1, 0])
def test_embedding_like_tables_get_the_rotated_frame_and_stay_on_the_8_bit_pass(amd, monkeypatch, metric):
    """The library's own choice (FLAT_AUTO, no switch): unit-norm rows with 8 dominant columns -> rotated frame, 8-bit pass, no overflow, the
    scan's answer; the same table with the frame forced to identity returns the same bits (whatever pass serves it) and re-ranks several
    times the rows."""
    n, d, nq = 300_000, 768, 256
    X, Q = embedding_like(n, d, 5), embedding_like(nq, d, 6)
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for it in range(3):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO), ref, "auto %d" % it)
        st = ix.stats()
        assert (st["main_kernel_bits"], st["i8_rotated"], st["overflow_queries"], st["i8_declined"]) == (8, 1, 0, 0), st
    rr_rot = st["rerank_rows"]
    same(ix.search(Q[:3], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), tuple(r[:3] for r in ref), "one-pass form, rotated frame")
    assert ix.stats()["one_pass"] == 1 and ix.stats()["i8_rotated"] == 1
    ix.close()
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "0")
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ref, "identity frame")
    st = ix.stats()
    assert st["i8_rotated"] == 0
    if st["main_kernel_bits"] == 8 and st["overflow_queries"] == 0:
        assert st["rerank_rows"] > 3 * rr_rot, (st["rerank_rows"], rr_rot)
    ix.close()


def test_uniform_tables_keep_the_identity_frame(amd):
    n, d, nq = 100_000, 768, 64
    X, Q = data(n, d, 41), data(nq, d, 42)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    st = ix.stats()
    assert st["main_kernel_bits"] == 8 and st["i8_rotated"] == 0, st
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM))
    ix.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("d", [192, 768, 1000])
def test_one_pass_search_in_the_rotated_frame(amd, monkeypatch, metric, d):
    """1 .. 16 queries, k up to 64: one pass over the rotated mirror + the re-rank == the scan (the prep launch rotates the queries)"""
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "1")
    n = 120_000
    X, Q = embedding_like(n, d, 50 + d), embedding_like(16, d, 51 + d)
    if metric == 0:
        X, Q = X * 3.0, Q * 3.0
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    for nq, k in ((1, 10), (2, 10), (3, 1), (4, 16), (8, 10), (16, 10), (1, 64), (5, 40)):
        a = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
        st = ix.stats()
        b = ix.search(Q[:nq], k, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
        same(a, b, "nq=%d k=%d" % (nq, k))
        assert st["i8_rotated"] == 1 and st["main_kernel_bits"] == 8, st
    for _ in range(5):   # (repeated single-query calls: a rotated table keeps the prep launch, the counters must still come back clean)
        same(ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ix.search(Q[:1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM))
    ix.close()


def test_rotated_mirror_is_extended_by_appended_rows(amd, monkeypatch):
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "1")
    n0, n1, d, nq = 80_000, 30_000, 256, 128
    X0, X1, Q = data(n0, d, 1), data(n1, d, 2) * 0.98 + 0.01, data(nq, d, 3)
    ix = amd.GpuIndex(d, 0)
    ix.attach_rows(X0)
    ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    ix.append_rows(X1)
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "0")           # (read when the mirror is first built only: the extension keeps the table's frame)
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    assert ix.stats()["i8_rotated"] == 1
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "after append")
    assert (a[0] >= n0).any()
    ix.append_rows(data(5_000, d, 4) * 1.5 - 0.25)          # far outside the grid: clamped, their residuals enter the bound
    a = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8)
    same(a, ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM), "after an out-of-grid append")
    ix.close()


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("n,d,T,L", [(20000, 128, 1, 300), (6000, 768, 4, 500), (5000, 772, 3, 200)])
def test_traversal_prefilter_is_invisible_in_the_rotated_frame(amd, monkeypatch, n, d, T, L, metric):
    """step d0 of traverse2_kernel reads the mirror rows: with the table in the rotated frame the walk is still the walk without the prefilter,
    bit for bit (queue contents, evaluation and expansion counts)"""
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "1")
    X, Q = embedding_like(n, d, 3 + d), embedding_like(32, d, 4 + d)
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    ix.build(n)
    k = min(L, 100)
    res = {}
    for pf in ("0", "1"):
        monkeypatch.setenv("EPS_TRV_PREFILTER", pf)
        ids, dist, cnt = ix.search(Q, k, mode=amd.MODE_GRAPH, intra_threads=T, master_queue=L, local_queue=L)
        st = ix.stats()
        res[pf] = (ids.copy(), dist.copy(), cnt.copy(), st["dist_evals"], st["expansions"], st["rerank_rows"])
    a, b = res["0"], res["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4]
    assert b[5] < b[3], (b[5], b[3])
    ix.close()


@pytest.mark.parametrize("metric", [1, 0, 2])
def test_rotated_frame_with_a_cut_grid_folds_the_clamped_rows_margins_and_stays_exact(amd, monkeypatch, metric):
    """r6, late: on tables of 4M rows and more the rotated grid cuts 10^-6 of the values off each tail (profiles/r6_embedding_like_grid_cut.txt): rows
    with a clamped value carry their own residual and the margins are folded per batch.  The same path at a size the suite can afford: the cut
    forced by EPS_MIRROR_CLIP (10^-4: far more clamped rows than the shipped 10^-6) - folded margins, the 8-bit pass, the scan's answer bit for bit
    for a batch and for single-query calls (one pass over the mirror, r6 late: offers carry `accumulator - 2 x margin`), and the traversal's prefilter
    invisible on it."""
    monkeypatch.setenv("EPS_MIRROR_ROTATE", "1")
    monkeypatch.setenv("EPS_MIRROR_CLIP", "4")
    n, d, nq = 200_000, 768, 200
    X, Q = embedding_like(n, d, 15), embedding_like(nq, d, 16)
    if metric == 0:
        X, Q = X * 2.0, Q * 2.0
    ix = amd.GpuIndex(d, metric)
    ix.attach_rows(X)
    ref = ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    for it in range(2):
        same(ix.search(Q, 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), ref, "batch %d" % it)
        st = ix.stats()
        assert (st["main_kernel_bits"], st["i8_rotated"], st["i8_folded"]) == (8, 1, 1), st
    one = 0
    for q in range(6):   # (r6, late: the one-pass search serves such tables too - folded start values, margin-free thresholds)
        same(ix.search(Q[q:q + 1], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), tuple(r[q:q + 1] for r in ref), "single query %d" % q)
        one += ix.stats()["one_pass"]
    same(ix.search(Q[:8], 10, mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA_I8), tuple(r[:8] for r in ref), "8 queries per call")
    assert one >= 1, one
    # the traversal's prefilter on the same mirror (per-batch start values, no edge constants)
    ix.build(20_000)
    res = {}
    for pf in ("0", "1"):
        monkeypatch.setenv("EPS_TRV_PREFILTER", pf)
        ids, dist, cnt = ix.search(Q[:32], 50, mode=amd.MODE_GRAPH, intra_threads=4, master_queue=300, local_queue=300)
        st = ix.stats()
        res[pf] = (ids.copy(), dist.copy(), cnt.copy(), st["dist_evals"], st["expansions"])
    a, b = res["0"], res["1"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2]) and a[3:] == b[3:]
    ix.close()
