import numpy as np


def data(n, d, seed):
    return np.random.default_rng(seed).random((n, d), dtype=np.float32)


def assert_topk_match(ids, dist, ref_ids, ref_dist, rtol=1e-4, atol=1e-6, what=""):
    """GPU top-k vs oracle top-k for ONE query.  Distances must agree to rtol (north_star: 1e-4 relative).
    IDs must be identical position by position, except inside groups of candidates whose oracle distances are
    closer than the fp32 summation-order noise (|d_i - d_j| <= 4e-6 * max(|d|, 1)): there the two sides may
    order/choose differently, which the reference itself would do on another compiler."""
    ids, ref_ids = np.asarray(ids), np.asarray(ref_ids)
    dist, ref_dist = np.asarray(dist, np.float64), np.asarray(ref_dist, np.float64)
    assert len(ids) == len(ref_ids), "%s: %d results vs %d expected" % (what, len(ids), len(ref_ids))
    if len(ids) == 0:
        return
    assert np.allclose(dist, ref_dist, rtol=rtol, atol=atol), "%s: distances differ\n%s\n%s" % (what, dist, ref_dist)
    if np.array_equal(ids, ref_ids):
        return
    tie = 4e-6 * np.maximum(np.abs(ref_dist), 1.0)
    bad = np.nonzero(ids != ref_ids)[0]
    for i in bad:
        # the id we returned must be an (almost) tie with the expected one at this rank
        assert abs(dist[i] - ref_dist[i]) <= tie[i], "%s: rank %d id %d != %d (d=%g vs %g)" % (
            what, i, ids[i], ref_ids[i], dist[i], ref_dist[i])
    inter = len(set(ids.tolist()) & set(ref_ids.tolist()))
    assert inter >= len(ids) - 1 - len(bad) // 2, "%s: id sets differ beyond boundary ties" % what


def bitset(n, ids):
    b = np.zeros((n + 7) // 8, np.uint8)
    for i in ids:
        b[i >> 3] |= 1 << (i & 7)
    return b
