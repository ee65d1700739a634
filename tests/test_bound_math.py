"""The 8-bit lower bound, checked as arithmetic on the CPU (no GPU): a numpy restatement of what colmean_kernel / quant_mirror_kernel /
query_prep8_kernel / stage_threshold8 (vectordb_amd/csrc/mfma_filter.hip, device_common.hpp) compute, in float32 where the device
uses float32, and the one property every user of the mirror relies on - the flat engine's filter stages, the traversal's and the
build searches' prefilter:

    fp32 distance(q, x) <= thr   ==>   dot(qi, xi) + acc0[x] >= Tq(thr)            (a row that fails the test is PROVABLY farther)

for every row and query, with no assumption about how the values are distributed: uniform rows, Gaussian rows, rows far from the
origin, rows appended outside the grid (clamped codes), queries far outside the table's range, the three metrics.

r4: the grid is CENTRED - rows and queries are quantised as x - mu on a symmetric grid, mu = one value per column (column means of
a sample + the mid-range of what is left; ANY mu keeps the bound valid, a good one makes it tight).  The Cauchy-Schwarz margin then
scales with |q - mu| and |x - mu| instead of |q| and |x|: half the margin on U[0,1) rows, and nothing is lost on tables far from
the origin."""
import numpy as np
import pytest

F = np.float32
EPS24 = F(5.9604645e-8)


ACC_FORCE = 0x38000000   # start value of a row that must pass whatever the threshold (thresholds are clamped to <= TQ_MAX, |dot| < 2^27)
TQ_MAX = 0x30000000
TAIL = 1e-7   # share of the SAMPLE's values the grid may leave outside on either side (ensure_mirror8: centre_hist_kernel)


def col_centre(X, sample=None):
    """what ensure_mirror8 does on the first build: column means (fp32) of a strided sample; then the grid's range from a 4096-bin
    histogram of x - mean over the sample, clipped so that at most max(2, TAIL * values) sample values lie outside on either side (one
    outlier value must not stretch the grid for ten million rows: a clamped row pays with ITS residual, below); centre = mean + mid-range
    of the clipped interval.  Returns (mu, half-range)."""
    S = X if sample is None else X[sample]
    mean = (S.astype(F).sum(0, dtype=F) / F(len(S))).astype(F)
    c = (X - mean).astype(F)
    lo, hi = F(c.min()), F(c.max())
    tol = max(2, int(TAIL * S.size))

    def clip_range(mean, clo, chi):
        cs = (S - mean).astype(F)
        for _ in range(3):      # (an outlier thousands of grid widths away leaves the bulk in ONE bin: the histogram is taken again inside the cut)
            binw = (chi - clo) / F(4096.0)
            if not binw > 0:
                break
            b = np.clip(((cs - clo) * (F(1.0) / binw)).astype(np.int64), 0, 4095)
            hist = np.bincount(b.ravel(), minlength=4096)
            cum = np.cumsum(hist)
            blo = int(np.searchsorted(cum, tol, side="right"))              # first bin whose cumulative count exceeds tol
            cumr = np.cumsum(hist[::-1])
            bhi = 4095 - int(np.searchsorted(cumr, tol, side="right"))
            a_, b_ = clo + F(blo) * binw, clo + F(bhi + 1) * binw
            if not (blo < 4096 and bhi >= 0 and b_ > a_):
                break
            cut_most = (b_ - a_) < F(0.25) * (chi - clo)
            clo, chi = a_, b_
            if not cut_most:
                break
        return clo, chi
    clo, chi = clip_range(mean, lo, hi)
    if clo > lo or chi < hi:
        # something was cut: the column means it polluted are estimated again from values clamped into the cut, the range once more
        mean2 = (np.clip(S, mean + clo, mean + chi).astype(F).sum(0, dtype=F) / F(len(S))).astype(F)
        delta = F(np.abs(mean2 - mean).max())
        mean = mean2
        clo, chi = clip_range(mean, lo - delta, hi + delta)
    z0 = F(0.5) * clo + F(0.5) * chi
    return (mean + z0).astype(F), F(max(chi - z0, z0 - clo))


def mirror(X, metric, mu=None, step=None):
    if mu is None:
        mu, half = col_centre(X)
        if step is None:
            step = half / F(127.0)
    xc = (X - mu).astype(F)                                   # x' = fl(x - mu)
    if step is None:
        step = F(max(abs(F(xc.min())), abs(F(xc.max())))) / F(127.0)
    inv = F(1.0) / step
    xi = np.clip(np.rint(xc * inv), -127, 127).astype(np.int32)
    res = (xc - step * xi.astype(F)).astype(F)               # fma(-step, xi, x')
    xh = (step * xi.astype(F)).astype(F)
    s = F(2.0) if metric == 0 else F(1.0)
    u = s * step * step
    x2c = (xc * xc).sum(1, dtype=F)
    if metric == 0:
        R = x2c
    else:
        R = -(mu * xc).sum(1, dtype=F)
    a0 = np.ceil(-R / u) + 1
    # per ROW (r4): the two norms the Cauchy-Schwarz margin multiplies the query's with
    erow = (np.sqrt((res * res).sum(1, dtype=F)) * F(1.00001) + F(1.2e-7) * np.sqrt(x2c)).astype(F)   # + the rounding of x - mu
    hrow = (np.sqrt((xh * xh).sum(1, dtype=F)) * F(1.00001)).astype(F)
    # a row whose constant leaves the accumulator's range (an outlier far outside the clipped grid) is FORCED: its test is not evaluated,
    # it always passes (fold) - in approximate-key mode it never does (acc0 = -2^30, as on padding rows); it does not enter the maxima
    forced = ~(np.abs(a0) < 536870912.0)
    acc0 = np.where(forced, -(1 << 30), a0).astype(np.int64)
    erow = np.where(forced, F(np.inf), erow).astype(F)
    ok = ~forced
    xn = (X * X).sum(1, dtype=F)
    mx = lambda v: F(v[ok].max()) if ok.any() else F(0.0)
    scal = dict(e1max=mx(erow), nxhmax=mx(hrow), xnmax=mx(xn), rmax=mx(np.abs(R)), mun=F(np.sqrt((mu * mu).sum(dtype=F)) * F(1.00001)))
    scal["xcmax"] = F(scal["nxhmax"] + scal["e1max"])        # >= max |x - mu| of the rows that are tested
    return dict(mu=mu, step=step, inv=inv, xi=xi, acc0=acc0, u=u, s=s, scal=scal, erow=erow, hrow=hrow, forced=forced)


def fold(m, qss):
    """fold8_kernel: the margin of every ROW for a whole batch of queries, added to the row's accumulator start value - its own two
    norms times the batch's LARGEST query norms (>= every query's own margin for that row) - so that the thresholds carry none"""
    qn = F(max(q["nq"] for q in qss))
    eq = F(max(q["eq"] for q in qss))
    with np.errstate(invalid="ignore", over="ignore"):
        marg = (m["s"] * (qn * m["erow"] + eq * m["hrow"])).astype(F)
        add = np.ceil(marg / m["u"]) + 1
    force = m["forced"] | ~(add < 536870912.0)
    return np.where(force, ACC_FORCE, m["acc0"] + np.where(force, 0, add).astype(np.int64))


def query(q, m, metric):
    qc = (q - m["mu"]).astype(F)
    qi = np.clip(np.rint(qc * m["inv"]), -127, 127).astype(np.int32)
    res = (qc - m["step"] * qi.astype(F)).astype(F)
    s2c = F((qc * qc).sum(dtype=F))
    s2 = F((q * q).sum(dtype=F))
    qmu = F((q * m["mu"]).sum(dtype=F))
    Cq = s2c if metric == 0 else ((F(1.0) - qmu) if metric == 1 else -qmu)
    return qi, dict(qn2=s2, nq=F(np.sqrt(s2c) * F(1.000001)), eq=F(np.sqrt((res * res).sum(dtype=F)) * F(1.00001) + F(1.2e-7) * np.sqrt(s2c)), Cq=F(Cq))


def threshold(thr, qs, m, metric, slack, folded=False):
    sc = m["scal"]
    margin = F(0.0) if folded else m["s"] * (qs["nq"] * sc["e1max"] + qs["eq"] * sc["nxhmax"])
    if metric == 0:
        scale = abs(thr) + F(2.0) * abs(qs["Cq"]) + F(2.0) * sc["rmax"]
    else:
        qn = F(np.sqrt(qs["qn2"]))
        scale = abs(thr) + F(1.0) + qn * (F(np.sqrt(sc["xnmax"])) + sc["mun"]) + sc["mun"] * sc["xcmax"] + abs(qs["Cq"]) + sc["rmax"]
    t = thr + margin + F(slack) * scale + F(4.0) * m["u"]
    return int(np.clip(np.floor((qs["Cq"] - t) / m["u"]) - 2, -(1 << 30), TQ_MAX))


def dist(q, X, metric):
    if metric == 0:
        return ((X - q) ** 2).sum(1, dtype=F)
    d = (X * q).sum(1, dtype=F)
    return (F(1.0) - d) if metric == 1 else -d


def slack_of(d):
    return max(8e-6, 2.0 * (3.0 * (d / 64.0 + 6.0) + 6.0) * 5.9604645e-8)


CASES = {
    "uniform": lambda r, n, d: r.random((n, d), dtype=F),
    "gaussian": lambda r, n, d: r.standard_normal((n, d)).astype(F),
    "far from the origin": lambda r, n, d: (1000.0 + r.random((n, d))).astype(F),
    "tiny range": lambda r, n, d: (0.5 + 1e-4 * r.random((n, d))).astype(F),
    "heavy tail": lambda r, n, d: (r.standard_normal((n, d)) * np.exp(r.standard_normal((n, 1)))).astype(F),
    "columns with their own means": lambda r, n, d: (r.standard_normal((1, d)) * 5.0 + 0.3 * r.standard_normal((n, d))).astype(F),
    "a few outlier values": lambda r, n, d: _with_outliers(r, r.random((n, d), dtype=F)),
}


def _with_outliers(r, X):
    for _ in range(5):
        X[r.integers(len(X)), r.integers(X.shape[1])] = F(r.choice([-1.0, 1.0]) * r.uniform(20.0, 200.0))
    return X


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("case", sorted(CASES))
def test_rows_within_the_threshold_always_pass_the_8bit_test(case, metric):
    """both forms of the test: thresholds that carry the table-wide margin (the build's and any caller's that does not fold), and r4's
    per-row margins folded into the accumulators' start values for the batch (fold) with margin-free thresholds"""
    rng = np.random.default_rng(abs(hash((case, metric))) % (1 << 31))
    n, d = 4000, 96
    X = CASES[case](rng, n, d)
    if metric == 1:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    mu, half = col_centre(X, sample=slice(0, None, 7))
    m = mirror(X, metric, mu=mu, step=half / F(127.0))
    # rows appended after the grid was fixed, some far outside it: quantised with clamped codes, their residual enters the bound
    span = F(np.percentile(X, 99.9) - np.percentile(X, 0.1))
    Xa = np.concatenate([X[:200] + F(0.2) * span * np.sign(rng.standard_normal((200, d))).astype(F), X[200:400]])   # up to 20 % of the range outside
    if metric == 1:
        Xa /= np.linalg.norm(Xa, axis=1, keepdims=True)
    m2 = mirror(np.concatenate([X, Xa]), metric, mu=m["mu"], step=m["step"])
    slack = slack_of(d)
    worst = 1 << 40
    if m["forced"].mean() > 0.01:     # the device declines a table most of whose row constants leave int32 (the fp16 pass serves it)
        assert metric != 0 and case in ("far from the origin", "tiny range")   # (mu large against the step: |R| / u = |mu . x'| / step^2)
        pytest.skip("row constants beyond int32: no 8-bit mirror for this table")
    for mm, rows in ((m, X), (m2, np.concatenate([X, Xa]))):
        qs_all, per_q = [], []
        for qk in range(12):
            q = rows[rng.integers(len(rows))] + F(0.05) * rng.standard_normal(d).astype(F) if qk % 3 else CASES[case](rng, 1, d)[0] * F(3.0) - F(1.0)
            if metric == 1:
                q = q / np.linalg.norm(q)
            q = q.astype(F)
            qi, qs = query(q, mm, metric)
            qs_all.append(qs)
            per_q.append((q, qi, qs))
        acc0f = fold(mm, qs_all)            # ONE fold for the batch of 12 queries
        for q, qi, qs in per_q:
            dd = dist(q, rows, metric)
            dot = mm["xi"].astype(np.int64) @ qi.astype(np.int64)
            for frac in (0.001, 0.02, 0.3):
                thr = F(np.partition(dd, int(frac * len(dd)))[int(frac * len(dd))])
                inside = dd <= thr
                Tq = threshold(thr, qs, mm, metric, slack)   # (forced rows are only ever tested in the folded form)
                assert ((dot + mm["acc0"])[inside & ~mm["forced"]] >= Tq).all(), (case, metric, frac, "table-wide margin")
                Tf = threshold(thr, qs, mm, metric, slack, folded=True)
                assert ((dot + acc0f)[inside] >= Tf).all(), (case, metric, frac, "folded per-row margins")
                worst = min(worst, int(((dot + acc0f)[inside] - Tf).min()))
    assert worst >= 0


def test_any_centre_keeps_the_bound_valid():
    """mu only has to be the same vector for rows and queries: a useless one (random) costs tightness, never exactness"""
    rng = np.random.default_rng(11)
    n, d = 3000, 64
    X = rng.random((n, d), dtype=F)
    for metric in (0, 1, 2):
        for mu in (np.zeros(d, F), rng.standard_normal(d).astype(F), np.full(d, 0.5, F)):
            m = mirror(X, metric, mu=mu)
            if m["forced"].any():
                continue
            for _ in range(6):
                q = rng.random(d, dtype=F)
                qi, qs = query(q, m, metric)
                dd = dist(q, X, metric)
                lhs = m["xi"].astype(np.int64) @ qi.astype(np.int64) + m["acc0"]
                thr = F(np.partition(dd, 30)[30])
                assert (lhs[dd <= thr] >= threshold(thr, qs, m, metric, slack_of(d))).all()


def test_the_bound_is_not_vacuous_on_uniform_rows():
    """... and it is worth something: on U[0,1) rows at d = 768 the test rejects most rows beyond a top-5 % threshold; the centred
    grid (r4) passes far fewer rows than the grid with one zero for all columns did (r3: 0.35 was the bar here)"""
    rng = np.random.default_rng(3)
    n, d = 3000, 768
    X = rng.random((n, d), dtype=F)
    m = mirror(X, 0)
    q = rng.random(d, dtype=F)
    qi, qs = query(q, m, 0)
    dd = dist(q, X, 0)
    thr = F(np.partition(dd, n // 20)[n // 20])
    Tq = threshold(thr, qs, m, 0, slack_of(d))
    lhs = m["xi"].astype(np.int64) @ qi.astype(np.int64) + m["acc0"]
    passed = (lhs >= Tq).mean()
    assert (lhs[dd <= thr] >= Tq).all() and passed < 0.16, passed
    # the margin itself: 2 (|q'| e1 + |eq| |xh'|) ~ 1.0 key units where the uncentred grid had ~2.0
    margin = m["s"] * (qs["nq"] * m["scal"]["e1max"] + qs["eq"] * m["scal"]["nxhmax"])
    assert margin < 1.2, margin


def test_one_outlier_value_costs_its_row_not_the_table():
    """r4: the clipped grid + per-row margins.  One value of 100 in a U[0,1) table stretched r3's / early r4's grid 100 x and the
    table-wide margin with it (the 8-bit pass could not filter: the fp16 pass served the table).  Now the grid is set by the bulk of
    the values, the outlier's row is clamped, carries its own large residual and simply always passes; every other row is filtered as
    if the outlier were not there."""
    rng = np.random.default_rng(4)
    n, d = 3000, 768
    X = rng.random((n, d), dtype=F)
    clean = mirror(X, 0)
    X[1234, 5] = F(100.0)
    m = mirror(X, 0)
    assert m["step"] < F(1.05) * clean["step"]                       # the grid did not stretch
    assert not m["forced"].any() and m["erow"][1234] > 50 and np.delete(m["erow"], 1234).max() < 2 * clean["erow"].max()
    q = rng.random(d, dtype=F)
    qi, qs = query(q, m, 0)
    dd = dist(q, X, 0)
    thr = F(np.partition(dd, n // 20)[n // 20])
    lhs = m["xi"].astype(np.int64) @ qi.astype(np.int64) + fold(m, [qs])
    Tf = threshold(thr, qs, m, 0, slack_of(d), folded=True)
    passed = (lhs >= Tf).mean()
    assert (lhs[dd <= thr] >= Tf).all() and passed < 0.16, passed   # (the outlier row itself is provably far: it does not pass)
    # ... where the table-wide margin is useless on the same mirror
    Tq = threshold(thr, qs, m, 0, slack_of(d))
    X2 = rng.random((n, d), dtype=F)
    X2[1234, 5] = F(1000.0)         # so far out that its row constant leaves the accumulator's range: the row is FORCED (always passes)
    m3 = mirror(X2, 0)
    assert m3["forced"][1234] and m3["forced"].sum() == 1 and m3["step"] < F(1.05) * clean["step"]   # (the range histogram refines itself)
    X4 = rng.random((n, d), dtype=F)
    X4[7, 7] = F(30000.0)
    assert mirror(X4, 0)["step"] < F(1.05) * clean["step"]
    qi3, qs3 = query(q, m3, 0)
    dd3 = dist(q, X2, 0)
    thr3 = F(np.partition(dd3, n // 20)[n // 20])
    dot3 = m3["xi"].astype(np.int64) @ qi3.astype(np.int64)
    folded = (dot3 + fold(m3, [qs3]) >= threshold(thr3, qs3, m3, 0, slack_of(d), folded=True))
    tablewide = (dot3 + m3["acc0"] >= threshold(thr3, qs3, m3, 0, slack_of(d)))
    assert folded[dd3 <= thr3].all() and folded[1234] and folded.mean() < 0.16 and tablewide[(dd3 <= thr3) & ~m3["forced"]].all()


def upper_bound(acc, qs, m, metric, slack):
    """stream8_ub (vectordb_amd/csrc/stream8_kernel.hpp): the exact fp32 distance of a row whose accumulator is `acc` is AT MOST this - the
    Cauchy-Schwarz margin bounds |exact - approximate| on both sides.  The one-pass search of a handful of queries keeps the accumulators
    of k rows and uses the largest of their upper bounds where the staged chain uses the k-th best exact key."""
    sc = m["scal"]
    margin = m["s"] * (qs["nq"] * sc["e1max"] + qs["eq"] * sc["nxhmax"])
    dapx = F(qs["Cq"] - m["u"] * F(acc))
    if metric == 0:
        scale = abs(dapx) + margin + F(2.0) * abs(qs["Cq"]) + F(2.0) * sc["rmax"]
    else:
        qn = F(np.sqrt(qs["qn2"]))
        scale = abs(dapx) + margin + F(1.0) + qn * (F(np.sqrt(sc["xnmax"])) + sc["mun"]) + sc["mun"] * sc["xcmax"] + abs(qs["Cq"]) + sc["rmax"]
    return F(dapx + margin + F(2.0) * F(slack) * scale + F(4.0) * m["u"])


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("case", sorted(CASES))
def test_upper_bound_of_the_approximate_key(case, metric):
    """(i) every tested row's exact distance is at most the upper bound of its accumulator; (ii) the consequence the one-pass search relies
    on: with T = threshold(max upper bound of ANY k rows), every row of the exact top-k passes acc >= T."""
    rng = np.random.default_rng(abs(hash((case, metric, "ub"))) % (1 << 31))
    n, d, k = 3000, 96, 10
    X = CASES[case](rng, n, d)
    if metric == 1:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    mu, half = col_centre(X, sample=slice(0, None, 5))
    m = mirror(X, metric, mu=mu, step=half / F(127.0))
    if m["forced"].mean() > 0.01:
        pytest.skip("row constants beyond int32: no 8-bit mirror for this table")
    slack = slack_of(d)
    ok = ~m["forced"]
    for qk in range(10):
        q = X[rng.integers(n)] + F(0.05) * rng.standard_normal(d).astype(F) if qk % 3 else CASES[case](rng, 1, d)[0] * F(2.0) - F(0.5)
        if metric == 1:
            q = q / np.linalg.norm(q)
        q = q.astype(F)
        qi, qs = query(q, m, metric)
        acc = (m["xi"].astype(np.int64) @ qi.astype(np.int64)) + m["acc0"]
        exact = dist(q, X, metric)
        ub = np.array([upper_bound(int(a), qs, m, metric, slack) for a in acc[ok]], dtype=F)
        assert (exact[ok] <= ub).all(), (case, metric, float((exact[ok] - ub).max()))
        # any k tested rows (here: k random ones, and the k with the largest accumulators) bound the k-th best exact distance
        order = np.argsort(exact, kind="stable")
        for pick in (rng.choice(np.flatnonzero(ok), k, replace=False), np.flatnonzero(ok)[np.argsort(-acc[ok], kind="stable")[:k]]):
            thr = max(upper_bound(int(a), qs, m, metric, slack) for a in acc[pick])
            assert thr >= exact[order[k - 1]] or not ok[order[:k]].all()
            T = threshold(thr, qs, m, metric, slack)
            top = order[:k]
            assert (acc[top][ok[top]] >= T).all(), (case, metric)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 10, 16, 17, 40, 64])
def test_one_pass_table_never_loses_a_row_of_the_answer(metric, k):
    """The one-pass search's table as stream8_kernel.hpp keeps it: 64 slots (r5: 128 for k = 17..64), slot = ((row * 2654435761) >> 12) & 63
    (& 127) holds the largest accumulator among the VISIBLE rows hashed to it; the pass threshold = threshold(upper bound of the k-th largest slot).  Whatever
    subset of the rows has been offered when a row is tested (the table only tightens: here the empty table, a tenth of the rows, all of
    them), every visible row of the exact top-k passes - and against the final table not many more rows do than against the threshold of
    the exact k-th distance itself."""
    rng = np.random.default_rng(1000 * metric + k)
    n, d = 6000, 96
    X = rng.random((n, d), dtype=F)
    X[100:130] = X[99]                                   # a run of identical rows
    if metric == 1:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    mu, half = col_centre(X, sample=slice(0, None, 3))
    m = mirror(X, metric, mu=mu, step=half / F(127.0))
    slack = slack_of(d)
    visible = rng.random(n) > 0.2                        # a deleted bitset
    slots = 64 if k <= 16 else 128
    slot = ((np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(12) & np.uint64(slots - 1)
    for qk in range(6):
        q = (X[105] if qk == 0 else X[rng.integers(n)] + F(0.05) * rng.standard_normal(d).astype(F)).astype(F)
        if metric == 1:
            q = (q / np.linalg.norm(q)).astype(F)
        qi, qs = query(q, m, metric)
        acc = (m["xi"].astype(np.int64) @ qi.astype(np.int64)) + m["acc0"]
        exact = dist(q, X, metric)
        order = [i for i in np.lexsort((np.arange(n), exact)) if visible[i]][:k]     # the answer: (distance, id) order over visible rows
        for offered in (np.zeros(n, bool), (np.arange(n) % 10 == 0), np.ones(n, bool)):
            table = np.full(slots, -(1 << 31), dtype=np.int64)
            for i in np.flatnonzero(offered & visible):
                table[slot[i]] = max(table[slot[i]], acc[i])
            kth = np.sort(table)[::-1][k - 1]
            T = -(1 << 30) if kth == -(1 << 31) else threshold(upper_bound(int(kth), qs, m, metric, slack), qs, m, metric, slack)
            assert (acc[order] >= T).all(), (metric, k, qk, int(offered.sum()))
        # (T of the full table) how loose the table's threshold is against the best any method could use - the exact k-th distance
        passed = int((acc[visible] >= T).sum())
        best = int((acc[visible] >= threshold(exact[order[-1]], qs, m, metric, slack)).sum())
        # (the table's threshold carries the margin twice - once up to the bound, once down to the test; on this toy table of normalised
        # 96-d rows the margin is most of the spread of the distances, so only the L2 case is held to a number)
        assert metric != 0 or passed <= 4 * best + 40 * k + 60, (metric, k, qk, passed, best)


@pytest.mark.parametrize("k,slots", [(10, 64), (16, 64), (40, 128), (64, 128)])
def test_start_up_wait_reads_a_median_offer_not_the_worst(k, slots):
    """Why a workgroup of the one-pass search waits for max(2k, k + 8) filled table slots (capped at 7/8 of the table) before it reads its first
    thresholds (stream8_startup_need, csrc/stream8_kernel.hpp), not for k: every wavefront offers the best of its first 16 rows, the offers land in
    hash slots in no particular order, and the threshold is the k-th largest slot.  With EXACTLY k slots filled that is the worst row offered so far -
    once in a few thousand calls a below-median row of the table, which lets most of a small table through (the stress loop that found it:
    scripts/lab/one_pass_small_table_stress.py).  Simulated here: the quantile (0 = the best row of the table) of the row the first read would
    take for the k-th best, over 4000 start-ups."""
    rng = np.random.default_rng(k * 1000 + slots)
    want = max(2 * k, k + 8)
    need = min(want, slots - slots // 8)
    assert k <= need <= slots
    worst_old, worst_new = 0.0, 0.0
    at_old, at_new = [], []
    for trial in range(4000):
        n_off = 2048
        quant = rng.random((n_off, 16)).min(axis=1)          # the best of 16 rows, as a quantile of the table
        slot = rng.integers(0, slots, n_off)
        table = np.full(slots, np.inf)
        filled = 0
        seen_old = None
        for i in range(n_off):
            if table[slot[i]] == np.inf:
                filled += 1
            table[slot[i]] = min(table[slot[i]], quant[i])
            if seen_old is None and filled >= k:
                seen_old = np.sort(table)[k - 1]
            if filled >= need:
                at_new.append(np.sort(table)[k - 1])
                break
        at_old.append(seen_old)
    at_old, at_new = np.array(at_old), np.array(at_new)
    # the old rule's tail reaches rows a third of the way down the table and further; the new rule's stays in the best tenth, every time
    assert at_old.max() > 0.3, at_old.max()
    assert at_new.max() < 0.12, at_new.max()
    assert np.median(at_new) < 0.5 * np.median(at_old)


# ------------------------------------------------------------------------------------------------ r6: the grid in a rotated frame
# (device_common.hpp rot256_load, mfma_filter.hip ensure_mirror8).  Rows and queries are quantised as y = R x, R = blockdiag(H_256 / 16) . S . P
# (a fixed permutation of the zero-padded columns, signs, a 256-point Walsh-Hadamard transform per 256-column block): R is exactly orthogonal,
# distances do not change, and every y column is a signed mean of 256 values of the row - no column dominates, the step shrinks to what the
# row's norm needs.  The transform runs in fp64, y - mu is rounded to fp32 once.
def d_pad8_of(d):
    """the rotation's width: the row's columns rounded up to whole 256-column blocks (the mirror's row pitch is at least 512 bytes: columns beyond
    the rotation's width stay zero and enter nothing)"""
    return (d + 255) // 256 * 256


def rotation_table(d_pad8):
    """the library's fixed sequence (splitmix64 seeded with the width): source column of every rotated-input position, and its sign"""
    M = (1 << 64) - 1
    st = [0x9E3779B97F4A7C15 ^ d_pad8]

    def nxt():
        st[0] = (st[0] + 0x9E3779B97F4A7C15) & M
        z = st[0]
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    sp = list(range(d_pad8))
    for i in range(d_pad8 - 1, 0, -1):
        j = nxt() % (i + 1)
        sp[i], sp[j] = sp[j], sp[i]
    neg = [bool(nxt() & 1) for _ in range(d_pad8)]
    return np.array(sp), np.array(neg)


def rotate_rows(X):
    """fp64 image R x of every row, [n][d_pad8]: the butterflies in the kernel's order (two levels inside a lane's four values, six across lanes)"""
    n, d = X.shape
    dp = d_pad8_of(d)
    src, neg = rotation_table(dp)
    Z = np.zeros((n, dp), np.float64)
    Z[:, :d] = X
    Y = Z[:, src] * np.where(neg, -1.0, 1.0)
    Y = Y.reshape(n, dp // 256, 256)
    h = 1
    while h < 256:
        Y = Y.reshape(n, dp // 256, 256 // (2 * h), 2, h)
        Y = np.stack([Y[:, :, :, 0, :] + Y[:, :, :, 1, :], Y[:, :, :, 0, :] - Y[:, :, :, 1, :]], axis=3)
        Y = Y.reshape(n, dp // 256, 256)
        h *= 2
    return (Y * 0.0625).reshape(n, dp)


def mirror_rot(X, metric, mu=None, step=None):
    """mirror() in the rotated frame: x' = fl32(R x - mu) (ONE rounding), everything else as before; + 1e-12 |x| for the fp64 transform"""
    Y = rotate_rows(X)
    Y32 = Y.astype(F)
    if mu is None:
        mu, half = col_centre(Y32)
        if step is None:
            step = half / F(127.0)
    xc = (Y - mu.astype(np.float64)).astype(F)
    inv = F(1.0) / step
    xi = np.clip(np.rint(xc * inv), -127, 127).astype(np.int32)
    res = (xc - step * xi.astype(F)).astype(F)
    xh = (step * xi.astype(F)).astype(F)
    s = F(2.0) if metric == 0 else F(1.0)
    u = s * step * step
    x2c = (xc * xc).sum(1, dtype=F)
    R = x2c if metric == 0 else -(mu * xc).sum(1, dtype=F)
    a0 = np.ceil(-R / u) + 1
    xn = (Y32 * Y32).sum(1, dtype=F)
    erow = (np.sqrt((res * res).sum(1, dtype=F)) * F(1.00001) + F(1.2e-7) * np.sqrt(x2c) + F(1e-12) * np.sqrt(xn)).astype(F)
    hrow = (np.sqrt((xh * xh).sum(1, dtype=F)) * F(1.00001)).astype(F)
    forced = ~(np.abs(a0) < 536870912.0)
    acc0 = np.where(forced, -(1 << 30), a0).astype(np.int64)
    erow = np.where(forced, F(np.inf), erow).astype(F)
    ok = ~forced
    mx = lambda v: F(v[ok].max()) if ok.any() else F(0.0)
    scal = dict(e1max=mx(erow), nxhmax=mx(hrow), xnmax=mx(xn), rmax=mx(np.abs(R)), mun=F(np.sqrt((mu * mu).sum(dtype=F)) * F(1.00001)))
    scal["xcmax"] = F(scal["nxhmax"] + scal["e1max"])
    return dict(mu=mu, step=step, inv=inv, xi=xi, acc0=acc0, u=u, s=s, scal=scal, erow=erow, hrow=hrow, forced=forced)


def query_rot(q, m, metric):
    y = rotate_rows(q[None, :])[0]
    y32 = y.astype(F)
    qc = (y - m["mu"].astype(np.float64)).astype(F)
    qi = np.clip(np.rint(qc * m["inv"]), -127, 127).astype(np.int32)
    res = (qc - m["step"] * qi.astype(F)).astype(F)
    s2c = F((qc * qc).sum(dtype=F))
    s2 = F((y32 * y32).sum(dtype=F))
    qmu = F((y32 * m["mu"]).sum(dtype=F))
    Cq = s2c if metric == 0 else ((F(1.0) - qmu) if metric == 1 else -qmu)
    return qi, dict(qn2=s2, nq=F(np.sqrt(s2c) * F(1.000001)),
                    eq=F(np.sqrt((res * res).sum(dtype=F)) * F(1.00001) + F(1.2e-7) * np.sqrt(s2c) + F(1e-12) * np.sqrt(s2)), Cq=F(Cq))


def test_the_rotation_is_orthogonal_and_spreads_every_column():
    rng = np.random.default_rng(21)
    for d in (33, 256, 300, 768, 1000):
        X = rng.standard_normal((40, d))
        Y = rotate_rows(X)
        assert Y.shape[1] == d_pad8_of(d)
        assert np.allclose(Y @ Y.T, X @ X.T, rtol=0, atol=1e-11 * d)                # R^T R = I: every inner product survives
        src, neg = rotation_table(d_pad8_of(d))
        assert sorted(src.tolist()) == list(range(d_pad8_of(d))) and 0.3 < neg.mean() < 0.7
        e = np.zeros((1, d))
        e[0, d // 2] = 1.0                                                          # one column's energy lands on exactly one 256-block, evenly
        y = rotate_rows(e)[0]
        assert np.count_nonzero(y) == 256 and np.allclose(np.abs(y[y != 0]), 1.0 / 16.0)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("case", sorted(CASES))
def test_rows_within_the_threshold_always_pass_the_8bit_test_in_the_rotated_frame(case, metric):
    """the implication every user of the mirror relies on, for a table quantised in the rotated frame: same cases, same two forms of the test
    (table-wide margin in the threshold; per-row margins folded into the start values), rows appended outside the grid included"""
    rng = np.random.default_rng(abs(hash((case, metric, "rot"))) % (1 << 31))
    n, d = 2500, 96
    X = CASES[case](rng, n, d)
    if metric == 1:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    m = mirror_rot(X, metric)
    span = F(np.percentile(X, 99.9) - np.percentile(X, 0.1))
    Xa = np.concatenate([X[:150] + F(0.2) * span * np.sign(rng.standard_normal((150, d))).astype(F), X[150:300]])
    if metric == 1:
        Xa /= np.linalg.norm(Xa, axis=1, keepdims=True)
    rows2 = np.concatenate([X, Xa])
    m2 = mirror_rot(rows2, metric, mu=m["mu"], step=m["step"])
    slack = slack_of(d_pad8_of(d))
    if m["forced"].mean() > 0.01:
        pytest.skip("row constants beyond int32: no 8-bit mirror for this table")
    worst = 1 << 40
    for mm, rows in ((m, X), (m2, rows2)):
        per_q = []
        for qk in range(9):
            q = rows[rng.integers(len(rows))] + F(0.05) * rng.standard_normal(d).astype(F) if qk % 3 else CASES[case](rng, 1, d)[0] * F(3.0) - F(1.0)
            if metric == 1:
                q = q / np.linalg.norm(q)
            q = q.astype(F)
            per_q.append((q,) + query_rot(q, mm, metric))
        acc0f = fold(mm, [qs for _, _, qs in per_q])
        for q, qi, qs in per_q:
            dd = dist(q, rows, metric)                    # the distance in the ORIGINAL frame, fp32: what the re-rank computes
            dot = mm["xi"].astype(np.int64) @ qi.astype(np.int64)
            for frac in (0.001, 0.02, 0.3):
                thr = F(np.partition(dd, int(frac * len(dd)))[int(frac * len(dd))])
                inside = dd <= thr
                Tq = threshold(thr, qs, mm, metric, slack)
                assert ((dot + mm["acc0"])[inside & ~mm["forced"]] >= Tq).all(), (case, metric, frac, "table-wide margin")
                Tf = threshold(thr, qs, mm, metric, slack, folded=True)
                assert ((dot + acc0f)[inside] >= Tf).all(), (case, metric, frac, "folded per-row margins")
                worst = min(worst, int(((dot + acc0f)[inside] - Tf).min()))
    assert worst >= 0


def _embedding_like(rng, n, d):
    scale = np.ones(d, F)
    scale[:8] = 4.0
    X = rng.standard_normal((n, d)).astype(F) * scale
    return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(F)


def test_the_rotated_frame_is_what_embedding_like_rows_need_and_uniform_rows_do_not():
    """bench.py's embedding-like set (unit-norm rows, 8 dominant dimensions of 768), COSINE.  In the identity frame the dominant columns set
    the step for all 768 and the margin equals the spread of the distances (r5: the fp16 pass served such tables at half the matrix rate); in
    the rotated frame the step is a third, and an order of magnitude fewer rows survive the same threshold.  U[0,1) rows fill the identity
    grid evenly: there the rotated frame's step is ~2.9 x LARGER.  The library's rule (rotated iff step_r sqrt(d_pad8) < 0.75 step_i sqrt(d))
    separates the two."""
    rng = np.random.default_rng(31)
    n, d, k = 20_000, 768, 10
    X = _embedding_like(rng, n, d)
    Q = _embedding_like(rng, 6, d)
    mi = mirror(X, 1)
    mr = mirror_rot(X, 1)
    assert mr["step"] * np.sqrt(d_pad8_of(d)) < 0.5 * mi["step"] * np.sqrt(d)
    assert mr["erow"].mean() < 0.45 * mi["erow"].mean()
    passed = {"identity": 0, "rotated": 0}
    for q in Q:
        dd = dist(q, X, 1)
        thr = F(np.partition(dd, k - 1)[k - 1])
        for name, mm, (qi, qs) in (("identity", mi, query(q, mi, 1)), ("rotated", mr, query_rot(q, mr, 1))):
            lhs = mm["xi"].astype(np.int64) @ qi.astype(np.int64) + mm["acc0"]
            Tq = threshold(thr, qs, mm, 1, slack_of(d))
            assert (lhs[dd <= thr] >= Tq).all()
            passed[name] += int((lhs >= Tq).sum())
    assert passed["rotated"] * 5 < passed["identity"], passed
    U = rng.random((4000, d), dtype=F)
    ui, ur = mirror(U, 0), mirror_rot(U, 0)
    assert ur["step"] * np.sqrt(d_pad8_of(d)) > 2.0 * ui["step"] * np.sqrt(d)
