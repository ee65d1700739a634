"""Seeded random configurations: whatever engine FLAT_AUTO picks (stream scan, MFMA filter v7<1>/v7<2>/v3, in slices or
not, seeded or not, with or without a deleted bitset / attribute filter) must return the stream engine's bits."""
import numpy as np
import pytest

from helpers import data


@pytest.fixture(scope="module")
def amd():
    import vectordb_amd
    from vectordb_amd.build import build
    build()
    return vectordb_amd


def _configs():
    rng = np.random.default_rng(20240924)
    dims = [64, 96, 128, 200, 256, 320, 384, 512, 640, 1000]
    out = []
    for i in range(18):
        out.append(dict(n=int(rng.integers(70_000, 260_000)), d=int(rng.choice(dims)), nq=int(rng.choice([1, 5, 9, 40, 129, 300, 600, 2100])),
                        k=int(rng.choice([1, 5, 10, 37, 100, 128])), metric=int(rng.integers(0, 3)),
                        flt=str(rng.choice(["none", "none", "deleted50", "deleted90", "int"])), seed=1000 + i))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("c", _configs(), ids=lambda c: "n%d-d%d-q%d-k%d-m%d-%s" % (c["n"], c["d"], c["nq"], c["k"], c["metric"], c["flt"]))
def test_auto_engine_equals_stream_engine(amd, c):
    rng = np.random.default_rng(c["seed"])
    X, Q = data(c["n"], c["d"], c["seed"]), data(c["nq"], c["d"], c["seed"] + 1)
    if c["metric"] == 1:
        X = amd.normalize_rows(X, only_if_nonzero=True)
        Q = amd.normalize_rows(Q, only_if_nonzero=False)
    ix = amd.GpuIndex(c["d"], c["metric"])
    ix.attach_rows(X)
    vis = np.ones(c["n"], bool)
    if c["flt"].startswith("deleted"):
        dele = rng.random(c["n"]) < (0.5 if c["flt"] == "deleted50" else 0.9)
        ix.set_deleted(np.packbits(dele, bitorder="little"))
        vis &= ~dele
    elif c["flt"] == "int":
        col = rng.integers(-1000, 1000, c["n"]).astype(np.int32)
        ix.set_int_filter(col, "<", 250)
        vis &= col < 250
    a = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_AUTO)
    b = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_STREAM)
    m = ix.search(Q, c["k"], mode=amd.MODE_FLAT, flat_engine=amd.FLAT_MFMA)
    for got in (a, m):
        assert np.array_equal(got[0], b[0]), "%d result ids differ" % (got[0] != b[0]).sum()
        assert np.array_equal(got[1], b[1]) and np.array_equal(got[2], b[2])
    ids = b[0][b[0] >= 0]
    assert vis[ids].all() and (b[2] == min(c["k"], int(vis.sum()))).all()
    ix.close()
