"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly what include/*.h
declares; the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from vectordb_amd.build import build
    return build()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "epsilla_gfx950.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(eps_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(built)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from vectordb_amd import _lib
    assert set(_lib.EXPORTS) == declared


def test_defaults_mirror_reference_config(built):
    from vectordb_amd import _lib
    L = _lib.load()
    p = _lib.SearchParams()
    L.eps_default_search_params(ctypes.byref(p))
    # config/config.hpp:17-25
    assert (p.intra_threads, p.master_queue, p.local_queue, p.sync_interval, p.prefilter) == (4, 500, 500, 15, 0)
    b = _lib.BuildParams()
    L.eps_default_build_params(ctypes.byref(b))
    # NSGConfig(45, 50, 300, 100), db/ann_graph_segment.cpp:29
    assert (b.search_length, b.out_degree, b.candidate_pool_size, b.knng, b.seed) == (45, 50, 300, 100, 100)


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vectordb_amd
    with pytest.raises(vectordb_amd.EpsillaError) as e:
        vectordb_amd.GpuIndex(16)
    assert e.value.code == 40001


def test_product_never_imports_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "vectordb_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle|libepsilla_oracle|libepsilla_ref", txt, re.M):
                    bad.append(f)
    assert not bad, bad


def test_python_graph_file_roundtrip(tmp_path, oracle):
    """vectordb_amd.ANNGraphSegment reads/writes the reference's ann_graph_<field>.bin layout."""
    import numpy as np
    from vectordb_amd import ANNGraphSegment
    seg = ANNGraphSegment(skip_sync_disk=False)
    seg.record_number_ = 3
    seg.offset_table_ = np.array([0, 2, 3, 5], np.int64)
    seg.neighbor_list_ = np.array([1, 2, 0, 0, 1], np.int64)
    seg.navigation_point_ = 2
    os.makedirs(tmp_path / "5")
    assert seg.SaveANNGraph(str(tmp_path), 5, 1) == 0
    off, nbr, nav, fid = oracle.graph_read(str(tmp_path / "5" / "ann_graph_1.bin"))
    assert nav == 2 and fid == 0 and list(off) == [0, 2, 3, 5] and list(nbr) == [1, 2, 0, 0, 1]
    seg2 = ANNGraphSegment(str(tmp_path), 5, 1)
    assert seg2.record_number_ == 3 and seg2.navigation_point_ == 2
    assert np.array_equal(seg2.offset_table_, off) and np.array_equal(seg2.neighbor_list_, nbr)


def test_public_header_is_plain_c_and_struct_layouts_match_ctypes(tmp_path, built):
    """include/epsilla_gfx950.h is what a cgo / JNI / N-API binding would include: it must compile as C99 without torch or C++,
    and the ctypes mirrors in vectordb_amd/_lib.py (what the Python side passes by pointer) must have the C sizes and offsets."""
    import ctypes as C
    import shutil
    import subprocess
    from vectordb_amd import _lib
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("needs gcc")
    src = tmp_path / "t.c"
    src.write_text("""
#include <stdio.h>
#include <stddef.h>
#include "epsilla_gfx950.h"
int main(void) {
  printf("search %zu build %zu filter_op %zu %zu %zu %zu layout %zu\\n", sizeof(eps_search_params), sizeof(eps_build_params), sizeof(eps_filter_op),
         offsetof(eps_filter_op, arg), offsetof(eps_filter_op, ival), offsetof(eps_filter_op, dval), sizeof(eps_table_layout));
  return 0;
}
""")
    exe = tmp_path / "t"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()
    c = dict(search=int(out[1]), build=int(out[3]), fop=[int(x) for x in out[5:9]], layout=int(out[10]))
    assert C.sizeof(_lib.SearchParams) == c["search"] and C.sizeof(_lib.BuildParams) == c["build"]
    assert [C.sizeof(_lib.FilterOp), _lib.FilterOp.arg.offset, _lib.FilterOp.ival.offset, _lib.FilterOp.dval.offset] == c["fop"]
    assert C.sizeof(_lib.TableLayout) == c["layout"]



def test_traversal_gather_bytes_accounting():
    """bench.py / scripts/bench_graph.py price the traversal kernel with this: without the 8-bit prefilter every evaluation reads
    its fp32 row; with it (rerank_rows = fp32 rows read in step d) seeds and survivors read 4d bytes, every neighbour evaluation
    the mirror row (d rounded up to 16) + its 4-byte constant.  The two forms agree when every neighbour passes the prefilter,
    up to the mirror bytes themselves."""
    import vectordb_amd as amd
    d, deg, seeds = 768, 50.0, 500 * 1024
    off = {"dist_evals": 32_000 * 1024, "expansions": 590 * 1024, "rerank_rows": 0}
    assert amd.traversal_gather_bytes(off, d, deg, seeds) == off["dist_evals"] * (4.0 * d + 4) + off["expansions"] * (8 + 4 * deg)
    on = dict(off, rerank_rows=5_700 * 1024)
    want = (seeds + on["rerank_rows"]) * 4.0 * d + (on["dist_evals"] - seeds) * (768 + 4.0) + on["dist_evals"] * 4 + on["expansions"] * (8 + 4 * deg)
    assert amd.traversal_gather_bytes(on, d, deg, seeds) == want
    assert 0.40 < want / amd.traversal_gather_bytes(off, d, deg, seeds) < 0.50      # (the 10M x 768 measurement: 44.6 of 101 GB)
    allpass = dict(off, rerank_rows=off["dist_evals"] - seeds)
    extra = (off["dist_evals"] - seeds) * (768 + 4.0)
    assert amd.traversal_gather_bytes(allpass, d, deg, seeds) == amd.traversal_gather_bytes(off, d, deg, seeds) + extra
    assert amd.traversal_gather_bytes(dict(off, rerank_rows=1), 100, deg, 0) > 0     # d not a multiple of 16: mirror row rounded up to 112


def test_product_library_reads_no_environment_and_ships_no_ablation(built):
    """a production library must not change its behaviour - let alone its answers - because of an environment variable.
    The product .so imports no getenv at all (its engine switches are entries of the eps_set_tuning table), and the kernel-ablation
    switches, which make answers wrong on purpose, exist in lab builds (-DEPS_LAB) only."""
    import subprocess
    undefined = subprocess.run(["nm", "-D", "--undefined-only", built], capture_output=True, text=True).stdout
    assert not re.search(r"\b(secure_)?getenv\b", undefined), [ln for ln in undefined.splitlines() if "getenv" in ln]
    blob = open(built, "rb").read()
    for s in (b"EPS_S8_ABLATE", b"EPS_MFMA_ABLATE", b"EPS_V7_ABL"):
        assert s not in blob, s
    src = ""
    for f in os.listdir(os.path.join(ROOT, "vectordb_amd", "csrc")):
        src += open(os.path.join(ROOT, "vectordb_amd", "csrc", f), errors="ignore").read()
    # the only getenv left in the sources is the lab build's fallback inside tune_env
    assert len(re.findall(r"\bgetenv\(", src)) == 1 and "#ifdef EPS_LAB\n  return getenv(name);" in src


def test_tuning_table(built):
    from vectordb_amd import _lib
    L = _lib.load()
    assert L.eps_set_tuning(b"EPS_TRV_PREFILTER", b"1") == 0
    assert L.eps_set_tuning(b"EPS_TRV_PREFILTER", None) == 0
    assert L.eps_set_tuning(None, None) == 0


def test_dropin_builds_without_the_oracle_directory():
    """the drop-in (dropin/Makefile) must not reach into oracle/ (test infrastructure): its stand-in headers and its C entry points are its own"""
    mk = open(os.path.join(ROOT, "dropin", "Makefile")).read()
    assert "oracle" not in re.sub(r"#.*", "", mk)
    for f in os.listdir(os.path.join(ROOT, "dropin")):
        if f.endswith((".cpp", ".hpp")):
            assert not re.search(r"#include\s+[\"<][^\">]*oracle", open(os.path.join(ROOT, "dropin", f)).read()), f


def test_caller_provided_result_buffers_are_checked_before_the_c_abi_sees_them():
    """GpuIndex.search(out=...) hands raw pointers to eps_index_search, which writes nq x k ids / distances and nq counts: a buffer of
    another dtype, shape or layout is refused in Python (no index, no GPU needed: the check is a plain function)"""
    from vectordb_amd.index import _check_out
    _check_out(np.empty((3, 10), np.int64), "ids", (3, 10), "int64")
    for bad in (np.empty((3, 10), np.int32), np.empty((3, 9), np.int64), np.empty((10, 3), np.int64).T, np.empty((3, 20), np.int64)[:, ::2]):
        with pytest.raises(ValueError):
            _check_out(bad, "ids", (3, 10), "int64")
    import torch
    _check_out(torch.empty((3, 10), dtype=torch.float32), "dist", (3, 10), "float32")
    with pytest.raises(ValueError):
        _check_out(torch.empty((3, 10), dtype=torch.float64), "dist", (3, 10), "float32")
    with pytest.raises(ValueError):
        _check_out(torch.empty((10, 3), dtype=torch.float32).t(), "dist", (3, 10), "float32")


def test_error_class_is_part_of_the_abi(built):
    """r6: callers branch on eps_index_last_error_class (EPS_ERRCLASS_DEVICE_RANGE), never on the wording of eps_index_last_error"""
    import ctypes as C
    from vectordb_amd import _lib
    L = _lib.load()
    assert L.eps_index_last_error_class(None) == 0
    hdr = open(os.path.join(ROOT, "include", "epsilla_gfx950.h")).read()
    assert "#define EPS_ERRCLASS_DEVICE_RANGE 1" in hdr
    src = open(os.path.join(ROOT, "dropin", "vec_search_executor.cpp")).read()
    assert "eps_index_last_error_class" in src and 'strstr(why' not in src
