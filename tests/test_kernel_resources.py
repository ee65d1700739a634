"""The filter kernel's schedule is hand-counted (vmcnt/lgkmcnt waits, one instruction per MFMA shadow): if hipcc starts
spilling in it, every scratch reload adds a compiler-inserted vmcnt(0) that drains the LDS-DMA ring and the kernel
silently loses several x.  Compile it (no GPU needed) and pin the resource usage."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_filter_kernels_do_not_spill(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(ROOT, "vectordb_amd", "csrc", "mfma_filter.hip"), "-o", str(tmp_path / "mf.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    v7 = {k: v for k, v in usage.items() if "mfma_filter_kernel_v7" in k}
    # {128, 256}-query tiles x {ids, keys, dense} epilogues x {fp16, int8} operands + the two 128-row-tile forms (two workgroups per CU)
    assert len(v7) == 14, list(usage)
    for k, u in v7.items():
        assert u["ScratchSize [bytes/lane]"] == 0, (k, u)
        # r4 (EPS_V7_VI = 7): 7 of the 8 row blocks' accumulators live in arch VGPRs (the epilogue reads them in place), the operand
        # fragments and the last row block in the accumulator file; one wavefront per SIMD either way
        assert 64 <= u["AGPRs"] <= 200 and u["VGPRs"] <= 256 and u["Occupancy [waves/SIMD]"] >= 1, (k, u)
        if "ELi4EEE" in k:   # NRB = 4: must fit two wavefronts per SIMD (128 arch + 128 accumulator-file registers), or the form is pointless
            assert u["AGPRs"] <= 128 and u["VGPRs"] <= 128 and u["Occupancy [waves/SIMD]"] >= 2, (k, u)
    for k, u in usage.items():
        if "mfma_filter_kernel_v3" in k:
            assert u["ScratchSize [bytes/lane]"] == 0, (k, u)
    # r4, the one-pass search of a handful of queries: 3 row widths x 3 query counts (+ r5: 3 x 2 forms that quantise 1-2 queries in the kernel); no scratch (a function call in the kernel cost every
    # wavefront of a 2048-wavefront launch its scratch set-up: 15 us of a 0.2 ms call), at least two wavefronts per SIMD
    s8 = {k: v for k, v in usage.items() if "stream8_kernel" in k}
    assert len(s8) == 21, list(usage)   # (r5: + 1-2 queries quantised by the pass itself; + 3 x 2 forms with the 128-slot table of k = 17..64)
    for k, u in s8.items():
        assert u["ScratchSize [bytes/lane]"] == 0 and u["Occupancy [waves/SIMD]"] >= 2, (k, u)
    # r5: the same pass for 5..16 queries on the matrix cores (3 row widths; the slot count of the table is a run-time value there)
    s8m = {k: v for k, v in usage.items() if "stream8m_kernel" in k}
    assert len(s8m) == 6, list(usage)   # (r6: x 2 - one and two 16-query column blocks)
    for k, u in s8m.items():
        assert u["ScratchSize [bytes/lane]"] == 0 and u["Occupancy [waves/SIMD]"] >= 2, (k, u)
    # r5: filter programs reach the pass as a bitset - the evaluator (a function with a 16-entry stack) lives in this launch, not in the pass
    assert any("filter_mask_kernel" in k for k in usage), list(usage)


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_traversal_kernel_keeps_four_waves_per_simd(tmp_path):
    """graph_search plans its launch (workgroups per CU) for 4 wavefronts per SIMD, i.e. <= 128 VGPRs; the kernel is compiled for
    exactly that and the few registers that do not fit are spilled outside the gather loops - a handful of dwords, pinned here."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(ROOT, "vectordb_amd", "csrc", "traverse.hip"), "-o", str(tmp_path / "tr.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    trv = {k: v for k, v in usage.items() if "traverse2_kernel" in k}
    assert len(trv) == 24, list(usage)   # {float4, scalar} row loads x {4, 8, 16} wavefronts x queues in {LDS, HBM} x {without, with} the prefilter
    for k, u in trv.items():
        assert u["VGPRs"] <= 128 and u["Occupancy [waves/SIMD]"] >= 4, (k, u)
        with_prefilter = "ELb1EEEvNS_8Trv2ArgsE" in k or k.rstrip(">").endswith("true")
        # r5 (3 mirror rows x 3 pieces and 3 fp32 rows x 3 pieces in flight per lane group): 9 dwords in the form every batch search runs
        # (float4 rows, 4 wavefronts, queues in LDS), at most 21 in the forms with the queues in HBM / scalar row loads
        hot = k.startswith("_ZN3eps16traverse2_kernelILb1ELi4ELb0ELb1E")
        assert u["ScratchSize [bytes/lane]"] <= ((48 if hot else 96) if with_prefilter else 0), (k, u)


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_one_pass_tail_kernel_fits_a_full_workgroup(tmp_path):
    """s8_rerank_kernel (r5) is launched with 1024 threads per workgroup - 128 registers per lane at most - and keeps 12 pieces of 16 bytes per lane
    in flight (every piece of four rows at once): no scratch, and its LDS (ids, keys, the query, the rank slots) stays under the 64 KB a workgroup
    gets by default.  The re-rank kernels the batched stages use (KPL = 1, 2) stay free of scratch as well."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(ROOT, "vectordb_amd", "csrc", "flat_kernels.hip"), "-o", str(tmp_path / "fk.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    tail = {k: v for k, v in usage.items() if "s8_rerank_kernel" in k}
    assert len(tail) == 2, list(usage)
    for k, u in tail.items():
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs"] <= 128, (k, u)
        # static LDS (ids + rank slots + counters) + the dynamic part the launch asks for: the query (<= 4 KB at d <= 1024) + 4096 keys of 8 bytes
        assert u["LDS Size [bytes/block]"] + 4096 + 4096 * 8 <= 65536, (k, u)
    for k, u in usage.items():
        if ("rerank_kernelILi1ELb1E" in k or "rerank_kernelILi2ELb1E" in k or "rerank_split_kernel" in k):
            assert u["ScratchSize [bytes/lane]"] == 0, (k, u)
