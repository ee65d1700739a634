"""bench.py's output contract (one JSON line with the driver's keys, `roofline` and `cpu_baseline`), on a small workload."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "300000",
                        "--batch", "256", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 256 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"] + 1e-6
    assert "workload" in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and 0 < roof["frac"] < 1
    # r4: the clock / power leg after the timed region (a failure of the SMI library is reported, never fatal)
    ul = roof.get("under_load")
    assert ul is not None and ("failed" in ul or (ul["steps"] > 0 and ul["ms_per_step_sustained"] > 0)), ul
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port")
    assert d["recall_at_10"] >= 0.999
    # r5: N = 1 lines carry no scaling claim; the same steps end to end (host queries in, host results out, inside the timed region) beside the
    # device-resident value (the contract's timed region), and the reference's own BruteForceSearch answers compared with the GPU's
    assert d["scaling"] is None and d["value_device_resident"] == d["value"]
    e = d["end_to_end"]
    assert e["last_step_equals_device_resident_run"] is True and e["value"] > 0
    if cb["kind"] == "reference":
        bf = [l for l in cb["legs"] if l["leg"] == "bruteforce"][0]
        assert bf["gpu_headline_answers_equal"] == bf["queries"], bf


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_merge_equals_unsharded_scan():
    """The N > 1 path of bench.py end to end with product code on both sides: two ranks (gloo, both on this GPU), each
    with its own shard (rows i = local*2 + rank), ONE all-gather of the packed per-shard top-k, eps_merge_topk_packed;
    recall is measured against the merged exact stream scans of the shards, so anything short of 1.0 is a sharding /
    id-map / merge bug."""
    env = dict(os.environ, EPS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rows", "150000", "--batch", "96", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "rows_total=300000" in d["config"]["workload"]
    assert d["recall_at_10"] == 1.0 and d["recall_check"]["ground_truth_vs_torch_fp32_scan"] == 1.0
    _check_exchange(d, 2, 96, 10)
    # r6: 96 queries x 10 x 12 B fit a mailbox slot - the exchange step ran without a collective (peer stores + flag between the two processes)
    assert "mailbox" in d["exchange"]["collective"], d["exchange"]["collective"]


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """BASELINE configs[4]'s shape (8 shards, one packed all-gather, k-way merge) with the driver's launch line, on the one GPU of
    this box: eight gloo ranks, each with its own shard of the rows (row i = local * 8 + rank).  Exercises what the 8-GPU run
    exercises - rank-local shards, the id map, the gather layout for 8 contributions, the merge - except RCCL itself."""
    env = dict(os.environ, EPS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--rows", "80000", "--batch", "160", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "rows_total=640000" in d["config"]["workload"]
    assert d["recall_at_10"] == 1.0 and d["recall_check"]["ground_truth_vs_torch_fp32_scan"] == 1.0
    _check_exchange(d, 8, 160, 10)
    assert d["exchange"]["collective"] == "all_gather (host staged)"      # (160 queries: beyond a mailbox slot, and gloo ranks on one device cannot form an RCCL communicator)


def _check_exchange(d, world, batch, k):
    """r5: an N > 1 line proves by itself that N ranks took part and what the one exchange step cost: backend, world size,
    every rank's device, the all-gather and merge times of every timed step, the bytes a rank contributes (12 B x k x batch)."""
    x = d["exchange"]
    assert x["world_size"] == world and x["backend"] in ("gloo", "nccl")
    assert x["bytes_per_rank_per_step"] == (batch * k * 12 + 7) // 8 * 8 and x["bytes_gathered_per_rank_per_step"] == world * x["bytes_per_rank_per_step"]
    assert len(x["all_gather_us_per_step"]) == d["steps"] == len(x["merge_us_per_step"]) and all(v >= 0 for v in x["merge_us_per_step"])
    assert x["all_gather_us_mean"] is not None and x["merge_us_mean"] > 0
    assert sorted(r["rank"] for r in x["ranks"]) == list(range(world))
    assert len(set(r["pid"] for r in x["ranks"])) == world
    for r in x["ranks"]:
        assert r["device_name"] and r["main_kernel_ms"] is not None and r["search_rows"] > 0
    e = d["end_to_end"]
    assert e["last_step_equals_device_resident_run"] is True and 0 < e["value"] and 0 < e["frac_of_device_resident"] < 1.5
    assert d["value_device_resident"] == d["value"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`--gpus N` is what the line reports as n_gpus: under a launcher that started a different number of ranks bench.py stops
    instead of measuring fewer GPUs than asked for (r3's flag was parsed and never read).  CPU-only: fails before any device call."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_when_started_plain():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver used for N = 1): bench.py re-executes itself under
    torch.distributed.run with two ranks; gloo here, both on this box's one GPU."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EPS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "120000", "--batch", "64",
                        "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "rows_total=240000" in d["config"]["workload"] and d["recall_at_10"] == 1.0


@pytest.mark.gpu
def test_bench_inproc_shard_group():
    """--inproc: the one-process form of SURVEY 8e (eps_index_create_sharded, rows attached per shard on the shard's device, queries
    and results in device memory, merge on device 0) under the same JSON contract; two shards on this box's one GPU."""
    env = dict(os.environ, EPS_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inproc", "--steps", "2", "--warmup", "1", "--rows", "120000",
                        "--batch", "64"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "rows_total=240000" in d["config"]["workload"] and d["recall_at_10"] == 1.0
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key


@pytest.mark.gpu
def test_live_pmc_traffic_and_kernel_trace_of_a_child_run():
    """r6: bench.py measures `roofline.traffic` itself - child runs of the launch under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE), the longest dispatch of the
    kernel, gfx950's FETCH_SIZE x 2 - and the profiler's view of the launch duration under `--kernel-trace --stats`.  Here on a launch whose bytes are known: the
    one-pass search over a 300 000 x 768 mirror streams 300 000 x (768 + 4) bytes."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 not installed")
    sys.path.insert(0, ROOT)
    from bench_legs import pmc_bytes, profiler_kernel_us
    child = [sys.executable, os.path.join(ROOT, "scripts", "prof_single_query.py"), "300000", "768"]
    t = pmc_bytes(child, "stream8_kernel")
    assert "bytes" in t, t
    alg = 300000 * 772.0
    assert 0.98 * alg <= t["bytes"] <= 1.10 * alg, (t, alg)
    kt = profiler_kernel_us(child, "stream8_kernel")
    assert "median_us" in kt and 20.0 < kt["median_us"] < 200.0 and kt["dispatches"] >= 10, kt
