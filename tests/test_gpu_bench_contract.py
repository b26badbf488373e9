"""bench.py's output contract on a real GPU: one JSON line with the fields the
driver reads, for the default workload, the side configs and the extra
workloads (short runs), plus the N > 1 control flow with two ranks sharing
this box's one GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config"]


def _run(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_roofline(r):
    """An LDS-bound kernel: a fraction of a real bound, so 0 < frac <= 1; the
    metric's logical HBM figure is kept beside it (and may exceed 1)."""
    assert r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == 256.0 * 256 * 2.4
    assert r["launches"] > 0 and r["avg_launch_us"] > 0
    assert 0.0 < r["frac"] <= 1.0, r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["logical_hbm_frac"] > 0
    assert abs(r["logical_hbm_frac"] - r["logical_hbm_gbs"] / 8000.0) < 1e-9


def test_default_line():
    d = _run("--steps", "2", "--warmup", "1",
             env={"CSM_BENCH_SCANS": "256", "CSM_BENCH_DISTINCT": "128", "CSM_BENCH_WINDOWS": "64",
                  "CSM_BENCH_CONFIGS": "config2_single_query,config3,config5"})
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 1e8 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["config"]["workload"].startswith("configs[1]")
    assert d["config"]["poses_found"] == d["config"]["scans_per_step"] == 256
    r = d["roofline"]
    _check_roofline(r)
    if r["traffic"] is not None:
        assert "committed" in r["traffic_source"] and 0 < r["hbm_frac_measured"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    a = c["all_cores"]
    assert a["cores"] >= 1 and a["value"] > 0 and "OpenMP" in a["sample"]
    assert "configs" not in d                   # the driver's parser keeps "config", not extra top-level keys
    assert d["config"]["verified"] >= 8 and "fine_level" in d["config"]
    cf = d["config"]["side_runs"]
    assert set(d["config"]["side_runs_summary"]) == set(cf)
    assert cf["config2_single_query"]["latency_ms_median"] > 0
    c3 = cf["config3"]
    assert c3["value"] > 1e8 and c3["pyramid_build_ms"] > 0 and c3["end_to_end_value"] < c3["value"]
    assert c3["leaves_per_step"] == 256 * 59 * 52 * 52 and c3["found"] > 0
    _check_roofline(c3["roofline"])
    c5 = cf["config5"]
    assert c5["candidates"] % (804 * 804) == 0 and c5["candidates"] > 9.2e8
    assert c5["value"] > 1e8 and c5["found"] == 1
    ev = c5["evaluated"]
    assert ev["coarse_nodes"] == c5["candidates"] // (804 * 804) * 201 * 201
    assert 0 < ev["fine_candidates"] < c5["candidates"] // 4 and ev["fine_blocks_skipped"] > ev["fine_blocks_scored"]
    r5 = c5["roofline"]
    assert r5["bound"] == "lds" and 0 < r5["frac"] < 1 and abs(r5["frac"] - r5["achieved"] / r5["peak"]) < 1e-9


@pytest.mark.parametrize("workload,steps", [("config5", "3"), ("latency", "200")])
def test_side_workloads_as_lines_of_their_own(workload, steps):
    d = _run("--workload", workload, "--steps", steps, "--warmup", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == int(steps) and d["value"] > 1e8
    assert d["config"]["workload"].startswith("configs[4]" if workload == "config5" else "configs[1]")
    assert d["roofline"]["bound"] == "lds" and 0 < d["roofline"]["frac"] < 1


@pytest.mark.parametrize("workload", ["loop", "map"])
def test_extra_workloads(workload):
    d = _run("--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] > 0 and "workload" in d["config"]
    if workload == "loop":
        _check_roofline(d["roofline"])
        assert d["scaling"] == "weak" and d["config"]["queries_total"] == 256


def _two_ranks(port, *args, env=None):
    e = dict(os.environ, CSM_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", *args], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=e)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_rehearsal_default_is_the_sharded_loop_batch():
    """N > 1 defaults to configs[3]: the query batch sharded in contiguous blocks
    + one all-gather of the records per step (rendezvous, barrier, max-over-ranks
    timing, rank-0 line). bench.py itself asserts that each rank's block of the
    gathered buffer equals its local records. Two ranks share this box's one GPU
    over gloo (CSM_BENCH_REHEARSE=1) on a 96-query batch; the numbers mean nothing."""
    d = _two_ranks(29533, env={"CSM_BENCH_LOOP_TOTAL": "96"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["workload"].startswith("configs[3]")
    assert d["config"]["queries_total"] == 96 and d["config"]["queries_per_rank"] == 48
    assert d["config"]["leaves_per_step"] == 96 * 59 * 52 * 52
    assert "cpu_baseline" not in d


def test_two_rank_rehearsal_csm_replicas():
    """--workload csm at N > 1: configs[1] per rank + all-gather of the per-scan
    records; bench.py asserts that the gathered slice of every rank equals the
    records it wrote (the ordering of the collective against the scoring stream)."""
    d = _two_ranks(29534, "--workload", "csm", env={"CSM_BENCH_SCANS": "128", "CSM_BENCH_DISTINCT": "64", "CSM_BENCH_WINDOWS": "64"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["poses_found"] == 128
