"""bench.py's output contract on a real GPU: one JSON line with the fields the
driver reads, for the default workload and the two extra ones (short runs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config"]


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line():
    d = _run("--steps", "3", "--warmup", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 1e8 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["poses_found"] == d["config"]["scans_per_step"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["launches"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["sample"]


@pytest.mark.parametrize("workload", ["loop", "map"])
def test_extra_workloads(workload):
    d = _run("--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] > 0 and "workload" in d["config"]


def test_two_rank_rehearsal():
    """The N > 1 control flow of bench.py (rendezvous, barrier, max-over-ranks
    timing, all-gather of the records, rank-0 line) with two ranks sharing this
    box's one GPU over gloo (CSM_BENCH_REHEARSE=1); the numbers mean nothing."""
    env = dict(os.environ, CSM_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "cpu_baseline" not in d
