"""GPU parity through the C++ host side: my-lidar-graph-slam-v2_amd/host/
csm_adapters.hpp (the mirror of the reference's ScanMatcher / LoopDetector
interfaces) driven by host/adapter_demo, a plain g++ program linked against
libcsm_hip.so, checked against the CPU oracle."""
import json
import math
import os
import struct
import subprocess

import numpy as np
import pytest

from csm_hip import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "my-lidar-graph-slam-v2_amd", "host", "adapter_demo")


def _write_case(path, mode, case, inits, param_i, ranges3, thr, steps=None):
    g = np.ascontiguousarray(case["grid"], np.uint16)
    n = len(case["angles"])
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", mode, g.shape[0], g.shape[1], n, len(inits), param_i))
        f.write(struct.pack("<8d", *case["geom"], *ranges3, *thr))
        if steps is not None:
            f.write(struct.pack("<3d", *steps))
        f.write(struct.pack("<3d", *case["rel_pose"]))
        f.write(np.asarray(inits, np.float64).tobytes())
        f.write(np.asarray(case["angles"], np.float64).tobytes())
        f.write(np.asarray(case["ranges"], np.float64).tobytes())
        f.write(g.tobytes())


def _run(path, env=None):
    import __graft_entry__ as ge
    ge.build()
    out = subprocess.run([DEMO, path], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr + out.stdout
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_cpp_scan_matcher_adapter(tmp_path, oracle):
    case = synth.csm_case(11, rel_pose=(0.12, -0.03, 0.05))
    p = str(tmp_path / "csm.bin")
    _write_case(p, 0, case, [case["init_pose"]], 4, (1.0, 1.0, math.radians(10)), (0.0, 0.0))
    got = _run(p)
    lit = oracle.csm(case, 1.0, 1.0, math.radians(10), 4)
    assert got["found"] == lit["found"]
    assert [float.fromhex(v) for v in got["pose"]] == lit["estimatedPose"]
    assert float.fromhex(got["score"]) == lit["scoreMax"]
    assert got["win"] == [lit["winX"], lit["winY"], lit["winT"]]


def test_cpp_loop_detector_adapter(tmp_path, oracle):
    case = synth.csm_case(20, init_error=(0.4, -0.3, 0.06))
    rng = np.random.RandomState(9)
    inits = [tuple(np.asarray(case["truth"]) + rng.uniform(-0.5, 0.5, 3) * (1, 1, 0.2)) for _ in range(5)]
    inits.append((30.0, 30.0, 0.0))      # off the map: must be dropped from the results
    p = str(tmp_path / "bnb.bin")
    _write_case(p, 1, case, inits, 2, (2.5, 2.5, 0.5), (0.3, 0.5))
    got = _run(p)["results"]
    want = []
    for i, init in enumerate(inits):
        c = dict(case)
        c["init_pose"] = init
        r = oracle.bnb(c, 2.5, 2.5, 0.5, 2, 0.3, 0.5)
        if r["found"]:
            want.append((i, r["estimatedPose"], r["scoreMax"]))
    assert [g["node"] for g in got] == [w[0] for w in want]
    for g, w in zip(got, want):
        assert [float.fromhex(v) for v in g["pose"]] == w[1]
        assert float.fromhex(g["score"]) == w[2]
    # the same detector over a device list (Create(..., deviceIds)): a one-entry list, two
    # members on GPU 0 (threads + host-staged exchange) and a one-rank RCCL communicator
    for env in ({"CSM_DEMO_DEVICES": "0"}, {"CSM_DEMO_DEVICES": "0,0"},
                {"CSM_DEMO_DEVICES": "0", "CSM_GROUP_FORCE_RCCL": "1"}):
        assert _run(p, env)["results"] == got, env
    # with the detector's final matcher (UseFinalScanMatcher): pose, cost and covariance of
    # ScanMatcherLinearSolver on the search's estimate, at the header's tolerance
    fin = _run(p, {"CSM_DEMO_REFINE": "1", "CSM_DEMO_DEVICES": "0,0"})["results"]
    assert [g["node"] for g in fin] == [w[0] for w in want]
    alloc = (case["grid"].reshape(25, 16, 25, 16).max(axis=(1, 3)) > 0).astype(np.uint8)
    lam = 1e-4
    for g, w in zip(fin, want):
        r = oracle.linear_solver(case["grid"], case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                                 w[1], lambda_=lam, alloc=alloc)
        lam = r["lambda_"]                     # the solver object keeps its damping factor
        got_pose = np.array([float.fromhex(v) for v in g["pose"]])
        assert np.all(np.abs(got_pose - np.array(r["estimated_pose"])) < 1e-6)
        assert abs(float.fromhex(g["cost"]) - r["normalized_cost"]) < 1e-8
        assert abs(float.fromhex(g["cov00"]) - r["covariance"][0, 0]) < 1e-6 * abs(r["covariance"][0, 0])


def test_cpp_loop_detector_correlative_adapter(tmp_path, oracle):
    case = synth.csm_case(21, init_error=(0.4, -0.3, 0.06))
    rng = np.random.RandomState(10)
    inits = [tuple(np.asarray(case["truth"]) + rng.uniform(-0.5, 0.5, 3) * (1, 1, 0.2)) for _ in range(4)]
    inits.append((-30.0, 30.0, 0.0))     # off the map
    p = str(tmp_path / "ldc.bin")
    _write_case(p, 2, case, inits, 5, (2.5, 2.5, 0.5), (0.3, 0.5))
    got = _run(p)["results"]
    want = []
    for i, init in enumerate(inits):
        c = dict(case)
        c["init_pose"] = init
        r = oracle.csm(c, 2.5, 2.5, 0.5, 5, 0.3, 0.5)
        if r["found"]:
            want.append((i, r["estimatedPose"], r["scoreMax"]))
    assert want, "test data should produce at least one detection"
    assert [g["node"] for g in got] == [w[0] for w in want]
    for g, w in zip(got, want):
        assert [float.fromhex(v) for v in g["pose"]] == w[1]
        assert float.fromhex(g["score"]) == w[2]


def test_cpp_grid_search_adapter(tmp_path, oracle):
    case = synth.csm_case(22, n_beams=180, rel_pose=(0.1, 0.02, -0.04))
    p = str(tmp_path / "gs.bin")
    window = (0.8, 0.6, 0.2)
    steps = (0.05, 0.04, 0.01)
    _write_case(p, 3, case, [case["init_pose"]], 0, window, (0.3, 0.5), steps)
    got = _run(p)
    want = oracle.grid_search(case, *window, *steps, 0.3, 0.5)
    assert got["found"] == want["found"] == 1
    assert [float.fromhex(v) for v in got["pose"]] == want["estimatedPose"]
    assert float.fromhex(got["score"]) == want["scoreMax"]


def test_cpp_grid_map_builder_adapter(tmp_path, oracle):
    """GridMapBuilderHIP::UpdateLatestMap (last 10 of 12 nodes) + the matcher on
    the device-resident result, through the C++ adapters."""
    case = synth.map_case(8, n_scans=12, n_beams=360, rel_pose=(0.05, 0.0, 0.01))
    nodes = case["nodes"]
    latest = nodes[-10:]
    map_pose = latest[0]["pose"]
    want_shape, want_grid, stats = oracle.construct_map(case["shape"], map_pose, latest)
    last = nodes[-1]
    c, s_ = math.cos(map_pose[2]), math.sin(map_pose[2])
    dx, dy = last["pose"][0] + 0.07 - map_pose[0], last["pose"][1] - 0.05 - map_pose[1]
    init = (c * dx + s_ * dy, -s_ * dx + c * dy, last["pose"][2] + 0.01 - map_pose[2])
    p = str(tmp_path / "gmb.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<6i", 4, len(nodes), 360, 16, 10, 4))
        f.write(struct.pack("<8d", 0.05, 0.01, 20.0, 0.62, 0.46, 1.0, 1.0, 0.25))
        f.write(struct.pack("<3d", *last["rel_pose"]))
        f.write(struct.pack("<3d", *init))
        for nd in nodes:
            f.write(struct.pack("<5d", *nd["pose"], nd["min_range"], nd["max_range"]))
            f.write(np.asarray(nd["angles"], np.float64).tobytes())
            f.write(np.asarray(nd["ranges"], np.float64).tobytes())
    got = _run(p)
    assert (got["rows"], got["cols"]) == (want_shape["rows"], want_shape["cols"])
    assert [float.fromhex(v) for v in got["off"]] == [want_shape["off_x"], want_shape["off_y"]]
    h = 1469598103934665603
    for b in want_grid.astype("<u2").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert got["hash"] == "%016x" % h
    assert (got["rays"], got["updates"]) == (stats["rays"], stats["updates"])
    assert [float.fromhex(v) for v in got["map_pose"]] == list(map_pose)
    geom = (want_shape["res"], want_shape["off_x"], want_shape["off_y"])
    want = oracle.csm(dict(grid=want_grid, geom=geom, angles=last["angles"], ranges=last["ranges"],
                           rel_pose=last["rel_pose"], init_pose=init), 1.0, 1.0, 0.25, 4)
    assert got["found"] == want["found"] == 1
    assert [float.fromhex(v) for v in got["pose"]] == want["estimatedPose"]
    assert float.fromhex(got["score"]) == want["scoreMax"]
    # the local map grown scan by scan (CreateLocalMap + UpdateGridMap)
    shape = case["shape"]
    grid = np.zeros((shape["rows"], shape["cols"]), np.uint16)
    for nd in nodes:
        shape, grid, _ = oracle.update_map(shape, grid, nodes[0]["pose"], nd)
    assert (got["local"]["rows"], got["local"]["cols"]) == (shape["rows"], shape["cols"])
    assert [float.fromhex(v) for v in got["local"]["off"]] == [shape["off_x"], shape["off_y"]]
    h = 1469598103934665603
    for b in grid.astype("<u2").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert got["local"]["hash"] == "%016x" % h
