"""CPU suite: the host-side logic inside libcsm_hip.so against the oracle, and
the C ABI surface (the library loads without a GPU and exports every symbol
include/csm_hip.h declares; no compute call is made here)."""
import math
import os
import re

import numpy as np
import pytest

from csm_hip import _lib as L
from csm_hip import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "csm_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(csm_[a-z0-9_]+)\s*\(", text))
    lib = L.load()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)


def test_no_gpu_means_enodev_not_a_fallback():
    import ctypes as C
    lib = L.load()
    ctx = C.c_void_p()
    cfg = L.Config()
    rc = lib.csm_create(C.byref(cfg), C.byref(ctx))
    if rc == 0:           # running on a GPU box: fine, clean up
        lib.csm_destroy(ctx)
    else:
        assert rc == L.CSM_ENODEV


def test_search_step_window_min_known(oracle):
    rng = np.random.RandomState(1)
    for _ in range(50):
        res = float(rng.choice([0.025, 0.05, 0.1]))
        ranges = rng.uniform(0.3, 30, size=rng.randint(1, 400))
        assert api.host_search_step(res, ranges) == oracle.search_step(res, ranges)
    sx, sy, st = api.host_search_step(0.05, [1.0, 5.7296, 3.0])
    assert api.host_window(1.0, sx) == 10
    assert api.host_window(math.radians(10), st) == int(math.ceil(0.5 * math.radians(10) / st))
    for n in (1, 7, 360, 1080):
        for thr in (0.0, 0.1, 0.5, 0.6, 0.999, 1.0):
            k = api.host_min_known(n, thr)
            assert k == min([j for j in range(n + 2) if j / n > thr] + [n + 1])


def test_pose_algebra_and_lut_match_oracle(oracle):
    rng = np.random.RandomState(2)
    for _ in range(100):
        a = rng.uniform(-10, 10, 3)
        b = rng.uniform(-3, 3, 3)
        assert list(api.host_compound(a, b)) == list(oracle.compound(a, b))
        assert list(api.host_inverse_compound(a, b)) == list(oracle.inverse_compound(a, b))
        assert list(api.host_move_backward(a, b)) == list(oracle.move_backward(a, b))
    assert np.array_equal(api.host_probability_lut(), oracle.lut())


def test_projection_matches_oracle_bit_for_bit(oracle):
    case = synth.csm_case(3, n_beams=257)
    sx, sy, st = api.host_search_step(case["geom"][0], case["ranges"])
    sensor = api.host_compound(case["init_pose"], (0.1, -0.2, 0.05))
    wt = 7
    col, row, rc, rs = api.host_project(case["geom"], sensor, st, wt, case["angles"], case["ranges"], True)
    for t in range(-wt, wt + 1):
        pose = (sensor[0], sensor[1], sensor[2] + st * t)
        c, r = oracle.project(case["geom"], pose, case["angles"], case["ranges"])
        assert np.array_equal(col[t + wt], c)
        assert np.array_equal(row[t + wt], r)
    assert np.array_equal(rc[wt], case["ranges"] * np.cos(sensor[2] + case["angles"]))


def test_host_map_resize_matches_oracle():
    """csm_host_map_resize (GridMap::Resize / Expand on index boxes) against the
    CPU restatement, over frames with negative indices and odd block sizes."""
    import math
    from oracle import oracle
    from csm_hip import api
    rng = np.random.RandomState(5)
    for _ in range(300):
        res = float(rng.choice([0.05, 0.1, 0.025]))
        lb = int(rng.choice([2, 4, 5]))
        shape = dict(res=res, off_x=float(rng.uniform(-5, 5)), off_y=float(rng.uniform(-5, 5)),
                     rows=(1 << lb) * int(rng.randint(1, 9)), cols=(1 << lb) * int(rng.randint(1, 9)),
                     log2_block=lb)
        # two sensor positions are the only points: their box drives the resize
        pts = rng.uniform(-8, 8, (2, 2))
        nodes = [dict(pose=(float(p[0]), float(p[1]), 0.0), angles=np.zeros(1), ranges=np.zeros(1),
                      min_range=1.0, max_range=2.0) for p in pts]
        try:
            want, _, _ = oracle.construct_map(shape, (0.0, 0.0, 0.0), nodes)
        except ValueError:
            continue
        tiny = np.finfo(np.float64).tiny                     # the reference's initial maximum
        xs, ys = list(pts[:, 0]), list(pts[:, 1])
        box = [math.floor((min(xs) - res - shape["off_x"]) / res), math.floor((min(ys) - res - shape["off_y"]) / res),
               math.floor((max(xs + [tiny]) + res - shape["off_x"]) / res),
               math.floor((max(ys + [tiny]) + res - shape["off_y"]) / res)]
        got, shift = api.host_map_resize(shape, box)
        assert got == want
        assert shift[0] % (1 << lb) == 0 and shift[1] % (1 << lb) == 0
        # Expand: a box inside the map changes nothing, one outside joins the extent
        same, shift0 = api.host_map_resize(got, [1, 1, got["cols"] - 2, got["rows"] - 2], expand=True)
        assert same == got and shift0 == (0, 0)
        grown, sh1 = api.host_map_resize(got, [-3, 2, 5, got["rows"] + 1], expand=True)
        assert grown["cols"] >= got["cols"] + (1 << lb) and grown["rows"] >= got["rows"] + (1 << lb)
        assert sh1[1] < 0 and sh1[0] == 0
    with pytest.raises(api.CsmError):
        api.host_map_resize(shape, [5, 5, 4, 9])             # empty box: the reference asserts


def test_summary_array_record_bytes_layout():
    """SummaryArray.record_bytes() lifts the 48-byte csm_result out of each
    csm_summary (what the ranks all-gather), matching parallel.records_to_bytes."""
    import ctypes as C
    from csm_hip import parallel
    n = 5
    arr = (L.Summary * n)()
    for i in range(n):
        arr[i].raw.found = i % 2
        arr[i].raw.best_x, arr[i].raw.best_y, arr[i].raw.best_theta = i, -i, 2 * i
        arr[i].raw.key = 1000 + i
        arr[i].raw.score = 0.25 * i
        arr[i].candidates = 7 * i
    sa = api.SummaryArray(arr)
    b = sa.record_bytes()
    assert b.size == n * C.sizeof(L.Result) == n * 48
    want = parallel.records_to_bytes([api.result_to_dict(arr[i].raw) for i in range(n)]).reshape(-1)
    assert np.array_equal(b, want)
    assert sa.total("candidates") == 7 * sum(range(n)) and len(sa) == n
    assert sa[3]["raw"]["best_theta"] == 6


def test_shard_bounds_match_the_python_detector_and_cover_exactly_once():
    """csm_shard_bounds (the C ABI's contiguous blocks, loop_detector_fpga_parallel.cpp:42-46)
    against csm_hip.parallel.shard_bounds, for the sizes the configs use and ragged ones."""
    from csm_hip import api, parallel
    for n in list(range(0, 40)) + [255, 256, 257, 2047, 2048]:
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = api.host_shard_bounds(n, r, world)
                assert (lo, hi) == parallel.shard_bounds(n, r, world)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert api.host_shard_bounds(2048, 3, 8) == (768, 1024)
    assert api.host_shard_bounds(10, 5, 4) == (0, 0)       # no such member


def test_group_create_without_a_gpu_fails_like_a_context():
    import ctypes as C
    import numpy as np
    import torch
    from csm_hip import _lib
    lib = _lib.load()
    g = C.c_void_p()
    ids = np.zeros(1, np.int32)
    rc = lib.csm_group_create(ids.ctypes.data_as(C.c_void_p), 1, C.byref(g))
    if not torch.cuda.is_available():
        assert rc == _lib.CSM_ENODEV and not g.value
    elif rc == 0:
        lib.csm_group_destroy(g)
    assert lib.csm_group_create(None, 0, C.byref(g)) == _lib.CSM_EINVAL
