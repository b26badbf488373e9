"""Randomised GPU parity: many small, odd-shaped cases (tiny grids, windows
larger than the map, 1-30 beams, L from 1 to 9, thresholds, relative sensor
poses, maps that are mostly unknown) against the literal CPU restatement.
Deterministic seeds; every case must agree on found / best indices / f64 score
/ estimated pose, for the correlative matcher (single and batched) and for
branch and bound."""
import math

import numpy as np
import pytest

from csm_hip import api

pytestmark = pytest.mark.gpu


def _random_case(rng):
    rows = int(rng.choice([8, 16, 24, 48, 80, 136]))
    cols = int(rng.choice([8, 16, 40, 56, 96, 152]))
    res = float(rng.choice([0.05, 0.1, 0.025]))
    grid = rng.randint(1, 65535, size=(rows, cols)).astype(np.uint16)
    mode = rng.randint(0, 4)
    if mode == 0:
        grid[rng.rand(rows, cols) < 0.6] = 0                      # mostly unknown
    elif mode == 1:
        grid = (grid // 16384 * 16384 + 1).astype(np.uint16)      # 4 levels: ties
        grid[rng.rand(rows, cols) < 0.2] = 0
    elif mode == 2:
        grid[:, : cols // 2] = 0                                  # half the map unknown
    off = (-cols * res * rng.rand(), -rows * res * rng.rand())
    geom = (res, off[0], off[1])
    n = int(rng.choice([1, 2, 3, 7, 16, 30]))
    angles = np.sort(rng.uniform(-math.pi, math.pi, n))
    ranges = rng.uniform(0.2, max(0.6, 0.45 * min(rows, cols) * res), n)
    cx, cy = off[0] + cols * res * rng.uniform(0.2, 0.8), off[1] + rows * res * rng.uniform(0.2, 0.8)
    init = (cx, cy, rng.uniform(-math.pi, math.pi))
    rel = (0.0, 0.0, 0.0) if rng.rand() < 0.5 else tuple(rng.uniform(-0.2, 0.2, 3))
    return dict(grid=grid, geom=geom, angles=angles, ranges=ranges, init_pose=init, rel_pose=rel)


def _params(rng, case):
    res = case["geom"][0]
    rx = res * rng.randint(2, 14)
    ry = res * rng.randint(2, 14)
    rt = rng.uniform(0.05, 0.6)
    thr = [(0.0, 0.0), (0.2, 0.3), (0.5, 0.6), (0.05, 0.95)][rng.randint(0, 4)]
    return rx, ry, rt, thr


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_correlative(gpu_ctx, oracle, seed):
    rng = np.random.RandomState(1000 + seed)
    case = _random_case(rng)
    rx, ry, rt, thr = _params(rng, case)
    rows, cols = case["grid"].shape
    Lr = int(rng.randint(1, min(9, rows, cols) + 1))
    lit = oracle.csm(case, rx, ry, rt, Lr, thr[0], thr[1])
    if (2 * lit["winT"] + 1) > 400:
        pytest.skip("theta window too large for a quick fuzz case")
    m = api.ScanMatcherCorrelativeHIP("fuzz", Lr, rx, ry, rt, ctx=gpu_ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"], score_threshold=thr[0],
                          known_rate_threshold=thr[1])
    raw = out["raw"]
    assert out["pose_found"] == lit["found"], (raw, lit)
    assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"]), (raw, lit)
    assert raw["score"] == lit["scoreMax"]
    assert out["estimated_pose"] == lit["estimatedPose"]
    # the batched entry must agree with the single-query one
    gpu_ctx.upload_grid(4242, case["grid"])
    q = dict(map_id=4242, geom=case["geom"], angles=case["angles"], ranges=case["ranges"],
             rel_pose=case["rel_pose"], init_pose=case["init_pose"])
    b = gpu_ctx.correlative_match_batch([q, q], rx, ry, rt, Lr, thr[0], thr[1])
    for o in b:
        assert o["pose_found"] == lit["found"]
        assert (o["raw"]["best_x"], o["raw"]["best_y"], o["raw"]["best_theta"]) == \
            (lit["bestX"], lit["bestY"], lit["bestT"])
        assert o["raw"]["score"] == lit["scoreMax"]
    gpu_ctx.release_grid(4242)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_branch_and_bound(gpu_ctx, oracle, seed):
    rng = np.random.RandomState(5000 + seed)
    case = _random_case(rng)
    rx, ry, rt, thr = _params(rng, case)
    rows, cols = case["grid"].shape
    hmax = int(math.floor(math.log2(min(rows, cols))))
    H = int(rng.randint(0, min(5, hmax) + 1))
    thr = (max(thr[0], 0.01), max(thr[1], 0.01))
    want = oracle.bnb(case, rx, ry, rt, H, thr[0], thr[1])
    if (2 * want["winT"] + 1) > 400:
        pytest.skip("theta window too large for a quick fuzz case")
    gpu_ctx.upload_grid(4343, case["grid"])
    q = dict(map_id=4343, geom=case["geom"], angles=case["angles"], ranges=case["ranges"],
             rel_pose=case["rel_pose"], init_pose=case["init_pose"])
    out = gpu_ctx.bnb_match_batch([q], rx, ry, rt, H, thr[0], thr[1])[0]
    raw = out["raw"]
    assert out["pose_found"] == want["found"], (raw, want)
    assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (want["bestX"], want["bestY"], want["bestT"]), (raw, want)
    assert raw["score"] == want["scoreMax"]
    assert out["estimated_pose"] == want["estimatedPose"]
    gpu_ctx.release_grid(4343)
