"""GPU tests at BASELINE's large sizes, through properties that need no full
CPU oracle run, plus the LoopDetectorCorrelative wrapper (default L = 5)."""
import math

import numpy as np
import pytest

from csm_hip import api, parallel, synth

pytestmark = pytest.mark.gpu


def test_config5_exhaustive_global_window(gpu_ctx, oracle):
    """configs[4]: 2000x2000 grid @ 2.5 cm, +-10 m / +-180 deg at 2.5 cm / 0.25 deg,
    1080 beams, L = 4: 1441 x 804 x 804 = 9.3e8 candidate poses (2.0 TB of
    algorithmic gathers). Checked: window arithmetic, the winner's exact f64
    score and integer sums against the oracle's Score() at that pose, that no
    candidate of a 3x3x3 neighbourhood + 2000 random ones scores higher
    (oracle, including the offset nearest to the pose the scan was cast from --
    the room is symmetric and most beams are range-clipped, so the global
    optimum need not be that pose), idempotence."""
    case = synth.csm_case(7, rows=2000, cols=2000, res=0.025, n_beams=1080, fov=1.5 * math.pi,
                          max_range=5.7296, init_error=(3.1, -2.7, 1.3), n_boxes=10)
    rx, ry, rt, L = 20.0, 20.0, 2 * math.pi, 4
    m = api.ScanMatcherCorrelativeHIP("cfg5", L, rx, ry, rt, ctx=gpu_ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"], map_id=555)
    raw = out["raw"]
    assert (out["win_x"], out["win_y"]) == (400, 400)
    assert out["win_theta"] == api.host_window(rt, out["step_theta"])
    assert out["candidates"] == (2 * out["win_theta"] + 1) * 804 * 804
    assert out["pose_found"] == 1
    # exact score of the winner
    s, known = oracle.score_at(case["grid"], case["geom"], case["angles"], case["ranges"],
                               out["best_sensor_pose"])
    assert raw["score"] == s
    assert raw["known"] == known
    assert raw["key"] == 32268 * raw["known"] + 499 * raw["sum_values"]
    # nothing nearby or at random beats it (integer offsets: CSM projection)
    sx, sy, st = out["step_x"], out["step_y"], out["step_theta"]
    rng = np.random.RandomState(0)
    cands = [(dx, dy, dt) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dt in (-1, 0, 1)]
    cands += [(int(rng.randint(-400, 404)) - raw["best_x"], int(rng.randint(-400, 404)) - raw["best_y"],
               int(rng.randint(-out["win_theta"], out["win_theta"] + 1)) - raw["best_theta"])
              for _ in range(300)]
    # the offset nearest to the true pose
    truth = np.asarray(case["truth"])
    tx = int(round((truth[0] - out["sensor_pose"][0]) / sx)) - raw["best_x"]
    ty = int(round((truth[1] - out["sensor_pose"][1]) / sy)) - raw["best_y"]
    tt = int(round((truth[2] - out["sensor_pose"][2]) / st)) - raw["best_theta"]
    cands.append((tx, ty, tt))
    for dx, dy, dt in cands:
        t = raw["best_theta"] + dt
        pose_t = (out["sensor_pose"][0], out["sensor_pose"][1], out["sensor_pose"][2] + st * t)
        col, row = oracle.project(case["geom"], pose_t, case["angles"], case["ranges"])
        r = row + raw["best_y"] + dy
        c = col + raw["best_x"] + dx
        ok = (r >= 0) & (r < 2000) & (c >= 0) & (c < 2000)
        v = np.where(ok, case["grid"][np.clip(r, 0, 1999), np.clip(c, 0, 1999)], 0).astype(np.int64)
        key = 32268 * int((v != 0).sum()) + 499 * int(v.sum())
        assert key <= raw["key"]
    again = m.optimize_pose(None, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                            case["init_pose"], map_id=555)
    assert again["raw"] == raw
    gpu_ctx.release_grid(555)


def test_loop_detector_correlative_default_settings(gpu_ctx, oracle):
    """LoopDetectorCorrelative with the reference's default settings
    (launcher_settings_default.json:101-114: L = 5, 2.5 m x 2.5 m x 0.5 rad,
    thresholds 0.55 / 0.6 -- lowered here so that some queries are found)."""
    cases = [synth.csm_case(110 + i, n_beams=720, init_error=(0.5, -0.4, 0.08)) for i in range(5)]
    queries, grids = [], {}
    for i, c in enumerate(cases):
        grids[700 + i] = c["grid"]
        queries.append(dict(map_id=700 + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=c["rel_pose"], init_pose=c["init_pose"]))
    det = parallel.LoopDetectorCorrelativeHIP("ldc", gpu_ctx, 5, 2.5, 2.5, 0.5, 0.35, 0.6)
    outs, found = det.detect(queries, grids)
    want_found = []
    for i, (c, o) in enumerate(zip(cases, outs)):
        lit = oracle.csm(c, 2.5, 2.5, 0.5, 5, 0.35, 0.6)
        assert o["pose_found"] == lit["found"]
        assert (o["raw"]["best_x"], o["raw"]["best_y"], o["raw"]["best_theta"]) == \
            (lit["bestX"], lit["bestY"], lit["bestT"])
        assert o["raw"]["score"] == lit["scoreMax"]
        assert o["estimated_pose"] == lit["estimatedPose"]
        if lit["found"]:
            want_found.append(i)
    assert found == want_found
    for k in grids:
        gpu_ctx.release_grid(k)


def _loop_batch(first_seed, n, id_base):
    rng = np.random.RandomState(3)
    queries, grids, cases = [], {}, []
    for i in range(n):
        c = synth.csm_case(first_seed + i, n_beams=1080, fov=1.5 * math.pi)
        c["init_pose"] = tuple(np.asarray(c["truth"]) + rng.uniform(-0.6, 0.6, 3) * (1, 1, 0.15))
        cases.append(c)
        grids[id_base + i] = c["grid"]
        queries.append(dict(map_id=id_base + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=(0.0, 0.0, 0.0), init_pose=c["init_pose"]))
    return queries, grids, cases


def _check_loop_records(records, found, cases, oracle, every):
    assert found == [i for i, r in enumerate(records) if r["found"]]
    for r in records:
        assert r["key"] == 32268 * r["known"] + 499 * r["sum_values"]
        if r["found"]:
            assert r["score"] > 0.55 and r["known"] / 1080 > 0.6
    for i in range(0, len(records), every):
        want = oracle.bnb(cases[i], 2.5, 2.5, 0.5, 2, 0.55, 0.6)
        r = records[i]
        assert r["found"] == want["found"], i
        assert (r["best_x"], r["best_y"], r["best_theta"]) == (want["bestX"], want["bestY"], want["bestT"]), i
        assert r["score"] == want["scoreMax"], i


def test_config3_batch_of_256_submaps_all_against_oracle(gpu_ctx, oracle):
    """configs[2]: 1080-beam scans vs 256 candidate submaps, 3-level grids
    (H = 2), 2.5 m x 2.5 m x 0.5 rad, thresholds 0.55 / 0.6, through the sharding
    detector class (world size 1 here). EVERY query is checked against the
    literal CPU search (std::priority_queue): found flag, best indices, f64 score."""
    n = 256
    queries, grids, cases = _loop_batch(1000, n, 20000)
    det = parallel.LoopDetectorBranchBoundHIP("ld", gpu_ctx, 2.5, 2.5, 0.5, 2, 0.55, 0.6)
    records, found = det.detect(queries, grids)
    assert len(records) == n and len(found) > 0
    _check_loop_records(records, found, cases, oracle, 1)
    for k in grids:
        gpu_ctx.release_grid(k)


def test_config4_batch_of_2048_submaps_world_size_1(oracle):
    """configs[3] at world size 1: all 2048 queries of the sharded batch through
    LoopDetectorBranchBoundHIP.detect on one GPU (what every rank does for its
    block), every 64th checked against the literal CPU search, all records
    self-consistent; the device copy of the records (the all-gather's send
    buffer, csm_copy_last_batch_records) must equal the host summaries."""
    import torch
    n = 2048
    ctx = api.Context(0)
    queries, grids, cases = _loop_batch(5000, n, 0)
    det = parallel.LoopDetectorBranchBoundHIP("ld", ctx, 2.5, 2.5, 0.5, 2, 0.55, 0.6)
    records, found = det.detect(queries, grids)
    assert len(records) == n and len(found) > 0
    _check_loop_records(records, found, cases, oracle, 64)
    dev = torch.zeros(n * parallel.RECORD_BYTES, dtype=torch.uint8, device="cuda:0")
    ctx.copy_last_batch_records(dev.data_ptr())
    ctx.synchronize()
    assert parallel.bytes_to_records(dev.cpu().numpy()) == records
    ctx.close()
