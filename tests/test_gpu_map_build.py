"""GPU parity of the map building step (csm_construct_map_from_scans) against
the literal CPU restatement of GridMapBuilder::ConstructMapFromScans: resized
geometry, every cell value, and the update / saturation counters."""
import math

import numpy as np
import pytest

from csm_hip import api, synth

pytestmark = pytest.mark.gpu


def _check(gpu_ctx, oracle, case, map_id, **kw):
    want_shape, want_grid, stats = oracle.construct_map(case["shape"], case["map_pose"], case["nodes"], **{
        {"usable_range_max": "usable_max", "usable_range_min": "usable_min", "prob_hit": "prob_hit",
         "prob_miss": "prob_miss", "subpixel_scale": "subpixel"}[k]: v for k, v in kw.items()})
    shape, info = gpu_ctx.construct_map_from_scans(map_id, case["shape"], case["map_pose"], case["nodes"], **kw)
    assert shape == want_shape
    got = gpu_ctx.download_level(map_id, 0)
    assert got.shape == want_grid.shape
    bad = np.argwhere(got != want_grid)
    assert bad.size == 0, (len(bad), bad[:5], got[tuple(bad[0])], want_grid[tuple(bad[0])])
    assert info["rays"] == stats["rays"]
    assert info["cell_updates"] == stats["updates"]
    assert info["saturated_reads"] == stats["oob_reads"]
    ys, xs = np.nonzero(want_grid)
    assert (info["first_known_row"], info["first_known_col"]) == (ys.min(), xs.min())
    return shape, got, info


@pytest.mark.parametrize("seed,n_scans,n_beams", [(0, 1, 90), (1, 3, 360), (2, 10, 1080), (3, 10, 360)])
def test_map_matches_literal_builder(gpu_ctx, oracle, seed, n_scans, n_beams):
    case = synth.map_case(seed, n_scans=n_scans, n_beams=n_beams,
                          rel_pose=(0.1, -0.05, 0.02) if seed % 2 else (0.0, 0.0, 0.0))
    _check(gpu_ctx, oracle, case, 300 + seed)
    gpu_ctx.release_grid(300 + seed)


def test_map_params_and_rebuild_in_place(gpu_ctx, oracle):
    """Other builder settings, then a second build under the same id in the
    frame the first one left (the latest-map cycle of UpdateLatestMap)."""
    case = synth.map_case(11, n_scans=6, n_beams=720, noise=0.01)
    shape, _, _ = _check(gpu_ctx, oracle, case, 310, usable_range_max=4.0, prob_hit=0.7, prob_miss=0.4,
                         subpixel_scale=10)
    nxt = dict(case)
    nxt["shape"] = shape
    nxt["nodes"] = case["nodes"][2:] + synth.map_case(11, n_scans=8, n_beams=720)["nodes"][6:]
    nxt["map_pose"] = nxt["nodes"][0]["pose"]
    _check(gpu_ctx, oracle, nxt, 310)
    gpu_ctx.release_grid(310)


def test_map_feeds_the_matcher(gpu_ctx, oracle):
    """Build on the device, match against the resident result, compare with
    the CPU matcher on the CPU-built map (coarse level rebuilt after each build)."""
    case = synth.map_case(21, n_scans=10, n_beams=720)
    for rnd in range(2):
        shape, grid, _ = _check(gpu_ctx, oracle, case, 320)
        geom = (shape["res"], shape["off_x"], shape["off_y"])
        nd = case["nodes"][-1]
        init = (nd["pose"][0] + 0.08, nd["pose"][1] - 0.06, nd["pose"][2] + 0.015)
        # the matcher works in the map-local frame: map pose = first node's pose
        mp = case["map_pose"]
        c, s_ = math.cos(mp[2]), math.sin(mp[2])
        dx, dy = init[0] - mp[0], init[1] - mp[1]
        local = (c * dx + s_ * dy, -s_ * dx + c * dy, init[2] - mp[2])
        out = gpu_ctx.correlative_match(320, geom, nd["angles"], nd["ranges"], nd["rel_pose"], local,
                                        1.0, 1.0, 0.25, 4, 0.0, 0.0)
        want = oracle.csm(dict(grid=grid, geom=geom, angles=nd["angles"], ranges=nd["ranges"],
                               rel_pose=nd["rel_pose"], init_pose=local), 1.0, 1.0, 0.25, 4)
        assert out["pose_found"] == want["found"] == 1
        assert list(out["estimated_pose"]) == want["estimatedPose"]
        assert out["raw"]["score"] == want["scoreMax"]
        case = dict(case)
        case["shape"] = shape
        case["nodes"] = case["nodes"][1:]
        case["map_pose"] = case["nodes"][0]["pose"]
    gpu_ctx.release_grid(320)


def test_map_saturation_is_counted(gpu_ctx, oracle):
    """Many scans from one place drive wall cells to 65535 and beyond: the
    library and the restatement extend the reference's odds table the same way
    and report how often."""
    case = synth.map_case(5, n_scans=30, n_beams=720, step=0.0)
    _, grid, info = _check(gpu_ctx, oracle, case, 330)
    assert info["saturated_reads"] > 0 and (grid == 65535).any()
    gpu_ctx.release_grid(330)


def test_map_rejects_bad_input(gpu_ctx):
    case = synth.map_case(1, n_scans=2, n_beams=90)
    with pytest.raises(api.CsmError):
        gpu_ctx.construct_map_from_scans(340, case["shape"], case["map_pose"], [])
    with pytest.raises(api.CsmError):
        gpu_ctx.construct_map_from_scans(340, case["shape"], case["map_pose"], case["nodes"], subpixel_scale=0)
    assert not gpu_ctx.has_grid(340)
