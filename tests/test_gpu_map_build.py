"""GPU parity of the map building step (csm_construct_map_from_scans) against
the literal CPU restatement of GridMapBuilder::ConstructMapFromScans: resized
geometry, every cell value, and the update / saturation counters."""
import math

import numpy as np
import pytest

from csm_hip import _lib as L, api, synth

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
    first = (ys.min(), xs.min()) if ys.size else want_grid.shape      # nothing known: rows / cols
    assert (info["first_known_row"], info["first_known_col"]) == first
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


@pytest.mark.parametrize("seed", range(24))
def test_map_randomised(gpu_ctx, oracle, seed):
    """Random trajectories, resolutions, block sizes, sub-pixel scales, sensor
    offsets, range noise and starting frames; every cell must agree."""
    rng = np.random.RandomState(1000 + seed)
    res = float(rng.choice([0.025, 0.05, 0.1, 0.2]))
    case = synth.map_case(seed, n_scans=int(rng.randint(1, 8)), n_beams=int(rng.choice([64, 181, 360, 500])),
                          fov=float(rng.choice([math.pi, 1.5 * math.pi, 2 * math.pi])),
                          max_range=float(rng.choice([3.0, 6.0, 12.0])), res=res,
                          step=float(rng.choice([0.0, 0.05, 0.3])),
                          rel_pose=tuple(rng.uniform(-0.2, 0.2, 3)), noise=float(rng.choice([0.0, 0.02])))
    shape = dict(case["shape"])
    shape["log2_block"] = int(rng.choice([2, 3, 4, 5]))
    n = 1 << shape["log2_block"]
    shape["rows"] = shape["cols"] = -(-int(math.ceil(1.0 / res)) // n) * n
    # a frame left behind by earlier builds: any offset on the block lattice or off it
    shape["off_x"] = float(rng.choice([0.0, -3.2, 1.6, 0.0137]))
    shape["off_y"] = float(rng.choice([0.0, 4.8, -0.8, -0.0219]))
    case["shape"] = shape
    kw = dict(usable_range_max=float(rng.choice([2.5, 20.0])), usable_range_min=float(rng.choice([0.01, 0.5])),
              prob_hit=float(rng.choice([0.55, 0.62, 0.9])), prob_miss=float(rng.choice([0.1, 0.46, 0.49])),
              subpixel_scale=int(rng.choice([1, 7, 100, 1000])))
    try:
        oracle.construct_map(case["shape"], case["map_pose"], case["nodes"], usable_min=kw["usable_range_min"],
                             usable_max=kw["usable_range_max"])
    except ValueError:
        # nothing usable: the bounding box is empty and the reference asserts
        with pytest.raises(api.CsmError):
            gpu_ctx.construct_map_from_scans(400 + seed, case["shape"], case["map_pose"], case["nodes"], **kw)
        return
    _check(gpu_ctx, oracle, case, 400 + seed, **kw)
    gpu_ctx.release_grid(400 + seed)


def test_map_projection_paths(gpu_ctx, oracle):
    """The hit points come from the device under a certificate; forcing the host
    projection must give the same map."""
    case = synth.map_case(31, n_scans=5, n_beams=500, noise=0.01)
    _, grid, info = _check(gpu_ctx, oracle, case, 450)
    assert info["device_projection"] == 1
    host_ctx = api.Context(0, tuning_off=L.TUNE_MAP_HOST_PROJECTION)
    _, grid2, info2 = _check(host_ctx, oracle, case, 450)
    assert info2["device_projection"] == 0 and np.array_equal(grid, grid2)
    host_ctx.close()
    gpu_ctx.release_grid(450)


def test_map_aligned_geometry(gpu_ctx, oracle):
    """Sensor and walls on exact multiples of the resolution: hit points on cell
    edges (the device cannot certify them: they are redone on the host, or, past
    the list's capacity, everything is), rays through cell corners."""
    segs = [(-2.0, -1.5, 2.0, -1.5), (2.0, -1.5, 2.0, 1.5), (2.0, 1.5, -2.0, 1.5), (-2.0, 1.5, -2.0, -1.5)]
    nodes = []
    for k, pose in enumerate([(0.0, 0.0, 0.0), (0.25, 0.0, math.pi / 2), (0.25, 0.25, math.pi / 4)]):
        angles, ranges = synth.cast_scan(segs, pose, 720, 2 * math.pi, 10.0)
        nodes.append(dict(pose=pose, angles=angles, ranges=ranges, rel_pose=(0.0, 0.0, 0.0),
                          min_range=0.0, max_range=9.0))
    case = dict(nodes=nodes, map_pose=(0.0, 0.0, 0.0),
                shape=dict(res=0.25, off_x=0.0, off_y=0.0, rows=8, cols=8, log2_block=2))
    for scale in (1, 2, 100):
        _, _, info = _check(gpu_ctx, oracle, case, 440, subpixel_scale=scale)
        assert info["device_projection"] == 1
    small = api.Context(0, map_uncertain_cap=3)
    _, _, info = _check(small, oracle, case, 440, subpixel_scale=100)
    assert info["device_projection"] == 0           # more than 3 beams sit on cell edges
    small.close()
    gpu_ctx.release_grid(440)


def test_map_degenerate_boxes(gpu_ctx, oracle):
    """One scan whose only usable beams point along one axis: the bounding box
    is decided from exact host values."""
    angles = np.array([0.0, 0.0, 0.0])              # all along +x: sin(0) = 0 exactly
    ranges = np.array([2.0, 3.0, 1.5])
    node = dict(pose=(0.3, 0.2, 0.0), angles=angles, ranges=ranges, rel_pose=(0.0, 0.0, 0.0),
                min_range=0.0, max_range=10.0)
    shape = dict(res=0.05, off_x=0.0, off_y=0.0, rows=32, cols=32, log2_block=4)
    case = dict(nodes=[node], map_pose=(0.0, 0.0, 0.0), shape=shape)
    with pytest.raises(ValueError):
        oracle.construct_map(shape, case["map_pose"], case["nodes"])     # y extent is empty: Assert
    with pytest.raises(api.CsmError):
        gpu_ctx.construct_map_from_scans(460, shape, case["map_pose"], case["nodes"])
    two = dict(case)
    two["nodes"] = [node, dict(node, pose=(0.3, 0.7, 0.0))]              # a second sensor position opens it
    _check(gpu_ctx, oracle, two, 460)
    gpu_ctx.release_grid(460)


def _check_update(gpu_ctx, oracle, map_id, shape, grid, map_pose, node, **kw):
    okw = {{"usable_range_max": "usable_max", "usable_range_min": "usable_min", "prob_hit": "prob_hit",
            "prob_miss": "prob_miss", "subpixel_scale": "subpixel"}[k]: v for k, v in kw.items()}
    want_shape, want_grid, stats = oracle.update_map(shape, grid, map_pose, node, **okw)
    got_shape, info = gpu_ctx.update_map_with_scan(map_id, shape, map_pose, node, **kw)
    assert got_shape == want_shape
    got = gpu_ctx.download_level(map_id, 0)
    bad = np.argwhere(got != want_grid)
    assert bad.size == 0, (len(bad), bad[:5])
    assert (info["rays"], info["cell_updates"], info["saturated_reads"]) == \
        (stats["rays"], stats["updates"], stats["oob_reads"])
    ys, xs = np.nonzero(want_grid)
    first = (ys.min(), xs.min()) if ys.size else want_grid.shape
    assert (info["first_known_row"], info["first_known_col"]) == first
    return want_shape, want_grid, stats


@pytest.mark.parametrize("seed,n_beams", [(0, 360), (1, 1080), (2, 181)])
def test_local_map_grows_scan_by_scan(gpu_ctx, oracle, seed, n_beams):
    """GridMapBuilder::UpdateGridMap: a fresh 1 m x 1 m local map takes one scan
    after the other; it expands when a scan does not fit and keeps its cells."""
    case = synth.map_case(seed, n_scans=8, n_beams=n_beams, step=0.4,
                          rel_pose=(0.08, 0.0, 0.0) if seed else (0.0, 0.0, 0.0))
    shape = case["shape"]
    grid = np.zeros((shape["rows"], shape["cols"]), np.uint16)
    gpu_ctx.upload_grid(500 + seed, grid)
    grew = 0
    for nd in case["nodes"]:
        shape, grid, stats = _check_update(gpu_ctx, oracle, 500 + seed, shape, grid, case["map_pose"], nd,
                                           usable_range_max=6.0)
        grew += (stats["row_min"], stats["col_min"]) != (0, 0) or False
    assert grew >= 1
    # the finished local map serves the loop detector without an upload
    geom = (shape["res"], shape["off_x"], shape["off_y"])
    nd = case["nodes"][3]
    mp = case["map_pose"]
    c, s_ = math.cos(mp[2]), math.sin(mp[2])
    dx, dy = nd["pose"][0] + 0.2 - mp[0], nd["pose"][1] - 0.15 - mp[1]
    init = (c * dx + s_ * dy, -s_ * dx + c * dy, nd["pose"][2] + 0.03 - mp[2])
    q = dict(map_id=500 + seed, geom=geom, angles=nd["angles"], ranges=nd["ranges"], rel_pose=nd["rel_pose"],
             init_pose=init)
    out = gpu_ctx.bnb_match_batch([q], 1.5, 1.5, 0.3, 2, 0.3, 0.5)[0]
    want = oracle.bnb(dict(grid=grid, geom=geom, angles=nd["angles"], ranges=nd["ranges"],
                           rel_pose=nd["rel_pose"], init_pose=init), 1.5, 1.5, 0.3, 2, 0.3, 0.5)
    assert out["pose_found"] == want["found"]
    if want["found"]:
        assert list(out["estimated_pose"]) == want["estimatedPose"]
        assert out["raw"]["score"] == want["scoreMax"]
    gpu_ctx.release_grid(500 + seed)


def test_update_onto_uploaded_map_and_errors(gpu_ctx, oracle):
    """An uploaded (host-built) map as the starting point; growth towards
    negative indices; shape mismatch and missing map are refused."""
    case = synth.map_case(9, n_scans=4, n_beams=360)
    shape, grid, _ = oracle.construct_map(case["shape"], case["map_pose"], case["nodes"][:2])
    gpu_ctx.upload_grid(510, grid)
    far = dict(case["nodes"][3], pose=(case["nodes"][3]["pose"][0] - 6.0, case["nodes"][3]["pose"][1] - 5.0,
                                       case["nodes"][3]["pose"][2]))
    shape2, grid2, stats = _check_update(gpu_ctx, oracle, 510, shape, grid, case["map_pose"], far)
    assert stats["row_min"] < 0 or stats["col_min"] < 0
    _check_update(gpu_ctx, oracle, 510, shape2, grid2, case["map_pose"], case["nodes"][2])
    with pytest.raises(api.CsmError):
        gpu_ctx.update_map_with_scan(510, shape, case["map_pose"], case["nodes"][2])     # stale shape
    with pytest.raises(api.CsmError):
        gpu_ctx.update_map_with_scan(511, shape2, case["map_pose"], case["nodes"][2])    # not resident
    gpu_ctx.release_grid(510)


def test_map_many_scans_fine_resolution(gpu_ctx, oracle):
    """60 scans x 1080 beams at 2.5 cm (65k rays, windows larger than the LDS
    budget, cells with hundreds of hits)."""
    case = synth.map_case(40, n_scans=60, n_beams=1080, res=0.025, max_range=12.0, step=0.05)
    shape, grid, info = _check(gpu_ctx, oracle, case, 470)
    assert info["rays"] > 60000 and info["cell_updates"] > 4e6
    gpu_ctx.release_grid(470)


def test_frontend_loop_over_a_trajectory(gpu_ctx, oracle):
    """25 scans in a row, as the frontend runs them: rebuild the latest map from
    the last 10 scan nodes in the frame the previous build left, match the new
    scan against it, move on. Every step is compared with the CPU pipeline."""
    case = synth.map_case(77, n_scans=26, n_beams=360, step=0.15)
    nodes = case["nodes"]
    shape_dev = shape_cpu = case["shape"]
    for k in range(1, 26):
        window = nodes[max(0, k - 10):k]
        map_pose = window[0]["pose"]
        shape_cpu, grid, stats = oracle.construct_map(shape_cpu, map_pose, window)
        shape_dev, info = gpu_ctx.construct_map_from_scans(600, shape_dev, map_pose, window)
        assert shape_dev == shape_cpu, k
        assert np.array_equal(gpu_ctx.download_level(600, 0), grid), k
        assert info["cell_updates"] == stats["updates"]
        new = nodes[k]
        c, s_ = math.cos(map_pose[2]), math.sin(map_pose[2])
        dx, dy = new["pose"][0] + 0.04 - map_pose[0], new["pose"][1] - 0.03 - map_pose[1]
        init = (c * dx + s_ * dy, -s_ * dx + c * dy, new["pose"][2] + 0.01 - map_pose[2])
        geom = (shape_cpu["res"], shape_cpu["off_x"], shape_cpu["off_y"])
        out = gpu_ctx.correlative_match(600, geom, new["angles"], new["ranges"], new["rel_pose"], init,
                                        0.5, 0.5, 0.2, 4, 0.0, 0.0)
        want = oracle.csm(dict(grid=grid, geom=geom, angles=new["angles"], ranges=new["ranges"],
                               rel_pose=new["rel_pose"], init_pose=init), 0.5, 0.5, 0.2, 4)
        assert out["pose_found"] == want["found"], k
        assert list(out["estimated_pose"]) == want["estimatedPose"], k
        assert out["raw"]["score"] == want["scoreMax"], k
    gpu_ctx.release_grid(600)
