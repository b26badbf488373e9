"""Row A3 on its own: the device projection (k_project, certificate per entry)
against ComputeScanIndices as the host computes it with glibc
(csm_host_project; src/mapping/scan_matcher_correlative.cpp:277-297 +
include/.../sensor/sensor_data.hpp:189-203) and against the oracle's
restatement, entry by entry. The contract: every entry the device does NOT list
as uncertified equals the host's index; hence every mismatch is in the list."""
import math

import numpy as np
import pytest

from csm_hip import api, synth

pytestmark = pytest.mark.gpu


def _compare(ctx, oracle, geom, pose, step_theta, win_theta, angles, ranges):
    col, row, unc, count = ctx.project_scan(geom, pose, step_theta, win_theta, angles, ranges)
    hcol, hrow = api.host_project(geom, pose, step_theta, win_theta, angles, ranges)
    assert count == unc.size, "uncertified list overflowed the test's capacity"
    listed = np.zeros(col.size, bool)
    listed[unc] = True
    diff = ((col != hcol) | (row != hrow)).reshape(-1)
    # certified entries are exact; mismatches are a subset of the list
    assert not np.any(diff & ~listed), "a certified entry differs from the host's index"
    # the oracle's own projection agrees with the library's host projection (spot rows)
    n = len(angles)
    for t in (0, win_theta, 2 * win_theta):
        p = (pose[0], pose[1], pose[2] + step_theta * (t - win_theta))
        oc, orow = oracle.project(geom, p, angles, ranges)
        assert np.array_equal(oc, hcol[t]) and np.array_equal(orow, hrow[t])
    return col.size, int(unc.size), int(diff.sum())


def test_random_scans_one_million_entries(gpu_ctx, oracle):
    total = listed = wrong = 0
    for seed in range(8):
        c = synth.csm_case(300 + seed, n_beams=1080, fov=1.5 * math.pi)
        sx, sy, st = api.host_search_step(c["geom"][0], c["ranges"])
        wt = api.host_window(math.radians(60.0), st)
        a, b, d = _compare(gpu_ctx, oracle, c["geom"], c["init_pose"], st, wt, c["angles"], c["ranges"])
        total, listed, wrong = total + a, listed + b, wrong + d
    assert total >= 1000000
    # the certificate is tight: a few entries per million, not a sizeable share
    assert listed < total // 1000


def test_cell_edge_aligned_geometry(gpu_ctx, oracle):
    """Everything on exact multiples of the resolution: hit points sit on cell
    edges, where the two libms may floor differently. Those entries must be
    listed (the list is allowed to be long here)."""
    for seed in (1, 2, 3):
        c = synth.csm_case(400 + seed, n_beams=720, origin="aligned", truth=(0.0, 0.0, 0.0),
                           init_error=(0.05, -0.10, 0.0))
        sx, sy, st = api.host_search_step(c["geom"][0], c["ranges"])
        wt = api.host_window(math.radians(20.0), st)
        total, listed, wrong = _compare(gpu_ctx, oracle, c["geom"], c["init_pose"], st, wt, c["angles"],
                                        c["ranges"])
        assert listed > 0          # axis-parallel beams do land on edges


def test_large_offsets_and_angles(gpu_ctx, oracle):
    """|offset| ~ 1e3 m (ulp of the coordinate grows 1e3-fold) and |theta| ~ 1e2 rad
    (argument reduction of sin / cos): the certificate must scale with both."""
    rng = np.random.RandomState(5)
    for k in range(6):
        c = synth.csm_case(500 + k, n_beams=1080, fov=1.5 * math.pi)
        shift = (1000.0 + 37.0 * k, -2000.0 + 11.0 * k)
        geom = (c["geom"][0], c["geom"][1] + shift[0], c["geom"][2] + shift[1])
        turn = (100.0 + k) * (1 if k % 2 else -1)
        pose = (c["init_pose"][0] + shift[0], c["init_pose"][1] + shift[1], c["init_pose"][2] + turn)
        angles = c["angles"]                        # beams keep their sensor-frame angles
        sx, sy, st = api.host_search_step(geom[0], c["ranges"])
        wt = api.host_window(math.radians(30.0), st)
        _compare(gpu_ctx, oracle, geom, pose, st, wt, angles, c["ranges"] * (0.5 + rng.rand()))


def test_non_finite_ranges_are_rejected(gpu_ctx):
    c = synth.csm_case(9, n_beams=360)
    for bad in (float("inf"), float("nan")):
        r = c["ranges"].copy()
        r[17] = bad
        with pytest.raises(api.CsmError) as e:
            gpu_ctx.project_scan(c["geom"], c["init_pose"], 0.01, 3, c["angles"], r)
        assert e.value.code == -22
        gpu_ctx.upload_grid(4242, c["grid"])
        with pytest.raises(api.CsmError) as e:
            gpu_ctx.correlative_match(4242, c["geom"], c["angles"], r, c["rel_pose"], c["init_pose"],
                                      1.0, 1.0, 0.2, 4)
        assert e.value.code == -22
        gpu_ctx.release_grid(4242)


def test_non_finite_scan_in_a_large_batch_names_the_first_offender(gpu_ctx):
    """Batches of >= 256 queries check their scans on several host threads: the
    error must still name the FIRST bad query, and it comes before any map lookup."""
    c = synth.csm_case(11, n_beams=360)
    good_a, good_r = c["angles"], c["ranges"]
    bad_nan, bad_inf = good_r.copy(), good_r.copy()
    bad_nan[5] = float("nan")
    bad_inf[300] = float("inf")
    qs = []
    for i in range(600):
        r = bad_inf if i == 217 else bad_nan if i == 431 else good_r
        qs.append(dict(map_id=990000 + i, geom=c["geom"], angles=good_a, ranges=r, rel_pose=c["rel_pose"],
                       init_pose=c["init_pose"]))
    for call in (lambda: gpu_ctx.bnb_match_batch(qs, 1.0, 1.0, 0.2, 2, 0.3, 0.5),
                 lambda: gpu_ctx.correlative_match_batch(qs, 1.0, 1.0, 0.2, 4, 0.3, 0.5)):
        with pytest.raises(api.CsmError) as e:
            call()
        assert e.value.code == -22 and "query 217" in str(e.value)
