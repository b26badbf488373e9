"""GPU parity on the inputs that break a naive parallel arg-max: integer-key
ties, coarse nodes that fail to bound their fine candidates (negative edge
band), cell-edge-aligned geometry (branch-and-bound per-node projection),
nothing found, scans off the map, ragged sizes. The checker is always the
LITERAL sequential restatement (oracle.csm / oracle.bnb)."""
import math

import numpy as np
import pytest

from csm_hip import _lib as L
from csm_hip import api, synth

pytestmark = pytest.mark.gpu

stats = {"tie": 0, "band": 0, "delta": 0, "literal": 0, "band_differs": 0}


def _check_csm(ctx, oracle, case, rx, ry, rt, Lr, score_thr=0.0, known_thr=0.0, map_id=50):
    m = api.ScanMatcherCorrelativeHIP("edge", Lr, rx, ry, rt, ctx=ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"], score_threshold=score_thr,
                          known_rate_threshold=known_thr)
    lit = oracle.csm(case, rx, ry, rt, Lr, score_thr, known_thr)
    raw = out["raw"]
    assert out["pose_found"] == lit["found"], (raw, lit)
    assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"]), (raw, lit)
    assert raw["score"] == lit["scoreMax"]
    assert out["estimated_pose"] == lit["estimatedPose"]
    return raw, lit


@pytest.mark.parametrize("seed,levels", [(40, 2), (41, 3), (42, 4), (43, 8), (44, 2), (45, 3)])
def test_csm_integer_key_ties(gpu_ctx, oracle, seed, levels):
    case = synth.csm_case(seed, levels=levels, interior_unknown=0.0)
    raw, _ = _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), 4)
    stats["tie"] += bool(raw["flags"] & L.FLAG_KEY_TIE)


@pytest.mark.parametrize("seed,Lr", [(50, 4), (51, 4), (52, 5), (53, 8), (54, 3), (55, 4), (56, 2), (57, 6)])
def test_csm_negative_edge_band(gpu_ctx, oracle, seed, Lr):
    case = synth.csm_case(seed, rows=256, cols=288, origin="low_edge", half_x=5.2, half_y=4.4,
                          init_error=(0.23, 0.19, 0.03))
    raw, lit = _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), Lr)
    cf = oracle.csm_closed_form(case, 1.0, 1.0, math.radians(10), Lr)
    stats["band"] += bool(raw["flags"] & L.FLAG_EDGE_BAND)
    stats["literal"] += bool(raw["flags"] & L.FLAG_LITERAL)
    stats["band_differs"] += (cf["bestX"], cf["bestY"], cf["bestT"]) != (lit["bestX"], lit["bestY"], lit["bestT"])


def test_csm_thresholds_and_not_found(gpu_ctx, oracle):
    case = synth.csm_case(60)
    raw, lit = _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), 4, 0.95, 0.0)
    assert lit["found"] == 0 and raw["found"] == 0
    _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), 4, 0.2, 0.9)
    _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), 4, 0.1, 0.99)
    _check_csm(gpu_ctx, oracle, case, 0.5, 1.5, math.radians(4), 5, 0.3, 0.5)


def test_csm_scan_off_the_map(gpu_ctx, oracle):
    case = synth.csm_case(61)
    case["init_pose"] = (40.0, -35.0, 0.3)
    raw, lit = _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), 4)
    assert raw["found"] == 0


@pytest.mark.parametrize("n_beams,rows,cols,Lr", [(37, 112, 96, 3), (1, 64, 64, 2), (513, 176, 240, 7),
                                                   (2049, 128, 128, 4)])
def test_csm_ragged_sizes(gpu_ctx, oracle, n_beams, rows, cols, Lr):
    case = synth.csm_case(62, rows=rows, cols=cols, n_beams=n_beams, max_range=2.2,
                          half_x=cols * 0.05 * 0.3, half_y=rows * 0.05 * 0.3, n_boxes=1,
                          init_error=(0.06, -0.04, 0.01))
    _check_csm(gpu_ctx, oracle, case, 0.6, 0.4, math.radians(6), Lr)


def _check_bnb(ctx, oracle, cases, H, thr, rng=(2.5, 2.5, 0.5), base_id=3000):
    qs = []
    for i, c in enumerate(cases):
        ctx.upload_grid(base_id + i, c["grid"])
        qs.append(dict(map_id=base_id + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                       rel_pose=c["rel_pose"], init_pose=c["init_pose"]))
    outs = ctx.bnb_match_batch(qs, rng[0], rng[1], rng[2], H, thr[0], thr[1])
    for c, o in zip(cases, outs):
        want = oracle.bnb(c, rng[0], rng[1], rng[2], H, thr[0], thr[1])
        raw = o["raw"]
        assert o["pose_found"] == want["found"], (raw, want)
        assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (want["bestX"], want["bestY"], want["bestT"]), (raw, want)
        assert raw["score"] == want["scoreMax"]
        assert o["estimated_pose"] == want["estimatedPose"]
        stats["delta"] += bool(raw["flags"] & L.FLAG_PROJ_DELTA)
        stats["tie"] += bool(raw["flags"] & L.FLAG_KEY_TIE)
        stats["band"] += bool(raw["flags"] & L.FLAG_EDGE_BAND)
        stats["literal"] += bool(raw["flags"] & L.FLAG_LITERAL)
    for i in range(len(cases)):
        ctx.release_grid(base_id + i)
    return outs


def test_bnb_ties_on_quantised_maps(gpu_ctx, oracle):
    cases = [synth.csm_case(70 + i, levels=lv, interior_unknown=0.0, init_error=(0.3, 0.2, 0.04))
             for i, lv in enumerate([2, 3, 4, 2])]
    _check_bnb(gpu_ctx, oracle, cases, 2, (0.3, 0.5))


def test_bnb_negative_edge_band(gpu_ctx, oracle):
    cases = [synth.csm_case(80 + i, rows=256, cols=288, origin="low_edge", half_x=5.2, half_y=4.4,
                            init_error=(0.23, 0.19, 0.03)) for i in range(4)]
    _check_bnb(gpu_ctx, oracle, cases, 3, (0.3, 0.5), rng=(1.5, 1.5, 0.3))


def test_bnb_twenty_queries_some_in_the_edge_band(gpu_ctx, oracle):
    """>= 16 queries in one batch: the coarse levels are launched with the theta
    axis folded (a workgroup loops over slices) and normally exit at once; the
    low-edge queries here make those workgroups run their loop."""
    cases = []
    for i in range(20):
        if i % 3 == 0:
            cases.append(synth.csm_case(300 + i, n_beams=360, rows=256, cols=288, origin="low_edge", half_x=5.2,
                                        half_y=4.4, init_error=(0.23, 0.19, 0.03)))
        else:
            cases.append(synth.csm_case(300 + i, n_beams=360, rows=256, cols=288, half_x=5.2, half_y=4.4,
                                        init_error=(0.2, -0.1, 0.02)))
    before = stats["band"]
    _check_bnb(gpu_ctx, oracle, cases, 2, (0.3, 0.5), rng=(1.5, 1.5, 0.3), base_id=3300)
    assert stats["band"] > before


def test_bnb_cell_edge_aligned_projection(gpu_ctx, oracle):
    """Offsets, walls and poses on exact multiples of the resolution: hit points
    sit on cell edges, where sensor + x*step + r*cos and (sensor + r*cos) + x
    can floor differently (scan_matcher_branch_bound.cpp:156-176)."""
    cases = []
    for i in range(4):
        c = synth.csm_case(90 + i, origin="aligned", truth=(0.0, 0.0, 0.0),
                           init_error=(0.25, -0.15, 0.0))
        cases.append(c)
    c = synth.csm_case(95, origin="aligned", truth=(0.5, -0.25, math.pi / 2),
                       init_error=(0.0, 0.0, 0.0))
    cases.append(c)
    _check_bnb(gpu_ctx, oracle, cases, 2, (0.3, 0.5))


def test_csm_cell_edge_aligned(gpu_ctx, oracle):
    case = synth.csm_case(96, origin="aligned", truth=(0.0, 0.0, 0.0), init_error=(0.25, -0.15, 0.0))
    _check_csm(gpu_ctx, oracle, case, 1.0, 1.0, math.radians(10), 4)


def test_zz_edge_paths_were_exercised():
    """The inputs above must actually reach the exact paths."""
    print("edge-path statistics:", stats)
    assert stats["tie"] > 0
    assert stats["literal"] > 0


def test_correlative_batch_mixes_fast_and_exact_paths(gpu_ctx, oracle):
    """csm_correlative_match_batch: ordinary, tie-prone, edge-band and
    cell-edge-aligned queries in ONE batch; each must equal the literal sweep."""
    cases = [
        synth.csm_case(120, n_beams=540),
        synth.csm_case(121, n_beams=540, levels=3, interior_unknown=0.0),
        synth.csm_case(122, n_beams=540, rows=256, cols=288, origin="low_edge", half_x=5.2,
                       half_y=4.4, init_error=(0.23, 0.19, 0.03)),
        synth.csm_case(123, n_beams=540, origin="aligned", truth=(0.0, 0.0, 0.0),
                       init_error=(0.25, -0.15, 0.0)),
        synth.csm_case(124, n_beams=540),
    ]
    for Lr, thr in ((4, (0.0, 0.0)), (5, (0.3, 0.5)), (1, (0.2, 0.3))):
        qs = []
        for i, c in enumerate(cases):
            gpu_ctx.upload_grid(8000 + i, c["grid"])
            qs.append(dict(map_id=8000 + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                           rel_pose=c["rel_pose"], init_pose=c["init_pose"]))
        outs = gpu_ctx.correlative_match_batch(qs, 1.0, 1.0, math.radians(10), Lr, thr[0], thr[1])
        for c, o in zip(cases, outs):
            lit = oracle.csm(c, 1.0, 1.0, math.radians(10), Lr, thr[0], thr[1])
            raw = o["raw"]
            assert o["pose_found"] == lit["found"], (Lr, raw, lit)
            assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == \
                (lit["bestX"], lit["bestY"], lit["bestT"]), (Lr, raw, lit)
            assert raw["score"] == lit["scoreMax"]
            assert o["estimated_pose"] == lit["estimatedPose"]
        for i in range(len(cases)):
            gpu_ctx.release_grid(8000 + i)


def test_correlative_batch_of_twenty_with_edge_band_queries(gpu_ctx, oracle):
    """As above for csm_correlative_match_batch with a known-rate threshold of 0
    (the coarse pass then only runs for queries whose beams reach the band)."""
    cases = []
    for i in range(20):
        kw = dict(origin="low_edge", init_error=(0.23, 0.19, 0.03)) if i % 4 == 1 else dict(init_error=(0.1, 0.2, -0.02))
        cases.append(synth.csm_case(340 + i, n_beams=360, rows=256, cols=288, half_x=5.2, half_y=4.4, **kw))
    qs = []
    for i, c in enumerate(cases):
        gpu_ctx.upload_grid(8100 + i, c["grid"])
        qs.append(dict(map_id=8100 + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                       rel_pose=c["rel_pose"], init_pose=c["init_pose"]))
    outs = gpu_ctx.correlative_match_batch(qs, 1.0, 1.0, math.radians(10), 4, 0.0, 0.0)
    band = 0
    for c, o in zip(cases, outs):
        lit = oracle.csm(c, 1.0, 1.0, math.radians(10), 4, 0.0, 0.0)
        raw = o["raw"]
        assert o["pose_found"] == lit["found"], (raw, lit)
        assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"]), (raw, lit)
        assert raw["score"] == lit["scoreMax"]
        band += bool(raw["flags"] & L.FLAG_EDGE_BAND)
    assert band > 0
    for i in range(len(cases)):
        gpu_ctx.release_grid(8100 + i)


def _scattered_case(seed, n_beams):
    """Returns scattered over a disc of 3 m (not a room scan: only the arithmetic is
    under test): the endpoint tile under the disc's centre gets thousands of entries."""
    c = synth.csm_case(seed, rows=128, cols=128, n_beams=n_beams, max_range=3.0, half_x=1.9, half_y=1.7,
                       n_boxes=1, init_error=(0.06, -0.04, 0.01))
    rng = np.random.RandomState(seed)
    r = 0.2 + 2.8 * np.sqrt(rng.rand(n_beams))
    r[int(np.argmax(r))] = 3.0
    c["ranges"] = r.astype(np.float64)
    return c


def test_batch_with_tiles_split_into_several_records(gpu_ctx, oracle):
    """Endpoint tiles with more than 1024 entries are cut into several records by
    k_bin. The batch kernel walks the records of two theta slices together (tile
    number, chunk number): every chunk must be gathered, whether or not the
    neighbouring slice has a chunk of the same number."""
    cases = [_scattered_case(400 + i, n) for i, n in enumerate([9000, 7000] + [2500] * 15)]
    # the premise: some tile of some slice holds more than 1024 distinct row-pair cells
    c = cases[0]
    sx, sy, st = api.host_search_step(c["geom"][0], c["ranges"])
    wx, wy, wt = api.host_window(0.6, sx), api.host_window(0.4, sy), api.host_window(math.radians(6), st)
    col, row = api.host_project(c["geom"], c["init_pose"], st, wt, c["angles"], c["ranges"])
    ny = -(-(2 * wy + 1) // 4) * 4
    nx = -(-(2 * wx + 1) // 4) * 4
    rr, cc = row[0] + (-wy + ny - 1) + ((ny - 1) & 1), col[0] + (-wx + nx - 1)
    tile, cell = (rr // 64) * 1000 + cc // 64, (rr >> 1) * 4096 + cc
    assert max(np.unique(cell[tile == t]).size for t in np.unique(tile)) > 1024
    qs = []
    for i, c in enumerate(cases):
        gpu_ctx.upload_grid(8200 + i, c["grid"])
        qs.append(dict(map_id=8200 + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                       rel_pose=c["rel_pose"], init_pose=c["init_pose"]))
    for Lr, thr in ((4, (0.0, 0.0)), (3, (0.05, 0.1))):
        outs = gpu_ctx.correlative_match_batch(qs, 0.6, 0.4, math.radians(6), Lr, thr[0], thr[1])
        for c, o in zip(cases, outs):
            lit = oracle.csm(c, 0.6, 0.4, math.radians(6), Lr, thr[0], thr[1])
            raw = o["raw"]
            assert o["pose_found"] == lit["found"], (Lr, raw, lit)
            assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"]), (Lr, raw, lit)
            assert raw["score"] == lit["scoreMax"]
    for i in range(len(cases)):
        gpu_ctx.release_grid(8200 + i)


def test_largest_scan_and_one_beam_more(gpu_ctx, oracle):
    """10240 beams (csm_hip.h's limit: the binning kernel's tables fill the LDS) match
    the literal sweep; 10241 are refused with CSM_EINVAL."""
    case = _scattered_case(420, 10240)
    _check_csm(gpu_ctx, oracle, case, 0.6, 0.4, math.radians(6), 4)
    over = _scattered_case(421, 10241)
    gpu_ctx.upload_grid(8300, over["grid"])
    with pytest.raises(api.CsmError) as err:
        gpu_ctx.correlative_match(8300, over["geom"], over["angles"], over["ranges"], over["rel_pose"],
                                  over["init_pose"], 0.6, 0.4, math.radians(6), 4)
    assert err.value.code == L.CSM_EINVAL and "beams" in str(err.value)
    gpu_ctx.release_grid(8300)
