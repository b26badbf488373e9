"""The coarse-first search of a single window (csm_phase_kernels.hip): every coarse node scored
on the phase-major copy of the box-max level, the fine level only on the candidate blocks whose
coarse bound reaches the best fine score under the best coarse node. Forced on for windows of
every size (CSM_TUNE_FORCE_TWO_PHASE) and checked against the oracle's LITERAL sweep with its
running-maximum pruning (scan_matcher_correlative.cpp:161-197, 339-368 restated): ties, the
negative edge band, thresholds, odd and even coarse window sizes, coarse windows 2..8; then a
window large enough to take the path by itself."""
import math

import numpy as np
import pytest

from csm_hip import _lib as L, api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def forced():
    c = api.Context(0, tuning_off=L.TUNE_FORCE_TWO_PHASE)
    yield c
    c.close()


def _check(ctx, oracle, case, rx, ry, rt, Lr, score_thr=0.0, known_thr=0.0):
    m = api.ScanMatcherCorrelativeHIP("two_phase", Lr, rx, ry, rt, ctx=ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"], score_threshold=score_thr,
                          known_rate_threshold=known_thr)
    lit = oracle.csm(case, rx, ry, rt, Lr, score_thr, known_thr)
    raw = out["raw"]
    assert out["pose_found"] == lit["found"], (raw, lit)
    assert (raw["best_x"], raw["best_y"], raw["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"]), (raw, lit)
    assert raw["score"] == lit["scoreMax"]
    assert out["estimated_pose"] == lit["estimatedPose"]
    return out, ctx.last_search_info()


@pytest.mark.parametrize("seed,Lr,rx,ry", [(0, 4, 1.0, 1.0), (1, 4, 1.1, 0.9), (2, 2, 1.0, 1.0), (3, 5, 1.0, 1.3),
                                           (4, 3, 0.7, 1.0), (5, 8, 1.0, 1.0), (6, 6, 1.5, 1.2), (7, 4, 2.0, 2.0)])
def test_forced_two_phase_equals_literal_sweep(forced, oracle, seed, Lr, rx, ry):
    case = synth.csm_case(seed, n_beams=360 + 90 * (seed % 4))
    out, info = _check(forced, oracle, case, rx, ry, math.radians(10), Lr)
    assert info["two_phase"] == 1 and info["coarse_nodes_scored"] > 0
    assert info["blocks_scored"] >= 1
    assert info["fine_candidates_scored"] <= info["nominal_candidates"] * 1.5      # whole blocks


@pytest.mark.parametrize("seed,levels", [(40, 2), (41, 3), (43, 8)])
def test_two_phase_integer_key_ties(forced, oracle, seed, levels):
    case = synth.csm_case(seed, levels=levels, interior_unknown=0.0)
    _check(forced, oracle, case, 1.0, 1.0, math.radians(10), 4)


@pytest.mark.parametrize("seed,Lr", [(50, 4), (51, 4), (52, 5), (53, 8), (54, 3), (56, 2)])
def test_two_phase_negative_edge_band(forced, oracle, seed, Lr):
    """Beams at negative level indices: the phase-major copy reads 0 there exactly like the
    reference's lookup, the bound is void, every block is kept and the literal path decides."""
    case = synth.csm_case(seed, rows=256, cols=288, origin="low_edge", half_x=5.2, half_y=4.4,
                          init_error=(0.23, 0.19, 0.03))
    _check(forced, oracle, case, 1.0, 1.0, math.radians(10), Lr)


def test_two_phase_thresholds_and_not_found(forced, oracle):
    case = synth.csm_case(60)
    out, _ = _check(forced, oracle, case, 1.0, 1.0, math.radians(10), 4, 0.95, 0.0)
    assert out["pose_found"] == 0
    _check(forced, oracle, case, 1.0, 1.0, math.radians(10), 4, 0.2, 0.9)
    _check(forced, oracle, case, 1.0, 1.0, math.radians(10), 4, 0.1, 0.99)
    _check(forced, oracle, case, 0.5, 1.5, math.radians(4), 5, 0.3, 0.5)
    empty = dict(case, grid=np.zeros_like(case["grid"]))
    out, _ = _check(forced, oracle, empty, 1.0, 1.0, math.radians(10), 4)
    assert out["pose_found"] == 0


def test_large_window_takes_the_path_by_itself_and_prunes(gpu_ctx, oracle):
    """+-6 m / +-180 deg at 5 cm / 0.5 deg on a 640 x 640 map: 4.3e7 fine candidates. The default
    context searches it coarse-first; most candidate blocks are never scored."""
    case = synth.csm_case(71, rows=640, cols=640, n_beams=720, fov=1.5 * math.pi, init_error=(2.1, -1.7, 1.1),
                          n_boxes=8)
    out, info = _check(gpu_ctx, oracle, case, 12.0, 12.0, 2 * math.pi, 4)
    assert out["pose_found"] == 1
    assert info["two_phase"] == 1
    assert info["blocks_skipped"] > info["blocks_scored"] > 0, info
    assert info["fine_candidates_scored"] < info["nominal_candidates"] // 2
    # and the exhaustive search of the same window gives the same record
    plain = api.Context(0, tuning_off=L.TUNE_NO_TWO_PHASE)
    out2, info2 = _check(plain, oracle, case, 12.0, 12.0, 2 * math.pi, 4)
    assert info2["two_phase"] == 0 and out2["raw"] == out["raw"]
    plain.close()
