"""GPU parity of the brute-force matcher (ScanMatcherGridSearch) against the
literal CPU restatement: accumulated-double offsets, per-pose projection, own
known-rate test, first strict maximum in (dy, dx, dt) order."""
import pytest

from csm_hip import api, parallel, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,steps,thr", [
    (0, (0.05, 0.05, 0.005), (0.0, 0.0)),
    (1, (0.05, 0.05, 0.01), (0.3, 0.5)),
    (2, (0.03, 0.07, 0.013), (0.2, 0.4)),       # steps that are not multiples of the resolution
    (3, (0.1, 0.1, 0.02), (0.9, 0.5)),          # nothing passes the score threshold
    (4, (0.025, 0.025, 0.05), (0.1, 0.95)),     # known-rate threshold bites
])
def test_grid_search_matches_literal_loops(gpu_ctx, oracle, seed, steps, thr):
    case = synth.csm_case(seed, n_beams=240)
    rng3 = (0.6, 0.5, 0.2)
    gpu_ctx.upload_grid(60, case["grid"])
    out = gpu_ctx.grid_search_match(60, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                                    case["init_pose"], *rng3, *steps, thr[0], thr[1])
    want = oracle.grid_search(case, *rng3, *steps, thr[0], thr[1])
    assert out["candidates"] == want["evaluations"]
    assert out["pose_found"] == want["found"]
    assert [out["raw"]["best_x"], out["raw"]["best_y"], out["raw"]["best_theta"]] == want["bestIdx"]
    assert out["raw"]["score"] == want["scoreMax"]
    assert list(out["best_sensor_pose"]) == want["bestSensorPose"]
    assert list(out["estimated_pose"]) == want["estimatedPose"]
    gpu_ctx.release_grid(60)


def test_grid_search_default_loop_detector_window(gpu_ctx, oracle):
    """The default matcher settings of the grid-search loop detector: 2.5 m x
    2.5 m x 0.5 rad at 0.05 / 0.05 / 0.005 = 51 x 51 x 101 poses; 120 beams keep
    the CPU loops short."""
    case = synth.csm_case(7, n_beams=120)
    gpu_ctx.upload_grid(61, case["grid"])
    args = (2.5, 2.5, 0.5, 0.05, 0.05, 0.005, 0.3, 0.5)
    out = gpu_ctx.grid_search_match(61, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                                    case["init_pose"], *args)
    want = oracle.grid_search(case, *args)
    assert out["candidates"] == want["evaluations"]
    assert out["pose_found"] == want["found"]
    assert [out["raw"]["best_x"], out["raw"]["best_y"], out["raw"]["best_theta"]] == want["bestIdx"]
    assert out["raw"]["score"] == want["scoreMax"]
    assert list(out["estimated_pose"]) == want["estimatedPose"]
    gpu_ctx.release_grid(61)


def test_grid_search_wrappers(gpu_ctx, oracle):
    """ScanMatcherGridSearchHIP with a throw-away map and the loop detector
    wrapper over two queries (one off the map)."""
    case = synth.csm_case(9, n_beams=150, rel_pose=(0.03, 0.0, 0.01))
    m = api.ScanMatcherGridSearchHIP("gs", 0.5, 0.5, 0.1, 0.05, 0.05, 0.01, ctx=gpu_ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"])
    want = oracle.grid_search(case, 0.5, 0.5, 0.1, 0.05, 0.05, 0.01)
    assert list(out["estimated_pose"]) == want["estimatedPose"]
    assert out["raw"]["score"] == want["scoreMax"]

    det = parallel.LoopDetectorGridSearchHIP("ldgs", gpu_ctx, 1.0, 1.0, 0.2, 0.05, 0.05, 0.01, 0.3, 0.5)
    qs = []
    for init in (case["init_pose"], (40.0, -40.0, 0.0)):
        qs.append(dict(map_id=77, geom=case["geom"], angles=case["angles"], ranges=case["ranges"],
                       rel_pose=case["rel_pose"], init_pose=init))
    outs, found = det.detect(qs, grids={77: case["grid"]})
    assert found == [0]
    want = oracle.grid_search(case, 1.0, 1.0, 0.2, 0.05, 0.05, 0.01, 0.3, 0.5)
    assert list(outs[0]["estimated_pose"]) == want["estimatedPose"]
    assert outs[1]["pose_found"] == 0 and list(outs[1]["best_sensor_pose"]) == list(outs[1]["sensor_pose"])
    gpu_ctx.release_grid(77)
    with pytest.raises(ValueError):
        parallel.LoopDetectorGridSearchHIP("bad", gpu_ctx, 1, 1, 1, 0.1, 0.1, 0.1, 0.0, 0.5)
