"""The two-pass fine level of the batch entries: a packed-fp32 bound pass over every
candidate, then the exact integer kernel on the candidate blocks that can still hold the
winner (csm_joint_kernels.hip). These cases are built to defeat a careless bound:
landscapes where thousands of candidates tie exactly or lie within the fp32 rounding of
each other. The records must equal the oracle's literal sweep either way, and on a flat
landscape the exact kernel must not skip a single block."""
import math

import numpy as np
import pytest
import torch

from csm_hip import _lib as L, api, synth

pytestmark = pytest.mark.gpu


def _run(ctx, case, rx, ry, rt, Lr, n_copies=3):
    dev = torch.device("cuda", 0)
    sx, sy, st = api.host_search_step(case["geom"][0], case["ranges"])
    wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
    col, row = api.host_project(case["geom"], case["init_pose"], st, wt, case["angles"], case["ranges"])
    n = len(case["angles"])
    ctx.upload_grid(9, case["grid"])
    ctx.build_pyramid(9, [1, Lr])
    w = ctx.make_window(2 * wt + 1, n, wx, wy, Lr, 1, api.host_min_known(n, 0.0), 0.0)
    c_d, r_d = torch.from_numpy(col).to(dev), torch.from_numpy(row).to(dev)
    out = torch.zeros(n_copies * 48, dtype=torch.uint8, device=dev)
    prepared = ctx.prepare_windows([9] * n_copies, [w] * n_copies, [c_d.data_ptr()] * n_copies,
                                   [r_d.data_ptr()] * n_copies)
    ctx.bound_pass_stats()
    ctx.score_windows_dev(prepared, out.data_ptr())
    torch.cuda.synchronize(dev)
    stats = ctx.bound_pass_stats()
    recs = [L.Result.from_buffer_copy(out.cpu().numpy()[48 * k:48 * (k + 1)].tobytes()) for k in range(n_copies)]
    final = []
    for r in recs:
        if r.flags & (L.FLAG_EDGE_BAND | L.FLAG_KEY_TIE):      # finished by the exact single-window paths
            d = ctx.score_window(9, w, col, row)
            final.append((d["found"], d["best_x"], d["best_y"], d["best_theta"], d["score"]))
        else:
            final.append((r.found, r.best_x, r.best_y, r.best_theta, r.score))
    ctx.release_grid(9)
    return final, stats


@pytest.fixture(scope="module")
def ctx():
    dev = torch.device("cuda", 0)
    c = api.Context(0)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    yield c
    c.close()


def _want(oracle, case, rx, ry, rt, Lr):
    lit = oracle.csm(case, rx, ry, rt, Lr)
    return (lit["found"], lit["bestX"], lit["bestY"], lit["bestT"], lit["scoreMax"])


def test_flat_landscape_skips_nothing(ctx, oracle):
    """Every cell 30000: every candidate whose beams stay inside the map has the same key.
    All blocks hold a maximum; the first candidate in traversal order wins."""
    case = synth.csm_case(41, n_beams=720)
    case["grid"] = np.full_like(case["grid"], 30000)
    rx, ry, rt, Lr = 1.6, 1.6, math.radians(6), 4
    final, (scored, skipped) = _run(ctx, case, rx, ry, rt, Lr)
    assert all(f == _want(oracle, case, rx, ry, rt, Lr) for f in final)
    assert skipped == 0 and scored > 0


@pytest.mark.parametrize("seed,kind", [(42, "checker"), (43, "noise"), (44, "steps"), (45, "sparse")])
def test_near_tie_landscapes(ctx, oracle, seed, kind):
    """Grids on which many candidates differ by less than the fp32 rounding of the bound pass:
    a two-valued checkerboard, noise of +-2 around one value, value steps of 1, and a mostly
    unknown map with few known cells."""
    case = synth.csm_case(seed, n_beams=900)
    g = case["grid"]
    rng = np.random.RandomState(seed)
    rr, cc = np.indices(g.shape)
    if kind == "checker":
        g = np.where((rr + cc) & 1, 40000, 40001).astype(np.uint16)
    elif kind == "noise":
        g = (50000 + rng.randint(-2, 3, g.shape)).astype(np.uint16)
    elif kind == "steps":
        g = (20000 + (rr // 7 + cc // 5) % 3).astype(np.uint16)
    else:
        g = np.where(rng.rand(*g.shape) < 0.02, 65535, 0).astype(np.uint16)
    case["grid"] = g
    rx, ry, rt, Lr = 1.6, 1.6, math.radians(5), 4
    final, stats = _run(ctx, case, rx, ry, rt, Lr)
    assert all(f == _want(oracle, case, rx, ry, rt, Lr) for f in final), (kind, final[0], stats)


def test_bound_pass_off_is_identical(oracle):
    """The same batch with CSM_TUNE_NO_BOUND_PASS: the records do not change."""
    dev = torch.device("cuda", 0)
    case = synth.csm_case(46, n_beams=1080, fov=1.5 * math.pi)
    rx, ry, rt, Lr = 2.0, 2.0, math.radians(8), 4
    outs = []
    for off in (0, L.TUNE_NO_BOUND_PASS):
        c = api.Context(0, tuning_off=off)
        c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        final, stats = _run(c, case, rx, ry, rt, Lr)
        outs.append(final)
        assert (stats[1] > 0) == (off == 0)
        c.close()
    assert outs[0] == outs[1]
    assert outs[0][0] == _want(oracle, case, rx, ry, rt, Lr)


@pytest.mark.parametrize("n_beams,expect_bound_pass", [(1500, True), (2600, True), (6000, False)])
def test_many_beams_joint_and_fallback(ctx, oracle, n_beams, expect_bound_pass):
    """Above ~2,200 beams the joint hash table of a slice pair takes a CU's LDS alone; above ~4,200
    the batch goes back to per-slice lists and the exact kernel on every block. Same records."""
    case = synth.csm_case(47, n_beams=n_beams, max_range=4.0)
    rx, ry, rt, Lr = 1.2, 1.2, math.radians(6), 4
    final, (scored, skipped) = _run(ctx, case, rx, ry, rt, Lr, n_copies=2)
    assert all(f == _want(oracle, case, rx, ry, rt, Lr) for f in final)
    assert (skipped > 0) == expect_bound_pass


def test_thresholded_correlative_batch_uses_two_rounds(ctx, oracle):
    """LoopDetectorCorrelative's thresholds (score 0.55, known rate 0.6, launcher_settings_default.json:
    69-70): the coarse node's known count decides eligibility, the exact pass runs in two rounds behind
    the bound pass, and the records equal the literal sweep's."""
    qs, cases = [], []
    for i, seed in enumerate([91, 92, 93, 94, 95, 96]):
        case = synth.csm_case(seed, n_beams=1080, fov=1.5 * math.pi, init_error=(0.25, -0.2, 0.03))
        cases.append(case)
        qs.append(dict(map_id=1400 + i, geom=case["geom"], angles=case["angles"], ranges=case["ranges"],
                       rel_pose=case["rel_pose"], init_pose=case["init_pose"]))
        ctx.upload_grid(1400 + i, case["grid"])
    rx, ry, rt, Lr = 2.0, 2.0, math.radians(20), 4
    ctx.bound_pass_stats()
    outs = ctx.correlative_match_batch(qs, rx, ry, rt, Lr, 0.55, 0.6)
    scored, skipped = ctx.bound_pass_stats()
    assert skipped > scored > 0
    for c, o in zip(cases, outs):
        want = oracle.csm(c, rx, ry, rt, Lr, 0.55, 0.6)
        assert o["pose_found"] == want["found"]
        if want["found"]:
            assert (o["raw"]["best_x"], o["raw"]["best_y"], o["raw"]["best_theta"]) == \
                (want["bestX"], want["bestY"], want["bestT"])
            assert o["raw"]["score"] == want["scoreMax"]
        assert o["estimated_pose"] == want["estimatedPose"]
    for q in qs:
        ctx.release_grid(q["map_id"])
