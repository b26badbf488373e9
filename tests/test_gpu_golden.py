"""GPU parity against the committed fixtures (tests/golden/csm_cases.json),
through the C ABI, plus size-independent properties at BASELINE config 2."""
import json
import math
import os
import struct

import numpy as np
import pytest

from csm_hip import api, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unhex(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


def _cases():
    with open(os.path.join(GOLD, "csm_cases.json")) as f:
        return json.load(f)


def _fix_kw(kw):
    kw = dict(kw)
    for k in ("init_error", "truth", "rel_pose"):
        if k in kw:
            kw[k] = tuple(kw[k])
    return kw


@pytest.mark.parametrize("rec", _cases()["csm"], ids=lambda r: r["name"])
def test_csm_fixture(gpu_ctx, rec):
    case = synth.csm_case(**_fix_kw(rec["synth"]))
    rx, ry, rt, L, st, kt = rec["params"]
    m = api.ScanMatcherCorrelativeHIP("gold", int(L), rx, ry, rt, ctx=gpu_ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"], score_threshold=st,
                          known_rate_threshold=kt)
    e = rec["expect"]
    assert out["pose_found"] == e["found"]
    assert [out["raw"]["best_x"], out["raw"]["best_y"], out["raw"]["best_theta"]] == e["best"]
    assert [out["win_x"], out["win_y"], out["win_theta"]] == e["win"]
    assert out["raw"]["score"] == unhex(e["score"])
    assert out["estimated_pose"] == [unhex(v) for v in e["estimated_pose"]]


@pytest.mark.parametrize("rec", _cases()["bnb"], ids=lambda r: r["name"])
def test_bnb_fixture(gpu_ctx, rec):
    case = synth.csm_case(**_fix_kw(rec["synth"]))
    rx, ry, rt, H, st, kt = rec["params"]
    gpu_ctx.upload_grid(900, case["grid"])
    q = dict(map_id=900, geom=case["geom"], angles=case["angles"], ranges=case["ranges"],
             rel_pose=case["rel_pose"], init_pose=case["init_pose"])
    out = gpu_ctx.bnb_match_batch([q], rx, ry, rt, int(H), st, kt)[0]
    e = rec["expect"]
    assert out["pose_found"] == e["found"]
    assert [out["raw"]["best_x"], out["raw"]["best_y"], out["raw"]["best_theta"]] == e["best"]
    assert out["raw"]["score"] == unhex(e["score"])
    assert out["estimated_pose"] == [unhex(v) for v in e["estimated_pose"]]
    gpu_ctx.release_grid(900)


def test_config2_full_size_properties(gpu_ctx, oracle):
    """BASELINE configs[1] at full size (1080 beams, +-2 m / +-30 deg, L = 4):
    the literal CPU sweep still finishes in a second thanks to pruning, so the
    winner is checked directly; plus properties that need no oracle:
    translation equivariance (shifting the initial pose by whole cells shifts
    the best offsets back by the same cells) and idempotence."""
    case = synth.csm_case(5, n_beams=1080, fov=1.5 * math.pi, init_error=(0.31, -0.27, 0.08))
    rx, ry, rt, L = 4.0, 4.0, math.radians(60), 4
    m = api.ScanMatcherCorrelativeHIP("cfg2", L, rx, ry, rt, ctx=gpu_ctx)
    args = (case["grid"], case["geom"], case["angles"], case["ranges"], case["rel_pose"])
    a = m.optimize_pose(*args, case["init_pose"], map_id=77)
    lit = oracle.csm(case, rx, ry, rt, L)
    assert (a["raw"]["best_x"], a["raw"]["best_y"], a["raw"]["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"])
    assert a["raw"]["score"] == lit["scoreMax"]
    assert a["candidates"] == (2 * a["win_theta"] + 1) * 84 * 84
    b = m.optimize_pose(None, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                        case["init_pose"], map_id=77)
    assert a["raw"] == b["raw"]
    # the sums are over integer cell offsets, so the winner's raw sums are a
    # function of the absolute best pose only
    assert a["raw"]["key"] == 32268 * a["raw"]["known"] + 499 * a["raw"]["sum_values"]
    gpu_ctx.release_grid(77)


def _map_cases():
    with open(os.path.join(GOLD, "map_cases.json")) as f:
        return json.load(f)


def _fnv64(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a, dtype="<u2").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


_BUILDER_NAMES = {"usable_max": "usable_range_max", "usable_min": "usable_range_min", "prob_hit": "prob_hit",
                  "prob_miss": "prob_miss", "subpixel": "subpixel_scale"}


@pytest.mark.parametrize("rec", _map_cases(), ids=lambda r: r["name"])
def test_map_fixture(gpu_ctx, rec):
    """Map builds against the committed fixture (no oracle run on this box)."""
    kw = dict(rec["synth"])
    if "rel_pose" in kw:
        kw["rel_pose"] = tuple(kw["rel_pose"])
    case = synth.map_case(**kw)
    bkw = {_BUILDER_NAMES[k]: v for k, v in rec["builder"].items()}
    shape, info = gpu_ctx.construct_map_from_scans(900, case["shape"], case["map_pose"], case["nodes"], **bkw)
    want = rec["batch"]
    assert (shape["rows"], shape["cols"]) == (want["rows"], want["cols"])
    assert [shape["off_x"], shape["off_y"]] == [unhex(v) for v in want["off"]]
    assert _fnv64(gpu_ctx.download_level(900, 0)) == want["hash"]
    assert (info["rays"], info["cell_updates"], info["saturated_reads"]) == \
        (want["rays"], want["updates"], want["saturated"])
    shape = case["shape"]
    gpu_ctx.upload_grid(901, np.zeros((shape["rows"], shape["cols"]), np.uint16))
    for nd in case["nodes"]:
        shape, _ = gpu_ctx.update_map_with_scan(901, shape, case["map_pose"], nd, **bkw)
    want = rec["incremental"]
    assert (shape["rows"], shape["cols"]) == (want["rows"], want["cols"])
    assert [shape["off_x"], shape["off_y"]] == [unhex(v) for v in want["off"]]
    assert _fnv64(gpu_ctx.download_level(901, 0)) == want["hash"]
    gpu_ctx.release_grid(900)
    gpu_ctx.release_grid(901)
