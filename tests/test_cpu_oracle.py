"""CPU suite (-m "not gpu"): pins of the CPU oracle.

1. against the committed outputs of the reference's OWN header-only code
   (tests/golden/ref_geometry.json, generated from oracle/_ref) and, when the
   reference-built library is present, against it live;
2. against the committed regression fixtures (tests/golden/csm_cases.json);
3. internal consistency: literal sweep == closed form where no edge band is
   touched, literal sliding max == clamped-window formula, integer key order ==
   f64 order."""
import json
import math
import os
import struct

import numpy as np
import pytest

from csm_hip import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unhex(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


@pytest.fixture(scope="module")
def refgeo():
    with open(os.path.join(GOLD, "ref_geometry.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["compound", "inverse_compound", "move_backward"])
def test_pose_algebra_bit_exact_vs_reference(oracle, refgeo, name):
    fn = getattr(oracle, name)
    for rec in refgeo[name]:
        a = [unhex(v) for v in rec["a"]]
        b = [unhex(v) for v in rec["b"]]
        want = [unhex(v) for v in rec["out"]]
        assert list(fn(a, b)) == want


def test_hit_points_bit_exact_vs_reference(oracle, refgeo):
    import ctypes as C
    lib = oracle.lib()
    for rec in refgeo["hit_points"]:
        pose = np.array([unhex(v) for v in rec["pose"]])
        ang = [unhex(v) for v in rec["angles"]]
        rg = [unhex(v) for v in rec["ranges"]]
        want = [unhex(v) for v in rec["xy"]]
        for i in range(len(ang)):
            out = np.zeros(2)
            lib.orc_hit_point(pose.ctypes.data_as(C.c_void_p), C.c_double(rg[i]), C.c_double(ang[i]),
                              out.ctypes.data_as(C.c_void_p))
            assert list(out) == want[2 * i:2 * i + 2]


def test_probability_lut_bit_exact_vs_reference(oracle, refgeo):
    lut = oracle.lut()
    assert lut[0] == 0.0
    for v, h in refgeo["value_to_probability"]:
        assert lut[v] == unhex(h)
    assert np.all(np.diff(lut[1:]) > 0)


def test_live_reference_library_when_present(oracle):
    ref = oracle.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    lut = oracle.lut()
    for v in range(1, 65535):
        assert lut[v] == ref.ref_value_to_probability(v)


def _load_cases():
    with open(os.path.join(GOLD, "csm_cases.json")) as f:
        return json.load(f)


def _fix_kw(kw):
    kw = dict(kw)
    for k in ("init_error", "truth", "rel_pose"):
        if k in kw:
            kw[k] = tuple(kw[k])
    return kw


@pytest.mark.parametrize("rec", _load_cases()["csm"], ids=lambda r: r["name"])
def test_csm_golden(oracle, rec):
    case = synth.csm_case(**_fix_kw(rec["synth"]))
    rx, ry, rt, L, st, kt = rec["params"]
    assert int(case["grid"].astype(np.uint64).sum()) == rec["grid_sum"]
    coarse = oracle.boxmax(case["grid"], int(L))
    assert int(coarse.astype(np.uint64).sum()) == rec["coarse_sum"]
    res = oracle.csm(case, rx, ry, rt, int(L), st, kt, coarse=coarse)
    e = rec["expect"]
    assert res["found"] == e["found"]
    assert [res["bestX"], res["bestY"], res["bestT"]] == e["best"]
    assert [res["winX"], res["winY"], res["winT"]] == e["win"]
    assert res["scoreMax"] == unhex(e["score"])
    assert res["estimatedPose"] == [unhex(v) for v in e["estimated_pose"]]
    assert (res["ignoredNodes"], res["processedNodes"]) == (e["ignored"], e["processed"])


@pytest.mark.parametrize("rec", _load_cases()["bnb"], ids=lambda r: r["name"])
def test_bnb_golden(oracle, rec):
    case = synth.csm_case(**_fix_kw(rec["synth"]))
    rx, ry, rt, H, st, kt = rec["params"]
    res = oracle.bnb(case, rx, ry, rt, int(H), st, kt)
    e = rec["expect"]
    assert res["found"] == e["found"]
    assert [res["bestX"], res["bestY"], res["bestT"]] == e["best"]
    assert res["scoreMax"] == unhex(e["score"])
    assert res["estimatedPose"] == [unhex(v) for v in e["estimated_pose"]]


@pytest.mark.parametrize("seed,L", [(0, 4), (1, 1), (2, 3), (3, 5), (4, 7), (5, 4)])
def test_literal_sweep_equals_closed_form_without_edge_band(oracle, seed, L):
    case = synth.csm_case(seed)
    a = oracle.csm(case, 1.0, 1.0, math.radians(10), L)
    b = oracle.csm_closed_form(case, 1.0, 1.0, math.radians(10), L)
    assert b["touchesBand"] == 0
    assert (a["bestX"], a["bestY"], a["bestT"], a["scoreMax"]) == (b["bestX"], b["bestY"], b["bestT"], b["scoreMax"])
    # the best pose may lie beyond +win (extended domain, SURVEY 7 hard parts)
    nx = -(-(2 * a["winX"] + 1) // L) * L
    assert -a["winX"] <= a["bestX"] < -a["winX"] + nx


@pytest.mark.parametrize("win", [1, 2, 3, 4, 5, 7, 8, 16, 33])
def test_sliding_max_equals_clamped_window_formula(oracle, win):
    rng = np.random.RandomState(win)
    g = rng.randint(0, 65535, size=(37, 53)).astype(np.uint16)
    g[rng.rand(37, 53) < 0.3] = 0
    got = oracle.boxmax(g, win)
    want = np.zeros_like(g)
    for r in range(g.shape[0]):
        r0 = min(r, g.shape[0] - win)
        for c in range(g.shape[1]):
            c0 = min(c, g.shape[1] - win)
            want[r, c] = g[r0:r0 + win, c0:c0 + win].max()
    assert np.array_equal(got, want)


def test_integer_key_orders_like_the_f64_sum(oracle):
    """key = 32268*K + 499*S reproduces the order of the beam-order f64 sum
    whenever keys differ."""
    lut = oracle.lut()
    rng = np.random.RandomState(7)
    n = 1080
    base = rng.randint(0, 65535, size=n)
    base[rng.rand(n) < 0.2] = 0
    keys, sums = [], []
    for _ in range(300):
        v = base.copy()
        idx = rng.randint(0, n, size=3)
        v[idx] = np.clip(v[idx] + rng.randint(-2, 3, size=3), 0, 65534)
        s = 0.0
        for p in lut[v]:
            if p != 0.0:
                s += p
        keys.append(32268 * int((v != 0).sum()) + 499 * int(v.sum()))
        sums.append(s / n)
    order = np.argsort(keys, kind="stable")
    for a, b in zip(order[:-1], order[1:]):
        if keys[a] < keys[b]:
            assert sums[a] < sums[b]
