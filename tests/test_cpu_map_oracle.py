"""CPU checks of the map-building restatement (oracle/map_oracle.cpp): the
step-by-step ray walk against an independent closed form, the binary-Bayes cell
update, and the resize geometry."""
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


def test_bayes_primitives_bit_exact_vs_reference(oracle):
    """The conversions the cell update is made of, against outputs of the
    reference's own inline functions (grid_map_new/grid_values.hpp:11-58, built
    into oracle/_ref; tests/golden/ref_geometry.json)."""
    with open(os.path.join(GOLD, "ref_geometry.json")) as f:
        gold = json.load(f)
    lib = oracle.lib()
    for v, h in gold["value_to_odds"]:
        assert lib.orc_bb_value_to_odds(v) == unhex(h)
    for h, v in gold["probability_to_value"]:
        assert lib.orc_bb_probability_to_value(unhex(h)) == v
    for h, o in gold["probability_to_odds"]:
        assert lib.orc_bb_probability_to_odds(unhex(h)) == unhex(o)
    for h, p in gold["odds_to_probability"]:
        want = min(max(unhex(p), 1e-3), 1.0 - 1e-3)      # the wrapper's clamp, grid_binary_bayes.cpp:380
        assert lib.orc_bb_odds_to_probability(unhex(h)) == want


def test_bayes_primitives_against_live_reference_library(oracle):
    ref = oracle.ref()
    if ref is None or not hasattr(ref, "ref_value_to_odds"):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    lib = oracle.lib()
    for v in range(1, 65535):
        assert lib.orc_bb_value_to_odds(v) == ref.ref_value_to_odds(v)
    for p in np.linspace(1e-3, 1.0 - 1e-3, 20001):
        assert lib.orc_bb_probability_to_value(float(p)) == ref.ref_probability_to_value(float(p))


def closed_form_cells(sx, sy, ex, ey, s):
    """Per-column closed form of the sub-pixel ray walk: with N the ray's height
    in units of 1/(2 s dx) cell, column j holds the rows between its entry and
    exit heights; a ray through an exact cell corner steps diagonally."""
    if sx > ex:
        sx, sy, ex, ey = ex, ey, sx, sy
    X0, Y0, X1, Y1 = sx // s, sy // s, ex // s, ey // s
    if X0 == X1:
        return {(X0, y) for y in range(min(Y0, Y1), max(Y0, Y1) + 1)}
    dx, dy = ex - sx, ey - sy
    den = 2 * s * dx
    n = Y0 * den + (2 * (sy % s) + 1) * dx
    first = 2 * s - (2 * (sx % s) + 1)
    last = 2 * (ex % s) + 1
    m = X1 - X0
    cells = set()
    enter = Y0
    for j in range(m + 1):
        n_out = n + dy * (first + 2 * s * j) if j < m else n + dy * (first + 2 * s * (m - 1) + last)
        if dy > 0:
            top = -(-n_out // den) - 1
            cells.update((X0 + j, r) for r in range(enter, top + 1))
            enter = n_out // den
        else:
            bot = n_out // den
            cells.update((X0 + j, r) for r in range(bot, enter + 1))
            enter = -(-n_out // den) - 1
    return cells


def test_ray_walk_matches_closed_form(oracle):
    rng = np.random.RandomState(0)
    for scale in (1, 2, 3, 4, 100):
        for _ in range(1500):
            hi = 40 * scale
            sx, sy, ex, ey = (int(v) for v in rng.randint(0, hi, 4))
            walk = oracle.ray_cells(sx, sy, ex, ey, scale)
            assert len(set(walk)) == len(walk)                       # no cell twice
            assert set(walk) == closed_form_cells(sx, sy, ex, ey, scale), (sx, sy, ex, ey, scale)
            assert (sx // scale, sy // scale) in walk and (ex // scale, ey // scale) in walk
            for a, b in zip(walk, walk[1:]):                         # 8-connected, one step at a time
                assert max(abs(a[0] - b[0]), abs(a[1] - b[1])) == 1


def test_ray_walk_corner_cases(oracle):
    # through exact cell corners (scale 1: sub-pixel centres at .5): diagonal steps
    assert oracle.ray_cells(0, 0, 3, 3, 1) == [(0, 0), (1, 1), (2, 2), (3, 3)]
    assert oracle.ray_cells(3, 3, 0, 0, 1) == [(0, 0), (1, 1), (2, 2), (3, 3)]      # swapped ends
    assert oracle.ray_cells(0, 3, 3, 0, 1) == [(0, 3), (1, 2), (2, 1), (3, 0)]
    assert oracle.ray_cells(2, 5, 2, 1, 1) == [(2, 1), (2, 2), (2, 3), (2, 4), (2, 5)]   # vertical
    assert oracle.ray_cells(1, 4, 6, 4, 1) == [(x, 4) for x in range(1, 7)]            # horizontal
    assert oracle.ray_cells(7, 7, 7, 7, 1) == [(7, 7)]
    # same full-pixel column at sub-pixel scale takes the vertical branch
    assert oracle.ray_cells(110, 120, 190, 480, 100) == [(1, y) for y in range(1, 5)]


def test_bayes_update_table(oracle):
    hit, miss = 0.62, 0.46
    v0 = oracle.bayes_update(0, hit)
    assert v0 == int(1 + (hit - 1e-3) * 65534.0 / (1.0 - 1e-3 - 1e-3))
    assert oracle.bayes_update(0, miss) < 32768 < v0
    # repeated hits climb to ValueMax = 65535 in about 13 steps and stay there.
    # NOTE: the reference's odds table has 65535 entries (grid_values.cpp:74-77),
    # so its next update of such a cell reads one entry past the end (undefined
    # behaviour); the restatement extends the table's formula to 65535 and counts
    # those reads (stats["oob_reads"]). Repeated misses fall to 1.
    v, seen = v0, 0
    for _ in range(200):
        nv = oracle.bayes_update(v, hit)
        assert nv >= v
        seen += nv != v
        v = nv
    assert v == oracle.bayes_update(v, hit) == 65535 and seen < 20
    assert oracle.bayes_update(65535, miss) < 65535
    v = oracle.bayes_update(0, miss)
    for _ in range(200):
        nv = oracle.bayes_update(v, miss)
        assert nv <= v
        v = nv
    assert v == 1
    # hit then miss differs from miss then hit somewhere: order matters
    diff = sum(oracle.bayes_update(oracle.bayes_update(u, hit), miss) !=
               oracle.bayes_update(oracle.bayes_update(u, miss), hit) for u in range(1, 65535, 97))
    assert diff > 0


def test_construct_map_geometry_and_content(oracle):
    case = synth.map_case(3, n_scans=4, n_beams=360)
    shape, grid, stats = oracle.construct_map(case["shape"], case["map_pose"], case["nodes"])
    assert stats["end_missing"] == 0
    assert stats["rays"] > 0.9 * 4 * 360 - 8
    assert shape["rows"] % 16 == 0 and shape["cols"] % 16 == 0
    # the offset moved by whole blocks of the ORIGINAL frame (offset 0)
    assert abs(shape["off_x"] / (16 * 0.05) - round(shape["off_x"] / (16 * 0.05))) < 1e-9
    assert grid.shape == (shape["rows"], shape["cols"])
    known = grid != 0
    assert 0.05 < known.mean() < 0.9
    assert (grid[known] > 40000).sum() > 100 and (grid[known] < 30000).sum() > 1000
    # every hit point lies inside the map with the one-cell margin of Resize
    ys, xs = np.nonzero(known)
    assert xs.min() >= 1 and ys.min() >= 1 and xs.max() <= shape["cols"] - 2 and ys.max() <= shape["rows"] - 2
    # IndexToBlock sends a negative index one block further than floor would
    # (grid_map.cpp:809-812), so the first resize out of the original frame
    # keeps a spare block row / column; from a frame where every index is
    # non-negative the same scans give the tight box, and that one is stable
    shape2, grid2, _ = oracle.construct_map(shape, case["map_pose"], case["nodes"])
    assert shape2["rows"] == shape["rows"] - 16 and shape2["cols"] == shape["cols"] - 16
    shape3, grid3, _ = oracle.construct_map(shape2, case["map_pose"], case["nodes"])
    assert shape3 == shape2 and np.array_equal(grid2, grid3)
    # same cells, shifted by the spare block
    dy = int(round((shape2["off_y"] - shape["off_y"]) / 0.05))
    dx = int(round((shape2["off_x"] - shape["off_x"]) / 0.05))
    assert np.array_equal(grid[dy:dy + shape2["rows"], dx:dx + shape2["cols"]], grid2)


def test_construct_map_respects_usable_range(oracle):
    case = synth.map_case(5, n_scans=2, n_beams=180)
    _, _, full = oracle.construct_map(case["shape"], case["map_pose"], case["nodes"])
    _, _, short = oracle.construct_map(case["shape"], case["map_pose"], case["nodes"], usable_max=2.0)
    assert 0 < short["rays"] < full["rays"]


def _map_cases():
    with open(os.path.join(GOLD, "map_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("rec", _map_cases(), ids=lambda r: r["name"])
def test_map_golden(oracle, rec):
    """Regression pin of the restatement (tests/golden/map_cases.json)."""
    kw = dict(rec["synth"])
    if "rel_pose" in kw:
        kw["rel_pose"] = tuple(kw["rel_pose"])
    case = synth.map_case(**kw)
    shape, grid, stats = oracle.construct_map(case["shape"], case["map_pose"], case["nodes"], **rec["builder"])
    h = 1469598103934665603
    for b in grid.astype("<u2").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    want = rec["batch"]
    assert "%016x" % h == want["hash"]
    assert (shape["rows"], shape["cols"], stats["rays"], stats["updates"], stats["oob_reads"]) == \
        (want["rows"], want["cols"], want["rays"], want["updates"], want["saturated"])
    assert [shape["off_x"], shape["off_y"]] == [unhex(v) for v in want["off"]]
