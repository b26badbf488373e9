#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run in the build container only (needs /root/reference for ref_geometry.json):
    python tests/golden/make_golden.py

ref_geometry.json  OUTPUTS OF THE REFERENCE'S OWN CODE: oracle/_ref/libref_geom.so
                   is compiled from the reference's header-only pose.hpp,
                   sensor/sensor_data.hpp and grid_map_new/grid_values.hpp
                   (oracle/Makefile); doubles are stored as hex bit patterns.
csm_cases.json     Outputs of the CPU oracle (oracle/csm_oracle.cpp) on seeded
                   synthetic cases: regression pins for the restatement and
                   expected answers for the HIP path on the GPU box. They are
                   NOT reference outputs (the reference matcher cannot be built
                   here: it needs Eigen3 and Boost).
map_cases.json     Outputs of the CPU oracle (oracle/map_oracle.cpp) for map builds:
                   geometry, FNV-1a hash of the cells, update counters. Same status
                   as csm_cases.json.
Fixtures are data only: inputs are regenerated from seeds by csm_hip/synth.py.
"""
import json
import math
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))

from oracle import oracle as O  # noqa: E402
from csm_hip import synth  # noqa: E402
import ctypes as C  # noqa: E402


def hexd(x):
    return struct.pack(">d", float(x)).hex()


def ref_geometry():
    ref = O.ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_geom.so missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.RandomState(12345)
    out = {"compound": [], "inverse_compound": [], "move_backward": [], "hit_points": [],
           "value_to_probability": []}

    def call3(fn, a, b):
        o = np.zeros(3)
        fn(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
        return o
    for _ in range(64):
        a = rng.uniform(-20, 20, 3) * (1, 1, 0.2)
        b = rng.uniform(-3, 3, 3)
        for name, fn in (("compound", ref.ref_compound), ("inverse_compound", ref.ref_inverse_compound),
                         ("move_backward", ref.ref_move_backward)):
            o = call3(fn, a, b)
            out[name].append({"a": [hexd(v) for v in a], "b": [hexd(v) for v in b],
                              "out": [hexd(v) for v in o]})
    for _ in range(8):
        pose = rng.uniform(-10, 10, 3) * (1, 1, 0.3)
        n = 97
        ang = rng.uniform(-math.pi, math.pi, n)
        rg = rng.uniform(0.1, 30, n)
        xy = np.zeros(2 * n)
        ref.ref_hit_points(pose.ctypes.data_as(C.c_void_p), ang.ctypes.data_as(C.c_void_p),
                           rg.ctypes.data_as(C.c_void_p), n, xy.ctypes.data_as(C.c_void_p))
        out["hit_points"].append({"pose": [hexd(v) for v in pose], "angles": [hexd(v) for v in ang],
                                  "ranges": [hexd(v) for v in rg], "xy": [hexd(v) for v in xy]})
    vals = list(range(1, 65536, 257)) + [1, 2, 3, 32767, 32768, 65533, 65534]
    for v in sorted(set(vals)):
        out["value_to_probability"].append([v, hexd(ref.ref_value_to_probability(v))])
    # the primitives of the binary-Bayes cell update (grid_values.hpp:11-58)
    out["value_to_odds"] = [[v, hexd(ref.ref_value_to_odds(v))] for v in sorted(set(vals))]
    probs = [1e-3, 1.0 - 1e-3, 0.46, 0.5, 0.62] + list(rng.uniform(1e-3, 1.0 - 1e-3, 200))
    out["probability_to_value"] = [[hexd(p), int(ref.ref_probability_to_value(p))] for p in probs]
    out["probability_to_odds"] = [[hexd(p), hexd(ref.ref_probability_to_odds(p))] for p in probs]
    odds = [0.0, 1.0, 0.46 / 0.54, 0.62 / 0.38] + list(np.exp(rng.uniform(-8, 8, 200)))
    out["odds_to_probability"] = [[hexd(o), hexd(ref.ref_odds_to_probability(o))] for o in odds]
    return out


CSM_CASES = [
    # name, synth kwargs, (range_x, range_y, range_theta, L), (score_thr, known_thr)
    ("cfg1_seed0", dict(seed=0), (1.0, 1.0, math.radians(10), 4), (0.0, 0.0)),
    ("cfg1_seed1_L5", dict(seed=1), (1.0, 1.0, math.radians(10), 5), (0.0, 0.0)),
    ("cfg1_seed2_thr", dict(seed=2), (1.0, 1.0, math.radians(10), 4), (0.3, 0.6)),
    ("ties_levels3", dict(seed=41, levels=3, interior_unknown=0.0), (1.0, 1.0, math.radians(10), 4), (0.0, 0.0)),
    ("low_edge", dict(seed=50, rows=256, cols=288, origin="low_edge", half_x=5.2, half_y=4.4,
                      init_error=(0.23, 0.19, 0.03)), (1.0, 1.0, math.radians(10), 4), (0.0, 0.0)),
    ("cfg2_small", dict(seed=5, n_beams=1080, fov=1.5 * math.pi), (2.0, 2.0, math.radians(20), 4), (0.0, 0.0)),
]
BNB_CASES = [
    ("bnb_h2", dict(seed=20, init_error=(0.4, -0.3, 0.06)), (2.5, 2.5, 0.5, 2), (0.3, 0.5)),
    ("bnb_h3", dict(seed=21, init_error=(0.4, -0.3, 0.06)), (2.5, 2.5, 0.5, 3), (0.3, 0.5)),
    ("bnb_h6_default_thr", dict(seed=22, init_error=(0.4, -0.3, 0.06)), (2.5, 2.5, 0.5, 6), (0.55, 0.6)),
    ("bnb_aligned", dict(seed=90, origin="aligned", truth=(0.0, 0.0, 0.0), init_error=(0.25, -0.15, 0.0)),
     (2.5, 2.5, 0.5, 2), (0.3, 0.5)),
]


def pack(res):
    return {"found": res["found"], "best": [res["bestX"], res["bestY"], res["bestT"]],
            "win": [res["winX"], res["winY"], res["winT"]],
            "step_theta": hexd(res["stepT"]), "score": hexd(res["scoreMax"]),
            "estimated_pose": [hexd(v) for v in res["estimatedPose"]],
            "ignored": res["ignoredNodes"], "processed": res["processedNodes"]}


def csm_cases():
    out = {"csm": [], "bnb": []}
    for name, kw, (rx, ry, rt, L), (st, kt) in CSM_CASES:
        case = synth.csm_case(**kw)
        res = O.csm(case, rx, ry, rt, L, st, kt)
        coarse = O.boxmax(case["grid"], L)
        out["csm"].append({"name": name, "synth": kw, "params": [rx, ry, rt, L, st, kt],
                           "grid_sum": int(case["grid"].astype(np.uint64).sum()),
                           "coarse_sum": int(coarse.astype(np.uint64).sum()),
                           "coarse_xor": int(np.bitwise_xor.reduce(coarse.ravel().astype(np.uint64) *
                                                                   (np.arange(coarse.size, dtype=np.uint64) % 65521 + 1))),
                           "expect": pack(res)})
    for name, kw, (rx, ry, rt, H), (st, kt) in BNB_CASES:
        case = synth.csm_case(**kw)
        res = O.bnb(case, rx, ry, rt, H, st, kt)
        out["bnb"].append({"name": name, "synth": kw, "params": [rx, ry, rt, H, st, kt],
                           "expect": pack(res)})
    return out


MAP_CASES = [
    # name, synth.map_case kwargs, builder kwargs (oracle names)
    ("latest_10x360", dict(seed=0, n_scans=10, n_beams=360), {}),
    ("latest_10x1080", dict(seed=2, n_scans=10, n_beams=1080), {}),
    ("offset_sensor_noise", dict(seed=3, n_scans=6, n_beams=500, rel_pose=[0.1, -0.05, 0.02], noise=0.01),
     dict(usable_max=4.0, prob_hit=0.7, prob_miss=0.4, subpixel=10)),
]


def fnv64(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a, dtype="<u2").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def map_cases():
    """Map builds (oracle/map_oracle.cpp): the batch build of all scans, then the
    same scans one by one into a fresh local map."""
    out = []
    for name, kw, bkw in MAP_CASES:
        skw = dict(kw)
        if "rel_pose" in skw:
            skw["rel_pose"] = tuple(skw["rel_pose"])
        case = synth.map_case(**skw)
        shape, grid, stats = O.construct_map(case["shape"], case["map_pose"], case["nodes"], **bkw)
        inc_shape = case["shape"]
        inc = np.zeros((inc_shape["rows"], inc_shape["cols"]), np.uint16)
        for nd in case["nodes"]:
            inc_shape, inc, _ = O.update_map(inc_shape, inc, case["map_pose"], nd, **bkw)
        out.append({"name": name, "synth": kw, "builder": bkw,
                    "batch": {"rows": shape["rows"], "cols": shape["cols"], "off": [hexd(shape["off_x"]), hexd(shape["off_y"])],
                              "hash": fnv64(grid), "rays": stats["rays"], "updates": stats["updates"],
                              "saturated": stats["oob_reads"]},
                    "incremental": {"rows": inc_shape["rows"], "cols": inc_shape["cols"],
                                    "off": [hexd(inc_shape["off_x"]), hexd(inc_shape["off_y"])], "hash": fnv64(inc)}})
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "map_cases.json"), "w") as f:
        json.dump(map_cases(), f, indent=1)
    with open(os.path.join(HERE, "ref_geometry.json"), "w") as f:
        json.dump(ref_geometry(), f)
    with open(os.path.join(HERE, "csm_cases.json"), "w") as f:
        json.dump(csm_cases(), f, indent=1)
    print("golden fixtures written")
