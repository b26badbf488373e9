import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))
import __graft_entry__ as ge
ge.build()
from csm_hip import api, synth

def run(case, rx, ry, rt, L, pairs, merge=0):
    os.environ["CSM_FINE_PAIRS"] = "1" if pairs else "0"
    ctx = api.Context(0)
    sx, sy, st = api.host_search_step(case["geom"][0], case["ranges"])
    wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
    sensor = api.host_compound(case["init_pose"], case["rel_pose"])
    col, row = api.host_project(case["geom"], sensor, st, wt, case["angles"], case["ranges"])
    ctx.upload_grid(1, case["grid"])
    ctx.build_pyramid(1, [1, L])
    w = ctx.make_window(2 * wt + 1, len(case["angles"]), wx, wy, L, 1, api.host_min_known(len(case["angles"]), 0.0), 0.0, merge)
    res, S, K, CK = ctx.score_window(1, w, col, row, dump=True)
    ctx.close()
    return res, S, K, CK

for (seed, nb, rx, rt, L) in [(0, 360, 1.0, 10, 4), (3, 1080, 4.0, 60, 4)]:
    case = synth.csm_case(seed, n_beams=nb, fov=(2 * math.pi if nb == 360 else 1.5 * math.pi))
    for merge in (0, 1):
        r0, S0, K0, C0 = run(case, rx, rx, math.radians(rt), L, False, merge)
        r1, S1, K1, C1 = run(case, rx, rx, math.radians(rt), L, True, merge)
        d = np.argwhere(S0 != S1)
        print("seed", seed, "merge", merge, "shape", S0.shape, "S diff", len(d), "K diff", int((K0 != K1).sum()),
              "coarse K diff", int((C0 != C1).sum()), "res", r0["best_x"], r0["best_y"], r0["best_theta"], "|",
              r1["best_x"], r1["best_y"], r1["best_theta"])
        if len(d):
            print(" t:", np.unique(d[:, 0])[:20], "\n x:", np.unique(d[:, 1])[:90], "\n y:", np.unique(d[:, 2])[:90])
            for t, x, y in d[:12]:
                print("  ", t, x, y, int(S0[t, x, y]), int(S1[t, x, y]), int(K0[t, x, y]), int(K1[t, x, y]),
                      "dS", int(S1[t, x, y]) - int(S0[t, x, y]))
