"""Single-query launch chains replayed as HIP graphs (csm_correlative_match, from the third query
of a launch shape on): the records must equal those of the kernel-by-kernel path and the oracle,
for varying scans and poses of one shape, for alternating shapes, and across a map re-upload."""
import math

import numpy as np
import pytest

from csm_hip import _lib as L, api, synth

pytestmark = pytest.mark.gpu


def _match(ctx, case, sc, rx, ry, rt, Lr):
    return ctx.correlative_match(1, case["geom"], sc["angles"], sc["ranges"], case["rel_pose"], sc["init_pose"],
                                 rx, ry, rt, Lr, 0.0, 0.0)


def test_graph_replay_equals_plain_launches(oracle):
    case = synth.csm_case(31, n_beams=720)
    rng = np.random.RandomState(5)
    scans = []
    for k in range(10):
        truth = (0.3 * (rng.rand() - 0.5), 0.3 * (rng.rand() - 0.5), 0.2 * (rng.rand() - 0.5))
        angles, ranges = synth.cast_scan(case["segs"], truth, 720, 2 * math.pi, 5.7296)
        init = (truth[0] + 0.1, truth[1] - 0.08, truth[2] + 0.02)
        scans.append(dict(angles=angles, ranges=ranges, init_pose=init))
    graphs = api.Context(0)
    plain = api.Context(0, tuning_off=L.TUNE_NO_GRAPHS)
    for c in (graphs, plain):
        c.upload_grid(1, case["grid"])
    shapes = [(1.0, 1.0, math.radians(10), 4), (0.6, 0.8, math.radians(6), 4)]
    for rep in range(3):                       # every shape is seen often enough to be recorded and replayed
        for sc in scans:
            for rx, ry, rt, Lr in shapes:
                a = _match(graphs, case, sc, rx, ry, rt, Lr)
                b = _match(plain, case, sc, rx, ry, rt, Lr)
                assert a["raw"] == b["raw"] and a["estimated_pose"] == b["estimated_pose"]
                if rep == 2:
                    c2 = dict(case, angles=sc["angles"], ranges=sc["ranges"], init_pose=sc["init_pose"])
                    lit = oracle.csm(c2, rx, ry, rt, Lr)
                    assert (a["raw"]["best_x"], a["raw"]["best_y"], a["raw"]["best_theta"], a["raw"]["score"]) == \
                        (lit["bestX"], lit["bestY"], lit["bestT"], lit["scoreMax"])
    # a new map under the same id (the frontend re-uploads its latest map before every match)
    case2 = synth.csm_case(32, n_beams=720)
    for c in (graphs, plain):
        c.upload_grid(1, case2["grid"])
    for sc in scans[:4]:
        a = _match(graphs, case2, sc, *shapes[0])
        b = _match(plain, case2, sc, *shapes[0])
        assert a["raw"] == b["raw"]
    graphs.close()
    plain.close()
