#!/usr/bin/env python3
"""End-to-end (host-inclusive) numbers quoted in DESIGN.md: not the bench
metric. Run on the GPU box:  python tools/measure_e2e.py [n_submaps]

1. config 2 through csm_correlative_match(): host projection (glibc sin/cos,
   1 thread) + H2D of the hit indices + kernels + D2H of the 48-byte record.
2. config 3 through csm_bnb_match_batch(): 1 scan vs N submaps (default 256),
   3-level pyramids (H = 2), 2.5 m x 2.5 m x 0.5 rad, thresholds 0.55 / 0.6;
   host projection (threaded), H2D of r*cos / r*sin, pyramids, kernels.
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))

import numpy as np  # noqa: E402

from csm_hip import api, synth  # noqa: E402


def main():
    n_sub = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ctx = api.Context(0)
    out = {}

    case = synth.csm_case(5, n_beams=1080, fov=1.5 * math.pi, init_error=(0.31, -0.27, 0.08))
    ctx.upload_grid(1, case["grid"])
    m = api.ScanMatcherCorrelativeHIP("e2e", 4, 4.0, 4.0, math.radians(60), ctx=ctx)
    for _ in range(3):
        r = m.optimize_pose(None, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                            case["init_pose"], map_id=1)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        r = m.optimize_pose(None, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                            case["init_pose"], map_id=1)
    dt = (time.perf_counter() - t0) / reps
    out["config2_correlative_match"] = dict(ms_per_call=dt * 1e3, candidates=r["candidates"],
                                            poses_per_s=r["candidates"] / dt,
                                            found=r["pose_found"])

    # config 2, batched: the same window for 8 / 64 scans in one call
    rng2 = np.random.RandomState(11)
    for nb in (8, 64):
        qs = []
        for i in range(nb):
            init = tuple(np.asarray(case["truth"]) + rng2.uniform(-0.3, 0.3, 3) * (1, 1, 0.2))
            qs.append(dict(map_id=1, geom=case["geom"], angles=case["angles"], ranges=case["ranges"],
                           rel_pose=case["rel_pose"], init_pose=init))
        qs = ctx.prepare_queries(qs)          # the csm_loop_query[] a C++ caller holds
        ctx.correlative_match_batch(qs, 4.0, 4.0, math.radians(60), 4, 0.0, 0.0)
        t0 = time.perf_counter()
        outs = ctx.correlative_match_batch(qs, 4.0, 4.0, math.radians(60), 4, 0.0, 0.0)
        dt = time.perf_counter() - t0
        cands = sum(o["candidates"] for o in outs)
        out["config2_batch_%d" % nb] = dict(ms_per_batch=dt * 1e3, poses_per_s=cands / dt,
                                            found=sum(o["pose_found"] for o in outs),
                                            flagged=sum(1 for o in outs if o["raw"]["flags"]))

    # config 3: one scan against n_sub submaps
    base = synth.csm_case(1000, n_beams=1080, fov=1.5 * math.pi)
    rng = np.random.RandomState(3)
    queries, grids = [], {}
    for i in range(n_sub):
        c = synth.csm_case(1000 + i, n_beams=1080, fov=1.5 * math.pi)
        grids[5000 + i] = c["grid"]
        init = tuple(np.asarray(c["truth"]) + rng.uniform(-0.6, 0.6, 3) * (1, 1, 0.15))
        queries.append(dict(map_id=5000 + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=(0.0, 0.0, 0.0), init_pose=init))
    t0 = time.perf_counter()
    for k, g in grids.items():
        ctx.upload_grid(k, g)
    t_up = time.perf_counter() - t0
    queries = ctx.prepare_queries(queries)
    t0 = time.perf_counter()
    outs = ctx.bnb_match_batch(queries, 2.5, 2.5, 0.5, 2, 0.55, 0.6)     # builds pyramids
    t_first = time.perf_counter() - t0
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    outs = ctx.bnb_match_batch(queries, 2.5, 2.5, 0.5, 2, 0.55, 0.6)     # pyramids cached
    t_cached = time.perf_counter() - t0
    leaves = sum(o["candidates"] for o in outs)
    kt = {k: ctx.kernel_time(k) for k in ("bnb_index", "bin", "score_coarse", "score_fine", "finalize")}
    out["config3_bnb_batch"] = dict(
        submaps=n_sub, leaves=leaves, found=sum(o["pose_found"] for o in outs),
        flagged=sum(1 for o in outs if o["raw"]["flags"]),
        upload_s=t_up, first_call_s=t_first, cached_call_s=t_cached,
        leaves_per_s_cached=leaves / t_cached,
        kernel_ms={k: v[0] for k, v in kt.items()},
        device_only_leaves_per_s=leaves / (sum(v[0] for v in kt.values()) * 1e-3))
    # config 5: exhaustive global window on a 2000 x 2000 grid
    for k in grids:
        ctx.release_grid(k)
    big = synth.csm_case(7, rows=2000, cols=2000, res=0.025, n_beams=1080, fov=1.5 * math.pi,
                         max_range=5.7296, init_error=(3.1, -2.7, 1.3), n_boxes=10)
    ctx.upload_grid(9, big["grid"])
    m5 = api.ScanMatcherCorrelativeHIP("e2e5", 4, 20.0, 20.0, 2 * math.pi, ctx=ctx)
    r = m5.optimize_pose(None, big["geom"], big["angles"], big["ranges"], big["rel_pose"],
                         big["init_pose"], map_id=9)
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    r = m5.optimize_pose(None, big["geom"], big["angles"], big["ranges"], big["rel_pose"],
                         big["init_pose"], map_id=9)
    dt = time.perf_counter() - t0
    kt = {k: ctx.kernel_time(k) for k in ("project", "bin", "score_coarse", "score_fine", "finalize")}
    out["config5_global_window"] = dict(
        s_per_call=dt, candidates=r["candidates"], poses_per_s=r["candidates"] / dt,
        algorithmic_TB=r["candidates"] * 2160 / 1e12, kernel_ms={k: v[0] for k, v in kt.items()},
        fine_algorithmic_GBs=r["candidates"] * 2160 / (kt["score_fine"][0] * 1e-3) / 1e9,
        flags=r["raw"]["flags"], found=r["pose_found"])
    # brute-force matcher at the grid-search loop detector's default window
    ctx.upload_grid(1, case["grid"])
    gs = (2.5, 2.5, 0.5, 0.05, 0.05, 0.005, 0.3, 0.5)
    r = ctx.grid_search_match(1, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                              case["init_pose"], *gs)
    ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    r = ctx.grid_search_match(1, case["geom"], case["angles"], case["ranges"], case["rel_pose"],
                              case["init_pose"], *gs)
    dt = time.perf_counter() - t0
    out["grid_search_default_window"] = dict(
        ms_per_call=dt * 1e3, candidates=r["candidates"], poses_per_s=r["candidates"] / dt,
        kernel_ms=ctx.kernel_time("grid_search")[0], found=r["pose_found"])
    # map building: the frontend's latest map, 10 scans x 1080 beams
    mc = synth.map_case(2, n_scans=10, n_beams=1080)
    shape, info = ctx.construct_map_from_scans(77, mc["shape"], mc["map_pose"], mc["nodes"])
    ctx.reset_kernel_timing()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        _, info = ctx.construct_map_from_scans(77, shape, mc["map_pose"], mc["nodes"])
    dt = (time.perf_counter() - t0) / reps
    out["map_build_10x1080"] = dict(ms_per_call=dt * 1e3, rays=info["rays"], cell_updates=info["cell_updates"],
                                    updates_per_s=info["cell_updates"] / dt, host_us=info["host_us"],
                                    device_us=info["device_us"],
                                    kernel_ms=ctx.kernel_time("map_build")[0] / reps)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
