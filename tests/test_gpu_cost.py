"""SURVEY 8(f) rank 3 on the device: cost, covariance and the linear-solver
refinement, batched (csm_cost_covariance_batch / csm_linear_solver_batch),
against the CPU restatement (oracle/cost_oracle.cpp) at the tolerance
include/csm_hip.h states -- f64, not bit-exact: device sin / cos, tree
reductions, restated 3x3 solves."""
import math

import numpy as np
import pytest

from csm_hip import api, synth

pytestmark = pytest.mark.gpu

REL_COST = 1e-10        # costs, Hessian entries (relative)
REL_COV = 1e-8          # covariance entries, relative to the largest entry
ABS_POSE = 1e-7         # refined pose (m, rad) at equal iteration counts


def _batch(n, seed0=3000, n_beams=1080):
    rng = np.random.RandomState(17)
    queries, cases = [], []
    for i in range(n):
        c = synth.csm_case(seed0 + i, n_beams=n_beams, fov=1.5 * math.pi,
                           rel_pose=(0.1 * (i % 3), -0.03, 0.02 * (i % 2)))
        init = tuple(np.asarray(c["truth"]) + rng.uniform(-0.04, 0.04, 3) * (1, 1, 0.2))
        cases.append(c)
        queries.append(dict(map_id=5000 + i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=c["rel_pose"], init_pose=init))
    return queries, cases


def _close(a, b, rel):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= rel * np.maximum(np.abs(b).max(), 1e-300))


def test_cost_and_covariance_batch(gpu_ctx, oracle):
    queries, cases = _batch(24)
    for q, c in zip(queries, cases):
        gpu_ctx.upload_grid(q["map_id"], c["grid"])
    poses = [api.host_compound(q["init_pose"], q["rel_pose"]) for q in queries]
    got = gpu_ctx.cost_covariance_batch(queries, poses, 1e4)
    for q, c, p, g in zip(queries, cases, poses, got):
        # the derived allocation bitmap: a 16x16 block is allocated iff it holds a known cell
        blocks = c["grid"].reshape(25, 16, 25, 16).max(axis=(1, 3)) > 0
        alloc = blocks.astype(np.uint8)
        want_cost = oracle.cost(c["grid"], c["geom"], c["angles"], c["ranges"], p, alloc=alloc)
        h, _ = oracle.hessian_residual(c["grid"], c["geom"], c["angles"], c["ranges"], p, alloc=alloc)
        cov = oracle.covariance(c["grid"], c["geom"], c["angles"], c["ranges"], p, 1e4, alloc=alloc)
        assert abs(g["normalized_cost"] * len(c["angles"]) - want_cost) <= REL_COST * want_cost
        assert g["normalized_initial_cost"] == g["normalized_cost"] and g["iterations"] == 0
        assert _close(g["hessian"], h, REL_COST)
        assert _close(g["covariance"], cov, REL_COV)
        assert list(g["best_sensor_pose"]) == list(p)
    for q in queries:
        gpu_ctx.release_grid(q["map_id"])


def test_linear_solver_batch(gpu_ctx, oracle):
    queries, cases = _batch(32, seed0=3100)
    for q, c in zip(queries, cases):
        gpu_ctx.upload_grid(q["map_id"], c["grid"])
    got = gpu_ctx.linear_solver_batch(queries, 10, 1e-4, 1e-4, 1e4)
    same = 0
    for q, c, g in zip(queries, cases, got):
        alloc = (c["grid"].reshape(25, 16, 25, 16).max(axis=(1, 3)) > 0).astype(np.uint8)
        w = oracle.linear_solver(c["grid"], c["geom"], c["angles"], c["ranges"], q["rel_pose"],
                                 q["init_pose"], 10, 1e-4, 1e-4, 1e4, alloc=alloc)
        assert g["sensor_pose"] == w["sensor_pose"]            # Compound on the host: bit-exact
        assert abs(g["normalized_initial_cost"] - w["normalized_initial_cost"]) <= REL_COST * w["normalized_initial_cost"]
        # (the reference accepts every damped Gauss-Newton step: the cost may also rise)
        if g["iterations"] != w["iterations"]:
            continue        # |cost change| on the convergence threshold: allowed, must stay rare
        same += 1
        assert np.all(np.abs(np.asarray(g["best_sensor_pose"]) - np.asarray(w["best_sensor_pose"])) <= ABS_POSE)
        assert np.all(np.abs(np.asarray(g["estimated_pose"]) - np.asarray(w["estimated_pose"])) <= ABS_POSE)
        assert abs(g["normalized_cost"] - w["normalized_cost"]) <= 1e-9 * w["normalized_cost"]
        assert g["lambda_"] == w["lambda_"]
        assert _close(g["covariance"], w["covariance"], 1e-6)   # the pose differs by up to ABS_POSE
    assert same >= len(queries) - 1
    for q in queries:
        gpu_ctx.release_grid(q["map_id"])


def test_block_allocation_bitmap_changes_the_reads(gpu_ctx, oracle):
    """A caller-provided bitmap (GridMap::IsAllocated per block) replaces the
    derived one: marking every block allocated turns the 0.5 of empty blocks into
    the 0 of unknown cells, on both sides alike."""
    queries, cases = _batch(4, seed0=3200, n_beams=720)
    for q, c in zip(queries, cases):
        gpu_ctx.upload_grid(q["map_id"], c["grid"])
    poses = [api.host_compound(q["init_pose"], q["rel_pose"]) for q in queries]
    derived = gpu_ctx.cost_covariance_batch(queries, poses, 1e4)
    for q in queries:
        gpu_ctx.set_block_allocation(q["map_id"], 4, np.ones((25, 25), np.uint8))
    full = gpu_ctx.cost_covariance_batch(queries, poses, 1e4)
    changed = 0
    for q, c, p, d, f in zip(queries, cases, poses, derived, full):
        want = oracle.cost(c["grid"], c["geom"], c["angles"], c["ranges"], p, alloc=None)
        assert abs(f["normalized_cost"] * len(c["angles"]) - want) <= REL_COST * want
        changed += f["normalized_cost"] != d["normalized_cost"]
        gpu_ctx.set_block_allocation(q["map_id"], 4, None)      # back to the derived rule
    assert changed > 0
    again = gpu_ctx.cost_covariance_batch(queries, poses, 1e4)
    assert [a["normalized_cost"] for a in again] == [d["normalized_cost"] for d in derived]
    for q in queries:
        gpu_ctx.release_grid(q["map_id"])


def test_refinement_after_a_device_search(gpu_ctx, oracle):
    """The reference's sequence for one loop-detection query: branch-and-bound search,
    then the final matcher on its estimate (loop_detector_branch_bound.cpp:101-127)."""
    queries, cases = _batch(6, seed0=3300)
    rng = np.random.RandomState(3)
    for q, c in zip(queries, cases):
        gpu_ctx.upload_grid(q["map_id"], c["grid"])
        q["init_pose"] = tuple(np.asarray(c["truth"]) + rng.uniform(-0.5, 0.5, 3) * (1, 1, 0.15))
    found = gpu_ctx.bnb_match_batch(queries, 2.5, 2.5, 0.5, 2, 0.3, 0.5)
    hits = [(q, c, f) for q, c, f in zip(queries, cases, found) if f["pose_found"]]
    assert hits
    second = [dict(q, init_pose=tuple(f["estimated_pose"])) for q, c, f in hits]
    refined = gpu_ctx.linear_solver_batch(second)
    for (q, c, f), s, r in zip(hits, second, refined):
        alloc = (c["grid"].reshape(25, 16, 25, 16).max(axis=(1, 3)) > 0).astype(np.uint8)
        w = oracle.linear_solver(c["grid"], c["geom"], c["angles"], c["ranges"], q["rel_pose"],
                                 s["init_pose"], alloc=alloc)
        if r["iterations"] == w["iterations"]:
            assert np.all(np.abs(np.asarray(r["estimated_pose"]) - np.asarray(w["estimated_pose"])) <= ABS_POSE)
        # the refinement stays within a cell or two of the search's answer
        assert np.all(np.abs(np.asarray(r["estimated_pose"])[:2] - np.asarray(f["estimated_pose"])[:2]) < 0.15)
    for q in queries:
        gpu_ctx.release_grid(q["map_id"])
