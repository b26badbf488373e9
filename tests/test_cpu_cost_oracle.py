"""CPU suite: the oracle's restatement of CostSquareError / ScanMatcherLinearSolver
(oracle/cost_oracle.cpp; parity unpinned against the reference, which needs
Eigen3) pinned internally: analytic gradient against central differences of
Cost(), the two restated Eigen calls against numpy, the ProbabilityOr(.., 0.5)
rules, and that the refinement lowers the cost."""
import math

import numpy as np

from csm_hip import synth
from oracle import oracle as O


def _case(seed):
    c = synth.csm_case(seed, n_beams=540, fov=1.5 * math.pi)
    pose = np.asarray(c["truth"]) + np.array([0.031, -0.024, 0.006])
    return c, pose


def test_gradient_is_minus_two_times_residual():
    for seed in (1, 2, 3):
        c, pose = _case(seed)
        h, r = O.hessian_residual(c["grid"], c["geom"], c["angles"], c["ranges"], pose)
        assert np.allclose(h, h.T)
        eps = 1e-6
        for k in range(3):
            d = np.zeros(3)
            d[k] = eps
            fd = (O.cost(c["grid"], c["geom"], c["angles"], c["ranges"], pose + d) -
                  O.cost(c["grid"], c["geom"], c["angles"], c["ranges"], pose - d)) / (2 * eps)
            assert abs(fd - (-2.0 * r[k])) < 1e-4 * max(1.0, abs(fd))


def test_restated_eigen_calls():
    rng = np.random.RandomState(0)
    for _ in range(50):
        a = rng.randn(3, 3)
        m = a @ a.T + np.diag(rng.rand(3) * 1e-3)
        b = rng.randn(3)
        assert np.allclose(O.inverse3(m) @ m, np.eye(3), atol=1e-9)
        assert np.allclose(O.solve3(m, b), np.linalg.solve(m, b), rtol=1e-9, atol=1e-12)
    # pivoting: a matrix whose first column is tiny
    m = np.array([[1e-12, 2.0, 0.0], [0.0, 1.0, 3.0], [4.0, 0.0, 1.0]])
    assert np.allclose(O.solve3(m, [1.0, 2.0, 3.0]), np.linalg.solve(m, [1.0, 2.0, 3.0]))


def test_covariance_is_scaled_inverse_hessian():
    c, pose = _case(4)
    h, _ = O.hessian_residual(c["grid"], c["geom"], c["angles"], c["ranges"], pose)
    cov = O.covariance(c["grid"], c["geom"], c["angles"], c["ranges"], pose, 1e4)
    assert np.allclose(cov @ h / 1e4, np.eye(3), atol=1e-9)


def test_probability_or_half_rules():
    """Outside the map and in unallocated blocks a read gives 0.5; an unknown cell
    of an allocated block gives 0 (grid_map.cpp:423-436)."""
    grid = np.zeros((32, 32), np.uint16)
    grid[8:24, 8:24] = 40000
    geom = (0.05, -0.8, -0.8)
    angles, ranges = np.array([0.0]), np.array([0.2])
    # the beam ends at (x, y): cost = (1 - bilinear)^2
    def cost_at(x, y, alloc=None):
        return O.cost(grid, geom, angles, ranges, (x - 0.2, y, 0.0), alloc=alloc, log2_block=3)
    p_wall = O.lut()[40000]
    inside = cost_at(-0.8 + 0.05 * 12.5, -0.8 + 0.05 * 12.5)
    assert abs(inside - (1 - p_wall) ** 2) < 1e-12
    # far outside, mid cell (index 116.5): the reference clamps only the UPPER neighbour
    # (xc1 = min(xc0 + 1, cols - 1), cost_function_square_error.cpp:330-333), so three of
    # the four reads are outside the map (0.5) and the fourth is cell (31, 31): unknown, 0
    far_outside = cost_at(-0.8 + 0.05 * 116.5, -0.8 + 0.05 * 116.5)
    assert abs(far_outside - (1 - 0.25 * 1.5) ** 2) < 1e-12
    unknown_allocated = cost_at(-0.8 + 0.05 * 2.5, -0.8 + 0.05 * 2.5)
    assert abs(unknown_allocated - 1.0) < 1e-12      # probability 0 -> error 1
    alloc = np.ones((4, 4), np.uint8)
    alloc[0, 0] = 0
    unallocated = cost_at(-0.8 + 0.05 * 2.5, -0.8 + 0.05 * 2.5, alloc)
    assert abs(unallocated - 0.25) < 1e-12


def test_refinement_lowers_the_cost_and_stops():
    for seed in (5, 6, 7):
        c, pose = _case(seed)
        r = O.linear_solver(c["grid"], c["geom"], c["angles"], c["ranges"], (0.08, -0.02, 0.01), tuple(pose))
        assert 1 <= r["iterations"] <= 10
        assert r["normalized_cost"] < r["normalized_initial_cost"]
        assert 1e-8 <= r["lambda_"] <= 1e-4
        assert np.allclose(r["covariance"], r["covariance"].T, rtol=1e-9, atol=1e-12)
