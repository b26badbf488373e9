"""Block-sparse upload (csm_upload_grid_blocks): the reference keeps a grid as blocks of
2^k x 2^k cells, allocated on demand (inc/grid_map_new/grid_map.hpp:255-263). The device
de-blocking must give, byte for byte, the dense level csm_upload_grid gets from
GridMap::CopyValues' output, and the allocation bitmap the cost function reads."""
import math

import numpy as np
import pytest

from csm_hip import api, synth

pytestmark = pytest.mark.gpu


def _blocks_of(grid, k, allocated):
    bs = 1 << k
    br, bc = grid.shape[0] // bs, grid.shape[1] // bs
    blocks = []
    for r in range(br):
        for c in range(bc):
            blocks.append(grid[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs].copy() if allocated[r, c] else None)
    return blocks, br, bc


@pytest.mark.parametrize("k,br,bc,seed", [(4, 13, 11, 0), (3, 25, 30, 1), (5, 4, 7, 2), (0, 9, 17, 3), (4, 25, 25, 4)])
def test_deblocked_level_equals_dense_upload(gpu_ctx, k, br, bc, seed):
    rng = np.random.RandomState(seed)
    bs = 1 << k
    grid = rng.randint(0, 65536, (br * bs, bc * bs)).astype(np.uint16)
    grid[rng.rand(*grid.shape) < 0.3] = 0
    allocated = rng.rand(br, bc) < 0.7
    if seed == 4:
        allocated[:3, :] = False            # first known row / column far from the origin
        allocated[:, :2] = False
    dense = grid.copy()
    for r in range(br):
        for c in range(bc):
            if not allocated[r, c]:
                dense[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs] = 0      # what CopyValues hands out
    blocks, _, _ = _blocks_of(grid, k, allocated)
    gpu_ctx.upload_grid_blocks(801, blocks, br, bc, k)
    got = gpu_ctx.download_level(801, 0)
    assert np.array_equal(got, dense)
    gpu_ctx.upload_grid(802, dense)
    assert np.array_equal(gpu_ctx.download_level(802, 0), got)
    # the box maximum built from either is the same level
    w = min(4, dense.shape[0], dense.shape[1])
    gpu_ctx.build_pyramid(801, [1, w])
    gpu_ctx.build_pyramid(802, [1, w])
    assert np.array_equal(gpu_ctx.download_level(801, 1), gpu_ctx.download_level(802, 1))
    gpu_ctx.release_grid(801)
    gpu_ctx.release_grid(802)


def test_block_upload_matches_dense_upload_in_search_and_cost(gpu_ctx, oracle):
    """A match and the cost / covariance tail on a block-uploaded map equal those on the dense
    upload with the same allocation bitmap (unallocated blocks read 0.5 in the cost function,
    cost_function_square_error.cpp:330-333 through ProbabilityOr)."""
    case = synth.csm_case(5)
    grid = case["grid"]
    k = 4
    bs = 1 << k
    assert grid.shape[0] % bs == 0 and grid.shape[1] % bs == 0
    br, bc = grid.shape[0] // bs, grid.shape[1] // bs
    allocated = np.array([[bool(grid[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs].any()) for c in range(bc)]
                          for r in range(br)])
    allocated[br // 2, bc // 2] = True         # an allocated block that may hold only unknown cells
    blocks, _, _ = _blocks_of(grid, k, allocated)
    gpu_ctx.upload_grid_blocks(811, blocks, br, bc, k)
    gpu_ctx.upload_grid(812, grid)
    gpu_ctx.set_block_allocation(812, k, allocated.astype(np.uint8))
    args = (case["geom"], case["angles"], case["ranges"], case["rel_pose"], case["init_pose"],
            1.0, 1.0, math.radians(10), 4, 0.0, 0.0)
    a = gpu_ctx.correlative_match(811, *args)
    b = gpu_ctx.correlative_match(812, *args)
    assert a["raw"] == b["raw"] and a["estimated_pose"] == b["estimated_pose"]
    lit = oracle.csm(case, 1.0, 1.0, math.radians(10), 4)
    assert (a["raw"]["best_x"], a["raw"]["best_y"], a["raw"]["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"])
    q = [dict(map_id=m, geom=case["geom"], angles=case["angles"], ranges=case["ranges"], rel_pose=case["rel_pose"],
              init_pose=case["init_pose"]) for m in (811, 812)]
    ra, rb = gpu_ctx.cost_covariance_batch(q, [a["best_sensor_pose"], b["best_sensor_pose"]])
    assert ra.keys() == rb.keys()
    for key in ra:
        assert np.array_equal(np.asarray(ra[key]), np.asarray(rb[key])), key
    gpu_ctx.release_grid(811)
    gpu_ctx.release_grid(812)
