"""Multi-GPU loop detection: one process per GPU, queries sharded in contiguous
blocks (the reference's own two-core split is first half / second half,
src/my_lidar_graph_slam/mapping/loop_detector_fpga_parallel.cpp:42-46), one
all-gather of the fixed-size best records (48 B per query) where the reference
concatenates the per-core result vectors (loop_detector_fpga_parallel.cpp:53-56).

torch.distributed is only the transport ("nccl" = RCCL over xGMI on the GPU
node, "gloo" in the CPU tests); the records are plain bytes.
"""
import ctypes as C

import numpy as np

from . import _lib as L

RECORD_BYTES = C.sizeof(L.Result)   # 48


def shard_bounds(n_queries, rank, world):
    """Contiguous block of rank `rank`: sizes differ by at most one, earlier
    ranks take the larger blocks."""
    base, rem = divmod(n_queries, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# csm_result as a numpy record: the same 48 bytes, no per-record Python loop
RECORD_DTYPE = np.dtype([("found", "<i4"), ("best_x", "<i4"), ("best_y", "<i4"), ("best_theta", "<i4"),
                         ("key", "<u8"), ("sum_values", "<u4"), ("known", "<u4"), ("tie_count", "<u4"),
                         ("flags", "<u4"), ("score", "<f8")])
assert RECORD_DTYPE.itemsize == RECORD_BYTES


def records_to_bytes(raw_results):
    """list of dict (api.result_to_dict) -> uint8 [n, 48] in csm_result layout."""
    arr = np.zeros(len(raw_results), RECORD_DTYPE)
    for name in RECORD_DTYPE.names:
        arr[name] = [r[name] for r in raw_results]
    return arr.view(np.uint8).reshape(len(raw_results), RECORD_BYTES)


def bytes_to_records(buf):
    arr = np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1, RECORD_BYTES).view(RECORD_DTYPE).reshape(-1)
    cols = {name: arr[name].tolist() for name in RECORD_DTYPE.names}
    return [dict(zip(cols, vals)) for vals in zip(*cols.values())]


def allgather_records(local, n_queries, group=None, device=None):
    """All-gather the per-rank record blocks into query order.

    local: uint8 [m, 48] for this rank's shard_bounds block. Blocks are padded
    to the largest block so a single fixed-size all_gather suffices."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_queries, rank, world)
    assert local.shape == (hi - lo, RECORD_BYTES), (local.shape, hi - lo)
    max_block = -(-n_queries // world)
    send = torch.zeros(max_block * RECORD_BYTES, dtype=torch.uint8, device=device)
    if hi > lo:
        send[:(hi - lo) * RECORD_BYTES] = torch.from_numpy(local.reshape(-1)).to(send.device)
    recv = torch.zeros(world * max_block * RECORD_BYTES, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.cpu().numpy().reshape(world, max_block, RECORD_BYTES)
    out = np.zeros((n_queries, RECORD_BYTES), np.uint8)
    for r in range(world):
        a, b = shard_bounds(n_queries, r, world)
        out[a:b] = recv[r, :b - a]
    return out


class LoopDetectorBranchBoundHIP:
    """Search part of LoopDetectorBranchBound::Detect
    (src/my_lidar_graph_slam/mapping/loop_detector_branch_bound.cpp:59-156),
    constructor arguments as in src/my_lidar_graph_slam/loop_detector_factory.cpp:161-183.

    `scorer(queries) -> uint8 [m, 48] records (or a list of raw result dicts)`
    defaults to the HIP batch; the CPU (gloo) tests pass a stand-in so that the
    sharding and gather logic can run without a GPU."""

    def __init__(self, name, ctx, range_x, range_y, range_theta, node_height_max,
                 score_threshold, known_rate_threshold, group=None, scorer=None, device=None):
        self.name = name
        self.ctx = ctx
        self.params = (range_x, range_y, range_theta, node_height_max)
        self.score_threshold = score_threshold
        self.known_rate_threshold = known_rate_threshold
        self.group = group
        self.device = device
        self._scorer = scorer or self._hip_scorer

    def _hip_scorer(self, queries):
        rx, ry, rt, H = self.params
        outs = self.ctx.bnb_match_batch(queries, rx, ry, rt, H, self.score_threshold,
                                        self.known_rate_threshold, as_records=True)
        return outs.record_bytes().reshape(-1, RECORD_BYTES)

    def detect(self, queries, grids=None):
        """queries: list of dict(map_id, geom, angles, ranges, rel_pose, init_pose);
        grids: optional {map_id: uint16 grid} uploaded once per id (the
        mPrecompMaps cache of loop_detector_branch_bound.hpp:98).
        Returns (all raw records in query order, indices of the found ones)."""
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(self.group) if distributed else 1
        rank = dist.get_rank(self.group) if distributed else 0
        n = len(queries)
        lo, hi = shard_bounds(n, rank, world)
        mine = queries[lo:hi]
        if grids is not None and self.ctx is not None:
            for q in mine:
                if not self.ctx.has_grid(q["map_id"]):
                    self.ctx.upload_grid(q["map_id"], grids[q["map_id"]])
        local = self._scorer(mine) if mine else np.zeros((0, RECORD_BYTES), np.uint8)
        if not isinstance(local, np.ndarray):
            local = records_to_bytes(local)
        if distributed and world > 1:
            allrec = allgather_records(local, n, self.group, self.device)
        else:
            allrec = local
        records = bytes_to_records(allrec)
        found = [i for i, r in enumerate(records) if r["found"]]
        return records, found


class LoopDetectorCorrelativeHIP:
    """Search part of LoopDetectorCorrelative::Detect
    (src/my_lidar_graph_slam/mapping/loop_detector_correlative.cpp:59-156): the
    correlative matcher with a coarse map cached per local-map id and the
    detector's two thresholds."""

    def __init__(self, name, ctx, low_resolution, range_x, range_y, range_theta,
                 score_threshold, known_rate_threshold):
        self.name = name
        self.ctx = ctx
        self.low_resolution = low_resolution
        self.ranges = (range_x, range_y, range_theta)
        self.score_threshold = score_threshold
        self.known_rate_threshold = known_rate_threshold

    def detect(self, queries, grids=None):
        """Returns (summaries in query order, indices of the found ones). All
        queries go to the device in one batch (csm_correlative_match_batch)."""
        for q in queries:
            if not self.ctx.has_grid(q["map_id"]):
                self.ctx.upload_grid(q["map_id"], grids[q["map_id"]])
        outs = self.ctx.correlative_match_batch(
            queries, self.ranges[0], self.ranges[1], self.ranges[2], self.low_resolution,
            self.score_threshold, self.known_rate_threshold) if queries else []
        return outs, [i for i, o in enumerate(outs) if o["pose_found"]]


class LoopDetectorGridSearchHIP:
    """Search part of LoopDetectorGridSearch::Detect
    (src/my_lidar_graph_slam/mapping/loop_detector_grid_search.cpp:52-156): the
    brute-force matcher with the detector's two thresholds, one query after the
    other as the reference runs them."""

    def __init__(self, name, ctx, range_x, range_y, range_theta, step_x, step_y, step_theta,
                 score_threshold, known_rate_threshold):
        if not (0.0 < score_threshold <= 1.0 and 0.0 < known_rate_threshold <= 1.0):
            raise ValueError("thresholds must lie in (0, 1]")   # loop_detector_grid_search.cpp:47-48
        self.name = name
        self.ctx = ctx
        self.window = (range_x, range_y, range_theta, step_x, step_y, step_theta)
        self.score_threshold = score_threshold
        self.known_rate_threshold = known_rate_threshold

    def detect(self, queries, grids=None):
        outs = []
        for q in queries:
            if not self.ctx.has_grid(q["map_id"]):
                self.ctx.upload_grid(q["map_id"], grids[q["map_id"]])
            outs.append(self.ctx.grid_search_match(
                q["map_id"], q["geom"], q["angles"], q["ranges"], q["rel_pose"], q["init_pose"],
                *self.window, self.score_threshold, self.known_rate_threshold))
        return outs, [i for i, o in enumerate(outs) if o["pose_found"]]
