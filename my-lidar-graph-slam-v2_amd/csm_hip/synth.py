"""Seeded synthetic inputs of the shapes BASELINE.json / SURVEY.md section 8(d)
name: a rectangular room with clutter rasterised into a uint16 occupancy grid
(0 = unknown, 1..65534 known) and an analytic ray-cast scan. Pure numpy; used
by tests, bench.py and smoke().
"""
import math

import numpy as np


def _segments_of_box(x0, y0, x1, y1):
    return [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]


def make_room(seed, rows=400, cols=400, res=0.05, half_x=None, half_y=None,
              n_boxes=6, levels=None, interior_unknown=0.03, origin="center"):
    """Returns (grid uint16 [rows, cols], geom (res, offX, offY), segments).

    levels: None -> walls in [50000, 52000], free in [3000, 5000] (many distinct
    values); an int n -> values quantised to n levels (tie-prone maps).
    """
    rng = np.random.RandomState(seed)
    ext_x, ext_y = cols * res, rows * res
    # a non-round offset keeps scan points off cell edges by default
    off_x = -0.5 * ext_x + 0.0137 + 0.01 * rng.rand()
    off_y = -0.5 * ext_y - 0.0219 + 0.01 * rng.rand()
    hx = half_x if half_x is not None else ext_x * (0.26 + 0.08 * rng.rand())
    hy = half_y if half_y is not None else ext_y * (0.20 + 0.08 * rng.rand())
    if origin == "low_edge":
        # the room's low walls sit 1-2 cells inside the map's low edges, so
        # scan points project into / next to the negative edge band
        off_x = -hx - 1.6 * res
        off_y = -hy - 1.4 * res
    elif origin == "aligned":
        # everything on exact multiples of the resolution: points on cell edges
        off_x = -0.5 * ext_x
        off_y = -0.5 * ext_y
        hx = round(hx / res) * res
        hy = round(hy / res) * res
    segs = _segments_of_box(-hx, -hy, hx, hy)
    boxes = []
    for _ in range(n_boxes):
        w, h = 0.3 + 0.9 * rng.rand(), 0.3 + 0.9 * rng.rand()
        cx = (rng.rand() * 2 - 1) * (hx - w - 0.4)
        cy = (rng.rand() * 2 - 1) * (hy - h - 0.4)
        if abs(cx) < 1.0 and abs(cy) < 1.0:
            continue  # keep the sensor neighbourhood free
        boxes.append((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2))
        segs += _segments_of_box(*boxes[-1])

    xs = off_x + (np.arange(cols) + 0.5) * res
    ys = off_y + (np.arange(rows) + 0.5) * res
    X, Y = np.meshgrid(xs, ys)
    inside = (np.abs(X) < hx) & (np.abs(Y) < hy)
    wall = (np.abs(np.abs(X) - hx) <= res) & (np.abs(Y) <= hy + res) | \
           (np.abs(np.abs(Y) - hy) <= res) & (np.abs(X) <= hx + res)
    in_box = np.zeros_like(inside)
    box_wall = np.zeros_like(inside)
    for (x0, y0, x1, y1) in boxes:
        ib = (X > x0) & (X < x1) & (Y > y0) & (Y < y1)
        edge = ib & ~((X > x0 + res) & (X < x1 - res) & (Y > y0 + res) & (Y < y1 - res))
        in_box |= ib
        box_wall |= edge
    grid = np.zeros((rows, cols), np.uint16)
    free = inside & ~in_box & ~wall
    free_v = rng.randint(3000, 5001, size=(rows, cols))
    wall_v = rng.randint(50000, 52001, size=(rows, cols))
    if levels is not None:
        q = 65534 // max(1, levels - 1)
        free_v = np.maximum(1, (free_v // q) * q)
        wall_v = np.minimum(65534, (wall_v // q) * q)
    grid[free] = free_v[free]
    occ = wall | box_wall
    grid[occ] = wall_v[occ]
    if interior_unknown > 0:
        drop = free & (rng.rand(rows, cols) < interior_unknown)
        grid[drop] = 0
    return grid, (res, off_x, off_y), np.asarray(segs, np.float64)


def cast_scan(segs, pose, n_beams=360, fov=2 * math.pi, max_range=5.7296, noise=0.0, seed=0):
    """Analytic ray casting against axis-aligned segments. angles are
    -fov/2 + fov*i/n (360 deg) or linspace over the fov otherwise. The longest
    beam is pinned to max_range so that the reference's step-theta formula
    gives the configured angular step."""
    px, py, pth = pose
    if abs(fov - 2 * math.pi) < 1e-12:
        angles = -math.pi + 2 * math.pi * np.arange(n_beams) / n_beams
    else:
        angles = np.linspace(-fov / 2, fov / 2, n_beams)
    dx, dy = np.cos(angles + pth), np.sin(angles + pth)
    best = np.full(n_beams, np.inf)
    for (x0, y0, x1, y1) in segs:
        ex, ey = x1 - x0, y1 - y0
        den = dx * ey - dy * ex
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((x0 - px) * ey - (y0 - py) * ex) / den
            u = ((x0 - px) * dy - (y0 - py) * dx) / den
        ok = (np.abs(den) > 1e-12) & (t > 1e-6) & (u >= 0) & (u <= 1)
        best = np.where(ok & (t < best), t, best)
    ranges = np.minimum(best, max_range)
    if noise > 0:
        ranges = ranges + np.random.RandomState(seed).randn(n_beams) * noise
        ranges = np.clip(ranges, 0.05, max_range)
    ranges[int(np.argmax(ranges))] = max_range
    return angles.astype(np.float64), ranges.astype(np.float64)


def csm_case(seed, rows=400, cols=400, res=0.05, n_beams=360, fov=2 * math.pi,
             max_range=5.7296, levels=None, init_error=(0.17, -0.12, 0.02),
             rel_pose=(0.0, 0.0, 0.0), truth=None, origin="center", **room_kw):
    """One scan-vs-map case: grid + geometry + scan + initial pose."""
    grid, geom, segs = make_room(seed, rows, cols, res, levels=levels, origin=origin, **room_kw)
    rng = np.random.RandomState(seed + 7919)
    if truth is None:
        truth = (0.013 + 0.4 * (rng.rand() - 0.5), -0.021 + 0.4 * (rng.rand() - 0.5),
                 0.03 + 0.2 * (rng.rand() - 0.5))
    angles, ranges = cast_scan(segs, truth, n_beams, fov, max_range)
    init = (truth[0] + init_error[0], truth[1] + init_error[1], truth[2] + init_error[2])
    return dict(grid=grid, geom=geom, angles=angles, ranges=ranges, truth=truth,
                init_pose=init, rel_pose=rel_pose, segs=segs)


def map_case(seed, n_scans=10, n_beams=1080, fov=1.5 * math.pi, max_range=8.0, res=0.05,
             step=0.12, rel_pose=(0.0, 0.0, 0.0), noise=0.0, **room_kw):
    """Input of a map build (GridMapBuilder::ConstructMapFromScans): a short
    trajectory of scan nodes inside a synthetic room. Returns dict(nodes,
    map_pose, shape) with shape = the freshly constructed 1 m x 1 m map the
    reference starts from (grid_map_builder.cpp:80)."""
    _, _, segs = make_room(seed, 400, 400, res, **room_kw)
    rng = np.random.RandomState(seed + 4242)
    x, y, th = 0.3 * (rng.rand() - 0.5), 0.3 * (rng.rand() - 0.5), rng.rand() * 2 * math.pi
    nodes = []
    for k in range(n_scans):
        pose = (x, y, th)
        sensor = (x + math.cos(th) * rel_pose[0] - math.sin(th) * rel_pose[1],
                  y + math.sin(th) * rel_pose[0] + math.cos(th) * rel_pose[1], th + rel_pose[2])
        angles, ranges = cast_scan(segs, sensor, n_beams, fov, max_range, noise=noise, seed=seed + k)
        nodes.append(dict(pose=pose, angles=angles, ranges=ranges, rel_pose=rel_pose,
                          min_range=0.05, max_range=max_range))
        th += 0.15 * (rng.rand() - 0.5)
        x += step * math.cos(th)
        y += step * math.sin(th)
    block = 16
    n = int(math.ceil(1.0 / res))
    n = (n + block - 1) // block * block
    shape = dict(res=res, off_x=0.0, off_y=0.0, rows=n, cols=n, log2_block=4)
    return dict(nodes=nodes, map_pose=nodes[0]["pose"], shape=shape, segs=segs)
