"""tools/t_single.py: configs[1] one query per call (csm_correlative_match), 200 calls;
run under rocprofv3 --kernel-trace --stats to see what the 0.16 ms are made of."""
import sys
import time

import torch  # noqa: F401

sys.path[:0] = [".", "my-lidar-graph-slam-v2_amd"]
import bench  # noqa: E402
from csm_hip import api  # noqa: E402

wl = bench.make_workload(0, 8)
rx, ry, rt, L = wl["params"]
ctx = api.Context(0)
ctx.upload_grid(1, wl["grid"])
lat = []
for k in range(208):
    sc = wl["scans"][k % 8]
    t = time.perf_counter()
    ctx.correlative_match(1, wl["geom"], sc["angles"], sc["ranges"], sc["rel_pose"], sc["init_pose"], rx, ry, rt, L)
    lat.append(time.perf_counter() - t)
lat = sorted(lat[8:])
print("median %.1f us, min %.1f us" % (lat[len(lat) // 2] * 1e6, lat[0] * 1e6), file=sys.stderr)
