"""tools/t_group.py N M: the N-query branch-and-bound batch through a csm_group of M
members that all sit on GPU 0 (each member prepares and launches its block from its
own host thread and stream): does overlapping one member's host work with another's
kernels beat one context? Prints ms per call for M members and for a plain context."""
import sys
import time

import torch  # noqa: F401  (first: its bundled HIP runtime must be the one the library binds)

sys.path[:0] = [".", "my-lidar-graph-slam-v2_amd"]
import bench  # noqa: E402
from csm_hip import api  # noqa: E402

n, m = int(sys.argv[1]), int(sys.argv[2])
grp = api.Group([0] * m)
queries = []
for k, ctx in enumerate(grp.members):
    lo, hi = api.host_shard_bounds(n, k, m)
    q, _ = bench.make_loop_queries(ctx, lo, hi)
    queries += q
best = None
for rep in range(4):
    t = time.perf_counter()
    out = grp.bnb_match_batch(queries, *bench.LOOP_PARAMS, as_records=True)
    dt = (time.perf_counter() - t) * 1e3
    best = dt if best is None else min(best, dt)
    print("group of %d: call %.3f ms (marshalling included)" % (m, dt), file=sys.stderr)
found_g = sum(o["pose_found"] for o in out)
grp.close()
ctx = api.Context(0)
q, _ = bench.make_loop_queries(ctx, 0, n)
prep = ctx.prepare_queries(q)
for rep in range(4):
    t = time.perf_counter()
    o = ctx.bnb_match_batch(prep, *bench.LOOP_PARAMS, as_records=True)
    print("one context: call %.3f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
print("found", found_g, sum(x["pose_found"] for x in o), file=sys.stderr)
