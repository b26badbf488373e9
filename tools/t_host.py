import sys, os, math, time
sys.path[:0]=['.','my-lidar-graph-slam-v2_amd']
import bench
from csm_hip import api
n=int(sys.argv[1])
ctx=api.Context(0)
q,_=bench.make_loop_queries(ctx,0,n)
prep=ctx.prepare_queries(q)
for i in range(3):
    t=time.perf_counter(); o=ctx.bnb_match_batch(prep,*bench.LOOP_PARAMS,as_records=True); print("call %.3f ms"%((time.perf_counter()-t)*1e3), file=sys.stderr)
