#!/usr/bin/env python3
"""Map building alone (csm_construct_map_from_scans), for profiling:
python tools/bench_map.py [n_scans] [n_beams] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))

from csm_hip import api, synth  # noqa: E402


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n_beams = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    ctx = api.Context(0)
    mc = synth.map_case(2, n_scans=n_scans, n_beams=n_beams)
    shape, info = ctx.construct_map_from_scans(77, mc["shape"], mc["map_pose"], mc["nodes"])
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        _, info = ctx.construct_map_from_scans(77, shape, mc["map_pose"], mc["nodes"])
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps(dict(n_scans=n_scans, n_beams=n_beams, rows=shape["rows"], cols=shape["cols"],
                          ms_per_call=dt * 1e3, rays=info["rays"], cell_updates=info["cell_updates"],
                          updates_per_s=info["cell_updates"] / dt, host_us=info["host_us"],
                          device_us=info["device_us"], kernel_ms=ctx.kernel_time("map_build")[0] / reps)))


if __name__ == "__main__":
    main()
