#!/usr/bin/env python3
"""bench.py -- candidate poses scored per second by the HIP correlative path.

Workload at every N: BASELINE.json configs[1] per GPU -- frontend CSM, 1080-beam
scan over 270 deg, 400x400 grid @ 5 cm, +-2 m / +-30 deg window at 5 cm / 0.5 deg,
L = 4 (121 x 84 x 84 = 853,776 candidate poses per scan, 2160 algorithmic bytes
each). One step = SCANS_PER_STEP (64) independent scans, hit indices and grid
already resident in HBM, scored by one batched launch chain
(csm_score_windows_dev; CSM_BENCH_MODE=streams scores them one launch chain per
scan instead). With N > 1 every rank scores its own
scans (weak scaling, no data-path collective) and the per-scan best records
(48 B) are all-gathered over RCCL at the end of each step, as the loop
detector's result exchange does.

Prints ONE JSON line on rank 0 (contract in the task statement), including
"roofline" (dominant kernel = fine scoring kernel, HIP events inside the
library on its launch stream) and "cpu_baseline" (CPU oracle, 1 core, bounded
sample; rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))

SCANS_PER_STEP = int(os.environ.get("CSM_BENCH_SCANS", "64"))   # independent scans per step
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def make_workload(rank, n_scans):
    """configs[1]: one room map, n_scans scans from different true poses."""
    import numpy as np
    from csm_hip import api, synth
    grid, geom, segs = synth.make_room(1000 + rank, rows=400, cols=400, res=0.05)
    rng = np.random.RandomState(4242 + rank)
    rx, ry, rt, L = 4.0, 4.0, math.radians(60.0), 4
    scans = []
    for i in range(n_scans):
        truth = (0.5 * (rng.rand() - 0.5), 0.5 * (rng.rand() - 0.5), 0.3 * (rng.rand() - 0.5))
        angles, ranges = synth.cast_scan(segs, truth, n_beams=1080, fov=1.5 * math.pi,
                                         max_range=5.7296)
        init = (truth[0] + 0.31 * (rng.rand() - 0.5), truth[1] + 0.31 * (rng.rand() - 0.5),
                truth[2] + 0.1 * (rng.rand() - 0.5))
        sx, sy, st = api.host_search_step(geom[0], ranges)
        wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
        col, row = api.host_project(geom, init, st, wt, angles, ranges)
        scans.append(dict(angles=angles, ranges=ranges, init_pose=init, truth=truth,
                          rel_pose=(0.0, 0.0, 0.0), col=col, row=row, win=(wx, wy, wt)))
    return dict(grid=grid, geom=geom, scans=scans, params=(rx, ry, rt, L))


def pmc_traffic():
    """HBM bytes per fine-kernel launch from the committed PMC passes
    (profiles/r01_pmc_fine_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate runs of this same bench, FETCH_SIZE doubled per the gfx950
    note). A live bench run cannot collect counters itself; null when the file
    is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fine_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(wl, budget_s=12.0, max_scans=400):
    """The CPU oracle (literal ScanMatcherCorrelative sweep with pruning), one
    core, on the first scans of the same workload."""
    from oracle import oracle as O
    rx, ry, rt, L = wl["params"]
    coarse = O.boxmax(wl["grid"], L)
    t0 = time.time()
    n_done, cands, fine = 0, 0, 0
    for i in range(max_scans):
        sc = wl["scans"][i % len(wl["scans"])]
        case = dict(grid=wl["grid"], geom=wl["geom"], angles=sc["angles"], ranges=sc["ranges"],
                    rel_pose=sc["rel_pose"], init_pose=sc["init_pose"])
        r = O.csm(case, rx, ry, rt, L, coarse=coarse)
        wx, wy, wt = r["winX"], r["winY"], r["winT"]
        nx = -(-(2 * wx + 1) // L) * L
        ny = -(-(2 * wy + 1) // L) * L
        cands += (2 * wt + 1) * nx * ny
        fine += r["fineEvaluated"]
        n_done += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return dict(value=cands / dt, unit="candidate poses/s", cores=1, kind="port",
                sample="%d scan(s) of the same workload, %.1f s wall, coarse pruning on: "
                       "%.3g window poses/s nominal, %.3g fully evaluated fine poses/s"
                       % (n_done, dt, cands / dt, fine / dt))


def run_loop_workload(args, rank, world, dev, dev_index, rehearse):
    """configs[2] per GPU (configs[3] at 8 GPUs): one 1080-beam scan against 256
    candidate submaps (400x400 @ 5 cm, 3-level pyramids), 2.5 m x 2.5 m x 0.5 rad,
    thresholds 0.55 / 0.6. Maps and pyramids resident; a step = one
    csm_bnb_match_batch call on this rank's 256 queries + the all-gather of the
    48-byte records."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from csm_hip import api, parallel, synth
    n_sub = 256
    ctx = api.Context(dev_index)
    rng = np.random.RandomState(77 + rank)
    queries = []
    for i in range(n_sub):
        c = synth.csm_case(100000 * rank + i, n_beams=1080, fov=1.5 * math.pi)
        ctx.upload_grid(i, c["grid"])
        init = tuple(np.asarray(c["truth"]) + rng.uniform(-0.6, 0.6, 3) * (1, 1, 0.15))
        queries.append(dict(map_id=i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=(0.0, 0.0, 0.0), init_pose=init))
    params = (2.5, 2.5, 0.5, 2, 0.55, 0.6)
    # the query array is marshalled once, as a C++ caller holds it (csm_loop_query[])
    prepared = ctx.prepare_queries(queries)

    def step():
        outs = ctx.bnb_match_batch(prepared, *params, as_records=True)
        if world > 1:
            rec = torch.from_numpy(outs.record_bytes())
            if not rehearse:
                rec = rec.to(dev)
            out = torch.zeros(world * rec.numel(), dtype=torch.uint8, device=rec.device)
            dist.all_gather_into_tensor(out, rec)
        return outs

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(1, args.warmup)):
        outs = step()
    fence()
    ctx.lib.csm_enable_kernel_timing(ctx._ctx, 2)
    ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    fence()
    dt = time.perf_counter() - t0
    ctx.enable_kernel_timing(False)
    fine_ms, fine_n = ctx.kernel_time("score_fine")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        leaves = outs.total("candidates")
        outs = list(outs)
        alg = 2.0 * 1080 * leaves
        avg = fine_ms / max(1, fine_n) * 1e-3
        print(json.dumps({
            "metric": "candidate poses scored/sec (CSM+BnB), 1/2/4/8 GPU; % HBM roofline",
            "value": leaves * args.steps * world / dt, "unit": "candidate poses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[2] per GPU: branch-and-bound loop detection, 1 scan vs 256 "
                                   "submaps, 3-level grids, 2.5 m x 2.5 m x 0.5 rad, thresholds 0.55/0.6; "
                                   "host-inclusive batch call (scans in host memory, query array marshalled once, maps resident)",
                       "leaves_per_step_per_gpu": leaves, "found": sum(o["pose_found"] for o in outs),
                       "flagged": sum(1 for o in outs if o["raw"]["flags"]),
                       "parallelism": "queries sharded per GPU, all-gather of 48-B records"
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_score_batch (leaf level)",
                         "achieved": alg / avg / 1e9 if avg > 0 else 0.0, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg / avg / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0,
                         "traffic": None, "avg_launch_us": avg * 1e6, "launches": fine_n},
        }), flush=True)
    ctx.close()


def run_map_workload(args, rank, world, dev, dev_index, rehearse):
    """SURVEY 8(f) rank 4, the frontend cycle: a step = rebuild the latest map
    from the last 10 scans (10 x 1080 beams, csm_construct_map_from_scans; scans
    arrive as host arrays like the reference's ScanData) + match the newest scan
    against the device-resident result (csm_correlative_match, +-0.5 m / 0.25 rad).
    Ranks are independent replicas. Not the BASELINE metric."""
    import torch
    import torch.distributed as dist
    from csm_hip import api, synth
    ctx = api.Context(dev_index)
    mc = synth.map_case(2 + rank, n_scans=10, n_beams=1080)
    shape, info = ctx.construct_map_from_scans(7, mc["shape"], mc["map_pose"], mc["nodes"])
    last = mc["nodes"][-1]
    mp = mc["map_pose"]
    c, s_ = math.cos(mp[2]), math.sin(mp[2])
    dx, dy = last["pose"][0] + 0.07 - mp[0], last["pose"][1] - 0.05 - mp[1]
    init = (c * dx + s_ * dy, -s_ * dx + c * dy, last["pose"][2] + 0.01 - mp[2])

    split = [0.0, 0.0]

    def step():
        t_a = time.perf_counter()
        sh, inf = ctx.construct_map_from_scans(7, shape, mc["map_pose"], mc["nodes"])
        t_b = time.perf_counter()
        out = ctx.correlative_match(7, (sh["res"], sh["off_x"], sh["off_y"]), last["angles"], last["ranges"],
                                    last["rel_pose"], init, 1.0, 1.0, 0.25, 4, 0.0, 0.0)
        split[0] += t_b - t_a
        split[1] += time.perf_counter() - t_b
        return inf, out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(1, args.warmup)):
        info, out = step()
    fence()
    # the per-step marshalling allocates enough Python objects to trigger a full
    # collection of torch's large heap (one ~50 ms pause per ~50 steps): keep
    # the collector out of the timed region
    import gc
    gc.collect()
    gc.disable()
    split[0] = split[1] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info, out = step()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    split_timed = [split[0] / args.steps * 1e3, split[1] / args.steps * 1e3]
    # kernel breakdown from a few extra steps with the event timers on (they
    # allocate events per launch, so they stay out of the timed region)
    k_steps = 10
    ctx.enable_kernel_timing(True)
    step()
    ctx.reset_kernel_timing()
    for _ in range(k_steps):
        step()
    fence()
    ctx.enable_kernel_timing(False)
    build_ms, build_n = ctx.kernel_time("map_build")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        line = {
            "metric": "map cell updates/sec (latest-map build + match per step)",
            "value": info["cell_updates"] * args.steps * world / dt, "unit": "cell updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "frontend cycle: ConstructMapFromScans of 10 x 1080 beams (%d x %d cells, "
                                   "%d rays, %d cell updates) + ScanMatcherCorrelative on the resident map; "
                                   "host-inclusive calls" % (shape["rows"], shape["cols"], info["rays"],
                                                             info["cell_updates"]),
                       "map_kernels_us_per_step": build_ms / max(1, build_n) * 1e3,
                       "build_ms_per_step": split_timed[0],
                       "match_ms_per_step": split_timed[1],
                       "match_kernels_us": {k: ctx.kernel_time(k)[0] / k_steps * 1e3 for k in
                                            ("boxmax", "project", "bin", "score_coarse", "score_fine",
                                             "finalize")},
                       "match_found": out["pose_found"], "match_flags": out["raw"]["flags"],
                       "match_tie_count": out["raw"]["tie_count"],
                       "match_setup_us": out["input_setup_us"], "match_optimization_us": out["optimization_us"],
                       "parallelism": "independent replicas" if world > 1 else "single GPU"},
        }
        if not args.no_cpu_baseline:
            from oracle import oracle
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 5.0:
                oracle.construct_map(mc["shape"], mc["map_pose"], mc["nodes"])
                reps += 1
            cdt = (time.perf_counter() - t0) / reps
            line["cpu_baseline"] = dict(value=info["cell_updates"] / cdt, unit="cell updates/s", cores=1,
                                        kind="port", sample="%d builds of the same 10 scans, map build only, "
                                        "%.2f ms each" % (reps, cdt * 1e3))
        print(json.dumps(line), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="also measure the host-inclusive batched detector entry on the same scans "
                         "(off by default: its launches share the dominant kernel's symbol and would "
                         "mix into a profile of this command)")
    ap.add_argument("--no-extras", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--workload", choices=["csm", "loop", "map"], default="csm",
                    help="csm (default): BASELINE configs[1]; loop: configs[2]/[3], 256 candidate "
                         "submaps per GPU through the branch-and-bound batch + all-gather; map: the "
                         "frontend cycle (latest-map build from 10 scans + match), not the BASELINE metric")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    # CSM_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend, to walk
    # the multi-process control flow on a one-GPU box (numbers meaningless)
    rehearse = os.environ.get("CSM_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    ge.build()          # file-locked: ranks take turns, later ones find it built
    from csm_hip import api, _lib

    if args.workload in ("loop", "map"):
        (run_loop_workload if args.workload == "loop" else run_map_workload)(
            args, rank, world, dev, dev_index, rehearse)
        if world > 1:
            dist.destroy_process_group()
        return

    wl = make_workload(rank, SCANS_PER_STEP)
    rx, ry, rt, L = wl["params"]
    # The scans of a step are independent: they alternate between N_STREAMS
    # matcher contexts (one HIP stream each), so one scan's small kernels (bin,
    # arg-max, finalize: a few workgroups each) run beside another scan's
    # full-chip scoring kernel instead of leaving the chip idle.
    # CSM_BENCH_MODE=batch (default): the step's scans go through the batched launch
    # chain in one call (csm_score_windows_dev); =streams: one launch chain per scan
    # (csm_score_window_dev), alternating between N_STREAMS contexts.
    batch_mode = os.environ.get("CSM_BENCH_MODE", "batch") == "batch"
    n_streams = 1 if batch_mode else max(1, min(SCANS_PER_STEP, int(os.environ.get("CSM_BENCH_STREAMS", "2"))))
    stream = torch.cuda.current_stream(dev)
    side_streams = [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
    ctxs = []
    for k in range(n_streams):
        c = api.Context(dev_index)
        c.set_stream((stream if k == 0 else side_streams[k - 1]).cuda_stream)
        c.upload_grid(1, wl["grid"])
        c.build_pyramid(1, [1, L])
        ctxs.append(c)
    ctx = ctxs[0]

    n_beams = 1080
    windows, cols, rows_ = [], [], []
    cands_per_step = 0
    for sc in wl["scans"]:
        wx, wy, wt = sc["win"]
        w = ctx.make_window(2 * wt + 1, n_beams, wx, wy, L, 1, api.host_min_known(n_beams, 0.0), 0.0)
        windows.append(w)
        cols.append(torch.from_numpy(sc["col"]).to(dev))
        rows_.append(torch.from_numpy(sc["row"]).to(dev))
        nx = -(-(2 * wx + 1) // L) * L
        ny = -(-(2 * wy + 1) // L) * L
        cands_per_step += (2 * wt + 1) * nx * ny
    rec_bytes = 48
    results = torch.zeros(SCANS_PER_STEP * rec_bytes, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(world * SCANS_PER_STEP * rec_bytes, dtype=torch.uint8, device=dev)

    prepared = ctx.prepare_windows([1] * SCANS_PER_STEP, windows, [c.data_ptr() for c in cols],
                                   [r.data_ptr() for r in rows_])

    def step():
        if batch_mode:
            ctx.score_windows_dev(prepared, results.data_ptr())
        else:
            for s2 in side_streams:
                s2.wait_stream(stream)       # the previous step's all-gather has read `results`
            for i in range(SCANS_PER_STEP):
                ctxs[i % n_streams].score_window_dev(1, windows[i], cols[i].data_ptr(), rows_[i].data_ptr(),
                                                     results.data_ptr() + i * rec_bytes)
            for s2 in side_streams:
                stream.wait_stream(s2)
        if world > 1:
            if rehearse:
                host = results.cpu()
                out = torch.zeros(world * host.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, host)
            else:
                dist.all_gather_into_tensor(gathered, results)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # events around the dominant kernel only inside the timed region; the
    # other kernels are timed in a short extra pass afterwards
    for c in ctxs:
        c.lib.csm_enable_kernel_timing(c._ctx, 2)
        c.reset_kernel_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0

    def timer_sum(name):
        ms = n = 0
        for c in ctxs:
            a, b = c.kernel_time(name)
            ms, n = ms + a, n + b
        return ms, n
    for c in ctxs:
        c.enable_kernel_timing(False)
    fine_ms, fine_n = timer_sum("score_fine")
    for c in ctxs:
        c.lib.csm_enable_kernel_timing(c._ctx, 1)
        c.reset_kernel_timing()
    step()
    fence()
    for c in ctxs:
        c.enable_kernel_timing(False)
    coarse_ms, coarse_n = timer_sum("score_coarse")
    bin_ms, bin_n = timer_sum("bin")
    fin_ms, fin_n = timer_sum("finalize")
    arg_ms, arg_n = timer_sum("argmax")

    # extra (not `value`): the same scans through the batched detector entry
    # (csm_correlative_match_batch: scans arrive as host arrays, projection on
    # the device, 64 queries per call) -- what LoopDetectorCorrelative-style
    # callers get when queries are independent
    batched = None
    if rank == 0 and args.extras:
        qs = []
        for rep in range(max(1, 64 // SCANS_PER_STEP)):
            for sc in wl["scans"]:
                init = (sc["init_pose"][0] + 0.01 * rep, sc["init_pose"][1] - 0.01 * rep, sc["init_pose"][2])
                qs.append(dict(map_id=1, geom=wl["geom"], angles=sc["angles"], ranges=sc["ranges"],
                               rel_pose=sc["rel_pose"], init_pose=init))
        ctx_b = api.Context(dev_index)       # its own stream, as a detector object has
        ctx_b.upload_grid(1, wl["grid"])
        qs = ctx_b.prepare_queries(qs)       # the csm_loop_query[] a C++ caller holds
        ctx_b.correlative_match_batch(qs, rx, ry, rt, L, 0.0, 0.0, as_records=True)
        samples = []
        for _ in range(9):
            tb0 = time.perf_counter()
            outs = ctx_b.correlative_match_batch(qs, rx, ry, rt, L, 0.0, 0.0, as_records=True)
            samples.append(time.perf_counter() - tb0)
        tb = sorted(samples)[len(samples) // 2]      # median: host calls jitter
        ctx_b.enable_kernel_timing(True)
        ctx_b.reset_kernel_timing()
        ctx_b.correlative_match_batch(qs, rx, ry, rt, L, 0.0, 0.0)
        kms = {k: ctx_b.kernel_time(k)[0] for k in ("project", "bin", "score_coarse", "score_fine", "finalize")}
        ctx_b.close()
        batched = {"value": outs.total("candidates") / tb, "unit": "candidate poses/s",
                   "queries_per_call": qs.n, "ms_per_call": tb * 1e3, "kernel_ms": kms,
                   "note": "median of 9 calls; host-inclusive: scans in host memory, projection + "
                           "search on device, summaries back on the host; one GPU"}

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # sanity: every scan found a pose and the record decodes
    rec = np.frombuffer(results.cpu().numpy().tobytes(), dtype=np.int32).reshape(SCANS_PER_STEP, 12)
    n_found = int(rec[:, 0].sum())

    if rank == 0:
        total = cands_per_step * args.steps * world
        value = total / dt
        windows_per_launch = SCANS_PER_STEP if batch_mode else 1
        cands_per_launch = cands_per_step / SCANS_PER_STEP * windows_per_launch
        alg_bytes = 2.0 * n_beams * cands_per_launch
        avg_fine_s = (fine_ms / max(1, fine_n)) * 1e-3
        achieved = alg_bytes / avg_fine_s / 1e9 if avg_fine_s > 0 else 0.0
        out = {
            "metric": "candidate poses scored/sec (CSM+BnB), 1/2/4/8 GPU; % HBM roofline",
            "value": value,
            "unit": "candidate poses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",   # uint16 cells, exact u32/u64 integer sums; f64 replay of the winner
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: frontend CSM, 1080-beam scan, 400x400@5cm grid, "
                            "+-2 m/+-30 deg window at 5 cm/0.5 deg, L=4",
                "scans_per_step": SCANS_PER_STEP,
                "mode": "batch: csm_score_windows_dev, one launch chain per step" if batch_mode
                        else "streams: csm_score_window_dev per scan on %d stream(s)" % n_streams,
                "streams": n_streams,
                "candidates_per_scan": cands_per_step / SCANS_PER_STEP,
                "beams": n_beams,
                "parallelism": "scans sharded per GPU, all-gather of 48-B best records" if world > 1
                               else "single GPU",
                "poses_found": n_found,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_score_batch (fine level, %d windows per launch)" % SCANS_PER_STEP if batch_mode
                          else "k_score (fine level)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(),
                "avg_launch_us": avg_fine_s * 1e6,
                "launches": fine_n,
                "algorithmic_bytes_per_launch": alg_bytes,
                "other_kernels_avg_us": {
                    "score_coarse": coarse_ms / max(1, coarse_n) * 1e3,
                    "bin": bin_ms / max(1, bin_n) * 1e3,
                    "finalize": fin_ms / max(1, fin_n) * 1e3,
                    "argmax": arg_ms / max(1, arg_n) * 1e3,
                },
            },
        }
        if batched is not None:
            out["batched"] = batched
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)

    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
