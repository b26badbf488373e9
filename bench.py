#!/usr/bin/env python3
"""bench.py -- candidate poses scored per second by the HIP correlative path.

N = 1 (default): BASELINE.json configs[1] -- frontend CSM, 1080-beam scan over
270 deg, 400x400 grid @ 5 cm, +-2 m / +-30 deg window at 5 cm / 0.5 deg, L = 4
(123 x 84 x 84 = 867,888 candidate poses per scan, 2160 algorithmic bytes
each). One step = SCANS_PER_STEP (3072) scans, hit indices and grid already
resident in HBM, scored 256 windows per batched launch chain
(csm_score_windows_dev), so that the K timed steps last seconds, not
milliseconds. The same line carries, under "configs", short driver-timed runs
of configs[2] (256 submaps, pyramid build reported separately), configs[3] at
N = 1 (2048 submaps on one GPU: the strong-scaling base), configs[4]
(exhaustive 2000x2000) and the host-inclusive single-query latency of
configs[1].

N > 1 (default): BASELINE.json configs[3] -- branch-and-bound loop detection,
2048 candidate submaps sharded in contiguous blocks of 2048 / N queries per
rank (loop_detector_fpga_parallel.cpp:42-46), one all-gather of the 48-byte
best records over RCCL per step (":53-56"). Total work is fixed: strong
scaling. `--workload csm` keeps configs[1] per GPU instead (independent
replicas, weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement), including
"roofline" (dominant kernel = fine / leaf scoring kernel, HIP events inside the
library on its launch stream) and, at N = 1, "cpu_baseline" (CPU oracle on one
core and on all cores, bounded samples).

Roofline: the fine kernel gathers from LDS, not from HBM (a 400x400 grid is
320 KB; measured HBM traffic is < 0.1 % of the algorithmic bytes), so the bound
reported is the LDS read rate: achieved = 4 B x (cell entries x candidates) per
launch / launch time against 256 B/clk/CU x 256 CUs x 2.4 GHz. The metric's own
figure (2 B x beams x candidates against 8 TB/s of HBM) is kept as
"logical_hbm_frac"; it exceeds 1 because those bytes never leave the chip.
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

WINDOWS_PER_LAUNCH = int(os.environ.get("CSM_BENCH_WINDOWS", "256"))   # scans per batched launch chain
SCANS_PER_STEP = int(os.environ.get("CSM_BENCH_SCANS", "3072"))     # scans per step
DISTINCT_SCANS = int(os.environ.get("CSM_BENCH_DISTINCT", "512"))    # different scans generated
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# MI355X_MICROARCH.md, LDS: 64 banks x 4 B per clock per CU (ds_read_b64 / b128
# rate), 256 CUs, 2.4 GHz max clock. ds_read_b32 reaches half of it.
LDS_PEAK_GBS = 256.0 * 256 * 2.4
METRIC = "candidate poses scored/sec (CSM+BnB), 1/2/4/8 GPU; % HBM roofline"
N_BEAMS = 1080
LOOP_PARAMS = (2.5, 2.5, 0.5, 2, 0.55, 0.6)    # launcher_settings_default.json:130-131, 143-146
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_fine_traffic.json")


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
        angles, ranges = synth.cast_scan(segs, truth, n_beams=N_BEAMS, fov=1.5 * math.pi,
                                         max_range=5.7296)
        init = (truth[0] + 0.31 * (rng.rand() - 0.5), truth[1] + 0.31 * (rng.rand() - 0.5),
                truth[2] + 0.1 * (rng.rand() - 0.5))
        sx, sy, st = api.host_search_step(geom[0], ranges)
        wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
        col, row = api.host_project(geom, init, st, wt, angles, ranges)
        scans.append(dict(angles=angles, ranges=ranges, init_pose=init, truth=truth,
                          rel_pose=(0.0, 0.0, 0.0), col=col, row=row, win=(wx, wy, wt)))
    return dict(grid=grid, geom=geom, scans=scans, params=(rx, ry, rt, L))


def cell_entries(col, row, rows, cols, x_lo, y_lo, x_hi, y_hi, max_mult=15):
    """Distinct (theta, cell) pairs a window's beams land on, counted the way
    the binning kernel forms its entries (a cell with more than 15 beams is
    split; beams that cannot touch the grid for any candidate are dropped):
    the LDS gathers one candidate needs."""
    import numpy as np
    nt = col.shape[0]
    total = 0
    for t in range(nt):
        r, c = row[t].astype(np.int64), col[t].astype(np.int64)
        ok = (r + y_hi >= 0) & (r <= rows - 1 - y_lo) & (c + x_hi >= 0) & (c <= cols - 1 - x_lo)
        key = (r[ok] + (1 << 20)) * (1 << 22) + (c[ok] + (1 << 20))
        _, cnt = np.unique(key, return_counts=True)
        total += int(((cnt + max_mult - 1) // max_mult).sum())
    return total


def pmc_traffic(path=None):
    """HBM bytes per fine-kernel launch from the committed PMC passes (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same command,
    FETCH_SIZE doubled per the gfx950 note). A live bench run cannot collect
    counters itself, so this is a committed measurement, labelled as such;
    (None, None) when the file is absent."""
    path = path or PMC_FILE
    try:
        with open(path) as f:
            d = json.load(f)
        src = "%s (committed; kernel %s, %s windows per launch, library %s)" % (
            os.path.relpath(path, ROOT), d.get("kernel", "?"), d.get("windows_per_launch", "?"),
            d.get("library_version", "?"))
        scale = 1.0
        wpl = d.get("windows_per_launch")
        if path == PMC_FILE and isinstance(wpl, int) and wpl > 0 and wpl != WINDOWS_PER_LAUNCH:
            scale = WINDOWS_PER_LAUNCH / float(wpl)
            src += "; scaled to %d windows per launch" % WINDOWS_PER_LAUNCH
        return float(d["hbm_bytes_per_launch"]) * scale, src
    except (OSError, KeyError, ValueError):
        return None, None


def verify_sample(wl, rec_raw, scans_per_step, n_distinct, n_check):
    """After the timed region: n_check records of the last step, spread over its
    launch chains, against the CPU oracle's literal ScanMatcherCorrelative sweep
    (the checker, never the thing measured). Raises on a mismatch; returns the
    number of records compared."""
    if n_check <= 0:
        return 0
    import numpy as np
    from csm_hip import _lib as L
    from oracle import oracle as O
    rx, ry, rt, Lr = wl["params"]
    coarse = O.boxmax(wl["grid"], Lr)
    picks = sorted(set(int(i) for i in np.linspace(0, scans_per_step - 1, n_check)))
    compared = 0
    for j in picks:
        sc = wl["scans"][j % n_distinct]
        r = L.Result.from_buffer_copy(rec_raw[48 * j:48 * (j + 1)])
        case = dict(grid=wl["grid"], geom=wl["geom"], angles=sc["angles"], ranges=sc["ranges"],
                    rel_pose=sc["rel_pose"], init_pose=sc["init_pose"])
        lit = O.csm(case, rx, ry, rt, Lr, coarse=coarse)
        if r.flags & (L.FLAG_EDGE_BAND | L.FLAG_KEY_TIE):
            continue            # to be finished by csm_resolve_window_dev: not final as it stands
        got = (r.found, r.best_x, r.best_y, r.best_theta, r.score)
        want = (lit["found"], lit["bestX"], lit["bestY"], lit["bestT"], lit["scoreMax"])
        if got != want:
            raise AssertionError("bench record %d differs from the oracle: %r != %r" % (j, got, want))
        compared += 1
    return compared


def cpu_baseline(wl, budget_s=10.0, max_scans=2000):
    """The CPU oracle (literal ScanMatcherCorrelative sweep with pruning) on the
    first scans of the same workload: one core (the reference is
    single-threaded on this path) and OpenMP over theta on all host cores."""
    from oracle import oracle as O
    rx, ry, rt, L = wl["params"]
    coarse = O.boxmax(wl["grid"], L)

    def run(fn):
        t0 = time.time()
        n_done, cands, fine, threads = 0, 0, 0, 1
        for i in range(max_scans):
            sc = wl["scans"][i % len(wl["scans"])]
            case = dict(grid=wl["grid"], geom=wl["geom"], angles=sc["angles"], ranges=sc["ranges"],
                        rel_pose=sc["rel_pose"], init_pose=sc["init_pose"])
            r = fn(case, rx, ry, rt, L, coarse=coarse)
            threads = r.get("threads", 1)
            wx, wy, wt = r["winX"], r["winY"], r["winT"]
            nx = -(-(2 * wx + 1) // L) * L
            ny = -(-(2 * wy + 1) // L) * L
            cands += (2 * wt + 1) * nx * ny
            fine += r["fineEvaluated"]
            n_done += 1
            if time.time() - t0 > budget_s:
                break
        dt = time.time() - t0
        return n_done, cands / dt, fine / dt, dt, threads

    O.csm_omp(dict(grid=wl["grid"], geom=wl["geom"], angles=wl["scans"][0]["angles"],
                   ranges=wl["scans"][0]["ranges"], rel_pose=(0.0, 0.0, 0.0),
                   init_pose=wl["scans"][0]["init_pose"]), rx, ry, rt, L, coarse=coarse)   # thread pool up
    n1, v1, f1, d1, _ = run(O.csm)
    nn, vn, fn_, dn, threads = run(O.csm_omp)
    return dict(value=v1, unit="candidate poses/s", cores=1, kind="port",
                sample="%d scan(s) of the same workload, %.1f s wall, coarse pruning on: "
                       "%.3g window poses/s nominal, %.3g fully evaluated fine poses/s"
                       % (n1, d1, v1, f1),
                all_cores=dict(value=vn, unit="candidate poses/s", cores=threads, kind="port",
                               sample="OpenMP over theta, %d threads, %d scan(s), %.1f s wall: %.3g window "
                                      "poses/s nominal, %.3g fully evaluated fine poses/s"
                                      % (threads, nn, dn, vn, fn_)))


# ------------------------------------------------------------------ loop detection (configs[2] / [3])

LOOP_GROUP = 256      # candidate submaps per query scan (BASELINE configs[2]: "1 scan vs 256 candidate submaps")
LOOP_ROOM = (4.4, 3.6)   # half extents of the room family, m (every wall within the 5.73 m range)


def loop_scan(group):
    """The query scan of block `group` of 256 candidate submaps: cast in the bare
    room of the family from a pose of its own; every submap of the block is the
    same room with its own clutter, map offset and initial-pose error."""
    import numpy as np
    from csm_hip import synth
    _, _, segs = synth.make_room(90000 + group, half_x=LOOP_ROOM[0], half_y=LOOP_ROOM[1], n_boxes=0)
    rng = np.random.RandomState(91000 + group)
    truth = (0.013 + 0.4 * (rng.rand() - 0.5), -0.021 + 0.4 * (rng.rand() - 0.5), 0.03 + 0.2 * (rng.rand() - 0.5))
    angles, ranges = synth.cast_scan(segs, truth, n_beams=N_BEAMS, fov=1.5 * math.pi, max_range=5.7296)
    return truth, angles, ranges


def make_loop_queries(ctx, lo, hi):
    """Submaps lo..hi-1 of the 2048-submap batch (seeds = query numbers, so every
    rank builds exactly its own block), uploaded under map id = query number.
    Queries i with the same i // 256 share one scan (the same arrays: the library
    stages a scan once per call however many maps it is matched against)."""
    import numpy as np
    from csm_hip import synth
    queries = []
    t_up = 0.0
    scans = {}
    for i in range(lo, hi):
        g = i // LOOP_GROUP
        if g not in scans:
            scans[g] = loop_scan(g)
        truth, angles, ranges = scans[g]
        grid, geom, _ = synth.make_room(100000 + i, half_x=LOOP_ROOM[0], half_y=LOOP_ROOM[1])
        rng = np.random.RandomState(77000 + i)
        t0 = time.perf_counter()
        ctx.upload_grid(i, grid)
        t_up += time.perf_counter() - t0
        init = tuple(np.asarray(truth) + rng.uniform(-0.6, 0.6, 3) * (1, 1, 0.15))
        queries.append(dict(map_id=i, geom=geom, angles=angles, ranges=ranges,
                            rel_pose=(0.0, 0.0, 0.0), init_pose=init))
    return queries, t_up


def loop_roofline(leaves, fine_ms, fine_n, entries_per_leaf_query=None, n_queries=None):
    alg = 2.0 * N_BEAMS * leaves
    avg = fine_ms / max(1, fine_n) * 1e-3
    # HBM bytes per leaf launch: the committed PMC passes of the 256-query batch (one map per query)
    traffic, traffic_src = pmc_traffic(os.path.join(ROOT, "profiles", "r03_loop_pmc_traffic.json"))
    if traffic is not None and n_queries not in (None, 256):
        traffic, traffic_src = traffic * n_queries / 256.0, traffic_src + "; scaled to %d queries" % n_queries
    d = {"bound": "lds", "kernel": "leaf level of the batch, all queries in one launch chain: k_score_jointf_batch<124, 6> (packed-fp32 bound pass) + k_bound_select + k_score_joint_list<124, 6> in two rounds (exact pass)", "unit": "GB/s", "peak": LDS_PEAK_GBS,
         "avg_launch_us": avg * 1e6, "launches": fine_n, "traffic": traffic, "traffic_source": traffic_src,
         "hbm_frac_measured": traffic / avg / 1e9 / HBM_PEAK_GBS if traffic is not None and avg > 0 else None,
         "logical_hbm_gbs": alg / avg / 1e9 if avg > 0 else 0.0,
         "logical_hbm_frac": alg / avg / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0}
    if entries_per_leaf_query is not None and avg > 0:
        d["achieved"] = 4.0 * entries_per_leaf_query / avg / 1e9
        d["frac"] = d["achieved"] / LDS_PEAK_GBS
    else:
        d["achieved"], d["frac"] = None, None
    return d


def loop_gathers(queries, prm):
    """4-byte LDS gathers one leaf pass over `queries` needs (cell entries x leaves)."""
    from csm_hip import api
    rx, ry, rt, H = prm[:4]
    total = 0
    stride = max(1, len(queries) // 64)          # a sample of <= ~64 queries, scaled up
    sample = queries[::stride]
    for q in sample:
        sx, sy, st = api.host_search_step(q["geom"][0], q["ranges"])
        wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
        big = 1 << H
        nx = -(-(2 * wx + 1) // big) * big
        ny = -(-(2 * wy + 1) // big) * big
        col, row = api.host_project(q["geom"], q["init_pose"], st, wt, q["angles"], q["ranges"])
        e = cell_entries(col, row, 400, 400, -wx, -wy, -wx + nx - 1, -wy + ny - 1)
        total += e * nx * ny
    return total * len(queries) / len(sample)


def run_loop_workload(args, rank, world, dev, dev_index, rehearse, stream, n_total, strong):
    """Branch-and-bound loop detection, n_total candidate submaps (400x400 @ 5 cm,
    3-level grids, one 1080-beam scan each), 2.5 m x 2.5 m x 0.5 rad, thresholds
    0.55 / 0.6, sharded in contiguous blocks over the ranks. Maps and pyramids
    resident; a step = one csm_bnb_match_batch call on this rank's block + ONE
    all-gather of the 48-byte records (device buffers, RCCL)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from csm_hip import api, parallel
    lo, hi = parallel.shard_bounds(n_total, rank, world)
    block = -(-n_total // world)
    ctx = api.Context(dev_index, tuning_off=args.tuning_off)
    ctx.set_stream(stream.cuda_stream)
    queries, _ = make_loop_queries(ctx, lo, hi)
    prepared = ctx.prepare_queries(queries)
    rec_bytes = parallel.RECORD_BYTES
    send = torch.zeros(block * rec_bytes, dtype=torch.uint8, device=dev)
    recv = torch.zeros(world * block * rec_bytes, dtype=torch.uint8, device=dev)

    gather_events = []          # (start, end) around every timed all-gather, on `stream`

    def step():
        outs = ctx.bnb_match_batch(prepared, *LOOP_PARAMS, as_records=True)
        if world > 1:
            ctx.copy_last_batch_records(send.data_ptr())      # device to device, on `stream`
            if rehearse:
                host = send.cpu()
                out = torch.zeros(world * host.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, host)
                recv.copy_(out)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                dist.all_gather_into_tensor(recv, send)       # RCCL; the current stream waits for it
                e1.record(stream)
                gather_events.append((e0, e1))
        return outs

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for _ in range(max(1, args.warmup)):
            outs = step()
        fence()
        ctx.lib.csm_enable_kernel_timing(ctx._ctx, 2)
        ctx.reset_kernel_timing()
        del gather_events[:]
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs = step()
        fence()
        dt = time.perf_counter() - t0
    ctx.enable_kernel_timing(False)
    fine_ms, fine_n = ctx.kernel_time("score_fine")
    bound_ms, bound_n = ctx.kernel_time("score_bound")      # the packed-fp32 bound pass of the leaf level
    if bound_n:
        fine_ms, fine_n = fine_ms + bound_ms, bound_n       # per batch call: bound pass + both exact rounds
    gather_ms = [a.elapsed_time(b) for a, b in gather_events]
    leaves_local = outs.total("candidates")
    if world > 1:
        tt = torch.tensor([dt, float(leaves_local)], dtype=torch.float64, device=None if rehearse else dev)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt, leaves = float(tmax[0].item()), int(tt[1].item())
        # every rank must now hold every record, in query order
        allrec = recv.cpu().numpy().reshape(world, block, rec_bytes)
        mine = np.frombuffer(outs.record_bytes(), np.uint8).reshape(-1, rec_bytes)
        assert np.array_equal(allrec[rank, :hi - lo], mine), "gathered block differs from the local records"
        found = 0
        for r in range(world):
            a, b = parallel.shard_bounds(n_total, r, world)
            found += int(allrec[r, :b - a, :4].copy().view(np.int32).sum())
    else:
        leaves = leaves_local
        found = sum(o["pose_found"] for o in outs)
    if rank == 0:
        rl = loop_roofline(leaves_local, fine_ms, fine_n, loop_gathers(queries, LOOP_PARAMS), len(queries))
        print(json.dumps({
            "metric": METRIC, "value": leaves * args.steps / dt, "unit": "candidate poses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[%d]: branch-and-bound loop detection, %d candidate submaps, one "
                                   "1080-beam query scan per 256 of them "
                                   "(400x400@5cm, 3-level grids), 2.5 m x 2.5 m x 0.5 rad, "
                                   "thresholds 0.55/0.6; host-inclusive batch call per rank (scans in host "
                                   "memory, query array marshalled once, maps and pyramids resident)"
                                   % (3 if strong else 2, n_total),
                       "queries_total": n_total, "queries_per_rank": hi - lo,
                       "single_gpu_base": "configs.config4_one_gpu of the N = 1 line is this workload on one "
                                          "GPU (the N = 1 headline itself is configs[1], another workload)"
                       if strong else None,
                       "leaves_per_step": leaves, "found": found,
                       "parallelism": "contiguous query blocks per GPU, one all-gather of 48-B records "
                                      "per step (%s)" % ("gloo rehearsal" if rehearse else "RCCL")
                       if world > 1 else "single GPU",
                       "exchange": {"collective": "all_gather_into_tensor (RCCL)", "bytes_per_rank": block * rec_bytes,
                                    "all_gather_ms_per_step_rank0": (sum(gather_ms) / len(gather_ms)) if gather_ms else None,
                                    "all_gather_ms_max_rank0": max(gather_ms) if gather_ms else None,
                                    "note": "HIP events on the scoring stream around the collective, rank 0"}
                       if world > 1 else None},
            "roofline": rl,
        }), flush=True)
    ctx.close()


def measure_loop_config(dev_index, n_sub, steps=5):
    """configs[2] (n_sub = 256) or configs[3] at N = 1 (2048) as a short run:
    upload, pyramid build (timed on its own), then `steps` batch calls."""
    import torch
    from csm_hip import api
    ctx = api.Context(dev_index)
    t0 = time.perf_counter()
    queries, t_up = make_loop_queries(ctx, 0, n_sub)
    t_gen = time.perf_counter() - t0 - t_up
    prepared = ctx.prepare_queries(queries)
    H = LOOP_PARAMS[3]
    ctx.synchronize()
    ctx.enable_kernel_timing(True)
    ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    ctx.build_pyramids(list(range(n_sub)), [1 << h for h in range(H + 1)])
    ctx.synchronize()
    t_pyr = time.perf_counter() - t0
    box_ms, box_n = ctx.kernel_time("boxmax")
    ctx.enable_kernel_timing(False)
    t0 = time.perf_counter()
    outs = ctx.bnb_match_batch(prepared, *LOOP_PARAMS, as_records=True)      # first call
    t_first = time.perf_counter() - t0
    ctx.lib.csm_enable_kernel_timing(ctx._ctx, 2)
    ctx.reset_kernel_timing()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        outs = ctx.bnb_match_batch(prepared, *LOOP_PARAMS, as_records=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.enable_kernel_timing(False)
    fine_ms, fine_n = ctx.kernel_time("score_fine")
    bound_ms, bound_n = ctx.kernel_time("score_bound")
    if bound_n:
        fine_ms, fine_n = fine_ms + bound_ms, bound_n       # per batch call: bound pass + both exact rounds
    leaves = outs.total("candidates")
    found = sum(1 for o in outs if o["pose_found"])
    flagged = sum(1 for o in outs if o["raw"]["flags"])
    rl = loop_roofline(leaves, fine_ms, fine_n, loop_gathers(queries, LOOP_PARAMS), len(queries))
    ctx.close()
    return {"workload": "configs[%d]%s: one 1080-beam query scan per 256 of %d candidate submaps (one room family, own clutter / map offset / initial pose per submap), 3-level grids, 2.5 m x 2.5 m x "
                        "0.5 rad, thresholds 0.55/0.6" % (2 if n_sub == 256 else 3,
                                                          "" if n_sub == 256 else " on ONE GPU (strong-scaling base)",
                                                          n_sub),
            "value": leaves / dt, "unit": "candidate poses/s", "ms_per_step": dt * 1e3, "steps": steps,
            "leaves_per_step": leaves, "found": found, "flagged": flagged,
            "pyramid_build_ms": t_pyr * 1e3, "pyramid_build_kernel_ms": box_ms, "pyramid_launches": box_n,
            "upload_ms": t_up * 1e3, "host_generation_s": t_gen, "first_call_ms": t_first * 1e3,
            "end_to_end_value": leaves / (dt + t_pyr), "end_to_end_note":
                "one search step + the pyramid build of all %d maps (paid once per map in the "
                "reference too, loop_detector_branch_bound.cpp:83-89)" % n_sub,
            "roofline": rl}


def measure_config5(dev_index, runs=2, warmup=1):
    """configs[4]: global CSM, 2000x2000 @ 2.5 cm, +-10 m / +-180 deg at 2.5 cm / 0.25 deg, 1080
    beams, L = 4: 9.3e8 candidate poses in the window. Searched coarse-first: every
    coarse node scored, the fine level only on the candidate blocks that can still win; value =
    NOMINAL window poses / s (what the reference's pruned sweep is quoted in), with the poses
    actually evaluated beside it."""
    import torch
    from csm_hip import api, synth
    case = synth.csm_case(7, rows=2000, cols=2000, res=0.025, n_beams=N_BEAMS, fov=1.5 * math.pi,
                          max_range=5.7296, init_error=(3.1, -2.7, 1.3), n_boxes=10)
    ctx = api.Context(dev_index)
    t0 = time.perf_counter()
    ctx.upload_grid(5, case["grid"])
    t_up = time.perf_counter() - t0
    args = (5, case["geom"], case["angles"], case["ranges"], case["rel_pose"], case["init_pose"],
            20.0, 20.0, 2 * math.pi, 4, 0.0, 0.0)
    for _ in range(max(1, warmup)):
        out = ctx.correlative_match(*args)          # the first call builds box-max(4) and its phase-major copy
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(runs):
        out = ctx.correlative_match(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / runs
    info = ctx.last_search_info()
    # per-kernel times from a short extra pass (HIP events around every launch cost host time)
    ctx.lib.csm_enable_kernel_timing(ctx._ctx, 1)
    ctx.reset_kernel_timing()
    for _ in range(3):
        ctx.correlative_match(*args)
    ctx.enable_kernel_timing(False)
    kern = {}
    for name in ("project", "bin", "score_coarse", "select", "score_fine", "argmax", "finalize"):
        ms, n = ctx.kernel_time(name)
        if n:
            kern[name] = ms / 3.0 * 1e3          # us per query (a name may cover two launches)
    cands = out["candidates"]
    wt = out["win_theta"]
    col, row = api.host_project(case["geom"], out["sensor_pose"], out["step_theta"], wt, case["angles"],
                                case["ranges"])
    nx = ny = 804
    ent = cell_entries(col, row, 2000, 2000, -out["win_x"], -out["win_y"], -out["win_x"] + nx - 1,
                       -out["win_y"] + ny - 1)
    per_slice = ent / (2 * wt + 1)
    coarse_us = kern.get("score_coarse", 0.0)
    fine_us = kern.get("score_fine", 0.0)
    two_phase = info["two_phase"] == 1
    if two_phase and coarse_us >= fine_us:
        dom, dom_us = ("k_score_joint_one (exact joint kernel over pairs of slices) on the phase-major box-max(4) copy "
                       "(coarse pass: every coarse node)"), coarse_us
        lds = 4.0 * per_slice * info["coarse_nodes_scored"]
    else:
        dom, dom_us = "k_score_pairs%s (fine level)" % ("_list" if two_phase else ""), fine_us
        lds = 4.0 * per_slice * info["fine_candidates_scored"]
    ctx.close()
    ach = lds / (dom_us * 1e-6) / 1e9 if dom_us > 0 else None
    return {"workload": "configs[4]: global CSM, 2000x2000@2.5cm, +-10 m/+-180 deg at 2.5 cm/0.25 deg, "
                        "1080 beams, L=4; host-inclusive csm_correlative_match (projection on device), "
                        "searched coarse-first" + ("" if two_phase else " -- NOT taken: exhaustive"),
            "value": cands / dt, "unit": "candidate poses/s (nominal window)", "ms_per_query": dt * 1e3, "runs": runs,
            "candidates": cands, "found": out["pose_found"], "upload_ms": t_up * 1e3,
            "evaluated": {"coarse_nodes": info["coarse_nodes_scored"], "fine_candidates": info["fine_candidates_scored"],
                          "fine_blocks_scored": info["blocks_scored"], "fine_blocks_skipped": info["blocks_skipped"],
                          "evaluated_poses_per_s": (info["coarse_nodes_scored"] + info["fine_candidates_scored"]) / dt,
                          "note": "fully evaluated N-beam scores: every coarse node + every candidate of the fine "
                                  "blocks scored; the rest of the nominal window is excluded by the box-max bound "
                                  "exactly as scan_matcher_correlative.cpp:181-182 excludes it"},
            "kernel_us_per_query": kern,
            "roofline": {"bound": "lds", "kernel": dom, "unit": "GB/s", "peak": LDS_PEAK_GBS,
                         "achieved": ach, "frac": ach / LDS_PEAK_GBS if ach else None,
                         "avg_launch_us": dom_us,
                         "traffic": pmc_traffic(os.path.join(ROOT, "profiles", "r03_config5_pmc_traffic.json"))[0],
                         "traffic_source": pmc_traffic(os.path.join(ROOT, "profiles", "r03_config5_pmc_traffic.json"))[1],
                         "note": "LDS gather bytes (4 B x cell entries x candidates the kernel scores) / its time; "
                                 "HBM-side traffic of this workload: profiles/ (PMC passes)"}}


def measure_config2_latency(dev_index, wl, reps=30, warmup=3):
    """What one frontend caller sees: csm_correlative_match per scan, scan in host
    memory, projection on the device, summary back on the host."""
    from csm_hip import api
    rx, ry, rt, L = wl["params"]
    ctx = api.Context(dev_index)
    ctx.upload_grid(1, wl["grid"])
    samples = []
    cands = 0
    t_all = 0.0
    for i in range(reps + warmup):
        if i == warmup:
            t_all = time.perf_counter()
        sc = wl["scans"][i % len(wl["scans"])]
        t0 = time.perf_counter()
        out = ctx.correlative_match(1, wl["geom"], sc["angles"], sc["ranges"], sc["rel_pose"],
                                    sc["init_pose"], rx, ry, rt, L, 0.0, 0.0)
        samples.append(time.perf_counter() - t0)
        if i >= warmup:
            cands += out["candidates"]
    t_all = time.perf_counter() - t_all
    # per-kernel times from a short extra pass: HIP events around every launch cost host time and
    # switch the graph replay off, so they are not collected inside the timed loop
    ctx.lib.csm_enable_kernel_timing(ctx._ctx, 1)
    ctx.reset_kernel_timing()
    n_k = min(200, reps)
    for i in range(n_k):
        sc = wl["scans"][i % len(wl["scans"])]
        ctx.correlative_match(1, wl["geom"], sc["angles"], sc["ranges"], sc["rel_pose"], sc["init_pose"],
                              rx, ry, rt, L, 0.0, 0.0)
    ctx.enable_kernel_timing(False)
    kernels = {}
    for name in ("project", "bin", "score_coarse", "score_fine", "argmax", "finalize"):
        ms, n = ctx.kernel_time(name)
        if n:
            kernels[name] = ms / n_k * 1e3
    ctx.close()
    samples = sorted(samples[warmup:])
    med = samples[len(samples) // 2]
    return {"workload": "configs[1], one query per call: csm_correlative_match, host-inclusive "
                        "(17 KB up, projection + search on device, 48-B record back)",
            "latency_ms_median": med * 1e3, "latency_ms_min": samples[0] * 1e3,
            "latency_ms_p90": samples[int(0.9 * (len(samples) - 1))] * 1e3,
            "value": out["candidates"] / med, "unit": "candidate poses/s", "reps": reps,
            "total_s": t_all, "candidates_total": cands, "kernel_us_per_query": kernels,
            "note": "the timed loop replays each launch shape's chain as a HIP graph (from the third query of a "
                    "shape on); kernel_us_per_query comes from a separate pass with HIP events around every launch"}


# ------------------------------------------------------------------ map workload (not the BASELINE metric)

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


# ------------------------------------------------------------------ frontend CSM (configs[1])

def run_csm_workload(args, rank, world, dev, dev_index, rehearse, stream):
    import numpy as np
    import torch
    import torch.distributed as dist
    from csm_hip import api

    n_chunks = max(1, SCANS_PER_STEP // WINDOWS_PER_LAUNCH)
    n_distinct = max(WINDOWS_PER_LAUNCH, min(DISTINCT_SCANS, n_chunks * WINDOWS_PER_LAUNCH))
    n_distinct -= n_distinct % WINDOWS_PER_LAUNCH
    scans_per_step = n_chunks * WINDOWS_PER_LAUNCH
    wl = make_workload(rank, n_distinct)
    rx, ry, rt, L = wl["params"]
    ctx = api.Context(dev_index, tuning_off=args.tuning_off)
    # a non-default torch stream: csm_set_stream(NULL) would mean "the context's
    # own stream", and the collectives below must be ordered against the scoring
    ctx.set_stream(stream.cuda_stream)
    ctx.upload_grid(1, wl["grid"])
    ctx.build_pyramid(1, [1, L])

    windows, cols, rows_, cands = [], [], [], []
    for sc in wl["scans"]:
        wx, wy, wt = sc["win"]
        windows.append(ctx.make_window(2 * wt + 1, N_BEAMS, wx, wy, L, 1, api.host_min_known(N_BEAMS, 0.0), 0.0))
        cols.append(torch.from_numpy(sc["col"]).to(dev))
        rows_.append(torch.from_numpy(sc["row"]).to(dev))
        nx = -(-(2 * wx + 1) // L) * L
        ny = -(-(2 * wy + 1) // L) * L
        cands.append((2 * wt + 1) * nx * ny)
    rec_bytes = 48
    results = torch.zeros(scans_per_step * rec_bytes, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(world * scans_per_step * rec_bytes, dtype=torch.uint8, device=dev)
    n_batches = n_distinct // WINDOWS_PER_LAUNCH
    prepared = []
    for b in range(n_batches):
        sl = slice(b * WINDOWS_PER_LAUNCH, (b + 1) * WINDOWS_PER_LAUNCH)
        prepared.append(ctx.prepare_windows([1] * WINDOWS_PER_LAUNCH, windows[sl],
                                            [c.data_ptr() for c in cols[sl]],
                                            [r.data_ptr() for r in rows_[sl]]))
    cands_per_batch = [sum(cands[b * WINDOWS_PER_LAUNCH:(b + 1) * WINDOWS_PER_LAUNCH]) for b in range(n_batches)]
    cands_per_step = sum(cands_per_batch[k % n_batches] for k in range(n_chunks))

    def step():
        for k in range(n_chunks):
            ctx.score_windows_dev(prepared[k % n_batches],
                                  results.data_ptr() + k * WINDOWS_PER_LAUNCH * rec_bytes)
        if world > 1:
            if rehearse:
                host = results.cpu()
                out = torch.zeros(world * host.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, host)
                gathered.copy_(out)
            else:
                dist.all_gather_into_tensor(gathered, results)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        fence()
        # events around the dominant kernel only inside the timed region; the
        # other kernels are timed in a short extra pass afterwards
        ctx.lib.csm_enable_kernel_timing(ctx._ctx, 2)
        ctx.reset_kernel_timing()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        ctx.enable_kernel_timing(False)
        fine_ms, fine_n = ctx.kernel_time("score_fine")
        bound_ms, bound_n = ctx.kernel_time("score_bound")      # the packed-fp32 bound pass of the fine level
        fine_ms += bound_ms
        blocks_scored, blocks_skipped = ctx.bound_pass_stats()
        ctx.lib.csm_enable_kernel_timing(ctx._ctx, 1)
        ctx.reset_kernel_timing()
        step()
        fence()
        ctx.enable_kernel_timing(False)
    others = {}
    for name in ("score_coarse", "bin", "finalize"):
        ms, n = ctx.kernel_time(name)
        others[name] = ms / max(1, n) * 1e3

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


# ------------------------------------------------------------------ frontend CSM (configs[1])

def run_csm_workload(args, rank, world, dev, dev_index, rehearse, stream):
    import numpy as np
    import torch
    import torch.distributed as dist
    from csm_hip import api

    n_chunks = max(1, SCANS_PER_STEP // WINDOWS_PER_LAUNCH)
    n_distinct = max(WINDOWS_PER_LAUNCH, min(DISTINCT_SCANS, n_chunks * WINDOWS_PER_LAUNCH))
    n_distinct -= n_distinct % WINDOWS_PER_LAUNCH
    scans_per_step = n_chunks * WINDOWS_PER_LAUNCH
    wl = make_workload(rank, n_distinct)
    rx, ry, rt, L = wl["params"]
    ctx = api.Context(dev_index, tuning_off=args.tuning_off)
    # a non-default torch stream: csm_set_stream(NULL) would mean "the context's
    # own stream", and the collectives below must be ordered against the scoring
    ctx.set_stream(stream.cuda_stream)
    ctx.upload_grid(1, wl["grid"])
    ctx.build_pyramid(1, [1, L])

    windows, cols, rows_, cands = [], [], [], []
    for sc in wl["scans"]:
        wx, wy, wt = sc["win"]
        windows.append(ctx.make_window(2 * wt + 1, N_BEAMS, wx, wy, L, 1, api.host_min_known(N_BEAMS, 0.0), 0.0))
        cols.append(torch.from_numpy(sc["col"]).to(dev))
        rows_.append(torch.from_numpy(sc["row"]).to(dev))
        nx = -(-(2 * wx + 1) // L) * L
        ny = -(-(2 * wy + 1) // L) * L
        cands.append((2 * wt + 1) * nx * ny)
    rec_bytes = 48
    results = torch.zeros(scans_per_step * rec_bytes, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(world * scans_per_step * rec_bytes, dtype=torch.uint8, device=dev)
    n_batches = n_distinct // WINDOWS_PER_LAUNCH
    prepared = []
    for b in range(n_batches):
        sl = slice(b * WINDOWS_PER_LAUNCH, (b + 1) * WINDOWS_PER_LAUNCH)
        prepared.append(ctx.prepare_windows([1] * WINDOWS_PER_LAUNCH, windows[sl],
                                            [c.data_ptr() for c in cols[sl]],
                                            [r.data_ptr() for r in rows_[sl]]))
    cands_per_batch = [sum(cands[b * WINDOWS_PER_LAUNCH:(b + 1) * WINDOWS_PER_LAUNCH]) for b in range(n_batches)]
    cands_per_step = sum(cands_per_batch[k % n_batches] for k in range(n_chunks))

    def step():
        for k in range(n_chunks):
            ctx.score_windows_dev(prepared[k % n_batches],
                                  results.data_ptr() + k * WINDOWS_PER_LAUNCH * rec_bytes)
        if world > 1:
            if rehearse:
                host = results.cpu()
                out = torch.zeros(world * host.numel(), dtype=torch.uint8)
                dist.all_gather_into_tensor(out, host)
                gathered.copy_(out)
            else:
                dist.all_gather_into_tensor(gathered, results)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        fence()
        # events around the dominant kernel only inside the timed region; the
        # other kernels are timed in a short extra pass afterwards
        ctx.lib.csm_enable_kernel_timing(ctx._ctx, 2)
        ctx.reset_kernel_timing()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        ctx.enable_kernel_timing(False)
        fine_ms, fine_n = ctx.kernel_time("score_fine")
        bound_ms, bound_n = ctx.kernel_time("score_bound")      # the packed-fp32 bound pass of the fine level
        fine_ms += bound_ms
        blocks_scored, blocks_skipped = ctx.bound_pass_stats()
        ctx.lib.csm_enable_kernel_timing(ctx._ctx, 1)
        ctx.reset_kernel_timing()
        step()
        fence()
        ctx.enable_kernel_timing(False)
    others = {}
    for name in ("score_coarse", "bin", "finalize"):
        ms, n = ctx.kernel_time(name)
        others[name] = ms / max(1, n) * 1e3

    # the same workload through the exact integer kernel alone (no bound pass): reported beside the headline
    exact_only = None
    if world == 1 and not args.no_configs and not (args.tuning_off & 256) and blocks_skipped:
        ctx2 = api.Context(dev_index, tuning_off=args.tuning_off | 256)
        ctx2.set_stream(stream.cuda_stream)
        ctx2.upload_grid(1, wl["grid"])
        ctx2.build_pyramid(1, [1, L])
        prep2 = [ctx2.prepare_windows([1] * WINDOWS_PER_LAUNCH, windows[b * WINDOWS_PER_LAUNCH:(b + 1) * WINDOWS_PER_LAUNCH],
                                      [c.data_ptr() for c in cols[b * WINDOWS_PER_LAUNCH:(b + 1) * WINDOWS_PER_LAUNCH]],
                                      [r.data_ptr() for r in rows_[b * WINDOWS_PER_LAUNCH:(b + 1) * WINDOWS_PER_LAUNCH]])
                 for b in range(n_batches)]
        with torch.cuda.stream(stream):
            for rep in range(3):
                if rep == 1:
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                for k in range(n_chunks):
                    ctx2.score_windows_dev(prep2[k % n_batches], results.data_ptr() + k * WINDOWS_PER_LAUNCH * rec_bytes)
            torch.cuda.synchronize(dev)
        exact_only = cands_per_step * 2 / (time.perf_counter() - t0)
        ctx2.close()

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=None if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # this rank's slice of the gathered buffer must be its own records
        mine = gathered[rank * results.numel():(rank + 1) * results.numel()]
        assert torch.equal(mine, results), "all-gather read the records before they were written"

    # sanity: every scan found a pose and the record decodes
    rec_raw = results.cpu().numpy().tobytes()
    rec = np.frombuffer(rec_raw, dtype=np.int32).reshape(scans_per_step, 12)
    n_found = int(rec[:, 0].sum())
    verified = verify_sample(wl, rec_raw, scans_per_step, n_distinct, args.verify) if rank == 0 else 0


    if rank == 0:
        total = cands_per_step * args.steps * world
        value = total / dt
        cands_per_launch = cands_per_step / n_chunks
        alg_bytes = 2.0 * N_BEAMS * cands_per_launch
        avg_fine_s = (fine_ms / max(1, fine_n)) * 1e-3
        # LDS gathers one launch needs: cell entries x candidates, summed over its windows
        gathers = 0
        for i in range(WINDOWS_PER_LAUNCH):
            sc = wl["scans"][i]
            wx, wy, wt = sc["win"]
            nx = -(-(2 * wx + 1) // L) * L
            ny = -(-(2 * wy + 1) // L) * L
            gathers += cell_entries(sc["col"], sc["row"], 400, 400, -wx, -wy, -wx + nx - 1, -wy + ny - 1) * nx * ny
        lds_bytes = 4.0 * gathers
        achieved = lds_bytes / avg_fine_s / 1e9 if avg_fine_s > 0 else 0.0
        logical = alg_bytes / avg_fine_s / 1e9 if avg_fine_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic()
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "candidate poses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # uint16 cells; packed-fp32 bound pass over every candidate, exact u32 / u64 integer sums for the
            # candidates within its bound of the maximum; f64 replay of the winner
            "dtype": "f32+u32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: frontend CSM, 1080-beam scan, 400x400@5cm grid, "
                            "+-2 m/+-30 deg window at 5 cm/0.5 deg, L=4",
                "scans_per_step": scans_per_step,
                "distinct_scans": n_distinct,
                "mode": "csm_score_windows_dev, %d windows per launch chain, %d chains per step"
                        % (WINDOWS_PER_LAUNCH, n_chunks),
                "candidates_per_scan": cands_per_step / scans_per_step,
                "beams": N_BEAMS,
                "parallelism": "independent replicas per GPU, all-gather of 48-B best records" if world > 1
                               else "single GPU",
                "poses_found": n_found,
                "fine_level": ("packed-fp32 bound pass over every candidate (relative error < (beams + 3) 2^-24, "
                               "proven) + exact integer kernel on the %.2f %% of candidate blocks within that bound of "
                               "the window's maximum" % (100.0 * blocks_scored / max(1, blocks_scored + blocks_skipped)))
                              if blocks_skipped else "exact integer kernel on every candidate block",
                "bound_pass_us_per_launch": bound_ms / max(1, bound_n) * 1e3,
                "exact_integer_kernel_only_poses_per_s": exact_only,
                "verified": verified,
                "verified_note": "records of the last timed step compared, after the timed region, with the CPU "
                                 "oracle's literal sweep (best x, y, theta and the f64 score at tolerance 0)",
            },
            "roofline": {
                "bound": "lds",
                "kernel": ("fine level of %d windows: k_score_jointf_batch<156, 8> + <156, 6> (packed-fp32 bound pass: the "
                           "48-row blocks, then the 36-row blocks) + k_bound_select + k_score_joint_list<156, 8> / <156, 6> "
                           "(exact kernel on the blocks kept); avg_launch_us spans all of them"
                           if blocks_skipped else
                           "fine level of %d windows: the exact kernel's launches (48-row blocks, then the 36-row blocks); "
                           "avg_launch_us spans them") % WINDOWS_PER_LAUNCH,
                "achieved": achieved,
                "peak": LDS_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / LDS_PEAK_GBS,
                "note": "LDS gather bytes (4 B x cell entries x candidates) per launch / launch time vs "
                        "256 B/clk/CU x 256 CUs x 2.4 GHz; the grid window lives in LDS, HBM is not the bound",
                "lds_gathers_per_launch": gathers,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "hbm_frac_measured": (traffic / avg_fine_s / 1e9 / HBM_PEAK_GBS)
                                     if traffic is not None and avg_fine_s > 0 else None,
                "logical_hbm_gbs": logical,
                "logical_hbm_frac": logical / HBM_PEAK_GBS,
                "avg_launch_us": avg_fine_s * 1e6,
                "launches": fine_n,
                "algorithmic_bytes_per_launch": alg_bytes,
                "other_kernels_avg_us": others,
            },
        }
        ctx.close()
        if world == 1 and not args.no_configs:
            cfgs = {}
            wanted = os.environ.get("CSM_BENCH_CONFIGS",
                                    "config2_single_query,config3,config4_one_gpu,config5").split(",")
            for name, fn in (("config2_single_query", lambda: measure_config2_latency(dev_index, wl)),
                             ("config3", lambda: measure_loop_config(dev_index, 256)),
                             ("config4_one_gpu", lambda: measure_loop_config(dev_index, 2048, steps=3)),
                             ("config5", lambda: measure_config5(dev_index))):
                if name not in wanted:
                    continue
                try:
                    cfgs[name] = fn()
                except Exception as e:          # a side measurement must not lose the headline line
                    cfgs[name] = {"error": repr(e)}
            # inside "config": the driver's parser keeps that object (a top-level "configs" key was dropped)
            out["config"]["side_runs"] = cfgs
            out["config"]["side_runs_summary"] = {
                k: {kk: v[kk] for kk in ("value", "unit", "ms_per_step", "ms_per_query", "latency_ms_median") if kk in v}
                for k, v in cfgs.items() if isinstance(v, dict)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    else:
        ctx.close()


def run_side_workload(args, dev_index, which):
    """`--workload config5` / `--workload latency` as lines of their own (the same contract):
    a step is one query (one csm_correlative_match call, host-inclusive)."""
    if which == "config5":
        d = measure_config5(dev_index, runs=args.steps, warmup=max(1, args.warmup))
        total_s = d["ms_per_query"] * 1e-3 * args.steps
        ms_step = d["ms_per_query"]
        value = d["value"]
        roof = d["roofline"]
    else:
        wl = make_workload(0, 64)
        d = measure_config2_latency(dev_index, wl, reps=args.steps, warmup=max(3, args.warmup))
        total_s = d["total_s"]
        ms_step = d["total_s"] / args.steps * 1e3
        value = d["candidates_total"] / d["total_s"]
        fine_us = d["kernel_us_per_query"].get("score_fine", 0.0)
        sc = wl["scans"][0]
        wx, wy, wt = sc["win"]
        L = wl["params"][3]
        nx, ny = -(-(2 * wx + 1) // L) * L, -(-(2 * wy + 1) // L) * L
        gathers = cell_entries(sc["col"], sc["row"], 400, 400, -wx, -wy, -wx + nx - 1, -wy + ny - 1) * nx * ny
        ach = 4.0 * gathers / (fine_us * 1e-6) / 1e9 if fine_us else None
        roof = {"bound": "lds", "kernel": "k_score_pairs (fine level of one window, tile-split)", "unit": "GB/s",
                "peak": LDS_PEAK_GBS, "achieved": ach, "frac": ach / LDS_PEAK_GBS if ach else None,
                "avg_launch_us": fine_us, "traffic": None,
                "note": "a single 0.87-M-pose window is latency-bound: six launches and two copies around "
                        "~50 us of scoring; frac is the fine launch's share of the LDS read peak"}
    print(json.dumps({
        "metric": METRIC, "value": value, "unit": "candidate poses/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": d["workload"], "step": "one csm_correlative_match call (host-inclusive)",
                   "timed_s": total_s, "detail": {k: v for k, v in d.items() if k not in ("roofline", "workload")}},
        "roofline": roof}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="default: 20 (csm, loop workloads), 40 queries (config5), 12000 queries (latency)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tuning-off", type=int, default=0,
                    help="csm_config.tuning_off bits (CSM_TUNE_*: A/B runs of single launch optimisations)")
    ap.add_argument("--verify", type=int, default=24,
                    help="records of the last timed step compared with the CPU oracle after the timed region "
                         "(configs[1] workload; 0 = none)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short side runs of configs[2], [3] at N = 1, [4] and the single-query latency")
    ap.add_argument("--workload", choices=["auto", "csm", "loop", "loop4", "map", "config5", "latency"], default="auto",
                    help="auto (default): csm at N = 1, loop4 at N > 1. csm: BASELINE configs[1] per GPU "
                         "(replicas). loop: configs[2], 256 candidate submaps per GPU. loop4: configs[3], 2048 "
                         "submaps sharded over the ranks (strong scaling) + all-gather. map: the frontend "
                         "cycle (latest-map build from 10 scans + match), not the BASELINE metric")
    args = ap.parse_args()

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

    workload = args.workload
    if workload == "auto":
        workload = "csm" if world == 1 else "loop4"
    if args.steps is None:
        args.steps = {"config5": 40, "latency": 12000}.get(workload, 20)
    stream = torch.cuda.Stream(dev)
    if workload in ("config5", "latency"):
        if world != 1:
            raise SystemExit("--workload %s is a single-GPU line" % workload)
        run_side_workload(args, dev_index, workload)
    elif workload == "map":
        run_map_workload(args, rank, world, dev, dev_index, rehearse)
    elif workload in ("loop", "loop4"):
        n_total = 2048 if workload == "loop4" else 256 * world
        if rehearse and "CSM_BENCH_LOOP_TOTAL" in os.environ:
            n_total = int(os.environ["CSM_BENCH_LOOP_TOTAL"])      # smaller rehearsal batches
        run_loop_workload(args, rank, world, dev, dev_index, rehearse, stream, n_total, workload == "loop4")
    else:
        run_csm_workload(args, rank, world, dev, dev_index, rehearse, stream)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
