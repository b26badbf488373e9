"""GPU parity of the launch shape bench.py's headline number rests on: 256
full-size BASELINE configs[1] windows (1080 beams over 270 deg, 400 x 400 grid @ 5 cm,
+-2 m / +-30 deg, L = 4: 121-123 slices x 84 x 84 candidates each) through
csm_score_windows_dev in ONE launch chain -- the batched fine kernels with the lane
table, the XCD-aware block order, the R = 6 tail launch and the joint two-slice
entry lists, each also switched off (csm_config.tuning_off).

Every record (found, best x / y / theta, f64 score at tolerance 0) is compared with
the oracle's LITERAL sequential sweep (orc_csm: scan_matcher_correlative.cpp:161-197,
339-368 restated), and the full per-candidate integer sums S, K of four windows with
the oracle's closed form. The oracle itself is "parity unpinned" for the sweep (the
reference holds no vectors)."""
import concurrent.futures
import math
import os

import numpy as np
import pytest
import torch

import bench
from csm_hip import _lib as L, api

pytestmark = pytest.mark.gpu

N_WIN = 256
N_DUMP = 4


@pytest.fixture(scope="module")
def headline(oracle):
    wl = bench.make_workload(0, N_WIN)
    rx, ry, rt, Lr = wl["params"]
    coarse = oracle.boxmax(wl["grid"], Lr)

    def case_of(sc):
        return dict(grid=wl["grid"], geom=wl["geom"], angles=sc["angles"], ranges=sc["ranges"],
                    rel_pose=sc["rel_pose"], init_pose=sc["init_pose"])

    # the literal sweep of every window; ctypes releases the GIL, so host threads help
    workers = max(1, min(16, (os.cpu_count() or 2) // 2))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        lits = list(pool.map(lambda sc: oracle.csm(case_of(sc), rx, ry, rt, Lr, coarse=coarse), wl["scans"]))
    dumps = [oracle.csm_closed_form(case_of(sc), rx, ry, rt, Lr, coarse=coarse, dump=True)
             for sc in wl["scans"][:N_DUMP]]
    return wl, lits, dumps


VARIANTS = [("default", 0), ("no_bound_pass", L.TUNE_NO_BOUND_PASS), ("no_pair_tail", L.TUNE_NO_PAIR_TAIL),
            ("no_lane_map", L.TUNE_NO_LANE_MAP), ("no_xcd_map", L.TUNE_NO_XCD_MAP), ("no_joint", L.TUNE_NO_JOINT),
            ("one_slice", L.TUNE_NO_TWO_SLICES | L.TUNE_NO_JOINT),
            ("all_off", L.TUNE_NO_PAIR_TAIL | L.TUNE_NO_LANE_MAP | L.TUNE_NO_XCD_MAP | L.TUNE_NO_JOINT)]


@pytest.mark.parametrize("name,tuning_off", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_headline_batch_every_record_and_dumps(headline, name, tuning_off):
    wl, lits, dumps = headline
    rx, ry, rt, Lr = wl["params"]
    dev = torch.device("cuda", 0)
    ctx = api.Context(0, tuning_off=tuning_off)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.upload_grid(1, wl["grid"])
    ctx.build_pyramid(1, [1, Lr])
    windows, cols, rows, keep = [], [], [], []
    for sc in wl["scans"]:
        wx, wy, wt = sc["win"]
        windows.append(ctx.make_window(2 * wt + 1, bench.N_BEAMS, wx, wy, Lr, 1,
                                       api.host_min_known(bench.N_BEAMS, 0.0), 0.0))
        c_d, r_d = torch.from_numpy(sc["col"]).to(dev), torch.from_numpy(sc["row"]).to(dev)
        keep += [c_d, r_d]
        cols.append(c_d.data_ptr())
        rows.append(r_d.data_ptr())
    out = torch.zeros(N_WIN * 48, dtype=torch.uint8, device=dev)
    ds, dk = [0] * N_WIN, [0] * N_WIN
    for k in range(N_DUMP):
        S, K = dumps[k][1], dumps[k][2]
        keep += [torch.zeros(S.size, dtype=torch.int32, device=dev), torch.zeros(K.size, dtype=torch.int16, device=dev)]
        ds[k], dk[k] = keep[-2].data_ptr(), keep[-1].data_ptr()
    prepared = ctx.prepare_windows([1] * N_WIN, windows, cols, rows)
    # fp32 keys of the bound pass for the windows after the integer dumps (a window with integer
    # dumps is scored exactly everywhere; these go through the skipping)
    df = [0] * N_WIN
    fkeep = []
    for k in range(N_DUMP, 2 * N_DUMP):
        wx, wy, wt = wl["scans"][k]["win"]
        nx, ny = -(-(2 * wx + 1) // Lr) * Lr, -(-(2 * wy + 1) // Lr) * Lr
        fkeep.append(torch.zeros((2 * wt + 1) * nx * ny, dtype=torch.float32, device=dev))
        df[k] = fkeep[-1].data_ptr()
    ctx.bound_pass_stats()
    ctx.score_windows_dump_dev(prepared, out.data_ptr(), ds, dk, df)
    torch.cuda.synchronize(dev)
    scored, skipped = ctx.bound_pass_stats()
    bound_pass = not (tuning_off & (L.TUNE_NO_BOUND_PASS | L.TUNE_NO_JOINT | L.TUNE_NO_TWO_SLICES))
    if bound_pass:
        # the bound pass must leave the exact kernel a small share of the blocks
        assert skipped > 20 * scored > 0, (scored, skipped)
    else:
        assert scored == skipped == 0
    rec = out.cpu().numpy().reshape(N_WIN, 48)
    n_flagged = 0
    for k in range(N_WIN):
        r = L.Result.from_buffer_copy(rec[k].tobytes())
        lit = lits[k]
        if r.flags & (L.FLAG_EDGE_BAND | L.FLAG_KEY_TIE):
            # finished by the exact single-window paths, as the header says
            n_flagged += 1
            sc = wl["scans"][k]
            d = ctx.score_window(1, windows[k], sc["col"], sc["row"])
            got = (d["found"], d["best_x"], d["best_y"], d["best_theta"], d["score"])
        else:
            got = (r.found, r.best_x, r.best_y, r.best_theta, r.score)
        assert got[0] == lit["found"] == 1, (name, k)
        assert got[1:4] == (lit["bestX"], lit["bestY"], lit["bestT"]), (name, k)
        assert got[4] == lit["scoreMax"], (name, k)          # f64, tolerance 0
    assert n_flagged <= N_WIN // 8
    for k in range(N_DUMP):
        want, oS, oK, _ = dumps[k]
        S = keep[2 * N_WIN + 2 * k].cpu().numpy().view(np.uint32).reshape(oS.shape)
        K = keep[2 * N_WIN + 2 * k + 1].cpu().numpy().view(np.uint16).reshape(oK.shape)
        assert np.array_equal(S, oS), (name, k)
        assert np.array_equal(K, oK), (name, k)
    if bound_pass:
        # the bound itself: |fp32 key - exact key| <= (n + 3) 2^-24 * key for every candidate
        from oracle import oracle as O
        for j, k in enumerate(range(N_DUMP, 2 * N_DUMP)):
            sc = wl["scans"][k]
            case = dict(grid=wl["grid"], geom=wl["geom"], angles=sc["angles"], ranges=sc["ranges"],
                        rel_pose=sc["rel_pose"], init_pose=sc["init_pose"])
            _, oS, oK, _ = O.csm_closed_form(case, rx, ry, rt, Lr, dump=True)
            key = 32268.0 * oK.astype(np.float64) + 499.0 * oS.astype(np.float64)
            got = fkeep[j].cpu().numpy().astype(np.float64).reshape(key.shape)
            err = np.abs(got - key)
            assert np.all(err <= (bench.N_BEAMS + 3) * 2.0 ** -24 * key + 1e-9), (name, k, float(err.max()))
            assert float(err.max()) > 0.0                     # it IS an approximation
    ctx.close()
