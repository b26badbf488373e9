"""Shapes that stress the joint binning kernel (k_binj, csm_joint_kernels.hip): its hash table of
any size, the lanes behind the last beam, more beams than one pass of five blocks, one-tile-wide
frames, scans that pile hundreds of beams on a few cells (beam counts above 15: several entries per
slot), windows at the map's low edge (edge band). Every case goes through csm_score_windows_dev in
a batch and is compared with the oracle: the winner's record with the literal sweep
(scan_matcher_correlative.cpp:161-197, 339-368 restated; "parity unpinned"),
the full per-candidate integer sums S, K of the first window with the closed form."""
import math

import numpy as np
import pytest
import torch

from csm_hip import _lib as L, api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    dev = torch.device("cuda", 0)
    c = api.Context(0)
    c.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    yield c
    c.close()


def _batch(ctx, oracle, cases, rx, ry, rt, Lr, dump_first=True):
    """cases share a scan length; one window per case, all in one launch chain"""
    dev = torch.device("cuda", 0)
    n = len(cases[0]["angles"])
    keep, windows, cols, rows, ids, hits = [], [], [], [], [], []
    for i, case in enumerate(cases):
        assert len(case["angles"]) == n
        sx, sy, st = api.host_search_step(case["geom"][0], case["ranges"])
        wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
        col, row = api.host_project(case["geom"], case["init_pose"], st, wt, case["angles"], case["ranges"])
        ctx.upload_grid(700 + i, case["grid"])
        ctx.build_pyramid(700 + i, [1, Lr])
        windows.append(ctx.make_window(2 * wt + 1, n, wx, wy, Lr, 1, api.host_min_known(n, 0.0), 0.0))
        c_d, r_d = torch.from_numpy(col).to(dev), torch.from_numpy(row).to(dev)
        keep += [c_d, r_d]
        cols.append(c_d.data_ptr())
        rows.append(r_d.data_ptr())
        ids.append(700 + i)
        hits.append((col, row))
    out = torch.zeros(len(cases) * 48, dtype=torch.uint8, device=dev)
    prepared = ctx.prepare_windows(ids, windows, cols, rows)
    want0 = None
    if dump_first:
        want0 = oracle.csm_closed_form(cases[0], rx, ry, rt, Lr, dump=True)
        S = torch.zeros(want0[1].size, dtype=torch.int32, device=dev)
        K = torch.zeros(want0[2].size, dtype=torch.int16, device=dev)
        ds, dk = [0] * len(cases), [0] * len(cases)
        ds[0], dk[0] = S.data_ptr(), K.data_ptr()
        ctx.score_windows_dump_dev(prepared, out.data_ptr(), ds, dk, [0] * len(cases))
    else:
        ctx.score_windows_dev(prepared, out.data_ptr())
    torch.cuda.synchronize(dev)
    rec = out.cpu().numpy().reshape(len(cases), 48)
    for i, case in enumerate(cases):
        r = L.Result.from_buffer_copy(rec[i].tobytes())
        if r.flags & (L.FLAG_EDGE_BAND | L.FLAG_KEY_TIE):      # finished by the exact single-window paths
            d = ctx.score_window(ids[i], windows[i], hits[i][0], hits[i][1])
            got = (d["found"], d["best_x"], d["best_y"], d["best_theta"], d["score"])
        else:
            got = (r.found, r.best_x, r.best_y, r.best_theta, r.score)
        lit = oracle.csm(case, rx, ry, rt, Lr)
        assert got[0] == lit["found"], i
        if lit["found"]:
            assert got[1:4] == (lit["bestX"], lit["bestY"], lit["bestT"]), i
            assert got[4] == lit["scoreMax"], i              # f64, tolerance 0
    if dump_first:
        _, oS, oK, _ = want0
        assert np.array_equal(S.cpu().numpy().view(np.uint32).reshape(oS.shape), oS)
        assert np.array_equal(K.cpu().numpy().view(np.uint16).reshape(oK.shape), oK)
    for i in ids:
        ctx.release_grid(i)


# beams per scan around the kernel's block edges: 512 lanes x 5 blocks per pass over the two slices
@pytest.mark.parametrize("n_beams", [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1279, 1280, 1281, 1300, 2047, 2100])
def test_beam_counts_around_block_edges(ctx, oracle, n_beams):
    cases = [synth.csm_case(300 + n_beams + k, n_beams=n_beams, fov=1.5 * math.pi, max_range=6.0) for k in range(2)]
    _batch(ctx, oracle, cases, 0.4, 0.35, math.radians(3), 4)


@pytest.mark.parametrize("rows,cols,Lr,n_beams", [(40, 40, 2, 90), (48, 200, 4, 180), (200, 48, 3, 180),
                                                  (640, 96, 4, 400), (96, 1000, 5, 400), (1200, 1200, 4, 720)])
def test_frame_shapes(ctx, oracle, rows, cols, Lr, n_beams):
    """one tile wide, one tile high, long and thin, many tiles (the key's column field widens)"""
    res = 0.05
    cases = [synth.csm_case(900 + rows + k, rows=rows, cols=cols, res=res, n_beams=n_beams,
                            max_range=0.45 * min(rows, cols) * res, init_error=(0.06, -0.04, 0.01)) for k in range(2)]
    _batch(ctx, oracle, cases, 0.3, 0.3, math.radians(2), Lr)


def _piled_scan(case, n_beams, spread):
    """beams in a fan of `spread` rad, all with nearly one range: hundreds of beams on a handful of cells"""
    rng = np.random.RandomState(n_beams)
    angles = np.sort(rng.uniform(-0.5 * spread, 0.5 * spread, n_beams))
    base = float(np.median(case["ranges"][np.isfinite(case["ranges"])]))
    ranges = base + rng.uniform(-0.02, 0.02, n_beams)
    out = dict(case)
    out["angles"], out["ranges"] = angles.astype(np.float64), ranges.astype(np.float64)
    return out


@pytest.mark.parametrize("n_beams,spread", [(300, 0.02), (1080, 0.05), (1080, 0.004), (2000, 0.2)])
def test_beams_piled_on_few_cells(ctx, oracle, n_beams, spread):
    """beam counts far above 15 per cell: a slot becomes ceil(count / 15) entries; with spread 0.004
    every beam of a slice lands on one or two cells"""
    cases = [_piled_scan(synth.csm_case(1200 + k, n_beams=64), n_beams, spread) for k in range(2)]
    _batch(ctx, oracle, cases, 0.25, 0.25, math.radians(2), 4)


@pytest.mark.parametrize("seed", [50, 51, 52, 53])
def test_windows_at_the_low_edge(ctx, oracle, seed):
    """scans whose windows reach below row / column 0 of the map (the cases of
    tests/test_gpu_edge.py::test_csm_negative_edge_band, here through the batch entry): the edge-band
    flag comes from k_binj's second look at the beams, the record from the exact
    single-window path"""
    cases = [synth.csm_case(seed + 10 * k, rows=256, cols=288, origin="low_edge", half_x=5.2, half_y=4.4,
                            init_error=(0.23, 0.19, 0.03)) for k in range(3)]
    _batch(ctx, oracle, cases, 1.0, 1.0, math.radians(10), 4, dump_first=False)
