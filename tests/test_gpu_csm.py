"""GPU parity: the HIP correlative path, through the C ABI, against the CPU
oracle on the same seeded inputs. Bar: integer sums and best-pose indices
bit-exact; the f64 score bit-exact too (tolerance 0: it is replayed in beam
order in f64 on the device)."""
import math

import numpy as np
import pytest

from csm_hip import _lib as Lb, api, synth

pytestmark = pytest.mark.gpu


def _window_for(case, rx, ry, rt, L, score_thr=0.0, known_thr=0.0):
    res = case["geom"][0]
    sx, sy, st = api.host_search_step(res, case["ranges"])
    wx, wy, wt = api.host_window(rx, sx), api.host_window(ry, sy), api.host_window(rt, st)
    sensor = api.host_compound(case["init_pose"], case["rel_pose"])
    col, row = api.host_project(case["geom"], sensor, st, wt, case["angles"], case["ranges"])
    n = len(case["angles"])
    return (wx, wy, wt), col, row, api.host_min_known(n, known_thr)


@pytest.mark.parametrize("win", [1, 2, 3, 4, 5, 8, 16, 64])
def test_boxmax_matches_oracle(gpu_ctx, oracle, win):
    grid, _, _ = synth.make_room(3, rows=208, cols=176)
    gpu_ctx.upload_grid(100, grid)
    gpu_ctx.build_pyramid(100, [1, win])
    got = gpu_ctx.download_level(100, 1)
    assert np.array_equal(gpu_ctx.download_level(100, 0), grid)
    assert np.array_equal(got, oracle.boxmax(grid, win))
    gpu_ctx.release_grid(100)


@pytest.mark.parametrize("seed,L", [(0, 4), (1, 4), (2, 1), (3, 5), (4, 3), (5, 8)])
def test_config1_all_candidates_and_winner(gpu_ctx, oracle, seed, L):
    case = synth.csm_case(seed)
    rx, ry, rt = 1.0, 1.0, math.radians(10)
    (wx, wy, wt), col, row, mk = _window_for(case, rx, ry, rt, L)
    gpu_ctx.upload_grid(1, case["grid"])
    gpu_ctx.build_pyramid(1, [1, L])
    w = gpu_ctx.make_window(2 * wt + 1, len(case["angles"]), wx, wy, L, 1, mk, 0.0)
    res, S, K, CK = gpu_ctx.score_window(1, w, col, row, dump=True)
    want, oS, oK, oCK = oracle.csm_closed_form(case, rx, ry, rt, L, dump=True)
    lit = oracle.csm(case, rx, ry, rt, L)
    assert np.array_equal(S, oS)
    assert np.array_equal(K, oK)
    if L > 1:
        assert np.array_equal(CK, oCK)
    assert (res["best_x"], res["best_y"], res["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"])
    assert res["found"] == lit["found"]
    assert res["score"] == lit["scoreMax"]          # bit-exact f64
    assert want["scoreMax"] == lit["scoreMax"]
    gpu_ctx.release_grid(1)


def test_config1_match_summary(gpu_ctx, oracle):
    case = synth.csm_case(11, rel_pose=(0.12, -0.03, 0.05))
    m = api.ScanMatcherCorrelativeHIP("csm", 4, 1.0, 1.0, math.radians(10), ctx=gpu_ctx)
    out = m.optimize_pose(case["grid"], case["geom"], case["angles"], case["ranges"],
                          case["rel_pose"], case["init_pose"])
    lit = oracle.csm(case, 1.0, 1.0, math.radians(10), 4)
    assert out["pose_found"] == lit["found"] == 1
    assert (out["win_x"], out["win_y"], out["win_theta"]) == (lit["winX"], lit["winY"], lit["winT"])
    assert out["estimated_pose"] == lit["estimatedPose"]      # bit-exact doubles
    assert out["raw"]["score"] == lit["scoreMax"]


@pytest.mark.parametrize("merge_mode", [0, 1])
def test_beam_merging_on_and_off_give_identical_sums(gpu_ctx, oracle, merge_mode):
    """k_bin either merges beams that share a cell into weighted entries or
    keeps one entry per beam (csm_window.merge_mode); both must reproduce every
    candidate's integer sums. Short ranges put many beams on one cell
    (multiplicities above kMaxMult = 15 are split)."""
    case = synth.csm_case(13, n_beams=4000, max_range=1.2)
    rx, ry, rt, L = 0.8, 0.8, math.radians(8), 4
    (wx, wy, wt), col, row, mk = _window_for(case, rx, ry, rt, L)
    dup = max(np.unique(np.stack([col[wt], row[wt]]), axis=1, return_counts=True)[1])
    assert dup > 15
    gpu_ctx.upload_grid(2, case["grid"])
    gpu_ctx.build_pyramid(2, [1, L])
    w = gpu_ctx.make_window(2 * wt + 1, len(case["angles"]), wx, wy, L, 1, mk, 0.0, merge_mode)
    res, S, K, CK = gpu_ctx.score_window(2, w, col, row, dump=True)
    want, oS, oK, oCK = oracle.csm_closed_form(case, rx, ry, rt, L, dump=True)
    assert np.array_equal(S, oS) and np.array_equal(K, oK) and np.array_equal(CK, oCK)
    lit = oracle.csm(case, rx, ry, rt, L)
    assert (res["best_x"], res["best_y"], res["best_theta"]) == (lit["bestX"], lit["bestY"], lit["bestT"])
    assert res["score"] == lit["scoreMax"]
    gpu_ctx.release_grid(2)


@pytest.mark.parametrize("n_beams,max_range,merge_mode", [(4000, 1.2, 0), (4000, 1.2, 1), (1080, 5.7, 0),
                                                           (2000, 0.3, 0)])
def test_saturated_grid_never_overflows_the_packed_accumulators(gpu_ctx, oracle, n_beams, max_range, merge_mode):
    """The fine kernel adds value and known count of a gather in one 32-bit
    multiply-add (23 + 9 bits) and empties those accumulators by a rule over the
    entry list (flags per 64-entry chunk: pairs_gather). Every cell at 65535 and
    many beams per cell (entries of up to 30 beams) is the worst case for it:
    any stretch of more than 128 beams between two flushes corrupts K."""
    case = synth.csm_case(17, n_beams=n_beams, max_range=max_range)
    case["grid"] = np.full_like(case["grid"], 65535)
    rx, ry, rt, L = 0.8, 0.8, math.radians(6), 4
    (wx, wy, wt), col, row, mk = _window_for(case, rx, ry, rt, L)
    gpu_ctx.upload_grid(3, case["grid"])
    gpu_ctx.build_pyramid(3, [1, L])
    w = gpu_ctx.make_window(2 * wt + 1, len(case["angles"]), wx, wy, L, 1, mk, 0.0, merge_mode)
    res, S, K, CK = gpu_ctx.score_window(3, w, col, row, dump=True)
    want, oS, oK, oCK = oracle.csm_closed_form(case, rx, ry, rt, L, dump=True)
    assert np.array_equal(K, oK) and np.array_equal(S, oS)
    assert int(K.max()) == n_beams and int(S.max()) == 65535 * n_beams
    gpu_ctx.release_grid(3)


def test_lane_table_on_and_off_give_identical_sums(gpu_ctx, oracle):
    """The pair kernels take the thread -> (lane group, column) assignment from a host
    table that keeps half-waves free of LDS bank conflicts (lane_map_for, csm_api.hip);
    CSM_TUNE_NO_LANE_MAP numbers the threads through the groups in order. Both must give
    every candidate's sums (84 x 84 window: 6 groups of 84 columns, the shape the table is for)."""
    case = synth.csm_case(21, n_beams=1080, fov=1.5 * math.pi)
    rx, ry, rt, L = 2.0, 2.0, math.radians(4), 4
    (wx, wy, wt), col, row, mk = _window_for(case, rx, ry, rt, L)
    want, oS, oK, oCK = oracle.csm_closed_form(case, rx, ry, rt, L, dump=True)
    plain = api.Context(0, tuning_off=Lb.TUNE_NO_LANE_MAP)
    for ctx in (gpu_ctx, plain):
        ctx.upload_grid(4, case["grid"])
        ctx.build_pyramid(4, [1, L])
        w = ctx.make_window(2 * wt + 1, len(case["angles"]), wx, wy, L, 1, mk, 0.0)
        res, S, K, CK = ctx.score_window(4, w, col, row, dump=True)
        assert np.array_equal(S, oS) and np.array_equal(K, oK)
        assert (res["best_x"], res["best_y"], res["best_theta"]) == (want["bestX"], want["bestY"], want["bestT"])
        ctx.release_grid(4)
    plain.close()


def _carry_case(n_theta):
    """Hit indices (the same for every slice) that put, on an all-65535 grid and an 84 x 84
    window, exactly the entry sequence that once lost a flush: the first endpoint tile holds
    36 one-beam even-row entries and then four 15-beam odd-row entries (the heavy group is
    flushed at its start, leaves 60 beams in the packed accumulators and the running count at
    a multiple of 96), the next tile 96 one-beam entries whose first bucket boundary is entry
    95. Without the carry of "heavy entry among the last four" across the tile boundary 152
    beams pile up before the next flush and the value sum (23 bits) runs into the count."""
    win, n_cand = 40, 84                      # nx = ny = 84 for L = 4
    x_hi = y_hi = -win + n_cand - 1           # 43; k_bin's frame: rr = r + y_hi + 1, cc = c + x_hi
    cells = []                                # (frame row, frame col, beams)
    cells += [(140, 130 + i, 1) for i in range(36)]
    cells += [(143, 130 + i, 15) for i in range(4)]
    cells += [(200, 128 + i, 1) for i in range(64)] + [(202, 128 + i, 1) for i in range(32)]
    row = np.concatenate([[rr - y_hi - 1] * m for rr, cc, m in cells]).astype(np.int32)
    col = np.concatenate([[cc - x_hi] * m for rr, cc, m in cells]).astype(np.int32)
    return win, np.tile(col, (n_theta, 1)), np.tile(row, (n_theta, 1))


def test_flush_carry_across_tile_boundary_single_window(gpu_ctx):
    """pairs_gather's carry across lists. Single-window kernel, not
    tile-split (2 x 193 workgroups >= 384), full S / K dump: every candidate sees all 192
    beams on saturated cells."""
    n_theta = 193
    win, col, row = _carry_case(n_theta)
    n = col.shape[1]
    assert n == 36 + 60 + 96
    grid = np.full((400, 400), 65535, np.uint16)
    gpu_ctx.upload_grid(5, grid)
    gpu_ctx.build_pyramid(5, [1, 4])
    w = gpu_ctx.make_window(n_theta, n, win, win, 4, 1, 0, 0.0)
    res, S, K, CK = gpu_ctx.score_window(5, w, col, row, dump=True)
    assert int(K.min()) == int(K.max()) == n
    assert int(S.min()) == int(S.max()) == 65535 * n
    gpu_ctx.release_grid(5)


def test_flush_carry_across_tile_boundary_batch(gpu_ctx):
    """The same sequence through the batched two-slice kernel (csm_score_windows_dump_dev)."""
    import torch
    dev = torch.device("cuda", 0)
    n_theta = 9
    win, col, row = _carry_case(n_theta)
    n = col.shape[1]
    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.upload_grid(6, np.full((400, 400), 65535, np.uint16))
    ctx.build_pyramid(6, [1, 4])
    nw = 3
    w = ctx.make_window(n_theta, n, win, win, 4, 1, 0, 0.0)
    c_d, r_d = torch.from_numpy(col).to(dev), torch.from_numpy(row).to(dev)
    out = torch.zeros(nw * 48, dtype=torch.uint8, device=dev)
    ds = [torch.zeros(n_theta * 84 * 84, dtype=torch.int32, device=dev) for _ in range(nw)]
    dk = [torch.zeros(n_theta * 84 * 84, dtype=torch.int16, device=dev) for _ in range(nw)]
    prepared = ctx.prepare_windows([6] * nw, [w] * nw, [c_d.data_ptr()] * nw, [r_d.data_ptr()] * nw)
    ctx.score_windows_dump_dev(prepared, out.data_ptr(), [t.data_ptr() for t in ds], [t.data_ptr() for t in dk])
    torch.cuda.synchronize(dev)
    for k in range(nw):
        S = ds[k].cpu().numpy().view(np.uint32)
        K = dk[k].cpu().numpy().view(np.uint16)
        assert int(K.min()) == int(K.max()) == n, k
        assert int(S.min()) == int(S.max()) == 65535 * n, k
    ctx.close()
