"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/_build/libcsm_oracle.so
(the CPU restatement in oracle/csm_oracle.cpp) and, where it was built, of
oracle/_ref/libref_geom.so (the reference's own header-only geometry).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module. The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libcsm_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_geom.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


class CsmParams(C.Structure):
    _fields_ = [("rangeX", C.c_double), ("rangeY", C.c_double), ("rangeT", C.c_double),
                ("lowRes", C.c_int), ("scoreThr", C.c_double), ("knownThr", C.c_double)]


class BnbParams(C.Structure):
    _fields_ = [("rangeX", C.c_double), ("rangeY", C.c_double), ("rangeT", C.c_double),
                ("nodeHeightMax", C.c_int), ("scoreThr", C.c_double), ("knownThr", C.c_double)]


class GridParams(C.Structure):
    _fields_ = [("rangeX", C.c_double), ("rangeY", C.c_double), ("rangeT", C.c_double),
                ("stepX", C.c_double), ("stepY", C.c_double), ("stepT", C.c_double),
                ("scoreThr", C.c_double), ("knownThr", C.c_double)]


class Result(C.Structure):
    _fields_ = [("found", C.c_int), ("bestX", C.c_int), ("bestY", C.c_int), ("bestT", C.c_int),
                ("winX", C.c_int), ("winY", C.c_int), ("winT", C.c_int),
                ("stepX", C.c_double), ("stepY", C.c_double), ("stepT", C.c_double),
                ("scoreMax", C.c_double), ("sensorPose", C.c_double * 3),
                ("bestSensorPose", C.c_double * 3), ("estimatedPose", C.c_double * 3),
                ("ignoredNodes", C.c_longlong), ("processedNodes", C.c_longlong),
                ("fineEvaluated", C.c_longlong)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        _lib = C.CDLL(ORACLE_SO)
        _lib.orc_value_to_probability.restype = C.c_double
        _lib.orc_value_to_probability.argtypes = [C.c_uint]
        _lib.orc_score_at.restype = C.c_double
        for name in ("orc_bb_probability_to_odds", "orc_bb_odds_to_probability"):
            getattr(_lib, name).restype = C.c_double
            getattr(_lib, name).argtypes = [C.c_double]
        _lib.orc_bb_probability_to_value.restype = C.c_uint
        _lib.orc_bb_probability_to_value.argtypes = [C.c_double]
        _lib.orc_bb_value_to_odds.restype = C.c_double
        _lib.orc_bb_value_to_odds.argtypes = [C.c_uint]
    return _lib


def ref():
    """The reference-built geometry library, or None when absent (GPU box)."""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        _ref = C.CDLL(REF_SO)
        _ref.ref_value_to_probability.restype = C.c_double
        _ref.ref_value_to_probability.argtypes = [C.c_uint]
        if hasattr(_ref, "ref_value_to_odds"):
            for name in ("ref_probability_to_odds", "ref_odds_to_probability"):
                getattr(_ref, name).restype = C.c_double
                getattr(_ref, name).argtypes = [C.c_double]
            _ref.ref_probability_to_value.restype = C.c_uint
            _ref.ref_probability_to_value.argtypes = [C.c_double]
            _ref.ref_value_to_odds.restype = C.c_double
            _ref.ref_value_to_odds.argtypes = [C.c_uint]
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def compound(a, b):
    out = np.zeros(3)
    lib().orc_compound(_p(_f64(a)), _p(_f64(b)), _p(out))
    return out


def inverse_compound(a, b):
    out = np.zeros(3)
    lib().orc_inverse_compound(_p(_f64(a)), _p(_f64(b)), _p(out))
    return out


def move_backward(a, b):
    out = np.zeros(3)
    lib().orc_move_backward(_p(_f64(a)), _p(_f64(b)), _p(out))
    return out


def lut():
    t = np.zeros(65536)
    lib().orc_lut(_p(t))
    return t


def boxmax(grid, win):
    g = np.ascontiguousarray(grid, dtype=np.uint16)
    out = np.zeros_like(g)
    rc = lib().orc_boxmax(_p(g), g.shape[0], g.shape[1], win, _p(out))
    if rc:
        raise ValueError("window does not fit")
    return out


def search_step(res, ranges):
    r = _f64(ranges)
    sx, sy, st = C.c_double(), C.c_double(), C.c_double()
    lib().orc_search_step(C.c_double(res), _p(r), r.size, C.byref(sx), C.byref(sy), C.byref(st))
    return sx.value, sy.value, st.value


def project(geom, pose, angles, ranges):
    a, r = _f64(angles), _f64(ranges)
    col = np.zeros(a.size, np.int32)
    row = np.zeros(a.size, np.int32)
    lib().orc_project(_p(_f64(geom)), _p(_f64(pose)), _p(a), _p(r), a.size, _p(col), _p(row))
    return col, row


def _csm_args(case, coarse):
    g = np.ascontiguousarray(case["grid"], dtype=np.uint16)
    c = np.ascontiguousarray(coarse, dtype=np.uint16)
    a, r = _f64(case["angles"]), _f64(case["ranges"])
    return g, c, a, r, _f64(case["geom"]), _f64(case["rel_pose"]), _f64(case["init_pose"])


def csm(case, range_x, range_y, range_t, low_res, score_thr=0.0, known_thr=0.0, coarse=None):
    """Literal ScanMatcherCorrelative::OptimizePose (sequential sweep + pruning)."""
    if coarse is None:
        coarse = boxmax(case["grid"], low_res)
    g, c, a, r, geom, rel, init = _csm_args(case, coarse)
    p = CsmParams(range_x, range_y, range_t, low_res, score_thr, known_thr)
    out = Result()
    lib().orc_csm(_p(g), _p(c), g.shape[0], g.shape[1], _p(geom), _p(a), _p(r), a.size,
                  _p(rel), _p(init), C.byref(p), C.byref(out))
    return out.as_dict()


def csm_omp(case, range_x, range_y, range_t, low_res, score_thr=0.0, known_thr=0.0, coarse=None):
    """orc_csm with the theta loop on all host cores (OpenMP). Returns the result
    dict plus "threads" = OpenMP threads used."""
    if coarse is None:
        coarse = boxmax(case["grid"], low_res)
    g, c, a, r, geom, rel, init = _csm_args(case, coarse)
    p = CsmParams(range_x, range_y, range_t, low_res, score_thr, known_thr)
    out = Result()
    threads = C.c_int(0)
    lib().orc_csm_omp(_p(g), _p(c), g.shape[0], g.shape[1], _p(geom), _p(a), _p(r), a.size,
                      _p(rel), _p(init), C.byref(p), C.byref(out), C.byref(threads))
    d = out.as_dict()
    d["threads"] = threads.value
    return d


def csm_closed_form(case, range_x, range_y, range_t, low_res, score_thr=0.0, known_thr=0.0,
                    coarse=None, dump=False):
    if coarse is None:
        coarse = boxmax(case["grid"], low_res)
    g, c, a, r, geom, rel, init = _csm_args(case, coarse)
    p = CsmParams(range_x, range_y, range_t, low_res, score_thr, known_thr)
    out = Result()
    band = C.c_int(0)
    S = K = CK = None
    if dump:
        sx, sy, st = search_step(geom[0], r)
        import math
        wx = int(math.ceil(0.5 * range_x / sx))
        wy = int(math.ceil(0.5 * range_y / sy))
        wt = int(math.ceil(0.5 * range_t / st))
        nxc, nyc = -(-(2 * wx + 1) // low_res), -(-(2 * wy + 1) // low_res)
        S = np.zeros((2 * wt + 1, nxc * low_res, nyc * low_res), np.uint32)
        K = np.zeros((2 * wt + 1, nxc * low_res, nyc * low_res), np.uint16)
        CK = np.zeros((2 * wt + 1, nxc, nyc), np.uint16)
    lib().orc_csm_closed_form(_p(g), _p(c), g.shape[0], g.shape[1], _p(geom), _p(a), _p(r),
                              a.size, _p(rel), _p(init), C.byref(p), C.byref(out),
                              C.byref(band), _p(S) if dump else None, _p(K) if dump else None,
                              _p(CK) if dump else None)
    d = out.as_dict()
    d["touchesBand"] = band.value
    return (d, S, K, CK) if dump else d


def pyramid(grid, node_height_max):
    return np.stack([boxmax(grid, 1 << h) for h in range(node_height_max + 1)])


def bnb(case, range_x, range_y, range_t, node_height_max, score_thr, known_thr, pyr=None):
    """Literal ScanMatcherBranchBound::OptimizePose (std::priority_queue)."""
    if pyr is None:
        pyr = pyramid(case["grid"], node_height_max)
    pyr = np.ascontiguousarray(pyr, dtype=np.uint16)
    a, r = _f64(case["angles"]), _f64(case["ranges"])
    geom, rel, init = _f64(case["geom"]), _f64(case["rel_pose"]), _f64(case["init_pose"])
    p = BnbParams(range_x, range_y, range_t, node_height_max, score_thr, known_thr)
    out = Result()
    lib().orc_bnb(_p(pyr), pyr.shape[1], pyr.shape[2], _p(geom), _p(a), _p(r), a.size,
                  _p(rel), _p(init), C.byref(p), C.byref(out))
    return out.as_dict()


def bnb_level_dump(case, level_grid, h, range_x, range_y, range_t, node_height_max):
    import math
    lv = np.ascontiguousarray(level_grid, dtype=np.uint16)
    a, r = _f64(case["angles"]), _f64(case["ranges"])
    geom, rel, init = _f64(case["geom"]), _f64(case["rel_pose"]), _f64(case["init_pose"])
    p = BnbParams(range_x, range_y, range_t, node_height_max, 0.0, 0.0)
    sx, sy, st = search_step(geom[0], r)
    wx = int(math.ceil(0.5 * range_x / sx))
    wy = int(math.ceil(0.5 * range_y / sy))
    wt = int(math.ceil(0.5 * range_t / st))
    big, s = 1 << node_height_max, 1 << h
    nx = -(-(2 * wx + 1) // big) * big // s
    ny = -(-(2 * wy + 1) // big) * big // s
    S = np.zeros((2 * wt + 1, nx, ny), np.uint32)
    K = np.zeros((2 * wt + 1, nx, ny), np.uint16)
    lib().orc_bnb_level_dump(_p(lv), lv.shape[0], lv.shape[1], _p(geom), _p(a), _p(r), a.size,
                             _p(rel), _p(init), C.byref(p), h, _p(S), _p(K))
    return S, K


def score_at(level_grid, geom, angles, ranges, pose):
    lv = np.ascontiguousarray(level_grid, dtype=np.uint16)
    a, r = _f64(angles), _f64(ranges)
    known = C.c_int(0)
    s = lib().orc_score_at(_p(lv), lv.shape[0], lv.shape[1], _p(_f64(geom)), _p(a), _p(r),
                           a.size, _p(_f64(pose)), C.byref(known))
    return s, known.value


def grid_search(case, range_x, range_y, range_t, step_x, step_y, step_t, score_thr=0.0, known_thr=0.0):
    """Literal ScanMatcherGridSearch::OptimizePose."""
    g = np.ascontiguousarray(case["grid"], dtype=np.uint16)
    a, r = _f64(case["angles"]), _f64(case["ranges"])
    geom, rel, init = _f64(case["geom"]), _f64(case["rel_pose"]), _f64(case["init_pose"])
    p = GridParams(range_x, range_y, range_t, step_x, step_y, step_t, score_thr, known_thr)
    out = Result()
    idx = (C.c_int * 3)()
    evals = C.c_longlong(0)
    lib().orc_grid_search(_p(g), g.shape[0], g.shape[1], _p(geom), _p(a), _p(r), a.size, _p(rel),
                          _p(init), C.byref(p), C.byref(out), idx, C.byref(evals))
    d = out.as_dict()
    d["bestIdx"] = list(idx)
    d["evaluations"] = evals.value
    return d


# ---- map building (oracle/map_oracle.cpp) ----

class MapShape(C.Structure):
    _fields_ = [("res", C.c_double), ("offX", C.c_double), ("offY", C.c_double),
                ("rows", C.c_int), ("cols", C.c_int), ("log2Block", C.c_int)]


class ScanNode(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("angles", C.c_void_p), ("ranges", C.c_void_p),
                ("n", C.c_int), ("rel", C.c_double * 3), ("minRange", C.c_double),
                ("maxRange", C.c_double)]


class BuilderParams(C.Structure):
    _fields_ = [("usableMin", C.c_double), ("usableMax", C.c_double), ("probHit", C.c_double),
                ("probMiss", C.c_double), ("subpixel", C.c_int)]


def _nodes(nodes):
    keep = []
    arr = (ScanNode * len(nodes))()
    for i, nd in enumerate(nodes):
        a, r = _f64(nd["angles"]), _f64(nd["ranges"])
        keep += [a, r]
        arr[i].pose[:] = list(nd["pose"])
        arr[i].angles = a.ctypes.data
        arr[i].ranges = r.ctypes.data
        arr[i].n = a.size
        arr[i].rel[:] = list(nd.get("rel_pose", (0.0, 0.0, 0.0)))
        arr[i].minRange = nd.get("min_range", 0.0)
        arr[i].maxRange = nd.get("max_range", 1e9)
    return arr, keep


def construct_map(shape, map_pose, nodes, usable_min=0.01, usable_max=20.0, prob_hit=0.62,
                  prob_miss=0.46, subpixel=100):
    """Literal GridMapBuilder::ConstructMapFromScans on a dense array. shape =
    dict(res, off_x, off_y, rows, cols, log2_block) of the map BEFORE the call.
    Returns (new shape dict, grid, stats dict)."""
    sh = MapShape(shape["res"], shape["off_x"], shape["off_y"], shape["rows"], shape["cols"],
                  shape["log2_block"])
    arr, keep = _nodes(nodes)
    prm = BuilderParams(usable_min, usable_max, prob_hit, prob_miss, subpixel)
    mp = _f64(map_pose)
    rc = lib().orc_map_resize(C.byref(sh), _p(mp), arr, len(nodes), C.byref(prm))
    if rc:
        raise ValueError("orc_map_resize failed: %d" % rc)
    grid = np.zeros((sh.rows, sh.cols), np.uint16)
    stats = (C.c_longlong * 4)()
    rc = lib().orc_map_integrate(C.byref(sh), _p(mp), arr, len(nodes), C.byref(prm), _p(grid), stats)
    if rc:
        raise ValueError("orc_map_integrate failed: %d" % rc)
    new_shape = dict(res=sh.res, off_x=sh.offX, off_y=sh.offY, rows=sh.rows, cols=sh.cols,
                     log2_block=sh.log2Block)
    return new_shape, grid, dict(rays=stats[0], updates=stats[1], oob_reads=stats[2],
                                 end_missing=stats[3])


def update_map(shape, grid, map_pose, node, usable_min=0.01, usable_max=20.0, prob_hit=0.62,
               prob_miss=0.46, subpixel=100):
    """Literal GridMapBuilder::UpdateGridMap for one scan node on a dense array:
    Expand (keeping the cells), then the ray casts. Returns (new shape, new
    grid, stats)."""
    sh = MapShape(shape["res"], shape["off_x"], shape["off_y"], shape["rows"], shape["cols"],
                  shape["log2_block"])
    arr, keep = _nodes([node])
    prm = BuilderParams(usable_min, usable_max, prob_hit, prob_miss, subpixel)
    mp = _f64(map_pose)
    r0, c0 = C.c_int(0), C.c_int(0)
    rc = lib().orc_map_expand(C.byref(sh), _p(mp), arr, C.byref(prm), C.byref(r0), C.byref(c0))
    if rc:
        raise ValueError("orc_map_expand failed: %d" % rc)
    old = np.ascontiguousarray(grid, dtype=np.uint16)
    new = np.zeros((sh.rows, sh.cols), np.uint16)
    new[-r0.value:-r0.value + old.shape[0], -c0.value:-c0.value + old.shape[1]] = old
    stats = (C.c_longlong * 4)()
    rc = lib().orc_map_integrate_keep(C.byref(sh), _p(mp), arr, C.byref(prm), _p(new), stats)
    if rc:
        raise ValueError("orc_map_integrate_keep failed: %d" % rc)
    new_shape = dict(res=sh.res, off_x=sh.offX, off_y=sh.offY, rows=sh.rows, cols=sh.cols,
                     log2_block=sh.log2Block)
    return new_shape, new, dict(rays=stats[0], updates=stats[1], oob_reads=stats[2],
                                end_missing=stats[3], row_min=r0.value, col_min=c0.value)


def ray_cells(sx, sy, ex, ey, scale=100, cap=1 << 16):
    out = np.zeros(2 * cap, np.int32)
    n = lib().orc_ray_cells(sx, sy, ex, ey, scale, _p(out), cap)
    return [tuple(out[2 * i:2 * i + 2]) for i in range(min(n, cap))]


def bayes_update(value, prob):
    lib().orc_bayes_update.restype = C.c_uint
    lib().orc_bayes_update.argtypes = [C.c_uint, C.c_double]
    return lib().orc_bayes_update(value, prob)


# ---- cost / covariance / linear-solver refinement (oracle/cost_oracle.cpp) ----

class CostGrid(C.Structure):
    _fields_ = [("v", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("res", C.c_double),
                ("offX", C.c_double), ("offY", C.c_double), ("alloc", C.c_void_p), ("log2Block", C.c_int)]


class RefineResult(C.Structure):
    _fields_ = [("normalizedInitialCost", C.c_double), ("normalizedCost", C.c_double),
                ("sensorPose", C.c_double * 3), ("bestSensorPose", C.c_double * 3),
                ("estimatedPose", C.c_double * 3), ("covariance", C.c_double * 9),
                ("lambda_", C.c_double), ("iterations", C.c_int)]


def _cost_grid(grid, geom, alloc=None, log2_block=4):
    g = np.ascontiguousarray(grid, dtype=np.uint16)
    a = None if alloc is None else np.ascontiguousarray(alloc, dtype=np.uint8)
    cg = CostGrid(g.ctypes.data, g.shape[0], g.shape[1], geom[0], geom[1], geom[2],
                  None if a is None else a.ctypes.data, log2_block)
    return cg, (g, a)


def cost(grid, geom, angles, ranges, sensor_pose, alloc=None, log2_block=4):
    """CostSquareError::Cost (sum of squared errors, not normalized)."""
    cg, keep = _cost_grid(grid, geom, alloc, log2_block)
    a, r = _f64(angles), _f64(ranges)
    lib().orc_cost.restype = C.c_double
    return lib().orc_cost(C.byref(cg), _p(a), _p(r), a.size, _p(_f64(sensor_pose)))


def hessian_residual(grid, geom, angles, ranges, sensor_pose, alloc=None, log2_block=4):
    cg, keep = _cost_grid(grid, geom, alloc, log2_block)
    a, r = _f64(angles), _f64(ranges)
    h, res = np.zeros(9), np.zeros(3)
    lib().orc_hessian_residual(C.byref(cg), _p(a), _p(r), a.size, _p(_f64(sensor_pose)), _p(h), _p(res))
    return h.reshape(3, 3), res


def covariance(grid, geom, angles, ranges, sensor_pose, covariance_scale=1e4, alloc=None, log2_block=4):
    cg, keep = _cost_grid(grid, geom, alloc, log2_block)
    a, r = _f64(angles), _f64(ranges)
    cov = np.zeros(9)
    lib().orc_covariance(C.byref(cg), _p(a), _p(r), a.size, _p(_f64(sensor_pose)),
                         C.c_double(covariance_scale), _p(cov))
    return cov.reshape(3, 3)


def inverse3(m):
    out = np.zeros(9)
    lib().orc_inverse3(_p(_f64(m).reshape(-1)), _p(out))
    return out.reshape(3, 3)


def solve3(m, b):
    x = np.zeros(3)
    lib().orc_solve3_colpiv_qr(_p(_f64(m).reshape(-1)), _p(_f64(b)), _p(x))
    return x


def linear_solver(grid, geom, angles, ranges, rel_pose, init_pose, iterations_max=10,
                  convergence_threshold=1e-4, lambda_=1e-4, covariance_scale=1e4, alloc=None,
                  log2_block=4):
    """ScanMatcherLinearSolver::OptimizePose; defaults as launcher_settings_default.json:28-35, 11-13."""
    cg, keep = _cost_grid(grid, geom, alloc, log2_block)
    a, r = _f64(angles), _f64(ranges)
    out = RefineResult()
    lib().orc_linear_solver(C.byref(cg), _p(a), _p(r), a.size, _p(_f64(rel_pose)), _p(_f64(init_pose)),
                            iterations_max, C.c_double(convergence_threshold), C.c_double(lambda_),
                            C.c_double(covariance_scale), C.byref(out))
    return dict(normalized_initial_cost=out.normalizedInitialCost, normalized_cost=out.normalizedCost,
                sensor_pose=list(out.sensorPose), best_sensor_pose=list(out.bestSensorPose),
                estimated_pose=list(out.estimatedPose),
                covariance=np.array(out.covariance).reshape(3, 3), lambda_=out.lambda_,
                iterations=out.iterations)
