"""Thin Python host layer over the C ABI (include/csm_hip.h).

Mirrors the reference's matcher interface for the hot path:
  ScanMatcherCorrelativeHIP.optimize_pose(...)   <- ScanMatcherCorrelative::OptimizePose
  (src/my_lidar_graph_slam/mapping/scan_matcher_correlative.cpp:92-244)
  LoopDetectorBranchBoundHIP.detect(...)         <- LoopDetectorBranchBound::Detect
  (src/my_lidar_graph_slam/mapping/loop_detector_branch_bound.cpp:59-156)
All arithmetic happens in libcsm_hip.so; this file only marshals numpy arrays.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class CsmError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("csm_hip error %d: %s" % (code, text))
        self.code = code


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def result_to_dict(r):
    return dict(found=int(r.found), best_x=int(r.best_x), best_y=int(r.best_y),
                best_theta=int(r.best_theta), key=int(r.key),
                sum_values=int(r.sum_values), known=int(r.known),
                tie_count=int(r.tie_count), flags=int(r.flags), score=float(r.score))


def summary_to_dict(s):
    return dict(pose_found=int(s.pose_found), win_x=s.win_x, win_y=s.win_y,
                win_theta=s.win_theta, step_x=s.step_x, step_y=s.step_y,
                step_theta=s.step_theta, sensor_pose=list(s.sensor_pose),
                best_sensor_pose=list(s.best_sensor_pose),
                estimated_pose=list(s.estimated_pose),
                input_setup_us=s.input_setup_us, optimization_us=s.optimization_us,
                candidates=int(s.candidates), raw=result_to_dict(s.raw))


class PreparedQueries:
    """Context.prepare_queries()'s result: the csm_loop_query array plus the
    numpy arrays its pointers refer to."""

    def __init__(self, arr, keep):
        self.arr, self.keep, self.n = arr, keep, len(arr)


class SummaryArray:
    """The csm_summary array of a batch call, converted on demand."""

    def __init__(self, out):
        self.out = out

    def __len__(self):
        return len(self.out)

    def __getitem__(self, i):
        return summary_to_dict(self.out[i])

    def __iter__(self):
        return (summary_to_dict(o) for o in self.out)

    def record_bytes(self):
        """The 48-byte csm_result of every query, back to back (what the ranks all-gather)."""
        n, size, off = len(self.out), C.sizeof(L.Summary), L.Summary.raw.offset
        flat = np.frombuffer(self.out, dtype=np.uint8).reshape(n, size)
        return np.ascontiguousarray(flat[:, off:off + C.sizeof(L.Result)]).reshape(-1)

    def total(self, field):
        return sum(int(getattr(o, field)) for o in self.out)


# ---- host-only helpers (no GPU needed) ----

def host_search_step(resolution, ranges):
    lib = L.load()
    r = _f64(ranges)
    sx, sy, st = C.c_double(), C.c_double(), C.c_double()
    rc = lib.csm_host_search_step(resolution, _ptr(r), r.size, C.byref(sx), C.byref(sy), C.byref(st))
    if rc:
        raise CsmError(rc, "csm_host_search_step")
    return sx.value, sy.value, st.value


def host_map_resize(shape, box, expand=False):
    """GridMap::Resize / Expand on an index box (min col, min row, max col, max
    row; inclusive). Returns (new shape dict, (first row, first col) in the old frame)."""
    sh = L.MapShape(shape["res"], shape["off_x"], shape["off_y"], shape["rows"], shape["cols"],
                    shape["log2_block"])
    b = np.ascontiguousarray(box, dtype=np.int32)
    shift = np.zeros(2, np.int32)
    rc = L.load().csm_host_map_resize(C.byref(sh), _ptr(b), 1 if expand else 0, _ptr(shift))
    if rc:
        raise CsmError(rc, "csm_host_map_resize")
    return (dict(res=sh.resolution, off_x=sh.offset_x, off_y=sh.offset_y, rows=sh.rows, cols=sh.cols,
                 log2_block=sh.log2_block_size), (int(shift[0]), int(shift[1])))


def host_window(rng, step):
    return L.load().csm_host_window(rng, step)


def host_min_known(n, thr):
    return L.load().csm_host_min_known(n, thr)


def host_compound(a, b):
    out = np.zeros(3)
    L.load().csm_host_compound(_ptr(_f64(a)), _ptr(_f64(b)), _ptr(out))
    return out


def host_inverse_compound(a, b):
    out = np.zeros(3)
    L.load().csm_host_inverse_compound(_ptr(_f64(a)), _ptr(_f64(b)), _ptr(out))
    return out


def host_move_backward(a, b):
    out = np.zeros(3)
    L.load().csm_host_move_backward(_ptr(_f64(a)), _ptr(_f64(b)), _ptr(out))
    return out


def host_project(geom, sensor_pose, step_theta, win_theta, angles, ranges, want_products=False):
    lib = L.load()
    a, r = _f64(angles), _f64(ranges)
    n, nt = a.size, 2 * win_theta + 1
    col = np.zeros((nt, n), np.int32)
    row = np.zeros((nt, n), np.int32)
    rc_ = np.zeros((nt, n)) if want_products else None
    rs_ = np.zeros((nt, n)) if want_products else None
    g = L.Geometry(*geom)
    sp = _f64(sensor_pose)
    rc = lib.csm_host_project(C.byref(g), _ptr(sp), step_theta, win_theta, _ptr(a), _ptr(r), n,
                              _ptr(col), _ptr(row),
                              _ptr(rc_) if want_products else None,
                              _ptr(rs_) if want_products else None)
    if rc:
        raise CsmError(rc, "csm_host_project")
    return (col, row, rc_, rs_) if want_products else (col, row)


def host_probability_lut():
    lut = np.zeros(65536)
    L.load().csm_host_probability_lut(_ptr(lut))
    return lut


class Context:
    """One csm_ctx: owns the device grids, workspaces and a stream."""

    def __init__(self, device_id=0, tuning_off=0, map_uncertain_cap=0):
        """tuning_off: L.TUNE_* bits (switch launch optimisations off: A/B runs, tests)."""
        self.lib = L.load()
        self._ctx = C.c_void_p()
        cfg = L.Config()
        cfg.device_id = device_id
        cfg.tuning_off = tuning_off
        cfg.map_uncertain_cap = map_uncertain_cap
        rc = self.lib.csm_create(C.byref(cfg), C.byref(self._ctx))
        if rc:
            self._ctx = C.c_void_p()
            raise CsmError(rc, "csm_create failed (no GPU?)")
        self.shapes = {}

    def close(self):
        if self._ctx:
            self.lib.csm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise CsmError(rc, self.lib.csm_last_error(self._ctx).decode())

    def set_stream(self, stream_ptr):
        self._check(self.lib.csm_set_stream(self._ctx, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self.lib.csm_synchronize(self._ctx))

    def upload_grid(self, map_id, grid):
        g = np.ascontiguousarray(grid, dtype=np.uint16)
        self._check(self.lib.csm_upload_grid(self._ctx, map_id, _ptr(g), g.shape[0], g.shape[1]))
        self.shapes[map_id] = g.shape

    def upload_grid_blocks(self, map_id, blocks, block_rows, block_cols, log2_block):
        """Block-sparse upload (the reference's own storage): blocks[br * block_cols + bc] is a
        (2^k, 2^k) uint16 array or None for an unallocated block."""
        keep = [None if b is None else np.ascontiguousarray(b, dtype=np.uint16) for b in blocks]
        ptrs = (C.c_void_p * len(keep))(*[None if b is None else b.ctypes.data for b in keep])
        self._check(self.lib.csm_upload_grid_blocks(self._ctx, map_id, ptrs, block_rows, block_cols, log2_block))
        self.shapes[map_id] = (block_rows << log2_block, block_cols << log2_block)

    def has_grid(self, map_id):
        return bool(self.lib.csm_has_grid(self._ctx, map_id))

    def release_grid(self, map_id):
        self._check(self.lib.csm_release_grid(self._ctx, map_id))
        self.shapes.pop(map_id, None)

    def build_pyramid(self, map_id, win_sizes):
        w = np.ascontiguousarray(win_sizes, dtype=np.int32)
        self._check(self.lib.csm_build_pyramid(
            self._ctx, map_id, w.ctypes.data_as(C.POINTER(C.c_int32)), w.size))

    def project_scan(self, geom, sensor_pose, step_theta, win_theta, angles, ranges, cap=1 << 20):
        """csm_project_scan: the device projection with its certificate. Returns
        (col [nt, n], row [nt, n], flat indices of the uncertified entries, their full count)."""
        a, r = _f64(angles), _f64(ranges)
        n, nt = a.size, 2 * win_theta + 1
        col = np.zeros((nt, n), np.int32)
        row = np.zeros((nt, n), np.int32)
        unc = np.zeros(cap, np.uint32)
        count = C.c_int32(0)
        g = L.Geometry(*geom)
        sp = _f64(sensor_pose)
        self._check(self.lib.csm_project_scan(self._ctx, C.byref(g), _ptr(sp), step_theta, win_theta,
                                              _ptr(a), _ptr(r), n, _ptr(col), _ptr(row), _ptr(unc), cap,
                                              C.byref(count)))
        return col, row, unc[:min(count.value, cap)].copy(), count.value

    def build_pyramids(self, map_ids, win_sizes):
        """csm_build_pyramids: box-max(win) of every listed map for every win, built where missing."""
        ids = np.ascontiguousarray(map_ids, dtype=np.uint64)
        w = np.ascontiguousarray(win_sizes, dtype=np.int32)
        self._check(self.lib.csm_build_pyramids(self._ctx, _ptr(ids), ids.size, _ptr(w), w.size))

    def copy_last_batch_records(self, dst_ptr):
        """Device-to-device copy of the last batch's records (query order) into dst_ptr."""
        self._check(self.lib.csm_copy_last_batch_records(self._ctx, C.c_void_p(dst_ptr)))

    def download_level(self, map_id, level):
        out = np.zeros(self.shapes[map_id], np.uint16)
        self._check(self.lib.csm_download_level(self._ctx, map_id, level, _ptr(out)))
        return out

    @staticmethod
    def make_window(n_theta, n_points, win_x, win_y, low_resolution, coarse_level,
                    min_known, score_threshold, merge_mode=0):
        w = L.Window()
        w.merge_mode = merge_mode
        w.n_theta, w.n_points = n_theta, n_points
        w.win_x, w.win_y = win_x, win_y
        w.low_resolution, w.coarse_level = low_resolution, coarse_level
        w.min_known, w.score_threshold = min_known, score_threshold
        return w

    def score_window(self, map_id, window, hit_col, hit_row, dump=False):
        col = np.ascontiguousarray(hit_col, dtype=np.int32)
        row = np.ascontiguousarray(hit_row, dtype=np.int32)
        res = L.Result()
        if not dump:
            self._check(self.lib.csm_score_window(self._ctx, map_id, C.byref(window),
                                                  _ptr(col), _ptr(row), C.byref(res)))
            return result_to_dict(res)
        Lr = window.low_resolution
        nxc = -(-(2 * window.win_x + 1) // Lr)
        nyc = -(-(2 * window.win_y + 1) // Lr)
        s = np.zeros((window.n_theta, nxc * Lr, nyc * Lr), np.uint32)
        k = np.zeros((window.n_theta, nxc * Lr, nyc * Lr), np.uint16)
        ck = np.zeros((window.n_theta, nxc, nyc), np.uint16)
        self._check(self.lib.csm_score_window_dump(self._ctx, map_id, C.byref(window),
                                                   _ptr(col), _ptr(row), C.byref(res),
                                                   _ptr(s), _ptr(k), _ptr(ck)))
        return result_to_dict(res), s, k, ck

    def score_window_dev(self, map_id, window, col_ptr, row_ptr, out_ptr):
        self._check(self.lib.csm_score_window_dev(self._ctx, map_id, C.byref(window),
                                                  C.c_void_p(col_ptr), C.c_void_p(row_ptr),
                                                  C.c_void_p(out_ptr)))

    def prepare_windows(self, map_ids, windows, col_ptrs, row_ptrs):
        """The argument arrays of score_windows_dev(), built once for a batch
        that is scored repeatedly."""
        n = len(windows)
        ids = (C.c_uint64 * n)(*map_ids)
        wins = (L.Window * n)(*windows)
        cols = (C.c_void_p * n)(*col_ptrs)
        rows = (C.c_void_p * n)(*row_ptrs)
        return n, ids, wins, cols, rows

    def score_windows_dev(self, prepared, out_ptr):
        """csm_score_window_dev for many windows in one launch chain; `prepared`
        from prepare_windows(); out_ptr: device pointer to n 48-byte records."""
        n, ids, wins, cols, rows = prepared
        self._check(self.lib.csm_score_windows_dev(self._ctx, n, ids, wins, cols, rows,
                                                   C.c_void_p(out_ptr)))

    def score_windows_dump_dev(self, prepared, out_ptr, dump_s_ptrs=None, dump_k_ptrs=None, dump_f_ptrs=None):
        """score_windows_dev() that also writes, for the windows whose device pointers
        (0 = none) are given, every candidate's integer sums (S uint32, K uint16) and / or
        its fp32 key from the bound pass, [n_theta][nx][ny] each."""
        n, ids, wins, cols, rows = prepared

        def arr(ptrs):
            return (C.c_void_p * n)(*[p or None for p in ptrs]) if ptrs is not None else None
        self._check(self.lib.csm_score_windows_dump_dev(self._ctx, n, ids, wins, cols, rows, C.c_void_p(out_ptr),
                                                        arr(dump_s_ptrs), arr(dump_k_ptrs), arr(dump_f_ptrs)))

    def last_search_info(self):
        """What the last correlative_match() evaluated (nominal / coarse nodes / fine candidates)."""
        info = L.SearchInfo()
        self._check(self.lib.csm_last_search_info(self._ctx, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}

    def bound_pass_stats(self):
        """(candidate blocks the exact kernel scored, blocks it skipped after the fp32 bound
        pass) since the last call."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.csm_bound_pass_stats(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def correlative_match(self, map_id, geom, angles, ranges, rel_pose, init_pose,
                          range_x, range_y, range_theta, low_resolution,
                          score_threshold=0.0, known_rate_threshold=0.0):
        a, r = _f64(angles), _f64(ranges)
        scan = L.Scan()
        scan.angles = a.ctypes.data_as(C.POINTER(C.c_double))
        scan.ranges = r.ctypes.data_as(C.POINTER(C.c_double))
        scan.n_points = a.size
        scan.relative_sensor_pose[:] = list(rel_pose)
        g = L.Geometry(*geom)
        p = L.CorrelativeParams()
        p.range_x, p.range_y, p.range_theta = range_x, range_y, range_theta
        p.low_resolution = low_resolution
        p.score_threshold, p.known_rate_threshold = score_threshold, known_rate_threshold
        init = _f64(init_pose)
        out = L.Summary()
        self._check(self.lib.csm_correlative_match(self._ctx, map_id, C.byref(g), C.byref(scan),
                                                   _ptr(init), C.byref(p), C.byref(out)))
        return summary_to_dict(out)

    def construct_map_from_scans(self, map_id, shape, map_pose, nodes, usable_range_min=0.01,
                                 usable_range_max=20.0, prob_hit=0.62, prob_miss=0.46,
                                 subpixel_scale=100):
        """GridMapBuilder::ConstructMapFromScans
        (src/my_lidar_graph_slam/mapping/grid_map_builder.cpp:561-695) on the
        device; the result becomes the resident grid `map_id`. shape = dict(res,
        off_x, off_y, rows, cols, log2_block) of the map before the call; nodes =
        dicts(pose, angles, ranges, rel_pose, min_range, max_range). Returns
        (new shape dict, info dict); defaults as launcher_settings_default.json:183-186."""
        return self._map_build(map_id, shape, map_pose, nodes, False, usable_range_min, usable_range_max,
                               prob_hit, prob_miss, subpixel_scale)

    def update_map_with_scan(self, map_id, shape, map_pose, node, usable_range_min=0.01,
                             usable_range_max=20.0, prob_hit=0.62, prob_miss=0.46, subpixel_scale=100):
        """The grid half of GridMapBuilder::UpdateGridMap
        (src/my_lidar_graph_slam/mapping/grid_map_builder.cpp:389-494): one scan
        node on top of the resident map `map_id`, which grows if it has to."""
        return self._map_build(map_id, shape, map_pose, [node], True, usable_range_min, usable_range_max,
                               prob_hit, prob_miss, subpixel_scale)

    def _map_build(self, map_id, shape, map_pose, nodes, keep_cells, usable_range_min, usable_range_max,
                   prob_hit, prob_miss, subpixel_scale):
        sh = L.MapShape(shape["res"], shape["off_x"], shape["off_y"], shape["rows"], shape["cols"],
                        shape["log2_block"])
        arr = (L.ScanNode * len(nodes))()
        keep = []
        for i, nd in enumerate(nodes):
            a_, r_ = _f64(nd["angles"]), _f64(nd["ranges"])
            keep += [a_, r_]
            arr[i].global_pose[:] = list(nd["pose"])
            arr[i].scan.angles = a_.ctypes.data_as(C.POINTER(C.c_double))
            arr[i].scan.ranges = r_.ctypes.data_as(C.POINTER(C.c_double))
            arr[i].scan.n_points = a_.size
            arr[i].scan.relative_sensor_pose[:] = list(nd.get("rel_pose", (0.0, 0.0, 0.0)))
            arr[i].min_range = nd.get("min_range", 0.0)
            arr[i].max_range = nd.get("max_range", 1e9)
        prm = L.MapBuilderParams(usable_range_min, usable_range_max, prob_hit, prob_miss, subpixel_scale)
        info = L.MapBuildInfo()
        mp = _f64(map_pose)
        if keep_cells:
            self._check(self.lib.csm_update_map_with_scan(self._ctx, map_id, C.byref(sh), _ptr(mp), arr,
                                                          C.byref(prm), C.byref(info)))
        else:
            self._check(self.lib.csm_construct_map_from_scans(self._ctx, map_id, C.byref(sh), _ptr(mp), arr,
                                                              len(nodes), C.byref(prm), C.byref(info)))
        self.shapes[map_id] = (sh.rows, sh.cols)
        new_shape = dict(res=sh.resolution, off_x=sh.offset_x, off_y=sh.offset_y, rows=sh.rows,
                         cols=sh.cols, log2_block=sh.log2_block_size)
        return new_shape, {name: getattr(info, name) for name, _ in L.MapBuildInfo._fields_}

    def grid_search_match(self, map_id, geom, angles, ranges, rel_pose, init_pose,
                          range_x, range_y, range_theta, step_x, step_y, step_theta,
                          score_threshold=0.0, known_rate_threshold=0.0):
        """ScanMatcherGridSearch::OptimizePose
        (src/my_lidar_graph_slam/mapping/scan_matcher_grid_search.cpp:69-190)."""
        a, r = _f64(angles), _f64(ranges)
        scan = L.Scan()
        scan.angles = a.ctypes.data_as(C.POINTER(C.c_double))
        scan.ranges = r.ctypes.data_as(C.POINTER(C.c_double))
        scan.n_points = a.size
        scan.relative_sensor_pose[:] = list(rel_pose)
        g = L.Geometry(*geom)
        p = L.GridSearchParams(range_x, range_y, range_theta, step_x, step_y, step_theta,
                               score_threshold, known_rate_threshold)
        init = _f64(init_pose)
        out = L.Summary()
        self._check(self.lib.csm_grid_search_match(self._ctx, map_id, C.byref(g), C.byref(scan),
                                                   _ptr(init), C.byref(p), C.byref(out)))
        return summary_to_dict(out)

    def prepare_queries(self, queries):
        """Flatten a list of dict(map_id, geom, angles, ranges, rel_pose,
        init_pose) into the C array the batch entry points take. A caller that
        repeats a batch (or a C++ caller, which holds such an array anyway)
        passes the result instead of the list and skips the per-call marshalling."""
        if isinstance(queries, PreparedQueries):
            return queries
        n = len(queries)
        arr = (L.LoopQuery * n)()
        keep = []
        for i, q in enumerate(queries):
            a, r = _f64(q["angles"]), _f64(q["ranges"])
            keep.append((a, r))
            arr[i].map_id = q["map_id"]
            arr[i].geometry = L.Geometry(*q["geom"])
            arr[i].scan.angles = a.ctypes.data_as(C.POINTER(C.c_double))
            arr[i].scan.ranges = r.ctypes.data_as(C.POINTER(C.c_double))
            arr[i].scan.n_points = a.size
            arr[i].scan.relative_sensor_pose[:] = list(q["rel_pose"])
            arr[i].initial_pose[:] = list(q["init_pose"])
        return PreparedQueries(arr, keep)

    def correlative_match_batch(self, queries, range_x, range_y, range_theta, low_resolution,
                                score_threshold, known_rate_threshold, as_records=False):
        """queries: list of dict(map_id, geom, angles, ranges, rel_pose, init_pose)
        or prepare_queries()'s result. as_records: return the C summaries
        (SummaryArray) instead of dicts."""
        prep = self.prepare_queries(queries)
        p = L.CorrelativeParams()
        p.range_x, p.range_y, p.range_theta = range_x, range_y, range_theta
        p.low_resolution = low_resolution
        p.score_threshold, p.known_rate_threshold = score_threshold, known_rate_threshold
        out = (L.Summary * prep.n)()
        self._check(self.lib.csm_correlative_match_batch(self._ctx, prep.arr, prep.n, C.byref(p), out))
        return SummaryArray(out) if as_records else [summary_to_dict(o) for o in out]

    def bnb_match_batch(self, queries, range_x, range_y, range_theta, node_height_max,
                        score_threshold, known_rate_threshold, as_records=False):
        """As correlative_match_batch, for the branch-and-bound detector."""
        prep = self.prepare_queries(queries)
        p = L.BnbParams()
        p.range_x, p.range_y, p.range_theta = range_x, range_y, range_theta
        p.node_height_max = node_height_max
        p.score_threshold, p.known_rate_threshold = score_threshold, known_rate_threshold
        out = (L.Summary * prep.n)()
        self._check(self.lib.csm_bnb_match_batch(self._ctx, prep.arr, prep.n, C.byref(p), out))
        return SummaryArray(out) if as_records else [summary_to_dict(o) for o in out]

    def set_block_allocation(self, map_id, log2_block_size, allocated):
        """csm_set_block_allocation: uint8 [block rows, block cols] (None: derive from the cells)."""
        a = None if allocated is None else np.ascontiguousarray(allocated, dtype=np.uint8)
        self._check(self.lib.csm_set_block_allocation(self._ctx, map_id, log2_block_size,
                                                      None if a is None else _ptr(a)))

    @staticmethod
    def _refine_to_dict(r):
        return dict(normalized_initial_cost=r.normalized_initial_cost, normalized_cost=r.normalized_cost,
                    sensor_pose=list(r.sensor_pose), best_sensor_pose=list(r.best_sensor_pose),
                    estimated_pose=list(r.estimated_pose),
                    covariance=np.array(r.covariance).reshape(3, 3),
                    hessian=np.array(r.hessian).reshape(3, 3), lambda_=r.lambda_, iterations=r.iterations)

    def cost_covariance_batch(self, queries, sensor_poses, covariance_scale=1e4):
        """CostSquareError::Cost / n and ComputeCovariance at the given sensor poses."""
        prep = self.prepare_queries(queries)
        sp = _f64(sensor_poses).reshape(prep.n, 3)
        out = (L.RefineResult * prep.n)()
        self._check(self.lib.csm_cost_covariance_batch(self._ctx, prep.arr, prep.n, _ptr(sp),
                                                       covariance_scale, out))
        return [self._refine_to_dict(o) for o in out]

    def linear_solver_batch(self, queries, iterations_max=10, convergence_threshold=1e-4, lambda_=1e-4,
                            covariance_scale=1e4):
        """ScanMatcherLinearSolver::OptimizePose per query (init_pose = robot pose to refine);
        defaults as launcher_settings_default.json:28-35, 11-13."""
        prep = self.prepare_queries(queries)
        p = L.RefineParams(covariance_scale, iterations_max, 0, convergence_threshold, lambda_)
        out = (L.RefineResult * prep.n)()
        self._check(self.lib.csm_linear_solver_batch(self._ctx, prep.arr, prep.n, C.byref(p), out))
        return [self._refine_to_dict(o) for o in out]

    def enable_kernel_timing(self, on=True):
        self._check(self.lib.csm_enable_kernel_timing(self._ctx, 1 if on else 0))

    def reset_kernel_timing(self):
        self._check(self.lib.csm_reset_kernel_timing(self._ctx))

    def kernel_time(self, name):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.csm_kernel_time(self._ctx, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def host_shard_bounds(n_queries, member, n_members):
    lo, hi = C.c_int32(), C.c_int32()
    L.load().csm_shard_bounds(n_queries, member, n_members, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class Group:
    """csm_group: one context per listed device inside this process; batches are cut
    into contiguous blocks, one host thread per member, one all-gather of the
    records (RCCL when the devices differ)."""

    def __init__(self, device_ids, force_rccl=False, tuning_off=0):
        self.lib = L.load()
        self._g = C.c_void_p()
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        cfg = L.Config()
        cfg.tuning_off = tuning_off
        rc = self.lib.csm_group_create_ex(_ptr(ids), ids.size, C.byref(cfg),
                                          L.GROUP_FORCE_RCCL if force_rccl else 0, C.byref(self._g))
        if rc:
            self._g = C.c_void_p()
            raise CsmError(rc, "csm_group_create failed")
        self.members = []
        for k in range(self.lib.csm_group_size(self._g)):
            ctx = Context.__new__(Context)          # a view of the member: the group owns it
            ctx.lib, ctx.shapes = self.lib, {}
            ctx._ctx = C.c_void_p(self.lib.csm_group_member(self._g, k))
            ctx.close = lambda: None
            self.members.append(ctx)

    def close(self):
        if self._g:
            for m in self.members:
                m._ctx = C.c_void_p()
            self.lib.csm_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise CsmError(rc, self.lib.csm_group_last_error(self._g).decode())

    def upload_grids(self, queries, grids):
        """Every query's map to the member whose block holds the query."""
        n, m = len(queries), len(self.members)
        for k, ctx in enumerate(self.members):
            lo, hi = host_shard_bounds(n, k, m)
            for q in queries[lo:hi]:
                if not ctx.has_grid(q["map_id"]):
                    ctx.upload_grid(q["map_id"], grids[q["map_id"]])

    def bnb_match_batch(self, queries, range_x, range_y, range_theta, node_height_max,
                        score_threshold, known_rate_threshold, as_records=False):
        prep = self.members[0].prepare_queries(queries)
        p = L.BnbParams()
        p.range_x, p.range_y, p.range_theta = range_x, range_y, range_theta
        p.node_height_max = node_height_max
        p.score_threshold, p.known_rate_threshold = score_threshold, known_rate_threshold
        out = (L.Summary * prep.n)()
        self._check(self.lib.csm_group_bnb_match_batch(self._g, prep.arr, prep.n, C.byref(p), out))
        return SummaryArray(out) if as_records else [summary_to_dict(o) for o in out]

    def gathered_records(self, member):
        """Member `member`'s device buffer after the exchange, read back: uint8 [n_members, block, 48]."""
        dev, block = C.c_void_p(), C.c_int32()
        self._check(self.lib.csm_group_gathered_records_dev(self._g, member, C.byref(dev), C.byref(block)))
        m = len(self.members)
        n_bytes = m * block.value * C.sizeof(L.Result)
        out = np.zeros(n_bytes, np.uint8)
        self.members[member].synchronize()
        # a plain device-to-host copy of a raw pointer
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), dev, ctypes.c_size_t(n_bytes), 2)
        if rc:
            raise CsmError(rc, "hipMemcpy failed")
        return out.reshape(m, block.value, C.sizeof(L.Result))

    def exchange_info(self):
        used, us = C.c_int32(), C.c_double()
        self._check(self.lib.csm_group_exchange_info(self._g, C.byref(used), C.byref(us)))
        return bool(used.value), us.value


class ScanMatcherCorrelativeHIP:
    """Drop-in for ScanMatcherCorrelative (constructor arguments as in
    src/my_lidar_graph_slam/scan_matcher_factory.cpp:173-177)."""

    def __init__(self, name, low_resolution, range_x, range_y, range_theta, ctx=None):
        self.name = name
        self.low_resolution = low_resolution
        self.range_x, self.range_y, self.range_theta = range_x, range_y, range_theta
        self.ctx = ctx or Context()
        self._nonce = 1 << 62

    def optimize_pose(self, grid, geom, angles, ranges, rel_pose, init_pose,
                      map_id=None, score_threshold=0.0, known_rate_threshold=0.0):
        """grid may be None when map_id is already resident (the per-LocalMapId
        cache of the loop detectors); a throw-away latest map gets a nonce id,
        like LocalMapId::Invalid in scan_matcher_correlative_fpga.cpp:177-184."""
        mid = map_id
        if mid is None:
            mid = self._nonce
        if grid is not None and (map_id is None or not self.ctx.has_grid(mid)):
            self.ctx.upload_grid(mid, grid)
        out = self.ctx.correlative_match(mid, geom, angles, ranges, rel_pose, init_pose,
                                         self.range_x, self.range_y, self.range_theta,
                                         self.low_resolution, score_threshold,
                                         known_rate_threshold)
        if map_id is None:
            self.ctx.release_grid(mid)
        return out


class ScanMatcherGridSearchHIP:
    """Drop-in for ScanMatcherGridSearch (constructor arguments as in
    src/my_lidar_graph_slam/scan_matcher_factory.cpp, "GridSearch" branch:
    ranges then steps)."""

    def __init__(self, name, range_x, range_y, range_theta, step_x, step_y, step_theta, ctx=None):
        if not (step_x > 0 and step_y > 0 and step_theta > 0):
            raise ValueError("steps must be positive")
        self.name = name
        self.ranges = (range_x, range_y, range_theta)
        self.steps = (step_x, step_y, step_theta)
        self.ctx = ctx or Context()
        self._nonce = (1 << 62) + 1

    def optimize_pose(self, grid, geom, angles, ranges, rel_pose, init_pose,
                      map_id=None, score_threshold=0.0, known_rate_threshold=0.0):
        mid = self._nonce if map_id is None else map_id
        if grid is not None and (map_id is None or not self.ctx.has_grid(mid)):
            self.ctx.upload_grid(mid, grid)
        out = self.ctx.grid_search_match(mid, geom, angles, ranges, rel_pose, init_pose,
                                         *self.ranges, *self.steps, score_threshold,
                                         known_rate_threshold)
        if map_id is None:
            self.ctx.release_grid(mid)
        return out
