"""ctypes binding of libcsm_hip.so (the C ABI declared in include/csm_hip.h).

The library is built in-tree (my-lidar-graph-slam-v2_amd/csrc/libcsm_hip.so) by
__graft_entry__.build(). There is no fallback: if the shared object is missing
the import fails, and without a GPU every compute entry point returns
CSM_ENODEV.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSM_HIP_LIB: a tuning build of the same library (tools/build_variant.sh), for A/B runs on one box
LIB_PATH = os.environ.get("CSM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libcsm_hip.so")

CSM_OK = 0
CSM_ENOENT = -2
CSM_EIO = -5
CSM_ENOMEM = -12
CSM_ENODEV = -19
CSM_EINVAL = -22

FLAG_EDGE_BAND = 1
FLAG_KEY_TIE = 2
FLAG_F64_TIE = 4
FLAG_LITERAL = 8
FLAG_PROJ_DELTA = 16


class Config(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("tuning_off", C.c_uint32), ("map_uncertain_cap", C.c_int32),
                ("reserved", C.c_int32 * 5)]


# csm_config.tuning_off bits (include/csm_hip.h)
TUNE_NO_LANE_MAP, TUNE_NO_XCD_MAP, TUNE_NO_PAIR_TAIL, TUNE_NO_TWO_SLICES = 1, 2, 4, 8
TUNE_NO_THETA_MAJOR, TUNE_NO_TILE_SPLIT, TUNE_MAP_HOST_PROJECTION, TUNE_NO_JOINT = 16, 32, 64, 128
TUNE_NO_BOUND_PASS = 256
TUNE_NO_TWO_PHASE, TUNE_FORCE_TWO_PHASE, TUNE_NO_GRAPHS = 512, 1024, 2048
GROUP_FORCE_RCCL = 1


class SearchInfo(C.Structure):
    _fields_ = [("nominal_candidates", C.c_int64), ("coarse_nodes_scored", C.c_int64),
                ("fine_candidates_scored", C.c_int64), ("two_phase", C.c_int32), ("reserved", C.c_int32),
                ("blocks_scored", C.c_int64), ("blocks_skipped", C.c_int64)]


class Geometry(C.Structure):
    _fields_ = [("resolution", C.c_double), ("offset_x", C.c_double),
                ("offset_y", C.c_double)]


class Scan(C.Structure):
    _fields_ = [("angles", C.POINTER(C.c_double)),
                ("ranges", C.POINTER(C.c_double)),
                ("n_points", C.c_int32), ("reserved", C.c_int32),
                ("relative_sensor_pose", C.c_double * 3)]


class CorrelativeParams(C.Structure):
    _fields_ = [("range_x", C.c_double), ("range_y", C.c_double),
                ("range_theta", C.c_double), ("low_resolution", C.c_int32),
                ("reserved", C.c_int32), ("score_threshold", C.c_double),
                ("known_rate_threshold", C.c_double)]


class BnbParams(C.Structure):
    _fields_ = [("range_x", C.c_double), ("range_y", C.c_double),
                ("range_theta", C.c_double), ("node_height_max", C.c_int32),
                ("reserved", C.c_int32), ("score_threshold", C.c_double),
                ("known_rate_threshold", C.c_double)]


class ScanNode(C.Structure):
    _fields_ = [("global_pose", C.c_double * 3), ("scan", Scan), ("min_range", C.c_double),
                ("max_range", C.c_double)]


class MapBuilderParams(C.Structure):
    _fields_ = [("usable_range_min", C.c_double), ("usable_range_max", C.c_double),
                ("prob_hit", C.c_double), ("prob_miss", C.c_double), ("subpixel_scale", C.c_int32)]


class MapShape(C.Structure):
    _fields_ = [("resolution", C.c_double), ("offset_x", C.c_double), ("offset_y", C.c_double),
                ("rows", C.c_int32), ("cols", C.c_int32), ("log2_block_size", C.c_int32)]


class MapBuildInfo(C.Structure):
    _fields_ = [("rays", C.c_int64), ("cell_updates", C.c_int64), ("saturated_reads", C.c_int64),
                ("first_known_row", C.c_int32), ("first_known_col", C.c_int32),
                ("device_projection", C.c_int32), ("reserved", C.c_int32),
                ("host_us", C.c_double), ("device_us", C.c_double)]


class GridSearchParams(C.Structure):
    _fields_ = [("range_x", C.c_double), ("range_y", C.c_double), ("range_theta", C.c_double),
                ("step_x", C.c_double), ("step_y", C.c_double), ("step_theta", C.c_double),
                ("score_threshold", C.c_double), ("known_rate_threshold", C.c_double)]


class Result(C.Structure):
    _fields_ = [("found", C.c_int32), ("best_x", C.c_int32),
                ("best_y", C.c_int32), ("best_theta", C.c_int32),
                ("key", C.c_uint64), ("sum_values", C.c_uint32),
                ("known", C.c_uint32), ("tie_count", C.c_uint32),
                ("flags", C.c_uint32), ("score", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("pose_found", C.c_int32), ("win_x", C.c_int32),
                ("win_y", C.c_int32), ("win_theta", C.c_int32),
                ("step_x", C.c_double), ("step_y", C.c_double),
                ("step_theta", C.c_double), ("sensor_pose", C.c_double * 3),
                ("best_sensor_pose", C.c_double * 3),
                ("estimated_pose", C.c_double * 3),
                ("input_setup_us", C.c_double), ("optimization_us", C.c_double),
                ("candidates", C.c_int64), ("raw", Result)]


class RefineParams(C.Structure):
    _fields_ = [("covariance_scale", C.c_double), ("iterations_max", C.c_int32), ("reserved", C.c_int32),
                ("convergence_threshold", C.c_double), ("lambda_", C.c_double)]


class RefineResult(C.Structure):
    _fields_ = [("normalized_initial_cost", C.c_double), ("normalized_cost", C.c_double),
                ("sensor_pose", C.c_double * 3), ("best_sensor_pose", C.c_double * 3),
                ("estimated_pose", C.c_double * 3), ("covariance", C.c_double * 9),
                ("hessian", C.c_double * 9), ("lambda_", C.c_double), ("iterations", C.c_int32),
                ("reserved", C.c_int32)]


class LoopQuery(C.Structure):
    _fields_ = [("map_id", C.c_uint64), ("geometry", Geometry), ("scan", Scan),
                ("initial_pose", C.c_double * 3)]


class Window(C.Structure):
    _fields_ = [("n_theta", C.c_int32), ("n_points", C.c_int32),
                ("win_x", C.c_int32), ("win_y", C.c_int32),
                ("low_resolution", C.c_int32), ("coarse_level", C.c_int32),
                ("min_known", C.c_int32), ("merge_mode", C.c_int32),
                ("score_threshold", C.c_double)]


# name -> (restype, argtypes); mirrors include/csm_hip.h one to one
_P = C.POINTER
_ctx = C.c_void_p
SIGNATURES = {
    "csm_create": (C.c_int, [_P(Config), _P(_ctx)]),
    "csm_destroy": (C.c_int, [_ctx]),
    "csm_last_error": (C.c_char_p, [_ctx]),
    "csm_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "csm_synchronize": (C.c_int, [_ctx]),
    "csm_upload_grid": (C.c_int, [_ctx, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32]),
    "csm_upload_grid_blocks": (C.c_int, [_ctx, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "csm_has_grid": (C.c_int, [_ctx, C.c_uint64]),
    "csm_release_grid": (C.c_int, [_ctx, C.c_uint64]),
    "csm_build_pyramid": (C.c_int, [_ctx, C.c_uint64, _P(C.c_int32), C.c_int32]),
    "csm_download_level": (C.c_int, [_ctx, C.c_uint64, C.c_int32, C.c_void_p]),
    "csm_build_pyramids": (C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "csm_copy_last_batch_records": (C.c_int, [_ctx, C.c_void_p]),
    "csm_host_search_step": (C.c_int, [C.c_double, C.c_void_p, C.c_int32,
                                       _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "csm_host_window": (C.c_int, [C.c_double, C.c_double]),
    "csm_host_min_known": (C.c_int, [C.c_int32, C.c_double]),
    "csm_host_compound": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "csm_host_inverse_compound": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "csm_host_move_backward": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "csm_host_project": (C.c_int, [_P(Geometry), C.c_void_p, C.c_double, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "csm_host_probability_lut": (None, [C.c_void_p]),
    "csm_project_scan": (C.c_int, [_ctx, _P(Geometry), C.c_void_p, C.c_double, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   _P(C.c_int32)]),
    "csm_score_window": (C.c_int, [_ctx, C.c_uint64, _P(Window), C.c_void_p,
                                   C.c_void_p, _P(Result)]),
    "csm_score_window_dev": (C.c_int, [_ctx, C.c_uint64, _P(Window), C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "csm_score_windows_dev": (C.c_int, [_ctx, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "csm_score_windows_dump_dev": (C.c_int, [_ctx, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "csm_bound_pass_stats": (C.c_int, [_ctx, _P(C.c_uint64), _P(C.c_uint64)]),
    "csm_last_search_info": (C.c_int, [_ctx, C.c_void_p]),
    "csm_resolve_window_dev": (C.c_int, [_ctx, C.c_uint64, _P(Window), C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "csm_score_window_dump": (C.c_int, [_ctx, C.c_uint64, _P(Window), C.c_void_p,
                                        C.c_void_p, _P(Result), C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "csm_correlative_match": (C.c_int, [_ctx, C.c_uint64, _P(Geometry), _P(Scan),
                                        C.c_void_p, _P(CorrelativeParams), _P(Summary)]),
    "csm_bnb_match_batch": (C.c_int, [_ctx, _P(LoopQuery), C.c_int32,
                                      _P(BnbParams), _P(Summary)]),
    "csm_correlative_match_batch": (C.c_int, [_ctx, _P(LoopQuery), C.c_int32,
                                              _P(CorrelativeParams), _P(Summary)]),
    "csm_grid_search_match": (C.c_int, [_ctx, C.c_uint64, _P(Geometry), _P(Scan), C.c_void_p,
                                        _P(GridSearchParams), _P(Summary)]),
    "csm_construct_map_from_scans": (C.c_int, [_ctx, C.c_uint64, _P(MapShape), C.c_void_p, _P(ScanNode),
                                               C.c_int32, _P(MapBuilderParams), _P(MapBuildInfo)]),
    "csm_host_map_resize": (C.c_int, [_P(MapShape), C.c_void_p, C.c_int32, C.c_void_p]),
    "csm_update_map_with_scan": (C.c_int, [_ctx, C.c_uint64, _P(MapShape), C.c_void_p, _P(ScanNode),
                                           _P(MapBuilderParams), _P(MapBuildInfo)]),
    "csm_set_block_allocation": (C.c_int, [_ctx, C.c_uint64, C.c_int32, C.c_void_p]),
    "csm_cost_covariance_batch": (C.c_int, [_ctx, _P(LoopQuery), C.c_int32, C.c_void_p, C.c_double,
                                            _P(RefineResult)]),
    "csm_linear_solver_batch": (C.c_int, [_ctx, _P(LoopQuery), C.c_int32, _P(RefineParams), _P(RefineResult)]),
    "csm_shard_bounds": (None, [C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "csm_group_create": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_void_p)]),
    "csm_group_create_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, _P(C.c_void_p)]),
    "csm_group_destroy": (C.c_int, [C.c_void_p]),
    "csm_group_size": (C.c_int32, [C.c_void_p]),
    "csm_group_member": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "csm_group_last_error": (C.c_char_p, [C.c_void_p]),
    "csm_group_bnb_match_batch": (C.c_int, [C.c_void_p, _P(LoopQuery), C.c_int32, _P(BnbParams), _P(Summary)]),
    "csm_group_correlative_match_batch": (C.c_int, [C.c_void_p, _P(LoopQuery), C.c_int32,
                                                    _P(CorrelativeParams), _P(Summary)]),
    "csm_allgather_results": (C.c_int, [C.c_void_p, C.c_void_p]),
    "csm_group_gathered_records_dev": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_void_p), _P(C.c_int32)]),
    "csm_group_exchange_info": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_double)]),
    "csm_enable_kernel_timing": (C.c_int, [_ctx, C.c_int32]),
    "csm_kernel_time": (C.c_int, [_ctx, C.c_char_p, _P(C.c_double), _P(C.c_int64)]),
    "csm_reset_kernel_timing": (C.c_int, [_ctx]),
    "csm_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libcsm_hip.so and declare every entry point. Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcsm_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(expected at %s)" % LIB_PATH)
    # A process that also uses PyTorch must end up with ONE HIP runtime: torch ships its
    # own libamdhip64, and whichever of the two runtimes initialises second finds no GPU.
    # Importing torch first makes libcsm_hip.so bind to the copy torch has already loaded
    # (the configuration every GPU test and bench.py run in). Without torch installed the
    # library uses /opt/rocm's runtime, as a C++ caller does.
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
