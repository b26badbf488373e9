/* csm_hip.h -- C ABI of libcsm_hip.so, the MI355X (gfx950) correlative
 * scan-matching backend.
 *
 * Drop-in boundary for the correlative / branch-and-bound matching path of
 * sterngerlach/my-lidar-graph-slam-v2. Each entry point cites the reference
 * interface it replaces (paths relative to the reference tree,
 * inc/ = include/my_lidar_graph_slam/, src/ = src/my_lidar_graph_slam/).
 *
 * Conventions
 *  - plain C types only; every pointer is a host pointer unless its name ends
 *    in _dev; inputs are borrowed for the duration of the call.
 *  - every function returns 0 on success or a negative errno-style code;
 *    csm_last_error() gives the text. Nothing throws across the boundary.
 *  - one csm_ctx per matcher / detector object, single caller per ctx (the
 *    reference never shares a matcher between its two threads,
 *    src/slam_module_factory.cpp:102-104 vs src/loop_detector_factory.cpp:180-183).
 *    All mutable state lives in the ctx. The one process-wide table is a
 *    mutex-guarded record of the dynamic-LDS limit already granted to each
 *    kernel function per device (a property of the function, not of a ctx).
 *  - occupancy values are the reference's raw uint16 cells: 0 = unknown,
 *    1..65535 <-> P in [0.001, 0.999] (inc/grid_map_new/grid_binary_bayes.hpp:163-176).
 *  - there is no CPU fallback: without a GPU every compute entry point fails
 *    with CSM_ENODEV.
 *  - scans must hold finite ranges and angles ("no return" beams filtered out
 *    upstream, as the reference's scan filters do): a non-finite value makes
 *    the matching entry points fail with CSM_EINVAL.
 *  - limits: at most 10240 beams per scan (the binning kernel's tables live in LDS); LowResolution / 2^NodeHeightMax up
 *    to 64 cells; grid + window up to ~2500 x 2500 cells per map (CSM_EINVAL
 *    beyond).
 */
#ifndef CSM_HIP_H
#define CSM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSM_OK       0
#define CSM_ENOENT  (-2)   /* unknown map id / level */
#define CSM_EIO     (-5)   /* HIP runtime error */
#define CSM_ENOMEM  (-12)
#define CSM_ENODEV  (-19)  /* no usable GPU */
#define CSM_EINVAL  (-22)

/* csm_result.flags */
#define CSM_FLAG_EDGE_BAND   1u  /* a coarse read fell in the negative edge band
                                    (SURVEY 8(a) A8): resolved by the literal
                                    sequential device path */
#define CSM_FLAG_KEY_TIE     2u  /* several candidates shared the best integer
                                    key: resolved by the f64 replay */
#define CSM_FLAG_F64_TIE     4u  /* several candidates share the best f64 score
                                    bit for bit (branch-and-bound: the reference's
                                    choice then depends on heap order) */
#define CSM_FLAG_LITERAL     8u  /* result produced by the literal path */
#define CSM_FLAG_PROJ_DELTA 16u  /* branch-and-bound: some per-node projection
                                    differed from base+offset and was corrected */

typedef struct csm_ctx csm_ctx;

/* csm_config.tuning_off: switches individual launch optimisations OFF (A/B
 * measurements, parity tests of both forms). 0 = the defaults. The library never
 * reads the environment on a launch path; forced launch shapes exist only in
 * tuning builds (-DCSM_TUNING), read once in csm_create. */
#define CSM_TUNE_NO_LANE_MAP         1u   /* threads numbered through the lane groups in order */
#define CSM_TUNE_NO_XCD_MAP          2u   /* identity workgroup -> XCD order in batch launches */
#define CSM_TUNE_NO_PAIR_TAIL        4u   /* a window's last row block stays in the R = 8 launch */
#define CSM_TUNE_NO_TWO_SLICES       8u   /* batch fine kernel: one theta slice per workgroup */
#define CSM_TUNE_NO_THETA_MAJOR     16u   /* large single-window launches stay block-major */
#define CSM_TUNE_NO_TILE_SPLIT      32u   /* small single windows are never tile-split */
#define CSM_TUNE_MAP_HOST_PROJECTION 64u  /* map building: hit points computed on the host */
#define CSM_TUNE_NO_JOINT          128u   /* batch fine kernel: per-slice entry lists (round-2 form) */
#define CSM_TUNE_NO_BOUND_PASS     256u   /* no packed-fp32 bound pass: the exact integer kernel scores
                                             every candidate block */
#define CSM_TUNE_NO_TWO_PHASE      512u   /* single windows are always searched exhaustively */
#define CSM_TUNE_FORCE_TWO_PHASE  1024u   /* ... always coarse-first (default: by window size) */
#define CSM_TUNE_NO_GRAPHS        2048u   /* single queries are always launched kernel by kernel */

typedef struct {
    int32_t  device_id;          /* HIP device ordinal */
    uint32_t tuning_off;         /* CSM_TUNE_* bits */
    int32_t  map_uncertain_cap;  /* > 0: capacity of the map builder's list of uncertified
                                    beams (tests of its overflow path); 0 = default */
    int32_t  reserved[5];
} csm_config;

/* Geometry of a grid map: inc/grid_map_new/grid_map_geometry.hpp:228-240 */
typedef struct {
    double  resolution;     /* metres per cell */
    double  offset_x;       /* mPosOffset.mX */
    double  offset_y;       /* mPosOffset.mY */
} csm_geometry;

/* One scan: inc/sensor/sensor_data.hpp (ScanData<double>: Angles(), Ranges(),
 * RelativeSensorPose()) */
typedef struct {
    const double* angles;
    const double* ranges;
    int32_t       n_points;
    int32_t       reserved;
    double        relative_sensor_pose[3];
} csm_scan;

/* ScanMatcherCorrelative constructor arguments
 * (inc/mapping/scan_matcher_correlative.hpp:58-66,
 *  src/scan_matcher_factory.cpp:173-177) plus the two thresholds of the
 * 6-argument OptimizePose overload (scan_matcher_correlative.hpp:75-81) */
typedef struct {
    double  range_x, range_y, range_theta;
    int32_t low_resolution;
    int32_t reserved;
    double  score_threshold;
    double  known_rate_threshold;
} csm_correlative_params;

/* ScanMatcherBranchBound constructor arguments
 * (inc/mapping/scan_matcher_branch_bound.hpp:109-118,
 *  src/scan_matcher_factory.cpp:22-26) plus thresholds */
typedef struct {
    double  range_x, range_y, range_theta;
    int32_t node_height_max;
    int32_t reserved;
    double  score_threshold;
    double  known_rate_threshold;
} csm_bnb_params;

/* Raw search result (the device-side record; 48 bytes, also the unit of the
 * multi-GPU all-gather). Offsets are in search steps relative to the sensor
 * pose, exactly the reference's bestWinX/Y/Theta
 * (src/mapping/scan_matcher_correlative.cpp:149-152, 203-206). */
typedef struct {
    int32_t  found;
    int32_t  best_x, best_y, best_theta;
    uint64_t key;          /* 32268*K + 499*S: exact integer order of the score */
    uint32_t sum_values;   /* S: sum of raw cell values over known hit cells */
    uint32_t known;        /* K: number of known hit cells */
    uint32_t tie_count;    /* candidates sharing the best key */
    uint32_t flags;
    double   score;        /* normalized score of the winner, f64, beam order:
                              the reference's scoreMax */
} csm_result;

/* Everything ScanMatchingSummary needs from the search
 * (inc/mapping/scan_matcher.hpp:53-82) plus the metric inputs of
 * src/mapping/scan_matcher_correlative.cpp:222-236. Cost and covariance
 * (lines 209-219) stay with the caller's CostFunction. */
typedef struct {
    int32_t    pose_found;
    int32_t    win_x, win_y, win_theta;
    double     step_x, step_y, step_theta;
    double     sensor_pose[3];       /* Compound(initial, relative sensor pose) */
    double     best_sensor_pose[3];
    double     estimated_pose[3];    /* MoveBackward(best, relative sensor pose) */
    double     input_setup_us;       /* upload / pyramid time */
    double     optimization_us;
    int64_t    candidates;           /* fine candidate poses fully scored */
    csm_result raw;
} csm_summary;

/* One loop-detection query (inc/mapping/loop_detector.hpp:27-55 flattened:
 * the reference local map is named by map_id, the query scan node by scan +
 * map-local initial pose, loop_detector_branch_bound.cpp:97-104). */
typedef struct {
    uint64_t     map_id;
    csm_geometry geometry;
    csm_scan     scan;
    double       initial_pose[3];
} csm_loop_query;

/* ---- life cycle (replaces the matcher constructors; device-init failure is
 * reported like LoadBitstream does, src/slam_launcher.cpp:83-107) ---- */
int  csm_create(const csm_config* cfg, csm_ctx** out);
int  csm_destroy(csm_ctx* ctx);
const char* csm_last_error(const csm_ctx* ctx);
/* Use an existing HIP stream (hipStream_t) for all work of this ctx; NULL =
 * the ctx's own (non-blocking) stream, NOT the legacy default stream: a caller
 * that wants its work ordered with the default stream passes hipStreamLegacy /
 * hipStreamPerThread, or better a stream of its own. */
int  csm_set_stream(csm_ctx* ctx, void* hip_stream);
int  csm_synchronize(csm_ctx* ctx);

/* ---- grid maps. Replaces GridMap::CopyValues + the per-LocalMapId cache
 * (src/grid_map_new/grid_map.cpp:439-457;
 *  inc/mapping/loop_detector_branch_bound.hpp:50-69, 98;
 *  src/mapping/scan_matcher_correlative_fpga.cpp:261-262) ---- */
int  csm_upload_grid(csm_ctx* ctx, uint64_t map_id, const uint16_t* dense,
                     int32_t rows, int32_t cols);
/* The same from the reference's own storage, without the host-side flatten: GridMap<T> keeps
 * block_rows x block_cols blocks of 2^log2_block x 2^log2_block uint16, row-major inside a block,
 * each allocated or not (inc/grid_map_new/grid_map.hpp:255-263: mBlocks, mLog2BlockSize, mBlockRows,
 * mBlockCols; inc/grid_map_new/grid_binary_bayes.hpp:197-202: mValues). blocks[br * block_cols + bc]
 * points at a block's values or is NULL for an unallocated block. The library packs the allocated
 * blocks into pinned staging (one copy), de-blocks on the device into the dense level
 * GridMap::CopyValues (src/grid_map_new/grid_map.cpp:289-350, 439-457) would have produced
 * (rows = block_rows << log2_block; unallocated blocks read 0), and keeps the allocation
 * bitmap for the cost function (what csm_set_block_allocation would be given). The blocks are
 * borrowed for the duration of the call. */
int  csm_upload_grid_blocks(csm_ctx* ctx, uint64_t map_id, const uint16_t* const* blocks,
                            int32_t block_rows, int32_t block_cols, int32_t log2_block);
int  csm_has_grid(csm_ctx* ctx, uint64_t map_id);   /* 1 / 0 */
int  csm_release_grid(csm_ctx* ctx, uint64_t map_id);

/* PrecomputeGridMap(s) on device (src/mapping/grid_map_builder.cpp:987-1065):
 * level i of map_id becomes the forward box-max with window win_sizes[i]
 * (win_sizes[0] is normally 1). Level 0 always aliases the uploaded grid when
 * win_sizes[0] == 1. */
int  csm_build_pyramid(csm_ctx* ctx, uint64_t map_id, const int32_t* win_sizes,
                       int32_t n_levels);
int  csm_download_level(csm_ctx* ctx, uint64_t map_id, int32_t level,
                        uint16_t* out /* rows*cols */);
/* PrecomputeGridMaps for many finished local maps at once (what
 * LoopDetectorBranchBound::Detect does lazily per query,
 * src/mapping/loop_detector_branch_bound.cpp:83-89): makes sure every map of
 * map_ids holds box-max(win_sizes[l]) for every l, building what is missing.
 * Levels that exist are kept (unlike csm_build_pyramid, which rebuilds).
 * Asynchronous on the ctx stream. */
int  csm_build_pyramids(csm_ctx* ctx, const uint64_t* map_ids, int32_t n_maps,
                        const int32_t* win_sizes, int32_t n_levels);

/* ---- host-side set-up pieces (pure CPU, exported so the adapter and the
 * tests share one implementation) ---- */
/* ComputeSearchStep: src/mapping/scan_matcher_correlative.cpp:255-274 */
int  csm_host_search_step(double resolution, const double* ranges, int32_t n,
                          double* step_x, double* step_y, double* step_theta);
/* Window half-widths: scan_matcher_correlative.cpp:141-146 */
int  csm_host_window(double range, double step);
/* Smallest known count K with double(K)/double(n) > known_rate_threshold
 * (the test of scan_matcher_correlative.cpp:181-182 as an integer bound) */
int  csm_host_min_known(int32_t n_points, double known_rate_threshold);
/* Compound / MoveBackward: inc/pose.hpp:154-166, 215-227 */
void csm_host_compound(const double start[3], const double diff[3], double out[3]);
void csm_host_inverse_compound(const double start[3], const double end[3], double out[3]);
void csm_host_move_backward(const double end[3], const double diff[3], double out[3]);
/* ComputeScanIndices for n_theta slices t = -win_theta..win_theta
 * (scan_matcher_correlative.cpp:161-168, 277-297): out arrays are
 * [n_theta][n_points]; also returns r*cos / r*sin per slice if non-NULL
 * (needed by the branch-and-bound projection). */
int  csm_host_project(const csm_geometry* geom, const double sensor_pose[3],
                      double step_theta, int32_t win_theta,
                      const double* angles, const double* ranges, int32_t n,
                      int32_t* hit_col, int32_t* hit_row,
                      double* r_cos, double* r_sin);
/* The same projection as the matchers run it: on the device, with a
 * per-entry certificate. hit_col / hit_row [2*win_theta+1][n] (host) receive
 * the device's indices; `uncertified` receives the flat indices t*n + i of the
 * entries whose cell coordinate lies too close to a cell edge for the device's
 * sin / cos to be trusted (at most uncertified_cap of them; *n_uncertified is
 * the full count). Contract checked by the tests: every entry NOT listed
 * equals csm_host_project's (glibc) index. The matchers recompute the listed
 * ones on the host. Non-finite ranges or angles are rejected (CSM_EINVAL);
 * the reference's upstream filters drop them before the matcher sees a scan. */
int  csm_project_scan(csm_ctx* ctx, const csm_geometry* geom, const double sensor_pose[3],
                      double step_theta, int32_t win_theta,
                      const double* angles, const double* ranges, int32_t n,
                      int32_t* hit_col, int32_t* hit_row, uint32_t* uncertified,
                      int32_t uncertified_cap, int32_t* n_uncertified);
/* value -> probability table, 65536 doubles
 * (inc/grid_map_new/grid_values.hpp:26-35) */
void csm_host_probability_lut(double* lut);

/* ---- the hot path ---- */

/* Pre-projected search window, CSM flavour: replaces the theta/x/y sweep of
 * src/mapping/scan_matcher_correlative.cpp:161-197 + 339-368 for one query. */
typedef struct {
    int32_t n_theta;          /* 2*win_theta+1 */
    int32_t n_points;
    int32_t win_x, win_y;
    int32_t low_resolution;   /* L; level `coarse_level` must be box-max(L) */
    int32_t coarse_level;
    int32_t min_known;        /* csm_host_min_known() */
    int32_t merge_mode;       /* 0: merge beams that land on the same cell into
                                 weighted entries (default); 1: one entry per beam
                                 (better when beams rarely share cells) */
    double  score_threshold;
} csm_window;

/* hit_col / hit_row: [n_theta][n_points] int32, host memory. */
int  csm_score_window(csm_ctx* ctx, uint64_t map_id, const csm_window* w,
                      const int32_t* hit_col, const int32_t* hit_row,
                      csm_result* out);
/* Same with device-resident inputs and output (asynchronous on the ctx
 * stream; no host synchronisation unless a tie / edge-band path is needed,
 * in which case out_dev->flags tells and csm_resolve_window_dev() finishes). */
int  csm_score_window_dev(csm_ctx* ctx, uint64_t map_id, const csm_window* w,
                          const int32_t* hit_col_dev, const int32_t* hit_row_dev,
                          csm_result* out_dev);
/* csm_score_window_dev for n independent windows in one launch chain (the
 * batched kernels of the loop detectors): windows[i] on map_ids[i] with the
 * device-resident hit indices hit_col_dev[i] / hit_row_dev[i], result i in
 * out_dev[i] (device). Asynchronous; a record that carries a key-tie or
 * edge-band flag is finished by scoring that window again with
 * csm_score_window_dev() + csm_resolve_window_dev(). Coarse levels as for the
 * single call (csm_build_pyramid; windows[i].coarse_level). */
int  csm_score_windows_dev(csm_ctx* ctx, int32_t n, const uint64_t* map_ids,
                           const csm_window* windows, const int32_t* const* hit_col_dev,
                           const int32_t* const* hit_row_dev, csm_result* out_dev);
/* The same launch chain with dumps for parity tests of the batched kernels, each a
 * device pointer per window or NULL (per window or for a whole array):
 * dump_s_dev[i] / dump_k_dev[i]: every candidate's integer sums S [n_theta][nx][ny]
 * uint32 and K [..] uint16 (nx = ceil((2*win_x+1)/L)*L); a window with either set is
 * scored by the exact kernel on every candidate block. dump_f_dev[i]: every
 * candidate's fp32 order key from the bound pass, [n_theta][nx][ny] float (written
 * only when the bound pass runs for the window's group). */
int  csm_score_windows_dump_dev(csm_ctx* ctx, int32_t n, const uint64_t* map_ids,
                                const csm_window* windows, const int32_t* const* hit_col_dev,
                                const int32_t* const* hit_row_dev, csm_result* out_dev,
                                uint32_t* const* dump_s_dev, uint16_t* const* dump_k_dev,
                                float* const* dump_f_dev);
/* What the last single-window search (csm_correlative_match) evaluated. Large windows are searched
 * coarse-first (scan_matcher_correlative.cpp:176-192 is the reference's pruning): every
 * coarse node is scored, then the fine level only on the candidate blocks that hold an eligible
 * coarse node reaching the best fine score found under the best coarse node. SURVEY 8(d) asks for
 * the EVALUATED poses of a pruned search next to the nominal window. */
typedef struct {
    int64_t nominal_candidates;        /* (2 win_theta + 1) * X_ext * Y_ext */
    int64_t coarse_nodes_scored;       /* 0: exhaustive search, no coarse pass needed */
    int64_t fine_candidates_scored;    /* candidates of the fine blocks scored (whole blocks) */
    int32_t two_phase;                 /* 1: coarse-first */
    int32_t reserved;
    int64_t blocks_scored, blocks_skipped;
} csm_search_info;
int  csm_last_search_info(csm_ctx* ctx, csm_search_info* out);
/* Two-pass fine level of the batch entries: candidate blocks the exact
 * integer kernel scored / skipped after the packed-fp32 bound pass since the last call
 * (synchronises the context's stream; resets the counters). */
int  csm_bound_pass_stats(csm_ctx* ctx, uint64_t* blocks_scored, uint64_t* blocks_skipped);
/* Synchronises, reads *out_dev and, when it carries a key tie or an edge-band
 * flag, runs the exact device paths (f64 tie replay / literal sequential
 * sweep) for the window just scored with csm_score_window_dev(); no-op
 * otherwise. csm_score_window() does this itself. */
int  csm_resolve_window_dev(csm_ctx* ctx, uint64_t map_id, const csm_window* w,
                            const int32_t* hit_col_dev, const int32_t* hit_row_dev,
                            csm_result* out_dev);
/* Optional dump of every candidate's integer sums for parity tests:
 * S [n_theta][nx][ny] uint32 and K [..] uint16 (nx = ceil((2*win_x+1)/L)*L),
 * coarse K [n_theta][nx/L][ny/L]. Host pointers, any may be NULL. */
int  csm_score_window_dump(csm_ctx* ctx, uint64_t map_id, const csm_window* w,
                           const int32_t* hit_col, const int32_t* hit_row,
                           csm_result* out, uint32_t* dump_s, uint16_t* dump_k,
                           uint16_t* dump_coarse_k);

/* ScanMatcherCorrelative::OptimizePose, both overloads
 * (src/mapping/scan_matcher_correlative.cpp:92-115, 118-244): uploads nothing;
 * the grid must be resident under map_id; builds box-max(L) if missing. */
int  csm_correlative_match(csm_ctx* ctx, uint64_t map_id,
                           const csm_geometry* geom, const csm_scan* scan,
                           const double initial_pose[3],
                           const csm_correlative_params* params,
                           csm_summary* out);

/* LoopDetectorBranchBound::Detect's search part for a batch of queries
 * (src/mapping/loop_detector_branch_bound.cpp:59-156 lines 68-108;
 *  ScanMatcherBranchBound::OptimizePose, scan_matcher_branch_bound.cpp:111-278).
 * Every map_id must be resident; pyramids are built and cached on first use.
 * out[i] corresponds to queries[i] (pose_found = 0 when the reference would
 * `continue`). */
int  csm_bnb_match_batch(csm_ctx* ctx, const csm_loop_query* queries,
                         int32_t n_queries, const csm_bnb_params* params,
                         csm_summary* out);

/* LoopDetectorCorrelative::Detect's search part for a batch of queries
 * (src/mapping/loop_detector_correlative.cpp:59-156, lines 68-108): the
 * correlative matcher with the detector's thresholds against resident maps,
 * one coarse map (box-max L) cached per map id. out[i] <-> queries[i]. */
int  csm_correlative_match_batch(csm_ctx* ctx, const csm_loop_query* queries,
                                 int32_t n_queries, const csm_correlative_params* params,
                                 csm_summary* out);

/* The raw records (csm_summary.raw) of the last csm_bnb_match_batch /
 * csm_correlative_match_batch call on this ctx, in query order, copied device
 * to device into dst_dev[n_queries] on the ctx stream (asynchronous): the
 * send buffer of the multi-GPU all-gather without a host round trip
 * (the concatenation of src/mapping/loop_detector_fpga_parallel.cpp:53-56). */
int  csm_copy_last_batch_records(csm_ctx* ctx, csm_result* dst_dev);

/* ---- the step after every search: cost, covariance, sub-cell refinement ----
 * CostSquareError (inc/mapping/cost_function_square_error.hpp;
 * src/mapping/cost_function_square_error.cpp:48-195, 232-341: bilinear
 * interpolation of the four nearest cells, squared error to 1, Gauss-Newton
 * Hessian, covariance = Hessian^-1 * CovarianceScale) and
 * ScanMatcherLinearSolver::OptimizePose
 * (src/mapping/scan_matcher_linear_solver.cpp:66-169: damped Gauss-Newton
 * steps, lambda halved / doubled between 1e-8 and 1e-4), which every matcher
 * and loop detector runs on the pose the search returns
 * (scan_matcher_correlative.cpp:209-219, loop_detector_branch_bound.cpp:123-127).
 * Batched: one launch for all queries of a Detect() call.
 *
 * TOLERANCE (f64, not bit-exact): hit points come from the device's sin / cos,
 * sums over the beams are tree reductions, and Eigen's 3x3 inverse / column-
 * pivoting QR are restated from their algorithms. Against the reference's
 * arithmetic: costs and Hessian entries agree to 1e-10 relative; covariance
 * entries to 1e-8 relative to the largest entry; a refined pose to 1e-7 (m,
 * rad) when both sides run the same number of iterations -- the count can
 * differ when |cost change| lies within 1e-10 of ConvergenceThreshold, or when
 * a hit point sits within ~1e-12 cells of a cell edge (the bilinear value is
 * continuous there, its gradient is not). The tests compare at these bounds.
 *
 * Map reads are GridMap::ProbabilityOr(row, col, 0.5)
 * (src/grid_map_new/grid_map.cpp:423-436): 0.5 outside the map or in a block
 * that was never allocated, the cell's probability (0 for unknown) otherwise.
 * Allocation is not part of the dense export: pass it with
 * csm_set_block_allocation. Without it a block counts as allocated iff it
 * holds a known cell -- exact for maps that were only ever updated (finished
 * local maps: every update leaves a value >= 1, grid_map.cpp:514-535), not for
 * a map that was cleared with ResetValues() (the frontend's latest map). */
typedef struct {
    double  covariance_scale;       /* CostSquareError: "CovarianceScale" */
    int32_t iterations_max;         /* "NumOfIterationsMax" */
    int32_t reserved;
    double  convergence_threshold;  /* "ConvergenceThreshold" */
    double  lambda;                 /* the solver object's damping factor when the call starts
                                       ("InitialLambda" on its first call); every query of a batch
                                       starts from it (the reference carries it from query to query:
                                       a change of <= 1e-4 on diagonals of 1e2 and more) */
} csm_refine_params;

typedef struct {
    double  normalized_initial_cost;   /* Cost(start) / n_points */
    double  normalized_cost;           /* Cost(final) / n_points: ScanMatchingSummary::mNormalizedCost */
    double  sensor_pose[3];            /* where the evaluation started */
    double  best_sensor_pose[3];
    double  estimated_pose[3];         /* MoveBackward(best sensor pose, relative sensor pose) */
    double  covariance[9];             /* row-major, map-local: mEstimatedCovariance */
    double  hessian[9];                /* at the final pose, undamped */
    double  lambda;                    /* damping factor after the call */
    int32_t iterations;
    int32_t reserved;
} csm_refine_result;

/* allocated: one byte per block (non-zero = allocated), row-major
 * [ceil(rows / 2^log2)][ceil(cols / 2^log2)]: GridMap::IsAllocated of any cell
 * of the block. Null: back to the rule above with this block size. Must be
 * repeated after the map is uploaded again. */
int  csm_set_block_allocation(csm_ctx* ctx, uint64_t map_id, int32_t log2_block_size,
                              const uint8_t* allocated);
/* Cost / n and ComputeCovariance at sensor_poses[i] (3 doubles per query: the
 * best sensor pose of the search), scan and map of queries[i] (initial_pose is
 * not read). */
int  csm_cost_covariance_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n_queries,
                               const double* sensor_poses, double covariance_scale,
                               csm_refine_result* out);
/* ScanMatcherLinearSolver::OptimizePose for every query: queries[i].initial_pose
 * is the map-local ROBOT pose to refine (the search's mEstimatedPose). */
int  csm_linear_solver_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n_queries,
                             const csm_refine_params* params, csm_refine_result* out);

/* ---- several GPUs behind one detector object, in one process ----
 * The reference's precedent is LoopDetectorFPGAParallel
 * (src/mapping/loop_detector_fpga_parallel.cpp:42-56): Detect() cuts the query
 * vector into contiguous halves, runs one std::thread per FPGA core and
 * concatenates the per-core results. A csm_group owns one csm_ctx per listed
 * device; its batch calls cut the queries into contiguous blocks
 * (csm_shard_bounds), run one host thread per member on its block and finish
 * with ONE exchange of the 48-byte best records (csm_allgather_results). */
typedef struct csm_group csm_group;

/* Block [*lo, *hi) of `member`: sizes differ by at most one, earlier members
 * take the larger blocks (n_members = 2: the first half / second half split of
 * loop_detector_fpga_parallel.cpp:42-46). Pure host arithmetic. */
void csm_shard_bounds(int32_t n_queries, int32_t member, int32_t n_members,
                      int32_t* lo, int32_t* hi);
/* One context per entry of device_ids (CSM_ENODEV if one cannot be opened,
 * like LoadBitstream's failure, src/slam_launcher.cpp:83-107). Listing a
 * device twice gives two members on it (two streams): meant for tests on a
 * one-GPU box. */
int  csm_group_create(const int32_t* device_ids, int32_t n_devices, csm_group** out);
/* The same with the members' configuration (member_cfg->device_id is ignored; NULL =
 * defaults) and CSM_GROUP_* flags. */
#define CSM_GROUP_FORCE_RCCL 1u  /* take the RCCL exchange also for a single member (a one-rank
                                    communicator: the call sequence on a one-GPU box) */
int  csm_group_create_ex(const int32_t* device_ids, int32_t n_devices, const csm_config* member_cfg,
                         uint32_t flags, csm_group** out);
int  csm_group_destroy(csm_group* group);
int32_t csm_group_size(const csm_group* group);
/* The member's context: upload / release the maps of the queries its block will
 * hold through it (csm_upload_grid, csm_has_grid, ...). Owned by the group. */
csm_ctx* csm_group_member(csm_group* group, int32_t member);
const char* csm_group_last_error(const csm_group* group);

/* csm_bnb_match_batch / csm_correlative_match_batch over the group: queries
 * [lo_k, hi_k) go to member k, whose context must hold their maps; out[i] <->
 * queries[i]. Ends with csm_allgather_results; out[i].raw is the record that
 * came back through the exchange. */
int  csm_group_bnb_match_batch(csm_group* group, const csm_loop_query* queries, int32_t n_queries,
                               const csm_bnb_params* params, csm_summary* out);
int  csm_group_correlative_match_batch(csm_group* group, const csm_loop_query* queries,
                                       int32_t n_queries, const csm_correlative_params* params,
                                       csm_summary* out);
/* The exchange step of the last group batch (where the reference concatenates
 * two vectors, src/mapping/loop_detector_fpga_parallel.cpp:53-56): every
 * member's block of device-resident records, padded to the largest block, is
 * all-gathered so that EVERY member's device buffer holds all records.
 * Members on distinct devices: ncclAllGather (RCCL over xGMI; librccl is
 * loaded on first use). One member, or members sharing a device: copies
 * through the host. host_out (may be null) receives the n_queries records in
 * query order, read back from member 0's gathered buffer. */
int  csm_allgather_results(csm_group* group, csm_result* host_out);
/* Member `member`'s gathered device buffer after csm_allgather_results:
 * n_members blocks of *block records each (block k = member k's queries, the
 * tail of a shorter block zero). Valid until the next group batch. */
int  csm_group_gathered_records_dev(csm_group* group, int32_t member,
                                    const csm_result** dev, int32_t* block);
/* used_rccl: 1 if the exchange runs over RCCL; last_gather_us: host wall time
 * of the last csm_allgather_results. */
int  csm_group_exchange_info(const csm_group* group, int32_t* used_rccl, double* last_gather_us);

/* ScanMatcherGridSearch constructor arguments + thresholds
 * (inc/mapping/scan_matcher_grid_search.hpp, src/scan_matcher_factory.cpp) */
typedef struct {
    double range_x, range_y, range_theta;
    double step_x, step_y, step_theta;
    double score_threshold;
    double known_rate_threshold;
} csm_grid_search_params;

/* ScanMatcherGridSearch::OptimizePose, both overloads
 * (src/mapping/scan_matcher_grid_search.cpp:69-190): brute force over
 * accumulated-double offsets, every pose projected on its own
 * (ScorePixelAccurate::Score), a pose counts only if its own known rate passes.
 * In the summary win_x / win_y / win_theta hold the number of x / y / theta
 * offsets and raw.best_x / best_y / best_theta the winner's loop indices
 * (-1 when nothing was found). */
int  csm_grid_search_match(csm_ctx* ctx, uint64_t map_id, const csm_geometry* geom,
                           const csm_scan* scan, const double initial_pose[3],
                           const csm_grid_search_params* params, csm_summary* out);

/* ---- map building: the producer of the matcher's input ---- */

/* One scan node of the window [scanNodeIdMin, scanNodeIdMax]
 * (ScanNode, inc/mapping/pose_graph.hpp; ScanData, inc/sensor/sensor_data.hpp) */
typedef struct {
    double   global_pose[3];      /* ScanNode::mGlobalPose */
    csm_scan scan;
    double   min_range, max_range;   /* ScanData::MinRange / MaxRange */
} csm_scan_node;

/* GridMapBuilder constructor arguments that the map update reads
 * (src/mapping/grid_map_builder.cpp:68-99) + its SubpixelScale constant
 * (inc/mapping/grid_map_builder.hpp:294) */
typedef struct {
    double  usable_range_min, usable_range_max;
    double  prob_hit, prob_miss;
    int32_t subpixel_scale;
} csm_map_builder_params;

/* GridMapGeometry + block size of a GridMap (inc/grid_map_new/grid_map.hpp).
 * In: the map BEFORE the call (Resize works in its frame). Out: after it. */
typedef struct {
    double  resolution, offset_x, offset_y;
    int32_t rows, cols;
    int32_t log2_block_size;
} csm_map_shape;

typedef struct {
    int64_t rays;               /* usable beams integrated */
    int64_t cell_updates;       /* hit + miss updates applied */
    int64_t saturated_reads;    /* updates of a cell already at 65535: the reference's odds table
                                   (grid_values.cpp:74-77) has no such entry (undefined behaviour
                                   there); this library extends the table's formula */
    int32_t first_known_row, first_known_col;
    int32_t device_projection;  /* 1: hit points computed on the device under a certificate (the few
                                   uncertifiable beams redone with glibc); 0: all on the host */
    int32_t reserved;
    double  host_us;            /* poses, projection round trip, resize */
    double  device_us;          /* upload + kernels of the update */
} csm_map_build_info;

/* Host only. GridMap<T>::Resize(BoundingBox<int>) (src/grid_map_new/grid_map.cpp:841-889)
 * on `shape`: box = { min col, min row, max col, max row } (inclusive cell
 * indices in the map's current frame), blocks per IndexToBlock (:804-814),
 * offsets per GridMapGeometry::Resize (grid_map_geometry.cpp:61-72). With
 * expand != 0 GridMap<T>::Expand (:915-936) runs first: nothing changes if the
 * box fits, else it is joined with the current extent. shift_out (may be null)
 * = first row, first column of the new map in the old frame. */
int  csm_host_map_resize(csm_map_shape* shape, const int32_t box[4], int32_t expand,
                         int32_t shift_out[2]);

/* GridMapBuilder::ConstructMapFromScans (src/mapping/grid_map_builder.cpp:561-695)
 * = UpdateLatestMap's work (:497-527): bounding box of the sensor positions and
 * usable hit points, GridMap::Resize + ResetValues, then a sub-pixel ray cast
 * per beam (BresenhamScaled, src/bresenham.cpp:58-237) with binary-Bayes miss /
 * hit updates in beam order. The finished map becomes (or replaces) the
 * resident grid `map_id`, ready for the matchers; csm_download_level(level 0)
 * returns it as CopyValues would. `info` may be null. */
int  csm_construct_map_from_scans(csm_ctx* ctx, uint64_t map_id, csm_map_shape* shape,
                                  const double global_map_pose[3], const csm_scan_node* nodes,
                                  int32_t n_nodes, const csm_map_builder_params* params,
                                  csm_map_build_info* info);

/* The grid half of GridMapBuilder::UpdateGridMap
 * (src/mapping/grid_map_builder.cpp:389-494): one new scan into the local map
 * that is being built. The bounding box starts at the sensor position
 * (ComputeBoundingBoxAndScanPointsMapLocal, :820-872); GridMap::Expand
 * (grid_map.cpp:915-961) leaves the map alone if the box fits and otherwise
 * resizes it to the union with its current extent, keeping the cells; then the
 * same ray casts as above on top of the existing values. `map_id` must be
 * resident (uploaded, or built by these calls) with the rows / cols of `shape`. */
int  csm_update_map_with_scan(csm_ctx* ctx, uint64_t map_id, csm_map_shape* shape,
                              const double global_map_pose[3], const csm_scan_node* node,
                              const csm_map_builder_params* params, csm_map_build_info* info);

/* ---- measurement hooks (bench.py) ---- */
/* enable = 1: every kernel launch is bracketed by HIP events on the ctx
 * stream; enable = 2: only the dominant (fine-level) scoring kernel, to keep
 * the timed region undisturbed; 0: off. */
int  csm_enable_kernel_timing(csm_ctx* ctx, int32_t enable);
/* Drains recorded events; returns total ms and launch count since the last
 * reset for kernel "score_fine" | "score_coarse" | "bin" | "finalize" | "boxmax". */
int  csm_kernel_time(csm_ctx* ctx, const char* name, double* total_ms,
                     int64_t* launches);
int  csm_reset_kernel_timing(csm_ctx* ctx);

/* Library / build identification */
const char* csm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CSM_HIP_H */
