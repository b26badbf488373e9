/* csm_device.hpp -- device-side data structures shared by the kernels and the
 * host launcher of libcsm_hip (gfx950 only). */
#ifndef CSM_DEVICE_HPP
#define CSM_DEVICE_HPP

#include <stdint.h>

namespace csm {

constexpr int kTile = 64;        /* endpoint tile edge, cells */
constexpr int kMaxRegionRows = 128;        /* LDS region rows of a stride-1 job */
constexpr int kMaxRegionRowsStrided = 128; /* ... of a strided (coarser level) job */
constexpr int kPairMaxCby = 56;  /* candidate rows per workgroup of the pair-row fine kernel */
constexpr int kPbMax = 1024;       /* entries per TileRec: k_bin splits fuller tiles */
constexpr int kJRec = 256;         /* entries per TileRec of a joint list (k_binj): the record's float beam counts
                                      (16 B per entry) live in LDS during the fp32 bound pass */
constexpr int kMaxMult = 15;       /* beams merged into one (cell, multiplicity) entry */
constexpr int kMaxPoints = 10240;  /* beams per scan: k_bin's hash table (16384 slots, load <= 2/3, 128 KB)
                                      and cell list must fit the CU's 160 KB of LDS */
constexpr int kBlock = 512;      /* threads per workgroup (8 wave64) */
constexpr int kBinBlock = 256;   /* threads per workgroup of the binning kernel */
constexpr int kMaxElig = 8;      /* eligibility levels per scoring job */
/* internal flag bit (never returned): some beam can reach the negative edge
 * band of a coarser level for some candidate offset */
constexpr uint32_t kFlagBandTouch = 1u << 16;

constexpr int kBoxTR = 32, kBoxTC = 64, kBoxMaxWin = 64;    /* k_boxmax_batch: output tile, largest window */
constexpr int kProjSlices = 256;     /* k_project: theta slices per workgroup (LDS table) */
constexpr int kBinDebugRows = 65536; /* -DCSM_BIN_TIMING builds: rows of k_bin's phase counters */

/* Where a pair-kernel launch sits among the candidate blocks of its window: a window's last
 * row block may be a launch of its own with fewer rows per lane (R = 6 for 36 rows instead of
 * R = 8 with a quarter of the lanes' rows outside the window: launch_score_batch). */
struct BlockBase {
    int row_base;       /* first candidate row of this launch's row block 0 */
    int cb_base;        /* this launch's block 0 in the window's numbering (BlockBest slots) */
    int ncb;            /* candidate blocks of the window, all launches */
};

/* One non-empty endpoint tile of one theta slice. */
struct TileRec {
    int32_t  r0, c0;     /* grid row / col of the first cell of the tile's
                            bounding box around its beams */
    uint32_t start;      /* first beam in the slice's sorted list */
    uint32_t count;
    int32_t  h, w;       /* bounding box extent, cells (<= kTile) */
    int32_t  pad[2];     /* [0]: entries of class "both rows" | "even row only" << 16 in this
                            record (the rest: "odd row only"); single mode: 0 | count << 16.
                            [1]: tile number << 4 | chunk of the tile (records of one slice
                            ascend in it: slices are merged tile by tile on it) */
};

static_assert(sizeof(TileRec) == 32, "copied into LDS as two uint4");

/* Best candidate of one workgroup (or of a reduction of several). */
struct BlockBest {
    unsigned long long key;    /* 0 = no eligible candidate */
    unsigned long long rank;   /* traversal rank of the first best candidate */
    uint32_t count;            /* candidates sharing `key` */
    uint32_t pad;
};

/* Known-count array of a coarser level: K[t][xi / div][yi / div]. */
struct EligLevel {
    const uint32_t* k;
    const uint32_t* s;         /* raw-value sums of the same nodes (bound check) */
    int32_t div;
    int32_t nxc, nyc;
};

/* Bin the beams of every theta slice by endpoint tile. */
struct BinJob {
    const int32_t* hit_col;    /* [n_theta][n_points] */
    const int32_t* hit_row;
    uint32_t* sorted_pb;       /* [n_theta][n_points] packed LDS offsets */
    uint32_t* sorted_rc;       /* optional: same order, (row << 16 | col) inside the
                                  tile's bounding box, for strided jobs */
    TileRec*  tiles;           /* [n_theta][max_tiles] */
    int32_t*  n_tiles;         /* [n_theta] */
    uint32_t* flags;           /* [1] CSM_FLAG_* accumulated with atomicOr */
    uint32_t* reserved_ptr;    /* unused (round 1 cleared the coarse accumulators here) */
    uint32_t* tuning_counters; /* CSM_BIN_TIMING builds: per-workgroup phase cycles; else null */
    int32_t   reserved_words;
    int32_t n_theta, n_points, max_tiles;
    int32_t rows, cols;
    int32_t x_lo, y_lo;        /* most negative candidate offset */
    int32_t x_hi, y_hi;        /* most positive candidate offset */
    int32_t tiles_x, tiles_y;
    int32_t hash_size;         /* slots of the LDS hash table: k_bin a power of two >= 3/2 n_points, k_binj
                                  binj_hash_size() (any size) */
    int32_t max_mult;          /* kMaxMult: merge same-cell beams; 1: one entry per beam */
    int32_t lstride;
    int32_t pair_mode;         /* 1: entries are aligned row pairs (pair-row fine kernel); 2: joint
                                  entries of two slices (k_binj: lists and records per pair of slices) */
    int32_t frame_shift;       /* 0 / 1 added to the tile frame's row origin: (y_lo + y_hi + shift) even,
                                  so that frame parity == grid-row parity of what the fine kernel reads */
    /* first row / column of the map that holds a known cell: a box that ends
     * before it is unknown on every level, so reading it as unknown is right */
    int32_t known_r0, known_c0;
    /* edge-band detection: coarse strides to test (box windows) */
    int32_t n_band;
    int32_t band_win[kMaxElig];
    int32_t band_nx[kMaxElig], band_ny[kMaxElig];
};

/* k_boxmax_batch: dst = box maximum (win x win, tail rule) of src; same rows / cols / pitch */
struct BoxJob {
    const uint16_t* src;
    uint16_t* dst;
    int32_t rows, cols, pitch, win;
};

/* k_zero_if_band: clear a[0..words) (and b) when the query's band flag is set */
struct ZeroJob {
    uint32_t* a;
    uint32_t* b;
    size_t words;
    const uint32_t* flags;
    int32_t always;
    int32_t pad;
};

/* Score every candidate of one level of one query. */
struct ScoreJob {
    const uint16_t* cells;     /* pitched grid level */
    int32_t rows, cols, pitch;
    /* pair-row fine kernel: the level as expanded cells v + (v != 0) << 23 in 8-byte
     * slots (row 2k, row 2k + 1) of one column, zero-padded by xg_pad cells on
     * every side, xg_pitch slots per pair row (k_expand_pairs) */
    const uint32_t* xg;
    int32_t xg_pitch, xg_pad;
    int32_t joint;             /* entry lists, records and record counts are per PAIR of theta slices
                                  (k_binj, csm_joint_kernels.hip): index t / 2, 2 n_points entries per
                                  pair, four beam counts per entry */
    const uint32_t* sorted_pb;
    const TileRec*  tiles;
    const int32_t*  n_tiles;
    int32_t n_theta, n_points, max_tiles;
    int32_t x_lo, y_lo;        /* cell offset of candidate index 0 */
    int32_t nx, ny;            /* candidates per axis */
    int32_t stride;            /* cells between neighbouring candidates (power of two
                                  for the strided kernel) */
    int32_t log2_stride;
    /* outputs (any may be null) */
    uint32_t*  dump_s;         /* [n_theta][nx][ny] */
    uint16_t*  dump_k;         /* [n_theta][nx][ny] */
    uint32_t*  acc_s;          /* [n_theta][nx][ny] atomic accumulate (tile-split launches) */
    uint32_t*  acc_k;
    int32_t    acc_x_major;    /* acc arrays laid out [n_theta][ny][nx] (coalesced atomics);
                                  2: plain stores instead of atomic adds, [n_theta][nx][ny] (a launch in
                                  which one lane computes a candidate's whole sum: the coarse pass of the
                                  two-phase search) */
    uint32_t* in_s;            /* [n_theta][ny][nx]; non-null: no gathering, the sums are read from here (the
                                  arg-max pass after a tile-split launch) */
    uint32_t* in_k;            /* (the pass clears what it has read) */
    BlockBest* block_best;     /* [n_theta][n_cand_blocks] */
    /* tie collection pass: append the rank of every eligible candidate whose
     * key equals *collect_key */
    const unsigned long long* collect_key;
    unsigned long long* tie_list;
    uint32_t*  tie_count;
    uint32_t   tie_cap;
    uint32_t*  flags;          /* [1] query flags (band touch in, edge band out) */
    /* eligibility for the argmax */
    int32_t n_elig;
    int32_t min_known;
    int32_t check_own_known;   /* also require the candidate's own K >= min_known */
    int32_t rank_l;            /* L of the traversal rank (1: plain t,x,y order) */
    /* min_known <= 1 only: when no beam can reach the negative edge band the
     * coarser level bounds every candidate by construction and passes the
     * known test whenever the candidate's own K >= 1, so a coarse job with
     * skip_unless_band exits at once and a fine job with elig_only_if_band
     * ignores its eligibility levels. */
    int32_t skip_unless_band;
    int32_t elig_only_if_band;
    EligLevel elig[kMaxElig];
    /* fp32 bound pass of the joint fine level (csm_joint_kernels.hip): every candidate's key
     * 32268 K + 499 S evaluated in packed fp32 from `xgf`, a copy of the level in the layout of
     * `xg` holding float(499 v + 32268 (v != 0)) per cell. The pass writes the greatest value of
     * each (slice, candidate block) to approx_best [n_theta][blocks]; the exact integer kernel
     * then skips the blocks that provably cannot hold the winner (approx_slack: relative margin
     * 4 (N + 3) 2^-24 covering both passes' rounding). Null / 0: no bound pass. */
    const float* xgf;
    float* approx_best;
    float* dump_f;             /* optional [n_theta][nx][ny]: every candidate's fp32 key (tests) */
    float approx_slack;
    int32_t pad1;
    uint32_t* bound_stats;     /* optional [2]: blocks the exact kernel scored / skipped after the bound pass */
    /* fp32 key below which no candidate can be reported as found (score threshold as a key, margin
     * included; 0: none): blocks below it are never scored */
    float key_floor;
    int32_t pad2;
    /* two-round exact pass (branch and bound: the winner must pass its own known-count test, so the
     * window's greatest fp32 key may belong to a leaf that does not count): the query's record after
     * round 1 (csm_result*), whose key is the best ELIGIBLE exact key so far */
    const void* round1_record;
};

/* Reduce block results, replay the winner in f64, write the result record. */
struct FinalJob {
    const BlockBest* block_best;
    int32_t n_entries;
    int32_t nx, ny, rank_l;
    int32_t x_lo, y_lo, win_theta;
    int32_t init_x, init_y, init_theta;   /* reported when nothing is found */
    const uint16_t* cells;
    int32_t rows, cols, pitch;
    const int32_t* hit_col;
    const int32_t* hit_row;
    int32_t n_points;
    double score_thr;
    const double* lut;
    const uint32_t* flags_in;
    uint32_t* flags_clear;     /* optional: the flag word of the NEXT query, cleared here */
    void* out;                 /* csm_result* (device) */
};

/* Device-side projection with a certificate. The device evaluates
 * ScanData::HitPoint + PositionToIndex with its own sin / cos; an entry is
 * trusted only where the cell coordinate is farther from a cell edge than a
 * bound on |host value - device value| (both libms are accurate to a few ulp),
 * so the floor is provably the one glibc would give. The few entries that
 * cannot be certified are listed for the host to recompute. */
struct ProjJob {
    const double* angles;      /* [n_points] */
    const double* ranges;
    int32_t* hit_col;          /* [n_theta][n_points] */
    int32_t* hit_row;
    uint32_t* unc_count;       /* [1] */
    uint32_t* unc_list;        /* [unc_cap] flat index t * n_points + i */
    uint32_t  unc_cap;
    int32_t n_theta, n_points, win_theta;
    double sensor_x, sensor_y, sensor_theta, step_theta;
    double off_x, off_y, res;
    /* branch and bound: also certify every per-node projection
     * floor((sensor + x*step + r*cos - off) / res) == base + x */
    int32_t check_nodes;
    int32_t flag_uncertain;    /* set CSM_FLAG_PROJ_DELTA in *flags instead of listing */
    int32_t x_lo, y_lo, nx, ny;
    double step_x, step_y;
    uint32_t* flags;           /* CSM_FLAG_PROJ_DELTA when a node cannot be certified */
};

/* Exhaustive f64 scores (beam order) of one level; literal / tie paths. */
struct ExactJob {
    const uint16_t* cells;
    int32_t rows, cols, pitch;
    const int32_t* hit_col;    /* integer-offset projection (CSM) */
    const int32_t* hit_row;
    const double* r_cos;       /* per-node projection (branch and bound) when non-null */
    const double* r_sin;
    double sensor_x, sensor_y, step_x, step_y, off_x, off_y, res;
    int32_t n_theta, n_points;
    int32_t x_lo, y_lo, nx, ny, stride;
    const double* lut;
    double*   out_score;       /* [n_theta][nx][ny] normalized score */
    uint32_t* out_k;           /* [n_theta][nx][ny] known count */
};

/* f64 replay of a list of tied candidates + pick (CSM order). */
struct TieJob {
    const unsigned long long* tie_list;
    const uint32_t* tie_count;
    uint32_t tie_cap;
    double* tie_score;         /* [tie_cap] */
    int32_t nx, ny, rank_l, x_lo, y_lo, win_theta;
    const uint16_t* cells;
    int32_t rows, cols, pitch;
    const int32_t* hit_col;
    const int32_t* hit_row;
    int32_t n_points;
    double score_thr;
    const double* lut;
    void* out;                 /* csm_result*, updated in place */
};

/* Literal sequential sweep of ScanMatcherCorrelative over precomputed exact
 * scores (src/mapping/scan_matcher_correlative.cpp:161-197, 339-368). */
struct LiteralJob {
    const double*   coarse_score;  /* [n_theta][nxc][nyc] */
    const uint32_t* coarse_k;
    const double*   fine_score;    /* [n_theta][nx][ny] */
    int32_t n_theta, nxc, nyc, L;
    int32_t x_lo, y_lo, win_theta;
    int32_t min_known;
    double score_thr;
    void* out;                     /* csm_result* */
};

/* Brute-force grid search (ScanMatcherGridSearch): every pose is projected on
 * its own in double, from host-computed r*cos / r*sin per theta value. */
struct GridSearchJob {
    const uint16_t* cells;
    int32_t rows, cols, pitch;
    const double* px;          /* [nx] sensor x + dx (accumulated doubles, host) */
    const double* py;          /* [ny] */
    const double* r_cos;       /* [nt][n_points] */
    const double* r_sin;
    double off_x, off_y, res;
    int32_t nx, ny, nt, n_points;
    int32_t min_known;
    double score_thr;
    const double* lut;
    double*   out_score;       /* [ny][nx][nt] traversal order */
    uint32_t* out_k;
    unsigned long long* best_bits;   /* [1] max score (as ordered bits) among eligible poses */
    unsigned long long* best_index;  /* [1] first pose index reaching it */
};

/* Map building from scans (GridMapBuilder::ConstructMapFromScans). A ray is one
 * usable beam of one scan node, numbered in the reference's update order. */
struct MapRay {
    double  hx, hy;            /* hit point, map-local */
    int32_t node;              /* its scan node */
    int32_t usable;            /* range inside (min, max): grid_map_builder.cpp:618-619 */
};
struct MapNode {
    double  x, y, theta;       /* sensor pose, map-local (host) */
    double  min_range, max_range;
    int32_t beam_base, n_beams;
    int32_t sx, sy;            /* sub-pixel index of the sensor position (set after the resize) */
};
struct MapRayRec {
    int32_t ex, ey;            /* sub-pixel index of the hit point */
    int32_t hit_cell;          /* row * cols + col, or -1 if the ray was rejected */
    int32_t slot;              /* arrival number among the hits of that cell */
};
enum MapCounter {
    kMapCursor = 0, kMapError, kMapSaturatedReads, kMapUpdates, kMapKnownRow, kMapKnownCol,
    kMapHitCells,
    /* update / saturation totals are striped over kMapStripes words each: same-address
     * atomics serialise at the memory side (~100 per microsecond) */
    kMapStripedUpdates, kMapStripedSaturated = kMapStripedUpdates + 64,
    kMapCounters = kMapStripedSaturated + 64
};
/* words from a hit cell's block start to its misses_between[] (16-byte aligned) */
__host__ __device__ inline uint32_t map_between_offset(uint32_t n) { return (2u * n + 3u) & ~3u; }
__host__ __device__ inline uint32_t map_block_words(uint32_t n)
{
    return map_between_offset(n) + ((n + 1u + 3u) & ~3u);
}

/* hit points on the device with a certificate (see k_map_project) */
struct MapProjJob {
    const double* angles;      /* all nodes' beams, concatenated */
    const double* ranges;
    const MapNode* nodes;
    int32_t n_nodes, n_beams;
    MapRay* rays;
    double off_x, off_y, res, scaled_res;    /* the map's frame BEFORE the resize */
    int32_t* box;              /* [4] min / max of floor((h -+ res - off) / res) over certified beams */
    uint32_t* unc_count;
    uint32_t* unc_list;
    uint32_t unc_cap;
};

constexpr int kMapStripes = 64;
struct MapJob {
    const MapRay* rays;
    const MapNode* nodes;
    MapRayRec* recs;
    int32_t n_rays;
    double  off_x, off_y, res, scaled_res;
    int32_t scale;
    int32_t rows, cols, pitch;
    uint32_t* n_hit;           /* [rows * cols] hits ending in the cell */
    uint32_t* n_miss;          /* [rows * cols] misses of cells no ray ends in */
    uint32_t* seg;             /* [rows * cols] start of the cell's block in `lists` */
    uint32_t* lists;           /* per cell with n hits: arrival[n] sorted[n] (padded to 4 words)
                                  misses_between[n + 1] (padded to 4 words); see map_between_offset */
    uint32_t* hit_cells;       /* [<= n_rays] the cells with hits, any order */
    unsigned long long* counters;   /* [kMapCounters] */
    const uint16_t* lut_hit;   /* value -> value after one hit / miss update */
    const uint16_t* lut_miss;
    uint16_t* cells;           /* output grid, rows * pitch */
    int32_t keep_cells;        /* 1: the updates go on top of the cells' values (UpdateGridMap) */
};

/* cost / covariance / linear-solver refinement (csm_cost_kernels.hip) */
struct CostOut {
    double initial_cost, cost;           /* sums over the beams, not normalized */
    double best_sensor_pose[3];
    double hessian[9], residual[3];      /* at the final pose, without the damping term */
    double covariance[9];
    double lambda;
    int32_t iterations, pad;
};
struct CostJob {
    const uint16_t* cells;
    int32_t rows, cols, pitch;
    int32_t log2_block, block_cols;
    const uint8_t* alloc;                /* one byte per block, null = all allocated */
    double res, off_x, off_y;
    const double* angles;
    const double* ranges;
    int32_t n, iterations_max;
    double sensor_pose[3];
    double convergence_threshold, lambda, covariance_scale;
    const double* lut;
    CostOut* out;
};

} /* namespace csm */
#endif
