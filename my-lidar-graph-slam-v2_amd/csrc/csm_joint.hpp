/* csm_joint.hpp -- host-side launch interface of csm_joint_kernels.hip (a translation unit
 * of its own): joint two-slice binning and the batched fine kernel that consumes it. Plain
 * arguments, HIP error codes back (-1: no kernel instantiated for the row pitch). */
#ifndef CSM_JOINT_HPP
#define CSM_JOINT_HPP

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "csm_device.hpp"

namespace csm {

struct JointLaunch {
    hipStream_t stream;
    int device;
    const ScoreJob* jobs_dev;
    dim3 grid;                  /* (candidate blocks of this launch, ceil(slices / 2), jobs) */
    size_t lds_bytes;
    int ls, R;                  /* row pitch (slots per pair row) and candidate rows per lane */
    int cbx, groups;
    const uint16_t* lane_map;
    int xcd_map;
    int row_base, cb_base, ncb; /* BlockBase: where this launch sits among the window's row blocks */
    int fp32;                   /* 1: the packed-fp32 bound pass (k_score_jointf_batch) */
    /* exact kernel over a work list (k_score_joint_list) instead of the grid: items / item_count
     * as k_bound_select wrote them, list_blocks workgroups share them */
    const uint32_t* items = nullptr;
    const uint32_t* item_count = nullptr;
    int list_blocks = 0;
};

/* slots of k_binj's hash table for n_points beams per slice (>= 4/3 * 2 * n_points, any size) and the
 * LDS bytes of k_binj for a frame of `tiles` endpoint tiles with that table */
int binj_hash_size(int n_points);
size_t binj_lds_bytes(int tiles, int n_points, int hash_size);

/* grid = (ceil(max slices / 2), jobs); BinJob.sorted_pb / sorted_rc hold 2 * n_points entries
 * per PAIR of slices, BinJob.tiles max_tiles records per pair, n_tiles one count per pair */
int launch_binj_batch(hipStream_t stream, int device, const BinJob* jobs_dev, int n_pairs_max, int n_jobs,
                      size_t lds_bytes);

int launch_joint_batch(const JointLaunch& launch);

/* One window with its jobs by value (no job array in device memory): joint binning over n_pairs pairs of
 * slices, then the exact joint kernel over launch.grid.x candidate blocks x n_pairs (launch.jobs_dev,
 * grid.y / z, fp32 and the list fields are not used). The coarse pass of a coarse-first search. */
int launch_binj_one(hipStream_t stream, int device, const BinJob& job, int n_pairs, size_t lds_bytes);
int launch_joint_one(const JointLaunch& launch, const ScoreJob& job, int n_pairs);

/* After the bound pass (approx_best of every job written): clears every job's BlockBest records and
 * lists the candidate blocks the exact kernel has to score: item = job << 18 | pair << 8 | block;
 * blocks >= split_cb go to items1 (the row block of the R = 6 launch). counts[2] must be zero.
 * round 1: as described; round 2 (two-round exact pass, ScoreJob.round1_record): the blocks not listed
 * in round 1 that can still reach the best eligible key of round 1. */
int launch_bound_select(hipStream_t stream, const ScoreJob* jobs_dev, int n_jobs, int ncb, int split_cb,
                        uint32_t* items0, uint32_t* items1, uint32_t* counts, uint32_t cap, int round);

/* xgf = the level's fp32 key copy in the layout of its pair-row copy (k_expand_pairs_f) */
int launch_expand_pairs_f(hipStream_t stream, const uint16_t* cells, int rows, int cols, int pitch, float* xgf,
                          int xg_prows, int xg_pitch, int pad);

} /* namespace csm */
#endif
