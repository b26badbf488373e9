/* csm_launch.hpp -- host-side launch interface of csm_launch.hip, the translation unit that holds the
 * per-slice kernels (csm_kernels.hip: binning, strided and pair-row scoring, box maximum, finalize,
 * exact paths, projection, grid search). csm_api.hip (planner, batch staging, C ABI) compiles without
 * device code and launches through these. Return values: a HIP error code (0 = launched), or -1 where
 * no kernel is instantiated for the requested shape. */
#ifndef CSM_LAUNCH_HPP
#define CSM_LAUNCH_HPP

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "csm_device.hpp"

namespace csm_launch {

using namespace csm;

/* which scoring kernel instantiation, and how it is launched */
struct ScoreLaunch {
    hipStream_t stream = nullptr;
    int device = 0;
    int lstride = 0, R = 0;
    int mode = 0;               /* strided kernels: 1 = power-of-two stride, 2 = any */
    bool weighted = true;
    int lists = 1;              /* pair batch kernels: 2 = two theta slices per workgroup */
    int cbx = 0, groups = 0;
    dim3 grid;
    size_t lds = 0;
    int n_buf = 1, n_slices = 1;
    int theta_major = 0, xcd_map = 0;
    const uint16_t* lane_map = nullptr;
    BlockBase bb = { 0, 0, 0 };
    int ncb = 0;                /* list launches: candidate blocks of the window */
    const uint32_t* items = nullptr;
    const uint32_t* count = nullptr;
    int blocks = 0;             /* list launches: workgroups sharing the list */
};

int score_strided(const ScoreLaunch& a, const ScoreJob& job);             /* k_score<LS, R, MODE, W> */
int score_strided_batch(const ScoreLaunch& a, const ScoreJob* jobs);      /* k_score_batch */
int score_pairs(const ScoreLaunch& a, const ScoreJob& job);               /* k_score_pairs */
int score_pairs_batch(const ScoreLaunch& a, const ScoreJob* jobs);        /* k_score_pairs_batch / pairs2_batch */
int score_pairs_list(const ScoreLaunch& a, const ScoreJob& job);          /* k_score_pairs_list */
int argmax(const ScoreLaunch& a, const ScoreJob& job);                    /* k_argmax<128, R> */

/* the other kernels: grid / block / dynamic LDS as the caller decides; *_lds variants raise the
 * kernel's dynamic-LDS limit first (process-wide table, only ever raised) */
int bin(hipStream_t s, int device, int n_theta, size_t lds, const BinJob& job);
int bin_batch(hipStream_t s, int device, int n_theta_max, int n_jobs, size_t lds, const BinJob* jobs);
int zero_if_band(hipStream_t s, int blocks, const ZeroJob& job);
int zero_if_band_batch(hipStream_t s, int blocks, int n_jobs, const ZeroJob* jobs);
int finalize(hipStream_t s, int device, size_t lds, const FinalJob& job);
int finalize_batch(hipStream_t s, int device, int n_jobs, size_t lds, const FinalJob* jobs);
int tie_replay_pick(hipStream_t s, int device, unsigned n, size_t lds, const TieJob& job);
int exact_scores(hipStream_t s, unsigned blocks, const ExactJob& job);
int literal_scan(hipStream_t s, const LiteralJob& job);
int boxmax_batch(hipStream_t s, dim3 grid, const BoxJob* jobs);
int expand_pairs(hipStream_t s, int blocks, const uint16_t* cells, int rows, int cols, int pitch, uint32_t* xg,
                 int prows, int xp, int pad);
int deblock(hipStream_t s, int blocks, const uint16_t* packed, const int32_t* slot, int log2_block, int block_cols,
            int rows, int cols, int pitch, uint16_t* cells, uint8_t* alloc, int n_blocks, int32_t* known_first);
int project(hipStream_t s, dim3 grid, const ProjJob& job);
int project_batch(hipStream_t s, dim3 grid, const ProjJob* jobs);
int grid_scores_pick(hipStream_t s, int blocks, const GridSearchJob& job);
int scatter_records(hipStream_t s, const csm_result* src, const int32_t* idx, csm_result* dst, int n);

} /* namespace csm_launch */
#endif
