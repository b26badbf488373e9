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
};

/* LDS bytes of k_binj for a frame of `tiles` endpoint tiles, n_points beams per slice and a
 * hash table of hash_size slots (a power of two >= 4/3 * 2 * n_points) */
size_t binj_lds_bytes(int tiles, int n_points, int hash_size);

/* grid = (ceil(max slices / 2), jobs); BinJob.sorted_pb / sorted_rc hold 2 * n_points entries
 * per PAIR of slices, BinJob.tiles max_tiles records per pair, n_tiles one count per pair */
int launch_binj_batch(hipStream_t stream, int device, const BinJob* jobs_dev, int n_pairs_max, int n_jobs,
                      size_t lds_bytes);

int launch_joint_batch(const JointLaunch& launch);

/* xgf = the level's fp32 key copy in the layout of its pair-row copy (k_expand_pairs_f) */
int launch_expand_pairs_f(hipStream_t stream, const uint16_t* cells, int rows, int cols, int pitch, float* xgf,
                          int xg_prows, int xg_pitch, int pad);

} /* namespace csm */
#endif
