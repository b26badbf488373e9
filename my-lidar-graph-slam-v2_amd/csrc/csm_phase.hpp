/* csm_phase.hpp -- host-side launch interface of csm_phase_kernels.hip: the pieces of the
 * two-phase (coarse-first) search of one large window around the scoring kernels. */
#ifndef CSM_PHASE_HPP
#define CSM_PHASE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csm_device.hpp"

namespace csm {

/* Phase-major copy of a box-max(L) level: block (a, b), a, b in [0, L), of hp x wp cells holds
 * D[pad + R][pad + C] = level[L R + a][L C + b] (0 where that lies outside the level or in the
 * padding), at rows a * hp .., columns b * wp .. of a (L hp) x (L wp) grid, row pitch `pitch`. */
int launch_phase_map(hipStream_t stream, const uint16_t* level, int rows, int cols, int level_pitch, int L,
                     int hp, int wp, int pad, uint16_t* out, int pitch);

/* Hit indices into the phase-major copy: beam (row, col) read by coarse candidate (xc, yc) at
 * level[row + y_lo + L yc][col + x_lo + L xc] reads phase map cell (row' + yc - win, col' + xc - win). */
int launch_phase_hits(hipStream_t stream, const int32_t* col, const int32_t* row, size_t n, int x_lo, int y_lo,
                      int L, int hp, int wp, int pad, int rows_c, int cols_c, int win_x, int win_y, int32_t* col_out,
                      int32_t* row_out);

struct TwoPhaseJob {
    /* coarse level sums [n_theta][nxs][nys] (nxs >= nxc: the phase pass may carry one extra column / row) */
    const uint32_t* coarse_s;
    const uint32_t* coarse_k;
    int32_t n_theta, nxc, nyc, nxs, nys, L, min_known;
    /* fine level */
    const uint16_t* cells;
    int32_t rows, cols, pitch;
    const int32_t* hit_col;      /* [n_theta][n_points] */
    const int32_t* hit_row;
    int32_t n_points, x_lo, y_lo;
    /* fine launch geometry: candidate blocks of cbx x cby, ncbx blocks per row of blocks */
    int32_t nx, ny, cbx, cby, ncbx, ncb;
    const uint32_t* flags;       /* query flags: kFlagBandTouch switches the pruning off */
    /* outputs */
    unsigned long long* best;    /* [2]: packed best eligible coarse node; best fine key under it */
    uint32_t* items;             /* work list of the fine launch: slice << 12 | block */
    uint32_t* count;             /* [1]: items = blocks kept */
    unsigned char* keep;         /* [n_theta * ncb] scratch of the marking, zero on entry */
    uint32_t cap;
};

/* best[0] = max over eligible coarse nodes of key << 26 | (2^26 - 1 - node index); needs best[0..1] = 0 */
int launch_coarse_best(hipStream_t stream, const TwoPhaseJob& job);
/* best[1] = greatest fine key among the L x L candidates under that node (exact integer sums) */
int launch_fine_under_best(hipStream_t stream, const TwoPhaseJob& job);
/* the fine blocks holding an eligible coarse node whose key reaches best[1] -> items / count (job.keep zeroed) */
int launch_mark_blocks(hipStream_t stream, const TwoPhaseJob& job);
/* the BlockBest records of the listed blocks (block_best[slice * ncb + block]) reduced to kReducedBest records */
constexpr int kReducedBest = 64;
int launch_reduce_items(hipStream_t stream, const BlockBest* block_best, const uint32_t* items, const uint32_t* count,
                        uint32_t cap, int ncb, BlockBest* out);

} /* namespace csm */
#endif
