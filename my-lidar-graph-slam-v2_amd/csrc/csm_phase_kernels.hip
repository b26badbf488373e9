/* csm_phase_kernels.hip -- two-phase (coarse-first) search of ONE large window (BASELINE configs[4]:
 * 9.3e8 fine candidates): the reference prunes with its coarse level as it sweeps
 * (src/mapping/scan_matcher_correlative.cpp:176-192: a coarse node whose score does not beat the
 * running maximum is skipped with its L x L fine candidates); a GPU has no running maximum, but the
 * same bound prunes in two passes:
 *   1. score EVERY coarse node (1 / L^2 of the fine work) -- with the fast fine kernel: a coarse
 *      candidate (xc, yc) reads box-max level cell (row + y_lo + L yc, col + x_lo + L xc), i.e. cell
 *      (R + yc, C + xc) of the level decimated by L at phase ((row + y_lo) mod L, (col + x_lo) mod L).
 *      k_phase_map lays the L x L decimated copies side by side in one grid, k_phase_hits moves every
 *      beam to its phase's copy, and the stride-1 pair kernel scores the nxc x nyc window on it.
 *      Reads at negative level indices give 0 exactly as the reference's lookup does (the negative
 *      edge band, SURVEY 8(a) A8, is reproduced, not avoided);
 *   2. the best eligible coarse node's own L x L fine candidates give a fine key F that the final
 *      winner can only exceed or equal; a fine candidate can reach F only under an eligible coarse
 *      node with key >= F (box maximum bounds every cell it covers), so
 *   3. the exact fine kernel runs on the candidate blocks that hold such a node (k_mark_blocks ->
 *      work list -> k_score_pairs_list) with its usual epilogue (eligibility, bound check, ties).
 * A beam that can reach the negative edge band voids the bound: then every block is kept. */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csm_phase.hpp"
#include "csm_score_common.hpp"

namespace csm {

__global__ __launch_bounds__(256) void k_phase_map(const uint16_t* __restrict__ level, int rows, int cols,
                                                  int level_pitch, int L, int hp, int wp, int pad,
                                                  uint16_t* __restrict__ out, int pitch)
{
    const size_t total = (size_t)L * hp * pitch;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int rr = (int)(i / pitch), cc = (int)(i % pitch);
        uint16_t v = 0;
        if (cc < L * wp) {
            const int a = rr / hp, R = rr - a * hp - pad;
            const int b = cc / wp, C = cc - b * wp - pad;
            const int r = L * R + a, c = L * C + b;
            if (R >= 0 && C >= 0 && r < rows && c < cols)
                v = level[(size_t)r * level_pitch + c];
        }
        out[i] = v;
    }
}

__device__ __forceinline__ int floor_div_i(int a, int b)
{
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0)))
        --q;
    return q;
}

__global__ __launch_bounds__(256) void k_phase_hits(const int32_t* __restrict__ col, const int32_t* __restrict__ row,
                                                   size_t n, int x_lo, int y_lo, int L, int hp, int wp, int pad,
                                                   int rows_c, int cols_c, int win_x, int win_y,
                                                   int32_t* __restrict__ col_out, int32_t* __restrict__ row_out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int u = row[i] + y_lo, v = col[i] + x_lo;
        int R = floor_div_i(u, L), C = floor_div_i(v, L);
        const int a = u - R * L, b = v - C * L;
        /* a beam whose every read falls outside the level reads zeros: keep it inside its block's
         * padding (pad cells of zeros before and after the rows_c x cols_c decimated cells) */
        R = max(-pad, min(R, rows_c));
        C = max(-pad, min(C, cols_c));
        row_out[i] = a * hp + pad + R + win_y;
        col_out[i] = b * wp + pad + C + win_x;
    }
}

__global__ __launch_bounds__(256) void k_coarse_best(TwoPhaseJob job)
{
    __shared__ unsigned long long red[4];
    const size_t per_t = (size_t)job.nxs * job.nys;
    const size_t total = per_t * job.n_theta;
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xc = (int)((i % per_t) / job.nys), yc = (int)(i % job.nys);
        if (xc >= job.nxc || yc >= job.nyc)
            continue;
        const uint32_t k = job.coarse_k[i];
        if ((int)k < job.min_known || k == 0)
            continue;
        const unsigned long long key = 32268ull * k + 499ull * (unsigned long long)job.coarse_s[i];
        const unsigned long long packed = (key << 26) | (unsigned long long)(((1u << 26) - 1u) - (uint32_t)min(i, (size_t)((1u << 26) - 1u)));
        best = best > packed ? best : packed;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(best >> 32), d, 64) << 32) |
                                     __shfl_xor((uint32_t)best, d, 64);
        best = best > o ? best : o;
    }
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            best = best > red[w] ? best : red[w];
        if (best)
            atomicMax(job.best, best);
    }
}

/* one workgroup of 16 waves: the L x L fine candidates under the best coarse node, exact integer keys */
__global__ __launch_bounds__(1024) void k_fine_under_best(TwoPhaseJob job)
{
    __shared__ unsigned long long red[16];
    const unsigned long long packed = job.best[0];
    unsigned long long fbest = 0;
    if (packed) {
        const size_t node = (size_t)(((1u << 26) - 1u) - (uint32_t)(packed & ((1ull << 26) - 1ull)));
        const size_t per_t = (size_t)job.nxs * job.nys;
        const int t = (int)(node / per_t);
        const int xc = (int)((node % per_t) / job.nys), yc = (int)(node % job.nys);
        const int32_t* col = job.hit_col + (size_t)t * job.n_points;
        const int32_t* row = job.hit_row + (size_t)t * job.n_points;
        const int L = job.L;
        /* a wave per fine candidate, lanes over beams */
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int f = wave; f < L * L; f += 16) {
            const int x = job.x_lo + xc * L + f / L, y = job.y_lo + yc * L + f % L;
            unsigned long long s = 0;
            uint32_t k = 0;
            for (int i = lane; i < job.n_points; i += 64) {
                const int r = row[i] + y, c = col[i] + x;
                if (r >= 0 && r < job.rows && c >= 0 && c < job.cols) {
                    const uint32_t v = job.cells[(size_t)r * job.pitch + c];
                    s += v;
                    k += v != 0;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                s += ((unsigned long long)__shfl_xor((uint32_t)(s >> 32), d, 64) << 32) | __shfl_xor((uint32_t)s, d, 64);
                k += __shfl_xor(k, d, 64);
            }
            const unsigned long long key = 32268ull * k + 499ull * s;
            fbest = fbest > key ? fbest : key;
        }
    }
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = fbest;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            fbest = fbest > red[w] ? fbest : red[w];
        job.best[1] = fbest;
    }
}

/* The fine blocks that can still win, in two steps. k_mark_nodes: one thread per coarse node in memory
 * order (466 MB of sums for configs[4], read once and coalesced; one wave per block reading its 21 x 12
 * nodes in 48-byte runs took 0.33 ms); a node that is eligible with key >= F sets the flag byte of every
 * fine block one of its L x L candidates lies in. k_compact_blocks: the flagged blocks (all of them when a
 * beam can reach the edge band) become the work list, one atomic per wave. */
__global__ __launch_bounds__(256) void k_mark_nodes(TwoPhaseJob job)
{
    const unsigned long long F = job.best[1];
    if (F == 0 || (*job.flags & kFlagBandTouch) != 0)
        return;
    const size_t per_t = (size_t)job.nxs * job.nys;
    const size_t total = per_t * job.n_theta;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t k = job.coarse_k[i];
        if ((int)k < job.min_known || k == 0)
            continue;
        if (32268ull * k + 499ull * (unsigned long long)job.coarse_s[i] < F)
            continue;
        const int t = (int)(i / per_t);
        const int xc = (int)((i % per_t) / job.nys), yc = (int)(i % job.nys);
        if (xc >= job.nxc || yc >= job.nyc)
            continue;
        const int bx0 = (xc * job.L) / job.cbx, bx1 = min(job.nx - 1, xc * job.L + job.L - 1) / job.cbx;
        const int by0 = (yc * job.L) / job.cby, by1 = min(job.ny - 1, yc * job.L + job.L - 1) / job.cby;
        for (int by = by0; by <= by1; ++by)
            for (int bx = bx0; bx <= bx1; ++bx)
                job.keep[(size_t)t * job.ncb + by * job.ncbx + bx] = 1;
    }
}

__global__ __launch_bounds__(256) void k_compact_blocks(TwoPhaseJob job)
{
    const int total = job.n_theta * job.ncb;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool all = (*job.flags & kFlagBandTouch) != 0;
    const bool keep = i < total && (all || job.keep[i] != 0);
    const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
    if (!km)
        return;
    uint32_t base = 0;
    if (lane == 0)
        base = atomicAdd(job.count, (uint32_t)__popcll(km));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (keep) {
        const uint32_t pos = base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
        const int t = i / job.ncb, cb = i - t * job.ncb;
        if (pos < job.cap)
            job.items[pos] = ((uint32_t)t << 12) | (uint32_t)cb;
    }
}

/* The BlockBest records of the work list's blocks reduced to kReducedBest records (k_finalize then reads
 * those instead of one record per block of the window: 262k for configs[4], a single workgroup's loop). */
__global__ __launch_bounds__(256) void k_reduce_items(const BlockBest* __restrict__ bb, const uint32_t* __restrict__ items,
                                                     const uint32_t* __restrict__ count, uint32_t cap, int ncb,
                                                     BlockBest* __restrict__ out)
{
    __shared__ unsigned long long red_key[4], red_rank[4];
    __shared__ uint32_t red_cnt[4];
    const uint32_t n = min(*count, cap);
    unsigned long long key = 0, rank = ~0ull;
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t it = items[i];
        const BlockBest b = bb[(size_t)(it >> 12) * ncb + (it & 4095u)];
        best_combine(key, rank, cnt, b.key, b.rank, b.count);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long k2 = shfl_xor_u64(key, m), r2 = shfl_xor_u64(rank, m);
        const uint32_t c2 = __shfl_xor(cnt, m, 64);
        best_combine(key, rank, cnt, k2, r2, c2);
    }
    if ((threadIdx.x & 63) == 0) {
        red_key[threadIdx.x >> 6] = key;
        red_rank[threadIdx.x >> 6] = rank;
        red_cnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            best_combine(key, rank, cnt, red_key[w], red_rank[w], red_cnt[w]);
        BlockBest r;
        r.key = key;
        r.rank = rank;
        r.count = cnt;
        r.pad = 0;
        out[blockIdx.x] = r;
    }
}

int launch_phase_map(hipStream_t stream, const uint16_t* level, int rows, int cols, int level_pitch, int L,
                     int hp, int wp, int pad, uint16_t* out, int pitch)
{
    const size_t total = (size_t)L * hp * pitch;
    const int blocks = (int)(total + 255 < (size_t)8192 * 256 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_phase_map, dim3(blocks), dim3(256), 0, stream, level, rows, cols, level_pitch, L, hp, wp,
                       pad, out, pitch);
    return (int)hipGetLastError();
}

int launch_phase_hits(hipStream_t stream, const int32_t* col, const int32_t* row, size_t n, int x_lo, int y_lo,
                      int L, int hp, int wp, int pad, int rows_c, int cols_c, int win_x, int win_y, int32_t* col_out,
                      int32_t* row_out)
{
    const int blocks = (int)(n + 255 < (size_t)4096 * 256 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_phase_hits, dim3(blocks), dim3(256), 0, stream, col, row, n, x_lo, y_lo, L, hp, wp, pad,
                       rows_c, cols_c, win_x, win_y, col_out, row_out);
    return (int)hipGetLastError();
}

int launch_coarse_best(hipStream_t stream, const TwoPhaseJob& job)
{
    const size_t total = (size_t)job.n_theta * job.nxs * job.nys;
    const int blocks = (int)(total + 255 < (size_t)2048 * 256 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_coarse_best, dim3(blocks), dim3(256), 0, stream, job);
    return (int)hipGetLastError();
}

int launch_fine_under_best(hipStream_t stream, const TwoPhaseJob& job)
{
    hipLaunchKernelGGL(k_fine_under_best, dim3(1), dim3(1024), 0, stream, job);
    return (int)hipGetLastError();
}

int launch_mark_blocks(hipStream_t stream, const TwoPhaseJob& job)
{
    const int total = job.n_theta * job.ncb;
    const size_t nodes = (size_t)job.n_theta * job.nxs * job.nys;
    const int blocks = (int)(nodes + 255 < (size_t)4096 * 256 ? (nodes + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_mark_nodes, dim3(blocks), dim3(256), 0, stream, job);
    hipLaunchKernelGGL(k_compact_blocks, dim3((total + 255) / 256), dim3(256), 0, stream, job);
    return (int)hipGetLastError();
}

int launch_reduce_items(hipStream_t stream, const BlockBest* block_best, const uint32_t* items, const uint32_t* count,
                        uint32_t cap, int ncb, BlockBest* out)
{
    hipLaunchKernelGGL(k_reduce_items, dim3(kReducedBest), dim3(256), 0, stream, block_best, items, count, cap, ncb, out);
    return (int)hipGetLastError();
}

} /* namespace csm */
