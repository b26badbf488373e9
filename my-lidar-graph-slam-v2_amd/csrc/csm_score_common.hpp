/* csm_score_common.hpp -- device functions shared by the scoring kernels of
 * csm_kernels.hip and csm_joint_kernels.hip (separate translation units of libcsm_hip.so):
 * the workgroup epilogue (eligibility, bound check, wave64 arg-max), the hand-issued LDS
 * reads, the flush bookkeeping of the packed accumulators and the XCD-aware block order. */
#ifndef CSM_SCORE_COMMON_HPP
#define CSM_SCORE_COMMON_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csm_device.hpp"
#include "../../include/csm_hip.h"

namespace csm {

__device__ __forceinline__ int floor_div(int a, int b)
{
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0)))
        --q;
    return q;
}

/* Is there k in [0, nk) with -(w-1) <= v = u + k*w <= -1 (a coarse read in the
 * negative edge band) whose box [v, v + w) reaches the first known row /
 * column `known_lo` of the map? Only then can the coarse level read "unknown"
 * where the box maximum is known. */
__device__ __forceinline__ bool band_hit(int u, int w, int nk, int known_lo)
{
    const int k0 = floor_div(-u, w);
    if (k0 < 0 || k0 >= nk)
        return false;
    const int v = u + k0 * w;
    return v <= -1 && v >= -(w - 1) && v + w - 1 >= known_lo;
}

/* acc + cell * mult as ONE v_mad_u32_u24. Left to itself the compiler pairs two
 * gathers into v_mul, v_mul, v_add3 (1.5 VALU per gather instead of 1). */
__device__ __forceinline__ uint32_t mad_u24(uint32_t cell, uint32_t mult, uint32_t acc)
{
    uint32_t out;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(out) : "v"(cell), "s"(mult), "v"(acc));
    return out;
}


__device__ __forceinline__ void best_combine(unsigned long long& key,
                                             unsigned long long& rank,
                                             uint32_t& count,
                                             unsigned long long k2,
                                             unsigned long long r2,
                                             uint32_t c2)
{
    if (k2 > key) {
        key = k2;
        rank = r2;
        count = c2;
    } else if (k2 == key) {
        rank = r2 < rank ? r2 : rank;
        count += c2;
    }
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

/* Shared tail of the scoring kernels: dumps, tile-split accumulation,
 * eligibility against the coarser levels, bound check, wave64 arg-max, one
 * record per workgroup. S / K: this lane's exact sums for its R candidates
 * (rows by * cby + g * R + r, column bx * cbx + dxi). */
template <int R>
__device__ __forceinline__ void score_epilogue(const ScoreJob& job, uint32_t (&S)[R], uint32_t (&K)[R],
                                               int t, int bx, int by, int cbx, int cby, int g, int dxi,
                                               bool lane_on, uint32_t qflags, int cb, int ncb)
{   /* cb / ncb: this workgroup's candidate block and the blocks per slice (the record's slot) */
    __shared__ unsigned long long red_key[kBlock / 64];
    __shared__ unsigned long long red_rank[kBlock / 64];
    __shared__ uint32_t red_cnt[kBlock / 64];
    const int tid = threadIdx.x;
    const int xi = bx * cbx + dxi;
    if (job.in_s && lane_on && xi < job.nx) {
        /* arg-max pass of a tile-split launch: the slices' sums are complete */
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yi = by * cby + g * R + r;
            if (yi < job.ny) {
                const size_t ai = ((size_t)t * job.ny + yi) * job.nx + xi;
                S[r] = job.in_s[ai];
                K[r] = job.in_k[ai];
                /* leave the accumulators clean for the next query */
                job.in_s[ai] = 0;
                job.in_k[ai] = 0;
            }
        }
    }
    unsigned long long bkey = 0, brank = ~0ull;
    uint32_t bcnt = 0;
    bool bound_broken = false;
    /* every job field the candidate loop needs, read once (`job` lives in global
     * memory and the stores below may alias it as far as the compiler knows) */
    BlockBest* const block_best = job.block_best;
    unsigned long long* const tie_list = job.tie_list;
    uint32_t* const dump_s = job.dump_s;
    uint16_t* const dump_k = job.dump_k;
    uint32_t* const acc_s = job.acc_s;
    uint32_t* const acc_k = job.acc_k;
    const int nx = job.nx, ny = job.ny, n_elig = job.n_elig, min_known = job.min_known;
    const bool acc_x_major = job.acc_x_major == 1;
    const bool acc_store = job.acc_x_major == 2;
    const bool check_known = job.check_own_known != 0;
    const bool use_elig = !job.elig_only_if_band || (qflags & kFlagBandTouch);
    const bool band_touch = (block_best || tie_list) && n_elig > 0 && (*job.flags & kFlagBandTouch) != 0;
    const unsigned long long collect_key = tie_list ? *job.collect_key : 0ull;
    if (lane_on && xi < nx) {
        /* the traversal rank of candidate (t, xi, yi): the divisions by the coarse
         * stride are done once per lane, the lane's R consecutive rows step the
         * quotient and remainder (eight candidates x four integer divisions by a
         * run-time divisor were a seventh of the kernel's vector instructions) */
        const int L = max(job.rank_l, 1);
        const int nxc = nx / L, nyc = ny / L;
        const int xq = xi / L, xm = xi - xq * L;
        const int y_first = by * cby + g * R;
        int yq = y_first / L, ym = y_first - yq * L - 1;
        const unsigned long long rank_x = ((unsigned long long)t * nxc + xq) * nyc;
        int best_yq = -1, best_ym = 0;      /* the lane's first best row: ranks grow with the row */
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yi = y_first + r;
            if (++ym == L) {
                ym = 0;
                ++yq;
            }
            if (yi >= ny)
                continue;
            const size_t ci = ((size_t)t * nx + xi) * ny + yi;
            if (dump_s)
                dump_s[ci] = S[r];
            if (dump_k)
                dump_k[ci] = (uint16_t)K[r];
            if (acc_s) {
                /* tile-split launch: slices add their partial integer sums.
                 * acc_x_major: consecutive lanes (dx) hit consecutive words */
                if (acc_store) {
                    acc_s[ci] = S[r];
                    acc_k[ci] = K[r];
                } else {
                    const size_t ai = acc_x_major ? ((size_t)t * ny + yi) * nx + xi : ci;
                    if (S[r])
                        atomicAdd(&acc_s[ai], S[r]);
                    if (K[r])
                        atomicAdd(&acc_k[ai], K[r]);
                }
            }
            if (!block_best && !tie_list)
                continue;
            bool ok = !(check_known || !use_elig) || (int)K[r] >= min_known;
            const unsigned long long key =
                32268ull * K[r] + 499ull * (unsigned long long)S[r];
            bool broken = false;
            for (int e = 0; e < n_elig && ok && use_elig; ++e) {
                const EligLevel& el = job.elig[e];
                const size_t ni =
                    ((size_t)t * el.nxc + xi / el.div) * el.nyc + yi / el.div;
                const uint32_t ck = el.k[ni];
                ok = (int)ck >= min_known;
                /* the coarser node must bound this candidate; it can fail to
                 * only through the negative edge band */
                const unsigned long long ckey =
                    32268ull * ck + 499ull * (unsigned long long)el.s[ni];
                broken |= key > ckey || (key == ckey && band_touch);
            }
            if (!ok)
                continue;
            bound_broken |= broken;
            if (key == 0)
                continue;
            if (tie_list && key != collect_key)
                continue;
            if (tie_list) {
                const unsigned long long rank = (((rank_x + yq) * L + xm) * L) + ym;
                const uint32_t pos = atomicAdd(job.tie_count, 1u);
                if (pos < job.tie_cap)
                    tie_list[pos] = rank;
                continue;
            }
            if (key > bkey) {
                bkey = key;
                bcnt = 1;
                best_yq = yq;
                best_ym = ym;
            } else if (key == bkey) {
                ++bcnt;
            }
        }
        if (best_yq >= 0)
            brank = (((rank_x + best_yq) * L + xm) * L) + best_ym;
    }
    if (!job.block_best)
        return;
    if (bound_broken)
        atomicOr(job.flags, CSM_FLAG_EDGE_BAND);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long k2 = shfl_xor_u64(bkey, m);
        const unsigned long long r2 = shfl_xor_u64(brank, m);
        const uint32_t c2 = __shfl_xor(bcnt, m, 64);
        best_combine(bkey, brank, bcnt, k2, r2, c2);
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        red_key[wave] = bkey;
        red_rank[wave] = brank;
        red_cnt[wave] = bcnt;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBlock / 64; ++w)
            best_combine(bkey, brank, bcnt, red_key[w], red_rank[w], red_cnt[w]);
        BlockBest bb;
        bb.key = bkey;
        bb.rank = brank;
        bb.count = bcnt;
        bb.pad = 0;
        job.block_best[(size_t)t * ncb + cb] = bb;
    }
}


/* ds_read_b64 with an immediate byte offset, issued by hand. The compiler does
 * not know that the result arrives later: every use must sit behind lds_wait,
 * which takes the registers as in/out operands so that nothing that reads them
 * can be scheduled above the wait. */
/* the 32-bit LDS address of a __shared__ object, for hand-written ds_* instructions */
__device__ __forceinline__ uint32_t lds_address(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

template <int OFFSET>
__device__ __forceinline__ void lds_read_b64(uint32_t addr, unsigned long long& q)
{
    static_assert(OFFSET >= 0 && OFFSET < 65536, "ds offset field");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "n"(OFFSET));
}

/* s_waitcnt lgkmcnt(LATER): returns once all LDS reads but the LATER youngest
 * have landed (LDS returns in order; an interleaved scalar load or a read the
 * compiler issued only makes the wait longer, never shorter). */
template <int LATER, int N>
__device__ __forceinline__ void lds_wait(unsigned long long (&q)[N])
{
    static_assert(N == 2 || N == 3 || N == 4 || N == 5, "registers to tie");
    if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]) : "n"(LATER));
    else if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(LATER));
    else if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) : "n"(LATER));
    else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(q[0]), "+v"(q[1]) : "n"(LATER));
}



/* Inclusive prefix sum over the wave's 64 lanes, data-parallel primitives only (no LDS
 * traffic: the gather counts its own LDS reads in lgkmcnt). */
__device__ __forceinline__ int wave_prefix_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);       /* row_shr:1 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);       /* row_shr:2 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);       /* row_shr:4 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);       /* row_shr:8 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);       /* row_bcast:15 -> rows 1, 3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);       /* row_bcast:31 -> rows 2, 3 */
    return x;
}

/* When the packed accumulators (value sum in 23 bits, known count above) must be emptied.
 * They hold 128 beams. Counting beams per group of entries cost ten scalar instructions per
 * group (7 % of the kernel); instead the 64 entries of a chunk of the list are looked at
 * once, one entry per lane: C = beams gathered so far including the entry (a running count
 * modulo 96 + a prefix sum). An entry is FLAGGED if C crosses a multiple of 96 at it, if it
 * is heavy (more than 8 beams), or if one of the four entries before it is heavy; a group
 * that holds a flagged entry empties the accumulators first. Between two flushes lie one
 * flagged group and groups without a flagged entry. If the flagged group has a heavy entry,
 * the next group is flagged too: <= 4 x 30 beams. Otherwise it has <= 32 beams and the
 * others stay inside one bucket of 96 (< 96 beams): < 128 in all. */
struct FlushState {
    int cum;        /* beams gathered so far, modulo 96 */
    int carry;      /* a heavy entry among the last four of the previous chunk */
};

/* After the last record: the accumulators' rest, then the value sum alone
 * (S held sum + count << 23 modulo 2^32; the sum itself is below 2^27). */
template <int R>
__device__ __forceinline__ void pairs_finish(uint32_t (&acc)[R], uint32_t (&S)[R], uint32_t (&K)[R])
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        K[r] += acc[r] >> 23;
        S[r] = S[r] + acc[r] - (K[r] << 23);
    }
}


/* Which (candidate block, slice, job) a workgroup of a batch launch works on. Workgroups are
 * handed to the 8 XCDs round-robin in the order of their linear id, and every XCD has an L2 of
 * its own: with the identity mapping the ~100 workgroups of one job -- which copy windows of
 * ONE map -- are spread over all eight L2s, and each of them fetches that map's windows from
 * the fabric. With xcd_map the jobs are dealt to the XCDs instead (job j on XCD j mod 8): linear
 * id L -> XCD L mod 8, the XCD's q-th workgroup (q = L / 8) -> job 8 (q / P) + L mod 8, part
 * q mod P of it (P = workgroups per job). Jobs beyond the last multiple of 8 keep the identity
 * mapping. (BASELINE configs[2], [3]: one map per job; configs[1]'s windows share one map.) */
__device__ __forceinline__ void xcd_block(int xcd_map, int& bx, int& by, int& bz)
{
    bx = (int)blockIdx.x;
    by = (int)blockIdx.y;
    bz = (int)blockIdx.z;
    if (!xcd_map)
        return;
    const uint32_t per_job = gridDim.x * gridDim.y;
    const uint32_t jobs8 = gridDim.z & ~7u;
    const uint32_t lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (lin >= per_job * jobs8)
        return;
    const uint32_t q = lin >> 3, slot = q / per_job, part = q - slot * per_job;
    bz = (int)(8u * slot + (lin & 7u));
    by = (int)(part / gridDim.x);
    bx = (int)(part - (uint32_t)by * gridDim.x);
}


/* PositionToIndex on device, IEEE double, no contraction: bit-identical to
 * src/grid_map_new/grid_map_geometry.cpp:113-122 */
__device__ __forceinline__ int cell_index(double pos, double off, double res)
{
    return (int)floor((pos - off) / res);
}

/* Bound on |q_host - q_device| for q = (sensor + r*trig - off) / res, in cells.
 * `trig_err`: absolute error of the device's cosine / sine against the host's
 * cos(arg) / sin(arg) (see proj_body); then one rounding per arithmetic step on
 * either side; the caller multiplies by a safety factor. */
__device__ __forceinline__ double proj_err_bound(double r, double hit, double off, double res,
                                                 double q, double trig_err)
{
    return (fabs(r) * trig_err + (fabs(hit) + fabs(off)) * 4e-16) / res + fabs(q) * 4e-16;
}

} /* namespace csm */
#endif
