/* csm_joint_kernels.hip -- the batched fine level on JOINT entries of two neighbouring
 * theta slices (round 3). A translation unit of its own inside libcsm_hip.so.
 *
 * Why. Slices 2p and 2p + 1 are 0.25 - 0.5 degrees apart: four fifths of the (row pair,
 * column) slots one of them hits are hit by the other too (configs[1]: 480 pair entries per
 * slice, 533 in the union of two slices). Round 2's batch kernel (k_score_pairs2_batch)
 * already staged ONE window per tile for the two slices but walked two entry lists: every
 * shared slot was read from LDS twice (R/2 + 1 ds_read_b64 each time), decoded twice, and
 * the lists came sorted by three classes whose 1 - 3 left-over entries ran through a loop
 * of single, latency-exposed entries. Here k_binj bins the beams of BOTH slices into one
 * hash table, an entry is a slot with four beam counts
 *     (even row, slice 0) (odd row, slice 0) (even row, slice 1) (odd row, slice 1)
 * and the gather reads the slot's R/2 + 1 row pairs once and feeds both accumulator sets:
 * 45 % fewer LDS reads and address / decode instructions per launch, no classes, no class
 * tails. A count of zero skips its R multiply-adds by a scalar branch inside the inline
 * assembly block (s_bfe_u32 sets SCC; the compiler sees one straight-line block, so the
 * hand-issued reads in flight never cross a basic-block edge).
 *
 * Replaces the sweep of src/mapping/scan_matcher_correlative.cpp:161-197, 301-368 and
 * ScorePixelAccurate::Score per leaf (src/mapping/score_function_pixel_accurate.cpp:16-58)
 * for batches; exact integer (S, K) per candidate as before.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "csm_score_common.hpp"
#include "csm_joint.hpp"

namespace csm {

/* ------------------------------------------------------------------ K0, joint */
#ifdef CSM_BIN_TIMING
/* tuning builds only (tools/bin_phases.py): cycles per phase of k_binj, one row of 16 counters per
 * workgroup (thread 0, plain stores) behind BinJob.tuning_counters */
#define BINJ_TICK(k)                                                                  \
    do {                                                                              \
        if (threadIdx.x == 0 && job.tuning_counters) {                                \
            const unsigned long long now_ = __builtin_readcyclecounter();             \
            reinterpret_cast<unsigned long long*>(job.tuning_counters)[dbg_row_ * 16 + (k)] += now_ - tick_; \
            tick_ = now_;                                                             \
        }                                                                             \
    } while (0)
#else
#define BINJ_TICK(k) do { } while (0)
#endif
constexpr int kBinjNoBeam = 0x3fffffff;    /* row / column of the lanes behind the last beam */
constexpr int kBinjBlock = 512;      /* threads per workgroup of the joint binning kernel: its hash table
                                        (12 B per slot, binj_hash_size(): 2880 slots for 2 x 1080 beams,
                                        40 KB with the rest) lets four workgroups share a CU */

/* One workgroup (kBinjBlock threads) per PAIR of theta slices (2p, 2p + 1; the last pair of
 * an odd number of slices holds one). Entry words:
 *   sorted_pb: m_o1 << 28 | m_e1 << 24 | m_o0 << 20 | m_e0 << 16 | byte offset of the slot
 *              = (pair_row * lstride + col) * 8 inside the tile's bounding box (< 2^16: at most 31
 *              pair rows of at most 262 slots)
 *   sorted_rc: the same counts | row << 7 | col (row even, rows / cols inside the bounding
 *              box; the strided kernels of the coarser levels pick the counts of their slice)
 * Lists, records and record counts are indexed by the pair: sorted_pb + p * 2 n_points,
 * tiles + p * max_tiles, n_tiles[p]. The structure follows k_bin (csm_kernels.hip): an LDS
 * hash table keyed by (tile, row pair, column), runs of neighbouring lanes with one key
 * inserted once, the probes of kAhead iterations travelling together, a list of the slots
 * claimed (a wave appends the new slots of its kAhead iterations behind one LDS counter with one
 * atomic). No classes: one count per tile.
 * The table has any size (slot = high word of hash x size), not a power of two: 4/3 of the beams
 * instead of up to 8/3 is what brings four workgroups to a CU. Passes B and C walk the list with
 * lanes 67 entries apart: the list is in beam order, so neighbouring entries belong to one tile and a
 * wave walking it in order would send its 64 atomics per instruction to one LDS address (that, not
 * the hash probes, was 90 % of the bank-conflict cycles of this kernel). */
__device__ __forceinline__ void k_binj_body(const BinJob& job)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_binj[];
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int tiles_x = uni(job.tiles_x);
    const int ntile = tiles_x * uni(job.tiles_y);
    const int hash_size = uni(job.hash_size);
    const int n_theta_job = uni(job.n_theta);
    const int p = blockIdx.x;
    const int t0 = 2 * p;
    if (t0 >= n_theta_job)
        return;
    const int n = uni(job.n_points);
    const int n2 = t0 + 1 < n_theta_job ? 2 * n : n;         /* beams of this pair */
    const int ntp = (ntile + 1) & ~1;
    unsigned long long* rowmask = reinterpret_cast<unsigned long long*>(sm_binj);      /* [ntp] */
    unsigned long long* colmask = rowmask + ntp;                                       /* [ntp] */
    unsigned long long* hval = colmask + ntp;        /* [hash_size] beams: e0 | o0 << 16 | e1 << 32 | o1 << 48 */
    uint32_t* hkey = reinterpret_cast<uint32_t*>(hval + hash_size);   /* [hash_size] (tile, cell) + 1, 0 = empty */
    uint32_t* cnt = hkey + hash_size;                /* [ntp] entries of the tile, later its cursor */
    uint16_t* list = reinterpret_cast<uint16_t*>(cnt + ntp);          /* [n2] occupied slots */
    __shared__ uint32_t list_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    /* slices t0 and t0 + 1 are contiguous in [n_theta][n_points]: beam i of the pair */
    const int32_t* col = job.hit_col + (size_t)t0 * n;
    const int32_t* row = job.hit_row + (size_t)t0 * n;
    const uint32_t hsize = (uint32_t)hash_size;
#ifdef CSM_BIN_TIMING
    unsigned long long tick_ = __builtin_readcyclecounter();
    const size_t dbg_row_ = min((size_t)blockIdx.y * gridDim.x + blockIdx.x, (size_t)kBinDebugRows - 1);
    if (threadIdx.x == 0 && job.tuning_counters) {
        unsigned long long* row_ = reinterpret_cast<unsigned long long*>(job.tuning_counters) + dbg_row_ * 16;
        for (int k = 0; k < 12; ++k)
            row_[k] = 0ull;
        row_[7] = 1ull;
    }
#endif

    /* (the job's fields first: their loads must not sit behind the barrier below, where waiting for them
     * would wait for the beams too) */
    bool band = false;
    int low_edge = 1;
    const int fs = uni(job.frame_shift);
    const int x_hi = uni(job.x_hi), y_hi = uni(job.y_hi), x_lo = uni(job.x_lo), y_lo = uni(job.y_lo);
    /* inside the frame: 0 <= r + y_hi <= rows - 1 - y_lo + y_hi, the columns alike */
    const uint32_t r_span = (uint32_t)(uni(job.rows) - 1 - y_lo + y_hi), c_span = (uint32_t)(uni(job.cols) - 1 - x_lo + x_hi);
    const int lw = 6 + (tiles_x > 1 ? 32 - __builtin_clz(tiles_x - 1) : 0);    /* frame columns < 2^lw */
    const unsigned long long from_me = ~0ull << lane, above_me = from_me << 1;
    const int n_band = uni(job.n_band);
    const int n_iter = (n2 + kBinjBlock - 1) / kBinjBlock;
    auto in_band = [&](int u, int w, int span, int known_lo) {
        if (u > 0 || u <= -span)
            return false;
        const int m = (-u) % w;
        return m != 0 && w - 1 - m >= known_lo;
    };
    const int known_r0 = uni(job.known_r0), known_c0 = uni(job.known_c0);

    constexpr int kAhead = 5;
    int rv[kAhead], cv[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
        const int i = u * kBinjBlock + tid;
        rv[u] = cv[u] = kBinjNoBeam;
        if (i < n2) {
            rv[u] = row[i];
            cv[u] = col[i];
        }
    }
    {
        /* masks, table and counts are contiguous: 16 bytes per lane and store (the last store may run
         * into the list, which is written after the barrier) */
        uint4* z = reinterpret_cast<uint4*>(sm_binj);
        const int n16 = (20 * ntp + 12 * hash_size + 15) >> 4;
        for (int i = tid; i < n16; i += kBinjBlock)
            z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0)
        list_n = 0u;
    /* the LDS stores only: the beams' global loads stay in flight across the barrier (__syncthreads()
     * would wait for them too; the first use below waits) */
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    BINJ_TICK(0);

    for (int it0 = 0; it0 < n_iter; it0 += kAhead) {
        if (it0) {
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int i = (it0 + u) * kBinjBlock + tid;
                rv[u] = cv[u] = kBinjNoBeam;
                if (i < n2) {
                    rv[u] = row[i];
                    cv[u] = col[i];
                }
            }
        }
        uint32_t key[kAhead], slot[kAhead];
        unsigned long long beams[kAhead];
        bool pend[kAhead], first[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            key[u] = 0xffffffffu;
            slot[u] = 0;
            beams[u] = 0ull;
            pend[u] = first[u] = false;
            /* (iterations behind the last beam run on kBinjNoBeam: no branch, the predicates stay in SGPRs) */
            const int i = (it0 + u) * kBinjBlock + tid;
            const int r = rv[u], c = cv[u];         /* kBinjNoBeam behind the last beam: invalid, far from the edge */
            const uint32_t ry = (uint32_t)(r + y_hi), cx = (uint32_t)(c + x_hi);
            const bool valid = ry <= r_span && cx <= c_span;
            const uint32_t sl = i >= n ? 1u : 0u;              /* which slice of the pair */
            /* key: (pair row of the frame, column of the frame) + 1; the tile comes out of it again in
             * passes B and C, where there is a quarter of the lanes to pay for it */
            const uint32_t rr = ry + (uint32_t)fs;
            key[u] = (((rr >> 1) << lw) | cx) + 1u;
            const bool odd = valid && (rr & 1u) != 0;
            const uint32_t cmp = valid ? key[u] | (sl << 31) : 0xffffffffu;   /* a run never spans the two slices */
            /* the neighbour below: one DPP move (wave_shr:1; lane 0 is a head anyway) */
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~cmp, (int)cmp, 0x138, 0xf, 0xf, false);
            const bool head = valid && (lane == 0 || cmp != prev);
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(head), vm = __builtin_amdgcn_ballot_w64(valid), om = __builtin_amdgcn_ballot_w64(odd);
            const unsigned long long above = (hm | ~vm) & above_me;
            const int e = above ? __builtin_ctzll(above) : 64;
            const unsigned long long run = (e == 64 ? ~0ull : (1ull << e) - 1ull) & from_me;
            const uint32_t co = (uint32_t)__popcll(run & om), ce = (uint32_t)__popcll(run) - co;
            beams[u] = (unsigned long long)(ce | (co << 16)) << (32u * sl);
            /* multiplicative hash on the full-rate 24-bit multiplier: 18 bits of key x C, scaled to the table */
            slot[u] = __umul24(__builtin_amdgcn_ubfe(__umul24(key[u], 0x3779b1u), 9, 18), hsize) >> 18;
            pend[u] = head;
            low_edge = min(low_edge, min(r + y_lo, c + x_lo));
        }
        BINJ_TICK(8);
        bool any = true;
        while (any) {
            uint32_t old[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u)
                old[u] = pend[u] ? atomicCAS(&hkey[slot[u]], 0u, key[u]) : 0u;
            any = false;
#pragma unroll
            for (int u = 0; u < kAhead; ++u)
                if (pend[u]) {
                    if (old[u] == 0u || old[u] == key[u]) {
                        first[u] = old[u] == 0u;
                        atomicAdd(&hval[slot[u]], beams[u]);
                        pend[u] = false;
                    } else {
                        slot[u] = slot[u] + 1u == hsize ? 0u : slot[u] + 1u;
                        any = true;
                    }
                }
            any = __builtin_amdgcn_ballot_w64(any) != 0ull;
        }
        BINJ_TICK(9);
        /* the lanes that claimed an empty slot append it to the list */
        {
            unsigned long long fm[kAhead];
            uint32_t claimed = 0;
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                fm[u] = __builtin_amdgcn_ballot_w64(first[u]);
                claimed += (uint32_t)__popcll(fm[u]);
            }
            if (claimed) {                      /* wave-uniform */
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&list_n, claimed);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    if (first[u])
                        list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm[u] >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)fm[u], 0u))] =
                            (uint16_t)slot[u];
                    base += (uint32_t)__popcll(fm[u]);
                }
            }
        }
        BINJ_TICK(10);
    }
    /* Edge band (A8): only beams within a window of the map's low edge can be in it. One minimum per beam
     * above; the modulo test reads the beams again, here, where little else is live (rare). */
    if (__builtin_amdgcn_ballot_w64(low_edge <= 0) != 0ull) {
        for (int i = tid; i < n2; i += kBinjBlock) {
            const int r = row[i], c = col[i];
            if (r + y_lo <= 0 || c + x_lo <= 0)
                for (int b = 0; b < n_band; ++b) {
                    const int w = job.band_win[b];
                    if (in_band(r + y_lo, w, job.band_ny[b] * w, known_r0) ||
                        in_band(c + x_lo, w, job.band_nx[b] * w, known_c0))
                        band = true;
                }
        }
    }
    if (band)
        atomicOr(job.flags, kFlagBandTouch);
    __syncthreads();
    BINJ_TICK(1);

    const uint32_t max_mult = (uint32_t)uni(job.max_mult);
    auto chunks = [&](uint32_t b) { return (b + max_mult - 1u) / max_mult; };
    const int n_cells = uni((int)list_n);
    auto cell_slot = [&](int e) { return (uint32_t)list[e]; };
    /* entries of a slot: its largest beam count in chunks of max_mult (15 for merged lists: a multiply) */
    auto entries_of = [&](unsigned long long hv) {
        const uint32_t lo = (uint32_t)hv, hi = (uint32_t)(hv >> 32);
        const uint32_t m = max(max(lo & 0xffffu, lo >> 16), max(hi & 0xffffu, hi >> 16));
        return max_mult == (uint32_t)kMaxMult ? __umul24(m + (uint32_t)kMaxMult - 1u, 0x8889u) >> 19 : chunks(m);
    };
    static_assert(kMaxMult == 15, "entries_of: (m + 14) * 0x8889 >> 19 is (m + 14) / 15 for m < 2^16");
    /* the key of a slot -> tile, pair row in the tile, column in the tile */
    auto decode = [&](uint32_t k1, int& tile, uint32_t& rk, uint32_t& cbk) {
        const uint32_t cx = k1 & ((1u << lw) - 1u), rp = k1 >> lw;
        tile = (int)((rp >> 5) * (uint32_t)tiles_x + (cx >> 6));
        rk = rp & 31u;
        cbk = cx & 63u;
    };

    /* Pass B: entries per tile, bounding boxes */
    const int walk = (tid * 67) & (kBinjBlock - 1);       /* a permutation of the block's list positions */
    for (int e0 = 0; e0 < n_cells; e0 += kBinjBlock) {
        const int e = e0 + walk;
        if (e >= n_cells)
            continue;
        const uint32_t sl = cell_slot(e);
        const uint32_t k1 = hkey[sl] - 1u;
        const unsigned long long hv = hval[sl];
        int tile;
        uint32_t rk, cbk;
        decode(k1, tile, rk, cbk);
        const bool any_even = (hv & 0x0000ffff0000ffffull) != 0, any_odd = (hv & 0xffff0000ffff0000ull) != 0;
        const uint32_t rlo = 2u * rk + (any_even ? 0u : 1u), rhi = 2u * rk + (any_odd ? 1u : 0u);
        atomicAdd(&cnt[tile], entries_of(hv));
        atomicOr(&rowmask[tile], (1ull << rlo) | (1ull << rhi));
        atomicOr(&colmask[tile], 1ull << cbk);
    }
    __syncthreads();
    BINJ_TICK(2);

    /* exclusive scan of entry counts and of the records per tile, then the records: the first wave
     * alone (64 tiles for configs[1]: one per lane), the others wait at the barrier */
    if (tid < 64) {
        const int chunk = (ntile + 63) / 64;
        const int lo = lane * chunk, hi = min(lo + chunk, ntile);
        uint32_t csum = 0, ne = 0;
        for (int i = lo; i < hi; ++i) {
            const uint32_t c = cnt[i];
            csum += c;
            ne += (c + kJRec - 1) / kJRec;
        }
        const uint32_t a = (uint32_t)wave_prefix_sum((int)csum), b = (uint32_t)wave_prefix_sum((int)ne);
        if (lane == 63)
            job.n_tiles[p] = (int32_t)b;
        BINJ_TICK(3);
        uint32_t off = a - csum, slot_rec = b - ne;
        TileRec* recs = job.tiles + (size_t)p * job.max_tiles;
        for (int i = lo; i < hi; ++i) {
            const uint32_t c = cnt[i];
            if (!c)
                continue;
            const unsigned long long rm = rowmask[i], cm = colmask[i];
            const int rlo = __builtin_ctzll(rm), rhi = 63 - __builtin_clzll(rm);
            const int clo = __builtin_ctzll(cm), chi = 63 - __builtin_clzll(cm);
            const int rmin = rlo & ~1;
            for (uint32_t done = 0; done < c; done += kJRec) {
                TileRec rec;
                rec.r0 = (i / tiles_x) * kTile - y_hi - fs + rmin;
                rec.c0 = (i % tiles_x) * kTile - x_hi + clo;
                rec.start = off + done;
                rec.count = min(c - done, (uint32_t)kJRec);
                rec.h = rhi - rmin + 1;
                rec.w = chi - clo + 1;
                rec.pad[0] = 0;
                rec.pad[1] = (int)(((uint32_t)i << 4) | (done / kJRec));
                recs[slot_rec++] = rec;
            }
            cnt[i] = off;                          /* the tile's cursor */
            off += c;
        }
    }
    __syncthreads();
    BINJ_TICK(4);

    /* Pass C: the entries */
    uint32_t* out = job.sorted_pb + (size_t)p * 2 * n;
    uint32_t* out_rc = job.sorted_rc ? job.sorted_rc + (size_t)p * 2 * n : nullptr;
    const uint32_t lstride = (uint32_t)uni(job.lstride);
    for (int e0 = 0; e0 < n_cells; e0 += kBinjBlock) {
        const int e = e0 + walk;
        if (e >= n_cells)
            continue;
        const uint32_t sl = cell_slot(e);
        const uint32_t k1 = hkey[sl] - 1u;
        const unsigned long long hv = hval[sl];
        int tile;
        uint32_t rk, cbk;
        decode(k1, tile, rk, cbk);
        const uint32_t rmin = (uint32_t)__builtin_ctzll(rowmask[tile]) & ~1u;
        const uint32_t rb = (rk << 1) - rmin;        /* even */
        const uint32_t cb = cbk - (uint32_t)__builtin_ctzll(colmask[tile]);
        uint32_t b0 = (uint32_t)hv & 0xffffu, b1 = ((uint32_t)hv >> 16), b2 = (uint32_t)(hv >> 32) & 0xffffu,
                 b3 = (uint32_t)(hv >> 48);
        const uint32_t total = entries_of(hv);
        uint32_t pos = atomicAdd(&cnt[tile], total);
        for (uint32_t k = 0; k < total; ++k, ++pos) {
            const uint32_t m0 = min(b0, max_mult), m1 = min(b1, max_mult), m2 = min(b2, max_mult),
                           m3 = min(b3, max_mult);
            b0 -= m0;
            b1 -= m1;
            b2 -= m2;
            b3 -= m3;
            const uint32_t mults = (m0 << 16) | (m1 << 20) | (m2 << 24) | (m3 << 28);
            out[pos] = mults | (((rb >> 1) * lstride + cb) << 3);       /* byte offset of the slot: < 2^16 */
            if (out_rc)
                out_rc[pos] = mults | (rb << 7) | cb;
        }
    }
    BINJ_TICK(5);
}

__global__ __launch_bounds__(kBinjBlock, 8) void k_binj_batch(const BinJob* jobs)
{
    k_binj_body(jobs[blockIdx.y]);
}

/* one window, its job by value (the coarse pass of a coarse-first search): grid = (pairs of slices) */
__global__ __launch_bounds__(kBinjBlock, 8) void k_binj_one(BinJob job)
{
    k_binj_body(job);
}

/* ------------------------------------------------------------------ K1, joint */

/* The multiply-adds of one entry: four blocks of R -- (even row, slice 0), (even row, slice 1),
 * (odd row, slice 0), (odd row, slice 1) -- each skipped by a scalar branch when its 4-bit beam
 * count in the entry word is zero (s_bfe_u32 sets SCC = result != 0). v[0 .. R] are the cells
 * of the lane's R candidate rows and the one below (even rows use v[r], odd rows v[r + 1]).
 * ONE asm statement: the compiler sees straight-line code (the reads of the next entry are in
 * flight across it), adds no hazard padding between the blocks and cannot reorder them. */
#define CSM_JMAD(a, v) "v_mad_u32_u24 %[" #a "], %[" #v "], %[m], %[" #a "]\n\t"
template <int R>
__device__ __forceinline__ void joint_mads(uint32_t w, const uint32_t (&v)[R + 2], uint32_t (&a)[R], uint32_t (&b)[R])
{
    static_assert(R == 6 || R == 8, "rows per lane");
    uint32_t m;
    if constexpr (R == 8) {
        asm("s_bfe_u32 %[m], %[w], 0x40010\n\t"
            "s_cbranch_scc0 1f\n\t"
            CSM_JMAD(a0, v0) CSM_JMAD(a1, v1) CSM_JMAD(a2, v2) CSM_JMAD(a3, v3)
            CSM_JMAD(a4, v4) CSM_JMAD(a5, v5) CSM_JMAD(a6, v6) CSM_JMAD(a7, v7)
            "1:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40018\n\t"
            "s_cbranch_scc0 2f\n\t"
            CSM_JMAD(b0, v0) CSM_JMAD(b1, v1) CSM_JMAD(b2, v2) CSM_JMAD(b3, v3)
            CSM_JMAD(b4, v4) CSM_JMAD(b5, v5) CSM_JMAD(b6, v6) CSM_JMAD(b7, v7)
            "2:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40014\n\t"
            "s_cbranch_scc0 3f\n\t"
            CSM_JMAD(a0, v1) CSM_JMAD(a1, v2) CSM_JMAD(a2, v3) CSM_JMAD(a3, v4)
            CSM_JMAD(a4, v5) CSM_JMAD(a5, v6) CSM_JMAD(a6, v7) CSM_JMAD(a7, v8)
            "3:\n\t"
            "s_bfe_u32 %[m], %[w], 0x4001c\n\t"
            "s_cbranch_scc0 4f\n\t"
            CSM_JMAD(b0, v1) CSM_JMAD(b1, v2) CSM_JMAD(b2, v3) CSM_JMAD(b3, v4)
            CSM_JMAD(b4, v5) CSM_JMAD(b5, v6) CSM_JMAD(b6, v7) CSM_JMAD(b7, v8)
            "4:"
            : [m] "=&s"(m), [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]),
              [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7]), [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]),
              [b3] "+v"(b[3]), [b4] "+v"(b[4]), [b5] "+v"(b[5]), [b6] "+v"(b[6]), [b7] "+v"(b[7])
            : [w] "s"(w), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]),
              [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8])
            : "scc");
    } else {
        asm("s_bfe_u32 %[m], %[w], 0x40010\n\t"
            "s_cbranch_scc0 1f\n\t"
            CSM_JMAD(a0, v0) CSM_JMAD(a1, v1) CSM_JMAD(a2, v2) CSM_JMAD(a3, v3) CSM_JMAD(a4, v4) CSM_JMAD(a5, v5)
            "1:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40018\n\t"
            "s_cbranch_scc0 2f\n\t"
            CSM_JMAD(b0, v0) CSM_JMAD(b1, v1) CSM_JMAD(b2, v2) CSM_JMAD(b3, v3) CSM_JMAD(b4, v4) CSM_JMAD(b5, v5)
            "2:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40014\n\t"
            "s_cbranch_scc0 3f\n\t"
            CSM_JMAD(a0, v1) CSM_JMAD(a1, v2) CSM_JMAD(a2, v3) CSM_JMAD(a3, v4) CSM_JMAD(a4, v5) CSM_JMAD(a5, v6)
            "3:\n\t"
            "s_bfe_u32 %[m], %[w], 0x4001c\n\t"
            "s_cbranch_scc0 4f\n\t"
            CSM_JMAD(b0, v1) CSM_JMAD(b1, v2) CSM_JMAD(b2, v3) CSM_JMAD(b3, v4) CSM_JMAD(b4, v5) CSM_JMAD(b5, v6)
            "4:"
            : [m] "=&s"(m), [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]),
              [a5] "+v"(a[5]), [b0] "+v"(b[0]), [b1] "+v"(b[1]), [b2] "+v"(b[2]), [b3] "+v"(b[3]), [b4] "+v"(b[4]),
              [b5] "+v"(b[5])
            : [w] "s"(w), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]),
              [v5] "v"(v[5]), [v6] "v"(v[6])
            : "scc");
    }
}
#undef CSM_JMAD

/* The entries of one record from the staged window into the accumulators of both slices.
 * Rules of the hand-issued reads as in pairs_gather (csm_kernels.hip): a read reaches its
 * lds_wait inside one basic block; groups of four entries, two entries' reads in flight. */
template <int LS, int R>
__device__ __forceinline__ void joint_gather(uint32_t lane_addr, const uint32_t* lpb, int lane, int cnt,
                                             uint32_t (&acc0)[R], uint32_t (&S0)[R], uint32_t (&K0)[R],
                                             FlushState& fs0, uint32_t (&acc1)[R], uint32_t (&S1)[R],
                                             uint32_t (&K1)[R], FlushState& fs1)
{
    constexpr int kRowBytes = LS * 8;
    constexpr int NQ = R / 2 + 1;
    auto flush = [](uint32_t (&acc)[R], uint32_t (&S)[R], uint32_t (&K)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S[r] += acc[r];
            K[r] += acc[r] >> 23;
            acc[r] = 0;
        }
    };
    auto issue = [&](uint32_t w, unsigned long long (&q)[NQ]) {
        const uint32_t addr = lane_addr + (w & 0xffffu);
        lds_read_b64<0 * kRowBytes>(addr, q[0]);
        lds_read_b64<1 * kRowBytes>(addr, q[1]);
        lds_read_b64<2 * kRowBytes>(addr, q[2]);
        lds_read_b64<3 * kRowBytes>(addr, q[3]);
        if constexpr (R >= 8)
            lds_read_b64<4 * kRowBytes>(addr, q[4]);
    };
    auto mads = [&](uint32_t w, const unsigned long long (&q)[NQ]) {
        uint32_t v[R + 2];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            v[2 * i] = (uint32_t)q[i];
            v[2 * i + 1] = (uint32_t)(q[i] >> 32);
        }
        joint_mads<R>(w, v, acc0, acc1);
    };
    uint32_t pb_cur;
    unsigned long long flags0, flags1;
    auto flags_of = [&](int b, int real, FlushState& fs) {
        const int c1 = fs.cum + wave_prefix_sum(b), c0 = c1 - b;
        const bool cross = ((uint32_t)c1 * 683u) >> 16 != ((uint32_t)c0 * 683u) >> 16;      /* floor(c / 96) */
        const unsigned long long heavy = __builtin_amdgcn_ballot_w64(b > 8);
        const unsigned long long f = __builtin_amdgcn_ballot_w64(cross) | heavy | heavy << 1 | heavy << 2 |
                                     heavy << 3 | heavy << 4 | (fs.carry ? 0xfull : 0ull);
        fs.carry = (heavy >> max(0, real - 4)) != 0;        /* among the last four real entries */
        fs.cum = __builtin_amdgcn_readlane(c1, 63) % 96;
        return f;
    };
    auto load_chunk = [&](int j0) {
        pb_cur = lpb[j0 + lane];
        const bool real = j0 + lane < cnt;
        const int b0 = real ? (int)(((pb_cur >> 16) & 15u) + ((pb_cur >> 20) & 15u)) : 0;
        const int b1 = real ? (int)(((pb_cur >> 24) & 15u) + (pb_cur >> 28)) : 0;
        const int nreal = min(64, cnt - j0);
        flags0 = flags_of(b0, nreal, fs0);
        flags1 = flags_of(b1, nreal, fs1);
    };
    int j = 0;
    load_chunk(0);
    while (j < cnt) {
        const int stop = min(cnt, (j | 63) + 1);
        for (; j + 4 <= stop; j += 4) {
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j & 63);
            const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + 1) & 63);
            const uint32_t o2 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + 2) & 63);
            const uint32_t o3 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + 3) & 63);
            if ((flags0 >> (j & 63)) & 0xfull)
                flush(acc0, S0, K0);
            if ((flags1 >> (j & 63)) & 0xfull)
                flush(acc1, S1, K1);
            unsigned long long qa[NQ], qb[NQ], qc[NQ], qd[NQ];
            issue(o0, qa);
            issue(o1, qb);
            lds_wait<NQ, NQ>(qa);
            mads(o0, qa);
            issue(o2, qc);
            lds_wait<NQ, NQ>(qb);
            mads(o1, qb);
            issue(o3, qd);
            lds_wait<NQ, NQ>(qc);
            mads(o2, qc);
            lds_wait<0, NQ>(qd);
            mads(o3, qd);
        }
        for (; j < stop; ++j) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j & 63);
            if ((flags0 >> (j & 63)) & 1ull)
                flush(acc0, S0, K0);
            if ((flags1 >> (j & 63)) & 1ull)
                flush(acc1, S1, K1);
            unsigned long long qa[NQ];
            issue(o, qa);
            lds_wait<0, NQ>(qa);
            mads(o, qa);
        }
        if ((j & 63) == 0 && j < cnt)
            load_chunk(j);
    }
}

/* One workgroup = (pair of theta slices, block of cbx x groups * R candidate offsets). Lane
 * <-> candidate mapping, window staging (LDS-DMA from the pair-row copy of the level), the
 * epilogue and the row-block split (BlockBase) as in score_body_pairs2; ONE record list. */
template <int LS, int R>
__device__ __forceinline__ void score_body_joint(const ScoreJob& job, int cbx, int groups, const uint16_t* lane_map,
                                                 int bid_x, int bid_y, BlockBase bb)
{
    static_assert(R % 2 == 0 && LS % 2 == 0, "pair rows, 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) uint16_t sm_tile[];
    const int t0 = 2 * bid_y, t1 = t0 + 1;
    const int n_theta = __builtin_amdgcn_readfirstlane(job.n_theta);
    if (t0 >= n_theta)
        return;
    const bool two = t1 < n_theta;
    const int tid = threadIdx.x;
    const int ncbx = (job.nx + cbx - 1) / cbx;
    const int bx = bid_x % ncbx, by = bid_x / ncbx;
    const int cby = groups * R;
    const int row0 = bb.row_base + by * cby;
    if (row0 >= job.ny)
        return;
    const uint32_t qflags = job.elig_only_if_band ? *job.flags : 0u;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int dxi = tid % cbx, g = tid / cbx;
    bool idle = false;
    if (lane_map) {
        const uint32_t m = lane_map[tid];
        dxi = (int)(m & 255u);
        g = (int)((m >> 8) & 127u);
        idle = (m >> 15) != 0;
    }
    const bool lane_on = !idle && g < groups;
    const int x0 = job.x_lo + bx * cbx;
    const int y0 = job.y_lo + row0;
    constexpr int kRowBytes = LS * 8;
    const int prows_full = (kTile + cby) / 2 + 1;
    const int max_pieces = (prows_full * kRowBytes + 1023) >> 10;
    uint32_t* sm_cells = reinterpret_cast<uint32_t*>(sm_tile);
    uint32_t* lpb = sm_cells + max_pieces * 256;
    const int tb = lane_on || (idle && g < groups && dxi < cbx) ? (g * (R / 2)) * kRowBytes + 8 * dxi : 0;
    const bool wave_live = __builtin_amdgcn_ballot_w64(lane_on && bx * cbx + dxi < job.nx &&
                                                       row0 + g * R < job.ny) != 0;

    uint32_t S0[R], K0[R], acc0[R], S1[R], K1[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        S0[r] = K0[r] = acc0[r] = S1[r] = K1[r] = acc1[r] = 0;
    FlushState fs0 = { 0, 0 }, fs1 = { 0, 0 };

    const int ntiles = __builtin_amdgcn_readfirstlane(job.in_s ? 0 : job.n_tiles[bid_y]);
    const TileRec* recs = job.tiles + (size_t)bid_y * job.max_tiles;
    const uint32_t* __restrict__ pbs = job.sorted_pb + (size_t)bid_y * 2 * job.n_points;
    const size_t xg_pitch = (size_t)job.xg_pitch;
    const size_t xg_row_bytes = xg_pitch * 8;
    const char* xg = reinterpret_cast<const char*>(job.xg);
    const int pad = job.xg_pad;

    constexpr int kMaxP = ((((kTile + kPairMaxCby) / 2 + 1) * kRowBytes + 1023) / 1024 + 7) / 8;
    uint32_t goff[kMaxP];
#pragma unroll
    for (int k = 0; k < kMaxP; ++k) {
        const uint32_t ob = (uint32_t)(wave + 8 * k) * 1024u + (uint32_t)lane * 16u;
        const uint32_t prow = ob / (uint32_t)kRowBytes, cb = ob - prow * (uint32_t)kRowBytes;
        goff[k] = prow * (uint32_t)xg_row_bytes + cb;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;

    TileRec rec;
    if (ntiles > 0)
        rec = recs[0];
    for (int ti = 0; ti < ntiles; ++ti) {
        const int c00 = __builtin_amdgcn_readfirstlane(rec.c0) + x0;
        const int gr0 = __builtin_amdgcn_readfirstlane(rec.r0) + y0;          /* even */
        const int a = c00 & 1;
        const int cnt = __builtin_amdgcn_readfirstlane((int)rec.count);
        const int start = __builtin_amdgcn_readfirstlane((int)rec.start);
        const int nprows = (__builtin_amdgcn_readfirstlane(rec.h) + cby) >> 1;
        const int npieces = (nprows * kRowBytes + 1023) >> 10;
        const char* src = xg + ((size_t)((gr0 + pad) >> 1) * xg_pitch + (size_t)((c00 & ~1) + pad)) * 8;
        if (ti + 1 < ntiles)
            rec = recs[ti + 1];
        __syncthreads();                                 /* previous tile consumed */
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int k = 0; k < kMaxP; ++k) {
            const int pc = wave + 8 * k;
            if (pc < npieces)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src + goff[k]), (lds_ptr)(sm_cells + pc * 256), 16, 0, 0);
        }
        static_assert(kJRec / 64 <= 8, "one list piece per wave");
        if (wave * 64 < cnt)
            __builtin_amdgcn_global_load_lds((glb_ptr)(pbs + start + wave * 64 + lane),
                                             (lds_ptr)(lpb + wave * 64), 4, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t lane_addr = lds_address(sm_cells) + (uint32_t)(tb + 8 * a);
        if (wave_live)
            joint_gather<LS, R>(lane_addr, lpb, lane, cnt, acc0, S0, K0, fs0, acc1, S1, K1, fs1);
    }
    pairs_finish<R>(acc0, S0, K0);
    pairs_finish<R>(acc1, S1, K1);
    score_epilogue<R>(job, S0, K0, t0, bx, 1, cbx, row0, g, dxi, lane_on, qflags, bid_x + bb.cb_base, bb.ncb);
    if (two) {
        __syncthreads();                                 /* the epilogue's reduction arrays */
        score_epilogue<R>(job, S1, K1, t1, bx, 1, cbx, row0, g, dxi, lane_on, qflags, bid_x + bb.cb_base, bb.ncb);
    }
}

/* ------------------------------------------------------------------ K1, joint, fp32 bound pass */
/* The same gather in PACKED fp32: v_pk_fma_f32 does two multiply-adds per lane and instruction at
 * the issue rate of v_mad_u32_u24 (tools/micro/pk_fma_bench.hip), and a ds_read_b64 of the
 * pair-row layout delivers exactly the operand pair of two neighbouring candidate rows. What is
 * summed is the candidate's ORDER KEY itself, 32268 K + 499 S = sum of beams * (499 v + 32268 (v != 0)),
 * one float per cell (k_expand_pairs_f): no separate known count, no flushes. The sum is not exact
 * (keys reach 2^35); every term is non-negative, so after n additions the relative error is below
 * (n + 1) 2^-24. The pass therefore does not pick the winner: it writes the greatest value of every
 * (slice, candidate block), and the exact integer kernel afterwards skips the blocks whose maximum
 * lies more than the proven margin below the window's maximum (score_body_joint, `approx_best`).
 *
 * Row pairing. A lane's even candidate rows pair up with the row pairs of the LDS layout for a hit on
 * an EVEN frame row: E[i] = (rows 2i, 2i + 1) += slot i. For a hit on the odd row the cells sit one
 * row lower; instead of re-pairing registers the lane keeps a second accumulator set shifted by one
 * row, O[i] = (rows 2i - 1, 2i) += slot i, i = 0 .. R/2 (rows -1 and R are never read). 4 + 5
 * instructions per (slice, parity) for 8 candidate rows instead of 8. */
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_expand_pairs_f(const uint16_t* __restrict__ cells, int rows, int cols,
                                                       int pitch, float2* __restrict__ xgf, int xg_prows,
                                                       int xg_pitch, int pad)
{
    const size_t total = (size_t)xg_prows * xg_pitch;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i / xg_pitch), c = (int)(i % xg_pitch) - pad;
        const int r0 = 2 * k - pad;
        uint32_t v0 = 0, v1 = 0;
        if (c >= 0 && c < cols) {
            if (r0 >= 0 && r0 < rows)
                v0 = cells[(size_t)r0 * pitch + c];
            if (r0 + 1 >= 0 && r0 + 1 < rows)
                v1 = cells[(size_t)(r0 + 1) * pitch + c];
        }
        /* one rounding (values above 2^24): part of the bound pass's error budget */
        xgf[i] = make_float2((float)(499u * v0 + 32268u * min(v0, 1u)), (float)(499u * v1 + 32268u * min(v1, 1u)));
    }
}

/* The packed multiply-adds of one entry. fa = (float beams even row, odd row) of slice 0, fb of
 * slice 1 (from the record's float table in LDS); zero counts are skipped by scalar branches on the
 * integer counts in the entry word. */
#define CSM_JFMA_E(acc, q, f) "v_pk_fma_f32 %[" #acc "], %[" #q "], %[" #f "], %[" #acc "] op_sel_hi:[1,0,1]\n\t"
#define CSM_JFMA_O(acc, q, f) "v_pk_fma_f32 %[" #acc "], %[" #q "], %[" #f "], %[" #acc "] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
template <int R>
__device__ __forceinline__ void joint_fmads(uint32_t w, const unsigned long long (&q)[R / 2 + 1], unsigned long long fa,
                                            unsigned long long fb, f32x2 (&ea)[R / 2], f32x2 (&oa)[R / 2 + 1],
                                            f32x2 (&eb)[R / 2], f32x2 (&ob)[R / 2 + 1])
{
    static_assert(R == 6 || R == 8, "rows per lane");
    uint32_t m;
    if constexpr (R == 8) {
        asm("s_bfe_u32 %[m], %[w], 0x40010\n\t"
            "s_cbranch_scc0 1f\n\t"
            CSM_JFMA_E(ea0, q0, fa) CSM_JFMA_E(ea1, q1, fa) CSM_JFMA_E(ea2, q2, fa) CSM_JFMA_E(ea3, q3, fa)
            "1:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40018\n\t"
            "s_cbranch_scc0 2f\n\t"
            CSM_JFMA_E(eb0, q0, fb) CSM_JFMA_E(eb1, q1, fb) CSM_JFMA_E(eb2, q2, fb) CSM_JFMA_E(eb3, q3, fb)
            "2:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40014\n\t"
            "s_cbranch_scc0 3f\n\t"
            CSM_JFMA_O(oa0, q0, fa) CSM_JFMA_O(oa1, q1, fa) CSM_JFMA_O(oa2, q2, fa) CSM_JFMA_O(oa3, q3, fa)
            CSM_JFMA_O(oa4, q4, fa)
            "3:\n\t"
            "s_bfe_u32 %[m], %[w], 0x4001c\n\t"
            "s_cbranch_scc0 4f\n\t"
            CSM_JFMA_O(ob0, q0, fb) CSM_JFMA_O(ob1, q1, fb) CSM_JFMA_O(ob2, q2, fb) CSM_JFMA_O(ob3, q3, fb)
            CSM_JFMA_O(ob4, q4, fb)
            "4:"
            : [m] "=&s"(m), [ea0] "+v"(ea[0]), [ea1] "+v"(ea[1]), [ea2] "+v"(ea[2]), [ea3] "+v"(ea[3]),
              [oa0] "+v"(oa[0]), [oa1] "+v"(oa[1]), [oa2] "+v"(oa[2]), [oa3] "+v"(oa[3]), [oa4] "+v"(oa[4]),
              [eb0] "+v"(eb[0]), [eb1] "+v"(eb[1]), [eb2] "+v"(eb[2]), [eb3] "+v"(eb[3]),
              [ob0] "+v"(ob[0]), [ob1] "+v"(ob[1]), [ob2] "+v"(ob[2]), [ob3] "+v"(ob[3]), [ob4] "+v"(ob[4])
            : [w] "s"(w), [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3]), [q4] "v"(q[4]),
              [fa] "v"(fa), [fb] "v"(fb)
            : "scc");
    } else {
        asm("s_bfe_u32 %[m], %[w], 0x40010\n\t"
            "s_cbranch_scc0 1f\n\t"
            CSM_JFMA_E(ea0, q0, fa) CSM_JFMA_E(ea1, q1, fa) CSM_JFMA_E(ea2, q2, fa)
            "1:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40018\n\t"
            "s_cbranch_scc0 2f\n\t"
            CSM_JFMA_E(eb0, q0, fb) CSM_JFMA_E(eb1, q1, fb) CSM_JFMA_E(eb2, q2, fb)
            "2:\n\t"
            "s_bfe_u32 %[m], %[w], 0x40014\n\t"
            "s_cbranch_scc0 3f\n\t"
            CSM_JFMA_O(oa0, q0, fa) CSM_JFMA_O(oa1, q1, fa) CSM_JFMA_O(oa2, q2, fa) CSM_JFMA_O(oa3, q3, fa)
            "3:\n\t"
            "s_bfe_u32 %[m], %[w], 0x4001c\n\t"
            "s_cbranch_scc0 4f\n\t"
            CSM_JFMA_O(ob0, q0, fb) CSM_JFMA_O(ob1, q1, fb) CSM_JFMA_O(ob2, q2, fb) CSM_JFMA_O(ob3, q3, fb)
            "4:"
            : [m] "=&s"(m), [ea0] "+v"(ea[0]), [ea1] "+v"(ea[1]), [ea2] "+v"(ea[2]),
              [oa0] "+v"(oa[0]), [oa1] "+v"(oa[1]), [oa2] "+v"(oa[2]), [oa3] "+v"(oa[3]),
              [eb0] "+v"(eb[0]), [eb1] "+v"(eb[1]), [eb2] "+v"(eb[2]),
              [ob0] "+v"(ob[0]), [ob1] "+v"(ob[1]), [ob2] "+v"(ob[2]), [ob3] "+v"(ob[3])
            : [w] "s"(w), [q0] "v"(q[0]), [q1] "v"(q[1]), [q2] "v"(q[2]), [q3] "v"(q[3]), [fa] "v"(fa), [fb] "v"(fb)
            : "scc");
    }
}
#undef CSM_JFMA_E
#undef CSM_JFMA_O

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

/* ftab_addr: LDS byte address of the record's float table (16 B per entry: beams of even / odd
 * row of slice 0, of slice 1); the two broadcast reads of an entry ride in the same lgkmcnt queue
 * as its R/2 + 1 slot reads. */
template <int LS, int R>
__device__ __forceinline__ void joint_gather_f(uint32_t lane_addr, uint32_t ftab_addr, const uint32_t* lpb, int lane,
                                               int cnt, f32x2 (&ea)[R / 2], f32x2 (&oa)[R / 2 + 1],
                                               f32x2 (&eb)[R / 2], f32x2 (&ob)[R / 2 + 1])
{
    constexpr int kRowBytes = LS * 8;
    constexpr int NQ = R / 2 + 1;
    constexpr int NP = NQ + 2;                      /* reads per entry */
    /* K: the entry's position in its group of four (its float-table row is an immediate offset from
     * the group's row: one address register per group) */
    auto issue = [&](uint32_t w, uint32_t faddr, auto kk, unsigned long long (&q)[NQ], unsigned long long (&f)[2]) {
        constexpr int K = decltype(kk)::value;
        const uint32_t addr = lane_addr + (w & 0xffffu);
        lds_read_b64<0 * kRowBytes>(addr, q[0]);
        lds_read_b64<1 * kRowBytes>(addr, q[1]);
        lds_read_b64<2 * kRowBytes>(addr, q[2]);
        lds_read_b64<3 * kRowBytes>(addr, q[3]);
        if constexpr (R >= 8)
            lds_read_b64<4 * kRowBytes>(addr, q[4]);
        lds_read_b64<16 * K>(faddr, f[0]);
        lds_read_b64<16 * K + 8>(faddr, f[1]);
    };
    /* waits until all but the `LATER` youngest LDS reads have landed; ties every register the
     * mads are about to read */
    auto wait = [&](auto later, unsigned long long (&q)[NQ], unsigned long long (&f)[2]) {
        constexpr int LATER = decltype(later)::value;
        if constexpr (R >= 8)
            asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(f[0]), "+v"(f[1]) : "n"(LATER));
        else
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(f[0]), "+v"(f[1]) : "n"(LATER));
    };
    using later_t = std::integral_constant<int, NP>;
    using now_t = std::integral_constant<int, 0>;
    uint32_t pb_cur = lpb[lane];
    int j = 0;
    /* N entries through the pipeline: two entries in flight from the first wait to the last; the reads drain
     * once per pass (the first wait of a pass sees the full LDS latency: 8 entries per pass instead of 4
     * took 2.2 % off the launch; 16 per pass gave half of that back, its code no longer sits well in the
     * instruction cache, and carrying two entries in flight from one pass into the next makes the register
     * sets loop-carried: 128 VGPRs and scratch). Register sets a, b, c, d in turn; entry u's float-table row is an immediate
     * offset from the pass's first row. */
    auto pass = [&](auto n_) {
        constexpr int N = decltype(n_)::value;
        uint32_t o[N];
#pragma unroll
        for (int u = 0; u < N; ++u)
            o[u] = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + u) & 63);
        const uint32_t fbase = ftab_addr + (uint32_t)j * 16u;
        unsigned long long q[4][NQ], f[4][2];
        issue(o[0], fbase, std::integral_constant<int, 0>(), q[0], f[0]);
        if constexpr (N > 1)
            issue(o[1], fbase, std::integral_constant<int, 1>(), q[1], f[1]);
        static_for<0, N>([&](auto u_) {
            constexpr int U = decltype(u_)::value;
            if constexpr (U + 1 < N)
                wait(later_t(), q[U % 4], f[U % 4]);
            else
                wait(now_t(), q[U % 4], f[U % 4]);
            joint_fmads<R>(o[U], q[U % 4], f[U % 4][0], f[U % 4][1], ea, oa, eb, ob);
            if constexpr (U + 2 < N)
                issue(o[U + 2], fbase, std::integral_constant<int, U + 2>(), q[(U + 2) % 4], f[(U + 2) % 4]);
        });
        j += N;
    };
    while (j < cnt) {
        const int stop = min(cnt, (j | 63) + 1);
#ifndef CSM_ABL_GROUP4
        while (j + 8 <= stop)
            pass(std::integral_constant<int, 8>());
#endif
        while (j + 4 <= stop)
            pass(std::integral_constant<int, 4>());
        while (j < stop)
            pass(std::integral_constant<int, 1>());
        if ((j & 63) == 0 && j < cnt)
            pb_cur = lpb[j + lane];
    }
}

/* Workgroup = (pair of theta slices, candidate block), as score_body_joint; writes
 * approx_best[t][block] = greatest fp32 key among the block's candidates of slice t (0 if none). */
template <int LS, int R>
__device__ __forceinline__ void score_body_jointf(const ScoreJob& job, int cbx, int groups, const uint16_t* lane_map,
                                                  int bid_x, int bid_y, BlockBase bb)
{
    static_assert(R % 2 == 0 && LS % 2 == 0, "pair rows, 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) uint16_t sm_tile[];
    __shared__ float red[2][kBlock / 64];
    const int t0 = 2 * bid_y, t1 = t0 + 1;
    const int n_theta = __builtin_amdgcn_readfirstlane(job.n_theta);
    if (t0 >= n_theta || !job.approx_best)
        return;
    const bool two = t1 < n_theta;
    const int tid = threadIdx.x;
    const int ncbx = (job.nx + cbx - 1) / cbx;
    const int bx = bid_x % ncbx, by = bid_x / ncbx;
    const int cby = groups * R;
    const int row0 = bb.row_base + by * cby;
    if (row0 >= job.ny)
        return;
    int dxi = tid % cbx, g = tid / cbx;
    bool idle = false;
    if (lane_map) {
        const uint32_t m = lane_map[tid];
        dxi = (int)(m & 255u);
        g = (int)((m >> 8) & 127u);
        idle = (m >> 15) != 0;
    }
    const bool lane_on = !idle && g < groups;
    const int x0 = job.x_lo + bx * cbx;
    const int y0 = job.y_lo + row0;
    constexpr int kRowBytes = LS * 8;
    const int prows_full = (kTile + cby) / 2 + 1;
    const int max_pieces = (prows_full * kRowBytes + 1023) >> 10;
    uint32_t* sm_cells = reinterpret_cast<uint32_t*>(sm_tile);
    uint32_t* lpb = sm_cells + max_pieces * 256;                 /* [kJRec] entry words */
    float4* ftab = reinterpret_cast<float4*>(lpb + kJRec);       /* [kJRec] float beam counts */
    const int tb = lane_on || (idle && g < groups && dxi < cbx) ? (g * (R / 2)) * kRowBytes + 8 * dxi : 0;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool wave_live = __builtin_amdgcn_ballot_w64(lane_on && bx * cbx + dxi < job.nx &&
                                                       row0 + g * R < job.ny) != 0;

    f32x2 ea[R / 2], oa[R / 2 + 1], eb[R / 2], ob[R / 2 + 1];
#pragma unroll
    for (int i = 0; i < R / 2; ++i)
        ea[i] = eb[i] = f32x2{ 0.f, 0.f };
#pragma unroll
    for (int i = 0; i <= R / 2; ++i)
        oa[i] = ob[i] = f32x2{ 0.f, 0.f };

    const int ntiles = __builtin_amdgcn_readfirstlane(job.n_tiles[bid_y]);
    const TileRec* recs = job.tiles + (size_t)bid_y * job.max_tiles;
    const uint32_t* __restrict__ pbs = job.sorted_pb + (size_t)bid_y * 2 * job.n_points;
    const size_t xg_pitch = (size_t)job.xg_pitch;
    const size_t xg_row_bytes = xg_pitch * 8;
    const char* xg = reinterpret_cast<const char*>(job.xgf);
    const int pad = job.xg_pad;

    constexpr int kMaxP = ((((kTile + kPairMaxCby) / 2 + 1) * kRowBytes + 1023) / 1024 + 7) / 8;
    uint32_t goff[kMaxP];
#pragma unroll
    for (int k = 0; k < kMaxP; ++k) {
        const uint32_t ob_ = (uint32_t)(wave + 8 * k) * 1024u + (uint32_t)lane * 16u;
        const uint32_t prow = ob_ / (uint32_t)kRowBytes, cb = ob_ - prow * (uint32_t)kRowBytes;
        goff[k] = prow * (uint32_t)xg_row_bytes + cb;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;

    TileRec rec;
    if (ntiles > 0)
        rec = recs[0];
    for (int ti = 0; ti < ntiles; ++ti) {
        const int c00 = __builtin_amdgcn_readfirstlane(rec.c0) + x0;
        const int gr0 = __builtin_amdgcn_readfirstlane(rec.r0) + y0;
        const int a = c00 & 1;
        const int cnt = __builtin_amdgcn_readfirstlane((int)rec.count);
        const int start = __builtin_amdgcn_readfirstlane((int)rec.start);
        const int nprows = (__builtin_amdgcn_readfirstlane(rec.h) + cby) >> 1;
        const int npieces = (nprows * kRowBytes + 1023) >> 10;
        const char* src = xg + ((size_t)((gr0 + pad) >> 1) * xg_pitch + (size_t)((c00 & ~1) + pad)) * 8;
        if (ti + 1 < ntiles)
            rec = recs[ti + 1];
        /* this thread's entry of the record, for the float table */
        uint32_t my_w = 0;
        if (tid < cnt)
            my_w = pbs[start + tid];
        __syncthreads();                                 /* previous tile consumed */
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int k = 0; k < kMaxP; ++k) {
            const int pc = wave + 8 * k;
            if (pc < npieces)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src + goff[k]), (lds_ptr)(sm_cells + pc * 256), 16, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        static_assert(kJRec <= kBlock, "one table entry per thread");
        if (tid < cnt) {
            lpb[tid] = my_w;
            ftab[tid] = make_float4((float)((my_w >> 16) & 15u), (float)((my_w >> 20) & 15u),
                                    (float)((my_w >> 24) & 15u), (float)(my_w >> 28));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t lane_addr = lds_address(sm_cells) + (uint32_t)(tb + 8 * a);
#ifndef CSM_ABL_NOGATHER       /* timing builds only (tools/build_variant.sh): what the tile loop costs without its gather */
        if (wave_live)
            joint_gather_f<LS, R>(lane_addr, lds_address(ftab), lpb, lane, cnt, ea, oa, eb, ob);
#else
        if (wave_live && cnt < 0)
            joint_gather_f<LS, R>(lane_addr, lds_address(ftab), lpb, lane, cnt, ea, oa, eb, ob);
#endif
    }

    /* this lane's candidates: column bx * cbx + dxi, rows row0 + g * R + r */
    const int xi = bx * cbx + dxi;
    float best0 = 0.f, best1 = 0.f;
    float* const dump_f = job.dump_f;
    if (lane_on && xi < job.nx) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yi = row0 + g * R + r;
            if (yi >= job.ny)
                continue;
            const float w0 = (r & 1 ? ea[r / 2].y + oa[(r + 1) / 2].x : ea[r / 2].x + oa[r / 2].y);
            const float w1 = (r & 1 ? eb[r / 2].y + ob[(r + 1) / 2].x : eb[r / 2].x + ob[r / 2].y);
            best0 = fmaxf(best0, w0);
            best1 = fmaxf(best1, w1);
            if (dump_f) {
                dump_f[((size_t)t0 * job.nx + xi) * job.ny + yi] = w0;
                if (two)
                    dump_f[((size_t)t1 * job.nx + xi) * job.ny + yi] = w1;
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        best0 = fmaxf(best0, __shfl_xor(best0, m, 64));
        best1 = fmaxf(best1, __shfl_xor(best1, m, 64));
    }
    if (lane == 0) {
        red[0][wave] = best0;
        red[1][wave] = best1;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            best0 = fmaxf(best0, red[0][w]);
            best1 = fmaxf(best1, red[1][w]);
        }
        const int cbg = bid_x + bb.cb_base;
        job.approx_best[(size_t)t0 * bb.ncb + cbg] = best0;
        if (two)
            job.approx_best[(size_t)t1 * bb.ncb + cbg] = best1;
    }
}

template <int LS, int R>
__global__ __launch_bounds__(kBlock, 4) void k_score_jointf_batch(const ScoreJob* jobs, int cbx, int groups,
                                                                  const uint16_t* lane_map, int xcd_map, BlockBase bb)
{
    int bx, by, bz;
    xcd_block(xcd_map, bx, by, bz);
    score_body_jointf<LS, R>(jobs[bz], cbx, groups, lane_map, bx, by, bb);
}

/* grid = (candidate blocks, ceil(theta slices / 2), jobs) */
template <int LS, int R>
__global__ __launch_bounds__(kBlock, 4) void k_score_joint_batch(const ScoreJob* jobs, int cbx, int groups,
                                                                 const uint16_t* lane_map, int xcd_map, BlockBase bb)
{
    int bx, by, bz;
    xcd_block(xcd_map, bx, by, bz);
    score_body_joint<LS, R>(jobs[bz], cbx, groups, lane_map, bx, by, bb);
}

/* One window, its job by value: grid.x = pairs of slices x candidate blocks, the pairs fastest
 * (neighbouring workgroups stage the same neighbourhood of the map: L2 hits, as k_score_pairs'
 * slice-major order). The coarse pass of a coarse-first search (csm_phase_kernels.hip) runs on it:
 * on the phase-major copy a tile holds the few beams of ONE phase, staging dominates, and a pair of
 * slices shares every staged window. */
template <int LS, int R>
__global__ __launch_bounds__(kBlock, 4) void k_score_joint_one(ScoreJob job, int cbx, int groups,
                                                               const uint16_t* lane_map, int n_pairs, BlockBase bb)
{
    const int by = (int)(blockIdx.x % (uint32_t)n_pairs), bx = (int)(blockIdx.x / (uint32_t)n_pairs);
    score_body_joint<LS, R>(job, cbx, groups, lane_map, bx, by, bb);
}

/* ---- after the bound pass: which candidate blocks the exact kernel still has to score ----
 * One workgroup per job (window). M = the window's greatest fp32 key. A block (pair of slices,
 * candidate block) whose own greatest key lies below M by more than the two passes' rounding
 * (approx_slack) can hold neither the winner nor a candidate that ties with it: its BlockBest
 * records stay "no candidate". Every other block becomes an item of the work list its launch
 * reads (list 0: row blocks of the R = 8 launch, list 1: the window's last row block when that is
 * an R = 6 launch). All blocks are kept when a beam can reach the negative edge band (eligibility
 * against the coarser level then decides, which the bound pass ignores) or when every candidate's
 * sums are wanted (dump_s / dump_k). item = job << 18 | pair << 8 | block.
 *
 * Blocks whose maximum lies below the score threshold's key (key_floor) can hold no candidate that
 * is reported as found and are never scored.
 *
 * round 2 (branch and bound). There the winner must also pass its own known-count test, which the
 * bound pass does not see: M may belong to a leaf that does not count, and the best ELIGIBLE leaf
 * may lie below M's margin. After round 1 (blocks near M, scored exactly, reduced by k_finalize)
 * the query's record holds the best eligible exact key B so far; a leaf can only reach or beat B
 * with an fp32 key >= B (1 - slack), so round 2 lists the blocks not scored yet whose maximum
 * reaches that (and the floor). After it no unscored block can hold the winner or a tie with it:
 * two rounds always suffice. Blocks listed in round 1 are marked by a negative approx_best. */
__global__ __launch_bounds__(256) void k_bound_select(const ScoreJob* jobs, int ncb, int split_cb,
                                                     uint32_t* items0, uint32_t* items1, uint32_t* counts,
                                                     uint32_t cap, int round)
{
    __shared__ float wmax[4];
    const ScoreJob& job = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const int n_theta = job.n_theta;
    const int total = n_theta * ncb;
    float* ab = job.approx_best;
    float m = 0.f;
    if (round == 1) {
        for (int i = tid; i < total; i += 256) {
            m = fmaxf(m, ab[i]);
            BlockBest none;
            none.key = 0;
            none.rank = ~0ull;
            none.count = 0;
            none.pad = 0;
            job.block_best[i] = none;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((tid & 63) == 0)
        wmax[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float thr = fmaxf(m * (1.0f - job.approx_slack), job.key_floor);
    const bool all_blocks = job.dump_s || job.dump_k || (job.elig_only_if_band && (*job.flags & kFlagBandTouch));
    const bool all = all_blocks && round == 1;
    if (round == 2) {
        if (all_blocks)
            return;                             /* everything was scored in round 1 */
        const unsigned long long best = reinterpret_cast<const csm_result*>(job.round1_record)->key;
        thr = fmaxf((float)best * (1.0f - job.approx_slack), job.key_floor);
    }
    const int n_pairs = (n_theta + 1) / 2;
    uint32_t kept = 0, dropped = 0;
    for (int i = tid; i < n_pairs * ncb; i += 256) {
        const int pr = i / ncb, cb = i - pr * ncb;
        const int t0 = 2 * pr;
        const float a0 = ab[(size_t)t0 * ncb + cb], a1 = t0 + 1 < n_theta ? ab[(size_t)(t0 + 1) * ncb + cb] : 0.f;
        if (a0 < 0.f)
            continue;                           /* scored in round 1 */
        const float mine = fmaxf(a0, a1);
        if (all || mine >= thr) {
            if (round == 1)
                ab[(size_t)t0 * ncb + cb] = -1.0f;
            const int which = cb >= split_cb ? 1 : 0;
            const uint32_t pos = atomicAdd(counts + which, 1u);
            if (pos < cap)
                (which ? items1 : items0)[pos] = ((uint32_t)blockIdx.x << 18) | ((uint32_t)pr << 8) | (uint32_t)cb;
            ++kept;
        } else {
            ++dropped;
        }
    }
    /* one atomic per counter and workgroup: same-address atomics serialise at the memory side
     * (63,000 of them cost 360 us per launch) */
    if (job.bound_stats) {
        __shared__ uint32_t tot[2];
        if (tid == 0)
            tot[0] = tot[1] = 0u;
        __syncthreads();
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            kept += __shfl_xor(kept, d, 64);
            dropped += __shfl_xor(dropped, d, 64);
        }
        if ((tid & 63) == 0) {
            atomicAdd(&tot[0], kept);
            atomicAdd(&tot[1], dropped);
        }
        __syncthreads();
        if (tid == 0) {
            if (tot[0])
                atomicAdd(job.bound_stats, tot[0]);
            if (round == 1 && tot[1])
                atomicAdd(job.bound_stats + 1, tot[1]);
            if (round == 2 && tot[0])
                atomicSub(job.bound_stats + 1, tot[0]);     /* counted as skipped after round 1 */
        }
    }
}

/* The exact kernel over a work list: a fixed grid of workgroups takes the items i = blockIdx.x,
 * blockIdx.x + gridDim.x, ... below *count (written by k_bound_select earlier on the stream). */
template <int LS, int R>
__global__ __launch_bounds__(kBlock, 4) void k_score_joint_list(const ScoreJob* jobs, int cbx, int groups,
                                                                const uint16_t* lane_map, BlockBase bb,
                                                                const uint32_t* items, const uint32_t* count)
{
    const uint32_t n = __builtin_amdgcn_readfirstlane((int)*count);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[i]);
        score_body_joint<LS, R>(jobs[it >> 18], cbx, groups, lane_map, (int)(it & 255u) - bb.cb_base,
                                (int)((it >> 8) & 1023u), bb);
        __syncthreads();
    }
}

} /* namespace csm */

/* ------------------------------------------------------------------ host launchers */

namespace {

/* Dynamic LDS above 64 KB needs the function attribute; process-wide, only ever raised. */
hipError_t grant_lds(int device, const void* fn, size_t bytes)
{
    if (bytes <= 64 * 1024)
        return hipSuccess;
    static std::mutex guard;
    static std::map<std::pair<int, const void*>, size_t> granted;
    std::lock_guard<std::mutex> lock(guard);
    size_t& have = granted[{ device, fn }];
    if (bytes > have) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess)
            return e;
        have = bytes;
    }
    return hipSuccess;
}

template <int LS, int R>
hipError_t launch_joint(const csm::JointLaunch& L)
{
    if (L.items) {
        auto list_kernel = csm::k_score_joint_list<LS, R>;
        const hipError_t e = grant_lds(L.device, reinterpret_cast<const void*>(list_kernel), L.lds_bytes);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(list_kernel, dim3(L.list_blocks), dim3(csm::kBlock), L.lds_bytes, L.stream, L.jobs_dev,
                           L.cbx, L.groups, L.lane_map, csm::BlockBase{ L.row_base, L.cb_base, L.ncb }, L.items,
                           L.item_count);
        return hipGetLastError();
    }
    auto kernel = L.fp32 ? csm::k_score_jointf_batch<LS, R> : csm::k_score_joint_batch<LS, R>;
    const hipError_t e = grant_lds(L.device, reinterpret_cast<const void*>(kernel), L.lds_bytes);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(kernel, L.grid, dim3(csm::kBlock), L.lds_bytes, L.stream, L.jobs_dev, L.cbx, L.groups,
                       L.lane_map, L.xcd_map, csm::BlockBase{ L.row_base, L.cb_base, L.ncb });
    return hipGetLastError();
}

template <int LS, int R>
hipError_t launch_joint_one_t(const csm::JointLaunch& L, const csm::ScoreJob& job, int n_pairs)
{
    auto kernel = csm::k_score_joint_one<LS, R>;
    const hipError_t e = grant_lds(L.device, reinterpret_cast<const void*>(kernel), L.lds_bytes);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_pairs * L.grid.x), dim3(csm::kBlock), L.lds_bytes, L.stream, job, L.cbx,
                       L.groups, L.lane_map, n_pairs, csm::BlockBase{ L.row_base, L.cb_base, L.ncb });
    return hipGetLastError();
}

} /* namespace */

namespace csm {

/* load factor <= 3/4 when every beam of the two slices lands on a cell of its own (in practice a quarter
 * of that: 533 cells of 2160 beams for configs[1]); < 2^16 slots: the list holds 16-bit slot numbers */
int binj_hash_size(int n_points)
{
    const int h = ((8 * n_points + 2) / 3 + 63) & ~63;
    return h < 512 ? 512 : h;
}

size_t binj_lds_bytes(int tiles, int n_points, int hash_size)
{
    const size_t ntp = (size_t)((tiles + 1) & ~1);
    return 20 * ntp + 12 * (size_t)hash_size + 2 * (size_t)(2 * n_points) + 16;
}

int launch_binj_batch(hipStream_t stream, int device, const BinJob* jobs_dev, int n_pairs_max, int n_jobs,
                      size_t lds_bytes)
{
    const hipError_t e = grant_lds(device, reinterpret_cast<const void*>(k_binj_batch), lds_bytes);
    if (e != hipSuccess)
        return (int)e;
    hipLaunchKernelGGL(k_binj_batch, dim3(n_pairs_max, n_jobs), dim3(kBinjBlock), lds_bytes, stream, jobs_dev);
    return (int)hipGetLastError();
}

int launch_binj_one(hipStream_t stream, int device, const BinJob& job, int n_pairs, size_t lds_bytes)
{
    const hipError_t e = grant_lds(device, reinterpret_cast<const void*>(k_binj_one), lds_bytes);
    if (e != hipSuccess)
        return (int)e;
    hipLaunchKernelGGL(k_binj_one, dim3(n_pairs, 1), dim3(kBinjBlock), lds_bytes, stream, job);
    return (int)hipGetLastError();
}

#define JOINT_CASE(LS)                                                                 \
    if (L.ls == LS && L.R == 8)                                                        \
        return (int)launch_joint<LS, 8>(L);                                            \
    if (L.ls == LS && L.R == 6)                                                        \
        return (int)launch_joint<LS, 6>(L);
#define JOINT_ONE_CASE(LS)                                                             \
    if (L.ls == LS && L.R == 8)                                                        \
        return (int)launch_joint_one_t<LS, 8>(L, job, n_pairs);                        \
    if (L.ls == LS && L.R == 6)                                                        \
        return (int)launch_joint_one_t<LS, 6>(L, job, n_pairs);

int launch_expand_pairs_f(hipStream_t stream, const uint16_t* cells, int rows, int cols, int pitch, float* xgf,
                          int xg_prows, int xg_pitch, int pad)
{
    const size_t total = (size_t)xg_prows * xg_pitch;
    const int blocks = (int)(total + 255 < 4096 * 256 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_expand_pairs_f, dim3(blocks), dim3(256), 0, stream, cells, rows, cols, pitch,
                       reinterpret_cast<float2*>(xgf), xg_prows, xg_pitch, pad);
    return (int)hipGetLastError();
}

int launch_bound_select(hipStream_t stream, const ScoreJob* jobs_dev, int n_jobs, int ncb, int split_cb,
                        uint32_t* items0, uint32_t* items1, uint32_t* counts, uint32_t cap, int round)
{
    hipLaunchKernelGGL(k_bound_select, dim3(n_jobs), dim3(256), 0, stream, jobs_dev, ncb, split_cb, items0, items1,
                       counts, cap, round);
    return (int)hipGetLastError();
}

int launch_joint_batch(const JointLaunch& L)
{
#ifdef CSM_FAST_BUILD
    JOINT_CASE(150) JOINT_CASE(156)
#else
    JOINT_CASE(86) JOINT_CASE(98) JOINT_CASE(118) JOINT_CASE(124) JOINT_CASE(130) JOINT_CASE(150)
    JOINT_CASE(156) JOINT_CASE(162) JOINT_CASE(182)
#endif
    return -1;      /* no instantiation for this row pitch */
}

int launch_joint_one(const JointLaunch& L, const ScoreJob& job, int n_pairs)
{
#ifdef CSM_FAST_BUILD
    JOINT_ONE_CASE(150) JOINT_ONE_CASE(156)
#else
    JOINT_ONE_CASE(86) JOINT_ONE_CASE(98) JOINT_ONE_CASE(118) JOINT_ONE_CASE(124) JOINT_ONE_CASE(130) JOINT_ONE_CASE(150)
    JOINT_ONE_CASE(156) JOINT_ONE_CASE(162) JOINT_ONE_CASE(182)
#endif
    return -1;
}

} /* namespace csm */
