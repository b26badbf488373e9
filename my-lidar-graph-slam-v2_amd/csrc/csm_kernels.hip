/* csm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the
 * correlative scan-matching hot path.
 *
 * K0 k_bin       merges the beams of one theta slice into (cell, multiplicity)
 *                entries (LDS hash table), sorts them by 64x64-cell endpoint
 *                tile, emits packed LDS offsets per entry.
 * K1 k_score     one workgroup = (theta slice, block of candidate offsets):
 *                for every non-empty endpoint tile it stages tile + window halo
 *                of the uint16 grid into LDS with 16-byte coalesced loads, then
 *                every lane walks the tile's beams and gathers from LDS for its
 *                own R candidate offsets, accumulating exact integer (S, K).
 *                Ends in a wave64 shuffle arg-max, one record per workgroup.
 *                Replaces ComputeScore inside the sweep of
 *                src/mapping/scan_matcher_correlative.cpp:161-197, 301-368 and
 *                ScorePixelAccurate::Score per branch-and-bound node
 *                (src/mapping/score_function_pixel_accurate.cpp:16-58).
 * K2 k_boxmax_batch  forward box maximum with the reference's "repeat the last
 *                window" edge rule (inc/util.hpp:369-424,
 *                src/mapping/grid_map_builder.cpp:918-984), all levels of all
 *                maps of a call in one launch.
 * K4 k_finalize  reduces the workgroup records, replays the winner in f64 in
 *                beam order (bit-exact scoreMax), writes the result record.
 *
 * Integer order key: sum P = (0.998/65534/499) * (32268*K + 499*S) with
 * S = sum of raw values over known cells, K = known count, so the u64 key
 * orders candidates exactly like the reference's double sum whenever keys
 * differ.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "csm_device.hpp"
#include "../../include/csm_hip.h"
#include "csm_score_common.hpp"

namespace csm {


#ifdef CSM_BIN_TIMING
/* tuning builds only: cycles per phase of k_bin, one row of 8 counters per workgroup
 * (thread 0, plain stores); the host points BinJob.tuning_counters at
 * kBinDebugRows rows */
#define BIN_TICK(k)                                                                   \
    do {                                                                              \
        if (threadIdx.x == 0 && job.tuning_counters) {                                         \
            const unsigned long long now_ = __builtin_readcyclecounter();             \
            reinterpret_cast<unsigned long long*>(job.tuning_counters)[dbg_row_ * 16 + (k)] = now_ - tick_; \
            tick_ = now_;                                                             \
        }                                                                             \
    } while (0)
#define BIN_SUB(k)                                                                    \
    do {                                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter();                 \
        sub_[k] += now_ - subt_;                                                      \
        subt_ = now_;                                                                 \
    } while (0)
#else
#define BIN_TICK(k) do { } while (0)
#define BIN_SUB(k) do { } while (0)
#endif

/* ------------------------------------------------------------------ K0 */
/* One workgroup (kBinBlock threads) per theta slice.
 *
 * Entries. Beams that land on the same cell of this slice add the same value to
 * every candidate, so they are merged through an LDS hash table into (cell,
 * multiplicity <= kMaxMult) entries. In pair mode (the pair-row fine kernel)
 * the unit is an ALIGNED ROW PAIR of one column -- rows (2k, 2k + 1) of the
 * tile frame -- with one multiplicity per row (m_even, m_odd): the fine kernel
 * reads both rows with one ds_read_b64, so a vertical run of hit cells costs
 * about half the LDS reads. Entries of a tile are sorted by class (both rows
 * hit / even row only / odd row only) so that the gather loops have no
 * per-entry branch; TileRec.pad[0] holds the first two class counts (16 bits
 * each), pad[1] the tile's number and the record's chunk number within it.
 *
 * Entry words:
 *   sorted_pb, single mode: mult << 16 | (row * lstride + col)
 *   sorted_pb, pair mode:   m_odd << 28 | m_even << 24 | (m_even + m_odd) << 19 |
 *                           byte offset of the slot = (pair_row * lstride + col) * 8
 *   sorted_rc (strided levels): m_odd << 28 | m_even << 24 | row << 16 | col
 * rows / cols relative to the tile's bounding box (TileRec.r0 / c0); in pair
 * mode r0 is rounded down to an even row of the tile frame, and the frame is
 * shifted by BinJob.frame_shift so that an even frame row + any candidate row
 * offset of the fine kernel's lanes is an even GRID row (the pair-row copy of
 * the grid pairs rows by their grid parity).
 *
 * How (round 2, 192 -> 92 us per 64-window launch of 7,744 slices): the kernel is
 * bound by instruction issue and LDS round trips, not by bandwidth -- LDS atomics
 * themselves run at 15-24 lanes per clock and CU (tools/micro/lds_atomic_bench.hip).
 * So: runs of neighbouring lanes with one key are inserted once with the run's
 * beam counts; the probes of all iterations of a thread travel together, one LDS
 * round trip per probe step; the lanes that claim an empty slot record it in a
 * per-wave list (no shared counter) and the later passes walk the distinct cells,
 * not the table; a tile's three class counts share one 64-bit word and its
 * bounding box is two 64-bit bit masks (3 atomics per cell, were 7); the cursor
 * of pass C is one returning 64-bit add; the edge-band test is skipped by whole
 * waves away from the map's low edges. */
__device__ __forceinline__ void k_bin_body(const BinJob& job)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_bin[];
    /* the job's fields are the same for every lane, but loaded through a reference the
     * compiler treats them as per-lane values: readfirstlane keeps the loop bounds, the
     * table sizes and everything derived from them in scalar registers and branches */
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int tiles_x = uni(job.tiles_x);
    const int ntile = tiles_x * uni(job.tiles_y);
    const int hash_size = uni(job.hash_size);
    const int n_theta_job = uni(job.n_theta);
    /* per tile: entry counts (later the cursors) of the three classes, 21 bits each,
     * in one 64-bit word -- total | both << 21 | even-only << 42 -- and the hit rows /
     * columns of the tile as bit masks (bounding box = lowest / highest set bit): three
     * 64-bit LDS atomics per distinct cell where seven 32-bit ones were needed */
    const int ntp = (ntile + 1) & ~1;    /* keeps the hash table 16-byte aligned */
    unsigned long long* cnt64 = reinterpret_cast<unsigned long long*>(sm_bin);   /* [ntp] */
    unsigned long long* rowmask = cnt64 + ntp;                                    /* [ntp] */
    unsigned long long* colmask = rowmask + ntp;                                  /* [ntp] */
    uint32_t* hkey = reinterpret_cast<uint32_t*>(colmask + ntp);   /* [hash_size] (tile, cell) + 1, 0 = empty */
    uint32_t* hval = hkey + hash_size;   /* [hash_size] beams on that cell: even row | odd row << 16 */
    /* occupied slots (< 32768), [n_points]: a segment per wave */
    uint16_t* list = reinterpret_cast<uint16_t*>(hval + hash_size);
    __shared__ uint32_t seg_n[kBinBlock / 64];   /* cells claimed by each wave */
    const int t = blockIdx.x;
    if (t >= n_theta_job)
        return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n = uni(job.n_points);
    const int32_t* col = job.hit_col + (size_t)t * n;
    const int32_t* row = job.hit_row + (size_t)t * n;
    const uint32_t hmask = (uint32_t)hash_size - 1u;
    const bool pairs = uni(job.pair_mode) != 0;
#ifdef CSM_BIN_TIMING
    unsigned long long tick_ = __builtin_readcyclecounter();
    const size_t dbg_row_ = min((size_t)blockIdx.y * gridDim.x + blockIdx.x, (size_t)kBinDebugRows - 1);
    if (threadIdx.x == 0 && job.tuning_counters)
        reinterpret_cast<unsigned long long*>(job.tuning_counters)[dbg_row_ * 16 + 7] = 1ull;
#endif

    /* The kernel is bound by latencies, not by throughput: a thread's beams (i = tid,
     * tid + kBinBlock, ...) are loaded kBinAhead iterations at a time so that pass A
     * waits for memory once per chunk (one load per iteration, even prefetched one
     * iteration ahead, was 58 % of the kernel); the first chunk is in flight while
     * the tables are cleared. */
    constexpr int kBinAhead = 6;
    int rv[kBinAhead], cv[kBinAhead];
#pragma unroll
    for (int u = 0; u < kBinAhead; ++u) {
        const int i = u * kBinBlock + tid;
        rv[u] = cv[u] = 0;
        if (i < n) {
            rv[u] = row[i];
            cv[u] = col[i];
        }
    }
    for (int i = tid; i < 3 * ntp; i += kBinBlock)
        cnt64[i] = 0ull;
    {
        uint4* h4 = reinterpret_cast<uint4*>(hkey);      /* hkey and hval are contiguous */
        for (int i = tid; i < hash_size / 2; i += kBinBlock)
            h4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    BIN_TICK(0);

    /* Pass A: count the beams per (tile, cell) -- per (tile, row pair, column) in pair
     * mode. Neighbouring beams are neighbouring lanes and mostly land on the same or
     * the next cell: a run of lanes with one key is inserted once, by its first lane,
     * with the run's beam counts (same-address LDS atomics of a wave serialise). The
     * lane that claims an empty slot appends it to `list`: passes B and C walk the
     * distinct cells, not the table. */
    const int r_max = uni(job.rows) - 1 - uni(job.y_lo);
    const int c_max = uni(job.cols) - 1 - uni(job.x_lo);
    bool band = false;
    const int fs = uni(job.frame_shift);     /* 0 / 1: see BinJob */
    const int x_hi = uni(job.x_hi), y_hi = uni(job.y_hi), x_lo = uni(job.x_lo), y_lo = uni(job.y_lo);
    const int n_band = uni(job.n_band);
    const int n_iter = (n + kBinBlock - 1) / kBinBlock;
#ifdef CSM_BIN_TIMING
    unsigned long long sub_[4] = { 0, 0, 0, 0 }, subt_ = 0;
#endif
    /* edge-band test of one coordinate, division-free for the beams that cannot be in
     * the band (all but those within one window of the map's low edge): band_hit() */
    auto in_band = [&](int u, int w, int span, int known_lo) {
        if (u > 0 || u <= -span)             /* k0 = floor(-u / w) outside [0, nk) */
            return false;
        const int m = (-u) % w;              /* v = -m */
        return m != 0 && w - 1 - m >= known_lo;
    };
    const int known_r0 = uni(job.known_r0), known_c0 = uni(job.known_c0);
    const int wave = tid >> 6;
    /* a wave can claim one slot per beam it handles (i = it * kBinBlock + wave * 64 + lane):
     * its segment of the list starts after the beams of the waves before it */
    auto seg_base = [&](int w) { return (n / kBinBlock) * 64 * w + min(n % kBinBlock, 64 * w); };
    uint16_t* my_list = list + seg_base(wave);
    uint32_t my_count = 0;                   /* wave-uniform */
    for (int it0 = 0; it0 < n_iter; it0 += kBinAhead) {
        if (it0) {
#pragma unroll
            for (int u = 0; u < kBinAhead; ++u) {
                const int i = (it0 + u) * kBinBlock + tid;
                if (i < n) {
                    rv[u] = row[i];
                    cv[u] = col[i];
                }
            }
        }
#ifdef CSM_BIN_TIMING
        subt_ = __builtin_readcyclecounter();
#endif
        /* keys, run heads and the beam counts of the runs */
        uint32_t key[kBinAhead], slot[kBinAhead], beams[kBinAhead];
        bool pend[kBinAhead], first[kBinAhead];
#pragma unroll
        for (int u = 0; u < kBinAhead; ++u) {
            key[u] = 0xffffffffu;
            slot[u] = beams[u] = 0;
            pend[u] = first[u] = false;
            if (it0 + u >= n_iter)          /* uniform */
                continue;
            const int i = (it0 + u) * kBinBlock + tid;
            const int r = rv[u], c = cv[u];
            const int rr = r + y_hi + fs, cc = c + x_hi;
            const bool valid = i < n && rr >= fs && r <= r_max && cc >= 0 && c <= c_max;
            bool odd = false;
            if (valid) {               /* rr, cc >= 0 */
                const uint32_t tile = ((uint32_t)rr / kTile) * (uint32_t)tiles_x + (uint32_t)cc / kTile;
                const uint32_t rb = (uint32_t)rr % kTile, cb = (uint32_t)cc % kTile;
                const uint32_t rkey = pairs ? rb >> 1 : rb;
                key[u] = ((tile << 12) | (rkey << 6) | cb) + 1u;
                odd = pairs && (rb & 1u);
            }
            const uint32_t prev = (uint32_t)__shfl_up((int)key[u], 1, 64);
            const bool head = valid && (lane == 0 || key[u] != prev);
            const unsigned long long hm = __ballot(head), vm = __ballot(valid), om = __ballot(odd);
            /* the run ends before the next head or the next lane without a cell */
            const unsigned long long above = (hm | ~vm) & ~((2ull << lane) - 1ull);
            const int e = above ? __builtin_ctzll(above) : 64;
            const unsigned long long run = (e == 64 ? ~0ull : (1ull << e) - 1ull) & ~((1ull << lane) - 1ull);
            const uint32_t co = (uint32_t)__popcll(run & om), ce = (uint32_t)__popcll(run) - co;
            beams[u] = ce | (co << 16);
            slot[u] = (key[u] * 2654435761u) >> 12 & hmask;
            pend[u] = head;
            /* every band lies at candidate offsets u <= 0: whole waves of beams away from
             * the map's low edges skip the test (and the scalar loads of its parameters) */
            const bool near_low_edge = i < n && (r + y_lo <= 0 || c + x_lo <= 0);
            if (__any(near_low_edge)) {
                if (near_low_edge)
                    for (int b = 0; b < n_band; ++b) {
                        const int w = job.band_win[b];
                        if (in_band(r + y_lo, w, job.band_ny[b] * w, known_r0) ||
                            in_band(c + x_lo, w, job.band_nx[b] * w, known_c0))
                            band = true;
                    }
            }
        }
        BIN_SUB(0);
        /* insertion: the probes of the chunk's iterations travel together, one LDS
         * round trip per probe step instead of one per step and iteration */
        bool any = true;
        while (any) {
            uint32_t old[kBinAhead];
#pragma unroll
            for (int u = 0; u < kBinAhead; ++u)
                old[u] = pend[u] ? atomicCAS(&hkey[slot[u]], 0u, key[u]) : 0u;
            any = false;
#pragma unroll
            for (int u = 0; u < kBinAhead; ++u)
                if (pend[u]) {
                    if (old[u] == 0u || old[u] == key[u]) {
                        first[u] = old[u] == 0u;
                        atomicAdd(&hval[slot[u]], beams[u]);
                        pend[u] = false;
                    } else {
                        slot[u] = (slot[u] + 1u) & hmask;
                        any = true;
                    }
                }
            any = __any(any);
        }
        BIN_SUB(1);
        /* the lanes that claimed an empty slot append it to the wave's segment of the list */
#pragma unroll
        for (int u = 0; u < kBinAhead; ++u) {
            const unsigned long long fm = __ballot(first[u]);
            if (first[u])
                my_list[my_count + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = (uint16_t)slot[u];
            my_count += (uint32_t)__popcll(fm);
        }
        BIN_SUB(2);
    }
    if (lane == 0)
        seg_n[wave] = my_count;
    if (band)
        atomicOr(job.flags, kFlagBandTouch);
#ifdef CSM_BIN_TIMING
    if (threadIdx.x == 0 && job.tuning_counters)
        for (int k = 0; k < 4; ++k)
            reinterpret_cast<unsigned long long*>(job.tuning_counters)[dbg_row_ * 16 + 8 + k] = sub_[k];
#endif
    __syncthreads();
    BIN_TICK(1);

    /* the host decides per query whether merging pays (job.max_mult 1 = off) */
    const uint32_t max_mult = (uint32_t)uni(job.max_mult);
    auto chunks = [&](uint32_t beams) { return (beams + max_mult - 1u) / max_mult; };
    static_assert(kBinBlock == 256, "four list segments");
    const uint32_t seg1 = (uint32_t)uni((int)seg_n[0]), seg2 = seg1 + (uint32_t)uni((int)seg_n[1]),
                   seg3 = seg2 + (uint32_t)uni((int)seg_n[2]);
    const int n_cells = uni((int)(seg3 + seg_n[3]));
    auto cell_slot = [&](int e) {
        const uint32_t ue = (uint32_t)e;
        const uint32_t sg = (ue >= seg1) + (ue >= seg2) + (ue >= seg3);
        const uint32_t first_of = sg == 0 ? 0u : sg == 1 ? seg1 : sg == 2 ? seg2 : seg3;
        return (uint32_t)list[(uint32_t)seg_base((int)sg) + (ue - first_of)];
    };

    /* Pass B: entries per tile and class (a cell with more than max_mult beams is
     * split) and the tile's bounding box, from the distinct cells */
    for (int e = tid; e < n_cells; e += kBinBlock) {
        const uint32_t sl = cell_slot(e);
        const uint32_t k1 = hkey[sl] - 1u, hv = hval[sl];
        const uint32_t ce = chunks(hv & 0xffffu), co = chunks(hv >> 16);
        const uint32_t both = min(ce, co);
        const int tile = (int)(k1 >> 12);
        const uint32_t rk = (k1 >> 6) & 63u, cbk = k1 & 63u;
        const uint32_t rlo = pairs ? 2u * rk + (ce ? 0u : 1u) : rk;
        const uint32_t rhi = pairs ? 2u * rk + (co ? 1u : 0u) : rk;
        atomicAdd(&cnt64[tile], (unsigned long long)max(ce, co) | ((unsigned long long)both << 21) |
                                    ((unsigned long long)(ce - both) << 42));
        atomicOr(&rowmask[tile], (1ull << rlo) | (1ull << rhi));
        atomicOr(&colmask[tile], 1ull << cbk);
    }
    __syncthreads();
    BIN_TICK(2);

    /* exclusive scan of entry counts and of the records per tile */
    const int chunk = (ntile + kBinBlock - 1) / kBinBlock;
    const int lo = tid * chunk, hi = min(lo + chunk, ntile);
    uint32_t cnt = 0, ne = 0, excl_cnt, excl_rec;
    for (int i = lo; i < hi; ++i) {
        const uint32_t c = (uint32_t)cnt64[i] & 0x1fffffu;
        cnt += c;
        ne += (c + kPbMax - 1) / kPbMax;
    }
    {
        /* wave64 shuffles, then the wave totals */
        uint32_t a = cnt, b = ne;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ua = __shfl_up(a, d, 64), ub = __shfl_up(b, d, 64);
            if (lane >= d) {
                a += ua;
                b += ub;
            }
        }
        __shared__ uint32_t wtot[2][kBinBlock / 64];
        if (lane == 63) {
            wtot[0][tid >> 6] = a;
            wtot[1][tid >> 6] = b;
        }
        __syncthreads();
        uint32_t ba = 0, bb2 = 0;
        for (int w = 0; w < (tid >> 6); ++w) {
            ba += wtot[0][w];
            bb2 += wtot[1][w];
        }
        excl_cnt = ba + a - cnt;
        excl_rec = bb2 + b - ne;
        if (tid == kBinBlock - 1)
            job.n_tiles[t] = (int32_t)(bb2 + b);
    }
    BIN_TICK(3);
    /* the records; the tile's word becomes its three cursors: both | even only | odd only */
    {
        uint32_t off = excl_cnt, slot_rec = excl_rec;
        TileRec* recs = job.tiles + (size_t)t * job.max_tiles;
        for (int i = lo; i < hi; ++i) {
            const unsigned long long w = cnt64[i];
            const uint32_t c = (uint32_t)w & 0x1fffffu;
            if (!c)
                continue;
            const uint32_t nb = (uint32_t)(w >> 21) & 0x1fffffu, nev = (uint32_t)(w >> 42);
            const unsigned long long rm = rowmask[i], cm = colmask[i];
            const int rlo = __builtin_ctzll(rm), rhi = 63 - __builtin_clzll(rm);
            const int clo = __builtin_ctzll(cm), chi = 63 - __builtin_clzll(cm);
            const int rmin = pairs ? (rlo & ~1) : rlo;
            for (uint32_t done = 0; done < c; done += kPbMax) {
                TileRec rec;
                rec.r0 = (i / tiles_x) * kTile - y_hi - fs + rmin;
                rec.c0 = (i % tiles_x) * kTile - x_hi + clo;
                rec.start = off + done;
                rec.count = min(c - done, (uint32_t)kPbMax);
                rec.h = rhi - rmin + 1;
                rec.w = chi - clo + 1;
                /* class counts of this chunk: entries [done, done + count) of the tile's
                 * list [both | even only | odd only] */
                const uint32_t end = done + rec.count;
                rec.pad[0] = (int)((min(end, nb) - min(done, nb)) |
                                   ((min(end, nb + nev) - min(max(done, nb), nb + nev)) << 16));
                rec.pad[1] = (int)(((uint32_t)i << 4) | (done / kPbMax));   /* tile, chunk of the tile */
                recs[slot_rec++] = rec;
            }
            cnt64[i] = (unsigned long long)off | ((unsigned long long)(off + nb) << 21) |
                       ((unsigned long long)(off + nb + nev) << 42);
            off += c;
        }
    }
    __syncthreads();
    BIN_TICK(4);

    /* Pass C: the entries */
    uint32_t* out = job.sorted_pb + (size_t)t * n;
    uint32_t* out_rc = job.sorted_rc ? job.sorted_rc + (size_t)t * n : nullptr;
    const uint32_t lstride = (uint32_t)uni(job.lstride);
    for (int e = tid; e < n_cells; e += kBinBlock) {
        const uint32_t sl = cell_slot(e);
        const uint32_t k1 = hkey[sl] - 1u, hv = hval[sl];
        const int tile = (int)(k1 >> 12);
        const uint32_t rlo = (uint32_t)__builtin_ctzll(rowmask[tile]);
        const uint32_t rmin = pairs ? rlo & ~1u : rlo;
        const uint32_t rkey = (k1 >> 6) & 63u;
        const uint32_t rb = (pairs ? rkey << 1 : rkey) - rmin;      /* even in pair mode */
        const uint32_t cb = (k1 & 63u) - (uint32_t)__builtin_ctzll(colmask[tile]);
        uint32_t be = hv & 0xffffu, bo = hv >> 16;
        const uint32_t ce = chunks(be), co = chunks(bo);
        const uint32_t both = min(ce, co);
        const unsigned long long cur = atomicAdd(&cnt64[tile], (unsigned long long)both |
                                                                   ((unsigned long long)(ce - both) << 21) |
                                                                   ((unsigned long long)(co - both) << 42));
        uint32_t pos_b = (uint32_t)cur & 0x1fffffu;
        uint32_t pos_x = ce > both ? (uint32_t)(cur >> 21) & 0x1fffffu : (uint32_t)(cur >> 42);
        const uint32_t total = max(ce, co);
        for (uint32_t k = 0; k < total; ++k) {
            const uint32_t me = min(be, max_mult), mo = min(bo, max_mult);
            be -= me;
            bo -= mo;
            const uint32_t pos = (me && mo) ? pos_b++ : pos_x++;
            if (pairs)      /* byte offset of the slot (8 B) | beams << 19 | m_even << 24 | m_odd << 28 */
                out[pos] = (mo << 28) | (me << 24) | ((me + mo) << 19) | (((rb >> 1) * lstride + cb) << 3);
            else
                out[pos] = (me << 16) | (rb * lstride + cb);
            if (out_rc)
                out_rc[pos] = (mo << 28) | (me << 24) | (rb << 16) | cb;
        }
    }
#ifdef CSM_BIN_TIMING
    __syncthreads();
    BIN_TICK(5);
#endif
}

__global__ __launch_bounds__(kBinBlock, 8) void k_bin(BinJob job)
{
    k_bin_body(job);
}

/* Clears the coarse level's atomic accumulators of a query, but only when some
 * beam can reach the negative edge band (or `always`): otherwise the coarse
 * pass exits without touching them (ScoreJob.skip_unless_band). One workgroup
 * column per query; replaces 27 MB of unconditional clearing per 64-window
 * launch. */
__device__ __forceinline__ void zero_body(const ZeroJob& z)
{
    if (!z.always && !(*z.flags & kFlagBandTouch))
        return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < z.words; i += (size_t)gridDim.x * 256) {
        z.a[i] = 0;
        if (z.b)
            z.b[i] = 0;
    }
}

__global__ __launch_bounds__(256) void k_zero_if_band(ZeroJob job)
{
    zero_body(job);
}

__global__ __launch_bounds__(256) void k_zero_if_band_batch(const ZeroJob* jobs)
{
    zero_body(jobs[blockIdx.y]);
}

/* ------------------------------------------------------------------ K1 */

/* LSTRIDE: LDS row pitch in cells. R: candidate rows per lane.
 * MODE 0: candidates one cell apart; 1: `stride` = 2^k cells apart (coarser
 * levels); 2: any stride (e.g. LowResolutionMapWinSize 5).
 * WEIGHTED: entries carry beam multiplicities (k_bin merged same-cell beams);
 * otherwise every entry is one beam and the multiply is dropped. */
template <int LSTRIDE, int R, int MODE, bool WEIGHTED>
__device__ __forceinline__ void score_body(const ScoreJob& job, int cbx, int groups,
                                           int slice, int n_slices, int n_buf, int t)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t sm_tile[];

    if (t >= job.n_theta)
        return;
    const int tid = threadIdx.x;
    constexpr bool STRIDED = MODE != 0;
    const int stride = STRIDED ? job.stride : 1;
    const int ncbx = (job.nx + cbx - 1) / cbx;
    const int bx = blockIdx.x % ncbx, by = blockIdx.x / ncbx;
    const int cby = groups * R;
    if (by * cby >= job.ny)
        return;

    const uint32_t qflags = (job.skip_unless_band || job.elig_only_if_band) ? *job.flags : 0u;
    if (job.skip_unless_band && !(qflags & kFlagBandTouch))
        return;

    const int dxi = tid % cbx, g = tid / cbx;
    const bool lane_on = g < groups;
    /* first candidate offset of this block, in cells */
    const int x0 = job.x_lo + bx * cbx * stride;
    const int y0 = job.y_lo + by * cby * stride;
    /* LDS: one region of expanded cells (u32), then the beam offsets of the
     * tile (kPbMax words). Stride-1 jobs keep the region row-major. Strided
     * jobs (candidates 2^k cells apart) store it phase-major in both axes --
     * region cell (rho, gamma) lives at row (rho mod s) * hd + rho div s,
     * column (gamma mod s) * wd + gamma div s -- so that the candidates of one
     * beam are contiguous again: conflict-free reads, R rows per lane. */
    const int sh = MODE == 1 ? job.log2_stride : 0;
    auto sdiv = [&](uint32_t x) { return MODE == 1 ? x >> sh : x / (uint32_t)stride; };
    auto smod = [&](uint32_t x) { return MODE == 1 ? x & (uint32_t)(stride - 1) : x % (uint32_t)stride; };
    const int hd = (kTile + stride - 1) / stride + cby - 1;   /* rows per row phase */
    const int wd = LSTRIDE / stride;                          /* columns per column phase */
    const int max_rows = STRIDED ? hd * stride : kTile + (cby - 1);
    /* n_buf == 2: two (region + beam offset) buffers used alternately, one
     * barrier per tile; n_buf == 1: one buffer, two barriers per tile */
    const int buf_words = max_rows * LSTRIDE + kPbMax;
    uint32_t* sm_cells = reinterpret_cast<uint32_t*>(sm_tile);
    uint32_t* lpb = sm_cells + max_rows * LSTRIDE;
    /* lane base inside the region (cells) */
    const int tb = lane_on ? (g * R) * LSTRIDE + dxi : 0;
    const int lane = tid & 63;

    /* acc packs (known count << 23) + (sum of values) of at most 128 beams
     * (counted with multiplicity); S, K are the exact totals */
    uint32_t S[R], K[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        S[r] = 0;
        K[r] = 0;
        acc[r] = 0;
    }
    int pending = 0;
    auto flush = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S[r] += acc[r] & 0x7fffffu;
            K[r] += acc[r] >> 23;
            acc[r] = 0;
        }
        pending = 0;
    };

    const int grid_rows = job.rows, grid_pitch = job.pitch;
    const uint16_t* __restrict__ cells = job.cells;
    /* joint lists (k_binj): records and entries of the PAIR of slices t belongs to; an entry
     * carries the beam counts of both slices (sorted_rc: counts << 16 | row << 7 | col) */
    const bool joint = STRIDED && job.joint != 0;
    const int tp = joint ? t >> 1 : t;
    const int jsh = joint ? 16 + 8 * (t & 1) : 24;          /* where this slice's (even, odd) counts sit */
    const int ntiles = job.in_s ? 0 : job.n_tiles[tp];
    const TileRec* recs = job.tiles + (size_t)tp * job.max_tiles;
    const uint32_t* __restrict__ pbs = job.sorted_pb + (size_t)tp * (joint ? 2 : 1) * job.n_points;
    /* chunks (8 cells, 16 B of the uint16 grid) one lane may have to fetch per tile */
    constexpr int kMaxCh =
        ((STRIDED ? kMaxRegionRowsStrided : kMaxRegionRows) * (LSTRIDE / 8) + kBlock - 1) / kBlock;
    constexpr int kPbRegs = kPbMax / kBlock;

    /* Software pipeline: the global loads of tile i+1 (cells and beam offsets)
     * are in flight in registers while the lanes gather tile i out of LDS;
     * nothing in the gather loop waits on vector or scalar memory. Only the
     * bounding box of the tile's beams (+ the block's candidate span) is
     * staged. Cells are expanded on the way into LDS to v + (v != 0) << 24,
     * so that one 32-bit add per gather accumulates both the value sum and the
     * known count. */
    uint4 pre[kMaxCh];
    uint32_t pre_pb[kPbRegs];
    int nch = 1, total = 0;
    int lr0 = 0, ch0 = 0, dq = 0, dr = 0;   /* (row, chunk) of this lane's first chunk; step per k */
    TileRec rec;
    auto fetch = [&](const TileRec& tr) {
        const int cs = (tr.c0 + x0) & ~7;           /* 16-byte aligned first col */
        const int a = (tr.c0 + x0) - cs;            /* 0..7 */
        const int nrows = tr.h + (cby - 1) * stride;
        nch = (a + tr.w + (cbx - 1) * stride + 7) >> 3;
        total = nrows * nch;
        const int gr0 = tr.r0 + y0;
        /* chunk idx = tid + k*kBlock -> (row, chunk-in-row), stepped without
         * a division per chunk */
        lr0 = tid / nch;
        ch0 = tid - lr0 * nch;
        dq = kBlock / nch;
        dr = kBlock - dq * nch;
        int lr = lr0, ch = ch0;
#pragma unroll
        for (int k = 0; k < kMaxCh; ++k) {
            const int idx = tid + k * kBlock;
            const int gr = gr0 + lr, gc = cs + ch * 8;
            const bool ok = idx < total && gr >= 0 && gr < grid_rows && gc >= 0 && gc < grid_pitch;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok)
                v = *reinterpret_cast<const uint4*>(cells + (size_t)gr * grid_pitch + gc);
            pre[k] = v;
            lr += dq;
            ch += dr;
            if (ch >= nch) {
                ch -= nch;
                ++lr;
            }
        }
#pragma unroll
        for (int q = 0; q < kPbRegs; ++q) {
            const uint32_t bi = tid + q * kBlock;
            pre_pb[q] = bi < tr.count ? pbs[tr.start + bi] : 0u;
        }
    };
    /* 24 bits, so that v_mad_u32_u24 can weight a cell by its beam count */
    auto expand = [](uint32_t v) { return v + (min(v, 1u) << 23); };
    int ti = slice;
    if (ti < ntiles) {
        rec = recs[ti];
        fetch(rec);
    }
    for (int it = 0; ti < ntiles; ti += n_slices, ++it) {
        if (n_buf == 1) {
            __syncthreads();                         /* previous tile consumed */
        } else {
            sm_cells = reinterpret_cast<uint32_t*>(sm_tile) + (it & 1) * buf_words;
            lpb = sm_cells + max_rows * LSTRIDE;
        }
        {
            int lr = lr0, ch = ch0;
#pragma unroll
            for (int k = 0; k < kMaxCh; ++k) {
                if (tid + k * kBlock < total) {
                    const uint4 w = pre[k];
                    uint4 lo4, hi4;
                    lo4.x = expand(w.x & 0xffffu);
                    lo4.y = expand(w.x >> 16);
                    lo4.z = expand(w.y & 0xffffu);
                    lo4.w = expand(w.y >> 16);
                    hi4.x = expand(w.z & 0xffffu);
                    hi4.y = expand(w.z >> 16);
                    hi4.z = expand(w.w & 0xffffu);
                    hi4.w = expand(w.w >> 16);
                    if (!STRIDED) {
                        uint4* dst = reinterpret_cast<uint4*>(sm_cells + lr * LSTRIDE + ch * 8);
                        dst[0] = lo4;
                        dst[1] = hi4;
                    } else {
                        uint32_t* drow = sm_cells + (smod(lr) * hd + sdiv(lr)) * LSTRIDE;
                        if (MODE == 1 && sh == 1) {
                            /* stride 2: even cells -> phase 0, odd -> phase 1,
                             * four consecutive dwords each */
                            *reinterpret_cast<uint4*>(drow + ch * 4) =
                                make_uint4(lo4.x, lo4.z, hi4.x, hi4.z);
                            *reinterpret_cast<uint4*>(drow + wd + ch * 4) =
                                make_uint4(lo4.y, lo4.w, hi4.y, hi4.w);
                        } else if (MODE == 1 && sh == 2) {
                            /* stride 4: cells j and j + 4 share phase j */
                            *reinterpret_cast<uint2*>(drow + ch * 2) = make_uint2(lo4.x, hi4.x);
                            *reinterpret_cast<uint2*>(drow + wd + ch * 2) = make_uint2(lo4.y, hi4.y);
                            *reinterpret_cast<uint2*>(drow + 2 * wd + ch * 2) = make_uint2(lo4.z, hi4.z);
                            *reinterpret_cast<uint2*>(drow + 3 * wd + ch * 2) = make_uint2(lo4.w, hi4.w);
                        } else {
                            const uint32_t e[8] = { lo4.x, lo4.y, lo4.z, lo4.w,
                                                    hi4.x, hi4.y, hi4.z, hi4.w };
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const uint32_t gm = ch * 8 + j;
                                drow[smod(gm) * wd + sdiv(gm)] = e[j];
                            }
                        }
                    }
                }
                lr += dq;
                ch += dr;
                if (ch >= nch) {
                    ch -= nch;
                    ++lr;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kPbRegs; ++q)
            lpb[tid + q * kBlock] = pre_pb[q];
        __syncthreads();
        const TileRec cur = rec;
        if (ti + n_slices < ntiles) {
            rec = recs[ti + n_slices];
            fetch(rec);
        }
        const int a = (cur.c0 + x0) & 7;
        const int cnt = (int)cur.count;
        const uint32_t* base = sm_cells + tb + a;
        /* entry = cell offset + number of beams on that cell (<= kMaxMult);
         * strided jobs read k_bin's sorted_rc words: m_odd << 28 | m_even << 24 |
         * row << 16 | col inside the bounding box, m_odd beams on the row below */
        auto gather_at = [&](const uint32_t* p, uint32_t m) {
            if (WEIGHTED) {                        /* weighted by the beam count */
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[r] = mad_u24(p[r * LSTRIDE], m, acc[r]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[r] += p[r * LSTRIDE];
            }
        };
        auto gather = [&](uint32_t pbv) {
            if (STRIDED) {
                const uint32_t cbm = (joint ? pbv & 127u : pbv & 0xffffu) + (uint32_t)a;
                const uint32_t coff = smod(cbm) * wd + sdiv(cbm);
                const uint32_t me = (pbv >> jsh) & 15u, mo = (pbv >> (jsh + 4)) & 15u;
                uint32_t rb = joint ? (pbv >> 7) & 127u : (pbv >> 16) & 0xffu;
                if (me)
                    gather_at(sm_cells + tb + (smod(rb) * hd + sdiv(rb)) * LSTRIDE + coff, me);
                if (mo) {
                    ++rb;
                    gather_at(sm_cells + tb + (smod(rb) * hd + sdiv(rb)) * LSTRIDE + coff, mo);
                }
            } else {
                gather_at(base + (pbv & 0xffffu), pbv >> 16);
            }
        };
        auto mult_of = [&](uint32_t pbv) {
            return STRIDED ? ((pbv >> jsh) & 15u) + ((pbv >> (jsh + 4)) & 15u) : !WEIGHTED ? 1u : pbv >> 16;
        };
        /* entries: 64 per LDS read, broadcast with v_readlane */
        uint32_t pb_cur = lpb[lane];
        for (int b0 = 0; b0 < cnt; b0 += 64) {
            const uint32_t pb_nxt = lpb[(b0 + 64 + lane) & (kPbMax - 1)];
            const int m = min(64, cnt - b0);
            int j = 0;
            for (; j + 4 <= m; j += 4) {
                const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j);
                const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j + 1);
                const uint32_t o2 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j + 2);
                const uint32_t o3 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j + 3);
                const int mm = (int)(mult_of(o0) + mult_of(o1) + mult_of(o2) + mult_of(o3));
                if (pending + mm > 128)
                    flush();
                pending += mm;
                gather(o0);
                gather(o1);
                gather(o2);
                gather(o3);
            }
            for (; j < m; ++j) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j);
                const int mm = (int)mult_of(o);
                if (pending + mm > 128)
                    flush();
                pending += mm;
                gather(o);
            }
            pb_cur = pb_nxt;
        }
    }
    flush();

    score_epilogue<R>(job, S, K, t, bx, by, cbx, cby, g, dxi, lane_on, qflags, (int)blockIdx.x, (int)gridDim.x);
}

/* grid = (candidate blocks, theta slices, tile slices) */
template <int LSTRIDE, int R, int MODE, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void k_score(ScoreJob job, int cbx, int groups, int n_buf)
{
    score_body<LSTRIDE, R, MODE, WEIGHTED>(job, cbx, groups, blockIdx.z, gridDim.z, n_buf, blockIdx.y);
}

/* The arg-max pass after a tile-split launch (job.in_s set): same lane <->
 * candidate mapping and epilogue, no gathering. A kernel of its own so that
 * profiles keep it apart from the gather launches. */
template <int LSTRIDE, int R>
__global__ __launch_bounds__(kBlock) void k_argmax(ScoreJob job, int cbx, int groups)
{
    score_body<LSTRIDE, R, 0, false>(job, cbx, groups, 0, 1, 1, blockIdx.y);
}

/* grid = (candidate blocks, theta slices or fewer, jobs * n_slices). A workgroup
 * takes the slices blockIdx.y, blockIdx.y + gridDim.y, ...: the host folds the
 * theta axis for levels that normally exit at once (skip_unless_band) -- 4,160
 * workgroups that each allocate a region of LDS only to read one flag and leave
 * took 47 us per 64-window launch. */
template <int LSTRIDE, int R, int MODE, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void k_score_batch(const ScoreJob* jobs, int cbx, int groups,
                                                       int n_slices, int n_buf)
{
    const ScoreJob& job = jobs[blockIdx.z / n_slices];
    if (job.skip_unless_band && !(*job.flags & kFlagBandTouch))
        return;
    for (int t = blockIdx.y; t < job.n_theta; t += gridDim.y) {
        score_body<LSTRIDE, R, MODE, WEIGHTED>(job, cbx, groups, blockIdx.z % n_slices, n_slices, n_buf, t);
        __syncthreads();
    }
}

/* ------------------------------------------------------------------ K1, pair-row layout */

/* Block-sparse upload (csm_upload_grid_blocks): the reference grid is block_rows x block_cols
 * blocks of 2^k x 2^k uint16, row-major inside a block, unallocated blocks absent
 * (inc/grid_map_new/grid_map.hpp:255-263, grid_binary_bayes.hpp:197-202). `packed` holds the allocated
 * blocks back to back, slot[b] the position of block b in it or -1. Writes the dense pitched level
 * (what GridMap::CopyValues, src/grid_map_new/grid_map.cpp:289-350, 439-457, would have produced on the
 * host: unallocated blocks read 0), the per-block allocation bytes (GridMap::IsAllocated) and the first
 * row / column that hold a known cell. */
__global__ __launch_bounds__(256) void k_deblock(const uint16_t* __restrict__ packed, const int32_t* __restrict__ slot,
                                                int log2_block, int block_cols, int rows, int cols, int pitch,
                                                uint16_t* __restrict__ cells, uint8_t* __restrict__ alloc,
                                                int n_blocks, int32_t* __restrict__ known_first)
{
    const int mask = (1 << log2_block) - 1;
    const size_t total = (size_t)rows * pitch;
    int kr = 0x7fffffff, kc = 0x7fffffff;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / pitch), c = (int)(i % pitch);
        uint16_t v = 0;
        if (c < cols) {
            const int s = slot[(r >> log2_block) * block_cols + (c >> log2_block)];
            if (s >= 0)
                v = packed[((size_t)s << (2 * log2_block)) + ((size_t)(r & mask) << log2_block) + (c & mask)];
        }
        cells[i] = v;
        if (v) {
            kr = min(kr, r);
            kc = min(kc, c);
        }
    }
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n_blocks; b += gridDim.x * 256)
        alloc[b] = slot[b] >= 0 ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kr = min(kr, __shfl_xor(kr, d, 64));
        kc = min(kc, __shfl_xor(kc, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (kr != 0x7fffffff)
            atomicMin(known_first, kr);
        if (kc != 0x7fffffff)
            atomicMin(known_first + 1, kc);
    }
}

/* Expanded, zero-padded, pair-row copy of a grid level: slot (k, c) = 8 bytes =
 * cells (2k - pad, c - pad) and (2k + 1 - pad, c - pad) as v + (v != 0) << 23 (24
 * bits: one v_mad_u32_u24 per gather adds value sum and known count together).
 * The fine kernel copies windows of it into LDS with global_load_lds (no
 * unpacking, no bounds tests, no VGPRs on the way). */
__global__ __launch_bounds__(256) void k_expand_pairs(const uint16_t* __restrict__ cells, int rows, int cols,
                                                     int pitch, uint2* __restrict__ xg, int xg_prows,
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
        xg[i] = make_uint2(v0 + (min(v0, 1u) << 23), v1 + (min(v1, 1u) << 23));
    }
}

/* The entries of one slice's record from the staged window; the rules of the hand-issued
 * reads: a read must reach its lds_wait without crossing a loop edge (a register copy the
 * compiler places on an edge would copy the register before the data has landed), so the
 * pipeline drains at the end of every group of four entries. One entry: `pbv` =
 * m_odd << 28 | m_even << 24 | beams << 19 | byte offset of its first slot. CLS 0: both rows
 * hit, 1: even row only, 2: odd row only. The reads are hand-issued ds_read_b64 (the compiler
 * would fuse two of them into a ds_read2_b64, which runs at half the rate). */
template <int LS, int R, bool WEIGHTED>
__device__ __forceinline__ void pairs_gather(uint32_t lane_addr, const uint32_t* lpb, int lane, int cnt,
                                             int end_both, int end_even, uint32_t (&acc)[R],
                                             uint32_t (&S)[R], uint32_t (&K)[R], FlushState& fs)
{
    constexpr int kRowBytes = LS * 8;
    /* S collects the whole packed word (value sum + known count << 23, modulo 2^32), K the
     * count; pairs_finish() takes the count back out. One instruction per row less than
     * splitting the word here (the flushes are 6 % of the kernel). */
    auto flush = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S[r] += acc[r];
            K[r] += acc[r] >> 23;
            acc[r] = 0;
        }
    };
    auto issue = [&](uint32_t pbv, auto cls, unsigned long long (&q)[R / 2 + 1]) {
        constexpr int CLS = decltype(cls)::value;
        const uint32_t addr = lane_addr + (pbv & 0x7ffffu);
#ifdef CSM_ABL_NOREADS
        q[0] = q[1] = q[R / 2] = addr;
        if (R >= 6) q[2] = addr;
        if (R >= 8) q[3] = addr;
        return;
#endif
        lds_read_b64<0 * kRowBytes>(addr, q[0]);
        lds_read_b64<1 * kRowBytes>(addr, q[1]);
        if (R >= 6)
            lds_read_b64<2 * kRowBytes>(addr, q[2]);
        if (R >= 8)
            lds_read_b64<3 * kRowBytes>(addr, q[3]);
        if (CLS != 1)
            lds_read_b64<(R / 2) * kRowBytes>(addr, q[R / 2]);
    };
    auto mads = [&](uint32_t pbv, auto cls, const unsigned long long (&q)[R / 2 + 1]) {
        constexpr int CLS = decltype(cls)::value;
        uint32_t v[R + 2];
#pragma unroll
        for (int i = 0; i < R / 2 + 1; ++i) {
            v[2 * i] = (uint32_t)q[i];
            v[2 * i + 1] = (uint32_t)(q[i] >> 32);
        }
        const uint32_t me = (pbv >> 24) & 15u, mo = pbv >> 28;
#ifdef CSM_ABL_NOMADS
        acc[0] += v[0] + v[2] + v[4] + v[R] + me + mo;
        if (R >= 6) acc[1] += v[6];
        return;
#endif
        if (WEIGHTED) {
            /* the two multiply-adds of one accumulator are kept R instructions
             * apart: back to back the compiler pads them with s_nop */
            if (CLS != 2) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[r] = mad_u24(v[r], me, acc[r]);
            }
            if (CLS != 1) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[r] = mad_u24(v[r + 1], mo, acc[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (CLS == 0)
                    acc[r] += v[r] + v[r + 1];       /* v_add3_u32 */
                else
                    acc[r] += v[CLS == 1 ? r : r + 1];
            }
        }
    };
    /* 64 entries of the list, one per lane, and their flush flags */
    uint32_t pb_cur;
    unsigned long long flags;
    auto load_chunk = [&](int j0) {
        pb_cur = lpb[j0 + lane];
        const int b = j0 + lane < cnt ? (int)((pb_cur >> 19) & 31u) : 0;
        const int c1 = fs.cum + wave_prefix_sum(b), c0 = c1 - b;        /* < 96 + 64 * 30 */
        const bool cross = ((uint32_t)c1 * 683u) >> 16 != ((uint32_t)c0 * 683u) >> 16;   /* floor(c / 96), c < 2900 */
        const unsigned long long heavy = __builtin_amdgcn_ballot_w64(b > 8);
        flags = __builtin_amdgcn_ballot_w64(cross) | heavy | heavy << 1 | heavy << 2 | heavy << 3 | heavy << 4 |
                (fs.carry ? 0xfull : 0ull);
        /* a heavy entry among the last four REAL entries of this chunk flags the first group
         * of the next chunk -- also when that chunk is the next tile's list (acc[] and cum
         * live on across tiles, and a list normally ends in a partial chunk: looking at
         * lanes 60..63 only lost the flag there and let 152 beams pile up) */
        const int last = min(64, cnt - j0);
        fs.carry = (heavy >> max(0, last - 4)) != 0;
        fs.cum = __builtin_amdgcn_readlane(c1, 63) % 96;
    };
    int j = 0;
    load_chunk(0);
    auto run = [&](int end, auto cls) {
        constexpr int CLS = decltype(cls)::value;
        constexpr int NP = CLS == 1 ? R / 2 : R / 2 + 1;     /* reads per entry */
        constexpr int NQ = R / 2 + 1;
        while (j < end) {
            const int stop = min(end, (j | 63) + 1);
            for (; j + 4 <= stop; j += 4) {
                const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j & 63);
                const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + 1) & 63);
                const uint32_t o2 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + 2) & 63);
                const uint32_t o3 = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, (j + 3) & 63);
                if ((flags >> (j & 63)) & 0xfull)
                    flush();
                unsigned long long qa[NQ], qb[NQ], qc[NQ], qd[NQ];
                issue(o0, cls, qa);
                issue(o1, cls, qb);
                lds_wait<NP, NQ>(qa);                 /* all but the NP reads just issued */
                mads(o0, cls, qa);
                issue(o2, cls, qc);
                lds_wait<NP, NQ>(qb);
                mads(o1, cls, qb);
                issue(o3, cls, qd);
                lds_wait<NP, NQ>(qc);
                mads(o2, cls, qc);
                lds_wait<0, NQ>(qd);
                mads(o3, cls, qd);
            }
            for (; j < stop; ++j) {
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)pb_cur, j & 63);
                if ((flags >> (j & 63)) & 1ull)
                    flush();
                unsigned long long qa[NQ];
                issue(o, cls, qa);
                lds_wait<0, NQ>(qa);
                mads(o, cls, qa);
            }
            if ((j & 63) == 0 && j < cnt)
                load_chunk(j);
        }
    };
    run(end_both, std::integral_constant<int, 0>());
    run(end_even, std::integral_constant<int, 1>());
    run(cnt, std::integral_constant<int, 2>());
}

/* The fine level (candidates one cell apart) with the LDS region stored in
 * ROW PAIRS: the 8-byte slot (pair row k, column c) holds the expanded cells of
 * region rows 2k and 2k + 1 at column c. A lane owns one candidate column and R
 * (even) consecutive candidate rows and fetches its cells with ds_read_b64 --
 * twice the bytes per LDS cycle of ds_read_b32 -- and one entry of k_bin's pair
 * mode (an aligned row pair of one column with a beam count per row) needs R/2
 * reads (even row only) or R/2 + 1 (odd row / both) for R or 2R multiply-adds.
 *
 * Staging is LDS-DMA: the region is a window of the level's pair-row copy
 * (k_expand_pairs: expanded and zero-padded once per map), LS slots per pair
 * row, moved by global_load_lds_dwordx4 in 1-KiB pieces (wave-uniform LDS
 * destination, per-lane source = piece-relative offset computed once per
 * kernel + a scalar base per tile). No staging VALU work, no staging VGPRs.
 * Entries arrive sorted by class, TileRec.pad = class counts. */
template <int LS, int R, bool WEIGHTED>
__device__ __forceinline__ void score_body_pairs(const ScoreJob& job, int cbx, int groups, int slice,
                                                 int n_slices, int t, int cb, const uint16_t* lane_map,
                                                 BlockBase bb)
{
    static_assert(R % 2 == 0 && LS % 2 == 0, "pair rows, 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) uint16_t sm_tile[];
    if (t >= job.n_theta)
        return;
    const int tid = threadIdx.x;
    const int ncbx = (job.nx + cbx - 1) / cbx;
    const int bx = cb % ncbx, by = cb / ncbx;
    const int cby = groups * R;
    const int row0 = bb.row_base + by * cby;         /* first candidate row of this workgroup */
    if (row0 >= job.ny)
        return;
    const uint32_t qflags = job.elig_only_if_band ? *job.flags : 0u;

    /* which candidate column and lane group this thread owns: in thread order, or by the
     * host's table that keeps half-waves free of bank conflicts (lane_map_for, csm_api.hip) */
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
    constexpr int kRowBytes = LS * 8;                   /* one pair row of the region */
    const int prows_full = (kTile + cby) / 2 + 1;
    const int max_pieces = (prows_full * kRowBytes + 1023) >> 10;
    uint32_t* sm_cells = reinterpret_cast<uint32_t*>(sm_tile);
    uint32_t* lpb = sm_cells + max_pieces * 256;
    /* idle lanes of the table read the slot it gives them (a bank pair of their own) */
    const int tb = lane_on || (idle && g < groups && dxi < cbx) ? (g * (R / 2)) * kRowBytes + 8 * dxi : 0;      /* bytes */
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    /* a wave none of whose lanes owns a candidate inside the window (the last row block
     * of 84 rows in blocks of 48: rows 88..95) copies windows and keeps the barriers,
     * but gathers nothing */
    const bool wave_live = __builtin_amdgcn_ballot_w64(lane_on && bx * cbx + dxi < job.nx &&
                                                       row0 + g * R < job.ny) != 0;

    uint32_t S[R], K[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        S[r] = 0;
        K[r] = 0;
        acc[r] = 0;
    }
    FlushState fs = { 0, 0 };

    const int ntiles = job.in_s ? 0 : job.n_tiles[t];
    const TileRec* recs = job.tiles + (size_t)t * job.max_tiles;
    const uint32_t* __restrict__ pbs = job.sorted_pb + (size_t)t * job.n_points;
    /* every job field the tile loop needs, read once: `job` lives in global memory and
     * a read inside the loop is a load per tile (a stray one cost 7 % of the kernel) */
    const size_t xg_pitch = (size_t)job.xg_pitch;
    const size_t xg_row_bytes = xg_pitch * 8;
    const char* xg = reinterpret_cast<const char*>(job.xg);
    const int pad = job.xg_pad;

    /* piece pc = wave + 8 k of the region (LDS bytes [1024 pc, 1024 pc + 1024)): this
     * lane's 16 bytes sit in pair row prow at byte cb of it */
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
    int ti = slice;
    if (ti < ntiles)
        rec = recs[ti];
    for (; ti < ntiles; ti += n_slices) {
        /* wave-uniform values, told to the compiler as such: the entry loops then run
         * on scalar registers and scalar branches */
        const int c00 = __builtin_amdgcn_readfirstlane(rec.c0) + x0;
        const int gr0 = __builtin_amdgcn_readfirstlane(rec.r0) + y0;          /* even */
        const int a = c00 & 1;
        const int cnt = __builtin_amdgcn_readfirstlane((int)rec.count);
        const int start = __builtin_amdgcn_readfirstlane((int)rec.start);
        const int nprows = (__builtin_amdgcn_readfirstlane(rec.h) + cby) >> 1;
        const int classes = __builtin_amdgcn_readfirstlane(rec.pad[0]);
        const int end_both = classes & 0xffff;
        const int end_even = end_both + (classes >> 16);
        const int npieces = (nprows * kRowBytes + 1023) >> 10;
        const char* src = xg + ((size_t)((gr0 + pad) >> 1) * xg_pitch + (size_t)((c00 & ~1) + pad)) * 8;
        if (ti + n_slices < ntiles)
            rec = recs[ti + n_slices];
        __syncthreads();                                 /* previous tile consumed */
#pragma unroll
        for (int k = 0; k < kMaxP; ++k) {
            const int pc = wave + 8 * k;
            if (pc < npieces)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src + goff[k]), (lds_ptr)(sm_cells + pc * 256), 16, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < kPbMax / 64 / 8; ++e) {
            const int pe = wave + 8 * e;
            if (pe * 64 < cnt)
                __builtin_amdgcn_global_load_lds((glb_ptr)(pbs + start + pe * 64 + lane),
                                                 (lds_ptr)(lpb + pe * 64), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        const uint32_t lane_addr = lds_address(sm_cells) + (uint32_t)(tb + 8 * a);
        if (wave_live)
            pairs_gather<LS, R, WEIGHTED>(lane_addr, lpb, lane, cnt, end_both, end_even, acc, S, K, fs);
    }
    pairs_finish<R>(acc, S, K);
    /* the epilogue wants the row block as (by, cby) and uses only their product */
    score_epilogue<R>(job, S, K, t, bx, 1, cbx, row0, g, dxi, lane_on, qflags, cb + bb.cb_base, bb.ncb);
}

/* ---- two theta slices per workgroup ---------------------------------------
 * Neighbouring slices (0.25 - 0.5 degrees apart) put their beams on the same
 * endpoint tiles, one or two cells apart. A workgroup of the batch kernel takes
 * slices 2k and 2k + 1, walks the two record lists together (records ascend in
 * TileRec.pad[1] = tile, chunk), stages ONE window per tile -- the union of the
 * two bounding boxes -- and gathers the entries of both slices from it into two
 * sets of accumulators: the barriers, the DMA wait and the window copy of a tile
 * are paid once per two slices (they were 29 % + 6 % of the single-slice kernel). */

template <int LS, int R, bool WEIGHTED>
__device__ __forceinline__ void score_body_pairs2(const ScoreJob& job, int cbx, int groups, const uint16_t* lane_map,
                                                  int bid_x, int bid_y, BlockBase bb)
{   /* bid_x: candidate block, bid_y: pair of theta slices (the kernel's, after its XCD mapping) */
    static_assert(R % 2 == 0 && LS % 2 == 0, "pair rows, 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) uint16_t sm_tile[];
    const int t0 = 2 * bid_y, t1 = t0 + 1;
    /* values loaded from the job are uniform, but only readfirstlane tells the compiler:
     * everything derived from them then stays in scalar registers and scalar branches */
    const int n_theta = __builtin_amdgcn_readfirstlane(job.n_theta);
    if (t0 >= n_theta)
        return;
    const bool two = t1 < n_theta;
    const int tid = threadIdx.x;
    const int ncbx = (job.nx + cbx - 1) / cbx;
    const int bx = bid_x % ncbx, by = bid_x / ncbx;
    const int cby = groups * R;
    const int row0 = bb.row_base + by * cby;         /* first candidate row of this workgroup */
    if (row0 >= job.ny)
        return;
    const uint32_t qflags = job.elig_only_if_band ? *job.flags : 0u;

    /* which candidate column and lane group this thread owns: in thread order, or by the
     * host's table that keeps half-waves free of bank conflicts (lane_map_for, csm_api.hip) */
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
    uint32_t* lpb0 = sm_cells + max_pieces * 256;
    uint32_t* lpb1 = lpb0 + kPbMax;
    /* idle lanes of the table read the slot it gives them (a bank pair of their own) */
    const int tb = lane_on || (idle && g < groups && dxi < cbx) ? (g * (R / 2)) * kRowBytes + 8 * dxi : 0;      /* bytes */
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    /* see score_body_pairs: waves without a candidate inside the window do not gather */
    const bool wave_live = __builtin_amdgcn_ballot_w64(lane_on && bx * cbx + dxi < job.nx &&
                                                       row0 + g * R < job.ny) != 0;

    uint32_t S0[R], K0[R], acc0[R], S1[R], K1[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        S0[r] = K0[r] = acc0[r] = S1[r] = K1[r] = acc1[r] = 0;
    FlushState fs0 = { 0, 0 }, fs1 = { 0, 0 };

    const bool gather = !job.in_s;
    const int n0 = __builtin_amdgcn_readfirstlane(gather ? job.n_tiles[t0] : 0);
    const int n1 = __builtin_amdgcn_readfirstlane(gather && two ? job.n_tiles[t1] : 0);
    const TileRec* recs0 = job.tiles + (size_t)t0 * job.max_tiles;
    const TileRec* recs1 = job.tiles + (size_t)t1 * job.max_tiles;
    const uint32_t* __restrict__ pbs0 = job.sorted_pb + (size_t)t0 * job.n_points;
    const uint32_t* __restrict__ pbs1 = job.sorted_pb + (size_t)t1 * job.n_points;
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

    TileRec ra, rb;
    int i0 = 0, i1 = 0;
    if (n0 > 0)
        ra = recs0[0];
    if (n1 > 0)
        rb = recs1[0];
    const int kNone = 0x7fffffff;
    while (i0 < n0 || i1 < n1) {
        /* wave-uniform values of the records at the two cursors */
        const int key0 = i0 < n0 ? __builtin_amdgcn_readfirstlane(ra.pad[1]) : kNone;
        const int key1 = i1 < n1 ? __builtin_amdgcn_readfirstlane(rb.pad[1]) : kNone;
        const int key = min(key0, key1);
        const bool use0 = key0 == key, use1 = key1 == key;
        const int r00 = __builtin_amdgcn_readfirstlane(ra.r0), c00_ = __builtin_amdgcn_readfirstlane(ra.c0);
        const int h0 = __builtin_amdgcn_readfirstlane(ra.h);
        const int r01 = __builtin_amdgcn_readfirstlane(rb.r0), c01_ = __builtin_amdgcn_readfirstlane(rb.c0);
        const int h1 = __builtin_amdgcn_readfirstlane(rb.h);
        const int cnt0 = use0 ? __builtin_amdgcn_readfirstlane((int)ra.count) : 0;
        const int cnt1 = use1 ? __builtin_amdgcn_readfirstlane((int)rb.count) : 0;
        const int start0 = __builtin_amdgcn_readfirstlane((int)ra.start);
        const int start1 = __builtin_amdgcn_readfirstlane((int)rb.start);
        const int cls0 = __builtin_amdgcn_readfirstlane(ra.pad[0]);
        const int cls1 = __builtin_amdgcn_readfirstlane(rb.pad[0]);
        /* the union of the bounding boxes (r0 even in both) */
        const int ru = use0 && use1 ? min(r00, r01) : use0 ? r00 : r01;
        const int cu = use0 && use1 ? min(c00_, c01_) : use0 ? c00_ : c01_;
        const int re = use0 && use1 ? max(r00 + h0, r01 + h1) : use0 ? r00 + h0 : r01 + h1;
        const int c00 = cu + x0;
        const int gr0 = ru + y0;
        const int a = c00 & 1;
        const int nprows = (re - ru + cby) >> 1;
        const int npieces = (nprows * kRowBytes + 1023) >> 10;
        /* where each slice's box starts inside the staged window, bytes */
        const int shift0 = ((r00 - ru) >> 1) * kRowBytes + (c00_ - cu) * 8;
        const int shift1 = ((r01 - ru) >> 1) * kRowBytes + (c01_ - cu) * 8;
        const char* src = xg + ((size_t)((gr0 + pad) >> 1) * xg_pitch + (size_t)((c00 & ~1) + pad)) * 8;
        if (use0 && ++i0 < n0)
            ra = recs0[i0];
        if (use1 && ++i1 < n1)
            rb = recs1[i1];
        __syncthreads();                                 /* previous tile consumed */
        __builtin_amdgcn_s_setprio(3);                   /* the copy's requests go out ahead of the other
                                                            workgroup's gather instructions (-0.35 %) */
#pragma unroll
        for (int k = 0; k < kMaxP; ++k) {
            const int pc = wave + 8 * k;
            if (pc < npieces)
                __builtin_amdgcn_global_load_lds((glb_ptr)(src + goff[k]), (lds_ptr)(sm_cells + pc * 256), 16, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < kPbMax / 64 / 8; ++e) {
            const int pe = wave + 8 * e;
            if (pe * 64 < cnt0)
                __builtin_amdgcn_global_load_lds((glb_ptr)(pbs0 + start0 + pe * 64 + lane),
                                                 (lds_ptr)(lpb0 + pe * 64), 4, 0, 0);
            if (pe * 64 < cnt1)
                __builtin_amdgcn_global_load_lds((glb_ptr)(pbs1 + start1 + pe * 64 + lane),
                                                 (lds_ptr)(lpb1 + pe * 64), 4, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t lane_addr = lds_address(sm_cells) + (uint32_t)(tb + 8 * a);
        if (wave_live && cnt0 > 0)
            pairs_gather<LS, R, WEIGHTED>(lane_addr + (uint32_t)shift0, lpb0, lane, cnt0, cls0 & 0xffff,
                                          (cls0 & 0xffff) + (cls0 >> 16), acc0, S0, K0, fs0);
        if (wave_live && cnt1 > 0)
            pairs_gather<LS, R, WEIGHTED>(lane_addr + (uint32_t)shift1, lpb1, lane, cnt1, cls1 & 0xffff,
                                          (cls1 & 0xffff) + (cls1 >> 16), acc1, S1, K1, fs1);
    }
    pairs_finish<R>(acc0, S0, K0);
    pairs_finish<R>(acc1, S1, K1);
    score_epilogue<R>(job, S0, K0, t0, bx, 1, cbx, row0, g, dxi, lane_on, qflags, bid_x + bb.cb_base, bb.ncb);
    if (two) {
        __syncthreads();                                 /* the epilogue's reduction arrays */
        score_epilogue<R>(job, S1, K1, t1, bx, 1, cbx, row0, g, dxi, lane_on, qflags, bid_x + bb.cb_base, bb.ncb);
    }
}

/* grid = (candidate blocks, theta slices, tile slices), or with theta_major
 * (theta slices, candidate blocks, 1): the workgroups that run at the same time then
 * belong to ONE candidate block and neighbouring slices and copy (nearly) the same
 * windows of the map -- for maps far larger than an L2 (configs[4]: 56 MB) that turns
 * the window copies from fabric traffic into L2 hits. */
template <int LS, int R, bool WEIGHTED>
__global__ __launch_bounds__(kBlock, 4) void k_score_pairs(ScoreJob job, int cbx, int groups, int theta_major,
                                                           const uint16_t* lane_map)
{
    if (theta_major) {
        /* theta_major & 2: candidate blocks dealt to the XCDs (block b on XCD b mod 8), so that the
         * windows of a block -- nearly the same for neighbouring slices -- are fetched into one L2
         * instead of all eight: linear id L -> XCD L mod 8, its q-th workgroup (q = L / 8) ->
         * block 8 (q / T) + L mod 8, slice q mod T (T slices); blocks beyond the last multiple of
         * 8 keep the identity order */
        uint32_t t = blockIdx.x, cb = blockIdx.y;
        const uint32_t n_t = gridDim.x, lin = blockIdx.y * gridDim.x + blockIdx.x;
        if ((theta_major & 2) && lin < n_t * (gridDim.y & ~7u)) {
            const uint32_t q = lin >> 3, slot = q / n_t;
            t = q - slot * n_t;
            cb = 8u * slot + (lin & 7u);
        }
        score_body_pairs<LS, R, WEIGHTED>(job, cbx, groups, 0, 1, (int)t, (int)cb, lane_map,
                                          BlockBase{ 0, 0, (int)gridDim.y });
    } else
        score_body_pairs<LS, R, WEIGHTED>(job, cbx, groups, blockIdx.z, gridDim.z, (int)blockIdx.y,
                                          (int)blockIdx.x, lane_map, BlockBase{ 0, 0, (int)gridDim.x });
}

/* The single-window fine kernel over a work list (two-phase search: only the candidate blocks whose
 * coarse bound can still reach the best fine score; k_mark_blocks, csm_phase_kernels.hip): a fixed
 * grid of workgroups takes the items i = blockIdx.x, blockIdx.x + gridDim.x, ... below *count;
 * item = slice << 12 | candidate block. */
template <int LS, int R, bool WEIGHTED>
__global__ __launch_bounds__(kBlock, 4) void k_score_pairs_list(ScoreJob job, int cbx, int groups, int ncb,
                                                                const uint16_t* lane_map, const uint32_t* items,
                                                                const uint32_t* count)
{
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)*count);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)items[i]);
        score_body_pairs<LS, R, WEIGHTED>(job, cbx, groups, 0, 1, (int)(it >> 12), (int)(it & 4095u), lane_map,
                                          BlockBase{ 0, 0, ncb });
        __syncthreads();
    }
}

/* grid = (candidate blocks, theta slices, jobs) */
template <int LS, int R, bool WEIGHTED>
__global__ __launch_bounds__(kBlock, 4) void k_score_pairs_batch(const ScoreJob* jobs, int cbx, int groups,
                                                                 const uint16_t* lane_map, int xcd_map, BlockBase bb)
{
    int bx, by, bz;
    xcd_block(xcd_map, bx, by, bz);
    score_body_pairs<LS, R, WEIGHTED>(jobs[bz], cbx, groups, 0, 1, by, bx, lane_map, bb);
}

/* grid = (candidate blocks, ceil(theta slices / 2), jobs) */
template <int LS, int R, bool WEIGHTED>
__global__ __launch_bounds__(kBlock, 4) void k_score_pairs2_batch(const ScoreJob* jobs, int cbx, int groups,
                                                                  const uint16_t* lane_map, int xcd_map, BlockBase bb)
{
    int bx, by, bz;
    xcd_block(xcd_map, bx, by, bz);
    score_body_pairs2<LS, R, WEIGHTED>(jobs[bz], cbx, groups, lane_map, bx, by, bb);
}

/* ------------------------------------------------------------------ K2 */
/* Forward box maximum with the reference's tail rule, both passes in one
 * kernel, any number of (map, window) jobs in one launch:
 *   out[r][c] = max{ in[r'][c'] : s(r) <= r' < s(r) + W, s(c) <= c' < s(c) + W },
 *   s(i) = min(i, n - W)   ("repeat the last full window", inc/util.hpp:421-423)
 * which is SlidingWindowMaxRow then SlidingWindowMaxCol of
 * src/mapping/grid_map_builder.cpp:918-984 (a maximum of maxima; both orders
 * give the same bytes). A workgroup computes a 32 x 64 output tile: the input
 * rows / columns it needs form one contiguous range of at most 32 + W - 1 /
 * 64 + W - 1, staged in LDS; vertical maxima into a second LDS array, then
 * horizontal maxima to memory. Pad columns (cols..pitch) are written 0.
 * grid = (column tiles, row tiles, jobs). */

__global__ __launch_bounds__(256) void k_boxmax_batch(const BoxJob* jobs)
{
    __shared__ uint16_t tile[(kBoxTR + kBoxMaxWin - 1) * (kBoxTC + kBoxMaxWin)];
    __shared__ uint16_t mid[kBoxTR * (kBoxTC + kBoxMaxWin)];
    const BoxJob j = jobs[blockIdx.z];
    const int r0 = blockIdx.y * kBoxTR, c0 = blockIdx.x * kBoxTC;
    if (r0 >= j.rows || c0 >= j.pitch)
        return;
    const int w = j.win;
    const int tr = min(kBoxTR, j.rows - r0);              /* output rows of this tile */
    const int tc = min(kBoxTC, j.pitch - c0);             /* output columns incl. pad columns */
    const int tcv = max(0, min(kBoxTC, j.cols - c0));     /* ... that hold cells */
    const int in_r0 = min(r0, j.rows - w);
    const int nr = min(r0 + tr - 1, j.rows - w) + w - in_r0;
    const int in_c0 = min(c0, j.cols - w);
    const int nc = tcv > 0 ? min(c0 + tcv - 1, j.cols - w) + w - in_c0 : 0;
    const int ncp = kBoxTC + kBoxMaxWin;                  /* LDS row pitch */
    for (int i = threadIdx.x; i < nr * nc; i += 256) {
        const int r = i / nc, c = i - r * nc;
        tile[r * ncp + c] = j.src[(size_t)(in_r0 + r) * j.pitch + in_c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tr * nc; i += 256) {
        const int r = i / nc, c = i - r * nc;
        const int s = min(r0 + r, j.rows - w) - in_r0;
        uint16_t m = 0;
        for (int k = 0; k < w; ++k)
            m = max(m, tile[(s + k) * ncp + c]);
        mid[r * ncp + c] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tr * tc; i += 256) {
        const int r = i / tc, c = i - r * tc;
        uint16_t m = 0;
        if (c < tcv) {
            const int s = min(c0 + c, j.cols - w) - in_c0;
            for (int k = 0; k < w; ++k)
                m = max(m, mid[r * ncp + s + k]);
        }
        j.dst[(size_t)(r0 + r) * j.pitch + c0 + c] = m;
    }
}

/* ------------------------------------------------------------------ K4 */
__device__ __forceinline__ void k_finalize_body(const FinalJob& job)
{
    extern __shared__ double sm_p[];            /* [n_points] probabilities */
    __shared__ unsigned long long red_key[kBlock];
    __shared__ unsigned long long red_rank[kBlock];
    __shared__ uint32_t red_cnt[kBlock];
    __shared__ uint32_t red_s[kBlock], red_k[kBlock];
    const int tid = threadIdx.x;

    unsigned long long bkey = 0, brank = ~0ull;
    uint32_t bcnt = 0;
    for (int i = tid; i < job.n_entries; i += kBlock) {
        const BlockBest bb = job.block_best[i];
        best_combine(bkey, brank, bcnt, bb.key, bb.rank, bb.count);
    }
    red_key[tid] = bkey;
    red_rank[tid] = brank;
    red_cnt[tid] = bcnt;
    __syncthreads();
    for (int s = kBlock / 2; s >= 1; s >>= 1) {
        if (tid < s) {
            best_combine(bkey, brank, bcnt, red_key[tid + s], red_rank[tid + s], red_cnt[tid + s]);
            red_key[tid] = bkey;
            red_rank[tid] = brank;
            red_cnt[tid] = bcnt;
        }
        __syncthreads();
    }
    bkey = red_key[0];
    brank = red_rank[0];
    bcnt = red_cnt[0];

    csm_result* out = reinterpret_cast<csm_result*>(job.out);
    const uint32_t flags_in = job.flags_in ? (*job.flags_in & 0xffffu) : 0u;
    if (job.flags_clear && tid == 0)
        *job.flags_clear = 0u;
    if (bkey == 0) {
        if (tid == 0) {
            csm_result r;
            r.found = 0;
            r.best_x = job.init_x;
            r.best_y = job.init_y;
            r.best_theta = job.init_theta;
            r.key = 0;
            r.sum_values = 0;
            r.known = 0;
            r.tie_count = 0;
            r.flags = flags_in;
            r.score = job.score_thr;
            *out = r;
        }
        return;
    }
    /* decode the traversal rank */
    const int L = job.rank_l;
    const int nxc = job.nx / L, nyc = job.ny / L;
    unsigned long long q = brank;
    const int fy = (int)(q % L); q /= L;
    const int fx = (int)(q % L); q /= L;
    const int yc = (int)(q % nyc); q /= nyc;
    const int xc = (int)(q % nxc); q /= nxc;
    const int t = (int)q;
    const int x = job.x_lo + xc * L + fx, y = job.y_lo + yc * L + fy;

    /* f64 replay of the winner: gather in parallel, sum in beam order */
    const int32_t* col = job.hit_col + (size_t)t * job.n_points;
    const int32_t* row = job.hit_row + (size_t)t * job.n_points;
    uint32_t s = 0, k = 0;
    for (int i = tid; i < job.n_points; i += kBlock) {
        const int r = row[i] + y, c = col[i] + x;
        uint32_t v = 0;
        if (r >= 0 && r < job.rows && c >= 0 && c < job.cols)
            v = job.cells[(size_t)r * job.pitch + c];
        sm_p[i] = job.lut[v];
        s += v;
        k += v != 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m, 64);
        k += __shfl_xor(k, m, 64);
    }
    if ((tid & 63) == 0) {
        red_s[tid >> 6] = s;
        red_k[tid >> 6] = k;
    }
    __syncthreads();
    if (tid == 0) {
        /* beam order, one rounding per add; adding the 0.0 of an unknown
         * cell is exact, so no skip is needed */
        double sum = 0.0;
        int i = 0;
        for (; i + 8 <= job.n_points; i += 8) {
            double p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                p[j] = sm_p[i + j];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                sum += p[j];
        }
        for (; i < job.n_points; ++i)
            sum += sm_p[i];
        uint32_t st = 0, kt = 0;
        for (int i = 0; i < kBlock / 64; ++i) {
            st += red_s[i];
            kt += red_k[i];
        }
        const double score = sum / (double)job.n_points;
        csm_result r;
        r.found = score > job.score_thr ? 1 : 0;
        r.best_x = x;
        r.best_y = y;
        r.best_theta = t - job.win_theta;
        r.key = bkey;
        r.sum_values = st;
        r.known = kt;
        r.tie_count = bcnt;
        r.flags = flags_in | (bcnt > 1 ? CSM_FLAG_KEY_TIE : 0u);
        r.score = r.found ? score : job.score_thr;
        if (!r.found) {
            r.best_x = job.init_x;
            r.best_y = job.init_y;
            r.best_theta = job.init_theta;
        }
        *out = r;
    }
}

__global__ __launch_bounds__(kBlock) void k_finalize(FinalJob job)
{
    k_finalize_body(job);
}

/* ------------------------------------------------------------------ exact paths */

/* Every candidate of one level, f64 in beam order: what the reference's
 * ComputeScore / ScorePixelAccurate::Score returns. One lane per candidate. */
__global__ __launch_bounds__(kBlock) void k_exact_scores(ExactJob job)
{
    const long total = (long)job.n_theta * job.nx * job.ny;
    const long gid = (long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= total)
        return;
    const int yi = (int)(gid % job.ny);
    const int xi = (int)((gid / job.ny) % job.nx);
    const int t = (int)(gid / ((long)job.ny * job.nx));
    const int x = job.x_lo + xi * job.stride, y = job.y_lo + yi * job.stride;
    const size_t base = (size_t)t * job.n_points;
    double sum = 0.0;
    uint32_t known = 0;
    const bool per_node = job.r_cos != nullptr;
    const double px = job.sensor_x + x * job.step_x;
    const double py = job.sensor_y + y * job.step_y;
    for (int i = 0; i < job.n_points; ++i) {
        int r, c;
        if (per_node) {
            c = cell_index(px + job.r_cos[base + i], job.off_x, job.res);
            r = cell_index(py + job.r_sin[base + i], job.off_y, job.res);
        } else {
            c = job.hit_col[base + i] + x;
            r = job.hit_row[base + i] + y;
        }
        uint32_t v = 0;
        if (r >= 0 && r < job.rows && c >= 0 && c < job.cols)
            v = job.cells[(size_t)r * job.pitch + c];
        sum += job.lut[v];
        known += v != 0;
    }
    job.out_score[gid] = sum / (double)job.n_points;
    job.out_k[gid] = known;
}

/* One workgroup per tied candidate: exact score, as k_finalize does. */
__global__ __launch_bounds__(kBlock) void k_tie_replay(TieJob job)
{
    extern __shared__ double sm_tp[];
    const uint32_t n = min(*job.tie_count, job.tie_cap);
    if (blockIdx.x >= n)
        return;
    const int L = job.rank_l;
    const int nxc = job.nx / L, nyc = job.ny / L;
    unsigned long long q = job.tie_list[blockIdx.x];
    const int fy = (int)(q % L); q /= L;
    const int fx = (int)(q % L); q /= L;
    const int yc = (int)(q % nyc); q /= nyc;
    const int xc = (int)(q % nxc); q /= nxc;
    const int t = (int)q;
    const int x = job.x_lo + xc * L + fx, y = job.y_lo + yc * L + fy;
    const int32_t* col = job.hit_col + (size_t)t * job.n_points;
    const int32_t* row = job.hit_row + (size_t)t * job.n_points;
    for (int i = threadIdx.x; i < job.n_points; i += kBlock) {
        const int r = row[i] + y, c = col[i] + x;
        uint32_t v = 0;
        if (r >= 0 && r < job.rows && c >= 0 && c < job.cols)
            v = job.cells[(size_t)r * job.pitch + c];
        sm_tp[i] = job.lut[v];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int i = 0; i < job.n_points; ++i)
            sum += sm_tp[i];
        job.tie_score[blockIdx.x] = sum / (double)job.n_points;
    }
}

/* Pick among the tied candidates: highest f64 score, then first in traversal
 * order (the strict `<` update of scan_matcher_correlative.cpp:358). */
__global__ void k_tie_pick(TieJob job)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    csm_result* out = reinterpret_cast<csm_result*>(job.out);
    const uint32_t n = min(*job.tie_count, job.tie_cap);
    if (n == 0)
        return;
    double best = job.tie_score[0];
    unsigned long long brank = job.tie_list[0];
    uint32_t same = 1;
    for (uint32_t i = 1; i < n; ++i) {
        const double s = job.tie_score[i];
        const unsigned long long r = job.tie_list[i];
        if (s > best) {
            best = s;
            brank = r;
            same = 1;
        } else if (s == best) {
            ++same;
            if (r < brank)
                brank = r;
        }
    }
    const int L = job.rank_l;
    const int nxc = job.nx / L, nyc = job.ny / L;
    unsigned long long q = brank;
    const int fy = (int)(q % L); q /= L;
    const int fx = (int)(q % L); q /= L;
    const int yc = (int)(q % nyc); q /= nyc;
    const int xc = (int)(q % nxc); q /= nxc;
    const int t = (int)q;
    csm_result r = *out;
    r.flags |= CSM_FLAG_KEY_TIE | (same > 1 ? CSM_FLAG_F64_TIE : 0u);
    if (*job.tie_count > job.tie_cap)
        r.flags |= CSM_FLAG_EDGE_BAND;   /* overflow: let the literal path decide */
    r.tie_count = *job.tie_count;
    if (best > job.score_thr) {
        r.found = 1;
        r.best_x = job.x_lo + xc * L + fx;
        r.best_y = job.y_lo + yc * L + fy;
        r.best_theta = t - job.win_theta;
        r.score = best;
    }
    *out = r;
}

/* The reference's sweep, literally, over precomputed exact scores. One
 * wave64: lanes test 64 coarse nodes at a time against the running maximum;
 * a passing node is expanded (its L*L fine scores, first strict maximum) and
 * the remaining lanes are re-tested against the new maximum. */
__global__ __launch_bounds__(64) void k_csm_literal_scan(LiteralJob job)
{
    const int lane = threadIdx.x;
    const long per_t = (long)job.nxc * job.nyc;
    const long n_nodes = per_t * job.n_theta;
    const int L = job.L;
    const int ny = job.nyc * L, nx = job.nxc * L;
    double m = job.score_thr;
    long best_node = -1;
    int best_f = 0;
    for (long base = 0; base < n_nodes; base += 64) {
        const long i = base + lane;
        double c = 0.0;
        bool kok = false;
        if (i < n_nodes) {
            c = job.coarse_score[i];
            kok = (int)job.coarse_k[i] >= job.min_known;
        }
        int from = 0;
        while (true) {
            const bool pass = lane >= from && kok && c > m;
            const unsigned long long mask = __ballot(pass);
            if (mask == 0)
                break;
            const int first = __ffsll((long long)mask) - 1;
            const long node = base + first;
            const int t = (int)(node / per_t);
            const int xc = (int)((node % per_t) / job.nyc);
            const int yc = (int)(node % job.nyc);
            /* expand: lanes over the L*L fine candidates in (fx, fy) order */
            double fbest = m;
            int farg = -1;
            for (int f0 = 0; f0 < L * L; f0 += 64) {
                const int f = f0 + lane;
                double s = -1.0;
                if (f < L * L) {
                    const int fx = f / L, fy = f % L;
                    s = job.fine_score[((size_t)t * nx + xc * L + fx) * ny + yc * L + fy];
                }
                /* wave arg-max: greatest score, smallest index */
                double bs = s;
                int bi = f < L * L ? f : 0x7fffffff;
                for (int sh = 32; sh >= 1; sh >>= 1) {
                    const double os = __shfl_xor(bs, sh, 64);
                    const int oi = __shfl_xor(bi, sh, 64);
                    if (os > bs || (os == bs && oi < bi)) {
                        bs = os;
                        bi = oi;
                    }
                }
                if (bs > fbest) {
                    fbest = bs;
                    farg = bi;
                }
            }
            if (farg >= 0) {
                m = fbest;
                best_node = node;
                best_f = farg;
            }
            from = first + 1;
        }
    }
    if (lane == 0) {
        csm_result* out = reinterpret_cast<csm_result*>(job.out);
        csm_result r = *out;
        r.flags |= CSM_FLAG_LITERAL;
        if (best_node >= 0) {
            const int t = (int)(best_node / per_t);
            const int xc = (int)((best_node % per_t) / job.nyc);
            const int yc = (int)(best_node % job.nyc);
            r.found = 1;
            r.best_x = job.x_lo + xc * L + best_f / L;
            r.best_y = job.y_lo + yc * L + best_f % L;
            r.best_theta = t - job.win_theta;
            r.score = m;
        } else {
            r.found = 0;
            r.best_x = job.x_lo;
            r.best_y = job.y_lo;
            r.best_theta = -job.win_theta;
            r.score = job.score_thr;
        }
        *out = r;
    }
}

/* ------------------------------------------------------------------ grid search */
/* ScanMatcherGridSearch (src/mapping/scan_matcher_grid_search.cpp:84-142): one
 * lane per pose (dy, dx, dt); ScorePixelAccurate::Score for that pose in f64,
 * beam order. Lanes run over dx fastest so a wave shares one theta's products. */
__global__ __launch_bounds__(kBlock) void k_grid_scores(GridSearchJob job)
{
    const long total = (long)job.nx * job.ny * job.nt;
    const long gid = (long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= total)
        return;
    const int ix = (int)(gid % job.nx);
    const int iy = (int)((gid / job.nx) % job.ny);
    const int it = (int)(gid / ((long)job.nx * job.ny));
    const double px = job.px[ix], py = job.py[iy];
    const double* rc = job.r_cos + (size_t)it * job.n_points;
    const double* rs = job.r_sin + (size_t)it * job.n_points;
    double sum = 0.0;
    uint32_t known = 0;
    for (int i = 0; i < job.n_points; ++i) {
        const int c = cell_index(px + rc[i], job.off_x, job.res);
        const int r = cell_index(py + rs[i], job.off_y, job.res);
        uint32_t v = 0;
        if (r >= 0 && r < job.rows && c >= 0 && c < job.cols)
            v = job.cells[(size_t)r * job.pitch + c];
        sum += job.lut[v];
        known += v != 0;
    }
    const double score = sum / (double)job.n_points;
    const size_t p = ((size_t)iy * job.nx + ix) * job.nt + it;   /* dy, dx, dt loop order */
    job.out_score[p] = score;
    job.out_k[p] = known;
    if ((int)known >= job.min_known && score > job.score_thr)
        atomicMax(job.best_bits, (unsigned long long)__double_as_longlong(score) + 1ull);   /* 0 = none yet */
}

/* first pose, in loop order, that reaches the maximum (the strict `>` update) */
__global__ __launch_bounds__(kBlock) void k_grid_pick(GridSearchJob job)
{
    const long total = (long)job.nx * job.ny * job.nt;
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= total)
        return;
    const unsigned long long best = *job.best_bits;
    if (best == 0ull)
        return;
    if ((int)job.out_k[p] >= job.min_known &&
        (unsigned long long)__double_as_longlong(job.out_score[p]) + 1ull == best)
        atomicMin(job.best_index, (unsigned long long)p);
}

/* ------------------------------------------------------------------ batch */
__global__ __launch_bounds__(kBinBlock, 8) void k_bin_batch(const BinJob* jobs)
{
    /* k_bin reads blockIdx.x as the theta slice */
    k_bin_body(jobs[blockIdx.y]);
}

__global__ __launch_bounds__(kBlock) void k_finalize_batch(const FinalJob* jobs)
{
    k_finalize_body(jobs[blockIdx.x]);
}

/* dst[idx[i]] = src[i]: the records of one shape group into query order */
__global__ __launch_bounds__(256) void k_scatter_records(const csm_result* src, const int32_t* idx,
                                                        csm_result* dst, int n)
{
    static_assert(sizeof(csm_result) == 48, "record layout");
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const uint4* s = reinterpret_cast<const uint4*>(src + i);
    uint4* d = reinterpret_cast<uint4*>(dst + idx[i]);
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}


/* grid = (beam blocks, theta groups, jobs): a thread owns one beam and walks the
 * theta slices blockIdx.y, blockIdx.y + gridDim.y, ...
 *
 * The reference evaluates cos / sin of arg = (sensor.theta + stepTheta * t) + a_i
 * for every slice and beam (scan_matcher_correlative.cpp:163-166, 277-297). Here
 * a thread calls the library once, for B = sensor.theta + a_i, a workgroup once
 * per slice for D = stepTheta * t, and the slice's values come from the addition
 * theorems: 2 + 2/beams calls per beam and slice became 2/slices + 2/beams. The
 * result is NOT the host's value bit for bit and does not have to be: an entry
 * counts only if floor() cannot flip within the error bound (the certificate),
 * everything else is recomputed on the host. Error of the cosine (sine alike):
 *   |arg_host - (sensor.theta + D + a_i)| <= 1.5 ulp of the larger angle (the host
 *   rounds theta_t and arg, the device rounds B)      -> 4e-16 * (|theta| + |D| + |a|)
 *   device libm <= 3 ulp on each of cosB, sinB, cosD, sinD, two products, one sum
 *                                                     -> 2.4e-15 */

__device__ __forceinline__ void proj_body(const ProjJob& job)
{
    __shared__ double tab_c[kProjSlices], tab_s[kProjSlices];
    const int n_local = (job.n_theta - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;   /* slices of this group */
    if (n_local <= 0)
        return;
    for (int k = threadIdx.x; k < n_local; k += kBlock) {
        const int tt = ((int)blockIdx.y + k * (int)gridDim.y) - job.win_theta;
        const double d = job.step_theta * tt;
        tab_c[k] = cos(d);
        tab_s[k] = sin(d);
    }
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= job.n_points)
        return;
    /* every job field the slice loop needs, read once (the stores below may alias the
     * job as far as the compiler knows); 1 / res replaces the divisions: f64 division
     * was most of this kernel, and the quotient only has to be good to the bound */
    const double a = job.angles[i];
    const double r = job.ranges[i];
    const double sensor_x = job.sensor_x, sensor_y = job.sensor_y, sensor_theta = job.sensor_theta;
    const double off_x = job.off_x, off_y = job.off_y, res = job.res, inv_res = 1.0 / job.res;
    const double step_theta = job.step_theta, step_x = job.step_x, step_y = job.step_y;
    const int n_points = job.n_points, win_theta = job.win_theta;
    const int x_lo = job.x_lo, y_lo = job.y_lo, nx = job.nx, ny = job.ny;
    const bool check_nodes = job.check_nodes != 0, flag_uncertain = job.flag_uncertain != 0;
    int32_t* const hit_col = job.hit_col;
    int32_t* const hit_row = job.hit_row;
    const double bb = sensor_theta + a;
    const double cb = cos(bb), sb = sin(bb);
    const double ang = fabs(sensor_theta) + fabs(a);
    /* branch and bound: the widest node offsets, once per beam */
    const double xr = fmax(fabs((double)x_lo), fabs((double)(x_lo + nx - 1))) * step_x;
    const double yr = fmax(fabs((double)y_lo), fabs((double)(y_lo + ny - 1))) * step_y;
    const bool unit_step = step_x == res && step_y == res;
    /* a quotient formed as (h - off) * (1 / res): two roundings more than a division */
    auto err_bound = [&](double hit, double off, double q, double trig_err) {
        return (fabs(r) * trig_err + (fabs(hit) + fabs(off)) * 4e-16) * inv_res + fabs(q) * 8e-16;
    };
    for (int k = 0; k < n_local; ++k) {
        const int t = (int)blockIdx.y + k * (int)gridDim.y;
        const double cd = tab_c[k], sd = tab_s[k];
        const double trig_err = 2.4e-15 + 4e-16 * (ang + fabs(step_theta * (t - win_theta)));
        const double rc = r * (cb * cd - sb * sd);
        const double rs = r * (sb * cd + cb * sd);
        const double hx = sensor_x + rc, hy = sensor_y + rs;
        const double qx = (hx - off_x) * inv_res, qy = (hy - off_y) * inv_res;
        const double fx = floor(qx), fy = floor(qy);
        const double mx = 64.0 * err_bound(hx, off_x, qx, trig_err);
        const double my = 64.0 * err_bound(hy, off_y, qy, trig_err);
        const size_t idx = (size_t)t * n_points + i;
        const int col = (int)fx, row = (int)fy;
        hit_col[idx] = col;
        hit_row[idx] = row;
        bool uncertain = !(qx - fx > mx && qx - fx < 1.0 - mx && qy - fy > my && qy - fy < 1.0 - my);
        if (check_nodes && !uncertain) {
            /* appendNode's pose (scan_matcher_branch_bound.cpp:156-176):
             * floor(((sensor + x*step) + r*trig - off) / res) must equal base + x for
             * every node offset x. The search step IS the resolution
             * (scan_matcher_branch_bound.cpp:293-312), so in exact arithmetic the
             * node coordinate is q + x; every rounding on the way is bounded, hence
             * one test per beam and axis certifies all offsets at once. Only beams
             * within that (slightly wider) margin of a cell edge walk the offsets. */
            const double wx = 64.0 * ((fabs(r) * trig_err + (fabs(hx) + xr + fabs(off_x)) * 8e-16) * inv_res +
                                      (fabs(qx) + xr * inv_res) * 1.2e-15);
            const double wy = 64.0 * ((fabs(r) * trig_err + (fabs(hy) + yr + fabs(off_y)) * 8e-16) * inv_res +
                                      (fabs(qy) + yr * inv_res) * 1.2e-15);
            const bool sure_x = unit_step && qx - fx > wx && qx - fx < 1.0 - wx;
            const bool sure_y = unit_step && qy - fy > wy && qy - fy < 1.0 - wy;
            for (int xi = 0; xi < nx && !sure_x && !uncertain; ++xi) {
                const int x = x_lo + xi;
                const double h = (sensor_x + x * step_x) + rc;
                const double q = (h - off_x) * inv_res;
                const double f = floor(q);
                const double m = 64.0 * err_bound(h, off_x, q, trig_err);
                uncertain = !((int)f == col + x && q - f > m && q - f < 1.0 - m);
            }
            for (int yi = 0; yi < ny && !sure_y && !uncertain; ++yi) {
                const int y = y_lo + yi;
                const double h = (sensor_y + y * step_y) + rs;
                const double q = (h - off_y) * inv_res;
                const double f = floor(q);
                const double m = 64.0 * err_bound(h, off_y, q, trig_err);
                uncertain = !((int)f == row + y && q - f > m && q - f < 1.0 - m);
            }
        }
        if (uncertain) {
            if (check_nodes || flag_uncertain) {
                atomicOr(job.flags, CSM_FLAG_PROJ_DELTA);
            } else {
                const uint32_t pos = atomicAdd(job.unc_count, 1u);
                if (pos < job.unc_cap)
                    job.unc_list[pos] = (uint32_t)idx;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_project(ProjJob job)
{
    proj_body(job);
}

__global__ __launch_bounds__(kBlock) void k_project_batch(const ProjJob* jobs)
{
    proj_body(jobs[blockIdx.z]);
}

} /* namespace csm */
