/* csm_internal.hpp -- what the translation units of libcsm_hip.so share on the host side: the
 * context (device grids, workspaces, tuning switches, graphs), error / allocation / timing helpers.
 * csm_api.hip (matchers), csm_map_api.hip (map updates), csm_cost_api.hip (cost / covariance /
 * refinement) and csm_group.hip (several GPUs in one process) are compiled on their own. */
#ifndef CSM_INTERNAL_HPP
#define CSM_INTERNAL_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <limits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <tuple>
#include <atomic>
#include <vector>

#include "csm_device.hpp"
#include "../../include/csm_hip.h"

using namespace csm;

/* ------------------------------------------------------------------ ctx */

namespace csm_host {

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
};

struct Level {
    int       win = 1;
    uint16_t* cells = nullptr;   /* pitched rows*pitch */
    bool      owned = false;
    bool      stale = false;     /* derived from a base that was rebuilt since */
    size_t    cap = 0;           /* bytes allocated (owned levels) */
};

struct DeviceGrid;

/* Phase-major copy of a box-max(L) level of a map (k_phase_map): the grid the coarse pass of the
 * two-phase search scores on. */
struct PhaseMap {
    std::unique_ptr<DeviceGrid> grid;
    int hp = 0, wp = 0, pad = 0;
    const uint16_t* built_from = nullptr;   /* the level's buffer and the base epoch it was built at */
    uint64_t epoch = 0;
};

struct DeviceGrid {
    int rows = 0, cols = 0, pitch = 0;
    uint64_t base_epoch = 0;          /* bumped whenever level 0's cells change */
    std::map<int, PhaseMap> phase;    /* by box-max window L */
    int known_r0 = 0, known_c0 = 0;   /* first row / column holding a known cell */
    std::vector<Level> levels;   /* levels[0] is the uploaded grid */
    /* expanded, zero-padded pair-row copy of level 0 for the fine kernel's LDS-DMA
     * staging (k_expand_pairs); rebuilt when the base changes or a window needs more padding */
    uint32_t* xg = nullptr;
    size_t xg_cap = 0;
    int xg_pad = 0, xg_pitch = 0;
    bool xg_stale = true;
    /* the same layout holding float(499 v + 32268 (v != 0)) per cell: source of the fp32 bound
     * pass of the joint fine level (k_expand_pairs_f); follows xg */
    float* xgf = nullptr;
    size_t xgf_cap = 0;
    bool xgf_valid = false;
    /* block-allocation bitmap for the cost function's ProbabilityOr(.., 0.5): one byte per
     * block; the caller's (csm_set_block_allocation) or derived from the cells */
    uint8_t* alloc = nullptr;
    size_t alloc_cap = 0;
    int alloc_log2 = 0, alloc_bcols = 0;
    bool alloc_user = false, alloc_stale = true;
};

struct TimedSpan {
    hipEvent_t a, b;
};

struct KernelTimer {
    std::vector<TimedSpan> spans;
    double  total_ms = 0.0;
    int64_t launches = 0;
};

} /* namespace csm_host */
using namespace csm_host;

/* Launch-shape switches and forced shapes. Resolved ONCE, in csm_create: the switches from
 * csm_config.tuning_off (CSM_TUNE_NO_*, A/B measurements and tests); the forced shapes only
 * in tuning builds (-DCSM_TUNING, tools/build_variant.sh), from the environment. Nothing on a
 * launch path reads the environment. */
struct Tuning {
    bool lane_map = true;      /* conflict-free thread -> candidate table (lane_map_for) */
    bool xcd_map = true;       /* a job's workgroups on one XCD (xcd_block) */
    bool pair_tail = true;     /* a window's last row block as an R = 6 launch */
    bool two_slices = true;    /* batch fine kernel takes two theta slices per workgroup */
    bool joint = true;         /* ... on joint entry lists of the two slices (k_binj / k_score_joint_batch) */
    bool bound_pass = true;    /* ... preceded by the packed-fp32 bound pass; the exact kernel skips blocks that cannot win */
    int  two_phase = 0;        /* single large windows coarse-first: 0 by size, 1 always, -1 never */
    bool graphs = true;        /* repeated single-query launch chains replayed as HIP graphs */
    bool tile_split = true;    /* small single windows: tile list split over blockIdx.z */
    bool map_host_projection = false;   /* map building: hit points on the host */
    int  theta_major = -1;     /* -1: by launch size */
    int  fine_slices = 0, force_r = 0, pair_r = 0, pair_ncbx = 0, pair_groups = 0, pair_ls = 0,
         pair_tail_ls = 0, nbuf = 0, map_unc_cap = 0;
    bool plan_debug = false, host_timing = false;
};

struct csm_ctx {
    int device = 0;
    Tuning tune;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<uint64_t, DeviceGrid> grids;
    double* lut_dev = nullptr;
    /* workspaces */
    DevBuf hits, sorted, tiles, ntiles, misc, coarse_s, coarse_k, best, dump_s, dump_k, scratch;
    DevBuf b_prod, b_hits, b_sorted, b_tiles, b_ntiles, b_lvl, b_best, b_jobs, b_out, b_abest, bound_stats, b_items, tp_items, ph_hits;
    /* single-query launch chains as HIP graphs (csm_correlative_match): one per launch shape, keyed by
     * everything that is baked into the nodes; alloc_epoch changes whenever a device buffer the
     * nodes point at may have moved */
    uint64_t alloc_epoch = 0;
    bool capturing = false;
    std::map<std::vector<uint64_t>, hipGraphExec_t> graphs;
    std::map<std::vector<uint64_t>, int> graph_seen;
    void* q_pin = nullptr;           /* pinned: [ProjJob | angles | ranges] up, [record | uncertified count] back */
    size_t q_pin_cap = 0;
    DevBuf q_dev;
    const uint32_t* tp_count_dev = nullptr;      /* blocks kept (= items of the work list) by the last two-phase search */
    int64_t tp_blocks_total = 0;                 /* ... of this many */
    int64_t last_coarse_nodes = 0, last_fine_candidates = 0, last_nominal = 0, last_block_candidates = 0;   /* csm_last_search_info */
    DevBuf fine_s, fine_k, tie, ex_fine, ex_fine_k, ex_coarse, ex_coarse_k, scan_dev, unc, sorted_rc, b_sorted_rc;
    std::map<std::array<int, 4>, uint16_t*> lane_maps;   /* lane_map_for(): (cbx, groups, R, LS) -> device table */
    void* pin = nullptr;          /* pinned staging of csm_upload_grid */
    size_t pin_cap = 0;
    void* pin_scans = nullptr;    /* pinned staging of a batch's scans */
    size_t pin_scans_cap = 0;
    /* cost / refinement batches: device scans + job table, host staging */
    DevBuf c_scans, c_jobs, box_jobs;
    std::vector<csm::BoxJob> box_stage;
    std::vector<double> c_stage;
    std::vector<csm::CostJob> c_job_stage;
    /* the final records of the last batch call in query order (csm_copy_last_batch_records) */
    DevBuf rec_dev;
    int rec_n = 0;
    std::vector<csm_result> rec_patch;            /* host copies of records fixed up after the device pass */
    /* map building */
    DevBuf m_rays, m_recs, m_cell, m_lists, m_cnt, m_lut;
    double m_lut_hit = -1.0, m_lut_miss = -1.0;   /* probabilities the update tables were built for */
    bool m_apply_attr = false;
    hipEvent_t m_ev[2] = { nullptr, nullptr };    /* device_us of csm_map_build_info */
    std::vector<double> stage;                    /* host staging of one scan (angles, ranges) */
    /* job tables of csm_score_windows_dev calls (pageable sources of asynchronous
     * uploads), each kept until the event recorded behind its launch chain has fired */
    std::vector<std::pair<hipEvent_t, std::shared_ptr<void>>> resident_hold;
    /* the fine-level job of the last csm window, for the tie collection pass */
    csm::ScoreJob last_fine;
    unsigned flag_toggle = 0;     /* two flag words, used alternately: k_finalize of query i
                                     clears the word of query i + 1 */
    bool flags_ready = false;
    bool fine_acc_dirty = false;  /* a tile-split launch was issued but its arg-max pass (which
                                     clears the accumulators) was not: clear before reuse */
    int timing = 0;               /* 0 off, 1 every kernel, 2 the fine scoring kernel only */
    std::map<std::string, KernelTimer> timers;
    std::vector<hipEvent_t> event_pool;
    /* pinned staging blocks of the batch entries' job tables, reused once the copy
     * that reads them has run */
    std::vector<std::pair<void*, size_t>> pin_free;
};

namespace csm_host {


inline int fail(csm_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                        \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail(ctx, CSM_EIO, "%s failed: %s (%s:%d)", #expr,             \
                        hipGetErrorString(e_), __FILE__, __LINE__);               \
    } while (0)

inline int ensure(csm_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap)
        return CSM_OK;
    if (ctx->capturing)
        return fail(ctx, CSM_EIO, "internal: a workspace would grow during graph capture");
    ++ctx->alloc_epoch;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (b.p)
        HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess)
        return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", want);
    b.cap = want;
    return CSM_OK;
}

struct ScopedTimer {
    csm_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    ScopedTimer(csm_ctx* c, const char* n) : ctx(c), name(n)
    {
        if (!ctx->timing || ctx->capturing ||
            (ctx->timing == 2 && std::strcmp(n, "score_fine") != 0 && std::strcmp(n, "score_bound") != 0))
            return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!ctx->event_pool.empty()) {
                e = ctx->event_pool.back();
                ctx->event_pool.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, ctx->stream);
    }
    ~ScopedTimer()
    {
        if (!a)
            return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->timers[name].spans.push_back({ a, b });
    }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

/* A "no return" beam (inf / NaN range) has no hit point; the reference's scan
 * filters drop such beams before a matcher sees the scan. The library refuses
 * them instead of converting a non-finite coordinate to an int. */
inline bool scan_is_finite(const csm_scan* scan)
{
    for (int i = 0; i < scan->n_points; ++i)
        if (!std::isfinite(scan->ranges[i]) || !std::isfinite(scan->angles[i]))
            return false;
    return std::isfinite(scan->relative_sensor_pose[0]) && std::isfinite(scan->relative_sensor_pose[1]) &&
           std::isfinite(scan->relative_sensor_pose[2]);
}

inline DeviceGrid* find_grid(csm_ctx* ctx, uint64_t id)
{
    auto it = ctx->grids.find(id);
    return it == ctx->grids.end() ? nullptr : &it->second;
}

/* defined in csm_api.hip */
void free_levels(DeviceGrid& g, bool keep_base);

} /* namespace csm_host */
using namespace csm_host;

#endif
