/* csm_window.hip -- one search window at a time (host code only): the CSM launch chain (run_window), the
 * coarse-first search of large windows (search_window, csm_phase_kernels.hip), tie / literal resolution,
 * and the single-query entry points of the C ABI (csm_score_window*, csm_correlative_match with its graph
 * replay, csm_grid_search_match, csm_project_scan). */
#include "csm_matchers.hpp"

namespace csm_host {


/* Mode 1 on JOINT entries of slice pairs (csm_joint_kernels.hip: k_binj_one + k_score_joint_one), jobs
 * by value. On the phase-major copy a tile holds the beams of one phase only (configs[4]: ~6 entries per
 * staged window against ~53 at the fine level), so the pass is bound by staging; a pair of neighbouring
 * slices shares every staged window. Returns kNotJoint where the joint tables do not fit (the caller
 * then takes the per-slice pair kernel). */
const int kNotJoint = -1000;

int run_level_pass_joint(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                         const int32_t* hit_col_dev, const int32_t* hit_row_dev, uint32_t* flags, TwoPhaseCtl* tp)
{
    if (!ctx->tune.joint || !ctx->tune.two_slices || !p.fine.pairs || p.L != 1)
        return kNotJoint;
    PassPlan jp;
    const int hash_size = csm::binj_hash_size(p.n);
    const size_t binj_lds = csm::binj_lds_bytes(p.tiles_x * p.tiles_y, p.n, hash_size);
    /* the exact joint kernel keeps kJRec entry words next to the window copy (not two kPbMax lists) */
    if (binj_lds > 150 * 1024 || !plan_pass_pairs(ctx->tune, p.nx, p.ny, &jp, true, kJRec * 4) || jp.lists != 2)
        return kNotJoint;
    jp.joint = true;
    jp.weighted = true;
    int rc;
    const int n_pairs = (p.n_theta + 1) / 2;
    const int max_tiles = std::min(2 * p.n, p.tiles_x * p.tiles_y) + 2 * p.n / kJRec + 1;
    if ((rc = ensure(ctx, ctx->sorted, (size_t)n_pairs * 2 * p.n * 4 + 256))) return rc;
    if ((rc = ensure(ctx, ctx->tiles, (size_t)n_pairs * max_tiles * sizeof(TileRec)))) return rc;
    if ((rc = ensure(ctx, ctx->ntiles, (size_t)n_pairs * 8))) return rc;
    BinJob bj;
    std::memset(&bj, 0, sizeof(bj));
    bj.hit_col = hit_col_dev;
    bj.hit_row = hit_row_dev;
    bj.sorted_pb = reinterpret_cast<uint32_t*>(ctx->sorted.p);
    bj.tiles = reinterpret_cast<TileRec*>(ctx->tiles.p);
    bj.n_tiles = reinterpret_cast<int32_t*>(ctx->ntiles.p);
    bj.flags = flags;
    bj.n_theta = p.n_theta;
    bj.n_points = p.n;
    bj.max_tiles = max_tiles;
    bj.rows = g.rows;
    bj.cols = g.cols;
    bj.x_lo = p.x_lo;
    bj.y_lo = p.y_lo;
    bj.x_hi = p.x_hi;
    bj.y_hi = p.y_hi;
    bj.tiles_x = p.tiles_x;
    bj.tiles_y = p.tiles_y;
    bj.known_r0 = g.known_r0;
    bj.known_c0 = g.known_c0;
    bj.hash_size = hash_size;
    bj.max_mult = kMaxMult;
    bj.lstride = jp.lstride;
    bj.pair_mode = 2;
    bj.frame_shift = (p.ny - 1) & 1;
    {
        ScopedTimer tm(ctx, "bin");
        if ((rc = launched_ok(ctx, csm::launch_binj_one(ctx->stream, ctx->device, bj, n_pairs, binj_lds), "joint binning")))
            return rc;
    }
    ScoreJob fj;
    std::memset(&fj, 0, sizeof(fj));
    fj.rows = g.rows;
    fj.cols = g.cols;
    fj.pitch = g.pitch;
    fj.sorted_pb = bj.sorted_pb;
    fj.tiles = bj.tiles;
    fj.n_tiles = bj.n_tiles;
    fj.n_theta = p.n_theta;
    fj.n_points = p.n;
    fj.max_tiles = max_tiles;
    fj.x_lo = p.x_lo;
    fj.y_lo = p.y_lo;
    fj.flags = flags;
    fj.min_known = w->min_known;
    fj.cells = g.levels[0].cells;
    fj.xg = g.xg;
    fj.xg_pitch = g.xg_pitch;
    fj.xg_pad = g.xg_pad;
    fj.nx = p.nx;
    fj.ny = p.ny;
    fj.stride = 1;
    fj.rank_l = p.L;
    fj.joint = 1;
    /* every candidate's sums, stored; no arg-max, no record */
    fj.acc_s = reinterpret_cast<uint32_t*>(ctx->coarse_s.p);
    fj.acc_k = reinterpret_cast<uint32_t*>(ctx->coarse_k.p);
    fj.acc_x_major = 2;
    tp->level_s = fj.acc_s;
    tp->level_k = fj.acc_k;
    tp->nxs = p.nx;
    tp->nys = p.ny;
    const uint16_t* lane_map = nullptr;
    if ((rc = lane_map_for(ctx, jp, &lane_map)))
        return rc;
    csm::JointLaunch L{};
    L.stream = ctx->stream;
    L.device = ctx->device;
    L.grid = dim3(jp.ncb(), 1, 1);
    L.lds_bytes = pass_lds_bytes(jp);
    L.ls = jp.lstride;
    L.R = jp.R;
    L.cbx = jp.cbx;
    L.groups = jp.groups;
    L.lane_map = lane_map;
    L.ncb = jp.ncb();
    ScopedTimer tm(ctx, "score_coarse");
    return launched_ok(ctx, csm::launch_joint_one(L, fj, n_pairs), "joint level pass");
}

int run_window(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
               const int32_t* hit_col_dev, const int32_t* hit_row_dev,
               csm_result* out_dev, const WindowOutputs* dumps, bool force_coarse,
               TwoPhaseCtl* tp)
{
    const int tp_mode = tp ? tp->mode : 0;
    if (w->coarse_level < 0 || w->coarse_level >= (int)g.levels.size())
        return fail(ctx, CSM_ENOENT, "coarse level %d not built", w->coarse_level);
    if (g.levels[w->coarse_level].stale)
        return fail(ctx, CSM_ENOENT, "coarse level %d is stale: the map was rebuilt", w->coarse_level);
    if (g.levels[w->coarse_level].win != p.L)
        return fail(ctx, CSM_EINVAL, "level %d holds box-max(%d), window asks L=%d",
                    w->coarse_level, g.levels[w->coarse_level].win, p.L);
    int rc;
    const size_t nt = p.n_theta;
    if ((rc = ensure(ctx, ctx->sorted, nt * p.n * 4 + 256))) return rc;   /* + 64 entries: the LDS-DMA of a
                                                                             tile's list reads whole 64-entry pieces */
    if (p.fine.pairs && (rc = ensure_xgrid(ctx, g, xgrid_pad_for(p.nx, p.ny)))) return rc;
    if ((rc = ensure(ctx, ctx->tiles, nt * p.max_tiles * sizeof(TileRec)))) return rc;
    if ((rc = ensure(ctx, ctx->ntiles, nt * 8))) return rc;
    if ((rc = ensure(ctx, ctx->misc, 256))) return rc;
    if ((rc = ensure(ctx, ctx->coarse_s, nt * p.nxc * p.nyc * 4))) return rc;
    if ((rc = ensure(ctx, ctx->coarse_k, nt * p.nxc * p.nyc * 4))) return rc;
    const int ncb = p.fine.ncb();
    if ((rc = ensure(ctx, ctx->best, nt * ncb * sizeof(BlockBest)))) return rc;
    if ((rc = ensure(ctx, ctx->sorted_rc, nt * p.n * 4))) return rc;

    /* tile-split fine launch when the window gives fewer than ~1.5 workgroups
     * per CU (config 2: 246); CSM_TUNE_NO_TILE_SPLIT: never */
    int fine_slices = 1;
    {
        const long blocks = (long)ncb * p.n_theta;
        if (blocks < 384)
            fine_slices = (int)std::min<long>(4, std::max<long>(1, 492 / std::max<long>(1, blocks)));
        if (!ctx->tune.tile_split || tp_mode)
            fine_slices = 1;
        else if (ctx->tune.fine_slices)
            fine_slices = std::max(1, std::min(8, ctx->tune.fine_slices));
        if (fine_slices > 1) {
            /* the accumulators are zero between queries: cleared once when
             * (re)allocated, then by the arg-max pass as it reads them */
            const size_t words = nt * (size_t)p.nx * p.ny;
            const void* old_s = ctx->fine_s.p;
            const void* old_k = ctx->fine_k.p;
            if ((rc = ensure(ctx, ctx->fine_s, words * 4))) return rc;
            if ((rc = ensure(ctx, ctx->fine_k, words * 4))) return rc;
            if (ctx->fine_s.p != old_s || ctx->fine_acc_dirty)
                HIP_TRY(ctx, hipMemsetAsync(ctx->fine_s.p, 0, ctx->fine_s.cap, ctx->stream));
            if (ctx->fine_k.p != old_k || ctx->fine_acc_dirty)
                HIP_TRY(ctx, hipMemsetAsync(ctx->fine_k.p, 0, ctx->fine_k.cap, ctx->stream));
            ctx->fine_acc_dirty = false;
        }
    }
    uint32_t* flag_words = reinterpret_cast<uint32_t*>(ctx->misc.p);
    if (!ctx->flags_ready && !ctx->capturing) {
        HIP_TRY(ctx, hipMemsetAsync(flag_words, 0, 16, ctx->stream));
        ctx->flags_ready = true;
    }
    uint32_t* flags = flag_words + (ctx->flag_toggle & 1u);
    uint32_t* flags_next = flag_words + ((ctx->flag_toggle + 1u) & 1u);
    if (ctx->capturing) {
        /* a graph bakes its pointers: a flag word of its own, cleared by a node of the graph */
        flags = flag_words + 2;
        flags_next = nullptr;
        HIP_TRY(ctx, hipMemsetAsync(flags, 0, 4, ctx->stream));
    } else if (tp_mode != 1) {  /* the level pass sets no flag and has no finalize to clear one */
        ctx->flag_toggle++;
    }

    if (tp_mode == 1) {
        const int rcj = run_level_pass_joint(ctx, g, w, p, hit_col_dev, hit_row_dev, flags, tp);
        if (rcj != kNotJoint)
            return rcj;
    }

    BinJob bj;
    std::memset(&bj, 0, sizeof(bj));
    bj.hit_col = hit_col_dev;
    bj.hit_row = hit_row_dev;
    bj.sorted_pb = reinterpret_cast<uint32_t*>(ctx->sorted.p);
    bj.tiles = reinterpret_cast<TileRec*>(ctx->tiles.p);
    bj.n_tiles = reinterpret_cast<int32_t*>(ctx->ntiles.p);
    bj.flags = flags;
    bj.n_theta = p.n_theta;
    bj.n_points = p.n;
    bj.max_tiles = p.max_tiles;
    bj.rows = g.rows;
    bj.cols = g.cols;
    bj.x_lo = p.x_lo;
    bj.y_lo = p.y_lo;
    bj.x_hi = p.x_hi;
    bj.y_hi = p.y_hi;
    bj.tiles_x = p.tiles_x;
    bj.tiles_y = p.tiles_y;
    bj.known_r0 = g.known_r0;
    bj.known_c0 = g.known_c0;
    bj.hash_size = bin_hash_size(p.n);
    bj.max_mult = p.fine.weighted ? kMaxMult : 1;
    bj.lstride = p.fine.lstride;
    bj.pair_mode = p.fine.pairs ? 1 : 0;
    bj.frame_shift = p.fine.pairs ? ((p.ny - 1) & 1) : 0;
    bj.sorted_rc = p.L > 1 ? reinterpret_cast<uint32_t*>(ctx->sorted_rc.p) : nullptr;
    const bool coarse_exits = w->min_known <= 1 && !force_coarse && tp_mode != 2;   /* unless a beam reaches the band */
    if (p.L > 1) {
        bj.n_band = 1;
        bj.band_win[0] = p.L;
        bj.band_nx[0] = p.nxc;
        bj.band_ny[0] = p.nyc;
    }
    {
        const size_t lds = bin_lds_bytes(p.tiles_x * p.tiles_y, p.n);
        ScopedTimer tm(ctx, "bin");
        if ((rc = launched_ok(ctx, csm_launch::bin(ctx->stream, ctx->device, p.n_theta, lds, bj), "binning"))) return rc;
    }
    if (p.L > 1 && tp_mode != 2) {
        /* the coarse pass accumulates with atomics: its sums are cleared first, but
         * only when it is going to run (k_zero_if_band reads the band flag k_bin set) */
        ZeroJob zj;
        zj.a = reinterpret_cast<uint32_t*>(ctx->coarse_s.p);
        zj.b = reinterpret_cast<uint32_t*>(ctx->coarse_k.p);
        zj.words = nt * p.nxc * p.nyc;
        zj.flags = flags;
        zj.always = coarse_exits ? 0 : 1;
        zj.pad = 0;
        const int zb = (int)std::min<size_t>(256, (zj.words + 255) / 256);
        if ((rc = launched_ok(ctx, csm_launch::zero_if_band(ctx->stream, std::max(1, zb), zj), "edge-band clear"))) return rc;
    }

    ScoreJob base;
    std::memset(&base, 0, sizeof(base));
    base.rows = g.rows;
    base.cols = g.cols;
    base.pitch = g.pitch;
    base.sorted_pb = bj.sorted_pb;
    base.tiles = bj.tiles;
    base.n_tiles = bj.n_tiles;
    base.n_theta = p.n_theta;
    base.n_points = p.n;
    base.max_tiles = p.max_tiles;
    base.x_lo = p.x_lo;
    base.y_lo = p.y_lo;
    base.flags = flags;
    base.min_known = w->min_known;

    if (p.L > 1 && tp_mode != 2) {
        ScoreJob cj = base;
        cj.cells = g.levels[w->coarse_level].cells;
        cj.nx = p.nxc;
        cj.ny = p.nyc;
        cj.stride = p.L;
        cj.log2_stride = p.coarse.log2s;
        cj.sorted_pb = bj.sorted_rc;
        cj.acc_s = reinterpret_cast<uint32_t*>(ctx->coarse_s.p);
        cj.acc_k = reinterpret_cast<uint32_t*>(ctx->coarse_k.p);
        cj.rank_l = 1;
        cj.skip_unless_band = coarse_exits;
        const size_t nodes = nt * p.nxc * p.nyc;
        (void)nodes;
        ScopedTimer tm(ctx, "score_coarse");
        /* few candidates per slice: split the tile list over blockIdx.z so
         * enough workgroups are in flight to hide the staging latency -- unless
         * the pass only runs when a beam reaches the edge band (rare): then one
         * slice, so that the launch that normally exits at once stays small */
        if ((rc = launch_score(ctx, cj, p.coarse, p.n_theta, coarse_exits ? 1 : kCoarseSlices)))
            return rc;
    }

    const BlockBest* tp_reduced = nullptr;
    ScoreJob fj = base;
    fj.cells = g.levels[0].cells;
    fj.xg = g.xg;
    fj.xg_pitch = g.xg_pitch;
    fj.xg_pad = g.xg_pad;
    fj.nx = p.nx;
    fj.ny = p.ny;
    fj.stride = 1;
    fj.block_best = reinterpret_cast<BlockBest*>(ctx->best.p);
    (void)fine_slices;
    fj.rank_l = p.L;
    if (dumps) {
        fj.dump_s = dumps->dump_s;
        fj.dump_k = dumps->dump_k;
    }
    if (p.L > 1) {
        fj.n_elig = 1;
        fj.elig[0].k = reinterpret_cast<const uint32_t*>(ctx->coarse_k.p);
        fj.elig[0].s = reinterpret_cast<const uint32_t*>(ctx->coarse_s.p);
        fj.elig[0].div = p.L;
        fj.elig[0].nxc = p.nxc;
        fj.elig[0].nyc = p.nyc;
        fj.elig_only_if_band = coarse_exits;
        if (tp_mode == 2) {
            fj.elig[0].k = tp->level_k;
            fj.elig[0].s = tp->level_s;
            fj.elig[0].nxc = tp->nxs;
            fj.elig[0].nyc = tp->nys;
        }
    } else {
        fj.check_own_known = 1;
    }
    if (tp_mode == 1) {
        /* every candidate's sums, stored; no arg-max, no record */
        fj.block_best = nullptr;
        fj.check_own_known = 0;
        fj.acc_s = reinterpret_cast<uint32_t*>(ctx->coarse_s.p);
        fj.acc_k = reinterpret_cast<uint32_t*>(ctx->coarse_k.p);
        fj.acc_x_major = 2;
        tp->level_s = fj.acc_s;
        tp->level_k = fj.acc_k;
        tp->nxs = p.nx;
        tp->nys = p.ny;
        ScopedTimer tm(ctx, "score_coarse");
        return launch_score(ctx, fj, p.fine, p.n_theta, 1);
    }
    if (tp_mode == 2) {
        /* the blocks whose coarse bound reaches the best fine key under the best coarse node */
        if (p.fine.ncb() > 4096 || (size_t)p.n_theta * tp->nxs * tp->nys >= (1u << 26) || p.n > 4096)
            return fail(ctx, CSM_EINVAL, "internal: window too large for the two-phase work list");
        const size_t n_blocks = nt * ncb;
        if ((rc = ensure(ctx, ctx->tp_items, 64 + csm::kReducedBest * sizeof(BlockBest) + n_blocks * 5))) return rc;
        unsigned long long* best2 = reinterpret_cast<unsigned long long*>(ctx->tp_items.p);
        uint32_t* count = reinterpret_cast<uint32_t*>(best2 + 2);
        BlockBest* reduced = reinterpret_cast<BlockBest*>(reinterpret_cast<char*>(ctx->tp_items.p) + 64);
        uint32_t* items = reinterpret_cast<uint32_t*>(reduced + csm::kReducedBest);
        unsigned char* keep = reinterpret_cast<unsigned char*>(items + n_blocks);
        HIP_TRY(ctx, hipMemsetAsync(ctx->tp_items.p, 0, 64, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(keep, 0, n_blocks, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->best.p, 0, n_blocks * sizeof(BlockBest), ctx->stream));
        csm::TwoPhaseJob J;
        std::memset(&J, 0, sizeof(J));
        J.coarse_s = tp->level_s;
        J.coarse_k = tp->level_k;
        J.n_theta = p.n_theta;
        J.nxc = p.nxc;
        J.nyc = p.nyc;
        J.nxs = tp->nxs;
        J.nys = tp->nys;
        J.L = p.L;
        J.min_known = w->min_known;
        J.cells = g.levels[0].cells;
        J.rows = g.rows;
        J.cols = g.cols;
        J.pitch = g.pitch;
        J.hit_col = hit_col_dev;
        J.hit_row = hit_row_dev;
        J.n_points = p.n;
        J.x_lo = p.x_lo;
        J.y_lo = p.y_lo;
        J.nx = p.nx;
        J.ny = p.ny;
        J.cbx = p.fine.cbx;
        J.cby = p.fine.groups * p.fine.R;
        J.ncbx = p.fine.ncbx;
        J.ncb = ncb;
        J.flags = flags;
        J.best = best2;
        J.items = items;
        J.count = count;
        J.keep = keep;
        J.cap = (uint32_t)n_blocks;
        {
            ScopedTimer tm(ctx, "select");
            int e = csm::launch_coarse_best(ctx->stream, J);
            if (!e) e = csm::launch_fine_under_best(ctx->stream, J);
            if (!e) e = csm::launch_mark_blocks(ctx->stream, J);
            if (e)
                return fail(ctx, CSM_EIO, "two-phase select launch failed: %s", hipGetErrorString((hipError_t)e));
        }
        {
            ScopedTimer tm(ctx, "score_fine");
            if ((rc = launch_score_list(ctx, fj, p.fine, items, count, (int)std::min<size_t>(n_blocks, 2048))))
                return rc;
        }
        /* k_finalize reads kReducedBest records instead of one per block of the window */
        if ((rc = launched_ok(ctx, csm::launch_reduce_items(ctx->stream, fj.block_best, items, count, (uint32_t)n_blocks,
                                                            ncb, reduced), "record reduction")))
            return rc;
        tp_reduced = reduced;
        ctx->tp_count_dev = count;
        ctx->tp_blocks_total = (int64_t)n_blocks;
    }
    if (tp_mode == 2) {
        /* launched above */
    } else if (fine_slices > 1) {
        /* small windows: too few workgroups to fill the chip, so the tile list
         * is split over blockIdx.z, the slices add their exact integer sums
         * with atomics, and a second pass does the arg-max */
        ScoreJob sj = fj;
        sj.block_best = nullptr;
        sj.dump_s = nullptr;
        sj.dump_k = nullptr;
        sj.acc_s = reinterpret_cast<uint32_t*>(ctx->fine_s.p);
        sj.acc_k = reinterpret_cast<uint32_t*>(ctx->fine_k.p);
        sj.acc_x_major = 1;
        ctx->fine_acc_dirty = true;
        {
            ScopedTimer tm(ctx, "score_fine");
            if ((rc = launch_score(ctx, sj, p.fine, p.n_theta, fine_slices)))
                return rc;
        }
        ScoreJob aj = fj;
        aj.in_s = sj.acc_s;
        aj.in_k = sj.acc_k;
        ScopedTimer tm(ctx, "argmax");
        if ((rc = launch_argmax(ctx, aj, p.fine, p.n_theta)))
            return rc;
        ctx->fine_acc_dirty = false;
    } else {
        ScopedTimer tm(ctx, "score_fine");
        if ((rc = launch_score(ctx, fj, p.fine, p.n_theta, 1)))
            return rc;
    }
    ctx->last_fine = fj;

    FinalJob fin;
    std::memset(&fin, 0, sizeof(fin));
    fin.block_best = fj.block_best;
    fin.n_entries = p.n_theta * ncb;
    if (tp_reduced) {
        fin.block_best = tp_reduced;
        fin.n_entries = csm::kReducedBest;
    }
    fin.nx = p.nx;
    fin.ny = p.ny;
    fin.rank_l = p.L;
    fin.x_lo = p.x_lo;
    fin.y_lo = p.y_lo;
    fin.win_theta = (p.n_theta - 1) / 2;
    fin.init_x = -p.win_x;
    fin.init_y = -p.win_y;
    fin.init_theta = -fin.win_theta;
    fin.cells = g.levels[0].cells;
    fin.rows = g.rows;
    fin.cols = g.cols;
    fin.pitch = g.pitch;
    fin.hit_col = hit_col_dev;
    fin.hit_row = hit_row_dev;
    fin.n_points = p.n;
    fin.score_thr = w->score_threshold;
    fin.lut = ctx->lut_dev;
    fin.flags_in = flags;
    fin.flags_clear = flags_next;
    fin.out = out_dev;
    {
        const size_t lds = (size_t)p.n * 8;
        ScopedTimer tm(ctx, "finalize");
        if ((rc = launched_ok(ctx, csm_launch::finalize(ctx->stream, ctx->device, lds, fin), "finalize"))) return rc;
    }
    return CSM_OK;
}


/* The phase-major copy of box-max level `level` of g for coarse windows of up to `need` candidates
 * per axis (its zero padding), built on first use and whenever the level changed. */
int ensure_phase_map(csm_ctx* ctx, DeviceGrid& g, int level, int need, PhaseMap** out)
{
    const int L = g.levels[level].win;
    PhaseMap& pm = g.phase[L];
    const uint16_t* src = g.levels[level].cells;
    if (pm.grid && pm.built_from == src && pm.epoch == g.base_epoch && pm.pad >= need + 2) {
        *out = &pm;
        return CSM_OK;
    }
    const int pad = std::max(need + 2, pm.pad);
    const int rows_c = ceil_div(g.rows, L), cols_c = ceil_div(g.cols, L);
    const int hp = rows_c + 2 * pad, wp = cols_c + 2 * pad;
    if (!pm.grid)
        pm.grid.reset(new DeviceGrid());
    DeviceGrid& pg = *pm.grid;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_levels(pg, false);
    pg.rows = L * hp;
    pg.cols = L * wp;
    pg.pitch = (pg.cols + 7) & ~7;
    pg.known_r0 = 0;
    pg.known_c0 = 0;
    Level base;
    const size_t bytes = (size_t)pg.rows * pg.pitch * 2;
    if (hipMalloc(reinterpret_cast<void**>(&base.cells), bytes) != hipSuccess)
        return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
    base.win = 1;
    base.owned = true;
    base.cap = bytes;
    pg.levels.push_back(base);
    const int e = csm::launch_phase_map(ctx->stream, src, g.rows, g.cols, g.pitch, L, hp, wp, pad, base.cells, pg.pitch);
    if (e)
        return fail(ctx, CSM_EIO, "k_phase_map launch failed: %s", hipGetErrorString((hipError_t)e));
    pm.hp = hp;
    pm.wp = wp;
    pm.pad = pad;
    pm.built_from = src;
    pm.epoch = g.base_epoch;
    *out = &pm;
    return CSM_OK;
}

/* Is this window searched coarse-first? Large windows only (the coarse pass, the selection and a
 * second binning cost more than they save on a window the exhaustive kernel finishes in 50 us). */
bool wants_two_phase(const csm_ctx* ctx, const Plan& p)
{
    if (ctx->tune.two_phase < 0 || p.L < 2 || !p.fine.pairs || p.fine.ncb() > 4096 || p.n > 4096)
        return false;
    const size_t nodes = (size_t)p.n_theta * (p.nxc + 1) * (p.nyc + 1);
    if (nodes >= (1u << 26))
        return false;
    return ctx->tune.two_phase > 0 || (double)p.n_theta * p.nx * p.ny >= 3.0e7;
}

/* One window, device-resident hit indices: exhaustive (run_window) or coarse-first. */
int search_window(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p, const int32_t* col_dev,
                  const int32_t* row_dev, csm_result* out_dev)
{
    ctx->last_nominal = (int64_t)p.n_theta * p.nx * p.ny;
    ctx->last_coarse_nodes = 0;
    ctx->last_fine_candidates = ctx->last_nominal;
    ctx->tp_count_dev = nullptr;
    if (!wants_two_phase(ctx, p))
        return run_window(ctx, g, w, p, col_dev, row_dev, out_dev, nullptr);
    int rc;
    PhaseMap* pm = nullptr;
    if ((rc = ensure_phase_map(ctx, g, w->coarse_level, std::max(p.nxc, p.nyc) + 1, &pm))) return rc;
    /* the coarse window on the phase-major copy: candidate (xc, yc) = offsets (xc - wcx, yc - wcy) */
    csm_window wc = *w;
    wc.win_x = p.nxc / 2;
    wc.win_y = p.nyc / 2;
    wc.low_resolution = 1;
    wc.coarse_level = 0;
    Plan pc;
    if ((rc = make_plan(ctx, *pm->grid, &wc, &pc))) return rc;
    const size_t hn = (size_t)p.n_theta * p.n;
    if ((rc = ensure(ctx, ctx->ph_hits, hn * 8 + 256))) return rc;
    int32_t* pcol = reinterpret_cast<int32_t*>(ctx->ph_hits.p);
    int32_t* prow = pcol + hn;
    {
        ScopedTimer tm(ctx, "project");
        const int e = csm::launch_phase_hits(ctx->stream, col_dev, row_dev, hn, p.x_lo, p.y_lo, p.L, pm->hp, pm->wp,
                                             pm->pad, ceil_div(g.rows, p.L), ceil_div(g.cols, p.L), wc.win_x, wc.win_y,
                                             pcol, prow);
        if (e)
            return fail(ctx, CSM_EIO, "k_phase_hits launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    TwoPhaseCtl tp;
    tp.mode = 1;
    if ((rc = run_window(ctx, *pm->grid, &wc, pc, pcol, prow, nullptr, nullptr, false, &tp))) return rc;
    tp.mode = 2;
    if ((rc = run_window(ctx, g, w, p, col_dev, row_dev, out_dev, nullptr, false, &tp))) return rc;
    ctx->last_coarse_nodes = (int64_t)p.n_theta * p.nxc * p.nyc;
    ctx->last_fine_candidates = -1;         /* from the device counters, on request (csm_last_search_info) */
    ctx->last_block_candidates = (int64_t)p.fine.cbx * p.fine.groups * p.fine.R;
    return CSM_OK;
}

const uint32_t kTieCap = 1u << 16;
const uint32_t kUncCap = 4096;

/* Several candidates share the best integer key: collect them with a second
 * fine pass, replay each in f64, pick like the reference's strict `<`. */
int resolve_ties(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                 const int32_t* col_dev, const int32_t* row_dev, csm_result* out_dev)
{
    int rc;
    if ((rc = ensure(ctx, ctx->tie, (size_t)kTieCap * 16 + 64))) return rc;
    unsigned long long* list = reinterpret_cast<unsigned long long*>(ctx->tie.p);
    double* score = reinterpret_cast<double*>(list + kTieCap);
    uint32_t* count = reinterpret_cast<uint32_t*>(score + kTieCap);
    HIP_TRY(ctx, hipMemsetAsync(count, 0, 4, ctx->stream));
    ScoreJob cj = ctx->last_fine;
    cj.block_best = nullptr;
    cj.dump_s = nullptr;
    cj.dump_k = nullptr;
    cj.collect_key = reinterpret_cast<const unsigned long long*>(
        reinterpret_cast<const char*>(out_dev) + offsetof(csm_result, key));
    cj.tie_list = list;
    cj.tie_count = count;
    cj.tie_cap = kTieCap;
    if ((rc = launch_score(ctx, cj, p.fine, p.n_theta, 1)))
        return rc;
    TieJob tj;
    std::memset(&tj, 0, sizeof(tj));
    tj.tie_list = list;
    tj.tie_count = count;
    tj.tie_cap = kTieCap;
    tj.tie_score = score;
    tj.nx = p.nx;
    tj.ny = p.ny;
    tj.rank_l = p.L;
    tj.x_lo = p.x_lo;
    tj.y_lo = p.y_lo;
    tj.win_theta = (p.n_theta - 1) / 2;
    tj.cells = g.levels[0].cells;
    tj.rows = g.rows;
    tj.cols = g.cols;
    tj.pitch = g.pitch;
    tj.hit_col = col_dev;
    tj.hit_row = row_dev;
    tj.n_points = p.n;
    tj.score_thr = w->score_threshold;
    tj.lut = ctx->lut_dev;
    tj.out = out_dev;
    uint32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    n = std::min(n, kTieCap);
    if (n == 0)
        return CSM_OK;
    const size_t lds = (size_t)p.n * 8;
    return launched_ok(ctx, csm_launch::tie_replay_pick(ctx->stream, ctx->device, (unsigned)n, lds, tj), "tie replay");
}

/* The reference's sequential sweep over device-computed exact scores: used
 * when some coarse node fails to bound its fine candidates (negative edge
 * band) or the tie list overflows. */
int resolve_literal(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                    const int32_t* col_dev, const int32_t* row_dev, csm_result* out_dev)
{
    int rc;
    const size_t nf = (size_t)p.n_theta * p.nx * p.ny;
    const size_t nc = (size_t)p.n_theta * p.nxc * p.nyc;
    if ((rc = ensure(ctx, ctx->ex_fine, nf * 8))) return rc;
    if ((rc = ensure(ctx, ctx->ex_fine_k, nf * 4))) return rc;
    if ((rc = ensure(ctx, ctx->ex_coarse, nc * 8))) return rc;
    if ((rc = ensure(ctx, ctx->ex_coarse_k, nc * 4))) return rc;
    ExactJob ej;
    std::memset(&ej, 0, sizeof(ej));
    ej.rows = g.rows;
    ej.cols = g.cols;
    ej.pitch = g.pitch;
    ej.hit_col = col_dev;
    ej.hit_row = row_dev;
    ej.n_theta = p.n_theta;
    ej.n_points = p.n;
    ej.x_lo = p.x_lo;
    ej.y_lo = p.y_lo;
    ej.lut = ctx->lut_dev;
    ExactJob cj = ej;
    cj.cells = g.levels[w->coarse_level].cells;
    cj.nx = p.nxc;
    cj.ny = p.nyc;
    cj.stride = p.L;
    cj.out_score = reinterpret_cast<double*>(ctx->ex_coarse.p);
    cj.out_k = reinterpret_cast<uint32_t*>(ctx->ex_coarse_k.p);
    if (int e = csm_launch::exact_scores(ctx->stream, (unsigned)((nc + kBlock - 1) / kBlock), cj))
        return launched_ok(ctx, e, "exact score");
    ExactJob fj = ej;
    fj.cells = g.levels[0].cells;
    fj.nx = p.nx;
    fj.ny = p.ny;
    fj.stride = 1;
    fj.out_score = reinterpret_cast<double*>(ctx->ex_fine.p);
    fj.out_k = reinterpret_cast<uint32_t*>(ctx->ex_fine_k.p);
    if (int e = csm_launch::exact_scores(ctx->stream, (unsigned)((nf + kBlock - 1) / kBlock), fj))
        return launched_ok(ctx, e, "exact score");
    LiteralJob lj;
    std::memset(&lj, 0, sizeof(lj));
    lj.coarse_score = cj.out_score;
    lj.coarse_k = cj.out_k;
    lj.fine_score = fj.out_score;
    lj.n_theta = p.n_theta;
    lj.nxc = p.nxc;
    lj.nyc = p.nyc;
    lj.L = p.L;
    lj.x_lo = p.x_lo;
    lj.y_lo = p.y_lo;
    lj.win_theta = (p.n_theta - 1) / 2;
    lj.min_known = w->min_known;
    lj.score_thr = w->score_threshold;
    lj.out = out_dev;
    return launched_ok(ctx, csm_launch::literal_scan(ctx->stream, lj), "literal sweep");
}

/* Finish a window whose fast-path record carries a tie or an edge-band flag. */
int resolve_window(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                   const int32_t* col_dev, const int32_t* row_dev, csm_result* out_dev,
                   const csm_result* have, bool* changed)
{   /* have: the record as already read back by the caller (saves a copy and a wait per query) */
    csm_result r;
    if (have) {
        r = *have;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(&r, out_dev, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (changed)
        *changed = (!(r.flags & CSM_FLAG_EDGE_BAND) && r.tie_count > 1) || (r.flags & CSM_FLAG_EDGE_BAND);
    int rc;
    if (!(r.flags & CSM_FLAG_EDGE_BAND) && r.tie_count > 1) {
        if ((rc = resolve_ties(ctx, g, w, p, col_dev, row_dev, out_dev))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(&r, out_dev, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (r.flags & CSM_FLAG_EDGE_BAND)
        if ((rc = resolve_literal(ctx, g, w, p, col_dev, row_dev, out_dev))) return rc;
    return CSM_OK;
}

} /* namespace csm_host */

extern "C" {

int csm_score_window_dev(csm_ctx* ctx, uint64_t map_id, const csm_window* w,
                         const int32_t* hit_col_dev, const int32_t* hit_row_dev, csm_result* out_dev)
{
    if (!ctx || !w || !hit_col_dev || !hit_row_dev || !out_dev)
        return fail(ctx, CSM_EINVAL, "csm_score_window_dev: bad arguments");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Plan p;
    int rc = make_plan(ctx, *g, w, &p);
    if (rc)
        return rc;
    return run_window(ctx, *g, w, p, hit_col_dev, hit_row_dev, out_dev, nullptr);
}

int csm_resolve_window_dev(csm_ctx* ctx, uint64_t map_id, const csm_window* w,
                           const int32_t* hit_col_dev, const int32_t* hit_row_dev,
                           csm_result* out_dev)
{
    if (!ctx || !w || !hit_col_dev || !hit_row_dev || !out_dev)
        return fail(ctx, CSM_EINVAL, "csm_resolve_window_dev: bad arguments");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    Plan p;
    int rc = make_plan(ctx, *g, w, &p);
    if (rc)
        return rc;
    return resolve_window(ctx, *g, w, p, hit_col_dev, hit_row_dev, out_dev);
}

int csm_score_window_dump(csm_ctx* ctx, uint64_t map_id, const csm_window* w, const int32_t* hit_col,
                          const int32_t* hit_row, csm_result* out, uint32_t* dump_s,
                          uint16_t* dump_k, uint16_t* dump_coarse_k)
{
    if (!ctx || !w || !hit_col || !hit_row || !out)
        return fail(ctx, CSM_EINVAL, "csm_score_window: bad arguments");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Plan p;
    int rc = make_plan(ctx, *g, w, &p);
    if (rc)
        return rc;
    const size_t hn = (size_t)p.n_theta * p.n;
    if ((rc = ensure(ctx, ctx->hits, hn * 8 + 256))) return rc;
    int32_t* col_dev = reinterpret_cast<int32_t*>(ctx->hits.p);
    int32_t* row_dev = col_dev + hn;
    csm_result* res_dev = reinterpret_cast<csm_result*>(row_dev + hn);
    HIP_TRY(ctx, hipMemcpyAsync(col_dev, hit_col, hn * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(row_dev, hit_row, hn * 4, hipMemcpyHostToDevice, ctx->stream));
    WindowOutputs dumps;
    const size_t nc = (size_t)p.n_theta * p.nx * p.ny;
    if (dump_s) {
        if ((rc = ensure(ctx, ctx->dump_s, nc * 4))) return rc;
        dumps.dump_s = reinterpret_cast<uint32_t*>(ctx->dump_s.p);
    }
    if (dump_k) {
        if ((rc = ensure(ctx, ctx->dump_k, nc * 2))) return rc;
        dumps.dump_k = reinterpret_cast<uint16_t*>(ctx->dump_k.p);
    }
    rc = run_window(ctx, *g, w, p, col_dev, row_dev, res_dev, (dump_s || dump_k) ? &dumps : nullptr,
                    dump_coarse_k != nullptr);
    if (rc)
        return rc;
    if ((rc = resolve_window(ctx, *g, w, p, col_dev, row_dev, res_dev)))
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out, res_dev, sizeof(csm_result), hipMemcpyDeviceToHost, ctx->stream));
    if (dump_s)
        HIP_TRY(ctx, hipMemcpyAsync(dump_s, dumps.dump_s, nc * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dump_k)
        HIP_TRY(ctx, hipMemcpyAsync(dump_k, dumps.dump_k, nc * 2, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint32_t> ck32;
    if (dump_coarse_k && p.L > 1) {
        ck32.resize((size_t)p.n_theta * p.nxc * p.nyc);
        HIP_TRY(ctx, hipMemcpyAsync(ck32.data(), ctx->coarse_k.p, ck32.size() * 4,
                                    hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ck32.size(); ++i)
        dump_coarse_k[i] = (uint16_t)ck32[i];
    return CSM_OK;
}

int csm_score_window(csm_ctx* ctx, uint64_t map_id, const csm_window* w, const int32_t* hit_col,
                     const int32_t* hit_row, csm_result* out)
{
    return csm_score_window_dump(ctx, map_id, w, hit_col, hit_row, out, nullptr, nullptr, nullptr);
}

int csm_correlative_match(csm_ctx* ctx, uint64_t map_id, const csm_geometry* geom,
                          const csm_scan* scan, const double initial_pose[3],
                          const csm_correlative_params* prm, csm_summary* out)
{
    if (!ctx || !geom || !scan || !initial_pose || !prm || !out || scan->n_points < 1 ||
        prm->low_resolution < 1)
        return fail(ctx, CSM_EINVAL, "csm_correlative_match: bad arguments");
    if (!scan->angles || !scan->ranges || !scan_is_finite(scan))
        return fail(ctx, CSM_EINVAL, "csm_correlative_match: scan holds a non-finite range or angle");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof(*out));
    const auto t0 = std::chrono::steady_clock::now();
    int level = 0;
    int rc = level_for_window(ctx, *g, prm->low_resolution, &level);
    if (rc)
        return rc;
    /* no wait here: a rebuilt coarse level is ordered before the search on the stream;
     * input_setup_us is the host side of the set-up */
    const auto t1 = std::chrono::steady_clock::now();

    /* scan_matcher_correlative.cpp:130-146 */
    csm_host_compound(initial_pose, scan->relative_sensor_pose, out->sensor_pose);
    csm_host_search_step(geom->resolution, scan->ranges, scan->n_points, &out->step_x,
                         &out->step_y, &out->step_theta);
    out->win_x = csm_host_window(prm->range_x, out->step_x);
    out->win_y = csm_host_window(prm->range_y, out->step_y);
    out->win_theta = csm_host_window(prm->range_theta, out->step_theta);

    csm_window w;
    std::memset(&w, 0, sizeof(w));
    w.n_theta = 2 * out->win_theta + 1;
    w.n_points = scan->n_points;
    w.win_x = out->win_x;
    w.win_y = out->win_y;
    w.low_resolution = prm->low_resolution;
    w.coarse_level = level;
    w.min_known = csm_host_min_known(scan->n_points, prm->known_rate_threshold);
    w.score_threshold = prm->score_threshold;
    w.merge_mode = merging_pays(scan->angles, scan->ranges, scan->n_points, geom->resolution) ? 0 : 1;

    /* Projection on the device with a per-entry certificate; the host
     * recomputes (glibc) only the entries that could not be certified. */
    const size_t hn = (size_t)w.n_theta * w.n_points;
    const int n = scan->n_points;
    Plan p;
    if ((rc = make_plan(ctx, *g, &w, &p))) return rc;
    if ((rc = ensure(ctx, ctx->hits, hn * 8 + 256))) return rc;
    if ((rc = ensure(ctx, ctx->unc, 16 + (size_t)kUncCap * 4))) return rc;
    int32_t* col_dev = reinterpret_cast<int32_t*>(ctx->hits.p);
    int32_t* row_dev = col_dev + hn;
    /* result record and the uncertified-entry count sit side by side: one read-back */
    struct Tail {
        csm_result res;
        uint32_t n_unc, pad[3];
    };
    Tail* tail_dev = reinterpret_cast<Tail*>(row_dev + hn);
    csm_result* res_dev = &tail_dev->res;
    uint32_t* unc_count = &tail_dev->n_unc;
    uint32_t* unc_list = reinterpret_cast<uint32_t*>(ctx->unc.p) + 4;
    /* One query's stream work: [projection job | angles | ranges] up from a pinned block, the
     * projection, the search, [record | uncertified count] back into the pinned block. The same
     * sequence for every query of one launch shape, so from the third query of a shape on it is
     * replayed as a HIP graph (one launch instead of nine; every varying input lives in the pinned
     * block or in device memory the nodes point at). */
    const size_t job_bytes = (sizeof(ProjJob) + 255) & ~(size_t)255;
    const size_t up_bytes = job_bytes + (size_t)n * 16;
    const size_t pin_bytes = up_bytes + 256;
    if (pin_bytes > ctx->q_pin_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->q_pin)
            (void)hipHostFree(ctx->q_pin);
        ctx->q_pin = nullptr;
        ctx->q_pin_cap = 0;
        ++ctx->alloc_epoch;
        if (hipHostMalloc(&ctx->q_pin, pin_bytes + pin_bytes / 4, hipHostMallocDefault) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipHostMalloc(%zu) failed", pin_bytes);
        ctx->q_pin_cap = pin_bytes + pin_bytes / 4;
    }
    if ((rc = ensure(ctx, ctx->q_dev, up_bytes))) return rc;
    char* pin = reinterpret_cast<char*>(ctx->q_pin);
    char* qd = reinterpret_cast<char*>(ctx->q_dev.p);
    double* ang_dev = reinterpret_cast<double*>(qd + job_bytes);
    double* rng_dev = ang_dev + n;
    Tail* tail_pin = reinterpret_cast<Tail*>(pin + ((up_bytes + 63) & ~(size_t)63));
    ProjJob pj;
    std::memset(&pj, 0, sizeof(pj));
    pj.angles = ang_dev;
    pj.ranges = rng_dev;
    pj.hit_col = col_dev;
    pj.hit_row = row_dev;
    pj.unc_count = unc_count;
    pj.unc_list = unc_list;
    pj.unc_cap = kUncCap;
    pj.n_theta = w.n_theta;
    pj.n_points = n;
    pj.win_theta = out->win_theta;
    pj.sensor_x = out->sensor_pose[0];
    pj.sensor_y = out->sensor_pose[1];
    pj.sensor_theta = out->sensor_pose[2];
    pj.step_theta = out->step_theta;
    pj.off_x = geom->offset_x;
    pj.off_y = geom->offset_y;
    pj.res = geom->resolution;
    std::memcpy(pin, &pj, sizeof(pj));
    std::memcpy(pin + job_bytes, scan->angles, (size_t)n * 8);
    std::memcpy(pin + job_bytes + (size_t)n * 8, scan->ranges, (size_t)n * 8);
    const bool two_phase = wants_two_phase(ctx, p);
    auto enqueue = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(qd, pin, up_bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(unc_count, 0, 16, ctx->stream));
        {
            ScopedTimer tm(ctx, "project");
            const int pb = ceil_div(n, kBlock);
            if (int e = csm_launch::project_batch(ctx->stream, dim3(pb, proj_theta_groups(w.n_theta, pb), 1),
                                                  reinterpret_cast<const ProjJob*>(qd)))
                return launched_ok(ctx, e, "projection");
        }
        int rc2 = search_window(ctx, *g, &w, p, col_dev, row_dev, res_dev);
        if (rc2)
            return rc2;
        HIP_TRY(ctx, hipMemcpyAsync(tail_pin, tail_dev, sizeof(Tail), hipMemcpyDeviceToHost, ctx->stream));
        return CSM_OK;
    };
    /* what a graph of this chain has baked in */
    std::vector<uint64_t> key = {
        ctx->alloc_epoch, (uint64_t)(uintptr_t)ctx->stream, (uint64_t)(uintptr_t)g->levels[0].cells,
        (uint64_t)(uintptr_t)g->levels[level].cells, (uint64_t)(uintptr_t)g->xg, (uint64_t)g->xg_pad,
        (uint64_t)g->rows, (uint64_t)g->cols, (uint64_t)g->known_r0, (uint64_t)g->known_c0,
        (uint64_t)w.n_theta, (uint64_t)n, (uint64_t)w.win_x, (uint64_t)w.win_y, (uint64_t)w.low_resolution,
        (uint64_t)(uint32_t)w.min_known, (uint64_t)w.merge_mode, 0 };
    std::memcpy(&key.back(), &w.score_threshold, 8);
    bool launched = false;
    if (!two_phase && !ctx->timing && ctx->tune.graphs && !g->xg_stale) {
        auto it = ctx->graphs.find(key);
        if (it != ctx->graphs.end()) {
            HIP_TRY(ctx, hipGraphLaunch(it->second, ctx->stream));
            ctx->last_nominal = (int64_t)p.n_theta * p.nx * p.ny;
            ctx->last_coarse_nodes = 0;
            ctx->last_fine_candidates = ctx->last_nominal;
            ctx->tp_count_dev = nullptr;
            launched = true;
        } else if (++ctx->graph_seen[key] >= 3) {
            /* third query of this shape: every workspace has its size; record the chain */
            if (ctx->graphs.size() >= 8) {
                for (auto& kv : ctx->graphs)
                    (void)hipGraphExecDestroy(kv.second);
                ctx->graphs.clear();
                ctx->graph_seen.clear();
            }
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
                ctx->capturing = true;
                const int rc_cap = enqueue();
                ctx->capturing = false;
                const hipError_t e_end = hipStreamEndCapture(ctx->stream, &graph);
                if (rc_cap == CSM_OK && e_end == hipSuccess && graph &&
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    ctx->graphs[key] = exec;
                    HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
                    launched = true;
                }
                if (graph)
                    (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
            }
        }
    }
    if (!launched && (rc = enqueue()))
        return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    Tail tail = *tail_pin;
    {
        bool changed = false;
        if ((rc = resolve_window(ctx, *g, &w, p, col_dev, row_dev, res_dev, &tail.res, &changed))) return rc;
        if (changed) {
            HIP_TRY(ctx, hipMemcpyAsync(&tail.res, res_dev, sizeof(csm_result), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    out->raw = tail.res;
    const uint32_t n_unc = tail.n_unc;
    if (n_unc > 0) {
        /* recompute the uncertified entries exactly as the reference does */
        std::vector<int32_t> col(hn), row(hn);
        HIP_TRY(ctx, hipMemcpy(col.data(), col_dev, hn * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(row.data(), row_dev, hn * 4, hipMemcpyDeviceToHost));
        bool patched = false;
        if (n_unc > kUncCap) {
            std::vector<int32_t> c2(hn), r2(hn);
            csm_host_project(geom, out->sensor_pose, out->step_theta, out->win_theta, scan->angles,
                             scan->ranges, n, c2.data(), r2.data(), nullptr, nullptr);
            patched = c2 != col || r2 != row;
            col.swap(c2);
            row.swap(r2);
        } else {
            std::vector<uint32_t> list(n_unc);
            HIP_TRY(ctx, hipMemcpy(list.data(), unc_list, (size_t)n_unc * 4, hipMemcpyDeviceToHost));
            for (uint32_t idx : list) {
                const int t = (int)(idx / (uint32_t)n) - out->win_theta;
                const int i = (int)(idx % (uint32_t)n);
                const double theta = out->sensor_pose[2] + out->step_theta * t;
                const double hx = out->sensor_pose[0] + scan->ranges[i] * std::cos(theta + scan->angles[i]);
                const double hy = out->sensor_pose[1] + scan->ranges[i] * std::sin(theta + scan->angles[i]);
                const int32_t c = static_cast<int>(std::floor((hx - geom->offset_x) / geom->resolution));
                const int32_t r = static_cast<int>(std::floor((hy - geom->offset_y) / geom->resolution));
                if (c != col[idx] || r != row[idx]) {
                    col[idx] = c;
                    row[idx] = r;
                    patched = true;
                }
            }
        }
        if (patched) {
            HIP_TRY(ctx, hipMemcpy(col_dev, col.data(), hn * 4, hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(row_dev, row.data(), hn * 4, hipMemcpyHostToDevice));
            if ((rc = search_window(ctx, *g, &w, p, col_dev, row_dev, res_dev))) return rc;
            if ((rc = resolve_window(ctx, *g, &w, p, col_dev, row_dev, res_dev))) return rc;
            HIP_TRY(ctx, hipMemcpy(&out->raw, res_dev, sizeof(csm_result), hipMemcpyDeviceToHost));
        }
    }
    const auto t2 = std::chrono::steady_clock::now();

    out->pose_found = out->raw.found;
    /* scan_matcher_correlative.cpp:203-206, 214-216 */
    out->best_sensor_pose[0] = out->sensor_pose[0] + out->raw.best_x * out->step_x;
    out->best_sensor_pose[1] = out->sensor_pose[1] + out->raw.best_y * out->step_y;
    out->best_sensor_pose[2] = out->sensor_pose[2] + out->raw.best_theta * out->step_theta;
    csm_host_move_backward(out->best_sensor_pose, scan->relative_sensor_pose, out->estimated_pose);
    const int nx = ceil_div(2 * w.win_x + 1, w.low_resolution) * w.low_resolution;
    const int ny = ceil_div(2 * w.win_y + 1, w.low_resolution) * w.low_resolution;
    out->candidates = (int64_t)w.n_theta * nx * ny;
    out->input_setup_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    out->optimization_us = std::chrono::duration<double, std::micro>(t2 - t1).count();
    return CSM_OK;
}



/* ScanMatcherGridSearch::OptimizePose (scan_matcher_grid_search.cpp:69-190) */
int csm_grid_search_match(csm_ctx* ctx, uint64_t map_id, const csm_geometry* geom,
                          const csm_scan* scan, const double initial_pose[3],
                          const csm_grid_search_params* prm, csm_summary* out)
{
    if (!ctx || !geom || !scan || !initial_pose || !prm || !out || scan->n_points < 1 ||
        !(prm->step_x > 0.0) || !(prm->step_y > 0.0) || !(prm->step_theta > 0.0))
        return fail(ctx, CSM_EINVAL, "csm_grid_search_match: bad arguments");
    if (!scan->angles || !scan->ranges || !scan_is_finite(scan))
        return fail(ctx, CSM_EINVAL, "csm_grid_search_match: scan holds a non-finite range or angle");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof(*out));
    const auto t0 = std::chrono::steady_clock::now();
    csm_host_compound(initial_pose, scan->relative_sensor_pose, out->sensor_pose);
    /* the three loops of scan_matcher_grid_search.cpp:118-120: accumulated doubles */
    const double rx = prm->range_x / 2.0, ry = prm->range_y / 2.0, rt = prm->range_theta / 2.0;
    std::vector<double> px, py, th;
    for (double dy = -ry; dy <= ry; dy += prm->step_y)
        py.push_back(out->sensor_pose[1] + dy);
    for (double dx = -rx; dx <= rx; dx += prm->step_x)
        px.push_back(out->sensor_pose[0] + dx);
    for (double dt = -rt; dt <= rt; dt += prm->step_theta)
        th.push_back(out->sensor_pose[2] + dt);
    const int nx = (int)px.size(), ny = (int)py.size(), nt = (int)th.size(), n = scan->n_points;
    out->win_x = nx;
    out->win_y = ny;
    out->win_theta = nt;
    out->step_x = prm->step_x;
    out->step_y = prm->step_y;
    out->step_theta = prm->step_theta;
    for (int k = 0; k < 3; ++k)
        out->best_sensor_pose[k] = out->sensor_pose[k];
    out->raw.best_x = out->raw.best_y = out->raw.best_theta = -1;
    out->raw.score = prm->score_threshold;
    const size_t total = (size_t)nx * ny * nt;
    out->candidates = (int64_t)total;
    if (total > 0) {
        /* ScanData::HitPoint's products per theta value, with glibc */
        std::vector<double> prod(2 * (size_t)nt * n);
        double* rc = prod.data();
        double* rs = rc + (size_t)nt * n;
        for (int k = 0; k < nt; ++k)
            for (int i = 0; i < n; ++i) {
                rc[(size_t)k * n + i] = scan->ranges[i] * std::cos(th[k] + scan->angles[i]);
                rs[(size_t)k * n + i] = scan->ranges[i] * std::sin(th[k] + scan->angles[i]);
            }
        int rc_ = 0;
        const size_t words = (size_t)nx + ny + prod.size();
        if ((rc_ = ensure(ctx, ctx->ex_coarse, words * 8 + 64))) return rc_;
        if ((rc_ = ensure(ctx, ctx->ex_fine, total * 8))) return rc_;
        if ((rc_ = ensure(ctx, ctx->ex_fine_k, total * 4))) return rc_;
        if ((rc_ = ensure(ctx, ctx->tie, 64))) return rc_;
        double* d_px = reinterpret_cast<double*>(ctx->ex_coarse.p);
        double* d_py = d_px + nx;
        double* d_rc = d_py + ny;
        double* d_rs = d_rc + (size_t)nt * n;
        unsigned long long* d_best = reinterpret_cast<unsigned long long*>(ctx->tie.p);
        const unsigned long long init_best[2] = { 0ull, ~0ull };
        HIP_TRY(ctx, hipMemcpyAsync(d_px, px.data(), (size_t)nx * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_py, py.data(), (size_t)ny * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_rc, prod.data(), prod.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_best, init_best, 16, hipMemcpyHostToDevice, ctx->stream));
        GridSearchJob gj;
        std::memset(&gj, 0, sizeof(gj));
        gj.cells = g->levels[0].cells;
        gj.rows = g->rows;
        gj.cols = g->cols;
        gj.pitch = g->pitch;
        gj.px = d_px;
        gj.py = d_py;
        gj.r_cos = d_rc;
        gj.r_sin = d_rs;
        gj.off_x = geom->offset_x;
        gj.off_y = geom->offset_y;
        gj.res = geom->resolution;
        gj.nx = nx;
        gj.ny = ny;
        gj.nt = nt;
        gj.n_points = n;
        gj.min_known = csm_host_min_known(n, prm->known_rate_threshold);
        gj.score_thr = prm->score_threshold;
        gj.lut = ctx->lut_dev;
        gj.out_score = reinterpret_cast<double*>(ctx->ex_fine.p);
        gj.out_k = reinterpret_cast<uint32_t*>(ctx->ex_fine_k.p);
        gj.best_bits = d_best;
        gj.best_index = d_best + 1;
        const unsigned blocks = (unsigned)((total + kBlock - 1) / kBlock);
        {
            ScopedTimer tm(ctx, "grid_search");
            if (int e = csm_launch::grid_scores_pick(ctx->stream, blocks, gj))
                return launched_ok(ctx, e, "grid search");
        }
        unsigned long long best[2] = { 0, 0 };
        HIP_TRY(ctx, hipMemcpyAsync(best, d_best, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (best[0] != 0ull && best[1] != ~0ull) {
            double score;
            const unsigned long long bits = best[0] - 1ull;
            std::memcpy(&score, &bits, 8);
            const size_t p = (size_t)best[1];
            const int it = (int)(p % nt), ix = (int)((p / nt) % nx), iy = (int)(p / ((size_t)nt * nx));
            out->pose_found = 1;
            out->raw.found = 1;
            out->raw.best_x = ix;
            out->raw.best_y = iy;
            out->raw.best_theta = it;
            out->raw.score = score;
            out->best_sensor_pose[0] = px[ix];
            out->best_sensor_pose[1] = py[iy];
            out->best_sensor_pose[2] = th[it];
        }
    }
    csm_host_move_backward(out->best_sensor_pose, scan->relative_sensor_pose, out->estimated_pose);
    out->optimization_us =
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return CSM_OK;
}

/* The device projection (k_project) on its own, for parity tests of A3 */
int csm_project_scan(csm_ctx* ctx, const csm_geometry* geom, const double sensor_pose[3],
                     double step_theta, int32_t win_theta, const double* angles, const double* ranges,
                     int32_t n, int32_t* hit_col, int32_t* hit_row, uint32_t* uncertified,
                     int32_t uncertified_cap, int32_t* n_uncertified)
{
    if (!ctx || !geom || !sensor_pose || !angles || !ranges || n < 1 || win_theta < 0 || !hit_col ||
        !hit_row || !n_uncertified || uncertified_cap < 0 || (uncertified_cap > 0 && !uncertified))
        return fail(ctx, CSM_EINVAL, "csm_project_scan: bad arguments");
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(ranges[i]) || !std::isfinite(angles[i]))
            return fail(ctx, CSM_EINVAL, "csm_project_scan: beam %d is not finite", i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n_theta = 2 * win_theta + 1;
    const size_t hn = (size_t)n_theta * n;
    int rc;
    if ((rc = ensure(ctx, ctx->hits, hn * 8 + 256))) return rc;
    if ((rc = ensure(ctx, ctx->scan_dev, (size_t)n * 16))) return rc;
    if ((rc = ensure(ctx, ctx->unc, 16 + (size_t)std::max(uncertified_cap, 1) * 4))) return rc;
    int32_t* col_dev = reinterpret_cast<int32_t*>(ctx->hits.p);
    int32_t* row_dev = col_dev + hn;
    double* ang_dev = reinterpret_cast<double*>(ctx->scan_dev.p);
    double* rng_dev = ang_dev + n;
    uint32_t* unc_count = reinterpret_cast<uint32_t*>(ctx->unc.p);
    uint32_t* unc_list = unc_count + 4;
    HIP_TRY(ctx, hipMemcpyAsync(ang_dev, angles, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(rng_dev, ranges, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(unc_count, 0, 16, ctx->stream));
    ProjJob pj;
    std::memset(&pj, 0, sizeof(pj));
    pj.angles = ang_dev;
    pj.ranges = rng_dev;
    pj.hit_col = col_dev;
    pj.hit_row = row_dev;
    pj.unc_count = unc_count;
    pj.unc_list = unc_list;
    pj.unc_cap = (uint32_t)uncertified_cap;
    pj.n_theta = n_theta;
    pj.n_points = n;
    pj.win_theta = win_theta;
    pj.sensor_x = sensor_pose[0];
    pj.sensor_y = sensor_pose[1];
    pj.sensor_theta = sensor_pose[2];
    pj.step_theta = step_theta;
    pj.off_x = geom->offset_x;
    pj.off_y = geom->offset_y;
    pj.res = geom->resolution;
    if (int e = csm_launch::project(ctx->stream, dim3(ceil_div(n, kBlock), proj_theta_groups(n_theta, ceil_div(n, kBlock))), pj))
        return launched_ok(ctx, e, "projection");
    uint32_t count = 0;
    HIP_TRY(ctx, hipMemcpyAsync(hit_col, col_dev, hn * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hit_row, row_dev, hn * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&count, unc_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_uncertified = (int32_t)count;
    const uint32_t have = std::min<uint32_t>(count, (uint32_t)uncertified_cap);
    if (have)
        HIP_TRY(ctx, hipMemcpy(uncertified, unc_list, (size_t)have * 4, hipMemcpyDeviceToHost));
    return CSM_OK;
}

} /* extern "C" */

