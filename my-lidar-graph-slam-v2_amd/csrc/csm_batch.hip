/* csm_batch.hip -- many windows in one launch chain (host code only): branch-and-bound and correlative
 * loop-detection batches (run_batch_group: staging, joint binning, bound pass + exact rounds), device-resident
 * window batches (csm_score_windows_dev), their entry points of the C ABI. */
#include "csm_matchers.hpp"

#ifdef CSM_BIN_TIMING
static unsigned long long* g_bin_debug = nullptr;
static uint32_t* bin_debug_buffer()
{
    const size_t bytes = (size_t)kBinDebugRows * 128;
    if (!g_bin_debug && (hipMalloc(reinterpret_cast<void**>(&g_bin_debug), bytes) != hipSuccess ||
                         hipMemset(g_bin_debug, 0, bytes) != hipSuccess))
        g_bin_debug = nullptr;
    return reinterpret_cast<uint32_t*>(g_bin_debug);
}
#endif

namespace csm_host {


struct BatchPrep {
    DeviceGrid* grid = nullptr;
    int level[kMaxElig] = { 0 };   /* index into grid->levels of box-max(2^h) */
    int n_theta = 0, n = 0;
    int win_x = 0, win_y = 0, win_t = 0;
    int nx = 0, ny = 0;
    int tiles_x = 0, tiles_y = 0, max_tiles = 0;
    size_t hit_off = 0, tile_off = 0, theta_off = 0, best_off = 0;
    size_t lvl_off[kMaxElig] = { 0 };
};


/* Node of the reference's best-first search
 * (inc/mapping/scan_matcher_branch_bound.hpp:67-106): ordered by score only. */
struct HeapNode {
    int x, y, t, h;
    double score, known_rate;
    bool operator<(const HeapNode& o) const { return score < o.score; }
};

/* Exact resolution of one flagged branch-and-bound query. The device computes
 * the f64 score (beam order, per-node projection in double) and known count of
 * EVERY node of every level; the host then runs the reference's queue
 * discipline (std::priority_queue, same push / pop order as
 * src/mapping/scan_matcher_branch_bound.cpp:156-231) reading those scores
 * instead of calling Score(). No score is computed on the CPU. */
int bnb_literal(csm_ctx* ctx, const csm_loop_query& q, const BatchPrep& p, const csm_summary& o,
                const csm_bnb_params* prm, csm_result* res)
{
    const int H = prm->node_height_max;
    const int nx = p.nx, ny = p.ny;
    /* the exact path needs the host's own r*cos / r*sin (glibc) */
    const size_t hn = (size_t)p.n_theta * p.n;
    std::vector<double> prod(2 * hn);
    {
        std::vector<int32_t> col(hn), row(hn);
        int prc = csm_host_project(&q.geometry, o.sensor_pose, o.step_theta, p.win_t, q.scan.angles,
                                   q.scan.ranges, p.n, col.data(), row.data(), prod.data(),
                                   prod.data() + hn);
        if (prc)
            return fail(ctx, prc, "projection failed");
    }
    int rc0 = ensure(ctx, ctx->ex_coarse, 2 * hn * 8);
    if (rc0)
        return rc0;
    double* d_rc = reinterpret_cast<double*>(ctx->ex_coarse.p);
    double* d_rs = d_rc + hn;
    HIP_TRY(ctx, hipMemcpyAsync(d_rc, prod.data(), 2 * hn * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<std::vector<double>> sc(H + 1);
    std::vector<std::vector<uint32_t>> kn(H + 1);
    int rc;
    for (int h = 0; h <= H; ++h) {
        const int nxh = nx >> h, nyh = ny >> h;
        const size_t n = (size_t)p.n_theta * nxh * nyh;
        if ((rc = ensure(ctx, ctx->ex_fine, n * 8))) return rc;
        if ((rc = ensure(ctx, ctx->ex_fine_k, n * 4))) return rc;
        ExactJob ej;
        std::memset(&ej, 0, sizeof(ej));
        ej.cells = p.grid->levels[p.level[h]].cells;
        ej.rows = p.grid->rows;
        ej.cols = p.grid->cols;
        ej.pitch = p.grid->pitch;
        ej.r_cos = d_rc;
        ej.r_sin = d_rs;
        ej.sensor_x = o.sensor_pose[0];
        ej.sensor_y = o.sensor_pose[1];
        ej.step_x = o.step_x;
        ej.step_y = o.step_y;
        ej.off_x = q.geometry.offset_x;
        ej.off_y = q.geometry.offset_y;
        ej.res = q.geometry.resolution;
        ej.n_theta = p.n_theta;
        ej.n_points = p.n;
        ej.x_lo = -p.win_x;
        ej.y_lo = -p.win_y;
        ej.nx = nxh;
        ej.ny = nyh;
        ej.stride = 1 << h;
        ej.lut = ctx->lut_dev;
        ej.out_score = reinterpret_cast<double*>(ctx->ex_fine.p);
        ej.out_k = reinterpret_cast<uint32_t*>(ctx->ex_fine_k.p);
        if (int e = csm_launch::exact_scores(ctx->stream, (unsigned)((n + kBlock - 1) / kBlock), ej))
            return launched_ok(ctx, e, "exact score");
        sc[h].resize(n);
        kn[h].resize(n);
        HIP_TRY(ctx, hipMemcpyAsync(sc[h].data(), ej.out_score, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(kn[h].data(), ej.out_k, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }

    const int win_x = p.win_x, win_y = p.win_y, win_t = p.win_t;
    double score_max = prm->score_threshold;
    int best_x = 0, best_y = 0, best_t = 0;
    std::priority_queue<HeapNode> queue;
    const double n_points = static_cast<double>(p.n);
    auto append_node = [&](int x, int y, int t, int h) {
        const int xi = (x + win_x) >> h, yi = (y + win_y) >> h;
        const size_t i = ((size_t)(t + win_t) * (nx >> h) + xi) * (ny >> h) + yi;
        const double score = sc[h][i];
        if (score > score_max)
            queue.push(HeapNode { x, y, t, h, score, static_cast<double>(kn[h][i]) / n_points });
    };
    const int win_size_max = 1 << H;
    for (int x = -win_x; x <= win_x; x += win_size_max)
        for (int y = -win_y; y <= win_y; y += win_size_max)
            for (int t = -win_t; t <= win_t; ++t)
                append_node(x, y, t, H);
    while (!queue.empty()) {
        const HeapNode cur = queue.top();
        if (cur.score <= score_max || cur.known_rate <= prm->known_rate_threshold) {
            queue.pop();
            continue;
        }
        if (cur.h == 0) {
            best_x = cur.x;
            best_y = cur.y;
            best_t = cur.t;
            score_max = cur.score;
            queue.pop();
        } else {
            const int h = cur.h - 1;
            const int wsz = 1 << h;
            queue.pop();
            append_node(cur.x, cur.y, cur.t, h);
            append_node(cur.x + wsz, cur.y, cur.t, h);
            append_node(cur.x, cur.y + wsz, cur.t, h);
            append_node(cur.x + wsz, cur.y + wsz, cur.t, h);
        }
    }
    res->found = score_max > prm->score_threshold ? 1 : 0;
    res->best_x = best_x;
    res->best_y = best_y;
    res->best_theta = best_t;
    res->score = score_max;
    res->flags |= CSM_FLAG_LITERAL;
    return CSM_OK;
}

/* The device copy of a batch's final records: sized here, filled by
 * run_batch_group, handed out by csm_copy_last_batch_records. */
int begin_batch_records(csm_ctx* ctx, int n_queries)
{
    ctx->rec_n = 0;
    int rc = ensure(ctx, ctx->rec_dev, (size_t)n_queries * sizeof(csm_result));
    if (rc)
        return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   /* patches of the previous call have landed */
    ctx->rec_patch.clear();
    ctx->rec_patch.reserve((size_t)n_queries);         /* no reallocation under a pending copy */
    ctx->rec_n = n_queries;
    return CSM_OK;
}

/* What distinguishes the two batched searches. */
struct BatchSpec {
    bool bnb = true;          /* branch and bound (leaf + 2^h levels) or correlative (fine + one
                                 box-max(L) level) */
    int H = 0;                /* number of coarser levels */
    int stride[kMaxElig] = { 1 };   /* stride[j] of level j (stride[0] = 1) */
    int unit = 1;             /* candidate domain is padded to a multiple of this */
    double range_x = 0, range_y = 0, range_theta = 0;
    double score_thr = 0, known_thr = 0;
    const csm_bnb_params* bnb_params = nullptr;
    const csm_correlative_params* csm_params = nullptr;
    const double* max_range = nullptr;   /* [n_queries] largest range of each query's scan */
};

/* csm_score_windows_dev: the windows and hit indices are given (device
 * resident, already projected), the results stay on the device, nothing waits. */
struct ResidentBatch {
    const csm_window* windows;            /* [n] */
    const int32_t* const* hit_col;        /* [n] device pointers, [n_theta][n_points] each */
    const int32_t* const* hit_row;
    csm_result* out_dev;                  /* [n] device */
    uint32_t* const* dump_s = nullptr;    /* optional [n] device pointers (any may be null): every candidate's */
    uint16_t* const* dump_k = nullptr;    /* integer sums, [n_theta][nx][ny] (parity tests) */
    float* const* dump_f = nullptr;       /* optional: every candidate's fp32 key of the bound pass */
};

/* One group of queries that share (nx, ny): the whole device pipeline. */
int run_batch_group(csm_ctx* ctx, const csm_loop_query* queries, const std::vector<int>& idx,
                    const std::vector<std::vector<int>>& levels, const BatchSpec& spec,
                    csm_summary* out, const ResidentBatch* resident = nullptr)
{
    const int H = spec.H;
    const int nq = (int)idx.size();
    const bool host_timing = ctx->tune.host_timing;
    auto tick = [&](const char* what) {
        static thread_local std::chrono::steady_clock::time_point last;
        const auto now = std::chrono::steady_clock::now();
        if (host_timing && what)
            fprintf(stderr, "[run_batch_group nq=%d] %-10s %8.3f ms\n", nq, what,
                    std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };
    tick(nullptr);
    std::vector<BatchPrep> pp(nq);
    std::vector<csm_summary> scratch_out;
    if (resident) {                       /* no host summaries in this mode */
        scratch_out.assign((size_t)*std::max_element(idx.begin(), idx.end()) + 1, csm_summary {});
        out = scratch_out.data();
    }
    int rc;

    /* ---- host set-up: window, projection products (threaded over queries) ---- */
    size_t hit_total = 0, tile_total = 0, theta_total = 0;
    int n_theta_max = 0, n_points_max = 0;
    size_t bin_lds = 0;
    for (int k = 0; k < nq; ++k) {
        const csm_loop_query& q = queries[idx[k]];
        csm_summary& o = out[idx[k]];
        BatchPrep& p = pp[k];
        p.grid = find_grid(ctx, q.map_id);
        for (int h = 0; h <= H; ++h)
            p.level[h] = levels[idx[k]][h];
        if (resident) {
            const csm_window& w = resident->windows[idx[k]];
            p.win_x = w.win_x;
            p.win_y = w.win_y;
            p.win_t = (w.n_theta - 1) / 2;
            p.n_theta = w.n_theta;
            p.n = w.n_points;
        } else {
            csm_host_compound(q.initial_pose, q.scan.relative_sensor_pose, o.sensor_pose);
            search_step_from_max(q.geometry.resolution, spec.max_range[idx[k]], &o.step_x, &o.step_y,
                                 &o.step_theta);
            o.win_x = p.win_x = csm_host_window(spec.range_x, o.step_x);
            o.win_y = p.win_y = csm_host_window(spec.range_y, o.step_y);
            o.win_theta = p.win_t = csm_host_window(spec.range_theta, o.step_theta);
            p.n_theta = 2 * p.win_t + 1;
            p.n = q.scan.n_points;
        }
        const int big = spec.unit;
        p.nx = ceil_div(2 * p.win_x + 1, big) * big;
        p.ny = ceil_div(2 * p.win_y + 1, big) * big;
        p.tiles_x = ceil_div(p.grid->cols + p.win_x + (-p.win_x + p.nx - 1), kTile);
        p.tiles_y = ceil_div(p.grid->rows + p.win_y + (-p.win_y + p.ny - 1) + 1, kTile);
        if (p.n > kMaxPoints)
            return fail(ctx, CSM_EINVAL, "query %d: more than %d beams per scan", idx[k], kMaxPoints);
        n_theta_max = std::max(n_theta_max, p.n_theta);
        n_points_max = std::max(n_points_max, p.n);
    }
    const int nx = pp[0].nx, ny = pp[0].ny;

    /* ---- launch geometry shared by the group ---- */
    std::vector<PassPlan> lp(H + 1);
    for (int h = 0; h <= H; ++h) {
        if (h == 0) {
            if (!plan_pass_pairs(ctx->tune, nx, ny, &lp[0], true))
                return fail(ctx, CSM_EINVAL, "no launch geometry for the fine level");
            continue;
        }
        if (!plan_pass(ctx->tune, nx / spec.stride[h], ny / spec.stride[h], spec.stride[h], &lp[h]))
            return fail(ctx, CSM_EINVAL, "no launch geometry for level %d (stride %d)", h,
                        spec.stride[h]);
    }
    if (resident) {
        lp[0].weighted = resident->windows[idx[0]].merge_mode == 0;
    } else {
        const csm_loop_query& q0 = queries[idx[0]];
        lp[0].weighted = merging_pays(q0.scan.angles, q0.scan.ranges, q0.scan.n_points,
                                      q0.geometry.resolution);
    }
    /* Joint entry lists of slice pairs (k_binj + k_score_joint_batch, csm_joint_kernels.hip): the
     * two-slice plan with merged (weighted) entries, when the joint hash table of every query
     * fits a CU's LDS. Otherwise round 2's per-slice lists. */
    size_t binj_lds = 0;
    for (int k = 0; k < nq; ++k)
        binj_lds = std::max(binj_lds, csm::binj_lds_bytes(pp[k].tiles_x * pp[k].tiles_y, pp[k].n,
                                                          csm::binj_hash_size(pp[k].n)));
    const bool joint = ctx->tune.joint && lp[0].pairs && lp[0].lists == 2 && lp[0].weighted &&
                       binj_lds <= 150 * 1024;     /* up to ~1,100 beams four binning workgroups share a CU, two up
                                                      to ~2,200; one (the fine level's gain outweighs the slower
                                                      binning) up to ~4,200 beams per scan */
    lp[0].joint = joint;
    /* The packed-fp32 bound pass in front of the exact kernel: only where the arg-max is over ALL
     * candidates of the window -- the correlative sweep with a known-rate threshold that the coarse
     * level passes whenever a fine candidate scores at all (min_known <= 1; a touched edge band
     * switches the skipping off per query on the device). Branch and bound tests every leaf's own
     * known count: exact kernel only. */
    bool bound_pass = joint && ctx->tune.bound_pass && nq < (1 << 14) && lp[0].ncb() <= 256 &&
                      (n_theta_max + 1) / 2 <= 1024;       /* the work list's item format */
    /* Where the winner must pass a known-count test the bound pass does not see -- branch and bound:
     * every leaf's own count; the correlative sweep with a known-rate threshold above one beam: the
     * coarse node's count (the reference's loop detectors run with 0.6) -- the window's greatest
     * fp32 key may belong to a candidate that does not count, and the exact pass runs in two rounds
     * (k_bound_select). */
    bool two_rounds = spec.bnb;
    for (int k = 0; k < nq; ++k)
        two_rounds = two_rounds || (resident ? resident->windows[idx[k]].min_known
                                             : csm_host_min_known(pp[k].n, spec.known_thr)) > 1;
    for (int k = 0; k < nq; ++k) {
        BatchPrep& p = pp[k];
        /* lists and records per slice, or per pair of slices (2 n entries each) */
        const int units = joint ? (p.n_theta + 1) / 2 : p.n_theta;
        const int per_unit = joint ? 2 * p.n : p.n;
        p.max_tiles = std::min(per_unit, p.tiles_x * p.tiles_y) + per_unit / (joint ? kJRec : kPbMax) + 1;
        bin_lds = std::max(bin_lds, bin_lds_bytes(p.tiles_x * p.tiles_y, p.n));
        p.hit_off = hit_total;
        p.tile_off = tile_total;
        p.theta_off = theta_total;
        hit_total += (size_t)(p.n_theta + 1) * p.n;         /* >= units * per_unit */
        tile_total += (size_t)units * p.max_tiles;
        theta_total += 2 * (size_t)p.n_theta;     /* record counts + merge flags */
    }
    if (bin_lds > 160 * 1024 - 64)
        return fail(ctx, CSM_EINVAL, "grid + window too large for the binning kernel (its per-tile words, hash table and cell list exceed the LDS)");

    /* scans go to the device as they are (angles, ranges); the projection runs
     * there with a per-entry certificate (k_project) */
    /* Queries that share a scan (one query scan node against many local maps: the
     * usual shape of a Detect() call) share its device copy. The staging buffer is
     * pinned and owned by the context: no clearing, one DMA. */
    std::vector<size_t> scan_off(nq);
    size_t scan_total = 0;
    {
        std::map<std::tuple<const double*, const double*, int>, size_t> seen;
        for (int k = 0; k < nq; ++k) {
            const csm_scan& sc = queries[idx[k]].scan;
            auto key = std::make_tuple(sc.angles, sc.ranges, pp[k].n);
            auto it = resident ? seen.end() : seen.find(key);
            if (it != seen.end()) {
                scan_off[k] = it->second;
                continue;
            }
            scan_off[k] = scan_total;
            if (!resident)
                seen.emplace(key, scan_total);
            scan_total += 2 * (size_t)pp[k].n;
        }
    }
    double* scans = nullptr;
    if (!resident) {
        const size_t need = scan_total * 8;
        if (need > ctx->pin_scans_cap) {
            if (ctx->pin_scans)
                (void)hipHostFree(ctx->pin_scans);
            ctx->pin_scans = nullptr;
            ctx->pin_scans_cap = 0;
            if (hipHostMalloc(&ctx->pin_scans, need + need / 4 + 64, hipHostMallocDefault) != hipSuccess)
                return fail(ctx, CSM_ENOMEM, "hipHostMalloc(%zu) failed", need);
            ctx->pin_scans_cap = need + need / 4 + 64;
        }
        scans = reinterpret_cast<double*>(ctx->pin_scans);
        size_t filled = 0;                  /* scans are laid out in first-use order */
        std::vector<int> first_use;
        for (int k = 0; k < nq; ++k) {
            if (scan_off[k] != filled)
                continue;                   /* a duplicate of an earlier query's scan */
            first_use.push_back(k);
            filled += 2 * (size_t)pp[k].n;
        }
        host_parallel_for((int)first_use.size(), 128, [&](int lo, int hi) {
            for (int j = lo; j < hi; ++j) {
                const int k = first_use[j];
                const csm_loop_query& q = queries[idx[k]];
                std::memcpy(scans + scan_off[k], q.scan.angles, (size_t)pp[k].n * 8);
                std::memcpy(scans + scan_off[k] + pp[k].n, q.scan.ranges, (size_t)pp[k].n * 8);
            }
        });
    }

    tick("setup");
    const int lstride = lp[0].lstride;
    const int ncb = lp[0].ncb();

    /* ---- workspaces ---- */
    size_t lvl_total = 0, best_total = 0;
    for (int k = 0; k < nq; ++k) {
        BatchPrep& p = pp[k];
        for (int h = 1; h <= H; ++h) {
            p.lvl_off[h] = lvl_total;
            lvl_total += (size_t)p.n_theta * (nx / spec.stride[h]) * (ny / spec.stride[h]);
        }
        p.best_off = best_total;
        best_total += (size_t)p.n_theta * ncb;
    }
    if ((rc = ensure(ctx, ctx->b_prod, (resident ? 0 : scan_total * 8) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->b_hits, (resident ? 0 : hit_total * 8) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->b_sorted, hit_total * 4 + 256))) return rc;
    if (lp[0].pairs)
        for (int k = 0; k < nq; ++k)
            if ((rc = ensure_xgrid(ctx, *pp[k].grid, xgrid_pad_for(nx, ny)))) return rc;
    if ((rc = ensure(ctx, ctx->b_sorted_rc, hit_total * 4))) return rc;
    if ((rc = ensure(ctx, ctx->b_tiles, tile_total * sizeof(TileRec)))) return rc;
    if ((rc = ensure(ctx, ctx->b_ntiles, theta_total * 4))) return rc;
    if ((rc = ensure(ctx, ctx->b_lvl, lvl_total * 8 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->b_best, best_total * sizeof(BlockBest)))) return rc;
    if (bound_pass) {
        if (!ctx->bound_stats.p) {
            if ((rc = ensure(ctx, ctx->bound_stats, 64))) return rc;
            HIP_TRY(ctx, hipMemsetAsync(ctx->bound_stats.p, 0, 64, ctx->stream));
        }
        if ((rc = ensure(ctx, ctx->b_abest, best_total * sizeof(float)))) return rc;
        /* work lists of the exact kernel: [2 counts, pad][items 0][items 1], one item per (pair, block) */
        size_t blocks_total = 0;
        for (int k = 0; k < nq; ++k)
            blocks_total += (size_t)((pp[k].n_theta + 1) / 2) * ncb;
        if ((rc = ensure(ctx, ctx->b_items, 64 + 2 * blocks_total * 4))) return rc;
        for (int k = 0; k < nq; ++k)
            if ((rc = ensure_xgrid_f(ctx, *pp[k].grid))) return rc;
    }
    if ((rc = ensure(ctx, ctx->b_out, (size_t)nq * (sizeof(csm_result) + 4)))) return rc;
    const size_t jobs_bytes = (size_t)nq * (sizeof(ProjJob) + sizeof(BinJob) + sizeof(FinalJob) +
                                            (size_t)(H + 1) * sizeof(ScoreJob) + (size_t)H * sizeof(ZeroJob));
    if ((rc = ensure(ctx, ctx->b_jobs, jobs_bytes + 1024))) return rc;

    double* d_scans = reinterpret_cast<double*>(ctx->b_prod.p);
    int32_t* d_col = reinterpret_cast<int32_t*>(ctx->b_hits.p);
    int32_t* d_row = d_col + hit_total;
    uint32_t* d_sorted = reinterpret_cast<uint32_t*>(ctx->b_sorted.p);
    uint32_t* d_sorted_rc = reinterpret_cast<uint32_t*>(ctx->b_sorted_rc.p);
    TileRec* d_tiles = reinterpret_cast<TileRec*>(ctx->b_tiles.p);
    int32_t* d_ntiles = reinterpret_cast<int32_t*>(ctx->b_ntiles.p);
    uint32_t* d_lvl_s = reinterpret_cast<uint32_t*>(ctx->b_lvl.p);
    uint32_t* d_lvl_k = d_lvl_s + lvl_total;
    BlockBest* d_best = reinterpret_cast<BlockBest*>(ctx->b_best.p);
    csm_result* d_out = reinterpret_cast<csm_result*>(ctx->b_out.p);
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(d_out + nq);

    if (!resident)
        HIP_TRY(ctx, hipMemcpyAsync(d_scans, scans, scan_total * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_flags, 0, (size_t)nq * 4, ctx->stream));

    tick("workspace");
    /* ---- job tables ---- */
    std::vector<ProjJob> ij(nq);
    std::vector<BinJob> bj(nq);
    std::vector<FinalJob> fj(nq);
    std::vector<std::vector<ScoreJob>> sj(H + 1, std::vector<ScoreJob>(nq));
    std::vector<ZeroJob> zj((size_t)nq * H);
    size_t zero_words_max = 0;
    for (int k = 0; k < nq; ++k) {
        const csm_loop_query& q = queries[idx[k]];
        const csm_summary& o = out[idx[k]];
        const BatchPrep& p = pp[k];
        const DeviceGrid& g = *p.grid;
        const int x_lo = -p.win_x, y_lo = -p.win_y;
        const int min_known = resident ? resident->windows[idx[k]].min_known
                                       : csm_host_min_known(p.n, spec.known_thr);

        ProjJob& I = ij[k];
        std::memset(&I, 0, sizeof(I));
        I.angles = d_scans + scan_off[k];
        I.ranges = d_scans + scan_off[k] + p.n;
        I.hit_col = resident ? const_cast<int32_t*>(resident->hit_col[idx[k]]) : d_col + p.hit_off;
        I.hit_row = resident ? const_cast<int32_t*>(resident->hit_row[idx[k]]) : d_row + p.hit_off;
        I.flags = d_flags + k;
        I.n_theta = p.n_theta;
        I.n_points = p.n;
        I.win_theta = p.win_t;
        I.sensor_x = o.sensor_pose[0];
        I.sensor_y = o.sensor_pose[1];
        I.sensor_theta = o.sensor_pose[2];
        I.step_theta = o.step_theta;
        I.off_x = q.geometry.offset_x;
        I.off_y = q.geometry.offset_y;
        I.res = q.geometry.resolution;
        I.check_nodes = spec.bnb ? 1 : 0;
        I.flag_uncertain = 1;
        I.x_lo = x_lo;
        I.y_lo = y_lo;
        I.nx = nx;
        I.ny = ny;
        I.step_x = o.step_x;
        I.step_y = o.step_y;

        BinJob& B = bj[k];
        std::memset(&B, 0, sizeof(B));
        B.hit_col = I.hit_col;
        B.hit_row = I.hit_row;
        B.sorted_pb = d_sorted + p.hit_off;
        B.sorted_rc = H > 0 ? d_sorted_rc + p.hit_off : nullptr;
        B.tiles = d_tiles + p.tile_off;
        B.n_tiles = d_ntiles + p.theta_off;
        B.flags = d_flags + k;
        B.n_theta = p.n_theta;
        B.n_points = p.n;
        B.max_tiles = p.max_tiles;
        B.rows = g.rows;
        B.cols = g.cols;
        B.x_lo = x_lo;
        B.y_lo = y_lo;
        B.x_hi = x_lo + nx - 1;
        B.y_hi = y_lo + ny - 1;
        B.tiles_x = p.tiles_x;
        B.tiles_y = p.tiles_y;
        B.known_r0 = g.known_r0;
        B.known_c0 = g.known_c0;
        B.hash_size = joint ? csm::binj_hash_size(p.n) : bin_hash_size(p.n);
        B.max_mult = lp[0].weighted ? kMaxMult : 1;
        B.lstride = lstride;
        B.pair_mode = joint ? 2 : lp[0].pairs ? 1 : 0;
#ifdef CSM_BIN_TIMING
        B.tuning_counters = bin_debug_buffer();
#endif
        B.frame_shift = lp[0].pairs ? ((ny - 1) & 1) : 0;
        B.n_band = H;
        for (int h = 1; h <= H; ++h) {
            B.band_win[h - 1] = spec.stride[h];
            B.band_nx[h - 1] = nx / spec.stride[h];
            B.band_ny[h - 1] = ny / spec.stride[h];
        }

        ScoreJob base;
        std::memset(&base, 0, sizeof(base));
        base.rows = g.rows;
        base.cols = g.cols;
        base.pitch = g.pitch;
        base.sorted_pb = B.sorted_pb;
        base.tiles = B.tiles;
        base.n_tiles = B.n_tiles;
        base.n_theta = p.n_theta;
        base.n_points = p.n;
        base.max_tiles = p.max_tiles;
        base.x_lo = x_lo;
        base.y_lo = y_lo;
        base.flags = d_flags + k;
        base.min_known = min_known;
        base.joint = joint ? 1 : 0;
        base.rank_l = spec.bnb ? 1 : spec.unit;
        for (int h = 1; h <= H; ++h) {
            ScoreJob& S = sj[h][k];
            S = base;
            S.cells = g.levels[p.level[h]].cells;
            S.nx = nx / spec.stride[h];
            S.ny = ny / spec.stride[h];
            S.stride = spec.stride[h];
            S.log2_stride = ilog2_exact(spec.stride[h]);
            /* a leaf's own known count bounds every ancestor's from below when
             * no read can fall in the edge band: the level passes are only
             * needed to detect (and then handle) that case */
            S.skip_unless_band = spec.bnb ? 1 : (min_known <= 1);
            S.sorted_pb = B.sorted_rc;
            S.acc_s = d_lvl_s + p.lvl_off[h];
            S.acc_k = d_lvl_k + p.lvl_off[h];
            /* the level's atomic accumulators: cleared only when the pass will run */
            ZeroJob& Z0 = zj[(size_t)k * H + (h - 1)];
            Z0.a = S.acc_s;
            Z0.b = S.acc_k;
            Z0.words = (size_t)p.n_theta * S.nx * S.ny;
            Z0.flags = d_flags + k;
            Z0.always = S.skip_unless_band ? 0 : 1;
            Z0.pad = 0;
            zero_words_max = std::max(zero_words_max, Z0.words);
        }
        ScoreJob& F = sj[0][k];
        F = base;
        F.cells = g.levels[p.level[0]].cells;
        F.xg = g.xg;
        F.xg_pitch = g.xg_pitch;
        F.xg_pad = g.xg_pad;
        F.nx = nx;
        F.ny = ny;
        F.stride = 1;
        F.block_best = d_best + p.best_off;
        if (bound_pass) {
            F.xgf = g.xgf;
            F.approx_best = reinterpret_cast<float*>(ctx->b_abest.p) + p.best_off;
            /* |fp32 key - key| <= (n + 2) 2^-24 * key for a sum of n non-negative terms (one rounding
             * per fused multiply-add, one for each cell's float, one for joining the two accumulator
             * sets), n <= beams. A candidate that reaches the winner's exact key has an fp32 key of at
             * least max_fp32 * (1 - 3 (n + 3) 2^-24); the kernel compares with 4 (n + 3) 2^-24. */
            F.approx_slack = 4.0f * (float)(p.n + 3) * 5.9604645e-08f;
            F.bound_stats = reinterpret_cast<uint32_t*>(ctx->bound_stats.p);
            /* found <=> sum of probabilities / n > threshold, and that sum is (0.998 / 65534 / 499) * key
             * up to the f64 rounding of the beam-order summation (1e-12 relative): a candidate below
             * this key cannot be reported */
            const double thr = resident ? resident->windows[idx[k]].score_threshold : spec.score_thr;
            F.key_floor = thr > 0.0 ? (float)(thr * p.n * (65534.0 * 499.0 / 0.998) * (1.0 - 1e-9)) *
                                          (1.0f - F.approx_slack)
                                    : 0.0f;
            F.round1_record = resident ? (const void*)(resident->out_dev + idx[k]) : (const void*)(d_out + k);
            if (resident && resident->dump_f)
                F.dump_f = resident->dump_f[idx[k]];
        }
        if (resident && resident->dump_s)
            F.dump_s = resident->dump_s[idx[k]];
        if (resident && resident->dump_k)
            F.dump_k = resident->dump_k[idx[k]];
        /* branch and bound tests every popped node, leaf included; the
         * correlative sweep tests the coarse node only */
        F.check_own_known = spec.bnb || H == 0;
        F.elig_only_if_band = spec.bnb ? 1 : (min_known <= 1);
        F.n_elig = H;
        for (int h = 1; h <= H; ++h) {
            F.elig[h - 1].k = d_lvl_k + p.lvl_off[h];
            F.elig[h - 1].s = d_lvl_s + p.lvl_off[h];
            F.elig[h - 1].div = spec.stride[h];
            F.elig[h - 1].nxc = nx / spec.stride[h];
            F.elig[h - 1].nyc = ny / spec.stride[h];
        }

        FinalJob& Z = fj[k];
        std::memset(&Z, 0, sizeof(Z));
        Z.block_best = F.block_best;
        Z.n_entries = p.n_theta * ncb;
        Z.nx = nx;
        Z.ny = ny;
        Z.rank_l = spec.bnb ? 1 : spec.unit;
        Z.x_lo = x_lo;
        Z.y_lo = y_lo;
        Z.win_theta = p.win_t;
        /* scan_matcher_branch_bound.cpp:144-146 / scan_matcher_correlative.cpp:149-152 */
        Z.init_x = spec.bnb ? 0 : -p.win_x;
        Z.init_y = spec.bnb ? 0 : -p.win_y;
        Z.init_theta = spec.bnb ? 0 : -p.win_t;
        Z.cells = F.cells;
        Z.rows = g.rows;
        Z.cols = g.cols;
        Z.pitch = g.pitch;
        Z.hit_col = I.hit_col;
        Z.hit_row = I.hit_row;
        Z.n_points = p.n;
        Z.score_thr = resident ? resident->windows[idx[k]].score_threshold : spec.score_thr;
        Z.lut = ctx->lut_dev;
        Z.flags_in = d_flags + k;
        Z.out = resident ? resident->out_dev + idx[k] : d_out + k;
    }
    /* upload the job tables: one device buffer with 256-byte aligned sections, filled
     * from one pinned block by ONE copy (five copies from pageable vectors cost 18 us per
     * 64-window chain and a staging pass each) */
    const size_t tables_cap = jobs_bytes + (size_t)nq * 4 + 256 * (size_t)(H + 10);
    if ((rc = ensure(ctx, ctx->b_jobs, tables_cap))) return rc;
    std::shared_ptr<std::pair<void*, size_t>> pin_block;
    {
        std::pair<void*, size_t> blk(nullptr, 0);
        for (size_t b = 0; b < ctx->pin_free.size(); ++b)
            if (ctx->pin_free[b].second >= tables_cap) {
                blk = ctx->pin_free[b];
                ctx->pin_free.erase(ctx->pin_free.begin() + b);
                break;
            }
        if (!blk.first) {
            const size_t cap = tables_cap + tables_cap / 4 + 4096;
            if (hipHostMalloc(&blk.first, cap, hipHostMallocDefault) != hipSuccess)
                return fail(ctx, CSM_ENOMEM, "hipHostMalloc(%zu) failed", cap);
            blk.second = cap;
        }
        csm_ctx* owner = ctx;
        pin_block = std::shared_ptr<std::pair<void*, size_t>>(
            new std::pair<void*, size_t>(blk), [owner](std::pair<void*, size_t>* b) {
                owner->pin_free.push_back(*b);
                delete b;
            });
    }
    char* const jb0 = reinterpret_cast<char*>(ctx->b_jobs.p);
    char* const hb0 = reinterpret_cast<char*>(pin_block->first);
    size_t tables_off = 0;
    auto put = [&](const void* src, size_t bytes, char** dev) -> hipError_t {
        *dev = jb0 + tables_off;
        std::memcpy(hb0 + tables_off, src, bytes);
        tables_off += (bytes + 255) & ~(size_t)255;
        return tables_off <= tables_cap ? hipSuccess : hipErrorInvalidValue;
    };
    char *d_ij, *d_bj, *d_fj, *d_idx = nullptr, *d_zj = nullptr;
    if (H > 0)
        HIP_TRY(ctx, put(zj.data(), zj.size() * sizeof(ZeroJob), &d_zj));
    if (!resident)
        HIP_TRY(ctx, put(idx.data(), (size_t)nq * sizeof(int), &d_idx));
    std::vector<char*> d_sj(H + 1);
    HIP_TRY(ctx, put(ij.data(), nq * sizeof(ProjJob), &d_ij));
    HIP_TRY(ctx, put(bj.data(), nq * sizeof(BinJob), &d_bj));
    HIP_TRY(ctx, put(fj.data(), nq * sizeof(FinalJob), &d_fj));
    for (int h = 0; h <= H; ++h)
        HIP_TRY(ctx, put(sj[h].data(), nq * sizeof(ScoreJob), &d_sj[h]));
    HIP_TRY(ctx, hipMemcpyAsync(jb0, hb0, tables_off, hipMemcpyHostToDevice, ctx->stream));

    tick("jobs");
    /* ---- launches ---- */
    if (!resident) {
        ScopedTimer tm(ctx, "project");
        const int pb = ceil_div(n_points_max, kBlock);
        if (int e = csm_launch::project_batch(ctx->stream, dim3(pb, proj_theta_groups(n_theta_max, (long)pb * nq), nq),
                                              reinterpret_cast<const ProjJob*>(d_ij)))
            return launched_ok(ctx, e, "projection");
    }
    if (joint) {
        ScopedTimer tm(ctx, "bin");
        const int e = csm::launch_binj_batch(ctx->stream, ctx->device, reinterpret_cast<const BinJob*>(d_bj),
                                             (n_theta_max + 1) / 2, nq, binj_lds);
        if (e != 0)
            return fail(ctx, CSM_EIO, "joint binning launch failed: %s", hipGetErrorString((hipError_t)e));
    } else {
        ScopedTimer tm(ctx, "bin");
        if ((rc = launched_ok(ctx, csm_launch::bin_batch(ctx->stream, ctx->device, n_theta_max, nq, bin_lds,
                                                         reinterpret_cast<const BinJob*>(d_bj)), "binning")))
            return rc;
    }
    if (H > 0) {
        const int zb = (int)std::min<size_t>(64, (zero_words_max + 255) / 256);
        if ((rc = launched_ok(ctx, csm_launch::zero_if_band_batch(ctx->stream, std::max(1, zb), nq * H,
                                                                  reinterpret_cast<const ZeroJob*>(d_zj)), "edge-band clear")))
            return rc;
    }
    for (int h = H; h >= 1; --h) {
        /* keep >= ~2k workgroups in flight: split the tile list when the
         * level has few candidate blocks */
        const long blocks = (long)lp[h].ncb() * n_theta_max * nq;
        /* a level that only runs when a beam reaches the edge band (rare) is launched
         * unsplit: the launch that normally exits at once stays small */
        bool all_exit = true;
        for (int k = 0; k < nq; ++k)
            all_exit = all_exit && sj[h][k].skip_unless_band;
        const int n_slices = (blocks >= 2048 || all_exit)
                                 ? 1 : (int)std::min<long>(8, ceil_div(2048, (int)std::max<long>(1, blocks)));
        ScopedTimer tm(ctx, "score_coarse");
        if ((rc = launch_score_batch(ctx, reinterpret_cast<const ScoreJob*>(d_sj[h]), nq, lp[h],
                                     n_theta_max, n_slices, all_exit && nq >= 16 ? 4 : 0)))
            return rc;
    }
    if (bound_pass) {
        PassPlan fp = lp[0];
        fp.fp32 = true;
        ScopedTimer tm(ctx, "score_bound");
        if ((rc = launch_score_batch(ctx, reinterpret_cast<const ScoreJob*>(d_sj[0]), nq, fp, n_theta_max, 1)))
            return rc;
    }
    auto finalize = [&]() -> int {
        const size_t lds = (size_t)n_points_max * 8;
        ScopedTimer tm(ctx, "finalize");
        return launched_ok(ctx, csm_launch::finalize_batch(ctx->stream, ctx->device, nq, lds,
                                                           reinterpret_cast<const FinalJob*>(d_fj)), "finalize");
    };
    if (bound_pass) {
        size_t blocks_total = 0;
        for (int k = 0; k < nq; ++k)
            blocks_total += (size_t)((pp[k].n_theta + 1) / 2) * ncb;
        uint32_t* counts = reinterpret_cast<uint32_t*>(ctx->b_items.p);
        uint32_t* items0 = counts + 16;
        uint32_t* items1 = items0 + blocks_total;
        JointList list;
        list.items[0] = items0;
        list.items[1] = items1;
        list.counts = counts;
        list.blocks = (int)std::min<size_t>(blocks_total, 2048);
        const int split_cb = tail_split(ctx, lp[0]) ? lp[0].ncbx * (lp[0].ncby - 1) : ncb;
        for (int round = 1; round <= (two_rounds ? 2 : 1); ++round) {
            {
                ScopedTimer tm(ctx, "score_fine");
                HIP_TRY(ctx, hipMemsetAsync(counts, 0, 8, ctx->stream));
                const int e = csm::launch_bound_select(ctx->stream, reinterpret_cast<const ScoreJob*>(d_sj[0]), nq, ncb,
                                                       split_cb, items0, items1, counts, (uint32_t)blocks_total, round);
                if (e != 0)
                    return fail(ctx, CSM_EIO, "k_bound_select launch failed: %s", hipGetErrorString((hipError_t)e));
                if ((rc = launch_score_batch(ctx, reinterpret_cast<const ScoreJob*>(d_sj[0]), nq, lp[0], n_theta_max, 1,
                                             0, &list)))
                    return rc;
            }
            if ((rc = finalize()))
                return rc;
        }
    } else {
        {
            ScopedTimer tm(ctx, "score_fine");
            if ((rc = launch_score_batch(ctx, reinterpret_cast<const ScoreJob*>(d_sj[0]), nq, lp[0], n_theta_max, 1)))
                return rc;
        }
        if ((rc = finalize()))
            return rc;
    }
    if (resident) {
        /* asynchronous: the records stay on the device. The pinned block the job tables
         * are copied from goes back to the pool when this chain has run. */
        hipEvent_t done = nullptr;
        if (!ctx->event_pool.empty()) {
            done = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            HIP_TRY(ctx, hipEventCreate(&done));
        }
        HIP_TRY(ctx, hipEventRecord(done, ctx->stream));
        ctx->resident_hold.emplace_back(done, std::shared_ptr<void>(pin_block));
        return CSM_OK;
    }
    /* device copy of the records in query order (csm_copy_last_batch_records) */
    csm_result* rec_dev = reinterpret_cast<csm_result*>(ctx->rec_dev.p);
    if (rec_dev) {
        if ((rc = launched_ok(ctx, csm_launch::scatter_records(ctx->stream, d_out, reinterpret_cast<const int32_t*>(d_idx),
                                                               rec_dev, nq), "record scatter")))
            return rc;
    }
    tick("launch");
    std::vector<csm_result> res(nq);
    HIP_TRY(ctx, hipMemcpyAsync(res.data(), d_out, (size_t)nq * sizeof(csm_result),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    tick("gpu wait");

    for (int k = 0; k < nq; ++k) {
        const csm_loop_query& q = queries[idx[k]];
        csm_summary& o = out[idx[k]];
        if (res[k].flags & (CSM_FLAG_EDGE_BAND | CSM_FLAG_KEY_TIE | CSM_FLAG_PROJ_DELTA)) {
            if (spec.bnb) {
                if ((rc = bnb_literal(ctx, q, pp[k], o, spec.bnb_params, &res[k])))
                    return rc;
            } else {
                /* exact single-query path (host-verified projection, tie replay,
                 * literal sweep) */
                const uint32_t why = res[k].flags;
                csm_summary one;
                if ((rc = csm_correlative_match(ctx, q.map_id, &q.geometry, &q.scan, q.initial_pose,
                                                spec.csm_params, &one)))
                    return rc;
                res[k] = one.raw;
                res[k].flags |= why & CSM_FLAG_PROJ_DELTA;
            }
            if (rec_dev) {
                /* fixed up on the host: patch the device copy (the source stays alive in the ctx) */
                ctx->rec_patch.push_back(res[k]);
                HIP_TRY(ctx, hipMemcpyAsync(rec_dev + idx[k], &ctx->rec_patch.back(), sizeof(csm_result),
                                            hipMemcpyHostToDevice, ctx->stream));
            }
        }
        o.raw = res[k];
        o.pose_found = o.raw.found;
        /* scan_matcher_branch_bound.cpp:238-247 */
        o.best_sensor_pose[0] = o.sensor_pose[0] + o.step_x * o.raw.best_x;
        o.best_sensor_pose[1] = o.sensor_pose[1] + o.step_y * o.raw.best_y;
        o.best_sensor_pose[2] = o.sensor_pose[2] + o.step_theta * o.raw.best_theta;
        csm_host_move_backward(o.best_sensor_pose, q.scan.relative_sensor_pose, o.estimated_pose);
        o.candidates = (int64_t)pp[k].n_theta * nx * ny;
    }
    tick("finish");
    return CSM_OK;
}

} /* namespace csm_host */

extern "C" {


int csm_bnb_match_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n_queries,
                        const csm_bnb_params* prm, csm_summary* out)
{
    if (!ctx || !queries || n_queries < 1 || !prm || !out || prm->node_height_max < 0 ||
        prm->node_height_max >= kMaxElig)
        return fail(ctx, CSM_EINVAL, "csm_bnb_match_batch: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int H = prm->node_height_max;
    const bool host_timing = ctx->tune.host_timing;
    const auto tb0 = std::chrono::steady_clock::now();
    {
        int rc = begin_batch_records(ctx, n_queries);
        if (rc)
            return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (host_timing)
        fprintf(stderr, "[bnb batch] begin_records %8.3f ms\n", std::chrono::duration<double, std::milli>(t0 - tb0).count());
    /* pyramids: build and cache per map id, as mPrecompMaps does
     * (loop_detector_branch_bound.cpp:83-89) */
    std::vector<std::vector<int>> levels(n_queries, std::vector<int>(H + 1, 0));
    std::vector<PendingBox> pending_levels;
    std::vector<double> max_range(n_queries, 0.0);
    {
        const int i = scans_finite_max(queries, n_queries, max_range.data());
        if (i >= 0 && (!queries[i].scan.angles || !queries[i].scan.ranges || queries[i].scan.n_points < 1))
            return fail(ctx, CSM_EINVAL, "query %d: empty scan", i);
        if (i >= 0)
            return fail(ctx, CSM_EINVAL, "query %d: scan holds a non-finite range or angle", i);
    }
    for (int i = 0; i < n_queries; ++i) {
        DeviceGrid* g = find_grid(ctx, queries[i].map_id);
        if (!g)
            return fail(ctx, CSM_ENOENT, "query %d: map %llu not resident", i,
                        (unsigned long long)queries[i].map_id);
        for (int h = 0; h <= H; ++h) {
            int rc = level_for_window(ctx, *g, 1 << h, &levels[i][h], &pending_levels);
            if (rc)
                return rc;
        }
    }
    {
        /* all missing levels of all maps: one launch */
        int rc = launch_box_jobs(ctx, pending_levels);
        if (rc)
            return rc;
    }
    const auto t1 = std::chrono::steady_clock::now();

    if (host_timing)
        fprintf(stderr, "[bnb batch] levels        %8.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    /* group queries by leaf-window shape */
    std::memset(out, 0, sizeof(csm_summary) * (size_t)n_queries);
    std::map<std::pair<int, int>, std::vector<int>> groups;
    for (int i = 0; i < n_queries; ++i) {
        double sx, sy, st;
        search_step_from_max(queries[i].geometry.resolution, max_range[i], &sx, &sy, &st);
        const int big = 1 << H;
        const int nx = ceil_div(2 * csm_host_window(prm->range_x, sx) + 1, big) * big;
        const int ny = ceil_div(2 * csm_host_window(prm->range_y, sy) + 1, big) * big;
        groups[{ nx, ny }].push_back(i);
    }
    for (auto& kv : groups) {
        BatchSpec spec;
        spec.bnb = true;
        spec.max_range = max_range.data();
        spec.H = H;
        for (int h = 0; h <= H; ++h)
            spec.stride[h] = 1 << h;
        spec.unit = 1 << H;
        spec.range_x = prm->range_x;
        spec.range_y = prm->range_y;
        spec.range_theta = prm->range_theta;
        spec.score_thr = prm->score_threshold;
        spec.known_thr = prm->known_rate_threshold;
        spec.bnb_params = prm;
        if (host_timing)
            fprintf(stderr, "[bnb batch] grouping      %8.3f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        int rc = run_batch_group(ctx, queries, kv.second, levels, spec, out);
        if (rc)
            return rc;
    }
    const auto t2 = std::chrono::steady_clock::now();
    const double setup = std::chrono::duration<double, std::micro>(t1 - t0).count();
    const double opt = std::chrono::duration<double, std::micro>(t2 - t1).count();
    for (int i = 0; i < n_queries; ++i) {
        out[i].input_setup_us = setup / n_queries;
        out[i].optimization_us = opt / n_queries;
    }
    return CSM_OK;
}

/* LoopDetectorCorrelative::Detect's search part for a batch of queries
 * (src/mapping/loop_detector_correlative.cpp:59-156 lines 68-108): one coarse
 * map per local map id, cached on the device like mPrecompMaps. */
int csm_correlative_match_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n_queries,
                                const csm_correlative_params* prm, csm_summary* out)
{
    if (!ctx || !queries || n_queries < 1 || !prm || !out || prm->low_resolution < 1)
        return fail(ctx, CSM_EINVAL, "csm_correlative_match_batch: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int L = prm->low_resolution;
    const int H = L > 1 ? 1 : 0;
    {
        int rc = begin_batch_records(ctx, n_queries);
        if (rc)
            return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<PendingBox> pending_levels;
    std::vector<std::vector<int>> levels(n_queries, std::vector<int>(H + 1, 0));
    std::vector<double> max_range(n_queries, 0.0);
    {
        const int i = scans_finite_max(queries, n_queries, max_range.data());
        if (i >= 0 && (!queries[i].scan.angles || !queries[i].scan.ranges || queries[i].scan.n_points < 1))
            return fail(ctx, CSM_EINVAL, "query %d: empty scan", i);
        if (i >= 0)
            return fail(ctx, CSM_EINVAL, "query %d: scan holds a non-finite range or angle", i);
    }
    for (int i = 0; i < n_queries; ++i) {
        DeviceGrid* g = find_grid(ctx, queries[i].map_id);
        if (!g)
            return fail(ctx, CSM_ENOENT, "query %d: map %llu not resident", i,
                        (unsigned long long)queries[i].map_id);
        if (H) {
            int rc = level_for_window(ctx, *g, L, &levels[i][1], &pending_levels);
            if (rc)
                return rc;
        }
    }
    {
        int rc = launch_box_jobs(ctx, pending_levels);
        if (rc)
            return rc;
    }
    const auto t1 = std::chrono::steady_clock::now();
    std::memset(out, 0, sizeof(csm_summary) * (size_t)n_queries);
    std::map<std::pair<int, int>, std::vector<int>> groups;
    for (int i = 0; i < n_queries; ++i) {
        double sx, sy, st;
        search_step_from_max(queries[i].geometry.resolution, max_range[i], &sx, &sy, &st);
        const int nx = ceil_div(2 * csm_host_window(prm->range_x, sx) + 1, L) * L;
        const int ny = ceil_div(2 * csm_host_window(prm->range_y, sy) + 1, L) * L;
        groups[{ nx, ny }].push_back(i);
    }
    for (auto& kv : groups) {
        BatchSpec spec;
        spec.bnb = false;
        spec.max_range = max_range.data();
        spec.H = H;
        spec.stride[0] = 1;
        spec.stride[1] = L;
        spec.unit = L;
        spec.range_x = prm->range_x;
        spec.range_y = prm->range_y;
        spec.range_theta = prm->range_theta;
        spec.score_thr = prm->score_threshold;
        spec.known_thr = prm->known_rate_threshold;
        spec.csm_params = prm;
        int rc = run_batch_group(ctx, queries, kv.second, levels, spec, out);
        if (rc)
            return rc;
    }
    const auto t2 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_queries; ++i) {
        out[i].input_setup_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / n_queries;
        out[i].optimization_us = std::chrono::duration<double, std::micro>(t2 - t1).count() / n_queries;
    }
    return CSM_OK;
}

/* csm_score_window_dev for many windows at once: one launch chain (k_bin_batch,
 * the batched scoring kernels, k_finalize_batch) over all of them. */
int csm_score_windows_dev(csm_ctx* ctx, int32_t n, const uint64_t* map_ids, const csm_window* windows,
                          const int32_t* const* hit_col_dev, const int32_t* const* hit_row_dev,
                          csm_result* out_dev)
{
    return csm_score_windows_dump_dev(ctx, n, map_ids, windows, hit_col_dev, hit_row_dev, out_dev, nullptr,
                                      nullptr, nullptr);
}

int csm_score_windows_dump_dev(csm_ctx* ctx, int32_t n, const uint64_t* map_ids, const csm_window* windows,
                               const int32_t* const* hit_col_dev, const int32_t* const* hit_row_dev,
                               csm_result* out_dev, uint32_t* const* dump_s_dev, uint16_t* const* dump_k_dev,
                               float* const* dump_f_dev)
{
    if (!ctx || n < 1 || !map_ids || !windows || !hit_col_dev || !hit_row_dev || !out_dev)
        return fail(ctx, CSM_EINVAL, "csm_score_windows_dev: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<csm_loop_query> queries((size_t)n);
    std::vector<std::vector<int>> levels((size_t)n, std::vector<int>(2, 0));
    /* windows that share the candidate domain, the coarse window and the merge mode go together */
    std::map<std::array<int, 4>, std::vector<int>> groups;
    for (int i = 0; i < n; ++i) {
        const csm_window& w = windows[i];
        if (w.n_theta < 1 || (w.n_theta & 1) == 0 || w.n_points < 1 || w.win_x < 0 || w.win_y < 0 ||
            w.low_resolution < 1 || !hit_col_dev[i] || !hit_row_dev[i])
            return fail(ctx, CSM_EINVAL, "window %d: bad window", i);
        DeviceGrid* g = find_grid(ctx, map_ids[i]);
        if (!g)
            return fail(ctx, CSM_ENOENT, "window %d: map %llu not resident", i,
                        (unsigned long long)map_ids[i]);
        const int L = w.low_resolution;
        if (L > 1) {
            if (w.coarse_level < 0 || w.coarse_level >= (int)g->levels.size() ||
                g->levels[w.coarse_level].stale || g->levels[w.coarse_level].win != L)
                return fail(ctx, CSM_ENOENT, "window %d: level %d does not hold box-max(%d)", i,
                            w.coarse_level, L);
            levels[i][1] = w.coarse_level;
        }
        std::memset(&queries[i], 0, sizeof(csm_loop_query));
        queries[i].map_id = map_ids[i];
        const int nx = ceil_div(2 * w.win_x + 1, L) * L, ny = ceil_div(2 * w.win_y + 1, L) * L;
        groups[{ nx, ny, L, w.merge_mode }].push_back(i);
    }
    ResidentBatch resident { windows, hit_col_dev, hit_row_dev, out_dev, dump_s_dev, dump_k_dev, dump_f_dev };
    /* drop the tables of earlier calls whose launch chains have completed */
    while (!ctx->resident_hold.empty()) {
        const bool full = ctx->resident_hold.size() >= 256;
        hipEvent_t ev = ctx->resident_hold.front().first;
        if (full)
            HIP_TRY(ctx, hipEventSynchronize(ev));
        else if (hipEventQuery(ev) != hipSuccess)
            break;
        ctx->event_pool.push_back(ev);
        ctx->resident_hold.erase(ctx->resident_hold.begin());
    }
    for (auto& kv : groups) {
        const int L = kv.first[2];
        BatchSpec spec;
        spec.bnb = false;
        spec.H = L > 1 ? 1 : 0;
        spec.stride[0] = 1;
        spec.stride[1] = L;
        spec.unit = L;
        int rc = run_batch_group(ctx, queries.data(), kv.second, levels, spec, nullptr, &resident);
        if (rc)
            return rc;
    }
    return CSM_OK;
}

int csm_copy_last_batch_records(csm_ctx* ctx, csm_result* dst_dev)
{
    if (!ctx || !dst_dev)
        return fail(ctx, CSM_EINVAL, "csm_copy_last_batch_records: bad arguments");
    if (ctx->rec_n < 1 || !ctx->rec_dev.p)
        return fail(ctx, CSM_ENOENT, "no batch has been scored on this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst_dev, ctx->rec_dev.p, (size_t)ctx->rec_n * sizeof(csm_result),
                                hipMemcpyDeviceToDevice, ctx->stream));
    return CSM_OK;
}

int csm_bound_pass_stats(csm_ctx* ctx, uint64_t* blocks_scored, uint64_t* blocks_skipped)
{
    if (!ctx)
        return CSM_EINVAL;
    uint32_t h[2] = { 0, 0 };
    if (ctx->bound_stats.p) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipMemcpyAsync(h, ctx->bound_stats.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->bound_stats.p, 0, 8, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (blocks_scored) *blocks_scored = h[0];
    if (blocks_skipped) *blocks_skipped = h[1];
    return CSM_OK;
}

} /* extern "C" */

#ifdef CSM_BIN_TIMING
/* tuning builds only (tools/build_variant.sh NAME -DCSM_BIN_TIMING): reads and clears k_bin's phase counters */
extern "C" int csm_debug_bin_cycles(unsigned long long* out16)
{
    /* out16[0..5]: cycles per phase (8..11: parts of pass A) summed over the workgroups of the LAST launch
     * pattern (rows are overwritten by every launch), out16[15]: workgroups */
    if (!g_bin_debug)
        return CSM_ENOENT;
    std::vector<unsigned long long> rows((size_t)kBinDebugRows * 16);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(rows.data(), g_bin_debug, rows.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return CSM_EIO;
    for (int k = 0; k < 16; ++k)
        out16[k] = 0;
    for (int r = 0; r < kBinDebugRows; ++r) {
        if (!rows[(size_t)r * 16 + 7])
            continue;
        for (int k = 0; k < 12; ++k)
            if (k != 7)
                out16[k] += rows[(size_t)r * 16 + k];
        out16[15] += 1;
    }
    return CSM_OK;
}
#endif

