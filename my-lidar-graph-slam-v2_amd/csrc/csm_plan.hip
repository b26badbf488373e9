/* csm_plan.hip -- the matchers' launch planning and level bookkeeping (host code only): bit-exact host
 * restatements (search step, probability table), the candidate-block planner of the scoring passes
 * (plan_pass, plan_pass_pairs, make_plan), the lane table, the launch helpers on top of csm_launch.hpp /
 * csm_joint.hpp, box-maximum levels and the pair-row copies of a grid. Declared in csm_matchers.hpp. */
#include "csm_matchers.hpp"

namespace csm_host {


/* ---- host restatements (bit-exact pieces) ---- */

/* inc/grid_map_new/grid_values.hpp:26-35 with ValueMin 1, ValueMax 65535,
 * ProbabilityMin 1e-3, ProbabilityMax 1-1e-3
 * (inc/grid_map_new/grid_binary_bayes.hpp:163-176). The reference table stops
 * at 65534 (src/grid_map_new/grid_values.cpp:32-33); 65535 follows the same
 * formula here. */
double value_to_probability(unsigned v)
{
    const double pmin = 1e-3;
    const double pmax = 1.0 - 1e-3;
    if (v == 0)
        return 0.0;
    return pmin + (pmax - pmin) * static_cast<double>(static_cast<int>(v) - 1) /
                      static_cast<double>(65535 - 1);
}


/* k_project's grid.y: workgroups per (beam block, job) that share the theta slices --
 * few (each pays two library calls per beam once), but enough workgroups to fill
 * the chip and at most kProjSlices slices each */
int proj_theta_groups(int n_theta, long blocks_xz)
{
    long g = std::min<long>(n_theta, std::max<long>(1, (1024 + blocks_xz - 1) / std::max<long>(1, blocks_xz)));
    g = std::max<long>(g, ceil_div(n_theta, kProjSlices));
    return (int)g;
}


/* The same check and the scan's largest range in one pass over the beams (the
 * batch entries need both for every query; x * 0 is NaN exactly when x is not
 * finite, which keeps the loop free of branches). */
bool scan_finite_max(const csm_scan* scan, double* max_range)
{
    double poison = 0.0, mx = scan->ranges[0];
    for (int i = 0; i < scan->n_points; ++i) {
        const double r = scan->ranges[i];
        poison += r * 0.0 + scan->angles[i] * 0.0;
        mx = r > mx ? r : mx;
    }
    *max_range = mx;
    return poison == 0.0 && std::isfinite(scan->relative_sensor_pose[0]) &&
           std::isfinite(scan->relative_sensor_pose[1]) && std::isfinite(scan->relative_sensor_pose[2]);
}

void search_step_from_max(double resolution, double max_range, double* step_x, double* step_y,
                          double* step_theta)
{
    const double theta = resolution / max_range;
    *step_x = resolution;
    *step_y = resolution;
    *step_theta = std::acos(1.0 - 0.5 * theta * theta);
}

/* scan_finite_max of every query; returns the first offending query or -1 */
int scans_finite_max(const csm_loop_query* queries, int n_queries, double* max_range)
{
    /* A loop detection matches ONE scan against many submaps: a query whose scan shares its arrays and its
     * sensor pose with the query before it reuses that query's result (256 x 1080 doubles walked once
     * instead of 256 times, and no worker threads: 0.1 ms of a 1.85 ms batch). */
    auto same_scan = [&](int a, int b) {
        const csm_scan& x = queries[a].scan;
        const csm_scan& y = queries[b].scan;
        return x.angles == y.angles && x.ranges == y.ranges && x.n_points == y.n_points &&
               std::memcmp(x.relative_sensor_pose, y.relative_sensor_pose, sizeof(x.relative_sensor_pose)) == 0;
    };
    int distinct = 1;
    for (int i = 1; i < n_queries; ++i)
        distinct += same_scan(i, i - 1) ? 0 : 1;
    std::atomic<int> bad(n_queries);
    auto range = [&](int lo, int hi) {
        bool prev_ok = false;
        for (int i = lo; i < hi; ++i) {
            const csm_scan& sc = queries[i].scan;
            bool ok;
            if (i > lo && same_scan(i, i - 1)) {
                ok = prev_ok;
                max_range[i] = max_range[i - 1];
            } else {
                ok = sc.angles && sc.ranges && sc.n_points >= 1 && scan_finite_max(&sc, &max_range[i]);
            }
            prev_ok = ok;
            if (!ok) {
                int cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {
                }
                return;
            }
        }
    };
    if (distinct * 8 <= n_queries)
        range(0, n_queries);
    else
        host_parallel_for(n_queries, 128, range);
    return bad.load() < n_queries ? bad.load() : -1;
}

/* Do enough beams share cells for merging to pay? A merged entry costs a
 * multiply per gather (~3x the vector work of the plain path) and saves LDS
 * reads in proportion to the duplicates: break-even near 1.4 beams per cell
 * (measured: config 2, 1.9 beams per cell, 108 -> 93 us; config 5, 1.15, 91 ->
 * 98 ms). Estimated from the scan alone: a beam of range r next to a neighbour
 * d_theta away opens a new cell with probability ~ min(1, r * d_theta / res). */
bool merging_pays(const double* angles, const double* ranges, int n, double res)
{
    if (n < 2)
        return false;
    double cells = 1.0;
    for (int i = 1; i < n; ++i) {
        const double arc = std::fabs(angles[i] - angles[i - 1]) * 0.5 * (ranges[i] + ranges[i - 1]);
        cells += std::min(1.0, arc / res);
    }
    return n >= 1.4 * cells;
}

/* k_bin's hash table: load factor <= 2/3 when every beam lands on a cell of its own */
int bin_hash_size(int n_points)
{
    int h = 1024;
    while (2 * h < 3 * n_points && h < 16384)      /* kMaxPoints = 10240 -> 16384 */
        h <<= 1;
    return h;
}

/* k_bin's LDS: three 64-bit words per tile, the hash table
 * (keys, beam counts) and the list of occupied slots (16 bits each, a segment per wave) */
size_t bin_lds_bytes(int tiles, int n_points)
{
    return ((size_t)6 * ((tiles + 1) & ~1) + 2 * (size_t)bin_hash_size(n_points)) * 4 + 2 * (size_t)n_points + 16;
}






int ilog2_exact(int v)
{
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return (1 << l) == v ? l : -1;
}

/* Pick the candidate block (cbx wide, groups * R tall) for nx x ny candidates
 * `stride` cells apart. Fails (returns false) if nothing fits the LDS limits. */
bool plan_pass(const Tuning& tune, int nx, int ny, int stride, PassPlan* out)
{
    PassPlan p;
    p.nx = nx;
    p.ny = ny;
    p.stride = stride;
    p.log2s = ilog2_exact(stride);
    const bool strided = stride > 1;
    const int max_ls = 192;
    /* columns: 7 (alignment) + tile + (cbx - 1) * stride + 1 <= lstride */
    const int max_cbx = std::min(120, (max_ls - kTile - 8) / stride + 1);
    if (max_cbx < 1)
        return false;
    const int nb = ceil_div(nx, max_cbx);
    p.cbx = ceil_div(nx, nb);
    p.ncbx = ceil_div(nx, p.cbx);
    const int need = kTile + 8 + (p.cbx - 1) * stride;
    const int cand_ls[] = { 96, 128, 160, 192 };
    for (int ls : cand_ls) {
        /* phase-major layout: column phase p owns floor(ls / stride) cells */
        const int need8 = (need + 7) & ~7;
        if ((ls / stride) * stride < need8 || (strided && ls != 128 && ls != 192))
            continue;
        p.lstride = ls;
        break;
    }
    if (!p.lstride)
        return false;
    /* rows: stride-1 regions hold kTile + cby - 1 rows, strided ones
     * (ceil(kTile / s) + cby - 1) * s */
    const int max_cby = strided ? kMaxRegionRowsStrided / stride - (kTile + stride - 1) / stride + 1
                                : kMaxRegionRows - kTile + 1;
    if (max_cby < 1)
        return false;
    int g = std::max(1, kBlock / p.cbx);
    static const int r_fine[] = { 4, 5, 6, 7, 8 };
    static const int r_strided[] = { 1, 2, 4 };
    const int* rs = strided ? r_strided : r_fine;
    const int nrs = strided ? 3 : 5;
    long best_cost = -1;
    for (int k = 0; k < nrs; ++k) {
        const int r = rs[k];
        if (tune.force_r && !strided && tune.force_r != r)
            continue;
        if (r > max_cby)
            continue;
        int gg = std::min(g, ceil_div(ny, r));
        gg = std::max(1, std::min(gg, max_cby / r));
        const int nby = ceil_div(ny, gg * r);
        /* per (tile, block): ~108 r instruction slots of gathering (~90 beams)
         * + staging that grows with the rows the block spans; times the
         * number of blocks along y. Calibrated on config 2 (R 7 < 4 < 8). */
        const long cost = (long)nby * (1080L * r + 1000L + 25L * gg * r * stride) +
                          (long)(kBlock - gg * p.cbx);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && r > p.R)) {
            best_cost = cost;
            p.R = r;
            p.groups = gg;
            p.ncby = nby;
        }
    }
    if (best_cost < 0)
        return false;
    *out = p;
    return true;
}

/* The pair-row fine kernel (k_score_pairs<LS, 8, W>): LS = slots per pair row of
 * the LDS region = alignment column + 64-cell tile + cbx - 1 candidates, even
 * (16-byte rows for the LDS-DMA pieces). Instantiated for these LS; a candidate
 * block may be any width cbx <= LS - 65 (124: the conflict-free pitch of R = 6, cbx = 52,
 * the branch-and-bound detector's default window; 156: that of R = 6, cbx = 84, the 36-row
 * tail block of the frontend window). */
const int kPairLS[] = { 86, 98, 118, 124, 130, 150, 156, 162, 182 };

size_t pair_lds_bytes(int ls, int cby, int lists)
{
    const size_t region = (size_t)((kTile + cby) / 2 + 1) * ls * 8;
    return ((region + 1023) / 1024) * 1024 + (size_t)lists * kPbMax * 4;
}

/* two_slices: plan for the batch kernel that takes two theta slices per workgroup
 * (a second entry list in LDS) */
bool plan_pass_pairs(const Tuning& tune, int nx, int ny, PassPlan* out, bool two_slices, int list_lds)
{
    const int lists = two_slices ? 2 : 1;
    if (!tune.two_slices && two_slices)
        return plan_pass_pairs(tune, nx, ny, out, false);
    /* LDS of a launch: the window copy + its entry lists (kPbMax words per list of the pair kernels; list_lds
     * bytes where the caller knows better: the joint kernels keep kJRec words) */
    auto lds_of = [&](int ls, int cby) {
        return list_lds >= 0 ? pair_lds_bytes(ls, cby, 0) + (size_t)list_lds : pair_lds_bytes(ls, cby, lists);
    };
    double best = -1.0;
    const int max_cbx = kPairLS[sizeof(kPairLS) / sizeof(kPairLS[0]) - 1] - 65;
    /* R = candidate rows per lane: 8, or 6 where that covers the rows with fewer
     * multiply-adds per wave (52 rows: 9 groups x 6 instead of 7 x 8) */
    const int force_r = tune.pair_r, force_ncbx = tune.pair_ncbx, force_g = tune.pair_groups;   /* tuning builds */
    for (int R : { 8, 6 })
    for (int ncbx = ceil_div(nx, max_cbx); ncbx <= ceil_div(nx, max_cbx) + 2; ++ncbx) {
        if ((force_r && R != force_r) || (force_ncbx && ncbx != force_ncbx))
            continue;
        PassPlan p;
        p.nx = nx;
        p.ny = ny;
        p.stride = 1;
        p.log2s = 0;
        p.pairs = true;
        p.lists = lists;
        p.list_lds = list_lds;
        p.R = R;
        p.ncbx = ncbx;
        p.cbx = ceil_div(nx, ncbx);
        p.lstride = 0;
        for (int ls : kPairLS)
            if (ls >= p.cbx + 65) {
                p.lstride = ls;
                break;
            }
        if (!p.lstride)
            continue;
        /* A half-wave that holds the end of one lane group and the start of the next
         * reads without a bank conflict when the next group's slots continue the bank
         * sequence: (R / 2) * LS = cbx (mod 32). Take such a pitch if one is instantiated
         * within 8 slots of the smallest (0.9 % of the branch-and-bound leaf kernel; no
         * even LS does it for R = 8, cbx = 84). */
        for (int ls : kPairLS)
            if (ls >= p.lstride && ls <= p.lstride + 8 && ((R / 2) * ls - p.cbx) % 32 == 0) {
                p.lstride = ls;
                break;
            }
        /* R = 8 needs an odd pitch for that (no such LS) and takes the lane table instead; but where
         * the window's last row block becomes an R = 6 launch (tail_split) BOTH launches read the
         * pitch the entries were binned for: take the R = 6 launch's conflict-free pitch if close */
        const int tail_ls_from = p.lstride;
        if (R == 8 && tune.pair_tail)
            for (int ls : kPairLS)
                if (ls >= tail_ls_from && ls <= tail_ls_from + 8 && (3 * ls - p.cbx) % 32 == 0) {
                    p.lstride = ls;
                    break;
                }
        int g = std::min(std::min(kBlock / p.cbx, ceil_div(ny, R)), kPairMaxCby / R);
        /* two workgroups per CU: at most 80 KB of LDS each INCLUDING the kernel's static
         * __shared__ arrays (a plan at exactly 80 KB of dynamic LDS ran one workgroup per
         * CU: configs[4] took 87 ms instead of 56) */
        while (g > 1 && lds_of(p.lstride, g * R) > 80 * 1024 - 1024)
            --g;
        if (force_g && force_g <= g)
            g = force_g;
        if (g < 1 || lds_of(p.lstride, g * R) > 160 * 1024 - 256)
            continue;
        p.ncby = ceil_div(ny, g * R);
        if (!force_g)
            g = ceil_div(ceil_div(ny, p.ncby), R);      /* balance the row blocks */
        p.groups = g;
        /* per (block, tile): the window copy grows with the region; the gather costs
         * every wave R multiply-adds + ~6 other instructions per entry, however many
         * of its lanes are useful; ~450 cycles of barriers and waits */
        const double cost = (double)p.ncbx * p.ncby *
                            (0.01 * ((kTile + g * R) / 2 + 1) * p.lstride + 46.0 * (R + 6) + 450.0);
        if (tune.plan_debug)
            fprintf(stderr, "[plan %dx%d] R %d ncbx %d cbx %d LS %d groups %d ncby %d lds %zu cost %.0f\n", nx, ny, R,
                    ncbx, p.cbx, p.lstride, g, p.ncby, lds_of(p.lstride, g * R), cost);
        if (best < 0 || cost < best) {
            best = cost;
            *out = p;
        }
    }
    if (best < 0)
        return false;
    if (tune.pair_ls) {                                 /* tuning builds: force the row pitch */
        bool have = false;
        for (int ls : kPairLS)
            have = have || ls == tune.pair_ls;
        if (have && tune.pair_ls >= out->cbx + 65 &&
            lds_of(tune.pair_ls, out->groups * out->R) <= 160 * 1024 - 256)
            out->lstride = tune.pair_ls;
    }
    return true;
}

/* padding (cells, every side) the pair-row copy of a grid needs for a window of nx x ny candidates */
int xgrid_pad_for(int nx, int ny)
{
    return (std::max(nx, ny) + kTile + kPairMaxCby + 8 + 31) & ~31;
}

/* Two LDS buffers (one barrier per tile, staging overlapped with the gather)
 * when they fit (forced in tuning builds). */
int pick_buffers(const Tuning& tune, size_t lds_one, long blocks)
{
    (void)blocks;
    if (tune.nbuf)
        return tune.nbuf == 2 && 2 * lds_one <= 160 * 1024 - 256 ? 2 : 1;
    /* measured (512-thread workgroups): no gain on config 2, and the halved
     * occupancy costs 25-35 % on configs 3 and 5 */
    return 1;
}

size_t pass_lds_bytes(const PassPlan& p)
{
    if (p.pairs)
        return p.list_lds >= 0 ? pair_lds_bytes(p.lstride, p.groups * p.R, 0) + (size_t)p.list_lds
                               : pair_lds_bytes(p.lstride, p.groups * p.R, p.lists);
    const int cby = p.groups * p.R;
    const int rows = p.stride > 1 ? ((kTile + p.stride - 1) / p.stride + cby - 1) * p.stride
                                  : kTile + cby - 1;
    return (size_t)rows * p.lstride * 4 + kPbMax * 4;
}

int make_plan(csm_ctx* ctx, const DeviceGrid& g, const csm_window* w, Plan* p)
{
    if (w->n_theta < 1 || w->n_points < 1 || w->win_x < 0 || w->win_y < 0 ||
        w->low_resolution < 1)
        return fail(ctx, CSM_EINVAL, "bad window");
    p->n_theta = w->n_theta;
    p->n = w->n_points;
    p->win_x = w->win_x;
    p->win_y = w->win_y;
    p->L = w->low_resolution;
    p->nxc = ceil_div(2 * w->win_x + 1, p->L);
    p->nyc = ceil_div(2 * w->win_y + 1, p->L);
    p->nx = p->nxc * p->L;
    p->ny = p->nyc * p->L;
    p->x_lo = -w->win_x;
    p->y_lo = -w->win_y;
    p->x_hi = p->x_lo + p->nx - 1;
    p->y_hi = p->y_lo + p->ny - 1;
    if (!plan_pass_pairs(ctx->tune, p->nx, p->ny, &p->fine))
        return fail(ctx, CSM_EINVAL, "internal: no launch geometry for the fine level");
    p->fine.weighted = w->merge_mode == 0;
    if (p->L > 1 && !plan_pass(ctx->tune, p->nxc, p->nyc, p->L, &p->coarse))
        return fail(ctx, CSM_EINVAL, "LowResolution %d too large for the coarse kernel", p->L);
    p->tiles_x = ceil_div(g.cols - p->x_lo + p->x_hi, kTile);
    p->tiles_y = ceil_div(g.rows - p->y_lo + p->y_hi + 1, kTile);    /* + 1: k_bin's frame shift */
    p->max_tiles = std::min(p->n, p->tiles_x * p->tiles_y) + p->n / kPbMax + 1;
    if (p->n > kMaxPoints)
        return fail(ctx, CSM_EINVAL, "more than %d beams per scan", kMaxPoints);
    const size_t bin_lds = bin_lds_bytes(p->tiles_x * p->tiles_y, p->n);
    if (bin_lds > 160 * 1024 - 64)
        return fail(ctx, CSM_EINVAL, "grid + window too large for the binning kernel (its per-tile words, hash table and cell list exceed the LDS)");
    return CSM_OK;
}

/* What the wrappers of csm_launch.hip return: a HIP error code, or -1 for "no kernel instantiated". */
int launched_ok(csm_ctx* ctx, int e, const char* what)
{
    if (e < 0)
        return fail(ctx, CSM_EINVAL, "internal: no %s kernel for this launch shape", what);
    if (e != 0)
        return fail(ctx, CSM_EIO, "%s kernel launch failed: %s", what, hipGetErrorString((hipError_t)e));
    return CSM_OK;
}

/* the fields of a csm_launch::ScoreLaunch a pass plan decides */
csm_launch::ScoreLaunch score_launch(const csm_ctx* ctx, const PassPlan& pp, dim3 grid, size_t lds)
{
    csm_launch::ScoreLaunch a;
    a.stream = ctx->stream;
    a.device = ctx->device;
    a.lstride = pp.lstride;
    a.R = pp.R;
    a.mode = pp.stride == 1 ? 0 : pp.log2s >= 0 ? 1 : 2;
    a.weighted = pp.weighted;
    a.lists = pp.lists;
    a.cbx = pp.cbx;
    a.groups = pp.groups;
    a.grid = grid;
    a.lds = lds;
    a.ncb = pp.ncb();
    return a;
}

/* Which candidate (lane group g, column dxi) a thread of a pair kernel owns. A ds_read_b64
 * serves a half-wave in one pass when its 32 slots cover the 64 banks once; slot (g, dxi) of an
 * entry sits at bank pair (dxi + (R / 2) * LS * g) mod 32. With threads numbered through the
 * groups in order (dxi = tid % cbx) every half-wave that holds the end of one group and the
 * start of the next takes two passes (6 of 16 for cbx = 84: a quarter of the LDS cycles of a
 * kernel the LDS read rate bounds). The table instead gives each group whole half-waves for its
 * first 32 * floor(cbx / 32) columns and deals the remaining columns of all groups to the
 * remaining half-waves so that a half-wave holds each bank pair once; what cannot be placed
 * that way is collected in the last half-waves (cbx = 84, 6 groups, LS = 150: 17 passes per
 * wave-round of reads instead of 21). Entry = idle << 15 | g << 8 | dxi; 0xffff = idle lane without
 * a slot of its own. Returns null
 * (threads in order) where the table would not save a pass. */
int lane_map_for(csm_ctx* ctx, const PassPlan& pp, const uint16_t** out)
{
    *out = nullptr;
    if (!ctx->tune.lane_map)
        return CSM_OK;
    const std::array<int, 4> key = { pp.cbx, pp.groups, pp.R, pp.lstride };
    auto it = ctx->lane_maps.find(key);
    if (it != ctx->lane_maps.end()) {
        *out = it->second;
        return CSM_OK;
    }
    uint16_t*& slot = ctx->lane_maps[key];
    slot = nullptr;
    const int nhw = kBlock / 32, nfull = pp.cbx / 32;
    auto pos = [&](int g, int c) { return (c + (pp.R / 2) * pp.lstride * g) % 32; };
    auto passes_of = [&](const std::vector<uint16_t>& t) {
        int total = 0;
        for (int h = 0; h < nhw; ++h) {
            int cnt[32] = { 0 }, worst = 0;
            for (int l = 0; l < 32; ++l)
                if (t[h * 32 + l] != 0xffff)
                    worst = std::max(worst, ++cnt[pos((t[h * 32 + l] >> 8) & 127, t[h * 32 + l] & 255)]);
            total += worst;
        }
        return total;
    };
    std::vector<uint16_t> linear(kBlock, 0xffff), table(kBlock, 0xffff);
    for (int tid = 0; tid < kBlock; ++tid)
        if (tid / pp.cbx < pp.groups)
            linear[tid] = (uint16_t)((tid / pp.cbx) << 8 | (tid % pp.cbx));
    if (pp.cbx > 255 || pp.groups > 127 || pp.groups * nfull >= nhw)
        return CSM_OK;
    int hw = 0;
    for (int g = 0; g < pp.groups; ++g)
        for (int k = 0; k < nfull; ++k, ++hw)
            for (int l = 0; l < 32; ++l)
                table[hw * 32 + l] = (uint16_t)(g << 8 | (32 * k + l));
    const int nrem = nhw - hw;
    std::vector<std::vector<uint16_t>> lists(nrem);
    std::vector<uint16_t> extra;
    int seen[32] = { 0 };
    for (int g = 0; g < pp.groups; ++g)
        for (int c = 32 * nfull; c < pp.cbx; ++c) {
            const int i = seen[pos(g, c)]++;
            const uint16_t v = (uint16_t)(g << 8 | c);
            if (i < nrem && lists[i].size() < 32)
                lists[i].push_back(v);
            else
                extra.push_back(v);
        }
    for (uint16_t v : extra) {
        int h = nrem - 1;
        while (h >= 0 && lists[h].size() >= 32)
            --h;
        if (h < 0)
            return CSM_OK;
        lists[h].push_back(v);
    }
    for (int h = 0; h < nrem; ++h) {
        bool used[32] = { false };
        for (size_t l = 0; l < lists[h].size(); ++l) {
            table[(hw + h) * 32 + l] = lists[h][l];
            used[pos(lists[h][l] >> 8, lists[h][l] & 255)] = true;
        }
        /* idle lanes read too (the instruction is the wave's): each gets a slot of its own on a
         * bank pair the half-wave does not use (group 0, column q), marked idle by bit 15 */
        int q = 0;
        for (size_t l = lists[h].size(); l < 32; ++l) {
            while (q < 32 && (used[q] || q >= pp.cbx))
                ++q;
            if (q < 32) {
                table[(hw + h) * 32 + l] = (uint16_t)(0x8000 | q);
                used[q] = true;
            }
        }
    }
    if (passes_of(table) >= passes_of(linear))
        return CSM_OK;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot), kBlock * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMemcpy(slot, table.data(), kBlock * sizeof(uint16_t), hipMemcpyHostToDevice));
    *out = slot;
    return CSM_OK;
}

int launch_score(csm_ctx* ctx, const ScoreJob& job, const PassPlan& pp, int n_theta, int n_slices)
{
    const dim3 grid(pp.ncb(), n_theta, n_slices);
    if (pp.pairs) {
        /* a launch far larger than the chip, not tile-split: slices fastest (see k_score_pairs) */
        int theta_major = (n_slices == 1 && (long)pp.ncb() * n_theta >= 4096 && pp.ncb() <= 65535) ? 1 : 0;
        if (ctx->tune.theta_major >= 0)
            theta_major = (ctx->tune.theta_major != 0 && n_slices == 1 && pp.ncb() <= 65535) ? 1 : 0;
        if (theta_major && ctx->tune.xcd_map)
            theta_major |= 2;        /* candidate blocks dealt to the XCDs (k_score_pairs) */
        const uint16_t* lane_map = nullptr;
        if (int rc = lane_map_for(ctx, pp, &lane_map))
            return rc;
        csm_launch::ScoreLaunch a = score_launch(ctx, pp, grid, pass_lds_bytes(pp));
        a.theta_major = theta_major;
        a.lane_map = lane_map;
        return launched_ok(ctx, csm_launch::score_pairs(a, job), "pair-row score");
    }
    size_t lds = pass_lds_bytes(pp);
    if (lds > 160 * 1024 - 256)
        return fail(ctx, CSM_EINVAL, "internal: LDS region too large");
    const int n_buf = pick_buffers(ctx->tune, lds, (long)grid.x * grid.y * grid.z);
    csm_launch::ScoreLaunch a = score_launch(ctx, pp, grid, lds * n_buf);
    a.n_buf = n_buf;
    return launched_ok(ctx, csm_launch::score_strided(a, job), "strided score");
}

/* the single-window pair kernel over a work list of (slice, candidate block) items */
int launch_score_list(csm_ctx* ctx, const ScoreJob& job, const PassPlan& pp, const uint32_t* items,
                      const uint32_t* count, int blocks)
{
    if (!pp.pairs)
        return fail(ctx, CSM_EINVAL, "internal: list launches need the pair kernel");
    const uint16_t* lane_map = nullptr;
    if (int rc = lane_map_for(ctx, pp, &lane_map))
        return rc;
    csm_launch::ScoreLaunch a = score_launch(ctx, pp, dim3(blocks), pass_lds_bytes(pp));
    a.lane_map = lane_map;
    a.items = items;
    a.count = count;
    a.blocks = blocks;
    return launched_ok(ctx, csm_launch::score_pairs_list(a, job), "pair-row list");
}

int launch_argmax(csm_ctx* ctx, const ScoreJob& job, const PassPlan& plan, int n_theta)
{
    /* only the lane <-> candidate mapping (cbx, groups, R) matters to this pass:
     * a pair plan borrows the R = 8 instantiation of the plain kernel */
    const csm_launch::ScoreLaunch a = score_launch(ctx, plan, dim3(plan.ncb(), n_theta, 1), 0);
    return launched_ok(ctx, csm_launch::argmax(a, job), "arg-max");     /* k_argmax<128, 6 | 8> exist */
}

/* a window's last row block as an R = 6 launch of its own? (launch_score_batch) */
bool tail_split(const csm_ctx* ctx, const PassPlan& pp)
{
    const int cby = pp.groups * pp.R, tail_rows = pp.ny - (pp.ncby - 1) * cby;
    return pp.pairs && pp.R == 8 && pp.ncby >= 2 && tail_rows > 0 && tail_rows <= pp.groups * 6 && ctx->tune.pair_tail;
}

int launch_pairs_batch(csm_ctx* ctx, const ScoreJob* jobs_dev, const PassPlan& pp, dim3 grid, BlockBase bb,
                       const JointList* list, int which)
{
    const size_t lds = pass_lds_bytes(pp);
    const uint16_t* lane_map = nullptr;
    if (int rc = lane_map_for(ctx, pp, &lane_map))
        return rc;
    /* one job's workgroups on one XCD (k_score_pairs*_batch, xcd_block); CSM_TUNE_NO_XCD_MAP: identity */
    const int xcd_map = ctx->tune.xcd_map ? 1 : 0;
    if (pp.joint) {
        csm::JointLaunch L;
        L.stream = ctx->stream;
        L.device = ctx->device;
        L.jobs_dev = jobs_dev;
        L.grid = dim3(grid.x, (grid.y + 1) / 2, grid.z);
        L.lds_bytes = lds;
        L.ls = pp.lstride;
        L.R = pp.R;
        L.cbx = pp.cbx;
        L.groups = pp.groups;
        L.lane_map = lane_map;
        L.xcd_map = xcd_map;
        L.row_base = bb.row_base;
        L.cb_base = bb.cb_base;
        L.ncb = bb.ncb;
        L.fp32 = pp.fp32 ? 1 : 0;
        if (list && !pp.fp32) {
            L.items = list->items[which];
            L.item_count = list->counts + which;
            L.list_blocks = list->blocks;
        }
        const int e = csm::launch_joint_batch(L);
        if (e < 0)
            return fail(ctx, CSM_EINVAL, "internal: no joint kernel for LS %d R %d", pp.lstride, pp.R);
        if (e != 0)
            return fail(ctx, CSM_EIO, "joint fine kernel launch failed: %s", hipGetErrorString((hipError_t)e));
        return CSM_OK;
    }
    csm_launch::ScoreLaunch a = score_launch(ctx, pp, grid, lds);
    a.lane_map = lane_map;
    a.xcd_map = xcd_map;
    a.bb = bb;
    return launched_ok(ctx, csm_launch::score_pairs_batch(a, jobs_dev), "pair-row batch");
}

int launch_score_batch(csm_ctx* ctx, const ScoreJob* jobs_dev, int n_jobs, const PassPlan& pp,
                       int n_theta_max, int n_slices, int theta_groups, const JointList* list)
{
    /* theta_groups > 0: that many workgroups per (block, job) share the theta slices */
    const dim3 grid(pp.ncb(), (theta_groups > 0 && !pp.pairs) ? std::min(theta_groups, n_theta_max) : n_theta_max,
                    n_jobs * n_slices);
    if (pp.pairs) {
        if (n_slices != 1)
            return fail(ctx, CSM_EINVAL, "internal: pair kernel batches are not tile-split");
        /* The last row block of a window rarely needs all R = 8 rows of its lanes (84 rows in
         * blocks of 48: the second block has 36). Where R = 6 covers it with the same lane
         * groups, that block is a launch of its own: three quarters of the reads and
         * multiply-adds per entry for half of the workgroups (CSM_TUNE_NO_PAIR_TAIL: one launch). */
        const int cby = pp.groups * pp.R;
        if (!tail_split(ctx, pp))
            return launch_pairs_batch(ctx, jobs_dev, pp, grid, BlockBase{ 0, 0, pp.ncb() }, list, 0);
        PassPlan tail = pp;
        tail.R = 6;
        /* The tail launch keeps the main launch's row pitch: k_bin wrote the entries' LDS offsets
         * for THAT pitch (BinJob.lstride). Round 2's last commit gave the tail its own
         * conflict-free pitch (156 instead of 150) and thereby scored candidate rows 48..83 of
         * every window on the wrong cells -- unnoticed because winners sit near the window's
         * centre; tests/test_gpu_headline.py (full S / K dumps of this launch shape) found it. */
        int rc = launch_pairs_batch(ctx, jobs_dev, pp, dim3(pp.ncbx * (pp.ncby - 1), grid.y, grid.z),
                                    BlockBase{ 0, 0, pp.ncb() }, list, 0);
        if (rc)
            return rc;
        return launch_pairs_batch(ctx, jobs_dev, tail, dim3(pp.ncbx, grid.y, grid.z),
                                  BlockBase{ (pp.ncby - 1) * cby, pp.ncbx * (pp.ncby - 1), pp.ncb() }, list, 1);
    }
    size_t lds = pass_lds_bytes(pp);
    if (lds > 160 * 1024 - 256)
        return fail(ctx, CSM_EINVAL, "internal: LDS region too large");
    const int n_buf = pick_buffers(ctx->tune, lds, (long)grid.x * grid.y * grid.z);
    csm_launch::ScoreLaunch a = score_launch(ctx, pp, grid, lds * n_buf);
    a.n_buf = n_buf;
    a.n_slices = n_slices;
    return launched_ok(ctx, csm_launch::score_strided_batch(a, jobs_dev), "strided batch");
}

/* One launch for all pending levels (k_boxmax_batch). The job table is uploaded
 * from context-owned host memory. */
int launch_box_jobs(csm_ctx* ctx, const std::vector<PendingBox>& pending)
{
    if (pending.empty())
        return CSM_OK;
    ctx->box_stage.resize(pending.size());
    int rows_max = 0, pitch_max = 0;
    for (size_t i = 0; i < pending.size(); ++i) {
        const DeviceGrid& g = *pending[i].grid;
        BoxJob& b = ctx->box_stage[i];
        b.src = g.levels[0].cells;
        b.dst = g.levels[pending[i].level].cells;
        b.rows = g.rows;
        b.cols = g.cols;
        b.pitch = g.pitch;
        b.win = g.levels[pending[i].level].win;
        rows_max = std::max(rows_max, g.rows);
        pitch_max = std::max(pitch_max, g.pitch);
    }
    int rc = ensure(ctx, ctx->box_jobs, pending.size() * sizeof(BoxJob));
    if (rc)
        return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->box_jobs.p, ctx->box_stage.data(), pending.size() * sizeof(BoxJob),
                                hipMemcpyHostToDevice, ctx->stream));
    ScopedTimer tm(ctx, "boxmax");
    for (size_t first = 0; first < pending.size(); first += 65535) {      /* grid.z limit */
        const unsigned nz = (unsigned)std::min<size_t>(65535, pending.size() - first);
        const int e = csm_launch::boxmax_batch(ctx->stream, dim3(ceil_div(pitch_max, kBoxTC), ceil_div(rows_max, kBoxTR), nz),
                                               reinterpret_cast<const BoxJob*>(ctx->box_jobs.p) + first);
        if (e)
            return launched_ok(ctx, e, "box-maximum");
    }
    return CSM_OK;
}

/* Prepares level `win` of g for building: allocates (or reuses) its buffer and
 * records it in `pending`; the caller launches. `reuse`: a buffer of at least
 * rows * pitch * 2 bytes to build into, or null to allocate one. */
int build_level(csm_ctx* ctx, DeviceGrid& g, int win, Level* out, uint16_t* reuse,
                size_t reuse_cap)
{
    if (win < 1 || win > g.rows || win > g.cols)
        return fail(ctx, CSM_EINVAL, "box-max window %d does not fit %dx%d", win, g.rows, g.cols);
    if (win > kBoxMaxWin)
        return fail(ctx, CSM_EINVAL, "box-max window %d exceeds %d", win, kBoxMaxWin);
    const size_t bytes = (size_t)g.rows * g.pitch * 2;
    uint16_t* dst = reuse;
    size_t cap = reuse_cap;
    if (!dst) {
        ++ctx->alloc_epoch;
        if (hipMalloc(reinterpret_cast<void**>(&dst), bytes) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
        cap = bytes;
    }
    out->win = win;
    out->cells = dst;
    out->owned = true;
    out->stale = false;
    out->cap = cap;
    return CSM_OK;
}

/* index of the level with this window; builds and appends it if missing. With
 * `pending` the launch is left to the caller (launch_box_jobs), so that many
 * levels of many maps share one launch. */
int level_for_window(csm_ctx* ctx, DeviceGrid& g, int win, int* index,
                     std::vector<PendingBox>* pending)
{
    std::vector<PendingBox> local;
    std::vector<PendingBox>& todo = pending ? *pending : local;
    auto finish = [&]() { return pending ? CSM_OK : launch_box_jobs(ctx, local); };
    for (size_t i = 0; i < g.levels.size(); ++i)
        if (g.levels[i].win == win) {
            Level& have = g.levels[i];
            if (have.stale) {
                /* the base was rebuilt (csm_construct_map_from_scans): redo the box
                 * maximum, into the old buffer when it is large enough */
                const size_t bytes = (size_t)g.rows * g.pitch * 2;
                const bool fits = have.owned && have.cap >= bytes;
                if (have.owned && !fits) {
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                    (void)hipFree(have.cells);
                    have.cells = nullptr;
                    have.cap = 0;
                }
                Level fresh;
                int rc = build_level(ctx, g, win, &fresh, fits ? have.cells : nullptr, have.cap);
                if (rc)
                    return rc;
                have = fresh;
                todo.push_back({ &g, (int)i });
            }
            *index = (int)i;
            return finish();
        }
    Level lv;
    int rc = build_level(ctx, g, win, &lv);
    if (rc)
        return rc;
    g.levels.push_back(lv);
    *index = (int)g.levels.size() - 1;
    todo.push_back({ &g, *index });
    return finish();
}

/* The pair-row copy of level 0 with at least `need_pad` cells of zero padding. */
int ensure_xgrid(csm_ctx* ctx, DeviceGrid& g, int need_pad)
{
    if (g.xg && !g.xg_stale && g.xg_pad >= need_pad)
        return CSM_OK;
    const int pad = std::max(need_pad, g.xg_pad);
    const int prows = (g.rows + 2 * pad + 1) / 2 + 1;
    const int xp = (g.cols + 2 * pad + 1) & ~1;
    const size_t bytes = (size_t)prows * xp * 8;
    if (bytes > g.xg_cap) {
        ++ctx->alloc_epoch;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (g.xg)
            (void)hipFree(g.xg);
        g.xg = nullptr;
        g.xg_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&g.xg), bytes) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
        g.xg_cap = bytes;
    }
    const size_t total = (size_t)prows * xp;
    const int blocks = (int)std::min<size_t>(4096, (total + 255) / 256);
    ScopedTimer tm(ctx, "expand");
    if (int rc = launched_ok(ctx, csm_launch::expand_pairs(ctx->stream, blocks, g.levels[0].cells, g.rows, g.cols, g.pitch,
                                                            g.xg, prows, xp, pad), "pair-row copy"))
        return rc;
    g.xg_pad = pad;
    g.xg_pitch = xp;
    g.xg_stale = false;
    g.xgf_valid = false;
    return CSM_OK;
}

/* The fp32 key copy in the layout of the (up-to-date) pair-row copy. */
int ensure_xgrid_f(csm_ctx* ctx, DeviceGrid& g)
{
    if (g.xgf && g.xgf_valid)
        return CSM_OK;
    if (!g.xg || g.xg_stale)
        return fail(ctx, CSM_EINVAL, "internal: pair-row copy missing");
    const int prows = (g.rows + 2 * g.xg_pad + 1) / 2 + 1;
    const size_t bytes = (size_t)prows * g.xg_pitch * 8;
    if (bytes > g.xgf_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (g.xgf)
            (void)hipFree(g.xgf);
        g.xgf = nullptr;
        g.xgf_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&g.xgf), bytes) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
        g.xgf_cap = bytes;
    }
    ScopedTimer tm(ctx, "expand");
    const int e = csm::launch_expand_pairs_f(ctx->stream, g.levels[0].cells, g.rows, g.cols, g.pitch, g.xgf, prows,
                                             g.xg_pitch, g.xg_pad);
    if (e != 0)
        return fail(ctx, CSM_EIO, "k_expand_pairs_f launch failed: %s", hipGetErrorString((hipError_t)e));
    g.xgf_valid = true;
    return CSM_OK;
}

} /* namespace csm_host */

