/* csm_api.hip -- host side of the matchers in libcsm_hip.so: the C ABI of include/csm_hip.h
 * (planner, batch staging, graph replay) on top of the gfx950 kernels. No device code here:
 * the kernels are launched through csm_launch.hpp (per-slice kernels, csm_kernels.hip),
 * csm_joint.hpp (batched fine level) and csm_phase.hpp (coarse-first search), each a
 * translation unit of its own, so an edit of the host logic recompiles in seconds.
 *
 * Host-side expressions that must agree bit for bit with the reference
 * (search step, window, projection, pose algebra) are restated here from the
 * cited reference lines and are built with -ffp-contract=off.
 */
#include "csm_internal.hpp"

#include "csm_launch.hpp"
#include "csm_joint.hpp"
#include "csm_phase.hpp"


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

void free_levels(DeviceGrid& g, bool keep_base)
{
    for (auto& kv : g.phase)
        if (kv.second.grid)
            free_levels(*kv.second.grid, false);
    g.phase.clear();
    ++g.base_epoch;
    if (!keep_base) {
        if (g.xg)
            (void)hipFree(g.xg);
        g.xg = nullptr;
        g.xg_cap = 0;
        g.xg_stale = true;
        if (g.xgf)
            (void)hipFree(g.xgf);
        g.xgf = nullptr;
        g.xgf_cap = 0;
        g.xgf_valid = false;
        if (g.alloc)
            (void)hipFree(g.alloc);
        g.alloc = nullptr;
        g.alloc_cap = 0;
        g.alloc_user = false;
        g.alloc_stale = true;
    }
    for (size_t i = keep_base ? 1 : 0; i < g.levels.size(); ++i)
        if (g.levels[i].owned && g.levels[i].cells)
            (void)hipFree(g.levels[i].cells);
    g.levels.resize(keep_base && !g.levels.empty() ? 1 : 0);
}

} /* namespace csm_host */

namespace {

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

/* Splits [0, n) over up to four host threads (the batch entries touch tens of
 * megabytes of scan data before anything can be launched); fn(lo, hi) must not
 * touch the context. */
template <class F>
void host_parallel_for(int n, int grain, F fn)
{
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int nt = std::min(std::min(4, hw), n / std::max(1, grain));
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> workers;
    for (int w = 1; w < nt; ++w)
        workers.emplace_back(fn, (int)((long)n * w / nt), (int)((long)n * (w + 1) / nt));
    fn(0, (int)((long)n / nt));
    for (auto& t : workers)
        t.join();
}

/* scan_finite_max of every query; returns the first offending query or -1 */
int scans_finite_max(const csm_loop_query* queries, int n_queries, double* max_range)
{
    std::atomic<int> bad(n_queries);
    host_parallel_for(n_queries, 128, [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i) {
            const csm_scan& sc = queries[i].scan;
            if (!sc.angles || !sc.ranges || sc.n_points < 1 || !scan_finite_max(&sc, &max_range[i])) {
                int cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {
                }
                return;
            }
        }
    });
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





const int kCoarseSlices = 8;

/* Launch geometry of one scoring pass (one level of one window shape). */
struct PassPlan {
    int nx = 0, ny = 0, stride = 1, log2s = 0;
    int cbx = 0, groups = 0, R = 0, ncbx = 0, ncby = 0, lstride = 0;
    bool weighted = true;     /* entries carry beam multiplicities */
    bool pairs = false;       /* pair-row fine kernel (k_score_pairs): lstride = slots per pair row */
    int lists = 1;            /* entry lists in LDS: 2 = the batch kernel that takes two slices per workgroup */
    bool joint = false;       /* ... on joint entries of the two slices (one list; csm_joint_kernels.hip) */
    bool fp32 = false;        /* this launch is the packed-fp32 bound pass of the joint kernel */
    int ncb() const { return ncbx * ncby; }
};

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
bool plan_pass_pairs(const Tuning& tune, int nx, int ny, PassPlan* out, bool two_slices = false)
{
    const int lists = two_slices ? 2 : 1;
    if (!tune.two_slices && two_slices)
        return plan_pass_pairs(tune, nx, ny, out, false);
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
        while (g > 1 && pair_lds_bytes(p.lstride, g * R, lists) > 80 * 1024 - 1024)
            --g;
        if (force_g && force_g <= g)
            g = force_g;
        if (g < 1 || pair_lds_bytes(p.lstride, g * R, lists) > 160 * 1024 - 256)
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
                    ncbx, p.cbx, p.lstride, g, p.ncby, pair_lds_bytes(p.lstride, g * R, lists), cost);
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
            pair_lds_bytes(tune.pair_ls, out->groups * out->R, lists) <= 160 * 1024 - 256)
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
        return pair_lds_bytes(p.lstride, p.groups * p.R, p.lists);
    const int cby = p.groups * p.R;
    const int rows = p.stride > 1 ? ((kTile + p.stride - 1) / p.stride + cby - 1) * p.stride
                                  : kTile + cby - 1;
    return (size_t)rows * p.lstride * 4 + kPbMax * 4;
}

/* Launch geometry of one search window. */
struct Plan {
    int n_theta = 0, n = 0;
    int win_x = 0, win_y = 0, L = 1;
    int nxc = 0, nyc = 0, nx = 0, ny = 0;
    int x_lo = 0, y_lo = 0, x_hi = 0, y_hi = 0;
    PassPlan fine, coarse;
    int tiles_x = 0, tiles_y = 0, max_tiles = 0;
};

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

/* One launch of the pair kernels over row blocks [first block of `pp`'s numbering ...) of a batch. */
/* Work list of the exact joint kernel after the bound pass (k_bound_select): items of the main
 * launch, items of the R = 6 tail launch, their counts, workgroups to share them. */
struct JointList {
    const uint32_t* items[2] = { nullptr, nullptr };
    const uint32_t* counts = nullptr;       /* [2] */
    int blocks = 0;
};

/* a window's last row block as an R = 6 launch of its own? (launch_score_batch) */
bool tail_split(const csm_ctx* ctx, const PassPlan& pp)
{
    const int cby = pp.groups * pp.R, tail_rows = pp.ny - (pp.ncby - 1) * cby;
    return pp.pairs && pp.R == 8 && pp.ncby >= 2 && tail_rows > 0 && tail_rows <= pp.groups * 6 && ctx->tune.pair_tail;
}

int launch_pairs_batch(csm_ctx* ctx, const ScoreJob* jobs_dev, const PassPlan& pp, dim3 grid, BlockBase bb,
                       const JointList* list = nullptr, int which = 0)
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
                       int n_theta_max, int n_slices, int theta_groups = 0, const JointList* list = nullptr)
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



/* Box-maximum levels to build: collected first, launched together (launch_box_jobs). */
struct PendingBox {
    DeviceGrid* grid;
    int level;          /* index into grid->levels: its cells are the destination */
};

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
int build_level(csm_ctx* ctx, DeviceGrid& g, int win, Level* out, uint16_t* reuse = nullptr,
                size_t reuse_cap = 0)
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
                     std::vector<PendingBox>* pending = nullptr)
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

struct WindowOutputs {
    uint32_t* dump_s = nullptr;     /* device */
    uint16_t* dump_k = nullptr;
};

/* The CSM pipeline on device-resident inputs; asynchronous. */
/* Two-phase search (csm_phase_kernels.hip). mode 1: score the window and STORE every candidate's
 * sums [n_theta][nx][ny] (the coarse pass, run on the level's phase-major copy), nothing else;
 * mode 2: the fine level, its eligibility from such sums (level_s / level_k with strides nxs, nys),
 * over the work list of the blocks that can still win. */
struct TwoPhaseCtl {
    int mode = 0;
    uint32_t* level_s = nullptr;
    uint32_t* level_k = nullptr;
    int nxs = 0, nys = 0;
    uint32_t stats[4] = { 0, 0, 0, 0 };      /* mode 2, filled on request: items, kept, dropped */
};

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
    if (binj_lds > 150 * 1024 || !plan_pass_pairs(ctx->tune, p.nx, p.ny, &jp, true) || jp.lists != 2)
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
               csm_result* out_dev, const WindowOutputs* dumps, bool force_coarse = false,
               TwoPhaseCtl* tp = nullptr)
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
                   const csm_result* have = nullptr, bool* changed = nullptr)
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

} /* namespace */

/* ------------------------------------------------------------------ C ABI */

extern "C" {

const char* csm_version(void) { return "csm_hip 0.1 (gfx950)"; }

int csm_create(const csm_config* cfg, csm_ctx** out)
{
    if (!out)
        return CSM_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return CSM_ENODEV;
    const int dev = cfg ? cfg->device_id : 0;
    if (dev < 0 || dev >= ndev)
        return CSM_ENODEV;
    csm_ctx* ctx = new csm_ctx();
    ctx->device = dev;
    {
        Tuning& t = ctx->tune;
        const uint32_t off = cfg ? cfg->tuning_off : 0u;
        t.lane_map = !(off & CSM_TUNE_NO_LANE_MAP);
        t.xcd_map = !(off & CSM_TUNE_NO_XCD_MAP);
        t.pair_tail = !(off & CSM_TUNE_NO_PAIR_TAIL);
        t.two_slices = !(off & CSM_TUNE_NO_TWO_SLICES);
        t.joint = !(off & CSM_TUNE_NO_JOINT);
        t.bound_pass = !(off & CSM_TUNE_NO_BOUND_PASS);
        t.two_phase = (off & CSM_TUNE_NO_TWO_PHASE) ? -1 : (off & CSM_TUNE_FORCE_TWO_PHASE) ? 1 : 0;
        t.graphs = !(off & CSM_TUNE_NO_GRAPHS);
        t.tile_split = !(off & CSM_TUNE_NO_TILE_SPLIT);
        t.map_host_projection = (off & CSM_TUNE_MAP_HOST_PROJECTION) != 0;
        if (off & CSM_TUNE_NO_THETA_MAJOR)
            t.theta_major = 0;
        t.map_unc_cap = cfg ? cfg->map_uncertain_cap : 0;
#ifdef CSM_TUNING
        /* tuning builds only (tools/build_variant.sh): forced launch shapes from the environment,
         * read here once -- never on a launch path */
        auto env_int = [](const char* name, int dflt) {
            const char* e = getenv(name);
            return e ? atoi(e) : dflt;
        };
        t.lane_map = env_int("CSM_LANE_MAP", t.lane_map) != 0;
        t.xcd_map = env_int("CSM_XCD_MAP", t.xcd_map) != 0;
        t.pair_tail = env_int("CSM_PAIR_TAIL", t.pair_tail) != 0;
        t.two_slices = env_int("CSM_PAIR_SLICES", t.two_slices ? 2 : 1) != 1;
        t.joint = env_int("CSM_JOINT", t.joint) != 0;
        t.bound_pass = env_int("CSM_BOUND_PASS", t.bound_pass) != 0;
        t.theta_major = env_int("CSM_THETA_MAJOR", t.theta_major);
        t.fine_slices = env_int("CSM_FINE_SLICES", 0);
        t.force_r = env_int("CSM_FORCE_R", 0);
        t.pair_r = env_int("CSM_PAIR_R", 0);
        t.pair_ncbx = env_int("CSM_PAIR_NCBX", 0);
        t.pair_groups = env_int("CSM_PAIR_GROUPS", 0);
        t.pair_ls = env_int("CSM_PAIR_LS", 0);
        t.pair_tail_ls = env_int("CSM_PAIR_TAIL_LS", 0);
        t.nbuf = env_int("CSM_NBUF", 0);
        t.plan_debug = env_int("CSM_PLAN_DEBUG", 0) != 0;
        t.host_timing = env_int("CSM_HOST_TIMING", 0) != 0;
#endif
    }
    if (hipSetDevice(dev) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return CSM_ENODEV;
    }
    ctx->stream = ctx->own_stream;
    std::vector<double> lut(65536);
    csm_host_probability_lut(lut.data());
    if (hipMalloc(reinterpret_cast<void**>(&ctx->lut_dev), 65536 * 8) != hipSuccess ||
        hipMemcpy(ctx->lut_dev, lut.data(), 65536 * 8, hipMemcpyHostToDevice) != hipSuccess) {
        delete ctx;
        return CSM_ENOMEM;
    }
    *out = ctx;
    return CSM_OK;
}

int csm_destroy(csm_ctx* ctx)
{
    if (!ctx)
        return CSM_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->graphs)
        (void)hipGraphExecDestroy(kv.second);
    ctx->graphs.clear();
    if (ctx->q_pin)
        (void)hipHostFree(ctx->q_pin);
    ctx->q_pin = nullptr;
    for (auto& kv : ctx->grids)
        free_levels(kv.second, false);
    DevBuf* bufs[] = { &ctx->hits, &ctx->sorted, &ctx->tiles, &ctx->ntiles, &ctx->misc,
                       &ctx->coarse_s, &ctx->coarse_k, &ctx->best, &ctx->dump_s, &ctx->dump_k,
                       &ctx->scratch, &ctx->b_prod, &ctx->b_hits, &ctx->b_sorted, &ctx->b_tiles,
                       &ctx->b_ntiles, &ctx->b_lvl, &ctx->b_best, &ctx->b_jobs, &ctx->b_out, &ctx->b_abest, &ctx->bound_stats, &ctx->b_items, &ctx->tp_items, &ctx->ph_hits, &ctx->q_dev,
                       &ctx->fine_s, &ctx->fine_k, &ctx->tie, &ctx->ex_fine, &ctx->ex_fine_k, &ctx->ex_coarse, &ctx->ex_coarse_k,
                       &ctx->scan_dev, &ctx->unc, &ctx->sorted_rc, &ctx->b_sorted_rc, &ctx->rec_dev, &ctx->c_scans, &ctx->c_jobs, &ctx->box_jobs,
                       &ctx->m_rays, &ctx->m_recs, &ctx->m_cell, &ctx->m_lists, &ctx->m_cnt, &ctx->m_lut };
    for (DevBuf* b : bufs)
        if (b->p)
            (void)hipFree(b->p);
    if (ctx->lut_dev)
        (void)hipFree(ctx->lut_dev);
    for (auto& kv : ctx->lane_maps)
        if (kv.second)
            (void)hipFree(kv.second);
    if (ctx->pin)
        (void)hipHostFree(ctx->pin);
    if (ctx->pin_scans)
        (void)hipHostFree(ctx->pin_scans);
    for (auto& kv : ctx->timers)
        for (auto& s : kv.second.spans) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
    for (auto& h : ctx->resident_hold)
        (void)hipEventDestroy(h.first);
    ctx->resident_hold.clear();           /* returns the blocks they hold to pin_free */
    for (auto& b : ctx->pin_free)
        (void)hipHostFree(b.first);
    for (hipEvent_t e : ctx->event_pool)
        (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->m_ev)
        if (e)
            (void)hipEventDestroy(e);
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return CSM_OK;
}

const char* csm_last_error(const csm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int csm_set_stream(csm_ctx* ctx, void* hip_stream)
{
    if (!ctx)
        return CSM_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return CSM_OK;
}

int csm_synchronize(csm_ctx* ctx)
{
    if (!ctx)
        return CSM_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CSM_OK;
}

int csm_upload_grid(csm_ctx* ctx, uint64_t map_id, const uint16_t* dense, int32_t rows, int32_t cols)
{
    if (!ctx || !dense || rows < 1 || cols < 1)
        return fail(ctx, CSM_EINVAL, "csm_upload_grid: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DeviceGrid& g = ctx->grids[map_id];
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_levels(g, false);
    g.rows = rows;
    g.cols = cols;
    g.pitch = (cols + 7) & ~7;
    /* first known row / column (tightens the edge-band test of k_bin) */
    g.known_r0 = rows;
    g.known_c0 = cols;
    for (int r = 0; r < rows; ++r) {
        const uint16_t* line = dense + (size_t)r * cols;
        for (int c = 0; c < cols; ++c)
            if (line[c] != 0) {
                if (r < g.known_r0)
                    g.known_r0 = r;
                if (c < g.known_c0)
                    g.known_c0 = c;
                break;          /* later cells of this row cannot lower known_c0 below c */
            }
    }
    Level base;
    const size_t bytes = (size_t)rows * g.pitch * 2;
    if (hipMalloc(reinterpret_cast<void**>(&base.cells), bytes) != hipSuccess) {
        ctx->grids.erase(map_id);
        return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
    }
    base.win = 1;
    base.owned = true;
    base.cap = bytes;
    g.levels.push_back(base);
    /* GridMap::CopyValues' output goes through a pinned staging buffer of the context,
     * already in the device layout (pitched rows, pad cells 0): one contiguous DMA, no
     * device-side clearing, and the caller's buffer is free again when the call returns */
    if (bytes > ctx->pin_cap) {
        if (ctx->pin)
            (void)hipHostFree(ctx->pin);
        ctx->pin = nullptr;
        ctx->pin_cap = 0;
        if (hipHostMalloc(&ctx->pin, bytes + bytes / 4, hipHostMallocDefault) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
        ctx->pin_cap = bytes + bytes / 4;
    }
    uint16_t* stage = reinterpret_cast<uint16_t*>(ctx->pin);
    for (int r = 0; r < rows; ++r) {
        std::memcpy(stage + (size_t)r * g.pitch, dense + (size_t)r * cols, (size_t)cols * 2);
        if (g.pitch > cols)
            std::memset(stage + (size_t)r * g.pitch + cols, 0, (size_t)(g.pitch - cols) * 2);
    }
    HIP_TRY(ctx, hipMemcpyAsync(base.cells, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      /* the staging buffer is reused by the next upload */
    return CSM_OK;
}

int csm_upload_grid_blocks(csm_ctx* ctx, uint64_t map_id, const uint16_t* const* blocks, int32_t block_rows,
                           int32_t block_cols, int32_t log2_block)
{
    if (!ctx || !blocks || block_rows < 1 || block_cols < 1 || log2_block < 0 || log2_block > 10 ||
        ((int64_t)block_rows << log2_block) > 65536 || ((int64_t)block_cols << log2_block) > 65536)
        return fail(ctx, CSM_EINVAL, "csm_upload_grid_blocks: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rows = block_rows << log2_block, cols = block_cols << log2_block;
    const int n_blocks = block_rows * block_cols;
    const size_t block_cells = (size_t)1 << (2 * log2_block);
    int n_alloc = 0;
    for (int b = 0; b < n_blocks; ++b)
        n_alloc += blocks[b] != nullptr;
    DeviceGrid& g = ctx->grids[map_id];
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_levels(g, false);
    g.rows = rows;
    g.cols = cols;
    g.pitch = (cols + 7) & ~7;
    Level base;
    const size_t bytes = (size_t)rows * g.pitch * 2;
    if (hipMalloc(reinterpret_cast<void**>(&base.cells), bytes) != hipSuccess) {
        ctx->grids.erase(map_id);
        return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
    }
    base.win = 1;
    base.owned = true;
    base.cap = bytes;
    g.levels.push_back(base);
    if (hipMalloc(reinterpret_cast<void**>(&g.alloc), (size_t)n_blocks + 64) != hipSuccess)
        return fail(ctx, CSM_ENOMEM, "hipMalloc(%d) failed", n_blocks + 64);
    g.alloc_cap = (size_t)n_blocks + 64;
    /* pinned staging: [first known row, column][slot of every block][the allocated blocks]; one DMA */
    const size_t head = (((size_t)(2 + n_blocks) * 4) + 255) & ~(size_t)255;
    const size_t stage_bytes = head + (size_t)n_alloc * block_cells * 2;
    if (stage_bytes > ctx->pin_cap) {
        if (ctx->pin)
            (void)hipHostFree(ctx->pin);
        ctx->pin = nullptr;
        ctx->pin_cap = 0;
        if (hipHostMalloc(&ctx->pin, stage_bytes + stage_bytes / 4, hipHostMallocDefault) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipHostMalloc(%zu) failed", stage_bytes);
        ctx->pin_cap = stage_bytes + stage_bytes / 4;
    }
    int32_t* h_head = reinterpret_cast<int32_t*>(ctx->pin);
    h_head[0] = rows;                       /* "no known cell": what csm_upload_grid reports */
    h_head[1] = cols;
    uint16_t* h_packed = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(ctx->pin) + head);
    int next = 0;
    for (int b = 0; b < n_blocks; ++b) {
        h_head[2 + b] = blocks[b] ? next : -1;
        if (blocks[b])
            std::memcpy(h_packed + (size_t)next++ * block_cells, blocks[b], block_cells * 2);
    }
    int rc = ensure(ctx, ctx->scratch, stage_bytes);
    if (rc)
        return rc;
    char* d_stage = reinterpret_cast<char*>(ctx->scratch.p);
    HIP_TRY(ctx, hipMemcpyAsync(d_stage, ctx->pin, stage_bytes, hipMemcpyHostToDevice, ctx->stream));
    const int grid_blocks = (int)std::min<size_t>(4096, ((size_t)rows * g.pitch + 255) / 256);
    if (int e = csm_launch::deblock(ctx->stream, grid_blocks, reinterpret_cast<const uint16_t*>(d_stage + head),
                                    reinterpret_cast<const int32_t*>(d_stage) + 2, log2_block, block_cols, rows, cols,
                                    g.pitch, base.cells, g.alloc, n_blocks, reinterpret_cast<int32_t*>(d_stage)))
        return launched_ok(ctx, e, "block upload");
    int32_t known[2] = { rows, cols };
    HIP_TRY(ctx, hipMemcpyAsync(known, d_stage, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    g.known_r0 = known[0];
    g.known_c0 = known[1];
    g.alloc_log2 = log2_block;
    g.alloc_bcols = block_cols;
    g.alloc_user = true;
    g.alloc_stale = false;
    return CSM_OK;
}

int csm_has_grid(csm_ctx* ctx, uint64_t map_id)
{
    return ctx && find_grid(ctx, map_id) ? 1 : 0;
}

int csm_release_grid(csm_ctx* ctx, uint64_t map_id)
{
    if (!ctx)
        return CSM_EINVAL;
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_levels(*g, false);
    ctx->grids.erase(map_id);
    return CSM_OK;
}

int csm_build_pyramid(csm_ctx* ctx, uint64_t map_id, const int32_t* win_sizes, int32_t n_levels)
{
    if (!ctx || !win_sizes || n_levels < 1)
        return fail(ctx, CSM_EINVAL, "csm_build_pyramid: bad arguments");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (win_sizes[0] != 1)
        return fail(ctx, CSM_EINVAL, "win_sizes[0] must be 1 (level 0 is the grid itself)");
    free_levels(*g, true);
    std::vector<Level> lv;
    lv.push_back(g->levels[0]);
    for (int i = 1; i < n_levels; ++i) {
        Level l;
        if (win_sizes[i] == 1) {
            l = g->levels[0];
            l.owned = false;
        } else {
            int rc = build_level(ctx, *g, win_sizes[i], &l);
            if (rc) {
                for (size_t j = 1; j < lv.size(); ++j)
                    if (lv[j].owned)
                        (void)hipFree(lv[j].cells);
                return rc;
            }
        }
        lv.push_back(l);
    }
    g->levels = lv;
    std::vector<PendingBox> pending;
    for (int i = 1; i < n_levels; ++i)
        if (g->levels[i].owned)
            pending.push_back({ g, i });
    int rc = launch_box_jobs(ctx, pending);
    if (rc)
        return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CSM_OK;
}

int csm_download_level(csm_ctx* ctx, uint64_t map_id, int32_t level, uint16_t* out)
{
    if (!ctx || !out)
        return fail(ctx, CSM_EINVAL, "csm_download_level: bad arguments");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g || level < 0 || level >= (int)g->levels.size() || g->levels[level].stale)
        return fail(ctx, CSM_ENOENT, "map %llu level %d not resident",
                    (unsigned long long)map_id, level);
    HIP_TRY(ctx, hipMemcpy2DAsync(out, (size_t)g->cols * 2, g->levels[level].cells,
                                  (size_t)g->pitch * 2, (size_t)g->cols * 2, g->rows,
                                  hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CSM_OK;
}

/* ---- host set-up pieces ---- */

int csm_host_search_step(double resolution, const double* ranges, int32_t n,
                         double* step_x, double* step_y, double* step_theta)
{
    if (!ranges || n < 1)
        return CSM_EINVAL;
    search_step_from_max(resolution, *std::max_element(ranges, ranges + n), step_x, step_y, step_theta);
    return CSM_OK;
}

int csm_host_window(double range, double step)
{
    return static_cast<int>(std::ceil(0.5 * range / step));
}

int csm_host_min_known(int32_t n_points, double thr)
{
    /* smallest K in [0, n+1] with double(K)/double(n) > thr (monotone in K) */
    int lo = 0, hi = n_points + 1;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (static_cast<double>(mid) / static_cast<double>(n_points) > thr)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

void csm_host_compound(const double s[3], const double d[3], double out[3])
{
    const double sin_t = std::sin(s[2]);
    const double cos_t = std::cos(s[2]);
    const double x = cos_t * d[0] - sin_t * d[1] + s[0];
    const double y = sin_t * d[0] + cos_t * d[1] + s[1];
    const double th = s[2] + d[2];
    out[0] = x;
    out[1] = y;
    out[2] = th;
}

void csm_host_inverse_compound(const double s[3], const double e[3], double out[3])
{
    const double sin_t = std::sin(s[2]);
    const double cos_t = std::cos(s[2]);
    const double dx = e[0] - s[0];
    const double dy = e[1] - s[1];
    const double dt = e[2] - s[2];
    out[0] = cos_t * dx + sin_t * dy;
    out[1] = -sin_t * dx + cos_t * dy;
    out[2] = dt;
}

void csm_host_move_backward(const double e[3], const double d[3], double out[3])
{
    const double th = e[2] - d[2];
    const double sin_t = std::sin(th);
    const double cos_t = std::cos(th);
    const double x = e[0] - cos_t * d[0] + sin_t * d[1];
    const double y = e[1] - sin_t * d[0] - cos_t * d[1];
    out[0] = x;
    out[1] = y;
    out[2] = th;
}

int csm_host_project(const csm_geometry* geom, const double sensor_pose[3], double step_theta,
                     int32_t win_theta, const double* angles, const double* ranges, int32_t n,
                     int32_t* hit_col, int32_t* hit_row, double* r_cos, double* r_sin)
{
    if (!geom || !sensor_pose || !angles || !ranges || n < 1 || win_theta < 0 || !hit_col || !hit_row)
        return CSM_EINVAL;
    for (int t = -win_theta; t <= win_theta; ++t) {
        const double theta = sensor_pose[2] + step_theta * t;
        const size_t base = (size_t)(t + win_theta) * n;
        for (int i = 0; i < n; ++i) {
            /* ScanData::HitPoint, inc/sensor/sensor_data.hpp:189-203 */
            const double cos_t = std::cos(theta + angles[i]);
            const double sin_t = std::sin(theta + angles[i]);
            const double rc = ranges[i] * cos_t;
            const double rs = ranges[i] * sin_t;
            const double hx = sensor_pose[0] + rc;
            const double hy = sensor_pose[1] + rs;
            /* PositionToIndex, src/grid_map_new/grid_map_geometry.cpp:113-122 */
            hit_col[base + i] = static_cast<int>(std::floor((hx - geom->offset_x) / geom->resolution));
            hit_row[base + i] = static_cast<int>(std::floor((hy - geom->offset_y) / geom->resolution));
            if (r_cos)
                r_cos[base + i] = rc;
            if (r_sin)
                r_sin[base + i] = rs;
        }
    }
    return CSM_OK;
}

void csm_host_probability_lut(double* lut)
{
    for (unsigned v = 0; v < 65536; ++v)
        lut[v] = value_to_probability(v);
}

/* ---- hot path ---- */

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

} /* extern "C" */

/* ---- branch-and-bound batch ---- */

namespace {

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

} /* namespace */

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

int csm_build_pyramids(csm_ctx* ctx, const uint64_t* map_ids, int32_t n_maps, const int32_t* win_sizes,
                       int32_t n_levels)
{
    if (!ctx || !map_ids || n_maps < 1 || !win_sizes || n_levels < 1)
        return fail(ctx, CSM_EINVAL, "csm_build_pyramids: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<PendingBox> pending;
    for (int i = 0; i < n_maps; ++i) {
        DeviceGrid* g = find_grid(ctx, map_ids[i]);
        if (!g)
            return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_ids[i]);
        for (int l = 0; l < n_levels; ++l) {
            int index = 0;
            int rc = level_for_window(ctx, *g, win_sizes[l], &index, &pending);
            if (rc)
                return rc;
        }
    }
    return launch_box_jobs(ctx, pending);      /* all levels of all maps: one launch */
}

/* ---- measurement hooks ---- */

int csm_last_search_info(csm_ctx* ctx, csm_search_info* out)
{
    if (!ctx || !out)
        return CSM_EINVAL;
    std::memset(out, 0, sizeof(*out));
    out->nominal_candidates = ctx->last_nominal;
    out->coarse_nodes_scored = ctx->last_coarse_nodes;
    out->fine_candidates_scored = ctx->last_fine_candidates;
    if (ctx->tp_count_dev) {
        uint32_t kept = 0;
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipMemcpyAsync(&kept, ctx->tp_count_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        out->two_phase = 1;
        out->blocks_scored = kept;
        out->blocks_skipped = ctx->tp_blocks_total - (int64_t)kept;
        out->fine_candidates_scored = (int64_t)kept * ctx->last_block_candidates;
    }
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

int csm_enable_kernel_timing(csm_ctx* ctx, int32_t enable)
{
    if (!ctx)
        return CSM_EINVAL;
    ctx->timing = enable;
    return CSM_OK;
}

static void drain_timer(csm_ctx* ctx, KernelTimer& kt)
{
    for (auto& s : kt.spans) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            kt.total_ms += ms;
            kt.launches += 1;
        }
        ctx->event_pool.push_back(s.a);
        ctx->event_pool.push_back(s.b);
    }
    kt.spans.clear();
}

int csm_kernel_time(csm_ctx* ctx, const char* name, double* total_ms, int64_t* launches)
{
    if (!ctx || !name)
        return CSM_EINVAL;
    auto it = ctx->timers.find(name);
    if (it == ctx->timers.end()) {
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = 0;
        return CSM_OK;
    }
    drain_timer(ctx, it->second);
    if (total_ms) *total_ms = it->second.total_ms;
    if (launches) *launches = it->second.launches;
    return CSM_OK;
}

int csm_reset_kernel_timing(csm_ctx* ctx)
{
    if (!ctx)
        return CSM_EINVAL;
    for (auto& kv : ctx->timers) {
        drain_timer(ctx, kv.second);
        kv.second.total_ms = 0.0;
        kv.second.launches = 0;
    }
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
