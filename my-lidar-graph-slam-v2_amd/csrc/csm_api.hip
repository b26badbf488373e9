/* csm_api.hip -- the context and grid part of the C ABI of include/csm_hip.h (host code only): create /
 * destroy, streams, grid upload (dense and block-sparse), pyramids, downloads, the bit-exact host
 * restatements exported for callers (search step, window, pose algebra, projection, probability table),
 * search statistics and kernel timing. The matchers themselves: csm_window.hip (one window at a time),
 * csm_batch.hip (batches), on csm_plan.hip (planner, launch helpers, levels); kernels behind
 * csm_launch.hpp, csm_joint.hpp, csm_phase.hpp.
 *
 * Host-side expressions that must agree bit for bit with the reference
 * (search step, window, projection, pose algebra) are restated here from the
 * cited reference lines and are built with -ffp-contract=off.
 */
#include "csm_matchers.hpp"


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

