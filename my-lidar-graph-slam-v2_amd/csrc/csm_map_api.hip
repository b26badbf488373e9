/* csm_map_api.hip -- host side of the map updates: csm_construct_map_from_scans
 * and csm_update_map_with_scan (include/csm_hip.h), with their kernels
 * (csm_map_kernels.hip). A translation unit of libcsm_hip.so of its own. */
#include "csm_internal.hpp"

#include "csm_map_kernels.hip"

extern "C" {

/* ---- map building ---- */

namespace {

/* GridBinaryBayes's conversions (src/grid_map_new/grid_binary_bayes.cpp:345-383,
 * inc/grid_map_new/grid_values.hpp:11-46) with its constants: values 1..65535
 * stand for probabilities 0.001..0.999, 0 = unknown. */
constexpr uint32_t kMapUncCap = 4096;    /* beams listed for exact recomputation per build */
const double kBayesProbMin = 1e-3;
const double kBayesProbMax = 1.0 - 1e-3;

double bayes_probability_to_odds(double prob)
{
    if (prob == 0.0)
        return 1.0;
    if (prob < kBayesProbMin)
        return kBayesProbMin / (1.0 - kBayesProbMin);
    if (prob > kBayesProbMax)
        return kBayesProbMax / (1.0 - kBayesProbMax);
    return prob / (1.0 - prob);
}

uint16_t bayes_value_after(uint32_t value, double odds)
{
    /* GridBinaryBayes::UpdateOddsUnchecked (grid_binary_bayes.cpp:302-321) */
    double now = odds;
    if (value != 0) {
        const double p = kBayesProbMin + (kBayesProbMax - kBayesProbMin) *
                         static_cast<double>(static_cast<int>(value) - 1) / 65534.0;
        now = (p / (1.0 - p)) * odds;
    }
    double prob = 0.0;
    if (!(now < 0.0))
        prob = std::min(std::max(now / (1.0 + now), kBayesProbMin), kBayesProbMax);
    if (prob == 0.0)
        return 0;
    if (prob < kBayesProbMin)
        return 1;
    if (prob > kBayesProbMax)
        return 65535;
    return static_cast<uint16_t>(1 + (prob - kBayesProbMin) * 65534.0 / (kBayesProbMax - kBayesProbMin));
}

/* GridMap<T>::IndexToBlock (src/grid_map_new/grid_map.cpp:804-814): a negative
 * index lands one block further out than a floor would put it */
int map_index_to_block(int idx, int log2_block)
{
    return idx >= 0 ? (idx >> log2_block) : ((idx >> log2_block) - 1);
}

} /* namespace */

/* GridMap<T>::Resize(BoundingBox<int>) (src/grid_map_new/grid_map.cpp:841-889) and, with
 * expand != 0, GridMap<T>::Expand (:915-936) in front of it, on index boxes: host only. */
int csm_host_map_resize(csm_map_shape* shape, const int32_t box[4], int32_t expand, int32_t shift_out[2])
{
    if (!shape || !box || shape->log2_block_size < 0 || shape->log2_block_size > 12 ||
        !(shape->resolution > 0.0))
        return CSM_EINVAL;
    const int lb = shape->log2_block_size, block = 1 << lb;
    long long i_min_x = box[0], i_min_y = box[1], i_max_x = (long long)box[2] + 1, i_max_y = (long long)box[3] + 1;
    if (shift_out)
        shift_out[0] = shift_out[1] = 0;
    if (i_min_x >= i_max_x || i_min_y >= i_max_y)
        return CSM_EINVAL;                  /* the reference asserts */
    if (expand) {
        auto inside = [shape](long long row, long long col) {
            return row >= 0 && row < shape->rows && col >= 0 && col < shape->cols;
        };
        if (inside(i_min_y, i_min_x) && inside(i_max_y - 1, i_max_x - 1))
            return CSM_OK;                  /* the box fits: nothing changes */
        i_min_x = std::min(0ll, i_min_x);
        i_min_y = std::min(0ll, i_min_y);
        i_max_x = std::max((long long)shape->cols, i_max_x);
        i_max_y = std::max((long long)shape->rows, i_max_y);
    }
    if (i_min_x < -(1ll << 30) || i_min_y < -(1ll << 30) || i_max_x > (1ll << 30) || i_max_y > (1ll << 30))
        return CSM_EINVAL;
    const int b_min_x = map_index_to_block((int)i_min_x, lb), b_min_y = map_index_to_block((int)i_min_y, lb);
    const int b_max_x = map_index_to_block((int)i_max_x + block - 1, lb);
    const int b_max_y = map_index_to_block((int)i_max_y + block - 1, lb);
    const long long rows = (long long)(b_max_y - b_min_y) << lb, cols = (long long)(b_max_x - b_min_x) << lb;
    if (rows < 1 || cols < 1 || rows * cols > (1ll << 28))
        return CSM_EINVAL;
    shape->rows = (int32_t)rows;
    shape->cols = (int32_t)cols;
    /* GridMapGeometry::Resize (src/grid_map_new/grid_map_geometry.cpp:61-72) */
    shape->offset_x += shape->resolution * (b_min_x * block);      /* b_min may be negative: no shift */
    shape->offset_y += shape->resolution * (b_min_y * block);
    if (shift_out) {
        shift_out[0] = b_min_y * block;
        shift_out[1] = b_min_x * block;
    }
    return CSM_OK;
}

/* Both map updates of GridMapBuilder. keep_cells = false: ConstructMapFromScans
 * (src/mapping/grid_map_builder.cpp:561-695): resize to the scans' bounding box,
 * reset, integrate. keep_cells = true: UpdateGridMap (:389-494): expand only if
 * the scan does not fit, keep the cells, integrate one scan on top. */
static int map_build(csm_ctx* ctx, uint64_t map_id, csm_map_shape* shape,
                     const double global_map_pose[3], const csm_scan_node* nodes,
                     int32_t n_nodes, const csm_map_builder_params* prm,
                     csm_map_build_info* info, bool keep_cells)
{
    if (!ctx || !shape || !global_map_pose || !nodes || n_nodes < 1 || !prm ||
        !(shape->resolution > 0.0) || shape->log2_block_size < 0 || shape->log2_block_size > 12 ||
        prm->subpixel_scale < 1 || prm->subpixel_scale > 1024)
        return fail(ctx, CSM_EINVAL, "map build: bad arguments");
    if (keep_cells) {
        const DeviceGrid* have = find_grid(ctx, map_id);
        if (!have || have->levels.empty())
            return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
        if (have->rows != shape->rows || have->cols != shape->cols)
            return fail(ctx, CSM_EINVAL, "shape %d x %d does not match the resident map %d x %d",
                        shape->rows, shape->cols, have->rows, have->cols);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    const int scale = prm->subpixel_scale;
    const double res = shape->resolution;
    const double scaled_res = res / scale;                  /* ScaledGeometry, grid_map_geometry.cpp:46-58 */
    auto to_index = [res](double p, double off) { return static_cast<int>(std::floor((p - off) / res)); };

    /* grid_map_builder.cpp:583-612: sensor poses, usable ranges */
    std::vector<MapNode> table(n_nodes);
    long long n_beams_ll = 0, usable = 0;
    for (int k = 0; k < n_nodes; ++k) {
        const csm_scan_node& nd = nodes[k];
        if (!nd.scan.angles || !nd.scan.ranges || nd.scan.n_points < 0)
            return fail(ctx, CSM_EINVAL, "scan node %d has no scan", k);
        double global_sensor[3], local_sensor[3];
        csm_host_compound(nd.global_pose, nd.scan.relative_sensor_pose, global_sensor);
        csm_host_inverse_compound(global_map_pose, global_sensor, local_sensor);
        MapNode& t = table[k];
        t.x = local_sensor[0];
        t.y = local_sensor[1];
        t.theta = local_sensor[2];
        t.min_range = std::max(prm->usable_range_min, nd.min_range);
        t.max_range = std::min(prm->usable_range_max, nd.max_range);
        t.beam_base = (int32_t)n_beams_ll;
        t.n_beams = nd.scan.n_points;
        t.sx = t.sy = 0;
        n_beams_ll += nd.scan.n_points;
        for (int i = 0; i < nd.scan.n_points; ++i) {
            const double r = nd.scan.ranges[i];
            usable += !(r >= t.max_range || r <= t.min_range);
        }
    }
    if (n_beams_ll > (1ll << 24))
        return fail(ctx, CSM_EINVAL, "%lld beams in one map build", n_beams_ll);
    const int n_beams = (int)n_beams_ll;
    const int n_rays = n_beams;             /* a ray's number = its beam's place in the update order */

    int rc = 0;
    if ((rc = ensure(ctx, ctx->m_rays, (size_t)std::max(n_rays, 1) * sizeof(MapRay) +
                                           (size_t)n_nodes * sizeof(MapNode) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->m_recs, (size_t)std::max(n_rays, 1) * sizeof(MapRayRec)))) return rc;
    /* per hit cell at most 3n + 7 words (csm_device.hpp: map_block_words), then the hit-cell list */
    const size_t list_words = 10 * (size_t)n_rays + 16;
    if ((rc = ensure(ctx, ctx->m_lists, (list_words + (size_t)n_rays + 4) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->m_cnt, kMapCounters * sizeof(unsigned long long) + 64 + kMapUncCap * 4))) return rc;
    MapRay* d_rays = reinterpret_cast<MapRay*>(ctx->m_rays.p);
    MapNode* d_nodes = reinterpret_cast<MapNode*>(d_rays + std::max(n_rays, 1));
    unsigned long long* d_counters = reinterpret_cast<unsigned long long*>(ctx->m_cnt.p);
    int32_t* d_box = reinterpret_cast<int32_t*>(d_counters + kMapCounters);   /* [4] + count + spread */
    uint32_t* d_unc = reinterpret_cast<uint32_t*>(d_box + 4);                  /* [0] count, [1] spread bits */
    uint32_t* d_unc_list = d_unc + 4;

    /* ---- hit points + bounding box (grid_map_builder.cpp:614-638) ----
     * In index form: Resize(BoundingBox<double>) (grid_map.cpp:892-913) takes
     * floor((min - res - off) / res) and floor((max + res - off) / res), and that
     * expression is monotone, so the box is the min / max of it over the points. */
    int box[4] = { 0x7fffffff, 0x7fffffff, -0x7fffffff - 1, -0x7fffffff - 1 };
    double min_x = std::numeric_limits<double>::max(), min_y = min_x;
    double max_x = std::numeric_limits<double>::min(), max_y = max_x;   /* as the reference: smallest positive */
    if (keep_cells) {
        /* ComputeBoundingBoxAndScanPointsMapLocal starts from the sensor position (:835-841) */
        min_x = max_x = table[0].x;
        min_y = max_y = table[0].y;
    }
    auto add_point = [&](double x, double y) {
        min_x = std::min(min_x, x);
        min_y = std::min(min_y, y);
        max_x = std::max(max_x, x);
        max_y = std::max(max_y, y);
    };
    for (const MapNode& t : table)
        add_point(t.x, t.y);
    bool device_projection = n_beams > 0 && !ctx->tune.map_host_projection;
    uint32_t unc_cap = kMapUncCap;
    if (ctx->tune.map_unc_cap > 0)          /* csm_config.map_uncertain_cap: tests of the overflow path */
        unc_cap = (uint32_t)std::min<long>(ctx->tune.map_unc_cap, kMapUncCap);
    bool spread_known = false;              /* the box of the certified beams is certainly not degenerate */
    if (device_projection) {
        /* scans to the device (one staging copy), projection there */
        std::vector<double> stage(2 * (size_t)n_beams);
        for (int k = 0; k < n_nodes; ++k) {
            std::memcpy(stage.data() + table[k].beam_base, nodes[k].scan.angles,
                        (size_t)table[k].n_beams * sizeof(double));
            std::memcpy(stage.data() + n_beams + table[k].beam_base, nodes[k].scan.ranges,
                        (size_t)table[k].n_beams * sizeof(double));
        }
        if ((rc = ensure(ctx, ctx->scan_dev, stage.size() * sizeof(double)))) return rc;
        double* d_scan = reinterpret_cast<double*>(ctx->scan_dev.p);
        const int32_t init_box[8] = { box[0], box[1], box[2], box[3], 0, 0, 0, 0 };
        HIP_TRY(ctx, hipMemcpyAsync(d_scan, stage.data(), stage.size() * sizeof(double),
                                    hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_nodes, table.data(), table.size() * sizeof(MapNode),
                                    hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_box, init_box, sizeof(init_box), hipMemcpyHostToDevice, ctx->stream));
        MapProjJob pj;
        std::memset(&pj, 0, sizeof(pj));
        pj.angles = d_scan;
        pj.ranges = d_scan + n_beams;
        pj.nodes = d_nodes;
        pj.n_nodes = n_nodes;
        pj.n_beams = n_beams;
        pj.rays = d_rays;
        pj.off_x = shape->offset_x;
        pj.off_y = shape->offset_y;
        pj.res = res;
        pj.scaled_res = scaled_res;
        pj.box = d_box;
        pj.unc_count = d_unc;
        pj.unc_list = d_unc_list;
        pj.unc_cap = unc_cap;
        {
            ScopedTimer tm(ctx, "map_project");
            hipLaunchKernelGGL(k_map_project, dim3((unsigned)ceil_div(n_beams, 256)), dim3(256), 0, ctx->stream, pj);
        }
        HIP_TRY(ctx, hipGetLastError());
        int32_t got[8];
        HIP_TRY(ctx, hipMemcpyAsync(got, d_box, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t n_unc = (uint32_t)got[4];
        spread_known = ((uint32_t)got[5] & 3u) == 3u;
        if (n_unc > unc_cap || !spread_known) {
            device_projection = false;      /* too many beams on cell edges, or a degenerate box: all on the host */
        } else {
            for (int k = 0; k < 4; ++k)
                box[k] = got[k];
            if (n_unc) {
                /* the beams the device could not certify: exactly as the reference, and patched in */
                std::vector<uint32_t> list(n_unc);
                std::vector<MapRay> exact(n_unc);
                HIP_TRY(ctx, hipMemcpy(list.data(), d_unc_list, (size_t)n_unc * 4, hipMemcpyDeviceToHost));
                for (uint32_t u = 0; u < n_unc; ++u) {
                    const uint32_t b = list[u];
                    int k = 0;
                    while (k + 1 < n_nodes && table[k + 1].beam_base <= (int32_t)b)
                        ++k;
                    const int i = (int)b - table[k].beam_base;
                    const double r = nodes[k].scan.ranges[i];
                    MapRay& ray = exact[u];
                    ray.hx = table[k].x + r * std::cos(table[k].theta + nodes[k].scan.angles[i]);
                    ray.hy = table[k].y + r * std::sin(table[k].theta + nodes[k].scan.angles[i]);
                    ray.node = k;
                    ray.usable = 1;
                    add_point(ray.hx, ray.hy);
                    HIP_TRY(ctx, hipMemcpyAsync(d_rays + b, &ray, sizeof(ray), hipMemcpyHostToDevice, ctx->stream));
                }
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   /* `exact` goes out of scope */
            }
        }
    }
    if (!device_projection) {
        /* host projection (ScanData::HitPoint, inc/sensor/sensor_data.hpp:189-203) */
        std::vector<MapRay> rays((size_t)std::max(n_rays, 1));
        for (int k = 0; k < n_nodes; ++k) {
            const MapNode& t = table[k];
            for (int i = 0; i < t.n_beams; ++i) {
                MapRay& ray = rays[(size_t)t.beam_base + i];
                ray.hx = ray.hy = 0.0;
                ray.node = k;
                ray.usable = 0;
                const double r = nodes[k].scan.ranges[i];
                if (r >= t.max_range || r <= t.min_range)
                    continue;
                ray.hx = t.x + r * std::cos(t.theta + nodes[k].scan.angles[i]);
                ray.hy = t.y + r * std::sin(t.theta + nodes[k].scan.angles[i]);
                ray.usable = 1;
                add_point(ray.hx, ray.hy);
            }
        }
        for (int k = 0; k < 4; ++k)
            box[k] = k < 2 ? 0x7fffffff : -0x7fffffff - 1;
        if (n_rays) {
            HIP_TRY(ctx, hipMemcpyAsync(d_rays, rays.data(), (size_t)n_rays * sizeof(MapRay),
                                        hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       /* `rays` goes out of scope */
        }
        spread_known = false;
    }
    /* Assert(min < max) of Resize: the host-side points decide unless the certified
     * beams are known to spread in both axes */
    if (!spread_known && (!(min_x < max_x) || !(min_y < max_y)))
        return fail(ctx, CSM_EINVAL, "empty bounding box (the reference asserts)");
    if (min_x <= max_x) {                   /* points the host holds as doubles (always: the sensors) */
        box[0] = std::min(box[0], to_index(min_x - res, shape->offset_x));
        box[1] = std::min(box[1], to_index(min_y - res, shape->offset_y));
        box[2] = std::max(box[2], to_index(max_x + res, shape->offset_x));
        box[3] = std::max(box[3], to_index(max_y + res, shape->offset_y));
    }

    /* GridMap::Resize(BoundingBox<int>) / GridMap::Expand on the CURRENT geometry */
    csm_map_shape next = *shape;
    int32_t shift[2] = { 0, 0 };            /* first row / column of the new map in the old frame */
    if (csm_host_map_resize(&next, box, keep_cells ? 1 : 0, shift) != CSM_OK)
        return fail(ctx, CSM_EINVAL, "resized map is out of range");
    const bool resized = !keep_cells || shift[0] != 0 || shift[1] != 0 || next.rows != shape->rows ||
                         next.cols != shape->cols;
    const int rows = next.rows, cols = next.cols;
    const double off_x = next.offset_x, off_y = next.offset_y;
    for (MapNode& t : table) {
        t.sx = static_cast<int>(std::floor((t.x - off_x) / scaled_res));
        t.sy = static_cast<int>(std::floor((t.y - off_y) / scaled_res));
    }
    const size_t n_cells = (size_t)rows * cols;

    /* the two value -> value tables of the cell update */
    if (!ctx->m_apply_attr) {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_map_apply_hits),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         65536 * (int)sizeof(uint16_t)));
        ctx->m_apply_attr = true;
    }
    if ((rc = ensure(ctx, ctx->m_lut, 2 * 65536 * sizeof(uint16_t)))) return rc;
    uint16_t* d_lut = reinterpret_cast<uint16_t*>(ctx->m_lut.p);
    if (ctx->m_lut_hit != prm->prob_hit || ctx->m_lut_miss != prm->prob_miss) {
        std::vector<uint16_t> tab(2 * 65536);
        const double odds_hit = bayes_probability_to_odds(prm->prob_hit);     /* grid_map_builder.cpp:95-96 */
        const double odds_miss = bayes_probability_to_odds(prm->prob_miss);
        for (uint32_t v = 0; v < 65536; ++v) {
            tab[v] = bayes_value_after(v, odds_hit);
            tab[65536 + v] = bayes_value_after(v, odds_miss);
        }
        HIP_TRY(ctx, hipMemcpyAsync(d_lut, tab.data(), tab.size() * 2, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->m_lut_hit = prm->prob_hit;
        ctx->m_lut_miss = prm->prob_miss;
    }

    /* the destination grid: keep the old allocation when it is large enough */
    DeviceGrid& g = ctx->grids[map_id];
    const int pitch = (cols + 7) & ~7;
    const size_t bytes = (size_t)rows * pitch * 2;
    if (keep_cells && resized) {
        /* GridMap::Resize moves the blocks (grid_map.cpp:866-879): the old cells, shifted */
        Level base;
        const size_t want = bytes + bytes / 2;
        if (hipMalloc(reinterpret_cast<void**>(&base.cells), want) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", want);
        base.win = 1;
        base.owned = true;
        base.cap = want;
        const int shift_r = -shift[0], shift_c = -shift[1];
        HIP_TRY(ctx, hipMemsetAsync(base.cells, 0, bytes, ctx->stream));
        HIP_TRY(ctx, hipMemcpy2DAsync(base.cells + (size_t)shift_r * pitch + shift_c, (size_t)pitch * 2,
                                      g.levels[0].cells, (size_t)g.pitch * 2, (size_t)g.cols * 2, g.rows,
                                      hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (g.levels[0].owned)
            (void)hipFree(g.levels[0].cells);
        g.levels[0] = base;
    } else if (!keep_cells && (g.levels.empty() || !g.levels[0].owned || g.levels[0].cap < bytes)) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        free_levels(g, false);
        Level base;
        const size_t want = bytes + bytes / 2;
        if (hipMalloc(reinterpret_cast<void**>(&base.cells), want) != hipSuccess) {
            ctx->grids.erase(map_id);
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", want);
        }
        base.win = 1;
        base.owned = true;
        base.cap = want;
        g.levels.push_back(base);
    }
    for (size_t i = 1; i < g.levels.size(); ++i) {
        if (g.levels[i].owned)
            g.levels[i].stale = true;
        else
            g.levels[i].cells = g.levels[0].cells;     /* an alias of the base (window 1) */
    }
    g.xg_stale = true;         /* the pair-row copy follows the base */
    g.alloc_stale = true;      /* and so does the derived block-allocation bitmap */
    g.alloc_user = false;
    g.rows = rows;
    g.cols = cols;
    g.pitch = pitch;
    g.known_r0 = 0;
    g.known_c0 = 0;

    if ((rc = ensure(ctx, ctx->m_cell, 3 * n_cells * sizeof(uint32_t)))) return rc;
    MapJob mj;
    std::memset(&mj, 0, sizeof(mj));
    mj.rays = d_rays;
    mj.nodes = d_nodes;
    mj.recs = reinterpret_cast<MapRayRec*>(ctx->m_recs.p);
    mj.n_rays = n_rays;
    mj.off_x = off_x;
    mj.off_y = off_y;
    mj.res = res;
    mj.scaled_res = scaled_res;
    mj.scale = scale;
    mj.rows = rows;
    mj.cols = cols;
    mj.pitch = pitch;
    mj.n_hit = reinterpret_cast<uint32_t*>(ctx->m_cell.p);
    mj.n_miss = mj.n_hit + n_cells;
    mj.seg = mj.n_miss + n_cells;
    mj.lists = reinterpret_cast<uint32_t*>(ctx->m_lists.p);
    mj.hit_cells = mj.lists + list_words;
    mj.counters = d_counters;
    mj.lut_hit = d_lut;
    mj.lut_miss = d_lut + 65536;
    mj.cells = g.levels[0].cells;
    mj.keep_cells = keep_cells ? 1 : 0;
    unsigned long long counters[kMapCounters] = { 0 };
    counters[kMapKnownRow] = counters[kMapKnownCol] = ~0ull;
    const auto t1 = std::chrono::steady_clock::now();
    if (info && !ctx->m_ev[0]) {
        HIP_TRY(ctx, hipEventCreate(&ctx->m_ev[0]));
        HIP_TRY(ctx, hipEventCreate(&ctx->m_ev[1]));
    }
    const hipEvent_t ev_a = ctx->m_ev[0], ev_b = ctx->m_ev[1];
    if (info)
        HIP_TRY(ctx, hipEventRecord(ev_a, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_nodes, table.data(), table.size() * sizeof(MapNode),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mj.counters, counters, sizeof(counters), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(mj.n_hit, 0, 2 * n_cells * sizeof(uint32_t), ctx->stream));
    {
        ScopedTimer tm(ctx, "map_build");
        const unsigned ray_blocks = (unsigned)ceil_div(std::max(n_rays, 1), 256);
        const unsigned cell_blocks = (unsigned)((n_cells + 255) / 256);
        if (n_rays) {
            hipLaunchKernelGGL(k_map_hits, dim3(ray_blocks), dim3(256), 0, ctx->stream, mj);
            hipLaunchKernelGGL(k_map_alloc, dim3(cell_blocks), dim3(256), 0, ctx->stream, mj);
            hipLaunchKernelGGL(k_map_fill_hits, dim3(ray_blocks), dim3(256), 0, ctx->stream, mj);
            hipLaunchKernelGGL(k_map_rank_hits, dim3(ray_blocks), dim3(256), 0, ctx->stream, mj);
            hipLaunchKernelGGL(k_map_walk, dim3((unsigned)ceil_div(n_rays, kMapGroup)), dim3(512), 0, ctx->stream, mj);
        }
        hipLaunchKernelGGL(k_map_apply, dim3((unsigned)(((size_t)rows * pitch + 255) / 256)), dim3(256), 0,
                           ctx->stream, mj);
        if (usable > 0) {
            /* one workgroup per CU at most; the kernel spreads the cells with hits over
             * their wavefronts (each has at least one usable ray) */
            const unsigned wgs = (unsigned)std::min<long long>(
                256, ceil_div((int)std::min<long long>(usable, (long long)n_cells), 4));
            hipLaunchKernelGGL(k_map_apply_hits, dim3(wgs), dim3(256), 65536 * sizeof(uint16_t), ctx->stream, mj);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    if (info)
        HIP_TRY(ctx, hipEventRecord(ev_b, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(counters, mj.counters, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float dev_ms = 0.f;
    if (info)
        (void)hipEventElapsedTime(&dev_ms, ev_a, ev_b);
    if (counters[kMapError]) {
        free_levels(g, false);          /* the cells may be half updated: drop the map */
        ctx->grids.erase(map_id);
        return fail(ctx, CSM_EINVAL, "a ray leaves the resized map (flags %llu): the reference asserts",
                    counters[kMapError]);
    }
    g.known_r0 = counters[kMapKnownRow] == ~0ull ? rows : (int)counters[kMapKnownRow];
    g.known_c0 = counters[kMapKnownCol] == ~0ull ? cols : (int)counters[kMapKnownCol];
    shape->rows = rows;
    shape->cols = cols;
    shape->offset_x = off_x;
    shape->offset_y = off_y;
    if (info) {
        info->rays = usable;
        info->cell_updates = info->saturated_reads = 0;
        for (int k = 0; k < kMapStripes; ++k) {
            info->cell_updates += (int64_t)counters[kMapStripedUpdates + k];
            info->saturated_reads += (int64_t)counters[kMapStripedSaturated + k];
        }
        info->first_known_row = g.known_r0;
        info->first_known_col = g.known_c0;
        info->device_projection = device_projection ? 1 : 0;
        info->host_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        info->device_us = dev_ms * 1e3;
    }
    return CSM_OK;
}

/* GridMapBuilder::ConstructMapFromScans (src/mapping/grid_map_builder.cpp:561-695) */
int csm_construct_map_from_scans(csm_ctx* ctx, uint64_t map_id, csm_map_shape* shape,
                                 const double global_map_pose[3], const csm_scan_node* nodes,
                                 int32_t n_nodes, const csm_map_builder_params* prm,
                                 csm_map_build_info* info)
{
    return map_build(ctx, map_id, shape, global_map_pose, nodes, n_nodes, prm, info, false);
}

/* the grid half of GridMapBuilder::UpdateGridMap (src/mapping/grid_map_builder.cpp:389-494) */
int csm_update_map_with_scan(csm_ctx* ctx, uint64_t map_id, csm_map_shape* shape,
                             const double global_map_pose[3], const csm_scan_node* node,
                             const csm_map_builder_params* prm, csm_map_build_info* info)
{
    return map_build(ctx, map_id, shape, global_map_pose, node, node ? 1 : 0, prm, info, true);
}

} /* extern "C" */
