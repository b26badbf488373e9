/* csm_cost_api.hip -- host side of the batched cost / covariance / linear-solver
 * entry points, with their kernels (csm_cost_kernels.hip). A translation unit of
 * libcsm_hip.so of its own. */
#include "csm_internal.hpp"

#include "csm_cost_kernels.hip"


namespace {

const int kDefaultLog2Block = 4;    /* "PatchSize": 16, launcher_settings_default.json:178 */

/* the allocation bitmap the cost function's ProbabilityOr(.., 0.5) needs */
int ensure_allocation(csm_ctx* ctx, DeviceGrid& g)
{
    if (g.alloc && (g.alloc_user || !g.alloc_stale))
        return CSM_OK;
    const int log2b = g.alloc_log2 > 0 ? g.alloc_log2 : kDefaultLog2Block;
    const int brows = (g.rows + (1 << log2b) - 1) >> log2b, bcols = (g.cols + (1 << log2b) - 1) >> log2b;
    const size_t bytes = (size_t)brows * bcols;
    if (bytes > g.alloc_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (g.alloc)
            (void)hipFree(g.alloc);
        g.alloc = nullptr;
        g.alloc_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&g.alloc), bytes + 64) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
        g.alloc_cap = bytes + 64;
    }
    HIP_TRY(ctx, hipMemsetAsync(g.alloc, 0, bytes, ctx->stream));
    hipLaunchKernelGGL(k_block_allocation, dim3((unsigned)bytes), dim3(256), 0, ctx->stream,
                       g.levels[0].cells, g.rows, g.cols, g.pitch, log2b, bcols, g.alloc);
    HIP_TRY(ctx, hipGetLastError());
    g.alloc_log2 = log2b;
    g.alloc_bcols = bcols;
    g.alloc_user = false;
    g.alloc_stale = false;
    return CSM_OK;
}

/* poses: n x 3 sensor poses (sensor_given) or null = Compound(query initial pose, relative sensor pose) */
int run_cost_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n, const double* sensor_poses,
                   const csm_refine_params* prm, bool refine, csm_refine_result* out)
{
    if (!ctx || !queries || n < 1 || !prm || !out || !(prm->covariance_scale > 0.0) ||
        (refine && (prm->iterations_max < 1 || !(prm->convergence_threshold >= 0.0))))
        return fail(ctx, CSM_EINVAL, "bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    size_t scan_total = 0;
    std::vector<size_t> scan_off((size_t)n);
    for (int i = 0; i < n; ++i) {
        const csm_loop_query& q = queries[i];
        if (!q.scan.angles || !q.scan.ranges || q.scan.n_points < 1 || !scan_is_finite(&q.scan))
            return fail(ctx, CSM_EINVAL, "query %d: empty scan or non-finite beam", i);
        DeviceGrid* g = find_grid(ctx, q.map_id);
        if (!g)
            return fail(ctx, CSM_ENOENT, "query %d: map %llu not resident", i, (unsigned long long)q.map_id);
        if ((rc = ensure_allocation(ctx, *g)))
            return rc;
        scan_off[i] = scan_total;
        scan_total += 2 * (size_t)q.scan.n_points;
    }
    if ((rc = ensure(ctx, ctx->c_scans, scan_total * 8 + 64))) return rc;
    if ((rc = ensure(ctx, ctx->c_jobs, (size_t)n * (sizeof(CostJob) + sizeof(CostOut)) + 256))) return rc;
    /* host staging owned by the context: the sources of the asynchronous uploads stay alive */
    ctx->c_stage.resize(scan_total);
    ctx->c_job_stage.resize((size_t)n);
    double* d_scans = reinterpret_cast<double*>(ctx->c_scans.p);
    CostJob* d_jobs = reinterpret_cast<CostJob*>(ctx->c_jobs.p);
    CostOut* d_out = reinterpret_cast<CostOut*>(d_jobs + n);
    for (int i = 0; i < n; ++i) {
        const csm_loop_query& q = queries[i];
        const int np = q.scan.n_points;
        std::memcpy(ctx->c_stage.data() + scan_off[i], q.scan.angles, (size_t)np * 8);
        std::memcpy(ctx->c_stage.data() + scan_off[i] + np, q.scan.ranges, (size_t)np * 8);
        const DeviceGrid& g = *find_grid(ctx, q.map_id);
        CostJob& J = ctx->c_job_stage[i];
        std::memset(&J, 0, sizeof(J));
        J.cells = g.levels[0].cells;
        J.rows = g.rows;
        J.cols = g.cols;
        J.pitch = g.pitch;
        J.alloc = g.alloc;
        J.log2_block = g.alloc_log2;
        J.block_cols = g.alloc_bcols;
        J.res = q.geometry.resolution;
        J.off_x = q.geometry.offset_x;
        J.off_y = q.geometry.offset_y;
        J.angles = d_scans + scan_off[i];
        J.ranges = d_scans + scan_off[i] + np;
        J.n = np;
        J.iterations_max = refine ? prm->iterations_max : 0;
        if (sensor_poses) {
            for (int k = 0; k < 3; ++k)
                J.sensor_pose[k] = sensor_poses[3 * i + k];
        } else {
            /* scan_matcher_linear_solver.cpp:78-81 */
            csm_host_compound(q.initial_pose, q.scan.relative_sensor_pose, J.sensor_pose);
        }
        J.convergence_threshold = prm->convergence_threshold;
        J.lambda = prm->lambda;
        J.covariance_scale = prm->covariance_scale;
        J.lut = ctx->lut_dev;
        J.out = d_out + i;
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_scans, ctx->c_stage.data(), scan_total * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_jobs, ctx->c_job_stage.data(), (size_t)n * sizeof(CostJob),
                                hipMemcpyHostToDevice, ctx->stream));
    {
        ScopedTimer tm(ctx, "cost_refine");
        hipLaunchKernelGGL(k_cost_refine, dim3(n), dim3(kCostBlock), 0, ctx->stream, d_jobs);
        HIP_TRY(ctx, hipGetLastError());
    }
    std::vector<CostOut> res((size_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(res.data(), d_out, (size_t)n * sizeof(CostOut), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) {
        const csm_loop_query& q = queries[i];
        const CostOut& o = res[i];
        csm_refine_result& r = out[i];
        std::memset(&r, 0, sizeof(r));
        const double np = static_cast<double>(q.scan.n_points);
        r.normalized_initial_cost = o.initial_cost / np;
        r.normalized_cost = o.cost / np;
        for (int k = 0; k < 3; ++k) {
            r.sensor_pose[k] = ctx->c_job_stage[i].sensor_pose[k];
            r.best_sensor_pose[k] = o.best_sensor_pose[k];
        }
        /* scan_matcher_linear_solver.cpp:118-119 / scan_matcher_correlative.cpp:214-216 */
        csm_host_move_backward(r.best_sensor_pose, q.scan.relative_sensor_pose, r.estimated_pose);
        for (int k = 0; k < 9; ++k) {
            r.covariance[k] = o.covariance[k];
            r.hessian[k] = o.hessian[k];
        }
        r.lambda = o.lambda;
        r.iterations = o.iterations;
    }
    return CSM_OK;
}

} /* namespace */

extern "C" {

int csm_set_block_allocation(csm_ctx* ctx, uint64_t map_id, int32_t log2_block_size, const uint8_t* allocated)
{
    if (!ctx || log2_block_size < 0 || log2_block_size > 12)
        return fail(ctx, CSM_EINVAL, "csm_set_block_allocation: bad arguments");
    DeviceGrid* g = find_grid(ctx, map_id);
    if (!g)
        return fail(ctx, CSM_ENOENT, "map %llu not resident", (unsigned long long)map_id);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    g->alloc_log2 = log2_block_size;
    g->alloc_user = false;
    g->alloc_stale = true;
    if (!allocated)
        return CSM_OK;              /* back to the rule "a block with a known cell is allocated" */
    const int bs = 1 << log2_block_size;
    const int brows = (g->rows + bs - 1) >> log2_block_size, bcols = (g->cols + bs - 1) >> log2_block_size;
    const size_t bytes = (size_t)brows * bcols;
    if (bytes > g->alloc_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (g->alloc)
            (void)hipFree(g->alloc);
        g->alloc = nullptr;
        g->alloc_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&g->alloc), bytes + 64) != hipSuccess)
            return fail(ctx, CSM_ENOMEM, "hipMalloc(%zu) failed", bytes);
        g->alloc_cap = bytes + 64;
    }
    HIP_TRY(ctx, hipMemcpyAsync(g->alloc, allocated, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    g->alloc_bcols = bcols;
    g->alloc_user = true;
    g->alloc_stale = false;
    return CSM_OK;
}

int csm_cost_covariance_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n_queries,
                              const double* sensor_poses, double covariance_scale, csm_refine_result* out)
{
    if (!sensor_poses)
        return fail(ctx, CSM_EINVAL, "csm_cost_covariance_batch: bad arguments");
    csm_refine_params prm {};
    prm.covariance_scale = covariance_scale;
    return run_cost_batch(ctx, queries, n_queries, sensor_poses, &prm, false, out);
}

int csm_linear_solver_batch(csm_ctx* ctx, const csm_loop_query* queries, int32_t n_queries,
                            const csm_refine_params* params, csm_refine_result* out)
{
    return run_cost_batch(ctx, queries, n_queries, nullptr, params, true, out);
}

} /* extern "C" */
