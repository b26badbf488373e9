/* csm_cost_kernels.hip -- the step that follows every search in the reference,
 * batched on the device: CostSquareError (Cost, ComputeHessianAndResidual,
 * ComputeCovariance; src/mapping/cost_function_square_error.cpp:48-195, 232-341)
 * and the ScanMatcherLinearSolver refinement
 * (src/mapping/scan_matcher_linear_solver.cpp:66-169). One workgroup per
 * query; lanes over beams; f64 throughout.
 *
 * NOT bit-exact (include/csm_hip.h states the tolerance): the hit points use
 * the device's sin / cos, the sums over beams are tree reductions instead of
 * the reference's sequential loop, and the two Eigen calls (3x3 inverse,
 * column-pivoting Householder QR) are restated from their algorithms.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csm_device.hpp"

namespace csm {

constexpr int kCostBlock = 256;

/* allocated[b] = 1 iff block b holds a known (non-zero) cell: exactly the
 * reference's allocation state for a map that was only ever updated (every
 * update leaves a value >= 1 and allocates the block on the way,
 * src/grid_map_new/grid_map.cpp:514-535). */
__global__ __launch_bounds__(256) void k_block_allocation(const uint16_t* __restrict__ cells, int rows, int cols,
                                                         int pitch, int log2_block, int block_cols,
                                                         uint8_t* __restrict__ allocated)
{
    const int b = blockIdx.x;
    const int br = b / block_cols, bc = b % block_cols;
    const int bs = 1 << log2_block;
    bool any = false;
    for (int i = threadIdx.x; i < bs * bs; i += 256) {
        const int r = (br << log2_block) + (i >> log2_block), c = (bc << log2_block) + (i & (bs - 1));
        if (r < rows && c < cols && cells[(size_t)r * pitch + c] != 0)
            any = true;
    }
    if (__syncthreads_or(any) && threadIdx.x == 0)
        allocated[b] = 1;
}

struct CostTerms {
    double cost, h00, h01, h02, h11, h12, h22, r0, r1, r2;
};

/* GridMap::ProbabilityOr(row, col, 0.5): src/grid_map_new/grid_map.cpp:423-436 */
__device__ __forceinline__ double probability_or_half(const CostJob& j, int row, int col)
{
    if (row < 0 || row >= j.rows || col < 0 || col >= j.cols)
        return 0.5;
    if (j.alloc && !j.alloc[(row >> j.log2_block) * j.block_cols + (col >> j.log2_block)])
        return 0.5;
    return j.lut[j.cells[(size_t)row * j.pitch + col]];
}

/* one beam's contribution at sensor pose (px, py, pt) */
__device__ __forceinline__ void beam_terms(const CostJob& j, double px, double py, double pt, int i,
                                           bool want_hessian, CostTerms& t)
{
    /* ScanData::HitPoint, inc/sensor/sensor_data.hpp:189-203 */
    const double a = pt + j.angles[i], rg = j.ranges[i];
    const double hx = px + rg * cos(a), hy = py + rg * sin(a);
    /* GetClosestMapValues, cost_function_square_error.cpp:318-341 */
    const double fx = (hx - j.off_x) / j.res, fy = (hy - j.off_y) / j.res;
    const double x0 = floor(fx), y0 = floor(fy);
    const double dx = fx - x0, dy = fy - y0;
    const int xc0 = max((int)x0, 0), yc0 = max((int)y0, 0);
    const int xc1 = min(xc0 + 1, j.cols - 1), yc1 = min(yc0 + 1, j.rows - 1);
    const double m00 = probability_or_half(j, yc0, xc0), m01 = probability_or_half(j, yc1, xc0);
    const double m10 = probability_or_half(j, yc0, xc1), m11 = probability_or_half(j, yc1, xc1);
    /* BilinearInterpolation, :28-37 */
    const double smoothed = dy * (dx * m11 + (1.0 - dx) * m01) + (1.0 - dy) * (dx * m10 + (1.0 - dx) * m00);
    const double err = 1.0 - smoothed;
    t.cost += err * err;
    if (!want_hessian)
        return;
    /* ComputeScaledMapGradMapPoint / SensorPose, :232-277 */
    const double sgx = dy * (m11 - m01) + (1.0 - dy) * (m10 - m00);
    const double sgy = dx * (m11 - m10) + (1.0 - dx) * (m01 - m00);
    const double rx = hx - px, ry = hy - py;
    const double inv = 1.0 / j.res;
    const double g0 = sgx * inv, g1 = sgy * inv, g2 = (-ry * sgx + rx * sgy) * inv;
    t.h00 += g0 * g0;
    t.h01 += g0 * g1;
    t.h02 += g0 * g2;
    t.h11 += g1 * g1;
    t.h12 += g1 * g2;
    t.h22 += g2 * g2;
    t.r0 += g0 * err;
    t.r1 += g1 * err;
    t.r2 += g2 * err;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

/* sum of the beams' terms over the workgroup; every thread gets the totals */
__device__ __forceinline__ CostTerms block_terms(const CostJob& j, double px, double py, double pt,
                                                 bool want_hessian, double (*red)[10])
{
    CostTerms t = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = threadIdx.x; i < j.n; i += kCostBlock)
        beam_terms(j, px, py, pt, i, want_hessian, t);
    double v[10] = { t.cost, t.h00, t.h01, t.h02, t.h11, t.h12, t.h22, t.r0, t.r1, t.r2 };
    const int wave = threadIdx.x >> 6;
    __syncthreads();                       /* red[] free again */
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        v[k] = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0)
            red[wave][k] = v[k];
    }
    __syncthreads();
    CostTerms o;
    double s[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        s[k] = 0.0;
        for (int w = 0; w < kCostBlock / 64; ++w)
            s[k] += red[w][k];
    }
    o.cost = s[0]; o.h00 = s[1]; o.h01 = s[2]; o.h02 = s[3]; o.h11 = s[4];
    o.h12 = s[5]; o.h22 = s[6]; o.r0 = s[7]; o.r1 = s[8]; o.r2 = s[9];
    return o;
}

/* inverse of a symmetric-or-not 3x3 (row-major): adjugate / determinant */
__device__ __forceinline__ void inverse3(const double a[9], double out[9])
{
    const double c00 = a[4] * a[8] - a[5] * a[7];
    const double c01 = a[5] * a[6] - a[3] * a[8];
    const double c02 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const double inv = 1.0 / det;
    out[0] = c00 * inv;
    out[1] = (a[2] * a[7] - a[1] * a[8]) * inv;
    out[2] = (a[1] * a[5] - a[2] * a[4]) * inv;
    out[3] = c01 * inv;
    out[4] = (a[0] * a[8] - a[2] * a[6]) * inv;
    out[5] = (a[2] * a[3] - a[0] * a[5]) * inv;
    out[6] = c02 * inv;
    out[7] = (a[1] * a[6] - a[0] * a[7]) * inv;
    out[8] = (a[0] * a[4] - a[1] * a[3]) * inv;
}

/* x = A^-1 b, Householder QR with column pivoting (colPivHouseholderQr().solve) */
__device__ __forceinline__ void solve3_colpiv_qr(const double a_in[9], const double b_in[3], double x[3])
{
    double a[3][3], b[3] = { b_in[0], b_in[1], b_in[2] };
    int perm[3] = { 0, 1, 2 };
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            a[r][c] = a_in[3 * r + c];
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double best_norm = -1.0;
        for (int c = k; c < 3; ++c) {
            double s = 0.0;
            for (int r = k; r < 3; ++r)
                s += a[r][c] * a[r][c];
            if (s > best_norm) {
                best_norm = s;
                best = c;
            }
        }
        if (best != k) {
            for (int r = 0; r < 3; ++r) {
                const double tmp = a[r][k];
                a[r][k] = a[r][best];
                a[r][best] = tmp;
            }
            const int tp = perm[k];
            perm[k] = perm[best];
            perm[best] = tp;
        }
        double norm = 0.0;
        for (int r = k; r < 3; ++r)
            norm += a[r][k] * a[r][k];
        norm = sqrt(norm);
        if (norm == 0.0)
            continue;
        const double alpha = a[k][k] > 0.0 ? -norm : norm;
        double v[3] = { 0.0, 0.0, 0.0 };
        for (int r = k; r < 3; ++r)
            v[r] = a[r][k];
        v[k] -= alpha;
        double vv = 0.0;
        for (int r = k; r < 3; ++r)
            vv += v[r] * v[r];
        if (vv == 0.0)
            continue;
        for (int c = k; c < 3; ++c) {
            double dot = 0.0;
            for (int r = k; r < 3; ++r)
                dot += v[r] * a[r][c];
            const double f = 2.0 * dot / vv;
            for (int r = k; r < 3; ++r)
                a[r][c] -= f * v[r];
        }
        double dot = 0.0;
        for (int r = k; r < 3; ++r)
            dot += v[r] * b[r];
        const double f = 2.0 * dot / vv;
        for (int r = k; r < 3; ++r)
            b[r] -= f * v[r];
    }
    double y[3];
    for (int k = 2; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < 3; ++c)
            s -= a[k][c] * y[c];
        y[k] = s / a[k][k];
    }
    for (int k = 0; k < 3; ++k)
        x[perm[k]] = y[k];
}

/* grid = jobs. iterations_max == 0: cost + covariance at the given sensor pose
 * (what the matchers do after the search, scan_matcher_correlative.cpp:209-219);
 * > 0: ScanMatcherLinearSolver::OptimizePose from it. */
__global__ __launch_bounds__(kCostBlock) void k_cost_refine(const CostJob* jobs)
{
    __shared__ double red[kCostBlock / 64][10];
    __shared__ double pose[3];
    __shared__ int stop;
    const CostJob j = jobs[blockIdx.x];
    double px = j.sensor_pose[0], py = j.sensor_pose[1], pt = j.sensor_pose[2];
    const double initial_cost = block_terms(j, px, py, pt, false, red).cost;
    double prev_cost = initial_cost, cost = initial_cost, lambda = j.lambda;
    int iterations = 0;
    if (j.iterations_max > 0) {
        while (true) {
            /* OptimizeStep: scan_matcher_linear_solver.cpp:141-167 */
            const CostTerms t = block_terms(j, px, py, pt, true, red);
            if (threadIdx.x == 0) {
                const double h[9] = { t.h00 + lambda, t.h01, t.h02, t.h01, t.h11 + lambda, t.h12,
                                      t.h02, t.h12, t.h22 + lambda };
                const double r[3] = { t.r0, t.r1, t.r2 };
                double d[3];
                solve3_colpiv_qr(h, r, d);
                pose[0] = px + d[0];
                pose[1] = py + d[1];
                pose[2] = pt + d[2];
            }
            __syncthreads();
            px = pose[0];
            py = pose[1];
            pt = pose[2];
            cost = block_terms(j, px, py, pt, false, red).cost;
            ++iterations;
            /* every thread holds the same cost: the same decision everywhere */
            if (iterations >= j.iterations_max || fabs(prev_cost - cost) < j.convergence_threshold)
                break;
            lambda = cost < prev_cost ? fmax(1e-8, lambda * 0.5) : fmin(1e-4, lambda * 2.0);
            prev_cost = cost;
        }
    }
    (void)stop;
    /* ComputeCovariance at the final pose: cost_function_square_error.cpp:131-147 */
    const CostTerms t = block_terms(j, px, py, pt, true, red);
    if (threadIdx.x == 0) {
        CostOut o;
        o.initial_cost = initial_cost;
        o.cost = cost;
        o.best_sensor_pose[0] = px;
        o.best_sensor_pose[1] = py;
        o.best_sensor_pose[2] = pt;
        const double h[9] = { t.h00, t.h01, t.h02, t.h01, t.h11, t.h12, t.h02, t.h12, t.h22 };
        for (int k = 0; k < 9; ++k)
            o.hessian[k] = h[k];
        o.residual[0] = t.r0;
        o.residual[1] = t.r1;
        o.residual[2] = t.r2;
        double inv[9];
        inverse3(h, inv);
        for (int k = 0; k < 9; ++k)
            o.covariance[k] = inv[k] * j.covariance_scale;
        o.lambda = lambda;
        o.iterations = iterations;
        o.pad = 0;
        *j.out = o;
    }
}

} /* namespace csm */
