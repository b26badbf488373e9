/* cost_oracle.cpp
 *
 * TEST INFRASTRUCTURE ONLY (see csm_oracle.cpp's header for who may use it).
 * CPU restatement of the step that follows every search in the reference:
 * CostSquareError (Cost, ComputeHessianAndResidual, ComputeCovariance) and the
 * ScanMatcherLinearSolver refinement, on the dense export of a grid plus its
 * block-allocation bitmap.
 *
 * PARITY STATUS: "parity unpinned". The reference has no fixtures for these
 * functions and its translation units need Eigen3, which this image lacks, so
 * they cannot be compiled here. The two Eigen calls on the path are restated
 * from their documented algorithms, not from Eigen's source:
 *   Matrix3d::inverse()                -> adjugate / determinant
 *   colPivHouseholderQr().solve(b)     -> Householder QR with column pivoting
 * so results agree with the reference only to rounding (not bit for bit); the
 * GPU tests compare at the tolerance include/csm_hip.h states. Internal pins:
 * the analytic gradient 2 * residual against central differences of Cost(),
 * inverse * matrix against the identity, QR solve against the inverse.
 *
 * All paths below are relative to /root/reference/. Plain IEEE double in source
 * order; build with -ffp-contract=off.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

extern "C" {

void orc_hit_point(const double pose[3], double range, double angle, double out[2]);
double orc_value_to_probability(unsigned value);
void orc_compound(const double s[3], const double d[3], double out[3]);
void orc_move_backward(const double e[3], const double d[3], double out[3]);

/* A grid as the cost function reads it: GridMap::ProbabilityOr(row, col, 0.5)
 * (src/my_lidar_graph_slam/grid_map_new/grid_map.cpp:423-436) = 0.5 for a cell
 * outside the map or in a block that was never allocated
 * (IsBlockAllocated, :795-801), else the cell's probability (0 for an unknown
 * cell of an allocated block). alloc: one byte per block, row-major
 * [ceil(rows / bs)][ceil(cols / bs)], or null = every block allocated. */
struct OrcCostGrid {
    const uint16_t* v;
    int rows, cols;
    double res, offX, offY;
    const uint8_t* alloc;
    int log2Block;
};

static double probability_or_half(const OrcCostGrid& g, int row, int col)
{
    if (row < 0 || row >= g.rows || col < 0 || col >= g.cols)
        return 0.5;
    if (g.alloc) {
        const int bs = 1 << g.log2Block;
        const int bcols = (g.cols + bs - 1) >> g.log2Block;
        if (!g.alloc[(row >> g.log2Block) * bcols + (col >> g.log2Block)])
            return 0.5;
    }
    return orc_value_to_probability(g.v[static_cast<size_t>(row) * g.cols + col]);
}

struct MapValues { double dx, dy, m00, m01, m10, m11; };

/* GetClosestMapValues: src/.../mapping/cost_function_square_error.cpp:318-341
 * with PositionToIndexF (grid_map_geometry.cpp:125-132) */
static MapValues closest_map_values(const OrcCostGrid& g, double px, double py)
{
    const double fx = (px - g.offX) / g.res;
    const double fy = (py - g.offY) / g.res;
    const double x0 = std::floor(fx);
    const double y0 = std::floor(fy);
    MapValues m;
    m.dx = fx - x0;
    m.dy = fy - y0;
    const int xc0 = std::max(static_cast<int>(x0), 0);
    const int yc0 = std::max(static_cast<int>(y0), 0);
    const int xc1 = std::min(xc0 + 1, g.cols - 1);
    const int yc1 = std::min(yc0 + 1, g.rows - 1);
    m.m00 = probability_or_half(g, yc0, xc0);
    m.m01 = probability_or_half(g, yc1, xc0);
    m.m10 = probability_or_half(g, yc0, xc1);
    m.m11 = probability_or_half(g, yc1, xc1);
    return m;
}

/* MapValues::BilinearInterpolation: cost_function_square_error.cpp:28-37 */
static double bilinear(const MapValues& m)
{
    return m.dy * (m.dx * m.m11 + (1.0 - m.dx) * m.m01) +
           (1.0 - m.dy) * (m.dx * m.m10 + (1.0 - m.dx) * m.m00);
}

/* CostSquareError::Cost: cost_function_square_error.cpp:48-77 */
double orc_cost(const OrcCostGrid* g, const double* angles, const double* ranges, int n,
                const double sensorPose[3])
{
    double cost = 0.0;
    for (int i = 0; i < n; ++i) {
        double hp[2];
        orc_hit_point(sensorPose, ranges[i], angles[i], hp);
        const double smoothed = bilinear(closest_map_values(*g, hp[0], hp[1]));
        cost += std::pow(1.0 - smoothed, 2.0);
    }
    return cost;
}

/* ComputeHessianAndResidual: cost_function_square_error.cpp:151-195 with
 * ComputeScaledMapGradMapPoint (:232-252) and ComputeScaledMapGradSensorPose
 * (:256-277). hessian row-major 3x3. */
void orc_hessian_residual(const OrcCostGrid* g, const double* angles, const double* ranges, int n,
                          const double sensorPose[3], double hessian[9], double residual[3])
{
    std::memset(hessian, 0, 9 * sizeof(double));
    std::memset(residual, 0, 3 * sizeof(double));
    const double reciprocalResolution = 1.0 / g->res;
    for (int i = 0; i < n; ++i) {
        double hp[2];
        orc_hit_point(sensorPose, ranges[i], angles[i], hp);
        const MapValues m = closest_map_values(*g, hp[0], hp[1]);
        const double rx = hp[0] - sensorPose[0];
        const double ry = hp[1] - sensorPose[1];
        const double sgx = m.dy * (m.m11 - m.m01) + (1.0 - m.dy) * (m.m10 - m.m00);
        const double sgy = m.dx * (m.m11 - m.m10) + (1.0 - m.dx) * (m.m01 - m.m00);
        const double sgt = -ry * sgx + rx * sgy;
        const double grad[3] = { sgx * reciprocalResolution, sgy * reciprocalResolution,
                                 sgt * reciprocalResolution };
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                hessian[3 * r + c] += grad[r] * grad[c];
        const double mapResidual = 1.0 - bilinear(m);
        for (int r = 0; r < 3; ++r)
            residual[r] += grad[r] * mapResidual;
    }
}

/* inverse of a 3x3 matrix: adjugate / determinant */
int orc_inverse3(const double a[9], double out[9])
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
    return det != 0.0 ? 0 : -1;
}

/* ComputeCovariance: cost_function_square_error.cpp:131-147 */
void orc_covariance(const OrcCostGrid* g, const double* angles, const double* ranges, int n,
                    const double sensorPose[3], double covarianceScale, double cov[9])
{
    double h[9], r[3];
    orc_hessian_residual(g, angles, ranges, n, sensorPose, h, r);
    orc_inverse3(h, cov);
    for (int i = 0; i < 9; ++i)
        cov[i] *= covarianceScale;
}

/* x = A^-1 b by Householder QR with column pivoting (what
 * hessianMat.colPivHouseholderQr().solve(residualVec) computes,
 * scan_matcher_linear_solver.cpp:158-159) */
void orc_solve3_colpiv_qr(const double a_in[9], const double b_in[3], double x[3])
{
    double a[3][3], b[3] = { b_in[0], b_in[1], b_in[2] };
    int perm[3] = { 0, 1, 2 };
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            a[r][c] = a_in[3 * r + c];
    for (int k = 0; k < 3; ++k) {
        /* pivot: the remaining column with the largest norm */
        int best = k;
        double bestNorm = -1.0;
        for (int c = k; c < 3; ++c) {
            double s = 0.0;
            for (int r = k; r < 3; ++r)
                s += a[r][c] * a[r][c];
            if (s > bestNorm) {
                bestNorm = s;
                best = c;
            }
        }
        if (best != k) {
            for (int r = 0; r < 3; ++r)
                std::swap(a[r][k], a[r][best]);
            std::swap(perm[k], perm[best]);
        }
        /* Householder reflector for column k, rows k..2 */
        double norm = 0.0;
        for (int r = k; r < 3; ++r)
            norm += a[r][k] * a[r][k];
        norm = std::sqrt(norm);
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

struct OrcRefineResult {
    double normalizedInitialCost, normalizedCost;
    double sensorPose[3], bestSensorPose[3], estimatedPose[3];
    double covariance[9];
    double lambda;          /* damping factor after the call (the reference keeps it in the object) */
    int iterations;
};

/* ScanMatcherLinearSolver::OptimizePose + OptimizeStep:
 * scan_matcher_linear_solver.cpp:66-169. lambda: the object's damping factor
 * when the call starts (InitialLambda on the first call). */
void orc_linear_solver(const OrcCostGrid* g, const double* angles, const double* ranges, int n,
                       const double rel[3], const double initialPose[3], int iterationsMax,
                       double convergenceThreshold, double lambda, double covarianceScale,
                       OrcRefineResult* out)
{
    orc_compound(initialPose, rel, out->sensorPose);
    const double initialCost = orc_cost(g, angles, ranges, n, out->sensorPose);
    out->normalizedInitialCost = initialCost / n;
    double prevCost = initialCost;
    double cost = std::numeric_limits<double>::max();
    double best[3] = { out->sensorPose[0], out->sensorPose[1], out->sensorPose[2] };
    int iterations = 0;
    while (true) {
        double h[9], r[3], delta[3];
        orc_hessian_residual(g, angles, ranges, n, best, h, r);
        h[0] += lambda;
        h[4] += lambda;
        h[8] += lambda;
        orc_solve3_colpiv_qr(h, r, delta);
        best[0] += delta[0];
        best[1] += delta[1];
        best[2] += delta[2];
        cost = orc_cost(g, angles, ranges, n, best);
        if (++iterations >= iterationsMax || std::fabs(prevCost - cost) < convergenceThreshold)
            break;
        if (cost < prevCost)
            lambda = std::max(1e-8, lambda * 0.5);
        else
            lambda = std::min(1e-4, lambda * 2.0);
        prevCost = cost;
    }
    out->normalizedCost = cost / n;
    std::memcpy(out->bestSensorPose, best, sizeof(best));
    orc_move_backward(best, rel, out->estimatedPose);
    orc_covariance(g, angles, ranges, n, best, covarianceScale, out->covariance);
    out->lambda = lambda;
    out->iterations = iterations;
}

} /* extern "C" */
