/* csm_oracle.cpp
 *
 * TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's correlative
 * scan-matching path, used as the checker by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg. Nothing in the product path
 * (my-lidar-graph-slam-v2_amd/) links, loads or calls this file.
 *
 * PARITY STATUS: "parity unpinned" for the matcher as a whole. The reference
 * ships no tests / golden vectors, and its matcher translation units need
 * Eigen3 and Boost headers that this image lacks, so the reference matcher
 * cannot be compiled here. Pinned pieces: the header-only geometry
 * (pose.hpp, sensor/sensor_data.hpp) and the value->probability formula
 * (grid_map_new/grid_values.hpp) DO compile from the reference's own files;
 * oracle/Makefile builds them into oracle/_ref/libref_geom.so and
 * tests/test_cpu_oracle.py checks this restatement against them bit for bit.
 * Everything else is restated from the cited lines and cross-checked by an
 * independent second formulation (literal sequential sweep vs closed form).
 *
 * Written in C++ (not C) for one reason: the branch-and-bound matcher's tie
 * order is whatever libstdc++'s std::priority_queue does, so the restatement
 * uses the same container.
 *
 * All paths below are relative to /root/reference/.
 * Floating point: the reference builds with g++ -O3 for baseline x86-64 (no
 * FMA), so every expression is plain IEEE double in source order; build this
 * file with -ffp-contract=off.
 */

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <queue>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

/* ---- pose algebra: include/my_lidar_graph_slam/pose.hpp:154-166, 183-200,
 * 215-227 ---- */
void orc_compound(const double s[3], const double d[3], double out[3])
{
    const double sinT = std::sin(s[2]);
    const double cosT = std::cos(s[2]);
    out[0] = cosT * d[0] - sinT * d[1] + s[0];
    out[1] = sinT * d[0] + cosT * d[1] + s[1];
    out[2] = s[2] + d[2];
}

void orc_inverse_compound(const double s[3], const double e[3], double out[3])
{
    const double sinT = std::sin(s[2]);
    const double cosT = std::cos(s[2]);
    const double dx = e[0] - s[0];
    const double dy = e[1] - s[1];
    const double dt = e[2] - s[2];
    out[0] = cosT * dx + sinT * dy;
    out[1] = -sinT * dx + cosT * dy;
    out[2] = dt;
}

void orc_move_backward(const double e[3], const double d[3], double out[3])
{
    const double theta = e[2] - d[2];
    const double sinT = std::sin(theta);
    const double cosT = std::cos(theta);
    out[0] = e[0] - cosT * d[0] + sinT * d[1];
    out[1] = e[1] - sinT * d[0] - cosT * d[1];
    out[2] = theta;
}

/* ---- hit point: include/my_lidar_graph_slam/sensor/sensor_data.hpp:189-203 */
void orc_hit_point(const double pose[3], double range, double angle,
                   double out[2])
{
    const double cosT = std::cos(pose[2] + angle);
    const double sinT = std::sin(pose[2] + angle);
    out[0] = pose[0] + range * cosT;
    out[1] = pose[1] + range * sinT;
}

/* ---- value -> probability: include/.../grid_map_new/grid_values.hpp:26-35
 * with the constants of grid_binary_bayes.hpp:163-176 (identical in
 * grid_constant.hpp:166-178): ValueMin 1, ValueMax 65535, ProbabilityMin 1e-3,
 * ProbabilityMax 1 - 1e-3. The reference table (grid_values.cpp:11-46) has
 * 65535 entries, so value 65535 is out of bounds there; this restatement
 * extends the same formula to 65535 and says so. */
static const double kProbMin = 1e-3;
static const double kProbMax = 1.0 - 1e-3;

double orc_value_to_probability(unsigned value)
{
    if (value == 0)
        return 0.0;
    return kProbMin + (kProbMax - kProbMin) *
           static_cast<double>(static_cast<int>(value) - 1) /
           static_cast<double>(65535 - 1);
}

void orc_lut(double* lut /* 65536 */)
{
    for (unsigned v = 0; v < 65536; ++v)
        lut[v] = orc_value_to_probability(v);
}

/* ---- dense grid view. Restates GridMap<T>::ProbabilityOr / ValueOr
 * (src/.../grid_map_new/grid_map.cpp:385-397, 424-436): out-of-map reads give
 * the caller's default (unknown). The dense array is what CopyValues
 * (grid_map.cpp:439-457) exports; unallocated blocks are 0 there. */
struct Grid {
    const uint16_t* v;
    int rows, cols;
    double res, offX, offY;
};

static inline unsigned grid_value_or0(const Grid& g, int row, int col)
{
    if (row < 0 || row >= g.rows || col < 0 || col >= g.cols)
        return 0;
    return g.v[static_cast<size_t>(row) * g.cols + col];
}

/* grid_map_geometry.cpp:113-122 */
static inline void position_to_index(const Grid& g, double x, double y,
                                     int* col, int* row)
{
    *col = static_cast<int>(std::floor((x - g.offX) / g.res));
    *row = static_cast<int>(std::floor((y - g.offY) / g.res));
}

/* ---- sliding window maximum: include/my_lidar_graph_slam/util.hpp:369-424
 * restated literally (monotonic index queue, "repeat the last window" tail). */
static void sliding_window_max(const uint16_t* in, size_t inStride,
                               uint16_t* out, size_t outStride,
                               int n, int win)
{
    std::deque<int> q;
    int idxIn = 0, idxOut = 0;
    auto at = [&](int i) { return in[static_cast<size_t>(i) * inStride]; };
    for (idxIn = 0; idxIn < win; ++idxIn) {
        while (!q.empty() && at(idxIn) >= at(q.back()))
            q.pop_back();
        q.push_back(idxIn);
    }
    for (; idxIn < n; ++idxIn) {
        out[static_cast<size_t>(idxOut++) * outStride] = at(q.front());
        while (!q.empty() && q.front() <= idxIn - win)
            q.pop_front();
        while (!q.empty() && at(idxIn) >= at(q.back()))
            q.pop_back();
        q.push_back(idxIn);
    }
    for (; idxOut < n; ++idxOut)
        out[static_cast<size_t>(idxOut) * outStride] = at(q.front());
}

/* PrecomputeGridMap: src/.../mapping/grid_map_builder.cpp:918-984, 1015-1065.
 * First pass runs down every column (SlidingWindowMaxRow), second along every
 * row (SlidingWindowMaxCol). Returns -1 if win does not fit. */
int orc_boxmax(const uint16_t* in, int rows, int cols, int win, uint16_t* out)
{
    if (win < 1 || win > rows || win > cols)
        return -1;
    std::vector<uint16_t> mid(static_cast<size_t>(rows) * cols);
    for (int c = 0; c < cols; ++c)
        sliding_window_max(in + c, cols, mid.data() + c, cols, rows, win);
    for (int r = 0; r < rows; ++r)
        sliding_window_max(mid.data() + static_cast<size_t>(r) * cols, 1,
                           out + static_cast<size_t>(r) * cols, 1, cols, win);
    return 0;
}

/* ---- search step: scan_matcher_correlative.cpp:255-274 (same text in
 * scan_matcher_branch_bound.cpp:293-312) */
void orc_search_step(double res, const double* ranges, int n,
                     double* stepX, double* stepY, double* stepT)
{
    const double maxRange = *std::max_element(ranges, ranges + n);
    const double theta = res / maxRange;
    *stepX = res;
    *stepY = res;
    *stepT = std::acos(1.0 - 0.5 * theta * theta);
}

struct OrcScan {
    const double* angles;
    const double* ranges;
    int n;
    double rel[3]; /* RelativeSensorPose */
};

struct OrcCsmParams {
    double rangeX, rangeY, rangeT; /* SearchRangeX/Y/Theta */
    int lowRes;                    /* LowResolutionMapWinSize */
    double scoreThr, knownThr;
};

struct OrcResult {
    int found;
    int bestX, bestY, bestT;
    int winX, winY, winT;
    double stepX, stepY, stepT;
    double scoreMax;      /* normalized score of the winner (or threshold) */
    double sensorPose[3]; /* Compound(initial, relative) */
    double bestSensorPose[3];
    double estimatedPose[3];
    long long ignoredNodes, processedNodes;
    long long fineEvaluated; /* fully evaluated fine poses (CSM) / leaves */
};

/* ComputeScanIndices: scan_matcher_correlative.cpp:277-297 */
static void scan_indices(const Grid& g, const double pose[3],
                         const OrcScan& s, int* col, int* row)
{
    for (int i = 0; i < s.n; ++i) {
        double hp[2];
        orc_hit_point(pose, s.ranges[i], s.angles[i], hp);
        position_to_index(g, hp[0], hp[1], &col[i], &row[i]);
    }
}

void orc_project(const double* grid_geom /* res, offX, offY */,
                 const double pose[3], const double* angles,
                 const double* ranges, int n, int* col, int* row)
{
    Grid g { nullptr, 0, 0, grid_geom[0], grid_geom[1], grid_geom[2] };
    OrcScan s { angles, ranges, n, { 0, 0, 0 } };
    scan_indices(g, pose, s, col, row);
}

struct ScoreSummary { double normalized, sum, knownRate; int known; };

/* ComputeScore: scan_matcher_correlative.cpp:301-336 */
static ScoreSummary compute_score(const Grid& g, const double* lut,
                                  const int* col, const int* row, int n,
                                  int offX, int offY)
{
    int known = 0;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        const double prob = lut[grid_value_or0(g, row[i] + offY, col[i] + offX)];
        if (prob == 0.0)
            continue;
        sum += prob;
        ++known;
    }
    ScoreSummary r;
    r.normalized = sum / static_cast<double>(n);
    r.sum = sum;
    r.knownRate = static_cast<double>(known) / static_cast<double>(n);
    r.known = known;
    return r;
}

static const double* shared_lut()
{
    static std::vector<double> lut;
    if (lut.empty()) {
        lut.resize(65536);
        orc_lut(lut.data());
    }
    return lut.data();
}

static void setup_window(const Grid& g, const OrcScan& s, const double init[3],
                         double rangeX, double rangeY, double rangeT,
                         OrcResult* out)
{
    orc_compound(init, s.rel, out->sensorPose);
    orc_search_step(g.res, s.ranges, s.n, &out->stepX, &out->stepY, &out->stepT);
    out->winX = static_cast<int>(std::ceil(0.5 * rangeX / out->stepX));
    out->winY = static_cast<int>(std::ceil(0.5 * rangeY / out->stepY));
    out->winT = static_cast<int>(std::ceil(0.5 * rangeT / out->stepT));
}

static void finish_pose(const OrcScan& s, OrcResult* out)
{
    out->bestSensorPose[0] = out->sensorPose[0] + out->bestX * out->stepX;
    out->bestSensorPose[1] = out->sensorPose[1] + out->bestY * out->stepY;
    out->bestSensorPose[2] = out->sensorPose[2] + out->bestT * out->stepT;
    orc_move_backward(out->bestSensorPose, s.rel, out->estimatedPose);
}

/* ScanMatcherCorrelative::OptimizePose, 6-argument overload, literal
 * sequential sweep with coarse pruning: scan_matcher_correlative.cpp:118-244,
 * 339-368. `coarse` is the PrecomputeGridMap(map, lowRes) output (pass the
 * result of orc_boxmax). Cost / covariance (lines 209-219) are outside the
 * hot path and not restated. */
int orc_csm(const uint16_t* grid, const uint16_t* coarse, int rows, int cols,
            const double* geom /* res, offX, offY */,
            const double* angles, const double* ranges, int n,
            const double* rel, const double* init,
            const OrcCsmParams* p, OrcResult* out)
{
    Grid g { grid, rows, cols, geom[0], geom[1], geom[2] };
    Grid gc { coarse, rows, cols, geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { rel[0], rel[1], rel[2] } };
    const double* lut = shared_lut();
    std::memset(out, 0, sizeof(*out));
    setup_window(g, s, init, p->rangeX, p->rangeY, p->rangeT, out);

    const int winX = out->winX, winY = out->winY, winT = out->winT;
    const int L = p->lowRes;
    double scoreMax = p->scoreThr;
    int bestX = -winX, bestY = -winY, bestT = -winT;
    std::vector<int> col(n), row(n);

    for (int t = -winT; t <= winT; ++t) {
        const double pose[3] = { out->sensorPose[0], out->sensorPose[1],
                                 out->sensorPose[2] + out->stepT * t };
        scan_indices(gc, pose, s, col.data(), row.data());
        for (int x = -winX; x <= winX; x += L) {
            for (int y = -winY; y <= winY; y += L) {
                const ScoreSummary c =
                    compute_score(gc, lut, col.data(), row.data(), n, x, y);
                if (c.normalized <= scoreMax || c.knownRate <= p->knownThr) {
                    out->ignoredNodes++;
                    continue;
                }
                for (int fx = x; fx < x + L; ++fx) {
                    for (int fy = y; fy < y + L; ++fy) {
                        const ScoreSummary f = compute_score(
                            g, lut, col.data(), row.data(), n, fx, fy);
                        out->fineEvaluated++;
                        if (scoreMax < f.normalized) {
                            scoreMax = f.normalized;
                            bestX = fx;
                            bestY = fy;
                            bestT = t;
                        }
                    }
                }
                out->processedNodes++;
            }
        }
    }
    out->found = scoreMax > p->scoreThr;
    out->bestX = bestX;
    out->bestY = bestY;
    out->bestT = bestT;
    out->scoreMax = scoreMax;
    finish_pose(s, out);
    return 0;
}

/* The same sweep with the theta loop spread over all host cores (OpenMP): the
 * all-core CPU baseline of BASELINE.md section 4(b). Every theta slice runs the
 * literal x / y sweep of orc_csm with a running maximum of its own plus a
 * shared bound (the best score any slice has found so far, tested strictly);
 * the slices' winners are then merged in theta order with the same strict `<`.
 * Same winner as orc_csm whenever the closed form of A5 holds (no edge band).
 * `threads` receives the number of OpenMP threads actually used. */
int orc_csm_omp(const uint16_t* grid, const uint16_t* coarse, int rows, int cols,
                const double* geom, const double* angles, const double* ranges, int n,
                const double* rel, const double* init,
                const OrcCsmParams* p, OrcResult* out, int* threads)
{
    Grid g { grid, rows, cols, geom[0], geom[1], geom[2] };
    Grid gc { coarse, rows, cols, geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { rel[0], rel[1], rel[2] } };
    const double* lut = shared_lut();
    std::memset(out, 0, sizeof(*out));
    setup_window(g, s, init, p->rangeX, p->rangeY, p->rangeT, out);
    const int winX = out->winX, winY = out->winY, winT = out->winT;
    const int L = p->lowRes;
    const int nT = 2 * winT + 1;
    struct Slice { double score; int x, y; long long ignored, processed, fine; };
    std::vector<Slice> best(nT);
    int used = 1;
    double shared = p->scoreThr;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#endif
        std::vector<int> col(n), row(n);
#pragma omp for schedule(dynamic, 1)
        for (int ti = 0; ti < nT; ++ti) {
            const int t = ti - winT;
            const double pose[3] = { out->sensorPose[0], out->sensorPose[1],
                                     out->sensorPose[2] + out->stepT * t };
            scan_indices(gc, pose, s, col.data(), row.data());
            Slice b { p->scoreThr, -winX, -winY, 0, 0, 0 };
            for (int x = -winX; x <= winX; x += L)
                for (int y = -winY; y <= winY; y += L) {
                    const ScoreSummary c =
                        compute_score(gc, lut, col.data(), row.data(), n, x, y);
                    /* `shared`: the best score any slice has found so far. A node
                     * strictly below it cannot hold the global winner; equality is
                     * kept so that the earliest theta still wins a tie. */
                    double bound;
#pragma omp atomic read
                    bound = shared;
                    if (c.normalized <= b.score || c.normalized < bound ||
                        c.knownRate <= p->knownThr) {
                        b.ignored++;
                        continue;
                    }
                    for (int fx = x; fx < x + L; ++fx)
                        for (int fy = y; fy < y + L; ++fy) {
                            const ScoreSummary f = compute_score(
                                g, lut, col.data(), row.data(), n, fx, fy);
                            b.fine++;
                            if (b.score < f.normalized) {
                                b.score = f.normalized;
                                b.x = fx;
                                b.y = fy;
                            }
                        }
                    if (b.score > bound) {
#pragma omp critical(orc_csm_omp_bound)
                        if (b.score > shared)
                            shared = b.score;
                    }
                    b.processed++;
                }
            best[ti] = b;
        }
    }
    double scoreMax = p->scoreThr;
    int bestX = -winX, bestY = -winY, bestT = -winT;
    for (int ti = 0; ti < nT; ++ti) {
        out->ignoredNodes += best[ti].ignored;
        out->processedNodes += best[ti].processed;
        out->fineEvaluated += best[ti].fine;
        if (scoreMax < best[ti].score) {
            scoreMax = best[ti].score;
            bestX = best[ti].x;
            bestY = best[ti].y;
            bestT = ti - winT;
        }
    }
    out->found = scoreMax > p->scoreThr;
    out->bestX = bestX;
    out->bestY = bestY;
    out->bestT = bestT;
    out->scoreMax = scoreMax;
    finish_pose(s, out);
    if (threads)
        *threads = used;
    return 0;
}

/* Independent second formulation of the same result (SURVEY.md section 8(a)
 * A5): exhaustive over the extended domain, no pruning by score; a candidate
 * counts iff its coarse node's known rate passes; first strict maximum in
 * (t, coarse x, coarse y, fine x, fine y) order. Valid only when no coarse
 * read falls in the negative edge band (A8); `touchesBand` reports that.
 * Optionally dumps per-candidate integer sums: S = sum of raw values over
 * known cells, K = known count, laid out [t][xi][yi] with xi = fx + winX. */
int orc_csm_closed_form(const uint16_t* grid, const uint16_t* coarse,
                        int rows, int cols, const double* geom,
                        const double* angles, const double* ranges, int n,
                        const double* rel, const double* init,
                        const OrcCsmParams* p, OrcResult* out,
                        int* touchesBand,
                        uint32_t* dumpS, uint16_t* dumpK,
                        uint16_t* dumpCoarseK)
{
    Grid g { grid, rows, cols, geom[0], geom[1], geom[2] };
    Grid gc { coarse, rows, cols, geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { rel[0], rel[1], rel[2] } };
    const double* lut = shared_lut();
    std::memset(out, 0, sizeof(*out));
    setup_window(g, s, init, p->rangeX, p->rangeY, p->rangeT, out);
    const int winX = out->winX, winY = out->winY, winT = out->winT;
    const int L = p->lowRes;
    const int nxc = (2 * winX + 1 + L - 1) / L, nyc = (2 * winY + 1 + L - 1) / L;
    const int nx = nxc * L, ny = nyc * L;
    double scoreMax = p->scoreThr;
    int bestX = -winX, bestY = -winY, bestT = -winT;
    std::vector<int> col(n), row(n);
    int band = 0;

    for (int t = -winT; t <= winT; ++t) {
        const double pose[3] = { out->sensorPose[0], out->sensorPose[1],
                                 out->sensorPose[2] + out->stepT * t };
        scan_indices(gc, pose, s, col.data(), row.data());
        for (int xc = 0; xc < nxc; ++xc) {
            for (int yc = 0; yc < nyc; ++yc) {
                const int x = -winX + xc * L, y = -winY + yc * L;
                for (int i = 0; i < n; ++i) {
                    const int rr = row[i] + y, cc = col[i] + x;
                    if ((rr < 0 && rr > -L) || (cc < 0 && cc > -L))
                        band = 1;
                }
                const ScoreSummary c =
                    compute_score(gc, lut, col.data(), row.data(), n, x, y);
                if (dumpCoarseK)
                    dumpCoarseK[(static_cast<size_t>(t + winT) * nxc + xc) * nyc + yc] =
                        static_cast<uint16_t>(c.known);
                const bool eligible = c.knownRate > p->knownThr;
                for (int fx = x; fx < x + L; ++fx) {
                    for (int fy = y; fy < y + L; ++fy) {
                        const ScoreSummary f = compute_score(
                            g, lut, col.data(), row.data(), n, fx, fy);
                        out->fineEvaluated++;
                        if (dumpS) {
                            uint32_t S = 0;
                            for (int i = 0; i < n; ++i)
                                S += grid_value_or0(g, row[i] + fy, col[i] + fx);
                            const size_t idx =
                                (static_cast<size_t>(t + winT) * nx + (fx + winX)) * ny + (fy + winY);
                            dumpS[idx] = S;
                            dumpK[idx] = static_cast<uint16_t>(f.known);
                        }
                        if (eligible && scoreMax < f.normalized) {
                            scoreMax = f.normalized;
                            bestX = fx;
                            bestY = fy;
                            bestT = t;
                        }
                    }
                }
            }
        }
    }
    out->found = scoreMax > p->scoreThr;
    out->bestX = bestX;
    out->bestY = bestY;
    out->bestT = bestT;
    out->scoreMax = scoreMax;
    if (touchesBand)
        *touchesBand = band;
    finish_pose(s, out);
    return 0;
}

/* ---- branch and bound ---- */

/* ScorePixelAccurate::Score: src/.../score_function_pixel_accurate.cpp:16-58 */
static ScoreSummary score_pixel_accurate(const Grid& g, const double* lut,
                                         const OrcScan& s, const double pose[3])
{
    double sum = 0.0;
    int known = 0;
    for (int i = 0; i < s.n; ++i) {
        double hp[2];
        orc_hit_point(pose, s.ranges[i], s.angles[i], hp);
        int col, row;
        position_to_index(g, hp[0], hp[1], &col, &row);
        const double prob = lut[grid_value_or0(g, row, col)];
        if (prob == 0.0)
            continue;
        sum += prob;
        ++known;
    }
    ScoreSummary r;
    r.normalized = sum / static_cast<double>(s.n);
    r.sum = sum;
    r.knownRate = static_cast<double>(known) / static_cast<double>(s.n);
    r.known = known;
    return r;
}

/* Node: include/.../mapping/scan_matcher_branch_bound.hpp:67-106 (ordering on
 * the normalized score only) */
struct BnbNode {
    int x, y, t, h;
    double score, knownRate;
    bool operator<(const BnbNode& o) const { return score < o.score; }
};

struct OrcBnbParams {
    double rangeX, rangeY, rangeT;
    int nodeHeightMax;
    double scoreThr, knownThr;
};

/* ScanMatcherBranchBound::OptimizePose:
 * src/.../scan_matcher_branch_bound.cpp:111-278. `pyramid` holds
 * nodeHeightMax+1 dense levels (level h = orc_boxmax with win 2^h),
 * concatenated. The leaf level reads pyramid level 0, as the reference does
 * (precompMaps.at(0)). */
int orc_bnb(const uint16_t* pyramid, int rows, int cols, const double* geom,
            const double* angles, const double* ranges, int n,
            const double* rel, const double* init,
            const OrcBnbParams* p, OrcResult* out)
{
    const int H = p->nodeHeightMax;
    std::vector<Grid> lv(H + 1);
    for (int h = 0; h <= H; ++h)
        lv[h] = Grid { pyramid + static_cast<size_t>(h) * rows * cols, rows, cols,
                       geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { rel[0], rel[1], rel[2] } };
    const double* lut = shared_lut();
    std::memset(out, 0, sizeof(*out));
    setup_window(lv[0], s, init, p->rangeX, p->rangeY, p->rangeT, out);
    const int winX = out->winX, winY = out->winY, winT = out->winT;

    double scoreMax = p->scoreThr;
    int bestX = 0, bestY = 0, bestT = 0;
    std::priority_queue<BnbNode> q;
    const int winSizeMax = 1 << H;

    auto appendNode = [&](int x, int y, int t, int h) {
        const double pose[3] = { out->sensorPose[0] + x * out->stepX,
                                 out->sensorPose[1] + y * out->stepY,
                                 out->sensorPose[2] + t * out->stepT };
        const ScoreSummary sc = score_pixel_accurate(lv[h], lut, s, pose);
        if (h == 0)
            out->fineEvaluated++;
        if (sc.normalized > scoreMax)
            q.push(BnbNode { x, y, t, h, sc.normalized, sc.knownRate });
        if (sc.normalized <= scoreMax)
            out->ignoredNodes++;
    };

    for (int x = -winX; x <= winX; x += winSizeMax)
        for (int y = -winY; y <= winY; y += winSizeMax)
            for (int t = -winT; t <= winT; ++t)
                appendNode(x, y, t, H);

    while (!q.empty()) {
        const BnbNode cur = q.top();
        if (cur.score <= scoreMax || cur.knownRate <= p->knownThr) {
            q.pop();
            out->ignoredNodes++;
            continue;
        }
        if (cur.h == 0) {
            bestX = cur.x;
            bestY = cur.y;
            bestT = cur.t;
            scoreMax = cur.score;
            q.pop();
            out->processedNodes++;
        } else {
            const int h = cur.h - 1;
            const int w = 1 << h;
            q.pop();
            out->processedNodes++;
            appendNode(cur.x, cur.y, cur.t, h);
            appendNode(cur.x + w, cur.y, cur.t, h);
            appendNode(cur.x, cur.y + w, cur.t, h);
            appendNode(cur.x + w, cur.y + w, cur.t, h);
        }
    }

    out->found = scoreMax > p->scoreThr;
    out->bestX = bestX;
    out->bestY = bestY;
    out->bestT = bestT;
    out->scoreMax = scoreMax;
    finish_pose(s, out);
    return 0;
}

/* Exhaustive per-node scores of one pyramid level with the branch-and-bound
 * projection (per-node pose in double, as appendNode does). Used to check the
 * device's per-level sums. Layout [t][xi][yi], xi, yi index nodes at stride
 * 2^h from -win over the extended domain (nRoot * 2^H cells per axis). */
int orc_bnb_level_dump(const uint16_t* level, int rows, int cols,
                       const double* geom, const double* angles,
                       const double* ranges, int n, const double* rel,
                       const double* init, const OrcBnbParams* p, int h,
                       uint32_t* dumpS, uint16_t* dumpK)
{
    Grid g { level, rows, cols, geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { rel[0], rel[1], rel[2] } };
    OrcResult w;
    std::memset(&w, 0, sizeof(w));
    setup_window(g, s, init, p->rangeX, p->rangeY, p->rangeT, &w);
    const int H = p->nodeHeightMax;
    const int big = 1 << H, st = 1 << h;
    const int nrx = (2 * w.winX + 1 + big - 1) / big, nry = (2 * w.winY + 1 + big - 1) / big;
    const int nx = nrx * big / st, ny = nry * big / st;
    for (int t = -w.winT; t <= w.winT; ++t)
        for (int xi = 0; xi < nx; ++xi)
            for (int yi = 0; yi < ny; ++yi) {
                const int x = -w.winX + xi * st, y = -w.winY + yi * st;
                const double pose[3] = { w.sensorPose[0] + x * w.stepX,
                                         w.sensorPose[1] + y * w.stepY,
                                         w.sensorPose[2] + t * w.stepT };
                uint32_t S = 0;
                int K = 0;
                for (int i = 0; i < n; ++i) {
                    double hp[2];
                    orc_hit_point(pose, ranges[i], angles[i], hp);
                    int col, row;
                    position_to_index(g, hp[0], hp[1], &col, &row);
                    const unsigned v = grid_value_or0(g, row, col);
                    S += v;
                    K += v != 0;
                }
                const size_t idx = (static_cast<size_t>(t + w.winT) * nx + xi) * ny + yi;
                dumpS[idx] = S;
                dumpK[idx] = static_cast<uint16_t>(K);
            }
    return 0;
}

/* ScanMatcherGridSearch::OptimizePose, 5-argument overload:
 * src/.../mapping/scan_matcher_grid_search.cpp:84-142. Offsets are accumulated
 * doubles (dy += sy ...), every pose is projected on its own, and a pose counts
 * only if its own known rate passes. bestIdx = (ix, iy, it) of the winner in
 * the three loops (or -1s), for comparison with the device's indices. */
struct OrcGridParams {
    double rangeX, rangeY, rangeT;
    double stepX, stepY, stepT;
    double scoreThr, knownThr;
};

int orc_grid_search(const uint16_t* grid, int rows, int cols, const double* geom,
                    const double* angles, const double* ranges, int n, const double* rel,
                    const double* init, const OrcGridParams* p, OrcResult* out, int* bestIdx,
                    long long* numEvaluations)
{
    Grid g { grid, rows, cols, geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { rel[0], rel[1], rel[2] } };
    const double* lut = shared_lut();
    std::memset(out, 0, sizeof(*out));
    orc_compound(init, s.rel, out->sensorPose);
    const double rx = p->rangeX / 2.0, ry = p->rangeY / 2.0, rt = p->rangeT / 2.0;
    const double sx = p->stepX, sy = p->stepY, st = p->stepT;
    double scoreMax = p->scoreThr;
    double best[3] = { out->sensorPose[0], out->sensorPose[1], out->sensorPose[2] };
    bestIdx[0] = bestIdx[1] = bestIdx[2] = -1;
    long long evals = 0;
    int iy = 0;
    for (double dy = -ry; dy <= ry; dy += sy, ++iy) {
        int ix = 0;
        for (double dx = -rx; dx <= rx; dx += sx, ++ix) {
            int it = 0;
            for (double dt = -rt; dt <= rt; dt += st, ++it) {
                const double pose[3] = { out->sensorPose[0] + dx, out->sensorPose[1] + dy,
                                         out->sensorPose[2] + dt };
                const ScoreSummary sc = score_pixel_accurate(g, lut, s, pose);
                ++evals;
                if (sc.normalized > scoreMax && sc.knownRate > p->knownThr) {
                    scoreMax = sc.normalized;
                    best[0] = pose[0];
                    best[1] = pose[1];
                    best[2] = pose[2];
                    bestIdx[0] = ix;
                    bestIdx[1] = iy;
                    bestIdx[2] = it;
                }
            }
        }
    }
    out->found = scoreMax > p->scoreThr;
    out->scoreMax = scoreMax;
    out->stepX = sx;
    out->stepY = sy;
    out->stepT = st;
    for (int k = 0; k < 3; ++k)
        out->bestSensorPose[k] = best[k];
    orc_move_backward(out->bestSensorPose, s.rel, out->estimatedPose);
    out->fineEvaluated = evals;
    if (numEvaluations)
        *numEvaluations = evals;
    return 0;
}

/* f64 score of one pose on one level with per-node projection (Score() as the
 * reference calls it); for score-value checks. */
double orc_score_at(const uint16_t* level, int rows, int cols,
                    const double* geom, const double* angles,
                    const double* ranges, int n, const double pose[3],
                    int* known)
{
    Grid g { level, rows, cols, geom[0], geom[1], geom[2] };
    OrcScan s { angles, ranges, n, { 0, 0, 0 } };
    const ScoreSummary sc = score_pixel_accurate(g, shared_lut(), s, pose);
    if (known)
        *known = sc.known;
    return sc.normalized;
}

} /* extern "C" */
