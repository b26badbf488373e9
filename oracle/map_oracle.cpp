/* map_oracle.cpp
 *
 * TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's map building
 * step that feeds the scan matcher:
 *   GridMapBuilder::ConstructMapFromScans   src/.../mapping/grid_map_builder.cpp:561-695
 *   GridMapBuilder::ComputeMissedIndicesScaled                       ...:891-911
 *   BresenhamScaled                          src/.../bresenham.cpp:58-237
 *   GridMap<T>::Resize (both overloads)      src/.../grid_map_new/grid_map.cpp:841-913
 *   GridMap<T>::IndexToBlock                                         ...:804-814
 *   GridMapGeometry::Resize / PositionToIndex / ScaledGeometry
 *                                            src/.../grid_map_new/grid_map_geometry.cpp:46-72, 113-122
 *   GridBinaryBayes::UpdateOddsUnchecked     src/.../grid_map_new/grid_binary_bayes.cpp:302-321
 *   GridBinaryBayes::ProbabilityToValue / ProbabilityToOdds / OddsToProbability  ...:345-383
 *   ComputeValueToOddsLookup                 src/.../grid_values.cpp:50-86
 * Used only by tests/. Nothing in my-lidar-graph-slam-v2_amd/ links or calls it.
 *
 * PARITY STATUS: "parity unpinned" -- the reference has no fixtures for this
 * step and bresenham.cpp / grid_map.cpp include util.hpp (Eigen), so they
 * cannot be compiled here. Pinned pieces: the conversions the cell update is
 * made of (ProbabilityToValue, ProbabilityToOdds, OddsToProbability,
 * ValueToOdds) are inline in grid_map_new/grid_values.hpp, which compiles from
 * the reference tree as it lies; tests/test_cpu_map_oracle.py checks the bb_*
 * functions below against them bit for bit (oracle/_ref and the committed
 * tests/golden/ref_geometry.json), as it does for the pose algebra and the hit
 * point. The ray walk is the reference's step-by-step formulation; tests
 * cross-check it against an independent closed form.
 *
 * The dense array stands for GridMap<GridBinaryBayes>: unallocated blocks read
 * as 0 (unknown) and ConstructMapFromScans resets every kept block, so the
 * block structure only matters for the geometry, which is restated exactly.
 * Build with -ffp-contract=off (the reference is plain x86-64, no FMA).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

extern "C" {
void orc_compound(const double s[3], const double d[3], double out[3]);
void orc_inverse_compound(const double s[3], const double e[3], double out[3]);
void orc_hit_point(const double pose[3], double range, double angle, double out[2]);
}

namespace {

const double kProbMin = 1e-3;
const double kProbMax = 1.0 - 1e-3;

/* grid_binary_bayes.cpp:359-369 */
double bb_probability_to_odds(double prob)
{
    if (prob == 0.0)
        return 1.0;
    if (prob < kProbMin)
        return kProbMin / (1.0 - kProbMin);
    if (prob > kProbMax)
        return kProbMax / (1.0 - kProbMax);
    return prob / (1.0 - prob);
}

/* grid_binary_bayes.cpp:372-383 */
double bb_odds_to_probability(double odds)
{
    if (odds < 0.0)
        return 0.0;
    const double prob = odds / (1.0 + odds);
    return std::clamp(prob, kProbMin, kProbMax);
}

/* grid_binary_bayes.cpp:345-356 + grid_values.hpp:11-23 (the double lands in a
 * std::uint16_t return value: truncation) */
uint16_t bb_probability_to_value(double prob)
{
    if (prob == 0.0)
        return 0;
    if (prob < kProbMin)
        return 1;
    if (prob > kProbMax)
        return 65535;
    return static_cast<uint16_t>(1 + (prob - kProbMin) * static_cast<double>(65535 - 1) /
                                         (kProbMax - kProbMin));
}

/* ValueToOddsLookup[v] (grid_values.cpp:50-86): 1.0 for unknown, else
 * ValueToOdds. The reference table has 65535 entries, so v = 65535 is out of
 * bounds there; the formula is extended and every such read is counted. */
double bb_value_to_odds(unsigned v, long long* oobReads)
{
    if (v == 0)
        return 1.0;
    if (v == 65535 && oobReads)
        ++*oobReads;
    const double prob = kProbMin + (kProbMax - kProbMin) *
                        static_cast<double>(static_cast<int>(v) - 1) /
                        static_cast<double>(65535 - 1);
    return prob / (1.0 - prob);
}

/* grid_binary_bayes.cpp:302-321 */
void bb_update(uint16_t* cell, double odds, long long* oobReads)
{
    if (*cell == 0) {
        *cell = bb_probability_to_value(bb_odds_to_probability(odds));
        return;
    }
    const double oldOdds = bb_value_to_odds(*cell, oobReads);
    *cell = bb_probability_to_value(bb_odds_to_probability(oldOdds * odds));
}

typedef std::pair<int, int> Cell;   /* (x = column, y = row) */

/* bresenham.cpp:58-237. The reference handles a start to the right of the end
 * by calling itself with the two swapped, and has one loop for rising and one
 * for non-rising rays that mirror each other (subY > denominator <-> subY < 0,
 * subY == denominator <-> subY == 0); here the falling case runs the rising
 * loop on the mirrored sub-row position. */
void ray_cells(int sx, int sy, int ex, int ey, int scale, std::vector<Cell>& out)
{
    out.clear();
    if (sx > ex) {
        std::swap(sx, ex);
        std::swap(sy, ey);
    }
    const int startX = sx / scale, startY = sy / scale;
    const int endX = ex / scale, endY = ey / scale;
    auto visit = [&out](int x, int y) {
        if (out.empty() || out.back() != Cell(x, y))
            out.emplace_back(x, y);
    };
    if (startX == endX) {                              /* :87-99 */
        for (int y = std::min(startY, endY); y <= std::max(startY, endY); ++y)
            visit(startX, y);
        return;
    }
    const int64_t dx = ex - sx;
    const int64_t dy = ey - sy;
    const int64_t denom = 2 * static_cast<int64_t>(scale) * dx;
    const int up = dy > 0 ? 1 : -1;
    const int64_t rise = dy > 0 ? dy : -dy;            /* per unit of x, towards `up` */
    /* distance of the sub-row centre from the cell edge the ray leaves behind */
    int64_t sub = (2 * (sy % scale) + 1) * dx;         /* :125 */
    if (up < 0)
        sub = denom - sub;
    const int headSpan = 2 * scale - (2 * (sx % scale) + 1);   /* :131-133 */
    const int tailSpan = 2 * (ex % scale) + 1;
    int x = startX, y = startY;
    visit(x, y);
    sub += rise * headSpan;                          /* :139 */
    for (;;) {                                         /* :144-166 / :192-217 */
        visit(x, y);
        while (sub > denom) {
            sub -= denom;
            y += up;
            visit(x, y);
        }
        if (sub == denom) {                            /* exactly through a corner */
            sub -= denom;
            y += up;
        }
        ++x;
        if (x == endX)
            break;
        sub += 2 * rise * scale;
    }
    sub += rise * tailSpan;                           /* :169-179 / :220-230 */
    visit(x, y);
    while (sub > denom) {
        sub -= denom;
        y += up;
        visit(x, y);
    }
}

struct Shape {
    double res, offX, offY;
    int rows, cols, log2Block;
};

int position_to_index(double p, double off, double res)   /* grid_map_geometry.cpp:113-122 */
{
    return static_cast<int>(std::floor((p - off) / res));
}

int index_to_block(int idx, int log2Block)                 /* grid_map.cpp:804-814 */
{
    return idx >= 0 ? (idx >> log2Block) : ((idx >> log2Block) - 1);
}

} /* namespace */

extern "C" {

struct OrcMapShape {
    double res, offX, offY;
    int rows, cols, log2Block;
};

struct OrcScanNode {
    double pose[3];              /* ScanNode::mGlobalPose */
    const double* angles;
    const double* ranges;
    int n;
    double rel[3];               /* ScanData::RelativeSensorPose */
    double minRange, maxRange;   /* ScanData::MinRange / MaxRange */
};

struct OrcBuilderParams {
    double usableMin, usableMax, probHit, probMiss;
    int subpixel;
};

/* Steps 1-2 of ConstructMapFromScans (grid_map_builder.cpp:583-645): the
 * bounding box of the sensor positions and the usable hit points, then
 * GridMap::Resize on the map's CURRENT geometry. `shape` is updated in place. */
int orc_map_resize(OrcMapShape* shape, const double mapPose[3], const OrcScanNode* nodes,
                   int nNodes, const OrcBuilderParams* prm)
{
    if (nNodes < 1)
        return 1;
    double minX = std::numeric_limits<double>::max(), minY = minX;
    double maxX = std::numeric_limits<double>::min(), maxY = maxX;   /* sic: the smallest positive double */
    for (int k = 0; k < nNodes; ++k) {
        const OrcScanNode& nd = nodes[k];
        double gs[3], ls[3];
        orc_compound(nd.pose, nd.rel, gs);
        orc_inverse_compound(mapPose, gs, ls);
        minX = std::min(minX, ls[0]);
        minY = std::min(minY, ls[1]);
        maxX = std::max(maxX, ls[0]);
        maxY = std::max(maxY, ls[1]);
        const double minRange = std::max(prm->usableMin, nd.minRange);
        const double maxRange = std::min(prm->usableMax, nd.maxRange);
        for (int i = 0; i < nd.n; ++i) {
            const double r = nd.ranges[i];
            if (r >= maxRange || r <= minRange)
                continue;
            double hp[2];
            orc_hit_point(ls, r, nd.angles[i], hp);
            minX = std::min(minX, hp[0]);
            minY = std::min(minY, hp[1]);
            maxX = std::max(maxX, hp[0]);
            maxY = std::max(maxY, hp[1]);
        }
    }
    if (!(minX < maxX) || !(minY < maxY))
        return 2;                                  /* Assert in Resize(BoundingBox<double>) */
    /* grid_map.cpp:892-913 */
    const int iMinX = position_to_index(minX - shape->res, shape->offX, shape->res);
    const int iMinY = position_to_index(minY - shape->res, shape->offY, shape->res);
    const int iMaxX = position_to_index(maxX + shape->res, shape->offX, shape->res) + 1;
    const int iMaxY = position_to_index(maxY + shape->res, shape->offY, shape->res) + 1;
    /* grid_map.cpp:841-889 */
    const int blockSize = 1 << shape->log2Block;
    const int bMinX = index_to_block(iMinX, shape->log2Block);
    const int bMinY = index_to_block(iMinY, shape->log2Block);
    const int bMaxX = index_to_block(iMaxX + blockSize - 1, shape->log2Block);
    const int bMaxY = index_to_block(iMaxY + blockSize - 1, shape->log2Block);
    /* the reference shifts (bMin << log2Block); a multiplication avoids shifting a negative int */
    const int rowMin = bMinY * blockSize, colMin = bMinX * blockSize;
    shape->rows = (bMaxY - bMinY) << shape->log2Block;
    shape->cols = (bMaxX - bMinX) << shape->log2Block;
    shape->offX += shape->res * colMin;            /* grid_map_geometry.cpp:61-72 */
    shape->offY += shape->res * rowMin;
    return 0;
}

/* Step 3 (grid_map_builder.cpp:647-692) on a zeroed dense array of the resized
 * shape. stats[0] = rays, [1] = cell updates, [2] = reads of table entry 65535
 * (out of bounds in the reference), [3] = rays whose end cell was not on the
 * walk (the reference asserts). Returns nonzero if an update leaves the map. */
static int map_integrate(const OrcMapShape* shape, const double mapPose[3], const OrcScanNode* nodes,
                         int nNodes, const OrcBuilderParams* prm, uint16_t* grid, long long* stats,
                         bool reset);

int orc_map_integrate(const OrcMapShape* shape, const double mapPose[3], const OrcScanNode* nodes,
                      int nNodes, const OrcBuilderParams* prm, uint16_t* grid, long long* stats)
{
    return map_integrate(shape, mapPose, nodes, nNodes, prm, grid, stats, true);
}

/* The update loop of GridMapBuilder::UpdateGridMap (grid_map_builder.cpp:446-477):
 * the same ray casts onto the cells the local map already holds. */
int orc_map_integrate_keep(const OrcMapShape* shape, const double mapPose[3], const OrcScanNode* node,
                           const OrcBuilderParams* prm, uint16_t* grid, long long* stats)
{
    return map_integrate(shape, mapPose, node, 1, prm, grid, stats, false);
}

/* UpdateGridMap's bounding box + GridMap::Expand (grid_map_builder.cpp:425-434,
 * 820-872; grid_map.cpp:915-961): the box starts at the sensor position, and
 * the map is resized to the union with its current extent only if the box does
 * not fit. `shape` is updated in place; *rowMin / *colMin = index of the new
 * map's first cell in the old frame (0 when the map did not change). */
int orc_map_expand(OrcMapShape* shape, const double mapPose[3], const OrcScanNode* node,
                   const OrcBuilderParams* prm, int* rowMin, int* colMin)
{
    double gs[3], ls[3];
    orc_compound(node->pose, node->rel, gs);
    orc_inverse_compound(mapPose, gs, ls);
    double minX = ls[0], minY = ls[1], maxX = ls[0], maxY = ls[1];
    const double minRange = std::max(prm->usableMin, node->minRange);
    const double maxRange = std::min(prm->usableMax, node->maxRange);
    for (int i = 0; i < node->n; ++i) {
        const double r = node->ranges[i];
        if (r >= maxRange || r <= minRange)
            continue;
        double hp[2];
        orc_hit_point(ls, r, node->angles[i], hp);
        minX = std::min(minX, hp[0]);
        minY = std::min(minY, hp[1]);
        maxX = std::max(maxX, hp[0]);
        maxY = std::max(maxY, hp[1]);
    }
    *rowMin = *colMin = 0;
    if (!(minX < maxX) || !(minY < maxY))
        return 2;                                   /* Assert in Expand(BoundingBox<double>) */
    const int bx0 = position_to_index(minX - shape->res, shape->offX, shape->res);
    const int by0 = position_to_index(minY - shape->res, shape->offY, shape->res);
    const int bx1 = position_to_index(maxX + shape->res, shape->offX, shape->res) + 1;
    const int by1 = position_to_index(maxY + shape->res, shape->offY, shape->res) + 1;
    auto inside = [shape](int row, int col) {
        return row >= 0 && row < shape->rows && col >= 0 && col < shape->cols;
    };
    if (inside(by0, bx0) && inside(by1 - 1, bx1 - 1))
        return 0;
    const int ux0 = std::min(0, bx0), uy0 = std::min(0, by0);
    const int ux1 = std::max(shape->cols, bx1), uy1 = std::max(shape->rows, by1);
    const int blockSize = 1 << shape->log2Block;
    const int bMinX = index_to_block(ux0, shape->log2Block), bMinY = index_to_block(uy0, shape->log2Block);
    const int bMaxX = index_to_block(ux1 + blockSize - 1, shape->log2Block);
    const int bMaxY = index_to_block(uy1 + blockSize - 1, shape->log2Block);
    *rowMin = bMinY * blockSize;
    *colMin = bMinX * blockSize;
    shape->rows = (bMaxY - bMinY) << shape->log2Block;
    shape->cols = (bMaxX - bMinX) << shape->log2Block;
    shape->offX += shape->res * *colMin;
    shape->offY += shape->res * *rowMin;
    return 0;
}

static int map_integrate(const OrcMapShape* shape, const double mapPose[3], const OrcScanNode* nodes,
                         int nNodes, const OrcBuilderParams* prm, uint16_t* grid, long long* stats,
                         bool reset)
{
    const double oddsHit = bb_probability_to_odds(prm->probHit);     /* grid_map_builder.cpp:95-96 */
    const double oddsMiss = bb_probability_to_odds(prm->probMiss);
    const int scale = prm->subpixel;
    const double scaledRes = shape->res / scale;                     /* grid_map_geometry.cpp:46-58 */
    long long rays = 0, updates = 0, oob = 0, noEnd = 0;
    if (reset)
        std::memset(grid, 0, sizeof(uint16_t) * shape->rows * shape->cols);
    std::vector<Cell> walk;
    auto inside = [shape](int x, int y) {
        return x >= 0 && x < shape->cols && y >= 0 && y < shape->rows;
    };
    for (int k = 0; k < nNodes; ++k) {
        const OrcScanNode& nd = nodes[k];
        double gs[3], ls[3];
        orc_compound(nd.pose, nd.rel, gs);
        orc_inverse_compound(mapPose, gs, ls);
        const int sX = position_to_index(ls[0], shape->offX, scaledRes);
        const int sY = position_to_index(ls[1], shape->offY, scaledRes);
        const double minRange = std::max(prm->usableMin, nd.minRange);
        const double maxRange = std::min(prm->usableMax, nd.maxRange);
        for (int i = 0; i < nd.n; ++i) {
            const double r = nd.ranges[i];
            if (r >= maxRange || r <= minRange)
                continue;
            double hp[2];
            orc_hit_point(ls, r, nd.angles[i], hp);
            const int hitX = position_to_index(hp[0], shape->offX, shape->res);
            const int hitY = position_to_index(hp[1], shape->offY, shape->res);
            const int eX = position_to_index(hp[0], shape->offX, scaledRes);
            const int eY = position_to_index(hp[1], shape->offY, scaledRes);
            if (sX < 0 || sY < 0 || eX < 0 || eY < 0)
                return 3;                                          /* Asserts of bresenham.cpp:73-76 */
            ray_cells(sX, sY, eX, eY, scale, walk);
            /* grid_map_builder.cpp:904-910: drop the first entry equal to the end cell */
            const Cell endCell(eX / scale, eY / scale);
            const auto it = std::find(walk.begin(), walk.end(), endCell);
            if (it == walk.end())
                ++noEnd;
            else
                walk.erase(it);
            for (const Cell& c : walk) {
                if (!inside(c.first, c.second))
                    return 4;
                bb_update(&grid[static_cast<size_t>(c.second) * shape->cols + c.first], oddsMiss, &oob);
                ++updates;
            }
            if (!inside(hitX, hitY))
                return 4;
            bb_update(&grid[static_cast<size_t>(hitY) * shape->cols + hitX], oddsHit, &oob);
            ++updates;
            ++rays;
        }
    }
    if (stats) {
        stats[0] = rays;
        stats[1] = updates;
        stats[2] = oob;
        stats[3] = noEnd;
    }
    return 0;
}

/* The cells of one ray, for the closed-form cross-check of the tests. Returns
 * the number of cells (written up to `cap`, x then y). */
int orc_ray_cells(int sx, int sy, int ex, int ey, int scale, int* out, int cap)
{
    std::vector<Cell> walk;
    ray_cells(sx, sy, ex, ey, scale, walk);
    for (size_t i = 0; i < walk.size() && static_cast<int>(i) < cap; ++i) {
        out[2 * i] = walk[i].first;
        out[2 * i + 1] = walk[i].second;
    }
    return static_cast<int>(walk.size());
}

/* the conversions on their own, for the checks against the reference's inline
 * primitives (oracle/_ref, tests/golden/ref_geometry.json) */
unsigned orc_bb_probability_to_value(double prob) { return bb_probability_to_value(prob); }
double orc_bb_probability_to_odds(double prob) { return bb_probability_to_odds(prob); }
double orc_bb_odds_to_probability(double odds) { return bb_odds_to_probability(odds); }
double orc_bb_value_to_odds(unsigned value) { return bb_value_to_odds(value, nullptr); }

/* one cell update, for table checks: returns the new value */
unsigned orc_bayes_update(unsigned value, double prob)
{
    uint16_t v = static_cast<uint16_t>(value);
    bb_update(&v, bb_probability_to_odds(prob), nullptr);
    return v;
}

} /* extern "C" */
