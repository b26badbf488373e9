/* adapter_demo.cpp -- drives the C++ adapters of csm_adapters.hpp from a
 * binary case file written by tests/test_gpu_adapter.py and prints one JSON
 * line. Built with g++ against libcsm_hip.so; exercises the same entry points a
 * reference-side ScanMatcher / LoopDetector subclass would.
 *
 * case file (little endian):
 *   int32 mode(0 csm, 1 bnb loop detector, 2 correlative loop detector, 3 grid search),
 *         rows, cols, n, n_queries, param_i (L or H)
 *   (mode 3 only: double step[3] right after the 8 doubles below)
 *   double res, offX, offY, rangeX, rangeY, rangeT, scoreThr, knownThr
 *   double rel[3]; double init[3 * n_queries]; double angles[n]; double ranges[n]
 *   uint16 grid[rows * cols]
 */
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "csm_adapters.hpp"

using namespace CsmHip;

template <typename T>
static bool rd(FILE* f, T* p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

/* mode 4: GridMapBuilderHIP + ScanMatcherCorrelativeHIP sharing one device
 * context. File: int32 4, n_nodes, n_beams, patch_size, n_latest, L;
 * double res, usable_min, usable_max, prob_hit, prob_miss, range_x, range_y, range_t;
 * double rel[3]; double init_local[3]; per node: double pose[3], min_range,
 * max_range, angles[n_beams], ranges[n_beams]. Builds the latest map, prints its
 * geometry and an FNV-1a hash of its cells, then matches the last node's scan. */
static int run_builder(FILE* f, const int32_t* hdr)
{
    const int nNodes = hdr[1], nBeams = hdr[2], patch = hdr[3], nLatest = hdr[4], lowRes = hdr[5];
    double prm[8], rel[3], init[3];
    if (!rd(f, prm, 8) || !rd(f, rel, 3) || !rd(f, init, 3))
        return 2;
    std::vector<std::vector<double>> angles(nNodes), ranges(nNodes);
    std::vector<ScanNodeView> nodes(nNodes);
    for (int k = 0; k < nNodes; ++k) {
        double head[5];
        angles[k].resize(nBeams);
        ranges[k].resize(nBeams);
        if (!rd(f, head, 5) || !rd(f, angles[k].data(), nBeams) || !rd(f, ranges[k].data(), nBeams))
            return 2;
        nodes[k].mNodeId = k;
        nodes[k].mGlobalPose = { head[0], head[1], head[2] };
        nodes[k].mMinRange = head[3];
        nodes[k].mMaxRange = head[4];
        nodes[k].mScanData.mAngles = angles[k].data();
        nodes[k].mScanData.mRanges = ranges[k].data();
        nodes[k].mScanData.mNumOfScans = nBeams;
        nodes[k].mScanData.mRelativeSensorPose = { rel[0], rel[1], rel[2] };
    }
    auto matcher = ScanMatcherCorrelativeHIP::Create("demo", lowRes, prm[5], prm[6], prm[7]);
    if (!matcher) {
        std::printf("{\"error\": \"no device\"}\n");
        return 3;
    }
    GridMapBuilderHIP builder(matcher->Context(), prm[0], patch, nLatest, prm[1], prm[2], prm[3], prm[4]);
    builder.UpdateLatestMap(nodes);
    const csm_map_build_info latestInfo = builder.LastBuildInfo();
    const GridMapView& map = builder.LatestMap();
    const std::vector<uint16_t> cells = builder.CopyLatestMapValues();
    uint64_t hash = 1469598103934665603ull;
    for (uint16_t v : cells) {
        hash = (hash ^ (v & 0xff)) * 1099511628211ull;
        hash = (hash ^ (v >> 8)) * 1099511628211ull;
    }
    ScanMatchingQuery q { map, nodes.back().mScanData, { init[0], init[1], init[2] } };
    const ScanMatchingSummary r = matcher->OptimizePose(q);
    /* a local map grown scan by scan (UpdateGridMap), its pose = the first node's */
    builder.CreateLocalMap(7);
    for (const ScanNodeView& nd : nodes)
        builder.UpdateGridMap(7, nodes.front().mGlobalPose, nd);
    const GridMapView local = builder.LocalMap(7);
    uint64_t localHash = 1469598103934665603ull;
    for (uint16_t v : builder.CopyLocalMapValues(7)) {
        localHash = (localHash ^ (v & 0xff)) * 1099511628211ull;
        localHash = (localHash ^ (v >> 8)) * 1099511628211ull;
    }
    std::printf("{\"rows\": %d, \"cols\": %d, \"off\": [\"%a\", \"%a\"], \"hash\": \"%016" PRIx64 "\", "
                "\"rays\": %lld, \"updates\": %lld, \"map_pose\": [\"%a\", \"%a\", \"%a\"], "
                "\"found\": %d, \"pose\": [\"%a\", \"%a\", \"%a\"], \"score\": \"%a\", "
                "\"local\": {\"rows\": %d, \"cols\": %d, \"off\": [\"%a\", \"%a\"], \"hash\": \"%016" PRIx64 "\"}}\n",
                map.mRows, map.mCols, map.mPosOffsetX, map.mPosOffsetY, hash,
                (long long)latestInfo.rays, (long long)latestInfo.cell_updates,
                builder.LatestMapPose().mX, builder.LatestMapPose().mY, builder.LatestMapPose().mTheta,
                r.mPoseFound ? 1 : 0, r.mEstimatedPose.mX, r.mEstimatedPose.mY, r.mEstimatedPose.mTheta,
                r.mScoreValue, local.mRows, local.mCols, local.mPosOffsetX, local.mPosOffsetY, localHash);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2)
        return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f)
        return 2;
    int32_t hdr[6];
    double prm[8], rel[3];
    double step[3] = { 0.0, 0.0, 0.0 };
    if (!rd(f, hdr, 6))
        return 2;
    if (hdr[0] == 4) {
        const int rc = run_builder(f, hdr);
        std::fclose(f);
        return rc;
    }
    if (false || !rd(f, prm, 8) || (hdr[0] == 3 && !rd(f, step, 3)) || !rd(f, rel, 3))
        return 2;
    const int mode = hdr[0], rows = hdr[1], cols = hdr[2], n = hdr[3], nq = hdr[4], pi = hdr[5];
    std::vector<double> init(3 * nq), angles(n), ranges(n);
    std::vector<uint16_t> grid((size_t)rows * cols);
    if (!rd(f, init.data(), init.size()) || !rd(f, angles.data(), n) || !rd(f, ranges.data(), n) ||
        !rd(f, grid.data(), grid.size()))
        return 2;
    std::fclose(f);

    GridMapView g;
    g.mValues = grid.data();
    g.mRows = rows;
    g.mCols = cols;
    g.mResolution = prm[0];
    g.mPosOffsetX = prm[1];
    g.mPosOffsetY = prm[2];
    ScanDataView s;
    s.mAngles = angles.data();
    s.mRanges = ranges.data();
    s.mNumOfScans = n;
    s.mRelativeSensorPose = { rel[0], rel[1], rel[2] };

    if (mode == 3) {
        auto m = ScanMatcherGridSearchHIP::Create("demo", prm[3], prm[4], prm[5], step[0], step[1], step[2]);
        if (!m) {
            std::printf("{\"error\": \"no device\"}\n");
            return 3;
        }
        ScanMatchingQuery q { g, s, { init[0], init[1], init[2] } };
        const ScanMatchingSummary r = m->OptimizePose(q, prm[6], prm[7]);
        std::printf("{\"found\": %d, \"pose\": [\"%a\", \"%a\", \"%a\"], \"score\": \"%a\", \"win\": [%d, %d, %d]}\n",
                    r.mPoseFound ? 1 : 0, r.mEstimatedPose.mX, r.mEstimatedPose.mY,
                    r.mEstimatedPose.mTheta, r.mScoreValue, r.mWinSizeX, r.mWinSizeY, r.mWinSizeTheta);
        return 0;
    }
    if (mode == 0) {
        auto m = ScanMatcherCorrelativeHIP::Create("demo", pi, prm[3], prm[4], prm[5]);
        if (!m) {
            std::printf("{\"error\": \"no device\"}\n");
            return 3;
        }
        ScanMatchingQuery q { g, s, { init[0], init[1], init[2] } };
        const ScanMatchingSummary r = (prm[6] == 0.0 && prm[7] == 0.0)
                                          ? m->OptimizePose(q)
                                          : m->OptimizePose(q, prm[6], prm[7]);
        int metricIds = 0;
        ReportScanMatcherMetrics(m->Name(), r, s.mNumOfScans, [&](const std::string& id, double) {
            metricIds += id.rfind("demo.", 0) == 0;
        });
        if (metricIds != 15)
            return 4;
        std::printf("{\"found\": %d, \"pose\": [\"%a\", \"%a\", \"%a\"], \"score\": \"%a\", \"win\": [%d, %d, %d]}\n",
                    r.mPoseFound ? 1 : 0, r.mEstimatedPose.mX, r.mEstimatedPose.mY,
                    r.mEstimatedPose.mTheta, r.mScoreValue, r.mWinSizeX, r.mWinSizeY, r.mWinSizeTheta);
        return 0;
    }
    /* CSM_DEMO_DEVICES="0,0": the detector over a device list (here two members on GPU 0) */
    std::vector<int> devices;
    if (const char* e = std::getenv("CSM_DEMO_DEVICES"))
        for (const char* p = e; *p;) {
            devices.push_back(std::atoi(p));
            while (*p && *p != ',')
                ++p;
            if (*p == ',')
                ++p;
        }
    if (devices.empty())
        devices.push_back(0);
    auto d = mode == 1 ? LoopDetectorBranchBoundHIP::Create("demo", pi, prm[3], prm[4], prm[5], prm[6], prm[7],
                                                            devices)
                       : nullptr;
    auto dc = mode == 2 ? LoopDetectorCorrelativeHIP::Create("demo", pi, prm[3], prm[4], prm[5], prm[6], prm[7])
                        : nullptr;
    if (!d && !dc) {
        std::printf("{\"error\": \"no device\"}\n");
        return 3;
    }
    const bool refine = std::getenv("CSM_DEMO_REFINE") != nullptr;   /* the detector's final matcher */
    if (d && refine)
        d->UseFinalScanMatcher();
    g.mId = 42;
    LoopDetectionQueryVector qs;
    for (int i = 0; i < nq; ++i) {
        LoopDetectionQuery q;
        q.mReferenceLocalMap = g;
        q.mQueryScanData = s;
        /* local map node at the origin: the scan node's global pose is its map-local pose */
        q.mReferenceLocalMapNodeGlobalPose = { 0.0, 0.0, 0.0 };
        q.mQueryScanNodeGlobalPose = { init[3 * i], init[3 * i + 1], init[3 * i + 2] };
        q.mQueryScanNodeId = i;
        qs.push_back(q);
    }
    const LoopDetectionResultVector rs = d ? d->Detect(qs) : dc->Detect(qs);
    std::printf("{\"results\": [");
    for (size_t i = 0; i < rs.size(); ++i)
        std::printf("%s{\"node\": %d, \"pose\": [\"%a\", \"%a\", \"%a\"], \"score\": \"%a\", "
                    "\"cost\": \"%a\", \"cov00\": \"%a\"}", i ? ", " : "",
                    rs[i].mScanNodeId, rs[i].mRelativePose.mX, rs[i].mRelativePose.mY,
                    rs[i].mRelativePose.mTheta, rs[i].mScoreValue, rs[i].mNormalizedCost,
                    rs[i].mEstimatedCovariance[0]);
    std::printf("]}\n");
    return 0;
}
