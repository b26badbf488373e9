/* csm_adapters.hpp -- header-only C++17 host side above the C ABI
 * (include/csm_hip.h). Mirrors the reference's plugin interfaces for the hot
 * path with the same names, argument meaning and error behaviour:
 *
 *   ScanMatcherCorrelativeHIP   <- ScanMatcherCorrelative
 *        inc/mapping/scan_matcher_correlative.hpp:53-125, scan_matcher.hpp:89-117
 *   LoopDetectorBranchBoundHIP  <- LoopDetectorBranchBound (search part)
 *        inc/mapping/loop_detector_branch_bound.hpp:71-112, loop_detector.hpp:97-116
 *   LoopDetectorCorrelativeHIP  <- LoopDetectorCorrelative (search part)
 *        inc/mapping/loop_detector_correlative.hpp, src/mapping/loop_detector_correlative.cpp:59-156
 *   ScanMatcherGridSearchHIP    <- ScanMatcherGridSearch
 *        inc/mapping/scan_matcher_grid_search.hpp, src/mapping/scan_matcher_grid_search.cpp:69-190
 *   GridMapBuilderHIP           <- GridMapBuilder (latest-map part)
 *        inc/mapping/grid_map_builder.hpp, src/mapping/grid_map_builder.cpp:497-527, 561-695
 *
 * The reference headers cannot be included in this image (Eigen3 / Boost are
 * absent), so the few value types the interfaces use are restated here in
 * namespace CsmHip. INTEGRATION.md shows the ~40-line glue a maintainer adds
 * inside the reference tree to derive these from the real
 * MyLidarGraphSlam::Mapping::ScanMatcher / LoopDetector.
 *
 * Error convention: like the reference's Assert() (inc/util.hpp:38-72) a
 * failed C-ABI call prints csm_last_error() with file:line and abort()s;
 * device-initialisation failure is reported from Create() as a null pointer,
 * like LoadBitstream's bool (src/slam_launcher.cpp:83-107).
 */
#ifndef CSM_ADAPTERS_HPP
#define CSM_ADAPTERS_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "csm_hip.h"

namespace CsmHip {

#define CSM_ASSERT_OK(ctx, expr)                                                   \
    do {                                                                           \
        const int rc_ = (expr);                                                    \
        if (rc_ != 0) {                                                            \
            std::fprintf(stderr, "Assertion failed: %s == 0 (rc %d: %s) at %s:%d\n", \
                         #expr, rc_, csm_last_error(ctx), __FILE__, __LINE__);     \
            std::abort();                                                          \
        }                                                                          \
    } while (0)

/* inc/pose.hpp:17-46 */
template <typename T>
struct RobotPose2D {
    T mX, mY, mTheta;
};

/* What a matcher reads of GridMap (inc/grid_map_new/grid_map.hpp): the dense
 * export of CopyValues (src/grid_map_new/grid_map.cpp:439-457) plus geometry
 * (inc/grid_map_new/grid_map_geometry.hpp:228-240). mId plays LocalMapId::mId;
 * kInvalidId marks a throw-away map (the frontend's latest map), as
 * LocalMapId::Invalid does in scan_matcher_correlative_fpga.cpp:177-184. */
struct GridMapView {
    static constexpr std::uint64_t kInvalidId = ~0ull;
    /* ids from here on are the adapters' own (throw-away maps): a caller's id must be smaller */
    static constexpr std::uint64_t kReservedIds = 1ull << 62;
    const std::uint16_t* mValues = nullptr;   /* row-major rows*cols */
    int mRows = 0, mCols = 0;
    double mResolution = 0.0;
    double mPosOffsetX = 0.0, mPosOffsetY = 0.0;
    std::uint64_t mId = kInvalidId;
    /* A map with an id is uploaded once and then read from the device (the
     * reference caches by LocalMapId because finished local maps never change,
     * loop_detector_branch_bound.hpp:98). A caller that passes a map which is
     * still growing under an id bumps mRevision whenever its cells change: the
     * adapters upload again when the revision differs from the one they hold. */
    std::uint64_t mRevision = 0;
};

/* What a matcher reads of Sensor::ScanData<double>
 * (inc/sensor/sensor_data.hpp:63-185) */
struct ScanDataView {
    const double* mAngles = nullptr;
    const double* mRanges = nullptr;
    std::size_t mNumOfScans = 0;
    RobotPose2D<double> mRelativeSensorPose { 0.0, 0.0, 0.0 };
};

/* inc/mapping/scan_matcher.hpp:27-50 */
struct ScanMatchingQuery {
    GridMapView mGridMap;
    ScanDataView mScanData;
    RobotPose2D<double> mMapLocalInitialPose;
};

/* inc/mapping/scan_matcher.hpp:53-82. Cost and covariance come from the
 * caller's CostFunction (scan_matcher_correlative.cpp:209-219): pass a
 * callback, or leave them zero. */
struct ScanMatchingSummary {
    bool mPoseFound = false;
    double mNormalizedCost = 0.0;
    RobotPose2D<double> mMapLocalInitialPose { 0, 0, 0 };
    RobotPose2D<double> mEstimatedPose { 0, 0, 0 };
    double mEstimatedCovariance[9] = { 0 };
    /* extras the reference observes as metrics
     * (scan_matcher_correlative.cpp:222-236) */
    RobotPose2D<double> mBestSensorPose { 0, 0, 0 };
    double mScoreValue = 0.0;
    int mWinSizeX = 0, mWinSizeY = 0, mWinSizeTheta = 0;
    double mStepSizeX = 0, mStepSizeY = 0, mStepSizeTheta = 0;
    double mInputSetupTime = 0, mOptimizationTime = 0;   /* micro seconds */
    long long mNumOfCandidates = 0;
    std::uint32_t mFlags = 0;
};

/* Cost / covariance hook: void(query, bestSensorPose, &normalizedCost, cov[9]) */
using CostCallback = void (*)(const ScanMatchingQuery&, const RobotPose2D<double>&, double*,
                              double*);

namespace detail {
/* which revision of which map id a context holds */
using RevisionMap = std::map<std::uint64_t, std::uint64_t>;

struct CtxDeleter {
    void operator()(csm_ctx* c) const { if (c) csm_destroy(c); }
};
using CtxPtr = std::unique_ptr<csm_ctx, CtxDeleter>;

inline CtxPtr MakeContext(int deviceId)
{
    csm_config cfg {};
    cfg.device_id = deviceId;
    csm_ctx* raw = nullptr;
    if (csm_create(&cfg, &raw) != 0)
        return CtxPtr();
    return CtxPtr(raw);
}

inline csm_scan ToScan(const ScanDataView& s)
{
    csm_scan out {};
    out.angles = s.mAngles;
    out.ranges = s.mRanges;
    out.n_points = static_cast<std::int32_t>(s.mNumOfScans);
    out.relative_sensor_pose[0] = s.mRelativeSensorPose.mX;
    out.relative_sensor_pose[1] = s.mRelativeSensorPose.mY;
    out.relative_sensor_pose[2] = s.mRelativeSensorPose.mTheta;
    return out;
}

inline void FillSummary(const csm_summary& s, const RobotPose2D<double>& initial,
                        ScanMatchingSummary* out)
{
    out->mPoseFound = s.pose_found != 0;
    out->mMapLocalInitialPose = initial;
    out->mEstimatedPose = { s.estimated_pose[0], s.estimated_pose[1], s.estimated_pose[2] };
    out->mBestSensorPose = { s.best_sensor_pose[0], s.best_sensor_pose[1], s.best_sensor_pose[2] };
    out->mScoreValue = s.raw.score;
    out->mWinSizeX = s.win_x;
    out->mWinSizeY = s.win_y;
    out->mWinSizeTheta = s.win_theta;
    out->mStepSizeX = s.step_x;
    out->mStepSizeY = s.step_y;
    out->mStepSizeTheta = s.step_theta;
    out->mInputSetupTime = s.input_setup_us;
    out->mOptimizationTime = s.optimization_us;
    out->mNumOfCandidates = s.candidates;
    out->mFlags = s.raw.flags;
}
} /* namespace detail */

/* The 15 value sequences a matcher registers with the MetricManager and observes
 * once per call (src/mapping/scan_matcher_correlative.cpp:37-70, 222-236), under
 * the same ids: observe("<matcherName>.InputSetupTime", value), ... Ignored /
 * processed nodes lose their meaning when every candidate is scored: reported
 * as 0 and the number of candidates. The glue in the reference tree passes
 * [&](const std::string& id, double v) { its ValueSequence for id ->Observe(v); }. */
template <typename Observe>
inline void ReportScanMatcherMetrics(const std::string& matcherName, const ScanMatchingSummary& s,
                                     std::size_t numOfScans, Observe&& observe)
{
    const double dx = s.mEstimatedPose.mX - s.mMapLocalInitialPose.mX;
    const double dy = s.mEstimatedPose.mY - s.mMapLocalInitialPose.mY;
    observe(matcherName + ".InputSetupTime", s.mInputSetupTime);
    observe(matcherName + ".OptimizationTime", s.mOptimizationTime);
    observe(matcherName + ".DiffTranslation", std::sqrt(dx * dx + dy * dy));
    observe(matcherName + ".DiffRotation", std::abs(s.mMapLocalInitialPose.mTheta - s.mEstimatedPose.mTheta));
    observe(matcherName + ".WinSizeX", static_cast<double>(s.mWinSizeX));
    observe(matcherName + ".WinSizeY", static_cast<double>(s.mWinSizeY));
    observe(matcherName + ".WinSizeTheta", static_cast<double>(s.mWinSizeTheta));
    observe(matcherName + ".StepSizeX", s.mStepSizeX);
    observe(matcherName + ".StepSizeY", s.mStepSizeY);
    observe(matcherName + ".StepSizeTheta", s.mStepSizeTheta);
    observe(matcherName + ".NumOfIgnoredNodes", 0.0);
    observe(matcherName + ".NumOfProcessedNodes", static_cast<double>(s.mNumOfCandidates));
    observe(matcherName + ".ScoreValue", static_cast<double>(static_cast<float>(s.mScoreValue)));   /* a float sequence */
    observe(matcherName + ".CostValue", s.mNormalizedCost);
    observe(matcherName + ".NumOfScans", static_cast<double>(numOfScans));
}

class ScanMatcherCorrelativeHIP final {
public:
    /* Constructor arguments as ScanMatcherCorrelative
     * (inc/mapping/scan_matcher_correlative.hpp:58-66); Create() returns null
     * when no usable GPU exists. */
    static std::unique_ptr<ScanMatcherCorrelativeHIP> Create(
        const std::string& scanMatcherName, int lowResolution, double rangeX, double rangeY,
        double rangeTheta, CostCallback costFunc = nullptr, int deviceId = 0)
    {
        detail::CtxPtr ctx = detail::MakeContext(deviceId);
        if (!ctx)
            return nullptr;
        return std::unique_ptr<ScanMatcherCorrelativeHIP>(new ScanMatcherCorrelativeHIP(
            scanMatcherName, lowResolution, rangeX, rangeY, rangeTheta, costFunc, std::move(ctx)));
    }

    ScanMatcherCorrelativeHIP(const ScanMatcherCorrelativeHIP&) = delete;
    ScanMatcherCorrelativeHIP& operator=(const ScanMatcherCorrelativeHIP&) = delete;

    const std::string& Name() const { return this->mName; }
    /* the device context, to share resident maps with a GridMapBuilderHIP */
    csm_ctx* Context() const { return this->mCtx.get(); }

    /* ScanMatcher::OptimizePose (scan_matcher_correlative.cpp:92-115): the whole
     * window, thresholds 0.0 / 0.0 */
    ScanMatchingSummary OptimizePose(const ScanMatchingQuery& queryInfo)
    {
        return this->OptimizePose(queryInfo, 0.0, 0.0);
    }

    /* The 6-argument overload the loop detectors call
     * (scan_matcher_correlative.cpp:118-244); the coarse map is built and cached
     * on the device instead of being passed in. */
    ScanMatchingSummary OptimizePose(const ScanMatchingQuery& q,
                                     const double normalizedScoreThreshold,
                                     const double knownRateThreshold)
    {
        csm_ctx* ctx = this->mCtx.get();
        const GridMapView& g = q.mGridMap;
        const bool temporary = g.mId == GridMapView::kInvalidId;
        if (!temporary && g.mId >= GridMapView::kReservedIds) {
            std::fprintf(stderr, "Assertion failed: map id below 2^62 at %s:%d\n", __FILE__, __LINE__);
            std::abort();
        }
        const std::uint64_t id = temporary ? GridMapView::kReservedIds : g.mId;
        auto held = this->mRevisions.find(id);
        /* mValues == nullptr: the map is already resident (built by a GridMapBuilderHIP on this context) */
        if (g.mValues && (temporary || !csm_has_grid(ctx, id) || held == this->mRevisions.end() ||
                          held->second != g.mRevision)) {
            CSM_ASSERT_OK(ctx, csm_upload_grid(ctx, id, g.mValues, g.mRows, g.mCols));
            this->mRevisions[id] = g.mRevision;
        }
        csm_geometry geom { g.mResolution, g.mPosOffsetX, g.mPosOffsetY };
        const csm_scan scan = detail::ToScan(q.mScanData);
        csm_correlative_params prm {};
        prm.range_x = this->mRangeX;
        prm.range_y = this->mRangeY;
        prm.range_theta = this->mRangeTheta;
        prm.low_resolution = this->mLowResolution;
        prm.score_threshold = normalizedScoreThreshold;
        prm.known_rate_threshold = knownRateThreshold;
        const double init[3] = { q.mMapLocalInitialPose.mX, q.mMapLocalInitialPose.mY,
                                 q.mMapLocalInitialPose.mTheta };
        csm_summary s {};
        CSM_ASSERT_OK(ctx, csm_correlative_match(ctx, id, &geom, &scan, init, &prm, &s));
        ScanMatchingSummary out;
        detail::FillSummary(s, q.mMapLocalInitialPose, &out);
        if (this->mCostFunc) {
            this->mCostFunc(q, out.mBestSensorPose, &out.mNormalizedCost, out.mEstimatedCovariance);
        } else if (this->mDeviceCovarianceScale > 0.0) {
            /* CostSquareError::Cost / ComputeCovariance at the best sensor pose on the device
             * (scan_matcher_correlative.cpp:209-219) */
            csm_loop_query cq {};
            cq.map_id = id;
            cq.geometry = geom;
            cq.scan = scan;
            csm_refine_result rr {};
            CSM_ASSERT_OK(ctx, csm_cost_covariance_batch(ctx, &cq, 1, s.best_sensor_pose,
                                                         this->mDeviceCovarianceScale, &rr));
            out.mNormalizedCost = rr.normalized_cost;
            for (int c = 0; c < 9; ++c)
                out.mEstimatedCovariance[c] = rr.covariance[c];
        }
        if (temporary && g.mValues)
            CSM_ASSERT_OK(ctx, csm_release_grid(ctx, id));
        return out;
    }

    /* Cost and covariance from the device's CostSquareError instead of a host callback
     * ("CovarianceScale", launcher_settings_default.json:11-13). */
    void UseDeviceCostFunction(double covarianceScale = 1e4) { this->mDeviceCovarianceScale = covarianceScale; }

private:
    ScanMatcherCorrelativeHIP(const std::string& name, int lowResolution, double rangeX,
                              double rangeY, double rangeTheta, CostCallback costFunc,
                              detail::CtxPtr ctx) :
        mName(name), mLowResolution(lowResolution), mRangeX(rangeX), mRangeY(rangeY),
        mRangeTheta(rangeTheta), mCostFunc(costFunc), mCtx(std::move(ctx)) { }

    const std::string mName;
    const int mLowResolution;
    const double mRangeX, mRangeY, mRangeTheta;
    const CostCallback mCostFunc;
    detail::CtxPtr mCtx;
    detail::RevisionMap mRevisions;
    double mDeviceCovarianceScale = 0.0;
};

/* ScanMatcherGridSearch (inc/mapping/scan_matcher_grid_search.hpp,
 * src/mapping/scan_matcher_grid_search.cpp:69-190): the brute-force matcher of
 * LoopDetectorGridSearch; the pixel-accurate score function is the device
 * kernel, the cost / covariance hook stays with the caller. */
class ScanMatcherGridSearchHIP final {
public:
    static std::unique_ptr<ScanMatcherGridSearchHIP> Create(
        const std::string& scanMatcherName, double rangeX, double rangeY, double rangeTheta,
        double stepX, double stepY, double stepTheta, CostCallback costFunc = nullptr,
        int deviceId = 0)
    {
        if (!(stepX > 0.0) || !(stepY > 0.0) || !(stepTheta > 0.0))
            return nullptr;
        detail::CtxPtr ctx = detail::MakeContext(deviceId);
        if (!ctx)
            return nullptr;
        return std::unique_ptr<ScanMatcherGridSearchHIP>(new ScanMatcherGridSearchHIP(
            scanMatcherName, rangeX, rangeY, rangeTheta, stepX, stepY, stepTheta, costFunc,
            std::move(ctx)));
    }

    const std::string& Name() const { return this->mName; }

    /* scan_matcher_grid_search.cpp:69-81 */
    ScanMatchingSummary OptimizePose(const ScanMatchingQuery& queryInfo)
    {
        return this->OptimizePose(queryInfo, 0.0, 0.0);
    }

    /* scan_matcher_grid_search.cpp:84-190 */
    ScanMatchingSummary OptimizePose(const ScanMatchingQuery& q,
                                     const double normalizedScoreThreshold,
                                     const double knownRateThreshold)
    {
        csm_ctx* ctx = this->mCtx.get();
        const GridMapView& g = q.mGridMap;
        const bool temporary = g.mId == GridMapView::kInvalidId;
        if (!temporary && g.mId >= GridMapView::kReservedIds) {
            std::fprintf(stderr, "Assertion failed: map id below 2^62 at %s:%d\n", __FILE__, __LINE__);
            std::abort();
        }
        const std::uint64_t id = temporary ? GridMapView::kReservedIds + 1 : g.mId;
        auto held = this->mRevisions.find(id);
        if (g.mValues && (temporary || !csm_has_grid(ctx, id) || held == this->mRevisions.end() ||
                          held->second != g.mRevision)) {
            CSM_ASSERT_OK(ctx, csm_upload_grid(ctx, id, g.mValues, g.mRows, g.mCols));
            this->mRevisions[id] = g.mRevision;
        }
        csm_geometry geom { g.mResolution, g.mPosOffsetX, g.mPosOffsetY };
        const csm_scan scan = detail::ToScan(q.mScanData);
        const csm_grid_search_params prm { this->mRangeX, this->mRangeY, this->mRangeTheta,
                                           this->mStepX, this->mStepY, this->mStepTheta,
                                           normalizedScoreThreshold, knownRateThreshold };
        const double init[3] = { q.mMapLocalInitialPose.mX, q.mMapLocalInitialPose.mY,
                                 q.mMapLocalInitialPose.mTheta };
        csm_summary s {};
        CSM_ASSERT_OK(ctx, csm_grid_search_match(ctx, id, &geom, &scan, init, &prm, &s));
        if (temporary)
            CSM_ASSERT_OK(ctx, csm_release_grid(ctx, id));
        ScanMatchingSummary out;
        detail::FillSummary(s, q.mMapLocalInitialPose, &out);
        if (this->mCostFunc)
            this->mCostFunc(q, out.mBestSensorPose, &out.mNormalizedCost, out.mEstimatedCovariance);
        return out;
    }

private:
    ScanMatcherGridSearchHIP(const std::string& name, double rangeX, double rangeY,
                             double rangeTheta, double stepX, double stepY, double stepTheta,
                             CostCallback costFunc, detail::CtxPtr ctx) :
        mName(name), mRangeX(rangeX), mRangeY(rangeY), mRangeTheta(rangeTheta), mStepX(stepX),
        mStepY(stepY), mStepTheta(stepTheta), mCostFunc(costFunc), mCtx(std::move(ctx)) { }

    const std::string mName;
    const double mRangeX, mRangeY, mRangeTheta;
    const double mStepX, mStepY, mStepTheta;
    const CostCallback mCostFunc;
    detail::CtxPtr mCtx;
    detail::RevisionMap mRevisions;
};

/* inc/mapping/loop_detector.hpp:27-55, flattened to what the search reads:
 * the reference local map (finished, immutable, keyed by LocalMapId) and the
 * query scan node's scan + its pose local to that map
 * (InverseCompound(localMapNode.mGlobalPose, scanNode.mGlobalPose),
 * loop_detector_branch_bound.cpp:97-98 -- the caller passes both global poses). */
struct LoopDetectionQuery {
    GridMapView mReferenceLocalMap;
    ScanDataView mQueryScanData;
    RobotPose2D<double> mQueryScanNodeGlobalPose;
    RobotPose2D<double> mReferenceLocalMapNodeGlobalPose;
    int mQueryScanNodeId = 0;
};
using LoopDetectionQueryVector = std::vector<LoopDetectionQuery>;

/* inc/mapping/loop_detector.hpp:58-92. Without a final matcher
 * (UseFinalScanMatcher) mRelativePose is the search's estimate and the
 * covariance is zero; with one they are ScanMatcherLinearSolver's
 * (loop_detector_branch_bound.cpp:123-135). */
struct LoopDetectionResult {
    RobotPose2D<double> mRelativePose;   /* estimated pose, map-local */
    RobotPose2D<double> mLocalMapPose;
    std::uint64_t mLocalMapNodeId;
    int mScanNodeId;
    double mScoreValue;
    std::uint32_t mFlags;
    double mEstimatedCovariance[9] = { 0 };   /* row-major */
    double mNormalizedCost = 0.0;
};
using LoopDetectionResultVector = std::vector<LoopDetectionResult>;

class LoopDetectorBranchBoundHIP final {
public:
    /* scoreThreshold / knownRateThreshold as LoopDetectorBranchBound
     * (src/mapping/loop_detector_branch_bound.cpp:38-56); nodeHeightMax and the
     * search ranges as its ScanMatcherBranchBound
     * (src/scan_matcher_factory.cpp:22-26). deviceIds: the GPUs the detector
     * spreads a Detect() call over -- contiguous blocks of the query vector, one
     * host thread per GPU inside the library, as LoopDetectorFPGAParallel does
     * with its two FPGA cores (src/mapping/loop_detector_fpga_parallel.cpp:42-56). */
    static std::unique_ptr<LoopDetectorBranchBoundHIP> Create(
        const std::string& loopDetectorName, int nodeHeightMax, double rangeX, double rangeY,
        double rangeTheta, double scoreThreshold, double knownRateThreshold,
        const std::vector<int>& deviceIds)
    {
        if (!(scoreThreshold > 0.0 && scoreThreshold <= 1.0) ||
            !(knownRateThreshold > 0.0 && knownRateThreshold <= 1.0) || deviceIds.empty())
            return nullptr;
        std::vector<std::int32_t> ids(deviceIds.begin(), deviceIds.end());
        csm_group* group = nullptr;
        if (csm_group_create(ids.data(), static_cast<std::int32_t>(ids.size()), &group) != 0)
            return nullptr;
        return std::unique_ptr<LoopDetectorBranchBoundHIP>(new LoopDetectorBranchBoundHIP(
            loopDetectorName, nodeHeightMax, rangeX, rangeY, rangeTheta, scoreThreshold,
            knownRateThreshold, group));
    }

    static std::unique_ptr<LoopDetectorBranchBoundHIP> Create(
        const std::string& loopDetectorName, int nodeHeightMax, double rangeX, double rangeY,
        double rangeTheta, double scoreThreshold, double knownRateThreshold, int deviceId = 0)
    {
        return Create(loopDetectorName, nodeHeightMax, rangeX, rangeY, rangeTheta, scoreThreshold,
                      knownRateThreshold, std::vector<int> { deviceId });
    }

    ~LoopDetectorBranchBoundHIP() { csm_group_destroy(this->mGroup); }
    LoopDetectorBranchBoundHIP(const LoopDetectorBranchBoundHIP&) = delete;
    LoopDetectorBranchBoundHIP& operator=(const LoopDetectorBranchBoundHIP&) = delete;

    const std::string& Name() const { return this->mName; }
    int NumOfDevices() const { return csm_group_size(this->mGroup); }

    /* The detector's final matcher, ScanMatcherLinearSolver on CostSquareError
     * ("FinalScanMatcherLinearSolver", launcher_settings_default.json:148-155, 11-13),
     * run on the device for all found queries of a Detect() call. */
    void UseFinalScanMatcher(int numOfIterationsMax = 10, double convergenceThreshold = 1e-4,
                             double initialLambda = 1e-4, double covarianceScale = 1e4)
    {
        this->mRefine = true;
        this->mRefineParams.iterations_max = numOfIterationsMax;
        this->mRefineParams.convergence_threshold = convergenceThreshold;
        this->mRefineParams.lambda = initialLambda;
        this->mRefineParams.covariance_scale = covarianceScale;
    }

    /* LoopDetector::Detect: results only for the queries where a pose was
     * found, in query order (loop_detector_branch_bound.cpp:107-135). */
    LoopDetectionResultVector Detect(const LoopDetectionQueryVector& queries)
    {
        LoopDetectionResultVector results;
        if (queries.empty())
            return results;
        const std::int32_t n = static_cast<std::int32_t>(queries.size());
        const std::int32_t members = csm_group_size(this->mGroup);
        std::vector<csm_loop_query> flat(queries.size());
        for (std::int32_t k = 0; k < members; ++k) {
            std::int32_t lo = 0, hi = 0;
            csm_shard_bounds(n, k, members, &lo, &hi);
            csm_ctx* ctx = csm_group_member(this->mGroup, k);
            for (std::int32_t i = lo; i < hi; ++i) {
                const LoopDetectionQuery& q = queries[i];
                const GridMapView& g = q.mReferenceLocalMap;
                /* a loop detector only sees finished local maps, which have an id and never
                 * change (Assert(localMap.mFinished), loop_detector_branch_bound.cpp:77) */
                if (g.mId == GridMapView::kInvalidId) {
                    std::fprintf(stderr, "Assertion failed: reference local map without an id at %s:%d\n",
                                 __FILE__, __LINE__);
                    std::abort();
                }
                /* upload once per id and member (mPrecompMaps, loop_detector_branch_bound.hpp:98) */
                if (!csm_has_grid(ctx, g.mId))
                    CSM_ASSERT_OK(ctx, csm_upload_grid(ctx, g.mId, g.mValues, g.mRows, g.mCols));
                csm_loop_query& f = flat[i];
                f.map_id = g.mId;
                f.geometry = { g.mResolution, g.mPosOffsetX, g.mPosOffsetY };
                f.scan = detail::ToScan(q.mQueryScanData);
                const double start[3] = { q.mReferenceLocalMapNodeGlobalPose.mX,
                                          q.mReferenceLocalMapNodeGlobalPose.mY,
                                          q.mReferenceLocalMapNodeGlobalPose.mTheta };
                const double end[3] = { q.mQueryScanNodeGlobalPose.mX, q.mQueryScanNodeGlobalPose.mY,
                                        q.mQueryScanNodeGlobalPose.mTheta };
                csm_host_inverse_compound(start, end, f.initial_pose);
            }
        }
        csm_bnb_params prm {};
        prm.range_x = this->mRangeX;
        prm.range_y = this->mRangeY;
        prm.range_theta = this->mRangeTheta;
        prm.node_height_max = this->mNodeHeightMax;
        prm.score_threshold = this->mScoreThreshold;
        prm.known_rate_threshold = this->mKnownRateThreshold;
        std::vector<csm_summary> out(queries.size());
        const int rc = csm_group_bnb_match_batch(this->mGroup, flat.data(), n, &prm, out.data());
        if (rc != 0) {
            std::fprintf(stderr, "Assertion failed: csm_group_bnb_match_batch == 0 (rc %d: %s) at %s:%d\n", rc,
                         csm_group_last_error(this->mGroup), __FILE__, __LINE__);
            std::abort();
        }
        /* the final matcher on every estimate that was found, member by member (each member
         * holds the maps of its own block): loop_detector_branch_bound.cpp:119-127 */
        std::vector<csm_refine_result> refined(queries.size());
        if (this->mRefine) {
            for (std::int32_t k = 0; k < members; ++k) {
                std::int32_t lo = 0, hi = 0;
                csm_shard_bounds(n, k, members, &lo, &hi);
                std::vector<csm_loop_query> second;
                std::vector<std::int32_t> index;
                for (std::int32_t i = lo; i < hi; ++i) {
                    if (!out[i].pose_found)
                        continue;
                    csm_loop_query q = flat[i];
                    for (int c = 0; c < 3; ++c)
                        q.initial_pose[c] = out[i].estimated_pose[c];
                    second.push_back(q);
                    index.push_back(i);
                }
                if (second.empty())
                    continue;
                std::vector<csm_refine_result> res(second.size());
                csm_ctx* ctx = csm_group_member(this->mGroup, k);
                CSM_ASSERT_OK(ctx, csm_linear_solver_batch(ctx, second.data(),
                                                           static_cast<std::int32_t>(second.size()),
                                                           &this->mRefineParams, res.data()));
                for (std::size_t j = 0; j < index.size(); ++j)
                    refined[index[j]] = res[j];
                /* the solver object keeps its damping factor between calls */
                this->mRefineParams.lambda = res.back().lambda;
            }
        }
        for (std::size_t i = 0; i < queries.size(); ++i) {
            if (!out[i].pose_found)
                continue;
            LoopDetectionResult r {
                { out[i].estimated_pose[0], out[i].estimated_pose[1], out[i].estimated_pose[2] },
                queries[i].mReferenceLocalMapNodeGlobalPose, queries[i].mReferenceLocalMap.mId,
                queries[i].mQueryScanNodeId, out[i].raw.score, out[i].raw.flags };
            if (this->mRefine) {
                r.mRelativePose = { refined[i].estimated_pose[0], refined[i].estimated_pose[1],
                                    refined[i].estimated_pose[2] };
                for (int c = 0; c < 9; ++c)
                    r.mEstimatedCovariance[c] = refined[i].covariance[c];
                r.mNormalizedCost = refined[i].normalized_cost;
            }
            results.push_back(r);
        }
        return results;
    }

private:
    LoopDetectorBranchBoundHIP(const std::string& name, int nodeHeightMax, double rangeX,
                               double rangeY, double rangeTheta, double scoreThreshold,
                               double knownRateThreshold, csm_group* group) :
        mName(name), mNodeHeightMax(nodeHeightMax), mRangeX(rangeX), mRangeY(rangeY),
        mRangeTheta(rangeTheta), mScoreThreshold(scoreThreshold),
        mKnownRateThreshold(knownRateThreshold), mGroup(group) { }

    const std::string mName;
    const int mNodeHeightMax;
    const double mRangeX, mRangeY, mRangeTheta;
    const double mScoreThreshold, mKnownRateThreshold;
    csm_group* mGroup;
    bool mRefine = false;
    csm_refine_params mRefineParams {};
};

/* LoopDetectorCorrelative (the reference's default "RealTimeCorrelative" loop
 * detector, launcher_settings_default.json:101-114, 394): the correlative
 * matcher with the detector's thresholds, one coarse map cached per local map. */
class LoopDetectorCorrelativeHIP final {
public:
    static std::unique_ptr<LoopDetectorCorrelativeHIP> Create(
        const std::string& loopDetectorName, int lowResolution, double rangeX, double rangeY,
        double rangeTheta, double scoreThreshold, double knownRateThreshold, int deviceId = 0)
    {
        /* src/mapping/loop_detector_correlative.cpp:38-56 */
        if (!(scoreThreshold > 0.0 && scoreThreshold <= 1.0) ||
            !(knownRateThreshold > 0.0 && knownRateThreshold <= 1.0) || lowResolution < 1)
            return nullptr;
        detail::CtxPtr ctx = detail::MakeContext(deviceId);
        if (!ctx)
            return nullptr;
        return std::unique_ptr<LoopDetectorCorrelativeHIP>(new LoopDetectorCorrelativeHIP(
            loopDetectorName, lowResolution, rangeX, rangeY, rangeTheta, scoreThreshold,
            knownRateThreshold, std::move(ctx)));
    }

    const std::string& Name() const { return this->mName; }

    LoopDetectionResultVector Detect(const LoopDetectionQueryVector& queries)
    {
        LoopDetectionResultVector results;
        if (queries.empty())
            return results;
        csm_ctx* ctx = this->mCtx.get();
        std::vector<csm_loop_query> flat(queries.size());
        for (std::size_t i = 0; i < queries.size(); ++i) {
            const LoopDetectionQuery& q = queries[i];
            const GridMapView& g = q.mReferenceLocalMap;
            if (!csm_has_grid(ctx, g.mId))
                CSM_ASSERT_OK(ctx, csm_upload_grid(ctx, g.mId, g.mValues, g.mRows, g.mCols));
            csm_loop_query& f = flat[i];
            f.map_id = g.mId;
            f.geometry = { g.mResolution, g.mPosOffsetX, g.mPosOffsetY };
            f.scan = detail::ToScan(q.mQueryScanData);
            const double start[3] = { q.mReferenceLocalMapNodeGlobalPose.mX,
                                      q.mReferenceLocalMapNodeGlobalPose.mY,
                                      q.mReferenceLocalMapNodeGlobalPose.mTheta };
            const double end[3] = { q.mQueryScanNodeGlobalPose.mX, q.mQueryScanNodeGlobalPose.mY,
                                    q.mQueryScanNodeGlobalPose.mTheta };
            csm_host_inverse_compound(start, end, f.initial_pose);
        }
        csm_correlative_params prm {};
        prm.range_x = this->mRangeX;
        prm.range_y = this->mRangeY;
        prm.range_theta = this->mRangeTheta;
        prm.low_resolution = this->mLowResolution;
        prm.score_threshold = this->mScoreThreshold;
        prm.known_rate_threshold = this->mKnownRateThreshold;
        std::vector<csm_summary> out(queries.size());
        CSM_ASSERT_OK(ctx, csm_correlative_match_batch(
                               ctx, flat.data(), static_cast<std::int32_t>(flat.size()), &prm, out.data()));
        for (std::size_t i = 0; i < queries.size(); ++i) {
            if (!out[i].pose_found)
                continue;
            results.push_back(LoopDetectionResult {
                { out[i].estimated_pose[0], out[i].estimated_pose[1], out[i].estimated_pose[2] },
                queries[i].mReferenceLocalMapNodeGlobalPose, queries[i].mReferenceLocalMap.mId,
                queries[i].mQueryScanNodeId, out[i].raw.score, out[i].raw.flags });
        }
        return results;
    }

private:
    LoopDetectorCorrelativeHIP(const std::string& name, int lowResolution, double rangeX,
                               double rangeY, double rangeTheta, double scoreThreshold,
                               double knownRateThreshold, detail::CtxPtr ctx) :
        mName(name), mLowResolution(lowResolution), mRangeX(rangeX), mRangeY(rangeY),
        mRangeTheta(rangeTheta), mScoreThreshold(scoreThreshold),
        mKnownRateThreshold(knownRateThreshold), mCtx(std::move(ctx)) { }

    const std::string mName;
    const int mLowResolution;
    const double mRangeX, mRangeY, mRangeTheta;
    const double mScoreThreshold, mKnownRateThreshold;
    detail::CtxPtr mCtx;
};

/* What the map update reads of a ScanNode (inc/mapping/pose_graph.hpp) and its
 * ScanData (inc/sensor/sensor_data.hpp:63-185) */
struct ScanNodeView {
    int mNodeId = 0;
    RobotPose2D<double> mGlobalPose { 0.0, 0.0, 0.0 };
    ScanDataView mScanData;
    double mMinRange = 0.0, mMaxRange = 0.0;
};

/* The latest-map half of GridMapBuilder (src/mapping/grid_map_builder.cpp): the
 * map the frontend matches every new scan against is rebuilt from the last
 * mNumOfScansForLatestMap scans on every call (UpdateLatestMap, :497-527).
 * Here it is built on the device and stays there under one map id, so the
 * matcher needs no upload; pass the matcher's Context() to share it. */
class GridMapBuilderHIP final {
public:
    /* constructor arguments as GridMapBuilder (grid_map_builder.cpp:68-99) minus
     * the local-map ones; `ctx` is borrowed */
    GridMapBuilderHIP(csm_ctx* ctx, double mapResolution, int patchSize, int numOfScansForLatestMap,
                      double usableRangeMin, double usableRangeMax, double probHit,
                      double probMiss, std::uint64_t latestMapId = (1ull << 61)) :
        mCtx(ctx), mNumOfScansForLatestMap(numOfScansForLatestMap), mLatestMapPose { 0.0, 0.0, 0.0 }
    {
        this->mParams = { usableRangeMin, usableRangeMax, probHit, probMiss, 100 };   /* SubpixelScale */
        /* GridMap(resolution, patchSize, 1.0, 1.0) (grid_map.cpp:75-98, 224-246) */
        int log2Block = 0;
        while ((1 << log2Block) < patchSize)
            ++log2Block;
        const int block = 1 << log2Block;
        const int desired = static_cast<int>(std::ceil(1.0 / mapResolution));
        const int cells = ((desired + block - 1) >> log2Block) << log2Block;
        this->mShape = { mapResolution, 0.0, 0.0, cells, cells, log2Block };
        this->mInitialCells = cells;
        this->mLatestMap.mId = latestMapId;
        this->SyncView();
    }

    /* geometry + id of the latest map; mValues is null: the cells live on the device */
    const GridMapView& LatestMap() const { return this->mLatestMap; }
    const RobotPose2D<double>& LatestMapPose() const { return this->mLatestMapPose; }
    const csm_map_build_info& LastBuildInfo() const { return this->mInfo; }

    /* GridMapBuilder::UpdateLatestMap (grid_map_builder.cpp:497-527); scanNodes in id order */
    void UpdateLatestMap(const std::vector<ScanNodeView>& scanNodes)
    {
        if (scanNodes.empty()) {
            std::fprintf(stderr, "Assertion failed: !scanNodes.empty() at %s:%d\n", __FILE__, __LINE__);
            std::abort();
        }
        const std::size_t count = std::min(scanNodes.size(),
                                           static_cast<std::size_t>(this->mNumOfScansForLatestMap));
        const ScanNodeView* first = scanNodes.data() + (scanNodes.size() - count);
        this->mLatestMapPose = first->mGlobalPose;
        this->ConstructMapFromScans(this->mLatestMapPose, first, count);
    }

    /* GridMapBuilder::ConstructMapFromScans (grid_map_builder.cpp:561-695) into the latest map */
    void ConstructMapFromScans(const RobotPose2D<double>& globalMapPose, const ScanNodeView* nodes,
                               std::size_t numOfNodes)
    {
        std::vector<csm_scan_node> flat(numOfNodes);
        for (std::size_t i = 0; i < numOfNodes; ++i) {
            flat[i].global_pose[0] = nodes[i].mGlobalPose.mX;
            flat[i].global_pose[1] = nodes[i].mGlobalPose.mY;
            flat[i].global_pose[2] = nodes[i].mGlobalPose.mTheta;
            flat[i].scan = detail::ToScan(nodes[i].mScanData);
            flat[i].min_range = nodes[i].mMinRange;
            flat[i].max_range = nodes[i].mMaxRange;
        }
        const double pose[3] = { globalMapPose.mX, globalMapPose.mY, globalMapPose.mTheta };
        CSM_ASSERT_OK(this->mCtx, csm_construct_map_from_scans(
                                      this->mCtx, this->mLatestMap.mId, &this->mShape, pose, flat.data(),
                                      static_cast<std::int32_t>(numOfNodes), &this->mParams, &this->mInfo));
        this->SyncView();
    }

    /* GridMap::CopyValues of the latest map (grid_map.cpp:439-457) */
    std::vector<std::uint16_t> CopyLatestMapValues() const
    {
        std::vector<std::uint16_t> values(static_cast<std::size_t>(this->mShape.rows) * this->mShape.cols);
        CSM_ASSERT_OK(this->mCtx, csm_download_level(this->mCtx, this->mLatestMap.mId, 0, values.data()));
        return values;
    }

    /* A new, empty local map on the device: GridMap(resolution, patchSize, 1.0, 1.0)
     * as UpdatePoseGraph creates it (grid_map_builder.cpp:251) */
    void CreateLocalMap(std::uint64_t localMapId)
    {
        csm_map_shape shape = this->mShape;
        shape.offset_x = shape.offset_y = 0.0;
        shape.rows = shape.cols = this->mInitialCells;
        const std::vector<std::uint16_t> empty(static_cast<std::size_t>(shape.rows) * shape.cols, 0);
        CSM_ASSERT_OK(this->mCtx, csm_upload_grid(this->mCtx, localMapId, empty.data(), shape.rows, shape.cols));
        this->mLocalShapes[localMapId] = shape;
    }

    /* the grid half of GridMapBuilder::UpdateGridMap (grid_map_builder.cpp:389-494):
     * the latest scan node into the local map that is being built */
    void UpdateGridMap(std::uint64_t localMapId, const RobotPose2D<double>& globalMapPose,
                       const ScanNodeView& latestScanNode)
    {
        auto it = this->mLocalShapes.find(localMapId);
        if (it == this->mLocalShapes.end()) {
            std::fprintf(stderr, "Assertion failed: local map %llu exists at %s:%d\n",
                         static_cast<unsigned long long>(localMapId), __FILE__, __LINE__);
            std::abort();
        }
        csm_scan_node flat {};
        flat.global_pose[0] = latestScanNode.mGlobalPose.mX;
        flat.global_pose[1] = latestScanNode.mGlobalPose.mY;
        flat.global_pose[2] = latestScanNode.mGlobalPose.mTheta;
        flat.scan = detail::ToScan(latestScanNode.mScanData);
        flat.min_range = latestScanNode.mMinRange;
        flat.max_range = latestScanNode.mMaxRange;
        const double pose[3] = { globalMapPose.mX, globalMapPose.mY, globalMapPose.mTheta };
        CSM_ASSERT_OK(this->mCtx, csm_update_map_with_scan(this->mCtx, localMapId, &it->second, pose, &flat,
                                                           &this->mParams, &this->mInfo));
    }

    /* geometry + id of a local map (cells on the device), e.g. for a LoopDetectionQuery */
    GridMapView LocalMap(std::uint64_t localMapId) const
    {
        const csm_map_shape& shape = this->mLocalShapes.at(localMapId);
        GridMapView view;
        view.mRows = shape.rows;
        view.mCols = shape.cols;
        view.mResolution = shape.resolution;
        view.mPosOffsetX = shape.offset_x;
        view.mPosOffsetY = shape.offset_y;
        view.mId = localMapId;
        return view;
    }

    std::vector<std::uint16_t> CopyLocalMapValues(std::uint64_t localMapId) const
    {
        const csm_map_shape& shape = this->mLocalShapes.at(localMapId);
        std::vector<std::uint16_t> values(static_cast<std::size_t>(shape.rows) * shape.cols);
        CSM_ASSERT_OK(this->mCtx, csm_download_level(this->mCtx, localMapId, 0, values.data()));
        return values;
    }

private:
    void SyncView()
    {
        this->mLatestMap.mRows = this->mShape.rows;
        this->mLatestMap.mCols = this->mShape.cols;
        this->mLatestMap.mResolution = this->mShape.resolution;
        this->mLatestMap.mPosOffsetX = this->mShape.offset_x;
        this->mLatestMap.mPosOffsetY = this->mShape.offset_y;
    }

    csm_ctx* mCtx;
    int mNumOfScansForLatestMap;
    RobotPose2D<double> mLatestMapPose;
    csm_map_builder_params mParams {};
    csm_map_shape mShape {};
    csm_map_build_info mInfo {};
    GridMapView mLatestMap;
    int mInitialCells = 0;
    std::map<std::uint64_t, csm_map_shape> mLocalShapes;
};

} /* namespace CsmHip */
#endif /* CSM_ADAPTERS_HPP */
