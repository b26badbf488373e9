/* ref_geom_exports.cpp
 *
 * TEST INFRASTRUCTURE ONLY. Build driver for oracle/_ref/libref_geom.so: it
 * includes the reference's OWN header-only files where they lie under
 * /root/reference (pose.hpp, sensor/sensor_data.hpp,
 * grid_map_new/grid_values.hpp -- the only hot-path files that compile without
 * Eigen3/Boost, which this image lacks) and re-exports their functions with C
 * linkage so tests can pin oracle/csm_oracle.cpp against them. No reference
 * source is copied into this repo; nothing here stands in for a missing
 * header. Built only in the container (the GPU box has no /root/reference).
 */
#include <memory>
#include <string>
#include <vector>

#include "my_lidar_graph_slam/pose.hpp"
#include "my_lidar_graph_slam/sensor/sensor_data.hpp"
#include "my_lidar_graph_slam/grid_map_new/grid_values.hpp"

using namespace MyLidarGraphSlam;

extern "C" {

void ref_compound(const double s[3], const double d[3], double out[3])
{
    const auto r = Compound(RobotPose2D<double>(s[0], s[1], s[2]),
                            RobotPose2D<double>(d[0], d[1], d[2]));
    out[0] = r.mX; out[1] = r.mY; out[2] = r.mTheta;
}

void ref_inverse_compound(const double s[3], const double e[3], double out[3])
{
    const auto r = InverseCompound(RobotPose2D<double>(s[0], s[1], s[2]),
                                   RobotPose2D<double>(e[0], e[1], e[2]));
    out[0] = r.mX; out[1] = r.mY; out[2] = r.mTheta;
}

void ref_move_backward(const double e[3], const double d[3], double out[3])
{
    const auto r = MoveBackward(RobotPose2D<double>(e[0], e[1], e[2]),
                                RobotPose2D<double>(d[0], d[1], d[2]));
    out[0] = r.mX; out[1] = r.mY; out[2] = r.mTheta;
}

/* ScanData<double>::HitPoint for every beam of a scan */
void ref_hit_points(const double pose[3], const double* angles,
                    const double* ranges, int n, double* outXY)
{
    const RobotPose2D<double> zero(0.0, 0.0, 0.0);
    Sensor::ScanData<double> scan(
        "ref", 0.0, zero, zero, zero, 0.0, 1e9, -4.0, 4.0,
        std::vector<double>(angles, angles + n),
        std::vector<double>(ranges, ranges + n));
    const RobotPose2D<double> p(pose[0], pose[1], pose[2]);
    for (int i = 0; i < n; ++i) {
        const auto hp = scan.HitPoint(p, static_cast<std::size_t>(i));
        outXY[2 * i] = hp.mX;
        outXY[2 * i + 1] = hp.mY;
    }
}

/* grid_values.hpp:26-35 with the constants the grid cell classes pass
 * (grid_binary_bayes.hpp:163-176) */
double ref_value_to_probability(unsigned value)
{
    return ValueToProbability(static_cast<std::uint16_t>(value), 1U, 65535U,
                              1e-3, 1.0 - 1e-3);
}

/* the inline primitives of grid_values.hpp:11-58 that the binary-Bayes cell
 * update is made of, with the constants of grid_binary_bayes.hpp:163-176 */
unsigned ref_probability_to_value(double prob)
{
    return ProbabilityToValue(prob, 1U, 65535U, 1e-3, 1.0 - 1e-3);
}

double ref_probability_to_odds(double prob) { return ProbabilityToOdds(prob); }

double ref_odds_to_probability(double odds) { return OddsToProbability(odds); }

double ref_value_to_odds(unsigned value)
{
    return ValueToOdds(static_cast<std::uint16_t>(value), 1U, 65535U, 1e-3, 1.0 - 1e-3);
}

} /* extern "C" */
