/* csm_matchers.hpp -- what the matchers' host translation units share (csm_plan.hip: planner, launch
 * helpers, levels; csm_window.hip: one window at a time; csm_batch.hip: batches; csm_api.hip: context,
 * grids, pyramids, host restatements, timing). All of it lives in namespace csm_host. */
#ifndef CSM_MATCHERS_HPP
#define CSM_MATCHERS_HPP

#include "csm_internal.hpp"

#include "csm_launch.hpp"
#include "csm_joint.hpp"
#include "csm_phase.hpp"

namespace csm_host {

const int kCoarseSlices = 8;

/* Launch geometry of one scoring pass (one level of one window shape). */
struct PassPlan {
    int nx = 0, ny = 0, stride = 1, log2s = 0;
    int cbx = 0, groups = 0, R = 0, ncbx = 0, ncby = 0, lstride = 0;
    bool weighted = true;     /* entries carry beam multiplicities */
    bool pairs = false;       /* pair-row fine kernel (k_score_pairs): lstride = slots per pair row */
    int lists = 1;            /* entry lists in LDS: 2 = the batch kernel that takes two slices per workgroup */
    bool joint = false;       /* ... on joint entries of the two slices (one list; csm_joint_kernels.hip) */
    bool fp32 = false;        /* this launch is the packed-fp32 bound pass of the joint kernel */
    int list_lds = -1;        /* LDS bytes of the entry lists if not lists * kPbMax words (plan_pass_pairs) */
    int ncb() const { return ncbx * ncby; }
};

/* Launch geometry of one search window. */
struct Plan {
    int n_theta = 0, n = 0;
    int win_x = 0, win_y = 0, L = 1;
    int nxc = 0, nyc = 0, nx = 0, ny = 0;
    int x_lo = 0, y_lo = 0, x_hi = 0, y_hi = 0;
    PassPlan fine, coarse;
    int tiles_x = 0, tiles_y = 0, max_tiles = 0;
};

/* Work list of the exact joint kernel after the bound pass (k_bound_select): items of the main
 * launch, items of the R = 6 tail launch, their counts, workgroups to share them. */
struct JointList {
    const uint32_t* items[2] = { nullptr, nullptr };
    const uint32_t* counts = nullptr;       /* [2] */
    int blocks = 0;
};

/* Box-maximum levels to build: collected first, launched together (launch_box_jobs). */
struct PendingBox {
    DeviceGrid* grid;
    int level;          /* index into grid->levels: its cells are the destination */
};

struct WindowOutputs {
    uint32_t* dump_s = nullptr;     /* device */
    uint16_t* dump_k = nullptr;
};

/* The CSM pipeline on device-resident inputs; asynchronous. */
/* Two-phase search (csm_phase_kernels.hip). mode 1: score the window and STORE every candidate's
 * sums [n_theta][nx][ny] (the coarse pass, run on the level's phase-major copy), nothing else;
 * mode 2: the fine level, its eligibility from such sums (level_s / level_k with strides nxs, nys),
 * over the work list of the blocks that can still win. */
struct TwoPhaseCtl {
    int mode = 0;
    uint32_t* level_s = nullptr;
    uint32_t* level_k = nullptr;
    int nxs = 0, nys = 0;
    uint32_t stats[4] = { 0, 0, 0, 0 };      /* mode 2, filled on request: items, kept, dropped */
};

/* Splits [0, n) over up to four host threads (the batch entries touch tens of
 * megabytes of scan data before anything can be launched); fn(lo, hi) must not
 * touch the context. */
template <class F>
void host_parallel_for(int n, int grain, F fn)
{
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int nt = std::min(std::min(4, hw), n / std::max(1, grain));
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> workers;
    for (int w = 1; w < nt; ++w)
        workers.emplace_back(fn, (int)((long)n * w / nt), (int)((long)n * (w + 1) / nt));
    fn(0, (int)((long)n / nt));
    for (auto& t : workers)
        t.join();
}

double value_to_probability(unsigned v);
int proj_theta_groups(int n_theta, long blocks_xz);
bool scan_finite_max(const csm_scan* scan, double* max_range);
void search_step_from_max(double resolution, double max_range, double* step_x, double* step_y,
                          double* step_theta);
int scans_finite_max(const csm_loop_query* queries, int n_queries, double* max_range);
bool merging_pays(const double* angles, const double* ranges, int n, double res);
int bin_hash_size(int n_points);
size_t bin_lds_bytes(int tiles, int n_points);
int ilog2_exact(int v);
bool plan_pass(const Tuning& tune, int nx, int ny, int stride, PassPlan* out);
size_t pair_lds_bytes(int ls, int cby, int lists);
bool plan_pass_pairs(const Tuning& tune, int nx, int ny, PassPlan* out, bool two_slices = false, int list_lds = -1);
int xgrid_pad_for(int nx, int ny);
int pick_buffers(const Tuning& tune, size_t lds_one, long blocks);
size_t pass_lds_bytes(const PassPlan& p);
int make_plan(csm_ctx* ctx, const DeviceGrid& g, const csm_window* w, Plan* p);
int launched_ok(csm_ctx* ctx, int e, const char* what);
csm_launch::ScoreLaunch score_launch(const csm_ctx* ctx, const PassPlan& pp, dim3 grid, size_t lds);
int lane_map_for(csm_ctx* ctx, const PassPlan& pp, const uint16_t** out);
int launch_score(csm_ctx* ctx, const ScoreJob& job, const PassPlan& pp, int n_theta, int n_slices);
int launch_score_list(csm_ctx* ctx, const ScoreJob& job, const PassPlan& pp, const uint32_t* items,
                      const uint32_t* count, int blocks);
int launch_argmax(csm_ctx* ctx, const ScoreJob& job, const PassPlan& plan, int n_theta);
bool tail_split(const csm_ctx* ctx, const PassPlan& pp);
int launch_pairs_batch(csm_ctx* ctx, const ScoreJob* jobs_dev, const PassPlan& pp, dim3 grid, BlockBase bb,
                       const JointList* list = nullptr, int which = 0);
int launch_score_batch(csm_ctx* ctx, const ScoreJob* jobs_dev, int n_jobs, const PassPlan& pp,
                       int n_theta_max, int n_slices, int theta_groups = 0, const JointList* list = nullptr);
int launch_box_jobs(csm_ctx* ctx, const std::vector<PendingBox>& pending);
int build_level(csm_ctx* ctx, DeviceGrid& g, int win, Level* out, uint16_t* reuse = nullptr,
                size_t reuse_cap = 0);
int level_for_window(csm_ctx* ctx, DeviceGrid& g, int win, int* index,
                     std::vector<PendingBox>* pending = nullptr);
int ensure_xgrid(csm_ctx* ctx, DeviceGrid& g, int need_pad);
int ensure_xgrid_f(csm_ctx* ctx, DeviceGrid& g);
int run_level_pass_joint(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                         const int32_t* hit_col_dev, const int32_t* hit_row_dev, uint32_t* flags, TwoPhaseCtl* tp);
int run_window(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
               const int32_t* hit_col_dev, const int32_t* hit_row_dev,
               csm_result* out_dev, const WindowOutputs* dumps, bool force_coarse = false,
               TwoPhaseCtl* tp = nullptr);
int ensure_phase_map(csm_ctx* ctx, DeviceGrid& g, int level, int need, PhaseMap** out);
bool wants_two_phase(const csm_ctx* ctx, const Plan& p);
int search_window(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p, const int32_t* col_dev,
                  const int32_t* row_dev, csm_result* out_dev);
int resolve_ties(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                 const int32_t* col_dev, const int32_t* row_dev, csm_result* out_dev);
int resolve_literal(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                    const int32_t* col_dev, const int32_t* row_dev, csm_result* out_dev);
int resolve_window(csm_ctx* ctx, DeviceGrid& g, const csm_window* w, const Plan& p,
                   const int32_t* col_dev, const int32_t* row_dev, csm_result* out_dev,
                   const csm_result* have = nullptr, bool* changed = nullptr);

} /* namespace csm_host */
#endif
