/* csm_group.hip -- several GPUs behind one detector object, inside ONE process
 * (a translation unit of libcsm_hip.so of its own, host code only).
 *
 * The reference's precedent is LoopDetectorFPGAParallel
 * (src/mapping/loop_detector_fpga_parallel.cpp:42-56): Detect() splits the query
 * vector into contiguous halves, runs one std::thread per FPGA core and
 * concatenates the per-core result vectors. Here: one csm_ctx per device, the
 * query vector cut into contiguous blocks (csm_shard_bounds), one host thread
 * per member running the ordinary batch entry point on its block, and ONE
 * exchange step (csm_allgather_results) after which every member's device
 * buffer holds all best records in query order:
 *   - members on distinct devices: ncclAllGather over RCCL (xGMI), the blocks
 *     padded to the largest block so that one fixed-size collective suffices.
 *     librccl is loaded on first use (dlopen), so a single-GPU user of the
 *     library never needs it.
 *   - one member, or members that share a device (only useful for tests on a
 *     one-GPU box: RCCL refuses duplicate devices): device copies through the
 *     host.
 * The summaries themselves are written by the member threads straight into the
 * caller's array: one address space, no gather needed for them.
 */
#include "csm_internal.hpp"

#include <dlfcn.h>

#include <set>
#include <type_traits>

/* The RCCL entry points are taken with dlsym (librccl is only needed when a group spans several
 * devices) through hand-declared pointer types. Where the header is present at build time the
 * declarations are checked against it, so that a changed signature breaks the build, not a call. */
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) &&
                  sizeof(ncclComm_t) == sizeof(void*) && (int)ncclSuccess == 0 && (int)ncclUint8 == 1,
              "csm_group's RCCL pointer types assume int-sized enums, ncclSuccess = 0, ncclUint8 = 1");
static_assert(std::is_same<decltype(&ncclAllGather),
                           ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value,
              "ncclAllGather signature changed");
static_assert(std::is_same<decltype(&ncclCommInitAll), ncclResult_t (*)(ncclComm_t*, int, const int*)>::value,
              "ncclCommInitAll signature changed");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value,
              "ncclCommDestroy signature changed");
static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value &&
                  std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value,
              "ncclGroupStart / ncclGroupEnd signature changed");
#endif

struct csm_group {
    std::vector<csm_ctx*> members;
    std::vector<int> devices;
    std::string err;
    /* exchange */
    bool distinct = false;           /* all members on different devices */
    bool use_rccl = false;
    void* rccl = nullptr;            /* dlopen handle */
    std::vector<void*> comms;        /* ncclComm_t per member */
    std::vector<DevBuf> send, recv;  /* per member: its padded block / all blocks */
    int last_n = 0;                  /* queries of the last batch */
    double last_gather_us = 0.0;
    /* RCCL entry points (rccl.h: ncclResult_t = int, ncclDataType_t ncclUint8 = 1) */
    int (*p_comm_init_all)(void**, int, const int*) = nullptr;
    int (*p_comm_destroy)(void*) = nullptr;
    int (*p_group_start)() = nullptr;
    int (*p_group_end)() = nullptr;
    int (*p_all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*p_error_string)(int) = nullptr;
};

namespace {

int gfail(csm_group* g, int code, const char* fmt, ...)
{
    if (g) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        g->err = buf;
    }
    return code;
}

int group_load_rccl(csm_group* g)
{
    if (g->rccl)
        return CSM_OK;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names)
        if ((g->rccl = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!g->rccl)
        return gfail(g, CSM_ENODEV, "librccl not found: %s", dlerror());
    auto sym = [&](const char* s) { return dlsym(g->rccl, s); };
    g->p_comm_init_all = reinterpret_cast<int (*)(void**, int, const int*)>(sym("ncclCommInitAll"));
    g->p_comm_destroy = reinterpret_cast<int (*)(void*)>(sym("ncclCommDestroy"));
    g->p_group_start = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
    g->p_group_end = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
    g->p_all_gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(
        sym("ncclAllGather"));
    g->p_error_string = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
    if (!g->p_comm_init_all || !g->p_comm_destroy || !g->p_group_start || !g->p_group_end ||
        !g->p_all_gather)
        return gfail(g, CSM_ENODEV, "librccl lacks an entry point");
    return CSM_OK;
}

int group_init_comms(csm_group* g)
{
    if (!g->comms.empty())
        return CSM_OK;
    int rc = group_load_rccl(g);
    if (rc)
        return rc;
    g->comms.assign(g->members.size(), nullptr);
    const int res = g->p_comm_init_all(g->comms.data(), (int)g->devices.size(), g->devices.data());
    if (res != 0) {
        g->comms.clear();
        return gfail(g, CSM_EIO, "ncclCommInitAll failed: %s",
                     g->p_error_string ? g->p_error_string(res) : "?");
    }
    return CSM_OK;
}

template <typename Params, typename Fn>
int group_run_batch(csm_group* g, const csm_loop_query* queries, int32_t n, const Params* prm,
                    csm_summary* out, Fn entry)
{
    if (!g || !queries || n < 1 || !prm || !out)
        return gfail(g, CSM_EINVAL, "bad arguments");
    const int m = (int)g->members.size();
    std::vector<int> rcs(m, CSM_OK);
    std::vector<std::thread> workers;
    for (int k = 0; k < m; ++k) {
        int32_t lo = 0, hi = 0;
        csm_shard_bounds(n, k, m, &lo, &hi);
        if (hi <= lo) {
            g->members[k]->rec_n = 0;
            continue;
        }
        /* one host thread per member, as loop_detector_fpga_parallel.cpp:42-50 */
        workers.emplace_back([=, &rcs]() {
            rcs[k] = entry(g->members[k], queries + lo, hi - lo, prm, out + lo);
        });
    }
    for (std::thread& w : workers)
        w.join();
    for (int k = 0; k < m; ++k)
        if (rcs[k])
            return gfail(g, rcs[k], "member %d (device %d): %s", k, g->devices[k],
                         csm_last_error(g->members[k]));
    g->last_n = n;
    /* the exchange step: afterwards every member's device buffer holds all records */
    std::vector<csm_result> gathered((size_t)n);
    int rc = csm_allgather_results(g, gathered.data());
    if (rc)
        return rc;
    for (int i = 0; i < n; ++i)
        out[i].raw = gathered[i];          /* what came through the gather is what is returned */
    return CSM_OK;
}

} /* namespace */

extern "C" {

void csm_shard_bounds(int32_t n_queries, int32_t member, int32_t n_members, int32_t* lo, int32_t* hi)
{
    if (n_members < 1 || member < 0 || member >= n_members || n_queries < 0) {
        *lo = *hi = 0;
        return;
    }
    const int32_t base = n_queries / n_members, rem = n_queries % n_members;
    *lo = member * base + std::min(member, rem);
    *hi = *lo + base + (member < rem ? 1 : 0);
}

int csm_group_create(const int32_t* device_ids, int32_t n_devices, csm_group** out)
{
    return csm_group_create_ex(device_ids, n_devices, nullptr, 0u, out);
}

int csm_group_create_ex(const int32_t* device_ids, int32_t n_devices, const csm_config* member_cfg,
                        uint32_t flags, csm_group** out)
{
    if (!out)
        return CSM_EINVAL;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return CSM_EINVAL;
    csm_group* g = new csm_group();
    for (int k = 0; k < n_devices; ++k) {
        csm_config cfg {};
        if (member_cfg)
            cfg = *member_cfg;
        cfg.device_id = device_ids[k];
        csm_ctx* ctx = nullptr;
        const int rc = csm_create(&cfg, &ctx);
        if (rc) {
            for (csm_ctx* c : g->members)
                csm_destroy(c);
            delete g;
            return rc;
        }
        g->members.push_back(ctx);
        g->devices.push_back(device_ids[k]);
    }
    std::set<int> uniq(g->devices.begin(), g->devices.end());
    g->distinct = uniq.size() == g->devices.size();
    /* RCCL when the members sit on different devices; CSM_GROUP_FORCE_RCCL also takes it
     * for a single member (a one-rank communicator: exercises the RCCL call sequence on
     * a one-GPU box) */
    g->use_rccl = g->distinct && (n_devices > 1 || (flags & CSM_GROUP_FORCE_RCCL));
    g->send.resize(n_devices);
    g->recv.resize(n_devices);
    *out = g;
    return CSM_OK;
}

int csm_group_destroy(csm_group* g)
{
    if (!g)
        return CSM_EINVAL;
    for (size_t k = 0; k < g->members.size(); ++k) {
        (void)hipSetDevice(g->devices[k]);
        (void)hipStreamSynchronize(g->members[k]->stream);
        if (k < g->comms.size() && g->comms[k] && g->p_comm_destroy)
            (void)g->p_comm_destroy(g->comms[k]);
        if (g->send[k].p)
            (void)hipFree(g->send[k].p);
        if (g->recv[k].p)
            (void)hipFree(g->recv[k].p);
    }
    for (csm_ctx* c : g->members)
        csm_destroy(c);
    if (g->rccl)
        dlclose(g->rccl);
    delete g;
    return CSM_OK;
}

int32_t csm_group_size(const csm_group* g) { return g ? (int32_t)g->members.size() : 0; }

csm_ctx* csm_group_member(csm_group* g, int32_t member)
{
    return g && member >= 0 && member < (int32_t)g->members.size() ? g->members[member] : nullptr;
}

const char* csm_group_last_error(const csm_group* g) { return g ? g->err.c_str() : "null group"; }

int csm_group_exchange_info(const csm_group* g, int32_t* used_rccl, double* last_gather_us)
{
    if (!g)
        return CSM_EINVAL;
    if (used_rccl)
        *used_rccl = g->use_rccl ? 1 : 0;
    if (last_gather_us)
        *last_gather_us = g->last_gather_us;
    return CSM_OK;
}

int csm_allgather_results(csm_group* g, csm_result* host_out)
{
    if (!g)
        return CSM_EINVAL;
    const int m = (int)g->members.size(), n = g->last_n;
    if (n < 1)
        return gfail(g, CSM_ENOENT, "no batch has been scored on this group");
    const auto t0 = std::chrono::steady_clock::now();
    const int block = (n + m - 1) / m;                       /* padded block, records */
    const size_t block_bytes = (size_t)block * sizeof(csm_result);
    /* An error return must not leave copies or a collective pending on the members' streams (the
     * caller may free or reuse the buffers): every member's stream is drained first. */
    auto drained = [&](int code) {
        for (int k = 0; k < m; ++k) {
            (void)hipSetDevice(g->devices[k]);
            (void)hipStreamSynchronize(g->members[k]->stream);
        }
        return code;
    };
    /* every member: its block (device records of its last batch call) into a padded send buffer */
    for (int k = 0; k < m; ++k) {
        csm_ctx* c = g->members[k];
        if (hipSetDevice(g->devices[k]) != hipSuccess)
            return gfail(g, CSM_EIO, "hipSetDevice(%d) failed", g->devices[k]);
        int rc;
        if ((rc = ensure(c, g->send[k], block_bytes)) || (rc = ensure(c, g->recv[k], block_bytes * m)))
            return drained(gfail(g, rc, "member %d: %s", k, csm_last_error(c)));
        int32_t lo = 0, hi = 0;
        csm_shard_bounds(n, k, m, &lo, &hi);
        if (hipMemsetAsync(g->send[k].p, 0, block_bytes, c->stream) != hipSuccess)
            return drained(gfail(g, CSM_EIO, "hipMemsetAsync failed"));
        if (hi > lo) {
            if (c->rec_n != hi - lo || !c->rec_dev.p)
                return drained(gfail(g, CSM_ENOENT, "member %d holds %d records, its block has %d", k, c->rec_n,
                                     hi - lo));
            if (hipMemcpyAsync(g->send[k].p, c->rec_dev.p, (size_t)(hi - lo) * sizeof(csm_result),
                               hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                return drained(gfail(g, CSM_EIO, "hipMemcpyAsync failed"));
        }
    }
    if (g->use_rccl) {
        int rc = group_init_comms(g);
        if (rc)
            return drained(rc);
        /* a group that was opened is always closed, whatever a call inside it returned */
        int res = g->p_group_start();
        const bool opened = res == 0;
        for (int k = 0; k < m && res == 0; ++k)
            res = g->p_all_gather(g->send[k].p, g->recv[k].p, block_bytes, /* ncclUint8 */ 1, g->comms[k],
                                  g->members[k]->stream);
        const int res_end = opened ? g->p_group_end() : 0;
        if (res != 0 || res_end != 0)
            return drained(gfail(g, CSM_EIO, "ncclAllGather failed: %s",
                                 g->p_error_string ? g->p_error_string(res ? res : res_end) : "?"));
    } else {
        /* one member, or members sharing a device: the blocks go through the host */
        std::vector<char> host(block_bytes * m);
        for (int k = 0; k < m; ++k) {
            (void)hipSetDevice(g->devices[k]);
            if (hipMemcpyAsync(host.data() + block_bytes * k, g->send[k].p, block_bytes,
                               hipMemcpyDeviceToHost, g->members[k]->stream) != hipSuccess ||
                hipStreamSynchronize(g->members[k]->stream) != hipSuccess)
                return drained(gfail(g, CSM_EIO, "device to host copy failed"));
        }
        for (int k = 0; k < m; ++k) {
            (void)hipSetDevice(g->devices[k]);
            if (hipMemcpy(g->recv[k].p, host.data(), block_bytes * m, hipMemcpyHostToDevice) != hipSuccess)
                return drained(gfail(g, CSM_EIO, "host to device copy failed"));
        }
    }
    /* member 0's gathered buffer, padding dropped, in query order */
    std::vector<csm_result> all((size_t)block * m);
    (void)hipSetDevice(g->devices[0]);
    if (hipMemcpyAsync(all.data(), g->recv[0].p, block_bytes * m, hipMemcpyDeviceToHost,
                       g->members[0]->stream) != hipSuccess)
        return drained(gfail(g, CSM_EIO, "device to host copy failed"));
    for (int k = 0; k < m; ++k) {
        (void)hipSetDevice(g->devices[k]);
        if (hipStreamSynchronize(g->members[k]->stream) != hipSuccess)
            return drained(gfail(g, CSM_EIO, "hipStreamSynchronize failed"));
    }
    if (host_out)
        for (int k = 0; k < m; ++k) {
            int32_t lo = 0, hi = 0;
            csm_shard_bounds(n, k, m, &lo, &hi);
            for (int i = lo; i < hi; ++i)
                host_out[i] = all[(size_t)k * block + (i - lo)];
        }
    g->last_gather_us =
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    g->err.clear();
    return CSM_OK;
}

int csm_group_gathered_records_dev(csm_group* g, int32_t member, const csm_result** dev, int32_t* block)
{
    if (!g || member < 0 || member >= (int32_t)g->members.size() || !dev || !block || g->last_n < 1)
        return gfail(g, CSM_EINVAL, "bad arguments");
    *dev = reinterpret_cast<const csm_result*>(g->recv[member].p);
    *block = (g->last_n + (int32_t)g->members.size() - 1) / (int32_t)g->members.size();
    return CSM_OK;
}

int csm_group_bnb_match_batch(csm_group* g, const csm_loop_query* queries, int32_t n_queries,
                              const csm_bnb_params* prm, csm_summary* out)
{
    return group_run_batch(g, queries, n_queries, prm, out, csm_bnb_match_batch);
}

int csm_group_correlative_match_batch(csm_group* g, const csm_loop_query* queries, int32_t n_queries,
                                      const csm_correlative_params* prm, csm_summary* out)
{
    return group_run_batch(g, queries, n_queries, prm, out, csm_correlative_match_batch);
}

} /* extern "C" */
