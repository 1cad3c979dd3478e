// fs_multi.hip — one process, one calling thread, several GPUs (include/fitslam_frontier.h, "one process, several GPUs").
//
// The reference calls its scorer in-process from the behaviour-tree thread (DEP/src/ExplorationBT.cpp:376-410 ->
// CostAssigner::getFrontierCosts, DEP/src/CostAssigner.cpp:73-119); a ROS node cannot turn itself into eight torchrun ranks.
// fs_multi is that call for a multi-GPU node: one fs_ctx per device, staging broadcast to all, the frontier list cut into
// contiguous blocks, every block launched asynchronously on its device's stream before the first one is waited for, the
// records of all blocks landed in the caller's buffer in list order.  Nothing here touches a kernel: it is the host-side
// schedule over the single-device C ABI.  No RCCL: candidates are independent and the records return over PCIe to the one
// host that asked for them (32 B x n); the multi-process bench keeps its all-gather for the ranks that each need the full list.
#include "fs_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

struct fs_multi {
    std::vector<fs_ctx *> ctx;
    std::vector<int> device;
    std::string err;
};

namespace {

int multi_fail(fs_multi *m, int code, const char *fmt, ...)
{
    if (m) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        m->err = buf;
    }
    return code;
}

// apply `call(ctx)` to every member; the first failure is reported with the member's own message
template <typename F>
int for_all(fs_multi *m, const char *what, F call)
{
    if (!m) return FS_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        const int rc = call(m->ctx[i]);
        if (rc != FS_OK) return multi_fail(m, rc, "%s on member %zu (device %d): %s", what, i, m->device[i], fs_last_error(m->ctx[i]));
    }
    return FS_OK;
}

// The same for calls that move megabytes to every device (map and cloud snapshots): one short-lived host thread per member,
// so that eight PCIe links copy at the same time instead of one after the other (the caller still sees ONE blocking call from
// its one thread; the members' contexts are independent, each thread binds its own device).
template <typename F>
int for_all_parallel(fs_multi *m, const char *what, F call)
{
    if (!m) return FS_E_INVALID;
    const size_t n = m->ctx.size();
    std::vector<int> rc(n, FS_OK);
    if (n == 1) {
        rc[0] = call(m->ctx[0]);
    } else {
        std::vector<std::thread> workers;
        workers.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            try {
                workers.emplace_back([&, i] { rc[i] = call(m->ctx[i]); });
            } catch (...) {                                   // no thread to be had: this member's copy runs here, after the others
                rc[i] = call(m->ctx[i]);
            }
        }
        for (std::thread &t : workers) t.join();
    }
    for (size_t i = 0; i < n; ++i)
        if (rc[i] != FS_OK) return multi_fail(m, rc[i], "%s on member %zu (device %d): %s", what, i, m->device[i], fs_last_error(m->ctx[i]));
    return FS_OK;
}

}  // namespace

extern "C" {

int fs_multi_shard_bounds(int32_t n, int n_shards, int shard, int32_t *lo, int32_t *hi)
{
    if (n < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !lo || !hi) return FS_E_INVALID;
    const int64_t per = ((int64_t)n + n_shards - 1) / n_shards;
    const int64_t l = std::min<int64_t>(n, (int64_t)shard * per);
    *lo = (int32_t)l;
    *hi = (int32_t)std::min<int64_t>(n, l + per);
    return FS_OK;
}

int fs_multi_create(const int *device_ids, int n_devices, fs_multi **out)
{
    if (!out) return FS_E_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return FS_E_INVALID;
    fs_multi *m = new fs_multi();
    for (int i = 0; i < n_devices; ++i) {
        fs_ctx *c = nullptr;
        const int rc = fs_ctx_create(device_ids[i], nullptr, &c);       // a stream of its own per member
        if (rc != FS_OK) {
            fs_multi_destroy(m);
            return rc;                                                  // FS_E_NO_DEVICE: no gfx950 under that ordinal — there is no CPU fallback
        }
        m->ctx.push_back(c);
        m->device.push_back(device_ids[i]);
    }
    *out = m;
    return FS_OK;
}

void fs_multi_destroy(fs_multi *m)
{
    if (!m) return;
    for (fs_ctx *c : m->ctx) fs_ctx_destroy(c);
    delete m;
}

int fs_multi_num_devices(const fs_multi *m) { return m ? (int)m->ctx.size() : 0; }

fs_ctx *fs_multi_ctx(fs_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }

const char *fs_multi_last_error(const fs_multi *m) { return m ? m->err.c_str() : "null multi-device scorer"; }

int fs_multi_set_option(fs_multi *m, const char *key, double value)
{
    return for_all(m, "fs_set_option", [&](fs_ctx *c) { return fs_set_option(c, key, value); });
}

int fs_multi_set_ray_params(fs_multi *m, const fs_ray_params *p)
{
    return for_all(m, "fs_set_ray_params", [&](fs_ctx *c) { return fs_set_ray_params(c, p); });
}

int fs_multi_upload_grid(fs_multi *m, const uint8_t *cells, int32_t nx, int32_t ny, int32_t nz, const double origin_xyz[3], double resolution)
{
    return for_all_parallel(m, "fs_upload_grid", [&](fs_ctx *c) { return fs_upload_grid(c, cells, nx, ny, nz, origin_xyz, resolution); });
}

int fs_multi_upload_landmarks(fs_multi *m, const float *xyz, int32_t n_landmarks)
{
    if (!m || (n_landmarks > 0 && !xyz) || n_landmarks < 0) return FS_E_INVALID;
    if (n_landmarks > 2000000) return multi_fail(m, FS_E_INVALID, "at most 2,000,000 landmarks per context");
    // the k-d ordering and the chunk spheres are a function of the cloud alone: once, on the calling thread (11.7 ms at C3);
    // then every device takes its copy at the same time
    FsStagedCloud staged;
    fs_stage_landmarks(xyz, n_landmarks, staged);
    return for_all_parallel(m, "fs_upload_landmarks", [&](fs_ctx *c) { return fs_upload_staged_landmarks(c, staged); });
}

int fs_multi_lookup_generate(fs_multi *m, const float bounds[6])
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    // the generator runs once (0.4 s of host float loops for the reference bounds); the others take its records
    int rc = fs_lookup_generate(m->ctx[0], bounds);
    if (rc != FS_OK) return multi_fail(m, rc, "fs_lookup_generate: %s", fs_last_error(m->ctx[0]));
    int64_t n = 0;
    fs_lookup_num_records(m->ctx[0], &n);
    std::vector<float> rec((size_t)n * 4);
    fs_lookup_get_records(m->ctx[0], rec.data());
    for (size_t i = 1; i < m->ctx.size(); ++i) {
        rc = fs_lookup_set_records(m->ctx[i], rec.data(), n);
        if (rc != FS_OK) return multi_fail(m, rc, "fs_lookup_set_records on member %zu: %s", i, fs_last_error(m->ctx[i]));
    }
    return FS_OK;
}

int fs_multi_lookup_load(fs_multi *m, const char *path)
{
    return for_all(m, "fs_lookup_load", [&](fs_ctx *c) { return fs_lookup_load(c, path); });
}

int fs_multi_set_fim_params(fs_multi *m, const fs_fim_params *p)
{
    return for_all(m, "fs_set_fim_params", [&](fs_ctx *c) { return fs_set_fim_params(c, p); });
}

int fs_multi_max_arrival(fs_multi *m, double *max_value, double *max_gt, double *min_gt)
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    double mv = 0, mx = 0, mn = 0;
    const int rc = fs_max_arrival(m->ctx[0], &mv, &mx, &mn);
    if (rc != FS_OK) return multi_fail(m, rc, "fs_max_arrival: %s", fs_last_error(m->ctx[0]));
    // candidate-independent (DEP/src/CostCalculator.cpp:123-191): computed once, cached everywhere
    for (size_t i = 1; i < m->ctx.size(); ++i) fs_set_arrival_limits(m->ctx[i], mx, mn);
    if (max_value) *max_value = mv;
    if (max_gt) *max_gt = mx;
    if (min_gt) *min_gt = mn;
    return FS_OK;
}

int fs_multi_score_arrival(fs_multi *m, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                           const uint8_t *blacklisted, const uint8_t *achievable_in,
                           int32_t *ray_counts, int32_t *arrival, int32_t *argmax, double *yaw,
                           uint8_t *achievable, int32_t *status)
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    if (n < 0 || (n > 0 && (!goal_xyz || !arrival || !argmax || !yaw || !achievable || !status))) return multi_fail(m, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    const int G = (int)m->ctx.size();
    int32_t ny = 0, ne = 0;
    if (ray_counts && fs_ray_fan_shape(m->ctx[0], &ny, &ne, nullptr) != FS_OK) return multi_fail(m, FS_E_STATE, "fs_set_ray_params has not been called");
    const size_t per = (size_t)ny * (size_t)ne;
    int first_error = FS_OK, launched = 0;
    for (int g = 0; g < G; ++g) {
        int32_t l = 0, h = 0;
        fs_multi_shard_bounds(n, G, g, &l, &h);
        const int rc = fs_score_arrival_begin(m->ctx[(size_t)g], h - l, goal_xyz + 3 * (size_t)l, frontier_size ? frontier_size + l : nullptr,
                                              blacklisted ? blacklisted + l : nullptr, achievable_in ? achievable_in + l : nullptr,
                                              ray_counts ? ray_counts + per * (size_t)l : nullptr, arrival + l, argmax + l, yaw + l, achievable + l, status + l);
        if (rc != FS_OK) {
            first_error = multi_fail(m, rc, "block %d (candidates %d..%d, device %d): %s", g, l, h, m->device[(size_t)g], fs_last_error(m->ctx[(size_t)g]));
            (void)fs_synchronize(m->ctx[(size_t)g]);     // a begin that failed half-way may have queued transfers out of this member's staging buffers: drained before anybody reuses them
            break;
        }
        ++launched;
    }
    for (int g = 0; g < launched; ++g) {
        const int rc = fs_score_arrival_end(m->ctx[(size_t)g]);
        if (rc != FS_OK && first_error == FS_OK)
            first_error = multi_fail(m, rc, "block %d (device %d): %s", g, m->device[(size_t)g], fs_last_error(m->ctx[(size_t)g]));
    }
    return first_error;
}

int fs_multi_score_candidates(fs_multi *m, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                              const uint8_t *blacklisted, const uint8_t *achievable_in, fs_record *records)
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    if (n < 0 || (n > 0 && (!goal_xyz || !records))) return multi_fail(m, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    const int G = (int)m->ctx.size();
    std::vector<int32_t> lo((size_t)G), hi((size_t)G);
    // ---- every block is launched before any is waited for: the devices work side by side under one host thread
    int first_error = FS_OK;
    int launched = 0;
    for (int g = 0; g < G; ++g) {
        fs_multi_shard_bounds(n, G, g, &lo[(size_t)g], &hi[(size_t)g]);
        const int32_t l = lo[(size_t)g], cnt = hi[(size_t)g] - l;
        const int rc = fs_score_candidates_begin(m->ctx[(size_t)g], cnt, goal_xyz + 3 * (size_t)l, frontier_size ? frontier_size + l : nullptr,
                                                 blacklisted ? blacklisted + l : nullptr, achievable_in ? achievable_in + l : nullptr);
        if (rc != FS_OK) {
            first_error = multi_fail(m, rc, "block %d (candidates %d..%d, device %d): %s", g, l, hi[(size_t)g], m->device[(size_t)g], fs_last_error(m->ctx[(size_t)g]));
            (void)fs_synchronize(m->ctx[(size_t)g]);     // (as in fs_multi_score_arrival)
            break;
        }
        ++launched;
    }
    // ---- collect in list order (what was launched is always waited for, also after a failure)
    for (int g = 0; g < launched; ++g) {
        const int32_t l = lo[(size_t)g], cnt = hi[(size_t)g] - l;
        const int rc = fs_score_candidates_end(m->ctx[(size_t)g], cnt, records + l);
        if (rc != FS_OK && first_error == FS_OK)
            first_error = multi_fail(m, rc, "block %d (device %d): %s", g, m->device[(size_t)g], fs_last_error(m->ctx[(size_t)g]));
    }
    return first_error;
}

}  // extern "C"
