// fs_multi.hip — one process, one calling thread, several GPUs (include/fitslam_frontier.h, "one process, several GPUs").
//
// The reference calls its scorer in-process from the behaviour-tree thread (DEP/src/ExplorationBT.cpp:376-410 ->
// CostAssigner::getFrontierCosts, DEP/src/CostAssigner.cpp:73-119); a ROS node cannot turn itself into eight torchrun ranks.
// fs_multi is that call for a multi-GPU node: one fs_ctx per device, staging broadcast to all, the frontier list cut into
// contiguous blocks, every block launched asynchronously on its device's stream before the first one is waited for, the
// records of all blocks landed in the caller's buffer in list order.  Nothing here touches a kernel: it is the host-side
// schedule over the single-device C ABI.
//
// Two ways home for the records.  fs_multi_score_candidates returns every block to the HOST (PCIe, 32 B x n) — for a caller that
// wants the records and ranks later.  fs_multi_get_frontier_costs is the reference's one call (CostAssigner::getFrontierCosts)
// on several GPUs and keeps the records on the devices: each member's block is moved DEVICE TO DEVICE into one list on member
// 0's GPU — hipMemcpyPeerAsync over xGMI on the member's own stream, right behind its kernels, plus an event member 0's stream
// waits for —, fs_rank_candidates_dev ranks the gathered list there, and one transfer brings records, costs and order to the host.
// That is a GATHER onto the one device that ranks, not an all-gather: in one process only one device needs the list, and on
// the fully connected xGMI mesh every peer has a link of its own to member 0 (7 x n/8 x 32 B at once: 560 KB for C4).  The
// multi-process bench (bench.py --gpus N, shard.py) keeps RCCL's all-gather, where every RANK needs the whole list.  Without
// peer access (hipDeviceCanAccessPeer says no) the blocks bounce through the members' page-locked buffers instead, and
// fs_multi_last_error says so.
#include "fs_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

enum { FS_GATHER_AUTO = 0, FS_GATHER_PEER = 1, FS_GATHER_HOST = 2, FS_GATHER_COPY = 3, FS_GATHER_PEER_CALL = 4 };

struct fs_multi {
    std::vector<fs_ctx *> ctx;
    std::vector<int> device;
    std::string err;
    // fs_multi_get_frontier_costs: one event per member (its block has reached the gathered list), and how blocks travel
    std::vector<hipEvent_t> block_done;
    int gather_mode = FS_GATHER_AUTO;       // what set-up found: FS_GATHER_PEER or FS_GATHER_HOST
    int gather_forced = FS_GATHER_AUTO;     // fs_multi_set_option("multi.gather", 1 | 2 | 3): tests of the paths one GPU cannot take by itself
    std::string gather_note;
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

// events and peer access for the device-to-device gather; idempotent
int gather_setup(fs_multi *m)
{
    if (m->gather_mode != FS_GATHER_AUTO) return FS_OK;
    const int dev0 = m->device[0];
    bool peer_ok = true;
    m->gather_note.clear();
    m->block_done.assign(m->ctx.size(), nullptr);
    for (size_t g = 0; g < m->ctx.size(); ++g) {
        const int dev = m->device[g];
        if (hipSetDevice(dev) != hipSuccess || hipEventCreateWithFlags(&m->block_done[g], hipEventDisableTiming) != hipSuccess)
            return multi_fail(m, FS_E_HIP, "event for member %zu (device %d): %s", g, dev, hipGetErrorString(hipGetLastError()));
        if (dev == dev0) continue;
        // both directions: the member's copy engine writes into device 0's list, device 0's stream waits for the member's event
        int can_a = 0, can_b = 0;
        (void)hipDeviceCanAccessPeer(&can_a, dev, dev0);
        (void)hipDeviceCanAccessPeer(&can_b, dev0, dev);
        if (!can_a || !can_b) {
            peer_ok = false;
            char buf[160];
            snprintf(buf, sizeof buf, "no peer access between device %d and device %d: record blocks bounce through page-locked host memory", dev, dev0);
            m->gather_note = buf;
            continue;
        }
        hipError_t e = hipDeviceEnablePeerAccess(dev0, 0);                 // (current device: dev)
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_ok = false;
        (void)hipGetLastError();
        if (hipSetDevice(dev0) == hipSuccess) {
            e = hipDeviceEnablePeerAccess(dev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peer_ok = false;
            (void)hipGetLastError();
        }
        if (!peer_ok && m->gather_note.empty()) m->gather_note = "hipDeviceEnablePeerAccess refused: record blocks bounce through page-locked host memory";
    }
    m->gather_mode = peer_ok ? FS_GATHER_PEER : FS_GATHER_HOST;
    if (!peer_ok) m->err = m->gather_note;                                // not an error code: the call works, fs_multi_last_error says how
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
    for (size_t g = 0; g < m->block_done.size(); ++g)
        if (m->block_done[g]) { (void)hipSetDevice(m->device[g]); (void)hipStreamSynchronize(fs_ctx_stream(m->ctx[g])); (void)hipEventDestroy(m->block_done[g]); }
    for (fs_ctx *c : m->ctx) fs_ctx_destroy(c);
    delete m;
}

int fs_multi_gather_mode(fs_multi *m)
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    const int rc = gather_setup(m);
    if (rc != FS_OK) return rc;
    return m->gather_forced != FS_GATHER_AUTO ? m->gather_forced : m->gather_mode;
}

int fs_multi_num_devices(const fs_multi *m) { return m ? (int)m->ctx.size() : 0; }

fs_ctx *fs_multi_ctx(fs_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }

const char *fs_multi_last_error(const fs_multi *m) { return m ? m->err.c_str() : "null multi-device scorer"; }

int fs_multi_set_option(fs_multi *m, const char *key, double value)
{
    if (m && key && std::string(key) == "multi.gather") {      // 0 auto | 1 peer copies | 2 host bounce | 3 device copies also between members of one GPU | 4 the same through hipMemcpyPeerAsync
        const int v = (int)value;
        if (v < FS_GATHER_AUTO || v > FS_GATHER_PEER_CALL) return multi_fail(m, FS_E_INVALID, "multi.gather takes 0 (auto), 1 (peer), 2 (host bounce), 3 (copy) or 4 (peer-copy call)");
        m->gather_forced = v;
        if (v == FS_GATHER_HOST) m->err = "multi.gather = 2: record blocks bounce through page-locked host memory";
        return FS_OK;
    }
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

int fs_multi_update_grid_region(fs_multi *m, int32_t x0, int32_t y0, int32_t z0, int32_t sx, int32_t sy, int32_t sz,
                                const uint8_t *cells, int64_t row_stride, int64_t slice_stride)
{
    return for_all_parallel(m, "fs_update_grid_region", [&](fs_ctx *c) { return fs_update_grid_region(c, x0, y0, z0, sx, sy, sz, cells, row_stride, slice_stride); });
}

int fs_multi_upload_landmarks(fs_multi *m, const float *xyz, int32_t n_landmarks)
{
    if (!m || (n_landmarks > 0 && !xyz) || n_landmarks < 0) return FS_E_INVALID;
    if (n_landmarks > 2000000) return multi_fail(m, FS_E_INVALID, "at most 2,000,000 landmarks per context");
    // the k-d ordering and the chunk spheres are a function of the cloud alone: once (4.3 ms at C3, the ordering's top levels on threads of their own);
    // then every device takes its copy at the same time
    // (the ordering on the device, "cloud.order": every member orders its own copy, all at once — the same order on each)
    if (!m->ctx.empty() && fs_ctx_cloud_on_device(m->ctx[0], n_landmarks))
        return for_all_parallel(m, "fs_upload_landmarks", [&](fs_ctx *c) { return fs_upload_landmarks(c, xyz, n_landmarks); });
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

int fs_multi_score_fim(fs_multi *m, int32_t n, const double *pose7, float *info_ref, float *fim21,
                       float *trace, float *logdet, int32_t *n_visible, int32_t *n_voxels)
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    if (n < 0 || (n > 0 && (!pose7 || !info_ref))) return multi_fail(m, FS_E_INVALID, "null pose or output pointer");
    if (n == 0) return FS_OK;
    const int G = (int)m->ctx.size();
    int first_error = FS_OK, launched = 0;
    for (int g = 0; g < G; ++g) {
        int32_t l = 0, h = 0;
        fs_multi_shard_bounds(n, G, g, &l, &h);
        const size_t o = (size_t)l;
        const int rc = fs_score_fim_begin(m->ctx[(size_t)g], h - l, pose7 + 7 * o, info_ref + o, fim21 ? fim21 + 21 * o : nullptr, trace ? trace + o : nullptr,
                                          logdet ? logdet + o : nullptr, n_visible ? n_visible + o : nullptr, n_voxels ? n_voxels + o : nullptr);
        if (rc != FS_OK) {
            first_error = multi_fail(m, rc, "block %d (poses %d..%d, device %d): %s", g, l, h, m->device[(size_t)g], fs_last_error(m->ctx[(size_t)g]));
            (void)fs_synchronize(m->ctx[(size_t)g]);
            break;
        }
        ++launched;
    }
    for (int g = 0; g < launched; ++g) {
        const int rc = fs_score_fim_end(m->ctx[(size_t)g]);
        if (rc != FS_OK && first_error == FS_OK)
            first_error = multi_fail(m, rc, "block %d (device %d): %s", g, m->device[(size_t)g], fs_last_error(m->ctx[(size_t)g]));
    }
    return first_error;
}

int fs_multi_get_frontier_costs(fs_multi *m, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                                const uint8_t *achievable_in, const double *path_length, const double *path_heading,
                                double alpha, double beta, double max_vx, double max_wz, int with_fisher_information,
                                fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order)
{
    if (!m || m->ctx.empty()) return FS_E_INVALID;
    if (n < 0 || (n > 0 && (!goal_xyz || !path_length || !path_heading || !records || !weighted_cost))) return multi_fail(m, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    int rc = gather_setup(m);
    if (rc != FS_OK) return rc;
    const int G = (int)m->ctx.size();
    const int mode = m->gather_forced != FS_GATHER_AUTO ? m->gather_forced : m->gather_mode;
    const bool with_fim = with_fisher_information != 0;
    fs_ctx *c0 = m->ctx[0];
    const int dev0 = m->device[0];
    hipStream_t s0 = fs_ctx_stream(c0);
    fs_record *d_list = nullptr;
    rc = fs_gather_begin(c0, n, blacklisted, path_length, path_heading, &d_list);
    if (rc != FS_OK) return multi_fail(m, rc, "gather set-up on device %d: %s", dev0, fs_last_error(c0));
    // ---- every member scores its block; blocks of other members travel to the list behind their kernels.  Nothing waits here.
    int first_error = FS_OK;
    std::vector<int> started;
    for (int g = 0; g < G && first_error == FS_OK; ++g) {
        int32_t l = 0, h = 0;
        fs_multi_shard_bounds(n, G, g, &l, &h);
        const int32_t cnt = h - l;
        if (cnt <= 0) continue;
        fs_ctx *cg = m->ctx[(size_t)g];
        const int dev = m->device[(size_t)g];
        const size_t bytes = sizeof(fs_record) * (size_t)cnt;
        // who may write the list itself: member 0, and (unless a test forces the copies) a member that shares member 0's GPU
        const bool direct = g == 0 || (dev == dev0 && mode == FS_GATHER_PEER);
        fs_record *d_block = nullptr;
        rc = fs_block_score_begin(cg, cnt, goal_xyz + 3 * (size_t)l, frontier_size ? frontier_size + l : nullptr, blacklisted ? blacklisted + l : nullptr,
                                  achievable_in ? achievable_in + l : nullptr, with_fim, direct ? d_list + l : nullptr, &d_block);
        started.push_back(g);
        if (rc != FS_OK) {
            first_error = multi_fail(m, rc, "block %d (candidates %d..%d, device %d): %s", g, l, h, dev, fs_last_error(cg));
            break;
        }
        if (g == 0) continue;                                   // member 0's kernels and the ranking share a stream: ordered by it
        hipStream_t sg = fs_ctx_stream(cg);
        const fs_record *h_block = nullptr;
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess && !direct) {
            if (mode == FS_GATHER_HOST) {
                rc = fs_block_records_to_host(cg, cnt, d_block, &h_block);
                if (rc != FS_OK) { first_error = multi_fail(m, rc, "block %d to host (device %d): %s", g, dev, fs_last_error(cg)); break; }
            } else if (dev != dev0 || mode == FS_GATHER_PEER_CALL) {
                // (mode 4: the peer-copy CALL also between two members of one GPU — the only way a one-GPU box executes this line)
                e = hipMemcpyPeerAsync(d_list + l, dev0, d_block, dev, bytes, sg);        // over xGMI, behind the block's kernels
            } else {
                e = hipMemcpyAsync(d_list + l, d_block, bytes, hipMemcpyDeviceToDevice, sg);
            }
        }
        if (e == hipSuccess) e = hipEventRecord(m->block_done[(size_t)g], sg);
        if (e == hipSuccess) e = hipSetDevice(dev0);
        if (e == hipSuccess) e = hipStreamWaitEvent(s0, m->block_done[(size_t)g], 0);
        if (e == hipSuccess && h_block) e = hipMemcpyAsync(d_list + l, h_block, bytes, hipMemcpyHostToDevice, s0);
        if (e != hipSuccess) {
            first_error = multi_fail(m, FS_E_HIP, "block %d (device %d) -> gathered list on device %d: %s", g, dev, dev0, hipGetErrorString(e));
            break;
        }
    }
    // ---- rank the gathered list on member 0's device, one transfer out
    if (first_error == FS_OK) {
        rc = fs_gather_rank(c0, n, alpha, beta, max_vx, max_wz);
        if (rc != FS_OK) first_error = multi_fail(m, rc, "ranking on device %d: %s", dev0, fs_last_error(c0));
    }
    if (first_error == FS_OK) {
        rc = fs_gather_end(c0, n, records, weighted_cost, arrival_utility, distance_utility, order);
        if (rc != FS_OK) first_error = multi_fail(m, rc, "%s", fs_last_error(c0));
    }
    if (first_error != FS_OK || mode == FS_GATHER_HOST) {
        // after a failure whatever was started is drained before anybody reuses the staging buffers; (host bounce: the member's
        // stream is idle once member 0's has passed its event — waited for here all the same, it costs nothing)
        for (int g : started) (void)fs_synchronize(m->ctx[(size_t)g]);
    }
    // a call that worked through the bounce says so every time (an earlier failure may have overwritten the note)
    if (first_error == FS_OK && mode == FS_GATHER_HOST && m->gather_forced == FS_GATHER_AUTO && !m->gather_note.empty()) m->err = m->gather_note;
    return first_error;
}

}  // extern "C"
