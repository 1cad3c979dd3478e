// fs_capi.hip — the C ABI of include/fitslam_frontier.h: context, staging of grid / landmarks / lookup
// table into HBM, host-side precomputation (fan directions, yaw rotations, crowding factors, dense
// table) and the launch sequences.  No CPU compute fallback exists: every scoring entry point
// launches the HIP kernels of fs_raymarch.hip / fs_fim.hip / fs_rank.hip or fails.
#include "fs_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

hipError_t fs_launch_rank(int32_t n, const fs_record *d_records, const uint8_t *d_black,
                          const double *d_len, const double *d_head, double alpha, double beta,
                          double max_vx, double max_wz, double max_gt, double *d_cost, double *d_au,
                          double *d_du, int32_t *d_order, int32_t *d_err, void **scratch, size_t *scratch_bytes,
                          hipStream_t s);

// Counts every (re)allocation of device / page-locked memory the library makes, in any context: a captured launch graph holds
// raw pointers, and is only replayed while this number is what it was at capture time.
std::atomic<uint64_t> fs_alloc_generation{0};     // (atomic: fs_multi stages its members from one host thread each)

namespace {

// Development build FS_POISON (`FS_POISON=1 python fit-slam_amd/_build.py`, a library of its own like every FS_DEV build): a
// buffer that grows is RETIRED, not freed — nothing allocated later can land on its address — and both the retired and the new
// memory are filled with 0xCD.  A launch that still holds a pointer taken before the growth then writes where nobody reads, and
// whoever reads the new buffer finds the pattern instead of a result: the stale-pointer defect of round 5 (fs_score_candidates_dev,
// DESIGN.md 0) fails every time under this build instead of only when the allocator happens to hand out a different address;
// so does anything that relies on fresh device memory being zero.  Leaks by design; test processes only.
#ifdef FS_POISON
static void poison_device(void *p, size_t bytes)
{
    (void)hipDeviceSynchronize();
    (void)hipMemset(p, 0xCD, bytes);
    (void)hipDeviceSynchronize();
}
#endif

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
#ifdef FS_POISON
        if (p) poison_device(p, cap * sizeof(T));                 // retired
#else
        if (p) (void)hipFree(p);
#endif
        p = nullptr; cap = 0;
        size_t want = std::max<size_t>(n, 64);
        ++fs_alloc_generation;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
#ifdef FS_POISON
        if (e == hipSuccess) poison_device(p, want * sizeof(T));
#endif
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};

// page-locked staging memory of a context: copies from / to it are true DMA transfers that overlap with the host, copies
// from pageable user memory are staged by the runtime in small synchronous pieces (measured on C3's 20 k candidates: 1.29 ms per
// fs_score_candidates call with pageable copies of the three input columns and the records)
// The buffer is also MAPPED into the device's address space (`dev`, nullptr if the runtime refuses): a small call lets its
// kernels read the candidate columns from it and write the results into it directly — a transfer of a few hundred bytes costs
// 4-5 us as an operation of its own on the stream (measured: five result columns against two, 46 against 33 us per one-pose
// call), a handful of PCIe reads and posted writes inside kernels that run anyway cost nothing that shows.
struct PinnedBuf {
    char *p = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
#ifdef FS_POISON
        if (p) { (void)hipDeviceSynchronize(); std::memset(p, 0xCD, cap); }      // retired (see DevBuf)
#else
        if (p) (void)hipHostFree(p);
#endif
        p = nullptr; dev = nullptr; cap = 0;
        const size_t want = std::max<size_t>(bytes, 4096);
        ++fs_alloc_generation;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocMapped);
#ifdef FS_POISON
        if (e == hipSuccess) std::memset(p, 0xCD, want);
#endif
        if (e != hipSuccess) {
            // a runtime / device that refuses MAPPED page-locked memory still gets plain page-locked staging: dev stays nullptr and
            // every entry point takes its transfer path (the sticky error of the refused call is cleared first)
            (void)hipGetLastError();
            p = nullptr;
            e = hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault);
            if (e != hipSuccess) { p = nullptr; return e; }
            cap = want;
            return hipSuccess;
        }
        cap = want;
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) dev = static_cast<char *>(d);
        else (void)hipGetLastError();
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr; dev = nullptr; cap = 0;
    }
};
#define FS_ZERO_COPY_MAX_N 1024   // candidates / poses up to which a host-buffer call reads and writes the mapped staging buffers in place

struct TimedLaunch {
    int kind;
    hipEvent_t start, stop;
};

// One captured launch sequence of a small host-buffer call: valid while the context's epoch (every staging /
// parameter call bumps it) and the allocation generation are what they were when it was captured.
struct GraphEntry {
    hipGraphExec_t exec = nullptr;
    uint64_t stamp = 0;          // epoch + allocation generation the graph was captured under
    uint64_t warm_stamp = 0;     // ... a plain call has run under (it made every allocation the sequence needs)
    bool broken = false;         // capture failed once: this sequence stays on plain launches
};

}  // namespace

struct fs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // ray parameters
    bool have_ray = false;
    fs_ray_params rp{};
    int32_t n_yaw = 0, n_elev = 0, window = 0;
    DevBuf<double> d_dir;
    DevBuf<float> d_yawR;
    double max_gt = 0.0, min_gt = 0.0;

    // grid
    bool have_grid = false;
    DevBuf<uint8_t> d_cells;                        // dense row-major image (FsGridDev)
    DevBuf<uint32_t> d_cls;                         // 2-bit class image, cut lazily for cls_ranges (long rays only)
    bool have_cls = false;
    DevBuf<uint32_t> d_cls_table, d_cls_pool;       // "ray.layout" 3: brick table + pool of distinct bricks (the sparse form of d_cls)
    bool have_sparse = false;
    uint64_t sparse_bricks = 0, sparse_pool_bricks = 0;
    int32_t cls_ranges[4] = {0, 0, 0, 0};
    int32_t nx = 0, ny = 0, nz = 0;
    double origin[3] = {0, 0, 0};
    double res = 0.0;

    // landmarks
    bool have_lm = false;
    int32_t m = 0, n_chunks = 0;
    DevBuf<float> d_lx, d_ly, d_lz, d_spheres;
    bool opt_cull = true;
    bool opt_learn = true;         // "fim.learn": the pass prediction uses the voxel ratio finished calls have shown (0: the fixed cap only — results then do not depend on the calls a context has served before)
    bool opt_special = true;       // "fim.specialise": the INFO_ONLY / YAW_ONLY workers where they apply (0: always the general worker)
    bool yaw_exact = false;        // every rotation of d_yawR is about Z with exact zeros / one (what YAW_ONLY relies on)
    int opt_bits1 = 14;            // development knobs (fs_set_option "fim.bits1", "fim.skip32")
    int opt_skip32 = 13;
    // "fim.headroom": what the pass prediction multiplies the learnt voxel ratio by, in 32nds.  The ratio is the LARGEST any big
    // pose showed and a pass is sized for a table 3/4 full, so 28/32 still leaves the worst pose seen at 86 % load; round 3's
    // 5/4 sent 7 312 of C3's 20 000 poses into a second pass at the reference's visibility request where 2 943 hold more voxels
    // than one pass takes (tools/ref_visibility_probe.py, profiles/r04/ref_visibility_pass_margin.jsonl: 1.87 -> 1.69 ms)
    int opt_headroom = 28;
    DevBuf<unsigned long long> d_counters;

    // lookup table
    bool have_table = false;
    std::vector<float> records;           // as in the .dat, [n][4]
    std::vector<float> dense;             // host copy of the dense table
    int32_t jx0 = 0, jy0 = 0, jz0 = 0, tx = 0, ty = 0, tz = 0;
    DevBuf<float> d_table, d_factor;
    bool have_factor = false;
    bool table_full = false;
    float fac[5] = {0, 0, 0, 0, 0};

    // fim parameters
    fs_fim_params fp{14.0, 1.0};

    // key-frames (computeInformationForPose)
    bool have_kf = false;
    int32_t n_kf = 0;
    int64_t n_kf_points = 0;
    std::vector<double> kf_pose;          // host copy [n_kf][7]: the check points depend on the call's parameters
    DevBuf<double> d_kf_check, d_kf_tri;
    DevBuf<int32_t> d_kf_off, d_kf_flagged, d_kf_cells, d_kf_points;
    DevBuf<float> d_kpx, d_kpy, d_kpz;
    DevBuf<uint32_t> d_kf_gtable;
    DevBuf<unsigned long long> d_kf_counters;

    // hash tables
    DevBuf<uint32_t> d_gtable;
    int ghash_bits = 0;
    static constexpr int kPool = 64;      // tier 3 (rare: no candidate of C3 or C5 reaches it): 64 x 1024 threads — an empty launch of 256 cost 7.5 us per call

    // per-candidate scratch
    DevBuf<double> d_goal, d_yaw, d_len, d_head, d_cost, d_au, d_du, d_sums;
    DevBuf<int32_t> d_fsize, d_arrival, d_argmax, d_status, d_nvis, d_nvox, d_raycounts, d_order, d_err;
    DevBuf<uint8_t> d_black, d_achin, d_ach;
    DevBuf<float> d_info, d_trace, d_logdet, d_fim21, d_Rt;
    DevBuf<uint32_t> d_overflow, d_tested, d_split_flags;
    int opt_split = 3;             // "fim.split": log2 of the workgroups ONE info-only pose is spread over when a call has few poses (0: off)
    DevBuf<int32_t> d_flagged;
    size_t tested_zeroed = 0;
    DevBuf<fs_record> d_records;
    void *rank_scratch = nullptr;
    size_t rank_scratch_bytes = 0;
    void *sort_scratch = nullptr;
    size_t sort_scratch_bytes = 0;
    DevBuf<int32_t> d_perm;
    PinnedBuf h_in, h_out;         // staging of the per-call candidate columns / of the records
    DevBuf<char> d_in;             // the candidate columns of the host-buffer entry points, packed as in h_in
    const double *in_goal = nullptr; const int32_t *in_fsize = nullptr; const uint8_t *in_black = nullptr, *in_achin = nullptr;
    // fs_score_arrival_begin -> _end: where the columns waiting in h_out go (caller's arrays) once the stream has drained
    struct PendingCol { void *host; size_t off, bytes; };
    std::vector<PendingCol> arrival_pending;
    std::vector<PendingCol> fim_pending;           // fs_score_fim_begin -> _end, likewise
    // ... and the finish of a split info-only call that fs_score_fim_end runs on the HOST (host_finish, fs_score_fim_begin): the
    // partial sums of every (pose, w) item land in mapped page-locked memory behind a flag word; the kernel arguments are kept for
    // the rare call that needs the HBM tier and the finish kernel after all
    PinnedBuf h_fin;
    bool fin_active = false;
    FsFimArgs fin_args{};
    size_t fin_off_info = 0, fin_off_nvox = 0;
    bool fin_want_nvox = false;
    bool opt_host_finish = true;   // "fim.hostfinish"
    // the gather role of fs_multi_get_frontier_costs' member 0 (its own block still goes through h_in / d_in): the planner's
    // path columns and the blacklist of the WHOLE list; the gathered records live in d_out, ahead of the ranking's columns
    PinnedBuf h_gin;
    DevBuf<char> d_gin;
    // scratch of the per-tick entry points (fs_trace_segments, fs_frontier_cells, fs_information_frontier_pair,
    // fs_upload_grid_bricks): owned by the context and grown on demand, never allocated and freed per call
    DevBuf<double> d_seg_start, d_seg_end, d_tri;
    DevBuf<uint8_t> d_seg_ok, d_seg_hit, d_mask, d_brick_cells, d_win;
    PinnedBuf h_win;               // fs_update_grid_region: the packed window
    DevBuf<int32_t> d_seg_traced, d_seg_unknown, d_seg_all, d_brick_xyz, d_bad;
    DevBuf<unsigned long long> d_count;
    // fs_frontier_clusters
    DevBuf<int32_t> d_fc_parent_t, d_fc_parent_f, d_fc_aux, d_fc_state, d_fc_labels;
    DevBuf<uint32_t> d_fc_queue;
    DevBuf<uint8_t> d_fc_visited;
    DevBuf<fs_frontier_cluster> d_fc_clusters;
    DevBuf<long long> d_fc_sums;
    // "cloud.order": where the landmark cloud is put into its k-d leaf order — 0 on the host, 2 on the device (fs_cloud.hip), 1 (default)
    // on the device from FS_CLOUD_DEVICE_FROM landmarks on: fs_upload_landmarks 1.2 ms at C3's 100 k landmarks, 4.4 ms at 500 k on the
    // device (the host form: 4.4 / 20.2 ms with its top levels on threads of their own, 13.3 / 78.9 ms on one thread —
    // tools/landmark_staging_probe.py); a level costs the device ~ 0.03 ms of launches whatever its size, so a 2 k cloud is faster
    // on the host (0.06 against 0.18 ms).  Scratch of the device path below.
    int opt_cloud_order = 1;
    DevBuf<float> d_cloud_raw;
    DevBuf<int32_t> d_cloud_perm, d_cloud_bounds;
    DevBuf<uint64_t> d_cloud_keys;
    DevBuf<uint32_t> d_cloud_bbox;
    DevBuf<char> d_cloud_temp;
    bool opt_sort = true;
    bool opt_sort_reverse = false; // development: blocks in reverse Morton order (order-sensitivity measurements)
    bool opt_costmap = true;       // the spatial sort puts the blocks that were expensive in the previous call first ("sort.costmap")
    const uint32_t *sort_keys = nullptr;   // this call's sort keys / the cost map inside sort_scratch (nullptr: list not sorted)
    uint32_t *sort_costmap = nullptr;
    int opt_layout = 0;            // "ray.layout": 0 by ray length, 1 row-major byte walk, 2 class-image walk, 3 sparse class image (experiment)

    // launch graphs of the small host-buffer calls ("graph" option; off while kernel timing is on)
    bool opt_zero_copy = true;     // "zerocopy": small host-buffer calls read / write the mapped page-locked buffers in place
    bool opt_graph = false;        // measured 5-7 us SLOWER per call than plain launches on ROCm 7.2 (profiles/r04/small_call_graphs.json): off by default
    uint64_t epoch = 1;
    std::map<uint64_t, GraphEntry> graphs;
    DevBuf<char> d_out;            // packed results of fs_get_frontier_costs (records | cost | utilities | order | error flag)

    // timing
    bool timing = false;
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int fail(fs_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define FS_HIP(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) return fail((c), FS_E_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// ---------------------------------------------------------------- launch graphs for the small host-buffer calls
// The reference scores tens of frontiers per behaviour-tree tick and ONE pose per isPoseSafe: such a call is a dozen launches of
// nearly empty kernels, and what it costs is launch latency.  `enqueue` puts the call's whole device-side sequence — one
// transfer in, the kernels, one transfer out; fixed sizes, fixed pointers, no allocation, no synchronisation — on the context's
// stream.  The first call in a given state runs it plainly (and makes every allocation it needs), the second captures it into a
// hipGraph, every later one replays the graph with ONE launch.  Any staging or parameter call (epoch) and any reallocation
// (fs_alloc_generation) sends the sequence through that cycle again; a capture that fails leaves it on plain launches.
// MEASURED (profiles/r04/small_call_graphs.json, alternating runs in one session): the replay costs 5-7 us MORE per call than the
// six plain launches it replaces (REF2D, one frontier: 61.0 -> 66.7 us; 50 frontiers: 86.5 -> 92.5 us) — hipGraphLaunch is the
// expensive launch on this runtime.  The "graph" option is therefore OFF by default; the path stays, tested, for runtimes
// where that changes.
#define FS_GRAPH_MAX_N 1024      // candidates per graphed call: buckets of 2^k up to here (below the spatial sort's threshold)

template <typename F>
int run_maybe_graphed(fs_ctx *c, uint64_t key, F enqueue)
{
    if (!c->opt_graph || c->timing) return enqueue();
    if (c->graphs.size() > 256) {                              // (a caller that varies the parameters baked into a key without end)
        for (auto &g : c->graphs) if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
        c->graphs.clear();
    }
    GraphEntry &g = c->graphs[key];
    const uint64_t stamp = (c->epoch << 32) ^ fs_alloc_generation;
    if (g.broken) return enqueue();
    if (g.exec && g.stamp == stamp) {
        if (hipGraphLaunch(g.exec, c->stream) == hipSuccess) return FS_OK;
        g.broken = true;
        return enqueue();
    }
    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    if (g.warm_stamp != stamp) {
        const int rc = enqueue();                              // plain: allocations, lazily cut images, attribute calls happen here
        g.warm_stamp = (c->epoch << 32) ^ fs_alloc_generation;
        return rc;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { g.broken = true; return enqueue(); }
    const int rc = enqueue();
    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
    const bool same_state = ((c->epoch << 32) ^ fs_alloc_generation) == stamp;
    if (rc != FS_OK || e_end != hipSuccess || !graph || !same_state ||
        hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        g.exec = nullptr; g.broken = true;
        (void)hipGetLastError();
        return rc != FS_OK ? rc : enqueue();                   // nothing has run yet: the capture only recorded
    }
    (void)hipGraphDestroy(graph);
    g.stamp = stamp;
    if (hipGraphLaunch(g.exec, c->stream) != hipSuccess) { g.broken = true; return enqueue(); }
    return FS_OK;
}

int graph_bucket(int32_t n)
{
    int b = 1;
    while (b < n) b <<= 1;
    return b;
}

double std_min(double a, double b) { return (b < a) ? b : a; }
double std_max(double a, double b) { return (a < b) ? b : a; }

hipEvent_t get_event(fs_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ScopedTimer {
    fs_ctx *c;
    TimedLaunch t{};
    bool on;
    hipStream_t s;
    ScopedTimer(fs_ctx *ctx, int kind, hipStream_t stream = nullptr) : c(ctx), on(ctx->timing), s(stream ? stream : ctx->stream)
    {
        if (!on) return;
        t.kind = kind;
        t.start = get_event(c);
        t.stop = get_event(c);
        (void)hipEventRecord(t.start, s);
    }
    ~ScopedTimer()
    {
        if (!on) return;
        (void)hipEventRecord(t.stop, s);
        c->launches.push_back(t);
    }
};

// ---------------------------------------------------------------- lookup-table math (host, float32)
// FIP/src/fisher_information/FisherInformationHelpers.cpp:71-96,114-123 with Q = I.
float information_of_point_local(const float p[3])
{
    const float n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const float inv_n = 1 / n, inv_n3 = 1 / (n * n * n);
    float dfdp[3][3], right[3][6], jac[3][6];
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) dfdp[r][col] = inv_n * (r == col ? 1.0f : 0.0f) - (inv_n3 * p[r]) * p[col];
    const float skew[3][3] = {{0, -p[2], p[1]}, {p[2], 0, -p[0]}, {-p[1], p[0], 0}};
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 3; ++col) {
            right[r][col] = (float)(-1.0) * (r == col ? 1.0f : 0.0f);
            right[r][col + 3] = skew[r][col];
        }
    for (int r = 0; r < 3; ++r)
        for (int col = 0; col < 6; ++col) {
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc += dfdp[r][k] * right[k][col];
            jac[r][col] = acc;
        }
    float trace = 0.0f;
    for (int col = 0; col < 6; ++col) {
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) acc += jac[k][col] * jac[k][col];
        trace += acc;
    }
    return trace;
}

const float kStepMin = 0.09f, kStepMax = 0.3f, kSubSampleUntil = -1.0f;   // FisherInfoManager.hpp:25-30

// FisherInfoManager.hpp:108-123
void voxel_coordinate(float x, float y, float z, float key[3], long lattice[3])
{
    double step;
    if (std::fabs(x) < kSubSampleUntil && std::fabs(y) < kSubSampleUntil && std::fabs(z) < kSubSampleUntil) step = kStepMin;
    else step = kStepMax;
    const double r[3] = {std::round(x * (1 / step)), std::round(y * (1 / step)), std::round(z * (1 / step))};
    for (int i = 0; i < 3; ++i) {
        key[i] = (float)(r[i] * step);
        if (lattice) lattice[i] = (long)r[i];
    }
}

struct KeyBits {
    uint32_t b[3];
    bool operator==(const KeyBits &o) const { return b[0] == o.b[0] && b[1] == o.b[1] && b[2] == o.b[2]; }
};
struct KeyBitsHash {
    size_t operator()(const KeyBits &k) const
    {
        size_t h = 0;
        for (uint32_t v : k.b) h ^= std::hash<uint32_t>{}(v) + 0x9e3779b9 + (h << 6) + (h >> 2);
        return h;
    }
};
KeyBits key_bits(const float k[3])
{
    KeyBits out;
    for (int i = 0; i < 3; ++i) {
        float f = (k[i] == 0.0f) ? 0.0f : k[i];     // -0 == +0 under float equality
        std::memcpy(&out.b[i], &f, 4);
    }
    return out;
}

// FisherInfoManager.cpp:117-229 — produces the file's record sequence.
void generate_records(float minX, float maxX, float minY, float maxY, float minZ, float maxZ, std::vector<float> &rec)
{
    rec.clear();
    float max_fi = -std::numeric_limits<float>::max();
    minX = std::floor(minX * (1 / kStepMax)) * kStepMax;
    minY = std::floor(minY * (1 / kStepMax)) * kStepMax;
    minZ = std::floor(minZ * (1 / kStepMax)) * kStepMax;
    maxX = std::ceil(maxX * (1 / kStepMax)) * kStepMax;
    maxY = std::ceil(maxY * (1 / kStepMax)) * kStepMax;
    maxZ = std::ceil(maxZ * (1 / kStepMax)) * kStepMax;
    std::unordered_set<KeyBits, KeyBitsHash> existing;
    float inc = kStepMin;
    for (float cx = minX; cx <= maxX; cx += inc) {
        if (cx > kSubSampleUntil + kStepMax) inc = kStepMax;
        for (float cy = minY; cy <= maxY; cy += inc) {
            for (float cz = minZ; cz <= maxZ; cz += inc) {
                float key[3];
                voxel_coordinate(cx, cy, cz, key, nullptr);
                if (!existing.insert(key_bits(key)).second) continue;
                const float value = information_of_point_local(key);
                if (std::isnan(value)) continue;
                max_fi = std::max(max_fi, value);
                rec.insert(rec.end(), {key[0], key[1], key[2], value});
            }
        }
    }
    rec.insert(rec.end(), {0.0f, 0.0f, 0.0f, max_fi});
}

// The FIM worker's pass prediction learns the cloud's voxels-per-landmark ratio from finished calls (counters[12], fs_fim.hip);
// what it learnt holds for one cloud, one table and one visibility volume.
static void reset_voxel_ratio(fs_ctx *c)
{
    if (!c->d_counters.p) return;
    (void)hipSetDevice(c->device);                         // (callers on the table path have not bound the device yet)
    (void)hipMemsetAsync(c->d_counters.p + 12, 0, 2 * sizeof(unsigned long long), c->stream);   // both ratios (FsFimArgs::ratio_slot)
}

// Dense re-indexing of the record list by the integer voxel lattice (what loadLookupTable's
// unordered_map resolves to: later duplicates overwrite, FisherInfoManager.cpp:245-251).
int build_dense(fs_ctx *c)
{
    reset_voxel_ratio(c);
    const int64_t n = (int64_t)c->records.size() / 4;
    if (n <= 0) return fail(c, FS_E_INVALID, "lookup table has no records");
    const double step = (double)kStepMax, inv = 1 / step;
    long lo[3] = {LONG_MAX, LONG_MAX, LONG_MAX}, hi[3] = {LONG_MIN, LONG_MIN, LONG_MIN};
    std::vector<long> lat((size_t)n * 3);
    for (int64_t i = 0; i < n; ++i) {
        const float *r = &c->records[4 * i];
        for (int a = 0; a < 3; ++a) {
            const long j = std::lround((double)r[a] * inv);
            const float back = (float)((double)j * step);
            if (!(back == r[a])) return fail(c, FS_E_INVALID, "lookup record %lld is off the 0.3 m voxel lattice", (long long)i);
            lat[3 * i + a] = j;
            lo[a] = std::min(lo[a], j);
            hi[a] = std::max(hi[a], j);
        }
    }
    const int64_t dx = hi[0] - lo[0] + 1, dy = hi[1] - lo[1] + 1, dz = hi[2] - lo[2] + 1;
    if (dx * dy * dz > (int64_t)FS_MAX_TABLE_CELLS) return fail(c, FS_E_INVALID, "lookup table lattice too large (%lld cells)", (long long)(dx * dy * dz));
    c->jx0 = (int32_t)lo[0]; c->jy0 = (int32_t)lo[1]; c->jz0 = (int32_t)lo[2];
    c->tx = (int32_t)dx; c->ty = (int32_t)dy; c->tz = (int32_t)dz;
    c->dense.assign((size_t)(dx * dy * dz), std::numeric_limits<float>::quiet_NaN());
    for (int64_t i = 0; i < n; ++i) {
        const size_t idx = ((size_t)(lat[3 * i] - lo[0]) * dy + (size_t)(lat[3 * i + 1] - lo[1])) * dz + (size_t)(lat[3 * i + 2] - lo[2]);
        c->dense[idx] = c->records[4 * i + 3];
    }
    c->table_full = true;
    for (float v : c->dense) if (!std::isfinite(v)) { c->table_full = false; break; }   // (NaN = absent; an infinite value also takes the guarded kernel)
    FS_HIP(c, c->d_table.ensure(c->dense.size()));
    FS_HIP(c, hipMemcpyAsync(c->d_table.p, c->dense.data(), c->dense.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (!c->have_factor) {
        // getFactorFromNum(num, 0.8f) — FisherInfoManager.hpp:102-106 (pow/exp in double, float return)
        std::vector<float> fac(FS_FACTOR_N, 0.0f);
        const float s = 0.8f;
        for (int k = 1; k < FS_FACTOR_N; ++k) fac[k] = (float)std::exp(1 - std::pow((double)k, (double)s));
        for (int k = 1; k <= 4; ++k) c->fac[k] = fac[k];
        FS_HIP(c, c->d_factor.ensure(fac.size()));
        FS_HIP(c, hipMemcpyAsync(c->d_factor.p, fac.data(), fac.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        FS_HIP(c, hipStreamSynchronize(c->stream));
        c->have_factor = true;
    }
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->have_table = true;
    ++c->epoch;
    return FS_OK;
}

// getTransformFromPose (FisherInformationHelpers.cpp:16-26): float translation, Eigen::Quaternionf -> rotation.
void pose_to_rt(const double pose7[7], float Rt[12])
{
    const float x = (float)pose7[3], y = (float)pose7[4], z = (float)pose7[5], w = (float)pose7[6];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Rt[0] = 1.0f - (tyy + tzz); Rt[1] = txy - twz;          Rt[2] = txz + twy;
    Rt[3] = txy + twz;          Rt[4] = 1.0f - (txx + tzz); Rt[5] = tyz - twx;
    Rt[6] = txz - twy;          Rt[7] = tyz + twx;          Rt[8] = 1.0f - (txx + tyy);
    Rt[9] = (float)pose7[0]; Rt[10] = (float)pose7[1]; Rt[11] = (float)pose7[2];
}

// Which image of the grid the arrival fan walks.  The class walk spends ~1.3x the instructions per step and touches 4-8x fewer
// cache lines (L2 -> L1 fill 4.4 GB -> 0.5 GB per C3 launch).  Measured (profiles/r03/ray_class_walk.json): on 3-D grids it wins
// at every ray length (C3: 0.190 against 0.229 ms at 40 cells, 0.63 against 1.12 ms at 160); on a 2-D costmap a short fan lives
// in L1 either way and the byte walk's cheaper set-up wins (REF2D: 0.059 against 0.073 ms) until the rays get long.
// "ray.layout" forces one.
#ifndef FS_CLASS_WALK_FROM
#define FS_CLASS_WALK_FROM 96.0
#endif
bool use_class_walk(const fs_ctx *c, double max_length_cells)
{
    // (the walk forms brick addresses with 24-bit multiplies: the largest brick stride, 512 * bricks_x * bricks_y, must fit)
    const uint64_t stride = 512ull * (uint64_t)((c->nx + 7) >> 3) * (uint64_t)((c->ny + 7) >> 3);
    // (... and it forms the cell address A in 32 bits: a grid thin in x / y and deep in z can pass the 2^31-cell limit and still
    // pad to 2^32 class cells or more — 4 x 4 x 2^26 does; such a grid keeps the byte walk)
    const uint64_t padded_cells = stride * (uint64_t)((c->nz + 7) >> 3);
    if (c->opt_layout == 1 || stride >= (1ull << 24) || padded_cells >= (1ull << 32)) return false;
    if (c->opt_layout >= 2) return true;
    return c->nz > 1 || max_length_cells >= FS_CLASS_WALK_FROM;
}

FsGridDev grid_dev(const fs_ctx *c)
{
    const uint32_t bx = (uint32_t)(c->nx + 7) >> 3, by = (uint32_t)(c->ny + 7) >> 3, bz = (uint32_t)(c->nz + 7) >> 3;
    return FsGridDev{c->d_cells.p, c->nx, c->ny, c->nz, c->origin[0], c->origin[1], c->origin[2], c->res, c->d_counters.p + 29,
                     c->have_cls ? c->d_cls.p : nullptr, {512u - 8u, 512u * bx - 64u, 512u * bx * by - 512u}, 512u * bx * by * bz, nullptr};
}

// every upload path ends here: images derived from the grid are cut again on next use
int retile_grid(fs_ctx *c, int32_t, int32_t, int32_t)
{
    c->have_cls = false;
    c->have_sparse = false;
    return FS_OK;
}

// The sparse form of the class image: the dense one (already cut on the device) is read back, every brick whose 512 cells are
// of ONE class points at the shared uniform brick of that class (pool slots 0..3), every other brick gets a slot of its own.
// A staging-time pass on the host — this is the experiment's set-up, not a path anybody waits for per tick.
int build_sparse_class_image(fs_ctx *c)
{
    const uint64_t bricks = (uint64_t)((c->nx + 7) >> 3) * (uint64_t)((c->ny + 7) >> 3) * (uint64_t)((c->nz + 7) >> 3);
    std::vector<uint32_t> dense(bricks * 32), table(bricks), pool;
    FS_HIP(c, hipMemcpyAsync(dense.data(), c->d_cls.p, dense.size() * 4, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    static const uint32_t uniform[4] = {0x00000000u, 0x55555555u, 0xAAAAAAAAu, 0xFFFFFFFFu};
    pool.reserve(4 * 32 + (size_t)(bricks / 2) * 32);
    for (int k = 0; k < 4; ++k) pool.insert(pool.end(), 32, uniform[k]);
    for (uint64_t b = 0; b < bricks; ++b) {
        const uint32_t *w = &dense[b * 32];
        bool same = true;
        for (int i = 1; i < 32 && same; ++i) same = w[i] == w[0];
        int k = -1;
        if (same) for (int u = 0; u < 4; ++u) if (w[0] == uniform[u]) k = u;
        if (k >= 0) { table[b] = (uint32_t)k; continue; }
        table[b] = (uint32_t)(pool.size() / 32);
        pool.insert(pool.end(), w, w + 32);
    }
    if (pool.size() / 32 >= (1ull << 23)) return fail(c, FS_E_INVALID, "sparse class image: more than 2^23 distinct bricks");   // (slot << 9 must stay in 32 bits)
    FS_HIP(c, c->d_cls_table.ensure(table.size())); FS_HIP(c, c->d_cls_pool.ensure(pool.size()));
    FS_HIP(c, hipMemcpyAsync(c->d_cls_table.p, table.data(), table.size() * 4, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(c->d_cls_pool.p, pool.data(), pool.size() * 4, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->sparse_bricks = bricks; c->sparse_pool_bricks = pool.size() / 32;
    c->have_sparse = true;
    ++c->epoch;
    return FS_OK;
}

int check_scoring_state(fs_ctx *c, bool need_rays, bool need_fim)
{
    if (need_rays && !c->have_ray) return fail(c, FS_E_STATE, "fs_set_ray_params has not been called");
    if (need_rays && !c->have_grid) return fail(c, FS_E_STATE, "fs_upload_grid has not been called");
    if (need_fim && !c->have_lm) return fail(c, FS_E_STATE, "fs_upload_landmarks has not been called");
    if (need_fim && !c->have_table) return fail(c, FS_E_STATE, "no lookup table: call fs_lookup_generate / fs_lookup_load");
    return FS_OK;
}

int fill_ray_args(fs_ctx *c, FsRayArgs &a, bool class_ok = true)
{
    const fs_ray_params &p = c->rp;
    a.grid = grid_dev(c);
    a.dir = c->d_dir.p;
    a.n_yaw = c->n_yaw; a.n_elev = c->n_elev; a.window = c->window;
    a.max_length = (unsigned int)(p.max_camera_depth / c->res);             // CostCalculator.cpp:28
    // Long fans walk the class image, cut for exactly this visitor (the ray parameters' ranges; a launch with another visitor
    // — fs_max_arrival — passes class_ok = false and walks the byte image).  It is cut on first use and again when the map or
    // the ranges change: one streaming pass over the grid.
    a.layout = 0;
    if (class_ok && use_class_walk(c, (double)a.max_length)) {
        const int32_t want[4] = {p.obst_min, p.obst_max, p.trace_min, p.trace_max};
        if (!c->have_cls || std::memcmp(want, c->cls_ranges, sizeof want) != 0) {
            if (c->d_cls.ensure(fs_class_image_words(c->nx, c->ny, c->nz)) != hipSuccess ||
                fs_launch_classify(c->d_cells.p, c->d_cls.p, c->nx, c->ny, c->nz, want[0], want[1], want[2], want[3], c->stream) != hipSuccess)
                return fail(c, FS_E_HIP, "could not stage the class image of the grid");
            std::memcpy(c->cls_ranges, want, sizeof want);
            c->have_cls = true;
            c->have_sparse = false;
            ++c->epoch;
        }
        a.grid = grid_dev(c);
        a.layout = 1;
        if (c->opt_layout == 3) {
            if (!c->have_sparse) { const int rc = build_sparse_class_image(c); if (rc) return rc; }
            a.grid.cls = c->d_cls_pool.p;
            a.grid.cls_table = c->d_cls_table.p;
            a.layout = 2;
        }
    }
    a.obst_min = p.obst_min; a.obst_max = p.obst_max; a.trace_min = p.trace_min; a.trace_max = p.trace_max;
    a.clamp = 1;
    // CostCalculator.cpp:47-48, getSizeInMeters = (size - 1 + 0.5) * resolution
    const double smx = (c->nx - 1 + 0.5) * c->res, smy = (c->ny - 1 + 0.5) * c->res, smz = (c->nz - 1 + 0.5) * c->res;
    a.lo_x = std_max(p.polygon[0], c->origin[0]);
    a.hi_x = std_min(p.polygon[2], c->origin[0] + smx);
    a.lo_y = std_max(p.polygon[1], c->origin[1]);
    a.hi_y = std_min(p.polygon[3], c->origin[1] + smy);
    a.lo_z = c->origin[2];
    a.hi_z = c->origin[2] + smz;
    a.footprint_radius = std::ceil(p.robot_radius / c->res);               // CostCalculator.cpp:77
    a.delta_theta = p.delta_theta;
    a.half_fov = p.camera_fov / 2;
    a.min_gt = c->min_gt;
    return FS_OK;
}

int fill_fim_args(fs_ctx *c, FsFimArgs &a)
{
    a.lx = c->d_lx.p; a.ly = c->d_ly.p; a.lz = c->d_lz.p;
    a.spheres = c->d_spheres.p;
    a.n_chunks = c->n_chunks;
    a.cull = c->opt_cull ? 1 : 0;
    a.table = c->d_table.p;
    a.jx0 = c->jx0; a.jy0 = c->jy0; a.jz0 = c->jz0;
    a.tx = c->tx; a.ty = c->ty; a.tz = c->tz;
    a.inv_step = 1 / (double)kStepMax;
    a.inv_step_f = (float)a.inv_step;
    a.factor = c->d_factor.p;
    a.fac1 = c->fac[1]; a.fac2 = c->fac[2]; a.fac3 = c->fac[3]; a.fac4 = c->fac[4];
    a.table_full = c->table_full ? 1 : 0;
    a.maxd2 = (float)(c->fp.max_dist * c->fp.max_dist);
    a.max_dist_f = (float)c->fp.max_dist * 1.0001f + 1.0e-3f;      // culling reach, rounded outwards
    a.far_lattice = !((double)a.max_dist_f * a.inv_step < 1000.0) ? 1 : 0;   // (NaN / inf / huge ranges: exact path for every landmark)
    {   // fl32(p * fl32(1/step)) is within |r| * 2^-23 of the double product the reference rounds (one rounding of the factor, one of
        // the product); twice that at the largest |r| a scored landmark can have, plus an absolute cushion, is the band around a
        // half-integer inside which the kernel re-evaluates in fp64 (0.4995 covered |r| < 2^10 wholesale: 40 times the lanes at 14 m)
        const double r_max = std::max(1.0, (double)a.max_dist_f * a.inv_step);
        const double band = 2.0 * r_max * 1.1920929e-7 + 1.0e-6;
        a.key_thr = a.far_lattice ? 0.0f : (float)(0.5 - band);
    }
    if (c->fp.max_angle >= M_PI) {
        a.cone_mode = 0; a.cos2 = 0.0f; a.cos_a = -1.0f; a.sin_a = 0.0f;
    } else {
        const float cs = (float)std::cos(c->fp.max_angle);
        a.cos2 = cs * cs;
        a.cone_mode = (cs >= 0.0f) ? 1 : 2;
        // chunk culling uses a slightly wider cone (alpha + 1e-3 rad) than the exact per-landmark predicate
        a.cos_a = (float)std::cos(c->fp.max_angle + 1.0e-3);
        a.sin_a = (float)std::sin(c->fp.max_angle + 1.0e-3);
        if (c->fp.max_angle + 1.0e-3 >= M_PI / 2) a.cone_mode = (a.cone_mode == 1) ? 3 : 2;   // 3: exact predicate of mode 1, no cone culling
    }
    // LDS tier: 2^14 slots (64 KiB, two 512-thread workgroups per CU) or fewer for small clouds; crowded poses are
    // scored in passes; the HBM tier behind it takes what still overflows
    int bits = 10;
    while (bits < c->opt_bits1 && (1 << bits) < 2 * c->m) ++bits;
    // the table's box in the camera frame: lattice index j holds |p / step - j| <= 0.5, so a landmark outside
    // [(j0 - 0.5) step, (j0 + t - 0.5) step] on any axis misses the table whatever the others are; 1 mm outwards covers the
    // rounding of the kernel's fp32 products (|p| <= max_dist).  Used by the INFO_ONLY worker only.
    {
        const double step = (double)kStepMax;
        const int32_t j0[3] = {c->jx0, c->jy0, c->jz0}, tn[3] = {c->tx, c->ty, c->tz};
        for (int k = 0; k < 3; ++k) {
            a.box_lo[k] = std::nextafter((float)(((double)j0[k] - 0.5) * step - 1.0e-3), -INFINITY);
            a.box_hi[k] = std::nextafter((float)(((double)(j0[k] + tn[k]) - 0.5) * step + 1.0e-3), INFINITY);
        }
    }
    a.info_only = 0; a.yaw_only = 0;
    a.learn = c->opt_learn ? 1 : 0;
    a.hash_bits = bits;
    a.skip32 = c->opt_skip32;
    a.headroom = c->opt_headroom;
    a.gtable = c->d_gtable.p;
    a.ghash_bits = c->ghash_bits;
    a.counters = c->d_counters.p;
    return FS_OK;
}

int ensure_candidate_scratch(fs_ctx *c, size_t n, bool want_fim21)
{
    FS_HIP(c, c->d_arrival.ensure(n)); FS_HIP(c, c->d_argmax.ensure(n)); FS_HIP(c, c->d_status.ensure(n));
    FS_HIP(c, c->d_yaw.ensure(n)); FS_HIP(c, c->d_ach.ensure(n));
    FS_HIP(c, c->d_info.ensure(n)); FS_HIP(c, c->d_trace.ensure(n)); FS_HIP(c, c->d_logdet.ensure(n));
    FS_HIP(c, c->d_nvis.ensure(n)); FS_HIP(c, c->d_nvox.ensure(n)); FS_HIP(c, c->d_overflow.ensure(n));
    FS_HIP(c, c->d_sums.ensure(n * 18));
    FS_HIP(c, c->d_flagged.ensure(n * 2));
    if (c->d_tested.cap < n) {
        FS_HIP(c, c->d_tested.ensure(n));
        FS_HIP(c, hipMemsetAsync(c->d_tested.p, 0, c->d_tested.cap * sizeof(uint32_t), c->stream));
    }
    if (want_fim21) FS_HIP(c, c->d_fim21.ensure(n * 21));
    return FS_OK;
}

}  // namespace

// ================================================================== C ABI

extern "C" {

int fs_abi_version(void) { return FS_ABI_VERSION; }

int fs_ctx_create(int device_id, void *stream, fs_ctx **out)
{
    if (!out) return FS_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return FS_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return FS_E_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FS_E_NO_DEVICE;   // kernels are built for gfx950 only
    if (hipSetDevice(device_id) != hipSuccess) return FS_E_NO_DEVICE;
    fs_ctx *c = new fs_ctx();
    c->device = device_id;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return FS_E_HIP;
        }
        c->own_stream = true;
    }
    // device-side counters (fs_get_counter; 29 / 30: range checks of FS_BOUNDS builds)
    if (c->d_counters.ensure(FS_N_COUNTERS) != hipSuccess ||
        hipMemsetAsync(c->d_counters.p, 0, FS_N_COUNTERS * sizeof(unsigned long long), c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        fs_ctx_destroy(c);
        return FS_E_HIP;
    }
    *out = c;
    return FS_OK;
}

void fs_ctx_destroy(fs_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &t : c->launches) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto &g : c->graphs) if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    c->d_out.release();
    c->d_dir.release(); c->d_yawR.release(); c->d_cells.release(); c->d_cls.release(); c->d_cls_table.release(); c->d_cls_pool.release();
    c->d_lx.release(); c->d_ly.release(); c->d_lz.release(); c->d_spheres.release(); c->d_counters.release();
    c->d_table.release(); c->d_factor.release(); c->d_gtable.release();
    c->d_kf_check.release(); c->d_kf_tri.release(); c->d_kf_off.release(); c->d_kf_flagged.release(); c->d_kf_cells.release();
    c->d_kf_points.release(); c->d_kpx.release(); c->d_kpy.release(); c->d_kpz.release(); c->d_kf_gtable.release();
    c->d_kf_counters.release();
    c->d_goal.release(); c->d_yaw.release(); c->d_len.release(); c->d_head.release();
    c->d_cost.release(); c->d_au.release(); c->d_du.release(); c->d_sums.release();
    c->d_fsize.release(); c->d_arrival.release(); c->d_argmax.release(); c->d_status.release();
    c->d_nvis.release(); c->d_nvox.release(); c->d_raycounts.release(); c->d_order.release(); c->d_err.release();
    c->d_black.release(); c->d_achin.release(); c->d_ach.release();
    c->d_info.release(); c->d_trace.release(); c->d_logdet.release(); c->d_fim21.release(); c->d_Rt.release();
    c->d_overflow.release(); c->d_tested.release(); c->d_split_flags.release(); c->d_flagged.release(); c->d_records.release();
    if (c->rank_scratch) (void)hipFree(c->rank_scratch);
    if (c->sort_scratch) (void)hipFree(c->sort_scratch);
    c->d_perm.release(); c->h_in.release(); c->h_out.release(); c->d_in.release(); c->h_gin.release(); c->d_gin.release(); c->h_fin.release(); c->h_win.release(); c->d_win.release();
    c->d_cloud_raw.release(); c->d_cloud_perm.release(); c->d_cloud_bounds.release(); c->d_cloud_keys.release(); c->d_cloud_temp.release(); c->d_cloud_bbox.release();
    c->d_seg_start.release(); c->d_seg_end.release(); c->d_tri.release(); c->d_seg_ok.release(); c->d_seg_hit.release();
    c->d_mask.release(); c->d_brick_cells.release(); c->d_seg_traced.release(); c->d_seg_unknown.release();
    c->d_seg_all.release(); c->d_brick_xyz.release(); c->d_bad.release(); c->d_count.release();
    c->d_fc_parent_t.release(); c->d_fc_parent_f.release(); c->d_fc_aux.release(); c->d_fc_state.release(); c->d_fc_labels.release();
    c->d_fc_queue.release(); c->d_fc_visited.release(); c->d_fc_clusters.release(); c->d_fc_sums.release();
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *fs_last_error(const fs_ctx *c) { return c ? c->err.c_str() : "null context"; }

int fs_synchronize(fs_ctx *c)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    return FS_OK;
}

int fs_enable_kernel_timing(fs_ctx *c, int enable)
{
    if (!c) return FS_E_INVALID;
    c->timing = enable != 0;
    return FS_OK;
}

int fs_kernel_time(fs_ctx *c, int kind, double *total_ms, int64_t *launches)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    int64_t cnt = 0;
    std::vector<TimedLaunch> keep;
    for (auto &t : c->launches) {
        if (t.kind == kind) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) { tot += ms; ++cnt; }
            c->event_pool.push_back(t.start);
            c->event_pool.push_back(t.stop);
        } else {
            keep.push_back(t);
        }
    }
    c->launches.swap(keep);
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    return FS_OK;
}

// ------------------------------------------------------------------ arrival information

int fs_set_ray_params(fs_ctx *c, const fs_ray_params *p)
{
    if (!c || !p) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!(p->delta_theta > 0.0) || !(p->max_camera_depth > 0.0) || !(p->camera_fov > 0.0))
        return fail(c, FS_E_INVALID, "max_camera_depth, delta_theta and camera_fov must be positive");
    if (p->n_elev < 1 || p->n_elev > FS_MAX_ELEV) return fail(c, FS_E_INVALID, "n_elev must be in [1,%d]", FS_MAX_ELEV);
    if (!std::isfinite(p->delta_theta) || !std::isfinite(p->max_camera_depth) || !std::isfinite(p->camera_fov) || !std::isfinite(p->robot_radius) ||
        !(p->robot_radius >= 0.0))
        return fail(c, FS_E_INVALID, "max_camera_depth, delta_theta, camera_fov and robot_radius must be finite (robot_radius >= 0)");
    for (int e = 0; e < p->n_elev; ++e)
        if (!std::isfinite(p->elev[e])) return fail(c, FS_E_INVALID, "elevation angles must be finite");
    for (int k = 0; k < 4; ++k)
        if (p->polygon[k] != p->polygon[k]) return fail(c, FS_E_INVALID, "polygon bounds must not be NaN");      // (+-DBL_MAX / +-inf = no clamp)
    // DEP/src/CostCalculator.cpp:36 — accumulated theta, `theta <= 2*pi`
    std::vector<double> theta;
    if (p->n_rays > 0) {
        double t = 0;
        for (int i = 0; i < p->n_rays; ++i) { theta.push_back(t); t += p->delta_theta; }
    } else {
        for (double t = 0; t <= (2 * M_PI); t += p->delta_theta) {
            theta.push_back(t);
            if (theta.size() > 4096) break;
        }
    }
    const int n_yaw = (int)theta.size();
    const int k = static_cast<int>(p->camera_fov / p->delta_theta);          // :87
    if (n_yaw > 4096) return fail(c, FS_E_INVALID, "more than 4096 yaw rays");
    if (k < 1 || n_yaw < k)
        return fail(c, FS_E_INVALID, "fewer yaw rays (%d) than the FOV window (%d): the reference would build a negative-size vector (CostCalculator.cpp:90)", n_yaw, k);
    std::vector<double> dir((size_t)n_yaw * p->n_elev * 3);
    for (int e = 0; e < p->n_elev; ++e) {
        const double d_h = p->max_camera_depth * std::cos(p->elev[e]);
        const double d_z = p->max_camera_depth * std::sin(p->elev[e]);
        for (int i = 0; i < n_yaw; ++i) {
            double *d = &dir[((size_t)e * n_yaw + i) * 3];
            d[0] = d_h * std::cos(theta[i]);                                 // :42
            d[1] = d_h * std::sin(theta[i]);                                 // :43
            d[2] = d_z;
        }
    }
    // rotation of the pose (goal, best yaw) for every possible argmax: orientationAroundZAxis(yaw)
    // (setRPY(0,0,yaw)) -> Quaternionf -> rotation matrix, as isPoseSafe(Point,Point) + getTransformFromPose
    const int n_win = n_yaw - k + 1;
    std::vector<float> yawR((size_t)n_win * 9);
    for (int i = 0; i < n_win; ++i) {
        const double yaw = (i * p->delta_theta) + (p->camera_fov / 2);       // :119
        const double half = yaw * 0.5;
        const double pose7[7] = {0, 0, 0, 0.0, 0.0, std::sin(half), std::cos(half)};
        float Rt[12];
        pose_to_rt(pose7, Rt);
        std::copy(Rt, Rt + 9, &yawR[(size_t)i * 9]);
    }
    bool yaw_exact = true;
    for (int i = 0; i < n_win; ++i) {
        const float *R = &yawR[(size_t)i * 9];
        yaw_exact = yaw_exact && R[2] == 0.0f && R[5] == 0.0f && R[6] == 0.0f && R[7] == 0.0f && R[8] == 1.0f;
    }
    FS_HIP(c, c->d_dir.ensure(dir.size()));
    FS_HIP(c, c->d_yawR.ensure(yawR.size()));
    FS_HIP(c, hipMemcpyAsync(c->d_dir.p, dir.data(), dir.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(c->d_yawR.p, yawR.data(), yawR.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->rp = *p;
    c->n_yaw = n_yaw; c->n_elev = p->n_elev; c->window = k;
    c->yaw_exact = yaw_exact;
    c->have_ray = true;
    ++c->epoch;
    c->max_gt = 0.0; c->min_gt = 0.0;
    return FS_OK;
}

int fs_ray_fan_shape(const fs_ctx *c, int32_t *n_yaw, int32_t *n_elev, int32_t *window)
{
    if (!c || !c->have_ray) return FS_E_STATE;
    if (n_yaw) *n_yaw = c->n_yaw;
    if (n_elev) *n_elev = c->n_elev;
    if (window) *window = c->window;
    return FS_OK;
}

int fs_upload_grid(fs_ctx *c, const uint8_t *cells, int32_t nx, int32_t ny, int32_t nz,
                   const double origin_xyz[3], double resolution)
{
    if (!c || !cells || !origin_xyz) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (nx <= 0 || ny <= 0 || nz <= 0 || !(resolution > 0.0) || !std::isfinite(resolution)) return fail(c, FS_E_INVALID, "bad grid shape or resolution");
    // (worldToMap with a NaN origin is a float-to-integer conversion of NaN: undefined in the reference, refused here)
    if (!std::isfinite(origin_xyz[0]) || !std::isfinite(origin_xyz[1]) || !std::isfinite(origin_xyz[2])) return fail(c, FS_E_INVALID, "grid origin must be finite");
    const uint64_t total = (uint64_t)nx * (uint64_t)ny * (uint64_t)nz;
    // (cell offsets are 32-bit unsigned in the walks, one z step is a signed 32-bit stride; everything else indexes in 64 bits)
    if (total >= (1ull << 32) || (uint64_t)nx * (uint64_t)ny >= (1ull << 31)) return fail(c, FS_E_INVALID, "dense grids are limited to 2^32 cells (and 2^31 per z slice)");
    FS_HIP(c, c->d_cells.ensure((size_t)total));
    FS_HIP(c, hipMemcpyAsync(c->d_cells.p, cells, (size_t)total, hipMemcpyHostToDevice, c->stream));
    {
        const int rc = retile_grid(c, nx, ny, nz);
        if (rc) return rc;
    }
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->origin[0] = origin_xyz[0]; c->origin[1] = origin_xyz[1]; c->origin[2] = origin_xyz[2];
    c->res = resolution;
    c->have_grid = true;
    c->max_gt = 0.0; c->min_gt = 0.0;
    ++c->epoch;
    return FS_OK;
}

// Layer::updateCosts' window of the master grid (include/fitslam_frontier.h): packed on the host into page-locked memory, one
// transfer, one scatter into the row-major image, and the class image re-cut for the bricks the window touches (a whole-map
// re-cut is a streaming pass over the grid — 1 GiB on C5 — for a window of a few thousand cells).
int fs_update_grid_region(fs_ctx *c, int32_t x0, int32_t y0, int32_t z0, int32_t sx, int32_t sy, int32_t sz,
                          const uint8_t *cells, int64_t row_stride, int64_t slice_stride)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!c->have_grid) return fail(c, FS_E_STATE, "fs_upload_grid has not been called");
    if (sx < 0 || sy < 0 || sz < 0) return fail(c, FS_E_INVALID, "negative window size");
    if (x0 < 0 || y0 < 0 || z0 < 0 || (int64_t)x0 + sx > c->nx || (int64_t)y0 + sy > c->ny || (int64_t)z0 + sz > c->nz)
        return fail(c, FS_E_INVALID, "window [%d,%lld) x [%d,%lld) x [%d,%lld) leaves the %d x %d x %d grid", x0, (long long)x0 + sx, y0, (long long)y0 + sy,
                    z0, (long long)z0 + sz, c->nx, c->ny, c->nz);
    if (sx == 0 || sy == 0 || sz == 0) return FS_OK;
    if (!cells) return FS_E_INVALID;
    if (row_stride == 0) row_stride = sx;
    if (slice_stride == 0) slice_stride = row_stride * (int64_t)sy;
    if (row_stride < sx || slice_stride < row_stride * (int64_t)(sy - 1) + sx) return fail(c, FS_E_INVALID, "window strides smaller than the window");
    const size_t total = (size_t)sx * (size_t)sy * (size_t)sz;
    FS_HIP(c, c->h_win.ensure(total));
    FS_HIP(c, c->d_win.ensure(total));
    for (int32_t z = 0; z < sz; ++z)
        for (int32_t y = 0; y < sy; ++y)
            std::memcpy(c->h_win.p + ((size_t)z * sy + y) * sx, cells + (size_t)z * (size_t)slice_stride + (size_t)y * (size_t)row_stride, (size_t)sx);
    FS_HIP(c, hipMemcpyAsync(c->d_win.p, c->h_win.p, total, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, fs_launch_window_scatter(c->d_win.p, c->d_cells.p, c->nx, c->ny, x0, y0, z0, sx, sy, sz, c->stream));
    if (c->have_cls) {
        const int b0[3] = {x0 >> 3, y0 >> 3, z0 >> 3};
        const int nb[3] = {((x0 + sx - 1) >> 3) - b0[0] + 1, ((y0 + sy - 1) >> 3) - b0[1] + 1, ((z0 + sz - 1) >> 3) - b0[2] + 1};
        FS_HIP(c, fs_launch_classify_region(c->d_cells.p, c->d_cls.p, c->nx, c->ny, c->nz, c->cls_ranges[0], c->cls_ranges[1], c->cls_ranges[2],
                                            c->cls_ranges[3], b0, nb, c->stream));
    }
    c->have_sparse = false;
    FS_HIP(c, hipStreamSynchronize(c->stream));                  // (the page-locked window is the next call's again)
    if (c->h_win.cap > ((size_t)64 << 20)) { c->h_win.release(); c->d_win.release(); }   // (a window of map size: not worth keeping)
    ++c->epoch;
    return FS_OK;
}

int fs_upload_grid_bricks(fs_ctx *c, int32_t nx, int32_t ny, int32_t nz, const double origin_xyz[3], double resolution,
                          uint8_t default_value, int64_t n_bricks, const int32_t *brick_xyz, const uint8_t *brick_cells)
{
    if (!c || !origin_xyz || n_bricks < 0 || (n_bricks > 0 && (!brick_xyz || !brick_cells))) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (nx <= 0 || ny <= 0 || nz <= 0 || !(resolution > 0.0) || !std::isfinite(resolution)) return fail(c, FS_E_INVALID, "bad grid shape or resolution");
    // (worldToMap with a NaN origin is a float-to-integer conversion of NaN: undefined in the reference, refused here)
    if (!std::isfinite(origin_xyz[0]) || !std::isfinite(origin_xyz[1]) || !std::isfinite(origin_xyz[2])) return fail(c, FS_E_INVALID, "grid origin must be finite");
    if ((nx & 7) || (ny & 7) || (nz & 7)) return fail(c, FS_E_INVALID, "brick upload needs dimensions that are multiples of 8");
    const uint64_t total = (uint64_t)nx * (uint64_t)ny * (uint64_t)nz;
    // (cell offsets are 32-bit unsigned in the walks, one z step is a signed 32-bit stride; everything else indexes in 64 bits)
    if (total >= (1ull << 32) || (uint64_t)nx * (uint64_t)ny >= (1ull << 31)) return fail(c, FS_E_INVALID, "dense grids are limited to 2^32 cells (and 2^31 per z slice)");
    FS_HIP(c, c->d_cells.ensure((size_t)total));
    FS_HIP(c, hipMemsetAsync(c->d_cells.p, default_value, (size_t)total, c->stream));
    if (n_bricks > 0) {
        DevBuf<int32_t> &d_xyz = c->d_brick_xyz, &d_bad = c->d_bad;
        DevBuf<uint8_t> &d_bc = c->d_brick_cells;
        FS_HIP(c, d_xyz.ensure((size_t)n_bricks * 3)); FS_HIP(c, d_bc.ensure((size_t)n_bricks * 512)); FS_HIP(c, d_bad.ensure(1));
        FS_HIP(c, hipMemsetAsync(d_bad.p, 0, 4, c->stream));
        FS_HIP(c, hipMemcpyAsync(d_xyz.p, brick_xyz, sizeof(int32_t) * 3 * (size_t)n_bricks, hipMemcpyHostToDevice, c->stream));
        FS_HIP(c, hipMemcpyAsync(d_bc.p, brick_cells, (size_t)n_bricks * 512, hipMemcpyHostToDevice, c->stream));
        FS_HIP(c, fs_launch_brick_scatter(n_bricks, d_xyz.p, d_bc.p, c->d_cells.p, nx, ny, nz, d_bad.p, c->stream));
        int32_t bad = 0;
        FS_HIP(c, hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, c->stream));
        FS_HIP(c, hipStreamSynchronize(c->stream));
        if (d_bc.cap > ((size_t)64 << 20)) { d_bc.release(); d_xyz.release(); }   // a whole-map brick list (C5: 385 MB) is not worth keeping
        if (bad) { c->have_grid = false; return fail(c, FS_E_INVALID, "a brick lies outside the grid"); }
    }
    {
        const int rc = retile_grid(c, nx, ny, nz);
        if (rc) return rc;
    }
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->nx = nx; c->ny = ny; c->nz = nz;
    c->origin[0] = origin_xyz[0]; c->origin[1] = origin_xyz[1]; c->origin[2] = origin_xyz[2];
    c->res = resolution;
    c->have_grid = true;
    c->max_gt = 0.0; c->min_gt = 0.0;
    ++c->epoch;
    return FS_OK;
}

int fs_frontier_cells(fs_ctx *c, int32_t lethal_threshold, uint8_t *mask, int64_t *count)
{
    if (!c || !count) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!c->have_grid) return fail(c, FS_E_STATE, "fs_upload_grid has not been called");
    const size_t total = (size_t)c->nx * c->ny * c->nz;
    DevBuf<uint8_t> &d_mask = c->d_mask;
    DevBuf<unsigned long long> &d_count = c->d_count;
    if (mask) FS_HIP(c, d_mask.ensure(total));
    FS_HIP(c, d_count.ensure(1));
    FS_HIP(c, hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), c->stream));
    {
        ScopedTimer t(c, 5);
        FS_HIP(c, fs_launch_frontier_cells(c->d_cells.p, c->nx, c->ny, c->nz, lethal_threshold, mask ? d_mask.p : nullptr, d_count.p, c->stream));
    }
    unsigned long long n = 0;
    FS_HIP(c, hipMemcpyAsync(&n, d_count.p, sizeof n, hipMemcpyDeviceToHost, c->stream));
    if (mask) FS_HIP(c, hipMemcpyAsync(mask, d_mask.p, total, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    *count = (int64_t)n;
    if (d_mask.cap > ((size_t)64 << 20)) d_mask.release();       // a whole-map mask (128 MB at 512^3, 1 GiB at 1024^3) is not worth keeping between ticks
    return FS_OK;
}

int fs_frontier_clusters(fs_ctx *c, const double robot_xy[2], int32_t lethal_threshold, double max_frontier_distance,
                         int32_t max_frontier_cluster_size, int32_t *labels, int32_t max_clusters,
                         fs_frontier_cluster *clusters, int32_t *n_clusters, int64_t *n_cells)
{
    if (!c || !robot_xy || !n_clusters || max_clusters < 0 || (max_clusters > 0 && !clusters)) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!c->have_grid) return fail(c, FS_E_STATE, "fs_upload_grid has not been called");
    if (c->nz != 1) return fail(c, FS_E_INVALID, "the frontier search is defined on a 2-D costmap (nz == 1)");
    *n_clusters = 0;
    if (n_cells) *n_cells = 0;
    const size_t cells = (size_t)c->nx * c->ny;
    // :26-33 — worldToMap of the robot position; off the map: no frontiers
    const double px = robot_xy[0], py = robot_xy[1];
    bool on_map = !(px < c->origin[0] || py < c->origin[1]);
    unsigned int mx = 0, my = 0;
    if (on_map) {
        const double qx = (px - c->origin[0]) / c->res, qy = (py - c->origin[1]) / c->res;
        on_map = qx < 4294967296.0 && qy < 4294967296.0;
        if (on_map) {
            mx = static_cast<unsigned int>(qx); my = static_cast<unsigned int>(qy);
            on_map = mx < (unsigned int)c->nx && my < (unsigned int)c->ny;
        }
    }
    if (!on_map) {
        if (labels) std::fill(labels, labels + cells, -1);
        return FS_OK;
    }
    FS_HIP(c, c->d_fc_parent_t.ensure(cells)); FS_HIP(c, c->d_fc_parent_f.ensure(cells)); FS_HIP(c, c->d_fc_aux.ensure(cells));
    FS_HIP(c, c->d_fc_queue.ensure(cells)); FS_HIP(c, c->d_fc_visited.ensure(cells)); FS_HIP(c, c->d_fc_state.ensure(8));
    if (labels) FS_HIP(c, c->d_fc_labels.ensure(cells));
    FS_HIP(c, c->d_fc_clusters.ensure((size_t)std::max(max_clusters, 1))); FS_HIP(c, c->d_fc_sums.ensure(2 * (size_t)std::max(max_clusters, 1)));
    const double reach = max_frontier_distance + (max_frontier_cluster_size * c->res * 1.414);      // :67
    {
        ScopedTimer t(c, 5);
        FS_HIP(c, fs_launch_frontier_clusters(c->d_cells.p, c->nx, c->ny, c->origin[0], c->origin[1], c->res, px, py,
                                              (int32_t)(my * (unsigned int)c->nx + mx), reach, lethal_threshold,
                                              c->d_fc_parent_t.p, c->d_fc_parent_f.p, c->d_fc_aux.p, c->d_fc_queue.p, c->d_fc_visited.p,
                                              c->d_fc_state.p, labels ? c->d_fc_labels.p : nullptr, max_clusters, c->d_fc_clusters.p,
                                              c->d_fc_sums.p, c->stream));
    }
    // results through the page-locked buffer: the counters, the first clusters (as many as a map of this size usually has) and
    // the labels are requested together and waited for once; only a map with more clusters costs a second round trip
    const size_t first = (size_t)std::min<int32_t>(max_clusters, 4096);
    const size_t o_state = 0, o_cl = 64, o_labels = (o_cl + first * sizeof(fs_frontier_cluster) + 63) & ~(size_t)63;
    FS_HIP(c, c->h_out.ensure(o_labels + (labels ? sizeof(int32_t) * cells : 0)));
    FS_HIP(c, hipMemcpyAsync(c->h_out.p + o_state, c->d_fc_state.p, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (first) FS_HIP(c, hipMemcpyAsync(c->h_out.p + o_cl, c->d_fc_clusters.p, first * sizeof(fs_frontier_cluster), hipMemcpyDeviceToHost, c->stream));
    if (labels) FS_HIP(c, hipMemcpyAsync(c->h_out.p + o_labels, c->d_fc_labels.p, sizeof(int32_t) * cells, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    int32_t state[8];
    std::memcpy(state, c->h_out.p + o_state, sizeof state);
    if (labels) std::memcpy(labels, c->h_out.p + o_labels, sizeof(int32_t) * cells);
    const int32_t stored = std::min(state[5], max_clusters);
    if (stored > 0) {
        std::memcpy(clusters, c->h_out.p + o_cl, sizeof(fs_frontier_cluster) * std::min<size_t>((size_t)stored, first));
        if ((size_t)stored > first) {
            FS_HIP(c, hipMemcpyAsync(clusters + first, c->d_fc_clusters.p + first, sizeof(fs_frontier_cluster) * ((size_t)stored - first), hipMemcpyDeviceToHost, c->stream));
            FS_HIP(c, hipStreamSynchronize(c->stream));
        }
        // slots were handed out in arrival order: present the clusters by ascending label
        std::sort(clusters, clusters + stored, [](const fs_frontier_cluster &u, const fs_frontier_cluster &v) { return u.label < v.label; });
    }
    *n_clusters = state[5];
    if (n_cells) *n_cells = state[6];
    if (c->d_fc_labels.cap * sizeof(int32_t) > ((size_t)64 << 20)) c->d_fc_labels.release();   // (same rule as the stencil mask and the brick list)
    return FS_OK;
}

int fs_set_arrival_limits(fs_ctx *c, double max_gt, double min_gt)
{
    if (!c) return FS_E_INVALID;
    c->max_gt = max_gt; c->min_gt = min_gt;
    ++c->epoch;
    return FS_OK;
}

int fs_max_arrival(fs_ctx *c, double *max_value, double *max_gt, double *min_gt)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    int rc = check_scoring_state(c, true, false);
    if (rc) return rc;
    FsRayArgs a{};
    if (const int rc_args = fill_ray_args(c, a, false)) return rc_args;     // (another visitor than the one a class image is cut for)
    // DEP/src/CostCalculator.cpp:140 visitor (260,260,0,255); :142-148 no clamping; start (0,0)
    a.obst_min = 260; a.obst_max = 260; a.trace_min = 0; a.trace_max = 255;
    a.clamp = 0;
    a.min_gt = 0.0;
    a.footprint_radius = 0.0;
    const double zcal = (c->nz > 1) ? c->origin[2] + 0.5 * c->nz * c->res : c->origin[2];
    const double goal[3] = {0.0, 0.0, zcal};
    FS_HIP(c, c->d_goal.ensure(3));
    rc = ensure_candidate_scratch(c, 1, false);
    if (rc) return rc;
    FS_HIP(c, hipMemcpyAsync(c->d_goal.p, goal, sizeof goal, hipMemcpyHostToDevice, c->stream));
    a.n = 1; a.goal = c->d_goal.p;
    a.arrival = c->d_arrival.p; a.argmax = c->d_argmax.p; a.status = c->d_status.p;
    a.yaw = c->d_yaw.p; a.achievable = c->d_ach.p;
    {
        ScopedTimer t(c, 0);
        FS_HIP(c, fs_launch_raymarch(a, c->stream));
    }
    int32_t arrival = 0, status = 0;
    FS_HIP(c, hipMemcpyAsync(&arrival, c->d_arrival.p, 4, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipMemcpyAsync(&status, c->d_status.p, 4, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    if (status != FS_STATUS_OK) {
        // :145-148 — returns 0 and leaves the limits unset
        if (max_value) *max_value = 0.0;
        if (max_gt) *max_gt = c->max_gt;
        if (min_gt) *min_gt = c->min_gt;
        return FS_OK;
    }
    c->max_gt = arrival * c->rp.factor_max;           // :186
    c->min_gt = c->rp.factor_min * c->max_gt;         // :188
    ++c->epoch;
    if (max_value) *max_value = (double)arrival;
    if (max_gt) *max_gt = c->max_gt;
    if (min_gt) *min_gt = c->min_gt;
    return FS_OK;
}

// Spatial processing order for the ray-march kernel (outputs stay in list order).  Small lists are not worth a sort.
static int maybe_sort(fs_ctx *c, FsRayArgs &a)
{
    a.perm = nullptr;
    c->sort_keys = nullptr; c->sort_costmap = nullptr;
    if (!c->opt_sort || a.n < 2048) return FS_OK;
    FS_HIP(c, c->d_perm.ensure(a.n));
    ScopedTimer t(c, 4);
    FS_HIP(c, fs_launch_sort_candidates(a.n, a.goal, a.grid, c->d_perm.p, &c->sort_scratch, &c->sort_scratch_bytes,
                                        c->d_counters.p + 10, &c->sort_keys, &c->sort_costmap, c->opt_costmap ? 1 : 0, c->opt_sort_reverse ? 1 : 0, c->stream));
    a.perm = c->d_perm.p;
    return FS_OK;
}

static int upload_candidates(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *fsize,
                             const uint8_t *black, const uint8_t *achin)
{
    // the columns are packed into the context's page-locked buffer (every host-buffer entry point synchronises before it
    // returns, so the previous call's transfer out of that buffer is over) and travel as ONE transfer into a device buffer of
    // the same layout: goal | frontier size | blacklist | achievable
    const size_t nn = (size_t)n, pad = (nn + 15) & ~(size_t)15;
    const size_t o_goal = 0, o_fsize = o_goal + 24 * nn, o_black = ((o_fsize + 4 * nn) + 15) & ~(size_t)15, o_achin = o_black + pad;
    const size_t total = o_achin + pad;
    FS_HIP(c, c->h_in.ensure(total));
    FS_HIP(c, c->d_in.ensure(total));
    std::memcpy(c->h_in.p + o_goal, goal_xyz, 24 * nn);
    if (fsize) std::memcpy(c->h_in.p + o_fsize, fsize, 4 * nn);
    if (black) std::memcpy(c->h_in.p + o_black, black, nn);
    if (achin) std::memcpy(c->h_in.p + o_achin, achin, nn);
    const size_t used = achin ? total : (black ? o_black + pad : (fsize ? o_fsize + 4 * nn : 24 * nn));
    // a short list is read by the kernels where it lies (mapped page-locked memory): no transfer of its own
    const bool in_place = c->opt_zero_copy && n <= FS_ZERO_COPY_MAX_N && c->h_in.dev;
    const char *base = in_place ? c->h_in.dev : c->d_in.p;
    if (!in_place) FS_HIP(c, hipMemcpyAsync(c->d_in.p, c->h_in.p, used, hipMemcpyHostToDevice, c->stream));
    c->in_goal = reinterpret_cast<const double *>(base + o_goal);
    c->in_fsize = fsize ? reinterpret_cast<const int32_t *>(base + o_fsize) : nullptr;
    c->in_black = black ? reinterpret_cast<const uint8_t *>(base + o_black) : nullptr;
    c->in_achin = achin ? reinterpret_cast<const uint8_t *>(base + o_achin) : nullptr;
    return FS_OK;
}

// fs_score_arrival in two halves (see fs_score_candidates_begin / _end): everything up to the copies into the caller's
// buffers is issued here ...
int fs_score_arrival_begin(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                           const uint8_t *blacklisted, const uint8_t *achievable_in,
                           int32_t *ray_counts, int32_t *arrival, int32_t *argmax, double *yaw,
                           uint8_t *achievable, int32_t *status)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    int rc = check_scoring_state(c, true, false);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!goal_xyz || !arrival || !argmax || !yaw || !achievable || !status)))
        return fail(c, FS_E_INVALID, "null output or input pointer");
    if (n == 0) return FS_OK;
    rc = upload_candidates(c, n, goal_xyz, frontier_size, blacklisted, achievable_in);
    if (rc) return rc;
    rc = ensure_candidate_scratch(c, n, false);
    if (rc) return rc;
    const size_t per = (size_t)c->n_yaw * c->n_elev;
    if (ray_counts) FS_HIP(c, c->d_raycounts.ensure((size_t)n * per));
    FsRayArgs a{};
    if (const int rc_args = fill_ray_args(c, a)) return rc_args;
    a.n = n; a.goal = c->in_goal;
    a.frontier_size = c->in_fsize;
    a.blacklisted = c->in_black;
    a.achievable_in = c->in_achin;
    a.ray_counts = ray_counts ? c->d_raycounts.p : nullptr;
    a.arrival = c->d_arrival.p; a.argmax = c->d_argmax.p; a.status = c->d_status.p;
    a.yaw = c->d_yaw.p; a.achievable = c->d_ach.p;
    // a short list: the kernel writes its five result columns straight into the mapped page-locked buffer (five transfers less;
    // the per-ray counts, if asked for, are the one column large enough to keep its transfer)
    {
        const size_t nn0 = (size_t)n;
        const size_t o_arr = 0, o_arg = o_arr + ((4 * nn0 + 15) & ~(size_t)15), o_st = o_arg + ((4 * nn0 + 15) & ~(size_t)15);
        const size_t o_yaw = o_st + ((4 * nn0 + 15) & ~(size_t)15), o_ach = o_yaw + ((8 * nn0 + 15) & ~(size_t)15), o_rc = o_ach + ((nn0 + 15) & ~(size_t)15);
        FS_HIP(c, c->h_out.ensure(o_rc + (ray_counts ? 4 * nn0 * per : 0)));
        if (c->opt_zero_copy && n <= FS_ZERO_COPY_MAX_N && c->h_out.dev) {
            char *hd = c->h_out.dev;
            a.arrival = reinterpret_cast<int32_t *>(hd + o_arr); a.argmax = reinterpret_cast<int32_t *>(hd + o_arg);
            a.status = reinterpret_cast<int32_t *>(hd + o_st); a.yaw = reinterpret_cast<double *>(hd + o_yaw);
            a.achievable = reinterpret_cast<uint8_t *>(hd + o_ach);
            rc = maybe_sort(c, a);
            if (rc) return rc;
            {
                ScopedTimer t(c, 0);
                FS_HIP(c, fs_launch_raymarch(a, c->stream));
            }
            c->arrival_pending.clear();
            if (ray_counts) {
                FS_HIP(c, hipMemcpyAsync(c->h_out.p + o_rc, c->d_raycounts.p, 4 * nn0 * per, hipMemcpyDeviceToHost, c->stream));
                c->arrival_pending.push_back({ray_counts, o_rc, 4 * nn0 * per});
            }
            c->arrival_pending.push_back({arrival, o_arr, 4 * nn0}); c->arrival_pending.push_back({argmax, o_arg, 4 * nn0});
            c->arrival_pending.push_back({status, o_st, 4 * nn0}); c->arrival_pending.push_back({yaw, o_yaw, 8 * nn0});
            c->arrival_pending.push_back({achievable, o_ach, nn0});
            return FS_OK;
        }
    }
    rc = maybe_sort(c, a);
    if (rc) return rc;
    {
        ScopedTimer t(c, 0);
        FS_HIP(c, fs_launch_raymarch(a, c->stream));
    }
    // The columns come back through the context's page-locked buffer: a copy into the caller's pageable arrays (std::vector in
    // the ROS adapter, numpy in the binding) would make every one of these calls host-synchronous — `begin` would return only
    // when this device is done, and fs_multi_score_arrival would run its devices one after the other.
    struct Col { void *host; const void *dev; size_t bytes; };
    const size_t nn = (size_t)n;
    const Col cols[6] = {{ray_counts, c->d_raycounts.p, 4 * nn * per}, {arrival, c->d_arrival.p, 4 * nn}, {argmax, c->d_argmax.p, 4 * nn},
                         {status, c->d_status.p, 4 * nn}, {yaw, c->d_yaw.p, 8 * nn}, {achievable, c->d_ach.p, nn}};
    size_t total = 0;
    for (const Col &col : cols) if (col.host) total += (col.bytes + 15) & ~(size_t)15;
    FS_HIP(c, c->h_out.ensure(total));
    c->arrival_pending.clear();
    size_t off = 0;
    for (const Col &col : cols) {
        if (!col.host) continue;
        FS_HIP(c, hipMemcpyAsync(c->h_out.p + off, col.dev, col.bytes, hipMemcpyDeviceToHost, c->stream));
        c->arrival_pending.push_back({col.host, off, col.bytes});
        off += (col.bytes + 15) & ~(size_t)15;
    }
    return FS_OK;
}

// ... and waited for here: one synchronisation, then the columns go to the caller's arrays
int fs_score_arrival_end(fs_ctx *c)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->arrival_pending.clear(); return fail(c, FS_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e)); }
    for (const fs_ctx::PendingCol &col : c->arrival_pending) std::memcpy(col.host, c->h_out.p + col.off, col.bytes);
    c->arrival_pending.clear();
    return FS_OK;
}

int fs_score_arrival(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                     const uint8_t *blacklisted, const uint8_t *achievable_in,
                     int32_t *ray_counts, int32_t *arrival, int32_t *argmax, double *yaw,
                     uint8_t *achievable, int32_t *status)
{
    const int rc = fs_score_arrival_begin(c, n, goal_xyz, frontier_size, blacklisted, achievable_in, ray_counts, arrival, argmax, yaw, achievable, status);
    if (rc || n == 0) return rc;
    return fs_score_arrival_end(c);
}

int fs_trace_segments(fs_ctx *c, int32_t n, const double *start_xyz, const double *end_xyz, double max_length_cells,
                      int32_t obst_min, int32_t obst_max, int32_t trace_min, int32_t trace_max,
                      uint8_t *ok, int32_t *traced, uint8_t *hit, int32_t *unknown, int32_t *all)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!c->have_grid) return fail(c, FS_E_STATE, "fs_upload_grid has not been called");
    if (n < 0 || (n > 0 && (!start_xyz || !end_xyz || !ok || !traced || !hit || !unknown || !all))) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    DevBuf<double> &d_s = c->d_seg_start, &d_e = c->d_seg_end;
    DevBuf<uint8_t> &d_ok = c->d_seg_ok, &d_hit = c->d_seg_hit;
    DevBuf<int32_t> &d_tr = c->d_seg_traced, &d_un = c->d_seg_unknown, &d_all = c->d_seg_all;
    FS_HIP(c, d_s.ensure((size_t)n * 3)); FS_HIP(c, d_e.ensure((size_t)n * 3));
    FS_HIP(c, d_ok.ensure(n)); FS_HIP(c, d_hit.ensure(n)); FS_HIP(c, d_tr.ensure(n)); FS_HIP(c, d_un.ensure(n)); FS_HIP(c, d_all.ensure(n));
    FS_HIP(c, hipMemcpyAsync(d_s.p, start_xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(d_e.p, end_xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    FsSegArgs a{};
    a.grid = grid_dev(c);
    a.n = n; a.start = d_s.p; a.end = d_e.p; a.max_length = max_length_cells;
    a.obst_min = obst_min; a.obst_max = obst_max; a.trace_min = trace_min; a.trace_max = trace_max;
    a.ok = d_ok.p; a.hit = d_hit.p; a.traced = d_tr.p; a.unknown = d_un.p; a.all = d_all.p;
    FS_HIP(c, fs_launch_segments(a, c->stream));
    FS_HIP(c, hipMemcpyAsync(ok, d_ok.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipMemcpyAsync(hit, d_hit.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipMemcpyAsync(traced, d_tr.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipMemcpyAsync(unknown, d_un.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipMemcpyAsync(all, d_all.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    return FS_OK;
}

// ------------------------------------------------------------------ Fisher information

}  // extern "C"

// Host half of fs_upload_landmarks: the cloud in k-d leaf order as SoA + one bounding sphere per chunk.  A function of the
// input alone — fs_multi runs it once and hands the result to every device.
void fs_stage_landmarks(const float *xyz, int32_t m, FsStagedCloud &out)
{
    // Chunks of 64 consecutive landmarks are the unit of visibility culling, so the cloud is put into the leaf order
    // of a k-d tree with exactly 64 landmarks per leaf: split the longest axis of the bounding box at the multiple of
    // 64 nearest the median, left part first; a remainder always goes right and ends up as the last, short chunk.
    // Against chunks cut from the Morton order (which straddle the curve's jumps) the bounding spheres shrink from
    // 1.78 m to 1.27 m on the C3 cloud and 20 % fewer landmarks survive the culling.  Non-finite points go last.
    std::vector<int32_t> order((size_t)m);
    int32_t n_finite = 0;
    {
        int32_t tail = m;
        for (int32_t i = 0; i < m; ++i) {
            const float *p = xyz + 3 * (size_t)i;
            // (usable: finite and within 1e17 m of the origin — see below; fabsf of a NaN compares false)
            if (std::fabs(p[0]) <= 1.0e17f && std::fabs(p[1]) <= 1.0e17f && std::fabs(p[2]) <= 1.0e17f) order[n_finite++] = i;
            else order[--tail] = i;
        }
        std::reverse(order.begin() + n_finite, order.end());      // keep the non-finite ones in input order
    }
    {
        // [lo, hi) ranges of `order`: split, then the left part, then the right one.  A range is worked on without looking at any
        // other, so the two halves of the top levels go to threads of their own (up to eight leaves of the recursion at once: this
        // ordering is what a new cloud costs per SLAM map update — fs_upload_landmarks 13.2 -> 4.3 ms at C3, 79.9 -> 20.6 ms at C5's
        // 500 k landmarks, tools/landmark_staging_probe.py — and the result is the same
        // permutation whoever computes it).
        const unsigned hw = std::thread::hardware_concurrency();
        int par_depth = hw >= 8 ? 3 : hw >= 4 ? 2 : hw >= 2 ? 1 : 0;
        if (const char *e = std::getenv("FS_KD_THREADS")) {       // measurement aid: 1 = everything on the calling thread
            const int t = std::atoi(e);
            par_depth = t >= 8 ? 3 : t >= 4 ? 2 : t >= 2 ? 1 : 0;
        }
        std::function<void(int32_t, int32_t, int)> build = [&](int32_t lo_i, int32_t hi_i, int depth) {
            for (;;) {
                const int32_t n = hi_i - lo_i;
                if (n <= FS_CHUNK) return;
                float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int32_t i = lo_i; i < hi_i; ++i) {
                    const float *p = xyz + 3 * (size_t)order[i];
                    for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], p[a]); bhi[a] = std::max(bhi[a], p[a]); }
                }
                int ax = 0;
                if (bhi[1] - blo[1] > bhi[ax] - blo[ax]) ax = 1;
                if (bhi[2] - blo[2] > bhi[ax] - blo[ax]) ax = 2;
                int32_t k = (n / 2) / FS_CHUNK * FS_CHUNK;
                if (k == 0) k = FS_CHUNK;
                std::nth_element(order.begin() + lo_i, order.begin() + lo_i + k, order.begin() + hi_i, [&](int32_t u, int32_t v) {
                    const float pu = xyz[3 * (size_t)u + ax], pv = xyz[3 * (size_t)v + ax];
                    return pu < pv || (pu == pv && u < v);        // ties by index: the order is a function of the input
                });
                if (depth < par_depth && n >= 16384) {
                    std::thread left;
                    try {
                        left = std::thread(build, lo_i, lo_i + k, depth + 1);
                    } catch (...) {                               // no thread to be had: the left part runs here
                        build(lo_i, lo_i + k, depth + 1);
                    }
                    build(lo_i + k, hi_i, depth + 1);
                    if (left.joinable()) left.join();
                    return;
                }
                build(lo_i, lo_i + k, depth + 1);
                lo_i += k;                                        // (the right part: this loop's next turn)
                ++depth;
            }
        };
        build(0, n_finite, 0);
    }
    const int32_t n_chunks = std::max<int32_t>(1, (m + FS_CHUNK - 1) / FS_CHUNK);
    const size_t mp = (size_t)n_chunks * FS_CHUNK;
    // SoA + far-away sentinels in the padding: (1e18)^2 is finite in fp32 and beyond any max_dist^2
    std::vector<float> &x = out.x, &y = out.y, &z = out.z, &sph = out.sph;
    x.assign(mp, 1.0e18f); y.assign(mp, 1.0e18f); z.assign(mp, 1.0e18f); sph.assign((size_t)n_chunks * 4, 0.0f);
    // A point with a NaN or an infinite coordinate lies outside every visibility volume (n^2 <= max_dist^2 is false for it), and the
    // kernels must never see it: their cone predicate is ONE v_min3_f32, which DROPS a NaN operand — a landmark at x = +inf seen from
    // a pose whose rotation has exact zeros (px = +inf, py = 0 * inf = NaN, n^2 = NaN) would pass as visible (found in round 5 by
    // tests/test_gpu_hardening.py::test_non_finite_and_far_away_landmarks_are_never_visible).  Such points keep their slots (the
    // cloud's size and order are the caller's) but are staged as the far-away sentinel of the padding.  The same for a FINITE
    // coordinate beyond 1e17 m: rotated into a camera frame it can overflow fp32 to +-inf, and an invisible landmark's terms are
    // removed by a factor q = 0, which inf * 0 = NaN defeats (NaN in the 6x6 sums of every pose with a general rotation).  Nothing
    // that far out is within any range fs_set_fim_params accepts (max_dist < 1e9 m).
    for (int32_t i = 0; i < n_finite; ++i) {
        const float *p = xyz + 3 * (size_t)order[i];
        x[i] = p[0]; y[i] = p[1]; z[i] = p[2];
    }
    for (int32_t ch = 0; ch < n_chunks; ++ch) {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        int cnt = 0;
        for (int k = 0; k < FS_CHUNK; ++k) {
            const size_t i = (size_t)ch * FS_CHUNK + k;
            if (i >= (size_t)n_finite) continue;                  // (padding and the non-finite points: sentinels, in no sphere)
            const double p[3] = {x[i], y[i], z[i]};
            for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], p[a]); bhi[a] = std::max(bhi[a], p[a]); }
            ++cnt;
        }
        float *s4 = &sph[4 * (size_t)ch];
        if (cnt == 0) { s4[0] = s4[1] = s4[2] = 0.0f; s4[3] = -1.0e30f; continue; }   // never accepted, holds nothing visible
        const float ctr[3] = {(float)(0.5 * (blo[0] + bhi[0])), (float)(0.5 * (blo[1] + bhi[1])), (float)(0.5 * (blo[2] + bhi[2]))};
        double r2 = 0.0;
        for (int k = 0; k < FS_CHUNK; ++k) {
            const size_t i = (size_t)ch * FS_CHUNK + k;
            if (i >= (size_t)n_finite) continue;                  // (padding and the non-finite points: sentinels, in no sphere)
            const double dx = (double)x[i] - ctr[0], dy = (double)y[i] - ctr[1], dz = (double)z[i] - ctr[2];
            r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
        }
        s4[0] = ctr[0]; s4[1] = ctr[1]; s4[2] = ctr[2];
        s4[3] = (float)(std::sqrt(r2) * 1.01 + 2.0e-3);          // safety margin: culling must never drop a visible landmark
    }
    out.m = m; out.n_chunks = n_chunks;
}

// Device half: four copies and the HBM-tier tables.
int fs_upload_staged_landmarks(fs_ctx *c, const FsStagedCloud &st)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    const int32_t m = st.m, n_chunks = st.n_chunks;
    const size_t mp = (size_t)n_chunks * FS_CHUNK;
    const std::vector<float> &x = st.x, &y = st.y, &z = st.z, &sph = st.sph;
    FS_HIP(c, c->d_lx.ensure(mp)); FS_HIP(c, c->d_ly.ensure(mp)); FS_HIP(c, c->d_lz.ensure(mp));
    FS_HIP(c, c->d_spheres.ensure(sph.size()));
    FS_HIP(c, hipMemcpyAsync(c->d_lx.p, x.data(), sizeof(float) * mp, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(c->d_ly.p, y.data(), sizeof(float) * mp, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(c->d_lz.p, z.data(), sizeof(float) * mp, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(c->d_spheres.p, sph.data(), sizeof(float) * sph.size(), hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->m = m; c->n_chunks = n_chunks;
    reset_voxel_ratio(c);
    // HBM hash tables for tier 3: one per pool workgroup, 2x the landmark count
    int gb = 12;
    while ((1ll << gb) < 2ll * std::max(m, 1)) ++gb;
    c->ghash_bits = gb;
    FS_HIP(c, c->d_gtable.ensure((size_t)fs_ctx::kPool << gb));
    c->have_lm = true;
    ++c->epoch;
    return FS_OK;
}

#define FS_CLOUD_DEVICE_FROM 4096
bool fs_ctx_cloud_on_device(const fs_ctx *c, int32_t m) { return c && (c->opt_cloud_order == 2 || (c->opt_cloud_order == 1 && m >= FS_CLOUD_DEVICE_FROM)); }

// "cloud.order" 1: the raw cloud goes up as it is and the device puts it into k-d leaf order (fs_cloud.hip) — the host only finds
// out how many landmarks are usable (the level layout depends on that number) and which, if any, are not.
static int upload_landmarks_device_order(fs_ctx *c, const float *xyz, int32_t m)
{
    FS_HIP(c, hipSetDevice(c->device));
    auto ok = [&](int32_t i) {                                    // as fs_stage_landmarks: finite and within 1e17 m (fabsf of a NaN compares false)
        const float *p = xyz + 3 * (size_t)i;
        return std::fabs(p[0]) <= 1.0e17f && std::fabs(p[1]) <= 1.0e17f && std::fabs(p[2]) <= 1.0e17f;
    };
    int32_t n_usable = 0;
    for (int32_t i = 0; i < m; ++i) n_usable += ok(i) ? 1 : 0;
    std::vector<int32_t> usable;                                  // the usable landmarks in input order — only when some are not (rare)
    if (n_usable != m) {
        usable.reserve((size_t)n_usable);
        for (int32_t i = 0; i < m; ++i) if (ok(i)) usable.push_back(i);
    }
    const bool all_usable = n_usable == m;
    c->have_lm = false;                                          // (until the new cloud is whole: an error on the way leaves no cloud, not half of one)
    const int32_t n_chunks = std::max<int32_t>(1, (m + FS_CHUNK - 1) / FS_CHUNK);
    const size_t mp = (size_t)n_chunks * FS_CHUNK;
    std::vector<int32_t> bounds, level_off, level_nodes, level_largest;
    fs_cloud_levels(n_usable, bounds, level_off, level_nodes, level_largest);
    const size_t temp_bytes = fs_cloud_sort_temp_bytes(n_usable, c->stream);
    FS_HIP(c, c->d_lx.ensure(mp)); FS_HIP(c, c->d_ly.ensure(mp)); FS_HIP(c, c->d_lz.ensure(mp));
    FS_HIP(c, c->d_spheres.ensure((size_t)n_chunks * 4));
    FS_HIP(c, c->d_cloud_raw.ensure(3 * (size_t)std::max(m, 1)));
    FS_HIP(c, c->d_cloud_perm.ensure(2 * (size_t)std::max(n_usable, 1)));
    FS_HIP(c, c->d_cloud_keys.ensure(2 * (size_t)std::max(n_usable, 1)));
    FS_HIP(c, c->d_cloud_bounds.ensure(std::max<size_t>(bounds.size(), 1)));
    FS_HIP(c, c->d_cloud_temp.ensure(std::max<size_t>(temp_bytes, 256)));
    FS_HIP(c, c->d_cloud_bbox.ensure(fs_cloud_top_bbox_words()));
    if (m > 0) FS_HIP(c, hipMemcpyAsync(c->d_cloud_raw.p, xyz, sizeof(float) * 3 * (size_t)m, hipMemcpyHostToDevice, c->stream));
    if (!bounds.empty()) FS_HIP(c, hipMemcpyAsync(c->d_cloud_bounds.p, bounds.data(), sizeof(int32_t) * bounds.size(), hipMemcpyHostToDevice, c->stream));
    int32_t *perm_a = c->d_cloud_perm.p, *perm_b = perm_a + std::max(n_usable, 1);
    if (all_usable) FS_HIP(c, fs_cloud_iota(perm_a, n_usable, c->stream));
    else if (n_usable > 0) FS_HIP(c, hipMemcpyAsync(perm_a, usable.data(), sizeof(int32_t) * (size_t)n_usable, hipMemcpyHostToDevice, c->stream));
    int32_t *perm = perm_a;
    FS_HIP(c, fs_cloud_order_device(c->d_cloud_raw.p, n_usable, c->d_cloud_bounds.p, level_off, level_nodes, level_largest, perm_a, perm_b, c->d_cloud_keys.p,
                                    c->d_cloud_keys.p + std::max(n_usable, 1), c->d_cloud_temp.p, temp_bytes, c->d_cloud_bbox.p, c->stream, &perm));
    FS_HIP(c, fs_cloud_finish(c->d_cloud_raw.p, perm, n_usable, n_chunks, c->d_lx.p, c->d_ly.p, c->d_lz.p, c->d_spheres.p, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));                  // (the caller's cloud and the vectors above are read until here)
    if (c->d_cloud_raw.cap > ((size_t)16 << 20)) { c->d_cloud_raw.release(); c->d_cloud_perm.release(); c->d_cloud_keys.release(); c->d_cloud_temp.release(); }
    c->m = m; c->n_chunks = n_chunks;
    reset_voxel_ratio(c);
    int gb = 12;
    while ((1ll << gb) < 2ll * std::max(m, 1)) ++gb;
    c->ghash_bits = gb;
    FS_HIP(c, c->d_gtable.ensure((size_t)fs_ctx::kPool << gb));
    c->have_lm = true;
    ++c->epoch;
    return FS_OK;
}

extern "C" {

int fs_upload_landmarks(fs_ctx *c, const float *xyz, int32_t m)
{
    if (!c || (m > 0 && !xyz) || m < 0) return FS_E_INVALID;
    // chunk masks live in LDS (one bit per chunk) next to the 64-KiB tier-1 table
    if (m > 2000000) return fail(c, FS_E_INVALID, "at most 2,000,000 landmarks per context");
    if (fs_ctx_cloud_on_device(c, m)) return upload_landmarks_device_order(c, xyz, m);
    FsStagedCloud st;
    fs_stage_landmarks(xyz, m, st);
    return fs_upload_staged_landmarks(c, st);
}

int fs_set_option(fs_ctx *c, const char *key, double value)
{
    if (!c || !key) return FS_E_INVALID;
    ++c->epoch;                                            // (whatever the knob is, captured launch sequences are taken again)
    if (std::strcmp(key, "graph") == 0) { c->opt_graph = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "zerocopy") == 0) { c->opt_zero_copy = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "fim.cull") == 0) { c->opt_cull = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "fim.specialise") == 0) { c->opt_special = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "fim.learn") == 0) { c->opt_learn = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "cloud.order") == 0 && value >= 0 && value <= 2) { c->opt_cloud_order = (int)value; return FS_OK; }
    if (std::strcmp(key, "ray.sort") == 0) { c->opt_sort = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "sort.costmap") == 0) { c->opt_costmap = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "sort.reverse") == 0) { c->opt_sort_reverse = value != 0.0; return FS_OK; }
    if (std::strcmp(key, "fim.hostfinish") == 0) { c->opt_host_finish = value != 0; ++c->epoch; return FS_OK; }
    if (std::strcmp(key, "fim.split") == 0 && value >= 0 && value <= 5) { c->opt_split = (int)value; ++c->epoch; return FS_OK; }
    if (std::strcmp(key, "ray.layout") == 0 && value >= 0 && value <= 3) { c->opt_layout = (int)value; return FS_OK; }
    if (std::strcmp(key, "fim.bits1") == 0 && value >= 10 && value <= 14) { c->opt_bits1 = (int)value; return FS_OK; }
    if (std::strcmp(key, "fim.skip32") == 0 && value >= 1 && value <= 32) { c->opt_skip32 = (int)value; return FS_OK; }
    if (std::strcmp(key, "fim.headroom") == 0 && value >= 8 && value <= 64) { c->opt_headroom = (int)value; return FS_OK; }
    return fail(c, FS_E_INVALID, "unknown option %s", key);
}

int fs_get_counter(fs_ctx *c, int which, int64_t *value, int reset)
{
    // host-side figures of the sparse class image ("ray.layout" 3): 1000 bricks of the grid, 1001 bricks its pool holds
    if (c && value && (which == 1000 || which == 1001)) { *value = (int64_t)(which == 1000 ? c->sparse_bricks : c->sparse_pool_bricks); return FS_OK; }
    if (!c || !value || which < 0 || which >= FS_N_COUNTERS) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    *value = 0;
    if (!c->d_counters.p) return FS_OK;
    unsigned long long v = 0;
    FS_HIP(c, hipMemcpyAsync(&v, c->d_counters.p + which, sizeof v, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    *value = (int64_t)v;
    if (reset) {
        FS_HIP(c, hipMemsetAsync(c->d_counters.p + which, 0, sizeof v, c->stream));
        FS_HIP(c, hipStreamSynchronize(c->stream));
    }
    return FS_OK;
}

int fs_lookup_generate(fs_ctx *c, const float bounds[6])
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    // DEP/src/fisher_information/GenerateLookupMain.cpp:9 — double literals narrowed to the float parameters
    const float def[6] = {(float)0.0, (float)21.0, (float)(-8.5 * 1.732), (float)(8.5 * 1.732), (float)(-8.5 * 1.732), (float)(8.5 * 1.732)};
    const float *b = bounds ? bounds : def;
    generate_records(b[0], b[1], b[2], b[3], b[4], b[5], c->records);
    return build_dense(c);
}

int fs_lookup_set_records(fs_ctx *c, const float *records, int64_t n)
{
    if (!c || !records || n <= 0) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    c->records.assign(records, records + 4 * n);
    return build_dense(c);
}

int fs_lookup_load(fs_ctx *c, const char *path)
{
    if (!c || !path) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(c, FS_E_IO, "Cannot load lookup table. Does it exist in the path? (%s)", path);
    std::vector<float> rec;
    float r[4];
    while (std::fread(r, sizeof(float), 4, f) == 4) rec.insert(rec.end(), r, r + 4);   // FisherInfoManager.cpp:247-248
    std::fclose(f);
    if (rec.empty()) return fail(c, FS_E_IO, "lookup table file %s holds no complete record", path);
    c->records.swap(rec);
    return build_dense(c);
}

int fs_lookup_save(fs_ctx *c, const char *path)
{
    if (!c || !path) return FS_E_INVALID;
    if (!c->have_table) return fail(c, FS_E_STATE, "no lookup table to save");
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(c, FS_E_IO, "Error opening file for writing (%s)", path);
    const size_t n = c->records.size();
    const size_t w = std::fwrite(c->records.data(), sizeof(float), n, f);
    std::fclose(f);
    return w == n ? FS_OK : fail(c, FS_E_IO, "short write to %s", path);
}

int fs_lookup_num_records(const fs_ctx *c, int64_t *n)
{
    if (!c || !n) return FS_E_INVALID;
    *n = (int64_t)c->records.size() / 4;
    return FS_OK;
}

int fs_lookup_get_records(const fs_ctx *c, float *records)
{
    if (!c || !records) return FS_E_INVALID;
    std::memcpy(records, c->records.data(), c->records.size() * sizeof(float));
    return FS_OK;
}

int fs_lookup_query(const fs_ctx *c, const float p[3], float *value)
{
    if (!c || !p || !value) return FS_E_INVALID;
    if (!c->have_table) return FS_E_STATE;
    float key[3];
    long j[3];
    voxel_coordinate(p[0], p[1], p[2], key, j);
    const long jx = j[0] - c->jx0, jy = j[1] - c->jy0, jz = j[2] - c->jz0;
    if (jx < 0 || jx >= c->tx || jy < 0 || jy >= c->ty || jz < 0 || jz >= c->tz) *value = std::numeric_limits<float>::quiet_NaN();
    else *value = c->dense[((size_t)jx * c->ty + jy) * c->tz + jz];
    return FS_OK;
}

int fs_set_fim_params(fs_ctx *c, const fs_fim_params *p)
{
    if (!c || !p) return FS_E_INVALID;
    if (!(p->max_dist > 0.0) || !(p->max_angle > 0.0)) return fail(c, FS_E_INVALID, "max_dist and max_angle must be positive");
    // (the padding of the landmark arrays and every unusable landmark sit at 1e18 m: max_dist^2 must stay far below (1e18)^2 in fp32)
    if (!(p->max_dist < 1.0e9)) return fail(c, FS_E_INVALID, "max_dist must be below 1e9 m");
    if (p->max_dist != c->fp.max_dist || p->max_angle != c->fp.max_angle) reset_voxel_ratio(c);
    c->fp = *p;
    ++c->epoch;
    return FS_OK;
}

// A call with few poses leaves most of the chip idle — ONE pose is one of 512 workgroup slots, and the reference's real call is one
// pose per tick (FisherInfoBTPlugin.cpp:24-57): spread each pose over W = 2^shift workgroups by voxel slab (fs_fim.hip, SPLIT), as
// long as all n * W items are resident at once and the cloud is big enough to be worth W workgroups' fixed cost (cull, table
// clear, reduction): at least 128 chunks (8 k landmarks) per workgroup — C1's 2 k landmarks measured 3 us SLOWER split eight ways.
// Same multiset of (voxel value, rank) terms; the partial sums are added in the finish kernel.  Sets a.split_shift / a.split_flags
// and grows the per-item scratch; a.cone_mode / a.table_full / a.info_only must be filled in.
static int maybe_split(fs_ctx *c, FsFimArgs &a, size_t n, bool want_fim21)
{
    a.split_shift = 0; a.split_flags = nullptr;
    if (c->opt_split <= 0 || !fs_fim_can_split(a)) return FS_OK;
    int shift = c->opt_split;
    while (shift > 0 && ((n << shift) > 256 || (c->n_chunks >> shift) < 128)) --shift;
    // (W = 1 — a cloud too small to be worth several workgroups — still goes through the split workers when the call asks for
    // info_ref alone and has few poses: they are what lets fs_score_fim_end do the finish on the host, one launch instead of three)
    if (shift == 0 && !(a.info_only && c->opt_host_finish && n <= 256)) return FS_OK;
    if (c->d_split_flags.cap < n) {
        FS_HIP(c, c->d_split_flags.ensure(n));
        FS_HIP(c, hipMemsetAsync(c->d_split_flags.p, 0, c->d_split_flags.cap * sizeof(uint32_t), c->stream));
    }
    const int rc = ensure_candidate_scratch(c, n << shift, want_fim21);
    if (rc) return rc;
    a.split_shift = shift; a.split_flags = c->d_split_flags.p;
    // The W contiguous slabs of the lattice along the camera's x axis (fs_fim.hip, slab_of): cut so that every slab holds the same
    // share of the visibility volume's cross-section over the stretch where the table and the range overlap (in front of the
    // camera only when there is a cone) — plane j at x = j * step shows a disc of radius^2 = min(max_dist^2 - x^2, (x tan(angle))^2).
    // Uniform landmark density assumed; what lies outside the stretch goes to the open-ended first / last slab.
    {
        const int W = 1 << shift;
        const double step = 1.0 / a.inv_step, R = c->fp.max_dist;
        const int reach = (int)std::ceil(R * a.inv_step) + 1;
        const bool cone = a.cone_mode == 1;
        int lo = std::max(c->jx0, cone ? 0 : -reach), hi = std::min(c->jx0 + c->tx - 1, reach);
        if (hi < lo) hi = lo;
        const double tan2 = cone ? std::pow(std::tan(std::min(c->fp.max_angle, 1.55)), 2) : 0.0;
        std::vector<double> cum((size_t)(hi - lo + 2), 0.0);
        for (int j = lo; j <= hi; ++j) {
            const double x = j * step;
            double r2 = std::max(R * R - x * x, 0.0);
            if (cone) r2 = std::min(r2, x * x * tan2);
            cum[(size_t)(j - lo + 1)] = cum[(size_t)(j - lo)] + r2 + 1e-9;      // (+ eps: strictly increasing, so every slab gets planes while there are any)
        }
        a.split_bound[0] = -(1 << 29);
        a.split_bound[W] = 1 << 29;
        int j = lo;
        for (int w = 1; w < W; ++w) {
            const double want = cum.back() * (double)w / (double)W;
            while (j < hi && cum[(size_t)(j - lo + 1)] < want) ++j;
            a.split_bound[w] = std::max(j, a.split_bound[w - 1] > -(1 << 28) ? a.split_bound[w - 1] : lo);
        }
    }
    return FS_OK;
}

static void bind_fim_outputs(fs_ctx *c, FsFimArgs &a)
{
    a.info_ref = c->d_info.p; a.trace = c->d_trace.p; a.logdet = c->d_logdet.p;
    a.n_visible = c->d_nvis.p; a.n_voxels = c->d_nvox.p; a.overflow = c->d_overflow.p;
    a.sums = c->d_sums.p;
    a.tested = c->d_tested.p;
    a.flagged = c->d_flagged.p;
    if (a.fim21) a.fim21 = c->d_fim21.p;       // (maybe_split may have grown — moved — the column since the caller asked for it)
}

// tier 1 on candidates [lo, lo + count) of the processing order
static int run_fim_tier1(fs_ctx *c, FsFimArgs &a, const int32_t *perm, int32_t lo, int32_t count)
{
    a.cand_perm = perm; a.cand_lo = lo; a.cand_count = count;
    ScopedTimer t(c, 1);
    FS_HIP(c, fs_launch_fim(a, c->stream));
    return FS_OK;
}

static int run_fim_rest(fs_ctx *c, FsFimArgs &a)
{
    {
        // the HBM tier exits at once unless the LDS tier flagged a candidate (device-side work list)
        ScopedTimer t(c, 2);
        FS_HIP(c, fs_launch_fim_overflow(a, fs_ctx::kPool, c->stream));
    }
    // (a launch of its own on purpose: running the finish inside the HBM-tier launch — its last workgroup, found with a sign-off
    // counter — measured 6-8 us SLOWER per small call than the launch it saves: profiles/EXPERIMENTS.md, round 5)
    FS_HIP(c, fs_launch_fim_finish(a, c->stream));
    return FS_OK;
}

// fs_score_fim in two halves (as fs_score_candidates_begin / _end): `begin` stages the poses, launches the kernels and requests
// the columns into the context's page-locked buffer, all asynchronous on the context's stream; `end` waits for that stream and
// copies the columns into the caller's arrays.  fs_multi_score_fim starts every member before it waits for the first.
int fs_score_fim_begin(fs_ctx *c, int32_t n, const double *pose7, float *info_ref, float *fim21,
                       float *trace, float *logdet, int32_t *n_visible, int32_t *n_voxels)
{
    if (!c) return FS_E_INVALID;
    c->fim_pending.clear();
    FS_HIP(c, hipSetDevice(c->device));
    int rc = check_scoring_state(c, false, true);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!pose7 || !info_ref))) return fail(c, FS_E_INVALID, "null pose or output pointer");
    if (n == 0) return FS_OK;
    // One pose per call is the reference's operating point (FisherInfoBTPlugin.cpp:24-57), so the call's fixed cost matters: the
    // pose records go through the context's page-locked buffer (one true DMA instead of a staged pageable copy) and every
    // requested output column comes back through the other one — asynchronous copies, ONE synchronisation.
    const size_t nn = (size_t)n;
    FS_HIP(c, c->h_in.ensure(nn * 12 * sizeof(float)));
    float *Rt = reinterpret_cast<float *>(c->h_in.p);
    for (int32_t i = 0; i < n; ++i) pose_to_rt(pose7 + 7 * (size_t)i, Rt + 12 * (size_t)i);
    FS_HIP(c, c->d_Rt.ensure(nn * 12));
    rc = ensure_candidate_scratch(c, n, fim21 != nullptr);
    if (rc) return rc;
    FsFimArgs a{};
    if (const int rc_args = fill_fim_args(c, a)) return rc_args;
    a.n = n; a.Rt = c->d_Rt.p;
    // chunk culling reasons in world space and needs R orthonormal: a non-unit quaternion (which the
    // reference would feed to Eigen unnormalised) switches this call to the brute-force scan
    for (int32_t i = 0; i < n && a.cull; ++i) {
        const double *q = pose7 + 7 * (size_t)i + 3;
        const double nq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (!(std::fabs(nq - 1.0) <= 1.0e-4)) a.cull = 0;
    }
    a.fim21 = fim21 ? c->d_fim21.p : nullptr;
    // isPoseSafe reads the scalar alone (FIP/src/fisher_information/FisherInfoManager.cpp:83-100): a call that asks for nothing
    // but info_ref (and, at no cost, n_voxels) takes the worker without the 6x6 sums and with the exact table-box cull
    a.info_only = (c->opt_special && !fim21 && !trace && !logdet && !n_visible) ? 1 : 0;
    // (the box cull leaves only chunks that can hold voxels of the table, so a pose shows more distinct voxels per landmark
    // scanned than the 13/32 the general worker caps its pass prediction at — C3, cone off: up to 0.5; an extra pass costs a
    // re-test of the landmarks, an overflow the HBM tier)
    if (a.info_only && a.skip32 < 20) a.skip32 = 20;
    rc = maybe_split(c, a, nn, fim21 != nullptr);
    if (rc) return rc;
    struct Col { void *host; const void *dev; size_t bytes; };
    const Col cols[6] = {{info_ref, c->d_info.p, 4 * nn}, {fim21, c->d_fim21.p, 84 * nn}, {trace, c->d_trace.p, 4 * nn},
                         {logdet, c->d_logdet.p, 4 * nn}, {n_visible, c->d_nvis.p, 4 * nn}, {n_voxels, c->d_nvox.p, 4 * nn}};
    size_t total = 0;
    uint64_t col_mask = 0;
    for (int k = 0; k < 6; ++k) if (cols[k].host) { total += (cols[k].bytes + 15) & ~(size_t)15; col_mask |= 1ull << k; }
    FS_HIP(c, c->h_out.ensure(total));
    // up to FS_ZERO_COPY_MAX_N poses the worker reads the pose records from, and the finish kernel writes the requested columns
    // into, the mapped page-locked buffers: the call is three launches and one synchronisation, no transfers
    const bool in_place = c->opt_zero_copy && n <= FS_ZERO_COPY_MAX_N && c->h_in.dev && c->h_out.dev;
    // ONE launch for the isPoseSafe call.  A split info-only call needs the finish kernel only to add W partial sums per pose, and the
    // HBM-tier launch only if an item ran out of table: both are launches of ~4.6 us of GPU timeline each.  Here the items write
    // their partial sums into mapped page-locked memory, fs_score_fim_end adds them on the host (the same double additions in the
    // same order as the finish kernel: same bits) — and only if an item raised the flag next to them does it launch the HBM tier
    // and the finish kernel after all and wait a second time.  (What the finish kernel also does per call — folding the test
    // counts into the running totals, zeroing per-call counters — is deferred to the next call that runs it: statistics only; the
    // work cursor is untouched by a call whose items all fit the grid, which split calls do by construction.)
    const bool host_finish = c->opt_host_finish && a.info_only && a.split_flags && in_place && !c->timing && !c->opt_graph;
    c->fin_active = false;
    if (host_finish) {
        FS_HIP(c, c->h_fin.ensure(16 + sizeof(double) * 18 * (nn << a.split_shift)));
        if (!c->h_fin.dev) return fail(c, FS_E_HIP, "mapped page-locked memory unavailable");
        std::memset(c->h_fin.p, 0, 16);
    }
    auto enqueue = [&]() -> int {
        if (in_place) a.Rt = reinterpret_cast<const float *>(c->h_in.dev);
        else FS_HIP(c, hipMemcpyAsync(c->d_Rt.p, c->h_in.p, nn * 12 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        bind_fim_outputs(c, a);
        if (in_place) {
            void **slot[6] = {reinterpret_cast<void **>(&a.info_ref), reinterpret_cast<void **>(&a.fim21), reinterpret_cast<void **>(&a.trace),
                              reinterpret_cast<void **>(&a.logdet), reinterpret_cast<void **>(&a.n_visible), reinterpret_cast<void **>(&a.n_voxels)};
            size_t o = 0;
            for (int k = 0; k < 6; ++k) {
                if (!cols[k].host) continue;
                *slot[k] = c->h_out.dev + o;
                o += (cols[k].bytes + 15) & ~(size_t)15;
            }
        }
        if (host_finish) {
            a.sums = reinterpret_cast<double *>(c->h_fin.dev + 16);
            a.host_flag = reinterpret_cast<uint32_t *>(c->h_fin.dev);
        }
        int r = run_fim_tier1(c, a, nullptr, 0, a.n << a.split_shift);      // (split: n * W work items)
        if (r) return r;
        if (host_finish) { c->fin_args = a; c->fin_active = true; return FS_OK; }
        r = run_fim_rest(c, a);
        if (r) return r;
        if (in_place) return FS_OK;
        size_t o = 0;
        for (const Col &col : cols) {
            if (!col.host) continue;
            FS_HIP(c, hipMemcpyAsync(c->h_out.p + o, col.dev, col.bytes, hipMemcpyDeviceToHost, c->stream));
            o += (col.bytes + 15) & ~(size_t)15;
        }
        return FS_OK;
    };
    // ONE pose is isPoseSafe's call: its launch sequence is captured (per number of poses up to 4, requested columns, cull mode)
    if (n <= 4) rc = run_maybe_graphed(c, (2ull << 40) | ((uint64_t)a.split_shift << 32) | ((uint64_t)n << 8) | (col_mask << 1) | (uint64_t)(a.cull ? 1 : 0) | (in_place ? 128u : 0u), enqueue);
    else rc = enqueue();
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    size_t off = 0;
    for (int k = 0; k < 6; ++k) {
        const Col &col = cols[k];
        if (!col.host) continue;
        if (k == 0) c->fin_off_info = off;
        if (k == 5) c->fin_off_nvox = off;
        c->fim_pending.push_back({col.host, off, col.bytes});
        off += (col.bytes + 15) & ~(size_t)15;
    }
    c->fin_want_nvox = n_voxels != nullptr;
    return FS_OK;
}

int fs_score_fim_end(fs_ctx *c)
{
    if (!c) return FS_E_INVALID;
    if (c->fim_pending.empty()) return FS_OK;
    FS_HIP(c, hipSetDevice(c->device));
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->fim_pending.clear(); c->fin_active = false; return fail(c, FS_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e)); }
    if (c->fin_active) {
        c->fin_active = false;
        const FsFimArgs &a = c->fin_args;
        uint32_t flag = 0;
        std::memcpy(&flag, c->h_fin.p, 4);
        if (flag == 0) {
            // the finish kernel's info-only branch (fs_fim.hip), on the host: W partial sums per pose, added in item order
            const double *S = reinterpret_cast<const double *>(c->h_fin.p + 16);
            float *info = reinterpret_cast<float *>(c->h_out.p + c->fin_off_info);
            int32_t *nvox = c->fin_want_nvox ? reinterpret_cast<int32_t *>(c->h_out.p + c->fin_off_nvox) : nullptr;
            const int W = 1 << a.split_shift;
            for (int32_t p = 0; p < a.n; ++p) {
                const double *S0 = S + ((size_t)p << a.split_shift) * 18;
                double s_info = S0[0], s_nvox = S0[17];
                for (int w = 1; w < W; ++w) { s_info += S0[(size_t)w * 18]; s_nvox += S0[(size_t)w * 18 + 17]; }
                info[p] = (float)s_info;
                if (nvox) nvox[p] = (int)(s_nvox + 0.5);
            }
        } else {
            // an item ran out of table: the HBM tier redoes the flagged poses, the finish kernel sorts out which result stands
            const int rc = run_fim_rest(c, c->fin_args);
            const hipError_t e2 = hipStreamSynchronize(c->stream);
            if (rc || e2 != hipSuccess) { c->fim_pending.clear(); return rc ? rc : fail(c, FS_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e2)); }
        }
    }
    for (const fs_ctx::PendingCol &col : c->fim_pending) std::memcpy(col.host, c->h_out.p + col.off, col.bytes);
    c->fim_pending.clear();
    return FS_OK;
}

int fs_score_fim(fs_ctx *c, int32_t n, const double *pose7, float *info_ref, float *fim21,
                 float *trace, float *logdet, int32_t *n_visible, int32_t *n_voxels)
{
    const int rc = fs_score_fim_begin(c, n, pose7, info_ref, fim21, trace, logdet, n_visible, n_voxels);
    if (rc) return rc;
    return fs_score_fim_end(c);
}

int fs_information_frontier_pair(fs_ctx *c, int32_t n, const double *est_pose7, const double *triangle_xy, float *information)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!c->have_lm) return fail(c, FS_E_STATE, "fs_upload_landmarks has not been called");
    if (n < 0 || (n > 0 && (!est_pose7 || !triangle_xy || !information))) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    std::vector<float> Rt((size_t)n * 12);
    for (int32_t i = 0; i < n; ++i) pose_to_rt(est_pose7 + 7 * (size_t)i, &Rt[12 * (size_t)i]);
    DevBuf<double> &d_tri = c->d_tri;
    FS_HIP(c, c->d_Rt.ensure(Rt.size())); FS_HIP(c, d_tri.ensure((size_t)n * 6)); FS_HIP(c, c->d_info.ensure(n));
    FS_HIP(c, hipMemcpyAsync(c->d_Rt.p, Rt.data(), Rt.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(d_tri.p, triangle_xy, sizeof(double) * 6 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    // the landmark arrays are padded with far-away sentinels: those never fall inside a triangle
    FS_HIP(c, fs_launch_frontier_pair(n, c->d_lx.p, c->d_ly.p, c->d_lz.p, c->n_chunks * FS_CHUNK, c->d_Rt.p, d_tri.p, c->d_info.p, c->stream));
    FS_HIP(c, hipMemcpyAsync(information, c->d_info.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    return FS_OK;
}

// ------------------------------------------------------------------ key-frame pose information (row a24)

namespace {

// quatToEuler(...)[2] (util.hpp:77-88): tf2::Matrix3x3(tf2::Quaternion).getRPY, yaw component.  tf2 is third party;
// setRotation / getEulerYPR as published for Humble (gimbal-lock branch returns yaw 0).
double yaw_of_quaternion(const double q[4])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double s = 2.0 / (x * x + y * y + z * z + w * w);
    const double ys = y * s, zs = z * s;
    const double m00 = 1.0 - (y * ys + z * zs), m10 = x * ys + w * zs, m20 = x * zs - w * ys;
    if (std::fabs(m20) >= 1.0) return 0.0;
    const double cp = std::cos(-std::asin(m20));
    return std::atan2(m10 / cp, m00 / cp);
}

// getVerticesOfFrustum2D (util.hpp:101-119)
void frustum_triangle(const double pose7[7], double depth, double hfov, double t[6])
{
    const double yaw = yaw_of_quaternion(pose7 + 3);
    t[0] = pose7[0]; t[1] = pose7[1];
    t[2] = pose7[0] + depth * std::cos(yaw - hfov / 2); t[3] = pose7[1] + depth * std::sin(yaw - hfov / 2);
    t[4] = pose7[0] + depth * std::cos(yaw + hfov / 2); t[5] = pose7[1] + depth * std::sin(yaw + hfov / 2);
}

}  // namespace

int fs_upload_keyframes(fs_ctx *c, int32_t n_kf, const double *kf_pose7, const int32_t *kf_offsets, const float *points_xyz)
{
    if (!c || n_kf < 0 || (n_kf > 0 && (!kf_pose7 || !kf_offsets))) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    const int64_t total = n_kf > 0 ? kf_offsets[n_kf] : 0;
    if (n_kf > 0 && kf_offsets[0] != 0) return fail(c, FS_E_INVALID, "kf_offsets[0] must be 0");
    for (int32_t k = 0; k < n_kf; ++k)
        if (kf_offsets[k + 1] < kf_offsets[k]) return fail(c, FS_E_INVALID, "kf_offsets must be non-decreasing");
    if (total > 0 && !points_xyz) return fail(c, FS_E_INVALID, "null pointer");
    std::vector<float> px((size_t)total), py((size_t)total), pz((size_t)total);
    for (int64_t i = 0; i < total; ++i) { px[i] = points_xyz[3 * i]; py[i] = points_xyz[3 * i + 1]; pz[i] = points_xyz[3 * i + 2]; }
    FS_HIP(c, c->d_kf_off.ensure((size_t)n_kf + 1));
    FS_HIP(c, c->d_kpx.ensure((size_t)total)); FS_HIP(c, c->d_kpy.ensure((size_t)total)); FS_HIP(c, c->d_kpz.ensure((size_t)total));
    const int32_t zero = 0;
    FS_HIP(c, hipMemcpyAsync(c->d_kf_off.p, n_kf > 0 ? kf_offsets : &zero, sizeof(int32_t) * ((size_t)n_kf + 1), hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        FS_HIP(c, hipMemcpyAsync(c->d_kpx.p, px.data(), sizeof(float) * (size_t)total, hipMemcpyHostToDevice, c->stream));
        FS_HIP(c, hipMemcpyAsync(c->d_kpy.p, py.data(), sizeof(float) * (size_t)total, hipMemcpyHostToDevice, c->stream));
        FS_HIP(c, hipMemcpyAsync(c->d_kpz.p, pz.data(), sizeof(float) * (size_t)total, hipMemcpyHostToDevice, c->stream));
    }
    FS_HIP(c, hipStreamSynchronize(c->stream));
    c->kf_pose.assign(kf_pose7, kf_pose7 + 7 * (size_t)n_kf);
    c->n_kf = n_kf; c->n_kf_points = total; c->have_kf = true;
    return FS_OK;
}

int fs_information_for_pose(fs_ctx *c, int32_t n, const double *pose7, const fs_keyframe_params *prm,
                            float *information, int32_t *n_cells, int32_t *n_points)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (!c->have_kf) return fail(c, FS_E_STATE, "fs_upload_keyframes has not been called");
    if (!c->have_grid) return fail(c, FS_E_STATE, "fs_upload_grid has not been called");
    if (n < 0 || !prm || (n > 0 && (!pose7 || !information))) return fail(c, FS_E_INVALID, "null pointer");
    if (!(prm->q_diag > 0.0f)) return fail(c, FS_E_INVALID, "q_diag must be positive");
    if (n == 0) return FS_OK;
    // host side: triangles (libm in double, like the reference) and poses
    std::vector<double> tri((size_t)n * 12);
    std::vector<float> Rt((size_t)n * 12);
    for (int32_t i = 0; i < n; ++i) {
        frustum_triangle(pose7 + 7 * (size_t)i, prm->max_depth, prm->hfov, &tri[12 * (size_t)i]);                          // :851
        frustum_triangle(pose7 + 7 * (size_t)i, prm->max_depth + prm->max_depth_error, prm->hfov, &tri[12 * (size_t)i + 6]);   // :174
        pose_to_rt(pose7 + 7 * (size_t)i, &Rt[12 * (size_t)i]);
    }
    // getVerticesToCheck (util.hpp:134-156) of every key-frame at depth + error
    std::vector<double> chk((size_t)c->n_kf * 12);
    for (int32_t k = 0; k < c->n_kf; ++k) {
        double *o = &chk[12 * (size_t)k];
        frustum_triangle(&c->kf_pose[7 * (size_t)k], prm->max_depth + prm->max_depth_error, prm->hfov, o);
        o[6] = (o[0] + o[2]) / 2;  o[7] = (o[1] + o[3]) / 2;
        o[8] = (o[2] + o[4]) / 2;  o[9] = (o[3] + o[5]) / 2;
        o[10] = (o[4] + o[0]) / 2; o[11] = (o[5] + o[1]) / 2;
    }
    // HBM tables of the fallback pass: every point could open its own cell
    int gbits = 10;
    const uint64_t worst = std::min<uint64_t>((uint64_t)c->n_kf_points, (uint64_t)c->nx * (uint64_t)c->ny);
    while (gbits < 30 && ((uint64_t)1 << gbits) < 2 * worst) ++gbits;
    int pool = 64;
    while (pool > 1 && ((size_t)pool * 3 << gbits) * sizeof(uint32_t) > ((size_t)1 << 30)) pool >>= 1;
    FS_HIP(c, c->d_kf_gtable.ensure((size_t)pool * 3 << gbits));
    FS_HIP(c, c->d_kf_tri.ensure(tri.size())); FS_HIP(c, c->d_Rt.ensure(Rt.size())); FS_HIP(c, c->d_kf_check.ensure(chk.size()));
    FS_HIP(c, c->d_info.ensure(n)); FS_HIP(c, c->d_kf_cells.ensure(n)); FS_HIP(c, c->d_kf_points.ensure(n));
    FS_HIP(c, c->d_kf_flagged.ensure(n)); FS_HIP(c, c->d_kf_counters.ensure(1));
    FS_HIP(c, hipMemcpyAsync(c->d_kf_tri.p, tri.data(), tri.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemcpyAsync(c->d_Rt.p, Rt.data(), Rt.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (!chk.empty()) FS_HIP(c, hipMemcpyAsync(c->d_kf_check.p, chk.data(), chk.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, hipMemsetAsync(c->d_kf_counters.p, 0, sizeof(unsigned long long), c->stream));
    FsKfArgs a{};
    a.n = n; a.tri = c->d_kf_tri.p; a.Rt = c->d_Rt.p;
    a.n_kf = c->n_kf; a.kf_check = c->d_kf_check.p; a.kf_offsets = c->d_kf_off.p;
    a.px = c->d_kpx.p; a.py = c->d_kpy.p; a.pz = c->d_kpz.p;
    a.radius = prm->radius;
    const float q = prm->q_diag;
    a.qinv = (q * q) * (1 / (q * (q * q)));                   // Eigen's 3x3 cofactor inverse of q*I (util.hpp:722)
    a.nx = c->nx; a.ny = c->ny; a.ox = c->origin[0]; a.oy = c->origin[1]; a.res = c->res;
    a.info = c->d_info.p; a.n_cells = c->d_kf_cells.p; a.n_points = c->d_kf_points.p;
    a.flagged = c->d_kf_flagged.p; a.counters = c->d_kf_counters.p;
    a.gtable = c->d_kf_gtable.p; a.gbits = gbits;
    FS_HIP(c, fs_launch_kf_info(a, pool, c->stream));
    FS_HIP(c, hipMemcpyAsync(information, c->d_info.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (n_cells) FS_HIP(c, hipMemcpyAsync(n_cells, c->d_kf_cells.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (n_points) FS_HIP(c, hipMemcpyAsync(n_points, c->d_kf_points.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    return FS_OK;
}

// ------------------------------------------------------------------ fused scoring

int fs_score_candidates_dev(fs_ctx *c, int32_t n, const double *d_goal_xyz, const int32_t *d_frontier_size,
                            const uint8_t *d_blacklisted, const uint8_t *d_achievable_in, fs_record *d_records)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    int rc = check_scoring_state(c, true, true);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_goal_xyz || !d_records))) return fail(c, FS_E_INVALID, "null device pointer");
    if (n == 0) return FS_OK;
    rc = ensure_candidate_scratch(c, n, false);
    if (rc) return rc;
    // What decides the size of the per-item scratch comes FIRST: a short list spreads each pose over several workgroups
    // (maybe_split), which may grow — and so move — every per-candidate column; no pointer into them is taken before that.
    FsFimArgs fa{};
    fill_fim_args(c, fa);
    fa.n = n;
    fa.fim21 = nullptr;
    fa.yaw_only = (c->opt_special && c->yaw_exact) ? 1 : 0;   // the ray-march kernel copies the pose's rotation out of d_yawR
    rc = maybe_split(c, fa, (size_t)n, false);                // a handful of frontiers: each pose over several workgroups
    if (rc) return rc;
    FsRayArgs ra{};
    if (const int rc_args = fill_ray_args(c, ra)) return rc_args;
    ra.n = n; ra.goal = d_goal_xyz;
    ra.frontier_size = d_frontier_size; ra.blacklisted = d_blacklisted; ra.achievable_in = d_achievable_in;
    ra.ray_counts = nullptr;
    ra.arrival = c->d_arrival.p; ra.argmax = c->d_argmax.p; ra.status = c->d_status.p;
    ra.yaw = c->d_yaw.p; ra.achievable = c->d_ach.p;
    FS_HIP(c, c->d_Rt.ensure((size_t)n * 12));
    ra.yawR = c->d_yawR.p; ra.pose12 = c->d_Rt.p;
    rc = maybe_sort(c, ra);
    if (rc) return rc;
    fa.Rt = c->d_Rt.p;
    fa.status = c->d_status.p;
    if (ra.perm && c->opt_costmap) { fa.costmap = c->sort_costmap; fa.cand_key = c->sort_keys; }   // heavy blocks first next time
    bind_fim_outputs(c, fa);
    {
        ScopedTimer t(c, 0);
        FS_HIP(c, fs_launch_raymarch(ra, c->stream));
    }
    // the FIM kernel visits the candidates in the same spatial order: neighbouring poses walk the same landmark chunks
    rc = run_fim_tier1(c, fa, ra.perm, 0, n << fa.split_shift);
    if (rc) return rc;
    // the finish kernel assembles the records (one launch less than a separate pack)
    fa.records = d_records;
    fa.rec_arrival = c->d_arrival.p; fa.rec_argmax = c->d_argmax.p; fa.rec_yaw = c->d_yaw.p; fa.rec_achievable = c->d_ach.p;
    rc = run_fim_rest(c, fa);
    if (rc) return rc;
    return FS_OK;
}

// fs_score_candidates in two halves, so that one host thread can keep several devices busy (fs_multi.hip): `begin` stages
// the candidate columns, launches the kernels and requests the records into the context's page-locked buffer — everything
// asynchronous on the context's stream; `end` waits for that stream and hands the records over.
int fs_score_candidates_begin(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                              const uint8_t *blacklisted, const uint8_t *achievable_in)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (n < 0 || (n > 0 && !goal_xyz)) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    int rc = check_scoring_state(c, true, true);
    if (rc) return rc;
    rc = upload_candidates(c, n, goal_xyz, frontier_size, blacklisted, achievable_in);
    if (rc) return rc;
    FS_HIP(c, c->d_records.ensure(n));
    FS_HIP(c, c->h_out.ensure(sizeof(fs_record) * (size_t)n));
    // a short list: the finish kernel writes the records straight into the mapped page-locked buffer
    const bool in_place = c->opt_zero_copy && n <= FS_ZERO_COPY_MAX_N && c->h_out.dev;
    rc = fs_score_candidates_dev(c, n, c->in_goal, c->in_fsize, c->in_black, c->in_achin,
                                 in_place ? reinterpret_cast<fs_record *>(c->h_out.dev) : c->d_records.p);
    if (rc) return rc;
    if (!in_place) FS_HIP(c, hipMemcpyAsync(c->h_out.p, c->d_records.p, sizeof(fs_record) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    return FS_OK;
}

int fs_score_candidates_end(fs_ctx *c, int32_t n, fs_record *records)
{
    if (!c) return FS_E_INVALID;
    if (n < 0 || (n > 0 && !records)) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    FS_HIP(c, hipSetDevice(c->device));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(records, c->h_out.p, sizeof(fs_record) * (size_t)n);
    return FS_OK;
}

}  // extern "C"

namespace {

// arrival information only, device columns in, arrival-only records out (the Fisher columns of a record stay zero)
int arrival_records_dev(fs_ctx *c, int32_t n, const double *d_goal, const int32_t *d_fsize, const uint8_t *d_black, const uint8_t *d_achin,
                        fs_record *d_records)
{
    int rc = check_scoring_state(c, true, false);
    if (rc) return rc;
    rc = ensure_candidate_scratch(c, n, false);
    if (rc) return rc;
    FsRayArgs a{};
    if (const int rc_args = fill_ray_args(c, a)) return rc_args;
    a.n = n; a.goal = d_goal; a.frontier_size = d_fsize; a.blacklisted = d_black; a.achievable_in = d_achin;
    a.arrival = c->d_arrival.p; a.argmax = c->d_argmax.p; a.status = c->d_status.p; a.yaw = c->d_yaw.p; a.achievable = c->d_ach.p;
    a.records = d_records;
    rc = maybe_sort(c, a);
    if (rc) return rc;
    ScopedTimer t(c, 0);
    FS_HIP(c, fs_launch_raymarch(a, c->stream));
    return FS_OK;
}

// The whole of a host-buffer scoring call — candidate columns (and, for ranking, the planner's path columns) in, records (and
// costs, utilities, order) out — as ONE transfer in, the kernels, ONE transfer out and one synchronisation; up to FS_GRAPH_MAX_N
// candidates the device-side sequence is a captured launch graph per power-of-two bucket (run_maybe_graphed): the list is
// padded to the bucket with blacklisted dummies, which no kernel spends work on and which a stable ascending sort leaves behind
// every real candidate.
int frontier_costs_core(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                        const uint8_t *achievable_in, const double *path_length, const double *path_heading,
                        double alpha, double beta, double max_vx, double max_wz, bool with_fim,
                        fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order)
{
    FS_HIP(c, hipSetDevice(c->device));
    int rc = check_scoring_state(c, true, with_fim);
    if (rc) return rc;
    const bool rank = path_length != nullptr;
    const bool graphed = c->opt_graph && !c->timing && n <= FS_GRAPH_MAX_N;
    const int32_t cap = graphed ? graph_bucket(n) : n;
    const size_t nn = (size_t)n, cc = (size_t)cap, pad = (cc + 15) & ~(size_t)15;
    // input block: goal | path length | path heading | frontier size | blacklist | achievable
    const size_t i_goal = 0, i_len = i_goal + 24 * cc, i_head = i_len + (rank ? 8 * cc : 0), i_fsize = i_head + (rank ? 8 * cc : 0);
    const size_t i_black = (i_fsize + 4 * cc + 15) & ~(size_t)15, i_achin = i_black + pad, total_in = i_achin + pad;
    // output block: records | cost | arrival utility | distance utility | order | range-error flag
    const size_t o_rec = 0, o_cost = o_rec + sizeof(fs_record) * cc, o_au = o_cost + (rank ? 8 * cc : 0), o_du = o_au + (rank ? 8 * cc : 0);
    const size_t o_order = o_du + (rank ? 8 * cc : 0), o_err = (o_order + (rank ? 4 * cc : 0) + 15) & ~(size_t)15, total_out = o_err + 16;
    FS_HIP(c, c->h_in.ensure(total_in)); FS_HIP(c, c->d_in.ensure(total_in));
    FS_HIP(c, c->h_out.ensure(total_out)); FS_HIP(c, c->d_out.ensure(total_out));
    char *h = c->h_in.p;
    std::memcpy(h + i_goal, goal_xyz, 24 * nn);
    if (rank) { std::memcpy(h + i_len, path_length, 8 * nn); std::memcpy(h + i_head, path_heading, 8 * nn); }
    if (frontier_size) std::memcpy(h + i_fsize, frontier_size, 4 * nn); else std::memset(h + i_fsize, 0, 4 * nn);
    if (blacklisted) std::memcpy(h + i_black, blacklisted, nn); else std::memset(h + i_black, 0, nn);
    if (achievable_in) std::memcpy(h + i_achin, achievable_in, nn); else std::memset(h + i_achin, 1, nn);
    if (cap > n) {                                              // the dummies: blacklisted, at the origin, nothing else
        std::memset(h + i_goal + 24 * nn, 0, 24 * (cc - nn));
        if (rank) { std::memset(h + i_len + 8 * nn, 0, 8 * (cc - nn)); std::memset(h + i_head + 8 * nn, 0, 8 * (cc - nn)); }
        std::memset(h + i_fsize + 4 * nn, 0, 4 * (cc - nn));
        std::memset(h + i_black + nn, 1, cc - nn);
        std::memset(h + i_achin + nn, 1, cc - nn);
    }
    // a short list is read and its results are written where they lie, in the mapped page-locked buffers: no transfers at all
    const bool in_place = c->opt_zero_copy && cap <= FS_ZERO_COPY_MAX_N && c->h_in.dev && c->h_out.dev;
    char *din = in_place ? c->h_in.dev : c->d_in.p, *dout = in_place ? c->h_out.dev : c->d_out.p;
    auto enqueue = [&]() -> int {
        if (!in_place) FS_HIP(c, hipMemcpyAsync(din, c->h_in.p, total_in, hipMemcpyHostToDevice, c->stream));
        const double *d_goal = reinterpret_cast<const double *>(din + i_goal);
        const int32_t *d_fsize = reinterpret_cast<const int32_t *>(din + i_fsize);
        const uint8_t *d_black = reinterpret_cast<const uint8_t *>(din + i_black), *d_achin = reinterpret_cast<const uint8_t *>(din + i_achin);
        fs_record *d_rec = reinterpret_cast<fs_record *>(dout + o_rec);
        int r = with_fim ? fs_score_candidates_dev(c, cap, d_goal, d_fsize, d_black, d_achin, d_rec)
                         : arrival_records_dev(c, cap, d_goal, d_fsize, d_black, d_achin, d_rec);
        if (r) return r;
        if (rank) {
            r = fs_rank_candidates_dev(c, cap, d_rec, d_black, reinterpret_cast<const double *>(din + i_len), reinterpret_cast<const double *>(din + i_head),
                                       alpha, beta, max_vx, max_wz, reinterpret_cast<double *>(dout + o_cost), reinterpret_cast<double *>(dout + o_au),
                                       reinterpret_cast<double *>(dout + o_du), reinterpret_cast<int32_t *>(dout + o_order), reinterpret_cast<int32_t *>(dout + o_err));
            if (r) return r;
        }
        if (!in_place) FS_HIP(c, hipMemcpyAsync(c->h_out.p, dout, rank ? total_out : sizeof(fs_record) * cc, hipMemcpyDeviceToHost, c->stream));
        return FS_OK;
    };
    if (graphed) {
        // what a captured sequence has baked in besides the pointers: the bucket, which kernels run, the ranking's parameters
        uint64_t key = ((uint64_t)cap << 8) | (with_fim ? 1u : 0u) | (rank ? 2u : 0u) | (in_place ? 4u : 0u) | (1ull << 40);
        if (rank) {
            const double prm[4] = {alpha, beta, max_vx, max_wz};
            uint64_t hsh = 1469598103934665603ull;
            for (size_t k = 0; k < sizeof prm; ++k) hsh = (hsh ^ reinterpret_cast<const unsigned char *>(prm)[k]) * 1099511628211ull;
            key ^= hsh << 41;
        }
        rc = run_maybe_graphed(c, key, enqueue);
    } else {
        rc = enqueue();
    }
    const hipError_t e_sync = hipStreamSynchronize(c->stream);      // (also after a failure: what was queued out of h_in has landed)
    if (rc) return rc;
    if (e_sync != hipSuccess) return fail(c, FS_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e_sync));
    const char *ho = c->h_out.p;
    std::memcpy(records, ho + o_rec, sizeof(fs_record) * nn);
    if (rank) {
        std::memcpy(weighted_cost, ho + o_cost, 8 * nn);
        if (arrival_utility) std::memcpy(arrival_utility, ho + o_au, 8 * nn);
        if (distance_utility) std::memcpy(distance_utility, ho + o_du, 8 * nn);
        if (order) std::memcpy(order, ho + o_order, 4 * nn);       // (stable sort: the dummies come after every real candidate)
        int32_t err = 0;
        std::memcpy(&err, ho + o_err, 4);
        if (err) return fail(c, FS_E_RANGE, "utility outside [0,1] (the reference throws: FrontierCostsManager.cpp:148-149,173-174)");
    }
    return FS_OK;
}

}  // namespace

// ------------------------------------------------------------------ pieces of fs_multi_get_frontier_costs (fs_multi.hip)
// One process, several GPUs, ONE call: every member scores its block of the frontier list on its own device and stream, the
// blocks' records are moved device to device into ONE list on member 0's device (fs_multi.hip: peer copies over xGMI, each on
// the member's stream behind its kernels, an event per member that member 0's stream waits for), fs_rank_candidates_dev runs
// there on the whole list, and what the caller asked for comes back in ONE transfer.  Nothing below synchronises except _end.

hipStream_t fs_ctx_stream(fs_ctx *c) { return c->stream; }
int fs_ctx_device(const fs_ctx *c) { return c->device; }

namespace {
struct GatherLayout {
    size_t i_len, i_head, i_black, total_in;
    size_t o_rec, o_cost, o_au, o_du, o_order, o_err, total_out;
    explicit GatherLayout(size_t n)
    {
        i_len = 0; i_head = 8 * n; i_black = 16 * n; total_in = i_black + ((n + 15) & ~(size_t)15);
        o_rec = 0; o_cost = sizeof(fs_record) * n; o_au = o_cost + 8 * n; o_du = o_au + 8 * n; o_order = o_du + 8 * n;
        o_err = (o_order + 4 * n + 15) & ~(size_t)15; total_out = o_err + 16;
    }
};
}  // namespace

// member 0, first: the whole list's path columns and blacklist go to its device (one transfer, on its stream); *d_list is where
// the n records of the gathered list will live
int fs_gather_begin(fs_ctx *c, int32_t n, const uint8_t *blacklisted, const double *path_length, const double *path_heading, fs_record **d_list)
{
    if (!c || n <= 0 || !path_length || !path_heading || !d_list) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    const size_t nn = (size_t)n;
    const GatherLayout L(nn);
    FS_HIP(c, c->h_gin.ensure(L.total_in)); FS_HIP(c, c->d_gin.ensure(L.total_in));
    FS_HIP(c, c->h_out.ensure(L.total_out)); FS_HIP(c, c->d_out.ensure(L.total_out));
    std::memcpy(c->h_gin.p + L.i_len, path_length, 8 * nn);
    std::memcpy(c->h_gin.p + L.i_head, path_heading, 8 * nn);
    if (blacklisted) std::memcpy(c->h_gin.p + L.i_black, blacklisted, nn); else std::memset(c->h_gin.p + L.i_black, 0, nn);
    FS_HIP(c, hipMemcpyAsync(c->d_gin.p, c->h_gin.p, L.total_in, hipMemcpyHostToDevice, c->stream));
    *d_list = reinterpret_cast<fs_record *>(c->d_out.p + L.o_rec);
    return FS_OK;
}

// any member: stage the block's candidate columns and score them — into d_dst when the member may write the gathered list
// itself (member 0; a member on member 0's device), else into its own record buffer; *d_block says where the records are
int fs_block_score_begin(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                         const uint8_t *achievable_in, bool with_fim, fs_record *d_dst, fs_record **d_block)
{
    if (!c || n <= 0 || !goal_xyz || !d_block) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    int rc = check_scoring_state(c, true, with_fim);
    if (rc) return rc;
    rc = upload_candidates(c, n, goal_xyz, frontier_size, blacklisted, achievable_in);
    if (rc) return rc;
    if (!d_dst) { FS_HIP(c, c->d_records.ensure(n)); d_dst = c->d_records.p; }
    rc = with_fim ? fs_score_candidates_dev(c, n, c->in_goal, c->in_fsize, c->in_black, c->in_achin, d_dst)
                  : arrival_records_dev(c, n, c->in_goal, c->in_fsize, c->in_black, c->in_achin, d_dst);
    if (rc) return rc;
    *d_block = d_dst;
    return FS_OK;
}

// the fallback without peer access: a block's records to the member's page-locked buffer (member 0's stream copies them on from there)
int fs_block_records_to_host(fs_ctx *c, int32_t n, const fs_record *d_block, const fs_record **h_block)
{
    if (!c || n <= 0 || !d_block || !h_block) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    FS_HIP(c, c->h_out.ensure(sizeof(fs_record) * (size_t)n));
    FS_HIP(c, hipMemcpyAsync(c->h_out.p, d_block, sizeof(fs_record) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    *h_block = reinterpret_cast<const fs_record *>(c->h_out.p);
    return FS_OK;
}

// member 0, once its stream waits for every block: U1 costs and order over the whole list, then the ONE transfer out
int fs_gather_rank(fs_ctx *c, int32_t n, double alpha, double beta, double max_vx, double max_wz)
{
    if (!c || n <= 0) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    const GatherLayout L((size_t)n);
    char *in = c->d_gin.p, *out = c->d_out.p;
    const int rc = fs_rank_candidates_dev(c, n, reinterpret_cast<const fs_record *>(out + L.o_rec), reinterpret_cast<const uint8_t *>(in + L.i_black),
                                          reinterpret_cast<const double *>(in + L.i_len), reinterpret_cast<const double *>(in + L.i_head),
                                          alpha, beta, max_vx, max_wz, reinterpret_cast<double *>(out + L.o_cost), reinterpret_cast<double *>(out + L.o_au),
                                          reinterpret_cast<double *>(out + L.o_du), reinterpret_cast<int32_t *>(out + L.o_order), reinterpret_cast<int32_t *>(out + L.o_err));
    if (rc) return rc;
    FS_HIP(c, hipMemcpyAsync(c->h_out.p, out, L.total_out, hipMemcpyDeviceToHost, c->stream));
    return FS_OK;
}

int fs_gather_end(fs_ctx *c, int32_t n, fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order)
{
    if (!c || n <= 0) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    const size_t nn = (size_t)n;
    const GatherLayout L(nn);
    const char *ho = c->h_out.p;
    if (records) std::memcpy(records, ho + L.o_rec, sizeof(fs_record) * nn);
    if (weighted_cost) std::memcpy(weighted_cost, ho + L.o_cost, 8 * nn);
    if (arrival_utility) std::memcpy(arrival_utility, ho + L.o_au, 8 * nn);
    if (distance_utility) std::memcpy(distance_utility, ho + L.o_du, 8 * nn);
    if (order) std::memcpy(order, ho + L.o_order, 4 * nn);
    int32_t err = 0;
    std::memcpy(&err, ho + L.o_err, 4);
    if (err) return fail(c, FS_E_RANGE, "utility outside [0,1] (the reference throws: FrontierCostsManager.cpp:148-149,173-174)");
    return FS_OK;
}

extern "C" {

int fs_score_candidates(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                        const uint8_t *blacklisted, const uint8_t *achievable_in, fs_record *records)
{
    if (!c) return FS_E_INVALID;
    if (n < 0 || (n > 0 && (!goal_xyz || !records))) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    // up to FS_GRAPH_MAX_N candidates — the reference's operating point — the call is one captured launch graph
    if (c->opt_graph && !c->timing && n <= FS_GRAPH_MAX_N)
        return frontier_costs_core(c, n, goal_xyz, frontier_size, blacklisted, achievable_in, nullptr, nullptr, 0, 0, 0, 0, true,
                                   records, nullptr, nullptr, nullptr, nullptr);
    const int rc = fs_score_candidates_begin(c, n, goal_xyz, frontier_size, blacklisted, achievable_in);
    if (rc) return rc;
    return fs_score_candidates_end(c, n, records);
}

int fs_get_frontier_costs(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                          const uint8_t *achievable_in, const double *path_length, const double *path_heading,
                          double alpha, double beta, double max_vx, double max_wz, int with_fisher_information,
                          fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order)
{
    if (!c) return FS_E_INVALID;
    if (n < 0 || (n > 0 && (!goal_xyz || !path_length || !path_heading || !records || !weighted_cost))) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    return frontier_costs_core(c, n, goal_xyz, frontier_size, blacklisted, achievable_in, path_length, path_heading, alpha, beta, max_vx, max_wz,
                               with_fisher_information != 0, records, weighted_cost, arrival_utility, distance_utility, order);
}

// ------------------------------------------------------------------ utility + ranking

int fs_rank_candidates_dev(fs_ctx *c, int32_t n, const fs_record *d_records, const uint8_t *d_blacklisted,
                           const double *d_path_length, const double *d_path_heading,
                           double alpha, double beta, double max_vx, double max_wz,
                           double *d_weighted_cost, double *d_arrival_utility, double *d_distance_utility,
                           int32_t *d_order, int32_t *d_range_error)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!d_records || !d_path_length || !d_path_heading || !d_weighted_cost))) return fail(c, FS_E_INVALID, "null device pointer");
    if (n == 0) return FS_OK;
    // columns the caller does not want still have to be written somewhere: the context's own scratch
    if (!d_arrival_utility) { FS_HIP(c, c->d_au.ensure(n)); d_arrival_utility = c->d_au.p; }
    if (!d_distance_utility) { FS_HIP(c, c->d_du.ensure(n)); d_distance_utility = c->d_du.p; }
    if (!d_order) { FS_HIP(c, c->d_order.ensure(n)); d_order = c->d_order.p; }
    if (!d_range_error) { FS_HIP(c, c->d_err.ensure(1)); d_range_error = c->d_err.p; }
    ScopedTimer t(c, 3);
    FS_HIP(c, fs_launch_rank(n, d_records, d_blacklisted, d_path_length, d_path_heading, alpha, beta, max_vx, max_wz, c->max_gt,
                             d_weighted_cost, d_arrival_utility, d_distance_utility, d_order, d_range_error,
                             &c->rank_scratch, &c->rank_scratch_bytes, c->stream));
    return FS_OK;
}

// The host-buffer form: one packed transfer in through the context's page-locked buffer, fs_rank_candidates_dev on the
// context's own device columns, the requested columns back through the other page-locked buffer, ONE synchronisation.
int fs_rank_candidates(fs_ctx *c, int32_t n, const fs_record *records, const uint8_t *blacklisted,
                       const double *path_length, const double *path_heading,
                       double alpha, double beta, double max_vx, double max_wz,
                       double *weighted_cost, double *arrival_utility, double *distance_utility,
                       int32_t *order)
{
    if (!c) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    if (n < 0 || (n > 0 && (!records || !path_length || !path_heading || !weighted_cost))) return fail(c, FS_E_INVALID, "null pointer");
    if (n == 0) return FS_OK;
    const size_t nn = (size_t)n, pad = (nn + 15) & ~(size_t)15;
    const size_t o_rec = 0, o_len = o_rec + sizeof(fs_record) * nn, o_head = o_len + 8 * nn, o_black = o_head + 8 * nn;
    const size_t total_in = o_black + pad;
    FS_HIP(c, c->h_in.ensure(total_in));
    FS_HIP(c, c->d_in.ensure(total_in));
    std::memcpy(c->h_in.p + o_rec, records, sizeof(fs_record) * nn);
    std::memcpy(c->h_in.p + o_len, path_length, 8 * nn);
    std::memcpy(c->h_in.p + o_head, path_heading, 8 * nn);
    if (blacklisted) std::memcpy(c->h_in.p + o_black, blacklisted, nn);
    FS_HIP(c, hipMemcpyAsync(c->d_in.p, c->h_in.p, blacklisted ? total_in : o_black, hipMemcpyHostToDevice, c->stream));
    FS_HIP(c, c->d_cost.ensure(n)); FS_HIP(c, c->d_au.ensure(n)); FS_HIP(c, c->d_du.ensure(n));
    FS_HIP(c, c->d_order.ensure(n)); FS_HIP(c, c->d_err.ensure(1));
    const int rc = fs_rank_candidates_dev(c, n, reinterpret_cast<const fs_record *>(c->d_in.p + o_rec),
                                          blacklisted ? reinterpret_cast<const uint8_t *>(c->d_in.p + o_black) : nullptr,
                                          reinterpret_cast<const double *>(c->d_in.p + o_len), reinterpret_cast<const double *>(c->d_in.p + o_head),
                                          alpha, beta, max_vx, max_wz, c->d_cost.p, c->d_au.p, c->d_du.p, c->d_order.p, c->d_err.p);
    if (rc) return rc;
    struct Col { void *host; const void *dev; size_t bytes; };
    const Col cols[5] = {{weighted_cost, c->d_cost.p, 8 * nn}, {arrival_utility, c->d_au.p, 8 * nn}, {distance_utility, c->d_du.p, 8 * nn},
                         {order, c->d_order.p, 4 * nn}, {nullptr, c->d_err.p, 4}};
    size_t total = 16;
    for (const Col &col : cols) if (col.host) total += (col.bytes + 15) & ~(size_t)15;
    FS_HIP(c, c->h_out.ensure(total));
    FS_HIP(c, hipMemcpyAsync(c->h_out.p, c->d_err.p, 4, hipMemcpyDeviceToHost, c->stream));
    size_t off = 16;
    for (const Col &col : cols) {
        if (!col.host) continue;
        FS_HIP(c, hipMemcpyAsync(c->h_out.p + off, col.dev, col.bytes, hipMemcpyDeviceToHost, c->stream));
        off += (col.bytes + 15) & ~(size_t)15;
    }
    FS_HIP(c, hipStreamSynchronize(c->stream));
    off = 16;
    for (const Col &col : cols) {
        if (!col.host) continue;
        std::memcpy(col.host, c->h_out.p + off, col.bytes);
        off += (col.bytes + 15) & ~(size_t)15;
    }
    int32_t err = 0;
    std::memcpy(&err, c->h_out.p, 4);
    if (err) return fail(c, FS_E_RANGE, "utility outside [0,1] (the reference throws: FrontierCostsManager.cpp:148-149,173-174)");
    return FS_OK;
}

// ------------------------------------------------------------------ self test

int fs_selftest_fp64(fs_ctx *c, int32_t max_abs, int64_t *mismatches)
{
    if (!c || !mismatches || max_abs < 1 || max_abs > 2048) return FS_E_INVALID;
    FS_HIP(c, hipSetDevice(c->device));
    const int side = max_abs + 1;
    const size_t total = (size_t)side * side;
    DevBuf<double> ds, dd;
    FS_HIP(c, ds.ensure(total)); FS_HIP(c, dd.ensure(total));
    FS_HIP(c, fs_launch_selftest(max_abs, ds.p, dd.p, c->stream));
    std::vector<double> hs(total), hd(total);
    FS_HIP(c, hipMemcpyAsync(hs.data(), ds.p, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipMemcpyAsync(hd.data(), dd.p, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FS_HIP(c, hipStreamSynchronize(c->stream));
    ds.release(); dd.release();
    int64_t bad = 0;
    for (int dx = 0; dx < side; ++dx)
        for (int dy = 0; dy < side; ++dy) {
            const size_t t = (size_t)dx * side + dy;
            // The reference calls std::hypot (Helpers.cpp:49); glibc's hypot is within 1 ulp but not
            // always correctly rounded, so the device (and this check) use the correctly rounded sqrt of
            // the exact integer dx^2+dy^2.  A 1-ulp change of dist cannot flip (unsigned)(scale*abs_da):
            // tests/test_oracle_raycast.py::test_hypot_vs_sqrt_never_changes_step_count proves it exhaustively.
            const double dist = std::sqrt((double)((long long)dx * dx + (long long)dy * dy));
            const double q = (dist == 0.0) ? 1.0 : 40.0 / dist;
            if (std::memcmp(&dist, &hs[t], 8) != 0) ++bad;
            if (std::memcmp(&q, &hd[t], 8) != 0) ++bad;
        }
    *mismatches = bad;
    return FS_OK;
}

}  // extern "C"
