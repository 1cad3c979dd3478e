// fs_internal.h — structures shared by the HIP kernels and the C-ABI host layer (not installed).
#ifndef FS_INTERNAL_H_
#define FS_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <vector>

#include "../../include/fitslam_frontier.h"
#include "../../include/fitslam_frontier_dev.h"

// Instrumentation paths (cycle stamps, schedule recorder, range checks) exist in FS_DEV builds only — the
// builds fit-slam_amd/_build.py makes under a library name of their own when one of its FS_* knobs is set.
#ifndef FS_DEV
#undef FS_FIM_STAMPS
#undef FS_FIM_STAMPS_PER_WAVE
#undef FS_FIM_SCHEDULE
#undef FS_FIM_BOUNDS
#undef FS_RAY_BOUNDS
#endif

// ---- ray-march kernel arguments ---------------------------------------------------------------
// The occupancy grid lives in HBM as the dense row-major image [nz][ny][nx] the uploads write — what short rays, the
// footprint disc, the segment tracer and the frontier-cell stencil read — and, for long rays, as the CLASS image: what the
// arrival visitor needs of a cell is two bits — in the trace range? in the obstacle range? (Helpers.hpp:64-71) — so the grid is
// classified once per (map, visitor ranges) and packed 16 cells to a dword in bricks of 8 x 8 x 8 cells = 128 B = one cache
// line: cell (x, y, z) is 2-bit field A & 15 of dword A >> 4,
//     A = x + (x >> 3) * cls_m[0] + (y << 3) + (y >> 3) * cls_m[1] + (z << 6) + (z >> 3) * cls_m[2]
// (brick-linear order [z >> 3][y >> 3][x >> 3], cell order [z & 7][y & 7][x & 7]; 512^3 -> 32 MiB).
struct FsGridDev {
    const uint8_t *cells;      // [nz][ny][nx]
    int32_t nx, ny, nz;
    double ox, oy, oz;         // origin
    double res;
    unsigned long long *dbg;   // range-checked builds (FS_BOUNDS=1): where a walk that left the grid is recorded
    const uint32_t *cls;       // class image, or nullptr while it has not been cut
    uint32_t cls_m[3];         // 512 - 8, 512 * bricks_x - 64, 512 * bricks_x * bricks_y - 512  (brick stride of the axis minus 8 << l)
    uint32_t cls_cells;        // cells the image holds (padded to whole bricks)
    // SPARSE form of the class image ("ray.layout" 3; BASELINE.json configs[4] "sparse-hashed voxel grid"): `cls` is then a POOL of
    // 128-B bricks and cls_table[brick-linear index] the slot of the brick's content in it — slots 0..3 are the four uniform
    // bricks (every cell of one class: all-unknown, all-free, ...), shared by every brick of that content.  nullptr: dense.
    const uint32_t *cls_table;
};

// dwords of the class image of an nx x ny x nz grid, and the kernel that fills it (bit 0: cost in [trace_min, trace_max], bit 1: in [obst_min, obst_max])
size_t fs_class_image_words(int nx, int ny, int nz);
hipError_t fs_launch_classify(const uint8_t *d_cells, uint32_t *d_cls, int nx, int ny, int nz, int obst_min, int obst_max,
                              int trace_min, int trace_max, hipStream_t s);
// ... only the n_bricks[3] bricks from brick brick0[3] on (what a rewritten window of the map touches), and the kernel that puts
// a packed window [sz][sy][sx] into the row-major image at (x0, y0, z0): fs_update_grid_region
hipError_t fs_launch_classify_region(const uint8_t *d_cells, uint32_t *d_cls, int nx, int ny, int nz, int obst_min, int obst_max,
                                     int trace_min, int trace_max, const int brick0[3], const int n_bricks[3], hipStream_t s);
hipError_t fs_launch_window_scatter(const uint8_t *d_window, uint8_t *d_grid, int nx, int ny, int x0, int y0, int z0,
                                    int sx, int sy, int sz, hipStream_t s);

struct FsRayArgs {
    FsGridDev grid;
    // fan geometry, precomputed on the host in double with libm (ray directions are
    // candidate-independent; device cos/sin could differ from libm in the last ulp and flip a
    // truncation at a cell boundary):  dir[(e*n_yaw + i)*3 + {0,1,2}] = D*cos(phi_e)*cos(theta_i), ... , D*sin(phi_e)
    const double *dir;
    int32_t n_yaw, n_elev, window;
    uint32_t max_length;       // (unsigned)(max_camera_depth / resolution), CostCalculator.cpp:28
    int32_t obst_min, obst_max, trace_min, trace_max;
    int32_t clamp;             // CostCalculator.cpp:47-48 (1) or setMaxArrivalInformation (0)
    int32_t layout;            // 0: row-major byte image (WalkLinear), 1: class image (WalkClass), 2: sparse class image (WalkSparse)
    double lo_x, hi_x, lo_y, hi_y, lo_z, hi_z;   // folded clamp bounds: max(poly_min, origin), min(poly_max, origin + sizeInMeters)
    double footprint_radius;   // ceil(robot_radius / resolution)
    double delta_theta, half_fov;
    double min_gt;             // min_arrival_info_gt_
    // candidates
    int32_t n;
    const double *goal;        // [n][3]
    const int32_t *frontier_size;   // or nullptr
    const uint8_t *blacklisted;     // or nullptr
    const uint8_t *achievable_in;   // or nullptr
    const int32_t *perm;            // spatial processing order (candidate ids) or nullptr; outputs stay in list order
    // outputs (device)
    int32_t *ray_counts;       // [n][n_elev][n_yaw] or nullptr
    int32_t *arrival, *argmax, *status;
    double *yaw;
    uint8_t *achievable;
    const float *yawR;         // [n_windows][9] rotation per argmax index (for pose12)
    float *pose12;             // [n][12] R (row-major) + t of the pose (goal, best yaw), or nullptr
    fs_record *records;        // [n] arrival-only records (Fisher columns zero), or nullptr: what fs_get_frontier_costs ranks when no FI is asked for
};

// ---- generic segment tracing (getTracedCells + a RayTracedCells visitor per segment)
struct FsSegArgs {
    FsGridDev grid;
    int32_t n;
    const double *start, *end;      // [n][3]
    double max_length;              // cells (the reference passes it as double)
    int32_t obst_min, obst_max, trace_min, trace_max;
    uint8_t *ok, *hit;
    int32_t *traced, *unknown, *all;
};
hipError_t fs_launch_segments(const FsSegArgs &a, hipStream_t s);
hipError_t fs_launch_brick_scatter(int64_t n_bricks, const int32_t *d_coords, const uint8_t *d_cells, uint8_t *d_grid,
                                   int nx, int ny, int nz, int *d_bad, hipStream_t s);
hipError_t fs_launch_frontier_pair(int n, const float *lx, const float *ly, const float *lz, int m, const float *d_Rt,
                                   const double *d_tri, float *d_out, hipStream_t s);
hipError_t fs_launch_frontier_cells(const uint8_t *d_grid, int nx, int ny, int nz, int lethal_threshold, uint8_t *d_mask,
                                    unsigned long long *d_count, hipStream_t s);

hipError_t fs_launch_frontier_clusters(const uint8_t *d_map, int nx, int ny, double ox, double oy, double res, double px, double py,
                                       int32_t start_pos, double reach, int32_t lethal_threshold, int32_t *d_parent_t, int32_t *d_parent_f,
                                       int32_t *d_aux, uint32_t *d_queue, uint8_t *d_visited, int32_t *d_state, int32_t *d_labels,
                                       int32_t max_clusters, fs_frontier_cluster *d_clusters, long long *d_sums, hipStream_t s);

// ---- key-frame pose information (computeInformationForPose)
struct FsKfArgs {
    int32_t n;                 // poses
    const double *tri;         // [n][12]: FOV triangle at max_depth, then at max_depth + max_depth_error (x0,y0,x1,y1,x2,y2)
    const float *Rt;           // [n][12]
    int32_t n_kf;
    const double *kf_check;    // [n_kf][12]: the key-frame frustum's three vertices and three edge midpoints (depth + error)
    const int32_t *kf_offsets; // [n_kf + 1] into the point arrays
    const float *px, *py, *pz; // key-frame world points, SoA
    double radius;             // < 0: no radius filter
    float qinv;                // Q^-1 diagonal
    int32_t nx, ny;
    double ox, oy, res;
    float *info;
    int32_t *n_cells, *n_points;
    int32_t *flagged;          // [n] poses redone with the HBM table
    unsigned long long *counters;   // [1]
    uint32_t *gtable;          // [pool][3][1 << gbits]
    int32_t gbits;
};
hipError_t fs_launch_kf_info(const FsKfArgs &a, int pool, hipStream_t s);

// ---- landmark staging (fs_capi.hip), split so that fs_multi orders a cloud once for all its devices
struct FsStagedCloud {
    int32_t m = 0, n_chunks = 0;
    std::vector<float> x, y, z;     // SoA in k-d leaf order, padded to whole chunks with far-away sentinels
    std::vector<float> sph;         // [n_chunks][4] bounding spheres (cx, cy, cz, r + safety margin)
};
void fs_stage_landmarks(const float *xyz, int32_t m, FsStagedCloud &out);
int fs_upload_staged_landmarks(fs_ctx *c, const FsStagedCloud &st);
// ... and the same ordering computed on the device ("cloud.order", fs_cloud.hip): every level's node boundaries (a function of the
// number of usable landmarks alone), the level loop (keys kernel + one stable radix sort per level), the SoA gather and the spheres
bool fs_ctx_cloud_on_device(const fs_ctx *c, int32_t m);
void fs_cloud_levels(int32_t n_usable, std::vector<int32_t> &bounds, std::vector<int32_t> &level_off, std::vector<int32_t> &level_nodes,
                     std::vector<int32_t> &level_largest);
size_t fs_cloud_sort_temp_bytes(int32_t n, hipStream_t s);
hipError_t fs_cloud_iota(int32_t *d_perm, int32_t n, hipStream_t s);
hipError_t fs_cloud_order_device(const float *d_raw, int32_t n_usable, const int32_t *d_bounds, const std::vector<int32_t> &level_off,
                                 const std::vector<int32_t> &level_nodes, const std::vector<int32_t> &level_largest, int32_t *d_perm_a, int32_t *d_perm_b, uint64_t *d_keys_a,
                                 uint64_t *d_keys_b, void *d_temp, size_t temp_bytes, uint32_t *d_top_bbox, hipStream_t s, int32_t **perm_out);
size_t fs_cloud_top_bbox_words();   // scratch of the top levels' bounding boxes (d_top_bbox; nullptr: one workgroup per node at every level)
hipError_t fs_cloud_finish(const float *d_raw, const int32_t *d_perm, int32_t n_usable, int32_t n_chunks, float *d_lx, float *d_ly, float *d_lz,
                           float *d_spheres, hipStream_t s);

// ---- FIM kernel arguments ---------------------------------------------------------------------
struct FsFimArgs {
    // landmarks: SoA in k-d leaf order, n_chunks chunks of 64 (the tail padded with far-away sentinels),
    // one bounding sphere (cx, cy, cz, r + safety margin) per chunk
    const float *lx, *ly, *lz;
    const float *spheres;      // [n_chunks][4]
    int32_t n_chunks;
    int32_t n_groups;          // passes of (waves x 64) chunks per workgroup; set by the launcher per kernel configuration
    int32_t cull;              // 0: every chunk is tested (brute force)
    // dense lookup table indexed by the integer voxel lattice
    const float *table;        // [tx][ty][tz], NaN = absent
    int32_t jx0, jy0, jz0;     // lattice index of table[0][0][0]
    int32_t tx, ty, tz;
    double inv_step;           // 1 / (double)0.3f  (FisherInfoManager.hpp:119)
    float inv_step_f;          // (float)inv_step, fast path of the voxel index
    int32_t far_lattice;       // 1: max_dist / step may reach 2^10 lattice cells — the fp32 fast path of the voxel index is not proven there
    float key_thr;             // fast path accepted while |r - rint(r)| < key_thr = 0.5 - 2 * (error bound of the fp32 product at the largest |r|)
    const float *factor;       // crowding factor by rank, [FS_FACTOR_N]; rank >= FS_FACTOR_N -> 0
    float fac1, fac2, fac3, fac4;   // factor[1..4]
    int32_t table_full;        // 1: only finite values inside the table box (true for every generated table)
    // visibility
    float maxd2;               // (float)(max_dist^2)
    float cos2;                // c*c, c = (float)cos(max_angle)
    int32_t cone_mode;         // 0 disabled, 1 c >= 0 (cone also culled per chunk), 2 c < 0, 3 c >= 0 but too wide to cull
    float max_dist_f, cos_a, sin_a;   // chunk culling (cone culled only in mode 1)
    // specialised workers (fs_fim.hip): info_only — the call reads info_ref / n_voxels only (fs_score_fim with NULL for the other
    // columns: what isPoseSafe needs); yaw_only — every pose record is a rotation about Z (checked on the host)
    int32_t info_only, yaw_only;
    int32_t learn;                    // 1: predict scoring passes with the voxel ratio learnt from finished calls (counters[ratio_slot]); 0: skip32 only
    // which learnt ratio this call predicts with and feeds: 12 — distinct voxels per landmark of the chunks in RANGE AND CONE (what the
    // cone workers hash from); 13 — per landmark of the chunks that can also meet the table's BOX (what the INFO_ONLY and the cone-off
    // workers hash from).  Two bases, two ratios: a pose shows up to twice as many voxels per landmark of the second kind.
    // Set by the launchers (fs_fim.hip), not by the caller.
    int32_t ratio_slot;
    // One pose over W = 2^split_shift workgroups (INFO_ONLY LDS worker; fs_fim.hip, SPLIT): cand_count = n * W work items, sums
    // [n * W][18], split_flags [n] (zero between calls: bit 0 a pose's item has handed the pose to the HBM tier, bit 1 the HBM
    // tier's result stands in the pose's first slot).  0: off.
    int32_t split_shift;
    uint32_t *split_flags;
    // slab w of a split pose = lattice x indices [split_bound[w], split_bound[w + 1]) (contiguous slabs; the first and the last one
    // are open-ended): cut by the host so that every slab holds the same share of the visibility volume's cross-section inside
    // the table (FS_SPLIT_MAX_W slabs at most)
    int32_t split_bound[33];
    // a split info-only call whose finish runs on the HOST (fs_capi.hip, host_finish): `sums` then points into mapped page-locked
    // memory, and an item that hands its pose to the HBM tier also raises this word there (plain store; nullptr: not such a call)
    uint32_t *host_flag;
    float box_lo[3], box_hi[3];       // the lookup table's box in the camera frame: half a voxel beyond the outermost lattice points, plus 1 mm
    // poses: Rt[n][12] (R row-major 9 + t 3), written by the host (explicit poses) or by the ray-march kernel
    int32_t n;
    const float *Rt;
    const int32_t *status;     // [n] or nullptr: status != 0 -> zero FI
    // tier 1 may be launched on a slice of the (spatially ordered) candidate list: workgroup b scores candidate
    // cand_perm[cand_lo + b] (cand_perm == nullptr: cand_lo + b), b < cand_count
    const int32_t *cand_perm;
    int32_t cand_lo, cand_count;
    // cost map of the spatial sort (fs_sort.hip), or nullptr: the landmark tests a candidate took are stored under its
    // block, costmap[cand_key[c] & (FS_COST_BINS - 1)], for the order of the next call
    uint32_t *costmap;
    const uint32_t *cand_key;
    // outputs (device)
    float *info_ref, *trace, *logdet, *fim21;   // fim21 may be nullptr
    int32_t *n_visible, *n_voxels;
    double *sums;              // [n][18] reduced per-candidate sums (info, 15 FIM block sums, n_visible, n_voxels)
    uint32_t *overflow;        // [n] tier that must re-score the candidate (0 = done)
    int32_t *flagged;          // [n] work list: candidates the LDS tier hands to the HBM tier
    uint32_t *tested;          // [n] landmark tests spent on the candidate (all tiers); zeroed by the finish kernel
    // fused scoring: the finish kernel also assembles the 32-byte records (nullptr: separate outputs only)
    fs_record *records;
    const int32_t *rec_arrival, *rec_argmax;
    const double *rec_yaw;
    const uint8_t *rec_achievable;
    unsigned long long *counters;   // [16]: 0 landmarks tested; per call 1 multi-pass candidates, 2 handed to the HBM tier, 3 unresolved; 4..6 their running totals; 8 / 9 work-list cursors of the LDS / HBM tier;
                                    // 10 / 11 landmark tests / candidates since the last spatial sort (its cost-map mean); 12 / 13 learnt voxel ratios (ratio_slot)
    // hash tables
    int32_t hash_bits;         // LDS tier (512-thread workgroups)
    int32_t skip32;            // pass-count prediction: distinct voxels <= skip32/32 of the landmarks scanned
    int32_t headroom;          // ... and, once a ratio has been learnt, headroom/32 of it (40 = 5/4) + 1/32 when that is smaller
    uint32_t *gtable;          // tier 3: HBM tables [pool][1 << ghash_bits]
    int32_t ghash_bits;
};

#define FS_COST_BINS   8192     // blocks of the sort's cost map (13 Morton bits)
#define FS_CHUNK       64       // landmarks per chunk (one wave)
#define FS_FACTOR_N    352      // (float)exp(1 - k^0.8) is exactly 0.0f from k = 337 on
// device-side counters of a context (fs_get_counter).  FS_FIM_SCHEDULE development builds append two words per candidate
// (< FS_SCHEDULE_MAX): the 100 MHz tick at which a workgroup of the persistent FIM grid started it, and
// duration | workgroup << 32 | passes << 56 (tools/fim_schedule.py).
#define FS_SCHEDULE_MAX 32768
#if defined(FS_FIM_STAMPS_PER_WAVE)
#define FS_N_COUNTERS 80
#elif defined(FS_FIM_SCHEDULE)
#define FS_N_COUNTERS (32 + 2 * FS_SCHEDULE_MAX)
#else
#define FS_N_COUNTERS 32
#endif
#define FS_SLOT_CNT_BITS 11     // slot = (key+1) << 11 | count
#define FS_SLOT_CNT_MASK ((1u << FS_SLOT_CNT_BITS) - 1u)
#define FS_SLOT_CNT_SAT  1024u  // counts beyond this contribute exactly 0.0f anyway
#define FS_MAX_TABLE_CELLS ((1u << (32 - FS_SLOT_CNT_BITS)) - 2u)


extern std::atomic<uint64_t> fs_alloc_generation;   // bumped by every device / page-locked (re)allocation of the library (fs_capi.hip): launch graphs hold raw pointers

// launchers (defined in the .hip files)
hipError_t fs_launch_raymarch(const FsRayArgs &a, hipStream_t s);
hipError_t fs_launch_sort_candidates(int32_t n, const double *d_goal, const FsGridDev &grid, int32_t *d_perm,
                                    void **scratch, size_t *scratch_bytes, unsigned long long *d_cost_acc,
                                    const uint32_t **d_keys, uint32_t **d_costmap, int use_costmap, int reverse, hipStream_t s);
hipError_t fs_launch_fim(const FsFimArgs &a, hipStream_t s);
bool fs_fim_can_split(const FsFimArgs &a);     // may split_shift be set for this call? (needs info_only, cone_mode, table_full filled in)
hipError_t fs_launch_fim_overflow(const FsFimArgs &a, int pool, hipStream_t s);
hipError_t fs_launch_fim_finish(const FsFimArgs &a, hipStream_t s);
hipError_t fs_launch_selftest(int32_t max_abs, double *d_sqrt, double *d_div, hipStream_t s);

// fs_score_candidates split for the multi-device scorer (fs_capi.hip): launch everything / wait and copy out
extern "C" int fs_score_candidates_begin(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                                         const uint8_t *blacklisted, const uint8_t *achievable_in);
extern "C" int fs_score_candidates_end(fs_ctx *c, int32_t n, fs_record *records);
extern "C" int fs_score_arrival_begin(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                                      const uint8_t *blacklisted, const uint8_t *achievable_in, int32_t *ray_counts, int32_t *arrival,
                                      int32_t *argmax, double *yaw, uint8_t *achievable, int32_t *status);
extern "C" int fs_score_arrival_end(fs_ctx *c);
extern "C" int fs_score_fim_begin(fs_ctx *c, int32_t n, const double *pose7, float *info_ref, float *fim21, float *trace, float *logdet,
                                  int32_t *n_visible, int32_t *n_voxels);
extern "C" int fs_score_fim_end(fs_ctx *c);

// pieces of fs_multi_get_frontier_costs (defined in fs_capi.hip, sequenced by fs_multi.hip)
hipStream_t fs_ctx_stream(fs_ctx *c);
int fs_ctx_device(const fs_ctx *c);
int fs_gather_begin(fs_ctx *c, int32_t n, const uint8_t *blacklisted, const double *path_length, const double *path_heading, fs_record **d_list);
int fs_block_score_begin(fs_ctx *c, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                         const uint8_t *achievable_in, bool with_fim, fs_record *d_dst, fs_record **d_block);
int fs_block_records_to_host(fs_ctx *c, int32_t n, const fs_record *d_block, const fs_record **h_block);
int fs_gather_rank(fs_ctx *c, int32_t n, double alpha, double beta, double max_vx, double max_wz);
int fs_gather_end(fs_ctx *c, int32_t n, fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order);

#endif
