/*
 * fitslam_frontier.h — C ABI of the MI355X-native frontier-scoring path.
 *
 * This is the drop-in boundary for ONE hot path of suchetanrs/FIT-SLAM: per-candidate arrival
 * information (ray fan through the occupancy grid, unknown-cell counts, FOV window max, best yaw)
 * and landmark Fisher information (voxel-lookup scalar with crowding discount + 6x6 FIM).
 * Plain pointers and sizes only; no C++ / torch / ROS types.  Every entry point cites the reference
 * interface it replaces.  Abbreviations (under the FIT-SLAM tree):
 *   DEP/ = dev_ws/src/DEPRECATED/frontier_exploration/frontier_exploration/
 *   FIP/ = dev_ws/src/fit-slam2/fisher_information_plugins/
 *
 * Conventions
 *   - every function returns FS_OK (0) or a negative FS_E_* code and never throws; the message of
 *     the last failure on a context is available from fs_last_error().
 *   - a context is single-caller (externally synchronised), owns one HIP stream (or borrows the
 *     one passed at creation) and owns device copies of grid, landmarks and lookup table.
 *     DIFFERENT contexts may be used from different host threads at the same time.
 *     Host buffers stay caller-owned; nothing is allocated across the ABI.
 *   - grid layout [nz][ny][nx] uint8, index = (z*ny + y)*nx + x; nz == 1 is the reference's
 *     nav2_costmap_2d::Costmap2D (cost constants: 255 unknown, 254 lethal, 253 inscribed, 0 free).
 *   - there is NO CPU fallback: if the HIP runtime or a gfx950 device is missing, fs_ctx_create
 *     fails with FS_E_NO_DEVICE.
 */
#ifndef FITSLAM_FRONTIER_H_
#define FITSLAM_FRONTIER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_ABI_VERSION 1

/* return codes */
#define FS_OK            0
#define FS_E_INVALID    -1   /* bad argument / parameters (e.g. fewer rays than the FOV window) */
#define FS_E_NO_DEVICE  -2   /* HIP runtime or device unavailable — no CPU fallback exists */
#define FS_E_HIP        -3   /* a HIP call failed */
#define FS_E_STATE      -4   /* call order: grid / landmarks / table / params not set yet */
#define FS_E_IO         -5   /* lookup-table file could not be read or written */
#define FS_E_RANGE      -6   /* utility outside [0,1] (the reference throws, FrontierCostsManager.cpp:148-149,173-174) */

/* per-candidate status (records.flags bits 8..15, status[] of fs_score_arrival) */
#define FS_STATUS_OK          0
#define FS_STATUS_OFF_MAP     1   /* a worldToMap failed: arrival 0, yaw 0 (DEP/src/CostCalculator.cpp:50-55) */
#define FS_STATUS_BLACKLISTED 2   /* DEP/src/FrontierCostsManager.cpp:77-86 */

#define FS_MAX_ELEV 16

typedef struct fs_ctx fs_ctx;

/* Replaces FrontierCostCalculator's constructor parameters (DEP/src/CostCalculator.cpp:5-21:
 * costCalculator/max_camera_depth, delta_theta, camera_fov, Costmap2DROS::getRobotRadius()),
 * the RayTracedCells ranges of DEP/src/CostCalculator.cpp:40, the limits factors of :186-188
 * (a parameter in fit-slam2: fit_slam2/params/active_slam_exploration_params.yaml:17) and
 * CostAssigner's polygon bbox (DEP/src/CostAssigner.cpp:148-165). */
typedef struct {
    double max_camera_depth;     /* 2.0  */
    double delta_theta;          /* 0.10 */
    double camera_fov;           /* 1.04 */
    double robot_radius;         /* 0.60 */
    int32_t n_rays;              /* 0: the reference loop `theta <= 2*pi` (63 rays at 0.10); >0: exactly n yaw rays */
    int32_t n_elev;              /* elevation rings of the 3-D extension; 1 with elev[0] == 0 is the reference */
    double elev[FS_MAX_ELEV];    /* radians */
    int32_t obst_min, obst_max;  /* 240, 254 */
    int32_t trace_min, trace_max;/* 255, 255 */
    double factor_max;           /* 1.2  */
    double factor_min;           /* 0.70 */
    double polygon[4];           /* minx, miny, maxx, maxy */
} fs_ray_params;

/* Replaces the GetLandmarksInView request fields (FIP/src/fisher_information/FisherInfoManager.cpp:60-65)
 * by an explicit visibility predicate (DESIGN.md "Visibility"). */
typedef struct {
    double max_dist;             /* 14.0; must be positive and below 1e9 m */
    double max_angle;            /* 1.0 rad from the camera +x axis; >= pi disables the cone */
} fs_fim_params;

/* Fixed-size per-candidate result record (the unit of the multi-GPU all-gather). 32 bytes. */
typedef struct {
    int32_t arrival;             /* Frontier::setArrivalInformation (DEP/src/CostCalculator.cpp:112) */
    int32_t argmax;              /* maxIndex of the FOV window (:98-107) */
    float   yaw;                 /* goal orientation, maxIndex*delta_theta + fov/2 (:119) */
    float   info_ref;            /* isPoseSafe's `information` (FIP/src/.../FisherInfoManager.cpp:100) */
    float   trace;               /* trace of the unit-weight 6x6 FIM over the visible landmarks */
    float   logdet;              /* log det of that FIM (D-optimality); -inf if singular */
    int32_t n_visible;           /* landmarks passing the visibility predicate */
    uint32_t flags;              /* bit0 achievable; bits 8..15 status; bits 16..31 min(n_voxels, 65535) */
} fs_record;

#define FS_FLAG_ACHIEVABLE 1u
#define FS_RECORD_STATUS(flags)   (((flags) >> 8) & 0xffu)
#define FS_RECORD_NVOXELS(flags)  (((flags) >> 16) & 0xffffu)

/* ---------------------------------------------------------------- context */

/* device_id: HIP device ordinal.  stream: a hipStream_t to borrow (e.g. the caller's current
 * stream) or NULL to create one owned by the context. */
int  fs_ctx_create(int device_id, void *stream, fs_ctx **out);
void fs_ctx_destroy(fs_ctx *ctx);
const char *fs_last_error(const fs_ctx *ctx);
int  fs_abi_version(void);
int  fs_synchronize(fs_ctx *ctx);

/* (per-kernel timing, device counters, tuning knobs and the fp64 self test — development and measurement aids that have no
 * counterpart in the reference — are declared in fitslam_frontier_dev.h) */

/* ---------------------------------------------------------------- arrival information (ray-cast) */

/* Replaces FrontierCostCalculator::FrontierCostCalculator (DEP/src/CostCalculator.cpp:5-21). */
int fs_set_ray_params(fs_ctx *ctx, const fs_ray_params *p);
/* number of yaw rays / FOV window the parameters produce (DEP/src/CostCalculator.cpp:36,87) */
int fs_ray_fan_shape(const fs_ctx *ctx, int32_t *n_yaw, int32_t *n_elev, int32_t *window);

/* Replaces the raw `nav2_costmap_2d::Costmap2D *exploration_costmap_` the scorer holds
 * (DEP/src/CostCalculator.cpp:10).  Copies the grid to HBM — the snapshot point; call it under the
 * costmap mutex.  Invalidates cached arrival limits. */
int fs_upload_grid(fs_ctx *ctx, const uint8_t *cells, int32_t nx, int32_t ny, int32_t nz,
                   const double origin_xyz[3], double resolution);

/* Same snapshot from a SPARSE map: n_bricks bricks of 8x8x8 cells (brick_xyz [n][3] in brick units, brick_cells
 * [n][512] with index (z*8 + y)*8 + x), everything else default_value (255 = unknown).  This is the wire format of
 * a hashed voxel map (BASELINE.json configs[4], 1024^3); in HBM the grid stays dense (1 GiB of 288 GB) so that the
 * ray walk needs no hash probe per cell.  Dimensions must be multiples of 8. */
int fs_upload_grid_bricks(fs_ctx *ctx, int32_t nx, int32_t ny, int32_t nz, const double origin_xyz[3], double resolution,
                          uint8_t default_value, int64_t n_bricks, const int32_t *brick_xyz, const uint8_t *brick_cells);

/* A WINDOW of the staged map rewritten in place — what a costmap update cycle does to the master grid: every layer's
 * `updateCosts(master_grid, min_i, min_j, max_i, max_j)` writes only inside the cycle's bounds through `getCharMap()` (the
 * reference's own layers: DEP/src/nav2_plugins/lethal_marker.cpp:305-325, fit_slam2_nav2_plugins/plugins/keepout_layer.cpp:279-300;
 * the costmap is a rolling / bounded-update one, fit_slam2/params/active_slam_nav2_params.yaml:124).  The window is cells
 * [x0, x0+sx) x [y0, y0+sy) x [z0, z0+sz) of the grid fs_upload_grid staged (same shape, origin and resolution: a map that
 * moved or was resized is a new snapshot).  `cells` points at the window's first cell, x fastest; row_stride / slice_stride are
 * the byte distances between its rows / z slices — pass `getCharMap() + y0 * size_x + x0` with row_stride = size_x to send a
 * window of the live costmap without packing it; 0 = tightly packed (sx, sx * sy).  Images derived from the grid are updated
 * for the bricks the window touches only; cached arrival limits stay (setMaxArrivalInformation's fan does not depend on the
 * cells, DEP/src/CostCalculator.cpp:123-191).  Scoring afterwards equals scoring after fs_upload_grid of the whole rewritten
 * map, bit for bit.  An empty window is a no-op; one that leaves the grid is refused (FS_E_INVALID), nothing written. */
int fs_update_grid_region(fs_ctx *ctx, int32_t x0, int32_t y0, int32_t z0, int32_t sx, int32_t sy, int32_t sz,
                          const uint8_t *cells, int64_t row_stride, int64_t slice_stride);

/* Frontier-cell predicate of FrontierSearch::isNewFrontierCell (DEP/src/FrontierSearch.cpp:218-249; isFree / isLethal /
 * isUnknown: DEP/include/.../FrontierSearch.hpp:129-142; frontierSearch/lethal_threshold 160) evaluated for every cell
 * of the staged grid, slice by slice: unknown cell, no lethal in-plane 4-neighbour, at least one free one.
 * mask [nz][ny][nx] (1 = frontier cell) may be NULL; *count = number of frontier cells.  The BFS clustering of the
 * reference (searchFrom / buildNewFrontier) consumes this mask on the host. */
int fs_frontier_cells(fs_ctx *ctx, int32_t lethal_threshold, uint8_t *mask, int64_t *count);

/* Frontier detection + clustering (second half).  Replaces, for a 2-D costmap (nz == 1),
 * std::vector<FrontierPtr> FrontierSearch::searchFrom(geometry_msgs::msg::Point position) (DEP/include/.../FrontierSearch.hpp:62,
 * DEP/src/FrontierSearch.cpp:21-96) with buildNewFrontier (:98-216), nearestFreeCell (DEP/src/Helpers.cpp:285-329) and
 * isNewFrontierCell (:218-249): the frontier cells the search collects — the 8-connected components of the frontier-cell
 * set that touch the region the outer search expands from the robot (cost < 254, within max_frontier_distance +
 * max_frontier_cluster_size * resolution * 1.414) — as clusters.
 *   labels   [ny][nx] or NULL: -1, or the cluster's label = the smallest cell index (y * nx + x) of its component
 *   clusters [max_clusters], ascending label; *n_clusters = clusters found (may exceed max_clusters: the first ones by label are
 *            stored); *n_cells = frontier cells found = every_frontier_list.size() of the reference
 * The reference then cuts a component into pieces of max_frontier_cluster_size + 1 cells in the order of its queue and takes
 * an angular median as each piece's goal point (:146-205): order-dependent steps that stay with the caller; a component of
 * `size` cells yields size / (max + 1) full pieces and, if size % (max + 1) > min_frontier_cluster_size, one more.
 * A robot position off the map gives no cluster (:28-33). */
typedef struct {
    int32_t label;               /* smallest cell index of the component */
    int32_t size;                /* cells in the component */
    double  centroid_x, centroid_y;   /* mean of the cell centres (mapToWorld), world frame */
    int32_t min_x, min_y, max_x, max_y;   /* bounding box in cells */
} fs_frontier_cluster;
int fs_frontier_clusters(fs_ctx *ctx, const double robot_xy[2], int32_t lethal_threshold, double max_frontier_distance,
                         int32_t max_frontier_cluster_size, int32_t *labels, int32_t max_clusters,
                         fs_frontier_cluster *clusters, int32_t *n_clusters, int64_t *n_cells);

/* Replaces double FrontierCostCalculator::setMaxArrivalInformation() (DEP/include/.../CostCalculator.hpp:58,
 * DEP/src/CostCalculator.cpp:123-191): geometric maximum of the FOV window on an obstacle-free fan from
 * world (0,0).  max_value = the window maximum (0 if (0,0) is off-map: limits stay unset, as the
 * reference); caches max_gt = factor_max*max_value, min_gt = factor_min*max_gt in the context. */
int fs_max_arrival(fs_ctx *ctx, double *max_value, double *max_gt, double *min_gt);
/* override the cached limits (e.g. restored from a running node) */
int fs_set_arrival_limits(fs_ctx *ctx, double max_gt, double min_gt);

/* Replaces void FrontierCostCalculator::setArrivalInformationForFrontier(FrontierPtr&, std::vector<double>&)
 * (DEP/include/.../CostCalculator.hpp:56, DEP/src/CostCalculator.cpp:23-121) applied to the whole
 * frontier list of FrontierCostsManager::assignCosts (DEP/src/FrontierCostsManager.cpp:74-119),
 * including its blacklist branch (:77-86).  Order-preserving.
 *   goal_xyz      [n][3] Frontier::getGoalPoint() (z: origin_z for 2-D grids)
 *   frontier_size [n]    Frontier::getSize() or NULL (0)
 *   blacklisted   [n]    or NULL (none)
 *   achievable_in [n]    Frontier::isAchievable() before the call, or NULL (true)
 *   ray_counts    [n][n_elev][n_yaw] information_along_ray, or NULL
 *   arrival, argmax [n]; yaw [n] (theta_s_star); achievable [n]; status [n] */
int fs_score_arrival(fs_ctx *ctx, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                     const uint8_t *blacklisted, const uint8_t *achievable_in,
                     int32_t *ray_counts, int32_t *arrival, int32_t *argmax, double *yaw,
                     uint8_t *achievable, int32_t *status);

/* Replaces bool getTracedCells(double sx, double sy, double wx, double wy, RayTracedCells&, double max_length,
 * Costmap2D*) (DEP/include/.../Helpers.hpp:125-126, DEP/src/Helpers.cpp:32-96) with a RayTracedCells visitor
 * (Helpers.hpp:20-111) for a batch of arbitrary segments — the form FrontierRoadMap::isConnectable
 * (DEP/src/planners/FrontierRoadmap.cpp:716-737, visitor (253,254,0,255)) and the recovery controller
 * (fit_slam2_recovery/src/recovery_controller.cpp:78-88, visitor (256,256,0,255)) use.
 *   start_xyz, end_xyz [n][3]; max_length_cells as the reference passes it (a double, in cells)
 *   ok [n]      the bool return (false: a worldToMap failed)        traced [n]  getCells().size()
 *   hit [n]     hasHitObstacle()     unknown [n] getNumUnknown()     all [n]     getCellsSize() */
int fs_trace_segments(fs_ctx *ctx, int32_t n, const double *start_xyz, const double *end_xyz, double max_length_cells,
                      int32_t obst_min, int32_t obst_max, int32_t trace_min, int32_t trace_max,
                      uint8_t *ok, int32_t *traced, uint8_t *hit, int32_t *unknown, int32_t *all);

/* ---------------------------------------------------------------- Fisher information */

/* Replaces the per-query service response `map_points` (FIP/src/.../FisherInfoManager.cpp:60-88)
 * by the whole landmark cloud staged once: xyz [m][3] float32, world frame.  A point with a NaN / infinite coordinate, or one
 * beyond 1e17 m, is visible from nowhere; it keeps its place in the count m and contributes to no pose. */
int fs_upload_landmarks(fs_ctx *ctx, const float *xyz, int32_t m);

/* Replaces generateLookupTable(minX,maxX,minY,maxY,minZ,maxZ) (FIP/include/.../FisherInfoManager.hpp:94,
 * FIP/src/.../FisherInfoManager.cpp:117-229).  bounds == NULL uses gen_fi_lookup's arguments
 * (DEP/src/fisher_information/GenerateLookupMain.cpp:9). */
int fs_lookup_generate(fs_ctx *ctx, const float bounds[6]);
/* Replaces loadLookupTable() (FisherInfoManager.hpp:96, FisherInfoManager.cpp:231-262): raw
 * {float key[3]; float value} records, host-endian, no header.  Missing file -> FS_E_IO (the reference throws). */
int fs_lookup_load(fs_ctx *ctx, const char *path);
int fs_lookup_save(fs_ctx *ctx, const char *path);        /* byte-compatible with the reference's .dat */
int fs_lookup_set_records(fs_ctx *ctx, const float *records /* [n][4] */, int64_t n);
int fs_lookup_num_records(const fs_ctx *ctx, int64_t *n);
int fs_lookup_get_records(const fs_ctx *ctx, float *records /* [n][4] */);
/* getInformationFromLookup(Eigen::Vector3f&, ...) (FisherInfoManager.cpp:264-285): plain table value, NaN on miss */
int fs_lookup_query(const fs_ctx *ctx, const float p_camera[3], float *value);

int fs_set_fim_params(fs_ctx *ctx, const fs_fim_params *p);

/* Replaces bool FisherInformationManager::isPoseSafe(geometry_msgs::msg::Pose&, bool, float& information)
 * (FIP/include/.../FisherInfoManager.hpp:125, FIP/src/.../FisherInfoManager.cpp:39-115) for a batch of
 * poses; `safe = info_ref[i] > threshold` stays with the caller (threshold 550, FisherInfoBTPlugin.cpp:20).
 *   pose7     [n][7] position xyz + orientation quaternion xyzw (geometry_msgs::Pose order)
 *   info_ref  [n]    the reference scalar: sum of table value x crowding factor over visible landmarks
 *   fim21     [n][21] upper triangle (row-major) of the unit-weight 6x6 FIM, or NULL
 *   trace, logdet [n] or NULL;  n_visible, n_voxels [n] or NULL
 * A call that passes NULL for fim21, trace, logdet and n_visible — i.e. asks for what isPoseSafe itself reads (:83-100), plus
 * n_voxels if wanted — is served by a worker that neither accumulates the 6x6 sums nor looks at landmarks outside the lookup
 * table's box (a table miss contributes nothing, :90-94): same info_ref (to the last bits) and the same n_voxels, less work.
 * The visibility volume is fs_set_fim_params'; the reference's own request is max_dist 14, max_angle 4.0 (cone off, :63-64). */
int fs_score_fim(fs_ctx *ctx, int32_t n, const double *pose7, float *info_ref, float *fim21,
                 float *trace, float *logdet, int32_t *n_visible, int32_t *n_voxels);

/* Replaces float computeInformationFrontierPair(std::vector<Point>& lndmrk_w, Pose& kf_pose_w, Pose& est_pose_w,
 * std::vector<Point2D>& FOVFrontierPair) (FIP/src/.../FisherInformationHelpers.cpp:125-143; isInside / onLeft:
 * FIP/include/.../FisherInformationHelpers.hpp:20-43) for a batch of (estimation pose, CCW triangle) pairs over the
 * staged landmark cloud: sum of the local-Jacobian trace (computeInformationOfPointLocal, :106-112) of the landmarks
 * whose (x, y) lies strictly inside the triangle.  triangle_xy [n][3][2].  (kf_pose_w only feeds an unused local in
 * the reference; the function itself is not called at run time there.) */
int fs_information_frontier_pair(fs_ctx *ctx, int32_t n, const double *est_pose7, const double *triangle_xy, float *information);

/* Key-frame pose information.  Replaces, for a batch of poses,
 *   std::pair<float, std::vector<Eigen::Vector3f>> frontier_exploration_information_affine::computeInformationForPose(
 *       Pose& pose, std::vector<Pose>& neighbouring_poses, std::vector<int> neighbouring_ids, slam_msgs::msg::MapData&,
 *       double max_depth, double hfov, double max_depth_error, Eigen::Matrix3f Q, bool pcl_return, Logger, Costmap2D*, bool)
 * (DEP/include/frontier_exploration/deprecated/util.hpp:840-916; dead code in the reference — its call site is the commented
 * block DEP/src/CostCalculator.cpp:326-365) together with getNodesInRadius (util.hpp:616-632), frustumOverlap (:172-185),
 * getVerticesOfFrustum2D (:101-119), isPointInsideTriangle (:49-66) and the affine computeInformationOfPoint (:687-759).
 * fs_upload_keyframes stages slam_msgs MapData: key-frame poses (graph.poses) and their world points (nodes[].word_pts)
 * as a CSR list.  The costmap geometry (x/y origin, resolution, size) is the one of fs_upload_grid. */
typedef struct fs_keyframe_params {
    double max_depth;          /* 2.0   (CostCalculator.cpp:358) */
    double hfov;               /* 1.089 */
    double max_depth_error;    /* 0.5 */
    float  q_diag;             /* Q = q_diag * I, 0.01f (CostCalculator.cpp:356) */
    double radius;             /* getNodesInRadius radius, 4.5; < 0: every key-frame is a neighbour */
} fs_keyframe_params;
int fs_upload_keyframes(fs_ctx *ctx, int32_t n_keyframes, const double *kf_pose7 /* [n][7] xyz + quat xyzw */,
                        const int32_t *kf_offsets /* [n + 1] */, const float *points_xyz /* [kf_offsets[n]][3] */);
/* information [n]: the pose information (.first of the pair); n_cells [n]: information_map.size(); n_points [n]: points that
 * were added (inside the FOV triangle and on the map); the last two may be NULL.  The point list of pcl_return is not produced. */
int fs_information_for_pose(fs_ctx *ctx, int32_t n, const double *pose7, const fs_keyframe_params *params,
                            float *information, int32_t *n_cells, int32_t *n_points);

/* ---------------------------------------------------------------- fused scoring */

/* Replaces the scoring half of bool CostAssigner::getFrontierCosts(req, res)
 * (DEP/include/.../CostAssigner.hpp:70, DEP/src/CostAssigner.cpp:73-119): arrival information for every
 * candidate, then Fisher information at the pose (goal, best yaw) built like
 * isPoseSafe(Point, Point, bool) builds one (FIP/src/.../FisherInfoManager.cpp:31-37,
 * DEP/include/.../util/GeometryUtils.hpp:112-124).  Candidates whose status != OK get zero FI.
 * records [n] host memory, same order as the input list. */
int fs_score_candidates(fs_ctx *ctx, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                        const uint8_t *blacklisted, const uint8_t *achievable_in, fs_record *records);

/* Same, with every buffer already resident in HBM (device pointers; frontier_size / blacklisted /
 * achievable_in may be NULL).  Asynchronous on the context's stream: call fs_synchronize or
 * synchronise the borrowed stream before reading d_records.  This is the form the multi-GPU
 * shard runner uses: d_records is the send buffer of the RCCL all-gather. */
int fs_score_candidates_dev(fs_ctx *ctx, int32_t n, const double *d_goal_xyz, const int32_t *d_frontier_size,
                            const uint8_t *d_blacklisted, const uint8_t *d_achievable_in, fs_record *d_records);

/* ---------------------------------------------------------------- one process, several GPUs */

/* The reference scores in-process, from its one behaviour-tree thread (DEP/src/main.cpp:9-24,
 * DEP/src/ExplorationBT.cpp:376-410: ProcessFrontierCostsBT::onStart -> CostAssigner::getFrontierCosts).  fs_multi keeps that
 * shape on a multi-GPU node: ONE object, ONE calling thread, no launcher and no second process.  It owns one fs_ctx per
 * entry of device_ids (an ordinal may repeat: two contexts on one GPU, each with its stream), every staging call is
 * applied to all of them (the grid, the cloud and the table are replicated: 128 MiB + 1.2 MB + 2.8 MB at 512^3), and
 * fs_multi_score_candidates cuts the frontier list into contiguous blocks of ceil(n / G) candidates (fs_multi_shard_bounds
 * — the same rule the multi-process bench uses), starts block g on device g without waiting, then collects the 32-byte
 * records of all blocks into the caller's buffer in list order.  fs_multi_get_frontier_costs is the one-call form: blocks
 * gathered device to device over xGMI onto member 0's GPU, ranked there, one transfer out (below). */
typedef struct fs_multi fs_multi;
int  fs_multi_create(const int *device_ids, int n_devices, fs_multi **out);
void fs_multi_destroy(fs_multi *m);
int  fs_multi_num_devices(const fs_multi *m);
fs_ctx *fs_multi_ctx(fs_multi *m, int i);                  /* member context i (options, counters, ranking); owned by m */
const char *fs_multi_last_error(const fs_multi *m);
/* block [*lo, *hi) of shard `shard` of `n_shards` over a list of n: lo = min(n, shard * ceil(n / n_shards)), hi = min(n, lo + ceil(n / n_shards)) */
int  fs_multi_shard_bounds(int32_t n, int n_shards, int shard, int32_t *lo, int32_t *hi);
/* each of these applies the fs_* call of the same name to every member (first failure is returned).  The two uploads reach
   all devices at the same time (one short-lived host thread per member; the cloud is k-d ordered once for all of them);
   the caller sees one blocking call. */
int  fs_multi_set_option(fs_multi *m, const char *key, double value);
int  fs_multi_set_ray_params(fs_multi *m, const fs_ray_params *p);
int  fs_multi_upload_grid(fs_multi *m, const uint8_t *cells, int32_t nx, int32_t ny, int32_t nz, const double origin_xyz[3], double resolution);
int  fs_multi_update_grid_region(fs_multi *m, int32_t x0, int32_t y0, int32_t z0, int32_t sx, int32_t sy, int32_t sz,
                                 const uint8_t *cells, int64_t row_stride, int64_t slice_stride);   /* the window on every device */
int  fs_multi_upload_landmarks(fs_multi *m, const float *xyz, int32_t n_landmarks);
int  fs_multi_lookup_generate(fs_multi *m, const float bounds[6]);
int  fs_multi_lookup_load(fs_multi *m, const char *path);
int  fs_multi_set_fim_params(fs_multi *m, const fs_fim_params *p);
/* setMaxArrivalInformation once (member 0), the limits handed to every member */
int  fs_multi_max_arrival(fs_multi *m, double *max_value, double *max_gt, double *min_gt);
/* fs_score_arrival over all members (same arguments; every output array in list order) */
int  fs_multi_score_arrival(fs_multi *m, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                            const uint8_t *blacklisted, const uint8_t *achievable_in,
                            int32_t *ray_counts, int32_t *arrival, int32_t *argmax, double *yaw,
                            uint8_t *achievable, int32_t *status);
/* fs_score_candidates over all members: host buffers in, records [n] out, list order; returns when every block is in */
int  fs_multi_score_candidates(fs_multi *m, int32_t n, const double *goal_xyz, const int32_t *frontier_size,
                               const uint8_t *blacklisted, const uint8_t *achievable_in, fs_record *records);
/* fs_score_fim over all members (FisherInformationManager::isPoseSafe's batch, FIP/src/fisher_information/FisherInfoManager.cpp:39-115):
   poses cut into the same contiguous blocks, every member started before the first is waited for, every column in list order */
int  fs_multi_score_fim(fs_multi *m, int32_t n, const double *pose7, float *info_ref, float *fim21,
                        float *trace, float *logdet, int32_t *n_visible, int32_t *n_voxels);
/* fs_get_frontier_costs — the whole of CostAssigner::getFrontierCosts (DEP/src/CostAssigner.cpp:73-119; the in-process call of
 * DEP/src/ExplorationBT.cpp:376-410) — over all members, with the records NEVER visiting the host between scoring and ranking:
 * member g scores block g on its device; its 32-byte records are moved device to device into ONE list on member 0's GPU
 * (hipMemcpyPeerAsync over xGMI on the member's stream behind its kernels, one event per member that member 0's stream waits
 * for; a member that shares member 0's GPU writes the list itself); fs_rank_candidates_dev ranks the gathered list there; ONE
 * transfer brings records, costs, utilities and order back.  Same arguments, same results (integers, costs and order bit for
 * bit) as fs_get_frontier_costs on one context.  A gather onto the ranking device, not an all-gather: in one process only
 * that device needs the list.  Where the runtime refuses peer access between a member and member 0 the blocks bounce through
 * page-locked host memory instead; the call still succeeds and fs_multi_last_error says so (fs_multi_gather_mode: 2). */
int  fs_multi_get_frontier_costs(fs_multi *m, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                                 const uint8_t *achievable_in, const double *path_length, const double *path_heading,
                                 double alpha, double beta, double max_vx, double max_wz, int with_fisher_information,
                                 fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order);
/* how fs_multi_get_frontier_costs moves the blocks: 1 device to device (peer access granted, or every member on one GPU),
   2 through page-locked host memory (peer access refused), 3 / forced values: fs_multi_set_option("multi.gather", 0 auto | 1 | 2 | 3
   = device copies even between members of one GPU | 4 = those copies through hipMemcpyPeerAsync), for tests of the paths a one-GPU
   box cannot take by itself.  < 0: error */
int  fs_multi_gather_mode(fs_multi *m);

/* ---------------------------------------------------------------- utility + ranking */

/* Replaces the U1 block of FrontierCostsManager::assignCosts (DEP/src/FrontierCostsManager.cpp:118,126-205)
 * and the four parallel vectors of GetFrontierCostsResponse (DEP/include/.../CostAssigner.hpp:51-59),
 * on the GPU.  records [n] host (from fs_score_candidates or gathered from all shards);
 * path_length / path_heading [n] from the planner (out of scope; inputs here).
 * Outputs: weighted_cost, arrival_utility, distance_utility [n]; order [n] = candidate indices by
 * ascending cost (stable).  Returns FS_E_RANGE where the reference would throw. */
int fs_rank_candidates(fs_ctx *ctx, int32_t n, const fs_record *records, const uint8_t *blacklisted,
                       const double *path_length, const double *path_heading,
                       double alpha, double beta, double max_vx, double max_wz,
                       double *weighted_cost, double *arrival_utility, double *distance_utility,
                       int32_t *order);

/* The whole of bool CostAssigner::getFrontierCosts(req, res) (DEP/include/.../CostAssigner.hpp:43-59,70; DEP/src/CostAssigner.cpp:73-119)
 * — FrontierCostsManager::assignCosts' arrival information for every frontier (DEP/src/FrontierCostsManager.cpp:74-119) followed by
 * its U1 block (:126-205) — as ONE call: frontier list and the planner's path columns in, scored and ranked candidates out; one
 * transfer each way and one synchronisation, the records never visit the host between scoring and ranking.  Up to 1024
 * frontiers (the reference handles tens per tick) the kernels read the inputs from, and write the results into, the context's
 * mapped page-locked buffers in place — plain launches, no transfers of their own.  (Replaying the sequence as a captured launch
 * graph exists behind fs_set_option("graph", 1) of fitslam_frontier_dev.h and is OFF by default: it measured 5-7 us slower per
 * call than the plain launches on this runtime; only with it on is the list padded to power-of-two buckets.)
 *   path_length, path_heading [n]  what the planner set on each frontier (Frontier::setPathLength / setPathHeading; inputs here).
 *                                  The reference plans a frontier only once its arrival information has left it achievable
 *                                  (FrontierCostsManager.cpp:88-91); here the columns come first — those of a frontier that turns
 *                                  out unachievable (or is blacklisted) are never read: same costs, the plan was spare work
 *   with_fisher_information        0: arrival information only — what the reference's assignCosts uses; the Fisher columns of the
 *                                  records stay zero.  1: also the Fisher information at the pose (goal, best yaw), as fs_score_candidates
 *   records [n]; weighted_cost [n]; arrival_utility, distance_utility, order [n] or NULL.  FS_E_RANGE where the reference throws. */
int fs_get_frontier_costs(fs_ctx *ctx, int32_t n, const double *goal_xyz, const int32_t *frontier_size, const uint8_t *blacklisted,
                          const uint8_t *achievable_in, const double *path_length, const double *path_heading,
                          double alpha, double beta, double max_vx, double max_wz, int with_fisher_information,
                          fs_record *records, double *weighted_cost, double *arrival_utility, double *distance_utility, int32_t *order);

/* Same, behind the scoring call on the device: every pointer is device memory (d_records = the records fs_score_candidates_dev
 * wrote, or the receive buffer of the all-gather), launched on the context's stream and NOT waited for — score and rank run
 * back to back without the records ever visiting the host ("fused after scoring").  d_arrival_utility,
 * d_distance_utility, d_order may be NULL (not wanted).  d_range_error (one int32, or NULL): non-zero after the kernels have run
 * where the reference would throw (FS_E_RANGE of the host form) — read it together with the results. */
int fs_rank_candidates_dev(fs_ctx *ctx, int32_t n, const fs_record *d_records, const uint8_t *d_blacklisted,
                           const double *d_path_length, const double *d_path_heading,
                           double alpha, double beta, double max_vx, double max_wz,
                           double *d_weighted_cost, double *d_arrival_utility, double *d_distance_utility,
                           int32_t *d_order, int32_t *d_range_error);

#ifdef __cplusplus
}
#endif
#endif
