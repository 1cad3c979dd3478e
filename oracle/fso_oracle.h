/*
 * fso_oracle.h — CPU ORACLE for the FIT-SLAM frontier-scoring hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (fit-slam_amd/csrc,
 * include/fitslam_frontier.h) never links, loads or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithms, each function citing the
 * reference file:line it follows.  Abbreviations (all under /root/reference/):
 *   DEP/ = dev_ws/src/DEPRECATED/frontier_exploration/frontier_exploration/
 *   FIP/ = dev_ws/src/fit-slam2/fisher_information_plugins/
 *
 * PARITY PIN STATUS: PARITY UNPINNED in the strict sense — the reference
 * has no automated tests, no golden vectors and no fixtures with expected values for this
 * path, and it cannot be compiled here (ROS 2 Humble, Eigen3, nav2, BT.CPP, two un-vendored
 * submodules), so no output of the reference itself is available to check against.  What
 * it does hold — the INPUTS of its three manual programs — is committed
 * (tests/golden/ref_held_inputs.npz) and replayed.  Short of that the oracle is pinned
 * by (i) the analytic known answers derivable from the reference's own manual test
 * inputs (DEP/tests/main_fim_computation.cpp:8-56, DEP/tests/fim_viz.cpp:70-100),
 * (ii) an independent slow Python transcription (oracle/pyref.py) and (iii) property
 * tests.  Third-party semantics (nav2_costmap_2d::Costmap2D, Eigen, the un-vendored
 * GetLandmarksInView server) are restated from their published behaviour:
 * "parity unpinned" at those boundaries.
 */
#ifndef FSO_ORACLE_H_
#define FSO_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-candidate status codes (shared numbering with include/fitslam_frontier.h) ---- */
#define FSO_STATUS_OK          0
#define FSO_STATUS_OFF_MAP     1   /* a worldToMap failed: arrival 0, yaw 0 (DEP/src/CostCalculator.cpp:50-55) */
#define FSO_STATUS_BLACKLISTED 2   /* DEP/src/FrontierCostsManager.cpp:77-86 */

/* uint8 grid, layout [nz][ny][nx], index = (z*ny + y)*nx + x.  nz == 1 is the reference's
 * nav2_costmap_2d::Costmap2D. */
typedef struct {
    int32_t nx, ny, nz;
    double origin_x, origin_y, origin_z;
    double resolution;
    const uint8_t *cells;
} fso_grid;

/* Parameters of the ray fan (DEP/src/CostCalculator.cpp:7-9,19; DEP/params/exploration.yaml:8-10). */
typedef struct {
    double max_camera_depth;   /* MAX_CAMERA_DEPTH, 2.0 */
    double delta_theta;        /* DELTA_THETA, 0.10 */
    double camera_fov;         /* CAMERA_FOV, 1.04 */
    double robot_radius;       /* getRobotRadius(), 0.60 */
    int32_t n_rays;            /* 0: reference loop `theta <= 2*pi`; >0: exactly that many yaw rays */
    int32_t n_elev;            /* elevation rings (3-D extension); 1 with elev[0]==0 is the reference */
    const double *elev;        /* [n_elev] radians; NULL means {0.0} */
    int32_t obst_min, obst_max;   /* RayTracedCells ranges, inclusive (DEP/include/.../Helpers.hpp:30-37) */
    int32_t trace_min, trace_max;
    int32_t clamp_to_polygon;  /* 1: CostCalculator.cpp:47-48; 0: setMaxArrivalInformation (no clamp) */
    double polygon[4];         /* minx, miny, maxx, maxy */
} fso_ray_params;

/* number of yaw rays the reference loop produces (DEP/src/CostCalculator.cpp:36). */
int32_t fso_num_yaw_rays(double delta_theta, int32_t n_rays_override);
/* accumulated theta list, out[n] (same loop). */
void fso_theta_list(double delta_theta, int32_t n, double *out);

/* App. B of SURVEY.md: nav2_costmap_2d::Costmap2D::worldToMap, extended with z. */
int fso_world_to_map(const fso_grid *g, double wx, double wy, double wz,
                     uint32_t *mx, uint32_t *my, uint32_t *mz);

/* One ray.  Follows getTracedCells + bresenham2D + RayTracedCells::operator()
 * (DEP/src/Helpers.cpp:7-96, DEP/include/.../Helpers.hpp:50-77).  `faithful` != 0 keeps the
 * reference's per-ray cell vector and O(k^2) dedupe scan; 0 counts only.  Returns 0 if a
 * worldToMap fails.  Outputs: traced = cells_.size(), hit = hasHitObstacle(),
 * unknown = getNumUnknown(), all = getCellsSize(). */
int fso_trace_ray(const fso_grid *g, double sx, double sy, double sz,
                  double wx, double wy, double wz, double max_length_cells,
                  int obst_min, int obst_max, int trace_min, int trace_max, int faithful,
                  int32_t *traced, int32_t *hit, int32_t *unknown, int32_t *all,
                  uint32_t *visited_offsets /* NULL or [>= max_length+1] */, int32_t *n_visited);

/* isRobotFootprintInLethal (DEP/src/Helpers.cpp:135-155); off-grid cells count as not lethal. */
int fso_footprint_in_lethal(const fso_grid *g, uint32_t cx, uint32_t cy, uint32_t cz, double radius_in_cells);

/* setArrivalInformationForFrontier for a batch (DEP/src/CostCalculator.cpp:23-121) plus the
 * blacklist branch of assignCosts (DEP/src/FrontierCostsManager.cpp:77-86).
 * goal_xyz [n][3]; frontier_size/blacklisted/achievable_in may be NULL (size 0 / 0 / 1).
 * ray_counts may be NULL, else [n][n_elev][n_yaw].  Returns 0, or <0 on invalid parameters
 * (rays < window: the reference would build a negative-size vector, CostCalculator.cpp:90). */
int fso_arrival_information(const fso_grid *g, const fso_ray_params *p, int32_t n,
                            const double *goal_xyz, const int32_t *frontier_size,
                            const uint8_t *blacklisted, const uint8_t *achievable_in,
                            double min_arrival_info_gt, int faithful, int n_threads,
                            int32_t *ray_counts, int32_t *arrival, int32_t *argmax,
                            double *yaw, uint8_t *achievable, int32_t *status);

/* setMaxArrivalInformation (DEP/src/CostCalculator.cpp:123-191).  Returns maxValue (0 if a
 * worldToMap failed, limits untouched); *max_gt = factor_max*maxValue, *min_gt = factor_min*max_gt. */
double fso_max_arrival_information(const fso_grid *g, const fso_ray_params *p,
                                   double factor_max, double factor_min,
                                   double *max_gt, double *min_gt);

/* ------------------------------------------------------------------ Fisher information */

/* Eigen-float32-order restatement of computeInformationOfPointLocal(p, Q=I)
 * (FIP/src/fisher_information/FisherInformationHelpers.cpp:71-96,114-123). */
float fso_information_of_point_local(const float p[3]);
/* Same through the 3-argument Jacobian: p_est = T^-1 p_w (FisherInformationHelpers.cpp:50-69,106-112).
 * pose7 = x y z qx qy qz qw (double, as geometry_msgs::Pose). */
float fso_information_of_point_local_world(const double pose7[7], const float p_w[3]);
/* 6x6 F = J^T J for one camera-frame point, float64 (SURVEY.md App. C.3 block form checked against it). */
void fso_fim_point_local_f64(const double p[3], double F[36]);

/* getVoxelCoordinate (FIP/include/.../FisherInfoManager.hpp:108-123): float key + integer lattice index. */
void fso_voxel_coordinate(float x, float y, float z, float key[3], int32_t idx[3]);
/* getFactorFromNum(num, 0.8f) (FisherInfoManager.hpp:102-106). */
float fso_factor_from_num(int32_t num);

/* Lookup table = the reference's .dat: records {float key[3]; float value} (16 B, no header). */
typedef struct fso_table fso_table;
/* generateLookupTable (FIP/src/.../FisherInfoManager.cpp:117-229) with the reference's float loops.
 * Returns a new table; *n_records (incl. the trailing (0,0,0) record). */
fso_table *fso_table_generate(float minX, float maxX, float minY, float maxY, float minZ, float maxZ);
/* loadLookupTable (FisherInfoManager.cpp:231-262): later duplicates overwrite. */
fso_table *fso_table_from_records(const float *records /* [n][4] */, int64_t n);
int64_t fso_table_num_records(const fso_table *t);           /* records as written to the file */
void fso_table_copy_records(const fso_table *t, float *out);  /* [n][4] */
int64_t fso_table_num_entries(const fso_table *t);           /* distinct keys after load */
float fso_table_find(const fso_table *t, const float key[3]); /* NaN on miss */
void fso_table_free(fso_table *t);

/* Visibility (build-defined: the reference delegates to an un-vendored service,
 * request fields at FIP/src/.../FisherInfoManager.cpp:60-65). */
typedef struct {
    double max_dist;    /* 14.0 */
    double max_angle;   /* 1.0 rad from +x; >= pi disables */
} fso_vis_params;

/* Pose -> float32 rotation (Eigen::Quaternionf::toRotationMatrix order) and translation,
 * getTransformFromPose (FisherInformationHelpers.cpp:16-26). R row-major [9]. */
void fso_pose_to_rt(const double pose7[7], float R[9], float t[3]);
/* yaw -> pose7 quaternion as nav2_util::geometry_utils::orientationAroundZAxis
 * (DEP/include/.../util/GeometryUtils.hpp:112-124, DEP/src/Frontier.cpp:50-56). */
void fso_yaw_to_quat(double yaw, double q_xyzw[4]);
/* camera-frame point p = R^T (w - t), the build-defined float32 op order. */
void fso_world_to_camera(const float R[9], const float t[3], const float w[3], float p[3]);
int fso_is_visible(const float p[3], const fso_vis_params *v);

/* isPoseSafe's accumulation for a batch of poses (FIP/src/.../FisherInfoManager.cpp:83-100,287-324)
 * over the landmarks visible under `vis`.
 *   info_ref   [n]      float32 sequential sum exactly as the reference accumulates it
 *   info_f64   [n]      order-independent float64 form  sum_v info_v * S(m_v)   (may be NULL)
 *   fim_f64    [n][36]  full 6x6 sum of F(p) over visible landmarks, float64    (may be NULL)
 *   trace_f64  [n]      its trace; logdet_f64 [n] log det (-inf if not PD)       (may be NULL)
 *   n_visible, n_voxels [n]  landmarks passing the predicate / distinct voxels hit (occupied_voxel_count_) */
int fso_pose_information(const fso_table *tab, const float *landmarks_xyz, int32_t m,
                         int32_t n, const double *pose7, const fso_vis_params *vis, int n_threads,
                         float *info_ref, double *info_f64, double *fim_f64,
                         double *trace_f64, double *logdet_f64,
                         int32_t *n_visible, int32_t *n_voxels);

/* computeInformationOfPointGlobal(p_c, p_w, T, Q = I) (FisherInformationHelpers.cpp:28-48,98-104):
 * J = df_dp(p_est) * R^-1 * [ I | -[p_w]x ], returns trace(J^T J), float32. */
float fso_information_of_point_global_world(const double pose7[7], const float p_w[3]);
/* computeInformationFrontierPair (FisherInformationHelpers.cpp:125-143) with isInside/onLeft
 * (FIP/include/.../FisherInformationHelpers.hpp:20-43): float32 running sum of the local trace at est_pose over
 * the landmarks whose (x, y) lies strictly inside the triangle tri = {ax,ay,bx,by,cx,cy}. */
float fso_information_frontier_pair(const float *landmarks_xyz, int32_t m, const double est_pose7[7], const double tri[6]);

/* ------------------------------------------------------------------ U1 utility (next-row §8f.1) */
/* assignCosts' utility loop + recomputeNormalizationFactors
 * (DEP/src/FrontierCostsManager.cpp:118,126-205; DEP/src/CostCalculator.cpp:512-520).
 * Returns 0, -2 if a utility leaves [0,1] (the reference throws, :148-149,173-174). */
int fso_u1_costs(int32_t n, const double *arrival, const uint8_t *achievable, const uint8_t *blacklisted,
                 const double *path_length, const double *path_heading,
                 double alpha, double beta, double max_vx, double max_wz, double max_arrival_gt,
                 double *weighted_cost, double *arrival_utility, double *distance_utility);

/* ------------------------------------------------------------------ key-frame pose information (§8a row a24) */
/* computeInformationForPose (DEP/include/.../deprecated/util.hpp:840-916; dead code in the reference, call site in the
 * commented block DEP/src/CostCalculator.cpp:326-365 with max_depth 2.0, hfov 1.089, max_depth_error 0.5, Q = 0.01 I,
 * neighbours within 4.5 m).  Helpers restated alongside: isPointInsideTriangle (:49-66), quatToEuler (:77-88; tf2
 * Matrix3x3::getRPY, third party: parity unpinned there), getVerticesOfFrustum2D (:101-119), getVerticesToCheck (:134-156),
 * frustumOverlap (:172-185), getNodesInRadius (:616-632), the float32 affine Jacobian/FIM/trace (:687-759). */
typedef struct {
    double max_depth;        /* 2.0 */
    double hfov;             /* 1.089 */
    double max_depth_error;  /* 0.5 */
    float q_diag;            /* Q = q_diag * I (0.01f) */
    double radius;           /* getNodesInRadius (4.5); < 0: every key-frame is a neighbour */
} fso_kf_params;
double fso_quat_to_yaw(const double q_xyzw[4]);                       /* quatToEuler(...)[2] */
void fso_frustum_vertices_2d(const double pose7[7], double max_depth, double hfov, double tri[6]);
int fso_point_in_triangle(double px, double py, const double tri[6]);
int fso_frustum_overlap(const double cur_pose7[7], const double check_pose7[7], double max_depth, double hfov, double err);
/* computeInformationOfPoint(p_c, p_w, T_w_c_est, Q) of the affine namespace, Q = q_diag * I, float32 */
float fso_information_of_point_affine(const double pose7[7], const float p_w[3], float q_diag);
/* For each of n poses: key-frames k (poses kf_pose7[n_kf][7], points points_xyz[kf_offsets[k] .. kf_offsets[k+1]))
 * in list order; per costmap cell the first point's value is computed once and added once per point in the cell.
 * info_ref: the reference's sequential float32 sum; info_f64: sum_cell count * value (float64 arithmetic, closed form);
 * n_cells: information_map.size(); n_points: points that passed triangle + worldToMap.  Uses the grid's x/y geometry only. */
int fso_information_for_pose(const fso_grid *g, int32_t n, const double *pose7, int32_t n_kf, const double *kf_pose7,
                             const int32_t *kf_offsets, const float *points_xyz, const fso_kf_params *prm, int n_threads,
                             float *info_ref, double *info_f64, int32_t *n_cells, int32_t *n_points);

#ifdef __cplusplus
}
#endif
#endif
