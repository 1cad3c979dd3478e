/*
 * fso_raycast.c — CPU ORACLE (test infrastructure only; see fso_oracle.h header).
 *
 * Arrival information: the per-candidate 2*pi ray fan, unknown-cell counting, footprint test,
 * sliding FOV window and best yaw.  Restates
 *   DEP/include/frontier_exploration/Helpers.hpp:20-116   (RayTracedCells, sign)
 *   DEP/src/Helpers.cpp:7-96,135-155                       (bresenham2D, getTracedCells, isRobotFootprintInLethal)
 *   DEP/src/CostCalculator.cpp:23-191                      (setArrivalInformationForFrontier, setMaxArrivalInformation)
 * and the nav2_costmap_2d::Costmap2D accessors listed in SURVEY.md App. B (third party, unpinned).
 *
 * 3-D extension (build-defined, SURVEY.md App. A.1 last paragraph): grid [nz][ny][nx], ray
 * directions = yaw fan x elevation rings, 3-axis integer Bresenham with the same dominant-axis /
 * error = abs_da/2 / end+1-visits convention.  With nz = 1, one ring at elevation 0 and z = origin_z
 * every expression below reduces to the 2-D reference expression.
 */
#include "fso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* DEP/src/CostCalculator.cpp:36 — `for (double theta = 0; theta <= (2 * M_PI); theta += DELTA_THETA)`.
 * theta is ACCUMULATED, never i*delta. */
int32_t fso_num_yaw_rays(double delta_theta, int32_t n_rays_override)
{
    if (n_rays_override > 0) return n_rays_override;
    if (!(delta_theta > 0.0)) return 0;
    int32_t n = 0;
    for (double theta = 0; theta <= (2 * M_PI); theta += delta_theta) {
        ++n;
        if (n > (1 << 20)) break;
    }
    return n;
}

void fso_theta_list(double delta_theta, int32_t n, double *out)
{
    double theta = 0;
    for (int32_t i = 0; i < n; ++i) {
        out[i] = theta;
        theta += delta_theta;
    }
}

/* Costmap2D::getSizeInMetersX() = (size_x - 1 + 0.5) * resolution. */
static double size_in_meters(int32_t cells, double res)
{
    return (cells - 1 + 0.5) * res;
}

/* Costmap2D::worldToMap.  `(unsigned)(double)` is undefined in C++ beyond
 * UINT_MAX; the oracle (and the HIP path) define any quotient >= 2^32 as off-map. */
int fso_world_to_map(const fso_grid *g, double wx, double wy, double wz,
                     uint32_t *mx, uint32_t *my, uint32_t *mz)
{
    if (wx < g->origin_x || wy < g->origin_y || wz < g->origin_z) return 0;
    double qx = (wx - g->origin_x) / g->resolution;
    double qy = (wy - g->origin_y) / g->resolution;
    double qz = (wz - g->origin_z) / g->resolution;
    if (!(qx < 4294967296.0) || !(qy < 4294967296.0) || !(qz < 4294967296.0)) return 0;
    *mx = (uint32_t)qx;
    *my = (uint32_t)qy;
    *mz = (uint32_t)qz;
    return (*mx < (uint32_t)g->nx) && (*my < (uint32_t)g->ny) && (*mz < (uint32_t)g->nz);
}

/* DEP/include/.../Helpers.hpp:113-116 — returns -1 for x == 0 as well. */
static int sign_ref(int x)
{
    return x > 0 ? 1 : -1;
}

/* State of one RayTracedCells visitor (DEP/include/.../Helpers.hpp:20-111). */
typedef struct {
    const fso_grid *g;
    int obst_min, obst_max, trace_min, trace_max;
    int faithful;
    /* faithful mode: the reference's std::vector<MapLocation>& cells_ */
    uint32_t *cells_xyz;   /* [cap][3] */
    int32_t n_cells;
    int hit_obstacle;
    int32_t unknown_cells;
    int32_t all_cells;
    uint32_t *visited;     /* optional record of every offset handed to operator() */
    int32_t n_visited;
} visitor;

/* RayTracedCells::operator()(offset) — DEP/include/.../Helpers.hpp:50-77. */
static void visit(visitor *v, uint32_t offset)
{
    const fso_grid *g = v->g;
    if (v->visited) v->visited[v->n_visited] = offset;
    ++v->n_visited;
    if (v->faithful) {
        /* indexToCells (App. B) extended with z, then the linear presence scan of :55-59 */
        uint32_t plane = (uint32_t)g->nx * (uint32_t)g->ny;
        uint32_t z = offset / plane;
        uint32_t rem = offset - z * plane;
        uint32_t y = rem / (uint32_t)g->nx;
        uint32_t x = rem - y * (uint32_t)g->nx;
        int present = 0;
        for (int32_t i = 0; i < v->n_cells; ++i) {
            if (v->cells_xyz[3 * i] == x && v->cells_xyz[3 * i + 1] == y && v->cells_xyz[3 * i + 2] == z)
                present = 1;
        }
        if (present) return;
        ++v->all_cells;
        int cost = (int)g->cells[offset];
        if (cost <= v->trace_max && cost >= v->trace_min && !v->hit_obstacle) {
            v->cells_xyz[3 * v->n_cells] = x;
            v->cells_xyz[3 * v->n_cells + 1] = y;
            v->cells_xyz[3 * v->n_cells + 2] = z;
            ++v->n_cells;
        }
        if (cost >= v->obst_min && cost <= v->obst_max) v->hit_obstacle = 1;
        if (cost == 255) ++v->unknown_cells;
    } else {
        /* a Bresenham walk never revisits a cell (it always advances along the dominant axis),
         * so the dedupe scan is a no-op and only the counters matter */
        ++v->all_cells;
        int cost = (int)g->cells[offset];
        if (cost <= v->trace_max && cost >= v->trace_min && !v->hit_obstacle) ++v->n_cells;
        if (cost >= v->obst_min && cost <= v->obst_max) v->hit_obstacle = 1;
        if (cost == 255) ++v->unknown_cells;
    }
}

/* bresenham2D (DEP/src/Helpers.cpp:7-30) with a second minor axis for the 3-D extension.
 * resolution_cut_factor == 1 at the only call site (Helpers.cpp:36), so every step visits. */
static void bresenham(visitor *at, uint32_t abs_da, uint32_t abs_db, uint32_t abs_dc,
                      int error_b, int error_c, int offset_a, int offset_b, int offset_c,
                      uint32_t offset, uint32_t max_length)
{
    uint32_t end = max_length < abs_da ? max_length : abs_da;
    for (uint32_t i = 0; i < end; ++i) {
        visit(at, offset);
        offset += (uint32_t)offset_a;
        error_b += (int)abs_db;
        if ((uint32_t)error_b >= abs_da) {
            offset += (uint32_t)offset_b;
            error_b -= (int)abs_da;
        }
        error_c += (int)abs_dc;
        if ((uint32_t)error_c >= abs_da) {
            offset += (uint32_t)offset_c;
            error_c -= (int)abs_da;
        }
    }
    visit(at, offset);
}

/* getTracedCells (DEP/src/Helpers.cpp:32-96); min_length = 0 so the start cell is (x0,y0). */
static int traced_cells(visitor *at, double sx, double sy, double sz,
                        double wx, double wy, double wz, double max_length)
{
    const fso_grid *g = at->g;
    uint32_t x1, y1, z1, x0, y0, z0;
    /* Helpers.cpp:40 — end point first, then start point */
    if (!fso_world_to_map(g, wx, wy, wz, &x1, &y1, &z1) || !fso_world_to_map(g, sx, sy, sz, &x0, &y0, &z0))
        return 0;

    int dx = (int)(x1 - x0);
    int dy = (int)(y1 - y0);
    int dz = (int)(z1 - z0);
    /* Helpers.cpp:49 — std::hypot(dx_full, dy_full); 3-D: sqrt of the exact integer sum */
    double dist = (dz == 0) ? hypot((double)dx, (double)dy)
                            : sqrt((double)((int64_t)dx * dx + (int64_t)dy * dy + (int64_t)dz * dz));

    uint32_t nx = (uint32_t)g->nx, ny = (uint32_t)g->ny;
    uint32_t offset = (z0 * ny + y0) * nx + x0;

    uint32_t abs_dx = (uint32_t)abs(dx);
    uint32_t abs_dy = (uint32_t)abs(dy);
    uint32_t abs_dz = (uint32_t)abs(dz);

    int offset_dx = sign_ref(dx);
    int offset_dy = sign_ref(dy) * (int)nx;
    int offset_dz = sign_ref(dz) * (int)(nx * ny);

    /* Helpers.cpp:78 */
    double scale = (dist == 0.0) ? 1.0 : fmin(1.0, max_length / dist);

    if (abs_dx >= abs_dy && abs_dx >= abs_dz) {
        /* Helpers.cpp:81-87 */
        int err = (int)(abs_dx / 2);
        bresenham(at, abs_dx, abs_dy, abs_dz, err, err, offset_dx, offset_dy, offset_dz, offset,
                  (uint32_t)(scale * abs_dx));
    } else if (abs_dy >= abs_dz) {
        /* Helpers.cpp:88-94 */
        int err = (int)(abs_dy / 2);
        bresenham(at, abs_dy, abs_dx, abs_dz, err, err, offset_dy, offset_dx, offset_dz, offset,
                  (uint32_t)(scale * abs_dy));
    } else {
        int err = (int)(abs_dz / 2);
        bresenham(at, abs_dz, abs_dx, abs_dy, err, err, offset_dz, offset_dx, offset_dy, offset,
                  (uint32_t)(scale * abs_dz));
    }
    return 1;
}

int fso_trace_ray(const fso_grid *g, double sx, double sy, double sz,
                  double wx, double wy, double wz, double max_length_cells,
                  int obst_min, int obst_max, int trace_min, int trace_max, int faithful,
                  int32_t *traced, int32_t *hit, int32_t *unknown, int32_t *all,
                  uint32_t *visited_offsets, int32_t *n_visited)
{
    visitor v;
    memset(&v, 0, sizeof v);
    v.g = g;
    v.obst_min = obst_min; v.obst_max = obst_max;
    v.trace_min = trace_min; v.trace_max = trace_max;
    v.faithful = faithful;
    v.visited = visited_offsets;
    uint32_t *cells = NULL;
    if (faithful) {
        size_t cap = (size_t)(max_length_cells > 0 ? max_length_cells : 0) + 4;
        cells = (uint32_t *)malloc(cap * 3 * sizeof(uint32_t));
        v.cells_xyz = cells;
    }
    int ok = traced_cells(&v, sx, sy, sz, wx, wy, wz, max_length_cells);
    if (traced) *traced = v.n_cells;
    if (hit) *hit = v.hit_obstacle;
    if (unknown) *unknown = v.unknown_cells;
    if (all) *all = v.all_cells;
    if (n_visited) *n_visited = v.n_visited;
    free(cells);
    return ok;
}

/* isRobotFootprintInLethal (DEP/src/Helpers.cpp:135-155).  The reference has no bounds check
 * (unsigned wrap -> out-of-bounds read near the border); the oracle treats off-grid cells as not
 * lethal.  Documented deviation (last paragraph). */
int fso_footprint_in_lethal(const fso_grid *g, uint32_t cx, uint32_t cy, uint32_t cz, double radius_in_cells)
{
    for (int dx = (int)(-radius_in_cells); dx <= radius_in_cells; ++dx) {
        for (int dy = (int)(-radius_in_cells); dy <= radius_in_cells; ++dy) {
            if (dx * dx + dy * dy <= radius_in_cells * radius_in_cells) {
                uint32_t x = cx + (uint32_t)dx;
                uint32_t y = cy + (uint32_t)dy;
                if (x >= (uint32_t)g->nx || y >= (uint32_t)g->ny) continue;
                uint32_t cost = g->cells[((size_t)cz * g->ny + y) * g->nx + x];
                if (cost == 254) return 1;
            }
        }
    }
    return 0;
}

static double dmax(double a, double b) { return a < b ? b : a; }   /* std::max */
static double dmin(double a, double b) { return b < a ? b : a; }   /* std::min */

/* The ray fan of one start point: counts[e][i].  Returns 0 if any worldToMap failed. */
static int ray_fan(const fso_grid *g, const fso_ray_params *p, int32_t n_yaw, const double *theta,
                   double sx, double sy, double sz, int faithful, uint32_t *scratch_cells,
                   int32_t *counts)
{
    /* DEP/src/CostCalculator.cpp:28 — unsigned int max_length = MAX_CAMERA_DEPTH / resolution */
    unsigned int max_length = (unsigned int)(p->max_camera_depth / g->resolution);
    const double zero = 0.0;
    const double *elev = p->elev ? p->elev : &zero;
    int32_t n_elev = p->elev ? p->n_elev : 1;
    for (int32_t e = 0; e < n_elev; ++e) {
        double d_h = p->max_camera_depth * cos(elev[e]);   /* horizontal reach; elev 0 -> D exactly */
        double d_z = p->max_camera_depth * sin(elev[e]);   /* elev 0 -> 0.0 */
        for (int32_t i = 0; i < n_yaw; ++i) {
            /* CostCalculator.cpp:42-43 */
            double wx = sx + (d_h * cos(theta[i]));
            double wy = sy + (d_h * sin(theta[i]));
            double wz = sz + d_z;
            if (p->clamp_to_polygon) {
                /* CostCalculator.cpp:47-48 */
                wx = dmax(p->polygon[0], dmax(g->origin_x, dmin(p->polygon[2], dmin(g->origin_x + size_in_meters(g->nx, g->resolution), wx))));
                wy = dmax(p->polygon[1], dmax(g->origin_y, dmin(p->polygon[3], dmin(g->origin_y + size_in_meters(g->ny, g->resolution), wy))));
                /* 3-D extension: z clamped to the map only (the polygon is a 2-D bbox) */
                wz = dmax(g->origin_z, dmin(g->origin_z + size_in_meters(g->nz, g->resolution), wz));
            }
            visitor v;
            memset(&v, 0, sizeof v);
            v.g = g;
            v.obst_min = p->obst_min; v.obst_max = p->obst_max;
            v.trace_min = p->trace_min; v.trace_max = p->trace_max;
            v.faithful = faithful;
            v.cells_xyz = scratch_cells;
            if (!traced_cells(&v, sx, sy, sz, wx, wy, wz, (double)max_length)) return 0;
            counts[e * n_yaw + i] = v.n_cells;   /* CostCalculator.cpp:57-58: getCells().size() */
        }
    }
    return 1;
}

/* CostCalculator.cpp:87-107 — window sum (no wrap), first maximum. Rings are summed per yaw
 * (3-D extension). */
static void window_max(const int32_t *counts, int32_t n_yaw, int32_t n_elev, int32_t k,
                       int32_t *max_value, int32_t *max_index)
{
    int32_t best = 0, best_i = 0;
    for (int32_t i = 0; i < n_yaw - k + 1; ++i) {
        int32_t s = 0;
        for (int32_t j = 0; j < k; ++j)
            for (int32_t e = 0; e < n_elev; ++e)
                s += counts[e * n_yaw + i + j];
        if (i == 0 || s > best) { best = s; best_i = i; }   /* :99-107: strict >, first maximum */
    }
    *max_value = best;
    *max_index = best_i;
}

int fso_arrival_information(const fso_grid *g, const fso_ray_params *p, int32_t n,
                            const double *goal_xyz, const int32_t *frontier_size,
                            const uint8_t *blacklisted, const uint8_t *achievable_in,
                            double min_arrival_info_gt, int faithful, int n_threads,
                            int32_t *ray_counts, int32_t *arrival, int32_t *argmax,
                            double *yaw, uint8_t *achievable, int32_t *status)
{
    int32_t n_yaw = fso_num_yaw_rays(p->delta_theta, p->n_rays);
    int32_t n_elev = p->elev ? p->n_elev : 1;
    /* CostCalculator.cpp:87 — static_cast<int>(CAMERA_FOV / DELTA_THETA) */
    int32_t k = (int32_t)(p->camera_fov / p->delta_theta);
    if (n_yaw <= 0 || n_elev <= 0 || k <= 0 || n_yaw < k) return -1;

    double *theta = (double *)malloc(sizeof(double) * (size_t)n_yaw);
    fso_theta_list(p->delta_theta, n_yaw, theta);
    unsigned int max_length = (unsigned int)(p->max_camera_depth / g->resolution);
    int32_t per = n_yaw * n_elev;
    (void)n_threads;

#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
    {
        int32_t *counts = (int32_t *)malloc(sizeof(int32_t) * (size_t)per);
        uint32_t *scratch = (uint32_t *)malloc(sizeof(uint32_t) * 3 * ((size_t)max_length + 4));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int32_t c = 0; c < n; ++c) {
            uint8_t ach = achievable_in ? achievable_in[c] : 1;
            if (ray_counts) memset(ray_counts + (size_t)c * per, 0, sizeof(int32_t) * (size_t)per);
            /* FrontierCostsManager.cpp:77-86 */
            if (blacklisted && blacklisted[c]) {
                arrival[c] = 0; argmax[c] = 0; yaw[c] = 0.0; achievable[c] = ach;
                status[c] = FSO_STATUS_BLACKLISTED;
                continue;
            }
            double sx = goal_xyz[3 * c], sy = goal_xyz[3 * c + 1], sz = goal_xyz[3 * c + 2];
            if (!ray_fan(g, p, n_yaw, theta, sx, sy, sz, faithful, scratch, counts)) {
                /* CostCalculator.cpp:50-55: arrival 0, orientation 0, return (achievability untouched) */
                arrival[c] = 0; argmax[c] = 0; yaw[c] = 0.0; achievable[c] = ach;
                status[c] = FSO_STATUS_OFF_MAP;
                continue;
            }
            if (ray_counts) memcpy(ray_counts + (size_t)c * per, counts, sizeof(int32_t) * (size_t)per);

            /* CostCalculator.cpp:75-82 */
            uint32_t sxm = 0, sym = 0, szm = 0;
            fso_world_to_map(g, sx, sy, sz, &sxm, &sym, &szm);
            double penalty = (double)fso_footprint_in_lethal(g, sxm, sym, szm, ceil(p->robot_radius / g->resolution));
            int32_t fsize = frontier_size ? frontier_size[c] : 0;
            if (1.0 - penalty == 0.0 && fsize < 10.0) ach = 0;

            int32_t max_value, max_index;
            window_max(counts, n_yaw, n_elev, k, &max_value, &max_index);
            arrival[c] = max_value;                         /* :112 */
            if ((double)max_value < min_arrival_info_gt) ach = 0;   /* :114-118 */
            argmax[c] = max_index;
            yaw[c] = (max_index * p->delta_theta) + (p->camera_fov / 2);   /* :119 */
            achievable[c] = ach;
            status[c] = FSO_STATUS_OK;
        }
        free(counts);
        free(scratch);
    }
    free(theta);
    return 0;
}

double fso_max_arrival_information(const fso_grid *g, const fso_ray_params *p,
                                   double factor_max, double factor_min,
                                   double *max_gt, double *min_gt)
{
    /* CostCalculator.cpp:123-191: fan from world (0,0) [z: origin plane of the candidate grid],
     * visitor (260,260,0,255), no endpoint clamping. */
    fso_ray_params q = *p;
    q.obst_min = 260; q.obst_max = 260; q.trace_min = 0; q.trace_max = 255;
    q.clamp_to_polygon = 0;
    int32_t n_yaw = fso_num_yaw_rays(q.delta_theta, q.n_rays);
    int32_t n_elev = q.elev ? q.n_elev : 1;
    int32_t k = (int32_t)(q.camera_fov / q.delta_theta);
    if (n_yaw <= 0 || k <= 0 || n_yaw < k) return 0.0;
    double *theta = (double *)malloc(sizeof(double) * (size_t)n_yaw);
    fso_theta_list(q.delta_theta, n_yaw, theta);
    int32_t *counts = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_yaw * n_elev));
    /* z of the calibration fan: the middle of the grid's z extent for nz > 1, origin_z for nz == 1 */
    double sz = (g->nz > 1) ? g->origin_z + 0.5 * g->nz * g->resolution : g->origin_z;
    int ok = ray_fan(g, &q, n_yaw, theta, 0.0, 0.0, sz, 0, NULL, counts);
    double ret = 0.0;
    if (ok) {
        int32_t max_value, max_index;
        window_max(counts, n_yaw, n_elev, k, &max_value, &max_index);
        *max_gt = max_value * factor_max;      /* :186 (1.2) */
        *min_gt = factor_min * (*max_gt);      /* :188 (0.70) */
        ret = (double)max_value;
    }
    free(counts);
    free(theta);
    return ret;
}

/* assignCosts' U1 block — DEP/src/FrontierCostsManager.cpp:118,126-205 with
 * recomputeNormalizationFactors/reset (DEP/src/CostCalculator.cpp:512-520,
 * DEP/include/.../CostCalculator.hpp:110-116).  getMaxArrivalInformation() returns
 * max_arrival_info_gt_ (CostCalculator.hpp:105-108). */
int fso_u1_costs(int32_t n, const double *arrival, const uint8_t *achievable, const uint8_t *blacklisted,
                 const double *path_length, const double *path_heading,
                 double alpha, double beta, double max_vx, double max_wz, double max_arrival_gt,
                 double *weighted_cost, double *arrival_utility, double *distance_utility)
{
    const double DMAX = 1.7976931348623157e308;
    double min_dist = DMAX, max_dist = -1.0, min_info = DMAX, max_info = -1.0;   /* reset() */
    for (int32_t i = 0; i < n; ++i) {
        if (blacklisted && blacklisted[i]) continue;       /* :77-86 `continue` skips :118 */
        if (!achievable[i]) continue;                       /* CostCalculator.cpp:514-515 */
        min_dist = dmin(min_dist, path_length[i]);
        max_dist = dmax(max_dist, path_length[i]);
        min_info = dmin(min_info, arrival[i]);
        max_info = dmax(max_info, arrival[i]);
    }
    (void)max_info;
    for (int32_t i = 0; i < n; ++i) {
        if (blacklisted && blacklisted[i]) {
            /* :84 weighted cost max(); blacklisted frontiers keep is_achievable, so the U1 loop
             * below would run on them with path_length = DBL_MAX; the reference's utility check
             * then throws unless they are also unachievable.  The oracle reports them as
             * max-cost / sentinel utilities like the unachievable branch. */
            weighted_cost[i] = DMAX;
            arrival_utility[i] = -69.8;
            distance_utility[i] = -1.8;
            continue;
        }
        if (!achievable[i]) {                               /* :129-135 */
            weighted_cost[i] = DMAX;
            arrival_utility[i] = -69.8;
            distance_utility[i] = -1.8;
            continue;
        }
        double au;
        if ((double)(max_arrival_gt - min_info) == 0.0) au = 0.0;            /* :139-140 */
        else au = (double)arrival[i] / (double)max_arrival_gt;               /* :145-146 */
        if (au > 1.0) return -2;                                             /* :148-149 */
        double pu;
        if ((double)((max_dist / max_vx + M_PI / max_wz) - (min_dist / max_vx + 0.0 / max_wz) == 0.0))   /* :152 */
            pu = 1.0;
        else
            pu = (double)(path_length[i] / max_vx + path_heading[i] / max_wz) /
                 (double)(max_dist / max_vx + M_PI / max_wz);                /* :158-159 */
        pu = 1.0 - pu;                                                        /* :160 */
        if (au > 1.0 || au < 0.0 || pu > 1.0 || pu < 0.0) return -2;         /* :173-174 */
        double utility = (alpha * au) + ((1.0 - alpha) * pu);                /* :176-177 */
        if (utility == 0.0) utility = 1e-16;                                  /* :178-182 */
        weighted_cost[i] = 1 / (beta * utility);                              /* :198 */
        arrival_utility[i] = au;
        distance_utility[i] = pu;
    }
    return 0;
}
