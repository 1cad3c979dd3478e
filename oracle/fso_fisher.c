/*
 * fso_fisher.c — CPU ORACLE (test infrastructure only; see fso_oracle.h header).
 *
 * Landmark Fisher information: per-point Jacobian / FIM trace, the voxel lookup table
 * (generate / load / query) and isPoseSafe's accumulation with the per-voxel crowding discount.
 * Restates
 *   FIP/src/fisher_information/FisherInformationHelpers.cpp:7-123
 *   FIP/include/fisher_information_plugins/fisher_information/FisherInfoManager.hpp:23-58,102-123
 *   FIP/src/fisher_information/FisherInfoManager.cpp:83-100,117-324
 *   DEP/src/fisher_information/GenerateLookupMain.cpp:9 (table bounds)
 *   DEP/include/frontier_exploration/util/GeometryUtils.hpp:112-124 (two-point pose)
 * Eigen3 float32 products are restated as plain k = 0,1,2 loops (third party, unpinned; the
 * difference in summation order is O(1e-7) relative, far inside the 1e-4 budget).
 *
 * Build-defined pieces (SURVEY.md App. A.3 — the reference delegates visibility to the
 * un-vendored ORB-SLAM3 GetLandmarksInView server, which also hands points over already in the
 * camera frame): world->camera transform  p = R^T (w - t)  in float32 with the op order of
 * fso_world_to_camera, and the visibility predicate of fso_is_visible.  The HIP path implements
 * exactly these two functions so that every integer decision (visible?, voxel key, rank in voxel)
 * is bit-identical; only float sums may differ in rounding.
 */
#include "fso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* FIP/include/.../FisherInfoManager.hpp:25-30 */
static const float step_min = 0.09f;
static const float step_max = 0.3f;
static const float subSampleVoxelUntil_m = -1.0f;

/* ------------------------------------------------------------------ per-point information */

/* getSkewMatrix — FisherInformationHelpers.cpp:7-14 */
static void skew_f(const float v[3], float S[9])
{
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}

/* computeJacobianForPointLocal(p) + computeFIM(J, I) + trace — FisherInformationHelpers.cpp:71-96,114-123 */
float fso_information_of_point_local(const float p[3])
{
    /* :74 — const float n = p.norm() */
    const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    /* :75-76 — df_dpc = (1/n) I - (1/(n*n*n)) p p^T */
    const float a = 1 / n;
    const float b = 1 / (n * n * n);
    float A[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = a * (i == j ? 1.0f : 0.0f) - (b * p[i]) * p[j];
    /* :81-84 — rightMat = [ -I | skew(p) ] */
    float S[9];
    skew_f(p, S);
    float Rm[18];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Rm[6 * i + j] = (float)(-1.0) * (i == j ? 1.0f : 0.0f);
            Rm[6 * i + 3 + j] = S[3 * i + j];
        }
    /* :88 — jacobian = df_dpc * dpc_dtwc  (3x3 * 3x6) */
    float J[18];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * Rm[6 * k + j];
            J[6 * i + j] = s;
        }
    /* :95 — J^T * Q.inverse() * J with Q = I; :122 trace */
    float tr = 0.0f;
    for (int j = 0; j < 6; ++j) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += J[6 * k + j] * J[6 * k + j];
        tr += s;
    }
    return tr;
}

void fso_fim_point_local_f64(const double p[3], double F[36])
{
    const double n = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    double A[9], Rm[18], J[18];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = (i == j ? 1.0 / n : 0.0) - p[i] * p[j] / (n * n * n);
    const double S[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Rm[6 * i + j] = (i == j ? -1.0 : 0.0);
            Rm[6 * i + 3 + j] = S[3 * i + j];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * Rm[6 * k + j];
            J[6 * i + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += J[6 * k + i] * J[6 * k + j];
            F[6 * i + j] = s;
        }
}

/* getTransformFromPose — FisherInformationHelpers.cpp:16-26: float32 translation and
 * Eigen::Quaternionf(w,x,y,z) -> rotation matrix (Eigen's toRotationMatrix operation order). */
void fso_pose_to_rt(const double pose7[7], float R[9], float t[3])
{
    t[0] = (float)pose7[0]; t[1] = (float)pose7[1]; t[2] = (float)pose7[2];
    const float x = (float)pose7[3], y = (float)pose7[4], z = (float)pose7[5], w = (float)pose7[6];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
    R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
}

/* nav2_util::geometry_utils::orientationAroundZAxis(theta) = tf2::Quaternion::setRPY(0,0,theta)
 * (third party; call sites DEP/.../GeometryUtils.hpp:123, DEP/src/Frontier.cpp:54). */
void fso_yaw_to_quat(double yaw, double q_xyzw[4])
{
    double halfYaw = yaw * 0.5;
    q_xyzw[0] = 0.0;
    q_xyzw[1] = 0.0;
    q_xyzw[2] = sin(halfYaw);
    q_xyzw[3] = cos(halfYaw);
}

/* p = R^T (w - t), float32, fixed operation order (build-defined; DESIGN.md "FIM kernel").
 * The reference's local Jacobian does T_w_c.inverse() * p_w (FisherInformationHelpers.cpp:53);
 * Eigen's Affine inverse uses a general 3x3 inverse rather than the transpose — an O(1e-7)
 * relative difference, documented as a deviation. */
void fso_world_to_camera(const float R[9], const float t[3], const float w[3], float p[3])
{
    const float dx = w[0] - t[0], dy = w[1] - t[1], dz = w[2] - t[2];
    p[0] = fmaf(R[0], dx, fmaf(R[3], dy, R[6] * dz));
    p[1] = fmaf(R[1], dx, fmaf(R[4], dy, R[7] * dz));
    p[2] = fmaf(R[2], dx, fmaf(R[5], dy, R[8] * dz));
}

/* Visibility: |p| <= max_dist and angle(p, +x) <= max_angle, evaluated
 * without sqrt/acos so that CPU and GPU decide identically:
 *   n2 = fma(px,px, fma(py,py, pz*pz)) <= (float)(max_dist^2)
 *   c = (float)cos(max_angle), c2 = c*c
 *   c >= 0:  px >= 0 && px*px >= c2*n2        c < 0:  px >= 0 || px*px <= c2*n2
 * max_angle >= pi disables the cone (the reference's request sends 4.0, FisherInfoManager.cpp:63). */
int fso_is_visible(const float p[3], const fso_vis_params *v)
{
    const float n2 = fmaf(p[0], p[0], fmaf(p[1], p[1], p[2] * p[2]));
    const float maxd2 = (float)(v->max_dist * v->max_dist);
    if (!(n2 <= maxd2)) return 0;
    if (v->max_angle >= M_PI) return 1;
    const float c = (float)cos(v->max_angle);
    const float c2 = c * c;
    const float xx = p[0] * p[0];
    const float rhs = c2 * n2;
    if (c >= 0.0f) return (p[0] >= 0.0f) && (xx >= rhs);
    return (p[0] >= 0.0f) || (xx <= rhs);
}

float fso_information_of_point_local_world(const double pose7[7], const float p_w[3])
{
    float R[9], t[3], p[3];
    fso_pose_to_rt(pose7, R, t);
    fso_world_to_camera(R, t, p_w, p);
    return fso_information_of_point_local(p);
}

/* computeJacobianForPointGlobal + computeFIM + trace (FisherInformationHelpers.cpp:28-48,93-104). */
float fso_information_of_point_global_world(const double pose7[7], const float p_w[3])
{
    float R[9], t[3], p[3];
    fso_pose_to_rt(pose7, R, t);
    fso_world_to_camera(R, t, p_w, p);                       /* :31 p_est = T^-1 p_w */
    const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const float a = 1 / n, b = 1 / (n * n * n);
    float A[9], L[9], Rm[18], D[18], J[18];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[3 * i + j] = a * (i == j ? 1.0f : 0.0f) - (b * p[i]) * p[j];   /* :35-36 */
            L[3 * i + j] = R[3 * j + i];                                      /* :39 T^-1.rotation() = R^T */
        }
    float S[9];
    skew_f(p_w, S);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Rm[6 * i + j] = (i == j ? 1.0f : 0.0f);                           /* :41 */
            Rm[6 * i + 3 + j] = (float)(-1.0) * S[3 * i + j];                 /* :42 */
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += L[3 * i + k] * Rm[6 * k + j];    /* :43 */
            D[6 * i + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * D[6 * k + j];     /* :45 */
            J[6 * i + j] = s;
        }
    float tr = 0.0f;
    for (int j = 0; j < 6; ++j) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += J[6 * k + j] * J[6 * k + j];
        tr += s;
    }
    return tr;
}

/* onLeft(float x, float y, q, r) — FisherInformationHelpers.hpp:26-29 (Point2D fields are double) */
static int on_left_f(float x, float y, double qx, double qy, double rx, double ry)
{
    return (qx - x) * (ry - y) - (qy - y) * (rx - x) > 0;
}

float fso_information_frontier_pair(const float *lm, int32_t m, const double est_pose7[7], const double tri[6])
{
    float pair_information = 0;                                               /* :129 */
    for (int32_t i = 0; i < m; ++i) {
        const float x = lm[3 * i], y = lm[3 * i + 1];
        if (on_left_f(x, y, tri[0], tri[1], tri[2], tri[3]) && on_left_f(x, y, tri[2], tri[3], tri[4], tri[5]) &&
            on_left_f(x, y, tri[4], tri[5], tri[0], tri[1]))                  /* :135 isInside */
            pair_information += fso_information_of_point_local_world(est_pose7, lm + 3 * i);   /* :139 */
    }
    return pair_information;
}

/* ------------------------------------------------------------------ voxel key, crowding factor */

/* getVoxelCoordinate — FisherInfoManager.hpp:108-123.  subSampleVoxelUntil_m = -1 makes the
 * first branch dead, so corrected_step is always (double)step_max = 0.300000011920929. */
void fso_voxel_coordinate(float x, float y, float z, float key[3], int32_t idx[3])
{
    double corrected_step;
    if (fabsf(x) < subSampleVoxelUntil_m && fabsf(y) < subSampleVoxelUntil_m && fabsf(z) < subSampleVoxelUntil_m)
        corrected_step = step_min;
    else
        corrected_step = step_max;
    const double rx = round(x * (1 / corrected_step));
    const double ry = round(y * (1 / corrected_step));
    const double rz = round(z * (1 / corrected_step));
    key[0] = (float)(rx * corrected_step);
    key[1] = (float)(ry * corrected_step);
    key[2] = (float)(rz * corrected_step);
    if (idx) {
        idx[0] = (int32_t)rx; idx[1] = (int32_t)ry; idx[2] = (int32_t)rz;
    }
}

/* getFactorFromNum(num, 0.8) — FisherInfoManager.hpp:102-106: std::pow(int, float) promotes to
 * double, exp in double, result narrowed to the float return type. */
float fso_factor_from_num(int32_t num)
{
    const float s = 0.8f;
    return (float)exp(1 - pow((double)num, (double)s));
}

/* ------------------------------------------------------------------ lookup table */

/* std::unordered_map<LookupKey, LookupValue> (FisherInfoManager.hpp:31-58,139): open addressing
 * on the three float bit patterns; -0.0f == 0.0f under std::array<float,3>::operator== and
 * std::hash<float> maps both zeros to the same bucket, so zeros are canonicalised. */
typedef struct {
    float key[3];
    double information;
    int32_t pointCount;
    uint32_t version;
    uint8_t used;
} slot_t;

struct fso_table {
    slot_t *slots;
    uint64_t cap;        /* power of two */
    int64_t n_entries;
    float *records;      /* as written to the file: [n_records][4] */
    int64_t n_records, cap_records;
};

static uint32_t fbits(float f)
{
    if (f == 0.0f) f = 0.0f + 0.0f;   /* canonical +0 */
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static uint64_t key_hash(const float k[3])
{
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 3; ++i) {
        h ^= (uint64_t)fbits(k[i]) + 0x9e3779b9ull + (h << 6) + (h >> 2);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    return h;
}

static int key_eq(const float a[3], const float b[3])
{
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
}

static fso_table *table_new(uint64_t cap)
{
    fso_table *t = (fso_table *)calloc(1, sizeof *t);
    t->cap = cap;
    t->slots = (slot_t *)calloc(cap, sizeof(slot_t));
    return t;
}

static slot_t *table_lookup(const fso_table *t, const float k[3], int insert)
{
    uint64_t h = key_hash(k) & (t->cap - 1);
    for (;;) {
        slot_t *s = &t->slots[h];
        if (!s->used) {
            if (!insert) return NULL;
            s->used = 1;
            memcpy(s->key, k, sizeof s->key);
            s->information = 0.0;
            s->pointCount = 0;
            s->version = 0xffffffffu;   /* LookupValue() — FisherInfoManager.hpp:44 */
            ((fso_table *)t)->n_entries++;
            return s;
        }
        if (key_eq(s->key, k)) return s;
        h = (h + 1) & (t->cap - 1);
    }
}

static void table_grow(fso_table *t)
{
    fso_table *b = table_new(t->cap * 2);
    for (uint64_t i = 0; i < t->cap; ++i)
        if (t->slots[i].used) {
            slot_t *s = table_lookup(b, t->slots[i].key, 1);
            *s = t->slots[i];
        }
    free(t->slots);
    t->slots = b->slots;
    t->cap = b->cap;
    free(b);
}

static slot_t *table_insert(fso_table *t, const float k[3])
{
    if ((uint64_t)(t->n_entries + 1) * 2 > t->cap) table_grow(t);
    return table_lookup(t, k, 1);
}

static void push_record(fso_table *t, const float k[3], float v)
{
    if (t->n_records == t->cap_records) {
        t->cap_records = t->cap_records ? t->cap_records * 2 : 1 << 16;
        t->records = (float *)realloc(t->records, sizeof(float) * 4 * (size_t)t->cap_records);
    }
    float *r = t->records + 4 * t->n_records++;
    r[0] = k[0]; r[1] = k[1]; r[2] = k[2]; r[3] = v;
}

/* generateLookupTable — FisherInfoManager.cpp:117-229.  All loop counters are float and are
 * advanced by repeated `+= increment_value`; the bounds are snapped in float (:130-135).  The
 * file content is the sequence of records written at :185-186 plus the trailing (0,0,0) record
 * (:190-197); loading (:245-251) lets that last record overwrite the (0,0,0) key's value. */
fso_table *fso_table_generate(float minX, float maxX, float minY, float maxY, float minZ, float maxZ)
{
    fso_table *t = table_new(1u << 21);
    float max_fi_value = -3.402823466e+38f;
    minX = floorf(minX * (1 / step_max)) * step_max;
    minY = floorf(minY * (1 / step_max)) * step_max;
    minZ = floorf(minZ * (1 / step_max)) * step_max;
    maxX = ceilf(maxX * (1 / step_max)) * step_max;
    maxY = ceilf(maxY * (1 / step_max)) * step_max;
    maxZ = ceilf(maxZ * (1 / step_max)) * step_max;
    float increment_value = step_min;                                   /* :140 */
    for (float counter_x = minX; counter_x <= maxX; counter_x += increment_value) {
        if (counter_x > subSampleVoxelUntil_m + step_max) increment_value = step_max;   /* :144-147 */
        for (float counter_y = minY; counter_y <= maxY; counter_y += increment_value) {
            for (float counter_z = minZ; counter_z <= maxZ; counter_z += increment_value) {
                float key[3];
                fso_voxel_coordinate(counter_x, counter_y, counter_z, key, NULL);   /* :158 */
                if (table_lookup(t, key, 0)) continue;                               /* :165-169 */
                slot_t *s = table_insert(t, key);                                    /* :170 */
                float value = fso_information_of_point_local(key);                   /* :164,176 */
                s->information = value;
                if (isnan(value)) { s->information = NAN; continue; }                /* :177-181 (key stays "existing") */
                if (value > max_fi_value) max_fi_value = value;                      /* :182 */
                push_record(t, key, value);                                          /* :185-186 */
            }
        }
    }
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    push_record(t, zero, max_fi_value);                                              /* :190-197 */
    /* turn the generation-time set into the table a loadLookupTable() of that file yields */
    float *rec = t->records;
    int64_t n = t->n_records;
    t->records = NULL; t->n_records = 0; t->cap_records = 0;
    fso_table *loaded = fso_table_from_records(rec, n);
    free(rec);
    fso_table_free(t);
    return loaded;
}

/* loadLookupTable — FisherInfoManager.cpp:231-262: lookup_table_fi_[key].information = value. */
fso_table *fso_table_from_records(const float *records, int64_t n)
{
    uint64_t cap = 1;
    while (cap < (uint64_t)n * 2 + 16) cap <<= 1;
    fso_table *t = table_new(cap);
    for (int64_t i = 0; i < n; ++i) {
        const float *r = records + 4 * i;
        slot_t *s = table_insert(t, r);
        s->information = r[3];
        push_record(t, r, r[3]);
    }
    return t;
}

int64_t fso_table_num_records(const fso_table *t) { return t->n_records; }
void fso_table_copy_records(const fso_table *t, float *out)
{
    memcpy(out, t->records, sizeof(float) * 4 * (size_t)t->n_records);
}
int64_t fso_table_num_entries(const fso_table *t) { return t->n_entries; }

float fso_table_find(const fso_table *t, const float key[3])
{
    const slot_t *s = table_lookup(t, key, 0);
    return s ? (float)s->information : NAN;
}

void fso_table_free(fso_table *t)
{
    if (!t) return;
    free(t->slots);
    free(t->records);
    free(t);
}

/* ------------------------------------------------------------------ pose information */

/* Cholesky log-determinant of a symmetric 6x6 (float64); -inf if not positive definite. */
static double logdet6(const double F[36])
{
    double L[36];
    memset(L, 0, sizeof L);
    double ld = 0.0;
    for (int j = 0; j < 6; ++j) {
        double d = F[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 1e-6 * F[6 * j + j])) return -INFINITY;   /* same singularity rule as the HIP path */
        double ljj = sqrt(d);
        L[6 * j + j] = ljj;
        ld += 2.0 * log(ljj);
        for (int i = j + 1; i < 6; ++i) {
            double s = F[6 * i + j];
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = s / ljj;
        }
    }
    return ld;
}

/* Private per-thread copy of the mutable part of the table (pointCount / version), so that the
 * batch can be evaluated with OpenMP while each pose still sees the reference's bookkeeping. */
typedef struct {
    int32_t *pointCount;
    uint32_t *version;
} counters_t;

int fso_pose_information(const fso_table *tab, const float *landmarks_xyz, int32_t m,
                         int32_t n, const double *pose7, const fso_vis_params *vis, int n_threads,
                         float *info_ref, double *info_f64, double *fim_f64,
                         double *trace_f64, double *logdet_f64,
                         int32_t *n_visible, int32_t *n_voxels)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
    {
        counters_t cs;
        cs.pointCount = (int32_t *)calloc(tab->cap, sizeof(int32_t));
        cs.version = (uint32_t *)malloc(tab->cap * sizeof(uint32_t));
        memset(cs.version, 0xff, tab->cap * sizeof(uint32_t));
        uint32_t latest_version = 1;                       /* FisherInfoManager.cpp:13 */
        uint64_t *touched = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(m > 0 ? m : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int32_t c = 0; c < n; ++c) {
            float R[9], t[3];
            fso_pose_to_rt(pose7 + 7 * c, R, t);
            ++latest_version;                              /* :47 */
            float total_information = 0.0f;                /* :83 */
            uint32_t occupied_voxel_count = 0;             /* :84 */
            int32_t visible = 0;
            int64_t n_touched = 0;
            double F[36];
            memset(F, 0, sizeof F);
            for (int32_t l = 0; l < m; ++l) {              /* :85 */
                float p[3];
                fso_world_to_camera(R, t, landmarks_xyz + 3 * l, p);
                if (!fso_is_visible(p, vis)) continue;     /* the GetLandmarksInView server's role */
                ++visible;
                /* :87-89 -> getInformationFromLookup(tf2::Vector3&, 0.3, version), :287-324 */
                float key[3];
                fso_voxel_coordinate(p[0], p[1], p[2], key, NULL);
                float info;
                uint64_t h = key_hash(key) & (tab->cap - 1);
                const slot_t *s = NULL;
                for (;;) {
                    const slot_t *q = &tab->slots[h];
                    if (!q->used) break;
                    if (key_eq(q->key, key)) { s = q; break; }
                    h = (h + 1) & (tab->cap - 1);
                }
                if (s) {
                    if (cs.version[h] == latest_version) {
                        cs.pointCount[h]++;                                    /* :296-298 */
                    } else {
                        cs.pointCount[h] = 1;                                  /* :299-304 */
                        cs.version[h] = latest_version;
                        occupied_voxel_count++;
                        touched[n_touched++] = h;
                    }
                    /* :318 — double information * float factor, narrowed to the float return */
                    info = (float)(s->information * fso_factor_from_num(cs.pointCount[h]));
                } else {
                    info = NAN;                                                /* :319-321 */
                }
                if (fim_f64 || trace_f64 || logdet_f64) {
                    /* plain (unit-weight) 6x6 sum over the visible set: the D-opt input */
                    double pd[3] = {p[0], p[1], p[2]}, Fp[36];
                    if (pd[0] != 0.0 || pd[1] != 0.0 || pd[2] != 0.0) {
                        fso_fim_point_local_f64(pd, Fp);
                        for (int i = 0; i < 36; ++i) F[i] += Fp[i];
                    }
                }
                if (isnan(info)) continue;                                     /* :90-94 */
                total_information += info;                                     /* :95 */
            }
            info_ref[c] = total_information;                                   /* :100 */
            if (n_visible) n_visible[c] = visible;
            if (n_voxels) n_voxels[c] = (int32_t)occupied_voxel_count;
            if (info_f64) {
                /* order-independent form: sum_v info_v * S(m_v) */
                double tot = 0.0;
                for (int64_t i = 0; i < n_touched; ++i) {
                    uint64_t hh = touched[i];
                    double iv = tab->slots[hh].information;
                    if (isnan(iv)) continue;
                    double S = 0.0;
                    for (int32_t k = 1; k <= cs.pointCount[hh]; ++k) S += (double)fso_factor_from_num(k);
                    tot += iv * S;
                }
                info_f64[c] = tot;
            }
            if (fim_f64) memcpy(fim_f64 + 36 * (size_t)c, F, sizeof F);
            if (trace_f64) {
                double tr = 0;
                for (int i = 0; i < 6; ++i) tr += F[7 * i];
                trace_f64[c] = tr;
            }
            if (logdet_f64) logdet_f64[c] = (visible >= 3) ? logdet6(F) : -INFINITY;
        }
        free(touched);
        free(cs.pointCount);
        free(cs.version);
    }
    return 0;
}

/* ------------------------------------------------------------------ key-frame pose information (row a24) */

/* quatToEuler (util.hpp:77-88): tf2::Quaternion(x,y,z,w) -> tf2::Matrix3x3 -> getRPY, yaw only.
 * tf2 is third party (not under /root/reference): setRotation and getEulerYPR restated from upstream Humble. */
double fso_quat_to_yaw(const double q[4])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double d = x * x + y * y + z * z + w * w;
    const double s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s;
    const double wy = w * ys, wz = w * zs;
    const double xx = x * xs, xy = x * ys, xz = x * zs;
    const double yy = y * ys, zz = z * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy;
    (void)xx;
    if (fabs(m20) >= 1.0) return 0.0;                                 /* gimbal lock branch: yaw = 0 */
    const double pitch = -asin(m20);
    return atan2(m10 / cos(pitch), m00 / cos(pitch));
}

/* getVerticesOfFrustum2D (util.hpp:101-119) */
void fso_frustum_vertices_2d(const double pose7[7], double max_depth, double hfov, double tri[6])
{
    const double yaw = fso_quat_to_yaw(pose7 + 3);
    tri[0] = pose7[0];
    tri[1] = pose7[1];
    tri[2] = pose7[0] + max_depth * cos(yaw - hfov / 2);
    tri[3] = pose7[1] + max_depth * sin(yaw - hfov / 2);
    tri[4] = pose7[0] + max_depth * cos(yaw + hfov / 2);
    tri[5] = pose7[1] + max_depth * sin(yaw + hfov / 2);
}

/* isPointInsideTriangle (util.hpp:49-66): barycentric, edges inclusive; a degenerate triangle gives inf/NaN -> false */
int fso_point_in_triangle(double px, double py, const double t[6])
{
    const double v0x = t[4] - t[0], v0y = t[5] - t[1];
    const double v1x = t[2] - t[0], v1y = t[3] - t[1];
    const double v2x = px - t[0], v2y = py - t[1];
    const double dot00 = v0x * v0x + v0y * v0y;
    const double dot01 = v0x * v1x + v0y * v1y;
    const double dot02 = v0x * v2x + v0y * v2y;
    const double dot11 = v1x * v1x + v1y * v1y;
    const double dot12 = v1x * v2x + v1y * v2y;
    const double inv_denom = 1.0 / (dot00 * dot11 - dot01 * dot01);
    const double u = (dot11 * dot02 - dot01 * dot12) * inv_denom;
    const double v = (dot00 * dot12 - dot01 * dot02) * inv_denom;
    return (u >= 0.0) && (v >= 0.0) && (u + v <= 1.0);
}

/* getVerticesToCheck (util.hpp:134-156): the three vertices and the three edge midpoints */
static void vertices_to_check(const double pose7[7], double depth, double hfov, double out[12])
{
    double t[6];
    fso_frustum_vertices_2d(pose7, depth, hfov, t);
    memcpy(out, t, sizeof t);
    out[6] = (t[0] + t[2]) / 2;  out[7] = (t[1] + t[3]) / 2;
    out[8] = (t[2] + t[4]) / 2;  out[9] = (t[3] + t[5]) / 2;
    out[10] = (t[4] + t[0]) / 2; out[11] = (t[5] + t[1]) / 2;
}

/* frustumOverlap (util.hpp:172-185) */
int fso_frustum_overlap(const double cur[7], const double chk[7], double max_depth, double hfov, double err)
{
    double t[6], c[12];
    fso_frustum_vertices_2d(cur, max_depth + err, hfov, t);
    vertices_to_check(chk, max_depth + err, hfov, c);
    for (int i = 0; i < 6; ++i)
        if (fso_point_in_triangle(c[2 * i], c[2 * i + 1], t)) return 1;
    return 0;
}

/* affine computeJacobianForPoint + computeFIM + trace (util.hpp:687-759).  Same Jacobian as
 * computeJacobianForPointGlobal; Q^-1 through Eigen's 3x3 cofactor inverse of q*I: (q*q) * (1 / (q*(q*q))). */
float fso_information_of_point_affine(const double pose7[7], const float p_w[3], float q)
{
    float R[9], t[3], p[3];
    fso_pose_to_rt(pose7, R, t);
    fso_world_to_camera(R, t, p_w, p);                       /* :690 */
    const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const float a = 1 / n, b = 1 / (n * n * n);
    float A[9], L[9], Rm[18], D[18], J[18], S[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[3 * i + j] = a * (i == j ? 1.0f : 0.0f) - (b * p[i]) * p[j];   /* :694-695 */
            L[3 * i + j] = R[3 * j + i];                                      /* :698 */
        }
    skew_f(p_w, S);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Rm[6 * i + j] = (i == j ? 1.0f : 0.0f);
            Rm[6 * i + 3 + j] = (float)(-1.0) * S[3 * i + j];                 /* :701 */
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += L[3 * i + k] * Rm[6 * k + j];
            D[6 * i + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * D[6 * k + j];
            J[6 * i + j] = s;
        }
    const float qinv = (q * q) * (1 / (q * (q * q)));                         /* :722 Q.inverse() */
    float tr = 0.0f;
    for (int j = 0; j < 6; ++j) {
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s += (J[6 * k + j] * qinv) * J[6 * k + j];
        tr += s;
    }
    return tr;
}

/* closed form of the same value in float64: [2 + 2|w|^2 - |w x v|^2] / (n^2 q), v = (w - t)/n */
static double info_point_affine_f64(const double pose7[7], const float p_w[3], float q)
{
    const double w[3] = {p_w[0], p_w[1], p_w[2]};
    const double d[3] = {w[0] - (double)(float)pose7[0], w[1] - (double)(float)pose7[1], w[2] - (double)(float)pose7[2]};
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const double n = sqrt(n2);
    const double v[3] = {d[0] / n, d[1] / n, d[2] / n};
    const double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    const double w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    return (2.0 + 2.0 * w2 - (c[0] * c[0] + c[1] * c[1] + c[2] * c[2])) / (n2 * (double)q);
}

int fso_information_for_pose(const fso_grid *g, int32_t n, const double *pose7, int32_t n_kf, const double *kf_pose7,
                             const int32_t *kf_offsets, const float *pts, const fso_kf_params *prm, int n_threads,
                             float *info_ref, double *info_f64, int32_t *n_cells, int32_t *n_points)
{
    if (!g || !pose7 || !prm || n < 0 || n_kf < 0) return -1;
    const size_t ncell = (size_t)g->nx * (size_t)g->ny;
    int bad = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        /* information_map: value and point count per costmap index, plus the list of touched indices */
        float *val = (float *)malloc(ncell * sizeof(float));
        double *val64 = (double *)malloc(ncell * sizeof(double));
        int32_t *cnt = (int32_t *)calloc(ncell, sizeof(int32_t));
        uint32_t *touched = (uint32_t *)malloc(ncell * sizeof(uint32_t));
        if (!val || !val64 || !cnt || !touched) {
#pragma omp atomic write
            bad = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int32_t c = 0; c < n; ++c) {
                const double *pose = pose7 + 7 * (size_t)c;
                double tri[6];
                fso_frustum_vertices_2d(pose, prm->max_depth, prm->hfov, tri);                    /* :851 */
                float pose_information = 0;                                                       /* :848 */
                size_t nt = 0;
                int32_t npts = 0;
                for (int32_t k = 0; k < n_kf; ++k) {
                    const double *kp = kf_pose7 + 7 * (size_t)k;
                    if (prm->radius >= 0.0) {                                                     /* getNodesInRadius :616-632 */
                        const double dx = kp[0] - pose[0], dy = kp[1] - pose[1];
                        if (!(sqrt(dx * dx + dy * dy) <= prm->radius)) continue;
                    }
                    if (!fso_frustum_overlap(pose, kp, prm->max_depth, prm->hfov, prm->max_depth_error)) continue;   /* :855 */
                    for (int32_t j = kf_offsets[k]; j < kf_offsets[k + 1]; ++j) {                 /* :860 word_pts */
                        const float *w = pts + 3 * (size_t)j;
                        if (!fso_point_in_triangle((double)w[0], (double)w[1], tri)) continue;    /* :863 */
                        uint32_t mx, my, mz;
                        if (!fso_world_to_map(g, (double)w[0], (double)w[1], g->origin_z, &mx, &my, &mz)) continue;   /* :866 */
                        const uint32_t index = my * (uint32_t)g->nx + mx;                         /* getIndex */
                        ++npts;
                        if (cnt[index] == 0) {                                                    /* :879 */
                            val[index] = fso_information_of_point_affine(pose, w, prm->q_diag);
                            val64[index] = info_point_affine_f64(pose, w, prm->q_diag);
                            touched[nt++] = index;
                        }
                        ++cnt[index];
                        pose_information += val[index];                                           /* :873,880 */
                    }
                }
                double s64 = 0.0;
                for (size_t i = 0; i < nt; ++i) {
                    s64 += (double)cnt[touched[i]] * val64[touched[i]];
                    cnt[touched[i]] = 0;
                }
                if (info_ref) info_ref[c] = pose_information;
                if (info_f64) info_f64[c] = s64;
                if (n_cells) n_cells[c] = (int32_t)nt;
                if (n_points) n_points[c] = npts;
            }
        }
        free(val); free(val64); free(cnt); free(touched);
    }
    return bad ? -3 : 0;
}
