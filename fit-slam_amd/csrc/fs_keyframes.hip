// fs_keyframes.hip — per-pose key-frame landmark information for gfx950.
//
// Batched form of computeInformationForPose (DEP/include/.../deprecated/util.hpp:840-916, dead code in the reference;
// its call site is the commented block DEP/src/CostCalculator.cpp:326-365): for one pose, every key-frame within
// `radius` (getNodesInRadius, util.hpp:616-632) whose 2-D frustum overlaps the pose's (frustumOverlap, :172-185)
// contributes its world points that lie inside the pose's FOV triangle (isPointInsideTriangle, :49-66) and on the
// costmap; the value of a costmap cell is computed once from the FIRST point that falls into it (the reference's
// information_map, :868-881) and added once per point of that cell.
//
// One 256-thread workgroup per pose.  The triangle vertices and the key-frames' six check points come from the host
// (libm cos/sin in double, like the ray direction tables); the point-in-triangle test and worldToMap run in fp64 on
// the device with the reference's operation order (+, -, x, / only; built with -ffp-contract=off).  information_map
// is an LDS hash table keyed by the cell index holding (smallest point index, point count): "first point in list
// order" == smallest index of the CSR point array, so the result does not depend on the order the lanes arrive in.
// Poses whose triangle covers more distinct cells than the LDS table holds are redone with a table in HBM.
#include "fs_internal.h"

#define FS_KF_THREADS 256
#define FS_KF_LDS_BITS 12

namespace {

__device__ __forceinline__ bool in_triangle(double px, double py, const double *t)
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

// affine computeJacobianForPoint + computeFIM + trace (util.hpp:687-759), float32, Q = q*I
__device__ float info_point_affine(const float *R, const float *t, float wx, float wy, float wz, float qinv)
{
    const float dx = wx - t[0], dy = wy - t[1], dz = wz - t[2];
    float p[3];
    p[0] = __fmaf_rn(R[0], dx, __fmaf_rn(R[3], dy, R[6] * dz));
    p[1] = __fmaf_rn(R[1], dx, __fmaf_rn(R[4], dy, R[7] * dz));
    p[2] = __fmaf_rn(R[2], dx, __fmaf_rn(R[5], dy, R[8] * dz));
    const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const float a = 1 / n, b = 1 / (n * n * n);
    const float w[3] = {wx, wy, wz};
    // right = [ I | -[w]x ]
    const float S[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
    float D[18], tr = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float r = (j < 3) ? (k == j ? 1.0f : 0.0f) : (-1.0f) * S[3 * k + (j - 3)];
                s += R[3 * k + i] * r;                               // left = R^T
            }
            D[6 * i + j] = s;
        }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float col = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float A = a * (i == k ? 1.0f : 0.0f) - (b * p[i]) * p[k];
                s += A * D[6 * k + j];
            }
            col += (s * qinv) * s;
        }
        tr += col;
    }
    return tr;
}

template <bool GLOBAL_TABLE>
__device__ void kf_pose(const FsKfArgs &a, int c, uint32_t *keys, uint32_t *ming, uint32_t *cnt, int bits)
{
    __shared__ int sh_accept[FS_KF_THREADS];
    __shared__ int sh_overflow, sh_running, sh_cells, sh_points;
    __shared__ double sh_sum[FS_KF_THREADS / 64];
    const int tid = threadIdx.x;
    const uint32_t slots = 1u << bits, smask = slots - 1u;
    for (uint32_t i = tid; i < slots; i += FS_KF_THREADS) { keys[i] = 0u; ming[i] = 0xffffffffu; cnt[i] = 0u; }
    if (tid == 0) { sh_overflow = 0; sh_running = 0; sh_cells = 0; sh_points = 0; }
    double tri[6], trie[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { tri[i] = a.tri[12 * (size_t)c + i]; trie[i] = a.tri[12 * (size_t)c + 6 + i]; }
    const double posx = tri[0], posy = tri[1];
    const uint32_t limit = GLOBAL_TABLE ? 0xffffffffu : (slots >> 2) * 3u;    // an LDS table is used up to 3/4 full
    int my_points = 0, my_cells = 0;
    bool failed = false;
    __syncthreads();

    for (int kb = 0; kb < a.n_kf && !failed; kb += FS_KF_THREADS) {
        // one key-frame per thread: getNodesInRadius (util.hpp:616-632) + frustumOverlap (:172-185)
        const int k = kb + tid;
        int acc = 0;
        if (k < a.n_kf) {
            const double *chk = a.kf_check + 12 * (size_t)k;
            bool near = true;
            if (a.radius >= 0.0) {
                const double dx = chk[0] - posx, dy = chk[1] - posy;
                near = sqrt(dx * dx + dy * dy) <= a.radius;
            }
            if (near) {
#pragma unroll 1
                for (int i = 0; i < 6 && !acc; ++i) acc = in_triangle(chk[2 * i], chk[2 * i + 1], trie) ? 1 : 0;
            }
        }
        sh_accept[tid] = acc;
        __syncthreads();
        const int kend = min(FS_KF_THREADS, a.n_kf - kb);
        for (int kk = 0; kk < kend; ++kk) {
            if (!sh_accept[kk]) continue;
            const int j0 = a.kf_offsets[kb + kk], j1 = a.kf_offsets[kb + kk + 1];
            for (int j = j0 + tid; j < j1; j += FS_KF_THREADS) {
                const float wx = a.px[j], wy = a.py[j];
                const double fx = (double)wx, fy = (double)wy;
                if (!in_triangle(fx, fy, tri)) continue;                          // :863
                // Costmap2D::worldToMap on (x, y) (:866) and getIndex
                if (fx < a.ox || fy < a.oy) continue;
                const double qx = (fx - a.ox) / a.res, qy = (fy - a.oy) / a.res;
                if (!(qx < 4294967296.0) || !(qy < 4294967296.0)) continue;
                const uint32_t mx = (uint32_t)qx, my = (uint32_t)qy;
                if (mx >= (uint32_t)a.nx || my >= (uint32_t)a.ny) continue;
                const uint32_t key = my * (uint32_t)a.nx + mx + 1u;
                ++my_points;
                uint32_t h = (key * 2654435761u) >> (32 - bits);
                bool placed = false;
                for (uint32_t probe = 0; probe < slots; ++probe) {
                    uint32_t cur = keys[h];
                    if (cur == 0u) {
                        cur = atomicCAS(&keys[h], 0u, key);
                        if (cur == 0u) { ++my_cells; cur = key; }
                    }
                    if (cur == key) { atomicMin(&ming[h], (uint32_t)j); atomicAdd(&cnt[h], 1u); placed = true; break; }
                    h = (h + 1u) & smask;
                }
                if (!placed) sh_overflow = 1;
            }
        }
        if (!GLOBAL_TABLE && my_cells) { atomicAdd(&sh_running, my_cells); my_cells = 0; }
        __syncthreads();
        failed = !GLOBAL_TABLE && ((uint32_t)sh_running > limit || sh_overflow != 0);
        __syncthreads();                                                          // sh_accept is rewritten next round
    }
    if (failed) {                                                                 // hand the pose to the HBM-table pass
        if (tid == 0) {
            const unsigned long long slot = atomicAdd(&a.counters[0], 1ull);
            a.flagged[slot] = c;
        }
        return;
    }

    // information_map (:868-881): value of the first point of every occupied cell, added once per point of the cell
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = a.Rt[12 * (size_t)c + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = a.Rt[12 * (size_t)c + 9 + i];
    double sum = 0.0;
    int cells = 0;
    for (uint32_t i = tid; i < slots; i += FS_KF_THREADS) {
        if (keys[i] == 0u) continue;
        const uint32_t j = ming[i];
        const float v = info_point_affine(R, t, a.px[j], a.py[j], a.pz[j], a.qinv);
        sum += (double)cnt[i] * (double)v;
        ++cells;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        cells += __shfl_xor(cells, d);
        my_points += __shfl_xor(my_points, d);
    }
    if ((tid & 63) == 0) {
        sh_sum[tid >> 6] = sum;
        atomicAdd(&sh_points, my_points);
        atomicAdd(&sh_cells, cells);
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < FS_KF_THREADS / 64; ++w) s += sh_sum[w];
        a.info[c] = (float)s;
        a.n_cells[c] = sh_cells;
        a.n_points[c] = sh_points;
    }
}

__global__ __launch_bounds__(FS_KF_THREADS)
void fs_kf_info_kernel(const FsKfArgs a)
{
    __shared__ uint32_t tab[3 << FS_KF_LDS_BITS];
    const int c = blockIdx.x;
    if (c >= a.n) return;
    kf_pose<false>(a, c, tab, tab + (1 << FS_KF_LDS_BITS), tab + (2 << FS_KF_LDS_BITS), FS_KF_LDS_BITS);
}

// poses the LDS pass flagged: a pool of workgroups, each with its own table in HBM (sized for every point of the map)
__global__ __launch_bounds__(FS_KF_THREADS)
void fs_kf_info_global_kernel(const FsKfArgs a)
{
    const int count = (int)a.counters[0];
    uint32_t *base = a.gtable + 3 * ((size_t)blockIdx.x << a.gbits);
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        kf_pose<true>(a, a.flagged[i], base, base + ((size_t)1 << a.gbits), base + ((size_t)2 << a.gbits), a.gbits);
        __syncthreads();
    }
}

}  // namespace

hipError_t fs_launch_kf_info(const FsKfArgs &a, int pool, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fs_kf_info_kernel, dim3(a.n), dim3(FS_KF_THREADS), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fs_kf_info_global_kernel, dim3(a.n < pool ? a.n : pool), dim3(FS_KF_THREADS), 0, s, a);
    return hipGetLastError();
}
