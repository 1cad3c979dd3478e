// fs_fim.hip — landmark Fisher-information accumulation for gfx950 (CDNA4).
//
// Replaces FisherInformationManager::isPoseSafe's landmark loop and getInformationFromLookup
// (FIP/src/fisher_information/FisherInfoManager.cpp:83-100,287-324) plus the per-landmark Jacobian/FIM
// of FisherInformationHelpers.cpp:71-123, batched over candidate poses.
//
// Mapping: one workgroup per candidate.  The landmark cloud (SoA fp32, staged once to HBM, L2-resident
// at these sizes) is streamed with 16-byte loads, four landmarks per lane per step.  For every
// landmark: p = R^T (w - t) in fp32 with explicit fma order, the visibility predicate (range + cone,
// multiplications and compares only), the voxel lattice index in fp64 (round(p * 1/0.3f), identical to
// getVoxelCoordinate, FisherInfoManager.hpp:108-123).  The reference's per-voxel bookkeeping
// (pointCount / version, :296-304) becomes an LDS hash table keyed by the dense lattice index:
// the returning LDS atomic that bumps the voxel's count yields the landmark's rank k in its voxel,
// hence its crowding factor exp(1 - k^0.8) (table, FisherInfoManager.hpp:102-106) — the same multiset
// of (info_v, k) terms the reference accumulates sequentially.  The unit-weight 6x6 FIM uses the block
// form F(p) = [[P/n^2, -S/n^2],[S/n^2, P]] (17 independent sums).  Sums are reduced with wave shuffles,
// then across waves through LDS; lane 0 finishes trace and log det (Cholesky, fp64).
// This is a scan + scatter-count: no dense contraction, so no MFMA.
#include "fs_internal.h"

#define FS_FIM_THREADS 1024
#define FS_FIM_WAVES (FS_FIM_THREADS / 64)
#define FS_NACC 18     // info, 15 FIM block sums, n_visible, n_voxels

namespace {

struct Acc {
    float info;
    float A[6];                   // sum P        , P = I - p^ p^T       (xx, xy, xz, yy, yz, zz)
    float B[6];                   // sum P / n^2
    float s[3];                   // sum p / n^2
    int nvis, nvox;
};

__device__ __forceinline__ uint32_t hash_key(uint32_t key, int bits)
{
    return (key * 2654435761u) >> (32 - bits);
}

// find-or-insert `key` and bump its count; returns the landmark's rank in the voxel (1-based),
// 0 if the table is full.  *is_new is set when this call created the voxel entry.
__device__ __forceinline__ uint32_t table_bump(uint32_t *table, int bits, uint32_t key, bool &is_new)
{
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t tag = (key + 1u) << FS_SLOT_CNT_BITS;
    uint32_t h = hash_key(key, bits);
    is_new = false;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint32_t cur = table[h];
        if (cur == 0u) {
            const uint32_t prev = atomicCAS(&table[h], 0u, tag | 1u);
            if (prev == 0u) { is_new = true; return 1u; }
            cur = prev;
        }
        if ((cur & ~FS_SLOT_CNT_MASK) == tag) {
            if ((cur & FS_SLOT_CNT_MASK) >= FS_SLOT_CNT_SAT) return FS_SLOT_CNT_SAT + 1u;   // factor is 0.0f out here
            const uint32_t old = atomicAdd(&table[h], 1u);
            return (old & FS_SLOT_CNT_MASK) + 1u;
        }
        h = (h + 1u) & mask;
    }
    return 0u;
}

__device__ __forceinline__ void visible_landmark(const FsFimArgs &a, uint32_t *table, int bits,
                                                 float px, float py, float pz, float n2,
                                                 Acc &acc, bool &overflow)
{
    acc.nvis += 1;
    // unit-weight FIM block sums, F(p) = [[P/n^2, -S/n^2],[S/n^2, P]].  P's diagonal is
    // formed as a sum of squares (1 - ux^2 = uy^2 + uz^2) so that no per-landmark cancellation occurs.
    // The zero vector has no Jacobian.
    if (n2 > 0.0f) {
        const float q = 1.0f / n2;
        const float qx = px * q, qy = py * q, qz = pz * q;
        const float xx = px * qx, yy = py * qy, zz = pz * qz;
        const float a[6] = {yy + zz, -(px * qy), -(px * qz), xx + zz, -(py * qz), xx + yy};
#pragma unroll
        for (int i = 0; i < 6; ++i) { acc.A[i] += a[i]; acc.B[i] += a[i] * q; }
        acc.s[0] += qx; acc.s[1] += qy; acc.s[2] += qz;
    }
    // getVoxelCoordinate: round(x * (1 / corrected_step)) in double, FisherInfoManager.hpp:119-121
    const int jx = (int)round((double)px * a.inv_step) - a.jx0;
    const int jy = (int)round((double)py * a.inv_step) - a.jy0;
    const int jz = (int)round((double)pz * a.inv_step) - a.jz0;
    if ((unsigned)jx >= (unsigned)a.tx || (unsigned)jy >= (unsigned)a.ty || (unsigned)jz >= (unsigned)a.tz) return;   // key miss -> NaN -> skipped (:90-94)
    const uint32_t key = ((uint32_t)jx * (uint32_t)a.ty + (uint32_t)jy) * (uint32_t)a.tz + (uint32_t)jz;
    const float info_v = a.table[key];
    if (info_v != info_v) return;                            // absent lattice point
    bool is_new;
    const uint32_t rank = table_bump(table, bits, key, is_new);
    if (rank == 0u) { overflow = true; return; }
    if (is_new) acc.nvox += 1;                               // occupied_voxel_count_++ (:304)
    const float fac = rank < FS_FACTOR_N ? a.factor[rank] : 0.0f;
    // :318 — (float)(double information * float factor): the double product of two floats is exact,
    // so one rounding to float == the fp32 product
    acc.info += info_v * fac;
}

template <bool GLOBAL_TABLE>
__device__ __forceinline__ void fim_candidate(const FsFimArgs &a, int c, uint32_t *table, int bits, float *red)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ int sh_overflow;

    // ---- pose
    float R[9], t[3];
    if (a.Rt) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = a.Rt[12 * (size_t)c + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = a.Rt[12 * (size_t)c + 9 + i];
    } else {
        const int idx = a.argmax[c];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = a.yawR[9 * idx + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = (float)a.goal[3 * (size_t)c + i];   // getTransformFromPose: float translation
    }

    // ---- clear the hash table
    const uint32_t slots = 1u << bits;
    for (uint32_t i = tid; i < slots; i += FS_FIM_THREADS) table[i] = 0u;
    if (tid == 0) sh_overflow = 0;
    __syncthreads();

    Acc acc;
    acc.info = 0.f; acc.nvis = 0; acc.nvox = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { acc.A[i] = 0.f; acc.B[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 3; ++i) acc.s[i] = 0.f;
    bool overflow = false;

    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(a.lx);
    const float4 *__restrict__ Y = reinterpret_cast<const float4 *>(a.ly);
    const float4 *__restrict__ Z = reinterpret_cast<const float4 *>(a.lz);
    const int n4 = a.m_padded >> 2;
    for (int i4 = tid; i4 < n4; i4 += FS_FIM_THREADS) {
        const float4 x4 = X[i4], y4 = Y[i4], z4 = Z[i4];
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
        const float ys[4] = {y4.x, y4.y, y4.z, y4.w};
        const float zs[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // p = R^T (w - t), fp32 with the operation order fixed by DESIGN.md "FIM accumulate"
            const float dx = xs[k] - t[0], dy = ys[k] - t[1], dz = zs[k] - t[2];
            const float px = __fmaf_rn(R[0], dx, __fmaf_rn(R[3], dy, R[6] * dz));
            const float py = __fmaf_rn(R[1], dx, __fmaf_rn(R[4], dy, R[7] * dz));
            const float pz = __fmaf_rn(R[2], dx, __fmaf_rn(R[5], dy, R[8] * dz));
            const float n2 = __fmaf_rn(px, px, __fmaf_rn(py, py, pz * pz));
            bool vis = (n2 <= a.maxd2);
            if (a.cone_mode == 1) vis = vis && (px >= 0.0f) && (px * px >= a.cos2 * n2);
            else if (a.cone_mode == 2) vis = vis && ((px >= 0.0f) || (px * px <= a.cos2 * n2));
            if (vis) visible_landmark(a, table, bits, px, py, pz, n2, acc, overflow);
        }
    }
    if (overflow) sh_overflow = 1;

    // ---- reduce: wave shuffles, then across waves through LDS
    float vals[FS_NACC];
    vals[0] = acc.info;
#pragma unroll
    for (int i = 0; i < 6; ++i) { vals[1 + i] = acc.A[i]; vals[7 + i] = acc.B[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) vals[13 + i] = acc.s[i];
    vals[16] = (float)acc.nvis;     // exact: < 2^24 per lane
    vals[17] = (float)acc.nvox;
#pragma unroll
    for (int i = 0; i < FS_NACC; ++i) {
        float x = vals[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        vals[i] = x;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < FS_NACC; ++i) red[wave * FS_NACC + i] = vals[i];
    }
    __syncthreads();
    // cross-wave sums in fp64, one thread per quantity; the 6x6 assembly and log det run in
    // fs_fim_finish_kernel so that this kernel carries no private arrays
    if (tid < FS_NACC) {
        double x = 0.0;
        for (int w = 0; w < FS_FIM_WAVES; ++w) x += (double)red[w * FS_NACC + tid];
        if (!(sh_overflow && !GLOBAL_TABLE)) a.sums[(size_t)c * FS_NACC + tid] = x;
    }
    if (tid == 0 && !GLOBAL_TABLE) a.overflow[c] = sh_overflow ? 1u : 0u;   // the overflow pass recomputes flagged candidates
    __syncthreads();
}

__device__ __forceinline__ void zero_sums(const FsFimArgs &a, int c)
{
    if (threadIdx.x < FS_NACC) a.sums[(size_t)c * FS_NACC + threadIdx.x] = 0.0;
    if (threadIdx.x == 0) a.overflow[c] = 0u;
}

__global__ __launch_bounds__(FS_FIM_THREADS)
void fs_fim_kernel(const FsFimArgs a)
{
    extern __shared__ uint32_t fs_fim_lds[];
    uint32_t *table = fs_fim_lds;
    float *red = reinterpret_cast<float *>(fs_fim_lds + (1u << a.hash_bits));
    const int c = blockIdx.x;
    if (c >= a.n) return;
    if (a.status && a.status[c] != FS_STATUS_OK) {           // blacklisted / off-map candidates carry zero FI
        zero_sums(a, c);
        return;
    }
    fim_candidate<false>(a, c, table, a.hash_bits, red);
}

// Second pass for candidates whose voxel set did not fit the LDS table: same code, table in HBM.
__global__ __launch_bounds__(FS_FIM_THREADS)
void fs_fim_overflow_kernel(const FsFimArgs a)
{
    __shared__ float red[FS_FIM_WAVES * FS_NACC];
    uint32_t *table = a.gtable + ((size_t)blockIdx.x << a.ghash_bits);
    for (int c = blockIdx.x; c < a.n; c += gridDim.x) {
        if (a.overflow[c] == 0u) continue;                   // uniform per workgroup
        fim_candidate<true>(a, c, table, a.ghash_bits, red);
    }
}

// One thread per candidate: assemble the 6x6 FIM from the 17 block sums, trace, log det.
__global__ void fs_fim_finish_kernel(const FsFimArgs a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n) return;
    const double *S = a.sums + (size_t)c * FS_NACC;
    const double *A = S + 1, *B = S + 7, *Sv = S + 13;
    // F = [[ B , -[s]x ], [ [s]x , A ]]   (block form of sum_k F(p_k))
    double F[6][6];
    const int ix[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            F[i][j] = B[ix[i][j]];
            F[i + 3][j + 3] = A[ix[i][j]];
        }
    // [s]x = [[0,-sz,sy],[sz,0,-sx],[-sy,sx,0]]
    const double sk[3][3] = {{0, -Sv[2], Sv[1]}, {Sv[2], 0, -Sv[0]}, {-Sv[1], Sv[0], 0}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            F[i][j + 3] = -sk[i][j];
            F[i + 3][j] = sk[i][j];
        }
    a.info_ref[c] = (float)S[0];
    a.trace[c] = (float)((A[0] + A[3] + A[5]) + (B[0] + B[3] + B[5]));
    if (a.fim21) {
        int o = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) a.fim21[21 * (size_t)c + o++] = (float)F[i][j];
    }
    // log det by Cholesky (fp64); -inf when singular: fewer than 3 landmarks can never give rank 6, and a
    // pivot below 1e-6 of its diagonal entry is rounding noise of the fp32 sums
    const int nvis = (int)(S[16] + 0.5);
    double L[6][6];
    double ld = 0.0;
    bool pd = nvis >= 3;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = F[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 1e-6 * F[j][j])) pd = false;
        const double ljj = sqrt(pd ? d : 1.0);
        L[j][j] = ljj;
        ld += 2.0 * log(ljj);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = F[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / ljj;
        }
    }
    a.logdet[c] = pd ? (float)ld : -INFINITY;
    a.n_visible[c] = nvis;
    a.n_voxels[c] = (int)(S[17] + 0.5);
}

}  // namespace

size_t fs_fim_lds_bytes(int hash_bits)
{
    return sizeof(uint32_t) * ((size_t)1 << hash_bits) + sizeof(float) * FS_FIM_WAVES * FS_NACC;
}

hipError_t fs_launch_fim(const FsFimArgs &a, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    const size_t lds = fs_fim_lds_bytes(a.hash_bits);
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fs_fim_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    hipLaunchKernelGGL(fs_fim_kernel, dim3(a.n), dim3(FS_FIM_THREADS), lds, s, a);
    return hipGetLastError();
}

hipError_t fs_launch_fim_overflow(const FsFimArgs &a, int pool, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    const int blocks = a.n < pool ? a.n : pool;
    hipLaunchKernelGGL(fs_fim_overflow_kernel, dim3(blocks), dim3(FS_FIM_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t fs_launch_fim_finish(const FsFimArgs &a, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fs_fim_finish_kernel, dim3((a.n + 127) / 128), dim3(128), 0, s, a);
    return hipGetLastError();
}
