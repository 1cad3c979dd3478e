// fs_rank.hip — U1 utility, weighted cost and ranking of the scored candidates on the GPU.
//
// Replaces the tail of FrontierCostsManager::assignCosts (DEP/src/FrontierCostsManager.cpp:118,126-205):
// running min/max of path length and arrival information over the achievable candidates
// (recomputeNormalizationFactors, DEP/src/CostCalculator.cpp:512-520), then per candidate
//   u = alpha * (arrival / max_arrival_gt) + (1 - alpha) * (1 - (len/vmax + heading/wmax) / (maxlen/vmax + pi/wmax)),
//   cost = 1 / (beta * u).
// All arithmetic is fp64 with the reference's operation order (basic IEEE ops only), so costs match
// the CPU bit for bit.  Ranking = stable ascending sort by cost: rocPRIM's radix sort on the order-preserving integer image
// of the double for long lists; for lists of up to 1024 candidates (the reference's operating point) ONE workgroup does the
// normalisation, the costs and a bitonic sort of (cost image, index) pairs in LDS in one launch.
#include "fs_internal.h"

#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

#define FS_DBL_MAX 1.7976931348623157e308

struct RankNorm {
    double min_dist, max_dist, min_info;
};

__device__ __forceinline__ double wave_min(double x)
{
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(x, d); x = (o < x) ? o : x; }
    return x;
}
__device__ __forceinline__ double wave_max(double x)
{
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(x, d); x = (x < o) ? o : x; }
    return x;
}

// reset() + recomputeNormalizationFactors over the list (CostCalculator.hpp:110-116, CostCalculator.cpp:512-520).
// Up to FS_NORM_BLOCKS workgroups reduce a slice each into `partial`; the last one to finish (a ticket that it resets to zero
// for the next call) folds the partials into `out` — min / max are exact whatever the order, so the result does not depend on it.
#define FS_NORM_BLOCKS 64
__global__ __launch_bounds__(1024)
void fs_rank_norm_kernel(int32_t n, const fs_record *rec, const uint8_t *black, const double *len,
                         RankNorm *out, RankNorm *partial, unsigned int *ticket, int32_t *err)
{
    __shared__ double s_min_d[16], s_max_d[16], s_min_i[16];
    __shared__ bool s_last;
    double min_d = FS_DBL_MAX, max_d = -1.0, min_i = FS_DBL_MAX;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (black && black[i]) continue;
        if (!(rec[i].flags & FS_FLAG_ACHIEVABLE)) continue;
        const double l = len[i], a = (double)rec[i].arrival;
        min_d = (l < min_d) ? l : min_d;
        max_d = (max_d < l) ? l : max_d;
        min_i = (a < min_i) ? a : min_i;
    }
    min_d = wave_min(min_d); max_d = wave_max(max_d); min_i = wave_min(min_i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_min_d[wave] = min_d; s_max_d[wave] = max_d; s_min_i[wave] = min_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            min_d = (s_min_d[w] < min_d) ? s_min_d[w] : min_d;
            max_d = (max_d < s_max_d[w]) ? s_max_d[w] : max_d;
            min_i = (s_min_i[w] < min_i) ? s_min_i[w] : min_i;
        }
        partial[blockIdx.x] = RankNorm{min_d, max_d, min_i};
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __threadfence();
        RankNorm r{FS_DBL_MAX, -1.0, FS_DBL_MAX};
        for (unsigned int b = 0; b < gridDim.x; ++b) {
            const volatile RankNorm *p = partial + b;
            const double a = p->min_dist, m = p->max_dist, i = p->min_info;
            r.min_dist = (a < r.min_dist) ? a : r.min_dist;
            r.max_dist = (r.max_dist < m) ? m : r.max_dist;
            r.min_info = (i < r.min_info) ? i : r.min_info;
        }
        *out = r;
        *err = 0;
        *ticket = 0u;
    }
}

__device__ __forceinline__ uint64_t sortable(double v)
{
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// One candidate's U1 utilities and weighted cost (FrontierCostsManager.cpp:126-205) — ONE definition for the list kernel and the
// small-list kernel below, so that both give the same bits.  Returns true where the reference throws ("Cost out of bounds").
__device__ __forceinline__ bool u1_cost(bool live, double arrival, double len, double head, double max_d, double min_d, double min_i,
                                        double alpha, double beta, double max_vx, double max_wz, double max_gt,
                                        double &c, double &au, double &pu)
{
    const double PI = 3.14159265358979323846;
    if (!live) {
        // FrontierCostsManager.cpp:84 (blacklisted) / :129-135 (not achievable)
        c = FS_DBL_MAX; au = -69.8; pu = -1.8;
        return false;
    }
    if ((double)(max_gt - min_i) == 0.0) au = 0.0;                                  // :139-140
    else au = arrival / (double)max_gt;                                               // :145-146
    if ((double)((max_d / max_vx + PI / max_wz) - (min_d / max_vx + 0.0 / max_wz) == 0.0)) pu = 1.0;   // :152-153
    else pu = (double)(len / max_vx + head / max_wz) / (double)(max_d / max_vx + PI / max_wz);           // :158-159
    pu = 1.0 - pu;                                                                    // :160
    const bool out_of_bounds = au > 1.0 || au < 0.0 || pu > 1.0 || pu < 0.0;          // :148-149,173-174
    double utility = (alpha * au) + ((1.0 - alpha) * pu);                             // :176-177
    if (utility == 0.0) utility = 1e-16;                                              // :178-182
    c = 1 / (beta * utility);                                                         // :198
    return out_of_bounds;
}

__global__ void fs_rank_cost_kernel(int32_t n, const fs_record *rec, const uint8_t *black, const double *len,
                                    const double *head, const RankNorm *norm, double alpha, double beta,
                                    double max_vx, double max_wz, double max_gt,
                                    double *cost, double *au_out, double *du_out,
                                    uint64_t *keys, int32_t *vals, int32_t *err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c, au, pu;
    const bool live = !(black && black[i]) && (rec[i].flags & FS_FLAG_ACHIEVABLE);
    if (u1_cost(live, (double)rec[i].arrival, live ? len[i] : 0.0, live ? head[i] : 0.0, norm->max_dist, norm->min_dist, norm->min_info,
                alpha, beta, max_vx, max_wz, max_gt, c, au, pu))
        atomicOr(err, 1);
    cost[i] = c; au_out[i] = au; du_out[i] = pu;
    keys[i] = sortable(c);
    vals[i] = i;
}

// The whole ranking of a SHORT list — the reference handles tens of frontiers per tick — in ONE launch of ONE workgroup:
// normalisation factors (block reduction), U1 costs, and the stable ascending order by a bitonic network over (cost image,
// index) pairs in LDS (the index breaks ties, which makes the order the stable one whatever the network does).  Replaces three
// launches (normalisation, costs, rocPRIM's block sort) for n <= FS_RANK_SMALL_MAX.
#define FS_RANK_SMALL_MAX 1024
__global__ __launch_bounds__(FS_RANK_SMALL_MAX)
void fs_rank_small_kernel(int32_t n, const fs_record *rec, const uint8_t *black, const double *len, const double *head,
                          double alpha, double beta, double max_vx, double max_wz, double max_gt,
                          double *cost, double *au_out, double *du_out, int32_t *order, int32_t *err)
{
    __shared__ uint64_t s_key[FS_RANK_SMALL_MAX];
    __shared__ int32_t s_idx[FS_RANK_SMALL_MAX];
    __shared__ double s_min_d[16], s_max_d[16], s_min_i[16];
    __shared__ int s_err;
    const int i = threadIdx.x, P = blockDim.x;                 // P: a power of two >= n (and >= 64)
    const int wave = i >> 6, lane = i & 63;
    if (i == 0) s_err = 0;
    bool live = false;
    double l = 0.0, h = 0.0, a = 0.0;
    if (i < n) {
        live = !(black && black[i]) && (rec[i].flags & FS_FLAG_ACHIEVABLE);
        a = (double)rec[i].arrival;
        if (live) { l = len[i]; h = head[i]; }
    }
    // reset() + recomputeNormalizationFactors over the live candidates (CostCalculator.cpp:512-520)
    double min_d = live ? l : FS_DBL_MAX, max_d = live ? l : -1.0, min_i = live ? a : FS_DBL_MAX;
    min_d = wave_min(min_d); max_d = wave_max(max_d); min_i = wave_min(min_i);
    if (lane == 0) { s_min_d[wave] = min_d; s_max_d[wave] = max_d; s_min_i[wave] = min_i; }
    __syncthreads();
    for (int w = 0; w < (P >> 6); ++w) {                       // (every thread folds the <= 16 partials itself: min / max are exact in any order)
        min_d = (s_min_d[w] < min_d) ? s_min_d[w] : min_d;
        max_d = (max_d < s_max_d[w]) ? s_max_d[w] : max_d;
        min_i = (s_min_i[w] < min_i) ? s_min_i[w] : min_i;
    }
    uint64_t key = ~0ull;                                       // padding sorts behind every candidate
    if (i < n) {
        double c, au, pu;
        if (u1_cost(live, a, l, h, max_d, min_d, min_i, alpha, beta, max_vx, max_wz, max_gt, c, au, pu)) atomicOr(&s_err, 1);
        cost[i] = c; au_out[i] = au; du_out[i] = pu;
        key = sortable(c);
    }
    s_key[i] = key; s_idx[i] = i;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = i ^ j;
            if (p > i) {
                const uint64_t ka = s_key[i], kb = s_key[p];
                const int32_t ia = s_idx[i], ib = s_idx[p];
                const bool a_after_b = ka > kb || (ka == kb && ia > ib);
                if (a_after_b == ((i & k) == 0)) { s_key[i] = kb; s_key[p] = ka; s_idx[i] = ib; s_idx[p] = ia; }
            }
            __syncthreads();
        }
    }
    if (i < n) order[i] = s_idx[i];
    if (i == 0) *err = s_err;
}

}  // namespace

hipError_t fs_launch_rank(int32_t n, const fs_record *d_records, const uint8_t *d_black,
                          const double *d_len, const double *d_head, double alpha, double beta,
                          double max_vx, double max_wz, double max_gt, double *d_cost, double *d_au,
                          double *d_du, int32_t *d_order, int32_t *d_err, void **scratch, size_t *scratch_bytes,
                          hipStream_t s)
{
    if (n <= FS_RANK_SMALL_MAX) {
        int threads = 64;
        while (threads < n) threads <<= 1;
        hipLaunchKernelGGL(fs_rank_small_kernel, dim3(1), dim3(threads), 0, s, n, d_records, d_black, d_len, d_head, alpha, beta, max_vx, max_wz,
                           max_gt, d_cost, d_au, d_du, d_order, d_err);
        return hipGetLastError();
    }
    size_t temp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, temp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0, 64, s);
    if (e != hipSuccess) return e;
    const size_t off_keys_in = 256 + sizeof(RankNorm) * FS_NORM_BLOCKS;      // [norm | ticket ... | partials][keys ...]
    const size_t off_keys_out = off_keys_in + ((sizeof(uint64_t) * (size_t)n + 255) & ~(size_t)255);
    const size_t off_vals_in = off_keys_out + ((sizeof(uint64_t) * (size_t)n + 255) & ~(size_t)255);
    const size_t off_temp = off_vals_in + ((sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255);
    const size_t need = off_temp + temp_bytes + 256;
    if (need > *scratch_bytes) {
        if (*scratch) (void)hipFree(*scratch);
        *scratch = nullptr; *scratch_bytes = 0;
        ++fs_alloc_generation;
        e = hipMalloc(scratch, need);
        if (e != hipSuccess) return e;
        *scratch_bytes = need;
        e = hipMemsetAsync(*scratch, 0, 256, s);                  // the norm kernel's ticket starts at zero (and returns there after every call)
        if (e != hipSuccess) return e;
    }
    char *base = static_cast<char *>(*scratch);
    RankNorm *norm = reinterpret_cast<RankNorm *>(base);
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(base + off_keys_in);
    uint64_t *keys_out = reinterpret_cast<uint64_t *>(base + off_keys_out);
    int32_t *vals_in = reinterpret_cast<int32_t *>(base + off_vals_in);
    void *temp = base + off_temp;

    unsigned int *ticket = reinterpret_cast<unsigned int *>(base + 128);
    RankNorm *partial = reinterpret_cast<RankNorm *>(base + 256);
    const int norm_blocks = std::min(FS_NORM_BLOCKS, (n + 1023) / 1024);
    hipLaunchKernelGGL(fs_rank_norm_kernel, dim3(norm_blocks), dim3(1024), 0, s, n, d_records, d_black, d_len, norm, partial, ticket, d_err);
    hipLaunchKernelGGL(fs_rank_cost_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_records, d_black, d_len,
                       d_head, norm, alpha, beta, max_vx, max_wz, max_gt, d_cost, d_au, d_du, keys_in, vals_in, d_err);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, d_order, (size_t)n, 0, 64, s);
}
