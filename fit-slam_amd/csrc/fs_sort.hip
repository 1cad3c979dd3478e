// fs_sort.hip — spatial processing order of the candidate list.
//
// The reference scores frontiers in list order (DEP/src/FrontierCostsManager.cpp:74); results are
// order-independent, so the kernels may visit the candidates in any order as long as outputs stay in list
// order.  Candidates are binned by the Morton code of their goal cell at 5 bits per axis (32 x 32 x 32 blocks of the
// map) with a three-launch counting sort — histogram, one-workgroup scan, scatter: neighbouring candidates then run
// in neighbouring wavefronts and share grid cache lines and landmark chunks (measured on C3: fs_raymarch_kernel
// 0.42 ms -> 0.23 ms, fs_fim_kernel -6 %).  The order inside a bin is whatever the scatter's atomics produce; no output
// depends on it.  (A full 30-bit radix sort of the keys with rocPRIM cost 46 us per call for the same effect.)
//
// Heavy blocks first.  The persistent FIM grid drains best when the expensive candidates are not the last ones it
// starts (measured on C3 by permuting the list on the host: heavy candidates last 0.941 ms, first 0.773 ms; DESIGN.md
// 4.2).  What a candidate costs is only known once its chunks are culled — but it is mostly a property of WHERE it stands
// in the cloud, and scoring calls come in sequences over the same map.  So the context keeps a cost map: the FIM worker
// records the landmark tests a candidate took under its block (13 Morton bits; atomicMax into FsFimArgs::costmap, cleared by
// the scan kernel once the key kernel has read it), the scan kernel turns the previous call's total into a mean, and the
// key kernel puts the blocks that held a candidate above that mean in front: key = class << 13 | Morton >> 2 (class 0: above twice that mean, 1: above half of it, 2: the rest) — what is
// still to be started when the list runs dry are candidates from blocks where nothing was expensive.  The first call on a context (all-zero map) is plain Morton
// order; a stale map costs nothing but the order.
#include "fs_internal.h"

#define FS_SORT_BITS 15
#define FS_SORT_BINS (1 << FS_SORT_BITS)
static_assert(FS_COST_BINS == (FS_SORT_BINS >> 2), "two key bits are the cost class");

namespace {

__device__ __forceinline__ uint32_t spread5(uint32_t v)
{
    v &= 0x1fu;
    v = (v | (v << 8)) & 0x100fu;  v = (v | (v << 4)) & 0x10c3u;  v = (v | (v << 2)) & 0x1249u;
    return v;
}

__global__ void fs_sortkey_kernel(int32_t n, const double *goal, FsGridDev g, int sx, int sy, int sz, uint32_t *keys, uint32_t *hist,
                                  const uint32_t *__restrict__ costmap, const uint32_t *__restrict__ cost_mean, uint32_t flip)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // ordering only: any monotone cell estimate will do (off-map / NaN goals clamp to the border)
    const double qx = (goal[3 * i] - g.ox) / g.res, qy = (goal[3 * i + 1] - g.oy) / g.res, qz = (goal[3 * i + 2] - g.oz) / g.res;
    const uint32_t cx = (uint32_t)fmin(fmax(qx, 0.0), (double)(g.nx - 1)) >> sx;
    const uint32_t cy = (uint32_t)fmin(fmax(qy, 0.0), (double)(g.ny - 1)) >> sy;
    const uint32_t cz = (uint32_t)fmin(fmax(qz, 0.0), (double)(g.nz - 1)) >> sz;
    // (flip = FS_COST_BINS - 1 walks the blocks in reverse: "sort.reverse", a knob for measuring the order sensitivity)
    const uint32_t block = ((spread5(cx) | (spread5(cy) << 1) | (spread5(cz) << 2)) >> 2) ^ flip;
    // class 0: a candidate of the block took more than twice the mean of the previous call, 1: more than half of it, 2: the rest
    const uint32_t cost = costmap[block], mean = *cost_mean;
    const uint32_t cls = cost > mean ? (cost > 2u * mean ? 0u : 1u) : (mean != 0xffffffffu && cost > mean / 2u ? 1u : 2u);
    const uint32_t key = cls * (uint32_t)FS_COST_BINS | block;
    keys[i] = key;
    atomicAdd(&hist[key], 1u);
}

// exclusive scan of the histogram by one workgroup: cursor[b] = first position of bin b; the histogram is cleared for
// the next call.  Wave w owns the bins [w * 2048, (w + 1) * 2048) as 8 rows of 64 uint4: every load and store is one
// contiguous kilobyte.
__global__ __launch_bounds__(1024)
void fs_sortscan_kernel(uint32_t *hist, uint32_t *cursor, uint32_t *costmap, uint32_t *cost_mean, unsigned long long *cost_acc)
{
    // the key kernel of this call has read the cost map: clear it for the maxima of this call's candidates
    for (int i = threadIdx.x; i < FS_COST_BINS / 4; i += 1024) reinterpret_cast<uint4 *>(costmap)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0 && cost_acc[1] != 0ull) {             // landmark tests / candidates of the calls since the last sort
        *cost_mean = (uint32_t)(cost_acc[0] / cost_acc[1]);    // read by the NEXT call's key kernel (this call's has run)
        cost_acc[0] = 0ull; cost_acc[1] = 0ull;
    }
    constexpr int ROWS = FS_SORT_BINS / (16 * 64 * 4);
    __shared__ uint32_t wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 *h4 = reinterpret_cast<uint4 *>(hist) + (size_t)wave * ROWS * 64 + lane;
    uint4 q[ROWS];
    uint32_t row_incl[ROWS];                                   // inclusive scan over the lanes of each row
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        q[j] = h4[j * 64];
        h4[j * 64] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint32_t carry = 0;                                        // bins of this wave before row j
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        uint32_t incl = (q[j].x + q[j].y) + (q[j].z + q[j].w);
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        row_incl[j] = carry + incl;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) wave_tot[wave] = carry;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    uint4 *c4 = reinterpret_cast<uint4 *>(cursor) + (size_t)wave * ROWS * 64 + lane;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        uint32_t run = base + row_incl[j] - ((q[j].x + q[j].y) + (q[j].z + q[j].w));
        uint4 o;
        o.x = run; run += q[j].x;
        o.y = run; run += q[j].y;
        o.z = run; run += q[j].z;
        o.w = run;
        c4[j * 64] = o;
    }
}

__global__ void fs_sortscatter_kernel(int32_t n, const uint32_t *keys, uint32_t *cursor, int32_t *perm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    perm[atomicAdd(&cursor[keys[i]], 1u)] = i;
}

int shift_for(int n)
{
    int s = 0;
    while ((n >> s) > 32) ++s;
    return s;
}

}  // namespace

hipError_t fs_launch_sort_candidates(int32_t n, const double *d_goal, const FsGridDev &grid, int32_t *d_perm,
                                    void **scratch, size_t *scratch_bytes, unsigned long long *d_cost_acc,
                                    const uint32_t **d_keys, uint32_t **d_costmap, int use_costmap, int reverse, hipStream_t s)
{
    const size_t a4 = (sizeof(uint32_t) * (size_t)n + 255) & ~(size_t)255;
    const size_t fixed = sizeof(uint32_t) * (2 * FS_SORT_BINS + FS_COST_BINS + 64);      // histogram, cursors, cost map, mean
    const size_t need = a4 + fixed;
    if (need > *scratch_bytes) {
        if (*scratch) (void)hipFree(*scratch);
        *scratch = nullptr; *scratch_bytes = 0;
        ++fs_alloc_generation;
        hipError_t e = hipMalloc(scratch, need);
        if (e != hipSuccess) return e;
        *scratch_bytes = need;
        e = hipMemsetAsync(*scratch, 0, need, s);              // the histogram starts (and is left) all zero; so does the cost map
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(static_cast<char *>(*scratch) + sizeof(uint32_t) * (2 * FS_SORT_BINS + FS_COST_BINS + 1), 0xff, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    char *base = static_cast<char *>(*scratch);
    uint32_t *hist = reinterpret_cast<uint32_t *>(base);
    uint32_t *cursor = hist + FS_SORT_BINS;
    uint32_t *costmap = cursor + FS_SORT_BINS;
    uint32_t *cost_mean = costmap + FS_COST_BINS;
    uint32_t *keys = reinterpret_cast<uint32_t *>(base + fixed);
    *d_keys = keys; *d_costmap = costmap;
    hipLaunchKernelGGL(fs_sortkey_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_goal, grid,
                       shift_for(grid.nx), shift_for(grid.ny), shift_for(grid.nz), keys, hist, costmap,
                       use_costmap ? cost_mean : cost_mean + 1,           // (the word after the mean holds 0xffffffff: nothing is "heavy")
                       reverse ? (uint32_t)(FS_COST_BINS - 1) : 0u);
    hipLaunchKernelGGL(fs_sortscan_kernel, dim3(1), dim3(1024), 0, s, hist, cursor, costmap, cost_mean, d_cost_acc);
    hipLaunchKernelGGL(fs_sortscatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, keys, cursor, d_perm);
    return hipGetLastError();
}
