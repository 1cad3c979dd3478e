// fs_sort.hip — spatial processing order of the candidate list.
//
// The reference scores frontiers in list order (DEP/src/FrontierCostsManager.cpp:74); results are
// order-independent, so the ray-march kernel may visit them in any order as long as outputs stay in list
// order.  Candidates are keyed by the Morton code of their goal cell (10 bits per axis) and radix-sorted
// (rocPRIM): neighbouring candidates then run in neighbouring wavefronts and share grid cache lines
// (measured on C3: fs_raymarch_kernel 0.42 ms -> 0.25 ms).
#include "fs_internal.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;  v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void fs_sortkey_kernel(int32_t n, const double *goal, FsGridDev g, int sx, int sy, int sz, uint32_t *keys, int32_t *vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // ordering only: any monotone cell estimate will do (off-map / NaN goals clamp to the border)
    const double qx = (goal[3 * i] - g.ox) / g.res, qy = (goal[3 * i + 1] - g.oy) / g.res, qz = (goal[3 * i + 2] - g.oz) / g.res;
    const uint32_t cx = (uint32_t)fmin(fmax(qx, 0.0), (double)(g.nx - 1)) >> sx;
    const uint32_t cy = (uint32_t)fmin(fmax(qy, 0.0), (double)(g.ny - 1)) >> sy;
    const uint32_t cz = (uint32_t)fmin(fmax(qz, 0.0), (double)(g.nz - 1)) >> sz;
    keys[i] = spread10(cx) | (spread10(cy) << 1) | (spread10(cz) << 2);
    vals[i] = i;
}

int shift_for(int n)
{
    int s = 0;
    while ((n >> s) > 1024) ++s;
    return s;
}

}  // namespace

hipError_t fs_launch_sort_candidates(int32_t n, const double *d_goal, const FsGridDev &grid, int32_t *d_perm,
                                    void **scratch, size_t *scratch_bytes, hipStream_t s)
{
    size_t temp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, temp_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0, 30, s);
    if (e != hipSuccess) return e;
    const size_t a4 = (sizeof(uint32_t) * (size_t)n + 255) & ~(size_t)255;
    const size_t need = 3 * a4 + temp_bytes + 256;
    if (need > *scratch_bytes) {
        if (*scratch) (void)hipFree(*scratch);
        *scratch = nullptr; *scratch_bytes = 0;
        e = hipMalloc(scratch, need);
        if (e != hipSuccess) return e;
        *scratch_bytes = need;
    }
    char *base = static_cast<char *>(*scratch);
    uint32_t *keys_in = reinterpret_cast<uint32_t *>(base);
    uint32_t *keys_out = reinterpret_cast<uint32_t *>(base + a4);
    int32_t *vals_in = reinterpret_cast<int32_t *>(base + 2 * a4);
    void *temp = base + 3 * a4;
    hipLaunchKernelGGL(fs_sortkey_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, d_goal, grid,
                       shift_for(grid.nx), shift_for(grid.ny), shift_for(grid.nz), keys_in, vals_in);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, d_perm, (size_t)n, 0, 30, s);
}
