// fs_gridops.hip — grid staging and the frontier-cell stencil.
//
// (1) Brick scatter: a sparse map arrives as a list of 8x8x8 bricks (the wire / host format of a hashed voxel
//     map); MI355X has 288 GB of HBM, so the scorer keeps the DENSE grid resident (1024^3 = 1 GiB) and the ray
//     walk needs no hash probe per cell.  fs_upload_grid_bricks fills the grid with the default cost and
//     scatters the bricks.
// (2) Frontier-cell predicate of FrontierSearch::isNewFrontierCell (DEP/src/FrontierSearch.cpp:218-249,
//     DEP/include/.../FrontierSearch.hpp:129-142): a cell is a frontier cell iff it is unknown (255), none of its
//     in-plane 4-neighbours is lethal (>= lethal_threshold and != 255) and at least one is free
//     (< lethal_threshold).  Edge cells simply have fewer neighbours (nhood4, DEP/src/Helpers.cpp:185-220).
//     The BFS clustering around it is graph traversal and stays on the host (out of scope).
#include "fs_internal.h"

#include <algorithm>

namespace {

__global__ void fs_brick_scatter_kernel(int64_t n_bricks, const int32_t *coords, const uint8_t *cells,
                                        uint8_t *grid, int nx, int ny, int nz, int *bad)
{
    // one 64-thread wave per brick: lane = (z, y) row of 8 cells
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n_bricks) return;
    const int lane = threadIdx.x & 63;
    const int bx = coords[3 * b], by = coords[3 * b + 1], bz = coords[3 * b + 2];
    const int x0 = bx * 8, y = by * 8 + (lane & 7), z = bz * 8 + (lane >> 3);
    if (bx < 0 || by < 0 || bz < 0 || x0 + 8 > nx || by * 8 + 8 > ny || bz * 8 + 8 > nz) {
        if (lane == 0) atomicOr(bad, 1);
        return;
    }
    const uint2 row = *reinterpret_cast<const uint2 *>(cells + b * 512 + lane * 8);
    *reinterpret_cast<uint2 *>(grid + ((size_t)z * ny + y) * nx + x0) = row;
}

__global__ __launch_bounds__(256)
void fs_frontier_cells_kernel(const uint8_t *grid, int nx, int ny, int nz, int lethal_threshold,
                              uint8_t *mask, unsigned long long *count)
{
    // one thread per 16 consecutive x cells of one (y, z) row
    const int groups_x = (nx + 15) >> 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)groups_x * ny * nz;
    int found = 0;
    if (t < total) {
        const int gx = (int)(t % groups_x);
        const long long row = t / groups_x;
        const int y = (int)(row % ny);
        const size_t base = (size_t)row * nx;
        const int x0 = gx * 16;
        const int lt = lethal_threshold;
        for (int k = 0; k < 16; ++k) {
            const int x = x0 + k;
            if (x >= nx) break;
            const uint8_t c = grid[base + x];
            uint8_t out = 0;
            if (c == 255) {
                bool has_free = false, has_lethal = false;
                auto look = [&](uint8_t v) {
                    if ((int)v < lt) has_free = true;
                    if ((int)v >= lt && v != 255) has_lethal = true;
                };
                if (x > 0) look(grid[base + x - 1]);
                if (x < nx - 1) look(grid[base + x + 1]);
                if (y > 0) look(grid[base + x - nx]);
                if (y < ny - 1) look(grid[base + x + nx]);
                out = (!has_lethal && has_free) ? 1 : 0;
            }
            if (mask) mask[base + x] = out;
            found += out;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) found += __shfl_xor(found, d);
    if ((threadIdx.x & 63) == 0 && found) atomicAdd(count, (unsigned long long)found);
}

// computeInformationFrontierPair (FIP/src/.../FisherInformationHelpers.cpp:125-143) for a batch: one workgroup per
// (estimation pose, FOV triangle); every landmark whose (x, y) lies strictly inside the triangle (onLeft of all
// three edges, evaluated in fp64 like the reference's Point2D arithmetic) adds the local-Jacobian trace
// 2 + 2/|p|^2 at p = R^T (w - t).  The reference never calls it at run time; kept for interface completeness.
__global__ __launch_bounds__(256)
void fs_frontier_pair_kernel(int n, const float *lx, const float *ly, const float *lz, int m, const float *Rt,
                             const double *tri, float *out)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    if (c >= n) return;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rt[12 * (size_t)c + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = Rt[12 * (size_t)c + 9 + i];
    const double ax = tri[6 * c], ay = tri[6 * c + 1], bx = tri[6 * c + 2], by = tri[6 * c + 3], cx = tri[6 * c + 4], cy = tri[6 * c + 5];
    float sum = 0.0f;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const float x = lx[i], y = ly[i], z = lz[i];
        const bool l1 = (ax - x) * (by - y) - (ay - y) * (bx - x) > 0;
        const bool l2 = (bx - x) * (cy - y) - (by - y) * (cx - x) > 0;
        const bool l3 = (cx - x) * (ay - y) - (cy - y) * (ax - x) > 0;
        if (l1 && l2 && l3) {
            const float dx = x - t[0], dy = y - t[1], dz = z - t[2];
            const float px = __fmaf_rn(R[0], dx, __fmaf_rn(R[3], dy, R[6] * dz));
            const float py = __fmaf_rn(R[1], dx, __fmaf_rn(R[4], dy, R[7] * dz));
            const float pz = __fmaf_rn(R[2], dx, __fmaf_rn(R[5], dy, R[8] * dz));
            const float n2 = __fmaf_rn(px, px, __fmaf_rn(py, py, pz * pz));
            sum += 2.0f + 2.0f / n2;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dense row-major grid -> 2-bit class image in 8 x 8 x 8 bricks (fs_internal.h, FsGridDev::cls): one thread classifies the 16
// cells of one dword — two x-rows of a brick — from the row-major image.  The launch covers the sub-box of sub.n bricks that
// starts at brick sub.b0 (the whole image, or the bricks a rewritten window of the map touches: fs_update_grid_region)
struct BrickBox { uint32_t b0[3], n[3]; };
__global__ void fs_classify_kernel(const uint8_t *__restrict__ cells, uint32_t *__restrict__ cls, int nx, int ny, int nz,
                                   int omin, int omax, int tmin, int tmax, BrickBox sub)
{
    const uint32_t bx = (uint32_t)(nx + 7) >> 3, by = (uint32_t)(ny + 7) >> 3;
    const long long total = (long long)sub.n[0] * sub.n[1] * sub.n[2] * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(i & 31);                    // dword of the brick: (z & 7) * 4 + (y & 7) / 2
        const long long brick = i >> 5;
        const uint32_t ix = sub.b0[0] + (uint32_t)(brick % sub.n[0]), iy = sub.b0[1] + (uint32_t)((brick / sub.n[0]) % sub.n[1]),
                       iz = sub.b0[2] + (uint32_t)(brick / ((long long)sub.n[0] * sub.n[1]));
        const int x0 = (int)(ix << 3), y0 = (int)((iy << 3) | ((d & 3u) << 1)), z = (int)((iz << 3) | (d >> 2));
        uint32_t word = 0u;
        if (z < nz) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int x = x0 + (k & 7), y = y0 + (k >> 3);
                const int c = (x < nx && y < ny) ? (int)cells[((size_t)z * ny + y) * nx + x] : -1;
                const uint32_t code = ((c >= tmin && c <= tmax) ? 1u : 0u) | ((c >= omin && c <= omax) ? 2u : 0u);
                word |= code << (2 * k);
            }
        }
        cls[((((size_t)iz * by + iy) * bx + ix) << 5) | d] = word;
    }
}

// a rewritten window of the map, packed [sz][sy][sx], into the row-major image at (x0, y0, z0): one thread per cell, x fastest
__global__ void fs_window_scatter_kernel(const uint8_t *__restrict__ win, uint8_t *__restrict__ grid, int nx, int ny,
                                         int x0, int y0, int z0, int sx, int sy, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % sx), y = (int)((i / sx) % sy), z = (int)(i / ((long long)sx * sy));
        grid[((size_t)(z0 + z) * ny + (size_t)(y0 + y)) * nx + (size_t)(x0 + x)] = win[i];
    }
}

}  // namespace

size_t fs_class_image_words(int nx, int ny, int nz)
{
    return (size_t)((nx + 7) >> 3) * (size_t)((ny + 7) >> 3) * (size_t)((nz + 7) >> 3) * 32;
}

hipError_t fs_launch_classify_region(const uint8_t *d_cells, uint32_t *d_cls, int nx, int ny, int nz, int obst_min, int obst_max,
                                     int trace_min, int trace_max, const int brick0[3], const int n_bricks[3], hipStream_t s)
{
    const BrickBox sub{{(uint32_t)brick0[0], (uint32_t)brick0[1], (uint32_t)brick0[2]}, {(uint32_t)n_bricks[0], (uint32_t)n_bricks[1], (uint32_t)n_bricks[2]}};
    const size_t words = (size_t)n_bricks[0] * (size_t)n_bricks[1] * (size_t)n_bricks[2] * 32;
    if (words == 0) return hipSuccess;
    const int blocks = (int)std::min<size_t>((words + 255) / 256, 65535u * 16u);
    hipLaunchKernelGGL(fs_classify_kernel, dim3(blocks), dim3(256), 0, s, d_cells, d_cls, nx, ny, nz, obst_min, obst_max, trace_min, trace_max, sub);
    return hipGetLastError();
}

hipError_t fs_launch_classify(const uint8_t *d_cells, uint32_t *d_cls, int nx, int ny, int nz, int obst_min, int obst_max,
                              int trace_min, int trace_max, hipStream_t s)
{
    const int b0[3] = {0, 0, 0}, nb[3] = {(nx + 7) >> 3, (ny + 7) >> 3, (nz + 7) >> 3};
    return fs_launch_classify_region(d_cells, d_cls, nx, ny, nz, obst_min, obst_max, trace_min, trace_max, b0, nb, s);
}

hipError_t fs_launch_window_scatter(const uint8_t *d_window, uint8_t *d_grid, int nx, int ny, int x0, int y0, int z0,
                                    int sx, int sy, int sz, hipStream_t s)
{
    const long long total = (long long)sx * sy * sz;
    if (total <= 0) return hipSuccess;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 65535ll * 16);
    hipLaunchKernelGGL(fs_window_scatter_kernel, dim3(blocks), dim3(256), 0, s, d_window, d_grid, nx, ny, x0, y0, z0, sx, sy, total);
    return hipGetLastError();
}

hipError_t fs_launch_frontier_pair(int n, const float *lx, const float *ly, const float *lz, int m, const float *d_Rt,
                                   const double *d_tri, float *d_out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fs_frontier_pair_kernel, dim3(n), dim3(256), 0, s, n, lx, ly, lz, m, d_Rt, d_tri, d_out);
    return hipGetLastError();
}

hipError_t fs_launch_brick_scatter(int64_t n_bricks, const int32_t *d_coords, const uint8_t *d_cells, uint8_t *d_grid,
                                   int nx, int ny, int nz, int *d_bad, hipStream_t s)
{
    if (n_bricks <= 0) return hipSuccess;
    const int per_block = 4;
    hipLaunchKernelGGL(fs_brick_scatter_kernel, dim3((unsigned)((n_bricks + per_block - 1) / per_block)), dim3(64 * per_block), 0, s,
                       n_bricks, d_coords, d_cells, d_grid, nx, ny, nz, d_bad);
    return hipGetLastError();
}

hipError_t fs_launch_frontier_cells(const uint8_t *d_grid, int nx, int ny, int nz, int lethal_threshold, uint8_t *d_mask,
                                    unsigned long long *d_count, hipStream_t s)
{
    const long long total = (long long)((nx + 15) >> 4) * ny * nz;
    hipLaunchKernelGGL(fs_frontier_cells_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       d_grid, nx, ny, nz, lethal_threshold, d_mask, d_count);
    return hipGetLastError();
}
