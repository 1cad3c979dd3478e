// fs_cloud.hip — the landmark cloud's k-d leaf order computed on the device ("cloud.order" 2, and 1 from 4096 landmarks on).
//
// What fs_stage_landmarks (fs_capi.hip) does on the host — the cloud in the leaf order of a k-d tree with exactly 64 landmarks per
// leaf, split along the longest axis of a node's bounding box at the multiple of 64 nearest the median — level by level on the GPU.
// The split POSITIONS are a function of the cloud's size alone (a node of n points splits at (n / 2) / 64 * 64), so the host lays
// out every level's node boundaries up front; per level one kernel finds each node's longest axis and writes a 64-bit key
// (node index, coordinate along that axis as a sortable word) per landmark, and one stable radix sort of (key, landmark) pairs
// puts every node's landmarks in coordinate order — the first k of a node are its left child.  Ties keep the order of the level
// before (the host form breaks them by landmark index): the two forms agree on every split whose coordinate is unique and may
// deal tied landmarks differently, so chunks — and the last bits of the float columns, which follow the order of summation — are
// not the host form's in general; integer outputs never depend on the order.  The result is a function of the input alone.
#include "fs_internal.h"

#include <algorithm>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace {

__device__ __forceinline__ uint32_t sortable(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // unsigned order == float order (-0 below +0: a fixed rule)
}

// One workgroup per node of the level: bounding box of its landmarks -> longest axis -> keys.  A node that no longer splits
// (<= 64 landmarks) writes keys that leave it where it is.
// (KEY = uint32_t: the key is the coordinate word alone — for the levels whose nodes are sorted as SEGMENTS, each by itself)
template <typename KEY>
__global__ __launch_bounds__(1024)
void fs_cloud_keys_kernel(const float *__restrict__ raw, const int32_t *__restrict__ perm, const int32_t *__restrict__ bounds,
                          KEY *__restrict__ keys)
{
    __shared__ float red[6][16];
    __shared__ int s_axis;
    const int j = blockIdx.x;
    const int lo = bounds[j], hi = bounds[j + 1];
    const int n = hi - lo;
    const KEY node = sizeof(KEY) == 8 ? (KEY)((uint64_t)(uint32_t)j << 32) : (KEY)0;
    if (n <= FS_CHUNK) {
        for (int p = lo + (int)threadIdx.x; p < hi; p += blockDim.x) keys[p] = node | (KEY)(uint32_t)(p - lo);
        return;
    }
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int p = lo + (int)threadIdx.x; p < hi; p += blockDim.x) {
        const float *q = raw + 3 * (size_t)perm[p];
#pragma unroll
        for (int a = 0; a < 3; ++a) { blo[a] = fminf(blo[a], q[a]); bhi[a] = fmaxf(bhi[a], q[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d >= 1; d >>= 1) { blo[a] = fminf(blo[a], __shfl_xor(blo[a], d)); bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], d)); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a][wave] = blo[a]; red[3 + a][wave] = bhi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int waves = blockDim.x >> 6;
        float l[3], h[3];
        for (int a = 0; a < 3; ++a) {
            l[a] = red[a][0]; h[a] = red[3 + a][0];
            for (int w = 1; w < waves; ++w) { l[a] = fminf(l[a], red[a][w]); h[a] = fmaxf(h[a], red[3 + a][w]); }
        }
        int ax = 0;                                               // the host form's rule, word for word (fs_stage_landmarks)
        if (h[1] - l[1] > h[ax] - l[ax]) ax = 1;
        if (h[2] - l[2] > h[ax] - l[ax]) ax = 2;
        s_axis = ax;
    }
    __syncthreads();
    const int ax = s_axis;
    for (int p = lo + (int)threadIdx.x; p < hi; p += blockDim.x) keys[p] = node | (KEY)sortable(raw[3 * (size_t)perm[p] + ax]);
}

// The top levels — few nodes, each a large share of the cloud — in two launches instead: every node's bounding box by many
// workgroups (a slice each, folded into the node's six words with atomic min / max on the sortable form), then one thread per
// landmark finds its node, the node's longest axis (the same rule) and writes the key.  One workgroup per node took 131 us for the
// root of a 100 k-landmark cloud and half of that for each of the next levels (profiles/r05/cloud_order_kernel_stats.csv).
#define FS_CLOUD_TOP_NODES 32                                      // levels of up to this many nodes take the two-launch form
#define FS_CLOUD_SEGMENT_MAX 4096                                  // levels whose largest node is at most this sort node by node (segments)
#define FS_CLOUD_TOP_LEVELS 6                                      // ... 1, 2, 4, 8, 16, 32 nodes: the first six levels at most

__device__ __forceinline__ float unsortable(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}

__global__ __launch_bounds__(256)
void fs_cloud_top_bbox_kernel(const float *__restrict__ raw, const int32_t *__restrict__ perm, const int32_t *__restrict__ bounds,
                              uint32_t *__restrict__ bbox /* [nodes][6]: min x y z, max x y z, sortable */)
{
    __shared__ float red[6][4];
    const int j = blockIdx.y;
    const int lo = bounds[j], hi = bounds[j + 1];
    if (hi - lo <= FS_CHUNK) return;
    const int per = (hi - lo + (int)gridDim.x - 1) / (int)gridDim.x;
    const int a0 = lo + (int)blockIdx.x * per, a1 = min(hi, a0 + per);
    if (a0 >= a1) return;
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int p = a0 + (int)threadIdx.x; p < a1; p += blockDim.x) {
        const float *q = raw + 3 * (size_t)perm[p];
#pragma unroll
        for (int a = 0; a < 3; ++a) { blo[a] = fminf(blo[a], q[a]); bhi[a] = fmaxf(bhi[a], q[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d >= 1; d >>= 1) { blo[a] = fminf(blo[a], __shfl_xor(blo[a], d)); bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], d)); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a][wave] = blo[a]; red[3 + a][wave] = bhi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = red[a][0];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[a][w]) : fmaxf(v, red[a][w]);
        // (a slice whose threads all idled keeps +-inf: harmless under min / max)
        if (a < 3) atomicMin(&bbox[6 * j + a], sortable(v));
        else atomicMax(&bbox[6 * j + a], sortable(v));
    }
}

__global__ void fs_cloud_top_reset_kernel(uint32_t *bbox, int n_words)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) bbox[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
}

__global__ __launch_bounds__(256)
void fs_cloud_top_keys_kernel(const float *__restrict__ raw, const int32_t *__restrict__ perm, const int32_t *__restrict__ bounds, int nodes,
                              const uint32_t *__restrict__ bbox, int32_t n_usable, uint64_t *__restrict__ keys)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_usable) return;
    int j = 0;                                                     // the node that holds position p: bounds[j] <= p < bounds[j + 1]
    for (int step = FS_CLOUD_TOP_NODES >> 1; step >= 1; step >>= 1)
        if (j + step < nodes && bounds[j + step] <= p) j += step;
    const int lo = bounds[j], hi = bounds[j + 1];
    const uint64_t node = (uint64_t)(uint32_t)j << 32;
    if (hi - lo <= FS_CHUNK) { keys[p] = node | (uint32_t)(p - lo); return; }
    float l[3], h[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { l[a] = unsortable(bbox[6 * j + a]); h[a] = unsortable(bbox[6 * j + 3 + a]); }
    int ax = 0;                                                    // the host form's rule, word for word (fs_stage_landmarks)
    if (h[1] - l[1] > h[ax] - l[ax]) ax = 1;
    if (h[2] - l[2] > h[ax] - l[ax]) ax = 2;
    keys[p] = node | sortable(raw[3 * (size_t)perm[p] + ax]);
}

__global__ void fs_cloud_iota_kernel(int32_t *perm, int32_t n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = i;
}

// SoA in the final order; everything behind the usable landmarks (unusable ones keep a slot, the padding of the last chunk) is
// the far-away sentinel of the host form
__global__ void fs_cloud_gather_kernel(const float *__restrict__ raw, const int32_t *__restrict__ perm, int32_t n_usable, int32_t padded,
                                       float *__restrict__ lx, float *__restrict__ ly, float *__restrict__ lz)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= padded) return;
    float x = 1.0e18f, y = 1.0e18f, z = 1.0e18f;
    if (p < n_usable) {
        const float *q = raw + 3 * (size_t)perm[p];
        x = q[0]; y = q[1]; z = q[2];
    }
    lx[p] = x; ly[p] = y; lz[p] = z;
}

// one wave per chunk: the host form's sphere (centre of the bounding box, largest distance * 1.01 + 2 mm), same arithmetic
__global__ __launch_bounds__(256)
void fs_cloud_spheres_kernel(const float *__restrict__ lx, const float *__restrict__ ly, const float *__restrict__ lz, int32_t n_usable,
                             int32_t n_chunks, float *__restrict__ spheres)
{
    const int ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ch >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    const int i = ch * FS_CHUNK + lane;
    const bool in = i < n_usable;
    const double x = in ? (double)lx[i] : 0.0, y = in ? (double)ly[i] : 0.0, z = in ? (double)lz[i] : 0.0;
    double l[3] = {in ? x : 1e300, in ? y : 1e300, in ? z : 1e300}, h[3] = {in ? x : -1e300, in ? y : -1e300, in ? z : -1e300};
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d >= 1; d >>= 1) { l[a] = fmin(l[a], __shfl_xor(l[a], d)); h[a] = fmax(h[a], __shfl_xor(h[a], d)); }
    const int cnt = __popcll(__ballot(in));
    if (cnt == 0) {
        if (lane == 0) { spheres[4 * ch] = 0.0f; spheres[4 * ch + 1] = 0.0f; spheres[4 * ch + 2] = 0.0f; spheres[4 * ch + 3] = -1.0e30f; }
        return;
    }
    const float c0 = (float)(0.5 * (l[0] + h[0])), c1 = (float)(0.5 * (l[1] + h[1])), c2 = (float)(0.5 * (l[2] + h[2]));
    const double dx = x - (double)c0, dy = y - (double)c1, dz = z - (double)c2;
    double r2 = in ? dx * dx + dy * dy + dz * dz : 0.0;
    for (int d = 32; d >= 1; d >>= 1) r2 = fmax(r2, __shfl_xor(r2, d));
    if (lane == 0) {
        spheres[4 * ch] = c0; spheres[4 * ch + 1] = c1; spheres[4 * ch + 2] = c2;
        spheres[4 * ch + 3] = (float)(sqrt(r2) * 1.01 + 2.0e-3);
    }
}

}  // namespace

// Every level's node boundaries, concatenated: level L holds level_nodes[L] + 1 ascending positions from 0 to n_usable; its largest
// node has level_largest[L] landmarks.
void fs_cloud_levels(int32_t n_usable, std::vector<int32_t> &bounds, std::vector<int32_t> &level_off, std::vector<int32_t> &level_nodes,
                     std::vector<int32_t> &level_largest)
{
    bounds.clear(); level_off.clear(); level_nodes.clear(); level_largest.clear();
    std::vector<int32_t> cur{0, n_usable};
    for (;;) {
        int32_t largest = 0;
        for (size_t j = 0; j + 1 < cur.size(); ++j) largest = std::max(largest, cur[j + 1] - cur[j]);
        if (largest <= FS_CHUNK) break;
        level_largest.push_back(largest);
        level_off.push_back((int32_t)bounds.size());
        level_nodes.push_back((int32_t)cur.size() - 1);
        bounds.insert(bounds.end(), cur.begin(), cur.end());
        std::vector<int32_t> next;
        next.reserve(2 * cur.size());
        for (size_t j = 0; j + 1 < cur.size(); ++j) {
            const int32_t lo = cur[j], n = cur[j + 1] - lo;
            next.push_back(lo);
            if (n > FS_CHUNK) {
                int32_t k = (n / 2) / FS_CHUNK * FS_CHUNK;
                if (k == 0) k = FS_CHUNK;
                next.push_back(lo + k);
            }
        }
        next.push_back(n_usable);
        cur.swap(next);
    }
}

size_t fs_cloud_top_bbox_words() { return (size_t)6 * FS_CLOUD_TOP_NODES * FS_CLOUD_TOP_LEVELS; }

size_t fs_cloud_sort_temp_bytes(int32_t n, hipStream_t s)
{
    size_t bytes = 0, seg = 0;
    const size_t size = (size_t)std::max(n, 1);
    if (rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, size, 0, 64, s) != hipSuccess)
        return 0;
    if (rocprim::segmented_radix_sort_pairs(nullptr, seg, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (unsigned int)size,
                                            (unsigned int)(size / FS_CHUNK + 2), (const int32_t *)nullptr, (const int32_t *)nullptr, 0, 32, s) != hipSuccess)
        return 0;
    return std::max(bytes, seg);
}

// perm_a holds the starting order of the usable landmarks (input order); returns the final order in *perm_out (one of the two buffers)
hipError_t fs_cloud_order_device(const float *d_raw, int32_t n_usable, const int32_t *d_bounds, const std::vector<int32_t> &level_off,
                                 const std::vector<int32_t> &level_nodes, const std::vector<int32_t> &level_largest, int32_t *d_perm_a, int32_t *d_perm_b, uint64_t *d_keys_a,
                                 uint64_t *d_keys_b, void *d_temp, size_t temp_bytes, uint32_t *d_top_bbox, hipStream_t s, int32_t **perm_out)
{
    int32_t *pin = d_perm_a, *pout = d_perm_b;
    for (size_t L = 0; L < level_off.size(); ++L) {
        const int32_t nodes = level_nodes[L];
        if (nodes <= FS_CLOUD_TOP_NODES && (int)L < FS_CLOUD_TOP_LEVELS && d_top_bbox) {
            const int slices = std::max(1, std::min(256, (n_usable / nodes + 2047) / 2048));
            if (L == 0) {                                        // every top level's boxes start at (+max, -max): one launch for all of them
                const int words = 6 * FS_CLOUD_TOP_NODES * FS_CLOUD_TOP_LEVELS;
                hipLaunchKernelGGL(fs_cloud_top_reset_kernel, dim3((words + 255) / 256), dim3(256), 0, s, d_top_bbox, words);
            }
            uint32_t *bbox = d_top_bbox + 6 * FS_CLOUD_TOP_NODES * L;
            hipLaunchKernelGGL(fs_cloud_top_bbox_kernel, dim3(slices, nodes), dim3(256), 0, s, d_raw, pin, d_bounds + level_off[L], bbox);
            hipLaunchKernelGGL(fs_cloud_top_keys_kernel, dim3((n_usable + 255) / 256), dim3(256), 0, s, d_raw, pin, d_bounds + level_off[L], nodes, bbox, n_usable, d_keys_a);
        } else if (level_largest[L] <= FS_CLOUD_SEGMENT_MAX) {
            // the deep levels — many nodes, each of a few thousand landmarks at most — sort every node as a SEGMENT of its own (one
            // launch, a workgroup per node) on the coordinate word alone: a sort of the whole array is a block sort and seven merge
            // passes at 100 k pairs, 63 us per level (profiles/r05/cloud_order_kernel_stats.csv)
            uint32_t *k32_in = reinterpret_cast<uint32_t *>(d_keys_a), *k32_out = reinterpret_cast<uint32_t *>(d_keys_b);
            hipLaunchKernelGGL(fs_cloud_keys_kernel<uint32_t>, dim3(nodes), dim3(256), 0, s, d_raw, pin, d_bounds + level_off[L], k32_in);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            size_t bytes = temp_bytes;
            e = rocprim::segmented_radix_sort_pairs(d_temp, bytes, k32_in, k32_out, pin, pout, (unsigned int)n_usable, (unsigned int)nodes,
                                                    d_bounds + level_off[L], d_bounds + level_off[L] + 1, 0, 32, s);
            if (e != hipSuccess) return e;
            std::swap(pin, pout);
            continue;
        } else {
            hipLaunchKernelGGL(fs_cloud_keys_kernel<uint64_t>, dim3(nodes), dim3(nodes <= 8 ? 1024 : nodes <= 64 ? 512 : 256), 0, s, d_raw, pin, d_bounds + level_off[L], d_keys_a);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        int node_bits = 1;
        while ((1 << node_bits) < nodes) ++node_bits;
        e = rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys_a, d_keys_b, pin, pout, (size_t)n_usable, 0, 32 + node_bits, s);
        if (e != hipSuccess) return e;
        std::swap(pin, pout);
    }
    *perm_out = pin;
    return hipSuccess;
}

hipError_t fs_cloud_iota(int32_t *d_perm, int32_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fs_cloud_iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_perm, n);
    return hipGetLastError();
}

hipError_t fs_cloud_finish(const float *d_raw, const int32_t *d_perm, int32_t n_usable, int32_t n_chunks, float *d_lx, float *d_ly, float *d_lz,
                           float *d_spheres, hipStream_t s)
{
    const int32_t padded = n_chunks * FS_CHUNK;
    hipLaunchKernelGGL(fs_cloud_gather_kernel, dim3((padded + 255) / 256), dim3(256), 0, s, d_raw, d_perm, n_usable, padded, d_lx, d_ly, d_lz);
    hipLaunchKernelGGL(fs_cloud_spheres_kernel, dim3((n_chunks + 3) / 4), dim3(256), 0, s, d_lx, d_ly, d_lz, n_usable, n_chunks, d_spheres);
    return hipGetLastError();
}
