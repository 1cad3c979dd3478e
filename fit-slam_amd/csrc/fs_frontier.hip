// fs_frontier.hip — frontier detection + clustering on the GPU, gfx950.
//
// Replaces the two nested breadth-first searches of FrontierSearch::searchFrom / buildNewFrontier
// (DEP/src/FrontierSearch.cpp:21-216) by order-independent set operations that give the same cells in the same clusters:
//   E  = the cells the outer search EXPANDS: the start cell plus every cell with cost < LETHAL_OBSTACLE (254) that lies
//        within the search radius and is 4-connected, through such cells, to a neighbour of the start cell (:59-73)
//   F  = the frontier cells (isNewFrontierCell, :218-249): unknown, no lethal 4-neighbour, at least one free one
//   a frontier cell 4-adjacent to a cell of E seeds buildNewFrontier (:75-78), which collects the whole 8-connected
//   component of F around it (:119-143) — so the search finds exactly the 8-connected components of F that touch E.
// Both connectivity problems are solved with a lock-free union-find over the cell array (roots = smallest cell index
// of a component: a canonical label), three passes over the grid each, instead of a serial queue.  The start cell is
// nearestFreeCell's (DEP/src/Helpers.cpp:285-329) — an exact emulation of its queue, bounded to the ring in which it ends
// (found by a map-wide reduction) and free in the normal case of a robot standing on a free cell.  What is NOT reproduced is order-dependent by construction in the
// reference: cutting a component into pieces of max_frontier_cluster_size + 1 cells in queue order (:146-178) and the
// angular-median goal point of each piece (:158-170); the number and sizes of the pieces follow from a component's
// size, and are reported as such (fs_frontier_cluster::size).
#include "fs_internal.h"

namespace {

// (agent-scope loads: links written by other CUs' atomics live in L2, a CU's L1 is never refreshed by them)
__device__ __forceinline__ int uf_find(const int32_t *parent, int x)
{
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}

// link the larger root under the smaller one; roots only ever decrease, so the smallest index of a component ends up its root
__device__ __forceinline__ void uf_union(int32_t *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

struct FcArgs {
    const uint8_t *map;
    int32_t nx, ny;
    double ox, oy, res;
    double px, py;              // robot position
    double reach;               // max_frontier_distance + max_frontier_cluster_size * resolution * 1.414 (:67)
    int32_t lethal_threshold;
    int32_t *parent_t, *parent_f, *aux;
    int32_t *state;             // [8]: 0 start cell, 1..4 roots of the start cell's expanded neighbours (-1: none), 5 clusters found, 6 cells found,
                                //      7 Chebyshev distance from the robot's cell to the nearest cell below the lethal threshold (INT_MAX: none)
    int32_t pos;                // the robot's cell
};

// per cell: is it expandable (T) / a frontier cell (F)?  One wave takes 64 consecutive cells of a row (grid: ny rows x
// ceil(nx / 64) segments).  F starts as singletons.  T starts linked by RUNS: the ballot of the T flags gives every lane the
// start of its horizontal run inside the segment (the nearest non-T lane below it) — a smaller index of its own component,
// written as its parent with a plain store before any union runs.
__global__ void fs_fc_classify_kernel(const FcArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int segs = (a.nx + 63) >> 6;
    const int y = wave / segs, seg = wave - y * segs;
    if (y >= a.ny) return;
    const int x = seg * 64 + lane;
    const bool inside = x < a.nx;
    const int i = y * a.nx + x;
    const int c = inside ? (int)a.map[i] : 254;
    // :62-68 — cost below LETHAL_OBSTACLE and inside the search radius (same fp64 expression: mapToWorld, pow(.,2), sqrt)
    const double wx = a.ox + (x + 0.5) * a.res, wy = a.oy + (y + 0.5) * a.res;
    const double dx = a.px - wx, dy = a.py - wy;
    const bool t = inside && c < 254 && sqrt(dx * dx + dy * dy) < a.reach;
    bool f = false;
    if (inside && c == 255) {                                               // :221
        bool has_free = false, has_lethal = false;
        auto look = [&](int j) {
            const int v = a.map[j];
            if (v < a.lethal_threshold) has_free = true;                    // isFree
            if (v >= a.lethal_threshold && v != 255) has_lethal = true;     // isLethal
        };
        if (x > 0) look(i - 1);
        if (x < a.nx - 1) look(i + 1);
        if (y > 0) look(i - a.nx);
        if (y < a.ny - 1) look(i + a.nx);
        f = !has_lethal && has_free;                                        // :241-246
    }
    const unsigned long long bt = __ballot(t);
    if (inside) {
        const unsigned long long gaps = ~bt & (lane ? (~0ull >> (64 - lane)) : 0ull);   // non-T lanes below mine
        const int start = gaps ? 64 - __builtin_clzll(gaps) : 0;                       // first lane of my run in this segment
        a.parent_t[i] = t ? i - (lane - start) : -1;
        a.parent_f[i] = f ? i : -1;
        a.aux[i] = -1;
    }
    if (wave == 0 && lane == 0) a.state[7] = 0x7fffffff;                    // (reduced by the next kernel)
}

// Unions.  T (the expandable cells, 4-connected) is large — the whole mapped free space — and a union per cell pair made
// this kernel 0.31 ms on a 512^2 costmap: every atomic of a big component ends at the same few roots.  T arrives linked by
// horizontal runs (classify kernel); what is left for atomics is one union per segment boundary a run crosses and one per
// STRETCH in which a run touches the row below (the first cell of every maximal stretch of vertically adjacent pairs) —
// runs, not cells.  F (the frontier cells, 8-connected) is
// a sparse set of thin curves: a union per forward neighbour (right, down, down-left, down-right), each pair once.
// Grid: ny rows x ceil(nx / 64) segments, one wave each.
__global__ void fs_fc_union_kernel(const FcArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int segs = (a.nx + 63) >> 6;
    const int y = wave / segs, seg = wave - y * segs;
    if (y >= a.ny) return;
    const int x = seg * 64 + lane;
    const bool inside = x < a.nx;
    const int i = y * a.nx + x;
    const bool down = y < a.ny - 1;
    // how far nearestFreeCell will have to look: its search is an unobstructed 8-connected flood from the robot's cell, i.e. it
    // reaches cells in rings of growing Chebyshev distance and stops in the first ring that holds a cell below the threshold
    if (inside && (int)a.map[i] < (a.lethal_threshold & 0xff)) {
        const int py = a.pos / a.nx, px = a.pos - py * a.nx;
        const int d = max(abs(x - px), abs(y - py));
        if (d < a.state[7]) atomicMin(&a.state[7], d);
    }
    const bool t0 = inside && a.parent_t[i] >= 0;
    const bool t1 = t0 && down && a.parent_t[i + a.nx] >= 0;
    const unsigned long long both = __ballot(t1);
    if (t0) {
        // a run that enters from the previous segment: one union at the boundary
        if (lane == 0 && seg > 0 && a.parent_t[i - 1] >= 0) uf_union(a.parent_t, i, i - 1);
        // first cell of a stretch of vertical pairs (the pair to my left is not one; across a segment boundary the stretch
        // simply starts again: one union more, never one less)
        if (t1 && (lane == 0 || !((both >> (lane - 1)) & 1ull))) uf_union(a.parent_t, i, i + a.nx);
    }
    if (inside && a.parent_f[i] >= 0) {
        const bool right = x < a.nx - 1, left = x > 0;
        if (right && a.parent_f[i + 1] >= 0) uf_union(a.parent_f, i, i + 1);
        if (down && a.parent_f[i + a.nx] >= 0) uf_union(a.parent_f, i, i + a.nx);
        if (down && left && a.parent_f[i + a.nx - 1] >= 0) uf_union(a.parent_f, i, i + a.nx - 1);
        if (down && right && a.parent_f[i + a.nx + 1] >= 0) uf_union(a.parent_f, i, i + a.nx + 1);
    }
}

__global__ void fs_fc_flatten_kernel(const FcArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nx * a.ny) return;
    if (a.parent_t[i] >= 0) a.parent_t[i] = uf_find(a.parent_t, i);
    if (a.parent_f[i] >= 0) a.parent_f[i] = uf_find(a.parent_f, i);
}

// One wave: the start cell of the outer search (:44-54) and the components of T its neighbours expand into.
// nearestFreeCell (DEP/src/Helpers.cpp:285-329) returns the first cell IN THE ORDER OF ITS QUEUE whose cost is below
// (unsigned char)lethal_threshold; the queue order is emulated exactly, by one lane — but only inside the box the search can
// reach before it ends: state[7] (reduced over the whole map by the kernels before) is the ring in which it ends, so the
// emulation touches at most (2 d + 3)^2 cells, the wave clears just that box of the visited flags, a robot on a free cell
// (d = 0, the normal case) costs nothing, and a map without any such cell (the reference walks its whole queue and gives
// up, :51-54) is answered at once.
__global__ void fs_fc_start_kernel(const FcArgs a, uint32_t *queue, uint8_t *visited)
{
    if (blockIdx.x != 0) return;
    const int pos = a.pos;
    const int val = a.lethal_threshold & 0xff;                              // the parameter is an unsigned char (Helpers.hpp:141)
    const int ring = a.state[7];
    int start = pos;
    if (ring != 0x7fffffff && ring > 0) {
        const int py = pos / a.nx, px = pos - py * a.nx;
        const int x0 = max(0, px - ring - 1), x1 = min(a.nx - 1, px + ring + 1), y0 = max(0, py - ring - 1), y1 = min(a.ny - 1, py + ring + 1);
        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
        for (int t = threadIdx.x; t < w * h; t += blockDim.x) visited[(y0 + t / w) * a.nx + x0 + t % w] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t head = 0, tail = 0;
            queue[tail++] = (uint32_t)pos;
            visited[pos] = 1;
            while (head < tail) {
                const int idx = (int)queue[head++];
                if (a.map[idx] < val) { start = idx; break; }
                const int y = idx / a.nx, x = idx - y * a.nx;
                const bool l = x > x0, r = x < x1, u = y > y0, d = y < y1;  // (cells beyond the box are never dequeued before the search ends)
                // nhood8's order (Helpers.cpp:199-252): -1, +1, -nx, +nx, -1-nx, -1+nx, +1-nx, +1+nx
                const int nb[8] = {l ? idx - 1 : -1, r ? idx + 1 : -1, u ? idx - a.nx : -1, d ? idx + a.nx : -1,
                                   (l && u) ? idx - 1 - a.nx : -1, (l && d) ? idx - 1 + a.nx : -1,
                                   (r && u) ? idx + 1 - a.nx : -1, (r && d) ? idx + 1 + a.nx : -1};
                for (int k = 0; k < 8; ++k) {
                    const int j = nb[k];
                    if (j >= 0 && !visited[j]) { queue[tail++] = (uint32_t)j; visited[j] = 1; }
                }
            }
        }
    }
    if (threadIdx.x != 0) return;
    a.state[0] = start;
    const int y = start / a.nx, x = start - y * a.nx;
    const int nb[4] = {x > 0 ? start - 1 : -1, x < a.nx - 1 ? start + 1 : -1, y > 0 ? start - a.nx : -1, y < a.ny - 1 ? start + a.nx : -1};
    for (int k = 0; k < 4; ++k) a.state[1 + k] = (nb[k] >= 0 && a.parent_t[nb[k]] >= 0) ? a.parent_t[nb[k]] : -1;
    a.state[5] = 0;
    a.state[6] = 0;
}

__device__ __forceinline__ bool fc_expanded(const FcArgs &a, int j)
{
    if (j == a.state[0]) return true;
    const int r = a.parent_t[j];
    return r >= 0 && (r == a.state[1] || r == a.state[2] || r == a.state[3] || r == a.state[4]);
}

// a frontier cell next to an expanded cell marks its component as found (aux[root] = 0)
__global__ void fs_fc_seed_kernel(const FcArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nx * a.ny) return;
    const int root = a.parent_f[i];
    if (root < 0) return;
    const int y = i / a.nx, x = i - y * a.nx;
    const bool seed = (x > 0 && fc_expanded(a, i - 1)) || (x < a.nx - 1 && fc_expanded(a, i + 1)) ||
                      (y > 0 && fc_expanded(a, i - a.nx)) || (y < a.ny - 1 && fc_expanded(a, i + a.nx));
    if (seed) a.aux[root] = 0;
}

// found roots get a dense cluster slot
__global__ void fs_fc_enumerate_kernel(const FcArgs a, int32_t max_clusters, fs_frontier_cluster *clusters, long long *sums)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nx * a.ny) return;
    if (a.parent_f[i] != i || a.aux[i] != 0) return;
    const int slot = atomicAdd(&a.state[5], 1);
    a.aux[i] = slot < max_clusters ? slot + 1 : -2;                         // slot + 1 (> 0), or -2: no room in the caller's array
    if (slot < max_clusters) {
        fs_frontier_cluster c;
        c.label = i; c.size = 0; c.centroid_x = 0.0; c.centroid_y = 0.0;
        c.min_x = a.nx; c.min_y = a.ny; c.max_x = -1; c.max_y = -1;
        clusters[slot] = c;
        sums[2 * slot] = 0; sums[2 * slot + 1] = 0;
    }
}

__global__ void fs_fc_collect_kernel(const FcArgs a, int32_t *labels, fs_frontier_cluster *clusters, long long *sums)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nx * a.ny) return;
    const int root = a.parent_f[i];
    int label = -1;
    if (root >= 0 && a.aux[root] != -1 && a.aux[root] != 0) {
        label = root;
        atomicAdd(&a.state[6], 1);
        const int slot = a.aux[root] - 1;
        if (slot >= 0) {
            const int y = i / a.nx, x = i - y * a.nx;
            atomicAdd(&clusters[slot].size, 1);
            atomicAdd(reinterpret_cast<unsigned long long *>(&sums[2 * slot]), (unsigned long long)x);
            atomicAdd(reinterpret_cast<unsigned long long *>(&sums[2 * slot + 1]), (unsigned long long)y);
            atomicMin(&clusters[slot].min_x, x); atomicMin(&clusters[slot].min_y, y);
            atomicMax(&clusters[slot].max_x, x); atomicMax(&clusters[slot].max_y, y);
        }
    }
    if (labels) labels[i] = label;
}

// centroid of the cell centres (mapToWorld of the mean index: exact integer sums, one division)
__global__ void fs_fc_finish_kernel(const FcArgs a, int32_t n, fs_frontier_cluster *clusters, const long long *sums)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || k >= a.state[5]) return;
    const double inv = 1.0 / (double)clusters[k].size;
    clusters[k].centroid_x = a.ox + ((double)sums[2 * k] * inv + 0.5) * a.res;
    clusters[k].centroid_y = a.oy + ((double)sums[2 * k + 1] * inv + 0.5) * a.res;
}

}  // namespace

// scratch: parent_t, parent_f, aux [cells] int32; queue [cells] uint32; visited [cells] uint8; state [8] int32; sums [2 * max_clusters] int64
hipError_t fs_launch_frontier_clusters(const uint8_t *d_map, int nx, int ny, double ox, double oy, double res, double px, double py,
                                       int32_t start_pos, double reach, int32_t lethal_threshold, int32_t *d_parent_t, int32_t *d_parent_f,
                                       int32_t *d_aux, uint32_t *d_queue, uint8_t *d_visited, int32_t *d_state, int32_t *d_labels,
                                       int32_t max_clusters, fs_frontier_cluster *d_clusters, long long *d_sums, hipStream_t s)
{
    FcArgs a{d_map, nx, ny, ox, oy, res, px, py, reach, lethal_threshold, d_parent_t, d_parent_f, d_aux, d_state, start_pos};
    const int n = nx * ny;
    const dim3 grid((n + 255) / 256), block(256);
    const long long row_waves = (long long)ny * ((nx + 63) >> 6);               // one wave per 64 cells of a row
    const dim3 row_grid((unsigned)((row_waves + 3) / 4));
    hipLaunchKernelGGL(fs_fc_classify_kernel, row_grid, block, 0, s, a);
    hipLaunchKernelGGL(fs_fc_union_kernel, row_grid, block, 0, s, a);
    hipLaunchKernelGGL(fs_fc_flatten_kernel, grid, block, 0, s, a);
    hipLaunchKernelGGL(fs_fc_start_kernel, dim3(1), dim3(256), 0, s, a, d_queue, d_visited);
    hipLaunchKernelGGL(fs_fc_seed_kernel, grid, block, 0, s, a);
    hipLaunchKernelGGL(fs_fc_enumerate_kernel, grid, block, 0, s, a, max_clusters, d_clusters, d_sums);
    hipLaunchKernelGGL(fs_fc_collect_kernel, grid, block, 0, s, a, d_labels, d_clusters, d_sums);
    if (max_clusters > 0) hipLaunchKernelGGL(fs_fc_finish_kernel, dim3((max_clusters + 255) / 256), block, 0, s, a, max_clusters, d_clusters, d_sums);
    return hipGetLastError();
}
