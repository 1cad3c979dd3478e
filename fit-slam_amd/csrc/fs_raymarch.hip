// fs_raymarch.hip — arrival-information kernel for gfx950 (CDNA4).
//
// Replaces the scalar loops of FrontierCostCalculator::setArrivalInformationForFrontier
// (DEP/src/CostCalculator.cpp:23-121) and its callees getTracedCells / bresenham2D /
// RayTracedCells::operator() / isRobotFootprintInLethal (DEP/src/Helpers.cpp:7-96,135-155,
// DEP/include/.../Helpers.hpp:50-77) for a whole frontier list at once.
//
// Mapping: one 64-lane wavefront per candidate, rays strided over the lanes (ray r = lane + 64*j),
// four candidates per 256-thread workgroup.  The grid and the fan-direction table were staged to HBM
// once (fs_upload_grid / fs_set_ray_params).  Per ray: the end point, the clamp, both worldToMap
// truncations, the step count `(unsigned)(scale*abs_da)` are evaluated in fp64 with the exact
// operation order of the reference (basic IEEE ops only: +, -, /, *, sqrt of an exact integer), so
// every integer cell index matches the CPU bit for bit; then an integer Bresenham walk reads one
// byte per step.  Per-yaw sums live in LDS; the FOV window maximum and its first argmax are a
// wave-shuffle reduction.  This is HBM/L2-bound byte gather: no MFMA.
#include "fs_internal.h"

#ifndef FS_RAY_WAVES
#define FS_RAY_WAVES 4          // candidates per workgroup
#endif
#ifndef FS_RAY_UNROLL
#define FS_RAY_UNROLL 4         // speculative cell loads in flight per lane
#endif

namespace {

__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }   // std::min(a,b)
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }   // std::max(a,b)

// (unsigned)((w - origin) / resolution), the quotient of Costmap2D::worldToMap, without the fp64 division in the common
// case.  Only the truncated quotient matters: t = a * (1 / res) lies within 4 ulp (< 1.5e-6 below 2^32) of the correctly
// rounded a / res, so both truncate to the same cell unless t sits within 1e-5 of an integer — then, and for quotients
// next to 2^32 or NaN, the division itself is evaluated (a few lanes in a million).  `q` is what the caller compares and
// truncates: identical decisions to the division in every case.
__device__ __forceinline__ double cell_quotient(double a, double res, double inv_res)
{
    const double t = a * inv_res;
    const double fr = __builtin_amdgcn_fract(t);                       // t - floor(t), in [0, 1)
    if (t < 4294967295.0 && fabs(fr - 0.5) < 0.5 - 1.0e-5) return t;
    return a / res;
}

// nav2_costmap_2d::Costmap2D::worldToMap with a z axis. Quotients >= 2^32 are off-map.
__device__ __forceinline__ bool world_to_map(const FsGridDev &g, double wx, double wy, double wz,
                                             uint32_t &mx, uint32_t &my, uint32_t &mz)
{
    if (wx < g.ox || wy < g.oy || wz < g.oz) return false;
    const double inv_res = 1.0 / g.res;                                // uniform: hoisted out of the ray loops
    const double qx = cell_quotient(wx - g.ox, g.res, inv_res);
    const double qy = cell_quotient(wy - g.oy, g.res, inv_res);
    const double qz = cell_quotient(wz - g.oz, g.res, inv_res);
    if (!(qx < 4294967296.0) || !(qy < 4294967296.0) || !(qz < 4294967296.0)) return false;
    mx = (uint32_t)qx;
    my = (uint32_t)qy;
    mz = (uint32_t)qz;
    return mx < (uint32_t)g.nx && my < (uint32_t)g.ny && mz < (uint32_t)g.nz;
}

__device__ __forceinline__ int sign_ref(int x) { return x > 0 ? 1 : -1; }   // Helpers.hpp:113-116

// Two interchangeable walks over the same cell sequence (selected per launch, FsRayArgs::layout; DESIGN.md 4.1 holds the
// measurements, including the five other formulations that were built, measured and removed):
//
// WalkLinear — the reference's own formulation on the dense row-major image: a linear offset, constant strides per
// axis, bresenham2D's body (DEP/src/Helpers.cpp:21-27) with a second minor axis.  Cheapest in instructions; rows only give
// x-major rays any cache-line reuse.  The winner for short rays (up to ~96 cells).
struct WalkLinear {
    uint32_t offset;
    uint32_t abs_da, abs_db, abs_dc;
    int err_b, err_c;
    int off_a, off_b, off_c;
    uint32_t end;            // min(max_length_steps, abs_da): loop visits, one more after the loop
};

// WalkClass — on the 2-bit class image (FsGridDev::cls: bricks of 8 x 8 x 8 cells = one 128-B cache line), which has no
// constant strides.  The walk keeps the cell coordinates IN THE RAY'S OWN AXIS ORDER (a = major, b, c = minors): the major
// coordinate steps unconditionally (the reference's `offset += offset_a`), the minors on their error terms exactly as
// bresenham2D's `error_b` (DEP/src/Helpers.cpp:21-27).  A cell's address is separable over the axes,
//     A = sum over v in (a, b, c) of v * l_v + (v >> 3) * m_v,
// with per-lane constants (l, m) that say which grid axis v is — a shift and two full-rate 24-bit multiply-adds per
// axis — and a cell arrives as its class (bit 0: in the trace range, bit 1: in the obstacle range), so the visitor is two
// bit tests.  More instructions per step than WalkLinear, 4-8x fewer cache lines per ray.
struct WalkClass {
    // cell coordinates along the major and the two minor axes, each PRE-SHIFTED by its axis' position inside a brick (x: 0, y: 3,
    // z: 6 bits): coordinate << s is the axis' share of the in-brick address, so the address needs no multiply for it
    uint32_t a, b, c;
    int sga, sgb, sgc;       // sign(d) with sign(0) = -1 (Helpers.hpp:113-116), shifted likewise: what a step adds
    uint32_t abs_da, adb, adc;
    int eb, ec;
    uint32_t ha, hb, hc;     // s + 3: (coordinate << s) >> (s + 3) = the axis' brick index
    uint32_t ma, mb, mc;     // FsGridDev::cls_m[axis]: brick stride of the axis minus (8 << s)
    uint32_t end;
};

// WalkSparse — WalkClass over the SPARSE class image (FsGridDev::cls_table): the same coordinates and steps, one more
// dependent load per cell (the brick's slot in the pool).  Built as the one measurement of BASELINE.json configs[4]'s "sparse"
// (profiles/EXPERIMENTS.md holds the outcome).
struct WalkSparse : WalkClass {};

// one minor axis of a walk: e += |d_minor|; if (e >= |d_major|) { v += sign; e -= |d_major|; } in five instructions —
// the subtraction's borrow IS the comparison (v_sub_co_u32), the smaller of e and e - |d_major| (unsigned wrap) is the new
// error term.  (The compiler spends a sixth on a separate compare.)
__device__ __forceinline__ void minor_step(uint32_t &v, int &e, uint32_t ad, uint32_t da, int sg)
{
    uint32_t t;
    asm("v_add_u32 %[e], %[e], %[ad]\n\t"
        "v_sub_co_u32 %[t], vcc, %[e], %[da]\n\t"
        "v_min_u32 %[e], %[e], %[t]\n\t"
        "v_cndmask_b32 %[t], %[sg], 0, vcc\n\t"
        "v_add_u32 %[v], %[v], %[t]"
        : [e] "+v"(e), [v] "+v"(v), [t] "=&v"(t)
        : [ad] "v"(ad), [da] "v"(da), [sg] "v"(sg)
        : "vcc");
}

__device__ __forceinline__ void walk_step(WalkLinear &w)
{
    w.offset += (uint32_t)w.off_a;
    minor_step(w.offset, w.err_b, w.abs_db, w.abs_da, w.off_b);
    minor_step(w.offset, w.err_c, w.abs_dc, w.abs_da, w.off_c);
}

__device__ __forceinline__ void walk_step(WalkClass &w)
{
    w.a += (uint32_t)w.sga;
    minor_step(w.b, w.eb, w.adb, w.abs_da, w.sgb);
    minor_step(w.c, w.ec, w.adc, w.abs_da, w.sgc);
}

__device__ __forceinline__ void walk_step(WalkSparse &w) { walk_step(static_cast<WalkClass &>(w)); }

// The cell under the walk.  Both end points are on the map (worldToMap succeeded) and a Bresenham walk between two
// cells never leaves their bounding box — each axis takes at most |d_axis| steps towards the end point — so every
// visit, including the speculative ones (they stay within `visits`), is inside the grid.
__device__ __forceinline__ int walk_cell(const FsGridDev &g, const WalkLinear &w)
{
#ifdef FS_RAY_BOUNDS   // development: verify the claim above on every visit instead of relying on it
    if (w.offset >= (uint32_t)g.nx * (uint32_t)g.ny * (uint32_t)g.nz) { atomicMax(g.dbg, 1ull); return 256; }
#endif
    return (int)g.cells[w.offset];
}
// the class (0..3) of the cell under the walk
__device__ __forceinline__ uint32_t walk_class(const FsGridDev &g, const WalkClass &w)
{
    // (24-bit multiplies run at full rate, 32-bit ones at a quarter of it; brick indices and strides are far below 2^24 —
    // fs_capi.hip only selects this walk where they are)
    // v * (1 << s) + (v >> 3) * m per axis: the first terms are the pre-shifted coordinates themselves (one three-operand add),
    // the second three shifts and three multiply-adds — seven instructions where products of the unshifted coordinates took eleven
    uint32_t A = w.a + w.b + w.c;
    A = (uint32_t)__umul24(w.c >> w.hc, w.mc) + A;
    A = (uint32_t)__umul24(w.b >> w.hb, w.mb) + A;
    A = (uint32_t)__umul24(w.a >> w.ha, w.ma) + A;
#ifdef FS_RAY_BOUNDS
    if (A >= g.cls_cells) { atomicMax(g.dbg, 6ull); return 0u; }
#endif
    const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(g.cls) + ((A >> 2) & ~3u));
    return __builtin_amdgcn_ubfe(word, A << 1, 2u);            // v_bfe_u32 takes the offset modulo 32: field A & 15
}

// ... through the brick table: A >> 9 is the brick (brick-linear order), A & 511 the cell inside it
__device__ __forceinline__ uint32_t walk_class(const FsGridDev &g, const WalkSparse &w)
{
    uint32_t A = w.a + w.b + w.c;
    A = (uint32_t)__umul24(w.c >> w.hc, w.mc) + A;
    A = (uint32_t)__umul24(w.b >> w.hb, w.mb) + A;
    A = (uint32_t)__umul24(w.a >> w.ha, w.ma) + A;
#ifdef FS_RAY_BOUNDS
    if (A >= g.cls_cells) { atomicMax(g.dbg, 6ull); return 0u; }
#endif
    const uint32_t slot = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(g.cls_table) + ((A >> 9) << 2));
    const uint32_t B = (slot << 9) | (A & 511u);
    const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(g.cls) + ((B >> 2) & ~3u));
    return __builtin_amdgcn_ubfe(word, A << 1, 2u);
}

// getTracedCells from the two map cells on (Helpers.cpp:46-94): `(unsigned)(scale * abs_da)` visits with
// scale = min(1, max_length / hypot(d)).  Evaluated as the reference does (fp64 square root, division, product) only
// where it could matter:
//   * |d|^2 <= max_length^2 in integers (max_length a whole number): hypot(d) <= max_length, the quotient is >= 1 and the
//     scale exactly 1;
//   * otherwise s = |d_major| max_length / |d| in fp32 (v_rsq_f32; relative error < 1e-6) truncates like the fp64 chain
//     (relative error 3e-16) unless it lies within 4e-6 s + 1e-6 of an integer — there the fp64 chain decides.
__device__ __forceinline__ uint32_t walk_visits(int dx, int dy, int dz, uint32_t abs_da, double max_length)
{
    const long long d2 = (long long)dx * dx + (long long)dy * dy + (long long)dz * dz;
    // (a whole number of cells below 2^26 — the arrival fan's case; fs_trace_segments may pass any double — squares exactly)
    const bool whole = max_length < 67108864.0 && max_length == floor(max_length);
    if (whole && d2 < (1ll << 52) && (double)d2 <= max_length * max_length) return abs_da;
    if (d2 < (1ll << 24) && max_length < 16777216.0) {
        const float s = ((float)abs_da * (float)max_length) * __builtin_amdgcn_rsqf((float)d2);
        const float fr = s - floorf(s);
        const float guard = 4.0e-6f * s + 1.0e-6f;
        if (fr > guard && fr < 1.0f - guard) {
            const uint32_t max_steps = (uint32_t)s;
            return max_steps < abs_da ? max_steps : abs_da;
        }
    }
    const double dist = sqrt((double)d2);          // == std::hypot(dx,dy) when dz == 0 (both correctly rounded)
    const double q = max_length / dist;
    const double scale = (dist == 0.0) ? 1.0 : ((q < 1.0) ? q : 1.0);       // std::min(1.0, max_length / dist)
    const uint32_t max_steps = (uint32_t)(scale * (double)abs_da);
    return max_steps < abs_da ? max_steps : abs_da;
}

__device__ __forceinline__ void walk_init(WalkLinear &w, const FsGridDev &g, uint32_t x0, uint32_t y0, uint32_t z0,
                                          uint32_t x1, uint32_t y1, uint32_t z1, double max_length)
{
    const int dx = (int)(x1 - x0), dy = (int)(y1 - y0), dz = (int)(z1 - z0);
    const uint32_t nx = (uint32_t)g.nx, ny = (uint32_t)g.ny;
    w.offset = (z0 * ny + y0) * nx + x0;
    const uint32_t adx = (uint32_t)abs(dx), ady = (uint32_t)abs(dy), adz = (uint32_t)abs(dz);
    const int odx = sign_ref(dx), ody = sign_ref(dy) * (int)nx, odz = sign_ref(dz) * (int)(nx * ny);
    if (adx >= ady && adx >= adz) {
        w.abs_da = adx; w.abs_db = ady; w.abs_dc = adz; w.off_a = odx; w.off_b = ody; w.off_c = odz;
    } else if (ady >= adz) {
        w.abs_da = ady; w.abs_db = adx; w.abs_dc = adz; w.off_a = ody; w.off_b = odx; w.off_c = odz;
    } else {
        w.abs_da = adz; w.abs_db = adx; w.abs_dc = ady; w.off_a = odz; w.off_b = odx; w.off_c = ody;
    }
    w.err_b = w.err_c = (int)(w.abs_da / 2);
    w.end = walk_visits(dx, dy, dz, w.abs_da, max_length);
}

__device__ __forceinline__ void walk_init(WalkClass &w, const FsGridDev &g, uint32_t x0, uint32_t y0, uint32_t z0,
                                          uint32_t x1, uint32_t y1, uint32_t z1, double max_length)
{
    const int dx = (int)(x1 - x0), dy = (int)(y1 - y0), dz = (int)(z1 - z0);
    const uint32_t adx = (uint32_t)abs(dx), ady = (uint32_t)abs(dy), adz = (uint32_t)abs(dz);
    const int sx = sign_ref(dx), sy = sign_ref(dy), sz = sign_ref(dz);
    // the axis order of walk_init(WalkLinear): the first largest extent is the major axis
    if (adx >= ady && adx >= adz) {
        w.a = x0; w.b = y0 << 3; w.c = z0 << 6; w.sga = sx; w.sgb = sy * 8; w.sgc = sz * 64; w.abs_da = adx; w.adb = ady; w.adc = adz;
        w.ha = 3; w.hb = 6; w.hc = 9; w.ma = g.cls_m[0]; w.mb = g.cls_m[1]; w.mc = g.cls_m[2];
    } else if (ady >= adz) {
        w.a = y0 << 3; w.b = x0; w.c = z0 << 6; w.sga = sy * 8; w.sgb = sx; w.sgc = sz * 64; w.abs_da = ady; w.adb = adx; w.adc = adz;
        w.ha = 6; w.hb = 3; w.hc = 9; w.ma = g.cls_m[1]; w.mb = g.cls_m[0]; w.mc = g.cls_m[2];
    } else {
        w.a = z0 << 6; w.b = x0; w.c = y0 << 3; w.sga = sz * 64; w.sgb = sx; w.sgc = sy * 8; w.abs_da = adz; w.adb = adx; w.adc = ady;
        w.ha = 9; w.hb = 3; w.hc = 6; w.ma = g.cls_m[2]; w.mb = g.cls_m[0]; w.mc = g.cls_m[1];
    }
    w.eb = w.ec = (int)(w.abs_da / 2);
    w.end = walk_visits(dx, dy, dz, w.abs_da, max_length);
}

__device__ __forceinline__ void walk_init(WalkSparse &w, const FsGridDev &g, uint32_t x0, uint32_t y0, uint32_t z0,
                                          uint32_t x1, uint32_t y1, uint32_t z1, double max_length)
{
    walk_init(static_cast<WalkClass &>(w), g, x0, y0, z0, x1, y1, z1, max_length);
}

// The start cell is the same for every ray of a fan: its worldToMap (three fp64 divisions) is done once per candidate;
// a start point off the map fails every ray, like the reference's `||` of the two conversions (Helpers.cpp:40).
template <typename Walk>
__device__ __forceinline__ bool ray_setup(const FsRayArgs &a, bool start_ok, uint32_t x0, uint32_t y0, uint32_t z0,
                                          double wx, double wy, double wz, Walk &w)
{
    uint32_t x1, y1, z1;
    if (!world_to_map(a.grid, wx, wy, wz, x1, y1, z1) || !start_ok) return false;
    walk_init(w, a.grid, x0, y0, z0, x1, y1, z1, (double)a.max_length);
    return true;
}

// RayTracedCells over the walk: number of cells in [trace_min,trace_max] seen before the first
// cell in [obst_min,obst_max]  (= cells_.size(), DEP/src/CostCalculator.cpp:57-58).
template <typename Walk>
__device__ __forceinline__ int ray_march(const FsRayArgs &a, Walk w)
{
    // the visitor's inclusive ranges, intersected with the byte range of a cell, as one unsigned compare each; an empty
    // range (e.g. the obstacle range (260, 260) of setMaxArrivalInformation) matches nothing
    const int omin = a.obst_min > 0 ? a.obst_min : 0, omax = a.obst_max < 255 ? a.obst_max : 255;
    const int tmin = a.trace_min > 0 ? a.trace_min : 0, tmax = a.trace_max < 255 ? a.trace_max : 255;
    const bool o_any = omax >= omin, t_any = tmax >= tmin;
    const uint32_t orange = o_any ? (uint32_t)(omax - omin) : 0u, trange = t_any ? (uint32_t)(tmax - tmin) : 0u;
    int count = 0;
    uint32_t visits = w.end + 1;          // loop visits + the one after the loop (Helpers.cpp:29)
    // full groups of FS_RAY_UNROLL visits first: every visit of the group exists, so the loads and steps need no per-step
    // bounds (one exec mask per group instead of one per step); the remainder (< FS_RAY_UNROLL visits) goes through the guarded
    // loop below.  Loads past an obstacle are speculative but always inside the start/end bounding box, hence in the grid.
    bool stopped = false;
    while (visits >= FS_RAY_UNROLL) {
        int c[FS_RAY_UNROLL];
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL; ++u) {
            c[u] = walk_cell(a.grid, w);
            walk_step(w);                   // (the step behind a ray's very last visit leads nowhere that is read)
        }
        bool hit = false;
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL; ++u) {
            const bool traced = t_any && !hit && (uint32_t)(c[u] - tmin) <= trange;
            count += traced ? 1 : 0;
            hit = hit || (o_any && (uint32_t)(c[u] - omin) <= orange);
        }
        if (hit) { stopped = true; break; }  // nothing is pushed after the first obstacle
        visits -= FS_RAY_UNROLL;
    }
    if (stopped) visits = 0;
    while (visits > 0) {
        // the last, partial group: up to FS_RAY_UNROLL - 1 independent byte loads, then classify them in order
        int c[FS_RAY_UNROLL];
        const uint32_t nb = visits < FS_RAY_UNROLL ? visits : FS_RAY_UNROLL;
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL; ++u) {
            c[u] = -1;
            if ((uint32_t)u < nb) {
                c[u] = walk_cell(a.grid, w);
                walk_step(w);
            }
        }
        // classify in order, without branches: a cell counts if no earlier cell of the ray was an obstacle
        // (Helpers.hpp:64-71: the traced test comes before hit_obstacle is set, so an obstacle cell itself still counts
        // when the two ranges overlap); slots past `nb` hold -1 and match no range
        bool hit = false;
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL; ++u) {
            const bool traced = t_any && !hit && (uint32_t)(c[u] - tmin) <= trange;
            count += traced ? 1 : 0;
            hit = hit || (o_any && (uint32_t)(c[u] - omin) <= orange);
        }
        if (hit) break;                    // nothing is pushed after the first obstacle
        visits -= nb;
    }
    return count;
}

// The visitor over the class image (the classes were cut with this launch's ranges: fs_capi.hip keeps them in step): bit 0
// counts while no earlier cell of the ray carried bit 1 (Helpers.hpp:64-71 — the traced test comes before hit_obstacle is set,
// so a cell in both ranges still counts).
template <typename W>
__device__ __forceinline__ int ray_march_classes(const FsRayArgs &a, W w)
{
    int count = 0;
    uint32_t visits = w.end + 1;
    uint32_t open = 1u;                              // 1 until the first obstacle
    while (visits >= FS_RAY_UNROLL) {
        uint32_t c[FS_RAY_UNROLL];
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL; ++u) {
            c[u] = walk_class(a.grid, w);
            walk_step(w);
        }
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL; ++u) {
            count += (int)(c[u] & open);
            open &= ~(c[u] >> 1);
        }
        if (!open) return count;
        visits -= FS_RAY_UNROLL;
    }
    if (visits > 0) {
        // the last, partial group as ONE batch of independent loads (a loop of single visits would expose a load latency
        // per step); slots past the ray's end hold class 0, which neither counts nor stops
        uint32_t c[FS_RAY_UNROLL - 1];
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL - 1; ++u) {
            c[u] = 0u;
            if ((uint32_t)u < visits) {
                c[u] = walk_class(a.grid, w);
                walk_step(w);
            }
        }
#pragma unroll
        for (int u = 0; u < FS_RAY_UNROLL - 1; ++u) {
            count += (int)(c[u] & open);
            open &= ~(c[u] >> 1);
        }
    }
    return count;
}

__device__ __forceinline__ int ray_march(const FsRayArgs &a, WalkClass w) { return ray_march_classes(a, w); }
__device__ __forceinline__ int ray_march(const FsRayArgs &a, WalkSparse w) { return ray_march_classes(a, w); }

// Every wave of the kernel below works on LDS of its own (per-yaw sums, tile, descriptors): what it wrote must be visible to
// its own later reads — which the LDS guarantees per wave, in program order — so the points where that matters need the
// compiler to keep the order, not a workgroup barrier.  (With s_barrier there the four fans of a workgroup waited for the
// slowest of them twice, and a finished wave kept its slot until the whole workgroup was done.)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#ifdef FS_RAY_WAVES_PER_EU
#define FS_RAY_OCCUPANCY __attribute__((amdgpu_waves_per_eu(FS_RAY_WAVES_PER_EU)))
#else
#define FS_RAY_OCCUPANCY
#endif
template <typename Walk>
__global__ __launch_bounds__(FS_RAY_WAVES * 64) FS_RAY_OCCUPANCY
void fs_raymarch_kernel(const FsRayArgs a)
{
    extern __shared__ int fs_ray_lds[];          // [FS_RAY_WAVES][n_yaw] per-yaw sums over the elevation rings
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    // Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2.  With the
    // candidate list in spatial (Morton) order, give every XCD a contiguous run of it so that neighbouring fans
    // share cache lines in ONE L2: position p of the sorted list is served by block (p % per) * 8 + p / per.
    const int per = (int)(gridDim.x >> 3);
    const int slot_block = a.perm ? (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int slot = slot_block * FS_RAY_WAVES + wave;
    const bool active = slot < a.n;
    const int c = active ? (a.perm ? a.perm[slot] : slot) : 0;
    int *ysum = fs_ray_lds + wave * a.n_yaw;
    const int n_rays = a.n_yaw * a.n_elev;

    for (int i = lane; i < a.n_yaw; i += 64) ysum[i] = 0;
    wave_lds_fence();

    bool black = false;
    double sx = 0, sy = 0, sz = 0;
    if (active) {
        black = a.blacklisted && a.blacklisted[c];
        sx = a.goal[3 * c]; sy = a.goal[3 * c + 1]; sz = a.goal[3 * c + 2];
    }
    bool fail = false;
    uint32_t sxm = 0, sym = 0, szm = 0;
    const bool start_ok = active && world_to_map(a.grid, sx, sy, sz, sxm, sym, szm);
    if (active && !black) {
        for (int r = lane; r < n_rays; r += 64) {
            // ray r = e * n_yaw + i: one elevation ring per wave instruction.  (Dealing the class walk's rays ring-fastest — a
            // sector of 64 / n_elev yaws per instruction, whose rays tend to end at the same wall — measured +-1 %: r03.)
            const int i = r % a.n_yaw;
            // DEP/src/CostCalculator.cpp:42-43: wx = sx + (MAX_CAMERA_DEPTH * cos(theta))
            double wx = sx + a.dir[3 * r];
            double wy = sy + a.dir[3 * r + 1];
            double wz = sz + a.dir[3 * r + 2];
            if (a.clamp) {                                  // :47-48
                wx = std_max(a.lo_x, std_min(a.hi_x, wx));
                wy = std_max(a.lo_y, std_min(a.hi_y, wy));
                wz = std_max(a.lo_z, std_min(a.hi_z, wz));
            }
            Walk w;
            int count = 0;
            if (ray_setup(a, start_ok, sxm, sym, szm, wx, wy, wz, w)) count = ray_march(a, w);
            else fail = true;
            if (count) atomicAdd(&ysum[i], count);
            if (a.ray_counts) a.ray_counts[(size_t)c * n_rays + r] = count;
        }
    }
    const bool any_fail = __any(fail);           // :50-55 — the first failing ray aborts the candidate
    wave_lds_fence();
    if (!active) return;

    uint8_t ach = a.achievable_in ? a.achievable_in[c] : (uint8_t)1;
    if (black || any_fail) {
        if (a.ray_counts)
            for (int r = lane; r < n_rays; r += 64) a.ray_counts[(size_t)c * n_rays + r] = 0;
        if (lane == 0) {
            a.arrival[c] = 0; a.argmax[c] = 0; a.yaw[c] = 0.0; a.achievable[c] = ach;
            a.status[c] = black ? FS_STATUS_BLACKLISTED : FS_STATUS_OFF_MAP;
            if (a.records) {
                fs_record r{};
                r.flags = (ach ? FS_FLAG_ACHIEVABLE : 0u) | ((uint32_t)(black ? FS_STATUS_BLACKLISTED : FS_STATUS_OFF_MAP) << 8);
                a.records[c] = r;
            }
        }
        return;
    }

    // footprint disc, DEP/src/Helpers.cpp:135-155 in the candidate's z slice; off-grid cells are not lethal
    const int ri = (int)a.footprint_radius;
    const int side = 2 * ri + 1;
    bool lethal = false;
    auto scan_disc = [&](auto cell_at) {
        const float inv_side = 1.0f / (float)side;
        for (int t = lane; t < side * side; t += 64) {
            // t / side and t % side without the integer division: the float quotient is within one of the true row for
            // t < 2^22 (side <= 2048; larger discs divide), and one step on the remainder makes it exact
            int row = side <= 2048 ? (int)(((float)t + 0.5f) * inv_side) : t / side;
            int col = t - row * side;
            if (col < 0) { --row; col += side; } else if (col >= side) { ++row; col -= side; }
            const int dx = row - ri, dy = col - ri;
            if ((double)(dx * dx + dy * dy) <= a.footprint_radius * a.footprint_radius) {
                const uint32_t x = sxm + (uint32_t)dx, y = sym + (uint32_t)dy;
                if (x < (uint32_t)a.grid.nx && y < (uint32_t)a.grid.ny && cell_at(x, y) == 254) lethal = true;
            }
        }
    };
    const uint8_t *slice = a.grid.cells + (size_t)szm * (size_t)a.grid.ny * (size_t)a.grid.nx;       // (needs the cost itself: 254)
    scan_disc([&](uint32_t x, uint32_t y) -> int { return slice[(size_t)y * (size_t)a.grid.nx + x]; });
    lethal = __any(lethal);
    const int fsize = a.frontier_size ? a.frontier_size[c] : 0;
    if (lethal && (double)fsize < 10.0) ach = 0;           // CostCalculator.cpp:77-82

    // FOV window, no wrap-around, first maximum (CostCalculator.cpp:87-107)
    const int k = a.window;
    int best = -1, best_i = 0x7fffffff;
    for (int i = lane; i <= a.n_yaw - k; i += 64) {
        int s = 0;
        for (int j = 0; j < k; ++j) s += ysum[i + j];
        if (s > best) { best = s; best_i = i; }            // ascending i per lane: strict > keeps the first
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int ob = __shfl_xor(best, d), oi = __shfl_xor(best_i, d);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    if (lane == 0) {
        a.arrival[c] = best;                                // :112
        if ((double)best < a.min_gt) ach = 0;              // :114-118
        a.argmax[c] = best_i;
        a.yaw[c] = ((double)best_i * a.delta_theta) + a.half_fov;   // :119, half_fov = CAMERA_FOV / 2
        a.achievable[c] = ach;
        a.status[c] = FS_STATUS_OK;
        if (a.records) {
            fs_record r{};
            r.arrival = best; r.argmax = best_i;
            r.yaw = (float)(((double)best_i * a.delta_theta) + a.half_fov);
            r.flags = (ach ? FS_FLAG_ACHIEVABLE : 0u) | ((uint32_t)FS_STATUS_OK << 8);
            a.records[c] = r;
        }
    }
    // pose (goal, best yaw) of the Fisher-information stage as one 48-byte record: rotation of
    // orientationAroundZAxis(yaw) (table indexed by the argmax) and the float32 translation of getTransformFromPose
    if (a.pose12 && lane < 12) {
        const float v = lane < 9 ? a.yawR[9 * best_i + lane] : (float)a.goal[3 * (size_t)c + (lane - 9)];
        a.pose12[12 * (size_t)c + lane] = v;
    }
}

// One lane per segment: getTracedCells(sx, sy, wx, wy, visitor, max_length, costmap) for arbitrary end points
// (DEP/src/Helpers.cpp:32-96) with every RayTracedCells accessor (Helpers.hpp:83-101).  Unlike the arrival
// fan the walk never stops early: unknown_cells_ and all_cells_count_ keep counting behind an obstacle.
template <typename Walk>
__global__ void fs_segments_kernel(const FsSegArgs s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    FsRayArgs a{};
    a.grid = s.grid;
    const FsGridDev &g = s.grid;
    uint32_t x1, y1, z1, x0, y0, z0;
    const double sx = s.start[3 * i], sy = s.start[3 * i + 1], sz = s.start[3 * i + 2];
    const double wx = s.end[3 * i], wy = s.end[3 * i + 1], wz = s.end[3 * i + 2];
    int traced = 0, unknown = 0, all = 0;
    bool hit = false, ok = false;
    if (world_to_map(g, wx, wy, wz, x1, y1, z1) && world_to_map(g, sx, sy, sz, x0, y0, z0)) {
        ok = true;
        Walk w;
        walk_init(w, g, x0, y0, z0, x1, y1, z1, s.max_length);
        for (uint32_t v = 0; v <= w.end; ++v) {
            const int c = walk_cell(g, w);
            {
                ++all;                                                        // Helpers.hpp:62
                if (c <= s.trace_max && c >= s.trace_min && !hit) ++traced;   // :64-67
                if (c >= s.obst_min && c <= s.obst_max) hit = true;           // :68-71
                if (c == 255) ++unknown;                                      // :72-75
            }
            walk_step(w);
        }
    }
    s.ok[i] = ok; s.hit[i] = hit;
    s.traced[i] = traced; s.unknown[i] = unknown; s.all[i] = all;
}

__global__ void fs_selftest_kernel(int32_t max_abs, double *out_sqrt, double *out_div)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int side = max_abs + 1;
    if (t >= side * side) return;
    const int dx = t / side, dy = t % side;
    const long long d2 = (long long)dx * dx + (long long)dy * dy;
    const double dist = sqrt((double)d2);
    out_sqrt[t] = dist;
    out_div[t] = (dist == 0.0) ? 1.0 : 40.0 / dist;
}

}  // namespace

hipError_t fs_launch_raymarch(const FsRayArgs &a, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    int blocks = (a.n + FS_RAY_WAVES - 1) / FS_RAY_WAVES;
    if (a.perm) blocks = (blocks + 7) / 8 * 8;               // whole XCD rounds for the remap above
    const size_t lds = sizeof(int) * (size_t)FS_RAY_WAVES * (size_t)a.n_yaw;
    if (a.layout == 2) hipLaunchKernelGGL(fs_raymarch_kernel<WalkSparse>, dim3(blocks), dim3(FS_RAY_WAVES * 64), lds, s, a);
    else if (a.layout == 1) hipLaunchKernelGGL(fs_raymarch_kernel<WalkClass>, dim3(blocks), dim3(FS_RAY_WAVES * 64), lds, s, a);
    else hipLaunchKernelGGL(fs_raymarch_kernel<WalkLinear>, dim3(blocks), dim3(FS_RAY_WAVES * 64), lds, s, a);
    return hipGetLastError();
}

hipError_t fs_launch_segments(const FsSegArgs &a, hipStream_t s)
{
    if (a.n <= 0) return hipSuccess;
    // (the segment visitor reports unknown cells separately from its ranges: it needs the costs, hence the byte image)
    hipLaunchKernelGGL(fs_segments_kernel<WalkLinear>, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t fs_launch_selftest(int32_t max_abs, double *d_sqrt, double *d_div, hipStream_t s)
{
    const int side = max_abs + 1;
    const int total = side * side;
    hipLaunchKernelGGL(fs_selftest_kernel, dim3((total + 255) / 256), dim3(256), 0, s, max_abs, d_sqrt, d_div);
    return hipGetLastError();
}
