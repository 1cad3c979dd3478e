// fs_fim.hip — landmark Fisher-information accumulation for gfx950 (CDNA4).
//
// Replaces FisherInformationManager::isPoseSafe's landmark loop and getInformationFromLookup
// (FIP/src/fisher_information/FisherInfoManager.cpp:83-100,287-324) plus the per-landmark Jacobian/FIM
// of FisherInformationHelpers.cpp:71-123, batched over candidate poses.
//
// Data: the landmark cloud is staged once (fs_upload_landmarks) in k-d leaf order as SoA fp32 and cut
// into chunks of 64 consecutive landmarks, each with a bounding sphere.  Persistent workgroups (two per CU) pull
// candidates from a device-side cursor over the processing order (fs_sort.hip); per candidate:
//   1. cull   — chunk k belongs to wave (k mod waves); every lane tests one chunk sphere against the
//               visibility volume (range sphere + cone, conservative); the ballots are the wave's accepted
//               chunks.  Only a few per cent of a 512^3 map's cloud survives, so the per-landmark work drops by an order of magnitude while the
//               visible set stays exactly the brute-force one (the exact predicate is re-evaluated
//               per landmark).
//   2. test   — the wave's ballot masks become a flat list of accepted chunk ids in one register (lane j = the j-th
//               accepted chunk) that is walked with v_readlane; per chunk coalesced 256-B loads (the next chunk's in
//               flight), p = R^T (w - t) in fp32 with the specified fma order, range + cone predicate as the sign
//               of one v_min3_f32 of three exact differences.
//   3. compact— visible lanes push p into the wave's LDS queue (ballot + mbcnt); whenever 64 are queued
//               the whole wave runs the expensive part at full lane utilisation.
//   4. score  — voxel lattice index (fp32 fast path, exact fp64 re-evaluation only next to a rounding
//               boundary: identical to getVoxelCoordinate, FisherInfoManager.hpp:108-123), dense-table
//               value, and the reference's per-voxel pointCount bookkeeping (:296-304) as an LDS hash
//               table keyed by the lattice index: the returning LDS atomic that bumps the voxel's
//               count is the landmark's rank k in its voxel -> crowding factor exp(1 - k^0.8)
//               (FisherInfoManager.hpp:102-106), looked up by the next call so that no call waits for its own
//               atomics.  Same multiset of (info_v, k) terms as the reference's sequential loop.
//               Unit-weight 6x6 FIM via the block form [[P/n^2,-S/n^2],[S/n^2,P]].
//   5. reduce — DPP row reductions, LDS across waves, fp64 per quantity; fs_fim_finish_kernel assembles F,
//               trace and the log det (pivots of the square-root-free Cholesky factorisation).
// Hash tables: 2^14 LDS slots per 512-thread workgroup (two workgroups per CU).  A candidate predicted to hold more
// distinct voxels than the table takes is scored by the same workgroup in 2, 4 or 8 voxel-partitioned PASSES; only a
// real overflow hands it to the table in HBM.  Scan + scatter-count has no dense contraction: no MFMA.
#include "fs_internal.h"

#include <atomic>

#define FS_NACC 18       // info, 15 FIM block sums, n_visible, n_voxels
#define FS_QCAP 128      // per-wave compaction queue (<= 63 left over + 64 new)
#define FS_MAX_PROBE 64  // LDS tiers give up (-> next tier) after this many bucket visits
#define FS_MAX_PARTS 8   // scoring passes per candidate within one LDS tier

#ifdef FS_FIM_STAMPS   // development: per-phase cycle counters of the tier-1 worker (tools/fim_stamps.py)
#define FS_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); st[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define FS_STAMP(k) do { } while (0)
#endif

#ifndef FS_T1_THREADS
#define FS_T1_THREADS 512
#endif
#ifndef FS_T1_WAVES_PER_EU
#define FS_T1_WAVES_PER_EU ((FS_T1_THREADS * 2 + 255) / 256)      // two workgroups per CU, four SIMDs
#endif
#define FS_HBM_THREADS 1024    // workgroup of the HBM tier (fs_fim_tier3_kernel)

#ifdef FS_FIM_BOUNDS   // development: every global access is range-checked, violations are recorded in counters[30] and skipped
#define FS_BOUND(ok, code) ((ok) ? true : (atomicMax(&a.counters[30], (unsigned long long)(code)), false))
#else
#define FS_BOUND(ok, code) true
#endif

namespace {

typedef float fs_f2 __attribute__((ext_vector_type(2)));
struct Acc {
    float info;
    // (sum P, sum P / n^2) per entry of P = I - p^ p^T (xx, xy, xz, yy, yz, zz): a pair per entry, so that one packed
    // v_pk_fma_f32 — (A, B) += v * (1, 1/n^2) — accumulates both (gfx950 issues it like a scalar fp32 instruction)
    fs_f2 AB[6];
    fs_f2 s01; float s2;          // sum p / n^2
    int nvis, nvox;
    // The last score call's (table value, count before my add) per lane, both possibly still in flight: the crowding factor is
    // looked up and the product added by the NEXT call (or by resolve_pending at the end of the pass), so no call waits for
    // its own returning add, factor read and table gather.  pend_old = 0xffffffff: nothing pending (factor index clamps to 0.0f).
    uint32_t pend_old; float pend_info;
};

// (rank = count + 1; ~0 -> the factor table's last entry, which is 0.0f.  With a table of finite values only — TABLE_FULL — a lane
// without a term may simply add value * 0.0f; a NaN hole must not reach the product.)
template <bool TABLE_FULL>
__device__ __forceinline__ void resolve_pending(Acc &acc, const float *lfac)
{
    const uint32_t cnt = acc.pend_old & FS_SLOT_CNT_MASK;
    const float fac = (lfac + 1)[cnt < FS_FACTOR_N - 2 ? cnt : FS_FACTOR_N - 2];
    if (TABLE_FULL) acc.info += acc.pend_info * fac;
    else acc.info += fac > 0.0f ? acc.pend_info * fac : 0.0f;
}

// base[idx] with a 32-bit BYTE offset: lets the load use the scalar-base + 32-bit vector-offset addressing form instead
// of building a 64-bit address per lane (every array indexed this way is far below 4 GiB)
__device__ __forceinline__ float ldg32(const float *__restrict__ base, uint32_t idx)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (idx << 2));
}

// bucket of a voxel key (< 2^21): multiplicative hashing with a 24-bit multiplier — v_mul_u32_u24 runs at full rate, a
// full 32-bit multiply at a quarter of it.  The multiplier was picked among a dozen odd constants on the voxel sets of
// the heaviest C3 candidates (longest probe chain at 89 % load: 12 bucket visits; 2654435761 as a 32-bit multiplier: 17).
__device__ __forceinline__ uint32_t hash_key(uint32_t key, int bits)
{
    return (uint32_t)__umul24(key, 0x7FEB35u) >> (32 - bits);     // the cast matters: HIP declares __umul24 as returning int
}

// Voxel lattice index of a camera-frame point -> key into the dense table: round(x * (1 / corrected_step)) of
// getVoxelCoordinate (FisherInfoManager.hpp:119-121), exactly.  The fp32 product is within |r| * 1.3e-7 of the fp64
// one (|r| < 2^10: every scored landmark lies within max_dist of the camera, and the host sets FsFimArgs::far_lattice when
// max_dist / step could reach 2^10 — then every lane takes the fp64 path), so away from a .5 boundary the fp32 nearest integer
// equals the fp64 half-away-from-zero result; lanes within FsFimArgs::key_thr' = 0.5 - key_thr of a boundary — twice the
// error bound at the largest |r| the visibility range allows, 1.2e-5 at 14 m — re-evaluate the fp64 expression.
// in_table: the voxel lies inside the table box.
__device__ __forceinline__ uint32_t voxel_key(const FsFimArgs &a, bool active, float px, float py, float pz, bool &in_table)
{
    const float rx = px * a.inv_step_f, ry = py * a.inv_step_f, rz = pz * a.inv_step_f;
    const float nx = rintf(rx), ny = rintf(ry), nz = rintf(rz);
    int jx = (int)nx, jy = (int)ny, jz = (int)nz;
    const float worst = fmaxf(fmaxf(fabsf(rx - nx), fabsf(ry - ny)), fabsf(rz - nz));
    if (active && (a.far_lattice || !(worst < a.key_thr))) {
        jx = (int)round((double)px * a.inv_step);
        jy = (int)round((double)py * a.inv_step);
        jz = (int)round((double)pz * a.inv_step);
    }
    jx -= a.jx0; jy -= a.jy0; jz -= a.jz0;
    in_table = active && (unsigned)jx < (unsigned)a.tx && (unsigned)jy < (unsigned)a.ty && (unsigned)jz < (unsigned)a.tz;
    // table extents are < 2^11 each and the cell count < 2^21 (FS_MAX_TABLE_CELLS): 24-bit multiplies are exact
    // (two v_mad_u32_u24 — left to itself the compiler widens the first product to a quarter-rate v_mad_u64_u32 under an exec mask)
    uint32_t row, cell;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(jx), "s"(a.ty), "v"(jy));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cell) : "v"(row), "s"(a.tz), "v"(jz));
    return in_table ? cell : 0u;
}

// Which of a candidate's n_parts (a power of two) scoring passes owns a landmark: the x index of its voxel, exactly as
// voxel_key computes it, modulo n_parts — whole voxels go to one pass, neighbouring slabs of voxels alternate between
// the passes (balanced), and only one coordinate has to be rounded.
// the x index of a landmark's voxel (absolute lattice index, before the table's offset), exactly as voxel_key computes it
__device__ __forceinline__ int voxel_jx(const FsFimArgs &a, bool active, float px)
{
    const float rx = px * a.inv_step_f;
    const float nx = rintf(rx);
    int jx = (int)nx;
    if (active && (a.far_lattice || !(fabsf(rx - nx) < a.key_thr))) jx = (int)round((double)px * a.inv_step);
    return jx;
}
__device__ __forceinline__ int voxel_part(const FsFimArgs &a, bool active, float px, int n_parts)
{
    return (voxel_jx(a, active, px) + 4096) & (n_parts - 1);
}

// Scores one queued landmark per lane (`active` false only for the empty lanes of the final partial batch): voxel key,
// dense-table gather, the FIM block sums, and the voxel's hash bucket — the reference's pointCount bookkeeping (:296-304).
// The table is organised in buckets of four slots read with ONE 16-byte LDS load: at the load factors used (< 0.5) a
// bucket almost never fills up, so nearly every landmark resolves with one read plus one or two atomics.
//
// Written with the scalar unit in mind.  Boolean logic on per-lane conditions compiles to mask arithmetic (s_and / s_or /
// s_andn2 on 64-bit lane masks), and a scalar instruction costs a wave 8 cycles alone and up to 16 with four waves per
// SIMD (tools/valu_mix_calib.hip) — more than any vector instruction here.  So the first bucket visit is straight-line
// code under ONE exec region (the lanes with a voxel), conditions implied by construction are not re-ANDed, the slot
// search and the saturation are integer minima, and the general probe loop with its loop-carried lane masks is entered
// only when a lane is left over.
//
// Nothing here waits for this call's own returning add, factor read or table gather: the pair (table value, count before my
// add) is parked in `acc` and turned into info_v * exp(1 - k^0.8) by the NEXT call (resolve_pending) — for tables without
// NaN holes (TABLE_FULL: every generated table) the gather is not waited for inside the call at all.
//
// INFO_ONLY: the caller asked for the reference's scalar alone (isPoseSafe reads nothing else, FisherInfoManager.cpp:83-100) — the
// 6x6 block sums and the visible count are not accumulated.
// WITH_SUMS false: likewise without the block sums — and COUNT_VIS false without the visible count — for the worker that adds
// both at TEST time (cone off: see fim_worker, SUMS_AT_TEST).
// The 6x6 block sums of one landmark per lane: (A, B) += P * (1, 1/n^2), s += p / n^2 with P = I - p^ p^T.  A lane with q = 0
// (not visible, or at the camera centre) adds exact zeros to every sum.
__device__ __forceinline__ void add_block_sums(Acc &acc, float px, float py, float pz, float q)
{
    const fs_f2 pxy = {px, py};
    const fs_f2 qxy = pxy * (fs_f2){q, q};
    const fs_f2 xxyy = pxy * qxy;
    const float qy = qxy.y, qz = pz * q;
    const float xx = xxyy.x, yy = xxyy.y, zz = pz * qz;
    const float v[6] = {yy + zz, -(px * qy), -(px * qz), xx + zz, -(py * qz), xx + yy};
    const fs_f2 one_q = {1.0f, q};
#pragma unroll
    for (int i = 0; i < 6; ++i) acc.AB[i] = __builtin_elementwise_fma((fs_f2){v[i], v[i]}, one_q, acc.AB[i]);
    acc.s01 += qxy; acc.s2 += qz;
}

template <bool TABLE_FULL, bool WITH_SUMS, bool COUNT_VIS = true>
__device__ __forceinline__ void score_visible(const FsFimArgs &a, uint32_t *table, const float *lfac, int bits, uint32_t max_probe,
                                              bool active, float px, float py, float pz, Acc &acc, bool &overflow)
{
    if (COUNT_VIS) acc.nvis += active ? 1 : 0;
    bool in_table;
    const uint32_t key = voxel_key(a, active, px, py, pz, in_table);
    const float info_t = FS_BOUND(key < (uint32_t)(a.tx * a.ty * a.tz), 1) ? ldg32(a.table, key) : 0.0f;
    bool valid = in_table;
    if (!TABLE_FULL) valid = valid && (info_t == info_t);
    const uint32_t tag = (key + 1u) << FS_SLOT_CNT_BITS;
    const uint32_t bmask = (1u << (bits - 2)) - 1u;
    uint32_t hb = hash_key(key, bits - 2);
    (void)FS_BOUND(hb <= bmask, 5);
    hb &= bmask;      // whatever the hash returns, the bucket index stays inside the table
    uint4 bk = *reinterpret_cast<const uint4 *>(table + (hb << 2));
    resolve_pending<TABLE_FULL>(acc, lfac);                           // the previous call's landmarks
    if (WITH_SUMS) {
        const float n2 = __fmaf_rn(px, px, __fmaf_rn(py, py, pz * pz));
        const float q = (active && n2 > 0.0f) ? __builtin_amdgcn_rcpf(n2) : 0.0f;
        add_block_sums(acc, px, py, pz, q);
        // (pins the sums HERE, in the shadow of the bucket read: left alone the compiler sinks them below the probe)
        asm volatile("" : "+v"(acc.AB[0]), "+v"(acc.AB[1]), "+v"(acc.AB[2]), "+v"(acc.AB[3]), "+v"(acc.AB[4]), "+v"(acc.AB[5]), "+v"(acc.s01), "+v"(acc.s2));
    }
    // ---- first bucket visit, straight-line
    uint32_t pend = 0xffffffffu;      // count before my add (rank - 1); ~0: no term (factor index clamps to an entry that is 0.0f)
    uint32_t n_won = 0u;
    uint32_t res = 0u;                // > FS_SLOT_CNT_MASK: this visit did not resolve the lane
    if (valid) {
        const uint32_t w0 = bk.x, w1 = bk.y, w2 = bk.z, w3 = bk.w;
        // a slot holds my voxel iff slot ^ tag is a bare count (an empty slot never is: tag >= 2^11).  The slot's index rides in
        // the two lowest bits of the difference — the match test looks at bits >= 11, saturation at bit 10, the exact count
        // comes back from the add — so the smallest of the four is the match AND says where it sits: no compare / select chain.
        const uint32_t y0 = (w0 ^ tag) & ~3u, y1 = ((w1 ^ tag) & ~3u) | 1u, y2 = ((w2 ^ tag) & ~3u) | 2u, y3 = ((w3 ^ tag) & ~3u) | 3u;
        const uint32_t y01 = y0 < y1 ? y0 : y1, y23 = y2 < y3 ? y2 : y3;                                // v_min_u32, v_min3_u32
        const uint32_t ymin = y01 < y23 ? y01 : y23;
        const bool has_match = ymin <= FS_SLOT_CNT_MASK;
        // slots fill in order: the first empty one is the number of occupied ones
        uint32_t u0, u1, u2, u3;
        asm("v_min_u32 %0, 1, %1" : "=v"(u0) : "v"(w0));
        asm("v_min_u32 %0, 1, %1" : "=v"(u1) : "v"(w1));
        asm("v_min_u32 %0, 1, %1" : "=v"(u2) : "v"(w2));
        asm("v_min_u32 %0, 1, %1" : "=v"(u3) : "v"(w3));
        const uint32_t used = (u0 + u1) + (u2 + u3);
        uint32_t *slot = table + (hb << 2) + (has_match ? (ymin & 3u) : used);
        uint32_t prev = 0xffffffffu;
        if (!has_match && used < 4u) prev = atomicCAS(slot, 0u, tag | 1u);    // 0: this lane created the entry
        // prev: 0 = created; another slot word (>= 2^11) = somebody was faster; ~0 = no CAS.  prev ^ tag is a bare count iff the
        // faster one brought MY voxel ("late": add to it).  None of this needs the masks of the conditions above re-ANDed:
        // ~0 is not 0, and ~0 ^ tag is no bare count (keys stop two short of 2^21).
        const uint32_t xprev = prev ^ tag;
        const uint32_t cnt = ymin < xprev ? ymin : xprev;                 // the count my voxel showed, wherever I found it
        // rank - 1 for lanes that do not add: created -> 0 (prev), saturated -> FS_SLOT_CNT_SAT (the factor is 0.0f from rank 337 on
        // and the count field must not run over); the add's return value replaces it for the others
        uint32_t pnd = cnt < prev ? cnt : prev;
        pnd = pnd < FS_SLOT_CNT_SAT ? pnd : FS_SLOT_CNT_SAT;
        if (cnt < FS_SLOT_CNT_SAT) pnd = atomicAdd(slot, 1u);
        pend = pnd;
        uint32_t w1_;                                                     // 1 iff prev == 0
        asm("v_min_u32 %0, 1, %1" : "=v"(w1_) : "v"(prev));
        n_won = w1_ ^ 1u;
        res = cnt < prev ? cnt : prev;                                    // resolved iff matched, late or created
    }
    asm volatile("" : "+v"(res));     // (the compare stays out here, one instruction whose mask is the ballot)
    const unsigned long long left_mask = __builtin_amdgcn_ballot_w64(res > FS_SLOT_CNT_MASK);
    if (left_mask != 0ull) {
        // ---- leftovers (a CAS lost to ANOTHER voxel, or a bucket with four other voxels), the general loop: it reads the
        // bucket again and moves on to the next one when it is full
        const bool left = __builtin_amdgcn_inverse_ballot_w64(left_mask);
        bool done = !left;
        pend = left ? 0xffffffffu : pend;
        for (uint32_t probe = 1; probe < max_probe; ++probe) {
            if (__all(done)) break;
            if (!done) bk = *reinterpret_cast<const uint4 *>(table + (hb << 2));
            const uint32_t w0 = bk.x, w1 = bk.y, w2 = bk.z, w3 = bk.w;
            const uint32_t x0 = w0 ^ tag, x1 = w1 ^ tag, x2 = w2 ^ tag, x3 = w3 ^ tag;
            const uint32_t xm3 = x0 < x1 ? (x0 < x2 ? x0 : x2) : (x1 < x2 ? x1 : x2);
            const uint32_t xmin = xm3 < x3 ? xm3 : x3;
            const bool has_match = xmin <= FS_SLOT_CNT_MASK;
            const uint32_t jm = x0 == xmin ? 0u : (x1 == xmin ? 1u : (x2 == xmin ? 2u : 3u));
            const uint32_t used = (w0 != 0u) + (w1 != 0u) + (w2 != 0u) + (w3 != 0u);
            const bool can_insert = !has_match && used < 4u;
            uint32_t *slot = table + (hb << 2) + (has_match ? jm : (used < 4u ? used : 0u));
            uint32_t prev = 0xffffffffu;
            if (!done && can_insert) prev = atomicCAS(slot, 0u, tag | 1u);
            const bool won = !done && can_insert && prev == 0u;
            const uint32_t xprev = prev ^ tag;
            const bool late = !done && can_insert && prev != 0u && xprev <= FS_SLOT_CNT_MASK;
            const bool do_add = !done && (has_match || late);
            const uint32_t cnt = has_match ? xmin : xprev;
            const bool sat = do_add && cnt >= FS_SLOT_CNT_SAT;
            pend = won ? 0u : pend;
            pend = sat ? FS_SLOT_CNT_SAT : pend;
            if (do_add && !sat) pend = atomicAdd(slot, 1u);
            n_won += won ? 1u : 0u;
            const bool is_full = !done && !has_match && used >= 4u;
            hb = is_full ? ((hb + 1u) & bmask) : hb;
            done = done || won || do_add;
        }
        if (!done) { overflow = true; pend = 0xffffffffu; }
    }
    acc.nvox += (int)n_won;                                            // occupied_voxel_count_++ (:304)
    acc.pend_old = pend;
    acc.pend_info = info_t;                                           // (raw: a lane without a term has pend = ~0 -> factor 0.0f)
}

// wave64 sums with DPP row operations (no LDS traffic): the totals land in lane 63.  N values at once, step by step: the two
// cross-row steps are written as ONE v_add_f32_dpp each with a partial row mask (rows outside the mask keep their value) —
// from the builtin the compiler makes a move of the identity, a DPP move and an add — and an instruction that reads a register
// through DPP must be two instructions behind the one that wrote it, which the step-major order provides (the assembler does
// not look into inline asm: hence the one s_nop in front of the first).
template <int N>
__device__ __forceinline__ void wave_sums_dpp(float (&v)[N])
{
    static_assert(N >= 3, "the step-major order is what separates a DPP read from the write before it");
#define FS_DPP_ADD(x, ctrl)                                                                                   \
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
#pragma unroll
    for (int i = 0; i < N; ++i) FS_DPP_ADD(v[i], 0x111);   // row_shr:1
#pragma unroll
    for (int i = 0; i < N; ++i) FS_DPP_ADD(v[i], 0x112);   // row_shr:2
#pragma unroll
    for (int i = 0; i < N; ++i) FS_DPP_ADD(v[i], 0x114);   // row_shr:4
#pragma unroll
    for (int i = 0; i < N; ++i) FS_DPP_ADD(v[i], 0x118);   // row_shr:8   -> lane 15 of each row holds the row sum
#undef FS_DPP_ADD
    asm volatile("s_nop 1");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[i]));   // into rows 1 and 3
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v[i]));   // into rows 2 and 3 -> lane 63
}

// dynamic-LDS carve-up of one workgroup: [per-wave compaction queues][factor copy][chunk masks][hash table]
template <int THREADS>
struct Lds {
    static constexpr int WAVES = THREADS / 64;
    static constexpr size_t fixed_words = (size_t)WAVES * 3 * FS_QCAP + FS_FACTOR_N;
    // One mask region per wave.  A wave culls the next candidate only after it has finished walking its own masks of
    // the current one, and no wave ever reads another wave's masks: one buffer is enough for the pipelined loop.
    // (`sets` = 2 for the worker that also keeps, per chunk, whether it can meet the lookup table's box: SUMS_AT_TEST)
    static __host__ __device__ size_t mask_words(int n_groups, int sets) { const size_t w = 2 * (size_t)WAVES * n_groups * sets; return w + (w & 2); }
    static __device__ __forceinline__ unsigned long long *masks(uint32_t *base, int n_groups)
    {
        return reinterpret_cast<unsigned long long *>(base + fixed_words);
    }
    static __device__ __forceinline__ float *factor(uint32_t *base) { return reinterpret_cast<float *>(base) + WAVES * 3 * FS_QCAP; }
    static __device__ __forceinline__ float *queue(uint32_t *base, int wave) { return reinterpret_cast<float *>(base) + wave * (3 * FS_QCAP); }
    static __device__ __forceinline__ uint32_t *table(uint32_t *base, int n_groups, int sets) { return base + fixed_words + mask_words(n_groups, sets); }
};

// The work of one persistent workgroup: items [0, count) of a list, handed out by a device-side counter.
struct FimWork {
    const int32_t *list;          // item -> candidate (nullptr: lo + item)
    int32_t lo, count;
    unsigned long long *counter;
};

// Scores candidates until the work list is empty.  The loop is software-pipelined ACROSS candidates: a wave that has
// finished its share of candidate i culls the chunks of candidate i+1 before it joins the reduction barrier, so the
// latency of the sphere loads and the imbalance between waves overlap instead of adding up.  Two barriers per
// candidate: after the table clear and after the reduction.
// A candidate the LDS worker cannot finish (GLOBAL_TABLE false) is appended to the work list of the HBM tier; the HBM
// worker itself (GLOBAL_TABLE true) has nobody to hand over to and only counts such a candidate (never observed).
// CONE: how the visibility cone enters the test loop — a template parameter, so that the loop carries ONE predicate instead of a
// scalar dispatch over the modes per chunk.  FS_CONE_OFF: no cone (FsFimArgs::cone_mode == 0 — what the reference itself
// requests: max_angle 4.0 > pi, FisherInfoManager.cpp:63-64), the predicate is the sign of maxd2 - n2.  FS_CONE_NARROW: the common
// cone (half-angle < pi/2, cone_mode == 1), one v_min3.  FS_CONE_ANY: whatever cone_mode says (wide cones; the HBM tier).
// INFO_ONLY: the call wants info_ref / n_voxels only (score_visible) — and then a landmark outside the lookup table's box
// contributes nothing at all (a miss is skipped, FisherInfoManager.cpp:90-94), so chunks are culled EXACTLY against that box
// in the camera frame and landmarks behind its near face are dropped by the test predicate: with the cone off this gives back
// most of what the cone cull gave.  n_visible is not produced in this mode.
// YAW_ONLY: every pose is a rotation about Z (the fused path builds them so, fs_raymarch.hip `yawR`): R2 = R5 = R6 = R7 = 0 and
// R8 = 1 exactly, hence fma(R3, dy, R6 * dz) is R3 * dy and pz is dz bit for bit (up to the sign of a zero, which no
// consumer sees) — the transform takes 7 instructions instead of 12.  The host checks the zeros before selecting it.
enum { FS_CONE_OFF = 0, FS_CONE_NARROW = 1, FS_CONE_ANY = 2 };
// SPLIT: ONE pose over W = 2^split_shift workgroups (the reference's real call is one pose per tick, FisherInfoBTPlugin.cpp:24-57 — one
// of 512 workgroup slots).  A work item is then (pose, w): item id = pose * W + w.  Workgroup w hashes only the landmarks whose
// voxel lies in ITS slab of the lattice along the camera's x axis (slab_of) — whole voxels stay in one workgroup (a voxel's ranks
// must be counted in one table), every landmark is scored exactly once, each workgroup needs 1/W of the table — and culls the chunk
// list with that slab, so it tests only what can reach it.  Partial sums go to slot pose * W + w; the finish kernel adds them.
// The 6x6 block sums and the visible count follow the landmark: where they are taken at scoring time they are already partitioned
// with the voxels; where they are taken at TEST time (cone off) the item that owns a landmark's CHUNK (chunk id mod W) takes them and
// the other items skip that code — every visible landmark is in exactly one chunk.
template <int THREADS, bool GLOBAL_TABLE, bool TABLE_FULL, int CONE, bool INFO_ONLY, bool YAW_ONLY, bool SPLIT = false>
__device__ __forceinline__ void fim_worker(const FsFimArgs &a, const FimWork work, uint32_t *lds, uint32_t *table, const int tier_bits)
{
    static_assert(!SPLIT || (!GLOBAL_TABLE && TABLE_FULL && CONE != FS_CONE_ANY), "the split workers exist for the LDS tier, finite tables and the two common cone modes");
    const int split_shift = SPLIT ? a.split_shift : 0;
    const int split_w_mask = (1 << split_shift) - 1;
    // item w of a split pose owns the voxels whose x index lies in [jlo, jhi): W contiguous slabs of the lattice along the camera's
    // x axis (FsFimArgs::split_bound, cut by the host into equal shares of the visibility volume inside the table), the
    // first and the last one open-ended — every landmark, inside the table or not, belongs to exactly one item.  Contiguous, so an
    // item CULLS with its own slab and tests only the chunks that can reach it: the test work is divided too, not only the hashing.
    // (The workers that take the 6x6 sums at test time — cone off, full columns — keep INTERLEAVED slabs, x index mod W: they have
    // to test every chunk in range for the sums anyway, and the even split of the hashing measured better there: 50 frontiers at the
    // reference's request 105 against 120 us.)
    auto slab_of = [&](int w, int &jlo, int &jhi) {
        // (constant indices: a dynamically indexed kernel argument would be copied to scratch; a chain of scalar selects, once per item)
        jlo = a.split_bound[0]; jhi = a.split_bound[1];
#pragma unroll
        for (int k = 1; k < 32; ++k) {
            jlo = (k == w) ? a.split_bound[k] : jlo;
            jhi = (k == w) ? a.split_bound[k + 1] : jhi;
        }
    };
    constexpr uint32_t fail_code = GLOBAL_TABLE ? 3u : 2u;
    // With the cone off, roughly half of what is visible (everything within range) lies outside the lookup table's box — behind the
    // camera, mostly — and only feeds the 6x6 sums and the visible count.  SUMS_AT_TEST adds those two for EVERY visible landmark
    // right at the test (27 instructions per chunk tested, at whatever lane utilisation the chunk has) and queues only what can
    // hit the table for the hash-table half of the scoring: the landmarks outside the box no longer pay a score call (165
    // instructions per 64).  Pays when most tested landmarks are visible and many of them miss the table — the cone-off volume;
    // with the cone nearly everything visible is in the box and the sums stay in the score call, at full lane utilisation.
    constexpr bool SUMS_AT_TEST = (CONE == FS_CONE_OFF) && !INFO_ONLY && !GLOBAL_TABLE;
    constexpr int WAVES = THREADS / 64;
    constexpr int STRIDE = WAVES * 64;
    const uint32_t max_probe = GLOBAL_TABLE ? (1u << tier_bits) : FS_MAX_PROBE;   // bucket visits incl. re-reads
    const int tid = threadIdx.x;
    // the wave index is uniform by construction; telling the compiler so (readfirstlane) keeps chunk ids, queue counts,
    // loop conditions and address bases in scalar registers instead of exec-masked vector code
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int sh_overflow[2], sh_wave_tested[2][WAVES], sh_wave_box[2][WAVES], sh_next[2];
    __shared__ __attribute__((aligned(16))) float sh_red[WAVES * FS_NACC];

    const float4 *__restrict__ spheres = reinterpret_cast<const float4 *>(a.spheres);
    const float *__restrict__ LX = a.lx, *__restrict__ LY = a.ly, *__restrict__ LZ = a.lz;
    float *lfac = Lds<THREADS>::factor(lds);
    float *qx = Lds<THREADS>::queue(lds, wave), *qy = qx + FS_QCAP, *qz = qy + FS_QCAP;

    // The LDS tier hands every workgroup its first two items by position (b and b + grid): two dependent device-scope
    // atomics less before the first landmark is tested; the shared cursor then counts from 2 * grid.
    const int preassigned = GLOBAL_TABLE ? 0 : 2 * (int)gridDim.x;
    auto item = [&](int i) -> int {
        if (i >= work.count) return -1;
        const int cid = work.list ? work.list[work.lo + i] : work.lo + i;
        return FS_BOUND(cid >= 0 && cid < (a.n << split_shift), 2) ? cid : -1;     // (SPLIT: an item id, pose * W + w)
    };
    auto fetch = [&]() -> int {                                    // thread 0 only
        return item((int)atomicAdd(work.counter, 1ull) + preassigned);
    };

    // ---- 1. cull: chunk k belongs to wave (k mod WAVES); one sphere per lane, 64 chunks of the wave per pass, four
    // sphere loads in flight per lane.  Blacklisted / off-map candidates keep nothing: zero FI.
    // The pose record (R row-major, t) read here is handed to the scoring pass of the same candidate in registers.
    // The pose record and status of a candidate are loaded one iteration before its cull (PoseLoad, issued at the top of
    // the previous candidate's scoring loop), and the chunk spheres a lane tests are the same for every candidate —
    // chunk (g * STRIDE + lane * WAVES + wave) for pass g — so the first four passes' spheres stay in registers for the
    // whole persistent loop: on clouds of up to 4 * THREADS chunks (C3: 1563) a cull issues no load at all.
    // ONE register per lane holds the record while it is in flight: lane k < 12 element k of (R row-major, t), lane 12 the
    // status, lane 13 the sort key.  The cull broadcasts the elements with v_readlane (they are scalars there anyway); fourteen
    // registers per lane for the same record were the difference between 128 VGPRs and a spill in the scoring loop.
    struct PoseLoad { uint32_t v; };
    auto load_pose = [&](int c) -> PoseLoad {
        PoseLoad p;
        if (!FS_BOUND(c >= 0 && c < a.n, 6)) c = 0;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.Rt + 12 * (size_t)c) + lane;
        if (lane == 12) src = reinterpret_cast<const uint32_t *>(a.status) + (a.status ? c : 0);
        if (lane == 13) src = (!GLOBAL_TABLE && a.costmap) ? a.cand_key + c : nullptr;
        p.v = (lane < 14 && src) ? *src : 0u;                       // (status absent: 0 = FS_STATUS_OK)
        return p;
    };
    auto pose_word = [&](const PoseLoad &pl, int k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)pl.v, k); };
    auto pose_elem = [&](const PoseLoad &pl, int k) -> float { return __builtin_bit_cast(float, pose_word(pl, k)); };
    constexpr bool RESIDENT_SPHERES = !GLOBAL_TABLE;              // (the 1024-thread HBM tier has no registers to spare)
    // passes whose spheres stay in registers (three where the cull also keeps the box masks: the fourth sphere's registers are
    // what that worker needs to stay within 128 without spilling, and one sphere load per 60-us candidate is nothing)
    constexpr int N_RES = SUMS_AT_TEST ? 3 : 4;
    float4 sp_res[N_RES];
#pragma unroll
    for (int u = 0; u < N_RES; ++u) {
        const int j = u * STRIDE + lane * WAVES + wave;
        sp_res[u] = (RESIDENT_SPHERES && u < a.n_groups && j < a.n_chunks)
                        ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(spheres) + ((uint32_t)j << 4))
                        : make_float4(0.f, 0.f, 0.f, -1.0e30f);
    }
    auto cull = [&](const PoseLoad &pl, int buf, float *Rn, float *tn, const int item_w) {
        // (SPLIT: the faces of the item's slab in the camera frame, half a voxel beyond its outermost lattice planes plus a millimetre
        // — the margin of the table's box below; open ends at +-1e30)
        float slab_lo = -1.0e30f, slab_hi = 1.0e30f;
        if (SPLIT) {
            int jlo, jhi;
            slab_of(item_w, jlo, jhi);
            if (jlo > -(1 << 28)) slab_lo = (float)(((double)jlo - 0.5) / a.inv_step - 1.0e-3);
            if (jhi < (1 << 28)) slab_hi = (float)(((double)jhi - 0.5) / a.inv_step + 1.0e-3);
        }
        unsigned long long *masks = Lds<THREADS>::masks(lds, a.n_groups) + wave * a.n_groups;
        unsigned long long *bmasks = masks + WAVES * a.n_groups;    // SUMS_AT_TEST: the accepted chunks that can meet the table's box
        const bool dead = (int)pose_word(pl, 12) != FS_STATUS_OK;
#pragma unroll
        for (int i = 0; i < 9; ++i) Rn[i] = pose_elem(pl, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) tn[i] = pose_elem(pl, 9 + i);
        const float ax = Rn[0], ay = Rn[3], az = Rn[6];             // the camera's +x axis in the world frame (R[0], R[3], R[6])
        const float t0 = tn[0], t1 = tn[1], t2 = tn[2];
        int tested = 0, tested_box = 0;
        // One sphere against the visibility volume, conservatively (s.w carries a safety margin), as the sign of ONE value: every
        // condition is "a difference is >= 0" (exact in floating point), AND is a minimum, OR a maximum — straight-line code, no
        // exec-mask nesting (three levels of it and their scalar bookkeeping in the branchy version).  A lane without a chunk
        // holds a sphere of radius -1e30: its reach is negative.
        // (returned as the float whose sign decides, so that the ballot is the mask of ONE compare — a boolean merged from the
        // two branches is materialised with a select and a second compare, and ANDing a "candidate is alive" mask on top costs
        // three scalar instructions per pass: a dead candidate writes empty masks instead)
        // the sphere's centre in the camera frame against the lookup table's box grown by the radius (a superset of "the sphere
        // meets the box"; the box is half a voxel wider than the outermost lattice points plus a millimetre): >= 0 iff it can.
        // What INFO_ONLY culls with — and what SUMS_AT_TEST keeps as a second mask: its later scoring passes only hash, and only
        // what is in the box can be hashed.
        auto box_one = [&](const float4 s) -> float {
            const float dx = s.x - t0, dy = s.y - t1, dz = s.z - t2;
            const float cx = dx * Rn[0] + dy * Rn[3] + (YAW_ONLY ? 0.0f : dz * Rn[6]);
            const float cy = dx * Rn[1] + dy * Rn[4] + (YAW_ONLY ? 0.0f : dz * Rn[7]);
            const float cz = YAW_ONLY ? dz : dx * Rn[2] + dy * Rn[5] + dz * Rn[8];
            const float bx = fminf((cx + s.w) - a.box_lo[0], a.box_hi[0] - (cx - s.w));
            const float by = fminf((cy + s.w) - a.box_lo[1], a.box_hi[1] - (cy - s.w));
            const float bz = fminf((cz + s.w) - a.box_lo[2], a.box_hi[2] - (cz - s.w));
            return fminf(bx, fminf(by, bz));
        };
        // ... and the same for the item's slab (SPLIT): >= 0 iff the sphere can hold a landmark whose voxel the item owns
        auto slab_one = [&](const float4 s) -> float {
            const float dx = s.x - t0, dy = s.y - t1, dz = s.z - t2;
            const float cx = dx * Rn[0] + dy * Rn[3] + (YAW_ONLY ? 0.0f : dz * Rn[6]);
            return fminf((cx + s.w) - slab_lo, slab_hi - (cx - s.w));
        };
        auto cull_one = [&](int j, const float4 s) -> float {
            if (!a.cull) return j < a.n_chunks ? 0.0f : -1.0f;        // brute force (wave-uniform)
            const float dx = s.x - t0, dy = s.y - t1, dz = s.z - t2;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const float reach = a.max_dist_f + s.w, r2 = s.w * s.w;
            float c = fminf(reach, reach * reach - d2);               // within reach of the range sphere
            if (INFO_ONLY) c = fminf(c, box_one(s));
            // (SPLIT: what is scored here is what the item owns.  The workers that take the 6x6 sums at test time keep the range mask
            // whole for the chunks whose sums are theirs and apply the slab to the second mask: below)
            if (SPLIT && !SUMS_AT_TEST) c = fminf(c, slab_one(s));
            if (CONE == FS_CONE_NARROW || (CONE == FS_CONE_ANY && a.cone_mode == 1)) {
                // sphere vs cone of half-angle alpha < pi/2: angle(d, axis) <= alpha + asin(r/|d|), i.e.
                // d.axis >= |d| cos(alpha + beta) = cos(alpha) sqrt(|d|^2 - r^2) - sin(alpha) r  (no division, one hardware
                // square root; the margin dwarfs its 1-ulp error) — or the camera sits inside the sphere (|d|^2 <= r^2)
                const float h = __builtin_amdgcn_sqrtf(fmaxf(d2 - r2, 0.0f));
                const float dot = YAW_ONLY ? dx * ax + dy * ay : dx * ax + dy * ay + dz * az;
                const float lhs = dot - (a.cos_a * h - a.sin_a * s.w - (1.0e-4f * reach + 1.0e-4f));     // |d| <= reach here
                c = fminf(c, fmaxf(lhs, r2 - d2));
            }
            return c;
        };
        if (dead) {
            if (lane == 0) {
                for (int g = 0; g < a.n_groups; ++g) masks[g] = 0ull;
                if (SUMS_AT_TEST) for (int g = 0; g < a.n_groups; ++g) bmasks[g] = 0ull;
                sh_wave_tested[buf][wave] = 0;
                if (SUMS_AT_TEST) sh_wave_box[buf][wave] = 0;
            }
            return;
        }
        for (int gb = 0; gb < a.n_groups; gb += 4) {
            float4 sp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int j = (gb + u) * STRIDE + lane * WAVES + wave;
                if (RESIDENT_SPHERES && gb == 0 && u < N_RES) sp[u] = sp_res[u < N_RES ? u : 0];
                else sp[u] = (gb + u < a.n_groups && j < a.n_chunks)
                                 ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(spheres) + ((uint32_t)j << 4))
                                 : make_float4(0.f, 0.f, 0.f, -1.0e30f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (gb + u < a.n_groups) {
                    const int j = (gb + u) * STRIDE + lane * WAVES + wave;
                    float c = cull_one(j, sp[u]);
                    asm volatile("" : "+v"(c));               // (keeps the compare out here, behind the merge of the two branches)
                    unsigned long long mask = __builtin_amdgcn_ballot_w64(c >= 0.0f);
                    if (SUMS_AT_TEST) {
                        // (a chunk outside the range sphere is outside both masks: the box value only matters where c >= 0)
                        float cb = a.cull ? box_one(sp[u]) : 0.0f;
                        asm volatile("" : "+v"(cb));
                        const unsigned long long bmask = __builtin_amdgcn_ballot_w64(cb >= 0.0f) & mask;
                        tested_box += __popcll(bmask);
                        if (lane == 0) bmasks[gb + u] = bmask;
                        // (SPLIT: the first pass tests the chunks whose test-time sums are this item's — chunk id mod W — and those anybody
                        // can hash from; nothing else in range concerns it)
                        if (SPLIT) mask &= __builtin_amdgcn_ballot_w64(((uint32_t)j & (uint32_t)split_w_mask) == (uint32_t)item_w) | bmask;
                    }
                    tested += __popcll(mask);
                    if (lane == 0) masks[gb + u] = mask;
                }
            }
        }
        if (lane == 0) sh_wave_tested[buf][wave] = tested;         // chunks this wave will scan
        if (SUMS_AT_TEST && lane == 0) sh_wave_box[buf][wave] = tested_box;   // ... and can hash from
    };

    // Pass prediction: distinct voxels <= skip32/32 of the landmarks scanned.  FsFimArgs::skip32 (13/32: p99 0.40 on C3) is the
    // cap for a cloud nothing is known about; once calls on this cloud have been finished, the largest ratio any candidate
    // showed (counters[ratio_slot] = 12 or 13 by what the worker hashes from, in 1/256, kept by the finish kernel; reset with the cloud, the table and the visibility volume)
    // times 5/4 plus 1/32 takes over when it is smaller (only candidates with >= 16 k landmarks scanned count) — a dense slab seen from inside (REF2D) holds many landmarks per voxel
    // and was being scored in two passes for nothing.  A candidate that outgrows the prediction overflows its table, goes to
    // the HBM tier (correct, slower) and raises the ratio for the next call.
    // The number of passes decides the order in which a candidate's fp32 terms are added: with the learnt ratio the LAST BITS of
    // info / trace / log det depend on the calls this context has finished before (and, under fs_multi, on which member scored
    // the candidate: each learns for itself); the integer outputs never do.  "fim.learn" 0 (FsFimArgs::learn) predicts with the
    // fixed cap alone for runs that must reproduce those bits.
    int skip32_eff = a.skip32;
    {
        const uint32_t seen = a.learn ? (uint32_t)a.counters[a.ratio_slot] : 0u;
        if (seen != 0u) {
            const int s = (int)((seen * (uint32_t)a.headroom + 255u) / 256u) + 1;      // headroom / 32 (5/4 by default) of the ratio, + 1/32
            skip32_eff = s < skip32_eff ? (s > 2 ? s : 2) : skip32_eff;
        }
        skip32_eff = __builtin_amdgcn_readfirstlane(skip32_eff);
    }
    // ---- prologue: first candidate, factor table
    // Work items are fetched two candidates ahead, by thread 0, behind an iteration's scoring loop (see there).
    if (tid == 0) {
        const int first = GLOBAL_TABLE ? fetch() : item((int)blockIdx.x);
        sh_next[0] = first;
        sh_next[1] = first >= 0 ? (GLOBAL_TABLE ? fetch() : item((int)blockIdx.x + (int)gridDim.x)) : -1;
        sh_overflow[0] = 0; sh_overflow[1] = 0;
    }
    // (the LDS tier knows its first item without the LDS round trip: its pose record is requested before the barrier, together
    // with the factor table and the resident spheres)
    const int first_item = GLOBAL_TABLE ? -1 : __builtin_amdgcn_readfirstlane(item((int)blockIdx.x));
    PoseLoad first_pose;
    if (!GLOBAL_TABLE && first_item >= 0) first_pose = load_pose(first_item >> split_shift);
    for (int i = tid; i < FS_FACTOR_N; i += THREADS) lfac[i] = a.factor[i];
    __syncthreads();
    // values read back from LDS flags are the same in every lane; readfirstlane tells the compiler so, which keeps the
    // whole persistent loop's control flow scalar (no exec-mask loops, loop-carried counters in SGPRs)
    int cur = __builtin_amdgcn_readfirstlane(sh_next[0]);
    int buf = 0;
    uint32_t cur_key = 0u;                                          // thread 0: sort key of `cur`
    float Rn[9], tn[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rn[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) tn[i] = 0.f;
    if (cur >= 0) {
        if (GLOBAL_TABLE) first_pose = load_pose(cur >> split_shift);
        cur_key = pose_word(first_pose, 13);
        cull(first_pose, 0, Rn, tn, cur & split_w_mask);
    }
    __syncthreads();

#ifdef FS_FIM_STAMPS
    unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 8: inside score calls of the loop, 9: their number
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    while (cur >= 0) {
        const int c = cur >> split_shift;                           // the pose; SPLIT: cur is the item id
        const int split_w = cur & split_w_mask;
        int slab_jlo = 0, slab_jhi = 0;
        if (SPLIT) slab_of(split_w, slab_jlo, slab_jhi);
#ifdef FS_FIM_SCHEDULE
        // development build: when did this workgroup start the candidate and how long did it take (plain stores by one
        // thread, no atomics: the schedule of the persistent grid is read back through fs_get_counter)
        unsigned long long t_cand;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_cand) :: "memory");
#endif
        int wg_tested = 0, wg_box = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) wg_tested += sh_wave_tested[buf][w];
        if (SUMS_AT_TEST) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) wg_box += sh_wave_box[buf][w];
        }
        wg_tested = __builtin_amdgcn_readfirstlane(wg_tested);
        // what the hash table sees: every accepted chunk — or, SUMS_AT_TEST, only those that can meet the table's box (the pass
        // prediction, the table size and the learnt voxel ratio are all relative to this number; later passes scan only these)
        const int wg_hash = SUMS_AT_TEST ? __builtin_amdgcn_readfirstlane(wg_box) : wg_tested;

        // ---- size and clear the hash table.  Distinct voxels are at most ~0.5 of the landmarks scanned (measured
        // p99 0.40, max 0.50): a table with as many slots as landmarks scanned stays below half full, so small
        // candidates clear only a small table.  The probe limit (FS_MAX_PROBE bucket visits) is the overflow detector.
        //
        // A candidate predicted to overfill the table is scored in n_parts PASSES over its accepted chunks: every pass
        // re-tests the landmarks but only queues those whose voxel belongs to the pass (voxel_part), so each landmark is
        // still scored exactly once and each pass needs 1/n_parts of the table.  That costs one extra test per landmark
        // and pass, against a table in HBM with two dependent global atomics per landmark (measured on C5: 3x faster per
        // candidate) or a hand-over to a second kernel with a larger LDS table (round 1: a 0.18 ms serial tail on C3).
        // Prediction: distinct voxels <= skip32/32 of the landmarks scanned (measured p99 0.40 on C3; 13/32 by default);
        // a candidate that overflows a pass all the same is handed to the HBM tier — correct, just slower.
        // (closed forms: written as loops the compiler unrolled and vectorised these few scalar steps into eighty instructions per
        // candidate and wave)
        const uint32_t capacity = 3u << (tier_bits - 2);                             // 3/4 of the largest table
        int n_parts = 1;
        bool skip_tier = false;
        if (!GLOBAL_TABLE) {
            // = landmarks scanned * skip32 / 32, in 32 bits (beyond 2^20 accepted chunks the answer is the HBM tier anyway)
            const uint32_t scanned2 = (uint32_t)(wg_hash < (1 << 20) ? wg_hash : (1 << 20)) * 2u;
            // (SPLIT: with contiguous slabs wg_hash already counts the item's own chunks — those that reach its slab; with interleaved
            // slabs it counts every chunk anybody hashes from, of which the item owns 1/W of the voxels)
            const uint32_t predicted = (scanned2 * (uint32_t)skip32_eff) >> (SUMS_AT_TEST ? split_shift : 0);
            const int k = (predicted > capacity ? 1 : 0) + (predicted > 2u * capacity ? 1 : 0) + (predicted > 4u * capacity ? 1 : 0) +
                          (predicted > 8u * capacity ? 1 : 0);
            n_parts = 1 << k;                                                         // 1, 2, 4, 8 passes
            skip_tier = n_parts > FS_MAX_PARTS;                                       // 16: hopeless here, HBM tier
        }
        int bits = tier_bits;
        if (!GLOBAL_TABLE && n_parts == 1) {
            // smallest table with at least as many slots as landmarks scanned, between 2^10 and the tier's size
            const uint32_t want = ((uint32_t)wg_hash * 64u) >> (SUMS_AT_TEST ? split_shift : 0);       // (3/4 of that measured 4 % slower: fuller buckets, more leftovers)
            const int need = want > 1024u ? 32 - __builtin_clz(want - 1u) : 10;
            bits = need < tier_bits ? need : tier_bits;
        }
        const uint32_t slots = 1u << bits;
        if (!skip_tier && wg_tested > 0) {
            if (GLOBAL_TABLE) {
                for (uint32_t i = tid; i < slots; i += THREADS) table[i] = 0u;
            } else {
                uint4 *t4 = reinterpret_cast<uint4 *>(table);
                for (uint32_t i = tid; i < (slots >> 2); i += THREADS) t4[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // ---- pose (read by this candidate's cull)
        float R[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = tn[i];
        // (vector registers on purpose: the test loop's multiply-adds with the pose as scalar operands measured 3 % slower)
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(R[i]));
#pragma unroll
        for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(t[i]));
        FS_STAMP(0);
        __syncthreads();                                           // table cleared
        FS_STAMP(1);
        const int nxt = __builtin_amdgcn_readfirstlane(sh_next[buf ^ 1]);
        PoseLoad next_pose;                                        // in flight during this candidate's scoring loop
        uint32_t next_key = 0u;
        if (nxt >= 0) next_pose = load_pose(nxt >> split_shift);
        if (tid == 0) sh_overflow[buf ^ 1] = 0;                    // flag of the candidate after this one
        // (the flag is read and written with workgroup-scope atomics on the LDS array itself: a `volatile int *` into it
        // decays to a generic pointer, and FLAT accesses count on vmcnt — every read drained the landmark prefetch)
        auto overflow_flag = [&]() -> int { return __hip_atomic_load(&sh_overflow[buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        auto raise_overflow = [&]() { __hip_atomic_store(&sh_overflow[buf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        bool overflow = false;
        if (skip_tier) { overflow = true; if (tid == 0) raise_overflow(); }

        // (vector copies: a scalar operand turns a 2-cycle multiply or subtract into a 3-cycle one — tools/valu_mix_calib.hip)
        float maxd2_v = a.maxd2, cos2_v = a.cos2, xlo_v = a.box_lo[0];
        asm volatile("" : "+v"(maxd2_v), "+v"(cos2_v), "+v"(xlo_v));
        Acc acc;
        acc.info = 0.f; acc.nvis = 0; acc.nvox = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc.AB[i] = (fs_f2){0.f, 0.f};
        acc.s01 = (fs_f2){0.f, 0.f}; acc.s2 = 0.f;
        acc.pend_old = 0xffffffffu; acc.pend_info = 0.f;
        int qcount = 0;                                            // wave-uniform
        const unsigned long long *masks = Lds<THREADS>::masks(lds, a.n_groups) + wave * a.n_groups;

        for (int part = 0; part < n_parts && !skip_tier; ++part) {
        // (SUMS_AT_TEST: the first pass tests every accepted chunk — sums and count —, the later ones only hash)
        if (SUMS_AT_TEST && part == 1) masks += WAVES * a.n_groups;
        if (part > 0) {
            // next pass: everybody is done with the table -> clear it
            __syncthreads();
            const bool stop = __builtin_amdgcn_readfirstlane(overflow_flag()) != 0;   // uniform: nobody writes the flag between these barriers
            uint4 *t4 = reinterpret_cast<uint4 *>(table);
            for (uint32_t i = tid; i < (slots >> 2); i += THREADS) t4[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            if (stop) break;                                       // an earlier pass ran out of table: next tier
        }
        // No barrier inside this loop.  The wave turns its ballot masks into a FLAT LIST of accepted chunk ids held in one register
        // (lane j = the j-th accepted chunk; compacted through 64 words of the wave's own queue that are free between two chunk
        // tests) and walks it with v_readlane: drawing the next chunk costs one vector and three scalar instructions instead of
        // the ten scalar ones of a find-first-bit walk, and the landmark prefetch no longer drains at every mask boundary.
        // Scalar instructions are the expensive kind here: one costs a wave 8 cycles alone and 16 with four waves per SIMD
        // (tools/valu_mix_calib.hip), more than any vector instruction of this loop.
        {
            uint32_t *scratch = reinterpret_cast<uint32_t *>(qz + 64);
            const uint32_t my_chunk = (uint32_t)(lane * WAVES + wave);
            // One chunk's landmarks in registers: test, compact, and score whenever 64 are queued.
            // (sums_here, wave-uniform: SUMS_AT_TEST under SPLIT — is this chunk one of those whose test-time sums this item takes?)
            auto body = [&](float wx, float wy, float wz, const bool sums_here) {
                // ---- 2. test: p = R^T (w - t), fp32 with the operation order fixed by DESIGN.md "FIM accumulate"
                const float dx = wx - t[0], dy = wy - t[1], dz = wz - t[2];
                const float px = YAW_ONLY ? __fmaf_rn(R[0], dx, R[3] * dy) : __fmaf_rn(R[0], dx, __fmaf_rn(R[3], dy, R[6] * dz));
                const float py = YAW_ONLY ? __fmaf_rn(R[1], dx, R[4] * dy) : __fmaf_rn(R[1], dx, __fmaf_rn(R[4], dy, R[7] * dz));
                const float pz = YAW_ONLY ? dz : __fmaf_rn(R[2], dx, __fmaf_rn(R[5], dy, R[8] * dz));
                const float n2 = __fmaf_rn(px, px, __fmaf_rn(py, py, pz * pz));
                float px2 = px * px;
                asm("" : "+v"(px2));      // (keeps the compiler from pairing this product with cos2 * n2 in a v_pk_mul_f32 that needs two moves)
                // The predicate is kept as ONE float whose sign decides (visible <=> m3 >= 0), so that the ballot below is the mask
                // of a single compare — a boolean merged from several paths is materialised with a select and a second compare.
                float m3;
                if (CONE == FS_CONE_OFF) {
                    // no cone: the sign of maxd2 - n2 alone; INFO_ONLY also drops what lies behind the table's near face (x index
                    // below the first lattice plane: a miss whatever y and z are)
                    m3 = maxd2_v - n2;
                    if (SUMS_AT_TEST) {
                        // the first pass of a candidate takes the sums and the count (later passes re-test the same landmarks)
                        if (part == 0 && (!SPLIT || sums_here)) {
                            const bool in_range = m3 >= 0.0f;
                            add_block_sums(acc, px, py, pz, (in_range && n2 > 0.0f) ? __builtin_amdgcn_rcpf(n2) : 0.0f);
                            acc.nvis += in_range ? 1 : 0;
                        }
                    }
                    if (INFO_ONLY || SUMS_AT_TEST) m3 = fminf(m3, px - xlo_v);
                } else if (CONE == FS_CONE_NARROW) {
                    // n2 <= maxd2 && px >= 0 && px2 >= cos2 * n2 in one value: the sign of a difference of two floats is exact, so each
                    // condition is "its difference is >= 0" and the three are "their minimum is >= 0" (-0.0 passes, as px >= 0 does):
                    // one v_min3_f32 and one compare instead of three compares and the scalar ANDs of their masks.
                    const float c1 = maxd2_v - n2, c3 = px2 - cos2_v * n2;
                    m3 = fminf(fminf(c1, px), c3);
                } else {
                    bool v = (n2 <= a.maxd2);
                    if (a.cone_mode == 1 || a.cone_mode == 3) v = v && (px >= 0.0f) && (px2 >= a.cos2 * n2);
                    else if (a.cone_mode == 2) v = v && ((px >= 0.0f) || (px2 <= a.cos2 * n2));
                    m3 = v ? 0.0f : -1.0f;
                }
                if (SPLIT && SUMS_AT_TEST) {
                    // (interleaved slabs: part id = pass << shift | w)
                    m3 = (voxel_part(a, true, px, n_parts << split_shift) == ((part << split_shift) | split_w)) ? m3 : -1.0f;
                } else if (SPLIT) {
                    // exact ownership: the voxel's x index inside the item's slab — and, when the item needs passes of its own, in this pass
                    const int jx = voxel_jx(a, true, px);
                    const bool mine = (uint32_t)(jx - slab_jlo) < (uint32_t)(slab_jhi - slab_jlo) && ((jx + 4096) & (n_parts - 1)) == part;
                    m3 = mine ? m3 : -1.0f;
                }
                else if (__builtin_expect(n_parts > 1, 0)) m3 = (voxel_part(a, true, px, n_parts) == part) ? m3 : -1.0f;   // wave-uniform branch (out of line: a taken branch costs a wave its instruction buffer); every lane evaluates
                const bool vis = m3 >= 0.0f;
                // ---- 3. compact
                const unsigned long long m = __builtin_amdgcn_ballot_w64(vis);
                if (m != 0ull) {
                    if (vis) {
                        const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        qx[pos] = px; qy[pos] = py; qz[pos] = pz;
                    }
                    qcount += __popcll(m);
                    if (qcount >= 64) {
                        qcount -= 64;                              // take the newest 64: the remainder stays in place
                        const float ex = qx[qcount + lane], ey = qy[qcount + lane], ez = qz[qcount + lane];
                        // ---- 4. score
#ifdef FS_FIM_STAMPS
                        const unsigned long long ts_ = __builtin_readcyclecounter();
#endif
                        score_visible<TABLE_FULL, !INFO_ONLY && !SUMS_AT_TEST, !SUMS_AT_TEST>(a, table, lfac, bits, max_probe, true, ex, ey, ez, acc, overflow);
#ifdef FS_FIM_STAMPS
                        st[8] += __builtin_readcyclecounter() - ts_; st[9] += 1;
#endif
                        if (__any(overflow)) raise_overflow();
                    }
                }
            };
            int g = 0;
            while (g < a.n_groups) {
                if (!GLOBAL_TABLE && __builtin_amdgcn_readfirstlane(overflow_flag())) break;   // another wave ran out of table: the HBM tier redoes it
                // ---- the next (up to) 64 accepted chunks: whole masks, as many as fit
                int n_ids = 0;
                do {
                    unsigned long long mask = masks[g];
                    mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mask) |
                           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mask >> 32)) << 32);   // wave-uniform copy
                    const int cnt = __popcll(mask);
                    if (n_ids + cnt > 64) break;                   // (never with n_ids == 0: a mask has 64 bits)
                    if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
                        const int pos = n_ids + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        scratch[pos] = (uint32_t)(g * STRIDE) + my_chunk;
                    }
                    n_ids += cnt;
                    ++g;
                } while (g < a.n_groups);
                if (n_ids == 0) break;                             // (only when every remaining mask was empty)
                const uint32_t ids = scratch[lane];                // lanes >= n_ids: stale words, never drawn
                auto fetch = [&](int j, float &x, float &y, float &z) -> int {
                    const int c = __builtin_amdgcn_readlane((int)ids, j);
                    const int cc = FS_BOUND(c >= 0 && c < a.n_chunks, 3) ? c : 0;
                    const uint32_t l = (uint32_t)(cc * 64 + lane);
                    x = ldg32(LX, l); y = ldg32(LY, l); z = ldg32(LZ, l);
                    return cc;
                };
                // The next chunk's landmarks are in flight while the current one is tested; three cheap moves hand them over (one
                // loop body, one inlined scoring call).  Past the end of the list the prefetch re-reads the last chunk (never used).
                float nx_, ny_, nz_;
                int chunk_next = fetch(0, nx_, ny_, nz_);
                const int last = n_ids - 1;
                for (int j = 0; j <= last; ++j) {
                    const float wx = nx_, wy = ny_, wz = nz_;
                    const int chunk_now = chunk_next;
                    chunk_next = fetch(min(j + 1, last), nx_, ny_, nz_);
                    // (SPLIT with the sums at test time: the CHUNK decides which of a pose's items takes a landmark's 6x6 sums and counts
                    // it as visible — chunk id mod W, wave-uniform, so the other items skip that code altogether; the hash-table half
                    // stays partitioned by voxel slab.  Every visible landmark is in exactly one chunk: counted exactly once.)
                    body(wx, wy, wz, !SPLIT || !SUMS_AT_TEST || (chunk_now & split_w_mask) == split_w);
                }
            }
        }
        FS_STAMP(2);
        // flush the queue remainder
        if (qcount > 0) {
            const bool have = lane < qcount;
            score_visible<TABLE_FULL, !INFO_ONLY && !SUMS_AT_TEST, !SUMS_AT_TEST>(a, table, lfac, bits, max_probe, have, have ? qx[lane] : 1.0f, have ? qy[lane] : 0.0f, have ? qz[lane] : 0.0f, acc, overflow);
            qcount = 0;
        }
        if (overflow) raise_overflow();
        FS_STAMP(3);
        }   // passes
        resolve_pending<TABLE_FULL>(acc, lfac);

        // ---- the next candidate's cull fills the time this wave would otherwise wait for the slower ones
        // The work item after the next one is drawn HERE, not at the top of the iteration: the device-scope atomic (a round trip
        // of microseconds) is in flight during the cull, and the wait for it and for the list lookup falls into the time this
        // wave would spend at the reduction barrier anyway — at the top, between the two barriers, it held up the whole
        // workgroup once per candidate.
        const bool draw = tid == 0 && nxt >= 0;
        unsigned long long ticket = 0ull;
        if (draw) ticket = atomicAdd(work.counter, 1ull);
        if (nxt >= 0) cull(next_pose, buf ^ 1, Rn, tn, nxt & split_w_mask);
        if (nxt >= 0) next_key = pose_word(next_pose, 13);
        if (tid == 0) sh_next[buf] = draw ? item((int)ticket + preassigned) : -1;   // read by the next iteration after its first barrier

        FS_STAMP(4);
        // ---- 5. reduce: DPP within the wave, then across waves through LDS
        // (INFO_ONLY: three quantities instead of eighteen — info, landmarks scored, voxels)
        constexpr int NRED = INFO_ONLY ? 3 : FS_NACC;
        float vals[NRED];
        vals[0] = acc.info;
        if (!INFO_ONLY) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { vals[1 + i] = acc.AB[i].x; vals[7 + i] = acc.AB[i].y; }
            vals[13] = acc.s01.x; vals[14] = acc.s01.y; vals[15] = acc.s2;
        }
        vals[NRED - 2] = (float)acc.nvis;     // exact: < 2^24 per lane
        vals[NRED - 1] = (float)acc.nvox;
        wave_sums_dpp(vals);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < NRED; ++i) sh_red[i * WAVES + wave] = vals[i];     // [quantity][wave]: one thread reads its row with wide loads
        }
        FS_STAMP(5);
        __syncthreads();
        FS_STAMP(6);
        // cross-wave sums in fp64, one thread per quantity; the 6x6 assembly and log det run in
        // fs_fim_finish_kernel so that this kernel carries no private arrays
        const bool failed = __builtin_amdgcn_readfirstlane(sh_overflow[buf]) != 0;
        // (the sums by the first lanes of wave 1, the flags and counters below by thread 0: the two halves of the epilogue run side
        // by side instead of one after the other on wave 0, which everybody waits for at the next barrier)
        int qi = tid - (WAVES > 1 ? 64 : 0);
        if (qi >= 0 && qi < NRED && !failed) {
            float part[WAVES];
#pragma unroll
            for (int w = 0; w < WAVES; ++w) part[w] = sh_red[qi * WAVES + w];
            double x = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) x += (double)part[w];
            const int q_out = (INFO_ONLY && qi > 0) ? FS_NACC - NRED + qi : qi;     // info stays sum 0, the two counts sums 16 and 17
            // (SPLIT: the partial sums of item `cur`; the HBM tier of a split call writes the whole pose into the pose's first slot)
            const size_t slot = SPLIT ? (size_t)cur : ((size_t)c << (GLOBAL_TABLE ? a.split_shift : 0));
            if (FS_BOUND(c >= 0 && c < a.n, 7)) a.sums[slot * FS_NACC + q_out] = x;
        }
        if (tid == 0) {
            // tier that must re-score the candidate (0: done) | chunks scanned per pass << 4 (the finish kernel's voxel ratio)
            // (the HBM tier keeps the count the LDS worker filed with the hand-over — chunks that can be HASHED, which for the box-culling
            // workers is fewer than the chunks in range this worker scans: the finish kernel's ratio must stay relative to the base
            // the next call's LDS worker predicts with, or a pose that overflowed teaches half its ratio and overflows again)
            uint32_t base = (uint32_t)(wg_hash < (1 << 27) ? wg_hash : (1 << 27));
            if (GLOBAL_TABLE) base = a.overflow[c] >> 4;
            a.overflow[c] = (failed ? fail_code : 0u) | (base << 4);
            const uint32_t lm_tested = SUMS_AT_TEST ? (uint32_t)wg_tested * 64u + (uint32_t)wg_hash * 64u * (uint32_t)(n_parts - 1)   // first pass + the later ones
                                                    : (uint32_t)wg_tested * 64u * (uint32_t)n_parts;
            if (!skip_tier && wg_tested > 0) atomicAdd(&a.tested[c], lm_tested);   // landmarks tested (M_tested), summed by the finish kernel
            // what this candidate cost, filed under its block of the map for the order of the next call (fs_sort.hip)
            // (the key was requested with the pose record: a load here would hold thread 0 — and with it the workgroup's next
            // barrier — for a memory round trip)
            if (!GLOBAL_TABLE && a.costmap) atomicMax(&a.costmap[cur_key & (FS_COST_BINS - 1)], lm_tested);
            if (!GLOBAL_TABLE && n_parts > 1 && !failed) atomicAdd(&a.counters[1], 1ull);   // scored in several passes
            // (SPLIT: the first of a pose's items to fail hands the WHOLE pose over, once; the HBM tier's result then replaces the
            // partial sums — flag 2 tells the finish kernel to read the pose's first slot alone)
            const bool hand_over = failed && !GLOBAL_TABLE && (!SPLIT || (atomicOr(&a.split_flags[c], 1u) & 1u) == 0u);
            if (SPLIT && failed && a.host_flag) __hip_atomic_store(a.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the host then runs the HBM tier and the finish kernel after all)
            if (hand_over) {
                const unsigned long long slot = atomicAdd(&a.counters[2], 1ull);
                if (FS_BOUND(slot < (unsigned long long)a.n, 4)) a.flagged[slot] = c;          // work list of the HBM tier
            } else if (failed && GLOBAL_TABLE) {
                atomicAdd(&a.counters[3], 1ull);
            }
            if (GLOBAL_TABLE && a.split_flags && !failed) atomicOr(&a.split_flags[c], 2u);
        }
        cur = nxt;
        cur_key = next_key;
        buf ^= 1;
        FS_STAMP(7);
#ifdef FS_FIM_SCHEDULE
        if (tid == 0 && !GLOBAL_TABLE && c < FS_SCHEDULE_MAX) {
            unsigned long long t_now;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now) :: "memory");
            a.counters[32 + 2 * c] = t_cand;                       // 100 MHz ticks
            a.counters[32 + 2 * c + 1] = ((t_now - t_cand) & 0xffffffffull) | ((unsigned long long)blockIdx.x << 32) | ((unsigned long long)n_parts << 56);
        }
#endif
    }
#ifdef FS_FIM_STAMPS
    if (lane == 0 && !GLOBAL_TABLE && THREADS == FS_T1_THREADS) {
        for (int k = 0; k < 10; ++k) atomicAdd(&a.counters[16 + k], st[k]);
        atomicAdd(&a.counters[31], 1ull);
#ifdef FS_FIM_STAMPS_PER_WAVE   // wait at the reduction barrier / scoring time by wave index (counters 32.. of an enlarged block)
        atomicAdd(&a.counters[32 + wave], st[6]);
        atomicAdd(&a.counters[48 + wave], st[2] + st[3]);
#endif
    }
#endif
}

// One thread per candidate: assemble the 6x6 FIM from the 15 block sums, trace, log det.  Called by every lane of whole
// waves (the per-wave reductions below shuffle over all 64 lanes; c >= a.n lanes only take part in those).
__device__ __forceinline__ void finish_body(const FsFimArgs &a, const int c)
{
    {   // landmark-test count of this block's candidates -> one atomic per block
        unsigned long long tsum = (c < a.n) ? (unsigned long long)a.tested[c] : 0ull;
        if (c < a.n) a.tested[c] = 0u;
        for (int d = 32; d >= 1; d >>= 1) tsum += __shfl_xor(tsum, d);
        if ((threadIdx.x & 63) == 0 && tsum) { atomicAdd(&a.counters[0], tsum); atomicAdd(&a.counters[10], tsum); }
    }
    {   // distinct voxels per landmark scanned, in 1/256: the largest any finished candidate showed (fim_worker's pass prediction)
        uint32_t r = 0u;
        if (c < a.n) {
            const uint32_t ov = a.overflow[c];
            const unsigned long long scanned = (unsigned long long)(ov >> 4) * 64ull;
            // (only candidates big enough to come near the decision count: a pose at the cloud's edge with one chunk of 64 landmarks
            // in 64 voxels says nothing about the crowded ones)
            if ((ov & 15u) == 0u && scanned >= 16384ull && a.split_shift == 0) {
                const uint32_t nv = (uint32_t)(a.sums[(size_t)c * FS_NACC + 17] + 0.5);
                r = (uint32_t)(((unsigned long long)nv * 256ull + scanned - 1ull) / scanned);
            }
        }
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(r, d); r = o > r ? o : r; }
        if ((threadIdx.x & 63) == 0 && r > (uint32_t)a.counters[a.ratio_slot]) atomicMax(&a.counters[a.ratio_slot], (unsigned long long)r);
    }
    if (c == 0) {                                            // per-call tier counters -> running totals
        a.counters[4] += a.counters[1];
        a.counters[5] += a.counters[2];
        a.counters[6] += a.counters[3];
        atomicAdd(&a.counters[11], (unsigned long long)a.n);
        a.counters[1] = 0ull; a.counters[2] = 0ull; a.counters[3] = 0ull; a.counters[7] = 0ull; a.counters[8] = 0ull; a.counters[9] = 0ull;
    }
    if (c >= a.n) return;
    if (a.info_only) {                                       // isPoseSafe's scalar and the voxel count: nothing else was accumulated
        const double *S1 = a.sums + ((size_t)c << a.split_shift) * FS_NACC;
        double info = S1[0], nvox = S1[17];
        if (a.split_flags) {
            // one pose over W workgroups: the W partial sums — unless the HBM tier redid the whole pose (flag 2: first slot alone)
            const uint32_t flags = a.split_flags[c];
            a.split_flags[c] = 0u;                           // (left clean for the next call)
            if (!(flags & 2u))
                for (int w = 1; w < (1 << a.split_shift); ++w) { info += S1[(size_t)w * FS_NACC]; nvox += S1[(size_t)w * FS_NACC + 17]; }
        }
        a.info_ref[c] = (float)info;
        a.n_voxels[c] = (int)(nvox + 0.5);
        return;
    }
    // (one pose over W workgroups: the W partial sums of every quantity — unless the HBM tier redid the whole pose, whose result
    // then stands alone in the pose's first slot)
    double S[FS_NACC];
    {
        const double *S0 = a.sums + ((size_t)c << a.split_shift) * FS_NACC;
#pragma unroll
        for (int q = 0; q < FS_NACC; ++q) S[q] = S0[q];
        if (a.split_flags) {
            const uint32_t flags = a.split_flags[c];
            a.split_flags[c] = 0u;
            if (!(flags & 2u))
                for (int w = 1; w < (1 << a.split_shift); ++w) {
#pragma unroll
                    for (int q = 0; q < FS_NACC; ++q) S[q] += S0[(size_t)w * FS_NACC + q];
                }
        }
    }
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
    // log det from the pivots of the square-root-free Cholesky factorisation F = L D L^T (fp64): log det = log(prod d_j),
    // one logarithm and six reciprocals instead of six logarithms, six square roots and fifteen divisions on one lane.
    // -inf when singular: fewer than 3 landmarks can never give rank 6, and a pivot below 1e-6 of its diagonal entry is
    // rounding noise of the fp32 sums.  (The product of six pivots of float32-range sums cannot leave the fp64 range.)
    const int nvis = (int)(S[16] + 0.5);
    double Lm[6][6];                                         // unit lower triangle, strictly below the diagonal
    double dv[6];
    double prod = 1.0;
    bool pd = nvis >= 3;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = F[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k] * dv[k];
        if (!(d > 1e-6 * F[j][j])) pd = false;
        const double dj = pd ? d : 1.0;
        dv[j] = dj;
        prod *= dj;
        const double inv = 1.0 / dj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = F[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k] * dv[k];
            Lm[i][j] = s * inv;
        }
    }
    const double ld = log(prod);
    const float logdet = pd ? (float)ld : -INFINITY;
    const int nvox = (int)(S[17] + 0.5);
    a.logdet[c] = logdet;
    a.n_visible[c] = nvis;
    a.n_voxels[c] = nvox;
    if (a.records) {                                         // same layout as fs_pack_kernel
        fs_record r;
        r.arrival = a.rec_arrival[c];
        r.argmax = a.rec_argmax[c];
        r.yaw = (float)a.rec_yaw[c];
        r.info_ref = (float)S[0];
        r.trace = (float)((A[0] + A[3] + A[5]) + (B[0] + B[3] + B[5]));
        r.logdet = logdet;
        r.n_visible = nvis;
        const uint32_t nv = (uint32_t)nvox;
        r.flags = (a.rec_achievable[c] ? FS_FLAG_ACHIEVABLE : 0u) | (((uint32_t)a.status[c] & 0xffu) << 8) |
                  ((nv > 65535u ? 65535u : nv) << 16);
        a.records[c] = r;
    }
}

__global__ void fs_fim_finish_kernel(const FsFimArgs a)
{
    finish_body(a, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// LDS tier: persistent workgroups (two per CU) pull candidates of the processing order from a device-side counter
// (two workgroups of 8 waves per CU = 4 waves per SIMD: the register allocator must stay within 128 VGPRs)
template <int THREADS, bool TABLE_FULL, int CONE, bool INFO_ONLY, bool YAW_ONLY, bool SPLIT = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(FS_T1_WAVES_PER_EU)))
void fs_fim_kernel(const FsFimArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t fs_fim_lds[];
    const FimWork work{a.cand_perm, a.cand_lo, a.cand_count, a.counters + 8};
    constexpr int MASK_SETS = (CONE == FS_CONE_OFF && !INFO_ONLY) ? 2 : 1;      // (= SUMS_AT_TEST of the worker)
    fim_worker<THREADS, false, TABLE_FULL, CONE, INFO_ONLY, YAW_ONLY, SPLIT>(a, work, fs_fim_lds, Lds<THREADS>::table(fs_fim_lds, a.n_groups, MASK_SETS), a.hash_bits);
}

// HBM tier: candidates the LDS tier appended to its work list; table in HBM with 2x the landmark count of slots and
// unbounded probing
template <int THREADS, bool TABLE_FULL>
__global__ __launch_bounds__(THREADS)
void fs_fim_tier3_kernel(const FsFimArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t fs_fim_lds[];
    const FimWork work{a.flagged, 0, (int)a.counters[2], a.counters + 9};
    uint32_t *table = a.gtable + ((size_t)blockIdx.x << a.ghash_bits);
    fim_worker<THREADS, true, TABLE_FULL, FS_CONE_ANY, false, false>(a, work, fs_fim_lds, table, a.ghash_bits);
}

template <int THREADS>
size_t lds_bytes(int hash_bits, bool global_table, int n_chunks, int *n_groups, int mask_sets)
{
    constexpr int WAVES = THREADS / 64;
    *n_groups = (n_chunks + WAVES * 64 - 1) / (WAVES * 64);
    const size_t mask_words = Lds<THREADS>::mask_words(*n_groups, mask_sets);   // table stays 16-byte aligned
    return sizeof(uint32_t) * (Lds<THREADS>::fixed_words + mask_words + (global_table ? 0 : ((size_t)1 << hash_bits)));
}

// workgroups of `kernel` that are resident on the device at once (persistent grids are launched with exactly that many)
template <typename K>
int resident_blocks(K kernel, int threads, size_t lds)
{
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel), threads, lds) != hipSuccess || per_cu <= 0) per_cu = 1;
    return cus * per_cu;
}

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace


namespace {

template <bool TABLE_FULL, int CONE, bool INFO_ONLY, bool YAW_ONLY, bool SPLIT = false>
hipError_t launch_tier1(FsFimArgs &a, hipStream_t s)
{
    const size_t lds = lds_bytes<FS_T1_THREADS>(a.hash_bits, false, a.n_chunks, &a.n_groups, (CONE == FS_CONE_OFF && !INFO_ONLY) ? 2 : 1);
    auto kernel = fs_fim_kernel<FS_T1_THREADS, TABLE_FULL, CONE, INFO_ONLY, YAW_ONLY, SPLIT>;
    hipError_t e = allow_lds(kernel, lds);
    if (e != hipSuccess) return e;
    // same kernel, same device, same LDS budget class: the occupancy query is made once (per instantiation) and remembered in ONE
    // atomic word — (device, LDS bytes, resident workgroups) — because contexts on different host threads launch through here at
    // the same time ("Threads"): a reader sees a complete triple or recomputes, never half of an update
    static std::atomic<uint64_t> cache{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = ((uint64_t)(dev & 0xff) << 44) | ((uint64_t)lds << 20);
    uint64_t seen = cache.load(std::memory_order_relaxed);
    if ((seen & ~0xfffffull) != key || (seen & 0xfffffull) == 0) {
        seen = key | (uint64_t)(resident_blocks(kernel, FS_T1_THREADS, lds) & 0xfffff);
        cache.store(seen, std::memory_order_relaxed);
    }
    const int resident = (int)(seen & 0xfffffull);
    hipLaunchKernelGGL(kernel, dim3(a.cand_count < resident ? a.cand_count : resident), dim3(FS_T1_THREADS), lds, s, a);
    return hipGetLastError();
}

template <bool TABLE_FULL>
hipError_t launch_overflow(FsFimArgs &a, int pool, hipStream_t s)
{
    const int blocks = a.n < pool ? a.n : pool;
    const size_t lds3 = lds_bytes<FS_HBM_THREADS>(0, true, a.n_chunks, &a.n_groups, 1);
    auto kernel3 = fs_fim_tier3_kernel<FS_HBM_THREADS, TABLE_FULL>;
    hipError_t e = allow_lds(kernel3, lds3);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel3, dim3(blocks), dim3(FS_HBM_THREADS), lds3, s, a);
    return hipGetLastError();
}

}  // namespace

// one pose over several workgroups (FsFimArgs::split_shift) exists for finite tables and the two common cone modes
bool fs_fim_can_split(const FsFimArgs &a)
{
    const int cone = a.cone_mode == 0 ? FS_CONE_OFF : (a.cone_mode == 1 ? FS_CONE_NARROW : FS_CONE_ANY);
    return a.table_full && cone != FS_CONE_ANY;
}

// which learnt voxel ratio a call works with (FsFimArgs::ratio_slot): 13 where the LDS worker that runs hashes from the chunks that
// can meet the table's box (cone off: SUMS_AT_TEST; INFO_ONLY), 12 where it hashes from everything in range and cone
static int ratio_slot_of(const FsFimArgs &a)
{
    const int cone = a.cone_mode == 0 ? FS_CONE_OFF : (a.cone_mode == 1 ? FS_CONE_NARROW : FS_CONE_ANY);
    const bool special = a.table_full && cone != FS_CONE_ANY;
    return (cone == FS_CONE_OFF || (special && a.info_only)) ? 13 : 12;
}

hipError_t fs_launch_fim(const FsFimArgs &a0, hipStream_t s)
{
    if (a0.n <= 0 || a0.cand_count <= 0) return hipSuccess;
    FsFimArgs a = a0;
    a.ratio_slot = ratio_slot_of(a);
    const int cone = a.cone_mode == 0 ? FS_CONE_OFF : (a.cone_mode == 1 ? FS_CONE_NARROW : FS_CONE_ANY);
    if (a.split_flags && !fs_fim_can_split(a)) return hipErrorInvalidValue;   // (the host asks fs_fim_can_split before it sets up a split call)
    // The specialised workers exist for finite tables (every generated one) and the two common cone modes; anything else runs the
    // general worker, which computes a superset (the finish kernel then hands out what was asked for).
    // (the cone-off workers size their passes by the chunks that can meet the table's box — SUMS_AT_TEST, INFO_ONLY —, and a pose
    // shows more distinct voxels per landmark of THOSE than of everything in range: the cap of the pass prediction follows)
    if (cone == FS_CONE_OFF && a.skip32 < 20) a.skip32 = 20;
    const bool special = a.table_full && cone != FS_CONE_ANY;
    if (a.split_flags) {                                       // few poses: each over W workgroups — W = 1 included (special holds: fs_fim_can_split)
        if (a.info_only) return cone == FS_CONE_OFF ? launch_tier1<true, FS_CONE_OFF, true, false, true>(a, s) : launch_tier1<true, FS_CONE_NARROW, true, false, true>(a, s);
        if (a.yaw_only) return cone == FS_CONE_OFF ? launch_tier1<true, FS_CONE_OFF, false, true, true>(a, s) : launch_tier1<true, FS_CONE_NARROW, false, true, true>(a, s);
        return cone == FS_CONE_OFF ? launch_tier1<true, FS_CONE_OFF, false, false, true>(a, s) : launch_tier1<true, FS_CONE_NARROW, false, false, true>(a, s);
    }
    if (!special || (!a.info_only && !a.yaw_only)) {
        if (cone == FS_CONE_OFF) return a.table_full ? launch_tier1<true, FS_CONE_OFF, false, false>(a, s) : launch_tier1<false, FS_CONE_OFF, false, false>(a, s);
        if (cone == FS_CONE_NARROW) return a.table_full ? launch_tier1<true, FS_CONE_NARROW, false, false>(a, s) : launch_tier1<false, FS_CONE_NARROW, false, false>(a, s);
        return a.table_full ? launch_tier1<true, FS_CONE_ANY, false, false>(a, s) : launch_tier1<false, FS_CONE_ANY, false, false>(a, s);
    }
    // (info_only comes from fs_score_fim, whose poses are general; the fused path — the one with yaw-only poses — always wants
    // the full records: the two flags never meet, and no INFO_ONLY x YAW_ONLY worker is instantiated)
    if (a.info_only) return cone == FS_CONE_OFF ? launch_tier1<true, FS_CONE_OFF, true, false>(a, s) : launch_tier1<true, FS_CONE_NARROW, true, false>(a, s);
    return cone == FS_CONE_OFF ? launch_tier1<true, FS_CONE_OFF, false, true>(a, s) : launch_tier1<true, FS_CONE_NARROW, false, true>(a, s);
}

hipError_t fs_launch_fim_overflow(const FsFimArgs &a0, int pool, hipStream_t s)
{
    if (a0.n <= 0) return hipSuccess;
    FsFimArgs a = a0;
    a.ratio_slot = ratio_slot_of(a);
    return a.table_full ? launch_overflow<true>(a, pool, s) : launch_overflow<false>(a, pool, s);
}

hipError_t fs_launch_fim_finish(const FsFimArgs &a0, hipStream_t s)
{
    if (a0.n <= 0) return hipSuccess;
    FsFimArgs a = a0;
    a.ratio_slot = ratio_slot_of(a);
    hipLaunchKernelGGL(fs_fim_finish_kernel, dim3((a.n + 63) / 64), dim3(64), 0, s, a);      // one wave per block: 20 k candidates spread over all CUs
    return hipGetLastError();
}
