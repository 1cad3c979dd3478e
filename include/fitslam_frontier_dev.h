/*
 * fitslam_frontier_dev.h — development and measurement entry points of libfitslam_frontier.so.
 *
 * Nothing here replaces a reference interface: these calls exist for bench.py (per-kernel device time, the M_tested counter of
 * the roofline figure), for the test-suite (the fp64 self test, forcing a code path) and for experiments.  A drop-in user of
 * the scoring path needs fitslam_frontier.h only.  Builds with instrumentation compiled in (per-phase cycle stamps, the
 * schedule recorder, range-checked accesses) are a separate matter: they are made by fit-slam_amd/_build.py when one of
 * its FS_* environment knobs is set, carry -DFS_DEV, and go to a library file of their own.
 */
#ifndef FITSLAM_FRONTIER_DEV_H_
#define FITSLAM_FRONTIER_DEV_H_

#include "fitslam_frontier.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel device time.  With timing enabled every kernel launch is bracketed by hipEvents on
 * the context's stream; fs_kernel_time returns and resets the accumulated (ms, launches) of
 * kernel kind: 0 ray-march, 1 FIM accumulate, 2 FIM HBM-table tier, 3 utility/rank, 4 candidate sort,
 * 5 frontier-cell stencil. */
int  fs_enable_kernel_timing(fs_ctx *ctx, int enable);
int  fs_kernel_time(fs_ctx *ctx, int kind, double *total_ms, int64_t *launches);

/* Tuning / diagnostic knobs (no reference counterpart).  Keys: "fim.cull" (default 1): per-chunk
 * bounding-sphere culling of the landmark cloud; 0 tests every landmark (brute force; identical results).
 * "ray.sort" (default 1): ray-march candidates in Morton order of their goal cell (lists of >= 2048).
 * "sort.costmap" (default 1): that sort puts the blocks of the map whose candidates were expensive in the previous call
 * first (the persistent FIM grid drains better; identical results); "sort.reverse" (default 0): reversed block order, a
 * knob for measuring how much the order matters.
 * "ray.layout" (default 0): which image of the grid the arrival fan walks — 0 picks by ray length (row-major byte image below
 * 96 cells on 2-D maps, the 2-bit class image in 8 x 8 x 8-cell bricks on 3-D grids and from there on: the measured winners), 1 / 2 force the byte / class
 * walk (identical results in every case; DESIGN.md 4.1 holds the measured table); 3 = the class image as a brick table + a pool
 * of distinct bricks (BASELINE.json configs[4]'s "sparse" form: 2.6x smaller, half the L2 traffic, 40 % SLOWER — an experiment kept
 * reproducible, profiles/EXPERIMENTS.md).
 * "fim.bits1", "fim.skip32": development knobs of the hash table (size; predicted share of distinct voxels among the
 * landmarks scanned, in 32nds, that decides the number of scoring passes — the cap under which the worker uses what
 * finished calls on the cloud have shown: DESIGN.md 4.2).
 * "fim.learn" (default 1): 0 = predict the passes with the fixed cap alone.  The number of passes decides the order in which a
 * candidate's float32 terms are added, so with learning on the LAST BITS of info / trace / log det depend on the calls a
 * context has finished before (and, under fs_multi, on the member that scored the candidate); integers never do.
 * "zerocopy" (default 1): host-buffer calls of up to 1024 candidates / poses let their kernels read the inputs from, and write
 * the results into, the context's page-locked staging buffers, which are mapped into the device's address space — no transfer
 * operations on the stream (one pose with five result columns: 66 -> 50 us; profiles/r04/small_call_in_place_staging.json);
 * 0 = transfers, as every larger call uses.  Identical results.
 * "graph" (default 0): 1 = the small host-buffer calls (fs_score_candidates / fs_get_frontier_costs up to 1024 candidates, fs_score_fim
 * up to 4 poses) replay a captured launch graph instead of launching their kernels one by one (identical results).  Measured
 * 5-7 us slower per call than plain launches on ROCm 7.2 (profiles/r04/small_call_graphs.json), hence off; never used while
 * kernel timing is enabled.
 * "fim.split" (default 3): a call with so few poses that the chip would idle — fs_score_fim with one or a few poses, fs_score_candidates
 * with a handful of frontiers — spreads each pose over up to 2^value workgroups by voxel slab (all n * W items resident, >= 128
 * landmark chunks per workgroup); 0 = one workgroup per pose.  Identical integers, the same multiset of information terms.  (One
 * isPoseSafe pose at the reference's visibility request: 60 -> 42 us — 34 us with "fim.hostfinish"; 50 frontiers through the fused path: 122 -> 104 us.)
 * "fim.hostfinish" (default 1): a split info-only call (the isPoseSafe call) is ONE launch — its items write their partial sums
 * into mapped page-locked memory and the host adds them after the synchronisation (same additions, same order as the finish
 * kernel); the HBM-tier and finish launches follow only if an item ran out of table.  0 = always three launches.
 * "cloud.order" (default 1): where fs_upload_landmarks puts the cloud into the leaf order of the k-d tree with 64 landmarks per leaf —
 * 2 on the device (one keys kernel + one sort per level, fs_cloud.hip: 0.7 ms at 100 k landmarks, 1.9 ms at 500 k),
 * 0 on the host (its top levels on threads of their own: 4.4 / 20.2 ms; FS_KD_THREADS=1 in the environment keeps it on the calling
 * thread: 13.3 / 78.9 ms), 1 on the device from 4096 landmarks on (a level costs the device its launches whatever its size: a 2 k
 * cloud takes 0.06 ms on the host, 0.18 ms on the device).  Same split rule; tied coordinates may land in different leaves, so the
 * last bits of the float columns differ between the two (the order of summation); integers never do.  Set before the upload.
 * "fim.specialise" (default 1): 0 = always the general FIM worker (no INFO_ONLY / YAW_ONLY instantiation; identical integers,
 * float sums to the last bits) — the A/B switch of tests/test_gpu_reference_visibility.py. */
int  fs_set_option(fs_ctx *ctx, const char *key, double value);
/* Device-side counters: 0 = landmark visibility tests performed (M_tested summed over candidates),
 * 4 = candidates scored in several voxel-partitioned passes, 5 = candidates re-scored with the table in HBM,
 * 6 = unresolved (always 0), 12 / 13 = the largest voxels-per-landmark ratio (in 1/256) a big candidate of the finished calls
 * showed (the pass prediction's input; reset with the cloud, the table and the visibility volume) — 12 per landmark of the
 * chunks in range and cone (the cone workers), 13 per landmark of the chunks that can also meet the lookup table's box (the
 * cone-off and info-only workers); 10 / 11 = landmark tests / candidates since the last spatially sorted call (the sort's own
 * accumulators: CLEARED by every call of 2048 candidates or more — not running totals).  Host-side: 1000 / 1001 = bricks of the grid / bricks in the pool of "ray.layout" 3. */
int  fs_get_counter(fs_ctx *ctx, int which, int64_t *value, int reset);

/* ---------------------------------------------------------------- self test */

/* Runs the fp64 primitives the ray set-up relies on (sqrt of exact small integers, division) on the
 * device and compares them with the host's correctly rounded results bit for bit.
 * *mismatches = 0 is required for bit-exact ray geometry. */
int fs_selftest_fp64(fs_ctx *ctx, int32_t max_abs, int64_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif
