"""Randomised differential test: random grids, fans, visitors, clouds and visibility volumes through the C ABI against the
oracle.  Same bar as test_gpu_parity.py: integers bit-exact, Fisher information within 1e-4."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-4
# by hand, for a wider sweep than the suite's 150 configurations: FS_FUZZ_SEEDS=3000 FS_FUZZ_BASE=100000 python -m pytest tests/test_gpu_fuzz.py -m gpu
N_SEEDS = int(os.environ.get("FS_FUZZ_SEEDS", "150"))
SEED_BASE = 9000 + int(os.environ.get("FS_FUZZ_BASE", "0"))
TINY = bool(os.environ.get("FS_FUZZ_TINY"))      # every map a tiny one


def _random_case(rng, fs):
    nx, ny = int(rng.integers(9, 90)), int(rng.integers(9, 90))
    nz = int(rng.choice([1, 1, 2, 5, 12]))
    if TINY or rng.random() < 0.04:                                         # maps of a few cells (1 x 1 x 1 included)
        nx, ny, nz = int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.choice([1, 1, 2, 3]))
    res = float(rng.choice([0.05, 0.1, 0.037]))
    origin = (float(rng.uniform(-3, 0.5)), float(rng.uniform(-3, 0.5)), float(rng.uniform(-0.5, 0.0)) if nz > 1 else 0.0)
    vals = np.array([0, 0, 0, 0, 255, 255, 255, 254, 253, 240, 100, 239], np.uint8)
    cells = rng.choice(vals, size=(nz, ny, nx))
    if rng.random() < 0.3:
        cells[:] = 255
        cells[:, ny // 2, :] = 254
    n = int(rng.integers(1, 70))
    lo = np.array(origin)
    hi = lo + np.array([nx, ny, nz]) * res
    goals = rng.uniform(lo - 0.2, hi + 0.2, size=(n, 3))
    if nz == 1:
        goals[:, 2] = origin[2]
    goals[: n // 2] = np.clip(goals[: n // 2], lo + 1e-6, hi - 1e-6)       # at least half on the map
    n_elev = 1 if nz == 1 else int(rng.integers(1, 4))
    elev = tuple(sorted(rng.uniform(-0.4, 0.4, size=n_elev).tolist())) if nz > 1 else (0.0,)
    use_delta = rng.random() < 0.5
    ray = dict(max_camera_depth=float(rng.uniform(0.3, 5.0)), delta_theta=float(rng.uniform(0.07, 0.6)),
               camera_fov=float(rng.uniform(0.3, 2.0)), robot_radius=float(rng.uniform(0.05, 0.8)),
               n_rays=0 if use_delta else int(rng.integers(8, 80)), elev=elev,
               obst=tuple(sorted(rng.integers(200, 262, size=2).tolist())) if rng.random() < 0.7 else (260, 260),
               trace=tuple(sorted(rng.integers(0, 258, size=2).tolist())),
               polygon=(lo[0] + rng.uniform(-1, 1), lo[1] + rng.uniform(-1, 1), hi[0] + rng.uniform(-1, 1), hi[1] + rng.uniform(-1, 1)))
    if not use_delta:
        ray["delta_theta"] = 2 * np.pi / ray["n_rays"]
    m = int(rng.choice([0, 1, 63, 64, 65, 300, 2000]))
    lm = rng.uniform(lo - 1.0, hi + 1.0, size=(m, 3)).astype(np.float32)
    if m > 10 and rng.random() < 0.3:
        lm[: m // 3] = lm[0]                                               # a crowded voxel
    fim = dict(max_dist=float(rng.uniform(0.5, 16.0)), max_angle=float(rng.choice([0.3, 1.0, 1.5, 1.5707963, 2.2, 3.2, 4.0])))
    return dict(cells=cells, origin=origin, res=res, goals=goals, ray=ray, lm=lm, fim=fim,
                fsize=rng.integers(1, 30, size=n).astype(np.int32), black=(rng.random(n) < 0.1).astype(np.uint8),
                achin=(rng.random(n) < 0.9).astype(np.uint8))


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_configuration(fs, oracle, scorer, ref_table, seed):
    rng = np.random.default_rng(SEED_BASE + seed)
    k = _random_case(rng, fs)
    G = oracle.Grid(k["cells"], origin=k["origin"], resolution=k["res"])
    P = oracle.RayParams(**k["ray"])
    n_yaw = oracle.num_yaw_rays(k["ray"]["delta_theta"], k["ray"]["n_rays"])
    window = int(k["ray"]["camera_fov"] / k["ray"]["delta_theta"])
    if n_yaw < max(window, 1) or window < 1:
        with pytest.raises(fs.FsError):
            scorer.set_ray_params(**k["ray"])
        return
    scorer.set_ray_params(**k["ray"])
    scorer.upload_grid(k["cells"], k["origin"], k["res"])
    scorer.upload_landmarks(k["lm"])
    scorer.set_fim_params(**k["fim"])
    layout = int(rng.integers(0, 3))
    scorer.set_option("ray.layout", layout)
    try:
        mx_o = oracle.max_arrival_information(G, P)
        assert scorer.max_arrival() == mx_o
        want = oracle.arrival_information(G, P, k["goals"], k["fsize"], k["black"], k["achin"], min_gt=mx_o["min_gt"], faithful=True)
        got = scorer.score_arrival(k["goals"], k["fsize"], k["black"], k["achin"])
        for f in ("status", "arrival", "argmax", "achievable", "ray_counts"):
            np.testing.assert_array_equal(got[f], want[f], err_msg=f)
        np.testing.assert_array_equal(got["yaw"], want["yaw"])
        # fused records: FIM at (goal, best yaw)
        rec = scorer.score_candidates(k["goals"], k["fsize"], k["black"], k["achin"])
        np.testing.assert_array_equal(rec["arrival"], want["arrival"])
        np.testing.assert_array_equal((rec["flags"] >> 8) & 0xFF, want["status"])
        ok = want["status"] == 0
        poses = oracle.poses_from_yaw(k["goals"], want["yaw"])
        fim = oracle.pose_information(ref_table, k["lm"], poses[ok], k["fim"]["max_dist"], k["fim"]["max_angle"], n_threads=4)
        np.testing.assert_array_equal(rec["n_visible"][ok], fim["n_visible"])
        np.testing.assert_array_equal(((rec["flags"] >> 16) & 0xFFFF)[ok], np.minimum(fim["n_voxels"], 65535))
        sc = np.maximum(np.abs(fim["info_f64"]), 1e-6)
        assert np.all(np.abs(rec["info_ref"][ok] - fim["info_f64"]) / sc <= REL)
        tr = np.maximum(np.abs(fim["trace"]), 1e-6)
        assert np.all(np.abs(rec["trace"][ok] - fim["trace"]) / tr <= REL)
        assert not rec["info_ref"][~ok].any() and not rec["n_visible"][~ok].any()
    finally:
        scorer.set_option("ray.layout", 0)


N_SEG_SEEDS = int(os.environ.get("FS_SEGMENT_SEEDS", "60"))


@pytest.mark.parametrize("seed", range(N_SEG_SEEDS))
def test_random_segments(fs, oracle, scorer, seed):
    """fs_trace_segments (the roadmap's isConnectable walk, DEP/src/planners/FrontierRoadmap.cpp:716-737; RayTracedCells with any
    visitor ranges) on the random maps of this file: random end points — on the map, off it, in one cell, axis-aligned, along the
    border —, random length limits and visitor ranges; every accessor (ok, cells traced, obstacle hit, unknown cells, all cells)
    against the oracle's single-ray trace."""
    rng = np.random.default_rng(SEED_BASE + 70_000 + seed)
    k = _random_case(rng, fs)
    G = oracle.Grid(k["cells"], origin=k["origin"], resolution=k["res"])
    nz, ny, nx = k["cells"].shape
    lo = np.array(k["origin"]); hi = lo + np.array([nx, ny, nz]) * k["res"]
    n = int(rng.integers(1, 120))
    a = rng.uniform(lo - 0.1, hi + 0.1, size=(n, 3)); b = rng.uniform(lo - 0.1, hi + 0.1, size=(n, 3))
    inside = rng.random(n) < 0.8
    a[inside] = np.clip(a[inside], lo + 1e-9, hi - 1e-9); b[inside] = np.clip(b[inside], lo + 1e-9, hi - 1e-9)
    same = rng.random(n) < 0.1
    b[same] = a[same]
    axis = rng.random(n) < 0.2
    b[axis, 1:] = a[axis, 1:]
    if nz == 1:
        a[:, 2] = b[:, 2] = k["origin"][2]
    border = rng.random(n) < 0.1
    a[border, 0] = lo[0]; b[border, 0] = hi[0] - 1e-9
    obst = tuple(sorted(rng.integers(0, 262, size=2).tolist()))
    trace = tuple(sorted(rng.integers(0, 262, size=2).tolist()))
    max_len = float(rng.choice([1, 3, 17, 183, 1000]))
    scorer.upload_grid(k["cells"], k["origin"], k["res"])
    got = scorer.trace_segments(a, b, max_len, obst=obst, trace=trace)
    for i in range(n):
        r = oracle.trace_ray(G, tuple(a[i]), tuple(b[i]), max_len, obst=obst, trace=trace, faithful=True)
        assert bool(got["ok"][i]) == r["ok"], (i, a[i], b[i])
        if r["ok"]:
            assert (got["traced"][i], bool(got["hit"][i]), got["unknown"][i], got["all"][i]) == (r["traced"], r["hit"], r["unknown"], r["all"]), \
                (i, a[i], b[i], max_len, obst, trace)
