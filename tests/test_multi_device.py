"""fs_multi — one process, one calling thread, several devices (include/fitslam_frontier.h).

not-gpu: the partition rule (fs_multi_shard_bounds, callable without a device) — contiguous, ordered, complete, equal to the
         rule of the multi-process path (shard.py); a device list without a gfx950 fails loudly.
gpu:     a device list [0, 0] / [0, 0, 0] (several contexts on the one GPU of the test box) returns the records of
         a single context (integers bit for bit) — the blocks are scored side by side and land in list order."""
import importlib

import numpy as np
import pytest


@pytest.mark.parametrize("n", [0, 1, 2, 7, 8, 9, 1000, 20_000, 160_000, 2_147_483_647])
@pytest.mark.parametrize("g", [1, 2, 3, 8, 64])
def test_shard_bounds_partition(fs, n, g):
    shard = importlib.import_module("fit-slam_amd.shard")
    prev = 0
    per = -(-n // g)
    for s in range(g):
        lo, hi = fs.capi.shard_bounds(n, g, s)
        assert lo == prev and lo <= hi <= n and hi - lo <= per       # contiguous, in list order, at most ceil(n / g) each
        assert (lo, hi) == tuple(shard.shard_bounds(n, g, s))        # the multi-process bench cuts the same blocks
        prev = hi
    assert prev == n                                                 # complete


def test_shard_bounds_rejects_bad_arguments(fs):
    for args in [(-1, 2, 0), (10, 0, 0), (10, 2, 2), (10, 2, -1)]:
        with pytest.raises(fs.FsError):
            fs.capi.shard_bounds(*args)


def test_multi_create_fails_loudly_without_a_device(fs):
    """No CPU fallback: on a box without a gfx950 (this container) every ordinal is refused; on the GPU box a bad ordinal is."""
    with pytest.raises(fs.FsError):
        fs.MultiScorer(devices=(10_000,))
    with pytest.raises(fs.FsError):
        fs.MultiScorer(devices=())


def _same_records(got, want):
    """Integer columns bit for bit; the float sums to a few ulp (which lane adds which landmark depends on the order in which
    racing LDS atomics hand out a voxel's ranks, so the last bit of a float sum is not a function of the input)."""
    for k in ("arrival", "argmax", "n_visible", "flags"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_array_equal(got["yaw"], want["yaw"])
    for k in ("info_ref", "trace", "logdet"):
        a, b = got[k].astype(np.float64), want[k].astype(np.float64)
        fin = np.isfinite(b)
        np.testing.assert_array_equal(np.isfinite(a), fin, err_msg=k)
        assert np.all(np.abs(a[fin] - b[fin]) <= 2e-6 * np.maximum(np.abs(b[fin]), 1.0)), k


def _stage(sc, w, kw):
    sc.set_ray_params(**kw)
    sc.upload_grid(w.cells, w.origin, w.resolution)
    sc.upload_landmarks(w.landmarks)
    sc.lookup_generate()
    sc.set_fim_params(14.0, 1.0)
    return sc.max_arrival()


@pytest.mark.gpu
@pytest.mark.parametrize("devices,n_cand", [((0, 0), 5000), ((0, 0, 0), 4999), ((0, 0, 0, 0), 3), ((0,), 257)])
def test_multi_equals_single_context(fs, devices, n_cand):
    w = fs.synth.make_workload("C2", n_cand=n_cand)
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    one = fs.FrontierScorer(device=0)
    mx1 = _stage(one, w, kw)
    want = one.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    one.close()
    m = fs.MultiScorer(devices=devices)
    assert m.n_devices == len(devices)
    mxm = _stage(m, w, kw)
    assert mxm == mx1
    for _ in range(2):                                               # twice: the second call re-uses every member's buffers
        got = m.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        _same_records(got, want)                                    # list order
    # fs_multi_score_arrival: every column, the per-ray counts included, bit for bit
    one = fs.FrontierScorer(device=0)
    _stage(one, w, kw)
    a1 = one.score_arrival(w.goals, w.frontier_size, w.blacklisted)
    one.close()
    am = m.score_arrival(w.goals, w.frontier_size, w.blacklisted, n_rays_total=a1["ray_counts"].shape[1] * a1["ray_counts"].shape[2])
    for k in ("arrival", "argmax", "yaw", "achievable", "status"):
        np.testing.assert_array_equal(am[k], a1[k], err_msg=k)
    np.testing.assert_array_equal(am["ray_counts"], a1["ray_counts"].reshape(am["ray_counts"].shape))
    # a call with fewer candidates than devices, and an empty one
    got = m.score_candidates(w.goals[:1], w.frontier_size[:1], w.blacklisted[:1])
    _same_records(got, want[:1])
    assert m.score_candidates(np.zeros((0, 3))).shape == (0,)
    m.close()


@pytest.mark.gpu
def test_multi_reports_the_failing_member(fs):
    m = fs.MultiScorer(devices=(0, 0))
    with pytest.raises(fs.FsError) as e:
        m.score_candidates(np.zeros((4, 3)))                         # nothing staged yet
    assert e.value.code == fs.capi.FS_E_STATE and "block 0" in str(e.value)
    m.close()
