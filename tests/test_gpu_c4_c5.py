"""BASELINE.json configs[3] and configs[4] on ONE GPU, through the C ABI.

C4 (512^3 grid, 160 k candidates, 100 k landmarks — sharded over 8 GPUs in the scaling run): the eight rank blocks
`shard_bounds(160000, 8, r)` scored one after the other and concatenated must equal the list scored whole — that is
the whole correctness content of the sharded run besides the all-gather (tests/test_shard_gloo.py, test_gpu_shard.py).
C5 (1024^3 map, 50 k candidates, 500 k landmarks): the map goes in through the sparse brick list of
fs_upload_grid_bricks (the wire format north_star names; dense 1 GiB + bricked 1 GiB in HBM), all 50 k candidates are
scored, and the size-independent properties of test_gpu_fullsize.py are checked.  Both get an oracle spot check."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _int_view(rec):
    return np.stack([rec["arrival"], rec["argmax"], rec["n_visible"], rec["flags"].astype(np.int64)], axis=1)


def _scorer(fs, w, bricks=None):
    s = fs.FrontierScorer(device=0)
    s.lookup_generate()
    s.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                     robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    if bricks is None:
        s.upload_grid(w.cells, w.origin, w.resolution)
    else:
        s.upload_grid_bricks(w.cells.shape, w.origin, w.resolution, bricks[0], bricks[1], default_value=255)
    s.upload_landmarks(w.landmarks)
    s.set_fim_params(14.0, 1.0)
    s.max_arrival()
    return s


def _oracle_spot_check(fs, oracle, ref_table, w, rec, idx):
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    mx = oracle.max_arrival_information(G, P)
    arr = oracle.arrival_information(G, P, w.goals[idx], w.frontier_size[idx], w.blacklisted[idx], min_gt=mx["min_gt"], n_threads=8)
    fim = oracle.pose_information(ref_table, w.landmarks, oracle.poses_from_yaw(w.goals[idx], arr["yaw"]), 14.0, 1.0, n_threads=8)
    r = rec[idx]
    ok = arr["status"] == 0
    np.testing.assert_array_equal(r["arrival"], arr["arrival"])
    np.testing.assert_array_equal(r["argmax"], arr["argmax"])
    np.testing.assert_array_equal(fs.capi.record_status(r), arr["status"])
    np.testing.assert_array_equal(fs.capi.record_achievable(r), arr["achievable"])
    np.testing.assert_array_equal(r["n_visible"][ok], fim["n_visible"][ok])
    np.testing.assert_array_equal(fs.capi.record_nvoxels(r)[ok], np.minimum(fim["n_voxels"][ok], 65535))
    sc = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
    assert np.max(np.abs(r["info_ref"][ok] - fim["info_f64"][ok]) / sc) <= 1e-4          # north_star tolerance
    tr = np.maximum(np.abs(fim["trace"][ok]), 1e-6)
    assert np.max(np.abs(r["trace"][ok] - fim["trace"][ok]) / tr) <= 1e-4
    return mx


def test_c4_blockwise_equals_whole_and_oracle_sample(fs, oracle, ref_table):
    shard = importlib.import_module("fit-slam_amd.shard")
    w = fs.synth.make_workload("C4")
    assert w.goals.shape[0] == 160_000 and w.cells.shape == (512, 512, 512) and w.landmarks.shape[0] == 100_000
    s = _scorer(fs, w)
    try:
        whole = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)                       # one list of 160 k
        assert whole.shape[0] == 160_000
        status = fs.capi.record_status(whole)
        assert (status == 2).sum() == int(w.blacklisted.sum()) and (status == 0).sum() > 150_000
        parts = []
        for r in range(8):                                                                         # what rank r of 8 scores
            lo, hi = shard.shard_bounds(160_000, 8, r)
            assert hi - lo == 20_000
            parts.append(s.score_candidates(w.goals[lo:hi], w.frontier_size[lo:hi], w.blacklisted[lo:hi]))
        joined = np.concatenate(parts)
        np.testing.assert_array_equal(_int_view(joined), _int_view(whole))
        np.testing.assert_array_equal(joined["yaw"], whole["yaw"])
        np.testing.assert_allclose(joined["info_ref"], whole["info_ref"], rtol=5e-6, atol=1e-6)    # same terms, other summation order
        np.testing.assert_allclose(joined["trace"], whole["trace"], rtol=5e-6, atol=1e-6)
        idx = np.sort(np.random.default_rng(4).choice(160_000, size=320, replace=False))
        _oracle_spot_check(fs, oracle, ref_table, w, whole, idx)
    finally:
        s.close()


def test_c5_brick_upload_full_list_properties_and_oracle_sample(fs, oracle, ref_table):
    w = fs.synth.make_workload("C5")
    assert w.cells.shape == (1024, 1024, 1024) and w.goals.shape[0] == 50_000 and w.landmarks.shape[0] == 500_000
    bxyz, bcells = fs.synth.dense_to_bricks(w.cells)
    assert 0 < bxyz.shape[0] < (1024 // 8) ** 3 and bcells.shape == (bxyz.shape[0], 512)         # genuinely sparse
    s = _scorer(fs, w, bricks=(bxyz, bcells))
    try:
        # the frontier-cell stencil sees the expanded grid: the staged map equals the dense one cell for cell
        _, n_frontier = s.frontier_cells(w.cells.shape, want_mask=False)
        base = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)                        # all 50 k candidates
        assert base.shape[0] == 50_000
        status = fs.capi.record_status(base)
        assert (status == 2).sum() == int(w.blacklisted.sum()) and (status == 0).sum() > 47_000
        assert base["n_visible"].max() > 5_000                                                    # crowded poses exist (several scoring passes)
        # culling off = brute force over all 500 k landmarks; sorting off; both walks of the grid: identical integers
        for key, val, back in (("fim.cull", 0, 1), ("ray.sort", 0, 1), ("ray.layout", 2, 0)):
            s.set_option(key, val)
            other = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)
            s.set_option(key, back)
            np.testing.assert_array_equal(_int_view(other), _int_view(base), err_msg=key)
            np.testing.assert_array_equal(other["yaw"], base["yaw"])
            np.testing.assert_allclose(other["info_ref"], base["info_ref"], rtol=5e-6, atol=1e-6)
        # block-wise == whole (the 8-GPU run replicates the grid and gives each rank 6 250 candidates)
        cuts = [0, 6_250, 12_500, 31_250, 50_000]
        joined = np.concatenate([s.score_candidates(w.goals[a:b], w.frontier_size[a:b], w.blacklisted[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        np.testing.assert_array_equal(_int_view(joined), _int_view(base))
        np.testing.assert_allclose(joined["info_ref"], base["info_ref"], rtol=5e-6, atol=1e-6)
        # the dense upload of the same map gives the same records and the same stencil count
        s.upload_grid(w.cells, w.origin, w.resolution)
        s.max_arrival()
        _, n_frontier_dense = s.frontier_cells(w.cells.shape, want_mask=False)
        assert n_frontier_dense == n_frontier
        sub = np.sort(np.random.default_rng(6).choice(50_000, size=5_000, replace=False))
        dense = s.score_candidates(w.goals[sub], w.frontier_size[sub], w.blacklisted[sub])
        np.testing.assert_array_equal(_int_view(dense), _int_view(base[sub]))
        idx = np.sort(np.random.default_rng(5).choice(50_000, size=320, replace=False))
        _oracle_spot_check(fs, oracle, ref_table, w, base, idx)
        assert s.get_counter(4) > 0 and s.get_counter(6) == 0                                     # crowded poses took several passes; nothing unresolved
    finally:
        s.close()


def test_brick_upload_rejects_what_it_cannot_stage(fs):
    s = fs.FrontierScorer(device=0)
    try:
        with pytest.raises(fs.FsError):                       # dimensions must be multiples of 8
            s.upload_grid_bricks((12, 16, 16), (0, 0, 0), 0.05, np.zeros((0, 3), np.int32), np.zeros((0, 512), np.uint8))
        with pytest.raises(fs.FsError):                       # a brick outside the grid
            s.upload_grid_bricks((16, 16, 16), (0, 0, 0), 0.05, np.array([[2, 0, 0]], np.int32), np.zeros((1, 512), np.uint8))
        with pytest.raises(fs.FsError):                       # dense staging is limited to 2^32 cells (documented hard limit)
            s.upload_grid_bricks((2048, 2048, 1024), (0, 0, 0), 0.05, np.zeros((0, 3), np.int32), np.zeros((0, 512), np.uint8))
    finally:
        s.close()


def test_grid_beyond_2_31_cells_walks_like_the_same_cells_in_a_small_grid(fs):
    """A 1024 x 1024 x 2560 grid (2.7 G cells; cell offsets past 2^31 — a shape whose brick strides still fit the class walk's 24-bit multiplies) that is unknown everywhere except for C1's 64^3 cells in its
    far corner: fans inside that corner, clamped to it by the polygon, must return exactly what the same cells return as a grid of
    their own (translation invariance; resolution 1/16 m so that every world <-> cell conversion is exact in both frames).  Both
    walks: the byte image (32-bit unsigned offsets up to 2^32) and the class image (brick addresses)."""
    w = fs.synth.make_workload("C1")
    res = 0.0625
    n = 64
    sub = np.ascontiguousarray(w.cells)                                 # [64][64][64]
    NX, NY, NZ = 1024, 1024, 2560
    x0, y0, z0 = NX - n - 8, NY - n - 16, NZ - n - 8                   # far corner: offsets > 2^31 (z0 * NX * NY = 2.6e9)
    big_origin = (-8.0, -8.0, -64.0)
    small_origin = (big_origin[0] + x0 * res, big_origin[1] + y0 * res, big_origin[2] + z0 * res)
    cell = ((w.goals - np.asarray(w.origin)) / w.resolution - 0.5).round().astype(np.int64)          # C1's goal cells
    goals = (cell + 0.5) * res + np.asarray(small_origin)
    poly = (small_origin[0], small_origin[1], small_origin[0] + (n - 1 + 0.5) * res, small_origin[1] + (n - 1 + 0.5) * res)
    kw = dict(max_camera_depth=40 * res, delta_theta=w.delta_theta, camera_fov=w.camera_fov, robot_radius=0.6 * res / 0.05,
              n_rays=w.n_yaw, elev=(0.0,), polygon=poly)
    s = fs.FrontierScorer(device=0)
    try:
        s.set_ray_params(**kw)
        s.upload_grid(sub, small_origin, res)
        s.set_arrival_limits(400.0, 40.0)
        want = {lay: None for lay in (1, 2)}
        for lay in (1, 2):
            s.set_option("ray.layout", lay)
            want[lay] = s.score_arrival(goals, w.frontier_size, w.blacklisted)
        big = np.full((NZ, NY, NX), 255, dtype=np.uint8)
        big[z0:z0 + n, y0:y0 + n, x0:x0 + n] = sub
        # the footprint disc may look past the corner block in the big grid: keep what it would see there equal to "off the map"
        # (never lethal) — unknown cells are not 254
        s.upload_grid(big, big_origin, res)
        del big
        s.set_arrival_limits(400.0, 40.0)
        assert (want[1]["arrival"] > 0).sum() > 50
        for lay in (1, 2):
            s.set_option("ray.layout", lay)
            got = s.score_arrival(goals, w.frontier_size, w.blacklisted)
            for k in ("status", "arrival", "argmax", "achievable", "ray_counts"):
                np.testing.assert_array_equal(got[k], want[lay][k], err_msg=f"layout {lay}: {k}")
            np.testing.assert_array_equal(got["yaw"], want[lay]["yaw"])
        np.testing.assert_array_equal(want[1]["ray_counts"], want[2]["ray_counts"])
        # ... and a costmap cycle's WINDOW out there (fs_update_grid_region: cell offsets past 2^31 in the scatter, brick addresses at
        # the far end of the class image in the partial re-cut): the corner block blanked to unknown, then written back in two
        # unaligned windows — the class image exists (layout 2 has just walked it), so it is the partial re-cut that runs
        s.update_grid_region(x0, y0, z0, np.full((n, n, n), 255, dtype=np.uint8))
        blank = s.score_arrival(goals, w.frontier_size, w.blacklisted)
        assert not np.array_equal(blank["ray_counts"], want[2]["ray_counts"]) and (blank["ray_counts"] >= want[2]["ray_counts"]).all()
        s.update_grid_region(x0, y0, z0, sub[:, :, :37])
        s.update_grid_region(x0 + 37, y0, z0, sub[:, :, 37:], view=True)
        for lay in (2, 1):
            s.set_option("ray.layout", lay)
            got = s.score_arrival(goals, w.frontier_size, w.blacklisted)
            for k in ("status", "arrival", "argmax", "achievable", "ray_counts", "yaw"):
                np.testing.assert_array_equal(got[k], want[lay][k], err_msg=f"after the windows, layout {lay}: {k}")
    finally:
        s.close()
