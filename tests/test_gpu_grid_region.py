"""fs_update_grid_region: a window of the staged map rewritten in place (what a costmap update cycle does to the master grid —
nav2_costmap_2d::Layer::updateCosts(master_grid, min_i, min_j, max_i, max_j); the reference's own layers:
DEP/src/nav2_plugins/lethal_marker.cpp:305-325, fit_slam2_nav2_plugins/plugins/keepout_layer.cpp:279-300).

The claim under test: scoring after a sequence of window updates equals scoring after fs_upload_grid of the whole rewritten map,
bit for bit — per-ray counts included — through both walks (row-major bytes and the 2-bit class image, whose bricks are re-cut
for the window only), in 2-D and 3-D, for windows that are not brick-aligned, touch the map's border or are handed over as a
strided view of the caller's whole map; and both equal the oracle on the rewritten map.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCALE = int(os.environ.get("FS_WINDOW_SCALE", "1"))      # by hand: FS_WINDOW_SCALE=100 multiplies the number of update cycles

FS_E_INVALID, FS_E_STATE = -1, -4


def _ray_kw(w, **over):
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    kw.update(over)
    return kw


def _random_window(rng, shape_zyx, max_side):
    nz, ny, nx = shape_zyx
    sx, sy = int(rng.integers(1, min(nx, max_side) + 1)), int(rng.integers(1, min(ny, max_side) + 1))
    sz = int(rng.integers(1, min(nz, max_side) + 1))
    # a third of the windows hug a border of the map
    x0 = int(rng.integers(0, nx - sx + 1)) if rng.random() > 0.33 else int(rng.choice([0, nx - sx]))
    y0 = int(rng.integers(0, ny - sy + 1)) if rng.random() > 0.33 else int(rng.choice([0, ny - sy]))
    z0 = int(rng.integers(0, nz - sz + 1))
    # what a layer writes: mostly free / lethal / unknown, some inflation values
    vals = rng.choice(np.array([0, 0, 0, 254, 254, 255, 255, 253, 240, 100], dtype=np.uint8), size=(sz, sy, sx))
    return x0, y0, z0, vals


def _equal_arrival(a, b):
    for k in ("status", "arrival", "argmax", "achievable", "yaw", "ray_counts"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("case", ["2d_bytes", "2d_class", "3d_class", "3d_bytes", "2d_odd_shape_class"])
def test_window_updates_equal_a_fresh_snapshot(fs, oracle, case):
    rng = np.random.default_rng({"2d_bytes": 1, "2d_class": 2, "3d_class": 3, "3d_bytes": 4, "2d_odd_shape_class": 5}[case])
    if case.startswith("2d"):
        n = 61 if "odd" in case else 96
        w = fs.synth.make_small_2d(301, n=n, n_cand=48, n_landmarks=50)
        layout = 2 if "class" in case else 1
    else:
        w = fs.synth.make_workload("C1", n_cand=60)                      # 64^3, 32 rays
        layout = 2 if "class" in case else 1
    cells = np.array(w.cells, dtype=np.uint8, copy=True)                 # [nz][ny][nx]
    upd, ref = fs.FrontierScorer(device=0), fs.FrontierScorer(device=0)
    try:
        for s in (upd, ref):
            s.set_option("ray.layout", layout)
            s.set_ray_params(**_ray_kw(w))
        upd.upload_grid(cells, w.origin, w.resolution)
        mx = upd.max_arrival()
        upd.score_arrival(w.goals, w.frontier_size, w.blacklisted)       # (the class image exists before the first window arrives)
        for step in range(12 * SCALE):
            k = int(rng.integers(1, 4))
            for _ in range(k):                                           # a cycle may rewrite several windows, overlapping ones too
                x0, y0, z0, vals = _random_window(rng, cells.shape, 23)
                cells[z0:z0 + vals.shape[0], y0:y0 + vals.shape[1], x0:x0 + vals.shape[2]] = vals
                if step % 2 == 0:
                    upd.update_grid_region(x0, y0, z0, vals)
                else:                                                    # ... straight from the caller's whole map, strides passed on
                    view = cells[z0:z0 + vals.shape[0], y0:y0 + vals.shape[1], x0:x0 + vals.shape[2]]
                    upd.update_grid_region(x0, y0, z0, view, view=True)
            got = upd.score_arrival(w.goals, w.frontier_size, w.blacklisted)
            ref.upload_grid(cells, w.origin, w.resolution)
            ref.set_arrival_limits(mx["max_gt"], mx["min_gt"])
            want = ref.score_arrival(w.goals, w.frontier_size, w.blacklisted)
            _equal_arrival(got, want)
            # the frontier predicate reads the same image
            m_got, c_got = upd.frontier_cells(cells.shape, 160)
            np.testing.assert_array_equal(m_got, oracle.frontier_cell_mask(cells, 160))
        # ... and the oracle on the final map (the limits survived every update: nobody called max_arrival again)
        G = oracle.Grid(cells, origin=w.origin, resolution=w.resolution)
        P = oracle.RayParams(**_ray_kw(w))
        assert oracle.max_arrival_information(G, P) == mx
        wo = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], n_threads=8)
        for kk in ("status", "arrival", "argmax", "achievable", "yaw", "ray_counts"):
            np.testing.assert_array_equal(got[kk], wo[kk], err_msg=kk)
        assert (got["arrival"] > 0).any()
    finally:
        upd.close(); ref.close()


def test_window_arguments(fs):
    """empty windows are no-ops, windows that leave the grid are refused and write nothing, a context without a map says so"""
    w = fs.synth.make_small_2d(302, n=64, n_cand=24, n_landmarks=50)
    s = fs.FrontierScorer(device=0)
    try:
        with pytest.raises(fs.capi.FsError) as e:
            s.update_grid_region(0, 0, 0, np.zeros((4, 4), dtype=np.uint8))
        assert e.value.code == FS_E_STATE
        s.set_ray_params(**_ray_kw(w))
        s.upload_grid(w.cells, w.origin, w.resolution)
        s.max_arrival()
        before = s.score_arrival(w.goals, w.frontier_size, w.blacklisted)
        for x0, y0, z0, shape in ((60, 0, 0, (1, 4, 8)), (0, 61, 0, (1, 4, 4)), (-1, 0, 0, (1, 2, 2)), (0, 0, 1, (1, 2, 2)), (0, 0, 0, (2, 2, 2)),
                                  (2 ** 31 - 4, 0, 0, (1, 1, 8))):
            with pytest.raises(fs.capi.FsError) as e:
                s.update_grid_region(x0, y0, z0, np.full(shape, 254, dtype=np.uint8))
            assert e.value.code == FS_E_INVALID, (x0, y0, z0, shape)
        s.update_grid_region(5, 5, 0, np.zeros((0, 7), dtype=np.uint8))         # empty: fine, nothing happens
        s.update_grid_region(64, 64, 0, np.zeros((0, 0), dtype=np.uint8))
        _equal_arrival(s.score_arrival(w.goals, w.frontier_size, w.blacklisted), before)
        # the whole map as one window == a snapshot
        other = np.ascontiguousarray(np.rot90(w.cells[0]))
        s.update_grid_region(0, 0, 0, other)
        t = fs.FrontierScorer(device=0)
        try:
            t.set_ray_params(**_ray_kw(w))
            t.upload_grid(other, w.origin, w.resolution)
            t.max_arrival()
            _equal_arrival(s.score_arrival(w.goals, w.frontier_size, w.blacklisted), t.score_arrival(w.goals, w.frontier_size, w.blacklisted))
        finally:
            t.close()
    finally:
        s.close()


def test_window_update_on_every_member_of_a_multi_device_scorer(fs):
    """fs_multi_update_grid_region: the window reaches every member (two contexts of the one GPU here); records equal one context's"""
    w = fs.synth.make_small_2d(303, n=96, n_cand=50, n_landmarks=400)
    cells = np.array(w.cells, dtype=np.uint8, copy=True).reshape(96, 96)
    one, multi = fs.FrontierScorer(device=0), fs.MultiScorer([0, 0])
    try:
        for s in (one, multi):
            s.set_ray_params(**_ray_kw(w))
            s.upload_grid(cells, w.origin, w.resolution)
            s.upload_landmarks(w.landmarks)
            s.lookup_generate()
            s.set_fim_params(14.0, 1.0)
            s.max_arrival()
        rng = np.random.default_rng(5)
        for _ in range(4):
            x0, y0, z0, vals = _random_window(rng, (1, 96, 96), 40)
            cells[y0:y0 + vals.shape[1], x0:x0 + vals.shape[2]] = vals[0]
            one.update_grid_region(x0, y0, 0, vals)
            multi.update_grid_region(x0, y0, 0, cells[y0:y0 + vals.shape[1], x0:x0 + vals.shape[2]], view=True)
            a = one.score_candidates(w.goals, w.frontier_size, w.blacklisted)
            b = multi.score_candidates(w.goals, w.frontier_size, w.blacklisted)
            for k in ("arrival", "argmax", "yaw", "n_visible", "flags"):
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    finally:
        one.close(); multi.close()


@pytest.mark.parametrize("name,seed", [("REF2D", 11), ("C2", 12)])
def test_map_windows_between_random_calls_equal_new_snapshots(fs, name, seed):
    """Random call sequences (every entry point, list lengths from 1 to 400, split settings, the optional routes of the small calls
    — captured launch graphs among them: a window update is a context epoch) with random map windows in between: the context that
    receives the windows must answer like one that takes a whole new snapshot each time."""
    from test_gpu_lifecycle import KINDS, _call, _compare, _make
    w = fs.synth.make_workload(name, n_cand=400)
    rng = np.random.default_rng(seed)
    cells = np.array(w.cells, dtype=np.uint8, copy=True)
    snap, limits = _make(fs, w)
    win, _ = _make(fs, w, limits)
    try:
        first = snap.score_arrival(w.goals, w.frontier_size, w.blacklisted)
        poses = fs.synth.poses_from_yaw(w.goals, first["yaw"])
        for step in range(40 * SCALE):
            if step % 2 == 0:
                for _ in range(int(rng.integers(1, 3))):
                    x0, y0, z0, vals = _random_window(rng, cells.shape, 48)
                    cells[z0:z0 + vals.shape[0], y0:y0 + vals.shape[1], x0:x0 + vals.shape[2]] = vals
                    win.update_grid_region(x0, y0, z0, cells[z0:z0 + vals.shape[0], y0:y0 + vals.shape[1], x0:x0 + vals.shape[2]], view=True)
                snap.upload_grid(cells, w.origin, w.resolution)
                snap.set_arrival_limits(limits["max_gt"], limits["min_gt"])
            kind = str(rng.choice(KINDS))
            n = int(rng.choice([1, 3, 9, 17, 50, 64, 200, 400]))
            pick = rng.choice(w.goals.shape[0], size=n, replace=False)
            split = int(rng.choice([0, 3, 4]))
            angle = float(rng.choice([1.0, 4.0]))
            opts = {k: int(rng.integers(0, 2)) for k in ("graph", "zerocopy", "fim.hostfinish")}
            for k, v in opts.items():
                win.set_option(k, v)
            if step % 5 == 0:
                win.set_option("ray.layout", int(rng.choice([0, 1, 2])))
            what = f"{name} seed {seed} step {step}: {kind} n={n} split={split} angle={angle} {opts}"
            _compare(fs, kind, _call(fs, win, kind, w, pick, poses, split, angle), _call(fs, snap, kind, w, pick, poses, split, angle), what)
    finally:
        snap.close(); win.close()
