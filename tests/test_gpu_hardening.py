"""Round-2 hardening checks: the host-side table generator against the oracle's record list
byte for byte, and the per-tick entry points on context-owned scratch (growing, shrinking and interleaved calls)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_generated_lookup_records_equal_the_oracle_byte_for_byte(fs, oracle, ref_table):
    """generateLookupTable (FIP/src/fisher_information/FisherInfoManager.cpp:117-229): same record sequence, same float32
    bits — key order, de-duplication, the NaN skip at the origin and the trailing (0,0,0) <- max record included."""
    s = fs.FrontierScorer(device=0)
    try:
        s.lookup_generate()                                               # gen_fi_lookup's bounds
        got = s.lookup_records()
        want = ref_table.records
        assert got.shape == want.shape == (710_000, 4)
        assert got.tobytes() == want.tobytes()
        for bounds in ((0.0, 3.0, -2.0, 2.0, -1.0, 1.0), (-1.2, 2.05, -0.31, 0.29, 0.0, 0.9), (0.0, 21.0, -14.7, 14.7, -14.7, 14.7)):
            s.lookup_generate(bounds)
            t = oracle.Table.generate(bounds)
            assert s.lookup_records().tobytes() == t.records.tobytes(), bounds
        # set_records / get_records round trip keeps the bytes (what fs_lookup_save writes)
        s.lookup_set_records(want[::-1].copy())
        assert s.lookup_records().tobytes() == want[::-1].tobytes()
    finally:
        s.close()


def test_per_tick_entry_points_reuse_context_scratch(fs, oracle):
    """fs_trace_segments / fs_frontier_cells / fs_information_frontier_pair keep their device buffers in the context:
    calls of growing, shrinking and interleaved sizes must all stay correct."""
    rng = np.random.default_rng(91)
    w = fs.synth.make_small_2d(61, n=160, n_cand=8, n_landmarks=700)
    s = fs.FrontierScorer(device=0)
    try:
        s.upload_grid(w.cells, w.origin, w.resolution)
        s.upload_landmarks(w.landmarks)
        G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
        lo = np.array(w.origin); hi = lo + np.array([160, 160, 1]) * w.resolution
        want_mask = oracle.frontier_cell_mask(w.cells, 160)
        for n in (5, 900, 37, 2_000, 1):
            a = rng.uniform(lo, hi, size=(n, 3)); b = a + rng.normal(scale=1.0, size=(n, 3))
            a[:, 2] = 0.0; b[:, 2] = 0.0
            got = s.trace_segments(a, b, 122, obst=(253, 254), trace=(0, 255))
            for i in range(0, n, max(1, n // 40)):
                r = oracle.trace_ray(G, a[i], b[i], 122, obst=(253, 254), trace=(0, 255), faithful=True)
                assert bool(got["ok"][i]) == r["ok"]
                if r["ok"]:
                    assert (got["traced"][i], bool(got["hit"][i]), got["unknown"][i], got["all"][i]) == (r["traced"], r["hit"], r["unknown"], r["all"])
            mask, count = s.frontier_cells(w.cells.shape, 160, want_mask=(n % 2 == 1))
            assert count == int(want_mask.sum())
            if mask is not None:
                np.testing.assert_array_equal(mask, want_mask)
            k = max(1, n // 50)
            poses = np.zeros((k, 7)); poses[:, 6] = 1.0
            poses[:, :2] = rng.uniform(-2, 2, size=(k, 2))
            tri = np.tile(np.array([[-3.0, -3.0, 3.0, -3.0, 0.0, 3.0]]), (k, 1)) + rng.normal(scale=0.2, size=(k, 6))
            info = s.information_frontier_pair(poses, tri)
            for i in range(k):
                want = oracle.information_frontier_pair(w.landmarks, poses[i], tri[i].reshape(3, 2))
                assert abs(info[i] - want) <= 1e-4 * max(1.0, abs(want))
    finally:
        s.close()


def _same_records(a, b):
    """integer fields bit for bit; the float32 sums to 1e-6 relative (which lane of a voxel's landmarks draws which rank is
    decided by a race between waves, so the order of a few float additions is not reproducible from run to run)"""
    for f in ("arrival", "argmax", "n_visible", "flags"):
        if not np.array_equal(a[f], b[f]):
            return False
    if a["yaw"].tobytes() != b["yaw"].tobytes():
        return False
    for f in ("info_ref", "trace", "logdet"):
        x, y = a[f].astype(np.float64), b[f].astype(np.float64)
        fin = np.isfinite(x)
        if not np.array_equal(fin, np.isfinite(y)) or not np.array_equal(x[~fin], y[~fin]):
            return False
        if not np.allclose(x[fin], y[fin], rtol=1e-6, atol=0.0):
            return False
    return True


def test_processing_order_never_changes_a_record(fs):
    """The spatial sort keeps a cost map between calls (fs_sort.hip: blocks that held expensive candidates go first) and
    can be told to walk the blocks in reverse; the order the kernels visit the candidates in must not change a record —
    across a cold map, a warm map, the map switched off, the reversed order and the sort switched off."""
    w = fs.synth.make_workload("C2", n_cand=6000)
    s = fs.FrontierScorer(device=0)
    try:
        s.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                         robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
        s.upload_grid(w.cells, w.origin, w.resolution)
        s.upload_landmarks(w.landmarks)
        s.lookup_generate()
        s.set_fim_params(14.0, 1.0)
        s.max_arrival()
        ref = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)           # cold map: plain Morton order
        assert int(ref["n_visible"].max()) > 0
        for _ in range(3):                                                             # the map warms up over the next calls
            assert _same_records(s.score_candidates(w.goals, w.frontier_size, w.blacklisted), ref)
        for key, value in (("sort.reverse", 1), ("sort.costmap", 0), ("sort.reverse", 0), ("ray.sort", 0), ("ray.sort", 1), ("sort.costmap", 1)):
            s.set_option(key, value)
            for _ in range(2):
                assert _same_records(s.score_candidates(w.goals, w.frontier_size, w.blacklisted), ref), (key, value)
        # a different list on the warm map of the old one, then the old list again
        sub = np.arange(0, 6000, 2)
        part = s.score_candidates(w.goals[sub], w.frontier_size[sub], w.blacklisted[sub])
        assert _same_records(part, ref[sub])
        assert _same_records(s.score_candidates(w.goals, w.frontier_size, w.blacklisted), ref)
    finally:
        s.close()


def test_landmark_cloud_at_the_documented_maximum(fs, oracle, ref_table):
    """The largest cloud a context takes is 2 000 000 landmarks (fs_upload_landmarks; four times configs[4]'s 500 k): at exactly
    the maximum every column of fs_score_fim and the info-only call must still be the oracle's — a clumped cloud, so that chunks
    of very different sizes, multi-pass poses and the HBM tier all occur — and one landmark more is refused with FS_E_INVALID,
    the staged cloud left as it was."""
    rng = np.random.default_rng(2_000_000)
    m = 2_000_000
    lm = rng.uniform(-40.0, 40.0, size=(m, 3)).astype(np.float32)
    lm[:, 2] = rng.uniform(-2.0, 3.0, size=m)
    lm[:300_000] = (np.array([6.0, 1.0, 0.5]) + rng.normal(scale=1.2, size=(300_000, 3))).astype(np.float32)      # a dense clump
    poses = np.zeros((10, 7)); poses[:, 6] = 1.0
    poses[:, 0] = np.linspace(-30.0, 30.0, 10); poses[:, 1] = np.linspace(-5.0, 5.0, 10)
    poses[4, :3] = (2.0, 1.0, 0.5)                                       # looking into the clump
    q = rng.normal(size=(3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[7:, 3:] = q                                                    # three general orientations
    s = fs.FrontierScorer(device=0)
    try:
        s.lookup_generate()
        s.upload_landmarks(lm)
        parity = __import__("importlib").import_module("fit-slam_amd.parity")
        for angle in (1.0, 4.0):
            s.set_fim_params(14.0, angle)
            want = oracle.pose_information(ref_table, lm, poses, 14.0, angle, n_threads=16)
            assert want["n_visible"].max() > 100_000 and want["n_voxels"].max() > 16384
            got = s.score_fim(poses)
            np.testing.assert_array_equal(got["n_visible"], want["n_visible"])
            np.testing.assert_array_equal(got["n_voxels"], want["n_voxels"])
            sc = np.maximum(np.abs(want["info_f64"]), 1e-6)
            assert np.max(np.abs(got["info_ref"] - want["info_f64"]) / sc) <= 1e-4
            assert np.max(np.abs(got["trace"] - want["trace"]) / np.maximum(np.abs(want["trace"]), 1e-6)) <= 1e-4
            gate = parity.logdet_gate(got["logdet"], want["logdet"], want["fim"], n_visible=want["n_visible"])
            assert gate["ok"], gate
            only = s.score_fim(poses, info_only=True)
            np.testing.assert_array_equal(only["n_voxels"], want["n_voxels"])
            assert np.max(np.abs(only["info_ref"] - want["info_f64"]) / sc) <= 1e-4
        with pytest.raises(fs.FsError):
            s.upload_landmarks(np.concatenate([lm, lm[:1]]))
        again = s.score_fim(poses[:2], info_only=True)                   # the refused upload left the staged cloud alone
        np.testing.assert_array_equal(again["n_voxels"], want["n_voxels"][:2])
    finally:
        s.close()


def test_ray_fan_at_the_documented_maxima(fs, oracle):
    """4096 yaw rays (fs_set_ray_params' limit) x 16 elevation rings (FS_MAX_ELEV) = 65 536 rays per candidate on a 3-D grid:
    per-ray counts, window maximum, first argmax, yaw and achievability against the oracle; 4097 rays and 17 rings are refused."""
    w = fs.synth.make_workload("C1", n_cand=24)
    n_yaw = 4096
    elev = tuple(np.linspace(-0.45, 0.45, 16).tolist())
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=2 * np.pi / n_yaw, camera_fov=w.camera_fov, robot_radius=w.robot_radius,
              n_rays=n_yaw, elev=elev, polygon=w.polygon)
    s = fs.FrontierScorer(device=0)
    try:
        s.set_ray_params(**kw)
        assert (s.n_yaw, s.n_elev, s.window) == (4096, 16, int(w.camera_fov / (2 * np.pi / n_yaw)))
        s.upload_grid(w.cells, w.origin, w.resolution)
        G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
        P = oracle.RayParams(**kw)
        mx = oracle.max_arrival_information(G, P)
        assert s.max_arrival() == mx
        want = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], n_threads=16)
        got = s.score_arrival(w.goals, w.frontier_size, w.blacklisted)
        for k in ("status", "arrival", "argmax", "achievable", "ray_counts"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        np.testing.assert_array_equal(got["yaw"], want["yaw"])
        assert want["arrival"].max() > 10_000 and got["ray_counts"].shape == (24, 16, 4096)
        with pytest.raises(fs.FsError):
            s.set_ray_params(**dict(kw, n_rays=4097, delta_theta=2 * np.pi / 4097))
        with pytest.raises(fs.FsError):
            s.set_ray_params(**dict(kw, elev=tuple(np.linspace(-0.45, 0.45, 17).tolist())))
    finally:
        s.close()


def test_lookup_table_at_the_documented_maximum(fs, oracle):
    """The dense lattice table holds up to 2^21 - 2 cells (the voxel key shares a 32-bit slot word with an 11-bit count): a table of
    127 x 128 x 128 = 2 080 768 cells — three times the reference's 71 x 100 x 100 — generated on both sides byte for byte, then
    scored; a lattice over the limit is refused."""
    step = 0.3
    bounds = (0.0, 125 * step + 0.01, -63 * step + 0.01, 63 * step + 0.01, -63 * step + 0.01, 63 * step + 0.01)
    t = oracle.Table.generate(bounds)
    assert t.records.shape[0] == 127 * 128 * 128
    rng = np.random.default_rng(21)
    lm = rng.uniform(-30.0, 30.0, size=(150_000, 3)).astype(np.float32)
    poses = np.zeros((6, 7)); poses[:, 6] = 1.0
    poses[:, 0] = np.linspace(-10.0, 10.0, 6)
    s = fs.FrontierScorer(device=0)
    try:
        s.lookup_generate(bounds)
        assert s.lookup_records().tobytes() == t.records.tobytes()
        s.upload_landmarks(lm)
        s.set_fim_params(40.0, 4.0)                                       # a range that reaches the far end of the table
        want = oracle.pose_information(t, lm, poses, 40.0, 4.0, n_threads=16)
        got = s.score_fim(poses)
        np.testing.assert_array_equal(got["n_voxels"], want["n_voxels"])
        np.testing.assert_array_equal(got["n_visible"], want["n_visible"])
        sc = np.maximum(np.abs(want["info_f64"]), 1e-6)
        assert np.max(np.abs(got["info_ref"] - want["info_f64"]) / sc) <= 1e-4
        assert want["n_voxels"].max() > 16384
        with pytest.raises(fs.FsError):
            s.lookup_generate((0.0, 127 * step + 0.01, -64 * step + 0.01, 63 * step + 0.01, -63 * step + 0.01, 63 * step + 0.01))     # 129 x 129 x 128 = 2 130 048 > 2^21 - 2
    finally:
        s.close()


@pytest.mark.parametrize("angle", [1.0, 4.0])
def test_non_finite_and_far_away_landmarks_are_never_visible(fs, oracle, ref_table, angle):
    """A cloud with NaN, +-inf and 1e30 coordinates sprinkled over it (a SLAM front end that has not culled its map): such a point is
    outside every visibility volume — the predicate n^2 <= max_dist^2 is false for it — so every column must equal the oracle's on
    the same cloud AND the columns of the cloud without those points; the k-d ordering (a comparator over coordinates) must not
    see them (fs_stage_landmarks puts them behind the finite ones)."""
    rng = np.random.default_rng(77)
    clean = rng.uniform(-8.0, 8.0, size=(6000, 3)).astype(np.float32)
    dirty = np.repeat(clean, 1, axis=0)
    bad = np.array([[np.nan, 0, 0], [0, np.nan, 1], [1, 2, np.nan], [np.inf, 0, 0], [0, -np.inf, 0], [np.nan, np.nan, np.nan],
                    [1e30, 0, 0], [0, -1e30, 1e30], [3.0e38, 3.0e38, 3.0e38]], dtype=np.float32)
    where = rng.choice(6000, size=400, replace=False)
    dirty = np.insert(dirty, where, bad[rng.integers(0, len(bad), size=400)], axis=0)
    poses = np.zeros((12, 7)); poses[:, 6] = 1.0
    poses[:, :2] = rng.uniform(-5.0, 5.0, size=(12, 2))
    q = rng.normal(size=(4, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[8:, 3:] = q
    s = fs.FrontierScorer(device=0)
    try:
        s.lookup_generate()
        s.set_fim_params(14.0, angle)
        s.upload_landmarks(clean)
        base = s.score_fim(poses)
        s.upload_landmarks(dirty)
        got = s.score_fim(poses)
        only = s.score_fim(poses, info_only=True)
        with np.errstate(all="ignore"):
            want = oracle.pose_information(ref_table, dirty, poses, 14.0, angle, n_threads=8)
        for k in ("n_visible", "n_voxels"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
            np.testing.assert_array_equal(got[k], base[k], err_msg=k)
        np.testing.assert_array_equal(only["n_voxels"], want["n_voxels"])
        sc = np.maximum(np.abs(want["info_f64"]), 1e-6)
        assert np.all(np.isfinite(got["info_ref"])) and np.all(np.isfinite(got["trace"])) and np.all(np.isfinite(got["fim21"]))
        assert np.max(np.abs(got["info_ref"] - want["info_f64"]) / sc) <= 1e-4
        assert np.max(np.abs(only["info_ref"] - want["info_f64"]) / sc) <= 1e-4
        assert np.max(np.abs(got["info_ref"] - base["info_ref"]) / sc) <= 1e-5
        assert want["n_visible"].min() > 100
        # a cloud with no usable point at all
        s.upload_landmarks(np.full((1000, 3), np.nan, dtype=np.float32))
        for r in (s.score_fim(poses), s.score_fim(poses, info_only=True)):
            assert not r["n_voxels"].any() and not r["info_ref"].any()
        s.upload_landmarks(clean)
        back = s.score_fim(poses)
        np.testing.assert_array_equal(back["n_visible"], base["n_visible"])
    finally:
        s.close()


@pytest.mark.parametrize("angle", [1.0, 4.0])
def test_unusable_poses_spoil_only_themselves(fs, oracle, ref_table, angle):
    """Poses nobody can stand at — NaN or infinite position, a position 1e30 m out, a NaN quaternion (and an all-zero one, which Eigen and the oracle read as the identity) — mixed into a batch
    of sound ones, through every column, the info-only call and a short (split) list: the call returns, nothing is read or
    written out of range (tests run under the FS_BOUNDS build too), the sound poses get exactly the results they get without the
    others, and an unusable pose sees nothing (no landmark is within range of it) or NaN — never a count out of thin air."""
    rng = np.random.default_rng(5)
    lm = rng.uniform(-8.0, 8.0, size=(20000, 3)).astype(np.float32)
    sound = np.zeros((6, 7)); sound[:, 6] = 1.0
    sound[:, :2] = rng.uniform(-4.0, 4.0, size=(6, 2))
    q = rng.normal(size=(3, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    sound[3:, 3:] = q
    bad = np.zeros((7, 7)); bad[:, 6] = 1.0
    bad[0, 0] = np.nan; bad[1, 1] = np.inf; bad[2, 2] = -np.inf; bad[3, :3] = 1e30
    bad[4, 3:] = np.nan; bad[5, 3:] = 0.0; bad[6, :3] = (3e38, -3e38, 3e38)
    mixed = np.concatenate([sound[:3], bad, sound[3:]])
    where_sound = np.r_[0:3, 10:13]
    s = fs.FrontierScorer(device=0)
    try:
        s.lookup_generate()
        s.upload_landmarks(lm)
        s.set_fim_params(14.0, angle)
        want = oracle.pose_information(ref_table, lm, sound, 14.0, angle, n_threads=8)
        sc = np.maximum(np.abs(want["info_f64"]), 1e-6)
        for batch, idx in ((mixed, where_sound), (mixed[1:5], np.r_[0:2]), (mixed[2:4], np.r_[0:1])):
            base = sound[:3][-len(idx):] if len(idx) < 3 else sound
            exp = {k: (v[:3][-len(idx):] if len(idx) < 3 else v) for k, v in want.items()}
            got = s.score_fim(batch)
            only = s.score_fim(batch, info_only=True)
            np.testing.assert_array_equal(got["n_visible"][idx], exp["n_visible"])
            np.testing.assert_array_equal(got["n_voxels"][idx], exp["n_voxels"])
            np.testing.assert_array_equal(only["n_voxels"][idx], exp["n_voxels"])
            e = np.maximum(np.abs(exp["info_f64"]), 1e-6)
            assert np.max(np.abs(got["info_ref"][idx] - exp["info_f64"]) / e) <= 1e-4
            assert np.max(np.abs(only["info_ref"][idx] - exp["info_f64"]) / e) <= 1e-4
            assert np.max(np.abs(got["trace"][idx] - exp["trace"]) / np.maximum(np.abs(exp["trace"]), 1e-6)) <= 1e-4
            others = np.setdiff1d(np.arange(len(batch)), idx)
            # (an all-zero quaternion is a rotation to Eigen — the identity, FisherInformationHelpers.cpp:20-24 — and to the oracle: it
            # is a sound pose at the origin, checked as such)
            zero_q = np.array([k for k in others if not np.any(batch[k, 3:]) and np.all(np.isfinite(batch[k, :3]))], dtype=int)
            if len(zero_q):
                z = oracle.pose_information(ref_table, lm, batch[zero_q], 14.0, angle, n_threads=2)
                np.testing.assert_array_equal(got["n_visible"][zero_q], z["n_visible"])
                assert np.max(np.abs(got["info_ref"][zero_q] - z["info_f64"]) / np.maximum(np.abs(z["info_f64"]), 1e-6)) <= 1e-4
            rest = np.setdiff1d(others, zero_q)
            assert np.all((got["n_visible"][rest] == 0) | ~np.isfinite(got["info_ref"][rest])), got["n_visible"][rest]
            assert np.all((only["info_ref"][rest] == 0) | ~np.isfinite(only["info_ref"][rest])), only["info_ref"][rest]
    finally:
        s.close()


def test_non_finite_parameters_are_refused(fs):
    """NaN / infinite set-up values would turn worldToMap into a float-to-integer conversion of NaN (undefined in the reference) or a
    ray fan of infinite length: refused with FS_E_INVALID, the context keeps what it had and goes on working."""
    w = fs.synth.make_small_2d(3, n=96, n_cand=16, n_landmarks=300)
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov, robot_radius=w.robot_radius,
              n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    s = fs.FrontierScorer(device=0)
    try:
        s.set_ray_params(**kw)
        s.upload_grid(w.cells, w.origin, w.resolution)
        s.upload_landmarks(w.landmarks); s.lookup_generate(); s.set_fim_params(14.0, 1.0)
        s.max_arrival()
        before = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        for origin, res in (((np.nan, 0.0, 0.0), 0.05), ((0.0, np.inf, 0.0), 0.05), ((0.0, 0.0, 0.0), np.inf), ((0.0, 0.0, 0.0), np.nan), ((0.0, 0.0, 0.0), 0.0), ((0.0, 0.0, 0.0), -0.05)):
            with pytest.raises(fs.FsError):
                s.upload_grid(w.cells, origin, res)
        for bad in (dict(max_camera_depth=np.inf), dict(max_camera_depth=np.nan), dict(delta_theta=np.nan), dict(camera_fov=np.inf), dict(robot_radius=np.nan),
                    dict(robot_radius=-0.1), dict(elev=(np.nan,)), dict(polygon=(np.nan, 0.0, 1.0, 1.0))):
            with pytest.raises(fs.FsError):
                s.set_ray_params(**dict(kw, **bad))
        for dist, angle in ((np.nan, 1.0), (np.inf, 1.0), (1.0e9, 1.0), (0.0, 1.0), (14.0, np.nan), (14.0, 0.0)):
            with pytest.raises(fs.FsError):
                s.set_fim_params(dist, angle)
        s.set_fim_params(9.9e8, 1.0); s.set_fim_params(14.0, 1.0)          # just inside the limit is fine
        after = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        for k in ("arrival", "argmax", "n_visible", "flags"):
            np.testing.assert_array_equal(after[k], before[k], err_msg=k)
    finally:
        s.close()
