"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer outputs (per-ray counts, arrival, argmax, achievability, status, visible /
voxel counts) must be bit-exact; Fisher information within 1e-4 relative (BASELINE.json north_star).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4   # north_star tolerance for FIM trace / D-optimality / info


def _parity():
    import importlib
    return importlib.import_module("fit-slam_amd.parity")


def _oracle_grid_params(oracle, w, **over):
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    kw.update(over)
    return G, oracle.RayParams(**kw)


def _setup_scorer(scorer, w, **over):
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    kw.update(over)
    scorer.set_ray_params(**kw)
    scorer.upload_grid(w.cells, w.origin, w.resolution)
    scorer.upload_landmarks(w.landmarks)
    scorer.set_fim_params(14.0, 1.0)


def _assert_arrival_equal(got, want):
    for k in ("status", "arrival", "argmax", "achievable"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_array_equal(got["yaw"], want["yaw"])          # same fp64 expression -> identical bits
    if got.get("ray_counts") is not None and want.get("ray_counts") is not None:
        np.testing.assert_array_equal(got["ray_counts"], want["ray_counts"])


def test_fp64_primitives_match_host(scorer):
    """sqrt of exact integers and division on the device == libm hypot / host division, bit for bit."""
    assert scorer.selftest_fp64(512) == 0


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_arrival_reference_defaults_2d(fs, oracle, scorer, seed):
    """The bit-exact 2-D slice: 63 rays (delta 0.10 accumulated), L = 40, window 10, polygon clamp."""
    w = fs.synth.make_small_2d(seed)
    G, P = _oracle_grid_params(oracle, w)
    _setup_scorer(scorer, w)
    assert (scorer.n_yaw, scorer.window) == (63, 10)
    mx_o = oracle.max_arrival_information(G, P)
    mx_g = scorer.max_arrival()
    assert mx_g == mx_o
    want = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx_o["min_gt"], faithful=True)
    got = scorer.score_arrival(w.goals, w.frontier_size, w.blacklisted)
    _assert_arrival_equal(got, want)
    assert (want["arrival"] > 0).sum() > 0


@pytest.mark.parametrize("name", ["C1_2D", "C1", "C2"])
def test_arrival_configs(fs, oracle, scorer, name):
    w = fs.synth.make_workload(name, n_cand=1500 if name == "C2" else None)
    G, P = _oracle_grid_params(oracle, w)
    _setup_scorer(scorer, w)
    mx_o = oracle.max_arrival_information(G, P)
    assert scorer.max_arrival() == mx_o
    want = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx_o["min_gt"], n_threads=8)
    got = scorer.score_arrival(w.goals, w.frontier_size, w.blacklisted)
    _assert_arrival_equal(got, want)


def test_arrival_edge_cases(fs, oracle, scorer):
    """off-map goals, goals on the border, NaN goal, blacklisted, achievable_in = 0, empty list."""
    w = fs.synth.make_small_2d(21, n=64, n_cand=32)
    G, P = _oracle_grid_params(oracle, w)
    _setup_scorer(scorer, w)
    lo = w.origin[0]
    hi = w.origin[0] + 64 * w.resolution
    goals = np.array([[lo - 1.0, 0.0, 0.0], [0.0, hi + 5.0, 0.0], [lo + 1e-9, lo + 1e-9, 0.0],
                      [hi - 1e-9, hi - 1e-9, 0.0], [np.nan, 0.0, 0.0], [0.0, 0.0, 0.0], [1e30, 0.0, 0.0],
                      [0.3, -0.2, 0.0]], dtype=np.float64)
    fsz = np.array([1, 2, 3, 4, 5, 50, 7, 9], dtype=np.int32)
    bl = np.array([0, 0, 0, 0, 0, 0, 0, 1], dtype=np.uint8)
    ai = np.array([1, 1, 0, 1, 1, 1, 1, 1], dtype=np.uint8)
    want = oracle.arrival_information(G, P, goals, fsz, bl, ai, min_gt=50.0, faithful=True)
    scorer.set_arrival_limits(100.0, 50.0)
    got = scorer.score_arrival(goals, fsz, bl, ai)
    _assert_arrival_equal(got, want)
    assert list(want["status"][[0, 1, 4, 6, 7]]) == [1, 1, 1, 1, 2]
    empty = scorer.score_arrival(np.zeros((0, 3)))
    assert empty["arrival"].shape == (0,)


def test_arrival_properties(fs, scorer):
    """Oracle-free properties: all-unknown grid -> every candidate reaches the geometric maximum;
    all-free grid -> zero; a wall of lethal cells blocks exactly the rays that cross it."""
    n = 128
    res = 0.05
    origin = (-n * res / 2, -n * res / 2, 0.0)
    scorer.set_ray_params(polygon=(-100, -100, 100, 100))
    goals = np.array([[0.0, 0.0, 0.0], [0.5, -0.3, 0.0], [-0.71, 0.42, 0.0]])
    scorer.upload_grid(np.full((n, n), 255, np.uint8), origin, res)
    mx = scorer.max_arrival()
    got = scorer.score_arrival(goals)
    assert mx["max_value"] > 0
    # interior goals on an all-unknown map see the same count geometry up to sub-cell offsets
    assert np.all(np.abs(got["arrival"] - mx["max_value"]) <= 12)
    assert got["arrival"][0] == mx["max_value"]
    scorer.upload_grid(np.zeros((n, n), np.uint8), origin, res)
    got = scorer.score_arrival(goals)
    assert np.all(got["arrival"] == 0) and np.all(got["ray_counts"] == 0)
    cells = np.full((n, n), 255, np.uint8)
    cells[:, n // 2 + 10] = 254            # a wall 10 cells to the +x side of the centre
    scorer.upload_grid(cells, origin, res)
    got = scorer.score_arrival(goals[:1])
    rc = got["ray_counts"][0, 0]
    assert rc[0] == 10                      # theta = 0: cells 0..9 unknown, cell 10 is the wall
    assert rc[31] >= 40                     # theta ~ pi: unobstructed, ~40 steps + the final visit


def test_fim_known_answers(scorer):
    """SURVEY.md App. C.1 closed form 2 + 2/|p|^2 through the table + single landmarks."""
    pts = np.array([[0.3, 0, 0], [0.3, 0.3, 0.3], [1.2, -0.9, 0.3], [3, 0.3, -0.6]], dtype=np.float32)
    expect = [24.222222, 9.407407, 2.854701, 2.211640]
    pose = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    scorer.set_fim_params(14.0, 1.3)
    for p, e in zip(pts, expect):
        scorer.upload_landmarks(p[None])
        r = scorer.score_fim(pose)
        assert r["n_visible"][0] == 1 and r["n_voxels"][0] == 1
        assert abs(r["info_ref"][0] - e) <= 2e-6 * e
        n2 = float(np.dot(p, p))
        assert abs(r["trace"][0] - (2 + 2 / n2)) <= 1e-5 * (2 + 2 / n2)
    # crowding: k landmarks in one voxel -> info * sum_{j<=k} exp(1 - j^0.8)   (App. C.4)
    scorer.upload_landmarks(np.tile(np.array([[1.2, -0.9, 0.3]], np.float32), (5, 1)))
    r = scorer.score_fim(pose)
    S5 = sum(np.exp(1 - j ** 0.8) for j in range(1, 6))
    assert r["n_visible"][0] == 5 and r["n_voxels"][0] == 1
    assert abs(r["info_ref"][0] - 2.854701 * S5) < 2e-4


def test_fim_voxel_count_saturates(fs, oracle, scorer, ref_table):
    """3 000 landmarks in ONE voxel (and 1 500 in a second): the slot's 11-bit count field stops at FS_SLOT_CNT_SAT = 1024 —
    beyond rank 337 the factor exp(1 - k^0.8) is 0.0f in float32 anyway — and must neither run over into the key bits nor
    open a second entry for the voxel."""
    rng = np.random.default_rng(17)
    a = np.array([1.2, -0.9, 0.3]) + rng.uniform(-0.05, 0.05, size=(3000, 3))
    b = np.array([3.0, 0.3, -0.6]) + rng.uniform(-0.05, 0.05, size=(1500, 3))
    lm = np.concatenate([a, b]).astype(np.float32)
    rng.shuffle(lm)
    pose = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    scorer.upload_landmarks(lm)
    scorer.set_fim_params(14.0, 1.3)
    got = scorer.score_fim(pose)
    want = oracle.pose_information(ref_table, lm, pose, 14.0, 1.3, n_threads=1)
    assert got["n_visible"][0] == 4500 and got["n_voxels"][0] == 2
    _check_fim(got, want, lm.shape[0])


def test_fim_learnt_pass_prediction_is_only_a_prediction(fs, oracle, scorer, ref_table):
    """The worker predicts a pose's distinct voxels from the largest voxels-per-landmark ratio finished calls have shown on this
    cloud (fs_fim.hip, counters[12]).  Train it on poses inside a dense clump (0.02 voxels per landmark), then look at a sparse
    region where nearly every landmark has a voxel of its own: the single pass it predicts overflows, the HBM tier takes the
    pose — same records as the oracle's — and the ratio it showed is what the next call predicts with."""
    rng = np.random.default_rng(23)
    dense = (np.array([4.0, 0.0, 0.0]) + rng.uniform(-1.5, 1.5, size=(60000, 3))).astype(np.float32)
    sparse = rng.normal(size=(60000, 3))
    sparse = sparse / np.linalg.norm(sparse, axis=1, keepdims=True) * (13.5 * rng.random(60000) ** (1 / 3))[:, None]
    sparse[:, 0] = np.abs(sparse[:, 0]) + 200.0                      # a half ball in front of (200, 0, 0)
    lm = np.concatenate([dense, sparse.astype(np.float32)])
    near = np.zeros((8, 7)); near[:, 6] = 1.0; near[:, 1] = np.linspace(-0.3, 0.3, 8)
    far = np.zeros((2, 7)); far[:, 6] = 1.0; far[:, 0] = 200.0; far[1, 2] = 0.2
    scorer.upload_landmarks(lm)
    scorer.set_fim_params(14.0, 1.5)
    scorer.set_option("fim.split", 0)                                 # (one workgroup per pose: a split call neither learns nor needs to)
    try:
        hbm0 = scorer.get_counter(5)
        got = scorer.score_fim(near)
        _check_fim(got, oracle.pose_information(ref_table, lm, near, 14.0, 1.5, n_threads=8), lm.shape[0])
        assert got["n_voxels"].max() * 20 < got["n_visible"].max()       # the clump is dense
        assert scorer.get_counter(5) == hbm0                              # ... and fitted its tables
        mixed = np.concatenate([near[:4], far])
        want = oracle.pose_information(ref_table, lm, mixed, 14.0, 1.5, n_threads=8)
        assert want["n_voxels"][4:].min() > 16384                         # more voxels than an LDS table has slots
        for _ in range(2):                                                # the second time with the ratio the first one raised
            _check_fim(scorer.score_fim(mixed), want, lm.shape[0])
        assert scorer.get_counter(5) > hbm0
    finally:
        scorer.set_option("fim.split", 3)


@pytest.mark.parametrize("angle", [1.0, 4.0])
def test_one_pose_over_several_workgroups(fs, oracle, scorer, ref_table, angle):
    """A call with few poses asks for info_ref alone (isPoseSafe: one pose per tick, FisherInfoBTPlugin.cpp:24-57) and spreads each
    pose over W = 2, 4, 8, 16 workgroups by voxel slab ("fim.split" = log2 W; fs_fim.hip SPLIT).  Whole voxels stay in one
    workgroup, so n_voxels is exact and the information is the same multiset of terms: equal to the unsplit call and the oracle
    for 1 .. 100 poses (beyond 256 / W poses the call falls back to fewer workgroups per pose, then to none)."""
    w = fs.synth.make_workload("REF2D", n_cand=100)
    rng = np.random.default_rng(5)
    poses = fs.synth.poses_from_yaw(w.goals, rng.uniform(-np.pi, np.pi, 100))
    poses[:, 2] = 0.4
    q = rng.normal(size=(6, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[:6, 3:] = q
    scorer.upload_landmarks(w.landmarks)
    scorer.lookup_generate()
    scorer.set_fim_params(14.0, angle)
    want = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, angle, n_threads=8)
    scale = np.maximum(np.abs(want["info_f64"]), 1e-6)
    try:
        for split in (0, 1, 2, 3, 4):
            scorer.set_option("fim.split", split)
            for n in (1, 2, 5, 31, 100):
                for _ in range(2):                                   # twice: the flags of a split call are left clean for the next
                    got = scorer.score_fim(poses[:n], info_only=True)
                    np.testing.assert_array_equal(got["n_voxels"], want["n_voxels"][:n], err_msg=f"split {split} n {n}")
                    assert np.max(np.abs(got["info_ref"] - want["info_f64"][:n]) / scale[:n]) <= REL, (split, n)
                # every column (the general workers: sums at scoring time with the cone, at test time without it)
                full = scorer.score_fim(poses[:n])
                _check_fim(full, {k: v[:n] for k, v in want.items()}, w.landmarks.shape[0])
                # the info-only call adds its W partial sums on the HOST (one launch); with the finish kernel instead the same
                # doubles are added in the same order (the terms INSIDE a partial sum may differ by the order of racing LDS atomics
                # from call to call: a few ulp between two calls, not bit equality)
                scorer.set_option("fim.hostfinish", 0)
                dev = scorer.score_fim(poses[:n], info_only=True)
                scorer.set_option("fim.hostfinish", 1)
                np.testing.assert_array_equal(dev["n_voxels"], got["n_voxels"])
                assert np.max(np.abs(dev["info_ref"].astype(np.float64) - got["info_ref"]) / np.maximum(np.abs(got["info_ref"]), 1.0)) <= 2e-6
    finally:
        scorer.set_option("fim.split", 3)
        scorer.set_option("fim.hostfinish", 1)


@pytest.mark.parametrize("name,angle", [("REF2D", 1.0), ("REF2D", 4.0), ("C2", 1.0), ("C2", 4.0)])
def test_fused_scoring_of_a_few_frontiers_spreads_each_pose_over_workgroups(fs, oracle, scorer, ref_table, name, angle):
    """fs_score_candidates with a handful of frontiers (the reference scores tens per tick) runs the YAW_ONLY workers with every
    pose over several workgroups (REF2D's 1 563 chunks: eight; C2's 782: four): the records must be those of the unsplit call
    (integers bit for bit) and the oracle's, whatever "fim.split" says."""
    w = fs.synth.make_workload(name, n_cand=40)
    G, P = _oracle_grid_params(oracle, w)
    _setup_scorer(scorer, w)
    scorer.set_fim_params(14.0, angle)
    mx = oracle.max_arrival_information(G, P)
    assert scorer.max_arrival() == mx
    arr = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], n_threads=8)
    fim = oracle.pose_information(ref_table, w.landmarks, oracle.poses_from_yaw(w.goals, arr["yaw"]), 14.0, angle, n_threads=8)
    try:
        for split in (0, 3, 2):
            scorer.set_option("fim.split", split)
            for n in (1, 3, 17, 40):
                rec = scorer.score_candidates(w.goals[:n], w.frontier_size[:n], w.blacklisted[:n])
                ok = arr["status"][:n] == 0
                np.testing.assert_array_equal(rec["arrival"], arr["arrival"][:n])
                np.testing.assert_array_equal(rec["n_visible"][ok], fim["n_visible"][:n][ok], err_msg=f"split {split} n {n}")
                np.testing.assert_array_equal(fs.capi.record_nvoxels(rec)[ok], np.minimum(fim["n_voxels"][:n][ok], 65535))
                sc = np.maximum(np.abs(fim["info_f64"][:n][ok]), 1e-6)
                assert np.max(np.abs(rec["info_ref"][ok] - fim["info_f64"][:n][ok]) / sc, initial=0.0) <= REL
                assert np.max(np.abs(rec["trace"][ok] - fim["trace"][:n][ok]) / np.maximum(fim["trace"][:n][ok], 1e-6), initial=0.0) <= REL
                gate = _parity().logdet_gate(rec["logdet"], fim["logdet"][:n], fim["fim"][:n], consider=ok, n_visible=fim["n_visible"][:n])
                assert gate["ok"], (split, n, gate)
    finally:
        scorer.set_option("fim.split", 3)
        scorer.set_fim_params(14.0, 1.0)


@pytest.mark.parametrize("name,n", [("C2", 17), ("REF2D", 9), ("REF2D", 33)])
def test_first_call_of_a_fresh_context_is_a_split_short_list(fs, oracle, ref_table, name, n):
    """The very FIRST scoring call of a context is a short list whose poses are spread over several workgroups: the per-item
    scratch (n * W items) outgrows what the call's first sizing (n candidates, 64 elements at least) allocated, so every
    per-candidate column moves while the call is being set up.  The ray-march kernel and the finish kernel must both see the
    columns where they ended up (round 5: fs_score_candidates_dev bound the ray-march outputs before the split had grown the
    columns — arrival 0 and stale argmax in every record; found by tests/fused_random.py, trial 6 of seed 3).  Same for
    fs_score_fim asking for the 6x6 upper triangles.  (With the production library this catches the defect only when the allocator
    hands the grown column a new address — it often re-uses the block it just got back; under the FS_POISON development build,
    which retires grown buffers, it fails every time: tests/test_gpu_lifecycle.py runs it there in a child process.)"""
    w = fs.synth.make_workload(name, n_cand=40)
    G, P = _oracle_grid_params(oracle, w)
    mx = oracle.max_arrival_information(G, P)
    arr = oracle.arrival_information(G, P, w.goals[:n], w.frontier_size[:n], w.blacklisted[:n], min_gt=mx["min_gt"], n_threads=8)
    poses = oracle.poses_from_yaw(w.goals[:n], arr["yaw"])
    fim = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
    ok = arr["status"] == 0
    for first in ("fused", "fim21"):
        sc = fs.FrontierScorer(0)                                    # fresh: nothing has sized the scratch yet
        try:
            _setup_scorer(sc, w)
            sc.lookup_generate()
            sc.set_arrival_limits(mx["max_gt"], mx["min_gt"])
            if first == "fim21":
                full = sc.score_fim(poses)
                _check_fim(full, fim, w.landmarks.shape[0])
            rec = sc.score_candidates(w.goals[:n], w.frontier_size[:n], w.blacklisted[:n])
            np.testing.assert_array_equal(rec["arrival"], arr["arrival"])
            np.testing.assert_array_equal(rec["argmax"], arr["argmax"])
            np.testing.assert_array_equal(fs.capi.record_status(rec), arr["status"])
            np.testing.assert_array_equal(rec["n_visible"][ok], fim["n_visible"][ok])
            scale = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
            assert np.max(np.abs(rec["info_ref"][ok] - fim["info_f64"][ok]) / scale, initial=0.0) <= REL
        finally:
            sc.close()


def test_split_pose_that_overflows_goes_to_the_hbm_tier_whole(fs, oracle, scorer, ref_table):
    """A split workgroup sizes its table for its share of the landmarks scanned, so it overflows only where even that share holds
    more distinct voxels than the largest LDS table has slots.  Poses in a sparse half ball of 200 k landmarks show > 2 x 16 384
    voxels each; spread over W = 2 workgroups on a ratio learnt from a dense clump (one pass predicted), both items of such a pose
    run out of table: the FIRST to fail hands the whole pose to the HBM tier — once —, and the finish takes that result instead
    of the partial sums.  The poses next to the clump stay split and in LDS."""
    rng = np.random.default_rng(23)
    dense = (np.array([4.0, 0.0, 0.0]) + rng.uniform(-1.5, 1.5, size=(60000, 3))).astype(np.float32)
    sparse = rng.normal(size=(200000, 3))
    sparse = sparse / np.linalg.norm(sparse, axis=1, keepdims=True) * (13.5 * rng.random(200000) ** (1 / 3))[:, None]
    sparse[:, 0] = np.abs(sparse[:, 0]) + 200.0
    lm = np.concatenate([dense, sparse.astype(np.float32)])
    near = np.zeros((8, 7)); near[:, 6] = 1.0; near[:, 1] = np.linspace(-0.3, 0.3, 8)
    far = np.zeros((2, 7)); far[:, 6] = 1.0; far[:, 0] = 200.0; far[1, 2] = 0.2
    scorer.upload_landmarks(lm)
    scorer.set_fim_params(14.0, 1.5)
    scorer.set_option("fim.split", 0)
    try:
        for _ in range(2):
            scorer.score_fim(np.repeat(near, 40, axis=0), info_only=True)        # 320 poses: unsplit, and big enough to teach the ratio
        mixed = np.concatenate([near[:3], far])
        want = oracle.pose_information(ref_table, lm, mixed, 14.0, 1.5, n_threads=8)
        assert want["n_voxels"][3:].min() > 2 * 16384
        sc = np.maximum(np.abs(want["info_f64"]), 1e-6)
        for split in (1, 3):
            scorer.set_option("fim.split", split)
            hbm0 = scorer.get_counter(5)
            for _ in range(2):
                got = scorer.score_fim(mixed, info_only=True)
                np.testing.assert_array_equal(got["n_voxels"], want["n_voxels"])
                assert np.max(np.abs(got["info_ref"] - want["info_f64"]) / sc) <= REL
            _check_fim(scorer.score_fim(mixed), want, lm.shape[0])       # every column through the same hand-over (general worker)
            if split == 1:
                assert scorer.get_counter(5) >= hbm0 + 2                  # the HBM tier took the far poses (each ONCE per call)
                assert scorer.get_counter(5) <= hbm0 + 6
    finally:
        scorer.set_option("fim.split", 3)


def test_fim_voxel_rounding_next_to_the_boundary(fs, oracle, scorer, ref_table):
    """getVoxelCoordinate rounds x / step in double; the kernel rounds an fp32 product and re-evaluates in fp64 only inside a band
    around the half-integers whose width follows the visibility range (FsFimArgs::key_thr).  Landmarks ON the boundaries and a
    few ulps either side of them, at every lattice index the range allows: the voxel of every one of them must be the oracle's."""
    step = np.float64(np.float32(0.3))
    k = np.arange(0, 46, dtype=np.float64)
    base = ((k + 0.5) * step).astype(np.float32)                     # x on (or one rounding off) a voxel boundary
    xs = [base]
    for ulps in (1, 2, 3, 5, 9, 17, 40, 100, 400):
        up, dn = base.copy(), base.copy()
        for _ in range(ulps if ulps < 40 else 0):
            up, dn = np.nextafter(up, np.float32(np.inf)), np.nextafter(dn, np.float32(-np.inf))
        if ulps >= 40:
            up, dn = base * np.float32(1 + ulps * 6e-8), base * np.float32(1 - ulps * 6e-8)
        xs += [up, dn]
    x = np.concatenate(xs)
    rng = np.random.default_rng(3)
    lm = np.stack([x, rng.uniform(-0.1, 0.1, x.size).astype(np.float32), rng.uniform(-0.1, 0.1, x.size).astype(np.float32)], axis=1)
    lm = np.concatenate([lm, lm[:, [1, 0, 2]] * np.float32([1, 0.3, 1]) + np.float32([2.0, 0, 0]), lm[:, [1, 2, 0]] * np.float32([1, 1, 0.3]) + np.float32([2.0, 0, 0])])
    pose = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    scorer.upload_landmarks(lm)
    for max_dist in (14.0, 30.0):
        scorer.set_fim_params(max_dist, 1.5)
        got = scorer.score_fim(pose)
        want = oracle.pose_information(ref_table, lm, pose, max_dist, 1.5, n_threads=1)
        assert want["n_visible"][0] > 500
        _check_fim(got, want, lm.shape[0])


def _check_fim(got, want, n_lm):
    np.testing.assert_array_equal(got["n_visible"], want["n_visible"])
    np.testing.assert_array_equal(got["n_voxels"], want["n_voxels"])
    scale = np.maximum(np.abs(want["info_f64"]), 1e-6)
    assert np.max(np.abs(got["info_ref"] - want["info_f64"]) / scale) <= REL
    # the reference's own float32 running sum drifts from the exact sum; allow its drift on top
    drift = np.abs(want["info_ref"] - want["info_f64"]) / scale
    assert np.max(np.abs(got["info_ref"] - want["info_ref"]) / scale - drift) <= REL
    tr = np.maximum(np.abs(want["trace"]), 1e-6)
    assert np.max(np.abs(got["trace"] - want["trace"]) / tr) <= REL
    # D-optimality has no reference counterpart: it is checked against the float64 oracle, under the ONE rule
    # bench.py's gates use too (fit-slam_amd/parity.py: plain 1e-4 plus kappa * 2^-24, the float32 F's own rounding)
    gate = _parity().logdet_gate(got["logdet"], want["logdet"], want["fim"], n_visible=want["n_visible"])
    assert gate["ok"], gate
    if got.get("fim21") is not None:
        iu = np.triu_indices(6)
        wantF = want["fim"][:, iu[0], iu[1]]
        mag = np.maximum(np.abs(wantF).max(axis=1, keepdims=True), 1e-6)
        assert np.max(np.abs(got["fim21"] - wantF) / mag) <= REL


@pytest.mark.parametrize("name,n_pose", [("C1", 200), ("C2", 300)])
def test_fim_explicit_poses(fs, oracle, scorer, ref_table, name, n_pose):
    w = fs.synth.make_workload(name, n_cand=n_pose)
    rng = np.random.default_rng(5)
    yaw = rng.uniform(-np.pi, np.pi, size=n_pose)
    poses = oracle.poses_from_yaw(w.goals, yaw)
    # a few general (non-yaw-only) orientations
    q = rng.normal(size=(8, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[:8, 3:] = q
    _setup_scorer(scorer, w)
    want = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
    got = scorer.score_fim(poses)
    _check_fim(got, want, w.landmarks.shape[0])


def test_fim_cone_modes_and_empty(fs, oracle, scorer, ref_table):
    w = fs.synth.make_workload("C1", n_cand=40)
    poses = oracle.poses_from_yaw(w.goals, np.linspace(-3, 3, 40))
    _setup_scorer(scorer, w)
    # (400 m: max_dist / 0.3 m passes 2^10 lattice cells, where the fp32 fast path of the voxel index is not proven — the kernel
    # then rounds every landmark in fp64: FsFimArgs::far_lattice)
    for max_dist, max_angle in [(14.0, 4.0), (3.0, 2.2), (1.0, 0.2), (6.0, np.pi / 2), (400.0, 1.0)]:
        scorer.set_fim_params(max_dist, max_angle)
        want = oracle.pose_information(ref_table, w.landmarks, poses, max_dist, max_angle, n_threads=4)
        got = scorer.score_fim(poses)
        _check_fim(got, want, w.landmarks.shape[0])
    scorer.set_fim_params(14.0, 1.0)
    scorer.upload_landmarks(np.zeros((0, 3), np.float32))
    got = scorer.score_fim(poses[:3])
    assert np.all(got["info_ref"] == 0) and np.all(got["n_visible"] == 0) and np.all(np.isneginf(got["logdet"]))


def test_fim_hash_overflow_pass(fs, oracle, scorer, ref_table):
    """More distinct voxels than the 32768-slot LDS table: the HBM-table pass must take over."""
    rng = np.random.default_rng(9)
    m = 110_000
    r = rng.uniform(0.5, 13.5, size=m) ** (1 / 1.0)
    u = rng.normal(size=(m, 3)); u[:, 0] = np.abs(u[:, 0]) * 3 + 0.5
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    lm = (u * r[:, None]).astype(np.float32)
    poses = np.array([[0, 0, 0, 0, 0, 0, 1.0], [0.1, -0.2, 0.05, 0, 0, 0.1, 0.995]])
    scorer.upload_landmarks(lm)
    scorer.set_fim_params(14.0, 1.3)
    want = oracle.pose_information(ref_table, lm, poses, 14.0, 1.3)
    assert want["n_voxels"].max() > 33_000
    got = scorer.score_fim(poses)
    _check_fim(got, want, m)


@pytest.mark.parametrize("name", ["C1", "C2", "REF2D"])
def test_fused_candidates(fs, oracle, scorer, ref_table, name):
    """REF2D: the reference's own 2-D operating point (512^2 costmap, 63 rays, a planar cloud with heights) — the first 300
    candidates hold poses with exactly three visible landmarks (kappa 1e5 and beyond), the ones bench.py's plain-1e-4 gate was
    red on in round 4."""
    w = fs.synth.make_workload(name, n_cand={"C2": 600, "REF2D": 300}.get(name))
    G, P = _oracle_grid_params(oracle, w)
    _setup_scorer(scorer, w)
    mx = oracle.max_arrival_information(G, P)
    assert scorer.max_arrival() == mx
    arr = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], n_threads=8)
    poses = oracle.poses_from_yaw(w.goals, arr["yaw"])
    fim = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
    rec = scorer.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    ok = arr["status"] == 0
    np.testing.assert_array_equal(rec["arrival"], arr["arrival"])
    np.testing.assert_array_equal(rec["argmax"], arr["argmax"])
    np.testing.assert_array_equal(fs.capi.record_status(rec), arr["status"])
    np.testing.assert_array_equal(fs.capi.record_achievable(rec), arr["achievable"])
    np.testing.assert_array_equal(rec["yaw"], arr["yaw"].astype(np.float32))
    np.testing.assert_array_equal(rec["n_visible"][ok], fim["n_visible"][ok])
    np.testing.assert_array_equal(fs.capi.record_nvoxels(rec)[ok], np.minimum(fim["n_voxels"][ok], 65535))
    assert np.all(rec["info_ref"][~ok] == 0) and np.all(rec["n_visible"][~ok] == 0)
    scale = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
    assert np.max(np.abs(rec["info_ref"][ok] - fim["info_f64"][ok]) / scale) <= REL
    assert np.max(np.abs(rec["trace"][ok] - fim["trace"][ok]) / np.maximum(fim["trace"][ok], 1e-6)) <= REL
    gate = _parity().logdet_gate(rec["logdet"], fim["logdet"], fim["fim"], consider=ok, n_visible=fim["n_visible"])
    assert gate["ok"], gate
    assert np.all(np.isneginf(rec["logdet"][~ok]))


def test_rank_matches_oracle(fs, oracle, scorer):
    w = fs.synth.make_workload("C1")
    G, P = _oracle_grid_params(oracle, w)
    _setup_scorer(scorer, w)
    # C1's 3.2 m map cannot hold the calibration fan (max_arrival = 0): set limits by hand
    scorer.set_arrival_limits(400.0, 40.0)
    rec = scorer.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    rng = np.random.default_rng(3)
    n = rec.shape[0]
    plen = rng.uniform(0.5, 30.0, size=n)
    phead = rng.uniform(0.0, np.pi, size=n)
    rc, want = oracle.u1_costs(rec["arrival"].astype(np.float64), fs.capi.record_achievable(rec), plen, phead, 400.0,
                               blacklisted=w.blacklisted)
    assert rc == 0
    got = scorer.rank_candidates(rec, plen, phead, blacklisted=w.blacklisted)
    for k in ("weighted_cost", "arrival_utility", "distance_utility"):
        np.testing.assert_array_equal(got[k], want[k])
    np.testing.assert_array_equal(got["order"], np.argsort(want["weighted_cost"], kind="stable"))
    # the device-resident form behind the scoring call: records straight from fs_score_candidates_dev, nothing visits the host
    import torch
    dev = torch.device("cuda", 0)
    d_goal = torch.from_numpy(w.goals).to(dev); d_fs = torch.from_numpy(w.frontier_size).to(dev); d_bl = torch.from_numpy(w.blacklisted).to(dev)
    d_rec = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    d_len = torch.from_numpy(plen).to(dev); d_head = torch.from_numpy(phead).to(dev)
    d_cost = torch.zeros(n, dtype=torch.float64, device=dev); d_au = torch.zeros_like(d_cost); d_du = torch.zeros_like(d_cost)
    d_order = torch.zeros(n, dtype=torch.int32, device=dev); d_err = torch.ones(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    scorer.score_candidates_dev(n, d_goal.data_ptr(), d_fs.data_ptr(), d_bl.data_ptr(), 0, d_rec.data_ptr())
    scorer.rank_candidates_dev(n, d_rec.data_ptr(), d_len.data_ptr(), d_head.data_ptr(), d_cost.data_ptr(), d_au.data_ptr(),
                               d_du.data_ptr(), d_order.data_ptr(), d_black=d_bl.data_ptr(), d_err=d_err.data_ptr())
    scorer.synchronize()
    assert int(d_err.cpu()[0]) == 0
    np.testing.assert_array_equal(d_cost.cpu().numpy(), want["weighted_cost"])
    np.testing.assert_array_equal(d_au.cpu().numpy(), want["arrival_utility"])
    np.testing.assert_array_equal(d_du.cpu().numpy(), want["distance_utility"])
    np.testing.assert_array_equal(d_order.cpu().numpy(), got["order"])
    # optional columns left out; a utility outside [0, 1] raises the flag instead of throwing (FrontierCostsManager.cpp:148-149)
    scorer.set_arrival_limits(1.0, 0.1)                               # arrival / max_gt > 1
    scorer.rank_candidates_dev(n, d_rec.data_ptr(), d_len.data_ptr(), d_head.data_ptr(), d_cost.data_ptr(), d_err=d_err.data_ptr())
    scorer.synchronize()
    assert int(d_err.cpu()[0]) != 0
    with pytest.raises(fs.capi.FsError):
        scorer.rank_candidates(rec, plen, phead, blacklisted=w.blacklisted)
    scorer.set_arrival_limits(400.0, 40.0)


def test_fim_culling_is_exact(fs, oracle, scorer, ref_table):
    """Chunk culling (Morton-ordered landmark chunks + bounding spheres) must leave the visible set, the
    voxel counts and the FI untouched; it only removes work."""
    w = fs.synth.make_workload("C2", n_cand=400)
    rng = np.random.default_rng(17)
    poses = oracle.poses_from_yaw(w.goals, rng.uniform(-np.pi, np.pi, size=400))
    _setup_scorer(scorer, w)
    out = {}
    for cull in (1, 0):
        scorer.set_option("fim.cull", cull)
        scorer.get_counter(0, reset=True)
        out[cull] = scorer.score_fim(poses)
        out[cull]["tested"] = scorer.get_counter(0, reset=True)
    scorer.set_option("fim.cull", 1)
    m_pad = -(-w.landmarks.shape[0] // 64) * 64
    # without culling every candidate tests the whole cloud once per scoring pass (a cloud of this size is predicted to
    # overfill one LDS table, so the brute-force mode runs a few passes per candidate)
    assert 400 * m_pad <= out[0]["tested"] <= 8 * 400 * m_pad and out[0]["tested"] % m_pad == 0
    assert out[1]["tested"] < 0.5 * out[0]["tested"]
    for k in ("n_visible", "n_voxels"):
        np.testing.assert_array_equal(out[0][k], out[1][k])
    np.testing.assert_allclose(out[0]["info_ref"], out[1]["info_ref"], rtol=2e-6)
    want = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
    _check_fim(out[1], want, w.landmarks.shape[0])
    # a non-unit quaternion (Eigen would not normalise it) silently falls back to the brute-force scan
    bad = poses[:4].copy(); bad[:, 3:] *= 1.3
    got = scorer.score_fim(bad)
    want = oracle.pose_information(ref_table, w.landmarks, bad, 14.0, 1.0)
    np.testing.assert_array_equal(got["n_visible"], want["n_visible"])


def test_trace_segments_matches_oracle(fs, oracle, scorer):
    """getTracedCells + RayTracedCells accessors for arbitrary segments (roadmap edge test, recovery controller)."""
    rng = np.random.default_rng(23)
    for w, is3d in ((fs.synth.make_small_2d(41, n=128, n_cand=8), False), (fs.synth.make_workload("C1", n_cand=8), True)):
        scorer.upload_grid(w.cells, w.origin, w.resolution)
        G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
        nz, ny, nx = w.cells.shape
        lo = np.array(w.origin); hi = lo + np.array([nx, ny, nz]) * w.resolution
        n = 600
        a = rng.uniform(lo - 0.1, hi + 0.1, size=(n, 3)); b = a + rng.normal(scale=1.5, size=(n, 3))
        if not is3d:
            a[:, 2] = 0.0; b[:, 2] = 0.0
        b[:20] = a[:20]                                           # zero-length segments: one visit
        for obst, trace, L in (((253, 254), (0, 255), 122), ((256, 256), (0, 255), 80), ((240, 254), (255, 255), 40)):
            got = scorer.trace_segments(a, b, L, obst=obst, trace=trace)
            for i in range(n):
                r = oracle.trace_ray(G, a[i], b[i], L, obst=obst, trace=trace, faithful=True)
                assert bool(got["ok"][i]) == r["ok"]
                if r["ok"]:
                    assert (got["traced"][i], bool(got["hit"][i]), got["unknown"][i], got["all"][i]) == \
                        (r["traced"], r["hit"], r["unknown"], r["all"]), i


def test_frontier_cell_stencil(fs, oracle, scorer):
    """isNewFrontierCell for every cell (the producer side of the candidate list)."""
    for w in (fs.synth.make_small_2d(51, n=200, n_cand=4), fs.synth.make_workload("C1", n_cand=4)):
        scorer.upload_grid(w.cells, w.origin, w.resolution)
        for thr in (160, 254, 1):
            mask, count = scorer.frontier_cells(w.cells.shape, thr)
            want = oracle.frontier_cell_mask(w.cells, thr)
            np.testing.assert_array_equal(mask, want)
            assert count == int(want.sum())
    edge = np.full((1, 5, 37), 255, np.uint8); edge[0, 0, 0] = 0; edge[0, 4, 36] = 0; edge[0, 2, 17] = 200; edge[0, 2, 19] = 0
    scorer.upload_grid(edge, (0, 0, 0), 0.05)
    mask, count = scorer.frontier_cells(edge.shape, 160)
    np.testing.assert_array_equal(mask, oracle.frontier_cell_mask(edge, 160))
    assert mask[0, 2, 18] == 0 and mask[0, 1, 19] == 1 and mask[0, 0, 1] == 1     # lethal neighbour vetoes; corners work


def test_brick_upload_equals_dense_upload(fs, scorer):
    """A sparse brick list (hashed-voxel-map wire format) must stage the same grid as the dense upload."""
    w = fs.synth.make_workload("C1", n_cand=120)
    _setup_scorer(scorer, w)
    dense = scorer.score_arrival(w.goals, w.frontier_size, w.blacklisted)
    nz, ny, nx = w.cells.shape
    b = w.cells.reshape(nz // 8, 8, ny // 8, 8, nx // 8, 8).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 512)
    bz, by, bx = np.meshgrid(np.arange(nz // 8), np.arange(ny // 8), np.arange(nx // 8), indexing="ij")
    xyz = np.stack([bx.ravel(), by.ravel(), bz.ravel()], axis=1)
    keep = ~(b == 255).all(axis=1)                     # all-unknown bricks are simply absent
    assert 0 < keep.sum() < keep.size
    scorer.upload_grid_bricks(w.cells.shape, w.origin, w.resolution, xyz[keep], b[keep], default_value=255)
    sparse = scorer.score_arrival(w.goals, w.frontier_size, w.blacklisted)
    for k in ("ray_counts", "arrival", "argmax", "achievable", "status"):
        np.testing.assert_array_equal(sparse[k], dense[k])
    with pytest.raises(fs.FsError):
        scorer.upload_grid_bricks(w.cells.shape, w.origin, w.resolution, [[nx // 8, 0, 0]], b[:1])


def test_information_frontier_pair(fs, oracle, scorer):
    w = fs.synth.make_workload("C1", n_cand=30)
    scorer.upload_landmarks(w.landmarks)
    rng = np.random.default_rng(31)
    poses = oracle.poses_from_yaw(w.goals, rng.uniform(-3, 3, size=30))
    tris = []
    for g, y in zip(w.goals, rng.uniform(-3, 3, size=30)):
        a = g[:2]
        b = a + 2.5 * np.array([np.cos(y - 0.5), np.sin(y - 0.5)])
        c = a + 2.5 * np.array([np.cos(y + 0.5), np.sin(y + 0.5)])
        tris.append([*a, *b, *c])
    got = scorer.information_frontier_pair(poses, tris)
    want = np.array([oracle.information_frontier_pair(w.landmarks, poses[i], tris[i]) for i in range(30)])
    assert (want > 0).sum() > 10
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def _yaw_poses(xy, yaw):
    p = np.zeros((len(yaw), 7))
    p[:, :2] = xy
    p[:, 5] = np.sin(np.asarray(yaw) / 2)
    p[:, 6] = np.cos(np.asarray(yaw) / 2)
    return p


def test_information_for_pose_matches_oracle(fs, oracle, scorer):
    """Row a24: counts bit-exact, information within 1e-4 of the float64 form (and of the reference's float32 sum)."""
    w = fs.synth.make_small_2d(21, n=128, n_cand=200, n_landmarks=4000)
    kf_pose, off, pts = fs.synth.make_keyframes(w, 300, seed=5, points_per_kf=250, reach=3.0)   # > 256: two key-frame rounds
    g = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    scorer.upload_grid(w.cells, w.origin, w.resolution)
    scorer.upload_keyframes(kf_pose, off, pts)
    rng = np.random.default_rng(2)
    poses = _yaw_poses(w.goals[:, :2], rng.uniform(-np.pi, np.pi, size=w.goals.shape[0]))
    for kw in (dict(), dict(radius=-1.0), dict(max_depth=1.0, hfov=0.6, max_depth_error=0.0, q_diag=1.0, radius=2.0)):
        got = scorer.information_for_pose(poses, **kw)
        want = oracle.information_for_pose(g, poses, kf_pose, off, pts, n_threads=4, **kw)
        assert want["n_points"].sum() > 1000
        np.testing.assert_array_equal(got["n_cells"], want["n_cells"])
        np.testing.assert_array_equal(got["n_points"], want["n_points"])
        np.testing.assert_allclose(got["information"], want["info_f64"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got["information"], want["info_ref"], rtol=1e-4, atol=1e-6)
    # no key-frames at all
    scorer.upload_keyframes(np.zeros((0, 7)), [0], np.zeros((0, 3), np.float32))
    e = scorer.information_for_pose(poses[:5])
    assert not e["information"].any() and not e["n_cells"].any()


def test_information_for_pose_hbm_table_pass(fs, oracle, scorer):
    """A FOV triangle covering more cells than the LDS table holds is redone with the HBM table: same results."""
    n = 256
    cells = np.zeros((1, n, n), np.uint8)
    origin = (-6.4, -6.4, 0.0)
    rng = np.random.default_rng(9)
    pts = np.concatenate([rng.uniform(-6.3, 6.3, size=(60000, 2)), rng.uniform(0.0, 2.0, size=(60000, 1))], axis=1).astype(np.float32)
    kf_pose = _yaw_poses(rng.uniform(-3, 3, size=(30, 2)), rng.uniform(-np.pi, np.pi, size=30))
    off = np.arange(0, 60001, 2000, dtype=np.int32)
    g = oracle.Grid(cells, origin=origin, resolution=0.05)
    scorer.upload_grid(cells, origin, 0.05)
    scorer.upload_keyframes(kf_pose, off, pts)
    poses = _yaw_poses(rng.uniform(-2, 2, size=(12, 2)), rng.uniform(-np.pi, np.pi, size=12))
    kw = dict(max_depth=6.0, hfov=2.0, max_depth_error=0.5, radius=-1.0)
    got = scorer.information_for_pose(poses, **kw)
    want = oracle.information_for_pose(g, poses, kf_pose, off, pts, n_threads=4, **kw)
    assert want["n_cells"].max() > 3072                       # beyond 3/4 of the 4096-slot LDS table
    np.testing.assert_array_equal(got["n_cells"], want["n_cells"])
    np.testing.assert_array_equal(got["n_points"], want["n_points"])
    np.testing.assert_allclose(got["information"], want["info_f64"], rtol=1e-4, atol=1e-6)


def test_grid_layouts_give_identical_walks(fs, oracle, scorer):
    """The row-major byte walk and the walk over the 2-bit class image (16 x 16 x 2-cell tiles) visit the same cells:
    bit-exact against the oracle and against each other, for short and long rays, 2-D and 3-D grids; arbitrary segments on
    odd grid sizes."""
    rng = np.random.default_rng(77)
    try:
        for name, depth_cells in (("C1_2D", 40), ("C1", 40), ("C2", 160)):
            w = fs.synth.make_workload(name, n_cand=400)
            w.max_camera_depth = depth_cells * w.resolution
            G, P = _oracle_grid_params(oracle, w)
            _setup_scorer(scorer, w)
            mx_o = oracle.max_arrival_information(G, P)
            want = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx_o["min_gt"], n_threads=8)
            for layout in (1, 2, 3, 0):                            # 1 row-major byte image, 2 class image, 3 sparse class image (brick table + pool), 0 chosen by ray length
                scorer.set_option("ray.layout", layout)
                assert scorer.max_arrival() == mx_o
                _assert_arrival_equal(scorer.score_arrival(w.goals, w.frontier_size, w.blacklisted), want)
        # odd sizes: nx, ny not multiples of 8, nz odd
        cells = rng.choice(np.array([0, 0, 0, 255, 255, 254, 250], np.uint8), size=(5, 37, 43))
        origin = (-1.0, -0.9, -0.1)
        G = oracle.Grid(cells, origin=origin, resolution=0.05)
        scorer.upload_grid(cells, origin, 0.05)
        lo = np.array(origin); hi = lo + np.array([43, 37, 5]) * 0.05
        a = rng.uniform(lo - 0.05, hi + 0.05, size=(500, 3)); b = rng.uniform(lo - 0.05, hi + 0.05, size=(500, 3))
        res = {}
        for layout in (1, 2):                                      # (segments always walk the byte image: the option must not matter)
            scorer.set_option("ray.layout", layout)
            res[layout] = scorer.trace_segments(a, b, 200, obst=(254, 254), trace=(0, 255))
        for k in ("ok", "traced", "hit", "unknown", "all"):
            np.testing.assert_array_equal(res[1][k], res[2][k])
        for i in range(0, 500, 5):
            r = oracle.trace_ray(G, a[i], b[i], 200, obst=(254, 254), trace=(0, 255), faithful=True)
            assert bool(res[2]["ok"][i]) == r["ok"]
            if r["ok"]:
                assert (res[2]["traced"][i], bool(res[2]["hit"][i]), res[2]["unknown"][i], res[2]["all"][i]) == \
                    (r["traced"], r["hit"], r["unknown"], r["all"]), i
    finally:
        scorer.set_option("ray.layout", 0)


def test_class_image_walk_equals_the_row_major_walk(fs, oracle, scorer):
    """The class-image walk (WalkClass): per-ray counts, arrival, argmax, yaw, achievability and status identical to the
    oracle and to the row-major walk — grids whose sides are not multiples of the 16 x 16 x 2 tile, goals at the map border
    and off the map, depths 1..64, a z slice of a 3-D grid, visitor ranges that overlap (a cell in both classes), and a change
    of the ranges between calls (the image must be cut again)."""
    rng = np.random.default_rng(301)
    try:
        for nx, ny, nz, depth in ((101, 67, 1, 40), (96, 96, 1, 64), (50, 131, 1, 7), (64, 64, 3, 40), (33, 35, 1, 1)):
            cells = rng.choice(np.array([0, 0, 0, 0, 255, 255, 255, 254, 253, 240], np.uint8), size=(nz, ny, nx))
            origin = (-0.05 * nx / 2, -0.05 * ny / 2, 0.0)
            n = 300
            goals = np.zeros((n, 3))
            goals[:, 0] = rng.uniform(origin[0] - 0.1, origin[0] + nx * 0.05 + 0.1, size=n)      # some off the map
            goals[:, 1] = rng.uniform(origin[1] - 0.1, origin[1] + ny * 0.05 + 0.1, size=n)
            goals[:40, 0] = origin[0] + rng.uniform(0.0, 0.2, size=40)                            # hugging the borders
            goals[40:80, 1] = origin[1] + ny * 0.05 - rng.uniform(0.0, 0.2, size=40)
            goals[:, 2] = rng.integers(0, nz, size=n) * 0.05 + 0.01
            fsize = rng.integers(1, 31, size=n).astype(np.int32)
            black = (rng.random(n) < 0.03).astype(np.uint8)
            for obst, trace in (((240, 254), (255, 255)), ((250, 255), (0, 255))):
                kw = dict(max_camera_depth=depth * 0.05, delta_theta=0.10, camera_fov=1.04, robot_radius=0.60, n_rays=0, elev=(0.0,),
                          obst=obst, trace=trace, polygon=(origin[0] + 0.3, origin[1] + 0.1, origin[0] + nx * 0.05 - 0.2, origin[1] + ny * 0.05 - 0.3))
                G = oracle.Grid(cells, origin=origin, resolution=0.05)
                P = oracle.RayParams(**kw)
                scorer.set_ray_params(**kw)
                scorer.upload_grid(cells, origin, 0.05)
                mx_o = oracle.max_arrival_information(G, P)
                want = oracle.arrival_information(G, P, goals, fsize, black, min_gt=mx_o["min_gt"], faithful=True)
                res = {}
                for layout in (2, 1):
                    scorer.set_option("ray.layout", layout)
                    assert scorer.max_arrival() == mx_o
                    res[layout] = scorer.score_arrival(goals, fsize, black, want_ray_counts=True)
                    _assert_arrival_equal(res[layout], want)
                    np.testing.assert_array_equal(res[layout]["yaw"], want["yaw"])
                    np.testing.assert_array_equal(res[layout]["ray_counts"], want["ray_counts"])
                assert (want["status"] == 1).sum() > 0 and (want["arrival"] > 0).sum() > 0
    finally:
        scorer.set_option("ray.layout", 0)


def test_fim_table_with_holes(fs, oracle, scorer, ref_table):
    """A loaded table may miss keys (the reference's lookup returns NaN and the landmark is skipped, no voxel is
    counted: FisherInfoManager.cpp:90-94,287-324).  Exercises the kernels compiled for tables with holes."""
    rng = np.random.default_rng(17)
    rec = ref_table.records
    keep = rng.random(rec.shape[0]) < 0.7
    keep[-1] = True                                           # the trailing (0,0,0) record
    holey = np.ascontiguousarray(rec[keep])
    w = fs.synth.make_workload("C1", n_cand=150)
    poses = oracle.poses_from_yaw(w.goals, rng.uniform(-np.pi, np.pi, size=150))
    _setup_scorer(scorer, w)
    try:
        scorer.lookup_set_records(holey)
        table = oracle.Table.from_records(holey)
        want = oracle.pose_information(table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
        full = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
        assert (want["n_voxels"] < full["n_voxels"]).sum() > 100      # the holes matter
        got = scorer.score_fim(poses)
        _check_fim(got, want, w.landmarks.shape[0])
        # the fused path uses the same table
        G, P = _oracle_grid_params(oracle, w)
        mx = scorer.max_arrival()
        rec_gpu = scorer.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        arr = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], n_threads=8)
        ok = arr["status"] == 0
        fim = oracle.pose_information(table, w.landmarks, oracle.poses_from_yaw(w.goals, arr["yaw"]), 14.0, 1.0, n_threads=8)
        np.testing.assert_array_equal(rec_gpu["n_visible"][ok], fim["n_visible"][ok])
        np.testing.assert_array_equal(((rec_gpu["flags"] >> 16) & 0xFFFF)[ok], fim["n_voxels"][ok])
        sc = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
        assert np.max(np.abs(rec_gpu["info_ref"][ok] - fim["info_f64"][ok]) / sc) <= REL
    finally:
        scorer.lookup_generate()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 5000, 32768, 32769, 70000])
def test_rank_order_is_the_stable_ascending_sort_at_every_size(fs, oracle, scorer, n):
    """Up to 1024 candidates ONE workgroup ranks the list in one launch (fs_rank_small_kernel: normalisation, costs, a bitonic
    network over (cost image, index) pairs in LDS), beyond that rocPRIM's sort: both must be numpy's stable argsort of the
    (bit-exact) costs — with thousands of equal costs (the unachievable and blacklisted candidates all carry DBL_MAX, and quantised
    path lengths make ties among the live ones)."""
    rng = np.random.default_rng(n)
    rec = np.zeros(n, dtype=fs.capi.RECORD_DTYPE)
    rec["arrival"] = rng.integers(0, 400, size=n)
    ach = rng.random(n) < 0.7
    rec["flags"] = ach.astype(np.uint32)
    black = (rng.random(n) < 0.05).astype(np.uint8)
    plen = np.round(rng.uniform(0.5, 30.0, size=n), 0)               # whole metres: many exact ties
    phead = np.round(rng.uniform(0.0, 3.0, size=n), 1)
    scorer.set_arrival_limits(400.0, 40.0)
    rc, want = oracle.u1_costs(rec["arrival"].astype(np.float64), ach.astype(np.uint8), plen, phead, 400.0, blacklisted=black)
    assert rc == 0
    got = scorer.rank_candidates(rec, plen, phead, blacklisted=black)
    np.testing.assert_array_equal(got["weighted_cost"], want["weighted_cost"])
    np.testing.assert_array_equal(got["order"], np.argsort(want["weighted_cost"], kind="stable"))
