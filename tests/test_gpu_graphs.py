"""The small host-buffer calls replay a captured launch graph (fs_capi.hip, run_maybe_graphed): first call plain, second call
captured, later calls one hipGraphLaunch.  What must hold: the same results as plain launches for every list length around the
bucket edges, across repeated calls, and — above all — after anything the graph has baked in has changed (map, cloud, limits,
parameters, buffer growth): a stale graph would silently score the old state.  Plus fs_get_frontier_costs
(CostAssigner::getFrontierCosts as one call) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _graphs_off_afterwards(scorer):
    yield
    scorer.set_option("graph", 0)


INT_COLS = ("arrival", "argmax", "n_visible", "flags")


def _kw(w):
    return dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)


def _stage(sc, w):
    sc.set_option("graph", 1)                # (off by default: measured slower than plain launches on this runtime)
    sc.set_ray_params(**_kw(w))
    sc.upload_grid(w.cells, w.origin, w.resolution)
    sc.upload_landmarks(w.landmarks)
    sc.set_fim_params(14.0, 1.0)
    sc.set_arrival_limits(400.0, 40.0)


def _same(a, b):
    for k in INT_COLS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(a["yaw"], b["yaw"])
    np.testing.assert_allclose(a["info_ref"], b["info_ref"], rtol=5e-6, atol=1e-6)
    np.testing.assert_allclose(a["trace"], b["trace"], rtol=5e-6, atol=1e-6)


@pytest.mark.parametrize("n", [1, 2, 3, 37, 64, 65, 1000, 1024, 1025])
def test_graphed_score_candidates_equals_plain_launches(fs, scorer, n):
    w = fs.synth.make_workload("C2", n_cand=1025)
    _stage(scorer, w)
    g, f, b = w.goals[:n], w.frontier_size[:n], w.blacklisted[:n]
    scorer.set_option("graph", 0)
    want = scorer.score_candidates(g, f, b)
    scorer.set_option("graph", 1)
    for _ in range(4):                                               # plain, capture, replay, replay
        _same(scorer.score_candidates(g, f, b), want)
    # another list of the same length through the same graph; optional columns left out
    g2 = w.goals[1025 - n:1025]
    scorer.set_option("graph", 0)
    want2 = scorer.score_candidates(g2)
    scorer.set_option("graph", 1)
    for _ in range(2):
        _same(scorer.score_candidates(g2), want2)


def test_a_graph_never_outlives_the_state_it_was_captured_in(fs, oracle, scorer, ref_table):
    w = fs.synth.make_workload("C2", n_cand=50)
    w1 = fs.synth.make_workload("C1", n_cand=50)
    _stage(scorer, w)
    g, f, b = w.goals, w.frontier_size, w.blacklisted

    def both():
        got = [scorer.score_candidates(g, f, b) for _ in range(3)][-1]      # replayed graph by the third call
        scorer.set_option("graph", 0)
        want = scorer.score_candidates(g, f, b)
        scorer.set_option("graph", 1)
        _same(got, want)
        return got

    base = both()
    scorer.set_arrival_limits(400.0, 390.0)                                   # min_gt is a kernel argument
    r = both()
    assert (fs.capi.record_achievable(r) != fs.capi.record_achievable(base)).any()
    scorer.set_arrival_limits(400.0, 40.0)
    scorer.upload_landmarks(w.landmarks[::3])                                 # another cloud (new chunk count, new pointers)
    r = both()
    assert (r["n_visible"] != base["n_visible"]).any()
    scorer.set_fim_params(6.0, 0.6)                                           # another visibility volume
    r2 = both()
    assert (r2["n_visible"] < r["n_visible"]).any()
    scorer.set_fim_params(14.0, 1.0)
    cells = w.cells.copy(); cells[:, :, : cells.shape[2] // 2] = 0           # another map, same shape
    scorer.upload_grid(cells, w.origin, w.resolution)
    r3 = both()
    assert (r3["arrival"] != base["arrival"]).any()
    # a big call in between grows (reallocates) the per-candidate buffers the small graph points into
    big = fs.synth.make_workload("C2", n_cand=5000)
    scorer.upload_grid(w.cells, w.origin, w.resolution); scorer.upload_landmarks(w.landmarks)
    both()
    scorer.score_candidates(big.goals, big.frontier_size, big.blacklisted)
    _same(both(), base)
    # and another workload altogether (other grid shape, other fan)
    _stage(scorer, w1)
    g, f, b = w1.goals, w1.frontier_size, w1.blacklisted
    both()


def test_graphed_one_pose_fim_follows_the_pose(fs, oracle, scorer, ref_table):
    w = fs.synth.make_workload("C2", n_cand=40)
    _stage(scorer, w)
    poses = oracle.poses_from_yaw(w.goals, np.linspace(-3, 3, 40))
    want = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 1.0, n_threads=8)
    for i in range(40):                                              # one graph, forty different poses through it
        r = scorer.score_fim(poses[i:i + 1], want_fim=False)
        assert r["n_visible"][0] == want["n_visible"][i] and r["n_voxels"][0] == want["n_voxels"][i]
        assert abs(r["info_ref"][0] - want["info_f64"][i]) <= 1e-4 * max(abs(want["info_f64"][i]), 1e-6)
        r = scorer.score_fim(poses[i:i + 1], info_only=True)
        assert r["n_voxels"][0] == want["n_voxels"][i]
        assert abs(r["info_ref"][0] - want["info_f64"][i]) <= 1e-4 * max(abs(want["info_f64"][i]), 1e-6)
    bad = poses[:1].copy(); bad[:, 3:] *= 1.2                         # non-unit quaternion: brute-force variant of the graph
    r = scorer.score_fim(bad, want_fim=False)
    wb = oracle.pose_information(ref_table, w.landmarks, bad, 14.0, 1.0)
    assert r["n_visible"][0] == wb["n_visible"][0]
    scorer.set_option("graph", 0)


@pytest.mark.parametrize("n,with_fim", [(1, False), (50, False), (50, True), (700, True), (3000, False), (3000, True)])
def test_get_frontier_costs_matches_oracle(fs, oracle, scorer, ref_table, n, with_fim):
    """arrival information (+ FI) + U1 + order in one call == oracle arrival + oracle U1 on the same path columns"""
    w = fs.synth.make_workload("C2", n_cand=n)
    _stage(scorer, w)
    scorer.set_arrival_limits(4000.0, 40.0)                           # (C2's four rings reach window sums beyond 400)
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(**_kw(w))
    arr = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=40.0, n_threads=8)
    rng = np.random.default_rng(n)
    plen, phead = rng.uniform(0.5, 30.0, size=n), rng.uniform(0.0, np.pi, size=n)
    rc, want = oracle.u1_costs(arr["arrival"].astype(np.float64), arr["achievable"], plen, phead, 4000.0, blacklisted=w.blacklisted)
    assert rc == 0
    for _ in range(3):
        got = scorer.get_frontier_costs(w.goals, plen, phead, w.frontier_size, w.blacklisted, with_fim=with_fim)
        rec = got["records"]
        np.testing.assert_array_equal(rec["arrival"], arr["arrival"])
        np.testing.assert_array_equal(rec["argmax"], arr["argmax"])
        np.testing.assert_array_equal(rec["yaw"], arr["yaw"].astype(np.float32))
        np.testing.assert_array_equal(fs.capi.record_status(rec), arr["status"])
        np.testing.assert_array_equal(fs.capi.record_achievable(rec), arr["achievable"])
        for k in ("weighted_cost", "arrival_utility", "distance_utility"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
        np.testing.assert_array_equal(got["order"], np.argsort(want["weighted_cost"], kind="stable"))
        if with_fim:
            _same(rec, scorer.score_candidates(w.goals, w.frontier_size, w.blacklisted))
        else:
            assert not rec["info_ref"].any() and not rec["n_visible"].any()
    # where the reference throws (utility outside [0, 1]) the call reports FS_E_RANGE
    scorer.set_arrival_limits(1.0, 0.1)
    if (arr["arrival"][(arr["achievable"] == 1) & (w.blacklisted == 0)] > 1).any():
        with pytest.raises(fs.capi.FsError) as e:
            scorer.get_frontier_costs(w.goals, plen, phead, w.frontier_size, w.blacklisted, with_fim=with_fim)
        assert e.value.code == fs.capi.FS_E_RANGE
    scorer.set_arrival_limits(400.0, 40.0)


def test_in_place_staging_equals_transfers(fs, oracle, scorer, ref_table):
    """Up to 1024 candidates / poses a host-buffer call reads its inputs from, and writes its results into, the mapped page-locked
    staging buffers ("zerocopy" 1, the default); 0 moves them with transfers as every larger call does.  Same results."""
    w = fs.synth.make_workload("C2", n_cand=900)
    _stage(scorer, w)
    scorer.set_option("graph", 0)
    scorer.set_arrival_limits(4000.0, 40.0)
    g, f, b = w.goals, w.frontier_size, w.blacklisted
    poses = oracle.poses_from_yaw(w.goals[:50], np.linspace(-3, 3, 50))
    rng = np.random.default_rng(1)
    plen, phead = rng.uniform(0.5, 30.0, size=900), rng.uniform(0.0, np.pi, size=900)
    out = {}
    for z in (1, 0):
        scorer.set_option("zerocopy", z)
        out[z] = dict(rec=scorer.score_candidates(g, f, b), arr=scorer.score_arrival(g, f, b), fim=scorer.score_fim(poses),
                      io=scorer.score_fim(poses, info_only=True), costs=scorer.get_frontier_costs(g, plen, phead, f, b, with_fim=True),
                      one=scorer.score_candidates(g[:1], f[:1], b[:1]))
    scorer.set_option("zerocopy", 1)
    _same(out[1]["rec"], out[0]["rec"]); _same(out[1]["one"], out[0]["one"]); _same(out[1]["costs"]["records"], out[0]["costs"]["records"])
    for k in ("arrival", "argmax", "status", "achievable", "yaw", "ray_counts"):
        np.testing.assert_array_equal(out[1]["arr"][k], out[0]["arr"][k], err_msg=k)
    for k in ("n_visible", "n_voxels"):
        np.testing.assert_array_equal(out[1]["fim"][k], out[0]["fim"][k])
    np.testing.assert_array_equal(out[1]["io"]["n_voxels"], out[0]["io"]["n_voxels"])
    for k in ("info_ref", "trace", "logdet", "fim21"):
        np.testing.assert_allclose(out[1]["fim"][k], out[0]["fim"][k], rtol=5e-6, atol=1e-5)
    np.testing.assert_allclose(out[1]["io"]["info_ref"], out[0]["io"]["info_ref"], rtol=5e-6, atol=1e-6)
    for k in ("weighted_cost", "arrival_utility", "distance_utility", "order"):
        np.testing.assert_array_equal(out[1]["costs"][k], out[0]["costs"][k], err_msg=k)
