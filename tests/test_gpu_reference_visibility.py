"""The visibility the reference itself requests — 14 m, observation-angle filter off (max_angle 4.0 > pi,
FIP/src/fisher_information/FisherInfoManager.cpp:63-64) — through every FIM worker instantiation that can serve it:

  general worker, cone off        fs_score_fim with every output column / fs_score_candidates with "fim.specialise" 0
  INFO_ONLY worker, cone off      fs_score_fim asked for info_ref (+ n_voxels) alone: what isPoseSafe reads (:83-100)
  YAW_ONLY worker, cone off       fs_score_candidates (poses are rotations about Z)

against the oracle: n_voxels bit-exact, info_ref within 1e-4 relative, on C1, C2, a sample of C3 and the three clouds the
reference's own manual programs hold.  The same at the build's 1.0 rad cone (INFO_ONLY / YAW_ONLY with the narrow cone)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-4
HERE = os.path.dirname(os.path.abspath(__file__))
REF_REQUEST = (14.0, 4.0)


def _check_info(got, want):
    np.testing.assert_array_equal(got["n_voxels"], want["n_voxels"])
    scale = np.maximum(np.abs(want["info_f64"]), 1e-6)
    assert np.max(np.abs(got["info_ref"] - want["info_f64"]) / scale) <= REL


def _both_workers(scorer, poses, want):
    """info-only call through the INFO_ONLY worker and through the general one; full call through the general one"""
    out = {}
    for special in (1, 0):
        scorer.set_option("fim.specialise", special)
        out[special] = scorer.score_fim(poses, info_only=True)
        _check_info(out[special], want)
    scorer.set_option("fim.specialise", 1)
    full = scorer.score_fim(poses, want_fim=False)
    np.testing.assert_array_equal(full["n_visible"], want["n_visible"])
    _check_info(full, want)
    # the two workers add the same terms in another order: last-bit differences only
    np.testing.assert_allclose(out[1]["info_ref"], out[0]["info_ref"], rtol=5e-6, atol=1e-6)
    return out


@pytest.mark.parametrize("name,n_pose", [("C1", 200), ("C2", 300), ("C3", 96)])
@pytest.mark.parametrize("vis", [REF_REQUEST, (14.0, 1.0)])
def test_info_only_worker_matches_oracle(fs, oracle, scorer, ref_table, name, n_pose, vis):
    w = fs.synth.make_workload(name, n_cand=n_pose)
    rng = np.random.default_rng(41)
    poses = oracle.poses_from_yaw(w.goals, rng.uniform(-np.pi, np.pi, size=n_pose))
    q = rng.normal(size=(6, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses[:6, 3:] = q                                               # a few general orientations (box cull in a tilted frame)
    scorer.upload_landmarks(w.landmarks)
    scorer.set_fim_params(*vis)
    want = oracle.pose_information(ref_table, w.landmarks, poses, vis[0], vis[1], n_threads=8)
    assert want["n_visible"].max() > 0
    _both_workers(scorer, poses, want)
    scorer.set_fim_params(14.0, 1.0)


def test_info_only_box_cull_removes_work_not_terms(fs, oracle, scorer, ref_table):
    """With the cone off the range sphere alone keeps every chunk within 14 m; the INFO_ONLY worker also culls against the
    table's box in the camera frame (x >= -0.15 m: nothing behind the camera can hit a voxel of the table).  Fewer landmark
    tests, identical voxels, same information."""
    w = fs.synth.make_workload("C2", n_cand=300)
    rng = np.random.default_rng(43)
    poses = oracle.poses_from_yaw(w.goals, rng.uniform(-np.pi, np.pi, size=300))
    lm = w.landmarks[::8]                                           # 6 k landmarks: one scoring pass per pose in either worker
    scorer.upload_landmarks(lm)
    scorer.set_fim_params(*REF_REQUEST)
    tested = {}
    res = {}
    for special in (1, 0):
        scorer.set_option("fim.specialise", special)
        scorer.get_counter(0, reset=True)
        res[special] = scorer.score_fim(poses, info_only=True)
        tested[special] = scorer.get_counter(0, reset=True)
    scorer.set_option("fim.specialise", 1)
    scorer.set_fim_params(14.0, 1.0)
    np.testing.assert_array_equal(res[1]["n_voxels"], res[0]["n_voxels"])
    _check_info(res[1], oracle.pose_information(ref_table, lm, poses, *REF_REQUEST, n_threads=8))
    assert tested[1] < 0.75 * tested[0], tested


@pytest.mark.parametrize("tag", ["mfc", "viz", "cone"])
def test_reference_held_clouds_at_the_reference_request(fs, oracle, scorer, ref_table, tag):
    z = np.load(os.path.join(HERE, "golden", "ref_held_inputs.npz"))
    if tag == "cone":
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_reference_inputs", os.path.join(HERE, "golden", "make_reference_inputs.py"))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        g, _ = gen.cone_lattice()
        lm = g[gen.reference_cone_mask(g)][::7].astype(np.float32)      # every 7th point of LoadLookupMain's sweep (142 k)
        pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
    else:
        pose, lm = z[f"{tag}_pose7"], z[f"{tag}_landmarks"]
    scorer.upload_landmarks(lm)
    scorer.set_fim_params(*REF_REQUEST)
    want = oracle.pose_information(ref_table, lm, pose[None], *REF_REQUEST, n_threads=8)
    _both_workers(scorer, pose[None], want)
    scorer.set_fim_params(14.0, 1.0)


@pytest.mark.parametrize("name", ["C1", "C2"])
@pytest.mark.parametrize("vis", [REF_REQUEST, (14.0, 1.0)])
def test_fused_path_yaw_only_worker(fs, oracle, scorer, ref_table, name, vis):
    """fs_score_candidates builds every pose as a rotation about Z and takes the YAW_ONLY worker (7-instruction transform);
    "fim.specialise" 0 sends the same call through the general one.  Identical arithmetic, and the oracle as in
    test_fused_candidates."""
    w = fs.synth.make_workload(name, n_cand=600 if name == "C2" else None)
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    scorer.set_ray_params(**kw)
    scorer.upload_grid(w.cells, w.origin, w.resolution)
    scorer.upload_landmarks(w.landmarks)
    scorer.set_fim_params(*vis)
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(**kw)
    mx = oracle.max_arrival_information(G, P)
    assert scorer.max_arrival() == mx
    recs = {}
    scorer.set_option("fim.learn", 0)       # the pass count — hence the order of the float sums — must not depend on the call before
    for special in (1, 0):
        scorer.set_option("fim.specialise", special)
        recs[special] = scorer.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    scorer.set_option("fim.specialise", 1)
    scorer.set_option("fim.learn", 1)
    scorer.set_fim_params(14.0, 1.0)
    # same arithmetic: everything that does not depend on which landmark won which rank in its voxel (the order of the LDS
    # atomics decides which lane adds which info_v * factor(k) term) agrees bit for bit; info_ref to the last bits
    for k in ("arrival", "argmax", "yaw", "trace", "logdet", "n_visible", "flags"):
        assert recs[1][k].tobytes() == recs[0][k].tobytes(), k
    np.testing.assert_allclose(recs[1]["info_ref"], recs[0]["info_ref"], rtol=5e-6, atol=1e-6)
    arr = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], n_threads=8)
    fim = oracle.pose_information(ref_table, w.landmarks, oracle.poses_from_yaw(w.goals, arr["yaw"]), vis[0], vis[1], n_threads=8)
    rec, ok = recs[1], arr["status"] == 0
    np.testing.assert_array_equal(rec["arrival"], arr["arrival"])
    np.testing.assert_array_equal(rec["n_visible"][ok], fim["n_visible"][ok])
    np.testing.assert_array_equal(fs.capi.record_nvoxels(rec)[ok], np.minimum(fim["n_voxels"][ok], 65535))
    scale = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
    assert np.max(np.abs(rec["info_ref"][ok] - fim["info_f64"][ok]) / scale) <= REL
    assert np.max(np.abs(rec["trace"][ok] - fim["trace"][ok]) / np.maximum(fim["trace"][ok], 1e-6)) <= REL


def test_info_only_box_cull_follows_the_table_it_was_given(fs, oracle, scorer):
    """The INFO_ONLY worker culls against the box of whatever lookup table is loaded, not against the reference's numbers: a small
    table that reaches BEHIND the camera (x from -3 m) and is narrow in z, general orientations, cone off and on."""
    bounds = np.array([-3.0, 6.0, -4.0, 4.0, -2.0, 2.0], dtype=np.float32)
    table = oracle.Table.generate(bounds)
    scorer.lookup_generate(bounds)
    try:
        w = fs.synth.make_workload("C2", n_cand=200)
        rng = np.random.default_rng(47)
        poses = oracle.poses_from_yaw(w.goals, rng.uniform(-np.pi, np.pi, size=200))
        q = rng.normal(size=(100, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        poses[:100, 3:] = q
        scorer.upload_landmarks(w.landmarks)
        for vis in (REF_REQUEST, (14.0, 1.0), (5.0, 4.0)):
            scorer.set_fim_params(*vis)
            want = oracle.pose_information(table, w.landmarks, poses, vis[0], vis[1], n_threads=8)
            assert want["n_voxels"].max() > 50
            _both_workers(scorer, poses, want)
    finally:
        scorer.lookup_generate()
        scorer.set_fim_params(14.0, 1.0)
