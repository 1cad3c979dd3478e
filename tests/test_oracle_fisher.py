"""Oracle tests for the Fisher-information half: pose construction, visibility, the per-voxel
crowding bookkeeping (FisherInfoManager.cpp:287-324) against the pure-Python transcription, and the
order-independent form of SURVEY.md App. A.2."""
import math

import numpy as np
import pytest


def test_pose_to_rt_and_yaw_quat(oracle, pyref):
    for yaw in (0.0, 0.52, -2.4, 3.1):
        q = oracle.yaw_to_quat(yaw)
        assert q[0] == 0 and q[1] == 0 and q[2] == math.sin(yaw * 0.5) and q[3] == math.cos(yaw * 0.5)
        R, t = oracle.pose_to_rt([1.5, -2.25, 0.5, *q])
        np.testing.assert_array_equal(R, pyref.quat_to_rot_f32(q))
        np.testing.assert_allclose(R, [[math.cos(yaw), -math.sin(yaw), 0], [math.sin(yaw), math.cos(yaw), 0], [0, 0, 1]], atol=2e-7)
        np.testing.assert_array_equal(t, np.array([1.5, -2.25, 0.5], np.float32))


def test_world_to_camera_and_visibility(oracle, pyref):
    rng = np.random.default_rng(2)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    pose = [0.3, -0.7, 1.1, *q]
    R, t = oracle.pose_to_rt(pose)
    for _ in range(300):
        w = rng.uniform(-12, 12, size=3).astype(np.float32)
        p = oracle.world_to_camera(R, t, w)
        np.testing.assert_array_equal(p, pyref.world_to_camera(R, t, w))
        for md, ma in ((14.0, 1.0), (6.0, 2.2), (14.0, 4.0), (9.0, math.pi / 2)):
            assert oracle.is_visible(p, md, ma) == pyref.is_visible(p, md, ma)
            # semantic check away from the boundaries
            n = float(np.linalg.norm(p.astype(np.float64)))
            ang = math.acos(max(-1.0, min(1.0, float(p[0]) / n))) if n > 0 else 0.0
            if abs(n - md) > 1e-3 and (ma >= math.pi or abs(ang - ma) > 1e-3):
                assert oracle.is_visible(p, md, ma) == (n <= md and (ma >= math.pi or ang <= ma))


@pytest.mark.parametrize("max_dist,max_angle", [(14.0, 1.0), (5.0, 2.0), (14.0, 4.0)])
def test_pose_information_matches_python_transcription(oracle, pyref, ref_table, max_dist, max_angle):
    rng = np.random.default_rng(7)
    # clustered landmarks so that voxels are shared (crowding ranks > 1) and some keys miss the table
    centres = rng.uniform(-6, 6, size=(12, 3))
    lm = (centres[rng.integers(0, 12, size=400)] + rng.normal(scale=0.25, size=(400, 3))).astype(np.float32)
    lm = np.concatenate([lm, np.array([[30, 0, 0], [0, 16.5, 0], [0.0, 0.0, 0.0]], np.float32)])
    poses = np.array([[0, 0, 0, 0, 0, 0, 1.0], [1.0, -2.0, 0.3, 0, 0, math.sin(0.4), math.cos(0.4)],
                      [-3.0, 2.0, -0.5, 0.1, -0.2, 0.3, 0.927]], dtype=np.float64)
    table = {tuple(float(v) + 0.0 for v in r[:3]): np.float32(r[3]) for r in ref_table.records}
    got = oracle.pose_information(ref_table, lm, poses, max_dist, max_angle)
    for i, pose in enumerate(poses):
        want = pyref.pose_information(table, lm, pose, max_dist, max_angle)
        assert got["n_visible"][i] == want["n_visible"] and got["n_voxels"][i] == want["n_voxels"]
        assert got["info_ref"][i] == want["info_ref"]            # same float32 sequential accumulation
        assert abs(got["info_f64"][i] - float(want["info_ref"])) <= 1e-5 * max(1.0, abs(got["info_f64"][i]))
    assert got["n_voxels"].max() < got["n_visible"].max()        # crowding was exercised


def test_order_independent_form(oracle, ref_table):
    """App. A.2: total = sum_v info_v * S(m_v); shuffling the landmarks changes only float rounding."""
    rng = np.random.default_rng(8)
    lm = rng.uniform(-8, 8, size=(3000, 3)).astype(np.float32)
    pose = np.array([[0.5, 0.5, 0.0, 0, 0, 0.2, 0.98]])
    a = oracle.pose_information(ref_table, lm, pose)
    b = oracle.pose_information(ref_table, lm[rng.permutation(3000)], pose)
    assert a["n_visible"][0] == b["n_visible"][0] and a["n_voxels"][0] == b["n_voxels"][0]
    assert abs(a["info_f64"][0] - b["info_f64"][0]) <= 1e-9 * a["info_f64"][0]
    assert abs(a["info_ref"][0] - b["info_ref"][0]) <= 1e-5 * a["info_f64"][0]
    assert abs(a["trace"][0] - b["trace"][0]) <= 1e-9 * a["trace"][0]
    np.testing.assert_allclose(a["fim"][0], a["fim"][0].T, atol=1e-9)
    assert abs(a["logdet"][0] - np.linalg.slogdet(a["fim"][0])[1]) < 1e-8


def test_threshold_decision(oracle, ref_table):
    """isPoseSafe returns total > threshold with threshold 550 (FisherInfoBTPlugin.cpp:20): the decision
    stays with the caller; here just the scalar's monotonicity in the landmark count."""
    rng = np.random.default_rng(9)
    lm = rng.uniform(-7, 7, size=(4000, 3)).astype(np.float32)
    pose = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    few = oracle.pose_information(ref_table, lm[:200], pose)["info_ref"][0]
    many = oracle.pose_information(ref_table, lm, pose)["info_ref"][0]
    assert many > few > 0


def test_global_jacobian_trace_and_frontier_pair(oracle):
    """a14/a16 of SURVEY.md 8(a): computeInformationOfPointGlobal and computeInformationFrontierPair."""
    rng = np.random.default_rng(12)
    pose = [0.4, -1.1, 0.2, 0, 0, np.sin(0.35), np.cos(0.35)]
    R, t = oracle.pose_to_rt(pose)
    for _ in range(30):
        w = rng.uniform(-6, 6, size=3).astype(np.float32)
        p = oracle.world_to_camera(R, t, w).astype(np.float64)
        n2 = p @ p
        # closed form of trace(J^T J), J = A R^T [I | -[w]x]:  tr(A^T A (I + [w]x^T [w]x)) with A^T A = P/n^2 in the camera frame
        P = (np.eye(3) - np.outer(p, p) / n2) / n2
        Rm = R.astype(np.float64)
        W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
        M = Rm @ P @ Rm.T
        want = np.trace(M) + np.trace(W.T @ M @ W)
        got = oracle.information_of_point_global_world(pose, w)
        assert abs(got - want) <= 2e-5 * want
    lm = rng.uniform(-5, 5, size=(500, 3)).astype(np.float32)
    tri = [0.0, 0.0, 4.0, -2.0, 4.0, 2.5]                      # counter-clockwise
    got = oracle.information_frontier_pair(lm, pose, tri)
    inside = [(tri[0] - x) * (tri[3] - y) - (tri[1] - y) * (tri[2] - x) > 0 and
              (tri[2] - x) * (tri[5] - y) - (tri[3] - y) * (tri[4] - x) > 0 and
              (tri[4] - x) * (tri[1] - y) - (tri[5] - y) * (tri[0] - x) > 0 for x, y in lm[:, :2].astype(np.float64)]
    want = sum(oracle.information_of_point_local_world(pose, lm[i]) for i in np.nonzero(inside)[0])
    assert sum(inside) > 10 and abs(got - want) <= 1e-5 * want
    assert oracle.information_frontier_pair(lm, pose, [0.0, 0.0, 4.0, 2.5, 4.0, -2.0]) == 0.0     # clockwise: nothing is "left"
