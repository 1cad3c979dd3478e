"""The landmark cloud's k-d leaf order computed on the device ("cloud.order" 2; the default, 1, takes it from 4096 landmarks on; fs_cloud.hip) against the host form
(fs_stage_landmarks): the order of the cloud is the library's own business — what must not change is what comes out.  Integer
columns (visible landmarks, voxels) are order-independent and must agree bit for bit; float columns follow the order of summation
and must agree within 1e-5 relative (north_star's tolerance is 1e-4); the chunk cull must test about as many landmarks (same split
rule, ties apart).  Checked against the oracle once as well, so that "both wrong in the same way" is excluded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(fs, lm, angle=1.0, multi=False):
    host = fs.FrontierScorer(device=0)
    dev = fs.MultiScorer([0, 0]) if multi else fs.FrontierScorer(device=0)
    host.set_option("cloud.order", 0)
    dev.set_option("cloud.order", 2)
    for s in (host, dev):
        s.upload_landmarks(lm)
        s.lookup_generate()
        s.set_fim_params(14.0, angle)
    return host, dev


def _poses(rng, n, lo, hi):
    p = np.zeros((n, 7))
    p[:, :3] = rng.uniform(lo, hi, size=(n, 3))
    q = rng.normal(size=(n, 4))
    q[: n // 2, :2] = 0.0                                   # half of them yaw-only
    p[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return p


def _same(a, b, what):
    np.testing.assert_array_equal(a["n_visible"], b["n_visible"], err_msg=what)
    np.testing.assert_array_equal(a["n_voxels"], b["n_voxels"], err_msg=what)
    for k in ("info_ref", "trace"):
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        assert np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-6), initial=0.0) <= 1e-5, (what, k)
    scale = np.maximum(np.abs(b["fim21"]).max(axis=1, keepdims=True), 1e-6)
    assert np.max(np.abs(a["fim21"] - b["fim21"]) / scale, initial=0.0) <= 1e-5, what


@pytest.mark.parametrize("name", ["C1", "C2", "REF2D"])
def test_device_order_scores_like_the_host_order(fs, oracle, ref_table, name):
    w = fs.synth.make_workload(name, n_cand=200)
    rng = np.random.default_rng(5)
    lo = np.asarray(w.origin) + 0.5
    hi = np.asarray(w.origin) + np.array([w.cells.shape[2], w.cells.shape[1], max(w.cells.shape[0], 40)]) * w.resolution - 0.5
    poses = _poses(rng, 160, lo, hi)
    for angle in (1.0, 4.0):
        host, dev = _pair(fs, w.landmarks, angle)
        try:
            host.get_counter(0, reset=True); dev.get_counter(0, reset=True)
            a, b = dev.score_fim(poses), host.score_fim(poses)
            _same(a, b, f"{name} angle {angle}")
            ta, tb = dev.get_counter(0), host.get_counter(0)
            assert abs(ta - tb) <= 0.02 * max(tb, 1), (name, angle, ta, tb)       # landmark tests behind the chunk cull
            c = dev.score_fim(poses[:40], info_only=True)
            np.testing.assert_array_equal(c["n_voxels"], b["n_voxels"][:40])
            if angle == 1.0:
                o = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, angle, n_threads=8)
                np.testing.assert_array_equal(a["n_visible"], o["n_visible"])
                np.testing.assert_array_equal(a["n_voxels"], o["n_voxels"])
                rel = np.abs(a["info_ref"] - o["info_f64"]) / np.maximum(np.abs(o["info_f64"]), 1e-6)
                assert rel.max() <= 1e-4
        finally:
            host.close(); dev.close()


@pytest.mark.parametrize("m", [0, 1, 63, 64, 65, 130, 1000, 4097, 70_000])
def test_device_order_on_every_cloud_size_and_with_unusable_and_tied_landmarks(fs, m):
    rng = np.random.default_rng(100 + m)
    lm = rng.uniform(-9.0, 9.0, size=(m, 3)).astype(np.float32)
    if m >= 64:
        lm[: m // 3, 0] = np.round(lm[: m // 3, 0])             # many ties along x
        lm[m // 2:, 2] = 1.0                                     # a plane: the z extent of many nodes is zero
        lm[rng.choice(m, size=max(1, m // 50), replace=False)] = np.array([np.nan, 1.0, 1.0], dtype=np.float32)
        lm[rng.choice(m, size=max(1, m // 70), replace=False), 1] = np.inf
        lm[rng.choice(m, size=max(1, m // 90), replace=False), 2] = 3.0e30
    poses = _poses(rng, 48, -8.0, 8.0)
    host, dev = _pair(fs, lm, 4.0)
    try:
        _same(dev.score_fim(poses), host.score_fim(poses), f"m {m}")
        # a second cloud on the same context (scratch grows / shrinks), then the first again
        other = rng.uniform(-5.0, 5.0, size=(max(1, 3 * m // 2), 3)).astype(np.float32)
        for s in (host, dev):
            s.upload_landmarks(other)
        _same(dev.score_fim(poses), host.score_fim(poses), f"m {m}: second cloud")
        for s in (host, dev):
            s.upload_landmarks(lm)
        _same(dev.score_fim(poses), host.score_fim(poses), f"m {m}: first cloud again")
    finally:
        host.close(); dev.close()


def test_device_order_is_a_function_of_the_input_and_works_on_every_member(fs):
    """two uploads of one cloud cut the same chunks (exactly as many landmark tests behind the cull; the float columns' last bits
    move from call to call anyway — racing LDS atomics); a two-member scorer orders the cloud on each member and
    scores like one context"""
    rng = np.random.default_rng(9)
    lm = rng.uniform(-10.0, 10.0, size=(30_000, 3)).astype(np.float32)
    poses = _poses(rng, 64, -8.0, 8.0)
    host, multi = _pair(fs, lm, 1.0, multi=True)
    one = fs.FrontierScorer(device=0)
    try:
        one.set_option("cloud.order", 2); one.set_option("fim.learn", 0)
        one.lookup_generate(); one.set_fim_params(14.0, 1.0)
        one.upload_landmarks(lm)
        one.get_counter(0, reset=True)
        a = one.score_fim(poses)
        ta = one.get_counter(0, reset=True)
        one.upload_landmarks(lm[::-1].copy())
        one.upload_landmarks(lm)
        one.get_counter(0, reset=True)
        b = one.score_fim(poses)
        tb = one.get_counter(0, reset=True)
        _same(a, b, "same cloud twice")
        assert ta == tb > 0                                      # the same chunks: exactly as many landmark tests behind the cull
        _same(multi.score_fim(poses), host.score_fim(poses), "two members")
    finally:
        host.close(); multi.close(); one.close()
