"""By hand on the GPU box: inputs nobody sends on purpose, through the entry points either side of the hot path — NaN / infinite /
far-away segment end points, robot positions, triangles, path columns, a truncated lookup-table file — each call wrapped so that an
error CODE is fine and a crash, a hang or a disagreement with the oracle (where the oracle defines the case) is reported.
    python tests/adversarial_probe.py
(Uses the oracle: test infrastructure.)"""
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

fs = importlib.import_module("fit-slam_amd")
problems = []


def attempt(what, fn):
    try:
        r = fn()
        print("ok   ", what)
        return r
    except fs.FsError as e:
        print("code ", what, "->", e)
    except AssertionError as e:
        print("DIFF ", what, "->", str(e)[:300])
        problems.append(what)
    return None


def main():
    w = fs.synth.make_small_2d(9, n=128, n_cand=40, n_landmarks=800)
    s = fs.FrontierScorer(0)
    s.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov, robot_radius=w.robot_radius,
                     n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    s.upload_grid(w.cells, w.origin, w.resolution)
    s.upload_landmarks(w.landmarks); s.lookup_generate(); s.set_fim_params(14.0, 1.0)
    mx = s.max_arrival()
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    lo = np.array(w.origin); hi = lo + np.array([128, 128, 1]) * w.resolution
    weird = [np.nan, np.inf, -np.inf, 1e30, -1e30, 1e300, lo[0] - 1e-12, hi[0], hi[0] - 1e-12, 0.0]

    # ---- roadmap edge tracing (fs_trace_segments) against oracle.trace_ray
    rng = np.random.default_rng(0)
    a = rng.uniform(lo, hi, size=(60, 3)); b = rng.uniform(lo, hi, size=(60, 3)); a[:, 2] = b[:, 2] = 0.0
    for k in range(60):
        if k % 2 == 0:
            a[k, k % 3 if k % 3 < 2 else 0] = weird[k % len(weird)]
        else:
            b[k, k % 2] = weird[k % len(weird)]

    def seg():
        got = s.trace_segments(a, b, 183)
        for k in range(60):
            with np.errstate(all="ignore"):
                r = oracle.trace_ray(G, tuple(a[k]), tuple(b[k]), 183, obst=(253, 254), trace=(0, 255), faithful=True)
            assert bool(got["ok"][k]) == bool(r["ok"]), (k, a[k], b[k], got["ok"][k], r["ok"])
            if r["ok"]:
                assert (got["hit"][k], got["unknown"][k]) == (int(r["hit"]), r["unknown"]), (k, a[k], b[k])
        return got
    attempt("fs_trace_segments with NaN / inf / far / boundary end points == oracle", seg)

    # ---- arrival information with weird goals
    goals = w.goals.copy()
    for k in range(len(goals)):
        goals[k, k % 2] = weird[k % len(weird)]

    def arr():
        with np.errstate(all="ignore"):
            want = oracle.arrival_information(G, oracle.RayParams(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                                                                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon),
                                              goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], faithful=True)
        got = s.score_arrival(goals, w.frontier_size, w.blacklisted)
        for k in ("status", "arrival", "argmax", "achievable"):
            assert np.array_equal(got[k], want[k]), (k, got[k], want[k])
        rec = s.score_candidates(goals, w.frontier_size, w.blacklisted)
        assert np.array_equal(rec["arrival"], want["arrival"])
        assert np.all(np.isfinite(rec["info_ref"])), rec["info_ref"]
    attempt("arrival information / fused records with NaN / inf / far / boundary goals == oracle", arr)

    # ---- frontier detection from weird robot positions
    for xy in ((np.nan, 0.0), (np.inf, 0.0), (1e30, 1e30), (lo[0] - 1.0, lo[1] - 1.0), (hi[0], hi[1])):
        def fc(xy=xy):
            labels, cl, n, cells = s.frontier_clusters(w.cells.shape, xy)
            if not np.all(np.isfinite(xy)) or max(abs(xy[0]), abs(xy[1])) > 1e9:
                # worldToMap of such a position is a float-to-integer conversion out of range: undefined in the reference (and in the
                # oracle, which inherits it).  Here: off the map, nothing found, no error.
                assert (n, cells) == (0, 0), (n, cells)
                return n
            ref = oracle.frontier_search(w.cells, w.origin, w.resolution, xy)
            if ref["ok"]:
                assert cells == ref["n_every"], (cells, ref["n_every"])
            return n
        attempt(f"fs_frontier_clusters from robot position {xy}", fc)

    # ---- frontier-pair information with weird triangles / poses
    pose = np.zeros((4, 7)); pose[:, 6] = 1.0; pose[1, 0] = np.nan; pose[2, 3:] = np.nan; pose[3, :3] = 1e30
    tri = np.tile(np.array([[-1.0, -1.0, 3.0, -1.0, 1.0, 3.0]]), (4, 1)); tri[0, 2] = np.nan; tri[3, 4] = np.inf

    def pair():
        got = s.information_frontier_pair(pose, tri)
        with np.errstate(all="ignore"):
            for k in range(4):
                want = oracle.information_frontier_pair(w.landmarks, pose[k], tri[k].reshape(3, 2))
                if np.isfinite(want):
                    assert abs(got[k] - want) <= 1e-4 * max(abs(want), 1e-6), (k, got[k], want)
        return got
    attempt("fs_information_frontier_pair with NaN / inf triangles and poses", pair)

    # ---- ranking with weird path columns
    rec = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    n = len(rec)
    for name, plen, phead, kw in (("NaN path length", np.where(np.arange(n) % 7 == 0, np.nan, 5.0), np.full(n, 1.0), {}),
                                  ("infinite path length", np.where(np.arange(n) % 7 == 0, np.inf, 5.0), np.full(n, 1.0), {}),
                                  ("negative heading", np.full(n, 5.0), np.full(n, -1.0), {}),
                                  ("max_vx = 0", np.full(n, 5.0), np.full(n, 1.0), dict(max_vx=0.0)),
                                  ("beta = 0", np.full(n, 5.0), np.full(n, 1.0), dict(beta=0.0))):
        def rk(plen=plen, phead=phead, kw=kw):
            r = s.rank_candidates(rec, plen, phead, w.blacklisted, **kw)
            assert np.array_equal(np.sort(r["order"]), np.arange(n)), "order is not a permutation"
            return r
        attempt(f"fs_rank_candidates with {name}", rk)

    # ---- a truncated / corrupt lookup-table file
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.dat")
        s.lookup_save(path)
        raw = open(path, "rb").read()
        for name, blob in (("truncated mid-record", raw[:len(raw) // 2 + 5]), ("empty", b""), ("16 bytes of NaN", np.full(4, np.nan, np.float32).tobytes())):
            open(path, "wb").write(blob)
            attempt(f"fs_lookup_load of a file that is {name}", lambda: s.lookup_load(path))
        attempt("fs_lookup_load of a missing file", lambda: s.lookup_load(os.path.join(d, "nope.dat")))
        s.lookup_generate()
    after = s.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    assert np.array_equal(after["arrival"], rec["arrival"]) and np.array_equal(after["n_visible"], rec["n_visible"]), "the context did not survive the probes"
    s.close()
    print("problems:", problems)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
