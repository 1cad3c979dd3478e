"""Random short frontier lists through the fused call (fs_score_candidates: ray-cast -> FIM -> records) and through the multi-device
one call (fs_multi_get_frontier_costs on two contexts of the one GPU, a random gather mode) against the oracle — the lists of a few
frontiers that the reference scores per tick are the ones whose poses are spread over several workgroups (fs_fim.hip, SPLIT).
By hand on the GPU box, for as many trials as one likes:

    python tests/fused_random.py [trials] [seed]

(Lives under tests/ because it uses the oracle: the checker is test infrastructure.)
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fs = importlib.import_module("fit-slam_amd")
parity = importlib.import_module("fit-slam_amd.parity")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402  (the checker)

REL = 1e-4


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    table = oracle.Table.generate()
    t0 = time.time()
    worlds = {}
    for name in ("REF2D", "C2"):
        w = fs.synth.make_workload(name, n_cand=400)
        kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
        sc = fs.FrontierScorer(0)
        m = fs.MultiScorer(devices=(0, 0))
        for s in (sc, m):
            s.set_ray_params(**kw); s.upload_grid(w.cells, w.origin, w.resolution); s.upload_landmarks(w.landmarks); s.lookup_generate()
        G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
        P = oracle.RayParams(**kw)
        mx = oracle.max_arrival_information(G, P)
        assert sc.max_arrival() == mx and m.max_arrival() == mx
        worlds[name] = (w, sc, m, G, P, mx)
    worst = 0.0
    for t in range(trials):
        name = str(rng.choice(["REF2D", "C2"]))
        w, sc, m, G, P, mx = worlds[name]
        n = int(rng.choice([1, 2, 3, 5, 9, 17, 32, 33, 64, 200]))
        pick = rng.choice(w.goals.shape[0], size=n, replace=False)
        goals, fsz, bl = w.goals[pick], w.frontier_size[pick], w.blacklisted[pick]
        angle = float(rng.choice([0.6, 1.0, 1.3, 4.0]))
        split = int(rng.choice([0, 1, 2, 3, 4]))
        sc.set_fim_params(14.0, angle); m.set_fim_params(14.0, angle)
        sc.set_option("fim.split", split); m.set_option("fim.split", split)
        arr = oracle.arrival_information(G, P, goals, fsz, bl, min_gt=mx["min_gt"], n_threads=16, want_ray_counts=False)
        fim = oracle.pose_information(table, w.landmarks, oracle.poses_from_yaw(goals, arr["yaw"]), 14.0, angle, n_threads=16)
        rec = sc.score_candidates(goals, fsz, bl)
        tag = f"trial {t}: {name} n={n} angle={angle} split={split}"
        ok = arr["status"] == 0
        if not (np.array_equal(rec["arrival"], arr["arrival"]) and np.array_equal(rec["argmax"], arr["argmax"])):
            bad = np.flatnonzero((rec["arrival"] != arr["arrival"]) | (rec["argmax"] != arr["argmax"]))
            again = sc.score_arrival(goals, fsz, bl)
            raise AssertionError((tag, bad.tolist(), rec["arrival"][bad].tolist(), arr["arrival"][bad].tolist(), rec["argmax"][bad].tolist(), arr["argmax"][bad].tolist(),
                                  fs.capi.record_status(rec)[bad].tolist(), arr["status"][bad].tolist(), goals[bad].tolist(), pick[bad].tolist(),
                                  "score_arrival:", again["arrival"][bad].tolist(), again["argmax"][bad].tolist()))
        assert np.array_equal(fs.capi.record_status(rec), arr["status"]) and np.array_equal(fs.capi.record_achievable(rec), arr["achievable"]), tag
        assert np.array_equal(rec["n_visible"][ok], fim["n_visible"][ok]), tag
        assert np.array_equal(fs.capi.record_nvoxels(rec)[ok], np.minimum(fim["n_voxels"][ok], 65535)), tag
        e = max(parity.rel_err(rec["info_ref"][ok], fim["info_f64"][ok]), parity.rel_err(rec["trace"][ok], fim["trace"][ok]))
        assert e <= REL, (tag, e)
        worst = max(worst, e)
        gate = parity.logdet_gate(rec["logdet"], fim["logdet"], fim["fim"], consider=ok, n_visible=fim["n_visible"])
        assert gate["ok"], (tag, gate)
        # the whole cost assignment as one call over two members, with Fisher information, through a random gather mode
        i = np.arange(n, dtype=np.float64)
        plen, phead = 0.5 + 29.5 * np.modf(i * 0.6180339887498949)[0], np.pi * np.modf(i * 0.7548776662466927)[0]
        rc, u1 = oracle.u1_costs(arr["arrival"].astype(np.float64), arr["achievable"], plen, phead, 4000.0, blacklisted=bl)
        assert rc == 0
        m.set_arrival_limits(4000.0, mx["min_gt"])
        m.set_option("multi.gather", int(rng.choice([0, 2, 3])))
        got = m.get_frontier_costs(goals, plen, phead, fsz, bl, with_fim=True)
        assert np.array_equal(got["records"]["arrival"], arr["arrival"]) and np.array_equal(got["records"]["n_visible"][ok], fim["n_visible"][ok]), tag
        for k in ("weighted_cost", "arrival_utility", "distance_utility"):
            assert np.array_equal(got[k], u1[k]), (tag, k)
        assert np.array_equal(got["order"], np.argsort(u1["weighted_cost"], kind="stable")), tag
        assert parity.rel_err(got["records"]["info_ref"][ok], fim["info_f64"][ok]) <= REL, tag
        if t % 20 == 0:
            print(f"ok {tag}  ({time.time() - t0:.0f} s)", flush=True)
    print(f"{trials} trials passed, worst relative error of info / trace {worst:.2e}")


if __name__ == "__main__":
    main()
