"""By hand on the GPU box: ONE call over three million candidates (150 times C3's list; C1's map and cloud) — the records must be those of
the same list scored in three blocks (integers bit for bit: nothing in the call depends on the list length), and ranking the three
million must give a permutation with ascending costs.  `python tests/big_list.py`.  (No oracle involved: a size-independent property.)
Measured round 5: 0.10 s for the scoring call (30 M candidate-goals/s through host buffers), 0.05 s for the ranking call."""
import importlib, sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload("C1")
sc = fs.FrontierScorer(0)
kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov, robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.set_ray_params(**kw); sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
rng = np.random.default_rng(0)
N = 3_000_000
idx = rng.integers(0, w.goals.shape[0], size=N)
goals = w.goals[idx] + rng.uniform(-0.02, 0.02, size=(N, 3)) * np.array([1, 1, 0])
fsz = w.frontier_size[idx]; bl = (rng.random(N) < 0.01).astype(np.uint8)
t = time.time(); rec = sc.score_candidates(goals, fsz, bl); print("one call", N, "candidates", round(time.time() - t, 2), "s")
bad = 0
for lo in range(0, N, 1_000_003):
    hi = min(N, lo + 1_000_003)
    part = sc.score_candidates(goals[lo:hi], fsz[lo:hi], bl[lo:hi])
    for k in ("arrival", "argmax", "n_visible", "flags"):
        bad += int((part[k] != rec[k][lo:hi]).sum())
    rel = np.abs(part["info_ref"] - rec["info_ref"][lo:hi]) / np.maximum(np.abs(rec["info_ref"][lo:hi]), 1e-6)
    assert rel.max() <= 1e-5, rel.max()
print("integer mismatches between the one call and three blocks:", bad, "; arrival > 0 on", int((rec["arrival"] > 0).sum()))
i = np.arange(N, dtype=np.float64)
plen, phead = 0.5 + 29.5 * np.modf(i * 0.6180339887498949)[0], np.pi * np.modf(i * 0.7548776662466927)[0]
sc.set_arrival_limits(4000.0, 1.0)
t = time.time(); r = sc.rank_candidates(rec, plen, phead, bl); print("rank", round(time.time() - t, 2), "s")
c = r["weighted_cost"][r["order"]]
assert np.all(np.diff(c) >= 0) and np.array_equal(np.sort(r["order"]), np.arange(N))
print("order is a permutation, costs ascending; OK" if bad == 0 else "MISMATCH")
