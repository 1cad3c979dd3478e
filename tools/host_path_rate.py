"""PCIe-inclusive rate of the host-buffer entry point fs_score_candidates (candidate columns in, records out) on a config."""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = fs.synth.make_workload(name)
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
t0 = time.perf_counter(); sc.upload_grid(w.cells, w.origin, w.resolution); t_grid = time.perf_counter() - t0
t0 = time.perf_counter(); sc.upload_landmarks(w.landmarks); t_lm = time.perf_counter() - t0
sc.lookup_generate(); sc.set_fim_params(14.0, 1.0); sc.max_arrival()
for _ in range(3): sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
t0 = time.perf_counter()
reps = 20
for _ in range(reps): rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
dt = (time.perf_counter() - t0) / reps
n = w.goals.shape[0]
print("%s: fs_score_candidates (host buffers in, records out) %.3f ms per %d candidates = %.2f M candidate-goals/s; "
      "one-time staging: grid %.1f ms (%d MiB, upload + re-tiling), landmarks %.1f ms (k-d ordering on the host + upload)"
      % (name, dt * 1e3, n, n / dt / 1e6, t_grid * 1e3, w.cells.size >> 20, t_lm * 1e3))
