"""Development tool: per-stage cost of fs_fim_kernel by ablation, interleaved rounds in one process."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = fs.synth.make_workload(name)
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
sc.enable_kernel_timing(True)
variants = [("full", 0), ("no score (cull+test+compact)", 1), ("no hash", 2), ("no table gather", 4), ("no FIM sums", 8),
            ("no hash+gather+sums", 14), ("cull only", 16 | 1)]
res = {k: [] for k, _ in variants}
for rnd in range(4):
    for k, mask in variants:
        sc.set_option("fim.ablate", mask)
        sc.kernel_time(0); sc.kernel_time(1); sc.kernel_time(2)
        sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        t1, _ = sc.kernel_time(1); t2, _ = sc.kernel_time(2); t0, _ = sc.kernel_time(0)
        if rnd: res[k].append((t1, t2, t0))
for k, _ in variants:
    a = np.array(res[k])
    print("%-32s fim %.3f ms (min %.3f)  tiers %.3f ms  ray %.3f ms" % (k, np.median(a[:, 0]), a[:, 0].min(), np.median(a[:, 1]), np.median(a[:, 2])))
