"""Workload statistics of a config on the GPU: visible fraction, distinct voxels, arrival distribution."""
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
print("max_arrival", sc.max_arrival())
rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
nv = rec["n_visible"]; vox = (rec["flags"] >> 16) & 0xFFFF
m = w.landmarks.shape[0]
print(name, "cells", w.cells.shape, "unk/free/obst", (w.cells == 255).mean(), (w.cells == 0).mean(), ((w.cells >= 240) & (w.cells < 255)).mean())
print("n_visible: mean %.0f (%.3f of M) p50 %.0f p99 %.0f max %d" % (nv.mean(), nv.mean() / m, np.percentile(nv, 50), np.percentile(nv, 99), nv.max()))
print("n_voxels : mean %.0f p50 %.0f p99 %.0f max %d (65535 = saturated)" % (vox.mean(), np.percentile(vox, 50), np.percentile(vox, 99), vox.max()))
print("arrival  : mean %.1f max %d ; achievable %.3f ; status counts %s" % (rec["arrival"].mean(), rec["arrival"].max(), (rec["flags"] & 1).mean(), np.bincount((rec["flags"] >> 8) & 0xFF)))
print("info_ref : mean %.1f max %.1f ; frac > 550: %.3f" % (rec["info_ref"].mean(), rec["info_ref"].max(), (rec["info_ref"] > 550).mean()))
