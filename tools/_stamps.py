import importlib, sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload("C3")
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
sc.set_option("fim.debug", 1)
rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
n = rec.shape[0]
raw = np.zeros(n * 64, dtype=np.uint64)
L = fs.load_library(); L.fs_debug_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
L.fs_debug_fetch(sc._h, n, raw.ctypes.data_as(C.c_void_p))
d = raw.reshape(n, 8, 8).astype(np.float64)
ok = d[:, 0, 0] > 0                      # candidates that ran tier 1 to the end (not skipped)
d = d[ok]
names = ["pose+cull", "barrier1", "size+clear+barrier2", "loop", "flush", "dpp+barrier3", "final"]
ph = d[:, :, :6]
print("candidates with stamps:", d.shape[0])
print("per-wave mean cycles: " + "  ".join("%s %.0f" % (k, v) for k, v in zip(["pose+cull", "publish+barrier1+size+clear+barrier2", "loop", "flush", "dpp+barrier3", "final"], ph.mean(axis=(0, 1)))))
loop = ph[:, :, 2]
print("loop: mean over waves %.0f, mean of max-over-waves %.0f, mean of min %.0f" % (loop.mean(), loop.max(axis=1).mean(), loop.min(axis=1).mean()))
tot = ph.sum(axis=2)
print("total per wave %.0f ; max over waves %.0f" % (tot.mean(), tot.max(axis=1).mean()))
nsc = d[:, :, 6]; nch = d[:, :, 7]
print("score calls per wave mean %.2f (max-min across waves mean %.2f); chunks per wave mean %.1f (max-min %.2f)" % (nsc.mean(), (nsc.max(axis=1) - nsc.min(axis=1)).mean(), nch.mean(), (nch.max(axis=1) - nch.min(axis=1)).mean()))
# regress loop cycles on calls and chunks
A = np.stack([nsc.ravel(), nch.ravel(), np.ones(nsc.size)], axis=1)
coef, *_ = np.linalg.lstsq(A, loop.ravel(), rcond=None)
print("loop cycles ~ %.0f * score_calls + %.0f * chunks + %.0f" % tuple(coef))
