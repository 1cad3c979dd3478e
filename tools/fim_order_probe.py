"""Development probe: how much of fs_fim_kernel's time is the drain of its persistent grid, and does the processing order
matter?  The device-side spatial sort is switched off and the candidate list itself is permuted on the host:
  morton        the host's own Morton order of the goal cells (the reference point: what the device sort does)
  heavy-first   candidates with n_visible above a percentile first (Morton order inside both groups)
  lpt           descending n_visible (longest processing time first; no spatial coherence left)
n_visible comes from a first, ordinary run.  Prints the hipEvent time of fs_fim_kernel / fs_raymarch_kernel per order."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "C3")


def scorer(sort):
    sc = fs.FrontierScorer(0)
    sc.set_option("ray.sort", 1.0 if sort else 0.0)
    sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                      robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
    sc.max_arrival()
    return sc


def timed(sc, order, reps=10):
    g, f, b = w.goals[order].copy(), w.frontier_size[order].copy(), w.blacklisted[order].copy()
    for _ in range(3): rec = sc.score_candidates(g, f, b)
    sc.enable_kernel_timing(True)
    for k in range(5): sc.kernel_time(k)
    for _ in range(reps): rec = sc.score_candidates(g, f, b)
    out = [sc.kernel_time(k) for k in (0, 1)]
    sc.enable_kernel_timing(False)
    return rec, out[0][0] / max(1, out[0][1]), out[1][0] / max(1, out[1][1])


def part1by2(v):
    v = v.astype(np.uint64) & 0x3FF
    v = (v | (v << 16)) & 0x30000FF
    v = (v | (v << 8)) & 0x300F00F
    v = (v | (v << 4)) & 0x30C30C3
    v = (v | (v << 2)) & 0x9249249
    return v


n = w.goals.shape[0]
ident = np.arange(n)
sc = scorer(True)
rec, ray0, fim0 = timed(sc, ident)
nvis = np.asarray(rec["n_visible"])
print("device sort:   ray %.4f ms  fim %.4f ms   (n_visible mean %.0f, p99 %.0f, max %d)" % (ray0, fim0, nvis.mean(), np.percentile(nvis, 99), nvis.max()))
sc.close()
cell = np.floor((w.goals - np.asarray(w.origin)[None, :]) / w.resolution).astype(np.int64)
key = part1by2(cell[:, 0]) | (part1by2(cell[:, 1]) << 1) | (part1by2(cell[:, 2]) << 2)
morton = np.argsort(key, kind="stable")
pos = np.empty(n, dtype=np.int64); pos[morton] = np.arange(n)
print("heavy candidates (top 5 %% by n_visible) sit at these deciles of the Morton order:",
      np.histogram(pos[nvis >= np.percentile(nvis, 95)], bins=10, range=(0, n))[0].tolist())
sc = scorer(False)
orders = {"morton": morton}
for pct in (50, 80, 95):
    heavy = nvis[morton] >= np.percentile(nvis, pct)
    orders["heavy-first p%d" % pct] = np.concatenate([morton[heavy], morton[~heavy]])
orders["lpt"] = np.argsort(-nvis, kind="stable")
orders["morton reversed"] = morton[::-1]
for name, o in orders.items():
    r, ray, fim = timed(sc, o)
    same = np.array_equal(np.asarray(r["n_visible"]), nvis[o])
    print("%-18s ray %.4f ms  fim %.4f ms   results follow the permutation: %s" % (name, ray, fim, same))
