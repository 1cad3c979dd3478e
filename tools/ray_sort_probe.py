"""Development probe: does spatially sorting the candidate list speed up fs_raymarch_kernel?"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload("C3")
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.max_arrival()
cell = ((w.goals - np.asarray(w.origin)[None]) / w.resolution).astype(np.int64)
def spread(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff; v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3; v = (v | (v << 2)) & 0x09249249
    return v
code = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
order = np.argsort(code, kind="stable")
nb = (len(order) + 3) // 4
# XCD-aware: block b runs on XCD b % 8; give every XCD a contiguous run of the sorted list
blocks = order[: nb * 4].reshape(nb, 4) if len(order) % 4 == 0 else None
variants = {"as given": np.arange(len(order)), "morton sorted": order}
if blocks is not None:
    per = (nb + 7) // 8
    bidx = np.array([(b % 8) * per + b // 8 for b in range(nb)])
    ok = bidx < nb
    remap = np.full(nb, -1); remap[ok] = bidx[ok]
    if ok.all():
        variants["morton + xcd remap"] = blocks[remap].reshape(-1)
sc.enable_kernel_timing(True)
for rnd in range(3):
    for k, perm in variants.items():
        sc.kernel_time(0)
        r = sc.score_arrival(w.goals[perm], w.frontier_size[perm], w.blacklisted[perm], want_ray_counts=False)
        ms, n = sc.kernel_time(0)
        if rnd: print("%-20s raymarch %.3f ms" % (k, ms / n))
