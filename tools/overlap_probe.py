"""Development probe: do successive batches overlap when they are scored on two contexts / streams?  (ray-march of batch
k+1 under the FIM kernel of batch k.)  Prints ms per batch for one context and for two alternating contexts."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload("C3")
dev = torch.device("cuda", 0)


def make():
    st = torch.cuda.Stream(device=dev)
    sc = fs.FrontierScorer(device=0, stream=st.cuda_stream)
    sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                      robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
    sc.max_arrival()
    rec = torch.zeros((w.goals.shape[0], 8), dtype=torch.int32, device=dev)
    return st, sc, rec


d_goal = torch.from_numpy(w.goals).to(dev); d_fs = torch.from_numpy(w.frontier_size).to(dev); d_bl = torch.from_numpy(w.blacklisted).to(dev)
ctxs = [make(), make()]
n = w.goals.shape[0]


def run(k_ctx, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        st, sc, rec = ctxs[i % k_ctx]
        sc.score_candidates_dev(n, d_goal.data_ptr(), d_fs.data_ptr(), d_bl.data_ptr(), 0, rec.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for k in (1, 2, 1, 2):
    run(k, 10)
    print("contexts %d: %.4f ms per batch" % (k, run(k, 100)))
