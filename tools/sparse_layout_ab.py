#!/usr/bin/env python3
"""BASELINE.json configs[4] says "sparse-hashed voxel grid"; the build keeps the grid dense in HBM.  ONE measurement of the
alternative: the ray-march kernel over the dense 2-bit class image ("ray.layout" 2) against the same
image as brick table + pool of distinct bricks ("ray.layout" 3), alternating A / B / A / B in one process.

    python tools/sparse_layout_ab.py [--workloads C3,C5] [--rounds 4] [--steps 10]

Prints one JSON object: per workload the kernel's hipEvent time per launch under either layout (median over the rounds),
the image sizes, and whether the records are identical.  The counters (TCC_HIT / MISS, FETCH_SIZE) come from
tools/pmc_collect.py --bench-arg=--option --bench-arg=ray.layout=3 --passes tcc,fetch on the same workload.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="C3,C5")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    fs = importlib.import_module("fit-slam_amd")
    shard = importlib.import_module("fit-slam_amd.shard")
    dev = torch.device("cuda", 0)
    out = {"what": "fs_raymarch_kernel per launch, dense class image (ray.layout 2) vs brick table + pool (ray.layout 3)", "workloads": {}}
    for name in args.workloads.split(","):
        t0 = time.time()
        w = fs.synth.make_workload(name)
        print(f"[ab] {name}: workload built in {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        st = torch.cuda.Stream(device=dev)
        sc = fs.FrontierScorer(device=0, stream=st.cuda_stream)
        sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                          robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
        sc.upload_grid(w.cells, w.origin, w.resolution)
        sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
        sc.max_arrival()
        n = w.goals.shape[0]
        d_goal = torch.from_numpy(np.ascontiguousarray(w.goals)).to(dev)
        d_fs = torch.from_numpy(w.frontier_size).to(dev); d_bl = torch.from_numpy(w.blacklisted).to(dev)
        d_rec = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        res = {2: [], 3: []}
        recs = {}
        with torch.cuda.stream(st):
            for rnd in range(args.rounds):
                for layout in (2, 3):
                    sc.set_option("ray.layout", layout)
                    for _ in range(3):
                        sc.score_candidates_dev(n, d_goal.data_ptr(), d_fs.data_ptr(), d_bl.data_ptr(), 0, d_rec.data_ptr())
                    torch.cuda.synchronize(dev)
                    sc.enable_kernel_timing(True)
                    sc.kernel_time(0)
                    for _ in range(args.steps):
                        sc.score_candidates_dev(n, d_goal.data_ptr(), d_fs.data_ptr(), d_bl.data_ptr(), 0, d_rec.data_ptr())
                    torch.cuda.synchronize(dev)
                    ms, cnt = sc.kernel_time(0)
                    sc.enable_kernel_timing(False)
                    res[layout].append(ms / max(cnt, 1))
                    if rnd == 0:
                        recs[layout] = shard.records_to_numpy(d_rec).copy()
        bricks, pool = sc.get_counter(1000), sc.get_counter(1001)
        dense_ms, sparse_ms = float(np.median(res[2])), float(np.median(res[3]))
        out["workloads"][name] = {
            "grid": list(w.cells.shape), "candidates": int(n), "rays_per_candidate": sc.n_yaw * sc.n_elev,
            "dense_class_image_ms": dense_ms, "sparse_class_image_ms": sparse_ms, "sparse_over_dense": sparse_ms / dense_ms,
            "rounds_ms": {"dense": res[2], "sparse": res[3]},
            "bricks": int(bricks), "pool_bricks": int(pool), "dense_image_MB": bricks * 128 / 1e6,
            "sparse_image_MB": (pool * 128 + bricks * 4) / 1e6,
            "records_identical": bool(np.array_equal(recs[2]["arrival"], recs[3]["arrival"]) and np.array_equal(recs[2]["argmax"], recs[3]["argmax"])
                                      and np.array_equal(recs[2]["flags"], recs[3]["flags"]))}
        sc.close()
        del w, d_goal, d_rec
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
