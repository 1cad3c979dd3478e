#!/usr/bin/env python3
"""How the pass prediction's margin trades scoring passes against HBM-tier hand-overs at the reference's own visibility request
(14 m, cone off) on C3: fs_score_fim over 20 k poses, info-only and full, for a range of "fim.headroom" / "fim.skip32" settings.
Prints one JSON line per setting (kernel ms from hipEvents, landmark tests per pose, multi-pass and HBM-tier poses per call)."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the HIP library: one runtime)

fs = importlib.import_module("fit-slam_amd")


def main():
    angle = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    w = fs.synth.make_workload("C3")
    sc = fs.FrontierScorer(device=0)
    sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                      robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate()
    sc.set_fim_params(14.0, angle); sc.max_arrival()
    rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    yaw = rec["argmax"].astype(np.float64) * w.delta_theta + w.camera_fov / 2
    half = yaw / 2
    poses = np.concatenate([w.goals, np.zeros((len(yaw), 2)), np.sin(half)[:, None], np.cos(half)[:, None]], axis=1)
    n = poses.shape[0]
    base = None
    for info_only in (True, False):
        for skip32, headroom in ((20, 40), (20, 32), (20, 26), (20, 20), (20, 16), (32, 40), (32, 32), (32, 24), (13, 40)):
            sc.set_option("fim.skip32", skip32); sc.set_option("fim.headroom", headroom)
            sc.set_fim_params(14.0, angle + 1e-9); sc.set_fim_params(14.0, angle)      # forget the learnt ratio
            for _ in range(3):
                r = sc.score_fim(poses, want_fim=False, info_only=info_only)
            sc.get_counter(0, reset=True)
            mp0, h0 = sc.get_counter(4), sc.get_counter(5)
            sc.enable_kernel_timing(True); sc.kernel_time(1); sc.kernel_time(2)
            reps = 5
            for _ in range(reps):
                r = sc.score_fim(poses, want_fim=False, info_only=info_only)
            k1, n1 = sc.kernel_time(1); k2, _ = sc.kernel_time(2)
            sc.enable_kernel_timing(False)
            if base is None:
                base = r["n_voxels"].copy()
            out = dict(info_only=info_only, skip32=skip32, headroom=headroom, fim_kernel_ms=k1 / n1, hbm_tier_ms=k2 / n1,
                       m_tested=sc.get_counter(0, reset=True) / (reps * n), multi_pass=(sc.get_counter(4) - mp0) / reps,
                       hbm_poses=(sc.get_counter(5) - h0) / reps, learnt=max(sc.get_counter(12), sc.get_counter(13)) / 256.0,
                       n_voxels_same=bool(np.array_equal(r["n_voxels"], base)))
            print(json.dumps(out), flush=True)
    q = np.percentile(base, [50, 90, 99, 100])
    print(json.dumps(dict(n_voxels_p50=q[0], p90=q[1], p99=q[2], max=q[3], over_12288=int((base > 12288).sum()), over_24576=int((base > 24576).sum()))))
    sc.close()


if __name__ == "__main__":
    main()
