#!/usr/bin/env python3
"""What a costmap update cycle costs the scorer: fs_upload_grid of the whole map against fs_update_grid_region of the window the
cycle rewrote (include/fitslam_frontier.h), each followed by the first scoring call (which re-cuts the class image — wholly after
a snapshot, only the touched bricks after a window).

    python tools/grid_region_probe.py [--workloads C3,REF2D] [--reps 9]

One JSON object: per workload and window size, the median wall time of the staging call, of the first fs_score_arrival after it
(2 000 candidates) and of a steady-state one; and whether the records after the windows equal those after a snapshot of the same
map (they must).  Host buffers are pageable numpy arrays, as a caller's costmap is.
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


def med(xs):
    return float(np.median(xs) * 1e3)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="C3,REF2D")
    ap.add_argument("--reps", type=int, default=9)
    args = ap.parse_args()
    fs = importlib.import_module("fit-slam_amd")
    out = {"what": "whole-map snapshot (fs_upload_grid) against a rewritten window (fs_update_grid_region); ms, medians", "workloads": {}}
    for name in args.workloads.split(","):
        w = fs.synth.make_workload(name, n_cand=2000)
        cells = np.array(w.cells, dtype=np.uint8, copy=True)
        nz, ny, nx = cells.shape
        kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
        s, t = fs.FrontierScorer(device=0), fs.FrontierScorer(device=0)
        for c in (s, t):
            c.set_ray_params(**kw)
            c.upload_grid(cells, w.origin, w.resolution)
        mx = s.max_arrival()
        t.set_arrival_limits(mx["max_gt"], mx["min_gt"])

        def score(c):
            t0 = time.perf_counter()
            r = c.score_arrival(w.goals, w.frontier_size, w.blacklisted, want_ray_counts=False)
            return time.perf_counter() - t0, r

        for _ in range(3):
            score(s)
        steady = [score(s)[0] for _ in range(args.reps)]
        rec = {"grid": [nx, ny, nz], "grid_bytes": int(cells.size), "steady_score_2000_ms": med(steady)}
        up, first = [], []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            s.upload_grid(cells, w.origin, w.resolution)
            up.append(time.perf_counter() - t0)
            s.set_arrival_limits(mx["max_gt"], mx["min_gt"])
            first.append(score(s)[0])
        rec["snapshot"] = {"stage_ms": med(up), "first_score_ms": med(first)}
        rng = np.random.default_rng(7)
        rec["windows"] = []
        sides = [(16, 16, 8), (64, 64, 16), (128, 128, 64), (256, 256, 128)] if nz > 1 else [(16, 16, 1), (64, 64, 1), (128, 128, 1), (256, 256, 1)]
        for sx, sy, sz in sides:
            sx, sy, sz = min(sx, nx), min(sy, ny), min(sz, nz)
            st_v, st_p, fi = [], [], []
            for r in range(args.reps):
                x0, y0, z0 = int(rng.integers(0, nx - sx + 1)), int(rng.integers(0, ny - sy + 1)), int(rng.integers(0, nz - sz + 1))
                vals = rng.choice(np.array([0, 0, 254, 255], dtype=np.uint8), size=(sz, sy, sx))
                cells[z0:z0 + sz, y0:y0 + sy, x0:x0 + sx] = vals
                view = cells[z0:z0 + sz, y0:y0 + sy, x0:x0 + sx]
                t0 = time.perf_counter()
                if r % 2:
                    s.update_grid_region(x0, y0, z0, view, view=True)         # straight from the caller's whole map
                    st_v.append(time.perf_counter() - t0)
                else:
                    s.update_grid_region(x0, y0, z0, vals)                    # a packed window
                    st_p.append(time.perf_counter() - t0)
                fi.append(score(s)[0])
            rec["windows"].append({"window": [sx, sy, sz], "bytes": sx * sy * sz, "stage_from_map_view_ms": med(st_v), "stage_packed_ms": med(st_p),
                                   "first_score_ms": med(fi)})
        # after all those windows the staged map must be the map: same records as a snapshot of it
        t.upload_grid(cells, w.origin, w.resolution)
        t.set_arrival_limits(mx["max_gt"], mx["min_gt"])
        a, b = score(s)[1], score(t)[1]
        rec["records_equal_a_snapshot_of_the_rewritten_map"] = bool(all(np.array_equal(a[k], b[k]) for k in ("status", "arrival", "argmax", "achievable", "yaw")))
        out["workloads"][name] = rec
        s.close(); t.close()
    print(json.dumps(out))
    return 0 if all(r["records_equal_a_snapshot_of_the_rewritten_map"] for r in out["workloads"].values()) else 1


if __name__ == "__main__":
    sys.exit(main())
