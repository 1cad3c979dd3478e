"""Soak run, by hand on the GPU box: the calls a FIT-SLAM node makes per tick — the whole cost assignment over tens of frontiers, one
isPoseSafe pose, a new costmap snapshot now and then, a new landmark cloud now and then, list lengths that wander — repeated for
`seconds` on ONE context and on a two-member multi-device scorer, watching host memory (RSS) and device memory (hipMemGetInfo through
torch) for growth.  A leak of one event, one graph or one staging buffer per call shows as a slope; the test of "no leak" is that
the second half of the run ends where the first half ended.

    python tests/soak/soak.py [seconds]

Prints one JSON line.  (Lives under tests/ because it is test infrastructure; it does not use the oracle.)
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    import psutil
    import torch                                     # before the library: one HIP runtime for both (as in bench.py)
    fs = importlib.import_module("fit-slam_amd")
    proc = psutil.Process()
    rng = np.random.default_rng(1)
    w = fs.synth.make_workload("REF2D", n_cand=2000)
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    sc = fs.FrontierScorer(0)
    m = fs.MultiScorer(devices=(0, 0))
    for s in (sc, m):
        s.set_ray_params(**kw); s.upload_grid(w.cells, w.origin, w.resolution); s.upload_landmarks(w.landmarks); s.lookup_generate()
        s.set_fim_params(14.0, 4.0)
    mx = sc.max_arrival()
    m.set_arrival_limits(mx["max_gt"], mx["min_gt"])
    first = sc.score_arrival(w.goals, w.frontier_size, w.blacklisted, want_ray_counts=False)
    poses = fs.synth.poses_from_yaw(w.goals, first["yaw"])

    def sample():
        free, total = torch.cuda.mem_get_info(0)
        return proc.memory_info().rss, total - free

    samples = []
    calls = 0
    t0 = time.time()
    next_sample = 0.0
    while True:
        now = time.time() - t0
        if now >= next_sample:
            samples.append((now, calls) + sample())
            next_sample += seconds / 20.0
            if now >= seconds:
                break
        n = int(rng.choice([1, 3, 8, 20, 50, 51, 120, 400, 1100, 2000], p=[.1, .1, .15, .2, .2, .05, .08, .06, .03, .03]))
        pick = rng.choice(w.goals.shape[0], size=n, replace=False)
        i = np.arange(n, dtype=np.float64)
        plen, phead = 0.5 + 29.5 * np.modf(i * 0.6180339887498949)[0], np.pi * np.modf(i * 0.7548776662466927)[0]
        target = sc if (calls % 3 or os.environ.get("SOAK_NO_MULTI")) else m
        target.get_frontier_costs(w.goals[pick], plen, phead, w.frontier_size[pick], w.blacklisted[pick], with_fim=bool(calls % 2))
        if not os.environ.get("SOAK_NO_FIM"):
            target.score_fim(poses[pick[:1]], info_only=True)
        if calls % 500 == 499 and not os.environ.get("SOAK_NO_SNAPSHOT"):                                   # a new snapshot, a new (slightly different) cloud
            k = int(rng.integers(1000, w.landmarks.shape[0]))
            for s in (sc, m):
                s.upload_grid(w.cells, w.origin, w.resolution)
                s.upload_landmarks(w.landmarks[:k])
            m.set_arrival_limits(mx["max_gt"], mx["min_gt"]); sc.set_arrival_limits(mx["max_gt"], mx["min_gt"])
        if calls % 2000 == 1999 and not os.environ.get("SOAK_NO_OPTIONS"):                                 # the optional routes too
            for key in ("graph", "zerocopy"):
                sc.set_option(key, float(rng.integers(0, 2)))
        calls += 2
    sc.set_option("graph", 0); sc.set_option("zerocopy", 1)
    half = len(samples) // 2
    rss = [s[2] for s in samples]; dev = [s[3] for s in samples]
    out = {
        "seconds": round(samples[-1][0], 1), "calls": calls,
        "host_rss_mb": {"start": rss[1] / 2**20, "middle": rss[half] / 2**20, "end": rss[-1] / 2**20},
        "device_used_mb": {"start": dev[1] / 2**20, "middle": dev[half] / 2**20, "end": dev[-1] / 2**20},
        "host_rss_growth_second_half_mb": (rss[-1] - rss[half]) / 2**20,
        "device_growth_second_half_mb": (dev[-1] - dev[half]) / 2**20,
        "every_sample": [{"t": round(s[0], 1), "calls": s[1], "rss_mb": round(s[2] / 2**20, 1), "device_mb": round(s[3] / 2**20, 1)} for s in samples],
    }
    out["ok"] = bool(out["host_rss_growth_second_half_mb"] < 16.0 and out["device_growth_second_half_mb"] < 16.0)
    print(json.dumps(out))
    sc.close(); m.close()
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
