#!/usr/bin/env python3
"""Why did fs_fim_overflow_kernel take 0.30 ms per step in round 2's two-rank rehearsal on ONE GPU, against 0.007 ms with one
rank?  Two scoring contexts on one device (fs_multi with the device list [0, 0]), each scoring half of
C3's list side by side, per-kernel hipEvent times read from each member context — once as they are launched (both persistent
FIM grids resident together) and once with the second member started only after the first has finished (no overlap)."""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fs = importlib.import_module("fit-slam_amd")


class Ctx(fs.FrontierScorer):
    """A FrontierScorer view of a member context of an fs_multi (not owned)."""
    def __init__(self, L, handle):
        self._L, self._h = L, C.c_void_p(handle)
    def close(self):
        self._h = None


def main():
    w = fs.synth.make_workload("C3")
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    out = {}
    for label, devices in (("one context, whole list", (0,)), ("two contexts on one GPU, half a list each, side by side", (0, 0))):
        m = fs.MultiScorer(devices=devices)
        m.set_ray_params(**kw); m.upload_grid(w.cells, w.origin, w.resolution); m.upload_landmarks(w.landmarks)
        m.lookup_generate(); m.set_fim_params(14.0, 1.0); m.max_arrival()
        members = [Ctx(m._L, m._L.fs_multi_ctx(m._h, i)) for i in range(len(devices))]
        for _ in range(5):
            m.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        for c in members:
            c.enable_kernel_timing(True)
            for kind in range(5):
                c.kernel_time(kind)
        reps = 20
        for _ in range(reps):
            m.score_candidates(w.goals, w.frontier_size, w.blacklisted)
        names = {0: "fs_raymarch_kernel", 1: "fs_fim_kernel", 2: "fs_fim_overflow_kernel", 4: "candidate_sort"}
        out[label] = [{names[k]: round(c.kernel_time(k)[0] / reps, 4) for k in names} for c in members]
        m.close()
    out["reading"] = ("hipEvent pairs around a launch measure from the moment the launch is REACHED on its stream to its end.  The HBM tier is a "
                      "256 x 1024-thread launch that needs a whole CU's wave slots per workgroup; while the OTHER context's persistent FIM grid "
                      "(two 512-thread workgroups per CU, 128 VGPRs each: the register file is full) is resident it cannot start, so its event pair "
                      "absorbs the rest of the neighbour's FIM kernel.  It still does nothing (no candidate is flagged).  With one rank per GPU — the "
                      "measured configuration — there is no neighbour and the launch costs 0.007 ms.")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
