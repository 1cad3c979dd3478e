#!/usr/bin/env python3
"""Generates tests/golden/*.npz — seeded inputs and the CPU oracle's outputs for them.

The reference holds no golden vectors for this path and cannot be built here
(§8c), so the fixtures are produced by oracle/ (itself pinned by tests/test_oracle_*.py) and
committed; the GPU tests replay the inputs through the HIP path and compare.  Fixtures are data
only (inputs + expected outputs).

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def one(name, w, O, table, elev=None, depth=None, max_angle=1.0, max_gt_for_costs=4000.0):
    kw = dict(max_camera_depth=depth or w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=elev or w.elev, polygon=w.polygon)
    G = O.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = O.RayParams(**kw)
    mx = O.max_arrival_information(G, P)
    arr = O.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], faithful=True)
    poses = O.poses_from_yaw(w.goals, arr["yaw"])
    fim = O.pose_information(table, w.landmarks, poses, 14.0, max_angle)
    # the planner's columns (inputs of the ranking: a low-discrepancy sequence, as bench.py's) and assignCosts' U1 block on them
    n = w.goals.shape[0]
    i = np.arange(n, dtype=np.float64)
    plen, phead = 0.5 + 29.5 * np.modf(i * 0.6180339887498949)[0], np.pi * np.modf(i * 0.7548776662466927)[0]
    rc, u1 = O.u1_costs(arr["arrival"].astype(np.float64), arr["achievable"], plen, phead, max_gt_for_costs, blacklisted=w.blacklisted)
    assert rc == 0
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        cells=w.cells, origin=np.array(w.origin), resolution=w.resolution, goals=w.goals,
        frontier_size=w.frontier_size, blacklisted=w.blacklisted, landmarks=w.landmarks,
        max_camera_depth=kw["max_camera_depth"], delta_theta=kw["delta_theta"], camera_fov=kw["camera_fov"],
        robot_radius=kw["robot_radius"], n_rays=kw["n_rays"], elev=np.array(kw["elev"]), polygon=np.array(kw["polygon"]),
        max_value=mx["max_value"], max_gt=mx["max_gt"], min_gt=mx["min_gt"],
        ray_counts=arr["ray_counts"], arrival=arr["arrival"], argmax=arr["argmax"], yaw=arr["yaw"],
        achievable=arr["achievable"], status=arr["status"],
        info_ref=fim["info_ref"], info_f64=fim["info_f64"], trace=fim["trace"], logdet=fim["logdet"],
        n_visible=fim["n_visible"], n_voxels=fim["n_voxels"], fim=fim["fim"], max_angle=max_angle,
        path_length=plen, path_heading=phead, max_gt_for_costs=max_gt_for_costs,
        weighted_cost=u1["weighted_cost"], arrival_utility=u1["arrival_utility"], distance_utility=u1["distance_utility"],
        order=np.argsort(u1["weighted_cost"], kind="stable").astype(np.int32))
    print(name, "written:", w.goals.shape[0], "candidates")


def main():
    fs = importlib.import_module("fit-slam_amd")
    import oracle as O
    O.build()
    table = O.Table.generate()
    one("small2d_reference_defaults", fs.synth.make_small_2d(11), O, table)
    w = fs.synth.make_workload("C1", n_cand=96, n_landmarks=1500)
    one("c1_3d_4rings", w, O, table, elev=(-0.30, -0.10, 0.10, 0.30), depth=1.5)
    # the reference's own visibility request (14 m, cone off: FisherInfoManager.cpp:63-64) on another small 2-D costmap
    one("small2d_reference_request", fs.synth.make_small_2d(23, n=128, n_cand=80, n_landmarks=900), O, table, max_angle=4.0)


if __name__ == "__main__":
    main()
