#!/usr/bin/env python3
"""Which poses miss north_star's plain 1e-4 on log det, and why.

    python tools/logdet_probe.py [--workload REF2D] [--n 2000] [--angle 1.0]

For the first --n candidates of the workload: the fused records of the HIP path, the float64 oracle's F, log det and
condition number, and for the worst poses the error next to kappa * 2^-24 (what the float32 inputs' own rounding can move
log det by) and next to what rounding the ORACLE's F to float32 entries alone would do.  Output: one JSON object on stdout.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="REF2D")
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--angle", type=float, default=1.0)
    ap.add_argument("--worst", type=int, default=12)
    args = ap.parse_args()
    fs = importlib.import_module("fit-slam_amd")
    import oracle as O   # the checker
    O.build()
    w = fs.synth.make_workload(args.workload, n_cand=None)
    n = min(args.n, w.goals.shape[0])
    kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
              robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    sc = fs.FrontierScorer(device=0)
    sc.set_ray_params(**kw); sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks)
    sc.lookup_generate(); sc.set_fim_params(14.0, args.angle)
    mx = sc.max_arrival()
    rec = sc.score_candidates(w.goals[:n], w.frontier_size[:n], w.blacklisted[:n])
    G = O.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    T = min(16, os.cpu_count() or 1)
    arr = O.arrival_information(G, O.RayParams(**kw), w.goals[:n], w.frontier_size[:n], w.blacklisted[:n], min_gt=mx["min_gt"],
                                faithful=False, n_threads=T, want_ray_counts=False)
    poses = O.poses_from_yaw(w.goals[:n], arr["yaw"])
    fim = O.pose_information(O.Table.generate(), w.landmarks, poses, 14.0, args.angle, n_threads=T, want_f64=True)
    got = sc.score_fim(poses)                     # the general worker, with the 21 entries
    ok = arr["status"] == 0
    fin = ok & np.isfinite(fim["logdet"]) & np.isfinite(rec["logdet"])
    lam = np.linalg.eigvalsh(fim["fim"])
    cond = lam[:, -1] / np.maximum(lam[:, 0], 1e-300)
    ld = fim["logdet"]
    err = np.abs(rec["logdet"].astype(np.float64) - ld)
    rel = err / np.maximum(1.0, np.abs(ld))
    err_g = np.abs(got["logdet"].astype(np.float64) - ld)
    # log det of the oracle's F after rounding its entries to float32 (what ANY float32 F can give at best)
    F32 = fim["fim"].astype(np.float32).astype(np.float64)
    ld32 = np.full(n, -np.inf)
    for i in np.flatnonzero(fin):
        s, v = np.linalg.slogdet(F32[i])
        ld32[i] = v if s > 0 else -np.inf
    iu = np.triu_indices(6)
    ent = np.abs(got["fim21"].astype(np.float64) - fim["fim"][:, iu[0], iu[1]]) / np.maximum(np.abs(fim["fim"]).max(axis=(1, 2))[:, None], 1e-300)
    idx = np.flatnonzero(fin)
    worst = idx[np.argsort(-rel[idx])][: args.worst]
    out = {"workload": args.workload, "n": int(n), "visibility": f"14 m, {args.angle} rad", "finite": int(fin.sum()),
           "share_within_1e-4": float(np.mean(rel[fin] <= 1e-4)),
           "share_within_1e-4_plus_floor": float(np.mean(err[fin] <= 1e-4 * np.maximum(1.0, np.abs(ld[fin])) + 2.0 ** -24 * cond[fin])),
           "max_rel_err": float(rel[fin].max()),
           "kappa_percentiles": {str(p): float(np.percentile(cond[fin], p)) for p in (50, 90, 99, 100)},
           "err_over_kappa_eps_max": float((err[fin] / (2.0 ** -24 * cond[fin])).max()),
           "worst": [{"candidate": int(i), "n_visible": int(fim["n_visible"][i]), "kappa": float(cond[i]),
                      "logdet_oracle": float(ld[i]), "logdet_fused": float(rec["logdet"][i]), "logdet_general": float(got["logdet"][i]),
                      "abs_err_fused": float(err[i]), "abs_err_general": float(err_g[i]), "rel_err": float(rel[i]),
                      "kappa_eps": float(2.0 ** -24 * cond[i]), "err_of_oracle_F_rounded_to_f32": float(abs(ld32[i] - ld[i])),
                      "fim21_max_entry_err_rel_to_largest": float(ent[i].max())} for i in worst]}
    print(json.dumps(out, indent=1))
    sc.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
