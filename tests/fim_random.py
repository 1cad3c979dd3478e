"""Random clouds / poses / visibility volumes / tables through fs_score_fim against the oracle — integers exactly, FI to the
tolerance of the parity tests.  test_gpu_fim_random.py runs a fixed-seed slice of it in the suite; by hand (GPU box), for as
many trials as one likes:

    python tests/fim_random.py [trials] [seed]

(Lives under tests/ because it uses the oracle: the checker is test infrastructure.)
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fs = importlib.import_module("fit-slam_amd")
parity = importlib.import_module("fit-slam_amd.parity")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402  (the checker; this tool is test infrastructure)

REL = 1e-4


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    sc = fs.FrontierScorer(0)
    full = oracle.Table.generate()
    rec = full.records
    worst = worst_ld = 0.0
    t0 = time.time()
    for t in range(trials):
        # table: the reference one, or one with 30 % holes (the guarded kernels)
        holes = rng.random() < 0.3
        if holes:
            keep = rng.random(rec.shape[0]) < 0.7
            keep[-1] = True
            r = np.ascontiguousarray(rec[keep])
            sc.lookup_set_records(r)
            table = oracle.Table.from_records(r)
        else:
            sc.lookup_generate()
            table = full
        # cloud: uniform box + a few dense clumps (crowded voxels, multi-pass poses)
        m = int(rng.choice([1, 63, 64, 65, 500, 5000, 40000, 150000, 400000]))
        ext = float(rng.choice([3.0, 10.0, 30.0, 80.0]))
        lm = rng.uniform(-ext, ext, size=(m, 3))
        lm[:, 2] *= rng.choice([0.1, 1.0])
        for _ in range(int(rng.integers(0, 4))):
            k = int(rng.integers(1, max(2, m // 3)))
            c = rng.uniform(-ext / 2, ext / 2, size=3)
            lm[rng.integers(0, m, size=k)] = c + rng.normal(scale=float(rng.choice([0.02, 0.3, 2.0])), size=(k, 3))
        lm = lm.astype(np.float32)
        n = int(rng.choice([1, 2, 7, 64, 300]))
        poses = np.zeros((n, 7))
        poses[:, :3] = rng.uniform(-ext / 2, ext / 2, size=(n, 3))
        q = rng.normal(size=(n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        poses[:, 3:] = q
        max_dist = float(rng.choice([1.0, 3.0, 14.0, 30.0, 400.0]))
        max_angle = float(rng.choice([0.2, 1.0, 1.5, np.pi / 2, 2.2, 4.0]))
        sc.upload_landmarks(lm)
        sc.set_fim_params(max_dist, max_angle)
        got = sc.score_fim(poses)
        want = oracle.pose_information(table, lm, poses, max_dist, max_angle, n_threads=16)
        tag = f"trial {t}: m={m} ext={ext} n={n} dist={max_dist} angle={max_angle:.2f} holes={holes}"
        assert np.array_equal(got["n_visible"], want["n_visible"]), tag
        assert np.array_equal(got["n_voxels"], want["n_voxels"]), tag
        scale = np.maximum(np.abs(want["info_f64"]), 1e-6)
        drift = np.abs(want["info_ref"] - want["info_f64"]) / scale
        err = np.abs(got["info_ref"] - want["info_f64"]) / scale
        assert np.all(err <= REL + drift), (tag, float(err.max()))
        tr = np.maximum(np.abs(want["trace"]), 1e-6)
        assert np.max(np.abs(got["trace"] - want["trace"]) / tr) <= REL, tag
        worst = max(worst, float(err.max()))
        # D-optimality: the one rule of fit-slam_amd/parity.py (plain 1e-4 of max(1, |log det|) plus kappa(F) * 2^-24)
        gate = parity.logdet_gate(got["logdet"], want["logdet"], want["fim"])
        both = np.isfinite(want["logdet"]) & np.isfinite(got["logdet"])
        e_ld = np.abs(got["logdet"][both] - want["logdet"][both])
        assert gate["logdet_share_within_rule"] == 1.0, (tag, gate)
        if both.any():
            cond = parity.fim_condition(want["fim"][both])
            worst_ld = max(worst_ld, float((e_ld / np.maximum(1.0, np.abs(want["logdet"][both])))[cond < 1e3].max(initial=0.0)))
        # the same poses asked for info_ref alone: the INFO_ONLY worker (exact table-box cull in a general camera frame) where the
        # table is finite and the cone common, the general one otherwise — n_voxels exactly, the information to the same tolerance
        io = sc.score_fim(poses, info_only=True)
        assert np.array_equal(io["n_voxels"], want["n_voxels"]), tag + " (info only)"
        err_io = np.abs(io["info_ref"] - want["info_f64"]) / scale
        assert np.all(err_io <= REL + drift), (tag + " (info only)", float(err_io.max()))
        worst = max(worst, float(err_io.max()))
        print(f"ok {tag}  visible max {int(want['n_visible'].max())}  voxels max {int(want['n_voxels'].max())}  "
              f"multi-pass {sc.get_counter(4)} hbm {sc.get_counter(5)}  ({time.time() - t0:.0f} s)", flush=True)
    print(f"{trials} trials passed, worst relative FI error {worst:.2e}, worst log det error at cond < 1e3 {worst_ld:.2e} of max(1, |ld|)")


if __name__ == "__main__":
    main()
