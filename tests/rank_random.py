"""By hand on the GPU box: random frontier lists through fs_rank_candidates (U1 utilities, weighted cost, stable ascending order —
FrontierCostsManager.cpp:126-205) against oracle.u1_costs + a stable argsort, bit for bit: list lengths on both sides of the
one-workgroup ranking (1024) and the radix sort, random weights and speed limits, blacklisted / unachievable frontiers, many exact
ties (the order must be the STABLE one), all-equal columns (the two degenerate-normalisation branches), arrival 0, utilities on the
edge of [0, 1] and beyond (FS_E_RANGE where the reference throws).
    python tests/rank_random.py [trials] [seed]
(Uses the oracle: test infrastructure.)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

fs = importlib.import_module("fit-slam_amd")


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    s = fs.FrontierScorer(0)
    thrown = 0
    for t in range(trials):
        n = int(rng.choice([1, 2, 3, 17, 63, 64, 65, 500, 1023, 1024, 1025, 2047, 2048, 2049, 5000, 20000]))
        max_gt = float(rng.choice([400.0, 4000.0, 1.0, 37.5]))
        rec = np.zeros(n, dtype=fs.capi.RECORD_DTYPE)
        arrival = rng.integers(0, int(max_gt) + 1, size=n)
        if rng.random() < 0.2:
            arrival[:] = arrival[0]                                     # max - min == 0 branch (:139-140) when it equals max_gt
        if rng.random() < 0.1:
            arrival[:] = int(max_gt)
        if rng.random() < 0.1:
            arrival[rng.integers(0, n)] = int(max_gt) + int(rng.integers(1, 5))     # au > 1: the reference throws
        ach = (rng.random(n) < rng.choice([0.0, 0.5, 0.9, 1.0])).astype(np.uint8)
        black = (rng.random(n) < rng.choice([0.0, 0.1, 0.5])).astype(np.uint8)
        rec["arrival"] = arrival
        rec["flags"] = np.where(ach != 0, 1, 0)                        # FS_FLAG_ACHIEVABLE
        plen = rng.uniform(0.5, 30.0, size=n)
        phead = rng.uniform(0.0, np.pi, size=n)
        mode = rng.integers(0, 5)
        if mode == 0:
            plen = np.round(plen)                                       # many exact ties
            phead = np.round(phead)
        elif mode == 1:
            plen[:] = plen[0]; phead[:] = 0.0                           # max == min distance with heading 0: the pu = 1 branch (:152-153) when pi / wz cancels
        elif mode == 2:
            phead = rng.uniform(0.0, 1.3 * np.pi, size=n)               # beyond pi: pu < 0 for the longest paths -> throws
        alpha = float(rng.choice([0.0, 0.25, 0.5, 1.0]))
        beta = float(rng.choice([1.0, 0.5, 3.0]))
        vx, wz = float(rng.choice([0.5, 0.26, 1.0])), float(rng.choice([0.5, 1.0, 1.82]))
        rc, want = oracle.u1_costs(arrival.astype(np.float64), ach, plen, phead, max_gt, blacklisted=black, alpha=alpha, beta=beta, max_vx=vx, max_wz=wz)
        s.set_arrival_limits(max_gt, 0.7 * max_gt)
        tag = f"trial {t}: n={n} max_gt={max_gt} alpha={alpha} beta={beta} vx={vx} wz={wz} mode={mode}"
        try:
            got = s.rank_candidates(rec, plen, phead, black, alpha=alpha, beta=beta, max_vx=vx, max_wz=wz)
        except fs.FsError as e:
            assert rc != 0 and e.code == -6, (tag, rc, e)                # FS_E_RANGE exactly where the oracle says the reference throws
            thrown += 1
            continue
        assert rc == 0, (tag, "the reference throws here, the library did not")
        for k in ("weighted_cost", "arrival_utility", "distance_utility"):
            assert np.array_equal(got[k], want[k]), (tag, k)
        assert np.array_equal(got["order"], np.argsort(want["weighted_cost"], kind="stable")), tag
    s.close()
    print(f"{trials} trials passed ({thrown} of them where the reference throws and the library returns FS_E_RANGE)")


if __name__ == "__main__":
    main()
