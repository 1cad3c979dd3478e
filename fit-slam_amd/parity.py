"""THE tolerance rule of the floating-point outputs — one definition, used by tests/ and by bench.py's gates alike.

north_star: "FIM trace / D-optimality within 1e-4 relative".  info_ref and the trace are sums of positive terms and meet
the plain figure.  log det is different in kind: for a symmetric positive definite F,

    d(log det F) = tr(F^-1 dF)   =>   |d log det| <= 6 * kappa(F) * max|dF_ij| / max|F_ij|,

so a relative perturbation eps of the ENTRIES moves log det by up to ~kappa * eps in absolute terms.  The path computes F in
float32 — as the reference does (Eigen::Matrix<float, 6, 6>, FIP/src/fisher_information/FisherInformationHelpers.cpp:99-123)
— so its entries carry eps = 2^-24 before any summation order enters.  Measured (tools/logdet_probe.py, profiles/r05/
logdet_probe_*.json): the only poses beyond plain 1e-4 are those with THREE OR FOUR visible landmarks (rank 6 needs three),
kappa 1e5 .. 5e5, and their error is what rounding the float64 oracle's own F to float32 entries gives (4.5e-4 against 4.3e-4;
1.4e-3 against 2.5e-3) — no accumulation order can do better with a float32 F.  Hence

    |ld_got - ld_want| <= 1e-4 * max(1, |ld_want|) + 2^-24 * kappa(F_want)

The second term is below 6e-5 up to kappa = 1e3 (plain 1e-4 holds there by itself) and only matters for nearly singular F.
Both shares — within plain 1e-4 and within the rule — are reported wherever the gate runs.
"""
from __future__ import annotations

import numpy as np

REL = 1e-4                 # north_star's relative tolerance
F32_EPS = 2.0 ** -24       # unit round-off of the float32 F entries


def fim_condition(F) -> np.ndarray:
    """kappa_2 of each 6x6 F (float64 oracle matrices [n][6][6])."""
    lam = np.linalg.eigvalsh(np.asarray(F, dtype=np.float64))
    return lam[:, -1] / np.maximum(lam[:, 0], 1e-300)


def logdet_tolerance(logdet_want, cond) -> np.ndarray:
    ld = np.asarray(logdet_want, dtype=np.float64)
    return REL * np.maximum(1.0, np.abs(ld)) + F32_EPS * np.asarray(cond, dtype=np.float64)


def logdet_gate(logdet_got, logdet_want, F_want, consider=None, n_visible=None) -> dict:
    """log det of the HIP path against the float64 oracle over the poses in `consider` (default: all).

    Returns the figures every report carries and `ok`:
      * both finite -> within logdet_tolerance;
      * the oracle finite and kappa < 1e5 -> the HIP path must be finite too (beyond that the 1e-6 pivot rule may fall
        either way on a float32 F);
      * fewer than three visible landmarks (`n_visible`, when given) can never give rank 6: the HIP path must say -inf.
    """
    got = np.asarray(logdet_got, dtype=np.float64)
    want = np.asarray(logdet_want, dtype=np.float64)
    sel = np.ones(want.shape, dtype=bool) if consider is None else np.asarray(consider, dtype=bool)
    cond = fim_condition(F_want)
    fin_w, fin_g = np.isfinite(want), np.isfinite(got)
    both = sel & fin_w & fin_g
    err = np.abs(got[both] - want[both])
    rel = err / np.maximum(1.0, np.abs(want[both]))
    tol = logdet_tolerance(want[both], cond[both])
    sure = sel & fin_w & (cond < 1e5)
    finite_where_sure = bool(np.all(fin_g[sure]))
    never_finite_where_singular = True if n_visible is None else bool(not np.any(fin_g[sel & (np.asarray(n_visible) < 3)]))
    worst = int(np.argmax(err / tol)) if err.size else -1
    out = {"logdet_max_rel_err": float(rel.max()) if rel.size else 0.0,
           "logdet_share_within_1e-4": float(np.mean(rel <= REL)) if rel.size else 1.0,
           "logdet_share_within_rule": float(np.mean(err <= tol)) if err.size else 1.0,
           "logdet_worst_err_over_tol": float((err / tol).max()) if err.size else 0.0,
           "logdet_worst_kappa": float(cond[both][worst]) if err.size else 0.0,
           "logdet_kappa_max": float(cond[both].max()) if err.size else 0.0,
           "logdet_finite_candidates": int(both.sum()),
           "logdet_singularity_agrees": bool(np.array_equal(fin_g[sel], fin_w[sel])),
           "logdet_rule": "|err| <= 1e-4 * max(1, |log det|) + 2^-24 * kappa(F)  (fit-slam_amd/parity.py)"}
    out["ok"] = bool((err.size == 0 or np.all(err <= tol)) and finite_where_sure and never_finite_where_singular)
    return out


def rel_err(got, want, floor=1e-6) -> float:
    """max |got - want| / max(|want|, floor) — the plain relative error of info_ref / trace columns."""
    g, w = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.max(np.abs(g - w) / np.maximum(np.abs(w), floor))) if w.size else 0.0
