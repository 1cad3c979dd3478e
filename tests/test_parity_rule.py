"""The ONE tolerance rule of the floating-point outputs (fit-slam_amd/parity.py) — used by the GPU tests and by every gate of
bench.py; here its behaviour on hand-made matrices, and bench.py's reaction to a red gate (no GPU involved)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spd(kappa, scale=1.0, seed=0):
    """A symmetric positive definite 6x6 with condition number kappa."""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
    lam = scale * np.geomspace(1.0, kappa, 6)
    return (q * lam) @ q.T


def test_plain_tolerance_for_well_conditioned_matrices_and_the_floor_beyond():
    P = importlib.import_module("fit-slam_amd.parity")
    F = np.stack([_spd(10.0, seed=1), _spd(1e3, seed=2), _spd(1e6, seed=3)])
    ld = np.array([np.linalg.slogdet(f)[1] for f in F])
    np.testing.assert_allclose(P.fim_condition(F), [10.0, 1e3, 1e6], rtol=1e-6)
    tol = P.logdet_tolerance(ld, P.fim_condition(F))
    np.testing.assert_allclose(tol, 1e-4 * np.maximum(1, np.abs(ld)) + 2.0 ** -24 * np.array([10.0, 1e3, 1e6]), rtol=1e-9)
    # within plain 1e-4 everywhere: green, both shares 1
    g = P.logdet_gate(ld + 0.5e-4 * np.maximum(1, np.abs(ld)), ld, F)
    assert g["ok"] and g["logdet_share_within_1e-4"] == 1.0 and g["logdet_share_within_rule"] == 1.0
    # 2e-4 on the well-conditioned one: red, whatever the others do
    bad = ld.copy(); bad[0] += 2e-4 * max(1, abs(ld[0]))
    g = P.logdet_gate(bad, ld, F)
    assert not g["ok"] and g["logdet_share_within_rule"] < 1.0
    # 1e-2 on the kappa = 1e6 one: beyond plain 1e-4, inside kappa * 2^-24 = 6e-2 — green, and the line says which share is which
    bad = ld.copy(); bad[2] += 1e-2
    g = P.logdet_gate(bad, ld, F)
    assert g["ok"] and g["logdet_share_within_1e-4"] < 1.0 and g["logdet_share_within_rule"] == 1.0
    assert abs(g["logdet_worst_kappa"] - 1e6) / 1e6 < 1e-6 and g["logdet_kappa_max"] >= 1e6 * (1 - 1e-6)


def test_singularity_rules():
    P = importlib.import_module("fit-slam_amd.parity")
    F = np.stack([_spd(50.0, seed=4), np.zeros((6, 6)), _spd(1e7, seed=5)])
    ld = np.array([np.linalg.slogdet(F[0])[1], -np.inf, np.linalg.slogdet(F[2])[1]])
    nvis = np.array([40, 2, 3])
    assert P.logdet_gate(ld.copy(), ld, F, n_visible=nvis)["ok"]
    # a finite value where fewer than three landmarks are visible is wrong
    got = ld.copy(); got[1] = 0.0
    assert not P.logdet_gate(got, ld, F, n_visible=nvis)["ok"]
    # -inf where the oracle is finite and the matrix is decently conditioned is wrong ...
    got = ld.copy(); got[0] = -np.inf
    assert not P.logdet_gate(got, ld, F, n_visible=nvis)["ok"]
    # ... but may happen beyond kappa = 1e5 (the 1e-6 pivot rule on a float32 F can fall either way)
    got = ld.copy(); got[2] = -np.inf
    g = P.logdet_gate(got, ld, F, n_visible=nvis)
    assert g["ok"] and not g["logdet_singularity_agrees"]
    # `consider` leaves poses out (off-map candidates)
    got = ld.copy(); got[0] += 1.0
    assert P.logdet_gate(got, ld, F, consider=np.array([False, True, True]), n_visible=nvis)["ok"]


def test_bench_finds_every_red_gate_in_a_line():
    sys.path.insert(0, ROOT)
    import bench
    line = {"parity": {"ok": True}, "cpu_baseline": {"value": 1.0},
            "other_operating_points": {"reference_request_visibility": {"parity": {"ok": False}, "fused_parity": {"ok": True}}},
            "strong_scaling": {"parity": {"ok": False}}, "ranked_step": None}
    assert sorted(bench.red_gates(line)) == ["other_operating_points.reference_request_visibility.parity", "strong_scaling.parity"]
    line["other_operating_points"]["reference_request_visibility"]["parity"]["ok"] = True
    line["strong_scaling"]["parity"]["ok"] = True
    assert bench.red_gates(line) == []


def test_host_cpu_topology_reports_what_the_box_grants():
    sys.path.insert(0, ROOT)
    import bench
    t = bench.host_cpu_topology()
    assert t["logical_cpus_usable"] >= 1 and (t["physical_cores_usable"] is None or 1 <= t["physical_cores_usable"] <= t["logical_cpus_usable"])
    assert t["cgroup_cpu_quota_cores"] is None or t["cgroup_cpu_quota_cores"] > 0
