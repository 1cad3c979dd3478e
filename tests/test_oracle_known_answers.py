"""Pins the CPU oracle against everything the reference's own (manual) tests and sources fix for
this path: the analytic inputs of DEP/tests/main_fim_computation.cpp:8-56 and DEP/tests/fim_viz.cpp:70-100,
the lookup-table generator's bounds (DEP/src/fisher_information/GenerateLookupMain.cpp:9) and the
constants of SURVEY.md Appendix C.  The reference holds no golden vectors with expected values
(SURVEY.md §4), so these closed forms are the pin."""
import math

import numpy as np
import pytest


@pytest.mark.parametrize("p,expect", [((0.3, 0, 0), 24.222222), ((0.3, 0.3, 0.3), 9.407407),
                                      ((1.2, -0.9, 0.3), 2.854701), ((3, 0.3, -0.6), 2.211640),
                                      ((21, 14.7, 14.7), 2.002290)])
def test_info_closed_form(oracle, pyref, p, expect):
    """App. C.1: trace(J^T J) = 2 + 2/|p|^2 (FisherInformationHelpers.cpp:71-96,114-123)."""
    v = oracle.information_of_point_local(p)
    n2 = sum(float(np.float32(c)) ** 2 for c in p)
    assert abs(v - expect) <= 2e-6 * expect
    assert abs(v - (2 + 2 / n2)) <= 5e-7 * (2 + 2 / n2)
    assert abs(float(pyref.info_point_local(p)) - v) <= 1e-6 * v


def test_info_origin_is_nan(oracle):
    assert math.isnan(oracle.information_of_point_local((0, 0, 0)))


def test_fim_block_form(oracle):
    """App. C.3: F(p) = [[P/n^2, -S/n^2],[S/n^2, P]], rank 2, trace 2 + 2/n^2."""
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = rng.uniform(-5, 5, size=3)
        F = oracle.fim_point_local_f64(p)
        n2 = p @ p
        P = np.eye(3) - np.outer(p, p) / n2
        S = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
        B = np.block([[P / n2, -S / n2], [S / n2, P]])
        np.testing.assert_allclose(F, B, atol=1e-12)
        assert np.linalg.matrix_rank(F, tol=1e-9) == 2
        assert abs(np.trace(F) - (2 + 2 / n2)) < 1e-12


def test_main_fim_computation_inputs(oracle):
    """DEP/tests/main_fim_computation.cpp: pose (34233, 32111, 0), yaw 0, landmarks on a 0.35 m grid.
    App. C.2: float32 world coordinates quantise to 1/256 m, so the reference itself is ~0.3 % off the
    ideal 2 + 2/(dx^2+dy^2); the float32-faithful oracle gives 10.135938 at (0.35, 0.35)."""
    pose = [34233.0, 32111.0, 0.0, 0, 0, 0, 1.0]
    v = oracle.information_of_point_local_world(pose, (np.float32(34233.0) + np.float32(0.35), np.float32(32111.0) + np.float32(0.35), 0.0))
    assert abs(v - 10.135938) < 5e-6
    worst = 0.0
    dx = np.float32(-5)
    while dx <= 5:
        dy = np.float32(-5)
        while dy <= 5:
            lx, ly = np.float32(34233.0) + dx, np.float32(32111.0) + dy
            ex, ey = float(lx) - 34233.0, float(ly) - 32111.0      # what float32 storage really holds
            if ex * ex + ey * ey > 1e-3:
                got = oracle.information_of_point_local_world(pose, (lx, ly, 0.0))
                want = 2 + 2 / (ex * ex + ey * ey)
                worst = max(worst, abs(got - want) / want)
            dy = np.float32(dy + np.float32(0.35))
        dx = np.float32(dx + np.float32(0.35))
    assert worst < 2e-4      # against the quantised offsets the float32 path is accurate


def test_fim_viz_inputs(oracle):
    """DEP/tests/fim_viz.cpp:70-100 — 74 landmarks around (3,2,1): info_sum = sum(2 + 2/|d|^2)."""
    pose = [3.0, 2.0, 1.0, 0, 0, 0, 1.0]
    total, ideal, count = 0.0, 0.0, 0
    for dx in (0.0, 2.5, 5.0):
        for dy in (-5.0, -2.5, 0.0, 2.5, 5.0):
            for dz in (-5.0, -2.5, 0.0, 2.5, 5.0):
                if dx == 0 and dy == 0 and dz == 0:
                    continue
                total += oracle.information_of_point_local_world(pose, (3 + dx, 2 + dy, 1 + dz))
                ideal += 2 + 2 / (dx * dx + dy * dy + dz * dz)
                count += 1
    assert count == 74
    assert abs(total - ideal) <= 1e-5 * ideal


def test_crowding_factor(oracle, pyref):
    """App. C.4 (FisherInfoManager.hpp:102-106)."""
    for k, e in [(1, 1.0), (2, 0.476589), (3, 0.244577), (5, 0.072520), (10, 0.004944)]:
        assert abs(oracle.factor_from_num(k) - e) < 1e-6
        assert oracle.factor_from_num(k) == float(pyref.crowding_factor(k))
    assert oracle.factor_from_num(400) == 0.0      # underflows to +0.0f: ranks beyond ~340 add nothing
    assert oracle.factor_from_num(330) > 0.0


def test_voxel_coordinate(oracle, pyref):
    """getVoxelCoordinate (FisherInfoManager.hpp:108-123): step is the double 0.300000011920929."""
    rng = np.random.default_rng(1)
    for _ in range(500):
        x, y, z = rng.uniform(-16, 22, size=3)
        key, idx = oracle.voxel_coordinate(x, y, z)
        assert tuple(key) == tuple(np.float32(v) for v in pyref.voxel_key(x, y, z))
        np.testing.assert_array_equal(key, (idx.astype(np.float64) * float(np.float32(0.3))).astype(np.float32))
    key, idx = oracle.voxel_coordinate(0.15, -0.15, 0.44)
    assert list(idx) == [1, -1, 1]                 # std::round: halves away from zero (0.15f*3.33 = 0.50000001)
    key, _ = oracle.voxel_coordinate(-0.1, 0.0, 0.0)
    assert key[0] == 0.0                           # -0.0f == 0.0f


def test_reference_table(ref_table, oracle):
    """generateLookupTable(0, 21, -8.5*1.732, 8.5*1.732, ...) with the reference's float loops
    (FisherInfoManager.cpp:117-229): 71 x 100 x 100 lattice minus the NaN origin plus the trailing
    (0,0,0) = max record -> 710 000 records; y/z reach -15.0 but only +14.7 (App. A.2 quirk)."""
    rec = ref_table.records
    assert rec.shape == (710_000, 4)
    assert ref_table.num_entries == 710_000
    body = rec[:-1]
    assert np.unique(body[:, 0]).size == 71 and np.unique(body[:, 1]).size == 100 and np.unique(body[:, 2]).size == 100
    assert body[:, 0].min() == 0.0 and abs(body[:, 0].max() - 21.0) < 1e-5
    assert abs(body[:, 1].min() + 15.0) < 1e-5 and abs(body[:, 1].max() - 14.7) < 1e-5
    np.testing.assert_array_equal(rec[-1, :3], [0, 0, 0])
    assert abs(rec[-1, 3] - 24.222222) < 2e-6 and rec[-1, 3] == body[:, 3].max()
    assert not np.isnan(rec[:, 3]).any()
    # every value is 2 + 2/|key|^2
    n2 = (body[:, :3].astype(np.float64) ** 2).sum(axis=1)
    np.testing.assert_allclose(body[:, 3], 2 + 2 / n2, rtol=1e-6)
    # lookups: -15.0 hits, +15.0 misses
    k_lo, _ = oracle.voxel_coordinate(3.0, -15.0, 0.0)
    k_hi, _ = oracle.voxel_coordinate(3.0, 15.0, 0.0)
    assert not math.isnan(ref_table.find(k_lo)) and math.isnan(ref_table.find(k_hi))
    assert abs(ref_table.find((0, 0, 0)) - 24.222222) < 2e-6


def test_table_generation_matches_python_transcription(oracle, pyref):
    bounds = (0.0, 1.5, -1.2, 1.2, -1.2, 1.2)
    t = oracle.Table.generate(bounds)
    want, n_rec = pyref.generate_table(bounds)
    rec = t.records
    assert rec.shape[0] == n_rec
    got = {tuple(float(v) + 0.0 for v in r[:3]): r[3] for r in rec}
    assert set(got) == set(want)
    for k, v in want.items():
        assert abs(got[k] - float(v)) <= 1e-6 * abs(float(v))


def test_table_file_format_roundtrip(oracle, tmp_path):
    """The .dat is raw {float key[3]; float value} records (FisherInfoManager.cpp:185-186,245-251)."""
    t = oracle.Table.generate((0.0, 0.9, -0.6, 0.6, -0.6, 0.6))
    p = tmp_path / "fi.dat"
    t.records.tofile(p)
    assert p.stat().st_size == 16 * t.records.shape[0]
    t2 = oracle.Table.from_records(np.fromfile(p, dtype=np.float32).reshape(-1, 4))
    np.testing.assert_array_equal(t.records, t2.records)
    # later duplicates overwrite: the trailing (0,0,0) record defines the origin's value
    assert t2.find((0, 0, 0)) == t.records[-1, 3]


def test_default_fan_shape(oracle):
    """App. C.5: delta_theta 0.10 accumulated -> 63 rays; window int(1.04/0.10) = 10; L = 40."""
    P = oracle.RayParams()
    assert P.n_yaw == 63 and P.window == 10
    th = oracle.theta_list(0.10, 63)
    acc = 0.0
    for i in range(63):
        assert th[i] == acc
        acc += 0.10
    assert th[62] != 62 * 0.10          # accumulated, not i*delta
    assert int(2.0 / 0.05) == 40
