"""Oracle unit tests for the ray-cast half: Costmap2D semantics (SURVEY.md App. B — third party,
pinned here), the Bresenham walk, the visitor, the window/argmax rule, and a cross-check of the C
oracle against the independent pure-Python transcription (oracle/pyref.py)."""
import math

import numpy as np
import pytest


def _grid(oracle, cells, origin=(0.0, 0.0, 0.0), res=0.05):
    return oracle.Grid(np.asarray(cells, np.uint8), origin=origin, resolution=res)


def test_world_to_map_semantics(oracle):
    g = _grid(oracle, np.zeros((10, 20)), origin=(-0.5, 1.0, 0.0), res=0.05)
    assert oracle.world_to_map(g, -0.5, 1.0) == (True, 0, 0, 0)
    assert oracle.world_to_map(g, -0.5 + 0.0999, 1.0 + 0.05)[1:3] == (1, 1)      # truncation, not rounding
    assert not oracle.world_to_map(g, -0.5000001, 1.2)[0]                          # wx < origin
    assert oracle.world_to_map(g, -0.5 + 20 * 0.05 - 1e-9, 1.0)[1] == 19
    assert not oracle.world_to_map(g, -0.5 + 20 * 0.05 + 1e-9, 1.0)[0]             # mx == size_x
    assert not oracle.world_to_map(g, float("nan"), 1.0)[0]
    assert not oracle.world_to_map(g, 1e300, 1.0)[0]


def test_sign_zero_is_minus_one_and_single_cell_ray(oracle):
    """sign(0) = -1 (Helpers.hpp:113-116); dist == 0 -> one visit (bresenham2D's trailing at(offset))."""
    g = _grid(oracle, np.full((8, 8), 255))
    r = oracle.trace_ray(g, (0.2, 0.2, 0.0), (0.2, 0.2, 0.0), 40)
    assert r["ok"] and r["traced"] == 1 and list(r["visited"]) == [4 * 8 + 4]
    # purely vertical ray: dx == 0 -> offset_dx = -1 but abs_db = 0 never steps in x
    r = oracle.trace_ray(g, (0.2, 0.0, 0.0), (0.2, 0.39, 0.0), 40)
    assert list(r["visited"]) == [y * 8 + 4 for y in range(8)]


def test_bresenham_visits_end_plus_one_and_max_length(oracle):
    g = _grid(oracle, np.full((64, 64), 255))
    r = oracle.trace_ray(g, (0.0, 0.0, 0.0), (3.0, 1.0, 0.0), 1000)
    assert r["all"] == 60 + 1 and r["traced"] == 61             # abs_dx = 60 -> 61 visits
    r = oracle.trace_ray(g, (0.0, 0.0, 0.0), (3.0, 1.0, 0.0), 40)
    dist = math.hypot(60, 20)
    assert r["all"] == int(min(1.0, 40 / dist) * 60) + 1


def test_visitor_rules(oracle):
    """RayTracedCells::operator() (Helpers.hpp:50-77): traced test happens before the obstacle flag is
    set; the obstacle cell is not traced with (240,254,255,255) but is with (253,254,0,255)."""
    row = np.array([[255, 255, 0, 255, 250, 255, 255, 255]], dtype=np.uint8)
    g = _grid(oracle, row)
    end = (0.39, 0.0, 0.0)
    r = oracle.trace_ray(g, (0.0, 0.0, 0.0), end, 40)
    assert (r["traced"], r["hit"], r["unknown"], r["all"]) == (3, True, 6, 8)
    r = oracle.trace_ray(g, (0.0, 0.0, 0.0), end, 40, obst=(253, 254), trace=(0, 255))
    assert (r["traced"], r["hit"]) == (8, False)
    row[0, 4] = 254
    r = oracle.trace_ray(_grid(oracle, row), (0.0, 0.0, 0.0), end, 40, obst=(253, 254), trace=(0, 255))
    assert (r["traced"], r["hit"]) == (5, True)                  # the obstacle cell itself is pushed


def test_hypot_vs_sqrt_never_changes_step_count():
    """The reference calls std::hypot (Helpers.cpp:49); glibc's hypot is within 1 ulp of, but not always
    equal to, the correctly rounded sqrt(dx^2+dy^2) the HIP path uses.  Exhaustive proof that
    (unsigned)(min(1, L/dist) * abs_da) is the same for both, for every reachable (dx, dy) and L."""
    n = 700
    d = np.arange(n, dtype=np.float64)
    dx, dy = np.meshgrid(d, d, indexing="ij")
    hyp = np.hypot(dx, dy)
    srt = np.sqrt(dx * dx + dy * dy)
    a = np.maximum(dx, dy)
    for L in (1.0, 7.0, 39.0, 40.0, 41.0, 100.0, 160.0, 320.0, 640.0):
        with np.errstate(divide="ignore"):
            s1 = np.where(hyp == 0, 1.0, np.minimum(1.0, L / hyp))
            s2 = np.where(srt == 0, 1.0, np.minimum(1.0, L / srt))
        np.testing.assert_array_equal((s1 * a).astype(np.uint32), (s2 * a).astype(np.uint32))
    # and the distance of scale*abs_da to the truncation boundary dwarfs 1 ulp whenever the two differ
    diff = hyp != srt
    v = (np.minimum(1.0, 40.0 / srt[diff]) * a[diff])
    assert np.min(np.abs(v - np.round(v))[np.abs(v - np.round(v)) > 0]) > 1e-9


def test_footprint_disc(oracle):
    """isRobotFootprintInLethal (Helpers.cpp:135-155): disc of radius ceil(0.6/0.05) = 12 cells; off-grid
    cells are not lethal (documented deviation from the reference's out-of-bounds read)."""
    cells = np.zeros((64, 64), np.uint8)
    g = _grid(oracle, cells)
    assert not oracle.footprint_in_lethal(g, 32, 32, 0, 12.0)
    cells[32 + 12, 32] = 254
    assert oracle.footprint_in_lethal(_grid(oracle, cells), 32, 32, 0, 12.0)
    cells[:] = 0
    cells[32 + 9, 32 + 9] = 254                                   # 81 + 81 > 144: outside the disc
    assert not oracle.footprint_in_lethal(_grid(oracle, cells), 32, 32, 0, 12.0)
    cells[32 + 8, 32 + 8] = 253                                   # only 254 counts
    assert not oracle.footprint_in_lethal(_grid(oracle, cells), 32, 32, 0, 12.0)
    assert not oracle.footprint_in_lethal(_grid(oracle, cells), 0, 0, 0, 12.0)   # border: no wrap-around reads


def test_window_first_maximum_and_yaw(oracle):
    """CostCalculator.cpp:87-119: window k = int(fov/delta), no wrap-around, strict > keeps the first
    maximum, yaw = maxIndex*delta + fov/2."""
    n = 128
    g = _grid(oracle, np.full((n, n), 255), origin=(-3.2, -3.2, 0.0))
    P = oracle.RayParams(polygon=(-10, -10, 10, 10))
    r = oracle.arrival_information(g, P, [[0.0, 0.0, 0.0]])
    rc = r["ray_counts"][0, 0]
    sums = np.array([rc[i:i + 10].sum() for i in range(63 - 10 + 1)])
    assert r["arrival"][0] == sums.max() and r["argmax"][0] == int(np.argmax(sums))
    assert r["yaw"][0] == (int(np.argmax(sums)) * 0.10) + (1.04 / 2)
    with pytest.raises(ValueError):
        oracle.arrival_information(g, oracle.RayParams(delta_theta=1.0, camera_fov=8.5), [[0.0, 0.0, 0.0]])


def test_max_arrival_and_limits(oracle):
    """setMaxArrivalInformation (CostCalculator.cpp:123-191): max_gt = 1.2 max, min_gt = 0.7 max_gt;
    off-map calibration point -> 0 and limits untouched."""
    n = 128
    g = _grid(oracle, np.zeros((n, n)), origin=(-3.2, -3.2, 0.0))
    P = oracle.RayParams()
    m = oracle.max_arrival_information(g, P)
    assert m["max_value"] > 300 and m["max_gt"] == m["max_value"] * 1.2 and m["min_gt"] == 0.70 * m["max_gt"]
    g2 = _grid(oracle, np.zeros((n, n)), origin=(1.0, 1.0, 0.0))
    assert oracle.max_arrival_information(g2, P) == dict(max_value=0.0, max_gt=0.0, min_gt=0.0)


def test_achievability_rules(oracle):
    n = 128
    cells = np.full((n, n), 255, np.uint8)
    cells[64 + 5, 64] = 254
    g = _grid(oracle, cells, origin=(-3.2, -3.2, 0.0))
    P = oracle.RayParams()
    goals = [[0.0, 0.0, 0.0]] * 3
    r = oracle.arrival_information(g, P, goals, frontier_size=[5, 10, 5], achievable_in=[1, 1, 1], min_gt=0.0)
    assert list(r["achievable"]) == [0, 1, 0]                    # lethal footprint and size < 10
    r = oracle.arrival_information(g, P, goals, frontier_size=[50, 50, 50], min_gt=1e9)
    assert list(r["achievable"]) == [0, 0, 0]                    # arrival < min_gt
    r = oracle.arrival_information(g, P, goals, frontier_size=[50, 50, 50], blacklisted=[0, 1, 0], min_gt=0.0)
    assert list(r["status"]) == [0, 2, 0] and r["arrival"][1] == 0 and r["yaw"][1] == 0.0


def test_faithful_equals_clean(fs, oracle):
    w = fs.synth.make_small_2d(5)
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(polygon=w.polygon)
    a = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=300, faithful=True)
    b = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=300, faithful=False, n_threads=4)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.parametrize("seed", [3, 4])
def test_c_oracle_matches_python_transcription_2d(fs, oracle, pyref, seed):
    w = fs.synth.make_small_2d(seed, n=72, n_cand=24)
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(polygon=w.polygon)
    got = oracle.arrival_information(G, P, w.goals, w.frontier_size, min_gt=250.0, faithful=True)
    cm = pyref.Costmap(w.cells, w.origin, w.resolution)
    for i in range(w.goals.shape[0]):
        r = pyref.arrival_information(cm, tuple(w.goals[i]), polygon=w.polygon, frontier_size=int(w.frontier_size[i]), min_gt=250.0)
        assert (r["arrival"], r["argmax"], r["status"], int(r["achievable"])) == \
            (got["arrival"][i], got["argmax"][i], got["status"][i], got["achievable"][i])
        assert r["yaw"] == got["yaw"][i]
        if r["counts"] is not None:
            np.testing.assert_array_equal(np.array(r["counts"]), got["ray_counts"][i])


def test_c_oracle_matches_python_transcription_3d(fs, oracle, pyref):
    w = fs.synth.make_workload("C1", n_cand=12)
    elev = (-0.3, 0.2)
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(max_camera_depth=1.0, delta_theta=w.delta_theta, n_rays=w.n_yaw, elev=elev, polygon=w.polygon)
    got = oracle.arrival_information(G, P, w.goals, w.frontier_size)
    cm = pyref.Costmap(w.cells, w.origin, w.resolution)
    for i in range(w.goals.shape[0]):
        r = pyref.arrival_information(cm, tuple(w.goals[i]), depth=1.0, delta_theta=w.delta_theta, n_rays=w.n_yaw,
                                      elev=elev, polygon=w.polygon, frontier_size=int(w.frontier_size[i]))
        assert (r["arrival"], r["argmax"], r["status"]) == (got["arrival"][i], got["argmax"][i], got["status"][i])
        np.testing.assert_array_equal(np.array(r["counts"]), got["ray_counts"][i])


def test_3d_reduces_to_2d(fs, oracle):
    """a 3-D grid whose slices all equal the 2-D costmap, one ring at elevation 0
    -> the same counts as the 2-D reference path on that costmap."""
    w = fs.synth.make_small_2d(8, n=80, n_cand=40)
    P = oracle.RayParams(polygon=w.polygon)
    g2 = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    a = oracle.arrival_information(g2, P, w.goals, w.frontier_size, min_gt=100.0)
    nz = 6
    cells3 = np.repeat(w.cells, nz, axis=0)
    g3 = oracle.Grid(cells3, origin=(w.origin[0], w.origin[1], -0.1), resolution=w.resolution)
    goals3 = w.goals.copy()
    goals3[:, 2] = 0.07            # inside slice 3
    b = oracle.arrival_information(g3, P, goals3, w.frontier_size, min_gt=100.0)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])


def test_u1_costs(oracle):
    """FrontierCostsManager.cpp:126-205 on a hand-checkable case."""
    arrival = np.array([100.0, 50.0, 0.0, 80.0])
    ach = np.array([1, 1, 0, 1], np.uint8)
    plen = np.array([2.0, 6.0, 1.0, 4.0])
    head = np.array([0.0, math.pi, 1.0, math.pi / 2])
    rc, r = oracle.u1_costs(arrival, ach, plen, head, max_arrival_gt=200.0)
    assert rc == 0
    denom = 6.0 / 0.5 + math.pi / 0.5
    au = arrival / 200.0
    pu = 1.0 - (plen / 0.5 + head / 0.5) / denom
    cost = 1 / (1.0 * (0.25 * au + 0.75 * pu))
    for i in (0, 1, 3):
        assert r["arrival_utility"][i] == au[i] and r["distance_utility"][i] == pu[i] and r["weighted_cost"][i] == cost[i]
    assert r["weighted_cost"][2] == np.finfo(np.float64).max and r["arrival_utility"][2] == -69.8
    rc, _ = oracle.u1_costs(np.array([300.0]), np.array([1], np.uint8), np.array([1.0]), np.array([0.0]), max_arrival_gt=200.0)
    assert rc == -2                                # the reference throws "ARRIVAL UTILITY ERROR"
