"""The ROS 2 adapter under fit-slam_amd/host/ros2/ LINKED AND RUN against test doubles.

tests/test_ros2_adapter_parse.py proves the three adapter sources parse against declarations; here the same sources — unchanged —
are compiled with every external #include answered by tests/ros2_fakes/ros2_fakes.hpp (definitions for exactly the shapes of
tests/ros2_decls/ros2_decls.hpp: a costmap that is an array, a node that is a parameter map, a blackboard, a builder factory, a tf
buffer with one transform, a subscription that keeps its callback, the reference's Frontier record as a struct and its EUCLIDEAN
planner restated), linked with tests/ros2_fakes/adapter_driver.cpp and the product library, and executed on the GPU:

* `CostAssignerGPU::getFrontierCosts(request, response)` through the three-step route and through `setFused(true)`, on the device
  list `[0, 0]` read from the node parameter — every frontier's arrival information, goal orientation, achievability, path
  length, weighted cost and both utilities against the ORACLE (bit for bit), the response columns against the frontiers, and the
  reference's refusals (no polygon / empty list: false; duplicate: throws);
* `FisherInformationManagerGPU`: table file from the node parameter, landmarks from a published `map_data` message (duplicates
  across key-frames and a NaN point dropped), `isPoseSafe` (both overloads) and the batch form against the oracle and the 550
  threshold;
* `FisherInfoBTPluginGPU::registerNodes` loaded instead of and next to the reference plugin (one builder per ID either way), the
  node built through the factory, `tick()`: throws without `latest_robot_pose`, SUCCESS / FAILURE by the verdict at the TF pose,
  the 700 ms back-off and `error_code_id` on FAILURE.

What this does NOT show: anything about ROS 2, nav2, tf2, pluginlib or BehaviorTree.CPP themselves — the doubles only let the
adapter's own code execute.  Row a23 stays "partial"; what changes is that every line of the adapter that talks to the C ABI
has now run, on real inputs, with checked results.
"""
import math
import os
import re
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fit-slam_amd", "host", "ros2")
FAKES = os.path.join(ROOT, "tests", "ros2_fakes", "ros2_fakes.hpp")
DRIVER_SRC = os.path.join(ROOT, "tests", "ros2_fakes", "adapter_driver.cpp")
SOURCES = ["CostAssignerGPU.cpp", "FisherInfoManagerGPU.cpp", "FisherInfoBTPluginGPU.cpp"]
OURS = ("fitslam_frontier.h", "fitslam_frontier_ros2/")


def build_driver(workdir, sanitize: bool = False) -> str:
    """g++ over the three adapter sources (unchanged) + the driver, external includes -> the test doubles; links the product library.
    sanitize: the adapter, the doubles and the driver instrumented with AddressSanitizer + UndefinedBehaviorSanitizer (gcc's; the
    product library itself stays as built)."""
    import importlib
    lib = importlib.import_module("fit-slam_amd")._build.build()
    files = [os.path.join(PKG, "src", s) for s in SOURCES] + [DRIVER_SRC]
    inc = os.path.join(PKG, "include", "fitslam_frontier_ros2")
    headers = [os.path.join(inc, f) for f in os.listdir(inc)]
    shim = os.path.join(str(workdir), "shim")
    for path in files + headers:
        for name in re.findall(r'^\s*#include\s*[<"]([^>"]+)[>"]', open(path).read(), flags=re.M):
            if name.startswith(OURS) or ("/" not in name and "." not in name):
                continue
            p = os.path.join(shim, name)
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w") as f:
                f.write(f'#include "{FAKES}"\n')
    exe = os.path.join(str(workdir), "adapter_driver_asan" if sanitize else "adapter_driver")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"] if sanitize else []
    cmd = ["g++", "-std=c++17", "-O1", "-g", *san, "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", shim, "-I", os.path.join(PKG, "include"),
           "-I", os.path.join(ROOT, "include"), *files, "-o", exe, "-L", os.path.dirname(lib), "-l" + os.path.basename(lib)[3:-3],
           "-Wl,-rpath," + os.path.dirname(lib), "-pthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-4000:]
    assert "warning" not in res.stderr, res.stderr[-4000:]
    return exe


def test_adapter_links_against_the_test_doubles(tmp_path):
    """No GPU needed: the adapter sources, the doubles and the driver compile without a warning and link against the product library
    (every fs_* symbol the adapter calls resolves)."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = build_driver(tmp_path)
    assert os.path.exists(exe)
    # the doubles define, they do not re-declare: the declarations file is included, not copied
    text = open(FAKES).read()
    assert '#include "../ros2_decls/ros2_decls.hpp"' in text and "struct Costmap2D {" not in text and "class Frontier {" not in text


def _euclid(start_xy, robot_yaw, goal, achievable):
    """setPlanForFrontierEuclidean (DEP/src/CostCalculator.cpp:445-484); robot_yaw = the yaw of the start pose's quaternion."""
    dmax = np.finfo(np.float64).max
    if not achievable:
        return False, dmax, dmax
    length = math.sqrt(math.pow(start_xy[0] - goal[0], 2) + math.pow(start_xy[1] - goal[1], 2))
    if length < 0.5:
        return False, dmax, dmax
    ry = robot_yaw + (2 * math.pi if robot_yaw < 0 else 0)
    gy = math.atan2(goal[1] - start_xy[1], goal[0] - start_xy[0])
    if gy < 0:
        gy += 2 * math.pi
    h = abs(ry - gy)
    if h > math.pi:
        h = 2 * math.pi - h
    return True, length, h


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_map,start", [(31, 128, (0.1, -0.2, 0.7)), (77, 256, (-2.3, 1.9, -1.1)), (5, 192, (3.05, 3.3, 2.9))])
def test_adapter_runs_on_the_gpu_and_matches_the_oracle(fs, oracle, ref_table, scorer, tmp_path, seed, n_map, start):
    exe = build_driver(tmp_path)
    w = fs.synth.make_small_2d(seed, n=n_map, n_cand=80, n_landmarks=900)
    poly32 = tuple(float(np.float32(v)) for v in w.polygon)        # the reference stores the polygon as Point32
    wl = tmp_path / "w.bin"
    with open(wl, "wb") as f:
        ny, nx = w.cells.shape[1:]
        f.write(struct.pack("<iiddd", nx, ny, w.resolution, w.origin[0], w.origin[1]))
        f.write(w.cells.tobytes())
        f.write(struct.pack("<i", w.goals.shape[0]))
        f.write(np.ascontiguousarray(w.goals[:, :2]).tobytes())
        f.write(w.frontier_size.tobytes())
        f.write(w.blacklisted.tobytes())
        f.write(struct.pack("<i", w.landmarks.shape[0]))
        f.write(w.landmarks.tobytes())
        f.write(struct.pack("<3d", *start))
        f.write(struct.pack("<4d", *w.polygon))
    table = tmp_path / "fisher_information_lookup_table.dat"
    scorer.lookup_generate()
    scorer.lookup_save(str(table))
    out = tmp_path / "r.bin"
    # (seed 77: the manager also REGENERATES the table file — a deliberately wrong file is put there first — and loads it back)
    regenerate = seed == 77
    if regenerate:
        scorer.lookup_generate((0.0, 3.0, -2.0, 2.0, -1.0, 1.0))
        scorer.lookup_save(str(table))
        scorer.lookup_generate()
    p = subprocess.run([exe, str(wl), str(table), str(out)] + (["regenerate"] if regenerate else []), capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    if regenerate:
        assert "check ok: generateLookupTable + loadLookupTable through the manager" in p.stdout
        assert open(table, "rb").read() == ref_table.records.tobytes()          # the file gen_fi_lookup would write, byte for byte
    if seed == 31:
        # the same run with the adapter, the doubles and the driver under ASan + UBSan: no report, the same bytes out
        exe_san = build_driver(tmp_path, sanitize=True)
        out_san = tmp_path / "r_san.bin"
        q = subprocess.run([exe_san, str(wl), str(table), str(out_san)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
        assert q.returncode == 0 and "ERROR: AddressSanitizer" not in q.stderr and "runtime error" not in q.stderr, q.stdout[-1500:] + q.stderr[-3000:]
        a_, b_ = np.fromfile(out, dtype=np.float64), np.fromfile(out_san, dtype=np.float64)
        n_ = w.goals.shape[0]
        np.testing.assert_array_equal(a_[:16 * n_], b_[:16 * n_])          # (the cost assignment: bit for bit; the FI columns may differ in last bits from call to call)
    assert "CHECK FAILED" not in p.stdout and "failures: 0" in p.stdout
    for line in ("no boundary polygon yet -> getFrontierCosts returns false", "getFrontierCosts (three-step route)", "getFrontierCosts (fused route)",
                 "duplicate frontier threw", "isPoseSafe before any map_data -> false", "plugin instead of the reference plugin: both IDs, once",
                 "plugin next to the reference plugin: both IDs, once", "tick without latest_robot_pose on the blackboard threw",
                 "tick far from every landmark: FAILURE after the 700 ms back-off"):
        assert "check ok: " + line in p.stdout, line
    raw = np.fromfile(out, dtype=np.float64)
    n = w.goals.shape[0]
    routes = raw[:16 * n].reshape(2, n, 8)
    tail = raw[-4:]
    n_poses = int(tail[2])
    assert tail[3] == 0 and raw.size == 16 * n + 2 * n_poses + 4
    safe = raw[16 * n:16 * n + 2 * n_poses].reshape(n_poses, 2)

    # ---- the expected values: the oracle's arrival information, the reference's Euclidean planner, the oracle's U1
    G = oracle.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = oracle.RayParams(polygon=poly32)
    mx = oracle.max_arrival_information(G, P)
    arr = oracle.arrival_information(G, P, w.goals, w.frontier_size, w.blacklisted, min_gt=mx["min_gt"], faithful=True)
    qz, qw = math.sin(start[2] * 0.5), math.cos(start[2] * 0.5)
    robot_yaw = math.atan2(2.0 * (qw * qz + 0.0 * 0.0), 1.0 - 2.0 * (0.0 * 0.0 + qz * qz))
    dmax = np.finfo(np.float64).max
    ach = arr["achievable"].copy()
    plen = np.zeros(n); phead = np.zeros(n)
    for i in range(n):
        if w.blacklisted[i]:
            plen[i] = dmax
            continue
        a, plen[i], phead[i] = _euclid(start, robot_yaw, w.goals[i], bool(ach[i]))
        ach[i] = a
    rc, u1 = oracle.u1_costs(arr["arrival"].astype(np.float64), ach, plen, phead, mx["max_gt"], blacklisted=w.blacklisted)
    assert rc == 0
    live = w.blacklisted == 0
    assert live.sum() > 10 and (~live).sum() > 0 and 0 < ach[live].sum() < live.sum()
    for r, name in ((0, "three-step"), (1, "fused")):
        got = routes[r]
        np.testing.assert_array_equal(got[live, 0], arr["arrival"][live].astype(np.float64), err_msg=name)
        np.testing.assert_array_equal(got[live, 1], arr["yaw"][live], err_msg=name)
        np.testing.assert_array_equal(got[live, 2], ach[live].astype(np.float64), err_msg=name)
        np.testing.assert_array_equal(got[:, 3], u1["weighted_cost"], err_msg=name)
        np.testing.assert_array_equal(got[live, 4], u1["arrival_utility"][live], err_msg=name)
        np.testing.assert_array_equal(got[live, 5], u1["distance_utility"][live], err_msg=name)
        np.testing.assert_array_equal(got[:, 6], plen, err_msg=name)
        np.testing.assert_array_equal(got[:, 7], got[:, 3], err_msg=name)                 # response.frontier_costs = the frontiers' weighted costs
        # blacklisted (FrontierCostsManager.cpp:77-86): zeros, maximal cost, no utilities
        assert not got[~live, 0].any() and not got[~live, 1].any() and np.all(got[~live, 3] == dmax) and np.all(got[~live, 4] == -1000.0)
    np.testing.assert_array_equal(routes[0], routes[1])                                   # the two routes agree on every column, bit for bit

    # ---- isPoseSafe at the reference's request (14 m, cone off: the adapter's default) against the oracle and the 550 threshold
    poses = oracle.poses_from_yaw(w.goals[:n_poses], routes[0][:n_poses, 1])
    fim = oracle.pose_information(ref_table, w.landmarks, poses, 14.0, 4.0)
    sc = np.maximum(np.abs(fim["info_f64"]), 1e-6)
    assert np.max(np.abs(safe[:, 1] - fim["info_f64"]) / sc) <= 1e-4
    clear = np.abs(fim["info_f64"] - 550.0) > 1e-3 * 550.0                               # (a verdict on the threshold itself may go either way)
    np.testing.assert_array_equal(safe[clear, 0], (fim["info_f64"][clear] > 550.0).astype(np.float64))
    # the BT node's verdict at the TF pose = pose 0's
    if clear[0]:
        assert tail[0] == tail[1] == float(fim["info_f64"][0] > 550.0)
