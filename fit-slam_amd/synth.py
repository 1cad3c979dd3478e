"""Deterministic synthetic workloads for the frontier-scoring path.

The reference ships no recorded costmap, landmark cloud or frontier list for this path, so tests,
bench.py and the golden-fixture script all draw their inputs from here.  Everything is a pure
function of (config, seed): numpy PCG64 with seed 0xF1751A00 + config index.

Grid values emulate a thresholded traversability costmap (nav2 cost constants):
255 unknown, 0 free, 254 lethal walls on ~70 % of the room borders, a 1-3 cell inflation band
253 -> 240.  3-D grids extrude the floor plan between a floor slab and a per-block ceiling and add
2 % floating obstacles.  Candidates are free cells 4-adjacent (in the plane) to an unknown cell —
the reference's frontier-cell predicate (DEP/src/FrontierSearch.cpp:218-249) — at cell centres.
Landmarks sit on obstacle cells with +-0.5 cell jitter.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

SEED_BASE = 0xF1751A00

# BASELINE.json:configs.  rays/cand = n_yaw * len(elev).  L = max_camera_depth / resolution.
CONFIGS = {
    # name: (index, N, nz, n_cand, n_landmarks, n_yaw, elev, depth_cells)
    "C1": dict(index=1, n=64, nz=64, n_cand=200, n_landmarks=2_000, n_yaw=32, elev=(0.0,), depth_cells=40),
    "C1_2D": dict(index=1, n=64, nz=1, n_cand=200, n_landmarks=2_000, n_yaw=32, elev=(0.0,), depth_cells=40),
    "C2": dict(index=2, n=256, nz=256, n_cand=5_000, n_landmarks=50_000, n_yaw=32,
               elev=(-0.30, -0.10, 0.10, 0.30), depth_cells=40),
    "C3": dict(index=3, n=512, nz=512, n_cand=20_000, n_landmarks=100_000, n_yaw=64,
               elev=(-0.30, -0.10, 0.10, 0.30), depth_cells=40),
    "C4": dict(index=4, n=512, nz=512, n_cand=160_000, n_landmarks=100_000, n_yaw=64,
               elev=(-0.30, -0.10, 0.10, 0.30), depth_cells=40),
    # configs[4]: 1024^3 map (sparse brick list on the wire, dense 1 GiB in HBM), 50 k candidates, 500 k landmarks
    "C5": dict(index=5, n=1024, nz=1024, n_cand=50_000, n_landmarks=500_000, n_yaw=64,
               elev=(-0.30, -0.10, 0.10, 0.30), depth_cells=40),
    # the reference's own operating point (DEP/src/CostCalculator.cpp:36, DEP/params/exploration.yaml:8-10): a 2-D costmap,
    # delta_theta 0.10 accumulated while theta <= 2 pi (n_yaw = 0 selects that loop: 63 rays), one ring, L = 40
    "REF2D": dict(index=6, n=512, nz=1, n_cand=20_000, n_landmarks=100_000, n_yaw=0, elev=(0.0,), depth_cells=40,
                  reach=6, jitter=True),
}

RESOLUTION = 0.05


@dataclass
class Workload:
    name: str
    cells: np.ndarray              # uint8 [nz][ny][nx]
    origin: tuple                  # (ox, oy, oz)
    resolution: float
    goals: np.ndarray              # float64 [n][3] world, cell centres
    frontier_size: np.ndarray      # int32 [n]
    blacklisted: np.ndarray        # uint8 [n]
    landmarks: np.ndarray          # float32 [m][3] world
    n_yaw: int
    elev: tuple
    max_camera_depth: float
    delta_theta: float
    camera_fov: float = 1.04
    robot_radius: float = 0.60
    polygon: tuple = field(default=None)

    @property
    def rays_per_candidate(self) -> int:
        return self.n_yaw * len(self.elev)


def _floor_plan(rng: np.random.Generator, n: int) -> np.ndarray:
    """2-D uint8 plan [n][n]."""
    plan = np.full((n, n), 255, dtype=np.uint8)
    lo, hi = max(3, n // 16), max(6, n // 4)
    target = 0.50
    guard = 0
    while (plan == 0).mean() < target and guard < 10_000:
        guard += 1
        w, h = rng.integers(lo, hi + 1, size=2)
        if rng.random() < 0.3:               # corridor
            if rng.random() < 0.5:
                h = max(2, n // 48)
            else:
                w = max(2, n // 48)
        x0 = int(rng.integers(1, max(2, n - w - 1)))
        y0 = int(rng.integers(1, max(2, n - h - 1)))
        plan[y0:y0 + h, x0:x0 + w] = 0
    free = plan == 0
    unk = plan == 255
    # free cells with an unknown 4-neighbour = room borders
    nb_unknown = np.zeros_like(free)
    nb_unknown[1:, :] |= unk[:-1, :]
    nb_unknown[:-1, :] |= unk[1:, :]
    nb_unknown[:, 1:] |= unk[:, :-1]
    nb_unknown[:, :-1] |= unk[:, 1:]
    border = free & nb_unknown
    wall = border & (rng.random((n, n)) < 0.7)
    plan[wall] = 254
    # inflation band: chebyshev distance 1..3 from a wall, free cells only, 253 -> 240
    dist = np.full((n, n), 99, dtype=np.int32)
    cur = wall.copy()
    for d in (1, 2, 3):
        grown = cur.copy()
        grown[1:, :] |= cur[:-1, :]
        grown[:-1, :] |= cur[1:, :]
        grown[:, 1:] |= cur[:, :-1]
        grown[:, :-1] |= cur[:, 1:]
        newly = grown & ~cur
        dist[newly & (dist == 99)] = d
        cur = grown
    band_val = {1: 253, 2: 246, 3: 240}
    for d, v in band_val.items():
        # keep some of the inflation sparse so that frontier cells survive next to walls
        sel = (dist == d) & (plan == 0) & (rng.random((n, n)) < 0.5)
        plan[sel] = v
    return plan


def make_grid(rng: np.random.Generator, n: int, nz: int) -> np.ndarray:
    plan = _floor_plan(rng, n)
    if nz == 1:
        return plan[None].copy()
    cells = np.full((nz, n, n), 255, dtype=np.uint8)
    blk = max(1, n // 16)
    nb = (n + blk - 1) // blk
    ceil_blk = rng.integers(int(0.45 * nz), int(0.9 * nz) + 1, size=(nb, nb))
    ceil = np.kron(ceil_blk, np.ones((blk, blk), dtype=np.int64))[:n, :n]
    floor = max(1, nz // 10)
    z = np.arange(nz)[:, None, None]
    inside = (z >= floor) & (z < ceil[None])
    np.copyto(cells, np.broadcast_to(plan[None], cells.shape), where=inside)
    cells[:floor][np.broadcast_to(plan[None] != 255, (floor, n, n))] = 254   # floor slab under mapped columns
    # 2 % floating obstacles among free cells
    n_float = int(0.02 * n * n * nz * 0.3)
    zi = rng.integers(floor, nz, size=n_float)
    yi = rng.integers(0, n, size=n_float)
    xi = rng.integers(0, n, size=n_float)
    sel = cells[zi, yi, xi] == 0
    cells[zi[sel], yi[sel], xi[sel]] = 254
    return cells


def _frontier_cells(rng: np.random.Generator, cells: np.ndarray, n_cand: int, reach: int = 1) -> np.ndarray:
    """Free cells with an unknown cell within `reach` 4-neighbour steps in the plane (reach 1 = the frontier-cell
    predicate itself; larger values add the cells around a frontier where cluster goal points fall)."""
    nz, ny, nx = cells.shape
    layers = np.arange(nz) if nz <= 64 else np.sort(rng.choice(nz, size=64, replace=False))
    found = []
    for z in layers:
        s = cells[z]
        free = s == 0
        if not free.any():
            continue
        unk = s == 255
        for _ in range(reach):
            nb = np.zeros_like(free)
            nb[1:, :] |= unk[:-1, :]
            nb[:-1, :] |= unk[1:, :]
            nb[:, 1:] |= unk[:, :-1]
            nb[:, :-1] |= unk[:, 1:]
            unk = unk | nb
        yy, xx = np.nonzero(free & nb)
        if yy.size:
            found.append(np.stack([xx, yy, np.full_like(xx, z)], axis=1))
    allc = np.concatenate(found, axis=0) if found else np.zeros((0, 3), dtype=np.int64)
    if allc.shape[0] == 0:
        raise RuntimeError("synthetic grid has no frontier cells")
    idx = rng.choice(allc.shape[0], size=n_cand, replace=allc.shape[0] < n_cand)
    return allc[idx]


def _landmarks(rng: np.random.Generator, cells: np.ndarray, origin, res, m: int) -> np.ndarray:
    nz, ny, nx = cells.shape
    out = np.zeros((0, 3), dtype=np.int64)
    guard = 0
    while out.shape[0] < m and guard < 200:
        guard += 1
        k = max(4 * m, 1 << 16)
        zi = rng.integers(0, nz, size=k)
        yi = rng.integers(0, ny, size=k)
        xi = rng.integers(0, nx, size=k)
        v = cells[zi, yi, xi]
        sel = (v >= 240) & (v <= 254)
        out = np.concatenate([out, np.stack([xi[sel], yi[sel], zi[sel]], axis=1)], axis=0)
    out = out[:m]
    jitter = rng.uniform(-0.5, 0.5, size=(out.shape[0], 3))
    w = (out + 0.5 + jitter) * res + np.asarray(origin)[None]
    return w.astype(np.float32)


def make_workload(name: str, *, n_cand: int | None = None, n_landmarks: int | None = None,
                  seed: int | None = None) -> Workload:
    cfg = CONFIGS[name]
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + cfg["index"] if seed is None else seed))
    n, nz = cfg["n"], cfg["nz"]
    res = RESOLUTION
    origin = (-n * res / 2, -n * res / 2, (-nz * res / 2) if nz > 1 else 0.0)
    cells = make_grid(rng, n, nz)
    nc = cfg["n_cand"] if n_cand is None else n_cand
    m = cfg["n_landmarks"] if n_landmarks is None else n_landmarks
    fc = _frontier_cells(rng, cells, nc, reach=cfg.get("reach", 1))
    goals = (fc + 0.5) * res + np.asarray(origin)[None]
    if nz == 1:
        goals[:, 2] = origin[2]
    if cfg.get("jitter"):
        goals[:, :2] += rng.uniform(-0.5, 0.5, size=(nc, 2)) * res      # goal points are not cell centres in the reference
    fsize = rng.integers(1, 31, size=nc).astype(np.int32)
    black = (rng.random(nc) < 0.01).astype(np.uint8)
    lm = _landmarks(rng, cells, origin, res, m)
    if nz == 1:
        lm[:, 2] = rng.uniform(0.0, 2.5, size=lm.shape[0]).astype(np.float32)      # heights U(0, 2.5 m)
    n_yaw = cfg["n_yaw"]
    return Workload(name=name, cells=cells, origin=origin, resolution=res,
                    goals=np.ascontiguousarray(goals, dtype=np.float64), frontier_size=fsize, blacklisted=black,
                    landmarks=lm, n_yaw=n_yaw, elev=tuple(cfg["elev"]),
                    max_camera_depth=cfg["depth_cells"] * res, delta_theta=(2 * np.pi / n_yaw) if n_yaw else 0.10,
                    polygon=(origin[0], origin[1], origin[0] + n * res, origin[1] + n * res))


def poses_from_yaw(goal_xyz, yaw) -> np.ndarray:
    """pose7 rows (x, y, z, qx, qy, qz, qw) of fs_score_fim for goal points and yaw angles: the quaternion
    nav2_util::geometry_utils::orientationAroundZAxis builds (tf2 setRPY(0, 0, yaw): (0, 0, sin(yaw / 2), cos(yaw / 2)); call
    sites DEP/include/.../util/GeometryUtils.hpp:112-124, DEP/src/Frontier.cpp:54) — libm's sin / cos in double, one by one."""
    import math
    goal = np.asarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((goal.shape[0], 7), dtype=np.float64)
    out[:, :3] = goal
    for i, y in enumerate(np.asarray(yaw, dtype=np.float64).reshape(-1)):
        h = float(y) * 0.5
        out[i, 5], out[i, 6] = math.sin(h), math.cos(h)
    return out


def candidate_block(w: Workload, name: str, block: int, n_cand: int | None = None):
    """Block `block` of a longer candidate list over the SAME map and cloud (weak scaling: rank r scores block r, and needs
    nothing of the other ranks' blocks).  Block 0 is the workload's own list; the others are drawn from the same frontier cells
    with a generator of their own, seeded by (config, block).  Returns (goals, frontier_size, blacklisted)."""
    cfg = CONFIGS[name]
    nc = cfg["n_cand"] if n_cand is None else n_cand
    if block == 0 and nc == w.goals.shape[0]:
        return w.goals, w.frontier_size, w.blacklisted
    rng = np.random.Generator(np.random.PCG64([SEED_BASE + cfg["index"], 0xB10C, int(block)]))
    fc = _frontier_cells(rng, w.cells, nc, reach=cfg.get("reach", 1))
    goals = (fc + 0.5) * w.resolution + np.asarray(w.origin)[None]
    if cfg["nz"] == 1:
        goals[:, 2] = w.origin[2]
    if cfg.get("jitter"):
        goals[:, :2] += rng.uniform(-0.5, 0.5, size=(nc, 2)) * w.resolution
    fsize = rng.integers(1, 31, size=nc).astype(np.int32)
    black = (rng.random(nc) < 0.01).astype(np.uint8)
    return np.ascontiguousarray(goals, dtype=np.float64), fsize, black


def dense_to_bricks(cells: np.ndarray, default_value: int = 255):
    """Sparse wire format of fs_upload_grid_bricks (BASELINE.json configs[4]): the 8x8x8 bricks that hold anything but
    `default_value`.  Returns (brick_xyz int32 [n][3] in brick units, brick_cells uint8 [n][512], index (z*8 + y)*8 + x)."""
    nz, ny, nx = cells.shape
    assert nz % 8 == 0 and ny % 8 == 0 and nx % 8 == 0
    c = cells.reshape(nz // 8, 8, ny // 8, 8, nx // 8, 8).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 512)
    idx = np.nonzero((c != default_value).any(axis=1))[0]
    bz, by, bx = np.unravel_index(idx, (nz // 8, ny // 8, nx // 8))
    return np.stack([bx, by, bz], axis=1).astype(np.int32), np.ascontiguousarray(c[idx])


def make_small_2d(seed: int, n: int = 96, n_cand: int = 64, n_landmarks: int = 600) -> Workload:
    """Reference-default fan (63 rays, L = 40) on a small 2-D costmap — the bit-exact parity slice."""
    rng = np.random.Generator(np.random.PCG64(seed))
    res = RESOLUTION
    origin = (-n * res / 2, -n * res / 2, 0.0)
    cells = make_grid(rng, n, 1)
    fc = _frontier_cells(rng, cells, n_cand)
    goals = (fc + 0.5) * res + np.asarray(origin)[None]
    goals[:, 2] = 0.0
    # a few goals off cell centres and near the border, to exercise truncation and clamping
    goals[: n_cand // 4, :2] += rng.uniform(-0.5, 0.5, size=(n_cand // 4, 2)) * res
    lm = _landmarks(rng, cells, origin, res, n_landmarks)
    lm[:, 2] = rng.uniform(0.0, 2.5, size=lm.shape[0]).astype(np.float32)
    return Workload(name=f"small2d_{seed}", cells=cells, origin=origin, resolution=res,
                    goals=np.ascontiguousarray(goals, dtype=np.float64),
                    frontier_size=rng.integers(1, 31, size=n_cand).astype(np.int32),
                    blacklisted=(rng.random(n_cand) < 0.05).astype(np.uint8),
                    landmarks=lm, n_yaw=0, elev=(0.0,), max_camera_depth=2.0, delta_theta=0.10,
                    polygon=(origin[0] + 0.2, origin[1] + 0.2, origin[0] + n * res - 0.2, origin[1] + n * res - 0.2))


def make_keyframes(w: Workload, n_kf: int, seed: int, points_per_kf: int = 300, reach: float = 3.0):
    """slam_msgs MapData stand-in for computeInformationForPose: key-frame poses (yaw-only, at candidate goals) and, per
    key-frame, the landmarks within `reach` metres in the plane (shared map points appear in several key-frames).
    Returns (kf_pose7 [n_kf][7], kf_offsets [n_kf + 1], points_xyz [total][3])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    idx = rng.choice(w.goals.shape[0], size=n_kf, replace=w.goals.shape[0] < n_kf)
    yaw = rng.uniform(-np.pi, np.pi, size=n_kf)
    pose = np.zeros((n_kf, 7), dtype=np.float64)
    pose[:, :2] = w.goals[idx, :2]
    pose[:, 5] = np.sin(yaw / 2)
    pose[:, 6] = np.cos(yaw / 2)
    offsets = [0]
    chunks = []
    lm = w.landmarks
    for k in range(n_kf):
        d2 = (lm[:, 0] - pose[k, 0]) ** 2 + (lm[:, 1] - pose[k, 1]) ** 2
        near = np.nonzero(d2 <= reach * reach)[0]
        if near.size > points_per_kf:
            near = np.sort(rng.choice(near, size=points_per_kf, replace=False))
        chunks.append(lm[near])
        offsets.append(offsets[-1] + near.size)
    pts = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 3), dtype=np.float32)
    return pose, np.asarray(offsets, dtype=np.int32), np.ascontiguousarray(pts, dtype=np.float32)
