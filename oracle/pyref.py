"""Second, independent, deliberately slow transcription of the hot path — TEST INFRASTRUCTURE ONLY.

Written straight from the reference sources in pure Python (numpy scalars for float32), sharing no
code with oracle/*.c, so that the C oracle can be cross-checked by something "obviously correct"
(option ii).  Small cases only.  Same reference citations as fso_oracle.h:
  DEP/ = dev_ws/src/DEPRECATED/frontier_exploration/frontier_exploration/
  FIP/ = dev_ws/src/fit-slam2/fisher_information_plugins/
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


class Costmap:
    """nav2_costmap_2d::Costmap2D accessors used by the path, plus a z axis."""

    def __init__(self, cells, origin, resolution):
        c = np.asarray(cells, dtype=np.uint8)
        if c.ndim == 2:
            c = c[None]
        self.cells = c
        self.nz, self.ny, self.nx = c.shape
        self.ox, self.oy, self.oz = (float(v) for v in origin)
        self.res = float(resolution)

    def world_to_map(self, wx, wy, wz):
        if wx < self.ox or wy < self.oy or wz < self.oz:
            return None
        mx = int((wx - self.ox) / self.res)   # truncation of a non-negative double
        my = int((wy - self.oy) / self.res)
        mz = int((wz - self.oz) / self.res)
        if mx < self.nx and my < self.ny and mz < self.nz:
            return mx, my, mz
        return None

    def size_in_meters(self, n):
        return (n - 1 + 0.5) * self.res

    def cost(self, x, y, z):
        return int(self.cells[z, y, x])


def _sign(v):                       # DEP/include/.../Helpers.hpp:113-116
    return 1 if v > 0 else -1


def walk_cells(cm: Costmap, s, w, max_length):
    """getTracedCells + bresenham2D (DEP/src/Helpers.cpp:7-96), as a list of (x,y,z) cells."""
    end = cm.world_to_map(*w)
    start = cm.world_to_map(*s)
    if end is None or start is None:
        return None
    d = [end[i] - start[i] for i in range(3)]
    dist = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) if d[2] != 0 else math.hypot(d[0], d[1])
    a = [abs(v) for v in d]
    scale = 1.0 if dist == 0.0 else min(1.0, max_length / dist)
    if a[0] >= a[1] and a[0] >= a[2]:
        order = (0, 1, 2)
    elif a[1] >= a[2]:
        order = (1, 0, 2)
    else:
        order = (2, 0, 1)
    ia, ib, ic = order
    steps = min(int(scale * a[ia]), a[ia])
    pos = list(start)
    err_b = a[ia] // 2
    err_c = a[ia] // 2
    out = []
    for _ in range(steps):
        out.append(tuple(pos))
        pos[ia] += _sign(d[ia])
        err_b += a[ib]
        if err_b >= a[ia]:
            pos[ib] += _sign(d[ib])
            err_b -= a[ia]
        err_c += a[ic]
        if err_c >= a[ia]:
            pos[ic] += _sign(d[ic])
            err_c -= a[ia]
    out.append(tuple(pos))
    return out


def count_ray(cm: Costmap, cells, obst=(240, 254), trace=(255, 255)):
    """RayTracedCells::operator() over a cell list (DEP/include/.../Helpers.hpp:50-77)."""
    kept = []
    hit = False
    for c in cells:
        if c in kept:
            continue
        cost = cm.cost(*c)
        if trace[0] <= cost <= trace[1] and not hit:
            kept.append(c)
        if obst[0] <= cost <= obst[1]:
            hit = True
    return len(kept), hit


def footprint_in_lethal(cm: Costmap, cx, cy, cz, radius):
    """DEP/src/Helpers.cpp:135-155; off-grid cells are not lethal (documented deviation)."""
    r = int(radius)
    for dx in range(-r, r + 1):
        for dy in range(-r, r + 1):
            if dx * dx + dy * dy <= radius * radius:
                x, y = cx + dx, cy + dy
                if 0 <= x < cm.nx and 0 <= y < cm.ny and cm.cost(x, y, cz) == 254:
                    return True
    return False


def theta_fan(delta_theta, n_rays=0):
    """DEP/src/CostCalculator.cpp:36 — accumulated theta."""
    out = []
    theta = 0.0
    if n_rays > 0:
        for _ in range(n_rays):
            out.append(theta)
            theta += delta_theta
        return out
    while theta <= 2 * math.pi:
        out.append(theta)
        theta += delta_theta
    return out


def arrival_information(cm: Costmap, goal, *, depth=2.0, delta_theta=0.10, fov=1.04, robot_radius=0.60,
                        n_rays=0, elev=(0.0,), polygon=(-1e300, -1e300, 1e300, 1e300), clamp=True,
                        obst=(240, 254), trace=(255, 255), frontier_size=0, min_gt=0.0, achievable=True):
    """setArrivalInformationForFrontier (DEP/src/CostCalculator.cpp:23-121)."""
    sx, sy, sz = goal
    max_length = int(depth / cm.res)
    thetas = theta_fan(delta_theta, n_rays)
    counts = []
    for e in elev:
        ring = []
        dh = depth * math.cos(e)
        dz = depth * math.sin(e)
        for th in thetas:
            wx = sx + (dh * math.cos(th))
            wy = sy + (dh * math.sin(th))
            wz = sz + dz
            if clamp:
                wx = max(polygon[0], max(cm.ox, min(polygon[2], min(cm.ox + cm.size_in_meters(cm.nx), wx))))
                wy = max(polygon[1], max(cm.oy, min(polygon[3], min(cm.oy + cm.size_in_meters(cm.ny), wy))))
                wz = max(cm.oz, min(cm.oz + cm.size_in_meters(cm.nz), wz))
            cells = walk_cells(cm, (sx, sy, sz), (wx, wy, wz), float(max_length))
            if cells is None:
                return dict(arrival=0, argmax=0, yaw=0.0, achievable=achievable, status=1, counts=None)
            ring.append(count_ray(cm, cells, obst, trace)[0])
        counts.append(ring)
    m = cm.world_to_map(sx, sy, sz)
    lethal = footprint_in_lethal(cm, m[0], m[1], m[2], math.ceil(robot_radius / cm.res))
    if lethal and frontier_size < 10.0:
        achievable = False
    k = int(fov / delta_theta)
    n = len(thetas)
    best, best_i = None, 0
    for i in range(n - k + 1):
        s = sum(counts[e][i + j] for j in range(k) for e in range(len(elev)))
        if best is None or s > best:
            best, best_i = s, i
    if best < min_gt:
        achievable = False
    return dict(arrival=best, argmax=best_i, yaw=(best_i * delta_theta) + (fov / 2), achievable=achievable,
                status=0, counts=counts)


# ---------------------------------------------------------------------- Fisher information

def info_point_local(p):
    """computeInformationOfPointLocal(p, I) in float32 (FIP/src/.../FisherInformationHelpers.cpp:71-96,114-123)."""
    p = np.asarray(p, dtype=f32)
    with np.errstate(all="ignore"):
        n = f32(np.sqrt(f32(p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]))
        A = (f32(1) / n) * np.eye(3, dtype=f32) - (f32(1) / (n * n * n)) * np.outer(p, p).astype(f32)
        S = np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]], dtype=f32)
        right = np.concatenate([-np.eye(3, dtype=f32), S], axis=1)
        J = (A @ right).astype(f32)
        F = (J.T @ J).astype(f32)
        return f32(np.trace(F))


def voxel_key(x, y, z):
    """getVoxelCoordinate (FIP/include/.../FisherInfoManager.hpp:108-123); step is (double)0.3f."""
    step = float(f32(0.3))

    def rnd(v):                      # std::round: half away from zero
        return math.floor(abs(v) + 0.5) * (1.0 if v >= 0 else -1.0)

    keys = []
    for v in (x, y, z):
        r = rnd(float(f32(v)) * (1 / step))
        keys.append(f32(r * step))
    return tuple(keys)


def crowding_factor(k):
    """getFactorFromNum (FisherInfoManager.hpp:102-106)."""
    return f32(math.exp(1 - math.pow(k, float(f32(0.8)))))


def generate_table(bounds):
    """generateLookupTable (FIP/src/.../FisherInfoManager.cpp:117-229) -> dict key->value, record count."""
    smax = f32(0.3)
    smin = f32(0.09)
    lo = [f32(np.floor(f32(bounds[i]) * (f32(1) / smax)) * smax) for i in (0, 2, 4)]
    hi = [f32(np.ceil(f32(bounds[i]) * (f32(1) / smax)) * smax) for i in (1, 3, 5)]
    table = {}
    seen = set()
    n_rec = 0
    vmax = f32(-3.4e38)
    inc = smin
    cx = lo[0]
    while cx <= hi[0]:
        if cx > f32(-1.0) + smax:
            inc = smax
        cy = lo[1]
        while cy <= hi[1]:
            cz = lo[2]
            while cz <= hi[2]:
                key = voxel_key(cx, cy, cz)
                kk = tuple(float(v) + 0.0 for v in key)
                if kk not in seen:
                    seen.add(kk)
                    val = info_point_local(key)
                    if not np.isnan(val):
                        vmax = max(vmax, val)
                        table[kk] = val
                        n_rec += 1
                cz = f32(cz + inc)
            cy = f32(cy + inc)
        cx = f32(cx + inc)
    table[(0.0, 0.0, 0.0)] = vmax
    return table, n_rec + 1


def quat_to_rot_f32(q):
    """Eigen::Quaternionf(w,x,y,z).toRotationMatrix() in float32 (FisherInformationHelpers.cpp:16-26)."""
    x, y, z, w = (f32(v) for v in q)
    tx, ty, tz = f32(2) * x, f32(2) * y, f32(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = f32(1)
    return np.array([[one - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, one - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, one - (txx + tyy)]], dtype=f32)


def _fma32(a, b, c):
    """correctly rounded float32 fma: exact rational value, then the nearest float32."""
    from fractions import Fraction
    s = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    best = f32(float(s))
    for cand in (np.nextafter(best, f32(-np.inf)), np.nextafter(best, f32(np.inf))):
        if abs(Fraction(float(cand)) - s) < abs(Fraction(float(best)) - s):
            best = cand
    return f32(best)


def world_to_camera(R, t, w):
    dx, dy, dz = f32(w[0]) - t[0], f32(w[1]) - t[1], f32(w[2]) - t[2]
    return np.array([_fma32(R[0, i], dx, _fma32(R[1, i], dy, R[2, i] * dz)) for i in range(3)], dtype=f32)


def is_visible(p, max_dist=14.0, max_angle=1.0):
    n2 = _fma32(p[0], p[0], _fma32(p[1], p[1], p[2] * p[2]))
    if not (n2 <= f32(max_dist * max_dist)):
        return False
    if max_angle >= math.pi:
        return True
    c = f32(math.cos(max_angle))
    c2 = c * c
    if c >= 0:
        return bool(p[0] >= 0 and p[0] * p[0] >= c2 * n2)
    return bool(p[0] >= 0 or p[0] * p[0] <= c2 * n2)


def pose_information(table, landmarks, pose7, max_dist=14.0, max_angle=1.0):
    """isPoseSafe's loop (FIP/src/.../FisherInfoManager.cpp:83-100) with getInformationFromLookup (:287-324)."""
    t = np.array(pose7[:3], dtype=f32)
    R = quat_to_rot_f32(pose7[3:7])
    counts = {}
    total = f32(0)
    nvis = 0
    for w in landmarks:
        p = world_to_camera(R, t, w)
        if not is_visible(p, max_dist, max_angle):
            continue
        nvis += 1
        key = tuple(float(v) + 0.0 for v in voxel_key(*p))
        if key not in table:
            continue
        counts[key] = counts.get(key, 0) + 1
        info = f32(float(table[key]) * float(crowding_factor(counts[key])))
        total = f32(total + info)
    return dict(info_ref=total, n_visible=nvis, n_voxels=len(counts))


# ---------------------------------------------------------------- key-frame pose information (§8a row a24)
# DEP/include/frontier_exploration/deprecated/util.hpp:49-66,77-119,134-185,616-632,687-759,840-916

def quat_to_yaw(q):
    """quatToEuler(...)[2]: tf2::Matrix3x3(q).getRPY (tf2 restated from upstream; third party)."""
    x, y, z, w = (float(v) for v in q)
    s = 2.0 / (x * x + y * y + z * z + w * w)
    m00 = 1.0 - (y * (y * s) + z * (z * s))
    m10 = x * (y * s) + w * (z * s)
    m20 = x * (z * s) - w * (y * s)
    if abs(m20) >= 1.0:
        return 0.0
    pitch = -math.asin(m20)
    return math.atan2(m10 / math.cos(pitch), m00 / math.cos(pitch))


def frustum_vertices_2d(pose7, depth, hfov):
    yaw = quat_to_yaw(pose7[3:7])
    x, y = float(pose7[0]), float(pose7[1])
    return [(x, y),
            (x + depth * math.cos(yaw - hfov / 2), y + depth * math.sin(yaw - hfov / 2)),
            (x + depth * math.cos(yaw + hfov / 2), y + depth * math.sin(yaw + hfov / 2))]


def point_in_triangle(p, tri):
    v0 = (tri[2][0] - tri[0][0], tri[2][1] - tri[0][1])
    v1 = (tri[1][0] - tri[0][0], tri[1][1] - tri[0][1])
    v2 = (p[0] - tri[0][0], p[1] - tri[0][1])
    d00 = v0[0] * v0[0] + v0[1] * v0[1]
    d01 = v0[0] * v1[0] + v0[1] * v1[1]
    d02 = v0[0] * v2[0] + v0[1] * v2[1]
    d11 = v1[0] * v1[0] + v1[1] * v1[1]
    d12 = v1[0] * v2[0] + v1[1] * v2[1]
    den = d00 * d11 - d01 * d01
    if den == 0.0:
        return False                    # 1/0 = inf -> inf * x is inf/NaN: every comparison chain fails or u + v <= 1 fails
    inv = 1.0 / den
    u = (d11 * d02 - d01 * d12) * inv
    v = (d00 * d12 - d01 * d02) * inv
    return u >= 0.0 and v >= 0.0 and u + v <= 1.0


def frustum_overlap(cur, chk, depth, hfov, err):
    tri = frustum_vertices_2d(cur, depth + err, hfov)
    v = frustum_vertices_2d(chk, depth + err, hfov)
    pts = list(v) + [((v[0][0] + v[1][0]) / 2, (v[0][1] + v[1][1]) / 2),
                     ((v[1][0] + v[2][0]) / 2, (v[1][1] + v[2][1]) / 2),
                     ((v[2][0] + v[0][0]) / 2, (v[2][1] + v[0][1]) / 2)]
    return any(point_in_triangle(p, tri) for p in pts)


def info_point_affine(pose7, p_w, q=0.01):
    """affine computeInformationOfPoint(p_c, p_w, T_w_c_est, Q) (util.hpp:687-759) with numpy float32 matrices."""
    t = np.array(pose7[:3], dtype=f32)
    R = quat_to_rot_f32(pose7[3:7])
    w = np.array(p_w, dtype=f32)
    p = world_to_camera(R, t, w)
    n = f32(np.sqrt(f32(p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]))
    A = (f32(1) / n) * np.eye(3, dtype=f32) - np.outer((f32(1) / (n * n * n)) * p, p).astype(f32)
    S = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=f32)
    right = np.concatenate([np.eye(3, dtype=f32), f32(-1.0) * S], axis=1)
    J = (A @ (R.T @ right)).astype(f32)
    qf = f32(q)
    qinv = (qf * qf) * (f32(1) / (qf * (qf * qf)))
    F = (J.T * qinv) @ J
    return f32(np.trace(F))


def information_for_pose(cm: Costmap, pose7, kf_pose7, kf_points, depth=2.0, hfov=1.089, err=0.5, q=0.01, radius=4.5):
    """computeInformationForPose (util.hpp:840-916); kf_points: one (m_k, 3) array per key-frame."""
    tri = frustum_vertices_2d(pose7, depth, hfov)
    info_map = {}
    total = f32(0)
    n_points = 0
    for kp, pts in zip(kf_pose7, kf_points):
        if radius >= 0 and not math.sqrt((kp[0] - pose7[0]) ** 2 + (kp[1] - pose7[1]) ** 2) <= radius:
            continue
        if not frustum_overlap(pose7, kp, depth, hfov, err):
            continue
        for w in np.asarray(pts, dtype=f32).reshape(-1, 3):
            if not point_in_triangle((float(w[0]), float(w[1])), tri):
                continue
            m = cm.world_to_map(float(w[0]), float(w[1]), cm.oz)
            if m is None:
                continue
            idx = m[1] * cm.nx + m[0]
            if idx not in info_map:
                info_map[idx] = info_point_affine(pose7, w, q)
            total = f32(total + info_map[idx])
            n_points += 1
    return float(total), len(info_map), n_points
