"""ctypes binding of include/fitslam_frontier.h (libfitslam_frontier.so).

This is the thin Python host layer over the C ABI: it owns no algorithm.  There is no CPU
fallback — if the HIP library cannot be built/loaded or no gfx950 device is present, creating a
`FrontierScorer` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

FS_OK, FS_E_INVALID, FS_E_NO_DEVICE, FS_E_HIP, FS_E_STATE, FS_E_IO, FS_E_RANGE = 0, -1, -2, -3, -4, -5, -6
STATUS_OK, STATUS_OFF_MAP, STATUS_BLACKLISTED = 0, 1, 2
FS_MAX_ELEV = 16

# every symbol include/fitslam_frontier.h declares
EXPORTED_SYMBOLS = [
    "fs_abi_version", "fs_ctx_create", "fs_ctx_destroy", "fs_last_error", "fs_synchronize",
    "fs_enable_kernel_timing", "fs_kernel_time", "fs_set_option", "fs_get_counter",
    "fs_set_ray_params", "fs_ray_fan_shape", "fs_upload_grid", "fs_upload_grid_bricks", "fs_update_grid_region", "fs_frontier_cells", "fs_frontier_clusters", "fs_max_arrival", "fs_set_arrival_limits",
    "fs_score_arrival", "fs_trace_segments",
    "fs_upload_landmarks", "fs_lookup_generate", "fs_lookup_load", "fs_lookup_save", "fs_lookup_set_records",
    "fs_lookup_num_records", "fs_lookup_get_records", "fs_lookup_query", "fs_set_fim_params", "fs_score_fim", "fs_information_frontier_pair",
    "fs_upload_keyframes", "fs_information_for_pose",
    "fs_score_candidates", "fs_score_candidates_dev", "fs_rank_candidates", "fs_rank_candidates_dev", "fs_get_frontier_costs", "fs_selftest_fp64",
    "fs_multi_create", "fs_multi_destroy", "fs_multi_num_devices", "fs_multi_ctx", "fs_multi_last_error", "fs_multi_shard_bounds",
    "fs_multi_set_option", "fs_multi_set_ray_params", "fs_multi_upload_grid", "fs_multi_update_grid_region", "fs_multi_upload_landmarks", "fs_multi_lookup_generate",
    "fs_multi_lookup_load", "fs_multi_set_fim_params", "fs_multi_max_arrival", "fs_multi_score_arrival", "fs_multi_score_candidates",
    "fs_multi_score_fim", "fs_multi_get_frontier_costs", "fs_multi_gather_mode",
]

RECORD_DTYPE = np.dtype([("arrival", "<i4"), ("argmax", "<i4"), ("yaw", "<f4"), ("info_ref", "<f4"),
                         ("trace", "<f4"), ("logdet", "<f4"), ("n_visible", "<i4"), ("flags", "<u4")])
assert RECORD_DTYPE.itemsize == 32


class KeyframeParamsC(C.Structure):
    _fields_ = [("max_depth", C.c_double), ("hfov", C.c_double), ("max_depth_error", C.c_double),
                ("q_diag", C.c_float), ("radius", C.c_double)]


class FrontierClusterC(C.Structure):
    _fields_ = [("label", C.c_int32), ("size", C.c_int32), ("centroid_x", C.c_double), ("centroid_y", C.c_double),
                ("min_x", C.c_int32), ("min_y", C.c_int32), ("max_x", C.c_int32), ("max_y", C.c_int32)]


CLUSTER_DTYPE = np.dtype([("label", "<i4"), ("size", "<i4"), ("centroid_x", "<f8"), ("centroid_y", "<f8"),
                          ("min_x", "<i4"), ("min_y", "<i4"), ("max_x", "<i4"), ("max_y", "<i4")])
assert CLUSTER_DTYPE.itemsize == C.sizeof(FrontierClusterC) == 40


class RayParamsC(C.Structure):
    _fields_ = [("max_camera_depth", C.c_double), ("delta_theta", C.c_double), ("camera_fov", C.c_double),
                ("robot_radius", C.c_double), ("n_rays", C.c_int32), ("n_elev", C.c_int32),
                ("elev", C.c_double * FS_MAX_ELEV),
                ("obst_min", C.c_int32), ("obst_max", C.c_int32), ("trace_min", C.c_int32), ("trace_max", C.c_int32),
                ("factor_max", C.c_double), ("factor_min", C.c_double), ("polygon", C.c_double * 4)]


class FimParamsC(C.Structure):
    _fields_ = [("max_dist", C.c_double), ("max_angle", C.c_double)]


class FsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fitslam_frontier error {code}: {msg}")
        self.code = code


_lib = None


def load_library(build: bool = True):
    """Load (building first if needed) the HIP library.  Raises if it is missing: no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if build and _build.needs_build():
        path = _build.build()
    if not os.path.exists(path):
        raise FsError(FS_E_NO_DEVICE, f"{path} is missing and could not be built; there is no CPU fallback")
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.fs_abi_version.restype = C.c_int
    L.fs_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.fs_ctx_destroy.argtypes = [vp]
    L.fs_ctx_destroy.restype = None
    L.fs_last_error.argtypes = [vp]
    L.fs_last_error.restype = C.c_char_p
    L.fs_synchronize.argtypes = [vp]
    L.fs_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.fs_kernel_time.argtypes = [vp, C.c_int, C.POINTER(dbl), C.POINTER(i64)]
    L.fs_set_option.argtypes = [vp, C.c_char_p, dbl]
    L.fs_get_counter.argtypes = [vp, C.c_int, C.POINTER(i64), C.c_int]
    L.fs_set_ray_params.argtypes = [vp, C.POINTER(RayParamsC)]
    L.fs_ray_fan_shape.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.fs_upload_grid.argtypes = [vp, vp, i32, i32, i32, C.POINTER(dbl * 3), dbl]
    L.fs_upload_grid_bricks.argtypes = [vp, i32, i32, i32, C.POINTER(dbl * 3), dbl, C.c_uint8, i64, vp, vp]
    L.fs_update_grid_region.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i64, i64]
    L.fs_multi_update_grid_region.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i64, i64]
    L.fs_frontier_cells.argtypes = [vp, i32, vp, C.POINTER(i64)]
    L.fs_frontier_clusters.argtypes = [vp, C.POINTER(dbl * 2), i32, dbl, i32, vp, i32, vp, C.POINTER(i32), C.POINTER(i64)]
    L.fs_max_arrival.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]
    L.fs_set_arrival_limits.argtypes = [vp, dbl, dbl]
    L.fs_score_arrival.argtypes = [vp, i32] + [vp] * 10
    L.fs_trace_segments.argtypes = [vp, i32, vp, vp, dbl, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.fs_upload_landmarks.argtypes = [vp, vp, i32]
    L.fs_lookup_generate.argtypes = [vp, vp]
    L.fs_lookup_load.argtypes = [vp, C.c_char_p]
    L.fs_lookup_save.argtypes = [vp, C.c_char_p]
    L.fs_lookup_set_records.argtypes = [vp, vp, i64]
    L.fs_lookup_num_records.argtypes = [vp, C.POINTER(i64)]
    L.fs_lookup_get_records.argtypes = [vp, vp]
    L.fs_lookup_query.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.fs_set_fim_params.argtypes = [vp, C.POINTER(FimParamsC)]
    L.fs_score_fim.argtypes = [vp, i32] + [vp] * 7
    L.fs_information_frontier_pair.argtypes = [vp, i32, vp, vp, vp]
    L.fs_upload_keyframes.argtypes = [vp, i32, vp, vp, vp]
    L.fs_information_for_pose.argtypes = [vp, i32, vp, C.POINTER(KeyframeParamsC), vp, vp, vp]
    L.fs_score_candidates.argtypes = [vp, i32] + [vp] * 5
    L.fs_score_candidates_dev.argtypes = [vp, i32] + [vp] * 5
    L.fs_rank_candidates.argtypes = [vp, i32, vp, vp, vp, vp, dbl, dbl, dbl, dbl, vp, vp, vp, vp]
    L.fs_rank_candidates_dev.argtypes = [vp, i32, vp, vp, vp, vp, dbl, dbl, dbl, dbl, vp, vp, vp, vp, vp]
    L.fs_get_frontier_costs.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, dbl, dbl, dbl, dbl, C.c_int, vp, vp, vp, vp, vp]
    L.fs_selftest_fp64.argtypes = [vp, i32, C.POINTER(i64)]
    L.fs_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.fs_multi_destroy.argtypes = [vp]
    L.fs_multi_destroy.restype = None
    L.fs_multi_num_devices.argtypes = [vp]
    L.fs_multi_ctx.argtypes = [vp, C.c_int]
    L.fs_multi_ctx.restype = vp
    L.fs_multi_last_error.argtypes = [vp]
    L.fs_multi_last_error.restype = C.c_char_p
    L.fs_multi_shard_bounds.argtypes = [i32, C.c_int, C.c_int, C.POINTER(i32), C.POINTER(i32)]
    L.fs_multi_set_option.argtypes = [vp, C.c_char_p, dbl]
    L.fs_multi_set_ray_params.argtypes = [vp, C.POINTER(RayParamsC)]
    L.fs_multi_upload_grid.argtypes = [vp, vp, i32, i32, i32, C.POINTER(dbl * 3), dbl]
    L.fs_multi_upload_landmarks.argtypes = [vp, vp, i32]
    L.fs_multi_lookup_generate.argtypes = [vp, vp]
    L.fs_multi_lookup_load.argtypes = [vp, C.c_char_p]
    L.fs_multi_set_fim_params.argtypes = [vp, C.POINTER(FimParamsC)]
    L.fs_multi_max_arrival.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]
    L.fs_multi_score_arrival.argtypes = [vp, i32] + [vp] * 10
    L.fs_multi_score_candidates.argtypes = [vp, i32] + [vp] * 5
    L.fs_multi_score_fim.argtypes = [vp, i32] + [vp] * 7
    L.fs_multi_get_frontier_costs.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, dbl, dbl, dbl, dbl, C.c_int, vp, vp, vp, vp, vp]
    L.fs_multi_gather_mode.argtypes = [vp]
    for name in EXPORTED_SYMBOLS:
        f = getattr(L, name)
        if name not in ("fs_ctx_destroy", "fs_last_error", "fs_multi_destroy", "fs_multi_last_error", "fs_multi_ctx"):
            f.restype = C.c_int
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _window_args(window, view):
    """(pointer, sx, sy, sz, row_stride, slice_stride, keep-alive) of a window given as its own array [sz][sy][sx] (or [sy][sx]),
    or — view=True — as a numpy VIEW into the caller's whole map: the view's strides are passed on and nothing is packed"""
    w = np.asarray(window, dtype=np.uint8)
    if w.ndim == 2:
        w = w[None]
    if not view or w.size == 0:
        w = np.ascontiguousarray(w)
        sz, sy, sx = w.shape
        return _p(w), sx, sy, sz, 0, 0, w
    if w.strides[2] != 1:
        raise ValueError("a window view must be contiguous along x")
    sz, sy, sx = w.shape
    return C.c_void_p(w.ctypes.data), sx, sy, sz, (w.strides[1] if sy > 1 else 0), (w.strides[0] if sz > 1 else 0), w


class FrontierScorer:
    """One scoring context = one GPU + one HIP stream.  Mirrors the call order of the reference:
    construct (parameters) -> grid snapshot -> max-arrival calibration -> score."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.fs_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != FS_OK:
            raise FsError(rc, "fs_ctx_create failed: no gfx950 device / HIP runtime (no CPU fallback exists)")
        self._h = h
        self._elev = (0.0,)
        self.n_yaw = self.n_elev = self.window = 0

    # -- plumbing
    def _check(self, rc):
        if rc != FS_OK:
            raise FsError(rc, (self._L.fs_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            if os.environ.get("FS_BOUNDS"):
                # library built with FS_BOUNDS=1 (range-checked global accesses): counters 29 (ray walks) and 30 (FIM
                # kernels) hold the code of a violated check, 0 if none
                ray, fim = self.get_counter(29), self.get_counter(30)
                if ray or fim:
                    raise FsError(f"FS_BOUNDS: range check violated (ray walk code {ray}, FIM kernel code {fim})")
            self._L.fs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._L.fs_synchronize(self._h))

    def enable_kernel_timing(self, on=True):
        self._check(self._L.fs_enable_kernel_timing(self._h, 1 if on else 0))

    def kernel_time(self, kind: int):
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._L.fs_kernel_time(self._h, kind, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def set_option(self, key: str, value: float):
        self._check(self._L.fs_set_option(self._h, key.encode(), float(value)))

    def get_counter(self, which: int, reset: bool = False) -> int:
        v = C.c_int64()
        self._check(self._L.fs_get_counter(self._h, int(which), C.byref(v), 1 if reset else 0))
        return v.value

    # -- arrival information
    def set_ray_params(self, max_camera_depth=2.0, delta_theta=0.10, camera_fov=1.04, robot_radius=0.60,
                       n_rays=0, elev=(0.0,), obst=(240, 254), trace=(255, 255), factor_max=1.2,
                       factor_min=0.70, polygon=(-1e300, -1e300, 1e300, 1e300)):
        p = _ray_params_c(max_camera_depth, delta_theta, camera_fov, robot_radius, n_rays, elev, obst, trace, factor_max, factor_min, polygon)
        self._check(self._L.fs_set_ray_params(self._h, C.byref(p)))
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.fs_ray_fan_shape(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.n_yaw, self.n_elev, self.window = a.value, b.value, c.value

    def upload_grid(self, cells: np.ndarray, origin, resolution: float):
        c = np.ascontiguousarray(cells, dtype=np.uint8)
        if c.ndim == 2:
            c = c[None]
        nz, ny, nx = c.shape
        o = (C.c_double * 3)(*[float(v) for v in origin])
        self._check(self._L.fs_upload_grid(self._h, _p(c), nx, ny, nz, C.byref(o), float(resolution)))

    def update_grid_region(self, x0, y0, z0, window, view=False):
        """fs_update_grid_region: `window` [sz][sy][sx] (or [sy][sx]) replaces the cells from (x0, y0, z0) on; view=True passes a
        numpy view into the caller's whole map with its strides (nothing packed on the Python side)"""
        ptr, sx, sy, sz, rs, ss, keep = _window_args(window, view)
        self._check(self._L.fs_update_grid_region(self._h, int(x0), int(y0), int(z0), sx, sy, sz, ptr, rs, ss))

    def upload_grid_bricks(self, shape_zyx, origin, resolution, brick_xyz, brick_cells, default_value=255):
        nz, ny, nx = shape_zyx
        xyz = np.ascontiguousarray(brick_xyz, dtype=np.int32).reshape(-1, 3)
        cells = np.ascontiguousarray(brick_cells, dtype=np.uint8).reshape(-1, 512)
        o = (C.c_double * 3)(*[float(v) for v in origin])
        self._check(self._L.fs_upload_grid_bricks(self._h, nx, ny, nz, C.byref(o), float(resolution), int(default_value),
                                                  xyz.shape[0], _p(xyz), _p(cells)))

    def frontier_cells(self, shape_zyx, lethal_threshold=160, want_mask=True):
        mask = np.zeros(shape_zyx, dtype=np.uint8) if want_mask else None
        n = C.c_int64()
        self._check(self._L.fs_frontier_cells(self._h, int(lethal_threshold), _p(mask), C.byref(n)))
        return mask, n.value

    def frontier_clusters(self, shape_yx, robot_xy, lethal_threshold=160, max_frontier_distance=50.0, max_frontier_cluster_size=20,
                          max_clusters=65536, want_labels=True):
        """FrontierSearch::searchFrom as clusters: returns (labels [ny][nx] or None, clusters (CLUSTER_DTYPE, ascending label),
        n_clusters, n_cells)."""
        ny, nx = shape_yx[-2], shape_yx[-1]
        labels = np.zeros((ny, nx), dtype=np.int32) if want_labels else None
        cl = np.zeros(max_clusters, dtype=CLUSTER_DTYPE)
        n, cells = C.c_int32(), C.c_int64()
        xy = (C.c_double * 2)(float(robot_xy[0]), float(robot_xy[1]))
        self._check(self._L.fs_frontier_clusters(self._h, C.byref(xy), int(lethal_threshold), float(max_frontier_distance),
                                                 int(max_frontier_cluster_size), _p(labels), int(max_clusters), _p(cl),
                                                 C.byref(n), C.byref(cells)))
        return labels, cl[:min(n.value, max_clusters)].copy(), n.value, cells.value

    def max_arrival(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self._L.fs_max_arrival(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(max_value=a.value, max_gt=b.value, min_gt=c.value)

    def set_arrival_limits(self, max_gt, min_gt):
        self._check(self._L.fs_set_arrival_limits(self._h, float(max_gt), float(min_gt)))

    def score_arrival(self, goal_xyz, frontier_size=None, blacklisted=None, achievable_in=None, want_ray_counts=True):
        goal = np.ascontiguousarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
        n = goal.shape[0]
        fs = None if frontier_size is None else np.ascontiguousarray(frontier_size, dtype=np.int32)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        ai = None if achievable_in is None else np.ascontiguousarray(achievable_in, dtype=np.uint8)
        rc_arr = np.zeros((n, self.n_elev, self.n_yaw), dtype=np.int32) if want_ray_counts else None
        arrival = np.zeros(n, dtype=np.int32); argmax = np.zeros(n, dtype=np.int32)
        yaw = np.zeros(n, dtype=np.float64); ach = np.zeros(n, dtype=np.uint8); status = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_score_arrival(self._h, n, _p(goal), _p(fs), _p(bl), _p(ai), _p(rc_arr),
                                             _p(arrival), _p(argmax), _p(yaw), _p(ach), _p(status)))
        return dict(ray_counts=rc_arr, arrival=arrival, argmax=argmax, yaw=yaw, achievable=ach, status=status)

    def trace_segments(self, start_xyz, end_xyz, max_length_cells, obst=(253, 254), trace=(0, 255)):
        a = np.ascontiguousarray(start_xyz, dtype=np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(end_xyz, dtype=np.float64).reshape(-1, 3)
        n = a.shape[0]
        ok = np.zeros(n, np.uint8); hit = np.zeros(n, np.uint8)
        traced = np.zeros(n, np.int32); unknown = np.zeros(n, np.int32); allc = np.zeros(n, np.int32)
        self._check(self._L.fs_trace_segments(self._h, n, _p(a), _p(b), float(max_length_cells), int(obst[0]), int(obst[1]),
                                              int(trace[0]), int(trace[1]), _p(ok), _p(traced), _p(hit), _p(unknown), _p(allc)))
        return dict(ok=ok, traced=traced, hit=hit, unknown=unknown, all=allc)

    # -- Fisher information
    def upload_landmarks(self, xyz):
        lm = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._check(self._L.fs_upload_landmarks(self._h, _p(lm), lm.shape[0]))

    def lookup_generate(self, bounds=None):
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.float32)
        self._check(self._L.fs_lookup_generate(self._h, _p(b)))

    def lookup_load(self, path: str):
        self._check(self._L.fs_lookup_load(self._h, path.encode()))

    def lookup_save(self, path: str):
        self._check(self._L.fs_lookup_save(self._h, path.encode()))

    def lookup_set_records(self, records):
        r = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 4)
        self._check(self._L.fs_lookup_set_records(self._h, _p(r), r.shape[0]))

    def lookup_records(self) -> np.ndarray:
        n = C.c_int64()
        self._check(self._L.fs_lookup_num_records(self._h, C.byref(n)))
        out = np.zeros((n.value, 4), dtype=np.float32)
        self._check(self._L.fs_lookup_get_records(self._h, _p(out)))
        return out

    def lookup_query(self, p) -> float:
        a = np.ascontiguousarray(p, dtype=np.float32)
        v = C.c_float()
        self._check(self._L.fs_lookup_query(self._h, _p(a), C.byref(v)))
        return v.value

    def set_fim_params(self, max_dist=14.0, max_angle=1.0):
        p = FimParamsC(max_dist, max_angle)
        self._check(self._L.fs_set_fim_params(self._h, C.byref(p)))

    def score_fim(self, pose7, want_fim=True, info_only=False):
        """info_only: what isPoseSafe itself needs — info_ref (and n_voxels); every other column is passed as NULL, which
        selects the worker without the 6x6 sums and with the exact table-box cull."""
        ps = np.ascontiguousarray(pose7, dtype=np.float64).reshape(-1, 7)
        n = ps.shape[0]
        info = np.zeros(n, dtype=np.float32)
        if info_only:
            nvox = np.zeros(n, dtype=np.int32)
            self._check(self._L.fs_score_fim(self._h, n, _p(ps), _p(info), None, None, None, None, _p(nvox)))
            return dict(info_ref=info, n_voxels=nvox)
        fim21 = np.zeros((n, 21), dtype=np.float32) if want_fim else None
        trace = np.zeros(n, dtype=np.float32); logdet = np.zeros(n, dtype=np.float32)
        nvis = np.zeros(n, dtype=np.int32); nvox = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_score_fim(self._h, n, _p(ps), _p(info), _p(fim21), _p(trace), _p(logdet), _p(nvis), _p(nvox)))
        return dict(info_ref=info, fim21=fim21, trace=trace, logdet=logdet, n_visible=nvis, n_voxels=nvox)

    def information_frontier_pair(self, est_pose7, triangles_xy):
        ps = np.ascontiguousarray(est_pose7, dtype=np.float64).reshape(-1, 7)
        tr = np.ascontiguousarray(triangles_xy, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(ps.shape[0], dtype=np.float32)
        self._check(self._L.fs_information_frontier_pair(self._h, ps.shape[0], _p(ps), _p(tr), _p(out)))
        return out

    # -- key-frame pose information (computeInformationForPose)
    def upload_keyframes(self, kf_pose7, kf_offsets, points_xyz):
        kf = np.ascontiguousarray(kf_pose7, dtype=np.float64).reshape(-1, 7)
        off = np.ascontiguousarray(kf_offsets, dtype=np.int32)
        pts = np.ascontiguousarray(points_xyz, dtype=np.float32).reshape(-1, 3)
        if off.shape[0] != kf.shape[0] + 1 or (off.shape[0] and off[-1] != pts.shape[0]):
            raise ValueError("kf_offsets must have n_keyframes + 1 entries ending at the number of points")
        self._check(self._L.fs_upload_keyframes(self._h, kf.shape[0], _p(kf), _p(off), _p(pts)))

    def information_for_pose(self, pose7, max_depth=2.0, hfov=1.089, max_depth_error=0.5, q_diag=0.01, radius=4.5):
        ps = np.ascontiguousarray(pose7, dtype=np.float64).reshape(-1, 7)
        n = ps.shape[0]
        prm = KeyframeParamsC(float(max_depth), float(hfov), float(max_depth_error), float(np.float32(q_diag)), float(radius))
        info = np.zeros(n, dtype=np.float32)
        cells = np.zeros(n, dtype=np.int32)
        pts = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_information_for_pose(self._h, n, _p(ps), C.byref(prm), _p(info), _p(cells), _p(pts)))
        return dict(information=info, n_cells=cells, n_points=pts)

    # -- fused
    def score_candidates(self, goal_xyz, frontier_size=None, blacklisted=None, achievable_in=None) -> np.ndarray:
        goal = np.ascontiguousarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
        n = goal.shape[0]
        fs = None if frontier_size is None else np.ascontiguousarray(frontier_size, dtype=np.int32)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        ai = None if achievable_in is None else np.ascontiguousarray(achievable_in, dtype=np.uint8)
        rec = np.zeros(n, dtype=RECORD_DTYPE)
        self._check(self._L.fs_score_candidates(self._h, n, _p(goal), _p(fs), _p(bl), _p(ai), _p(rec)))
        return rec

    def score_candidates_dev(self, n, d_goal, d_fsize, d_black, d_achin, d_records):
        """Device-pointer form (ints from tensor.data_ptr()); asynchronous on the context's stream."""
        vp = C.c_void_p
        self._check(self._L.fs_score_candidates_dev(self._h, int(n), vp(d_goal), vp(d_fsize) if d_fsize else None,
                                                    vp(d_black) if d_black else None,
                                                    vp(d_achin) if d_achin else None, vp(d_records)))

    def rank_candidates(self, records, path_length, path_heading, blacklisted=None,
                        alpha=0.25, beta=1.0, max_vx=0.5, max_wz=0.5):
        rec = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        n = rec.shape[0]
        pl = np.ascontiguousarray(path_length, dtype=np.float64)
        ph = np.ascontiguousarray(path_heading, dtype=np.float64)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        cost = np.zeros(n); au = np.zeros(n); du = np.zeros(n); order = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_rank_candidates(self._h, n, _p(rec), _p(bl), _p(pl), _p(ph), alpha, beta, max_vx, max_wz,
                                               _p(cost), _p(au), _p(du), _p(order)))
        return dict(weighted_cost=cost, arrival_utility=au, distance_utility=du, order=order)

    def get_frontier_costs(self, goal_xyz, path_length, path_heading, frontier_size=None, blacklisted=None, achievable_in=None,
                           with_fim=False, alpha=0.25, beta=1.0, max_vx=0.5, max_wz=0.5):
        """CostAssigner::getFrontierCosts as one call: arrival information (+ Fisher information) + U1 costs + order."""
        goal = np.ascontiguousarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
        n = goal.shape[0]
        pl = np.ascontiguousarray(path_length, dtype=np.float64)
        ph = np.ascontiguousarray(path_heading, dtype=np.float64)
        fs = None if frontier_size is None else np.ascontiguousarray(frontier_size, dtype=np.int32)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        ai = None if achievable_in is None else np.ascontiguousarray(achievable_in, dtype=np.uint8)
        rec = np.zeros(n, dtype=RECORD_DTYPE)
        cost = np.zeros(n); au = np.zeros(n); du = np.zeros(n); order = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_get_frontier_costs(self._h, n, _p(goal), _p(fs), _p(bl), _p(ai), _p(pl), _p(ph), alpha, beta, max_vx, max_wz,
                                                  1 if with_fim else 0, _p(rec), _p(cost), _p(au), _p(du), _p(order)))
        return dict(records=rec, weighted_cost=cost, arrival_utility=au, distance_utility=du, order=order)

    def rank_candidates_dev(self, n, d_records, d_path_length, d_path_heading, d_cost, d_au=0, d_du=0, d_order=0, d_black=0,
                            d_err=0, alpha=0.25, beta=1.0, max_vx=0.5, max_wz=0.5):
        """Device-pointer form (ints from tensor.data_ptr()); asynchronous on the context's stream."""
        vp = C.c_void_p
        o = lambda p: vp(p) if p else None
        self._check(self._L.fs_rank_candidates_dev(self._h, int(n), vp(d_records), o(d_black), vp(d_path_length), vp(d_path_heading),
                                                   alpha, beta, max_vx, max_wz, vp(d_cost), o(d_au), o(d_du), o(d_order), o(d_err)))

    def selftest_fp64(self, max_abs=256) -> int:
        bad = C.c_int64()
        self._check(self._L.fs_selftest_fp64(self._h, int(max_abs), C.byref(bad)))
        return bad.value


def _ray_params_c(max_camera_depth=2.0, delta_theta=0.10, camera_fov=1.04, robot_radius=0.60, n_rays=0, elev=(0.0,),
                  obst=(240, 254), trace=(255, 255), factor_max=1.2, factor_min=0.70, polygon=(-1e300, -1e300, 1e300, 1e300)):
    p = RayParamsC()
    p.max_camera_depth, p.delta_theta, p.camera_fov, p.robot_radius = max_camera_depth, delta_theta, camera_fov, robot_radius
    p.n_rays, p.n_elev = int(n_rays), len(elev)
    for i, e in enumerate(elev[:FS_MAX_ELEV]):
        p.elev[i] = float(e)
    p.obst_min, p.obst_max, p.trace_min, p.trace_max = int(obst[0]), int(obst[1]), int(trace[0]), int(trace[1])
    p.factor_max, p.factor_min = factor_max, factor_min
    for i in range(4):
        p.polygon[i] = float(polygon[i])
    return p


def shard_bounds(n: int, n_shards: int, shard: int):
    """fs_multi_shard_bounds: the block of `shard` — the partition rule of the multi-device scorer (needs no GPU)."""
    L = load_library()
    lo, hi = C.c_int32(), C.c_int32()
    rc = L.fs_multi_shard_bounds(int(n), int(n_shards), int(shard), C.byref(lo), C.byref(hi))
    if rc != FS_OK:
        raise FsError(rc, "fs_multi_shard_bounds: bad arguments")
    return lo.value, hi.value


class MultiScorer:
    """fs_multi: ONE process and ONE calling thread over several GPUs (or several contexts on one: repeat the ordinal).
    Staging calls are broadcast; score_candidates cuts the list into contiguous blocks, runs them side by side and returns
    the records in list order."""

    def __init__(self, devices=(0,)):
        self._L = load_library()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._L.fs_multi_create(ids, len(devices), C.byref(h))
        if rc != FS_OK:
            raise FsError(rc, "fs_multi_create failed: no gfx950 device under one of the ordinals (no CPU fallback exists)")
        self._h = h
        self.n_devices = len(devices)

    def _check(self, rc):
        if rc != FS_OK:
            raise FsError(rc, (self._L.fs_multi_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.fs_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._check(self._L.fs_multi_set_option(self._h, key.encode(), float(value)))

    def set_ray_params(self, **kw):
        p = _ray_params_c(**kw)
        self._check(self._L.fs_multi_set_ray_params(self._h, C.byref(p)))

    def upload_grid(self, cells, origin, resolution):
        c = np.ascontiguousarray(cells, dtype=np.uint8)
        if c.ndim == 2:
            c = c[None]
        nz, ny, nx = c.shape
        o = (C.c_double * 3)(*[float(v) for v in origin])
        self._check(self._L.fs_multi_upload_grid(self._h, _p(c), nx, ny, nz, C.byref(o), float(resolution)))

    def update_grid_region(self, x0, y0, z0, window, view=False):
        ptr, sx, sy, sz, rs, ss, keep = _window_args(window, view)
        self._check(self._L.fs_multi_update_grid_region(self._h, int(x0), int(y0), int(z0), sx, sy, sz, ptr, rs, ss))

    def upload_landmarks(self, xyz):
        lm = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._check(self._L.fs_multi_upload_landmarks(self._h, _p(lm), lm.shape[0]))

    def lookup_generate(self, bounds=None):
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.float32)
        self._check(self._L.fs_multi_lookup_generate(self._h, _p(b)))

    def set_fim_params(self, max_dist=14.0, max_angle=1.0):
        p = FimParamsC(max_dist, max_angle)
        self._check(self._L.fs_multi_set_fim_params(self._h, C.byref(p)))

    def max_arrival(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._check(self._L.fs_multi_max_arrival(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(max_value=a.value, max_gt=b.value, min_gt=c.value)

    def score_arrival(self, goal_xyz, frontier_size=None, blacklisted=None, achievable_in=None, n_rays_total=0):
        """n_rays_total = n_elev * n_yaw to get the per-ray counts back (0: not requested)."""
        goal = np.ascontiguousarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
        n = goal.shape[0]
        fs = None if frontier_size is None else np.ascontiguousarray(frontier_size, dtype=np.int32)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        ai = None if achievable_in is None else np.ascontiguousarray(achievable_in, dtype=np.uint8)
        rc_arr = np.zeros((n, n_rays_total), dtype=np.int32) if n_rays_total else None
        arrival = np.zeros(n, dtype=np.int32); argmax = np.zeros(n, dtype=np.int32)
        yaw = np.zeros(n, dtype=np.float64); ach = np.zeros(n, dtype=np.uint8); status = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_multi_score_arrival(self._h, n, _p(goal), _p(fs), _p(bl), _p(ai), _p(rc_arr),
                                                   _p(arrival), _p(argmax), _p(yaw), _p(ach), _p(status)))
        return dict(ray_counts=rc_arr, arrival=arrival, argmax=argmax, yaw=yaw, achievable=ach, status=status)

    def score_candidates(self, goal_xyz, frontier_size=None, blacklisted=None, achievable_in=None) -> np.ndarray:
        goal = np.ascontiguousarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
        n = goal.shape[0]
        fs = None if frontier_size is None else np.ascontiguousarray(frontier_size, dtype=np.int32)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        ai = None if achievable_in is None else np.ascontiguousarray(achievable_in, dtype=np.uint8)
        rec = np.zeros(n, dtype=RECORD_DTYPE)
        self._check(self._L.fs_multi_score_candidates(self._h, n, _p(goal), _p(fs), _p(bl), _p(ai), _p(rec)))
        return rec


    def score_fim(self, pose7, want_fim=True, info_only=False):
        """fs_multi_score_fim: FrontierScorer.score_fim over all members, every column in list order."""
        ps = np.ascontiguousarray(pose7, dtype=np.float64).reshape(-1, 7)
        n = ps.shape[0]
        info = np.zeros(n, dtype=np.float32)
        if info_only:
            nvox = np.zeros(n, dtype=np.int32)
            self._check(self._L.fs_multi_score_fim(self._h, n, _p(ps), _p(info), None, None, None, None, _p(nvox)))
            return dict(info_ref=info, n_voxels=nvox)
        fim21 = np.zeros((n, 21), dtype=np.float32) if want_fim else None
        trace = np.zeros(n, dtype=np.float32); logdet = np.zeros(n, dtype=np.float32)
        nvis = np.zeros(n, dtype=np.int32); nvox = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_multi_score_fim(self._h, n, _p(ps), _p(info), _p(fim21), _p(trace), _p(logdet), _p(nvis), _p(nvox)))
        return dict(info_ref=info, fim21=fim21, trace=trace, logdet=logdet, n_visible=nvis, n_voxels=nvox)

    def get_frontier_costs(self, goal_xyz, path_length, path_heading, frontier_size=None, blacklisted=None, achievable_in=None,
                           with_fim=False, alpha=0.25, beta=1.0, max_vx=0.5, max_wz=0.5):
        """fs_multi_get_frontier_costs: blocks scored on their devices, gathered device to device, ranked on member 0's GPU."""
        goal = np.ascontiguousarray(goal_xyz, dtype=np.float64).reshape(-1, 3)
        n = goal.shape[0]
        pl = np.ascontiguousarray(path_length, dtype=np.float64)
        ph = np.ascontiguousarray(path_heading, dtype=np.float64)
        fs = None if frontier_size is None else np.ascontiguousarray(frontier_size, dtype=np.int32)
        bl = None if blacklisted is None else np.ascontiguousarray(blacklisted, dtype=np.uint8)
        ai = None if achievable_in is None else np.ascontiguousarray(achievable_in, dtype=np.uint8)
        rec = np.zeros(n, dtype=RECORD_DTYPE)
        cost = np.zeros(n); au = np.zeros(n); du = np.zeros(n); order = np.zeros(n, dtype=np.int32)
        self._check(self._L.fs_multi_get_frontier_costs(self._h, n, _p(goal), _p(fs), _p(bl), _p(ai), _p(pl), _p(ph), alpha, beta, max_vx, max_wz,
                                                        1 if with_fim else 0, _p(rec), _p(cost), _p(au), _p(du), _p(order)))
        return dict(records=rec, weighted_cost=cost, arrival_utility=au, distance_utility=du, order=order)

    def gather_mode(self) -> int:
        rc = self._L.fs_multi_gather_mode(self._h)
        if rc < 0:
            self._check(rc)
        return rc

    def last_error(self) -> str:
        return (self._L.fs_multi_last_error(self._h) or b"").decode()

    def member(self, i: int):
        """fs_multi_ctx(m, i) as a borrowed handle for fs_* calls on one member (counters, options, limits)."""
        h = self._L.fs_multi_ctx(self._h, int(i))
        if not h:
            raise FsError(FS_E_INVALID, f"no member {i}")
        return C.c_void_p(h)

    def set_arrival_limits(self, max_gt, min_gt):
        for i in range(self.n_devices):
            rc = self._L.fs_set_arrival_limits(self.member(i), float(max_gt), float(min_gt))
            if rc != FS_OK:
                raise FsError(rc, f"fs_set_arrival_limits on member {i}")


def record_status(rec):
    return (rec["flags"] >> 8) & 0xFF


def record_achievable(rec):
    return (rec["flags"] & 1).astype(np.uint8)


def record_nvoxels(rec):
    return (rec["flags"] >> 16) & 0xFFFF
