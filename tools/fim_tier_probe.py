"""Development probe: tier-1 hash-table size / hand-over prediction vs. kernel times on a config."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = fs.synth.make_workload(name)
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
sc.enable_kernel_timing(True)
ref = None
for bits1, skip in [(14, 12), (14, 14), (14, 11), (14, 10), (14, 9), (14, 12)]:
    sc.set_option("fim.bits1", bits1); sc.set_option("fim.skip32", skip)
    for rnd in range(3):
        for k in (0, 1, 2): sc.kernel_time(k)
        sc.get_counter(4, True); sc.get_counter(5, True)
        rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    t = [sc.kernel_time(k) for k in (0, 1, 2)]
    t2, t3 = sc.get_counter(4, True), sc.get_counter(5, True)
    if ref is None: ref = rec.copy()
    same = all(np.array_equal(ref[f], rec[f]) for f in ("arrival", "argmax", "n_visible", "flags"))
    print("bits1 %d skip %2d/32: ray %.3f fim %.3f tiers %.3f ms ; multi-pass %d HBM tier %d ; ints equal %s ; info maxrel %.2e" % (
        bits1, skip, t[0][0] / max(1, t[0][1]), t[1][0] / max(1, t[1][1]), t[2][0] / max(1, t[2][1]), t2, t3, same,
        np.max(np.abs(ref["info_ref"] - rec["info_ref"]) / np.maximum(1, np.abs(ref["info_ref"])))), flush=True)
nv = ref["n_visible"]; vox = (ref["flags"] >> 16) & 0xFFFF
print("n_visible mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (nv.mean(), *np.percentile(nv, [50, 90, 99]), nv.max()))
print("n_voxels  mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (vox.mean(), *np.percentile(vox, [50, 90, 99]), vox.max()))
for thr in (1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288):
    print("  voxels <= %5d : %.3f of candidates" % (thr, (vox <= thr).mean()))
