"""Development tool: per-candidate workgroup durations of the FIM tiers (option fim.debug)."""
import importlib, sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload("C3")
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
sc.set_option("fim.debug", 1)
L = fs.load_library()
L.fs_debug_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
for mask in (2, 4, 8, 14):
    sc.set_option("fim.ablate", mask)
    r2 = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
    o2 = np.zeros((r2.shape[0], 8), dtype=np.uint64)
    assert L.fs_debug_fetch(sc._h, r2.shape[0], o2.ctypes.data_as(C.c_void_p)) == 0
    p2 = o2[:, 3:].astype(np.float64)
    print("ablate %2d: score clk per call %.0f, loop %.0f" % (mask, p2[:,3].sum()/max(p2[:,1].sum(),1), p2[:,2].mean()))
sc.set_option("fim.ablate", 0)
rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
n = rec.shape[0]
out = np.zeros((n, 8), dtype=np.uint64)
L = fs.load_library()
L.fs_debug_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
assert L.fs_debug_fetch(sc._h, n, out.ctypes.data_as(C.c_void_p)) == 0
us = out.astype(np.float64) / 100.0
ph = out[:, 3:].astype(np.float64)
print('wave0 clocks: cull %.0f  score calls %.2f  loop %.0f  score clk %.0f (%.0f per call)  reduce %.0f' % (ph[:,0].mean(), ph[:,1].mean(), ph[:,2].mean(), ph[:,3].mean(), ph[:,3].sum()/max(ph[:,1].sum(),1), ph[:,4].mean()))
z = rec['n_visible'] == 0

nv = rec["n_visible"]; vox = (rec["flags"] >> 16) & 0xFFFF
t1 = us[:, 0]
print("tier1 WG duration us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f ; sum/512 = %.1f us" % (t1.mean(), np.percentile(t1, 50), np.percentile(t1, 90), np.percentile(t1, 99), t1.max(), t1.sum() / 512))
for lo, hi in [(0, 1), (1, 500), (500, 2000), (2000, 5000), (5000, 10000), (10000, 20000), (20000, 40000)]:
    sel = (nv >= lo) & (nv < hi)
    if sel.any():
        print("  n_visible [%5d,%5d): %5d cands, tier1 mean %.2f us max %.2f us, voxels mean %.0f" % (lo, hi, sel.sum(), t1[sel].mean(), t1[sel].max(), vox[sel].mean()))
t2 = us[:, 1]; sel = t2 > 0
print("tier2: %d cands, mean %.2f us max %.2f us; their tier1 mean %.2f us; n_visible mean %.0f voxels mean %.0f" % (sel.sum(), t2[sel].mean() if sel.any() else 0, t2[sel].max() if sel.any() else 0, t1[sel].mean() if sel.any() else 0, nv[sel].mean() if sel.any() else 0, vox[sel].mean() if sel.any() else 0))
t3 = us[:, 2]; s3 = t3 > 0
print("tier3: %d cands, us %s, their tier2 us %s, n_visible %s voxels %s tested %s" % (s3.sum(), t3[s3], t2[s3], nv[s3], vox[s3], out[s3, 3] * 64))

tested = out[:, 3].astype(np.float64) * 64
ok = tested > 0
ratio = vox[ok] / tested[ok]
print("voxels/tested: mean %.3f p50 %.3f p90 %.3f p99 %.3f max %.3f" % (ratio.mean(), np.percentile(ratio,50), np.percentile(ratio,90), np.percentile(ratio,99), ratio.max()))
for thr in (8000, 10000, 12288):
    big = vox > thr
    print(" voxels > %d: %d cands; their tested: min %.0f p10 %.0f mean %.0f ; cands with tested >= that min: %d" % (thr, big.sum(), tested[big].min() if big.any() else 0, np.percentile(tested[big],10) if big.any() else 0, tested[big].mean() if big.any() else 0, (tested >= (tested[big].min() if big.any() else 1e18)).sum()))
print("tested: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (tested.mean(), np.percentile(tested,50), np.percentile(tested,90), np.percentile(tested,99), tested.max()))
