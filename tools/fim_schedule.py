"""Development probe (library built with FS_FIM_SCHEDULE=1): the schedule of the persistent FIM grid — which workgroup
started which candidate when, and how long it took.  Prints how long the grid keeps running after its work list is empty
(the drain) and what an ideal order would save.  Optional argument: workload (default C3); --reverse scores the list in
reversed spatial order (heavy candidates last on C3)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
w = fs.synth.make_workload(args[0] if args else "C3")
sc = fs.FrontierScorer(0)
for kv in [a[2:] for a in sys.argv[1:] if a.startswith("--") and "=" in a]:
    k, v = kv.split("=", 1)
    sc.set_option(k, float(v))
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
n = w.goals.shape[0]
for _ in range(3): rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
sc.enable_kernel_timing(True); sc.kernel_time(1)
rec = sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
fim_ms = sc.kernel_time(1)[0]
rows = np.array([[sc.get_counter(32 + 2 * c + k) for k in range(2)] for c in range(min(n, 32768))], dtype=np.int64)
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fim_schedule.npy"), rows)
t, w1 = rows[:, 0], rows[:, 1]
dur, wg, parts = (w1 & 0xFFFFFFFF) / 100.0, (w1 >> 32) & 0xFFFFFF, (w1 >> 56) & 0xFF
st = (t - t.min()) / 100.0
end = st + dur
n_wg = int(wg.max()) + 1
wg_end = np.array([end[wg == g].max() for g in range(n_wg) if (wg == g).any()])
last_fetch = st.max()
print("fs_fim_kernel %.1f us (hipEvent); %d workgroups; candidate time mean %.1f us, p50 %.1f, p99 %.1f, max %.1f; %d multi-pass" % (
    fim_ms * 1e3, n_wg, dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max(), int((parts > 1).sum())))
print("last candidate started at %.1f us; workgroups finish at p5/p50/p95/max = %.1f / %.1f / %.1f / %.1f us" % (
    last_fetch, np.percentile(wg_end, 5), np.percentile(wg_end, 50), np.percentile(wg_end, 95), wg_end.max()))
print("work = sum of candidate times / workgroups = %.1f us -> idle share of the grid %.1f %%" % (dur.sum() / n_wg, 100.0 * (1.0 - dur.sum() / n_wg / wg_end.max())))
for lo in range(0, int(end.max()) + 100, 100):
    m = (st >= lo) & (st < lo + 100)
    if m.any():
        print("  started in [%4d, %4d) us: %5d candidates, mean %.1f us, max %.1f us" % (lo, lo + 100, m.sum(), dur[m].mean(), dur[m].max()))
