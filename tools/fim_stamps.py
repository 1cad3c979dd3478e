"""Development probe (library built with FS_FIM_STAMPS=1): average cycles per candidate and wave in each phase of the tier-1 worker."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
w = fs.synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "C3")
sc = fs.FrontierScorer(0)
sc.set_ray_params(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks); sc.lookup_generate(); sc.set_fim_params(14.0, 1.0)
sc.max_arrival()
for _ in range(3): sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
for k in range(16, 32): sc.get_counter(k, True)
reps = 5
for _ in range(reps): sc.score_candidates(w.goals, w.frontier_size, w.blacklisted)
st = [sc.get_counter(k) for k in range(16, 32)]
waves = st[15]
names = ["top: sizes + table clear", "barrier after clear", "scoring loop", "flush", "cull of next candidate", "dpp reduce", "barrier after reduce", "outputs"]
n_iter = reps * w.goals.shape[0] * 8      # wave-iterations (8 waves per candidate)
tot = sum(st[:8])
print("waves %d, cycle-counter ticks per candidate and wave (total %.0f):" % (waves, tot / n_iter))
for k in range(8):
    print("  %-28s %9.0f  %5.1f %%" % (names[k], st[k] / n_iter, 100.0 * st[k] / tot))
print("  of the scoring loop: %.0f ticks in %.2f score calls (%.0f each), the rest = chunk tests + compaction" % (
    st[8] / n_iter, st[9] / n_iter, st[8] / max(1, st[9])))

try:
    per_wait = [sc.get_counter(32 + k) for k in range(8)]
    per_score = [sc.get_counter(48 + k) for k in range(8)]
    n_c = reps * w.goals.shape[0]
    print("by wave index (FS_FIM_STAMPS=wave builds): wait at the reduction barrier / scoring + flush, ticks per candidate")
    for k in range(8):
        print("  wave %d: wait %7.0f   scoring %7.0f" % (k, per_wait[k] / n_c, per_score[k] / n_c))
except Exception as e:
    pass
