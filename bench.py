#!/usr/bin/env python3
"""bench.py — candidate-goals scored per second (ray-cast + FIM) on a 512^3 grid (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (fs_score_candidates_dev: ray-march -> FIM accumulate -> record
pack) over this rank's block of candidates, followed — when N > 1 — by the single RCCL all-gather
of the 32-byte records.  Workload at N = 1: BASELINE.json configs[2] ("C3": 512^3 grid, 20 k
candidates, 100 k landmarks, 256 rays/candidate = 64 yaw x 4 elevation rings, L = 40 cells).
  --scaling weak   (default) every rank scores 20 k candidates of a 20 k*N list: per-GPU work fixed.
  --scaling strong configs[3] ("C4": 160 k candidates in total) split over the N ranks: total work fixed
                   (SURVEY.md §8(e) "Reporting").
Grid, landmarks, lookup table and the candidate arrays are resident in HBM before the timed region.

Launching: with --gpus N > 1 and no WORLD_SIZE in the environment this process is only a LAUNCHER: before anything
touches the GPU it starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a child,
relays rank 0's JSON line and exits with the child's code.  Under an external launcher (WORLD_SIZE set) --gpus must
equal WORLD_SIZE or the run fails.  Rank 0 prints ONE JSON line.

Timing: W untimed warm-up steps, then `--repeats` blocks of EXACTLY K steps each, every block bracketed by
barrier + torch.cuda.synchronize on both sides and reduced with MAX over the ranks; `value` / `ms_per_step` come from
the MEDIAN block, the minimum and every block's time are reported beside it.  Only the last block carries the
per-kernel hipEvent pairs the roofline needs (they cost ~3 % of a step), so the median is an event-free block.

Pipelining: --pipeline P (default 1: one stream, batches strictly back to back) stages the map and the cloud in P scorer
contexts, each on its own HIP stream, and sends batch k to context k mod P, so that the next batch's ray-march overlaps the
drain of the previous batch's persistent FIM workgroups.  Every step is still one complete batch.  Measured on C3
(profiles/r02/c3_pipeline.json): P = 2 gives 1.011 against 1.031 ms per batch, P = 3 loses; with P > 1 the FIM kernels of two
batches share the CUs, their hipEvent durations double and stop measuring the kernel — hence the default of 1.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
METRIC = "candidate-goals scored/sec (raycast+FIM) on 512^3 grid; % HBM roofline"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps each; value = the median block")
    ap.add_argument("--min-timed-seconds", type=float, default=0.5, help="keep adding timed blocks of --steps steps (beyond --repeats) until the timed GPU work adds up to this much: 5 blocks of 20 C3 steps are only 79 ms, too thin for a median (0 = exactly --repeats blocks)")
    ap.add_argument("--pipeline", type=int, default=1, help="scorer contexts (HIP streams) per GPU; successive batches alternate over them (1, the default = one stream, batches strictly back to back; 2 measured +2 %% on C3, but the two FIM kernels then share the CUs and their per-launch durations no longer measure the kernel)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--workload", default="C3", help="C3 (headline) | C5 | REF2D (the reference's own 2-D operating point: 63 rays, L = 40) | C2 | C1")
    ap.add_argument("--depth-cells", type=int, default=0, help="ray length L in cells (0: the config's 40 = 2 m; BASELINE.md's secondary throughput run uses 160)")
    ap.add_argument("--fim-angle", type=float, default=1.0, help="visibility cone half-angle in rad for the timed steps (1.0: the build's definition, SURVEY.md App. A.3; 4.0 = the reference's own request, FisherInfoManager.cpp:63-64: cone off).  The N = 1 line always carries the other one under other_operating_points.reference_request_visibility")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time (three runs, median); 0 disables")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling block (160 k candidates split over the ranks) that follows the weak blocks")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI; the measured configuration) | gloo (rehearsal of the N > 1 path on one GPU: ranks share the device, records are gathered through host memory)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="fs_set_option knobs, e.g. ray.layout=2")
    ap.add_argument("--latency", action="store_true", help="the drop-in operating point instead of the throughput line: p50 / p99 of ONE fs_score_candidates call (host buffers in, records out, synchronised) at 1 / 50 / 200 / 2000 frontiers on REF2D and C1, of fs_score_fim(n = 1) and of fs_frontier_clusters on a 512^2 costmap, each with the oracle's single-thread time beside it")
    ap.add_argument("--latency-calls", type=int, default=300, help="calls per latency figure")
    ap.add_argument("--rehearse-multi", action="store_true", help="one rank, but through every code path of the N > 1 run: a one-rank RCCL group, the asynchronous all-gather in every step, the ranking on the gathered list, the multi_gpu keys and the block-sampled parity gate (what a one-GPU box can rehearse of the multi-GPU line; never a scaling figure)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous rehearsal without a GPU: the ranks meet over gloo, all-gather a dummy record block and rank 0 prints a line marked dry_run (no value)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ launcher (never touches the GPU)

def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(args, argv, port=None):
    """The child command the launcher runs: one process per GPU under torch.distributed.run."""
    port = port or _free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args, argv) -> int:
    """--gpus N > 1 without a launcher around us: start N fresh ranks and relay rank 0's line.  This process has not
    initialised the GPU (no torch.cuda call, no HIP library loaded) and never does."""
    if not args.dry_run:
        # build (or find up to date) the HIP library HERE, once, before the ranks exist: hipcc needs no GPU, the build takes an
        # exclusive lock, and the ranks' own load_library() then finds the content stamp matching and compiles nothing
        importlib.import_module("fit-slam_amd._build").build()
    cmd = launcher_command(args, argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench launcher] {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("[bench launcher] the ranks exited without a result line", file=sys.stderr)
        return 3
    if line is not None:
        print(line, flush=True)
    return rc


# ------------------------------------------------------------------ helpers

def source_hash() -> str:
    """Identity of the kernels a counter profile belongs to: SHA-256 over the HIP sources and headers."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "fit-slam_amd", "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    for f in names:
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "fitslam_frontier.h"), "rb").read())
    return h.hexdigest()[:16]


def counter_profile(workload: str, depth_cells: int, fim_angle: float = 1.0):
    """PMC-derived figures (profiles/pmc_summary.json, written by tools/pmc_collect.py on the GPU box) — used ONLY when
    the file was measured on exactly these kernel sources and this workload; otherwise the line says unmeasured."""
    for name in ("pmc_summary.json", "pmc_summary_ref_request.json"):      # (the headline's, and the one collected at --fim-angle 4.0)
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if j.get("source_hash") != source_hash() or j.get("workload") != workload or int(j.get("depth_cells", 0)) != int(depth_cells):
            continue
        if abs(float(j.get("fim_angle", 1.0)) - float(fim_angle)) > 1e-9:  # (another visibility volume is another kernel instantiation)
            continue
        return j
    return None


def host_cpu_topology() -> dict:
    """What 'all host cores' means on this box: logical CPUs this process may run on, physical cores among them (distinct
    (package, core id) pairs of /proc/cpuinfo), SMT state, and the cgroup's CPU quota if there is one."""
    usable = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in usable:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in usable:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except (OSError, ValueError):
        pass
    smt = None
    try:
        smt = open("/sys/devices/system/cpu/smt/active").read().strip() == "1"
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    return {"logical_cpus_usable": len(usable), "physical_cores_usable": len(cores) or None, "smt_active": smt,
            "cgroup_cpu_quota_cores": quota}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def compare_records(r, arr, fim, fim_angle):
    """GPU records against the oracle's arrival / pose information of the same candidates."""
    n_s = int(r.shape[0])
    ok = arr["status"] == 0
    ints = (np.array_equal(r["arrival"], arr["arrival"]) and np.array_equal(r["argmax"], arr["argmax"]) and
            np.array_equal((r["flags"] >> 8) & 0xFF, arr["status"]) and
            np.array_equal(r["flags"] & 1, arr["achievable"]) and
            np.array_equal(r["n_visible"][ok], fim["n_visible"][ok]) and
            np.array_equal(((r["flags"] >> 16) & 0xFFFF)[ok], np.minimum(fim["n_voxels"][ok], 65535)))
    parity = importlib.import_module("fit-slam_amd.parity")      # the ONE tolerance rule (tests use the same functions)
    e_info = parity.rel_err(r["info_ref"][ok], fim["info_f64"][ok])
    e_tr = parity.rel_err(r["trace"][ok], fim["trace"][ok])
    # D-optimality: no reference counterpart (the reference keeps the trace only) — against the float64 oracle under
    # fit-slam_amd/parity.py's rule: plain 1e-4 of max(1, |log det|) plus kappa(F) * 2^-24, the rounding of a float32 F's
    # entries (only poses with three or four visible landmarks, kappa >= 1e5, need the second term: profiles/r05/logdet_probe_*.json).
    # The share within PLAIN 1e-4 is reported next to the share within the rule.
    gate = parity.logdet_gate(r["logdet"], fim["logdet"], fim["fim"], consider=ok, n_visible=fim["n_visible"])
    out = {"n": n_s, "integers_bit_exact": bool(ints), "info_max_rel_err": e_info, "trace_max_rel_err": e_tr}
    out.update({k: v for k, v in gate.items() if k != "ok"})
    out["visibility"] = f"14 m, {fim_angle} rad"
    out["ok"] = bool(ints and e_info <= parity.REL and e_tr <= parity.REL and gate["ok"])
    return out


def red_gates(obj, path="") -> list:
    """Every `parity`-like block of the result line whose `ok` is false (paths into the line)."""
    red = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            here = f"{path}.{k}" if path else k
            if isinstance(v, dict) and "parity" in k and v.get("ok") is False:
                red.append(here)
            red += red_gates(v, here)
    return red


def sample_parity(w, arrival_kw, mx, goals, fsize, black, rec, fim_angle, where):
    """N > 1: a sample spread over every rank's block, scored by the oracle on rank 0's host cores (outside every timed region)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O   # the checker
    O.build()
    T = min(16, os.cpu_count() or 1)
    G = O.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    arr = O.arrival_information(G, O.RayParams(**arrival_kw), goals, fsize, black, min_gt=mx["min_gt"], faithful=False, n_threads=T, want_ray_counts=False)
    fim = O.pose_information(O.Table.generate(), w.landmarks, O.poses_from_yaw(goals, arr["yaw"]), 14.0, fim_angle, n_threads=T, want_f64=True)
    out = compare_records(rec, arr, fim, fim_angle)
    out["sample"] = where
    return out


def cpu_baseline(w, arrival_kw, n_total, target_s, gpu_rec, mx, fim_angle=1.0):
    """Time the oracle (kind 'port') on a bounded sample of the same workload, single thread, three runs (median), and
    gate the GPU records of that sample against it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O   # the checker / CPU baseline only

    O.build()
    G = O.Grid(w.cells, origin=w.origin, resolution=w.resolution)
    P = O.RayParams(**arrival_kw)
    table = O.Table.generate()

    def run(lo, hi, threads, faithful):
        t0 = time.perf_counter()
        arr = O.arrival_information(G, P, w.goals[lo:hi], w.frontier_size[lo:hi], w.blacklisted[lo:hi],
                                    min_gt=mx["min_gt"], faithful=faithful, n_threads=threads, want_ray_counts=False)
        poses = O.poses_from_yaw(w.goals[lo:hi], arr["yaw"])
        fim = O.pose_information(table, w.landmarks, poses, 14.0, fim_angle, n_threads=threads, want_f64=True)
        return time.perf_counter() - t0, arr, fim

    probe = min(32, n_total)
    dt, _, _ = run(0, probe, 1, True)
    per_run = target_s * 0.2                       # 3 faithful runs + clean + all-cores ~ target_s
    n_s = int(max(probe, min(n_total, per_run / max(dt / probe, 1e-9))))
    times = []
    for _ in range(3):
        dt, arr, fim = run(0, n_s, 1, True)
        times.append(dt)
    med = float(np.median(times))
    out = {"value": n_s / med, "unit": "candidate-goals/s", "cores": 1, "kind": "port",
           "cpu_model": cpu_model(), "host_cores_visible": os.cpu_count(),
           "sample": f"first {n_s} of {n_total} candidates of the same workload, oracle/ C restatement "
                     f"(reference-faithful control flow), single thread, median of 3 runs of {med:.1f} s",
           "runs_s": times, "min_run_value": n_s / max(times), "max_run_value": n_s / min(times)}
    dt_c, _, _ = run(0, n_s, 1, False)
    out["clean_value"] = n_s / dt_c          # BASELINE.md `cpu_clean`: same results, no per-ray vectors / O(k^2) dedupe, one thread
    topo = host_cpu_topology()
    out["host"] = topo
    cores = min(16, topo["logical_cpus_usable"])     # the GPU box's CPU share for one GPU
    n_o = min(n_total, n_s * min(cores, 16))
    dt_o, _, _ = run(0, n_o, cores, False)
    out["omp_value"] = n_o / dt_o
    out["omp_cores"] = cores
    # OpenMP over candidates on ALL host cores of the box — every logical CPU this process may run on,
    # and (SMT on) one thread per physical core beside it.  A cgroup CPU quota (this pool: 16 cores' worth for a one-GPU box)
    # throttles such a run to the quota however many threads it starts; the figures are reported as measured, with the quota
    # beside them, and `best_*` names the fastest CPU configuration found.  Sample sized for ~1.5 s per run at the rate the
    # quota allows (the whole list when that is shorter), best of two runs (the first also pays the threads' creation).
    for key, thr in (("omp_all", topo["logical_cpus_usable"]), ("omp_physical", topo["physical_cores_usable"])):
        if not thr or thr <= cores or (key == "omp_physical" and thr == topo["logical_cpus_usable"]):
            continue
        eff = min(thr, topo["cgroup_cpu_quota_cores"] or thr)
        n_a = int(min(n_total, max(n_o, out["omp_value"] * (eff / cores) * 1.5, 8 * thr)))
        dt_a = min(run(0, n_a, thr, False)[0] for _ in range(2))
        out[key + "_value"] = n_a / dt_a
        out[key + "_cores"] = thr
        out[key + "_sample"] = f"first {n_a} of {n_total} candidates, {thr} OpenMP threads (schedule dynamic), best of 2 runs of {dt_a:.2f} s"
    best = max([("value", "cores"), ("omp_value", "omp_cores"), ("omp_all_value", "omp_all_cores"), ("omp_physical_value", "omp_physical_cores")],
               key=lambda kv: out.get(kv[0], 0.0))
    out["best_value"], out["best_threads"] = out[best[0]], out[best[1]]
    if topo["cgroup_cpu_quota_cores"] and topo["cgroup_cpu_quota_cores"] < topo["logical_cpus_usable"]:
        out["host_note"] = (f"the box shows {topo['logical_cpus_usable']} logical CPUs ({topo['physical_cores_usable']} cores) but its cgroup grants this job "
                            f"{topo['cgroup_cpu_quota_cores']:g} cores' worth of CPU time: runs with more threads than that are throttled, not faster")
    parity = compare_records(gpu_rec[:n_s], arr, fim, fim_angle) if gpu_rec is not None else None
    return out, parity


# ------------------------------------------------------------------ one rank

# ------------------------------------------------------------------ latency at the reference's operating point

def _percentiles(us):
    a = np.sort(np.asarray(us))
    return {"p50_us": float(a[len(a) // 2]), "p99_us": float(a[min(len(a) - 1, int(0.99 * len(a)))]), "min_us": float(a[0]), "calls": int(len(a))}


def run_latency(args) -> int:
    """The reference scores tens of frontiers per behaviour-tree tick (DEP/src/FrontierCostsManager.cpp:74-119) and ONE pose per
    isPoseSafe (FIP/src/fisher_information/FisherInfoBTPlugin.cpp:24-57): what a drop-in user feels is the latency of one call."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    fs = importlib.import_module("fit-slam_amd")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O   # CPU baseline beside every figure (never inside a timed GPU call)
    O.build()
    table = O.Table.generate()
    out = {"metric": "latency of one scoring call at the reference's operating point (host buffers in, results out, synchronised)",
           "unit": "us", "visibility": f"14 m, {args.fim_angle} rad" + (" (cone off: the reference's request)" if args.fim_angle >= np.pi else ""), "calls_per_figure": args.latency_calls, "workloads": {}}
    for wl in ("REF2D", "C1"):
        w = fs.synth.make_workload(wl, n_cand=2000)
        kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                  robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
        sc = fs.FrontierScorer(device=0)
        for kv in args.option:
            k, v = kv.split("=", 1)
            sc.set_option(k, float(v))
        sc.set_ray_params(**kw); sc.upload_grid(w.cells, w.origin, w.resolution); sc.upload_landmarks(w.landmarks)
        sc.lookup_generate(); sc.set_fim_params(14.0, args.fim_angle)
        mx = sc.max_arrival()
        G = O.Grid(w.cells, origin=w.origin, resolution=w.resolution)
        P = O.RayParams(**kw)
        res = {"grid": list(w.cells.shape), "landmarks": int(w.landmarks.shape[0]), "rays_per_candidate": sc.n_yaw * sc.n_elev, "score_candidates": {}}
        rec_all = None
        for n in (1, 50, 200, 2000):
            g, f, b = w.goals[:n], w.frontier_size[:n], w.blacklisted[:n]
            for _ in range(20):
                rec = sc.score_candidates(g, f, b)
            ts = []
            for _ in range(args.latency_calls):
                t0 = time.perf_counter()
                rec = sc.score_candidates(g, f, b)
                ts.append((time.perf_counter() - t0) * 1e6)
            if n == 2000:
                rec_all = rec
            t0 = time.perf_counter()
            arr = O.arrival_information(G, P, g, f, b, min_gt=mx["min_gt"], faithful=True, n_threads=1, want_ray_counts=False)
            fim = O.pose_information(table, w.landmarks, O.poses_from_yaw(g, arr["yaw"]), 14.0, args.fim_angle, n_threads=1, want_f64=True)
            cpu_us = (time.perf_counter() - t0) * 1e6
            ok = bool(np.array_equal(rec["arrival"], arr["arrival"]) and np.array_equal(rec["n_visible"][arr["status"] == 0], fim["n_visible"][arr["status"] == 0]))
            res["score_candidates"][str(n)] = dict(_percentiles(ts), cpu_oracle_1thread_us=cpu_us, parity_integers=ok)
        # the whole cost assignment of a tick as ONE call (fs_get_frontier_costs: arrival information + U1 + order; what the
        # reference's assignCosts computes), path columns as the Euclidean stand-in planner would set them
        res["get_frontier_costs"] = {}
        fc_by_n = {}
        for n in (50, 200, 2000):
            g, f, b = w.goals[:n], w.frontier_size[:n], w.blacklisted[:n]
            pl = np.hypot(g[:, 0], g[:, 1]) + 0.5
            ph = np.abs(np.arctan2(g[:, 1], g[:, 0]))
            sc.set_arrival_limits(max(mx["max_gt"], 1000.0) * 4, mx["min_gt"])      # (C1's 3.2 m map cannot hold the calibration fan: limits by hand)
            for _ in range(20):
                sc.get_frontier_costs(g, pl, ph, f, b, with_fim=False)
            ts = []
            for _ in range(args.latency_calls):
                t0 = time.perf_counter()
                fc = sc.get_frontier_costs(g, pl, ph, f, b, with_fim=False)
                ts.append((time.perf_counter() - t0) * 1e6)
            sc.set_arrival_limits(mx["max_gt"], mx["min_gt"])
            fc_by_n[n] = fc
            res["get_frontier_costs"][str(n)] = dict(_percentiles(ts), order_is_a_permutation=bool(np.array_equal(np.sort(fc["order"]), np.arange(n))))
        # the same call through the multi-device form on TWO contexts of this one GPU (fs_multi_get_frontier_costs: blocks scored side
        # by side, gathered on the device, ranked, one transfer out) — what the in-process multi-GPU path adds to a call, by the
        # way a block travels: 1 = written into the gathered list in place (members of one GPU), 3 = device copy on the member's
        # stream + event (the machinery a second GPU uses with hipMemcpyPeerAsync), 2 = the page-locked bounce (no peer access).
        # A rehearsal on one GPU: never a scaling figure.
        if wl == "REF2D":
            res["multi_get_frontier_costs_two_contexts_one_gpu"] = {}
            m = fs.MultiScorer(devices=(0, 0))
            m.set_ray_params(**kw); m.upload_grid(w.cells, w.origin, w.resolution)
            m.set_arrival_limits(max(mx["max_gt"], 1000.0) * 4, mx["min_gt"])
            for mode in (1, 3, 2):
                m.set_option("multi.gather", mode)
                for n in (50, 2000):
                    g, f, b = w.goals[:n], w.frontier_size[:n], w.blacklisted[:n]
                    pl = np.hypot(g[:, 0], g[:, 1]) + 0.5
                    ph = np.abs(np.arctan2(g[:, 1], g[:, 0]))
                    for _ in range(20):
                        m.get_frontier_costs(g, pl, ph, f, b, with_fim=False)
                    ts = []
                    for _ in range(args.latency_calls):
                        t0 = time.perf_counter()
                        fm = m.get_frontier_costs(g, pl, ph, f, b, with_fim=False)
                        ts.append((time.perf_counter() - t0) * 1e6)
                    res["multi_get_frontier_costs_two_contexts_one_gpu"][f"gather_mode_{mode}_n{n}"] = dict(
                        _percentiles(ts), equals_one_context=bool(np.array_equal(fm["weighted_cost"], fc_by_n[n]["weighted_cost"]) and
                                                                  np.array_equal(fm["order"], fc_by_n[n]["order"])))
            m.close()
        # isPoseSafe: ONE pose (the pose of candidate 0 at its best yaw)
        pose = fs.synth.poses_from_yaw(w.goals[:1], np.array([float(rec_all["yaw"][0])]))
        for _ in range(20):
            sc.score_fim(pose, want_fim=False)
        ts = []
        for _ in range(args.latency_calls):
            t0 = time.perf_counter()
            r1 = sc.score_fim(pose, want_fim=False)
            ts.append((time.perf_counter() - t0) * 1e6)
        t0 = time.perf_counter()
        f1 = O.pose_information(table, w.landmarks, pose, 14.0, args.fim_angle, n_threads=1, want_f64=False)
        res["score_fim_1_pose"] = dict(_percentiles(ts), cpu_oracle_1thread_us=(time.perf_counter() - t0) * 1e6,
                                       parity_integers=bool(r1["n_visible"][0] == f1["n_visible"][0]))
        # the same pose asked for info_ref alone — the call FisherInformationManager::isPoseSafe makes (INFO_ONLY worker)
        for _ in range(20):
            sc.score_fim(pose, info_only=True)
        ts = []
        for _ in range(args.latency_calls):
            t0 = time.perf_counter()
            r2 = sc.score_fim(pose, info_only=True)
            ts.append((time.perf_counter() - t0) * 1e6)
        res["score_fim_1_pose_info_only"] = dict(_percentiles(ts), parity_integers=bool(r2["n_voxels"][0] == f1["n_voxels"][0]),
                                                 info_rel_err=float(abs(r2["info_ref"][0] - f1["info_ref"][0]) / max(abs(float(f1["info_ref"][0])), 1e-6)))
        if wl == "REF2D":
            # frontier detection + clustering on the 512^2 costmap, robot on a free cell (fs_frontier_clusters), and the whole
            # FrontierSearch::searchFrom of the oracle (the reference's two nested breadth-first searches) beside it
            y, x = np.argwhere(w.cells[0] == 0)[1234]
            pos = (w.origin[0] + (x + 0.5) * w.resolution, w.origin[1] + (y + 0.5) * w.resolution)
            for _ in range(10):
                sc.frontier_clusters(w.cells.shape[1:], pos)
            for labels in (True, False):
                ts = []
                for _ in range(max(50, args.latency_calls // 3)):
                    t0 = time.perf_counter()
                    _, cl, n_cl, n_cells = sc.frontier_clusters(w.cells.shape[1:], pos, want_labels=labels)
                    ts.append((time.perf_counter() - t0) * 1e6)
                res["frontier_clusters_with_labels" if labels else "frontier_clusters_records_only"] = dict(_percentiles(ts), clusters=int(n_cl), cells=int(n_cells))
            t0 = time.perf_counter()
            fr = O.frontier_search(w.cells, w.origin, w.resolution, pos)
            res["frontier_clusters_with_labels"]["cpu_oracle_searchFrom_us"] = (time.perf_counter() - t0) * 1e6
            res["frontier_clusters_with_labels"]["frontier_records"] = int(len(fr["sizes"]))
        # One whole behaviour-tree tick of a LIVE system: the costmap cycle rewrote a window, the SLAM map changed (the whole cloud
        # is staged again: k-d ordering on the host + transfer), then ProcessFrontierCosts (the cost assignment of 50 frontiers as one
        # call) and EvaluateFisherInformation (isPoseSafe for one pose).  The window and the cloud hold what is already staged, so
        # every other figure of this run is unaffected.
        n = 50
        g, f, b = w.goals[:n], w.frontier_size[:n], w.blacklisted[:n]
        pl = np.hypot(g[:, 0], g[:, 1]) + 0.5
        ph = np.abs(np.arctan2(g[:, 1], g[:, 0]))
        sc.set_arrival_limits(max(mx["max_gt"], 1000.0) * 4, mx["min_gt"])
        side = min(128, w.cells.shape[2], w.cells.shape[1])
        win = w.cells[:, :side, :side]
        parts = {"window": [], "cloud": [], "costs": [], "pose": [], "tick": []}
        for it in range(20 + max(50, args.latency_calls // 5)):
            t0 = time.perf_counter()
            sc.update_grid_region(0, 0, 0, win, view=True)
            t1 = time.perf_counter()
            sc.upload_landmarks(w.landmarks)
            t2 = time.perf_counter()
            sc.get_frontier_costs(g, pl, ph, f, b, with_fim=False)
            t3 = time.perf_counter()
            sc.score_fim(pose, info_only=True)
            t4 = time.perf_counter()
            if it >= 20:
                for k, v in zip(("window", "cloud", "costs", "pose", "tick"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
                    parts[k].append(v * 1e6)
        sc.set_arrival_limits(mx["max_gt"], mx["min_gt"])
        res["live_tick_50_frontiers_1_pose"] = {"tick": _percentiles(parts["tick"]),
                                                "p50_us": {k: float(np.median(v)) for k, v in parts.items()},
                                                "window": [side, side, int(w.cells.shape[0])], "landmarks_restaged": int(w.landmarks.shape[0]),
                                                "note": "fs_update_grid_region + fs_upload_landmarks (whole cloud) + fs_get_frontier_costs (50 frontiers) + fs_score_fim (1 pose, info only), synchronised calls back to back"}
        sc.close()
        out["workloads"][wl] = res
    print(json.dumps(out), flush=True)
    return 0


def other_visibility(args, sc, w, n_local, d_goal, d_fsize, d_black, d_rec, dev, mx, arrival_kw):
    """The same workload at the OTHER visibility volume: the timed steps use --fim-angle (default: the build's 1.0 rad cone),
    this entry the reference's own request — 14 m, cone off (max_angle 4.0, FisherInfoManager.cpp:63-64; the ROS adapter's
    default) — or the other way round.  Two figures: the fused step (full records; YAW_ONLY worker) and ONE fs_score_fim call
    over the step's poses that asks for info_ref alone (the INFO_ONLY worker: what isPoseSafe reads), both with their kernel
    time, landmarks tested per candidate, pass and HBM-tier counts, and gated against the oracle on a sample."""
    import torch
    fs = importlib.import_module("fit-slam_amd")
    shard = importlib.import_module("fit-slam_amd.shard")
    angle = 4.0 if args.fim_angle < np.pi else 1.0
    sc.set_fim_params(14.0, angle)
    out = {"max_dist": 14.0, "max_angle": angle,
           "note": "same workload and build, other visibility volume; never `value`"}
    try:
        def fused():
            sc.score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_rec.data_ptr())
        for _ in range(3):
            fused()
        torch.cuda.synchronize(dev)
        for k in (0, 1, 2, 4):
            sc.kernel_time(k)
        sc.get_counter(0, reset=True)
        mp0, t30 = sc.get_counter(4), sc.get_counter(5)
        reps = max(5, args.steps)
        t0 = time.perf_counter()
        for _ in range(reps):
            fused()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        out["fused_step"] = {"ms_per_step": dt * 1e3, "candidate_goals_per_s": n_local / dt,
                             "m_tested_per_candidate": sc.get_counter(0, reset=True) / (reps * n_local),
                             "multi_pass_candidates_per_step": (sc.get_counter(4) - mp0) / reps,
                             "hbm_table_candidates_per_step": (sc.get_counter(5) - t30) / reps}
        sc.enable_kernel_timing(True)
        for _ in range(reps):
            fused()
        torch.cuda.synchronize(dev)
        ray_ms, ray_n = sc.kernel_time(0)
        fim_ms, fim_n = sc.kernel_time(1)
        ovf_ms, _ = sc.kernel_time(2)
        sc.kernel_time(4)
        out["fused_step"].update({"fs_raymarch_kernel_ms": ray_ms / max(ray_n, 1), "fs_fim_kernel_ms": fim_ms / max(fim_n, 1),
                                  "fs_fim_overflow_kernel_ms": ovf_ms / max(fim_n, 1), "fim_worker": "YAW_ONLY, cone " + ("off" if angle >= np.pi else "1.0 rad")})
        sc.enable_kernel_timing(False)
        rec = shard.records_to_numpy(d_rec[:n_local])
        # the step's poses through fs_score_fim, info_ref alone (host poses in, two columns out: PCIe included in ms_per_call)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O   # the checker of the sample below
        # (the pose of a candidate: goal + the yaw of its best window in DOUBLE, argmax * delta_theta + fov / 2 as
        # CostCalculator.cpp:119 — the record's float32 yaw would turn the camera by 1e-7 rad and move landmarks across voxel faces)
        poses = fs.synth.poses_from_yaw(w.goals[:n_local], rec["argmax"].astype(np.float64) * w.delta_theta + w.camera_fov / 2)
        for _ in range(2):
            r_info = sc.score_fim(poses, info_only=True)
        sc.get_counter(0, reset=True)
        mp0, t30 = sc.get_counter(4), sc.get_counter(5)
        sc.enable_kernel_timing(True)
        sc.kernel_time(1); sc.kernel_time(2)
        t0 = time.perf_counter()
        for _ in range(reps):
            r_info = sc.score_fim(poses, info_only=True)
        dt = (time.perf_counter() - t0) / reps
        fim_ms, fim_n = sc.kernel_time(1)
        ovf_ms, _ = sc.kernel_time(2)
        sc.enable_kernel_timing(False)
        out["info_only_call"] = {"ms_per_call": dt * 1e3, "poses_per_s": n_local / dt, "fs_fim_kernel_ms": fim_ms / max(fim_n, 1),
                                 "fs_fim_overflow_kernel_ms": ovf_ms / max(fim_n, 1),
                                 "m_tested_per_pose": sc.get_counter(0, reset=True) / (reps * n_local),
                                 "multi_pass_poses_per_call": (sc.get_counter(4) - mp0) / reps,
                                 "hbm_table_poses_per_call": (sc.get_counter(5) - t30) / reps,
                                 "learnt_voxels_per_landmark_scanned": sc.get_counter(13) / 256.0,
                                 "fim_worker": "INFO_ONLY (no 6x6 sums, exact table-box cull), cone " + ("off" if angle >= np.pi else "1.0 rad"),
                                 "note": "fs_score_fim(n = all poses of the step) with NULL for every column but info_ref / n_voxels — what FisherInformationManager::isPoseSafe reads; host poses in, PCIe both ways included in ms_per_call"}
        if args.cpu_seconds > 0:
            O.build()
            n_s = min(n_local, 192)
            ok = ((rec["flags"][:n_s] >> 8) & 0xFF) == 0
            fim = O.pose_information(O.Table.generate(), w.landmarks, poses[:n_s], 14.0, angle, n_threads=min(16, os.cpu_count() or 1), want_f64=True)
            sc_ = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
            nv = np.minimum(fim["n_voxels"], 65535)
            # (a pose whose candidate was off the map / blacklisted is still a pose for fs_score_fim: compared on all of them)
            sc_all = np.maximum(np.abs(fim["info_f64"]), 1e-6)
            out["parity"] = {"n": int(n_s),
                             "fused_n_visible_bit_exact": bool(np.array_equal(rec["n_visible"][:n_s][ok], fim["n_visible"][ok])),
                             "fused_n_voxels_bit_exact": bool(np.array_equal(((rec["flags"][:n_s] >> 16) & 0xFFFF)[ok], nv[ok])),
                             "fused_info_max_rel_err": float(np.max(np.abs(rec["info_ref"][:n_s][ok] - fim["info_f64"][ok]) / sc_)) if ok.any() else 0.0,
                             "info_only_n_voxels_bit_exact": bool(np.array_equal(r_info["n_voxels"][:n_s], fim["n_voxels"])),
                             "info_only_info_max_rel_err": float(np.max(np.abs(r_info["info_ref"][:n_s] - fim["info_f64"]) / sc_all))}
            p = out["parity"]
            p["ok"] = bool(p["fused_n_visible_bit_exact"] and p["fused_n_voxels_bit_exact"] and p["info_only_n_voxels_bit_exact"]
                           and p["fused_info_max_rel_err"] <= 1e-4 and p["info_only_info_max_rel_err"] <= 1e-4)
            # the CPU path at THIS visibility volume beside it (1 thread / 16 / every host core), and the full gate — log det
            # included — on the fused records of its sample
            out["cpu_baseline"], out["fused_parity"] = cpu_baseline(w, arrival_kw, n_local, args.cpu_seconds * 0.5, rec, mx, angle)
    finally:
        sc.set_fim_params(14.0, args.fim_angle)
    return out


def dry_run(args, rank, world) -> int:
    """Rendezvous + the one collective of the N > 1 path with dummy records on the CPU (gloo): what a box without
    GPUs can check of the launcher.  Scores nothing and reports no value."""
    import torch
    import torch.distributed as dist
    shard = importlib.import_module("fit-slam_amd.shard")
    n_total = 1000 * world + 7
    lo, hi = shard.shard_bounds(n_total, world, rank)
    cap = shard.shard_capacity(n_total, world)
    local = torch.zeros((cap, 8), dtype=torch.int32)
    local[: hi - lo, 0] = torch.arange(lo, hi, dtype=torch.int32)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        full = shard.gather_records(local, n_total)
        dist.barrier()
    else:
        full = local[:n_total]
    ok = bool(torch.equal(full[:, 0], torch.arange(n_total, dtype=torch.int32)))
    print(f"[bench] dry-run rank {rank}/{world}: gathered {full.shape[0]} records, ordered={ok}", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "value": None, "n_gpus": world, "scaling": args.scaling,
                          "gathered_records": int(full.shape[0]), "order_restored": ok}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 4


def run_rank(args) -> int:
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: refusing to report a number for the wrong rank count", file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    multi_path = world > 1 or args.rehearse_multi                  # the N > 1 data path (a one-rank group under --rehearse-multi)
    if multi_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout belongs to the one JSON line, so
        # file descriptor 1 points at stderr until the group has done its first collective
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        print(f"[bench] rank {rank}/{world} on cuda:{dev_index} ({args.backend})", file=sys.stderr, flush=True)

    fs = importlib.import_module("fit-slam_amd")
    shard = importlib.import_module("fit-slam_amd.shard")

    if args.scaling == "strong":
        # identical total work at every N: configs[3] — C4's 160 k candidates on C3's grid size — split over the ranks
        wl_name = "C4" if args.workload == "C3" else args.workload
        cfg = fs.synth.CONFIGS[wl_name]
        n_total = cfg["n_cand"]
        w = fs.synth.make_workload(wl_name)
    else:
        # every rank builds the same map and cloud (they are replicated) and ONLY its own block of the n x world list:
        # block r of the list is a function of (config, r) (synth.candidate_block; block 0 = the N = 1 list)
        wl_name = args.workload
        cfg = fs.synth.CONFIGS[wl_name]
        n_total = cfg["n_cand"] * world
        w = fs.synth.make_workload(wl_name)
    if args.depth_cells > 0:
        w.max_camera_depth = args.depth_cells * w.resolution
    lo, hi = shard.shard_bounds(n_total, world, rank)
    cap = shard.shard_capacity(n_total, world)
    if args.scaling == "strong":
        blk_goals, blk_fsize, blk_black = w.goals[lo:hi], w.frontier_size[lo:hi], w.blacklisted[lo:hi]
    else:
        assert (lo, hi) == (rank * cfg["n_cand"], (rank + 1) * cfg["n_cand"])
        blk_goals, blk_fsize, blk_black = fs.synth.candidate_block(w, wl_name, rank)

    # Explicit streams for everything: a scorer launches on the stream handle it is given and torch (and RCCL's
    # all-gather) orders its work after the CURRENT stream — so a batch is issued with its scorer's stream current, and
    # that stream is never the default one (handle 0, which the C ABI takes as "create your own").
    # --pipeline P: P scorer contexts, each with its own stream and its own staged copy of the map and the cloud; batch k
    # goes to context k mod P, so the ray-march of batch k+1 fills the CUs the persistent FIM workgroups of batch k leave
    # idle while they drain (measured: tools/overlap_probe.py, 1.024 -> 0.984 ms per C3 batch; 1.031 -> 1.011 here).
    # Every batch is still one complete pass of the hot path; the fences below wait for all streams.  Default P = 1.
    n_pipe = max(1, args.pipeline)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_pipe)]
    assert all(st.cuda_stream != 0 for st in streams)
    torch.cuda.set_stream(streams[0])
    scs = [fs.FrontierScorer(device=dev_index, stream=st.cuda_stream) for st in streams]
    sc = scs[0]
    arrival_kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                      robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    grid_note = "dense upload"
    bricks = fs.synth.dense_to_bricks(w.cells) if wl_name == "C5" else None
    for c in scs:
        for kv in args.option:
            k, v = kv.split("=", 1)
            c.set_option(k, float(v))
        c.set_ray_params(**arrival_kw)
        if bricks is not None:
            # configs[4]'s wire format: the non-unknown 8^3 bricks only; in HBM the grid is expanded to the dense image
            # (1 GiB of 288 GB: no hash probe per cell on the ray walk; + 256 MiB of class image when long rays ask for it).
            # Limit 2^31 cells.
            c.upload_grid_bricks(w.cells.shape, w.origin, w.resolution, bricks[0], bricks[1], default_value=255)
            grid_note = (f"sparse brick-list upload ({bricks[0].shape[0]} of {w.cells.size // 512} bricks), "
                         "dense 1 GiB in HBM")
        else:
            c.upload_grid(w.cells, w.origin, w.resolution)
        c.upload_landmarks(w.landmarks)
        c.lookup_generate()
        c.set_fim_params(14.0, args.fim_angle)
    del bricks
    mx = sc.max_arrival()
    for c in scs[1:]:                      # (the calibration fan ran on context 0: the others rank with the same limits)
        c.set_arrival_limits(mx["max_gt"], mx["min_gt"])

    # candidate columns of this rank's block, resident in HBM
    d_goal = torch.from_numpy(np.ascontiguousarray(blk_goals)).to(dev)
    d_fsize = torch.from_numpy(np.ascontiguousarray(blk_fsize)).to(dev)
    d_black = torch.from_numpy(np.ascontiguousarray(blk_black)).to(dev)
    # two record buffers per context: the all-gather of a batch runs on the communicator's stream while the next batch of
    # the same context is being scored
    n_buf = 2 * n_pipe
    d_recs = [torch.zeros((cap, 8), dtype=torch.int32, device=dev) for _ in range(n_buf)]
    pending = [None] * n_buf
    state = {"k": 0, "full": None}
    n_local = hi - lo

    def step():
        k = state["k"]
        state["k"] += 1
        c, b = k % n_pipe, k % n_buf
        d_rec = d_recs[b]
        with torch.cuda.stream(streams[c]):
            if pending[b] is not None:
                pending[b].wait()                                    # (on streams[c]: the scorer below overwrites the buffer that gather read)
                pending[b] = None
            scs[c].score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_rec.data_ptr())
            if multi_path:
                if args.backend == "nccl":
                    state["full"], pending[b] = shard.gather_records(d_rec, n_total, async_op=True)
                    return state["full"]
                streams[c].synchronize()
                state["full"] = shard.gather_records(d_rec.cpu(), n_total)   # gloo rehearsal: through host memory
                return state["full"]
        state["full"] = d_rec[:n_total]
        return state["full"]

    def fence():
        """barrier + synchronise; returns the moment THIS rank had finished its own work (before it met the others)"""
        for b in range(n_buf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize(dev)                                  # every stream of the device
        t_ready = time.perf_counter()
        if multi_path:
            dist.barrier()
            torch.cuda.synchronize(dev)
        return t_ready

    for _ in range(args.warmup):
        step()
    fence()
    for c in scs:
        c.get_counter(0, reset=True)
    # Per-kernel hipEvent pairs cost ~0.035 ms per step (measured: 1.06 vs 1.024 ms), so only the LAST timed block carries
    # them: the roofline's launch durations come from that block, `value` from the median block (an event-free one
    # whenever --repeats >= 3).  All blocks are listed in "timing".
    n_rep = max(1, args.repeats)
    if args.min_timed_seconds > 0:
        # one untimed calibration block decides the number of blocks — the SAME number on every rank (MAX over the ranks)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        cal = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if (multi_path and args.backend == "nccl") else "cpu")
        if multi_path:
            dist.all_reduce(cal, op=dist.ReduceOp.MAX)
        n_rep = int(min(500, max(n_rep, np.ceil(args.min_timed_seconds / max(float(cal[0]), 1e-6)))))
        for c in scs:
            c.get_counter(0, reset=True)
    block_s, local_s = [], []
    full = None
    for r in range(n_rep):
        if r == n_rep - 1:
            for c in scs:
                c.enable_kernel_timing(True)
                for kind in range(5):
                    c.kernel_time(kind)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full = step()
        t_ready = fence()
        block_s.append(time.perf_counter() - t0)
        local_s.append(t_ready - t0)                                 # this rank alone: its steps + its share of the gathers
    n_steps_all = args.steps * len(block_s)
    n_steps_ev = args.steps                                          # steps of the block that carried the events

    def kernel_sum(kind):
        ms, n = 0.0, 0
        for c in scs:
            a, b = c.kernel_time(kind)
            ms, n = ms + a, n + b
        return ms, n

    ray_ms, ray_n = kernel_sum(0)
    fim_ms, fim_n = kernel_sum(1)
    ovf_ms, ovf_n = kernel_sum(2)
    sort_ms, sort_n = kernel_sum(4)
    # landmark tests per candidate on this rank, over every timed step
    m_tested = sum(c.get_counter(0, reset=True) for c in scs) / max(1, n_steps_all * n_local)
    multipass, tier3 = sum(c.get_counter(4) for c in scs), sum(c.get_counter(5) for c in scs)
    for c in scs:
        c.enable_kernel_timing(False)

    # Two more operating points of the same workload, outside the timed region (N = 1 only), for the record:
    #  * host-buffer path: fs_score_candidates with the candidate columns in host memory and the records back in host memory —
    #    what a caller without device-resident buffers gets; PCIe both ways and one synchronisation per call included.
    #    This rate is never `value`.
    #  * cold cost map: the spatial sort puts blocks of the map whose candidates were expensive in the PREVIOUS call first
    #    (DESIGN.md 4.1); the timed loop scores one list over and over, so its map is perfectly trained.  A call whose
    #    predecessor recorded nothing (option sort.costmap off for one call) sees the empty map of a first call on a new snapshot.
    extra = {}
    if not multi_path and not args.no_parity:
        g_h, f_h, b_h = blk_goals, blk_fsize, blk_black
        for _ in range(3):
            sc.score_candidates(g_h, f_h, b_h)
        t0 = time.perf_counter()
        reps_h = max(5, args.steps)
        for _ in range(reps_h):
            sc.score_candidates(g_h, f_h, b_h)
        dt_h = (time.perf_counter() - t0) / reps_h
        extra["host_buffer_path"] = {"ms_per_call": dt_h * 1e3, "candidate_goals_per_s": n_local / dt_h,
                                     "note": "fs_score_candidates: candidate columns from host memory, records to host memory, one synchronisation per call (PCIe-inclusive; never `value`)"}
        cold = []
        for _ in range(7):
            sc.set_option("sort.costmap", 0)
            sc.score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_recs[0].data_ptr())
            sc.set_option("sort.costmap", 1)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            sc.score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_recs[0].data_ptr())
            torch.cuda.synchronize(dev)
            cold.append((time.perf_counter() - t0) * 1e3)
        warm = []
        for _ in range(7):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            sc.score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_recs[0].data_ptr())
            torch.cuda.synchronize(dev)
            warm.append((time.perf_counter() - t0) * 1e3)
        extra["cold_costmap_step"] = {"ms_cold_median": float(np.median(cold)), "ms_trained_median_same_method": float(np.median(warm)),
                                      "note": "single synchronised calls (launch latency included, unlike the back-to-back timed blocks): first call on an empty cost map against a call on the trained one"}
        # the caller's last step of the same pass: fs_rank_candidates over the records of one step (host columns in,
        # host columns out; path lengths as the reference's planner would supply them — here the straight-line distance)
        recs_h = sc.score_candidates(g_h, f_h, b_h)
        pl = np.hypot(g_h[:, 0] - w.robot_pose[0], g_h[:, 1] - w.robot_pose[1]).astype(np.float64) if hasattr(w, "robot_pose") else np.full(n_local, 5.0)
        ph = np.zeros(n_local, dtype=np.float64)
        for _ in range(2):
            sc.rank_candidates(recs_h, pl, ph, b_h)
        t0 = time.perf_counter()
        for _ in range(5):
            sc.rank_candidates(recs_h, pl, ph, b_h)
        extra["rank_candidates_call"] = {"ms_per_call": (time.perf_counter() - t0) / 5 * 1e3,
                                         "note": "fs_rank_candidates over one step's records: utilities, weighted cost and order, host columns both ways"}

    # ---- the map between two ticks: a costmap update cycle rewrites one window of the master grid (Layer::updateCosts' bounds,
    # DEP/src/nav2_plugins/lethal_marker.cpp:305-325); fs_update_grid_region sends that window alone, fs_upload_grid the whole map.
    # Each followed by one scoring step (the snapshot's re-cuts the whole class image, the window's nothing but its bricks).  The
    # window holds the cells the map already has there, so the timed loop's map — and every figure above — is unchanged.
    if not multi_path and not args.no_parity and wl_name != "C5":
        nz_, ny_, nx_ = w.cells.shape
        sx_, sy_, sz_ = min(64, nx_), min(64, ny_), min(16, nz_)
        x0_, y0_, z0_ = (nx_ - sx_) // 2, (ny_ - sy_) // 2, (nz_ - sz_) // 2
        view = w.cells[z0_:z0_ + sz_, y0_:y0_ + sy_, x0_:x0_ + sx_]

        def timed(fn, reps=5):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                fn()
                sc.score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_recs[0].data_ptr())
                torch.cuda.synchronize(dev)
                ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts))

        def snapshot():
            sc.upload_grid(w.cells, w.origin, w.resolution)
            sc.set_arrival_limits(mx["max_gt"], mx["min_gt"])

        t_none = timed(lambda: None)
        t_win = timed(lambda: sc.update_grid_region(x0_, y0_, z0_, view, view=True))
        t_snap = timed(snapshot)
        extra["map_update_then_step"] = {"step_alone_ms": t_none, "window_then_step_ms": t_win, "snapshot_then_step_ms": t_snap,
                                         "window": [sx_, sy_, sz_], "grid_bytes": int(w.cells.size),
                                         "note": "fs_update_grid_region of one window (from a strided view of the host map) / fs_upload_grid of the whole map, "
                                                 "each followed by one synchronised scoring step; medians of 5; host map in pageable memory"}

    if not multi_path and not args.no_parity:
        extra["reference_request_visibility" if args.fim_angle < np.pi else "build_cone_visibility"] = other_visibility(
            args, sc, w, n_local, d_goal, d_fsize, d_black, torch.zeros_like(d_recs[0]), dev, mx, arrival_kw)

    # ---- ranked step: the scoring step followed, on the same stream and without the records visiting the host, by
    # fs_rank_candidates_dev over the full (gathered) list — "scored / ranked candidates out".  Path lengths are an input of the
    # reference's ranking (its planner supplies them, DEP/src/FrontierCostsManager.cpp:103-117): resident columns here, the
    # straight-line distance from the map's centre and the bearing as heading.  One block of K steps, same bracketing; reported
    # next to `value`, never instead of it.  (Path columns: see below.)
    ranked = None
    if not multi_path or args.backend == "nccl":
        # (a function of the list index alone — a low-discrepancy sequence over [0.5, 30] m and [0, pi] rad — so that no rank
        # needs the other ranks' goal points and the columns are the same at every N)
        idx = np.arange(n_total, dtype=np.float64)
        d_len = torch.from_numpy(0.5 + 29.5 * np.modf(idx * 0.6180339887498949)[0]).to(dev)
        d_head = torch.from_numpy(np.pi * np.modf(idx * 0.7548776662466927)[0]).to(dev)
        # (one set of output columns per pipeline slot: the slots' streams are not ordered against each other)
        d_costs = [torch.zeros(n_total, dtype=torch.float64, device=dev) for _ in range(n_pipe)]
        d_orders = [torch.zeros(n_total, dtype=torch.int32, device=dev) for _ in range(n_pipe)]
        d_rerrs = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(n_pipe)]

        def ranked_step():
            full_ = step()
            k = state["k"] - 1
            c, b = k % n_pipe, k % n_buf
            with torch.cuda.stream(streams[c]):
                if pending[b] is not None:
                    pending[b].wait()                                # the ranking reads the gathered list
                    pending[b] = None
                scs[c].rank_candidates_dev(n_total, full_.data_ptr(), d_len.data_ptr(), d_head.data_ptr(), d_costs[c].data_ptr(),
                                           d_order=d_orders[c].data_ptr(), d_err=d_rerrs[c].data_ptr())

        for _ in range(max(2, args.warmup)):
            ranked_step()
        for c in scs:
            c.enable_kernel_timing(True)
            c.kernel_time(3)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ranked_step()
        fence()
        dt_r = time.perf_counter() - t0
        rank_ms, rank_n = kernel_sum(3)
        for c in scs:
            c.enable_kernel_timing(False)
            for kind in range(5):
                c.kernel_time(kind)
        checks = []
        for c in range(n_pipe):
            order = d_orders[c].cpu().numpy()
            cost = d_costs[c].cpu().numpy()
            checks.append((int(d_rerrs[c].cpu()[0]), bool(np.array_equal(np.sort(order), np.arange(n_total))), bool(np.all(np.diff(cost[order]) >= 0))))
        ranked = {"dt": dt_r, "rank_kernels_ms_per_step": rank_ms / max(rank_n, 1), "range_error": max(x[0] for x in checks),
                  "order_is_a_permutation": all(x[1] for x in checks), "costs_ascending": all(x[2] for x in checks)}

    # ---- strong scaling beside the weak line (N > 1, and the one-rank rehearsal): SURVEY.md 8(e) "Reporting: throughput at
    # G = 1/2/4/8 with identical total work".  configs[3]'s shape — 160 k candidates, 512^3 grid, 100 k landmarks — over the map
    # and cloud ALREADY staged for the weak blocks (nothing is re-staged): the list is blocks 0..7 of synth.candidate_block on
    # this workload, cut into `world` contiguous shares (fs_multi_shard_bounds' rule); a rank draws only the blocks its share
    # touches.  Same step (score + the one all-gather), same bracketing, MAX over the ranks, median of three blocks of K steps.
    strong = None
    if multi_path and args.scaling == "weak" and not args.no_strong:
        per_blk = cfg["n_cand"]
        n_tot_s = 8 * per_blk
        lo_s, hi_s = shard.shard_bounds(n_tot_s, world, rank)
        cap_s = shard.shard_capacity(n_tot_s, world)
        parts = []
        for b in range(lo_s // per_blk, (hi_s - 1) // per_blk + 1):
            g_, f_, k_ = fs.synth.candidate_block(w, wl_name, b)
            a0, a1 = max(lo_s, b * per_blk) - b * per_blk, min(hi_s, (b + 1) * per_blk) - b * per_blk
            parts.append((g_[a0:a1], f_[a0:a1], k_[a0:a1]))
        sg = np.ascontiguousarray(np.concatenate([x[0] for x in parts]))
        sf = np.ascontiguousarray(np.concatenate([x[1] for x in parts]))
        sb = np.ascontiguousarray(np.concatenate([x[2] for x in parts]))
        n_loc_s = hi_s - lo_s
        assert sg.shape[0] == n_loc_s
        ds_goal, ds_fsize, ds_black = torch.from_numpy(sg).to(dev), torch.from_numpy(sf).to(dev), torch.from_numpy(sb).to(dev)
        ds_recs = [torch.zeros((cap_s, 8), dtype=torch.int32, device=dev) for _ in range(2)]
        s_pending = [None, None]
        s_state = {"k": 0, "full": None}

        def strong_step():
            b = s_state["k"] % 2
            s_state["k"] += 1
            with torch.cuda.stream(streams[0]):
                if s_pending[b] is not None:
                    s_pending[b].wait()
                    s_pending[b] = None
                sc.score_candidates_dev(n_loc_s, ds_goal.data_ptr(), ds_fsize.data_ptr(), ds_black.data_ptr(), 0, ds_recs[b].data_ptr())
                if args.backend == "nccl":
                    s_state["full"], s_pending[b] = shard.gather_records(ds_recs[b], n_tot_s, async_op=True)
                else:
                    streams[0].synchronize()
                    s_state["full"] = shard.gather_records(ds_recs[b].cpu(), n_tot_s)

        def strong_fence():
            for b in range(2):
                if s_pending[b] is not None:
                    s_pending[b].wait()
                    s_pending[b] = None
            return fence()

        for _ in range(max(2, min(args.warmup, 3))):
            strong_step()
        strong_fence()
        s_block, s_local = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                strong_step()
            t_ready = strong_fence()
            s_block.append(time.perf_counter() - t0)
            s_local.append(t_ready - t0)
        red_dev = dev if args.backend == "nccl" else "cpu"
        tm = torch.tensor(s_block, dtype=torch.float64, device=red_dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        mine = torch.tensor([float(np.median(s_local))], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(x[0]) / args.steps * 1e3 for x in every]
        dt_s = float(np.median([float(x) for x in tm.tolist()]))
        strong = {"total_candidates": n_tot_s, "candidates_per_gpu": n_loc_s if world == 1 else cap_s, "ms_per_step": dt_s / args.steps * 1e3,
                  "candidate_goals_per_s": n_tot_s * args.steps / dt_s, "blocks_ms_per_step": [float(x) / args.steps * 1e3 for x in tm.tolist()],
                  "per_rank_ms_per_step": per_rank_ms, "per_rank_ms_per_step_min": min(per_rank_ms), "per_rank_ms_per_step_max": max(per_rank_ms),
                  "workload": f"configs[3] shape on the staged {wl_name} map and cloud: 8 x {per_blk} candidates (synth.candidate_block 0..7), "
                              f"contiguous shares of ceil(n / {world}); nothing re-staged; one all-gather of {n_tot_s} x 32 B per step",
                  "scaling": "strong", "note": "total work identical at every N; never `value` (the weak line is). A scaling figure only when the driver has run it on N physical GPUs"}
        # the gathered list is gated like the weak one: 48 candidates of each rank's share against the oracle (rank 0)
        if rank == 0 and not args.no_parity:
            rec_s = shard.records_to_numpy(s_state["full"][:n_tot_s])
            gs, fz, bl, ix = [], [], [], []
            blk_cache = {}
            for r in range(world):
                b_lo, b_hi = shard.shard_bounds(n_tot_s, world, r)
                pick = np.unique(np.linspace(b_lo, max(b_hi - 1, b_lo), num=min(48, b_hi - b_lo)).astype(np.int64))
                for i in pick:
                    if int(i) // per_blk not in blk_cache:
                        blk_cache[int(i) // per_blk] = fs.synth.candidate_block(w, wl_name, int(i) // per_blk)
                    g_, f_, k_ = blk_cache[int(i) // per_blk]
                    gs.append(g_[i % per_blk]); fz.append(f_[i % per_blk]); bl.append(k_[i % per_blk]); ix.append(int(i))
            strong["parity"] = sample_parity(w, arrival_kw, mx, np.asarray(gs), np.asarray(fz, dtype=np.int32), np.asarray(bl, dtype=np.uint8),
                                             rec_s[np.asarray(ix)], args.fim_angle, f"48 candidates evenly spaced in each of the {world} shares of the 160 k list")
        del ds_goal, ds_fsize, ds_black, ds_recs

    # ---- what a sub-linear curve would be made of (N > 1): every rank's own time per step, the spread with which the ranks
    # reach the closing barrier, and the all-gather by itself (blocking calls between an event pair on the scorer's stream)
    multi = None
    if multi_path:
        red_dev = dev if args.backend == "nccl" else "cpu"
        tmax = torch.tensor(block_s + ([ranked["dt"]] if ranked else []), dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in tmax.tolist()]
        block_s = vals[:len(block_s)]
        if ranked:
            ranked["dt"] = vals[-1]
        mine = torch.tensor(local_s, dtype=torch.float64, device=red_dev)
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        loc = np.array([[float(x) for x in t.tolist()] for t in everyone])        # [rank][block], seconds
        gather_ms = None
        if args.backend == "nccl":
            with torch.cuda.stream(streams[0]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(5):
                    shard.gather_records(d_recs[0], n_total)
                e0.record()
                for _ in range(20):
                    shard.gather_records(d_recs[0], n_total)
                e1.record()
            torch.cuda.synchronize(dev)
            g = torch.tensor([e0.elapsed_time(e1) / 20], dtype=torch.float64, device=dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather_ms = float(g[0])
        multi = {"per_rank_ms_per_step": [float(np.median(loc[r]) / args.steps * 1e3) for r in range(world)],
                 "per_rank_ms_per_step_min": float(np.median(loc, axis=1).min() / args.steps * 1e3),
                 "per_rank_ms_per_step_max": float(np.median(loc, axis=1).max() / args.steps * 1e3),
                 "barrier_skew_ms": float(np.median(loc.max(axis=0) - loc.min(axis=0)) * 1e3),
                 "barrier_skew_note": "per timed block: last rank ready minus first rank ready (each after its own K steps and gathers), median over the blocks",
                 "all_gather_ms": gather_ms,
                 "all_gather_note": f"one blocking all_gather_into_tensor of {n_total} x 32 B between an event pair on the scorer's stream, mean of 20, max over ranks (nccl backend only)"}

    exit_code = 0
    if rank == 0:
        dt = float(np.median(block_s))
        value = n_total * args.steps / dt
        R, E = w.n_yaw if w.n_yaw else sc.n_yaw, len(w.elev)
        L = int(w.max_camera_depth / w.resolution)
        r_fp = int(np.ceil(w.robot_radius / w.resolution))
        m = w.landmarks.shape[0]
        # SURVEY.md §8(d), strictly: B_cand = R*E*(L+1) + (2r+1)^2 + M_tested*12 + 32.  The dominant kernel (FIM accumulate)
        # owns the M_tested*12 + 32 part; M_tested = landmarks whose visibility predicate was evaluated (device counter).
        # The 16-B chunk spheres the culling scans are index bytes §8(d) does not list: reported separately, not counted.
        b_ray = R * E * (L + 1) + (2 * r_fp + 1) ** 2
        n_chunks = -(-m // 64)
        b_fim = m_tested * 12 + 32
        launches_per_step = fim_n / max(1, n_steps_ev)
        fim_avg_s = (fim_ms / max(fim_n, 1)) * 1e-3
        achieved = (n_local * b_fim / max(launches_per_step, 1e-9)) / fim_avg_s / 1e9 if fim_n else None
        prof = counter_profile(wl_name, args.depth_cells, args.fim_angle)
        gpu_rec = shard.records_to_numpy(full) if not args.no_parity else None
        cpu, parity = (None, None)
        if not multi_path and args.cpu_seconds > 0:
            cpu, parity = cpu_baseline(w, arrival_kw, n_total, args.cpu_seconds, gpu_rec, mx, args.fim_angle)
        elif multi_path and gpu_rec is not None:
            # no CPU baseline at N > 1 (rank 0, N = 1 only), but the gathered list is still gated: 96 candidates out of every
            # rank's block against the oracle
            per_blk = 96
            gs, fz, bl, ix = [], [], [], []
            for r in range(world):
                b_lo, b_hi = shard.shard_bounds(n_total, world, r)
                pick = np.unique(np.linspace(0, max(b_hi - b_lo - 1, 0), num=min(per_blk, b_hi - b_lo)).astype(np.int64))
                if args.scaling == "strong":
                    g_, f_, k_ = w.goals[b_lo:b_hi], w.frontier_size[b_lo:b_hi], w.blacklisted[b_lo:b_hi]
                else:
                    g_, f_, k_ = (blk_goals, blk_fsize, blk_black) if r == rank else fs.synth.candidate_block(w, wl_name, r)
                gs.append(g_[pick]); fz.append(f_[pick]); bl.append(k_[pick]); ix.append(b_lo + pick)
            ix = np.concatenate(ix)
            parity = sample_parity(w, arrival_kw, mx, np.concatenate(gs), np.concatenate(fz), np.concatenate(bl), gpu_rec[ix],
                                   args.fim_angle, f"{per_blk} candidates evenly spaced in each of the {world} rank blocks")
        per_rank = n_total // world if args.scaling == "strong" else cfg["n_cand"]
        roofline = {
            # the roofline the fraction below is priced against: the HBM one the metric asks for (algorithmic bytes / time / 8 TB/s).
            # What the counters say actually limits the dominant kernel is `limited_by` (+ `bound_evidence`): it reads its cloud
            # from L2 and is bound by instruction issue at four waves per SIMD, not by HBM.
            "bound": "hbm",
            "limited_by": (prof or {}).get("bound"),
            "bound_evidence": (prof or {}).get("bound_evidence", "no counter profile of these kernel sources is committed: nominal HBM roofline per SURVEY.md 8(d)"),
            "kernel": "fs_fim_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": (prof or {}).get("fs_fim_kernel_hbm_bytes_per_launch"),
            "algorithmic_bytes_per_candidate": b_fim, "m_tested_per_candidate": m_tested,
            "index_bytes_per_candidate_not_counted": n_chunks * 16,
            "bruteforce_bytes_per_candidate": m * 12 + 32,
            "candidates_per_launch": n_local / max(launches_per_step, 1e-9), "launches_per_step": launches_per_step,
            "avg_launch_ms": fim_ms / max(fim_n, 1), "launches": fim_n,
            "multi_pass_candidates": multipass, "hbm_table_candidates": tier3,
            "whole_step_frac": (n_local * (b_fim + b_ray) / (dt / args.steps)) / 1e9 / HBM_PEAK_GBS,
            "counters": ({k: prof[k] for k in ("source_hash", "valu_issue_utilisation", "valu_peak_wave_insts_per_cycle_per_simd",
                                                "fs_fim_kernel", "collected_with") if k in prof} if prof else None),
            "note": "achieved = candidates per launch x (M_tested*12 + 32) B / average hipEvent launch time of fs_fim_kernel, "
                    "measured live on the scorer's stream; the 1.2 MB cloud is served from L2 after the first touch, so "
                    "algorithmic GB/s is not HBM traffic (traffic = PMC bytes per launch when a matching profile exists)",
        }
        line = {
            "metric": METRIC,
            "value": value, "unit": "candidate-goals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8 grid walk (int32/fp64 set-up) + f32 FIM", "data": "synthetic",
            "config": {"workload": f"{wl_name}: {cfg['n']}^{3 if cfg['nz'] > 1 else 2} uint8 grid ({grid_note}), {per_rank} candidates/GPU "
                                   f"({n_total} total), {m} landmarks, {R * E} rays/candidate ({R} yaw x {E} elevation), "
                                   f"L={L} cells, visibility 14 m / {args.fim_angle} rad{' (cone off: the reference request)' if args.fim_angle >= np.pi else ''}, "
                                   f"chunk-culled (M_tested measured), reference 71x100x100 lookup table",
                       "candidates_per_gpu": per_rank, "total_candidates": n_total,
                       "sharding": f"contiguous blocks over {world} GPU(s), one all-gather of 32-B records ({args.backend})",
                       "pipeline": f"{n_pipe} scorer context(s)/HIP stream(s) per GPU; batch k runs whole on context k mod {n_pipe}"},
            "timing": {"repeats": len(block_s), "timed_seconds_total": float(sum(block_s)),
                       "block_ms_per_step": [round(b / args.steps * 1e3, 5) for b in block_s],
                       "block_ms_per_step_p10_p90": [float(np.percentile(block_s, 10) / args.steps * 1e3), float(np.percentile(block_s, 90) / args.steps * 1e3)],
                       "median_ms_per_step": dt / args.steps * 1e3, "min_ms_per_step": min(block_s) / args.steps * 1e3,
                       "value_from": "median block", "kernel_events_in_block": len(block_s) - 1, "value_at_min": n_total * args.steps / min(block_s)},
            "roofline": roofline,
            "kernels_ms_per_step": {"fs_raymarch_kernel": ray_ms / n_steps_ev, "fs_fim_kernel": fim_ms / n_steps_ev,
                                    "fs_fim_overflow_kernel": ovf_ms / n_steps_ev, "candidate_sort": sort_ms / n_steps_ev,
                                    "raymarch_algorithmic_GBps": (n_local * b_ray) / ((ray_ms / n_steps_ev) * 1e-3) / 1e9 if ray_n else None,
                                    "from": "the last timed block (the only one with per-kernel hipEvents); with --pipeline > 1 "
                                            "a kernel's duration includes the time it shares the chip with the other stream's kernels"},
            "cpu_baseline": cpu, "parity": parity,
            "ranked_step": ({"ms_per_step": ranked["dt"] / args.steps * 1e3, "candidate_goals_per_s": n_total * args.steps / ranked["dt"],
                             "rank_kernels_ms_per_step": ranked["rank_kernels_ms_per_step"], "range_error": ranked["range_error"],
                             "order_is_a_permutation": ranked["order_is_a_permutation"], "costs_ascending": ranked["costs_ascending"],
                             "note": "score (+ all-gather) + fs_rank_candidates_dev over the full list on the same stream, path columns resident; "
                                     "one block of K steps; never `value`"} if ranked else None),
            "multi_gpu": multi, "strong_scaling": strong, "rehearse_multi": bool(args.rehearse_multi) or None,
            "other_operating_points": extra or None,
        }
        print(json.dumps(line), flush=True)
        # a value next to a red gate must not look like success to whoever runs this: the line is printed (the figures are
        # evidence either way), the exit code says the gate failed
        red = red_gates(line)
        if red:
            print(f"[bench] PARITY GATE RED: {', '.join(red)} — exit code 5", file=sys.stderr, flush=True)
            exit_code = 5
    if multi_path:
        code = torch.tensor([exit_code], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(code, op=dist.ReduceOp.MAX)                 # every rank leaves with rank 0's verdict
        exit_code = int(code[0])
        dist.barrier()
        dist.destroy_process_group()
    for c in scs:
        c.close()
    return exit_code


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.latency:
        return run_latency(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
