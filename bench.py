#!/usr/bin/env python3
"""bench.py — candidate-goals scored per second (ray-cast + FIM) on a 512^3 grid (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (fs_score_candidates_dev: ray-march -> FIM accumulate -> record
pack) over this rank's block of candidates, followed — when N > 1 — by the single RCCL all-gather
of the 32-byte records.  Workload at N = 1: BASELINE.json configs[2] ("C3": 512^3 grid, 20 k
candidates, 100 k landmarks, 256 rays/candidate = 64 yaw x 4 elevation rings, L = 40 cells).  For
N > 1 every rank scores 20 k candidates of a 20 k*N list (configs[3] at N = 8): weak scaling.
Grid, landmarks, lookup table and the candidate arrays are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", help="C3 (headline) | C2 | C1 (smaller, for rehearsal)")
    ap.add_argument("--depth-cells", type=int, default=0, help="ray length L in cells (0: the config's 40 = 2 m; BASELINE.md's secondary throughput run uses 160)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time; 0 disables")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI; the measured configuration) | gloo (rehearsal of the N > 1 path on one GPU: ranks share the device, records are gathered through host memory)")
    return ap.parse_args()


def cpu_baseline(w, arrival_kw, n_total, target_s, gpu_rec, mx):
    """Time the oracle (kind 'port') on a bounded sample of the same workload, single thread, and
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
        fim = O.pose_information(table, w.landmarks, poses, 14.0, 1.0, n_threads=threads, want_f64=True)
        return time.perf_counter() - t0, arr, fim

    probe = min(32, n_total)
    dt, _, _ = run(0, probe, 1, True)
    n_s = int(max(probe, min(n_total, target_s / max(dt / probe, 1e-9))))
    dt, arr, fim = run(0, n_s, 1, True)
    out = {"value": n_s / dt, "unit": "candidate-goals/s", "cores": 1, "kind": "port",
           "sample": f"first {n_s} of {n_total} candidates of the same workload, oracle/ C restatement "
                     f"(reference-faithful control flow), single thread, {dt:.1f} s"}
    n_c = max(probe, n_s // 2)
    dt_c, _, _ = run(0, n_c, 1, False)
    out["clean_value"] = n_c / dt_c          # BASELINE.md `cpu_clean`: same results, no per-ray vectors / O(k^2) dedupe, one thread
    cores = min(16, os.cpu_count() or 1)     # the GPU box's CPU share for one GPU
    n_o = min(n_total, n_s * min(cores, 16))
    dt_o, _, _ = run(0, n_o, cores, False)
    out["omp_value"] = n_o / dt_o
    out["omp_cores"] = cores
    parity = None
    if gpu_rec is not None:
        r = gpu_rec[:n_s]
        ok = arr["status"] == 0
        ints = (np.array_equal(r["arrival"], arr["arrival"]) and np.array_equal(r["argmax"], arr["argmax"]) and
                np.array_equal((r["flags"] >> 8) & 0xFF, arr["status"]) and
                np.array_equal(r["flags"] & 1, arr["achievable"]) and
                np.array_equal(r["n_visible"][ok], fim["n_visible"][ok]))
        sc = np.maximum(np.abs(fim["info_f64"][ok]), 1e-6)
        e_info = float(np.max(np.abs(r["info_ref"][ok] - fim["info_f64"][ok]) / sc)) if ok.any() else 0.0
        e_tr = float(np.max(np.abs(r["trace"][ok] - fim["trace"][ok]) / np.maximum(fim["trace"][ok], 1e-6))) if ok.any() else 0.0
        parity = {"n": int(n_s), "integers_bit_exact": bool(ints), "info_max_rel_err": e_info,
                  "trace_max_rel_err": e_tr, "ok": bool(ints and e_info <= 1e-4 and e_tr <= 1e-4)}
    return out, parity


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    fs = importlib.import_module("fit-slam_amd")
    from importlib import import_module
    shard = import_module("fit-slam_amd.shard")

    cfg = fs.synth.CONFIGS[args.workload]
    per_rank = cfg["n_cand"]
    n_total = per_rank * world
    w = fs.synth.make_workload(args.workload, n_cand=n_total)
    if args.depth_cells > 0:
        w.max_camera_depth = args.depth_cells * w.resolution
    lo, hi = shard.shard_bounds(n_total, world, rank)
    cap = shard.shard_capacity(n_total, world)

    # One explicit stream for everything: the scorer launches on the stream handle it is given and torch (and RCCL's
    # all-gather) orders its work after the CURRENT stream — so the current stream must be that same, non-default one
    # (the default stream's handle is 0, which the C ABI takes as "create your own").
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    sc = fs.FrontierScorer(device=dev_index, stream=stream.cuda_stream)
    arrival_kw = dict(max_camera_depth=w.max_camera_depth, delta_theta=w.delta_theta, camera_fov=w.camera_fov,
                      robot_radius=w.robot_radius, n_rays=w.n_yaw, elev=w.elev, polygon=w.polygon)
    sc.set_ray_params(**arrival_kw)
    sc.upload_grid(w.cells, w.origin, w.resolution)
    sc.upload_landmarks(w.landmarks)
    sc.lookup_generate()
    sc.set_fim_params(14.0, 1.0)
    mx = sc.max_arrival()

    # candidate columns of this rank's block, resident in HBM
    d_goal = torch.from_numpy(w.goals[lo:hi].copy()).to(dev)
    d_fsize = torch.from_numpy(w.frontier_size[lo:hi].copy()).to(dev)
    d_black = torch.from_numpy(w.blacklisted[lo:hi].copy()).to(dev)
    # two record buffers: the all-gather of batch k runs on the communicator's stream while batch k+1 is being scored
    d_recs = [torch.zeros((cap, 8), dtype=torch.int32, device=dev) for _ in range(2)]
    pending = [None, None]
    state = {"k": 0, "full": None}
    n_local = hi - lo

    def step():
        b = state["k"] & 1
        state["k"] += 1
        if pending[b] is not None:
            pending[b].wait()                                        # the gather that last read this buffer is done
            pending[b] = None
        d_rec = d_recs[b]
        sc.score_candidates_dev(n_local, d_goal.data_ptr(), d_fsize.data_ptr(), d_black.data_ptr(), 0, d_rec.data_ptr())
        if world > 1:
            if args.backend == "nccl":
                state["full"], pending[b] = shard.gather_records(d_rec, n_total, async_op=True)
                return state["full"]
            torch.cuda.synchronize(dev)
            state["full"] = shard.gather_records(d_rec.cpu(), n_total)   # gloo rehearsal: through host memory
            return state["full"]
        state["full"] = d_rec[:n_total]
        return state["full"]

    def fence():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    sc.enable_kernel_timing(True)
    sc.kernel_time(0); sc.kernel_time(1); sc.kernel_time(2)
    sc.get_counter(0, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    fence()
    dt = time.perf_counter() - t0
    ray_ms, ray_n = sc.kernel_time(0)
    fim_ms, fim_n = sc.kernel_time(1)
    ovf_ms, ovf_n = sc.kernel_time(2)
    m_tested = sc.get_counter(0, reset=True) / max(1, args.steps * n_local)   # landmark tests per candidate on this rank
    tier2, tier3 = sc.get_counter(4), sc.get_counter(5)
    sc.enable_kernel_timing(False)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        value = n_total * args.steps / dt
        R, E = w.n_yaw, len(w.elev)
        L = int(w.max_camera_depth / w.resolution)
        r_fp = int(np.ceil(w.robot_radius / w.resolution))
        m = w.landmarks.shape[0]
        # B_cand = R*E*(L+1) + (2r+1)^2 + M_tested*12 + O_out.  With the chunk index M_tested is the
        # measured number of landmarks whose predicate was evaluated (device counter), plus the 16-B chunk spheres scanned.
        b_ray = R * E * (L + 1) + (2 * r_fp + 1) ** 2
        n_chunks = -(-m // 64)
        b_fim = m_tested * 12 + n_chunks * 16 + 32
        launches_per_step = fim_n / max(1, args.steps)
        fim_avg_s = (fim_ms / max(fim_n, 1)) * 1e-3
        achieved = (n_local * b_fim / max(launches_per_step, 1e-9)) / fim_avg_s / 1e9 if fim_n else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and args.depth_cells in (0, 40):
                    traffic = tj.get("fs_fim_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        valu_util = None
        vpath = os.path.join(ROOT, "profiles", "pmc_valu.json")
        if os.path.exists(vpath):
            try:
                vj = json.load(open(vpath))
                if vj.get("workload") == args.workload and args.depth_cells in (0, 40):
                    valu_util = vj["kernels"]["fs_fim_kernel"]["valu_utilisation"]
            except Exception:
                valu_util = None
        gpu_rec = shard.records_to_numpy(full) if not args.no_parity else None
        cpu, parity = (None, None)
        if world == 1 and args.cpu_seconds > 0:
            cpu, parity = cpu_baseline(w, arrival_kw, n_total, args.cpu_seconds, gpu_rec, mx)
        line = {
            "metric": "candidate-goals scored/sec (raycast+FIM) on 512^3 grid; % HBM roofline",
            "value": value, "unit": "candidate-goals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 grid walk (int32/fp64 set-up) + f32 FIM", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['n']}^3 uint8 grid, {per_rank} candidates/GPU "
                                   f"({n_total} total), {m} landmarks, {R * E} rays/candidate ({R} yaw x {E} elevation), "
                                   f"L={L} cells, chunk-culled visibility (M_tested measured), reference 71x100x100 lookup table",
                       "candidates_per_gpu": per_rank, "sharding": f"contiguous blocks over {world} GPU(s), one all-gather of 32-B records ({args.backend})"},
            "roofline": {"bound": "hbm", "kernel": "fs_fim_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_candidate": b_fim, "m_tested_per_candidate": m_tested,
                         "bruteforce_bytes_per_candidate": m * 12 + 32, "candidates_per_launch": n_local / max(launches_per_step, 1e-9),
                         "launches_per_step": launches_per_step,
                         "hash_tier2_candidates": tier2, "hash_tier3_candidates": tier3,
                         "avg_launch_ms": fim_ms / max(fim_n, 1), "launches": fim_n,
                         "valu_utilisation": valu_util,
                         "note": "algorithmic bytes (SURVEY 8(d), M_tested measured) / hipEvent time; the landmark cloud is "
                                 "served from L2 after the first touch, so algorithmic GB/s is not HBM traffic (see traffic); the kernel "
                                 "is bound by vector-ALU issue: valu_utilisation = SQ_ACTIVE_INST_VALU x 4 / SIMD cycles of the launch "
                                 "(profiles/pmc_valu.json)"},
            "kernels_ms_per_step": {"fs_raymarch_kernel": ray_ms / args.steps, "fs_fim_kernel": fim_ms / args.steps,
                                    "fs_fim_overflow_kernel": ovf_ms / args.steps,
                                    "raymarch_algorithmic_GBps": (n_local * b_ray) / ((ray_ms / args.steps) * 1e-3) / 1e9 if ray_n else None},
            "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sc.close()


if __name__ == "__main__":
    main()
