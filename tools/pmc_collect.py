#!/usr/bin/env python3
"""Collects the counter evidence behind bench.py's `roofline` object on the GPU box and writes profiles-ready summaries.

    python3 tools/pmc_collect.py [--out gpurun_out/pmc_r02] [--workload C3] [--skip-calib]

This driver never touches the GPU itself (no torch, no HIP): every measurement is a child `rocprofv3 ... -- python3 bench.py`
(or `-- tools/valu_calib`), one PMC pass per counter set as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes (8 SQ
slots, FETCH_SIZE and WRITE_SIZE in passes of their own), never combined with a trace domain other than --kernel-trace.
Outputs (under --out):
    kernel_stats.csv            rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0`
    pmc_<pass>.csv              per-dispatch counter rows of our kernels
    calib_stdout.jsonl          tools/valu_calib's own s_memtime measurement
    pmc_calib_<pass>.csv        the calibration kernels through the same counters
    pmc_summary.json            what bench.py replays (keyed by the kernel sources' hash): copy to profiles/pmc_summary.json
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SQ_PASSES = {
    "insts": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_FLAT"],
    "cycles": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"],
    "stalls": ["SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_THREAD_CYCLES_VALU", "SQ_INST_CYCLES_VMEM"],
    "grbm": ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
    # in-flight instruction levels: LEVEL / INSTS = the average latency of a vector-memory / LDS instruction in cycles
    "levels": ["SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM", "SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_ATOMIC_RETURN", "SQ_INSTS_LDS_ATOMIC", "SQ_WAVE_CYCLES"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum"],
}
CALIB_PASSES = {
    "insts": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    "cycles": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"],
    "grbm": ["GRBM_GUI_ACTIVE"],
}


def source_hash() -> str:
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "fit-slam_amd", "csrc")
    for f in sorted(x for x in os.listdir(csrc) if x.endswith((".hip", ".h"))):
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "fitslam_frontier.h"), "rb").read())
    return h.hexdigest()[:16]


def available_counters(out: str) -> set:
    path = os.path.join(out, "counters_available.txt")
    if not os.path.exists(path):
        r = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=300, cwd="/tmp")
        open(path, "w").write(r.stdout + r.stderr)
    text = open(path).read()
    return set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", text))


def run(cmd, log, timeout):
    t0 = time.time()
    print(f"[pmc] {' '.join(cmd)}", flush=True)
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        try:
            r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=timeout, cwd="/tmp", env=env)
            rc = r.returncode
        except subprocess.TimeoutExpired:
            print(f"[pmc] TIMEOUT after {timeout} s: stopping (no further GPU step after a hang)", flush=True)
            raise SystemExit(5)
    print(f"[pmc]   -> rc {rc} in {time.time() - t0:.0f} s", flush=True)
    return rc


def kernel_key(name: str) -> str:
    m = re.search(r"(fs_\w+|valu_stream)(<\d+)?", name)
    return m.group(0) if m else name[:60]


def collect_rows(d: str, dst_csv: str, keep=("fs_", "valu_stream")):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in keep)]
    if rows:
        fields = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count",
                  "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
        with open(dst_csv, "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=fields, extrasaction="ignore")
            w.writeheader()
            w.writerows(rows)
    return rows


def averages(rows, last=6):
    """kernel -> counter -> mean over the last `last` dispatches (the timed steps; earlier ones include warm-up)."""
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        d[kernel_key(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                                                   int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, cs in d.items():
        out[k] = {}
        for c, vals in cs.items():
            vals = sorted(vals)[-last:]
            out[k][c] = sum(v[1] for v in vals) / len(vals)
            out[k]["_duration_ns"] = sum(v[2] for v in vals) / len(vals)
            out[k]["_dispatches"] = len(cs[c])
    return out


def derive(summary, per_kernel, calib):
    """The figures bench.py replays, from the per-kernel counter means and the VALU calibration (pure arithmetic: also run by
    --rederive on a summary collected with --skip-calib, with the calibration of another collection of the same box)."""
    # ---- derived figures
    summary["kernels"] = per_kernel
    summary["calibration"] = calib
    # peak VALU issue rate of one SIMD, wave-instructions per cycle: from the calibration kernel's own s_memtime (W >= 2)
    # A SIMD arbitrates oldest-first, so the waves of one SIMD finish at very different times; what the SIMD sustains is
    # set by the LAST wave: W waves x insts / slowest wave's cycles.
    peak = peak4 = None
    if calib.get("s_memtime"):
        per_w = {c["waves_per_simd"]: c["slowest_wave_cycles"] / (c["wave_insts"] * c["waves_per_simd"]) for c in calib["s_memtime"]}
        peak = 1.0 / min(per_w.values())
        peak4 = 1.0 / per_w[4] if 4 in per_w else None
        summary["valu_peak_wave_insts_per_cycle_per_simd"] = peak
        summary["valu_peak_at_4_waves_per_simd"] = peak4
        summary["valu_simd_cycles_per_wave_inst"] = {str(w): v for w, v in sorted(per_w.items())}
        summary["in_kernel_clock_ghz"] = {str(c["waves_per_simd"]): c.get("in_kernel_clock_ghz") for c in calib["s_memtime"]}
    fim = next((v for k, v in per_kernel.items() if k.startswith("fs_fim_kernel")), None)
    if fim:
        n_simd = 256 * 4
        cyc = fim.get("GRBM_GUI_ACTIVE", 0) / 8.0                      # rocprofv3 sums the 8 XCDs
        res = {"duration_us": fim.get("_duration_ns", 0) / 1e3, "gpu_cycles": cyc}
        if cyc and "SQ_INSTS_VALU" in fim:
            res["valu_wave_insts_per_cycle_per_simd"] = fim["SQ_INSTS_VALU"] / (n_simd * cyc)
            if peak:
                res["valu_issue_utilisation"] = res["valu_wave_insts_per_cycle_per_simd"] / peak
                summary["valu_issue_utilisation"] = res["valu_issue_utilisation"]
            if peak4:
                res["valu_issue_utilisation_vs_4_wave_peak"] = res["valu_wave_insts_per_cycle_per_simd"] / peak4
        if cyc and "SQ_INSTS_SALU" in fim:
            res["salu_insts_per_cycle_per_cu"] = fim["SQ_INSTS_SALU"] / (256 * cyc)
            # every instruction class together: what a SIMD's four waves issue per cycle (each wave issues one at a time)
            res["all_insts_per_cycle_per_simd"] = sum(fim.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                                                                "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")) / (n_simd * cyc)
        wc = fim.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_SALU"):
                if c in fim:
                    res[f"{c}_share_of_wave_cycles"] = fim[c] / wc
        if "FETCH_SIZE" in fim or "WRITE_SIZE" in fim:
            f_kib, w_kib = fim.get("FETCH_SIZE", 0.0), fim.get("WRITE_SIZE", 0.0)
            # gfx950: FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read (MI355X_MICROARCH.md, HBM): doubled
            summary["fs_fim_kernel_hbm_bytes_per_launch"] = int((2 * f_kib + w_kib) * 1024)
            summary["fs_fim_kernel_hbm_bytes_per_launch_uncorrected"] = int((f_kib + w_kib) * 1024)
        if "TCC_HIT_sum" in fim:
            res["l2_hit_rate"] = fim["TCC_HIT_sum"] / max(1.0, fim["TCC_HIT_sum"] + fim.get("TCC_MISS_sum", 0.0))
        summary["fs_fim_kernel"] = res
        u = res.get("valu_issue_utilisation")
        w_any, w_inst = res.get("SQ_WAIT_ANY_share_of_wave_cycles"), res.get("SQ_WAIT_INST_ANY_share_of_wave_cycles")
        hbm = summary.get("fs_fim_kernel_hbm_bytes_per_launch")
        dur = res.get("duration_us")
        if u is not None and w_any is not None:
            # (the label only names what the figures below show; DESIGN.md 4.2 has the argument: at four waves per SIMD the
            # kernel's time follows its instruction count by kind — the waits a wave sees are covered by the CU's other workgroup)
            u4 = res.get("valu_issue_utilisation_vs_4_wave_peak")
            if u >= 0.75:
                summary["bound"] = "valu-issue"
            elif u4 is not None and u4 >= 0.5:
                summary["bound"] = "instruction issue at 4 waves per SIMD"
            elif w_any + (w_inst or 0.0) >= 0.5:
                summary["bound"] = "latency"      # waves parked on LDS-atomic / barrier / load waits
            else:
                summary["bound"] = "mixed"
            summary["bound_evidence"] = (
                f"fs_fim_kernel: VALU issue {u:.0%} of the SIMD peak measured by tools/valu_calib "
                f"({summary['valu_peak_wave_insts_per_cycle_per_simd']:.3f} wave-insts/cycle/SIMD)"
                + (f", {u4:.0%} of what four waves of independent v_fma_f32 reach" if u4 is not None else "")
                + f", waves parked {w_any:.0%} "
                f"(SQ_WAIT_ANY) + issue-stalled {(w_inst or 0.0):.0%} (SQ_WAIT_INST_ANY) of their cycles"
                + (f", L2 hit rate {res['l2_hit_rate']:.1%}" if "l2_hit_rate" in res else "")
                + (f", HBM traffic {hbm / 1e6:.1f} MB per launch = {hbm / (dur * 1e-6) / 1e9:.0f} GB/s of 8000" if hbm and dur else "")
                + ": not HBM-bound")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_r03"))
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--depth-cells", type=int, default=0)
    ap.add_argument("--skip-calib", action="store_true")
    ap.add_argument("--skip-stats", action="store_true")
    ap.add_argument("--passes", default=",".join(SQ_PASSES))
    ap.add_argument("--rederive", default="", help="no GPU work: recompute the derived figures of <dir>/pmc_summary.json (collected with --skip-calib) with the calibration of --calib-from")
    ap.add_argument("--calib-from", default="", help="a pmc_summary.json whose `calibration` block to use with --rederive (same box, same session)")
    ap.add_argument("--bench-arg", action="append", default=[], help="extra argument for bench.py (repeatable), e.g. --bench-arg=--option --bench-arg=ray.layout=2")
    args = ap.parse_args()
    if args.rederive:
        path = os.path.join(args.rederive, "pmc_summary.json")
        summary = json.load(open(path))
        calib = json.load(open(args.calib_from))["calibration"]
        summary["calibration"] = calib
        summary["calibration_from"] = "the collection " + os.path.basename(os.path.dirname(os.path.abspath(args.calib_from))) + " of the same gpurun session (this one ran with --skip-calib)"
        derive(summary, summary["kernels"], calib)
        json.dump(summary, open(path, "w"), indent=1, sort_keys=True, default=dict)
        print(json.dumps({k: summary.get(k) for k in ("source_hash", "fim_angle", "bound", "valu_issue_utilisation", "bound_evidence")}, indent=1))
        return
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    have = available_counters(out)
    # The profiled program is the interpreter itself, by its real path: rocprofv3's preloaded library initialises the GPU before
    # the program starts, so a PATH shim or wrapper (pyenv ...) would be an exec hop behind an initialised GPU.  For the same
    # reason bench.py must stay a single process here: --gpus N > 1 would make it a launcher that spawns ranks under the profiler.
    bad = [a for i, a in enumerate(args.bench_arg) if a.startswith("--gpus") and (a not in ("--gpus", "--gpus=1") or (a == "--gpus" and args.bench_arg[i + 1:i + 2] != ["1"]))]
    if bad:
        raise SystemExit("pmc_collect.py profiles one process: --gpus other than 1 is refused (profile the rank program, not the launcher)")
    bench = [os.path.realpath(sys.executable), os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--cpu-seconds", "0", "--no-parity"]
    if args.depth_cells:
        bench += ["--depth-cells", str(args.depth_cells)]
    bench += args.bench_arg
    fim_angle = 1.0
    for i, arg in enumerate(args.bench_arg):
        if arg == "--fim-angle" and i + 1 < len(args.bench_arg):
            fim_angle = float(args.bench_arg[i + 1])
        elif arg.startswith("--fim-angle="):
            fim_angle = float(arg.split("=", 1)[1])
    summary = {"source_hash": source_hash(), "workload": args.workload, "depth_cells": args.depth_cells, "fim_angle": fim_angle,
               "collected_with": "tools/pmc_collect.py: rocprofv3 --pmc <one counter set per pass> --kernel-trace --output-format csv -- "
                                 "python3 bench.py --steps 3 --warmup 2 --repeats 2 --cpu-seconds 0 --no-parity; means over the last 6 dispatches",
               "counters_missing": []}

    if not args.skip_stats:
        d = os.path.join(out, "stats")
        shutil.rmtree(d, ignore_errors=True)
        run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", *bench, "--steps", "20", "--warmup", "5", "--repeats", "5"],
            os.path.join(out, "stats.log"), 600)
        for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
        bl = [ln for ln in open(os.path.join(out, "stats.log")) if ln.startswith("{") and '"metric"' in ln]
        if bl:
            open(os.path.join(out, "bench_under_rocprof.json"), "w").write(bl[-1])

    per_kernel = collections.defaultdict(dict)
    for name in args.passes.split(","):
        ctrs = [c for c in SQ_PASSES[name] if c in have]
        summary["counters_missing"] += [c for c in SQ_PASSES[name] if c not in have]
        if not ctrs:
            continue
        d = os.path.join(out, f"pass_{name}")
        shutil.rmtree(d, ignore_errors=True)
        rc = run(["rocprofv3", "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", d, "--", *bench, "--steps", "3", "--warmup", "2", "--repeats", "2", "--min-timed-seconds", "0"],
                 os.path.join(out, f"pass_{name}.log"), 600)
        if rc != 0:
            summary.setdefault("failed_passes", []).append(name)
            continue
        rows = collect_rows(d, os.path.join(out, f"pmc_{name}.csv"))
        for k, v in averages(rows).items():
            per_kernel[k].update(v)
        shutil.rmtree(d, ignore_errors=True)

    calib = {}
    exe = os.path.join(ROOT, "tools", "valu_calib")
    if not args.skip_calib and os.path.exists(exe):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        open(os.path.join(out, "calib_stdout.jsonl"), "w").write(r.stdout + r.stderr)
        calib["s_memtime"] = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{\"waves")]
        ck = collections.defaultdict(dict)
        for name, ctrs in CALIB_PASSES.items():
            ctrs = [c for c in ctrs if c in have]
            d = os.path.join(out, f"calib_{name}")
            shutil.rmtree(d, ignore_errors=True)
            rc = run(["rocprofv3", "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", d, "--", exe], os.path.join(out, f"calib_{name}.log"), 300)
            if rc == 0:
                rows = collect_rows(d, os.path.join(out, f"pmc_calib_{name}.csv"))
                for k, v in averages(rows, last=1).items():      # the second (timed) launch of each configuration
                    ck[k].update(v)
            shutil.rmtree(d, ignore_errors=True)
        calib["counters"] = ck

    derive(summary, per_kernel, calib)
    json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True, default=dict)
    print(json.dumps({k: summary.get(k) for k in ("source_hash", "valu_peak_wave_insts_per_cycle_per_simd", "valu_issue_utilisation", "fs_fim_kernel",
                                                 "fs_fim_kernel_hbm_bytes_per_launch", "counters_missing", "failed_passes")}, indent=1, default=dict))


if __name__ == "__main__":
    main()
