#!/usr/bin/env python3
"""Copies the round-5 record set (gpurun_out/r05/final + the two pmc collections, made by tools/r05_bench_set.sh on the GPU box)
into profiles/: bench lines, kernel stats, counter rows (the last 24 dispatches per kernel: the means use the last 6), the
summaries bench.py replays.  The reference-request summary gets its derived figures from the C3 collection's calibration
(tools/pmc_collect.py --rederive).  No GPU needed."""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r05")
P = os.path.join(ROOT, "profiles", "r05")


def trimmed(src, dst, keep=24):
    rows = list(csv.DictReader(open(src)))
    if not rows:
        return
    ids = collections.defaultdict(set)
    for r in rows:
        ids[r["Kernel_Name"]].add(int(r["Dispatch_Id"]))
    last = {k: set(sorted(v)[-keep:]) for k, v in ids.items()}
    with open(dst, "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(r for r in rows if int(r["Dispatch_Id"]) in last[r["Kernel_Name"]])


def main():
    os.makedirs(P, exist_ok=True)
    for f in glob.glob(os.path.join(G, "final", "*.json")):
        shutil.copy(f, P)
    c3, ref = os.path.join(G, "pmc_c3"), os.path.join(G, "pmc_c3_ref")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_collect.py"), "--rederive", ref, "--calib-from", os.path.join(c3, "pmc_summary.json")],
                          stdout=subprocess.DEVNULL)
    shutil.copy(os.path.join(c3, "kernel_stats.csv"), os.path.join(P, "c3_kernel_stats.csv"))
    shutil.copy(os.path.join(c3, "bench_under_rocprof.json"), os.path.join(P, "c3_bench_under_rocprof.json"))
    shutil.copy(os.path.join(c3, "calib_stdout.jsonl"), os.path.join(P, "valu_calib_stdout.jsonl"))
    for f in glob.glob(os.path.join(c3, "pmc_calib_*.csv")):
        shutil.copy(f, os.path.join(P, "valu_calib_" + os.path.basename(f).replace("pmc_calib_", "pmc_")))
    for d, tag in ((c3, "c3"), (ref, "ref_request")):
        for f in glob.glob(os.path.join(d, "pmc_*.csv")):
            if "calib" in f:
                continue
            trimmed(f, os.path.join(P, f"{tag}_{os.path.basename(f)}"))
        shutil.copy(os.path.join(d, "pmc_summary.json"), os.path.join(P, f"{tag}_pmc_summary.json"))
    shutil.copy(os.path.join(c3, "pmc_summary.json"), os.path.join(ROOT, "profiles", "pmc_summary.json"))
    shutil.copy(os.path.join(ref, "pmc_summary.json"), os.path.join(ROOT, "profiles", "pmc_summary_ref_request.json"))
    sys.path.insert(0, ROOT)
    import bench
    import json
    for name in ("pmc_summary.json", "pmc_summary_ref_request.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        print(name, j["source_hash"], "== bench.source_hash()" if j["source_hash"] == bench.source_hash() else "!= " + bench.source_hash() + "  (STALE: sources changed since the collection)")


if __name__ == "__main__":
    main()
