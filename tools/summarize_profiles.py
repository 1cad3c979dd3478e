"""Turns the rocprofv3 outputs under gpurun_out/ (suffix given as argv[1], e.g. r1e) into the committed
summaries under profiles/: kernel stats, per-dispatch PMC rows of our kernels, pmc_traffic.json."""
import collections, csv, glob, json, os, re, shutil, sys
tag = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01"
name = sys.argv[3] if len(sys.argv) > 3 else "c3_final"
os.makedirs(dst, exist_ok=True)
res = collections.defaultdict(dict)
for kind in ("fetch", "write", "tcc"):
    for f in glob.glob(f"gpurun_out/pmc_{kind}_{tag}/runc/*_counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "fs_" in r["Kernel_Name"]]
        with open(f"{dst}/{name}_pmc_{kind}.csv", "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                              "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
            w.writeheader(); w.writerows(rows)
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            d[re.search(r"fs_\w+(<\d+)?", r["Kernel_Name"]).group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in d.items():
            for c, vals in v.items():
                vals = [x for x in vals if x > 0][-3:] or [0]
                res[k][c] = sum(vals) / len(vals)
for f in glob.glob(f"gpurun_out/prof_c3_{tag}/runc/*_kernel_stats.csv"):
    shutil.copy(f, f"{dst}/{name}_kernel_stats.csv")
if os.path.exists(f"gpurun_out/bench_c3_{tag}.json"):
    shutil.copy(f"gpurun_out/bench_c3_{tag}.json", f"{dst}/{name}_bench.json")
def pick(prefix):
    for k, v in res.items():
        if k.startswith(prefix):
            return v
    return {}
fim = pick("fs_fim_kernel<512"); ray = pick("fs_raymarch_kernel")
hit = lambda v: v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))
out = {"workload": "C3",
       "source": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (separate passes) --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-parity ({tag})",
       "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 64 B per 128-B request "
                "for wide coalesced reads, so reads are doubled; our loads are 4 B/lane (an uncalibrated width): the corrected figure is an upper bound",
       "fs_fim_kernel_fetch_kib": fim.get("FETCH_SIZE"), "fs_fim_kernel_write_kib": fim.get("WRITE_SIZE"),
       "fs_fim_kernel_hbm_bytes_per_launch": int((2 * fim.get("FETCH_SIZE", 0) + fim.get("WRITE_SIZE", 0)) * 1024),
       "fs_fim_kernel_hbm_bytes_per_launch_uncorrected": int((fim.get("FETCH_SIZE", 0) + fim.get("WRITE_SIZE", 0)) * 1024),
       "fs_fim_kernel_l2_hit_rate": hit(fim),
       "fs_raymarch_kernel_fetch_kib": ray.get("FETCH_SIZE"), "fs_raymarch_kernel_write_kib": ray.get("WRITE_SIZE"),
       "fs_raymarch_kernel_l2_hit_rate": hit(ray)}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
