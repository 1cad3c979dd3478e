#!/usr/bin/env python3
"""Rewrites the round-5 record table of DESIGN.md §5 from profiles/r05/*.json (so that the table is the record set, not a transcription of
it): `python tools/r05_table.py` prints the rows, `--write` replaces them in DESIGN.md."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r05")


def L(n):
    return json.load(open(os.path.join(P, n + ".json")))


def grp(x):
    """12345 -> '12 345' (thin grouping as in the rest of the document)"""
    s = f"{int(round(x)):,}".replace(",", " ")
    return s


def cpu(cb):
    if not cb:
        return "—"
    v = [cb.get("value"), cb.get("omp_value"), cb.get("omp_all_value")]
    return " / ".join(grp(x) if x else "—" for x in v)


def share(p):
    a, b = 100 * p["logdet_share_within_1e-4"], 100 * p["logdet_share_within_rule"]
    f = lambda x: f"{x:.0f} %" if abs(x - round(x)) < 0.05 else f"{x:.1f} %"
    return f(a), f(b)


def main():
    c3, c3r, r2, c5, l160, c4, reh = [L(n) for n in ("c3_bench", "c3_ref_request_bench", "ref2d_bench", "c5_bench", "c3_l160_bench", "c4_strong_n1_bench", "rehearse_multi_one_rank_rccl_bench")]
    stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(P, "c3_kernel_stats.csv")))}
    fim = next(v for k, v in stats.items() if "fs_fim_kernel<" in k)
    ray = next(v for k, v in stats.items() if "fs_raymarch_kernel<" in k)
    k = lambda j: j["kernels_ms_per_step"]
    M = lambda j: f"{j['value'] / 1e6:.1f} M" if j["value"] < 2e7 or True else ""
    rows = []
    t = c3["timing"]; p = c3["parity"]; a, b = share(p)
    rows.append(f"| `c3_bench.json` — **the headline** (`c3_kernel_stats.csv`: rocprofv3 {float(fim['AverageNs']) / 1e6:.4f} / {float(ray['AverageNs']) / 1e6:.4f} ms over {fim['Calls']} launches; `profiles/pmc_summary.json`) | C3, 14 m / 1.0 rad | **{M(c3)}** | {c3['ms_per_step']:.3f} ms ({t['repeats']} blocks, p10–p90 {t['block_ms_per_step_p10_p90'][0]:.3f}–{t['block_ms_per_step_p10_p90'][1]:.3f}; ranked on the device {c3['ranked_step']['ms_per_step']:.3f}) | {k(c3)['fs_fim_kernel']:.3f} ms | {k(c3)['fs_raymarch_kernel']:.3f} ms | **{c3['roofline']['frac']:.2f}**, M_tested {c3['roofline']['m_tested_per_candidate'] / 1e3:.1f} k | {grp(p['n'])}; {a} / {b} (max {p['logdet_max_rel_err']:.1e}) | {cpu(c3['cpu_baseline'])} |")
    p = c3r["parity"]; a, b = share(p); f = c3["other_operating_points"]["reference_request_visibility"]["fused_step"]
    rows.append(f"| `c3_ref_request_bench.json` (`profiles/pmc_summary_ref_request.json`) | C3, **the reference's request** (14 m, cone off) | {c3r['value'] / 1e6:.2f} M | {c3r['ms_per_step']:.3f} ms | {k(c3r)['fs_fim_kernel']:.3f} ms | {k(c3r)['fs_raymarch_kernel']:.3f} ms | **{c3r['roofline']['frac']:.2f}**, M_tested {c3r['roofline']['m_tested_per_candidate'] / 1e3:.1f} k ({grp(f['multi_pass_candidates_per_step'])} candidates in passes, {'none' if not f['hbm_table_candidates_per_step'] else grp(f['hbm_table_candidates_per_step'])} to the HBM tier) | {grp(p['n'])}; {a} / {b} (max {p['logdet_max_rel_err']:.0e}) | {cpu(c3r['cpu_baseline'])} |")
    p = r2["parity"]; a, b = share(p)
    rows.append(f"| `ref2d_bench.json` | REF2D (the reference's own 2-D operating point: 512² costmap, 63 rays), 1.0 rad | {M(r2)} | {r2['ms_per_step']:.3f} ms | {k(r2)['fs_fim_kernel']:.3f} ms | {k(r2)['fs_raymarch_kernel']:.3f} ms | {r2['roofline']['frac']:.2f}, M_tested {r2['roofline']['m_tested_per_candidate'] / 1e3:.1f} k | {grp(p['n'])}; **{a} / {b}** (two three-landmark poses, κ ≥ 1.4·10⁵: §2) | {cpu(r2['cpu_baseline'])} |")
    rr = r2["other_operating_points"]["reference_request_visibility"]; f = rr["fused_step"]
    rows.append(f"| same file, `reference_request_visibility` | REF2D, cone off | {f['candidate_goals_per_s'] / 1e6:.2f} M | {f['ms_per_step']:.3f} ms | {f['fs_fim_kernel_ms']:.3f} ms | — | M_tested {f['m_tested_per_candidate'] / 1e3:.1f} k, {'no passes' if not f['multi_pass_candidates_per_step'] else grp(f['multi_pass_candidates_per_step']) + ' candidates in passes'} | {rr['parity']['n']}; {'green' if rr['parity']['ok'] else 'RED'} | {cpu(rr.get('cpu_baseline'))} |")
    p = c5["parity"]; a, b = share(p)
    rows.append(f"| `c5_bench.json` | C5 (1024³ through the brick-list upload, 50 k candidates, 500 k landmarks), 1.0 rad | {M(c5)} | {c5['ms_per_step']:.3f} ms | {k(c5)['fs_fim_kernel']:.3f} ms | {k(c5)['fs_raymarch_kernel']:.3f} ms | {c5['roofline']['frac']:.2f}, M_tested {c5['roofline']['m_tested_per_candidate'] / 1e3:.1f} k | {grp(p['n'])}; {a} / {b} (max {p['logdet_max_rel_err']:.1e}) | {cpu(c5['cpu_baseline'])} |")
    rr = c5["other_operating_points"]["reference_request_visibility"]; f = rr["fused_step"]
    rows.append(f"| same file, `reference_request_visibility` | C5, cone off | {f['candidate_goals_per_s'] / 1e6:.2f} M | {f['ms_per_step']:.3f} ms | {f['fs_fim_kernel_ms']:.3f} ms | — | M_tested {f['m_tested_per_candidate'] / 1e3:.1f} k, {grp(f['multi_pass_candidates_per_step'])} candidates in passes | {rr['parity']['n']}; {'green' if rr['parity']['ok'] else 'RED'} | {cpu(rr.get('cpu_baseline'))} |")
    rows.append(f"| `c3_l160_bench.json` | C3 with 8 m rays (L = 160) | {M(l160)} | {l160['ms_per_step']:.3f} ms | {k(l160)['fs_fim_kernel']:.3f} ms | {k(l160)['fs_raymarch_kernel']:.3f} ms | {l160['roofline']['frac']:.2f} | — | — |")
    rows.append(f"| `c4_strong_n1_bench.json` | C4 as one list on one GPU (160 k candidates) | {M(c4)} | {c4['ms_per_step']:.3f} ms | {k(c4)['fs_fim_kernel']:.3f} ms | {k(c4)['fs_raymarch_kernel']:.3f} ms | {c4['roofline']['frac']:.2f} | — | — |")
    ss = reh["strong_scaling"]
    rows.append(f"| `rehearse_multi_one_rank_rccl_bench.json` | C3 through every N > 1 code path with ONE rank (RCCL group of one, all-gather behind every step {reh['multi_gpu']['all_gather_ms'] * 1e3:.0f} µs) | {M(reh)} | {reh['ms_per_step']:.3f} ms | {k(reh)['fs_fim_kernel']:.3f} ms | {k(reh)['fs_raymarch_kernel']:.3f} ms | {reh['roofline']['frac']:.2f} | {reh['parity']['n']} per block, {'green' if reh['parity']['ok'] else 'RED'}; `strong_scaling`: 160 k candidates {ss['ms_per_step']:.2f} ms = {ss['candidate_goals_per_s'] / 1e6:.1f} M/s, 48 per share, {'green' if (ss.get('parity') or {}).get('ok', True) else 'RED'} | — |")
    text = "\n".join(rows)
    if "--write" not in sys.argv:
        print(text)
        return
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    m = re.search(r"(\| `c3_bench\.json` — \*\*the headline\*\*.*?\n)(?=\n)", s, flags=re.S)
    assert m, "table not found"
    s = s[:m.start()] + text + "\n" + s[m.end():]
    open(path, "w").write(s)


if __name__ == "__main__":
    main()
