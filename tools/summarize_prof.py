#!/usr/bin/env python
"""Summarise a tools/profile.sh output directory into profiles/<tag>_*.{md,json} (committed evidence).

usage: python tools/summarize_prof.py gpurun_out/prof_<tag> <tag> [--flops-per-launch F]
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-units of
the TCC_EA0 request counters, collected in their own --pmc passes; on gfx950 FETCH_SIZE reports exactly
half of a wide coalesced streaming read, so the read side is doubled.
"""
import collections
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return (name[:90] + "...") if len(name) > 93 else name


lines = [f"# rocprofv3 summary `{tag}`", ""]
bench = None
for l in open(os.path.join(src, "kt.log")):
    if l.startswith('{"metric"'):
        bench = json.loads(l)
if bench:
    lines += ["Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-sweep "
              "--no-cpu-baseline --steps 3 --warmup 1` (tools/profile.sh); PMC in separate `--pmc` passes.", "",
              f"bench line under the profiler: value={bench['value']} {bench['unit']}, ms_per_step={bench['ms_per_step']}, "
              f"kernel_ms(HIP events)={bench['roofline']['kernel_ms']}", ""]

lines += ["## kernel stats (--kernel-trace --stats)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
kernel_avg_ns = None
for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))):
    lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | "
                 f"{float(r['MaxNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
    if "la_fwd" in r["Name"] and kernel_avg_ns is None:
        kernel_avg_ns = float(r["AverageNs"])

pmc = {}
meta = {}
for name in ("pmc_mfma", "pmc_wait", "pmc_fetch", "pmc_write"):
    f = os.path.join(src, name, f"{name}_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "la_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"],
                    "lds": r["LDS_Block_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]}
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)

summary = {"tag": tag, "kernel_avg_ms": None if kernel_avg_ns is None else kernel_avg_ns / 1e6, "pmc_per_launch": pmc,
           "kernel_resources": meta}
if pmc:
    lines += ["", "## PMC, forward kernel, average per launch", "", "| counter | value |", "|---|---|"]
    for k in sorted(pmc):
        lines.append(f"| {k} | {pmc[k]:.4g} |")
    d = {}
    if "GRBM_GUI_ACTIVE" in pmc and kernel_avg_ns:
        d["clock_GHz (GRBM_GUI_ACTIVE/8 XCDs / kernel time)"] = pmc["GRBM_GUI_ACTIVE"] / 8 / kernel_avg_ns
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        d["mfma_util (MFMA_BUSY / (1024 SIMDs x cycles))"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * pmc["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_LDS_IDX_ACTIVE" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        d["lds_util (LDS_IDX_ACTIVE / (256 CUs x cycles))"] = pmc["SQ_LDS_IDX_ACTIVE"] / (256 * pmc["GRBM_GUI_ACTIVE"] / 8)
    if all(k in pmc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
        tot = pmc["SQ_WAIT_ANY"] + pmc["SQ_WAIT_INST_ANY"] + pmc["SQ_ACTIVE_INST_ANY"]
        d["wave_state_fracs (wait_any / wait_inst / active)"] = [pmc["SQ_WAIT_ANY"] / tot, pmc["SQ_WAIT_INST_ANY"] / tot,
                                                                 pmc["SQ_ACTIVE_INST_ANY"] / tot]
    if "FETCH_SIZE" in pmc:
        d["hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2 gfx950 correction)"] = pmc["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in pmc:
        d["hbm_write_bytes (WRITE_SIZE KiB x 1024)"] = pmc["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in pmc:
        d["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        d["hbm_bytes_per_launch"] = pmc["FETCH_SIZE"] * 2048 + pmc["WRITE_SIZE"] * 1024
        if kernel_avg_ns:
            d["hbm_GBps"] = d["hbm_bytes_per_launch"] / kernel_avg_ns
    summary["derived"] = d
    lines += ["", "## derived", ""]
    for k, v in d.items():
        lines.append(f"- {k}: {v}")
    if meta:
        lines += ["", f"kernel resources: {meta}"]

open(os.path.join(out_dir, f"{tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out_dir, f"{tag}_rocprof_summary.json"), "w"), indent=1)
if "derived" in summary and "hbm_bytes_per_launch" in summary["derived"]:
    json.dump({"hbm_bytes_per_launch": summary["derived"]["hbm_bytes_per_launch"],
               "source": f"profiles/{tag}_rocprof_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)"},
              open(os.path.join(out_dir, "r01_pmc_summary_fp8.json" if "fp8" in tag else "r01_pmc_summary.json"), "w"), indent=1)
print("\n".join(lines))
