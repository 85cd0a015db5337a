#!/usr/bin/env python
"""Summarise tools/sched_sweep.sh (gpurun_out/sched_sweep) into profiles/r03_sched_sweep.{md,json}.

Per (variant, list): un-profiled ms and executed TFLOP/s (PROBE line of the plain run), L2 fills per launch (FETCH_SIZE, KiB, x2: the
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md for wide streaming reads), L2 hit rate (TCC_HIT / (HIT + MISS)), the
effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel time of that pass) and MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles)).
Counters are averaged over the probe's last three forward-kernel dispatches of a pass.
usage: python tools/summarize_sched_sweep.py gpurun_out/sched_sweep [tag]"""
import collections, csv, glob, json, os, re, sys
csv.field_size_limit(1 << 30)
src = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r03_sched_sweep"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(name, last=3):
    by = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"]:
                by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-last:]
    acc = collections.defaultdict(list)
    for i in ids:
        for k, v in by[i].items():
            acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def probe(path):
    try:
        m = re.search(r"PROBE .*sparsity=([\d.]+) ms=([\d.]+) executed_tflops=([\d.]+)", open(path).read())
        return (float(m.group(1)), float(m.group(2)), float(m.group(3))) if m else None
    except OSError:
        return None


rows = []
for plain in sorted(glob.glob(os.path.join(src, "*_plain.log"))):
    base = os.path.basename(plain)[: -len("_plain.log")]
    variant, thr = base.rsplit("_", 1)
    p = probe(plain)
    if p is None:
        continue
    r = {"variant": variant, "thr": float(thr), "sparsity": p[0], "ms": p[1], "executed_tflops": p[2]}
    f = counters(base + "_fetch")
    if "FETCH_SIZE" in f:
        r["l2_fills_GB"] = round(f["FETCH_SIZE"] * 1024 * 2 / 1e9, 1)
    h = counters(base + "_hit")
    if "TCC_HIT_sum" in h and h["TCC_HIT_sum"] + h.get("TCC_MISS_sum", 0) > 0:
        r["l2_hit"] = round(h["TCC_HIT_sum"] / (h["TCC_HIT_sum"] + h["TCC_MISS_sum"]), 4)
    b = counters(base + "_busy")
    pb = probe(os.path.join(src, base + "_busy.log"))
    if "GRBM_GUI_ACTIVE" in b and pb:
        cyc = b["GRBM_GUI_ACTIVE"] / 8
        r["clock_GHz"] = round(cyc / (pb[1] * 1e6), 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in b:
            r["mfma_busy"] = round(b["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
    rows.append(r)

out = {"what": "ticket-order sweep on the real step-49 lists (B=1 S=75600 H=40 D=128 bf16): chunk size C x heads in flight G; tree = C32 G4",
       "rows": rows}
with open(os.path.join(ROOT, "profiles", tag + ".json"), "w") as fjson:
    json.dump(out, fjson, indent=1)
lines = ["# Ticket-order sweep on real (fragmented) lists — " + tag, "",
         "`tools/sched_sweep.sh` on one box; `tree` = the committed order (chunks of 32 q-tiles, 4 heads interleaved). "
         "Lists: step 49 of the 50-step run at thr -4.22 (~44 % sparsity) and -2.46 (~78 %), frozen (thr = -inf).", "",
         "| list | variant | ms | executed TFLOP/s | vs tree | L2 fills GB | L2 hit | clock GHz | MFMA busy |", "|---|---|---|---|---|---|---|---|---|"]
for thr in sorted({r["thr"] for r in rows}):
    base = next((r for r in rows if r["thr"] == thr and r["variant"] == "tree"), None)
    for r in sorted((r for r in rows if r["thr"] == thr), key=lambda r: r["ms"]):
        rel = f"{(base['ms'] / r['ms'] - 1) * 100:+.1f} %" if base else ""
        lines.append(f"| thr {thr} ({r['sparsity']:.1%}) | {r['variant']} | {r['ms']:.2f} | {r['executed_tflops']:.0f} | {rel} | "
                     f"{r.get('l2_fills_GB', '')} | {r.get('l2_hit', '')} | {r.get('clock_GHz', '')} | {r.get('mfma_busy', '')} |")
with open(os.path.join(ROOT, "profiles", tag + ".md"), "w") as fmd:
    fmd.write("\n".join(lines) + "\n")
print("\n".join(lines))
