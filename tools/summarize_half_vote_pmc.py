#!/usr/bin/env python
"""tools/half_vote_pmc.sh output -> profiles/r06_half_vote_pmc.md (CPU). Counters are the mean of the probe's last three forward dispatches."""
import collections, csv, glob, os, re, sys
csv.field_size_limit(1 << 30)
src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(name):
    by = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd_x64_kernel" in r["Kernel_Name"]:
                by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-3:]
    acc = collections.defaultdict(list)
    for i in ids:
        for k, v in by[i].items():
            acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


rows = []
for vote in ("tile", "half"):
    for thr in ("-4.22", "-2.46"):
        probe = None
        for l in open(os.path.join(src, f"{vote}_{thr}_busy.log")):
            m = re.search(r"PROBE .*sparsity=([0-9.]+) ms=([0-9.]+) executed_tflops=([0-9.]+)", l)
            if m:
                probe = [float(x) for x in m.groups()]
        c = counters(f"{vote}_{thr}_busy")
        c.update(counters(f"{vote}_{thr}_fetch"))
        if not probe or "GRBM_GUI_ACTIVE" not in c:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        rows.append((vote, thr, probe[0], probe[1], probe[2], cyc / (probe[1] * 1e6), c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_VALU_MFMA_BUSY_CYCLES"],
                     c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("FETCH_SIZE", 0) * 2048 / 1e9))
md = ["# Round 6: the real step-49 lists under PMC in both list geometries (`tools/half_vote_pmc.sh`; one box, one session)", "",
      "B = 1, S = 75 600, H = 40, D = 128, bf16; the lists of the 50-step run at thr -4.22 / -2.46, frozen (thr = -inf) and timed; `tile` = the default 256-row vote,",
      "`half` = `LA_VOTE=half` (lists per 128-row half, union walk, waves sit out the tiles only the other half lists). Counters: mean of the last three forward dispatches;",
      "MFMA busy = `SQ_VALU_MFMA_BUSY_CYCLES` / (1024 x cycles), cycles = `GRBM_GUI_ACTIVE` / 8; L2 fills = `FETCH_SIZE` x 2048 (gfx950 correction).", "",
      "| vote | thr | sparsity of the list | ms (under PMC) | executed TFLOP/s | effective clock GHz | MFMA busy | MFMA busy cycles per launch | waves issuing / stalled / parked | L2 fills GB |",
      "|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    md.append(f"| {r[0]} | {r[1]} | {100 * r[2]:.1f} % | {r[3]:.2f} | {r[4]:.0f} | {r[5]:.2f} | {100 * r[6]:.1f} % | {r[7]:.4g} | {100 * r[8]:.1f} / {100 * r[9]:.1f} / {100 * r[10]:.1f} % | {r[11]:.1f} |")
by = {(r[0], r[1]): r for r in rows}
for thr in ("-4.22", "-2.46"):
    if ("tile", thr) in by and ("half", thr) in by:
        t, h = by[("tile", thr)], by[("half", thr)]
        md += ["", f"thr {thr}: the half vote lists {100 * (1 - (1 - h[2]) / (1 - t[2])):.1f} % fewer tiles, the matrix pipe is busy {100 * (1 - h[7] / t[7]):.1f} % fewer cycles, the launch takes "
               f"{100 * (1 - h[3] / t[3]):.1f} % less time at {h[5]:.2f} against {t[5]:.2f} GHz."]
open(os.path.join(ROOT, "profiles", "r06_half_vote_pmc.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
