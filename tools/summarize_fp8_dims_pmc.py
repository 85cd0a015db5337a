"""Summarises gpurun_out/fp8_dims_pmc/d<D>/ (tools/fp8_dims_pmc.sh): per head dim the launch time, the effective clock, MFMA busy and the split of
the waves' cycles. Counters are summed over the device by rocprofv3; GRBM_GUI_ACTIVE counts per XCD (8), SQ_VALU_MFMA_BUSY_CYCLES per SIMD (1 024)."""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
root = sys.argv[1]
print("| head_dim | ms / launch | clock GHz | MFMA busy | waves: issuing | stalled on issue | parked (waitcnt / barrier) | VALU active |")
print("|---|---|---|---|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(root, "d*")), key=lambda p: int(os.path.basename(p)[1:])):
    acc, n, dur = collections.defaultdict(float), collections.Counter(), []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd_x64_fp8" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd_x64_fp8" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if not dur or not n:
        print(f"| {os.path.basename(d)[1:]} | no data | | | | | | |")
        continue
    a = {k: acc[k] / n[k] for k in acc}
    dur.sort()
    ms = dur[len(dur) // 2]
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    wc = max(a.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"| {os.path.basename(d)[1:]} | {ms:.3f} | {cyc / ms / 1e6:.2f} | {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(cyc, 1):.1%} | "
          f"{a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1%} | {a.get('SQ_WAIT_INST_ANY', 0) / wc:.1%} | {a.get('SQ_WAIT_ANY', 0) / wc:.1%} | "
          f"{a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.1%} |")
