"""Summarises gpurun_out/trv_<name>_<thr>/ (tools/traffic_variants.sh): per (variant, real list) the kernel time, L2 fills
(FETCH_SIZE x 2 per the guide's gfx950 correction), hit rate, effective clock, MFMA busy. Counters: mean of the last 3 forward dispatches."""
import collections, csv, glob, os, sys
csv.field_size_limit(1 << 30)
root, specs = sys.argv[1], sys.argv[2:]


def last3(d):
    by = collections.defaultdict(dict)
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"]:
                by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-3:]
    acc = collections.defaultdict(list)
    for i in ids:
        for k, v in by[i].items():
            acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def probe_ms(log):
    try:
        line = [l for l in open(log) if "PROBE" in l][-1]
        return float(line.split("ms=")[1].split()[0]), line.strip()
    except Exception:
        return None, ""


print("| variant | list | ms | L2 fills GB | L2 hit | clock GHz | MFMA busy |")
print("|---|---|---|---|---|---|---|")
for spec in specs:
    name = spec.split("=")[0]
    for thr in ("-4.22", "-2.46"):
        o = os.path.join(root, f"trv_{name}_{thr}")
        f, w, b = last3(os.path.join(o, "fetch")), last3(os.path.join(o, "write")), last3(os.path.join(o, "busy"))
        ms, _ = probe_ms(os.path.join(o, "busy.log"))
        ms_f, line = probe_ms(os.path.join(o, "fetch.log"))
        fills = f.get("FETCH_SIZE", 0) * 2048 / 1e9
        hit = w.get("TCC_HIT_sum", 0) / max(1.0, w.get("TCC_HIT_sum", 0) + w.get("TCC_MISS_sum", 0))
        cyc = b.get("GRBM_GUI_ACTIVE", 0) / 8
        clk = cyc / (ms * 1e6) if ms else 0
        busy = b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(cyc, 1)
        print(f"| {name} | thr {thr} | {ms if ms else 0:.2f} | {fills:.1f} | {hit:.1%} | {clk:.2f} | {busy:.1%} |")
