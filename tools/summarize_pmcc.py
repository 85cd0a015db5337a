"""Summarises gpurun_out/pmcc_<name>/ (tools/pmc_cycles.sh): wall, effective clock, cycles, MFMA utilisation, wave-cycle split."""
import csv, glob, collections, sys, os
csv.field_size_limit(1 << 30)
for d in sys.argv[1:]:
    acc = collections.defaultdict(float); n = collections.Counter(); dur = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 1000000:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    a = {k: acc[k] / n[k] for k in acc}
    ms = sum(dur) / max(len(dur), 1)
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    steps = int(os.environ.get('WAVE_STEPS', 40960 * 256))
    g = lambda k: a.get(k, 0) / steps
    print(f"{os.path.basename(d)[5:]:12s} {ms:6.2f} ms  clk {cyc/ms/1e6:.2f} GHz  Mcyc {cyc/1e6:6.2f}  mfma_util {a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(cyc,1):.3f}"
          f"  quads/wave-step: total {g('SQ_WAVE_CYCLES'):.0f} active {g('SQ_ACTIVE_INST_ANY'):.0f} issue-stall {g('SQ_WAIT_INST_ANY'):.0f} parked {g('SQ_WAIT_ANY'):.0f} valu {g('SQ_ACTIVE_INST_VALU'):.0f}")
