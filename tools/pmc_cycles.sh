#!/bin/bash
# GPU box: cycles-vs-clock split for several library variants. usage: tools/pmc_cycles.sh lib1.so lib2.so ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so); OUT=$R/gpurun_out/pmcc_$name; mkdir -p $OUT
  LITEATTENTION_AMD_LIB=$R/$lib rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU \
     --kernel-trace --output-format csv -d $OUT -o p -- python $R/tools/abl_bench.py > $OUT/log.txt 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "la_fwd" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
dur = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "la_fwd" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
a = {k: acc[k] / n[k] for k in acc}
ms = sum(dur) / max(len(dur), 1)
cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
steps = 40960 * 256
print(f"$name: {ms:.2f} ms  clk {cyc/ms/1e6:.2f} GHz  Mcyc {cyc/1e6:.2f}  mfma_util {a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/cyc:.3f}  per wave-step quads: total {a.get('SQ_WAVE_CYCLES',0)/steps:.0f} active {a.get('SQ_ACTIVE_INST_ANY',0)/steps:.0f} issue-stall {a.get('SQ_WAIT_INST_ANY',0)/steps:.0f} parked {a.get('SQ_WAIT_ANY',0)/steps:.0f} valu {a.get('SQ_ACTIVE_INST_VALU',0)/steps:.0f}")
PY
done
