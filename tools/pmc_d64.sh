#!/bin/bash
# GPU box: wave-state / MFMA / LDS counters of the head_dim-64 kernel for several library variants. usage: tools/pmc_d64.sh lib.so ...  ("tree" = in-tree)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so); OUT=$R/gpurun_out/pmcd64_$name; mkdir -p $OUT
  if [ $lib = tree ]; then unset LITEATTENTION_AMD_LIB; else export LITEATTENTION_AMD_LIB=$R/$lib; fi
  python $R/tools/d64_bench.py 2>&1 | tail -1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU \
     --kernel-trace --kernel-include-regex "la_fwd" --output-format csv -d $OUT/a -o p -- python $R/tools/d64_bench.py > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE \
     --kernel-trace --kernel-include-regex "la_fwd" --output-format csv -d $OUT/b -o p -- python $R/tools/d64_bench.py > $OUT/b.log 2>&1
  python - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
res = {}
for sub in "ab":
    acc, n, dur = collections.defaultdict(float), collections.Counter(), []
    for f in glob.glob("$OUT/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for f in glob.glob("$OUT/" + sub + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    a = {k: acc[k] / n[k] for k in acc}
    ms = sum(dur) / max(len(dur), 1)
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    if sub == "a":
        wc = max(a.get("SQ_WAVE_CYCLES", 1), 1)
        print(f"$name: {ms:.2f} ms clk {cyc / ms / 1e6:.2f} GHz mfma_busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(cyc, 1):.3f} waves: issuing {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f} "
              f"issue-stalled {a.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} parked {a.get('SQ_WAIT_ANY', 0) / wc:.3f} valu-active {a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f} wave-cycles/cu-cycle {wc / max(cyc, 1) / 256:.2f}")
    else:
        print(f"   lds: insts {a.get('SQ_INSTS_LDS', 0):.3g} idx_active/cyc {a.get('SQ_LDS_IDX_ACTIVE', 0) / max(cyc, 1) / 256:.3f} bank_conflict/idx_active {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f} "
              f"wait_inst_lds {a.get('SQ_WAIT_INST_LDS', 0):.3g} valu insts {a.get('SQ_INSTS_VALU', 0):.3g} salu {a.get('SQ_INSTS_SALU', 0):.3g}")
PY
done
