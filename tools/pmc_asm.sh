#!/bin/bash
# GPU box: PMC passes over tools/abl_bench.py for the kernel selected by LA_FWD_KERNEL / LITEATTENTION_AMD_LIB.
# usage: tools/pmc_asm.sh <tag>
set -u
TAG=${1:-asm}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/abl_bench.py"
[ -f $R/gpurun_out/counters.txt ] || rocprofv3 --list-avail > $R/gpurun_out/counters.txt 2>&1
run_pmc() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run_pmc a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
run_pmc b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run_pmc c SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT
python - <<PY
import csv, glob, collections
for name in "abc":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "la_fwd" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(f"{k:32s} {acc[k]/n[k]:16.0f}  (avg over {n[k]} launches)")
PY
