#!/bin/bash
# GPU box: cycles-vs-clock split for several library variants. usage: tools/pmc_cycles.sh lib1.so lib2.so ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so); OUT=$R/gpurun_out/pmcc_$name; mkdir -p $OUT
  LITEATTENTION_AMD_LIB=$R/$lib rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU \
     --kernel-trace --output-format csv -d $OUT -o p -- python $R/tools/abl_bench.py > $OUT/log.txt 2>&1
  python $R/tools/summarize_pmcc.py $OUT
done
