#!/bin/bash
# GPU box: the fp8 kernel at head dims 64 / 128 / 192 / 256 under the cycle counters (dense S = 16 384, H = 40, the reference's arithmetic unless LA_FP8_P is set):
# effective clock, MFMA busy, and how a wave's cycles split into issuing / stalled / parked. One rocprofv3 --pmc pass per head dim (no tracing beside it).
#   usage: tools/fp8_dims_pmc.sh [dims...]   -> gpurun_out/fp8_dims_pmc/d<D>/, summary by tools/summarize_fp8_dims_pmc.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
DIMS=${@:-64 128 192 256}
for D in $DIMS; do
  OUT=$R/gpurun_out/fp8_dims_pmc/d$D; mkdir -p $OUT
  LA_PROBE_DIM=$D timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU \
     --kernel-trace --output-format csv -d $OUT -o p -- python $R/tools/debug/fp8_dim_launches.py > $OUT/log.txt 2>&1
done
python $R/tools/summarize_fp8_dims_pmc.py $R/gpurun_out/fp8_dims_pmc
