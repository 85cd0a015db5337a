#!/bin/bash
# GPU box, ONE session: the evidence behind "this kernel is at the socket's power limit".
#   1. MFMA-only loops (tools/mfma_power_bench.py): 32x32x16 and 16x16x32 bf16, random vs all-zero operands, 1 and 2 waves per SIMD:
#      ms, TFLOP/s, rocm-smi W and sclk; then GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES of the four 1-wave cases in their own PMC pass
#   2. the ablation table of HISTORY.md 4.2 regenerated on the current body (tools/pmc_cycles.sh: cycles, clock, MFMA busy) with the
#      power of every variant beside it (tools/variant_power.py), plus the priced levers (dotsum, mfmasum, halfskip, mfma16)
#   3. the headline configuration on THIS box: bench line (with its rocm-smi sample) + the MFMA PMC pass   (tools/box_probe.sh)
# Variants are built on the CPU side first:  python tools/asm_variants.py x_base=x64: ...   (see profiles/r04_power_ceiling.md)
set -u
TAG=${1:-r04_power}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=$R/build_variants
# 1
python $R/tools/mfma_power_bench.py run $OUT/mfma_power.json > $OUT/mfma_power.log 2>&1
for c in 32x32x16_w1_rand 32x32x16_w1_zero 16x16x32_w1_rand 16x16x32_w1_zero; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/mfma_pmc_$c -o p -- $V/mfma_power ${c} 1.0 > $OUT/mfma_pmc_$c.log 2>&1
done
# 2
NAMES="x_base x_nobar x_nobar2 x_nodma x_nosoftmax x_mfmaonly x_dotsum x_mfmasum x_hs8 x_hs4 x_m16qk x_m16pv x_m16both"
ARGS=""; LIBS=""
for n in $NAMES; do [ -f $V/$n.so ] && ARGS="$ARGS ${n#x_}=$V/$n.so" && LIBS="$LIBS build_variants/$n.so"; done
python $R/tools/variant_power.py $OUT/variant_power.json $ARGS > $OUT/variant_power.log 2>&1
(cd $R && bash tools/pmc_cycles.sh $LIBS > $OUT/pmc_cycles.txt 2>&1)
# 3
bash $R/tools/box_probe.sh $TAG/box > $OUT/box_probe.log 2>&1
find $R/gpurun_out -name "*.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
rm -rf $R/gpurun_out/pmcc_*/*/*agent_info* 2>/dev/null
tail -20 $OUT/pmc_cycles.txt; tail -16 $OUT/variant_power.log
