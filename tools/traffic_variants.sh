#!/bin/bash
# GPU box: bytes / hit rate / clock of the REAL step-49 lists (thr -4.22 ~ 44 %, thr -2.46 ~ 78 %) for several library variants
# (the cache policy of the K/V LDS-DMA stream on real lists). usage: tools/traffic_variants.sh name=lib.so ...
# Output: gpurun_out/trv_<name>_<thr>_{fetch,write,busy}/ + one summary line per (variant, list) by tools/summarize_traffic_variants.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
pmc() { local out=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --kernel-include-regex "la_fwd" --output-format csv -d $out -o p -- "$@" > $out.log 2>&1; }
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  [ "$lib" = tree ] && unset LITEATTENTION_AMD_LIB || export LITEATTENTION_AMD_LIB=$R/$lib
  for thr in -4.22 -2.46; do
    O=$R/gpurun_out/trv_${name}_${thr}; mkdir -p $O
    P="python $R/tools/traffic_probe.py --real $thr"
    pmc ${O}/fetch FETCH_SIZE -- $P
    pmc ${O}/write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $P
    pmc ${O}/busy GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- $P
    grep -h PROBE ${O}/fetch.log | tail -1 | sed "s/^/$name $thr: /"
  done
done
python $R/tools/summarize_traffic_variants.py $R/gpurun_out "$@"
