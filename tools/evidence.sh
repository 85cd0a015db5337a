#!/bin/bash
# GPU box: every profile the documents cite, in one call. Outputs under gpurun_out/ev_<tag>/ ; summarise with
#   python tools/summarize_evidence.py gpurun_out/ev_<tag> <tag>          (writes profiles/<tag>_*)
# PMC counters are collected in their own rocprofv3 passes (never together with trace domains), FETCH_SIZE alone (3 TCC slots).
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ev_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() { # dir-name counters... -- command
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --kernel-include-regex "la_fwd|la_prep" --output-format csv -d $OUT/$name -o p -- "$@" > $OUT/$name.log 2>&1
}
want() { [ "${EV_SECTIONS:-all}" = all ] || [[ " $EV_SECTIONS " == *" $1 "* ]]; }    # EV_SECTIONS="traffic_real other" re-runs parts
# 1. the driver's own command, un-profiled: the bench line of record
want bench && python $R/bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
want bench && python $R/bench.py --dtype fp16 --no-denoise --no-cpu-baseline > $OUT/bench_line_fp16.json 2>> $OUT/bench_line.err
# 2. kernel trace + stats of the same command (shorter loop)
want kt && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --no-sweep --no-cpu-baseline --no-denoise --no-head-dims --no-power --steps 5 --warmup 2 > $OUT/kt.log 2>&1
# 3. PMC passes, bf16 headline and fp8
for dt in bf16 fp8; do
  want pmc || continue
  B="python $R/bench.py --no-sweep --no-cpu-baseline --no-fp8 --no-verify --no-denoise --no-head-dims --no-power --steps 3 --warmup 1 --dtype $dt"
  pmc ${dt}_mfma SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU -- $B
  pmc ${dt}_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- $B
  pmc ${dt}_fetch FETCH_SIZE -- $B
  pmc ${dt}_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $B
done
# 4. bytes vs sparsity: imposed 0 / 42 / 77 % and the real (fragmented) lists of the 50-step run at ~44 % / ~78 %
for cfg in "imposed 0.0" "imposed 0.42" "imposed 0.77" "real -4.22" "real -2.46"; do
  set -- $cfg; n=traffic_$1_$2
  want traffic_$1 || continue
  P="python $R/tools/traffic_probe.py --$1 $2"
  pmc ${n}_fetch FETCH_SIZE -- $P
  pmc ${n}_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $P
  pmc ${n}_busy GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- $P
done
# 5. the other instantiations: bench lines + kernel stats + MFMA utilisation (d64: tools/d64_bench.py; 96 / 192 / 256: tools/d256_bench.py <D>)
# (d128: the headline body on the same shape, for the stall breakdown table: what the others are measured against)
for t in d64 d96 d128 d192 d256; do
  want other || continue
  if [ $t = d64 ]; then C="python $R/tools/d64_bench.py"; else C="python $R/tools/d256_bench.py ${t#d}"; fi
  $C > $OUT/${t}_bench.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${t}_kt -o kt -- $C > $OUT/${t}_kt.log 2>&1
  pmc ${t}_mfma SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -- $C
  # the stall breakdown: how much of the issue stall is the LDS, how busy the LDS array is, instruction counts
  pmc ${t}_lds SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- $C
done
# the hipcc-scheduled A/B kernels on the same box (head dims 192 / 96 zero-padded onto 256 / 128 by the host)
want other && LA_FWD_KERNEL=v2 python $R/tools/d256_bench.py > $OUT/v2_bench.txt 2>&1
# 5b. fp8: the three forms of P (LA_FP8_P=encoded / mfma_rowsum / the default = the reference's arithmetic; labelled default / exp / rowsum as in rounds 3-5) on the same box: banded
#     headline lists (ms, TFLOP/s) and the real step-49 lists with the error against fp32 torch on sampled rows
if want fp8forms; then
  { for m in default exp rowsum; do
      unset LA_FP8_P; [ $m = exp ] && export LA_FP8_P=mfma_rowsum; [ $m = default ] && export LA_FP8_P=encoded
      echo "P form $m: $(python $R/tools/fp8_quick.py 2>&1 | grep 's=')"
    done; unset LA_FP8_P
    python $R/tools/debug/fp8_tail_probe.py -4.22 2>&1 | grep "real lists"
    python $R/tools/debug/fp8_tail_probe.py -2.462 2>&1 | grep "real lists"; } > $OUT/fp8_p_forms.txt 2>&1
fi
# 6. socket power and clocks under the kernels (rocm-smi; HISTORY.md section 4.2)
want power && (cd $R && bash tools/power_probe.sh $OUT/power_probe.txt > /dev/null 2>&1)
ls $OUT | head -80
grep -h PROBE $OUT/traffic_*_fetch.log 2>/dev/null
[ -f $OUT/bench_line.json ] && tail -c 300 $OUT/bench_line.json
