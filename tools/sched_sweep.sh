#!/bin/bash
# GPU box: the ticket-order sweep of round 3 on the REAL step-49 lists of the 50-step run (thr -4.22 / -2.46):
# chunk size C (q-tiles of one head per XCD-queue chunk) x heads in flight G, one library per setting (built here with
#   python -m liteattention_amd.build -DLA_SCHED_C=<C> -DLA_SCHED_G=<G> --out=build_variants/sched_c<C>_g<G>.so ).
# Per variant and list: one un-profiled run (ms), FETCH_SIZE pass, TCC_HIT/MISS pass, GRBM_GUI_ACTIVE + MFMA busy pass.
#   summarise with python tools/summarize_sched_sweep.py gpurun_out/sched_sweep  -> profiles/r03_sched_sweep.{md,json}
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sched_sweep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VARIANTS=${VARIANTS:-"tree c16_g4 c64_g4 c32_g1 c32_g2 c32_g8 c16_g2 c64_g2 c8_g4"}
for thr in -4.22 -2.46; do
  python $R/tools/traffic_probe.py --real $thr > $OUT/warm_$thr.log 2>&1       # builds the list cache in /tmp
  for v in $VARIANTS; do
    if [ $v = tree ]; then unset LITEATTENTION_AMD_LIB; else export LITEATTENTION_AMD_LIB=$R/build_variants/sched_$v.so; fi
    n=${v}_$thr
    P="python $R/tools/traffic_probe.py --real $thr"
    $P > $OUT/${n}_plain.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "la_fwd" --output-format csv -d $OUT/${n}_fetch -o p -- $P > $OUT/${n}_fetch.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "la_fwd" --output-format csv -d $OUT/${n}_hit -o p -- $P > $OUT/${n}_hit.log 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "la_fwd" --output-format csv -d $OUT/${n}_busy -o p -- $P > $OUT/${n}_busy.log 2>&1
    grep -h PROBE $OUT/${n}_plain.log
  done
done
# keep only the counter CSVs (the merge back is capped at 64 MiB)
find $OUT -name "*.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $OUT
