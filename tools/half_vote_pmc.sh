#!/bin/bash
# GPU box (round 6): the real step-49 lists of the 50-step run at thr -4.22 / -2.46 under PMC, in BOTH list geometries - the default 256-row vote
# and LA_VOTE=half (lists per 128-row half: the waves of a half sit out the tiles only the other half lists). What the sitting-out returns: matrix-pipe
# busy cycles, the effective clock at the power cap, L2 fills. Summarise with: python tools/summarize_half_vote_pmc.py gpurun_out/hv_pmc
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/hv_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() { local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --kernel-include-regex "la_fwd" --output-format csv -d $OUT/$name -o p -- "$@" > $OUT/$name.log 2>&1; }
for vote in tile half; do
  if [ $vote = half ]; then export LA_VOTE=half; else unset LA_VOTE; fi
  for thr in -4.22 -2.46; do
    P="python $R/tools/traffic_probe.py --real $thr"
    pmc ${vote}_${thr}_busy GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -- $P
    pmc ${vote}_${thr}_fetch FETCH_SIZE -- $P
  done
done
unset LA_VOTE
grep -h PROBE $OUT/*_busy.log
