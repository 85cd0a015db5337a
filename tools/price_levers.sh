#!/bin/bash
# GPU box, one session: same-box prices of the bf16 energy levers of VERDICT r3 item 2 (variants built by tools/asm_variants.py):
#   x_dotsum   row sums of the rounded P by v_dot2c (32 VALU instructions fewer per step; a candidate for a caller-selected fast form; not built)
#   x_mfmasum  row sums from the matrix pipe (64 v_add_f32 fewer, 8 MFMAs more per step; pricing stand-in, wrong results)
#   x_hs8/4    every wave sits out one step in 8 / 4 (per-128-row-half lists walked as their union; pricing stand-in, wrong results)
# and the same dot-product row sums on the issue-bound head dims 64 / 96. Output: gpurun_out/<tag>/levers.txt
TAG=${1:-r04c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
V=build_variants
{
echo "== correctness of the variants that keep the results (abl_bench: max err vs fp32 torch, dense and with lists)"
for n in x_base x_dotsum; do LITEATTENTION_AMD_LIB=$R/$V/$n.so python tools/abl_bench.py; done
echo "== headline shape, interleaved subprocesses (tools/ab.py): dense / imposed 42 % / 77 %"
python tools/ab.py --reps 3 base=$V/x_base.so dotsum=$V/x_dotsum.so mfmasum=$V/x_mfmasum.so hs8=$V/x_hs8.so hs4=$V/x_hs4.so
echo "== head_dim 64 (tools/d64_bench.py), 96 (tools/d256_bench.py 96): base / dotsum, twice each, interleaved"
for r in 1 2; do
  for n in d64_base d64_dotsum; do echo -n "$n: "; LITEATTENTION_AMD_LIB=$R/$V/$n.so python tools/d64_bench.py; done
  for n in d96_base d96_dotsum; do echo -n "$n: "; LITEATTENTION_AMD_LIB=$R/$V/$n.so python tools/d256_bench.py 96 | tail -1; done
done
} > $OUT/levers.txt 2>&1
cat $OUT/levers.txt
