#!/bin/bash
# Builds A/B and pricing variants of the 16x16x32 body (gen_fwd_x64_m16.py) under build_variants/: usage tools/m16_variants.sh name=opts ...
# (opts = LA_X64_OPT of the head_dim-128 body; pricing options give wrong results and say so in la_build_info()).
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%=*}; opts=${spec#*=}
  ( LA_X64_OPT="$opts" python -m liteattention_amd.build -DLA_X64_M16=1 --out=build_variants/$name.so > /dev/null 2> build_variants/$name.log && echo "built $name [$opts]" || { echo "FAILED $name"; tail -5 build_variants/$name.log; } ) &
  while [ "$(jobs -r | wc -l)" -ge 2 ]; do sleep 1; done
done
wait
