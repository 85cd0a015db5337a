#!/bin/bash
# GPU box: A/B of head_dim-256 library variants (tools/asm_variants.py name=x64d256:<opts>), interleaved twice.
# usage: tools/d256_ab.sh build_variants/a.so build_variants/b.so ...
for rep in 1 2; do
  for so in "$@"; do
    echo -n "$(basename $so) : "
    LITEATTENTION_AMD_LIB=$PWD/$so python tools/d256_bench.py 2>/dev/null | grep "D=256"
  done
done
