#!/bin/bash
# GPU box: A/B of library variants at one head dim (tools/asm_variants.py name=x64d<D>:<opts>), interleaved twice.
# usage: tools/d256_ab.sh <D> build_variants/a.so build_variants/b.so ...
D=$1; shift
for rep in 1 2; do
  for so in "$@"; do
    echo -n "$(basename $so) : "
    LITEATTENTION_AMD_LIB=$PWD/$so python tools/d256_bench.py $D 2>/dev/null | grep "D=$D"
  done
done
