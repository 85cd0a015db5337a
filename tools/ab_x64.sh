#!/bin/bash
# GPU box: A/B the kernels on the quick bench (same box, interleaved).
for k in x64 asm v2 x64; do LA_FWD_KERNEL=$k timeout 300 python tools/quick_bench.py 2>&1 | tail -1; done
