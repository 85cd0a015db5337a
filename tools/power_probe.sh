#!/bin/bash
# GPU box: socket power and clocks (rocm-smi) sampled while one kernel runs in a loop - the direct evidence for HISTORY.md section 4.2.
# usage: tools/power_probe.sh <outfile> ; runs bf16 42 %, bf16 dense, fp8 42 %, an MFMA-idle memory copy loop for contrast
OUT=${1:-gpurun_out/power_probe.txt}
mkdir -p $(dirname $OUT)
probe() { # label, python snippet
  python - "$1" <<PY &
import sys, time, torch
sys.path.insert(0, "$PWD")
import liteattention_amd as L
from bench import banded_rows, impose_lists
S, H, D = 75600, 40, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
mode = sys.argv[1]
if mode == "copy":
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    t = time.time()
    while time.time() - t < 9: b.copy_(a); torch.cuda.synchronize()
    sys.exit(0)
if mode.startswith("fp8"):
    q, k, v = [x.to(torch.float8_e4m3fn) for x in (q, k, v)]
bm, bn = L.get_tile_sizes(D, q.element_size())
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
impose_lists(att, banded_rows(-(-S // bm), -(-S // bn), bm, bn, 0.0 if mode.endswith("dense") else 0.42))
t = time.time(); n = 0
torch.cuda.synchronize(); t0 = time.time()
while time.time() - t < 9:
    for _ in range(10): att(q, k, v)
    torch.cuda.synchronize(); n += 10
print(f"{mode}: {(time.time() - t0) / n * 1e3:.2f} ms per call over {n} calls")
PY
  PID=$!
  sleep 5
  echo "== $1" >> $OUT
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> $OUT; sleep 0.7; done
  wait $PID >> $OUT 2>&1
}
: > $OUT
rocm-smi --showmaxpower 2>/dev/null | grep -i power >> $OUT
for m in bf16_42 bf16_dense fp8_42 copy; do probe $m >> $OUT 2>&1; done
cat $OUT
