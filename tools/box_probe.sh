#!/bin/bash
# GPU box: what THIS box does on the headline configuration - the bench line (HIP-event kernel time, rocm-smi W / sclk under the timed
# loop) and one PMC pass (effective clock from GRBM_GUI_ACTIVE, MFMA busy). Run on several boxes (one gpurun call each) to see what
# varies from box to box: profiles/r04_power_ceiling.md section (iv).
set -u
TAG=${1:-box}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-sweep --no-cpu-baseline --no-fp8 --no-denoise --no-head-dims"
{ hostname; rocm-smi --showserial --showuniqueid 2>/dev/null | grep -iE "serial|unique"; rocm-smi --showmaxpower --showtemp 2>/dev/null | grep -iE "power|junction|edge"; } > $OUT/box_id.txt 2>&1
$B --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --kernel-include-regex "la_fwd" --output-format csv -d $OUT/pmc -o p -- $B --no-verify --no-power --steps 5 --warmup 2 > $OUT/pmc.log 2>&1
$B --steps 20 --warmup 5 > $OUT/bench_line_2.json 2>> $OUT/bench_line.err
find $OUT -name "*.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
python - <<PY
import csv, glob, json, collections
csv.field_size_limit(1 << 30)
out = {"box": open("$OUT/box_id.txt").read().split("\n")[:6]}
for i, f in enumerate(("$OUT/bench_line.json", "$OUT/bench_line_2.json")):
    try:
        b = json.loads([l for l in open(f) if l.startswith("{")][-1])
        out[f"bench_{i}"] = {"value": b["value"], "kernel_ms": b["roofline"]["kernel_ms"], "frac": b["roofline"]["frac"], "power": b.get("power"), "verified": (b.get("verified") or {}).get("ok")}
    except Exception as e:
        out[f"bench_{i}"] = {"error": repr(e)}
acc, n, dur = collections.defaultdict(float), collections.Counter(), []
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "la_fwd_x64_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob("$OUT/pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "la_fwd_x64_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
a = {k: acc[k] / n[k] for k in acc}
if dur and a.get("GRBM_GUI_ACTIVE"):
    ms = sum(dur) / len(dur); cyc = a["GRBM_GUI_ACTIVE"] / 8
    out["pmc"] = {"kernel_ms": round(ms, 3), "clock_ghz": round(cyc / ms / 1e6, 3), "mfma_busy": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, 4),
                  "waves_issuing": round(a.get("SQ_ACTIVE_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1), 4),
                  "waves_issue_stalled": round(a.get("SQ_WAIT_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1), 4)}
json.dump(out, open("$OUT/box_probe.json", "w"), indent=1)
print(json.dumps(out))
PY
