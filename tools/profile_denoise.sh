#!/bin/bash
# PMC passes over the 50-step denoise workload (real, fragmented skip lists) — last-step kernel statistics.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof_denoise; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/denoise_bench.py --alpha 6 --sink-gain 0.5 --tag prof --targets 0.42 --iters 1 --calib-heads 1"
# thr for 42% found earlier: fix it by making bisection start at it (lo=hi) is not supported; one iteration lands near -30 -> low sparsity.
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/hit -o hit -- $CMD > $OUT/hit.log 2>&1
python - <<'PY'
import csv, collections, os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof_denoise"
for name in ["fetch","hit"]:
    rows=[r for r in csv.DictReader(open(f"{out}/{name}/{name}_counter_collection.csv")) if "la_fwd" in r["Kernel_Name"] and int(r["Grid_Size"])==23640*256]
    byc=collections.defaultdict(list)
    for r in rows: byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"]), "Lb1" if "<true>" in r["Kernel_Name"] else "Lb0"))
    for c,v in byc.items():
        v.sort()
        print(name, c, "n=",len(v), "first(dense?)", v[0][1:], "last3", [x[1:] for x in v[-3:]])
PY
grep "^H=" $OUT/fetch.log
