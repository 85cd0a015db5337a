#!/usr/bin/env python
"""CPU: gpurun_out/{r04_power, r04_m16, r04c, r04_box*} (tools/power_ceiling.sh, tools/price_levers.sh, tools/box_probe.sh)
-> profiles/r04_power_ceiling.{json,md}: the evidence VERDICT r3 (next-round item 2) asked to see under profiles/.
usage: python tools/summarize_power_ceiling.py"""
import collections
import csv
import glob
import json
import os
import re

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
res = {}

# (i) (ii) MFMA-only loops
mp = json.load(open(os.path.join(G, "r04_power", "mfma_power.json")))
rows = []
for c in mp["cases"]:
    row = {k: c.get(k) for k in ("case", "instr", "waves_per_simd", "operands", "ms", "tflops", "socket_w", "sclk_mhz")}
    d = os.path.join(G, "r04_power", "mfma_pmc_" + c["case"])
    acc, n, dur = collections.defaultdict(float), collections.Counter(), []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if dur and n["GRBM_GUI_ACTIVE"]:
        ms = sum(dur) / len(dur)
        cyc = acc["GRBM_GUI_ACTIVE"] / n["GRBM_GUI_ACTIVE"] / 8
        row["pmc_pass"] = {"kernel_ms": round(ms, 3), "clock_ghz": round(cyc / ms / 1e6, 3),
                           "mfma_busy": round(acc["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 4)}
    row["frac_of_2500"] = round(c["tflops"] / 2500.0, 4) if c.get("tflops") else None
    row["pj_per_flop_socket"] = round(c["socket_w"] / c["tflops"], 4) if c.get("tflops") and c.get("socket_w") else None
    rows.append(row)
res["mfma_only"] = rows


# (iii) ablations + levers: cycles / clock / MFMA busy (pmc_cycles.sh) and W (variant_power.py), two sessions
def pmcc(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(\S+)\s+([\d.]+) ms\s+clk ([\d.]+) GHz\s+Mcyc\s+([\d.]+)\s+mfma_util ([\d.]+)\s+quads/wave-step: total (\d+) active (\d+) issue-stall (\d+) parked (\d+) valu (\d+)", ln)
        if m:
            out[m.group(1)[2:] if m.group(1).startswith("x_") else m.group(1)] = {
                "ms_under_pmc": float(m.group(2)), "clock_ghz": float(m.group(3)), "mcycles": float(m.group(4)), "mfma_busy": float(m.group(5)),
                "quads_per_wave_step": {"total": int(m.group(6)), "issuing": int(m.group(7)), "issue_stalled": int(m.group(8)),
                                        "parked": int(m.group(9)), "valu": int(m.group(10))}}
    return out


variants = {}
for sess in ("r04_power", "r04_m16"):
    pc = pmcc(os.path.join(G, sess, "pmc_cycles.txt"))
    vp = {r["variant"]: r for r in json.load(open(os.path.join(G, sess, "variant_power.json")))["rows"]}
    for name in list(pc) + [k for k in vp if k not in pc]:
        e = variants.setdefault(name if sess == "r04_power" or name not in variants else name + " (2nd session)", {"session": sess})
        e.update(pc.get(name, {}))
        if name in vp:
            e.update({"ms": vp[name]["ms"], "tflops_dense_equiv": vp[name]["tflops_dense_equiv"], "socket_w": vp[name]["socket_w"],
                      "sclk_mhz_smi": vp[name]["sclk_mhz"]})
res["variants"] = variants
res["variants_what"] = ("dense S=16384 H=80 bf16 d128 (tools/abl_bench.py shape); ms / W / sclk: tools/variant_power.py (median of 20 launches by HIP events, "
                        "rocm-smi under ~2.5 s of queued launches, 2 interleaved rounds); cycles, effective clock, MFMA busy: tools/pmc_cycles.sh")
lv = os.path.join(G, "r04c", "levers.txt")
if os.path.exists(lv):
    res["levers_headline_shape"] = [ln.rstrip() for ln in open(lv) if re.match(r"^(tree|base|dotsum|mfmasum|hs8|hs4)\s", ln)]
    res["levers_other_head_dims"] = [ln.rstrip().replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory", "").strip()
                                     for ln in open(lv) if re.search(r"TF", ln) and re.match(r"^(d64|D=96)", ln.strip())]
m16 = os.path.join(G, "r04_m16")
# (iv) boxes
boxes = []
for d in [os.path.join(G, "r04_power", "box")] + sorted(glob.glob(os.path.join(G, "r04_box*"))):
    f = os.path.join(d, "box_probe.json")
    if os.path.exists(f):
        b = json.load(open(f))
        uid = next((x.split(":")[-1].strip() for x in b["box"] if "Unique ID:" in x), "?")
        boxes.append({"unique_id": uid, "value_tflops": b["bench_0"].get("value"), "value_tflops_2nd_run": b.get("bench_1", {}).get("value"),
                      "kernel_ms": b["bench_0"].get("kernel_ms"), "roofline_frac": b["bench_0"].get("frac"),
                      "socket_w": (b["bench_0"].get("power") or {}).get("socket_w"), "sclk_mhz_smi": (b["bench_0"].get("power") or {}).get("sclk_mhz"),
                      "effective_clock_ghz_pmc": (b.get("pmc") or {}).get("clock_ghz"), "mfma_busy": (b.get("pmc") or {}).get("mfma_busy"),
                      "waves_issuing": (b.get("pmc") or {}).get("waves_issuing"), "verified": b["bench_0"].get("verified")})
res["boxes"] = boxes
json.dump(res, open(os.path.join(ROOT, "profiles", "r04_power_ceiling.json"), "w"), indent=1)

md = ["# r04 — the power ceiling of the bf16 kernel: evidence (generated by tools/summarize_power_ceiling.py)", "",
      "VERDICT r3 asked for four things under `profiles/`: (i) the MFMA-only loop with random vs all-zero operands, (ii) the same for",
      "`v_mfma_f32_16x16x32_bf16`, (iii) the ablation table of HISTORY.md 4.2 regenerated on the current body, (iv) the headline on >= 3 boxes",
      "with clock, watts and MFMA busy. Then two energy levers priced (row sums off the vector unit; per-half skipping). All of it is here;",
      "(i)-(iii) are ONE session on one box (`gpurun_out/r04_power`), the re-spaced 16x16x32 stand-ins a second one (`r04_m16`).", "",
      "## (i) (ii) The matrix pipe alone: `tools/mfma_power_bench.py`", "",
      "256 workgroups x 4 (8) waves, 64 independent MFMAs per loop iteration over 8 (16) accumulators, A / B fragments rotating; nothing else in the loop.", "",
      "| instruction | waves / SIMD | operands | TFLOP/s | of 2.5 PF | socket W | sclk (smi) MHz | pJ / FLOP (socket) | PMC pass: clock GHz, MFMA busy |", "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    p = r.get("pmc_pass")
    md.append(f"| {r['instr']} | {r['waves_per_simd']} | {r['operands']} | {r['tflops']} | {r['frac_of_2500']} | {r['socket_w']} | {r['sclk_mhz']} | {r['pj_per_flop_socket']} | "
              + (f"{p['clock_ghz']}, {100 * p['mfma_busy']:.1f} %" if p else "") + " |")
r_rand = next(r for r in rows if r["case"] == "32x32x16_w1_rand")
r_zero = next(r for r in rows if r["case"] == "32x32x16_w1_zero")
md += ["", f"Reading: with all-zero operands the pipe runs at the nameplate ({r_zero['sclk_mhz'] / 1000:.2f} GHz, {r_zero['frac_of_2500']:.2f} of 2.5 PF) at two thirds of the power cap; with N(0,1) bf16 operands the",
       f"SAME instruction stream is throttled to {r_rand['sclk_mhz'] / 1000:.2f} GHz = {r_rand['frac_of_2500']:.2f} of the nameplate. The ceiling of ANY bf16 kernel on random data on this part is therefore ~0.71-0.79,",
       "set by operand toggling, not by the instruction stream; the guide's 2495 TF figure is the zero-toggle case. `16x16x32` costs 8 % less energy per FLOP than",
       "`32x32x16` on random data (half the accumulator traffic per FLOP) and reaches 0.79.", "",
       "## (iii) Ablations and priced levers on the current body", "", res["variants_what"], "",
       "| variant | ms | dense-equiv TFLOP/s | socket W | sclk smi | ms under PMC | effective clock GHz | MFMA busy | quads per wave-step: total / issuing / issue-stalled / parked |", "|---|---|---|---|---|---|---|---|---|"]
for name, e in variants.items():
    q = e.get("quads_per_wave_step", {})
    md.append(f"| {name} | {e.get('ms', '')} | {e.get('tflops_dense_equiv', '')} | {e.get('socket_w', '')} | {e.get('sclk_mhz_smi', '')} | {e.get('ms_under_pmc', '')} | "
              f"{e.get('clock_ghz', '')} | {e.get('mfma_busy', '')} | {q.get('total', '')} / {q.get('issuing', '')} / {q.get('issue_stalled', '')} / {q.get('parked', '')} |")
md += ["", "Variants (tools/asm_variants.py options of gen_fwd_x64.py; all but `dotsum` compute wrong results and only price a component):",
       "`nobar` no per-step barrier; `nobar2` + no V^T waits, no vmcnt drain; `nodma` no LDS-DMA; `nosoftmax` no exp / sums / pack / row max; `mfmaonly` MFMAs + the loop skeleton;",
       "`dotsum` row sums of the rounded P by `v_dot2c_f32_bf16` (-32 VALU instructions per step, results kept: max err equal to base);",
       "`mfmasum` row sums from the matrix pipe (-64 `v_add_f32`, +8 MFMAs per step); `hs8` / `hs4` every wave sits out one step in 8 / 4 (per-128-row-half lists walked as their union);",
       "`m16qk` / `m16pv` / `m16both` every 32x32x16 MFMA of that GEMM replaced by two 16x16x32 on the same operands; `m16s` ... the same with the two re-spaced inside the gap.", ""]
if res.get("levers_headline_shape"):
    md += ["Same levers on the headline shape (S = 75 600, H = 40; dense / imposed 42 % / 77 %; `tools/ab.py`, 3 interleaved rounds, session `r04c`):", "", "```"] + res["levers_headline_shape"] + ["```", ""]
if res.get("levers_other_head_dims"):
    md += ["`dotsum` on the issue-bound head dims (base / dotsum, twice, interleaved):", "", "```"] + res["levers_other_head_dims"] + ["```", ""]
md += ["## (iv) The headline on four boxes (`tools/box_probe.sh`, one gpurun call each)", "",
       "| GPU unique id | TFLOP/s (2nd run) | kernel ms | roofline frac | socket W | sclk smi MHz | effective clock GHz (PMC) | MFMA busy | waves issuing |", "|---|---|---|---|---|---|---|---|---|"]
for b in boxes:
    md.append(f"| {b['unique_id']} | {b['value_tflops']} ({b['value_tflops_2nd_run']}) | {b['kernel_ms']} | {b['roofline_frac']} | {b['socket_w']} | {b['sclk_mhz_smi']} | "
              f"{b['effective_clock_ghz_pmc']} | {b['mfma_busy']} | {b['waves_issuing']} |")
md += ["", "MFMA busy and the wave-state split are IDENTICAL on every box (the kernel does the same cycles everywhere); TFLOP/s follows the effective clock 1 : 1.",
       "The clock rocm-smi samples is 4-6 % above the effective clock of the kernel (GRBM_GUI_ACTIVE / time) and moves independently of it from box to box, which is why round 3's",
       "driver box looked 'same clock, 5.5 % slower': its smi sample (1618 MHz) said nothing about its effective clock (53.4 ms at 79.3 % busy = 1.53 GHz).", ""]
open(os.path.join(ROOT, "profiles", "r04_power_ceiling.md"), "w").write("\n".join(md))
print("\n".join(md))
