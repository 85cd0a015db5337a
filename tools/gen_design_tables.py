#!/usr/bin/env python
"""Rewrites the measured tables of MEASUREMENTS.md from the committed evidence of ONE run (profiles/<tag>_*; tools/evidence.sh ->
tools/summarize_evidence.py): every block between `<!-- GEN:name -->` and `<!-- /GEN:name -->` is regenerated, nothing else is touched.

    python tools/gen_design_tables.py [tag]            rewrite MEASUREMENTS.md in place (default tag: the newest profiles/rNN_bench_line.json)
    python tools/gen_design_tables.py [tag] --check    exit 1 if MEASUREMENTS.md is not what the profiles say (tests/test_docs.py)

The prose around the blocks may interpret the numbers; it must not restate them from memory."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def load(name):
    p = os.path.join(PROF, name)
    return json.load(open(p)) if os.path.exists(p) else None


def f(x, nd=1):
    return "n/a" if x is None else f"{x:.{nd}f}"


def blocks(tag):
    b = load(f"{tag}_bench_line.json")
    rp, rp8 = load(f"{tag}_rocprof_summary.json"), load(f"{tag}_fp8_rocprof_summary.json")
    tr = load(f"{tag}_traffic_vs_sparsity.json")
    ss = load("r03_sched_sweep.json")
    ss4 = load("r04_sched_sweep.json")
    if ss and ss4:                                    # round 4 added the gang-scheduled A/B build (another box: compare it with ITS tree row)
        ss = {"rows": ss["rows"] + [dict(x, variant=x["variant"] + " (r04 session)") for x in ss4["rows"] if x["variant"] in ("gang", "tree")]}
    out = {}
    if b:
        r, pw = b["roofline"], b.get("power") or {}
        rows = ["| configuration | ms / step | executed TFLOP/s | of MFMA peak | notes |", "|---|---|---|---|---|",
                f"| **bf16 C3 imposed 42 % (headline)** | {b['ms_per_step']} | **{f(b['value'])}** | **{f(r['frac'], 3)}** | kernel {r['kernel_ms']} ms by HIP events; "
                f"{pw.get('socket_w', 'n/a')} W of {pw.get('cap_w', 'n/a')} W at {f((pw.get('sclk_mhz') or 0) / 1000, 2)} GHz; verified {b.get('verified', {}).get('ok')} "
                f"(max err {b.get('verified', {}).get('max_err')}, LSE {b.get('verified', {}).get('max_err_lse')}) |"]
        sw = b.get("sweep") or []
        if sw:
            rows.append("| bf16 C3 sweep " + " / ".join(f"{100 * s['sparsity']:.0f} %" for s in sw) + " | " + " / ".join(f(s["ms"]) for s in sw) + " | " +
                        " / ".join(f(s["executed_tflops"], 0) for s in sw) + " | | t(s)/t(0) = " + " / ".join(f(s["t_over_t0"], 3) for s in sw) +
                        " (reference " + " / ".join(f(s["reference_t_over_t0"], 3) for s in sw) + ") |")
        c1 = b.get("config1_dense_s32768")
        if c1 and "tflops" in c1:
            rows.append(f"| bf16 C2 dense S = 32 768, H = 40 (configs[1]) | {c1['ms']} | {f(c1['tflops'])} | {f(c1['frac_of_mfma_peak'], 3)} | verified {c1['verified']['ok']} |")
        f8 = b.get("fp8") or {}
        if f8.get("value"):
            p8 = f8.get("power") or {}
            note = f"{p8.get('socket_w', 'n/a')} W at {f((p8.get('sclk_mhz') or 0) / 1000, 2)} GHz; verified {f8.get('verified', {}).get('ok')} (max err {f8.get('verified', {}).get('max_err')})"
            if f8.get("exact_exp"):                    # lines of rounds 3-5: the encoded form was the default, the reference's arithmetic behind a flag
                note += f"; LA_FLAG_EXACT_EXP on the same box: {f(f8['exact_exp']['value'])} TFLOP/s"
            if f8.get("exact_rowsum"):
                note += (f"; **LA_FLAG_EXACT_ROWSUM (the reference's arithmetic): {f(f8['exact_rowsum']['value'])} TFLOP/s = {f(f8['exact_rowsum']['frac'], 3)}**, "
                         f"verified {(f8['exact_rowsum'].get('verified') or {}).get('ok')}")
            if f8.get("mfma_rowsum"):                  # round 6: the default IS the reference's arithmetic; the two opt-in forms beside it
                note += f"; opt-in LA_FLAG_FP8_MFMA_ROWSUM on the same box: {f(f8['mfma_rowsum']['value'])} TFLOP/s = {f(f8['mfma_rowsum']['frac'], 3)}"
            if f8.get("encoded_p"):
                note += (f"; opt-in LA_FLAG_FP8_ENCODED_P (block-scaled encoding of P, not the reference's arithmetic): {f(f8['encoded_p']['value'])} TFLOP/s = "
                         f"{f(f8['encoded_p']['frac'], 3)}, verified {(f8['encoded_p'].get('verified') or {}).get('ok')}")
            form = "default: the reference's arithmetic" if f8.get("mfma_rowsum") or f8.get("encoded_p") else "default block-scaled encoding of P"
            rows.append(f"| **fp8 C3 imposed 42 %** ({form}) | {f8['ms_per_step']} | **{f(f8['value'])}** | **{f(f8['roofline']['frac'], 3)}** of 5 PF | {note} |")
        for run in (b.get("other_head_dims") or {}).get("runs", []):
            rows.append(f"| bf16 head_dim {run['head_dim']}, dense S = 16 384 H = 40, tiles {run['tiles'][0]} x {run['tiles'][1]} | {run['ms']} | {f(run['tflops'])} | "
                        f"{f(run['frac_of_mfma_peak'], 3)} | verified {run['verified']['ok']} |")
        cb = b.get("cpu_baseline")
        if cb:
            rows.append(f"| CPU baseline: oracle port, {cb['cores']} host threads, bounded sample | | {cb['value']} | | eager torch CPU (the reference's path): "
                        f"{(cb.get('eager_torch') or {}).get('value')} |")
        out["headline"] = rows
        dn = b.get("denoise50")
        if dn and "runs" in dn:
            dv0 = dn.get("dense_vs_sweep0") or {}
            rows = [f"Dense kernel on the same tensors: {dn['dense_ms_per_step']} ms per step ({dn.get('dense_how', 'median of three launches')}; min / max "
                    f"{dn.get('dense_ms_min_max')}); against the dense point of the imposed-list sweep of the same run ({dv0.get('sweep0_kernel_ms')} ms, random q / k / v): "
                    f"ratio {dv0.get('ratio')}; the same random-tensor launch timed inside the loop: {dv0.get('dense_ms_on_the_sweeps_random_tensors_inside_the_loop')} ms "
                    f"(ratio {dv0.get('same_tensors_ratio')}: the launch context, an observation of either sign), structured / random inside the loop: {dv0.get('structured_over_random_inside_the_loop')}.", "",
                    "| target | thr (log2) | last-step sparsity (within 1 % of target) | last step ms | dense ms, this run | t / t_dense | ideal (1 - s) | reference t / t0 at the target | "
                    "50 steps ms | speed-up vs 50 dense calls | step-49 check (rows, max err / tol, LSE, write rows checked / bad, max ranges per row) | mean / max abs error vs dense output |",
                    "|---|---|---|---|---|---|---|---|---|---|---|---|"]
            for x in dn["runs"]:
                v = x.get("verified") or {}
                chk = (f"ok={v.get('ok')}: {v.get('rows')} rows, {v.get('max_err')} / {v.get('tol')}, {v.get('max_err_lse')}, {v.get('write_rows_checked')} / "
                       f"{v.get('write_rows_bad')}, {v.get('max_ranges_per_row')}") if v else "n/a"
                rows.append(f"| {x['target']} | {x['thr']} | {100 * x['sparsity_last_step']:.1f} % ({x.get('within_1pct_of_target')}) | {x['ms_last_step']} | {x.get('dense_ms_this_run')} | "
                            f"{x['t_last_over_dense']} | {x['ideal_1_minus_s']} | {x.get('reference_t_over_t0_at_target')} | "
                            f"{x['total_ms_50_steps']} | {x['speedup_vs_dense_50_steps']}x | {chk} | {x['mean_abs_err_vs_dense']} / {x['max_abs_err_vs_dense']} |")
            out["denoise50"] = rows
    if b:
        hv = b.get("half_vote") or {}
        if "at_fixed_thresholds" in hv:
            rows = ["| thr (target of the 256-row vote) | 50 steps ms: 256-row / 128-row (half) vote | ratio | last-step sparsity 256 / 128 | mean abs error vs dense 256 / 128 | max abs error 128 |",
                    "|---|---|---|---|---|---|"]
            tile = {r["target"]: r for r in (b.get("denoise50") or {}).get("runs", [])}
            for r in hv["denoise50"]["runs"]:
                t, x = tile.get(r["target"], {}), hv["at_fixed_thresholds"].get(r["target"], {})
                rows.append(f"| {r['thr']} ({r['target']}) | {t.get('total_ms_50_steps')} / {r['total_ms_50_steps']} | {x.get('total_ms_half_over_tile256')} | "
                            f"{100 * t.get('sparsity_last_step', 0):.1f} % / {100 * r['sparsity_last_step']:.1f} % | {t.get('mean_abs_err_vs_dense')} / {r['mean_abs_err_vs_dense']} | "
                            f"{r['max_abs_err_vs_dense']} |")
            sw = {round(s["sparsity"], 2): s for s in b.get("sweep") or []}
            rows += ["", "Imposed lists in the 128-row geometry (same box, same process) against the sweep of the 256-row vote: " +
                     "; ".join(f"{100 * e['sparsity']:.0f} %: {e['ms']} ms ({e['executed_tflops']} TFLOP/s, {e['frac_of_mfma_peak']} of the peak) against "
                               f"{(sw.get(round(e['sparsity'], 2)) or {}).get('kernel_ms')} ms" for e in hv.get("imposed_lists", [])) + "."]
            out["half_vote"] = rows
        sv = b.get("denoise50_survey") or {}
        if "runs" in sv:
            rows = ["| target | thr (log2, bisected over [-20, 0)) | last-step sparsity | last step ms | 50 steps ms | mean / max abs error vs dense output |", "|---|---|---|---|---|---|"]
            rows += [f"| {r['target']} | {r['thr']} | {100 * r['sparsity_last_step']:.1f} % | {r['ms_last_step']} | {r['total_ms_50_steps']} | {r['mean_abs_err_vs_dense']} / {r['max_abs_err_vs_dense']} |"
                     for r in sv["runs"]]
            out["denoise50_survey"] = rows
        f8 = b.get("fp8") or {}
        if any(f"head_dim_{hd}" in f8 for hd in (64, 96, 192, 256)):
            rows = ["| head_dim (tiles) | reference arithmetic (default): ms, TFLOP/s, of 5 PF | `LA_FLAG_FP8_MFMA_ROWSUM` | `LA_FLAG_FP8_ENCODED_P` | sampled rows ok |", "|---|---|---|---|---|"]
            for hd in (64, 96, 192, 256):
                x = f8.get(f"head_dim_{hd}") or {}
                fm = x.get("forms") or {}
                if not fm:
                    continue
                cell = lambda k: (f"{fm[k]['ms']} ms, {fm[k]['tflops']}, {fm[k]['frac_of_mfma_peak']}" if k in fm else "n/a")      # noqa: E731
                rows.append(f"| {hd} ({x['tiles'][0]} x {x['tiles'][1]}) | {cell('reference_arithmetic')} | {cell('mfma_rowsum')} | {cell('encoded_p')} | "
                            f"{all(v['verified']['ok'] for v in fm.values())} |")
            out["fp8_head_dims"] = rows
        jr = b.get("joint_recipe") or {}
        if "v2v" in jr:
            rows = ["| call | ms | TFLOP/s (GB/s for the merges) | of the MFMA (HBM) peak | launches timed |", "|---|---|---|---|---|"]
            for n in ("t2t", "t2v", "v2t", "v2v"):
                x = jr[n]
                rows.append(f"| {n}" + (f" ({x['num_splits']} key splits)" if x.get("num_splits", 1) > 1 else "") + f" | {x['ms']} | {x['tflops']} | {x['frac_of_mfma_peak']} | {x['launches_timed']} |")
            for n in ("merge_text", "merge_video"):
                x = jr[n]
                rows.append(f"| {n} | {x['ms']} | {x['gb_per_s']} GB/s | {x['frac_of_hbm_peak']} of HBM | {x['launches_timed']} |")
            rows.append(f"| **everything but v2v** | **{jr['everything_but_v2v_ms']}** | | **{100 * jr['everything_but_v2v_over_v2v']:.1f} % of the v2v call** | |")
            out["joint_recipe"] = rows
    for name, d in (("rocprof_bf16", rp), ("rocprof_fp8", rp8)):
        if not d:
            continue
        dv, pmc = d["derived"], d["pmc_per_launch"]
        ws = dv.get("wave_state_fracs_parked_stalled_issuing") or [None] * 3
        steps = None
        rows = [f"- kernel average (rocprofv3 --kernel-trace --stats): **{f(d['kernel_avg_ms'], 3)} ms**; effective clock {f(dv.get('clock_GHz'), 2)} GHz; "
                f"**MFMA busy {f(100 * dv.get('mfma_util', 0))} %**; LDS busy {f(100 * dv.get('lds_util', 0))} %, bank conflicts {pmc.get('SQ_LDS_BANK_CONFLICT', 0):.0f}",
                f"- waves: {f(100 * (ws[2] or 0))} % issuing / {f(100 * (ws[1] or 0))} % stalled / {f(100 * (ws[0] or 0))} % parked; "
                f"VALU active {f(100 * pmc.get('SQ_ACTIVE_INST_VALU', 0) / max(pmc.get('SQ_WAVE_CYCLES', 1), 1))} % of wave cycles "
                f"({pmc.get('SQ_INSTS_VALU', 0) / 1e9:.2f} G VALU instructions per launch)",
                f"- L2 hit {f(100 * dv.get('l2_hit_rate', 0))} %; L2 fills + writes {f(dv.get('hbm_bytes_per_launch', 0) / 1e9, 2)} GB per launch "
                f"({f(dv.get('hbm_GBps'), 0)} GB/s); resources {d.get('kernel_resources')}"]
        out[name] = rows
    if tr:
        rows = ["| list | sparsity s | ms | executed TFLOP/s | L2 fills GB | written GB | L2 hit | MFMA busy | clock GHz | fabric GB/s |", "|---|---|---|---|---|---|---|---|---|---|"]
        for x in tr["rows"]:
            rows.append(f"| {x['list']} | {x['sparsity']:.3f} | {x['kernel_ms_under_pmc']:.2f} | {x['executed_tflops']:.0f} | {x['hbm_read_GB']:.1f} | {x['hbm_write_GB']:.2f} | "
                        f"{100 * (x['l2_hit_rate'] or 0):.1f} % | {100 * (x['mfma_util'] or 0):.1f} % | {(x['clock_GHz'] or 0):.2f} | {(x['hbm_GBps'] or 0):.0f} |")
        out["traffic"] = rows
        by = {x["list"]: x for x in tr["rows"]}
        if all(k in by for k in ("imposed 0.42", "imposed 0.77", "real -4.22", "real -2.46")):
            g44 = 100 * (by["real -4.22"]["executed_tflops"] / by["imposed 0.42"]["executed_tflops"] - 1)
            g78 = 100 * (by["real -2.46"]["executed_tflops"] / by["imposed 0.77"]["executed_tflops"] - 1)
            out["real_gap"] = [f"Against the banded list of nearest sparsity the real lists are at {g44:+.1f} % (44 %) and {g78:+.1f} % (78 %) executed TFLOP/s in "
                               f"this session, at {by['real -4.22']['clock_GHz']:.2f} / {by['real -2.46']['clock_GHz']:.2f} GHz against "
                               f"{by['imposed 0.42']['clock_GHz']:.2f} / {by['imposed 0.77']['clock_GHz']:.2f} (round 2: -6 % / -10 %)."]
    pf = os.path.join(PROF, f"{tag}_fp8_p_forms.txt")
    if os.path.exists(pf):
        txt = open(pf).read()
        tf = {m.group(1): (float(m.group(2)), float(m.group(3))) for m in re.finditer(r"P form (\w+): .*?s=0.0: [0-9.]+ ms (\d+) TF \| s=0.42: [0-9.]+ ms (\d+) TF", txt)}
        err = {(m.group(1), m.group(2)): (m.group(3), m.group(4), m.group(5)) for m in
               re.finditer(r"(default|exp|rowsum)\s+real lists thr (-[0-9.]+): max\|O-ref\| ([0-9.]+) \(tol ([0-9.]+)\)\s+max\|LSE-ref\| ([0-9.]+)", txt)}
        if all(k in tf for k in ("default", "exp", "rowsum")):
            base = tf["rowsum"][1]
            rows = ["| form of P | C3 dense / imposed 42 %, TFLOP/s | vs the reference's form | real step-49 lists thr -4.22 / -2.46: max abs O error (bound), max abs LSE error |", "|---|---|---|---|"]
            for key, name in (("rowsum", "**default** (the reference's arithmetic; `LA_FLAG_EXACT_ROWSUM` until round 5)"), ("exp", "`LA_FLAG_FP8_MFMA_ROWSUM` (`LA_FLAG_EXACT_EXP` until round 5)"),
                              ("default", "`LA_FLAG_FP8_ENCODED_P` (block-scaled log-linear encoding; the default until round 5)")):
                e1, e2 = err.get((key, "-4.22")), err.get((key, "-2.462"))
                es = f"{e1[0]} ({e1[1]}), {e1[2]} / {e2[0]} ({e2[1]}), {e2[2]}" if e1 and e2 else "n/a"
                rows.append(f"| {name} | {tf[key][0]:.0f} / {tf[key][1]:.0f} | {100 * (tf[key][1] / base - 1):+.1f} % | {es} |")
            out["fp8_forms"] = rows
    if ss:
        rows = ["| list | variant (chunk C, heads interleaved G) | ms | executed TFLOP/s | L2 fills GB | L2 hit | clock GHz | MFMA busy |", "|---|---|---|---|---|---|---|---|"]
        for thr in sorted({x["thr"] for x in ss["rows"]}):
            for x in sorted((x for x in ss["rows"] if x["thr"] == thr), key=lambda x: x["ms"]):
                rows.append(f"| thr {thr} ({100 * x['sparsity']:.1f} %) | {x['variant']} | {x['ms']:.2f} | {x['executed_tflops']:.0f} | {x['l2_fills_GB']:.1f} | "
                            f"{100 * x['l2_hit']:.1f} % | {x['clock_GHz']:.2f} | {100 * x['mfma_busy']:.1f} % |")
        out["sched_sweep"] = rows
    return out


def latest_tag():
    tags = sorted(f[:3] for f in os.listdir(PROF) if re.fullmatch(r"r\d\d_bench_line\.json", f))
    return tags[-1] if tags else "r04"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else latest_tag()
    path = os.path.join(ROOT, "MEASUREMENTS.md")
    text = open(path).read()
    new = text
    for name, rows in blocks(tag).items():
        pat = re.compile(rf"(<!-- GEN:{name} -->\n).*?(<!-- /GEN:{name} -->)", re.S)
        if pat.search(new):
            new = pat.sub(lambda m: m.group(1) + "\n".join(rows) + "\n" + m.group(2), new)
    if "--check" in sys.argv:
        if new != text:
            print("MEASUREMENTS.md tables are stale: run python tools/gen_design_tables.py", tag)
            sys.exit(1)
        print("MEASUREMENTS.md tables match profiles/" + tag + "_*")
        return
    open(path, "w").write(new)
    print("MEASUREMENTS.md: regenerated blocks", sorted(blocks(tag)), "from", tag)


if __name__ == "__main__":
    main()
