"""CPU: the measured tables of MEASUREMENTS.md are the ones tools/gen_design_tables.py writes from the committed evidence of ONE run
(the newest profiles/rNN_*): prose may interpret the numbers, it may not drift from them. DESIGN.md is the short
current-state document (<= 300 lines, <= 120 columns), HISTORY.md the record of rounds 1-4."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_tables_are_generated_from_the_committed_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_design_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(os.path.join(ROOT, "MEASUREMENTS.md")).read()
    for name in ("headline", "rocprof_bf16", "rocprof_fp8", "traffic", "real_gap", "fp8_forms", "sched_sweep", "denoise50", "denoise50_survey",
                 "half_vote", "joint_recipe", "fp8_head_dims"):
        body = text.split(f"<!-- GEN:{name} -->")[1].split(f"<!-- /GEN:{name} -->")[0]
        assert body.strip(), f"generated block {name} is empty"


def test_committed_traffic_figure_belongs_to_the_committed_kernel_sources():
    """bench.py takes roofline.traffic from profiles/pmc_summary*.json only when the summary was measured on the kernel sources of the
    run; the committed summaries must be the ones of the tree (otherwise the driver's line says traffic: null)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    csrc = os.path.join(ROOT, "liteattention_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "la_fwd_x64_body.inc")):
        import pytest
        pytest.skip("generated bodies not built yet")
    sha = bench.kernel_source_hash()
    for fn in ("pmc_summary.json", "pmc_summary_fp8.json"):
        assert json.load(open(os.path.join(ROOT, "profiles", fn)))["kernel_source_sha16"] == sha, fn


def test_tools_readme_indexes_exactly_the_scripts_that_exist():
    """tools/ had grown to 45 scripts with overlapping purposes. Every script in tools/ (and tools/debug/) is named
    in tools/README.md, and every script the README names exists."""
    import re
    tools = os.path.join(ROOT, "tools")
    readme = open(os.path.join(tools, "README.md")).read()
    have = {f for f in os.listdir(tools) if f.endswith((".py", ".sh", ".hip"))} | \
           {"debug/" + f for f in os.listdir(os.path.join(tools, "debug")) if f.endswith((".py", ".sh", ".hip"))}
    named = set(re.findall(r"`((?:debug/)?[a-z0-9_]+\.(?:py|sh|hip))", readme))
    assert have - named == set(), f"scripts tools/README.md does not index: {sorted(have - named)}"
    assert named - have == set(), f"tools/README.md names scripts that do not exist: {sorted(named - have)}"


def test_design_md_is_the_short_current_state_document():
    """DESIGN.md = current state, at most 300 lines of at most 120 columns; the history lives in HISTORY.md."""
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().splitlines()
    assert len(lines) <= 300, len(lines)
    long = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 120]
    assert not long, long[:5]
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md")) and os.path.exists(os.path.join(ROOT, "MEASUREMENTS.md"))
