"""GPU: ``bench.py``'s N > 1 branch — process group, the trial step of the overlapped all-gather, the agreement all-reduce, the
``multi_gpu`` record, ``verified.ok_all_ranks`` — executed end to end on a ONE-rank RCCL group (``LA_BENCH_FORCE_DIST=1``), so the
driver's 2/4/8-GPU runs do not meet a code path that has never run. The partitioning it drives
is SURVEY.md §8(e) (heads sharded, one all-gather of O); the reference has no counterpart (README.md:199-250 is a caller recipe)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(extra, env_extra):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--seqlen", "8192", "--heads", "8", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-denoise", "--no-head-dims", "--no-fp8", "--no-power"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("windows", [3, 1])
def test_bench_multi_gpu_branch_on_a_one_rank_rccl_group(windows):
    env = {"LA_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0",
           "LOCAL_RANK": "0", "WORLD_SIZE": "1"}
    r = _run_bench(["--gpus", "1", "--overlap-windows", str(windows)], env)
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["scaling"] == "strong" and r["value"] > 0
    mg = r["multi_gpu"]
    assert mg["rccl_world_size"] == 1 and mg["backend"] == "nccl"
    assert mg["overlapped_form_kept"] is (windows > 1) and "overlap_note" not in r["config"]
    assert mg["overlap_windows"] == (windows if windows > 1 else 1) or mg["overlap_windows"] >= 1
    assert len(mg["kernel_ms_per_rank"]) == 1 and mg["kernel_ms_max"] >= mg["kernel_ms_min"] > 0
    assert r["verified"]["ok"] and r["verified"]["ok_all_ranks"]
    assert "all-gather" in r["config"]["parallelism"] and r["config"]["parallelism"].startswith("heads sharded 1x8")
    assert r["roofline"]["bound"] == "mfma" and 0 < r["roofline"]["frac"] < 1


def test_bench_self_launches_its_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` with NO RANK / WORLD_SIZE in the environment creates the ranks itself
    (torch.distributed.run on 127.0.0.1). On a 1-GPU box N = 1 with LA_BENCH_FORCE_DIST=1 takes exactly that road: the line comes
    from a rank the self-launcher made, over a real RCCL process group."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LA_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--seqlen", "8192", "--heads", "8", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-denoise", "--no-head-dims", "--no-fp8", "--no-power"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["config"]["launcher"].startswith("self-launched") and r["data"] == "synthetic"
    assert r["multi_gpu"]["rccl_world_size"] == 1 and r["multi_gpu"]["backend"] == "nccl" and r["multi_gpu"]["overlapped_form_kept"]
    assert r["verified"]["ok"] and r["verified"]["ok_all_ranks"] and r["n_gpus"] == 1


def test_bench_single_process_line_has_the_contract_fields():
    r = _run_bench(["--no-sweep"], {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "verified"):
        assert key in r, key
    assert "multi_gpu" not in r and r["dtype"] == "bf16" and r["config"]["workload"].startswith("QK-Skip")
    assert r["verified"]["ok"]
