"""CPU: bench.py's real ``main()`` with more than one rank.

``python bench.py --gpus 2`` started with NO launcher and no RANK / WORLD_SIZE in the environment must create its two ranks itself
(re-execution under torch.distributed.run on 127.0.0.1), print ONE JSON line from rank 0 and exit non-zero when a rank fails. Here the
device op is replaced through the ``LA_BENCH_STANDIN`` seam (tests/bench_standin.py: a torch restatement on the CPU over gloo), so what
runs is every other line of main(): parsing, self-launch, seeding, ``Hl = H // world``, agreement, timing, JSON merge, exit codes.
The partitioning is SURVEY.md 8(e); the reference has none (hopper/lite_attention.py:322-345 is bookkeeping only)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LA_BENCH_FORCE_DIST", "GROUP_RANK",
                        "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.update(LA_BENCH_STANDIN="tests.bench_standin", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), OMP_NUM_THREADS="2")
    env.update(extra)
    return env


def _bench(args, **env_extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--seqlen", "1000", "--heads", "4", "--steps", "2", "--warmup", "1", "--prewarm-steps", "1",
           "--no-cpu-baseline", "--no-power"] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_clean_env(**env_extra), cwd=ROOT)


def _line(res):
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]             # rank 0 only
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_creates_its_two_ranks_and_prints_one_line():
    r = _line(_bench(["--gpus", "2"]))
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "strong" and r["value"] > 0
    assert r["metric"].startswith("self-attn TFLOPS") and r["dtype"] == "bf16" and "STAND-IN" in r["data"]
    assert r["config"]["launcher"].startswith("self-launched")
    mg = r["multi_gpu"]
    assert mg["rccl_world_size"] == 2 and mg["backend"] == "gloo" and mg["heads_per_rank"] == [[0, 2], [2, 4]]
    assert mg["output_shard_bytes"] == 1 * 1000 * 2 * 128 * 2 and mg["bytes_received_per_rank_per_step"] == mg["output_shard_bytes"]
    assert len(mg["kernel_ms_per_rank"]) == 2 and mg["overlapped_form_kept"] is True and mg["overlap_windows"] >= 2
    assert r["config"]["parallelism"].startswith("heads sharded 2x2") and "all-gather" in r["config"]["parallelism"]
    assert r["verified"]["ok"] and r["verified"]["ok_all_ranks"] and r["verified"]["lists_fixed_point"]
    assert abs(r["config"]["sparsity"] - 0.42) < 0.08                   # 4 x 16 tiles: the imposed band at this size
    for absent in ("sweep", "fp8", "denoise50", "other_head_dims", "cpu_baseline"):      # 1-GPU sub-records stay out of an N > 1 line
        assert absent not in r


def test_the_plain_all_gather_form_and_the_external_launcher_form():
    r = _line(_bench(["--gpus", "2", "--overlap-windows", "1"]))
    assert r["multi_gpu"]["overlapped_form_kept"] is False and r["multi_gpu"]["overlap_windows"] == 1
    assert "1 RCCL all-gather of O per step" in r["config"]["parallelism"]
    # the driver's own N > 1 command: torch.distributed.run creates the ranks, bench.py must not launch again
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--seqlen", "1000", "--heads", "4", "--steps", "2",
           "--warmup", "1", "--prewarm-steps", "1", "--no-cpu-baseline", "--no-power"]
    r = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT))
    assert r["n_gpus"] == 2 and r["config"]["launcher"].startswith("external launcher") and r["multi_gpu"]["rccl_world_size"] == 2


def test_a_failing_rank_makes_the_whole_command_fail_and_print_no_line():
    res = _bench(["--gpus", "2", "--overlap-windows", "1"], LA_BENCH_STANDIN_FAIL_RANK="1")
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_a_world_that_is_not_gpus_wide_is_refused():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                         env=_clean_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"), cwd=ROOT)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr and not res.stdout.strip()
