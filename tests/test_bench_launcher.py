"""`python bench.py --gpus N` outside torchrun must start its own N ranks (the loop it shards is the reference's
sequential `for` of examples/benchmark.cpp:16).  CPU-only checks: the launcher runs two gloo ranks through the stub step and
forwards ONE JSON line with rccl_world_size = 2; without the stub the RANKS (not the launcher) refuse to run without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"] + extra,
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_launcher_starts_two_ranks_and_forwards_one_json_line():
    r = _run(["--stub-cpu", "--batch", "5", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["config"]["problems_total"] == 5
    assert d["gather_ok"] is True and d["steps"] == 2 and d["scaling"] == "strong"


def test_weak_scaling_total_is_per_rank_times_world():
    r = _run(["--stub-cpu", "--batch", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["problems_total"] == 8 and d["scaling"] == "weak"


def test_without_a_gpu_the_ranks_fail_loudly_not_the_launcher():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a CPU-only box")
    r = _run([])
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "launch with" not in (r.stdout + r.stderr)


def test_receding_horizon_workload_is_sharded_over_the_ranks():
    """configs[4] names 8 GPUs: `--workload rh` shards the instance list like the batch workloads (weak: B per rank, strong: B in
    the job), runs with no collective inside the loop, gathers ONE record per instance to rank 0 and prints ONE line with the true rank count.  CPU
    stand-in: two gloo ranks, the oracle loop on three instances; the gathered records equal a single-process run of the same instances in order"""
    r2 = _run(["--stub-cpu", "--workload", "rh", "--batch", "3", "--scaling", "strong", "--steps", "3"])
    assert r2.returncode == 0, r2.stderr[-2000:]
    lines = [l for l in r2.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r2.stdout
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["instances_total"] == 3 and d2["config"]["rccl_world_size"] == 2
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "0", "--stub-cpu", "--workload", "rh",
                         "--batch", "3", "--scaling", "strong"], capture_output=True, text=True, timeout=240, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["records"] == d2["records"] and len(d2["records"]) == 3
    assert d1["resolves_executed"] == d2["resolves_executed"] and d1["instances_arrived"] == d2["instances_arrived"]
    # weak scaling: B instances PER rank
    rw = _run(["--stub-cpu", "--workload", "rh", "--batch", "2", "--steps", "1"])
    assert rw.returncode == 0, rw.stderr[-2000:]
    dw = json.loads([l for l in rw.stdout.splitlines() if l.strip().startswith("{")][0])
    assert dw["config"]["instances_total"] == 4 and dw["scaling"] == "weak" and len(dw["records"]) == 4
