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
