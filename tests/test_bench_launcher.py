"""`python bench.py --gpus N` must start its own ranks (the driver calls it exactly like that): the launcher path is
rehearsed here on the CPU with --dry-launch (gloo process group, the oracle's step on a toy problem)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_launches_its_own_ranks_and_prints_one_json_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["steps"] == 3 and d["value"] > 0


def test_a_failing_rank_makes_the_launcher_exit_nonzero():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # without --dry-launch the ranks need a GPU: in this container they fail, and the launcher must say so with its exit code
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
