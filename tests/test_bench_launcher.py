"""`python bench.py --gpus N` from a bare shell starts its own ranks (the parent never touches the GPU) and
relays rank 0's single JSON line and the exit code.  Rehearsed here without a GPU: QK_BENCH_LAUNCH_TEST=1
makes every rank stop after the rendezvous (gloo) and one collective."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["QK_BENCH_LAUNCH_TEST"] = "1"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_self_launch_two_ranks_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1


def test_single_gpu_runs_inline():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    assert "torch.distributed.run" not in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE" in (r.stderr + r.stdout)
