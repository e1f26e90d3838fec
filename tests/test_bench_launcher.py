"""bench.py started the way the driver starts it (`python bench.py --gpus N ...`, no launcher, no WORLD_SIZE): for N > 1 it must
start its own ranks as a child torch.distributed.run and relay rank 0's line; on a node with fewer GPUs it must refuse before
touching a GPU.  --dry-launch replaces the render by a gloo all-reduce so that the launcher path runs without a GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=300)


def test_plain_start_with_two_gpus_launches_two_ranks():
    r = _run("--gpus", "2", "--steps", "4", "--warmup", "2", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE line, from rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["rank_sum"] == 3.0 and lines[0]["steps"] == 4 and lines[0]["warmup"] == 2


def test_more_gpus_than_the_node_has_is_refused_before_any_gpu_work():
    import torch
    have = torch.cuda.device_count()
    for extra in (["--multi", "process"], ["--multi", "single"]):
        r = _run("--gpus", str(have + 1), *extra)
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        assert "visible on this node" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
