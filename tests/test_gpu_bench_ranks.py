"""The rank code of `bench.py --gpus N` on the one GPU a test machine has.

* Two ranks that both render on GPU 0 and meet over gloo (host tensors): the launcher, the row sharding, the asynchronous render
  entry point, the single gather per step and the re-interleaving run as they do on N GPUs; the gathered film must be the film of
  the single-GPU run (equal clamped means: the films are equal bit for bit and the reduction runs over the same tensor shape).
* One rank on torch.distributed's nccl backend (= RCCL): RCCL refuses two ranks on one device, so the group has one member; the
  calls are the ones every rank makes (scripts/nccl_single_rank.py).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c3", "--spp", "4", "--steps", "1", "--warmup", "1", "--no-extras",
           "--cpu-baseline-seconds", "0", *flags]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_gather_the_single_gpu_film():
    one = _bench("--gpus", "1")
    two = _bench("--gpus", "2", "--rehearse-on-one-gpu")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["film_gather"]["gathers_per_step"] == 1 and two["film_gather"]["bytes_into_rank0_per_step"] == 2 * 400 * 800 * 3 * 4
    assert two["film_mean_clamped"] == one["film_mean_clamped"]
    samples = 2 * 800 * 800 * 4
    for line in (one, two):
        assert abs(line["value"] * line["ms_per_step"] * 1e3 - samples) < 1e-6 * samples   # Msamples/s x ms = samples of a step


def test_one_rank_group_on_the_rccl_backend_runs_the_gather_and_the_reductions():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_single_rank.py"), str(_free_port())], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["gathers_equal"] == [True, True] and line["max"] == 1.5 and line["sum"] == [3.0, 4.0]
    assert line["library"] == "librtx_hip.so"
