#!/usr/bin/env python3
"""The calls a rank of `bench.py --gpus N` makes on torch.distributed's nccl backend (= RCCL), in a group of ONE rank on GPU 0:
group set-up bound to the device, barrier, the film gather of parallel.ShardedFilms into row views of one receive buffer, and the
float64 MAX / SUM reductions of the timings.  A machine with one GPU cannot hold two RCCL ranks (RCCL refuses two ranks on one
device), so this is as much of that path as such a machine can run.  Prints one JSON line.

    python scripts/nccl_single_rank.py [port]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port = sys.argv[1] if len(sys.argv) > 1 else "29533"
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import rtx_pkg
    pkg = rtx_pkg.load()
    import importlib
    parallel = importlib.import_module("raytracer_amd.parallel")

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    dist.barrier()
    shapes = [(37, 20), (16, 48)]
    films = parallel.ShardedFilms(shapes, 0, 1, "cuda:0", collective=True)
    gen = torch.Generator(device="cpu").manual_seed(7)
    want = []
    for local in films.local:
        v = torch.rand(local.shape, generator=gen)
        local.copy_(v)
        want.append(v)
    equal = []
    for _ in range(2):                                   # the second gather reuses the buffers of the first
        got = films.gather()
        equal.append(all(bool((g.cpu() == w).all()) and tuple(g.shape) == (h, w_, 3) for g, w, (h, w_) in zip(got, want, shapes)))
    tt = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
    tot = torch.tensor([3.0, 4.0], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps({"backend": "nccl", "world": 1, "gathers_equal": equal, "max": float(tt.item()), "sum": [float(v) for v in tot.tolist()],
                      "library": os.path.basename(pkg.abi.LIB_PATH)}), flush=True)


if __name__ == "__main__":
    main()
