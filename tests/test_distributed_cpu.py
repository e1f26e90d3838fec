"""The N > 1 path on CPU: two gloo ranks each produce their row shard (the CPU oracle stands in
for the GPU kernel: same sharding rule, same counter-based sampler), rank 0 gathers through
raytracer_amd.parallel.gather_film, and the result must equal the single-rank film bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _worker(rank, world, port, H, W, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rtx_pkg
    pkg = rtx_pkg.load()
    import importlib
    parallel = importlib.import_module("raytracer_amd.parallel")
    import oracle_py as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = pkg.scene_loader.load(os.path.join(GOLDEN, "scenes", "cornell_box_quad.json"))
        osc = O.OracleScene(sd)
        st = sd.settings(width=W, height=H, samples_per_pixel=2, max_depth=6, seed=5, shard_index=rank, shard_count=world)
        film, _ = osc.render(sd.camera(W, H), st, O.RNG_COUNTER, threads=2)
        local = torch.from_numpy(film[rank::world].astype(np.float32).copy())   # packed rows of this shard, bottom-up
        assert local.shape[0] == parallel.shard_rows(H, rank, world)
        full = parallel.gather_film(local, H, W, rank, world)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [12, 13])   # 13: ranks own different row counts (padding path)
def test_two_rank_gather_equals_single_rank_film(H, pkg, oracle, load_scene, tmp_path):
    W = 10
    out = str(tmp_path / "film.npy")
    port = 29500 + (os.getpid() % 2000) + H
    mp.spawn(_worker, args=(2, port, H, W, out), nprocs=2, join=True)
    got = np.load(out)
    sd = load_scene("cornell_box_quad")
    full, _ = oracle.OracleScene(sd).render(sd.camera(W, H), sd.settings(width=W, height=H, samples_per_pixel=2, max_depth=6, seed=5),
                                            oracle.RNG_COUNTER, threads=2)
    assert np.array_equal(got, full.astype(np.float32))


def test_gather_film_world_one_is_identity(pkg):
    import importlib
    parallel = importlib.import_module("raytracer_amd.parallel")
    t = torch.arange(5 * 4 * 3, dtype=torch.float32).reshape(5, 4, 3)
    assert parallel.gather_film(t, 5, 4, 0, 1) is t
    assert [parallel.shard_rows(13, r, 4) for r in range(4)] == [4, 3, 3, 3]


def _sharded_worker(rank, world, port, shapes, out_path, via_host=False, collective=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import rtx_pkg
    rtx_pkg.load()
    import importlib
    parallel = importlib.import_module("raytracer_amd.parallel")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        films = parallel.ShardedFilms(shapes, rank, world, "cpu", via_host=via_host, collective=collective)
        for step in range(2):   # a second step reuses every buffer
            for i, ((h, w), local) in enumerate(zip(shapes, films.local)):
                whole = _pattern(h, w, i, step)
                local.copy_(whole[rank::world])
            full = films.gather()
            if rank == 0:
                assert [tuple(f.shape) for f in full] == [(h, w, 3) for h, w in shapes]
                for i, f in enumerate(full):
                    assert torch.equal(f, _pattern(*shapes[i], i, step)), (i, step)
            else:
                assert full is None
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


def _pattern(h, w, film, step):
    return (torch.arange(h * w * 3, dtype=torch.float32).reshape(h, w, 3) + 1000.0 * film + 0.5 * step)


# via_host: the rehearsal path (films staged through host memory for gloo); collective with one rank: a group of one goes through
# the collective all the same (what a one-GPU machine runs of the RCCL path)
@pytest.mark.parametrize("world,via_host,collective", [(2, False, None), (3, False, None), (2, True, None), (1, False, True), (1, True, True)])
def test_sharded_films_gather_several_films_in_one_exchange(world, via_host, collective, pkg, tmp_path):
    shapes = [(12, 5), (13, 4), (2, 3)]   # equal shares, unequal shares, fewer rows than ranks (world 3)
    out = str(tmp_path / "ok.txt")
    port = 31500 + (os.getpid() % 2000) + world + (7 if via_host else 0) + (11 if collective else 0)
    mp.spawn(_sharded_worker, args=(world, port, shapes, out, via_host, collective), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_sharded_films_world_one(pkg):
    import importlib
    parallel = importlib.import_module("raytracer_amd.parallel")
    films = parallel.ShardedFilms([(4, 3), (5, 2)], 0, 1, "cpu")
    films.local[1].fill_(2.0)
    out = films.gather()
    assert out[0].shape == (4, 3, 3) and out[1].shape == (5, 2, 3) and float(out[1].sum()) == 60.0
    assert films.bytes_into_root() == 0
