"""World-size-2 coverage of the tile-sharded path on CPU: gloo backend, kernels under the test-only host emulator.
Two ranks each sample half of a 2-tile "scene" and all-gather; the result must equal the single-process run of both
tiles bit for bit (noise is keyed by global tile index, tiles never mix)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_cases as gc
from ddif_testlib import make_diffusion, make_net, use_emulator


def _run(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HIPEMU_THREADS="4")
    torch.set_num_threads(2)
    use_emulator()
    from ddif.sharding import sample_sharded

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ds, H, T = "gf2", 8, 2
    cond = gc.tiles_for(ds, 2, H, H, seed=21)["cond"]
    d = make_diffusion(make_net(ds, "cpu"), gc.DATASETS[ds][0], T, H, "cpu")
    out = sample_sharded(d, cond, mode="ddpm_sample", seed=5)
    if rank == 0:
        q.put(out.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    return torch.from_numpy(out)


def test_two_rank_tile_shard_equals_single_process():
    one = _spawn(1, 29611)
    two = _spawn(2, 29612)
    assert one.shape == (2, 4, 8, 8)
    assert torch.equal(one, two)
    assert float(one.min()) >= 0.0 and float(one.max()) <= 1.0


def test_cut_and_stitch_roundtrip():
    from ddif.sharding import cut_tiles, shard_range, stitch_tiles

    scene = torch.arange(3 * 16 * 24, dtype=torch.float32).reshape(3, 16, 24)
    tiles = cut_tiles(scene, 8)
    assert tiles.shape == (6, 3, 8, 8)
    assert torch.equal(tiles[4], scene[:, 8:16, 8:16])  # tile (row 1, col 1)
    assert torch.equal(stitch_tiles(tiles, 2, 3), scene)
    assert shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
