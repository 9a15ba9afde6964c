"""Tile sharding across the GPUs of one node.

Tiles (batch entries) are independent through the whole sampler, so a scene's tiles are split into contiguous
blocks, one per rank; every rank runs the full sampler on its block with a full weight replica and NO data-path
collective.  The only exchange is one all-gather of the fused tiles (RCCL over xGMI; `nccl` backend) to stitch the
scene.  The reference itself never tiles (it feeds whole images, diffusion_engine.py:373-377), so tiles here are
non-overlapping and the stitch is a pure index permutation.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_tiles: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of tile indices owned by `rank`; n_tiles must divide evenly."""
    if n_tiles % world:
        raise ValueError(f"{n_tiles} tiles do not split evenly over {world} ranks")
    per = n_tiles // world
    return rank * per, (rank + 1) * per


def cut_tiles(scene: torch.Tensor, tile: int) -> torch.Tensor:
    """(C, H, W) -> (ny*nx, C, tile, tile), row-major over the tile grid."""
    C, H, W = scene.shape
    if H % tile or W % tile:
        raise ValueError("scene size must be a multiple of the tile size")
    ny, nx = H // tile, W // tile
    return scene.reshape(C, ny, tile, nx, tile).permute(1, 3, 0, 2, 4).reshape(ny * nx, C, tile, tile).contiguous()


def stitch_tiles(tiles: torch.Tensor, ny: int, nx: int) -> torch.Tensor:
    """Inverse of cut_tiles: (ny*nx, C, h, w) -> (C, ny*h, nx*w)."""
    n, C, h, w = tiles.shape
    assert n == ny * nx
    return tiles.reshape(ny, nx, C, h, w).permute(2, 0, 3, 1, 4).reshape(C, ny * h, nx * w).contiguous()


def _collectives_on(t: torch.Tensor) -> bool:
    """Run the stitch collective?  Yes with more than one rank; at world size 1 only when the group is RCCL ("nccl") and the tensor lives on a GPU -- a one-rank
    gloo group (a CPU-side launcher, a unit test) must not be handed CUDA tensors."""
    if not dist.is_initialized():
        return False
    if dist.get_world_size() > 1:
        return True
    return dist.get_backend() == "nccl" and t.is_cuda


def sample_sharded(diffusion, cond_all: torch.Tensor, mode: str = "ddpm_sample", seed: int = 0, x_T: torch.Tensor = None,
                   tile_base: int = 0, **kw) -> torch.Tensor:
    """Sample every tile of `cond_all` (n_tiles, 2C+4P, h, w; the same tensor on every rank) with the tiles split over
    the ranks of the default process group, then all-gather.  Returns sr = clip(residual + lms, 0, 1) for ALL tiles on
    every rank.  The noise stream is keyed by global tile index (`tile_base` + index in `cond_all`), so the result does
    not depend on the world size or on how a scene's tiles are batched.  `x_T` (n_tiles, C, h, w), optional, pins the
    initial noise (parity tests); by default it comes from the same counter-based generator."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(cond_all.shape[0], rank, world)
    cond = cond_all[lo:hi].contiguous()
    C = diffusion.channels
    if x_T is not None:
        kw = dict(kw, x_T=x_T[lo:hi].contiguous())
    res = diffusion(cond, mode=mode, seed=seed, tile0=tile_base + lo, device_rng=x_T is None, **kw)
    sr = (res + cond[:, :C]).clip(0, 1)  # diffusion_engine.py:446-447
    if not _collectives_on(sr):
        return sr
    # (with an RCCL process group the collective runs even at world size 1 -- a copy -- so that a 1-GPU run under a launcher exercises RCCL: tests/test_rccl_world1.py)
    out = torch.empty((cond_all.shape[0],) + tuple(sr.shape[1:]), dtype=sr.dtype, device=sr.device)
    dist.all_gather_into_tensor(out, sr.contiguous())
    return out


def scene_grid(n_tiles: int) -> Tuple[int, int]:
    """(ny, nx) with ny * nx == n_tiles and ny the largest divisor <= sqrt(n_tiles): 64 -> 8 x 8, 32 -> 4 x 8, 8 -> 2 x 4."""
    ny = int(n_tiles ** 0.5)
    while ny > 1 and n_tiles % ny:
        ny -= 1
    return max(1, ny), n_tiles // max(1, ny)


def sample_scene_dpmpp(net, diffusion, cond_all: torch.Tensor, x_T_all: torch.Tensor, steps: int = 50, order: int = 2, grid=None) -> torch.Tensor:
    """BASELINE configs[2]: ONE scene = `cond_all` (n_tiles, 2C+4P, h, w; the same on every rank) sampled with DPM-Solver++ multistep, the tiles
    split in contiguous blocks over the ranks of the default process group (strong scaling: n_tiles / world per GPU), one all-gather, then the
    stitch.  `x_T_all` (n_tiles, C, h, w) is the scene's initial noise (the same on every rank; each rank uses its block), so the fused scene does
    not depend on the number of GPUs.  Returns the fused scene (C, ny h, nx w) on every rank.

    Tiling is a DESIGN DECISION of this build, not a property of the reference: the reference feeds the whole scene through the network
    (diffusion_engine.py:373-377), where GroupNorm(1 group) statistics span the whole scene, the 3x3 convs see their neighbours across what
    would be tile edges (here: zero padding at every tile edge) and the H/8 self-attention attends over the whole scene.  A tiled run is
    therefore NOT the same function as the whole-scene run; each tile is the reference's computation ON THAT TILE
    (what `test_fn` does with 64 x 64 training-size inputs), and whole scenes up to the 4 GiB tensor limit can still go through one plan."""
    from .solver.dpm_solver import DPM_Solver, ImageSpaceClamp, NoiseScheduleVP, model_wrapper

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = cond_all.shape[0]
    ny, nx = grid if grid is not None else scene_grid(n)
    lo, hi = shard_range(n, rank, world)
    cond = cond_all[lo:hi].contiguous()
    C = diffusion.channels
    lms = cond[:, :C].contiguous()
    ns = NoiseScheduleVP("discrete", betas=diffusion.betas)
    fn = model_wrapper(net, ns, model_type="x_start", guidance_type="classifier-free", guidance_scale=1.0, condition=cond)
    solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=ImageSpaceClamp(lms, 0.0, 1.0))
    res = solver.sample(x_T_all[lo:hi].contiguous(), steps=steps, order=order, skip_type="time_uniform", method="multistep")
    sr = (res + lms).clip(0, 1)  # diffusion_engine.py:446-447
    if _collectives_on(sr):
        out = torch.empty((n,) + tuple(sr.shape[1:]), dtype=sr.dtype, device=sr.device)
        dist.all_gather_into_tensor(out, sr.contiguous())
        sr = out
    return stitch_tiles(sr, ny, nx)
