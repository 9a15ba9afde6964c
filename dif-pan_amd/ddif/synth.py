"""Deterministic synthetic weights and WV3/GF2/CAVE-shaped tiles.

No checkpoint or dataset ships with the reference, so parity fixtures, tests and
bench.py all use the same seeded generators.  Everything is drawn on the CPU with an explicit
`torch.Generator`, so the build container (where the reference is imported to make golden vectors) and the
GPU box produce bit-identical tensors.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import torch
import torch.nn.functional as F

from .layout import param_manifest


def synth_state_dict(cfg: dict, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """State dict with the reference key names.  Distribution family = torch's default inits
    (U(+-1/sqrt(fan_in)) for conv/linear weights and biases) with norm affines perturbed around (1, 0) and the
    zero-initialised `body.3` convs (models/sr3_dwt.py:386-387) drawn N(0, 0.02) so every branch is live."""
    sd: Dict[str, torch.Tensor] = {}
    manifest = param_manifest(cfg)
    shapes = dict(manifest)
    for key, shape in manifest:
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
        is_norm = any(s in key for s in (".block.0.", ".norm.", ".prenorm_x.", ".body.1."))
        if is_norm:
            if key.endswith("weight"):
                w = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                w = 0.1 * torch.randn(shape, generator=g)
        elif ".body.3." in key:
            w = 0.02 * torch.randn(shape, generator=g)
        else:
            if key.endswith("weight"):
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
            else:  # bias: fan_in of the sibling weight
                wshape = shapes[key[: -len("bias")] + "weight"]
                fan_in = 1
                for s in wshape[1:]:
                    fan_in *= s
            bound = 1.0 / math.sqrt(fan_in)
            w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[key] = w.to(torch.float32).contiguous()
    return sd


def haar_level1(x: torch.Tensor):
    """Level-1 Haar ("db1") analysis of (B,C,H,W) with even H,W -> LL, H (rows), V (cols), D, each (B,C,H/2,W/2).
    Definition as documented by PyWavelets: lo=[1,1]/sqrt2, hi detail = (x[2k]-x[2k+1])/sqrt2."""
    a, b = x[..., 0::2, 0::2], x[..., 0::2, 1::2]
    c, d = x[..., 1::2, 0::2], x[..., 1::2, 1::2]
    ll = (a + b + c + d) * 0.5
    ch = (a + b - c - d) * 0.5  # detail along rows axis (-2)
    cv = (a - b + c - d) * 0.5  # detail along cols axis (-1)
    cd = (a - b - c + d) * 0.5
    return ll, ch, cv, cd


def synth_tiles(B: int, C: int = 8, P: int = 1, H: int = 64, W: int = 64, seed: int = 7, order: str = "pan"):
    """Synthetic tiles in the PanCollection training-patch layout.
    Returns dict(gt, lms, pan, cond) with cond = cat[lms, pan, bilinear_up(wavelets)] of shape (B, 2C+4P, H, W)
    (diffusion_engine.py:221-228).  `order`: "pan" -> [LL, H, D, V] (dataset/pan_dataset.py:139-142),
    "hisr" -> [LL, H, V, D] (dataset/hisr.py:57-59)."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(B, C, H, W, generator=g)
    lms = F.interpolate(F.avg_pool2d(gt, 4), size=(H, W), mode="bilinear")
    pan = gt.mean(dim=1, keepdim=True).repeat(1, P, 1, 1)
    ll, _, _, _ = haar_level1(lms)
    _, ph, pv, pd = haar_level1(pan)
    wave = torch.cat([ll, ph, pd, pv] if order == "pan" else [ll, ph, pv, pd], dim=1)
    cond = torch.cat([lms, pan, F.interpolate(wave, size=(H, W), mode="bilinear")], dim=1)
    return dict(gt=gt, lms=lms, pan=pan, cond=cond.contiguous())
