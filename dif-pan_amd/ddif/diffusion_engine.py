"""Drop-in for the reference `diffusion_engine.py` entry points, WITHOUT its import-time side effects
(the reference runs a training and a test job at import, diffusion_engine.py:508-533).

    test_fn(...)        reference :352-505   inference over a test set: cond assembly -> sampler -> (sr + lms).clip(0,1)
    engine_google(...)  reference :52-348    training loop -- needs the backward pass, which this build does not have
                                             yet: raises DdifError instead of silently training in torch.
    norm / unorm / clamp_fn                  reference :33-49

plus the cond assembly the reference spreads over its datasets and engine:
    haar_dwt2       level-1 "db1" analysis as documented by PyWavelets (dataset/pan_dataset.py:73-81; PyWavelets itself is
                    not installed in the build image, so this restatement is pinned only by the Haar identities in
                    tests/test_engine.py -- "parity unpinned" at this boundary, which sits upstream of `cond`)
    assemble_cond   cond = cat[lms, pan, bilinear_up(wavelets)]   (diffusion_engine.py:221-228, 441-444)
Plotting, tensorboard logging and the MATLAB-style metric suite are out of scope.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import runtime as _rt
from .diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule
from .models.sr3_dwt import UNetSR3
from .runtime import DdifError
from .sharding import cut_tiles, sample_sharded, stitch_tiles
from .solver.dpm_solver import ImageSpaceClamp

DIVISION = {"wv3": 2047.0, "gf2": 1023.0, "qb": 2047.0, "cave": 1.0, "harvard": 1.0}  # diffusion_engine.py:107


def norm(x):
    return x * 2 - 1


def unorm(x):
    return (x + 1) / 2


def clamp_fn(g_sr, lo: float = 0.0, hi: float = 1.0):
    """x -> clamp(x + g_sr, lo, hi) - g_sr (reference :43-49), as an object the fused DPM-Solver path recognises."""
    return ImageSpaceClamp(g_sr, lo, hi)


def haar_dwt2(x: torch.Tensor):
    """Level-1 Haar analysis over the last two axes: returns (cA, (cH, cV, cD)) like pywt.wavedec2(x, "db1", level=1)."""
    a, b = x[..., 0::2, 0::2], x[..., 0::2, 1::2]
    c, d = x[..., 1::2, 0::2], x[..., 1::2, 1::2]
    return (a + b + c + d) * 0.5, ((a + b - c - d) * 0.5, (a - b + c - d) * 0.5, (a - b - c + d) * 0.5)


def wavelet_stack(lms: torch.Tensor, pan: torch.Tensor, dataset_name: str) -> torch.Tensor:
    """[lms_LL, pan_H, pan_D, pan_V] for the pansharpening sets (dataset/pan_dataset.py:139-142) and
    [hsi_LL, rgb_H, rgb_V, rgb_D] for CAVE / Harvard (dataset/hisr.py:57-59), at half resolution."""
    ll, _ = haar_dwt2(lms)
    _, (ph, pv, pd) = haar_dwt2(pan)
    parts = [ll, ph, pv, pd] if dataset_name in ("cave", "harvard") else [ll, ph, pd, pv]
    return torch.cat(parts, dim=1)


def assemble_cond(lms: torch.Tensor, pan: torch.Tensor, wavelets: Optional[torch.Tensor] = None,
                  dataset_name: str = "wv3") -> torch.Tensor:
    """cond = pack[lms, pan, bilinear_up(wavelets)] (reference :221-228) from NORMALISED lms / pan.
    Without `wavelets` (the reference's datasets precompute them on the CPU with PyWavelets) the whole assembly --
    Haar analysis, bilinear x2, channel pack -- is ONE HIP kernel (`ddif_cond_assemble`, csrc/kernels_aux.h) when the tensors
    live on the GPU; host tensors and dataset-provided wavelets take the torch expression the reference uses."""
    if wavelets is None and lms.device.type == "cuda":
        return _rt.cond_assemble(lms, pan, 1.0, 1 if dataset_name in ("cave", "harvard") else 0)
    if wavelets is None:
        wavelets = wavelet_stack(lms, pan, dataset_name)
    up = F.interpolate(wavelets, size=lms.shape[-1], mode="bilinear")
    return torch.cat([lms, pan, up], dim=1).contiguous()


def psnr(gt: torch.Tensor, pred: torch.Tensor, data_range: float = 1.0) -> float:
    mse = torch.mean((gt.double() - pred.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * float(np.log10(data_range ** 2 / mse))


def _dataset_shape(dataset_name: str):
    if dataset_name in ("harvard", "cave"):
        return 31, 3
    if dataset_name in ("wv3", "gf2", "qb"):
        return (8 if dataset_name == "wv3" else 4), 1
    raise NotImplementedError(f"dataset {dataset_name} not supported")


def _load_h5(path: str) -> Dict[str, np.ndarray]:
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise DdifError("h5py is not installed: pass the arrays with data={'lms':…, 'pan':…, 'gt':…} instead") from e
    with h5py.File(path) as f:
        return {k: np.asarray(f[k]) for k in f.keys()}


def build_model(dataset_name: str, n_steps: int, device, weight_path: Optional[str] = None,
                state_dict: Optional[dict] = None, image_size: int = 64):
    """The network + diffusion the engine builds for every dataset (reference :381-410)."""
    C, P = _dataset_shape(dataset_name)
    denoise_fn = UNetSR3(in_channel=C, out_channel=C, lms_channel=C, pan_channel=P, inner_channel=32, norm_groups=1,
                         channel_mults=(1, 2, 2, 4), attn_res=(8,), dropout=0.2, image_size=64,
                         self_condition=True).to(device)
    if state_dict is None and weight_path is not None:
        state_dict = torch.load(weight_path, map_location="cpu")  # checkpoint = bare state_dict (utils/misc.py:89-122)
        if "model" in state_dict and not any(k.startswith("downs.") for k in state_dict):
            state_dict = state_dict["model"]
    if state_dict is not None:
        state_dict = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        denoise_fn.load_state_dict(state_dict, strict=True)
    denoise_fn.eval()
    diffusion = GaussianDiffusion(denoise_fn, image_size=image_size, channels=C, pred_mode="x_start", loss_type="l1",
                                  device=device, clamp_range=(0, 1))
    diffusion.set_new_noise_schedule(betas=make_beta_schedule(schedule="cosine", n_timestep=n_steps, cosine_s=8e-3),
                                     device=device)
    return denoise_fn, diffusion.to(device)


@torch.no_grad()
def test_fn(test_data_path=None, weight_path=None, schedule_type="cosine", batch_size=320, n_steps=1500, show=False,
            device="cuda:0", full_res=False, dataset_name="gf2", division=1023, *, data: Optional[dict] = None,
            state_dict: Optional[dict] = None, sampler: str = "ddim_sample", section_counts: str = "ddim25",
            save_path: Optional[str] = None, seed: Optional[int] = None, tile: Optional[int] = None,
            x_T: Optional[torch.Tensor] = None):
    """Reference test_fn (:352-505): same positional / keyword arguments (schedule_type is ignored there too, :408);
    keyword-only extras let it run without h5 files (`data=`), without a checkpoint file (`state_dict=`) and with the
    DDPM sampler (`sampler="ddpm_sample"`).  Returns dict(sr=(N,C,H,W) array in raw units, psnr=[...], metrics=[...]).

    `tile=t` (new; the reference feeds whole 256x256 / 512x512 scenes through the network, :373-377): cond is assembled
    on the whole scene by the fused kernel, cut into non-overlapping t x t tiles, the tiles are sampled `batch_size` at a
    time -- sharded over the ranks of the default process group when one is initialised, with one RCCL all-gather per
    batch -- and stitched back.  Noise is keyed by (scene, tile) index, so the result does not depend
    on batch_size or on the number of GPUs.  `x_T` (tiled mode only, parity tests): (N * tiles, C, t, t) initial noise."""
    if show:
        raise DdifError("show=True (matplotlib grids) is out of scope of this build")
    d = data if data is not None else _load_h5(test_data_path)
    C, P = _dataset_shape(dataset_name)
    lms_all = torch.as_tensor(np.asarray(d["lms"]), dtype=torch.float32)  # RAW units: the cond-assembly kernel normalises
    pan_all = torch.as_tensor(np.asarray(d["pan"]), dtype=torch.float32)
    gt_all = None if (full_res or "gt" not in d) else torch.as_tensor(np.asarray(d["gt"]), dtype=torch.float32) / division
    if lms_all.shape[1] != C or pan_all.shape[1] != P:
        raise DdifError(f"{dataset_name}: expected lms with {C} and pan with {P} channels, got {tuple(lms_all.shape)} / {tuple(pan_all.shape)}")
    _, diffusion = build_model(dataset_name, n_steps, device, weight_path, state_dict, image_size=lms_all.shape[-1])
    if seed is not None:
        torch.manual_seed(seed)
    preds, scores, mets = [], [], []
    wave_order = 1 if dataset_name in ("cave", "harvard") else 0
    H, W = lms_all.shape[-2:]
    if tile is not None and (H > tile or W > tile):
        if H % tile or W % tile:
            raise DdifError(f"scene {H}x{W} is not a multiple of tile={tile}")
        ny, nx = H // tile, W // tile
        mode_kw = dict(section_counts=section_counts) if sampler == "ddim_sample" else {}
        base_seed = 0 if seed is None else seed
        for i in range(lms_all.shape[0]):  # one scene at a time: assemble on the scene, cut, sample, stitch
            raw_l, raw_p = lms_all[i:i + 1].to(device), pan_all[i:i + 1].to(device)
            cond_scene = _rt.cond_assemble(raw_l, raw_p, float(division), wave_order)[0]
            tiles = cut_tiles(cond_scene, tile)
            out_tiles = []
            for j in range(0, tiles.shape[0], batch_size):
                xt = None if x_T is None else x_T[i * ny * nx + j: i * ny * nx + j + batch_size].to(device)
                out_tiles.append(sample_sharded(diffusion, tiles[j:j + batch_size].contiguous(), mode=sampler, seed=base_seed, x_T=xt,
                                                tile_base=i * ny * nx + j, **mode_kw))
            sr = stitch_tiles(torch.cat(out_tiles, dim=0), ny, nx).unsqueeze(0)
            if gt_all is not None:
                gt = gt_all[i:i + 1].to(device)
                scores.append(psnr(gt.cpu(), sr.cpu()))
                mets.append(_rt.metrics(gt, sr, 4.0).cpu().numpy())  # SAM / ERGAS / PSNR / CC on the GPU (reference :449)
            preds.append((sr.cpu().numpy() * division).clip(0, division))
    else:
        for i in range(0, lms_all.shape[0], batch_size):
            raw_l, raw_p = lms_all[i:i + batch_size].to(device), pan_all[i:i + batch_size].to(device)
            cond = _rt.cond_assemble(raw_l, raw_p, float(division), wave_order)
            lms = cond[:, :C]
            if sampler == "ddim_sample":
                sr = diffusion(cond, mode="ddim_sample", section_counts=section_counts)  # respaces the schedule in place, once (SURVEY D-2)
            else:
                sr = diffusion(cond, mode="ddpm_sample")
            sr = (sr + lms).clip(0, 1)  # reference :446-447
            if gt_all is not None:
                gt = gt_all[i:i + batch_size].to(device)
                scores.append(psnr(gt.cpu(), sr.cpu()))
                mets.append(_rt.metrics(gt, sr, 4.0).cpu().numpy())
            preds.append((sr.cpu().numpy() * division).clip(0, division))
    out = dict(sr=np.concatenate(preds, axis=0), psnr=scores,
               metrics=(np.concatenate(mets, axis=0) if mets else np.zeros((0, 4), np.float32)))  # columns: SAM, ERGAS, PSNR (ref. sign), CC
    if save_path is not None:
        from scipy.io import savemat

        os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
        savemat(save_path, {k: v for k, v in dict(sr=out["sr"], lms=np.asarray(d["lms"]), pan=np.asarray(d["pan"]),
                                                  **({"gt": np.asarray(d["gt"])} if "gt" in d else {})).items()})
    return out


def engine_google(train_dataset_path, valid_dataset_path, dataset_name=None, image_n_channel=8, image_size=64,
                  schedule_type="cosine", n_steps=3_000, max_iterations=400_000, device="cuda:0", batch_size=128,
                  lr_d=1e-4, show_recon=False, pretrain_weight=None, pretrain_iterations=None, *, constrain_channel=None):
    """Reference engine_google (:52-348): training + periodic validation.  The training step needs the backward pass
    through the denoiser, AdamW, EMA and (multi-GPU) a gradient all-reduce; only the forward half of `p_losses`
    exists in this build.  Refusing loudly is deliberate: a silent torch fallback would not be this project's path."""
    raise DdifError("engine_google: the training step (backward pass, config 5) is not implemented by the HIP path yet; "
                    "sampling / validation are available through test_fn and GaussianDiffusion(mode='ddim_sample'|'ddpm_sample')")
