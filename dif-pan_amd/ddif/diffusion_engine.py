"""Drop-in for the reference `diffusion_engine.py` entry points, WITHOUT its import-time side effects
(the reference runs a training and a test job at import, diffusion_engine.py:508-533).

    test_fn(...)        reference :352-505   inference over a test set: cond assembly -> sampler -> (sr + lms).clip(0,1)
    engine_google(...)  reference :52-348    training loop: cond assembly -> p_losses -> loss.backward() (csrc/ddif_train.cpp) -> [DDP all-reduce]
                                             -> fused clip + AdamW + EMA step; periodic DDIM-25 validation with metrics
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
        # whole scenes through the network, as the reference does (:373-377).  The kernels address a tensor with 32-bit byte offsets: a batch
        # whose largest activation would reach 4 GiB is refused by the library -- the batch is then halved (loudly) and retried
        i, bs = 0, batch_size
        while i < lms_all.shape[0]:
            raw_l, raw_p = lms_all[i:i + bs].to(device), pan_all[i:i + bs].to(device)
            cond = _rt.cond_assemble(raw_l, raw_p, float(division), wave_order)
            lms = cond[:, :C]
            try:
                if sampler == "ddim_sample":
                    sr = diffusion(cond, mode="ddim_sample", section_counts=section_counts)  # respaces the schedule in place, once (SURVEY D-2)
                else:
                    sr = diffusion(cond, mode="ddpm_sample")
            except DdifError as e:
                if "4 GiB" in str(e) and bs > 1:
                    bs = max(1, bs // 2)
                    print(f"[ddif] test_fn: a batch of {raw_l.shape[0]} scenes of {H}x{W} exceeds the 4 GiB tensor limit; retrying with batch_size={bs}")
                    continue
                raise
            sr = (sr + lms).clip(0, 1)  # reference :446-447
            if gt_all is not None:
                gt = gt_all[i:i + bs].to(device)
                scores.append(psnr(gt.cpu(), sr.cpu()))
                mets.append(_rt.metrics(gt, sr, 4.0).cpu().numpy())
            preds.append((sr.cpu().numpy() * division).clip(0, division))
            i += raw_l.shape[0]
    out = dict(sr=np.concatenate(preds, axis=0), psnr=scores,
               metrics=(np.concatenate(mets, axis=0) if mets else np.zeros((0, 4), np.float32)))  # columns: SAM, ERGAS, PSNR (ref. sign), CC
    if save_path is not None:
        from scipy.io import savemat

        os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
        savemat(save_path, {k: v for k, v in dict(sr=out["sr"], lms=np.asarray(d["lms"]), pan=np.asarray(d["pan"]),
                                                  **({"gt": np.asarray(d["gt"])} if "gt" in d else {})).items()})
    return out


class _Batches:
    """Mini-batches (pan, lms, hr) of raw-count tensors out of an in-memory set {"pan", "lms", "gt"} (the three arrays PanDataset / HISRDataSets
    read from their h5 files, dataset/pan_dataset.py:37-66, dataset/hisr.py:22-46), reshuffled every epoch like DataLoader(shuffle=True).
    With `world > 1` every epoch's permutation comes from a generator seeded by (seed, epoch) -- the SAME on every rank -- and rank r takes
    order[r::world] (what DistributedSampler does): the ranks see disjoint samples whatever their own torch RNG state is.

    Position inside the epoch is part of the training state (`state()` / `load_state()`): a checkpoint written in the MIDDLE of an epoch
    resumes with the same permutation at the next batch.  With one rank the permutation comes from the global torch generator (as DataLoader's
    RandomSampler draws it), so the state keeps the generator state the permutation was drawn FROM; on resume it is re-drawn from that state
    without disturbing the (restored) global stream."""

    def __init__(self, data: Dict[str, torch.Tensor], batch_size: int, shuffle: bool = True, rank: int = 0, world: int = 1, seed: int = 0):
        self.pan, self.lms, self.gt = (torch.as_tensor(np.asarray(data[k]), dtype=torch.float32) for k in ("pan", "lms", "gt"))
        self.n, self.bs, self.shuffle = self.gt.shape[0], batch_size, shuffle
        self.rank, self.world, self.seed = rank, world, seed
        if world > 1 and self.n < world:
            raise DdifError("engine_google: %d training samples for %d ranks -- every rank needs at least one (the gradient all-reduce would "
                            "otherwise wait for ranks that have no batch)" % (self.n, world))
        self.epoch = 0        # epochs STARTED so far
        self.offset = 0       # batches of the current epoch already handed out
        self._perm_rng = None # world == 1: the global generator state the current epoch's permutation was drawn from
        self._resume = None

    def _order(self, epoch_index: int, redraw_from=None):
        if self.world > 1:
            g = torch.Generator().manual_seed(self.seed * 1_000_003 + epoch_index)
            order = torch.randperm(self.n, generator=g) if self.shuffle else torch.arange(self.n)
            n_even = (self.n // self.world) * self.world  # every rank the same number of samples (the tail is dropped, as drop_last does)
            return order[:n_even][self.rank::self.world]
        if not self.shuffle:
            return torch.arange(self.n)
        if redraw_from is not None:  # mid-epoch resume: the same permutation again, the global stream untouched
            keep = torch.get_rng_state()
            torch.set_rng_state(redraw_from)
            order = torch.randperm(self.n)
            torch.set_rng_state(keep)
            self._perm_rng = redraw_from
            return order
        self._perm_rng = torch.get_rng_state()
        return torch.randperm(self.n)

    def state(self) -> dict:
        return {"epoch": self.epoch, "offset": self.offset, "perm_rng": self._perm_rng}

    def load_state(self, st: dict):
        self.epoch, self._resume = int(st["epoch"]), dict(st)

    def __iter__(self):
        st, self._resume = self._resume, None
        nb = lambda order: -(-len(order) // self.bs)
        start = 0
        if st is not None and int(st.get("offset", 0)) > 0 and self.epoch > 0:
            order = self._order(self.epoch - 1, redraw_from=st.get("perm_rng"))  # the epoch that was in force when the state was taken
            if int(st["offset"]) < nb(order):
                start = int(st["offset"])
            else:
                order = None  # the checkpoint fell on the epoch's last batch: a fresh epoch starts, exactly as the uninterrupted run does
        else:
            order = None
        if order is None:
            order = self._order(self.epoch)
            self.epoch += 1
        self.offset = start
        for k in range(start * self.bs, len(order), self.bs):
            idx = order[k:k + self.bs]
            self.offset += 1  # counted when handed out: a state taken while the consumer works on this batch resumes at the next one
            yield self.pan[idx], self.lms[idx], self.gt[idx]


def _open_set(path_or_data):
    if isinstance(path_or_data, dict):
        return path_or_data
    try:
        import h5py  # not in the build image; the reference's datasets are h5 files (diffusion_engine.py:142-143)
    except ImportError as e:
        raise DdifError("engine_google: reading %r needs h5py; pass the arrays as a dict {'pan', 'lms', 'gt'} instead" % (path_or_data,)) from e
    f = h5py.File(path_or_data, "r")
    return {k: np.asarray(f[k]) for k in ("pan", "lms", "gt")}


_BUCKET_ALIGN = 64  # floats: every gradient starts on a 256-byte boundary, as a separately allocated tensor would (vector stores in the kernels)


def gradient_bucket(params):
    """One flat fp32 buffer holding every parameter's gradient (each on a 256-byte boundary; the gaps stay zero), and the per-parameter
    views into it.  The native training step and the fused optimizer work on the views (raw pointers), the DDP all-reduce on the flat
    buffer: no concatenation, no copy back."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += -(-p.numel() // _BUCKET_ALIGN) * _BUCKET_ALIGN
    flat = torch.zeros(off, dtype=torch.float32, device=params[0].device)
    return flat, [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]


def _flat_of(grads):
    """The flat tensor `grads` are the views of (gradient_bucket, in order), or None."""
    base = grads[0]._base
    if base is None or base.dim() != 1:
        return None
    off = 0
    for g in grads:
        if g._base is not base or g.storage_offset() != off or not g.is_contiguous():
            return None
        off += -(-g.numel() // _BUCKET_ALIGN) * _BUCKET_ALIGN
    return base if off == base.numel() else None


def average_gradients(grads, world: int):
    """DDP of config 5 (one process per GPU, `torch.distributed`; backend "nccl" = RCCL over xGMI): the gradients of all ranks are averaged
    in ONE flat bucket (7.1 M floats = 28 MB for the engine network: a single ring all-reduce, bandwidth-bound on the xGMI links rather
    than latency-bound on 702 small ones).  In place: the fused optimizer keeps the gradient pointers.  Gradients allocated by
    `gradient_bucket` are reduced where they lie; any other list goes through a temporary concatenation."""
    import torch.distributed as dist

    bucket = _flat_of(grads)
    flat = bucket if bucket is not None else torch.cat([g.reshape(-1) for g in grads])
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)  # RCCL averages inside the collective
    else:  # gloo (the CPU tests) has no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
    if bucket is not None:
        return
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def broadcast_parameters(tensors, src: int = 0):
    """What DistributedDataParallel does at construction: every rank starts from rank `src`'s parameters (and EMA copies), whatever its own
    RNG drew at initialisation.  One flat bucket, in place."""
    import torch.distributed as dist

    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


def lr_at(iteration: int, base: float, milestones=(100_000, 200_000, 350_000), gamma: float = 0.2) -> float:
    """MultiStepLR of the reference (diffusion_engine.py:210-212)"""
    return base * gamma ** sum(1 for m in milestones if iteration >= m)


def engine_google(train_dataset_path, valid_dataset_path, dataset_name=None, image_n_channel=8, image_size=64,
                  schedule_type="cosine", n_steps=3_000, max_iterations=400_000, device="cuda:0", batch_size=128,
                  lr_d=1e-4, show_recon=False, pretrain_weight=None, pretrain_iterations=None, *, constrain_channel=None,
                  add_n_channel=1, ema_start_iter=20_000, valid_every=5_000, save_dir=None, save_every=5_000, resume_state=None,
                  data_seed=0, log=print, log_every=1):
    """Reference engine_google (:52-348): the training loop.  Same keyword names and defaults; `train_dataset_path` / `valid_dataset_path`
    may also be dicts {"pan", "lms", "gt"} of raw-count arrays (h5py is not part of this image).  Per iteration, as in the reference
    (:218-241): cond assembly (one kernel), `diff_loss, recon = diffusion(hr - lms, cond=cond)`, `diff_loss.backward()` (the library's
    reverse pass), gradient all-reduce when torch.distributed is initialised (DDP of config 5: one process per GPU, RCCL; parameters
    broadcast from rank 0 at start, every epoch's permutation sharded by rank), then clip 0.003 + AdamW(lr, weight_decay 1e-4) +
    EMA(0.995 after `ema_start_iter`) as the fused three-launch optimizer step, MultiStepLR.  Every `valid_every` iterations: DDIM-25
    sampling of one validation batch on a SEPARATE diffusion object holding the EMA weights (the reference validates on
    `ema_updater.ema_model`, a deep copy, :277-329 -- the training schedule is never respaced) with SAM / ERGAS / PSNR / CC / SSIM from
    `ddif_metrics`.  Every `save_every` iterations (rank 0): `diffusion_{name}_iter_{N}.pth` and `ema_diffusion_{name}_iter_{N}.pth` as bare
    state_dicts like the reference (:333-340; `test_fn(weight_path=...)` loads either) plus `train_state_{name}_iter_{N}.pth` with the
    optimizer moments, step, iteration, RNG states and the data position (epoch, batch offset inside it, the generator state the epoch's
    permutation was drawn from); `resume_state=<that file>` continues such a run bit-identically, also from the middle of an epoch.
    `log_every` (not in the reference, default 1 = its behaviour): read the losses back and log every k-th iteration only -- the iteration
    itself never synchronises with the host.  Plotting, tensorboard and .mat dumps of the reference are out of scope.  Returns a dict with the loss history, the validation records,
    the model, the diffusion wrapper and the EMA weights."""
    import random as _random

    import torch.distributed as dist

    if schedule_type != "cosine":
        raise DdifError("engine_google: the reference trains on the cosine schedule")
    if show_recon:
        raise DdifError("engine_google: show_recon (matplotlib grids) is out of scope")
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)  # reference :78; the stateless ops launch on the current device
    name = dataset_name
    if name is None:
        if isinstance(train_dataset_path, dict):
            raise DdifError("engine_google: dataset_name is required with in-memory data")
        name = str(train_dataset_path).strip(".h5").split("_")[-1]
    if name not in DIVISION:
        raise DdifError("dataset %r not supported" % (name,))
    hisr = name in ("cave", "harvard")
    if hisr:
        add_n_channel = 3
    div = DIVISION[name]
    order = 1 if hisr else 0
    net_kw = dict(in_channel=image_n_channel, out_channel=image_n_channel, lms_channel=image_n_channel, pan_channel=add_n_channel, inner_channel=32,
                  norm_groups=1, channel_mults=(1, 2, 2, 4), attn_res=(8,), dropout=0.2, image_size=64, self_condition=True)
    net = UNetSR3(**net_kw).to(dev)
    if pretrain_weight is not None:
        sd = torch.load(pretrain_weight[0] if isinstance(pretrain_weight, (list, tuple)) else pretrain_weight, map_location="cpu")
        net.load_state_dict(sd.get("model", sd) if isinstance(sd, dict) and "model" in sd else sd, strict=isinstance(pretrain_weight, (list, tuple)))
    diffusion = GaussianDiffusion(net, image_size=image_size, channels=image_n_channel, pred_mode="x_start", loss_type="l1", device=dev, clamp_range=(0, 1))
    diffusion.set_new_noise_schedule(betas=make_beta_schedule(schedule="cosine", n_timestep=n_steps, cosine_s=8e-3), device=dev)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    train = _Batches(_open_set(train_dataset_path), batch_size, shuffle=True, rank=rank, world=world, seed=data_seed)
    valid = _Batches(_open_set(valid_dataset_path), 16, shuffle=False) if valid_dataset_path is not None else None
    params = [p for p in net.parameters()]
    _, grads = gradient_bucket(params)
    for p, g in zip(params, grads):
        p.grad = g  # autograd accumulates in place: the fused optimizer keeps these pointers
    if world > 1:
        broadcast_parameters(params)  # DDP: every replica starts from rank 0's initialisation
        net.mark_weights_dirty()
    ema = [p.detach().clone() for p in params]
    opt = _rt.FusedAdamW(params, grads, ema, lr=lr_d, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    iterations = int(pretrain_iterations) if pretrain_iterations is not None else 0
    history, records = [], []
    pending = []  # losses of the iterations since the last read-back (device scalars)

    def flush_losses():
        if pending:
            history.extend(float(v) for v in torch.stack(pending).cpu())
            pending.clear()

    resume_tiles = 0
    names = [n for n, _ in net.named_parameters()]
    if resume_state is not None:
        st = torch.load(resume_state, map_location="cpu", weights_only=False) if isinstance(resume_state, (str, os.PathLike)) else resume_state
        with torch.no_grad():
            for n, p, e in zip(names, params, ema):
                p.copy_(st["model"][n])
                e.copy_(st["ema"][n])
        net.mark_weights_dirty()
        opt.load_state_dict(st["optimizer"])
        iterations = int(st["iterations"])
        if "data" in st:
            train.load_state(st["data"])  # epoch in force, batch offset inside it, and the generator state its permutation came from
        else:
            train.epoch = int(st.get("epoch", 0))  # states written before the offset was recorded: exact only on an epoch boundary
        resume_tiles = int(st.get("tile_counter", 0))
        diffusion._mask_calls = int(st.get("mask_calls", 0))
        torch.set_rng_state(st["rng"]["torch"])
        _random.setstate(st["rng"]["python"])
        if dev.type == "cuda" and st["rng"].get("cuda") is not None:
            torch.cuda.set_rng_state(st["rng"]["cuda"], dev)

    # validation runs on its own network + diffusion (the reference's `ema_model` deep copy): ddim_sample_loop respaces ITS schedule
    # (3000 -> 25, in place, once); the training diffusion's schedule is never touched
    val = {}

    def validate():
        if not val:
            vnet = UNetSR3(**net_kw).to(dev).eval()
            vdiff = GaussianDiffusion(vnet, image_size=image_size, channels=image_n_channel, pred_mode="x_start", loss_type="l1", device=dev, clamp_range=(0, 1))
            vdiff.set_new_noise_schedule(betas=make_beta_schedule(schedule="cosine", n_timestep=n_steps, cosine_s=8e-3), device=dev)
            val["net"], val["diffusion"] = vnet, vdiff
        vnet, vdiff = val["net"], val["diffusion"]
        with torch.no_grad():
            for p, e in zip(vnet.parameters(), ema):
                p.copy_(e)
        vnet.mark_weights_dirty()
        pan, lms, gt = next(iter(valid))
        lms, pan, gt = lms.to(dev), pan.to(dev), gt.to(dev)
        cond = _rt.cond_assemble(lms, pan, div, wavelet_order=order)
        with torch.no_grad():
            sr = vdiff(cond, mode="ddim_sample", section_counts="ddim25")
            sr = (sr + cond[:, :image_n_channel]).clip(0, 1)
        gtn = (gt / div).contiguous()
        m = _rt.metrics(gtn, sr.contiguous(), ergas_ratio=4.0).mean(dim=0).cpu()
        ssim = float(_rt.ssim(gtn, sr.contiguous()).mean().cpu())
        return {"SAM": float(m[0]), "ERGAS": float(m[1]), "PSNR": float(m[2]), "CC": float(m[3]), "SSIM": ssim}

    def save(it):
        os.makedirs(save_dir, exist_ok=True)
        model_sd = {n: p.detach().cpu().clone() for n, p in zip(names, params)}
        ema_sd = {n: e.detach().cpu().clone() for n, e in zip(names, ema)}
        torch.save(model_sd, os.path.join(save_dir, f"diffusion_{name}_iter_{it}.pth"))      # bare state_dicts, reference :333-340
        torch.save(ema_sd, os.path.join(save_dir, f"ema_diffusion_{name}_iter_{it}.pth"))
        torch.save({"model": model_sd, "ema": ema_sd, "optimizer": opt.state_dict(), "iterations": it, "epoch": train.epoch, "data": train.state(),
                    "tile_counter": tile_counter, "mask_calls": int(getattr(diffusion, "_mask_calls", 0)),
                    "rng": {"torch": torch.get_rng_state(), "python": _random.getstate(),
                            "cuda": torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None}},
                   os.path.join(save_dir, f"train_state_{name}_iter_{it}.pth"))

    net.train()
    tile_counter = resume_tiles
    if world > 1:
        diffusion.train_mask_seed = (int(data_seed) + 1) * 1_000_003  # same mask stream on every rank, keyed by GLOBAL tile index below
    while iterations < max_iterations:
        for pan, lms, hr in train:
            pan, lms, hr = pan.to(dev), lms.to(dev), hr.to(dev)
            cond = _rt.cond_assemble(lms, pan, div, wavelet_order=order)
            lms_n = cond[:, :image_n_channel].contiguous()
            res = (hr / div - lms_n).contiguous()
            diffusion.train_tile0 = tile_counter * world + rank * res.shape[0]  # global index of this rank's first sample: masks do not depend on the split
            tile_counter += res.shape[0]
            diff_loss, recon_x = diffusion.train_step_into(res, cond, grads)  # gradients written straight into `grads` (no zeroing needed)
            if world > 1:
                average_gradients(grads, world)
            opt.lr = lr_at(iterations, lr_d)
            # EmaUpdater.update(iterations) runs BEFORE `iterations += 1` (reference :239-242): copy while that 0-based count <= start_iter, lerp after
            mode = 2 if iterations > ema_start_iter else 1
            # the reference prints the loss every iteration (one host read-back each); `log_every=k` reads the losses back k at a time instead,
            # so the stream is not drained in between (the step itself has no other synchronisation)
            want_log = log_every <= 1 or (iterations + 1) % log_every == 0 or iterations + 1 >= max_iterations
            gn = opt.step(max_grad_norm=0.003, ema_mode=mode, ema_decay=0.995, return_norm=want_log)
            iterations += 1
            net.mark_weights_dirty()  # the fused step wrote the parameters through raw pointers
            pending.append(diff_loss.detach())
            if want_log:
                flush_losses()
                log(f"[iter {iterations}/{max_iterations}: d_lr {opt.lr: .6f}] - denoise loss {history[-1]:.6f} (grad norm {gn:.4f})")
            if valid is not None and valid_every and iterations % valid_every == 0:
                flush_losses()
                rec = validate()
                records.append((iterations, rec))
                log(f"[iter {iterations}] validation: {rec}  (SSIM: PARITY-UNPINNED -- a restatement of skimage's defaults, skimage is not in this image; "
                    f"SAM / ERGAS / PSNR / CC are pinned to the reference's analysis_accu)")
            if save_dir and save_every and iterations % save_every == 0 and rank == 0:
                flush_losses()
                save(iterations)
            if iterations >= max_iterations:
                break
    flush_losses()
    net.eval()
    return {"loss": history, "validation": records, "model": net, "diffusion": diffusion, "ema": ema, "iterations": iterations,
            "optimizer": opt}
