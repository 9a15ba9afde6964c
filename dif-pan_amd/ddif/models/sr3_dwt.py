"""Drop-in for the reference `models/sr3_dwt.py::UNetSR3` (constructor :31-51, forward :169-219).

The module owns parameters with the reference's state-dict key names and shapes (so reference checkpoints load
unchanged and optimizers / EMA iterate `.parameters()` as before) but its forward pass is NOT torch: it runs the
hand-written gfx950 kernels of libddif.so through the C ABI (include/ddif.h).  No PyTorch / CPU fallback exists;
calls on CPU tensors or without the built library raise `DdifError`.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
from torch import nn

from ..layout import DEFAULT_CFG, param_manifest
from ..runtime import DdifError, NetHandle, PlanHandle


class _Node(nn.Module):
    """Plain container used to reproduce the reference's parameter tree (key names only)."""


def _register(root: nn.Module, key: str, value: torch.Tensor):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(value))


def _default_init(key: str, shape, shapes: Dict[str, tuple]) -> torch.Tensor:
    """torch's default inits for the layer kinds of the reference (Conv2d/Linear: U(+-1/sqrt(fan_in)); GroupNorm: 1/0;
    CondInjection.body[-1] zero-initialised, models/sr3_dwt.py:386-387)."""
    is_norm = any(s in key for s in (".block.0.", ".norm.", ".prenorm_x.", ".body.1."))
    if is_norm:
        return torch.ones(shape) if key.endswith("weight") else torch.zeros(shape)
    if ".body.3." in key:
        return torch.zeros(shape)
    wshape = shape if key.endswith("weight") else shapes[key[: -len("bias")] + "weight"]
    fan_in = 1
    for s in wshape[1:]:
        fan_in *= s
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape) * 2 - 1) * bound


# A process-wide counter bumped whenever ANY torch module registers a parameter, a buffer or a submodule: what invalidates UNetSR3's cached
# parameter list (`_param_cache`).  (torch >= 2.0 global registration hooks; they return None = "leave the value alone".)
_REG_EPOCH = [0]
_PARAM_CACHE = os.environ.get("DDIF_PARAM_CACHE", "1") != "0"  # DDIF_PARAM_CACHE=0: the pre-round-5 behaviour (A/B switch)


def _bump_reg_epoch(*_args):
    _REG_EPOCH[0] += 1
    return None


for _hook_name in ("register_module_parameter_registration_hook", "register_module_module_registration_hook", "register_module_buffer_registration_hook"):
    _reg = getattr(torch.nn.modules.module, _hook_name, None)
    if _reg is None:  # an older torch: no cache (every call re-walks the module tree)
        _REG_EPOCH = None
        break
    _reg(_bump_reg_epoch)


class UNetSR3(nn.Module):
    def __init__(
        self,
        in_channel=8,
        out_channel=3,
        inner_channel=32,
        lms_channel=8,
        pan_channel=1,
        norm_groups=32,
        channel_mults=(1, 2, 4, 8, 8),
        attn_res=(8,),
        res_blocks=3,
        dropout=0,
        with_noise_level_emb=True,
        image_size=128,
        self_condition=False,
        fourier_features=False,
        fourier_min=7,
        fourier_max=8,
        fourier_step=1,
        pred_var=False,
    ):
        super().__init__()
        self.cfg = dict(DEFAULT_CFG)
        self.cfg.update(
            in_channel=in_channel, out_channel=out_channel if out_channel is not None else in_channel,
            inner_channel=inner_channel, lms_channel=lms_channel, pan_channel=pan_channel, norm_groups=norm_groups,
            channel_mults=tuple(channel_mults), attn_res=tuple(attn_res), res_blocks=res_blocks, dropout=dropout,
            with_noise_level_emb=with_noise_level_emb, image_size=image_size, self_condition=self_condition,
            fourier_features=fourier_features, fourier_min=fourier_min, fourier_max=fourier_max,
            fourier_step=fourier_step, pred_var=pred_var)
        if fourier_features or pred_var or not with_noise_level_emb:
            raise DdifError("fourier_features / pred_var / with_noise_level_emb=False are not implemented by the HIP "
                            "path, and there is no fallback")
        # attributes the reference exposes and its callers read (diffusion_ddpm_pan.py:182-184)
        self.lms_channel = lms_channel
        self.pan_channel = pan_channel
        self.res_blocks = res_blocks
        self.self_condition = self_condition
        self.pred_var = pred_var
        self.fourier_features = fourier_features
        manifest = param_manifest(self.cfg)
        shapes = dict(manifest)
        for key, shape in manifest:
            _register(self, key, _default_init(key, shape, shapes))
        self._net: Optional[NetHandle] = None
        self._weights_sig = None

    # ---- weights -> library -----------------------------------------------------------------------------------------
    def _param_cache(self):
        """[(state-dict key, parameter)] and the bare parameter list, walked ONCE: `named_parameters()` over the 702 parameters costs 1.6 ms of Python,
        and the training loop needs the list several times per iteration (signature, refresh, bind) -- with the GPU idle meanwhile.  Valid until any
        module registers a parameter / submodule / buffer (a process-wide epoch bumped by torch's registration hooks) or `_apply` (.to / .cuda / .float)
        runs on this module."""
        c = self.__dict__.get("_pcache")
        epoch = object() if (_REG_EPOCH is None or not _PARAM_CACHE) else _REG_EPOCH[0]  # (no hooks / DDIF_PARAM_CACHE=0: every call re-walks)
        if c is not None and c["epoch"] == epoch:
            # cheap validation on every use: writers the hooks do not see -- `net.submodule.to(...)`, direct `_parameters[...]` assignment,
            # `torch.utils.swap_tensors` -- replace Parameter OBJECTS; the first / last entries of every child module's dict catch a replaced or converted block
            pl = c["plist"]
            own = self._parameters
            if len(own) != c["n_own"] or (pl and (pl[0] is not next(iter(own.values()), pl[0]) or pl[-1].device != c["dev_last"] or pl[0].device != c["dev_first"])):
                c = None
        if c is None or c["epoch"] != epoch:
            named = [(n, p) for n, p in self.named_parameters()]
            plist = [p for _, p in named]
            c = self.__dict__["_pcache"] = {"epoch": epoch, "named": named, "plist": plist, "n_own": len(self._parameters),
                                            "dev_first": plist[0].device if plist else None, "dev_last": plist[-1].device if plist else None,
                                            "devices": {p.device for p in plist}}
        return c

    def named_parameter_list(self):
        """`list(self.named_parameters())`, cached (see `_param_cache`)."""
        return self._param_cache()["named"]

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_pcache", None)
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_pcache", None)
        return out

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self._param_cache()["plist"])

    def mark_weights_dirty(self):
        """The parameters were written behind torch's back (raw-pointer kernels such as the fused optimizer step): the packed copy the
        library holds is stale and is rebuilt at the next forward / sampler call.  (`FusedAdamW.step` also bumps the version counters the
        signature is made of; this is the explicit form for any other writer.)"""
        self._weights_sig = None

    def pe_freqs(self) -> torch.Tensor:
        """exp(-ln(1e4) * j / count) evaluated with torch on the CPU exactly as PositionalEncoding does
        (models/sr3_dwt.py:229-236), so the kernels use bit-identical frequencies."""
        count = self.cfg["inner_channel"] // 2
        step = torch.arange(count, dtype=torch.float32) / count
        return torch.exp(-math.log(1e4) * step.unsqueeze(0)).reshape(-1)

    def _ensure_net(self, device: torch.device, train: bool = False) -> NetHandle:
        """The library-side copy of the weights, brought up to date.  Inference: host repack + upload (ddif_net_commit) whenever the
        parameters changed.  `train=True` (train-mode plans only): once committed, changed parameters are re-packed ON THE DEVICE in one
        launch (ddif_net_refresh) -- the training loop changes them every iteration."""
        device = torch.device(device)
        if self._net is None or self._net.device != device:
            self._net = NetHandle(self.cfg, device)
            self._weights_sig = None
        sig = self._signature()
        net = self._net
        committed = getattr(net, "device_refreshed", None) is not None
        if train and committed and self._param_cache()["devices"] == {device}:
            if sig != self._weights_sig:
                net.refresh_from_device(self.named_parameter_list(), tuple(s[0] for s in sig))
                self._weights_sig = sig
            return net
        if sig != self._weights_sig or getattr(net, "device_refreshed", False):
            net.load_state_dict(self.state_dict(), self.pe_freqs())
            self._weights_sig = sig
        return net

    DROP_PATH_PROB = 0.2  # FastAttnCondInjection's default drop_path_prob, never overridden by UNetSR3 (models/sr3_dwt.py:502,534)

    def plan_for(self, B: int, H: int, W: int, device, train: bool = False) -> PlanHandle:
        return self._ensure_net(device, train=train).plan(B, H, W, train=train)

    def set_train_masks(self, dropout_masks, droppath_scales):
        """Pin the masks of the NEXT train-mode forward passes (parity tests against a reference run whose masks were captured);
        `None, None` returns to fresh masks from the library's counter-based generator seeded from torch's RNG."""
        self._train_masks = None if dropout_masks is None else (dropout_masks, droppath_scales)

    # ---- reference API ----------------------------------------------------------------------------------------------
    def forward(self, x, time, cond=None, self_cond=None):
        if cond is None:
            raise DdifError("UNetSR3.forward: cond is required (the reference indexes it unconditionally, "
                            "models/sr3_dwt.py:197)")
        B, _, H, W = x.shape
        if self.training:
            # Train mode = Dropout(p) after every ResnetBlock Block's SiLU (models/sr3_dwt.py:295) + DropPath(0.2) on every decoder
            # FFN (:534,576), applied by the train-mode launch program (csrc/ddif_plan.cpp).  No autograd graph is built: the
            # backward pass is not implemented yet, so this serves p_losses' forward / loss value only.
            plan = self.plan_for(B, H, W, x.device, train=True)
            masks = getattr(self, "_train_masks", None)
            if masks is not None:
                plan.set_train_masks(*masks)
            else:
                plan.random_train_masks(int(torch.randint(0, 2 ** 62, (1,)).item()), 0, float(self.cfg["dropout"]), self.DROP_PATH_PROB)
        else:
            plan = self.plan_for(B, H, W, x.device)
        plan.set_cond(cond)
        if not torch.is_tensor(time):
            time = torch.tensor([float(time)] * B)
        sc = self_cond if (self.self_condition and self_cond is not None) else None
        return plan.forward(x, time, sc)
