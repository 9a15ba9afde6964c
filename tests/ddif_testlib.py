"""Shared helpers of the parity tests: build the drop-in network on the GPU (real library) or on the CPU against
the host-emulated build of the same kernel sources (tests only)."""
from __future__ import annotations

import os
import subprocess

import torch

import golden_cases as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dif-pan_amd")
EMU_LIB = os.path.join(PKG, "lib", "libddif_emu.so")
HIP_LIB = os.path.join(PKG, "lib", "libddif.so")

CTOR_KEYS = ("in_channel", "out_channel", "inner_channel", "lms_channel", "pan_channel", "norm_groups",
             "channel_mults", "attn_res", "res_blocks", "dropout", "image_size", "self_condition")


def ensure_emu_lib() -> str:
    # DDIF_EMU_LIB: another host build of the same sources, e.g. the AddressSanitizer one (`make -C dif-pan_amd emu-asan`,
    # run under LD_PRELOAD of clang's asan runtime -- tools/asan_emu.sh): GPU sanitizers are not available on the pool
    override = os.environ.get("DDIF_EMU_LIB")
    if override:
        return override
    subprocess.run(["make", "-C", PKG, "-j8", "emu"], check=True, stdout=subprocess.DEVNULL)
    return EMU_LIB


def use_emulator():
    import ddif

    return ddif.use_library(ensure_emu_lib())


def use_gpu_library():
    import ddif
    from ddif import runtime

    if runtime.library_loaded_path() != HIP_LIB:
        ddif.use_library(HIP_LIB)
    lib = ddif.get_lib()
    assert not lib.emulated, "GPU tests must run the gfx950 library"
    assert torch.cuda.is_available(), "-m gpu tests need a GPU; there is no CPU fallback to fall through to"
    return lib


def make_net(ds: str, device):
    from ddif.models.sr3_dwt import UNetSR3

    cfg = gc.cfg_for(ds)
    net = UNetSR3(**{k: cfg[k] for k in CTOR_KEYS})
    net.load_state_dict(gc.weights_for(ds))
    return net.to(device).eval()


def make_diffusion(net, C: int, T: int, size: int, device):
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule

    d = GaussianDiffusion(net, image_size=size, channels=C, pred_mode="x_start", loss_type="l1", device=device,
                          clamp_range=(0, 1))
    d.set_new_noise_schedule(betas=make_beta_schedule(schedule="cosine", n_timestep=T, cosine_s=8e-3), device=device)
    return d


def reference_noise_stream(seed: int, shape, n_steps: int):
    """x_T and the per-step noise exactly as the reference's p_sample_loop / ddim_sample_loop draws them from the
    global CPU generator after torch.manual_seed(seed): one randn(shape) for x_T, then one per step."""
    torch.manual_seed(seed)
    xT = torch.randn(shape)
    noise = torch.stack([torch.randn(shape) for _ in range(n_steps)], dim=0)
    return xT, noise
