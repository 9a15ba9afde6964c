"""ddif: MI355X-native drop-in for the DDIF (Dif-PAN) denoising hot path.

Module paths mirror the reference repository:
    ddif.models.sr3_dwt.UNetSR3                      <- models/sr3_dwt.py
    ddif.diffusion.diffusion_ddpm_pan.GaussianDiffusion, make_beta_schedule   <- diffusion/diffusion_ddpm_pan.py
    ddif.solver.dpm_solver.NoiseScheduleVP, model_wrapper, DPM_Solver         <- solver/dpm_solver.py
    ddif.diffusion_engine.engine_google, test_fn                              <- diffusion_engine.py
All tensor work runs in hand-written gfx950 kernels behind the C ABI of include/ddif.h (libddif.so).
"""
from .runtime import DdifError, get_lib, use_library  # noqa: F401

__all__ = ["DdifError", "get_lib", "use_library"]
