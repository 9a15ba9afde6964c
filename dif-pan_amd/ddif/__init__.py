"""ddif: MI355X-native drop-in for the DDIF (Dif-PAN) denoising hot path.

Module paths mirror the reference repository:
    ddif.models.sr3_dwt.UNetSR3                      <- models/sr3_dwt.py
    ddif.diffusion.diffusion_ddpm_pan.GaussianDiffusion, make_beta_schedule   <- diffusion/diffusion_ddpm_pan.py
    ddif.solver.dpm_solver.NoiseScheduleVP, model_wrapper, DPM_Solver         <- solver/dpm_solver.py
    ddif.diffusion_engine.engine_google, test_fn                              <- diffusion_engine.py
All tensor work runs in hand-written gfx950 kernels behind the C ABI of include/ddif.h (libddif.so).
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: the first scalar loads of every launch (its argument block) stop being a PCIe
# round trip -- 1.5 % of a denoising step of 146 dependent launches (profiles/r05/a_kernarg_ab.txt).  Read by the HIP runtime when it initialises,
# i.e. effective when this package is imported before the first GPU call of the process; a user's own setting wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .runtime import DdifError, get_lib, use_library  # noqa: F401

__all__ = ["DdifError", "get_lib", "use_library"]
