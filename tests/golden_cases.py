"""Seeded case definitions shared by tools/make_golden.py (which runs the real reference, build container
only) and the parity tests (which run the oracle / the HIP path).  Only seeds and shapes live here; inputs are
regenerated bit-identically on any machine with the same torch build from CPU generators."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dif-pan_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

from ddif.layout import engine_cfg  # noqa: E402
from ddif.synth import synth_state_dict, synth_tiles  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 1234

DATASETS = {  # name -> (C, P, wavelet order)
    "wv3": (8, 1, "pan"),
    "gf2": (4, 1, "pan"),
    "cave": (31, 3, "hisr"),
}


def cfg_for(name: str) -> dict:
    C, P, _ = DATASETS[name]
    return engine_cfg(C, P)


def weights_for(name: str):
    return synth_state_dict(cfg_for(name), WEIGHT_SEED)


def tiles_for(name: str, B: int, H: int, W: int, seed: int = 7):
    C, P, order = DATASETS[name]
    return synth_tiles(B, C, P, H, W, seed=seed, order=order)


# whole-forward cases: (case id, dataset, B, H, W, t values, float_t, with_self_cond)
FORWARD_CASES = [
    ("fwd_wv3_16_a", "wv3", 2, 16, 16, [0, 999], False, False),
    ("fwd_wv3_16_b", "wv3", 2, 16, 16, [5, 321], False, True),
    ("fwd_wv3_16_f", "wv3", 2, 16, 16, [17.25, 998.001], True, False),
    ("fwd_wv3_64", "wv3", 1, 64, 64, [5], False, True),
    ("fwd_wv3_24x40", "wv3", 1, 24, 40, [77], False, False),
    ("fwd_gf2_32", "gf2", 2, 32, 32, [3, 499], False, False),
    ("fwd_cave_32", "cave", 1, 32, 32, [1234], False, True),
]


def forward_inputs(case):
    cid, ds, B, H, W, tvals, float_t, with_sc = case
    C, P, _ = DATASETS[ds]
    g = torch.Generator().manual_seed(zlib_seed(cid))
    x = torch.randn(B, C, H, W, generator=g)
    sc = torch.randn(B, C, H, W, generator=g) if with_sc else None
    t = torch.tensor(tvals, dtype=torch.float32 if float_t else torch.long)
    cond = tiles_for(ds, B, H, W, seed=zlib_seed(cid) % 1000)["cond"]
    return x, t, cond, sc


def zlib_seed(s: str) -> int:
    import zlib

    return zlib.crc32(s.encode()) & 0x7FFFFFFF


SCHEDULE_T = [25, 500, 1000, 2000]
DDIM_FROM = [500, 1000]

# sampler cases: (case id, dataset, B, H, W, T, seed)
DDPM_CASES = [
    ("ddpm_wv3_16_T10", "wv3", 2, 16, 16, 10, 11),
    ("ddpm_wv3_16_T1000", "wv3", 1, 16, 16, 1000, 12),
    ("ddpm_wv3_64_T1000", "wv3", 1, 64, 64, 1000, 13),
    ("ddpm_gf2_32_T50", "gf2", 2, 32, 32, 50, 14),
]
DDPM_SNAPSHOTS = {"ddpm_wv3_16_T10": [1, 2, 10]}

# round 6: config-exact goldens from the real reference.  Kept out of DDPM_CASES / DPM_CASES so that the CPU oracle suite does not
# re-run minutes of 64 x 64 chains; the GPU suite compares the HIP path with them directly.
#   two more BASELINE configs[1] tiles (different cond / noise realisations) for the B = 64, T = 1000 test
DDPM_BIG_CASES = [
    ("ddpm_wv3_64_T1000_b", "wv3", 1, 64, 64, 1000, 15),
    ("ddpm_wv3_64_T1000_c", "wv3", 1, 64, 64, 1000, 16),
]
#   BASELINE configs[3] (CAVE, 128 x 128, T = 2000): the reference's full 2000-step chain, final `out` only
DDPM_FULL_CASES = [
    ("ddpm_cave_128_T2000", "cave", 1, 128, 128, 2000, 53),
]

DDIM_CASES = [  # (case id, dataset, B, H, W, T, section_counts, seed)
    ("ddim_wv3_32_T500_25", "wv3", 1, 32, 32, 500, "ddim25", 21),
    ("ddim_gf2_16_T1000_25", "gf2", 2, 16, 16, 1000, "ddim25", 22),
]

DPM_CASES = [  # (case id, dataset, H, W, T, steps, order, seed)   (B = 1: reference broadcast bug, SURVEY D-8)
    ("dpm_gf2_32_T1000_s10_o2", "gf2", 32, 32, 1000, 10, 2, 31),
    ("dpm_gf2_32_T1000_s50_o2", "gf2", 32, 32, 1000, 50, 2, 32),
    ("dpm_wv3_16_T500_s12_o3", "wv3", 16, 16, 500, 12, 3, 33),
    ("dpm_wv3_16_T500_s6_o3", "wv3", 16, 16, 500, 6, 3, 34),
]

# BASELINE configs[2] at its benchmarked tile size (GF2 64 x 64, DPM-Solver++ 2M, 50 evaluations), B = 1 (SURVEY D-8)
DPM_BIG_CASES = [
    ("dpm_gf2_64_T1000_s50_o2", "gf2", 64, 64, 1000, 50, 2, 36),
]

DPM_SKIP_CASES = [  # (case id, dataset, H, W, T, steps, order, seed, skip_type)
    ("dpm_wv3_16_T500_s10_o2_logsnr", "wv3", 16, 16, 500, 10, 2, 35, "logSNR"),
]

# BASELINE config 4 (CAVE, T = 2000): truncated DDPM runs -- the first / last `n` steps of the T-step loop
# (case id, dataset, B, H, W, T, which, n, seed)
DDPM_TRUNC_CASES = [
    ("ddpm_cave_64_T2000_first20", "cave", 1, 64, 64, 2000, "first", 20, 51),
    ("ddpm_cave_64_T2000_last20", "cave", 1, 64, 64, 2000, "last", 20, 52),
]

# big forward case of config 4's shape (multi-tile stem with scalar staging, C = 31 scalar-output epilogue, 256 tokens)
FORWARD_BIG_CASES = [
    ("fwd_cave_128", "cave", 1, 128, 128, [1500], False, True),
]

# train-mode forward (Dropout 0.2 + DropPath 0.2 live) with the reference's own masks captured: (case id, dataset, B, H, W, t values, seed)
TRAIN_FWD_CASES = [
    ("trainfwd_wv3_16", "wv3", 2, 16, 16, [7, 431], 61),
]

# one training forward + backward of the reference (tools/make_golden.py traingrad): grad norms of all parameters + these full gradients
TRAIN_GRAD_CASES = [
    ("traingrad_wv3_16", "wv3", 2, 16, 16, [7, 431], 62),
]
TRAIN_GRAD_FULL = ["downs.0.", "downs.1.cond_inj.body.0.weight", "downs.1.res_block.block2.block.3.weight", "downs.1.res_block.noise_func.noise_func.0.weight",
                   "noise_level_mlp.1.weight", "mid.0.attn.qkv.weight", "ups.0.cond_inj.q.0.weight", "ups.0.cond_inj.kv.1.weight", "ups.0.cond_inj.ffn.3.weight",
                   "ups.0.cond_inj.prenorm_x.weight", "final_conv.block.3.weight"]

LOSS_CASES = [  # (case id, dataset, B, H, W, T, t values, self-cond branch, seed)
    ("loss_wv3_16_sc0", "wv3", 2, 16, 16, 500, [3, 444], False, 41),
    ("loss_wv3_16_sc1", "wv3", 2, 16, 16, 500, [100, 7], True, 42),
]
