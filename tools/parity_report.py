#!/usr/bin/env python3
"""Print the measured parity margins of the GPU path against the committed golden vectors (same cases as
tests/test_gpu_parity.py, which only asserts the thresholds).  Run on a GPU box: python tools/parity_report.py [--full]
The math mode is whatever the environment selects (default: f16x2 split; DDIF_F16=0: bf16x3; DDIF_F16=0 DDIF_X3=0: exact fp32 MFMA) and is printed in the
header; tools/gpu_full.sh commits one report per mode under profiles/.  --full adds the 2000-step CAVE 128 x 128 chain."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "dif-pan_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import golden_cases as gc
import test_gpu_parity as T
from oracle import ddif_oracle as O

mode = "exact fp32 MFMA (DDIF_X3=0)" if os.environ.get("DDIF_X3") == "0" else ("bf16x3 split (DDIF_F16=0)" if os.environ.get("DDIF_F16") == "0" else "f16x2 split (default)")
print("math mode: %s" % mode)
print("every expected value below is the REAL reference's output (tools/make_golden.py, build container)")
print("case                          max |hip - reference|   (threshold)")
for case in gc.FORWARD_CASES:
    g = T._load(case[0])
    x, t, cond, sc = gc.forward_inputs(case)
    net = T.net_for(case[1])
    y = net(x.to(T.DEV), t.to(T.DEV), cond.to(T.DEV), None if sc is None else sc.to(T.DEV))
    print("%-28s  %.3e   (2e-5)" % (case[0], T._maxerr(y, torch.from_numpy(g["y"]))))
for case in gc.DDPM_CASES:
    g = T._load(case[0])
    out, outs, cond = T._run_ddpm(case, [])
    ref = torch.from_numpy(g["out"])
    C = gc.DATASETS[case[1]][0]
    lms = cond[:, :C]
    sr_hip, sr_ref = (out.cpu() + lms).clip(0, 1), (ref + lms).clip(0, 1)
    gt = gc.tiles_for(case[1], case[2], case[3], case[4], seed=case[6])["gt"]
    print("%-28s  %.3e   (1e-4)   PSNR diff %.2e dB (1e-3)" % (case[0], T._maxerr(out, ref), abs(O.psnr(sr_hip, gt) - O.psnr(sr_ref, gt))))

from ddif.solver.dpm_solver import DPM_Solver, ImageSpaceClamp, NoiseScheduleVP, model_wrapper  # noqa: E402
from ddif_testlib import make_diffusion, reference_noise_stream  # noqa: E402

for case in gc.DDPM_BIG_CASES:
    g = T._load(case[0])
    out, outs, cond = T._run_ddpm(case, [])
    ref = torch.from_numpy(g["out"])
    C = gc.DATASETS[case[1]][0]
    lms = cond[:, :C]
    sr_hip, sr_ref = (out.cpu() + lms).clip(0, 1), (ref + lms).clip(0, 1)
    gt = gc.tiles_for(case[1], case[2], case[3], case[4], seed=case[6])["gt"]
    print("%-28s  %.3e   (1e-4)   PSNR diff %.2e dB (1e-3)" % (case[0], T._maxerr(out, ref), abs(O.psnr(sr_hip, gt) - O.psnr(sr_ref, gt))))
for case in gc.DDIM_CASES:
    cid, ds, B, H, W, Tn, sect, seed = case
    g = T._load(cid)
    C = gc.DATASETS[ds][0]
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    d = make_diffusion(T.net_for(ds), C, Tn, H, T.DEV)
    n_keep = len(O.ddim_stride_set(Tn, sect))
    xT, noise = reference_noise_stream(seed, (B, C, H, W), n_keep)
    out = d(cond.to(T.DEV), mode="ddim_sample", section_counts=sect, x_T=xT.to(T.DEV), noise=noise.to(T.DEV))
    print("%-28s  %.3e   (1e-4)" % (cid, T._maxerr(out, torch.from_numpy(g["out"]))))
for case in gc.DPM_CASES + gc.DPM_BIG_CASES:
    cid, ds, H, W, Tn, steps, order, seed = case
    g = T._load(cid)
    C = gc.DATASETS[ds][0]
    cond = gc.tiles_for(ds, 1, H, W, seed=seed)["cond"].to(T.DEV)
    net = T.net_for(ds)
    d = make_diffusion(net, C, Tn, H, T.DEV)
    ns = NoiseScheduleVP("discrete", betas=d.betas)
    fn = model_wrapper(net, ns, model_type="x_start", guidance_type="classifier-free", guidance_scale=1.0, condition=cond)
    slv = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=ImageSpaceClamp(cond[:, :C].contiguous(), 0.0, 1.0))
    xT = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(seed))
    out = slv.sample(xT.to(T.DEV), steps=steps, order=order, skip_type="time_uniform", method="multistep")
    print("%-28s  %.3e   (1e-4)" % (cid, T._maxerr(out, torch.from_numpy(g["out"]))))
if "--full" in sys.argv:
    for case in gc.DDPM_FULL_CASES:
        cid, ds, B, H, W, Tn, seed = case
        g = T._load(cid)
        C = gc.DATASETS[ds][0]
        tiles = gc.tiles_for(ds, B, H, W, seed=seed)
        xT, noise = reference_noise_stream(seed, (B, C, H, W), Tn)
        d = make_diffusion(T.net_for(ds), C, Tn, H, T.DEV)
        out = d(tiles["cond"].to(T.DEV), mode="ddpm_sample", x_T=xT.to(T.DEV), noise=noise.to(T.DEV)).cpu()
        ref = torch.from_numpy(g["out"])
        lms = tiles["cond"][:, :C]
        sr_hip, sr_ref = (out + lms).clip(0, 1), (ref + lms).clip(0, 1)
        print("%-28s  %.3e   (1e-4)   PSNR diff %.2e dB (1e-3)" % (cid, T._maxerr(out, ref), abs(O.psnr(sr_hip, tiles["gt"]) - O.psnr(sr_ref, tiles["gt"]))))
