#!/usr/bin/env python3
"""Print the measured parity margins of the GPU path against the committed golden vectors (same cases as
tests/test_gpu_parity.py, which only asserts the thresholds).  Run on a GPU box: python tools/parity_report.py"""
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
