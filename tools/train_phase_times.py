"""Where a batch-32 training iteration spends its time (development aid, GPU only): the cond-only program, the train-mode forward, the
eval-mode forward of the same batch, the whole native step.  usage: python tools/train_phase_times.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dif-pan_amd"))

from ddif.models.sr3_dwt import UNetSR3  # noqa: E402
from ddif.synth import synth_tiles  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    C, P, H = 8, 1, 64
    torch.manual_seed(0)
    net = UNetSR3(in_channel=C, out_channel=C, lms_channel=C, pan_channel=P, inner_channel=32, norm_groups=1, channel_mults=(1, 2, 2, 4), attn_res=(8,), dropout=0.2,
                  image_size=64, self_condition=True).to(dev)
    tiles = synth_tiles(B, C, P, H, H, seed=1)
    cond = tiles["cond"].to(dev)
    x0 = torch.randn(B, C, H, H, device=dev)
    noise = torch.randn_like(x0)
    sc = torch.randn_like(x0)
    a = torch.full((B,), 0.8)
    s = torch.full((B,), 0.6)
    t = torch.full((B,), 0.5)
    out = {}
    # eval plan
    net.eval()
    pe = net.plan_for(B, H, H, dev)
    out["eval set_cond"] = timed(lambda: pe.set_cond(cond, force=True))
    out["eval q_sample_forward"] = timed(lambda: pe.q_sample_forward(x0, noise, a, s, t, sc))
    # train plan
    net.train()
    pt = net.plan_for(B, H, H, dev, train=True)
    out["train set_cond"] = timed(lambda: pt.set_cond(cond, force=True))
    out["train masks"] = timed(lambda: pt.random_train_masks(5, 0, 0.2, 0.2))
    out["train q_sample_forward"] = timed(lambda: pt.q_sample_forward(x0, noise, a, s, t, sc))
    named = [(n, torch.zeros_like(p)) for n, p in net.named_parameters()]
    pt.train_bind(named)
    net._net.refresh_from_device(net.named_parameters())  # fills the dgrad packs
    pt.set_cond(cond, force=True)
    out["device weight refresh"] = timed(lambda: net._net.refresh_from_device(net.named_parameters()))
    pt.set_cond(cond, force=True)
    out["train step (fwd + bwd)"] = timed(lambda: pt.train_step(x0, noise, a, s, t, sc, want_pred=False))
    # host side of the same call: how long the CPU needs to ISSUE one step's launches (the queue is drained first; no synchronisation inside)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pt.train_step(x0, noise, a, s, t, sc, want_pred=False)
    out["train step, host issue time"] = (time.perf_counter() - t0) / 5 * 1e3
    torch.cuda.synchronize()
    for k, v in out.items():
        print(f"{k:28s} {v:8.3f} ms")


if __name__ == "__main__":
    main()
