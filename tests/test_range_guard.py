"""Range behaviour of the f16x2 split (the default conv arithmetic of inference plans, csrc/ddif_dev.h) with TRAINED-LIKE magnitudes -- every other
parity test runs on small seeded random-init weights (ddif/synth.py).  The reference loads any checkpoint (utils/misc.py:89-122) and computes in
fp32, so this library must stay at fp32-class accuracy, or fall back, never emit NaN:

  * GroupNorm gains of 0.5 .. 8, conv weights x 20 on a third of the Block convs, a cond image x 4: forward + a 10-step DDPM chain against the oracle;
  * the two host-side guards (|w| >= 64 at commit; sqrt(N) max|gamma| + max|beta| >= 4094 at plan build) keep a conv on bf16x3;
  * a RAW conv input (no GroupNorm in front: the decoder's feed-forward pair) pushed past 4094: the plan's range watch fires
    (ddif_plan_range_status), the Python layer rebuilds the plan with those convs on bf16x3 and repeats the call -- finite, and equal to the oracle.

Each case runs on the host emulator (CPU) and, with -m gpu, on the MI355X (the guards had only ever run on the emulator)."""
import warnings

import pytest
import torch

import golden_cases as gc
from ddif_testlib import CTOR_KEYS, make_diffusion, use_emulator, use_gpu_library
from oracle import ddif_oracle as O

BACKENDS = [pytest.param("emu", id="emulated"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _dev(backend):
    if backend == "emu":
        use_emulator()
        return torch.device("cpu")
    use_gpu_library()
    return torch.device("cuda:0")


def _net(sd, ds, dev):
    from ddif.models.sr3_dwt import UNetSR3

    cfg = gc.cfg_for(ds)
    net = UNetSR3(**{k: cfg[k] for k in CTOR_KEYS})
    net.load_state_dict(sd)
    return net.to(dev).eval()


def trained_like(ds, seed, gamma_hi=8.0, conv_gain=20.0, ffn_gain=1.0):
    """The seeded fixture weights with trained-like magnitudes: every GroupNorm gain x U(0.5, gamma_hi), every third Block conv (the 3x3 behind a
    GroupNorm + SiLU) x conv_gain, and the decoder's attention-mix convs (attn_out / attn_res: their sum is the RAW input of the feed-forward pair)
    x ffn_gain."""
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in gc.weights_for(ds).items()}
    k3 = 0
    for k in sorted(sd):
        v = sd[k]
        if v.dim() == 1 and k.endswith(".weight"):  # GroupNorm gains (Linear / conv weights have more dimensions)
            sd[k] = v * (0.5 + (gamma_hi - 0.5) * torch.rand(v.shape, generator=g))
        elif k.endswith(".block.3.weight"):
            k3 += 1
            if k3 % 3 == 0:
                sd[k] = v * conv_gain
        elif ffn_gain != 1.0 and (k.endswith("cond_inj.attn_out.weight") or k.endswith("cond_inj.attn_res.weight") or k.endswith("cond_inj.attn_out.bias") or k.endswith("cond_inj.attn_res.bias")):
            sd[k] = v * ffn_gain
    return sd


def _inputs(ds, B, H, seed, cond_gain=4.0):
    C = gc.DATASETS[ds][0]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, H, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    cond = gc.tiles_for(ds, B, H, H, seed=seed + 1)["cond"] * cond_gain
    return C, x, t, cond


@pytest.mark.parametrize("backend", BACKENDS)
def test_forward_and_ddpm_with_trained_like_magnitudes_match_the_oracle(backend):
    dev = _dev(backend)
    ds, B, H = "wv3", (1 if backend == "emu" else 2), (16 if backend == "emu" else 32)
    sd = trained_like(ds, 5)
    C, x, t, cond = _inputs(ds, B, H, 31)
    net = _net(sd, ds, dev)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no range fallback is expected here: the watch must stay silent
        y = net(x.to(dev), t.to(dev), cond.to(dev)).cpu()
    with torch.no_grad():
        ref = O.unet_forward(sd, gc.cfg_for(ds), x, t, cond, None)
    assert torch.isfinite(y).all()
    assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    plan = net.plan_for(B, H, H, dev)
    assert plan.f16_raw and plan.range_fallbacks == 0
    # a 10-step DDPM chain (clamped x_0-prediction: the chain sees the large internal magnitudes at every step)
    T, steps = 50, 10
    d = make_diffusion(net, C, T, H, dev)
    g = torch.Generator().manual_seed(8)
    xT = torch.randn(B, C, H, H, generator=g)
    noise = torch.randn(steps, B, C, H, H, generator=g)
    p2 = d._plan(cond.to(dev))
    c1, c2 = d.posterior_mean_coef1.cpu(), d.posterior_mean_coef2.cpu()
    cz = (0.5 * d.posterior_log_variance_clipped.cpu()).exp()
    order = list(reversed(range(T)))[:steps]
    out = p2.sample_ddpm([float(i) for i in order], [float(c1[i]) for i in order], [float(c2[i]) for i in order], [float(cz[i]) for i in order],
                         xT.to(dev), noise.to(dev).contiguous(), 0, 0, (0.0, 1.0), dev).cpu()
    it = iter([xT] + [noise[k] for k in range(steps)])
    with torch.no_grad():
        ref = O.ddpm_sample(sd, gc.cfg_for(ds), cond, O.schedule_tables(O.cosine_betas(T)), noise_fn=lambda s: next(it), timesteps=order)
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max()) <= 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
def test_host_side_guards_keep_a_conv_on_bf16x3(backend):
    """|w| >= 64 (commit) and sqrt(N) max|gamma| + max|beta| >= 4094 (plan build): both convs stay on bf16x3 and the forward stays at fp32-class
    accuracy (outputs of O(100): relative bar).  Until round 5 this ran on the emulator only."""
    dev = _dev(backend)
    ds, B, H = "wv3", 1, 16
    C, x, t, cond = _inputs(ds, B, H, 21, cond_gain=1.0)
    for key, factor in (("downs.1.res_block.block1.block.0.weight", 500.0), ("downs.1.res_block.block2.block.3.weight", 100.0)):
        sd = {k: v.clone() for k, v in gc.weights_for(ds).items()}
        sd[key] = sd[key] * factor
        net = _net(sd, ds, dev)
        y = net(x.to(dev), t.to(dev), cond.to(dev)).cpu()
        with torch.no_grad():
            ref = O.unet_forward(sd, gc.cfg_for(ds), x, t, cond, None)
        assert torch.isfinite(y).all()
        assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), key


@pytest.mark.parametrize("backend", BACKENDS)
def test_raw_activation_beyond_the_half_range_falls_back_instead_of_nan(backend):
    """attn_out / attn_res x 3000 put the feed-forward pair's RAW input far past 4094: under f16x2 the staged half is inf and the image NaN.  The
    range watch must fire, the plan must be rebuilt on bf16x3 for raw-input convs, and the repeated call must match the oracle."""
    dev = _dev(backend)
    ds, B, H = "wv3", (1 if backend == "emu" else 2), 16
    sd = trained_like(ds, 9, gamma_hi=2.0, conv_gain=1.0, ffn_gain=3000.0)
    C, x, t, cond = _inputs(ds, B, H, 41, cond_gain=1.0)
    net = _net(sd, ds, dev)
    with pytest.warns(RuntimeWarning, match="left the f16x2 range"):
        y = net(x.to(dev), t.to(dev), cond.to(dev)).cpu()
    with torch.no_grad():
        ref = O.unet_forward(sd, gc.cfg_for(ds), x, t, cond, None)
    assert torch.isfinite(ref).all() and float(ref.abs().max()) < 1e30
    assert torch.isfinite(y).all()
    assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    plan = net.plan_for(B, H, H, dev)
    assert plan.range_fallbacks == 1 and not plan.f16_raw
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the rebuilt plan runs without another fallback
        y2 = net(x.to(dev), t.to(dev), cond.to(dev)).cpu()
    assert torch.equal(y, y2)
