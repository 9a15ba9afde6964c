"""Kernels either side of the denoising loop (csrc/kernels_aux.h) against their CPU
restatements in the oracle:
  cond assembly + level-1 Haar (reference dataset/pan_dataset.py:73-81,127-142, diffusion_engine.py:221-228),
  validation metrics SAM / ERGAS / PSNR / CC (utils/_metric_legacy.py:299-379; golden from the reference's own function),
  fused clip + AdamW + EMA (diffusion_engine.py:237-241; torch's CPU implementations are the reference's).
Each test runs on the CPU against the host-emulated build of the same kernel sources and, with -m gpu, on the MI355X."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
import ddif_testops
from ddif_testlib import use_emulator, use_gpu_library
from oracle import ddif_oracle as O

BACKENDS = [pytest.param("emu", id="emulated"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _dev(backend):
    if backend == "emu":
        use_emulator()
        return "cpu"
    use_gpu_library()
    return "cuda:0"


def _raw_pair(B, C, P, H, W, seed, division):
    g = torch.Generator().manual_seed(seed)
    lms = (torch.rand(B, C, H, W, generator=g) * division).round()  # raw sensor counts
    pan = (torch.rand(B, P, H, W, generator=g) * division).round()
    return lms, pan


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ds,shape", [("wv3", (2, 16, 16)), ("gf2", (1, 8, 8)), ("cave", (1, 12, 12))])  # square: the reference resizes to size=lms.shape[-1]
def test_cond_assemble_matches_oracle(backend, ds, shape):
    from ddif import runtime

    dev = _dev(backend)
    C, P, order = gc.DATASETS[ds]
    B, H, W = shape
    division = {"wv3": 2047.0, "gf2": 1023.0, "cave": 1.0}[ds]
    lms, pan = _raw_pair(B, C, P, H, W, 11, division if division > 1 else 1.0)
    if division == 1.0:
        lms, pan = lms * 0 + torch.rand(lms.shape, generator=torch.Generator().manual_seed(1)), pan * 0 + torch.rand(pan.shape, generator=torch.Generator().manual_seed(2))
    want = O.assemble_cond(lms, pan, division, hisr_order=(order == "hisr"))
    got = runtime.cond_assemble(lms.to(dev), pan.to(dev), division, 1 if order == "hisr" else 0).cpu()
    assert got.shape == want.shape == (B, 2 * C + 4 * P, H, W)
    assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
def test_cond_assemble_equals_the_fixture_generator(backend):
    """synth_tiles() builds cond from normalised data with torch ops (what every parity fixture uses): the kernel fed with
    the same data (division 1) reproduces it."""
    from ddif import runtime
    from ddif.synth import synth_tiles

    dev = _dev(backend)
    t = synth_tiles(2, 8, 1, 16, 16, seed=3)
    got = runtime.cond_assemble(t["lms"].to(dev), t["pan"].to(dev), 1.0, 0).cpu()
    assert float((got - t["cond"]).abs().max()) <= 1e-6


def test_metrics_oracle_matches_reference_golden():
    """oracle.analysis_accu against the values the reference's own analysis_accu produced (tools/make_golden.py, psnr.npz)."""
    g = np.load(os.path.join(gc.GOLDEN_DIR, "psnr.npz"))
    gen = torch.Generator().manual_seed(5)
    a = torch.rand(8, 33, 35, generator=gen)
    b = (a + 0.05 * torch.randn(8, 33, 35, generator=gen)).clamp(0, 1)
    m = O.analysis_accu(a.permute(1, 2, 0), b.permute(1, 2, 0), 4)
    assert abs(m["PSNR"] - float(g["ref_psnr"])) <= 1e-4
    assert abs(m["SAM"] - float(g["sam"])) <= 1e-4
    assert abs(m["ERGAS"] - float(g["ergas"])) <= 1e-4
    if "cc" in g:
        assert abs(m["CC"] - float(g["cc"])) <= 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_metrics_kernel_matches_oracle(backend):
    from ddif import runtime

    dev = _dev(backend)
    gen = torch.Generator().manual_seed(6)
    gt = torch.rand(3, 8, 33, 35, generator=gen)
    pred = (gt + 0.05 * torch.randn(3, 8, 33, 35, generator=gen)).clamp(0, 1)
    pred[2] = gt[2]  # identical pair: SAM 0, rmse 0 -> PSNR = -inf in the reference's sign, ERGAS 0
    got = runtime.metrics(gt.to(dev), pred.to(dev), 4.0).cpu()
    for b in range(2):
        m = O.analysis_accu(gt[b].permute(1, 2, 0), pred[b].permute(1, 2, 0), 4)
        want = torch.tensor([m["SAM"], m["ERGAS"], m["PSNR"], m["CC"]])
        assert torch.allclose(got[b], want, rtol=2e-5, atol=2e-5), (got[b], want)
    m = O.analysis_accu(gt[2].permute(1, 2, 0), pred[2].permute(1, 2, 0), 4)
    assert abs(float(got[2, 1]) - m["ERGAS"]) <= 1e-6 and float(got[2, 2]) == m["PSNR"] == float("-inf")
    assert abs(float(got[2, 0]) - m["SAM"]) <= 0.05  # acos near 1 amplifies fp32 rounding of the normalised dot product


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_adamw_clip_ema_matches_torch(backend):
    from ddif import runtime

    dev = _dev(backend)
    gen = torch.Generator().manual_seed(7)
    shapes = [(64, 32, 3, 3), (64,), (5000,), (3, 7), (1,)]
    params = [torch.randn(s, generator=gen) * 0.1 for s in shapes]
    steps = 4
    grads = [[torch.randn(s, generator=gen) * (0.01 if k % 2 else 1e-5) for s in shapes] for k in range(steps)]  # clipped and unclipped steps
    want_p, want_e, norms = O.optimizer_steps(params, grads, lr=1e-3, weight_decay=1e-2, max_norm=0.003, ema_decay=0.9, ema_start_iter=1)
    p = [t.clone().to(dev) for t in params]
    g = [torch.zeros_like(t) for t in p]
    e = [torch.zeros_like(t) for t in p]
    opt = runtime.FusedAdamW(p, g, e, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for it in range(steps):
        for gi, src in zip(g, grads[it]):
            gi.copy_(src)
        gn = opt.step(max_grad_norm=0.003, ema_mode=2 if it > 1 else 1, ema_decay=0.9, return_norm=True)
        assert abs(gn - norms[it]) <= 1e-6 * max(1.0, norms[it])
    for a, b in zip(p, want_p):
        assert float((a.cpu() - b).abs().max()) <= 2e-7
    for a, b in zip(e, want_e):
        assert float((a.cpu() - b).abs().max()) <= 2e-7


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 32, 32, 16, 16), (1, 64, 32, 8, 8), (2, 16, 48, 12, 20), (3, 8, 8, 5, 7), (2, 12, 40, 6, 24), (1, 32, 64, 5, 16), (1, 8, 16, 3, 128)],
                         ids=["32-32@16", "64-32@8", "16-48@12x20", "8-8@5x7", "12-40@6x24", "32-64@5x16", "8-16@3x128"])
def test_conv3x3_backward_matches_autograd(backend, shape):
    """dgrad / wgrad / dbias of the 3x3 conv (first building block) against torch autograd on the CPU in
    fp32 -- the reference's own backward IS autograd through nn.Conv2d (diffusion_engine.py:233).  Floating point: split-operand
    (bf16x3) MFMAs with fp32 accumulation and a different summation order; tolerance 2e-5 relative to the gradient's scale.
    Widths that are multiples of 8 (<= 128) take the bf16x3 weight-gradient kernel (kernels_bwd.h conv3x3_wgrad_x3_kernel): 24 = an odd number of
    8-pixel segments, 5 rows = a ragged last band, 12 / 40 channels = partial 32-channel blocks, 128 = the widest row it stages; 20 and 7 take the
    exact-fp32 kernel."""
    from ddif import runtime

    dev = _dev(backend)
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).requires_grad_()
    b = torch.zeros(Cout, requires_grad=True)
    dy = torch.randn(B, Cout, H, W, generator=g)
    y = torch.nn.functional.conv2d(x, w, b, padding=1)
    y.backward(dy)
    op = ddif_testops.Conv3x3Backward(B, Cin, Cout, H, W, dev)
    dx, dw, db = op(x.detach().to(dev), w.detach().to(dev), dy.to(dev))
    for got, want, nm in ((dx, x.grad, "dx"), (dw, w.grad, "dw"), (db, b.grad, "db")):
        err = float((got.cpu() - want).abs().max())
        assert err <= 2e-5 * max(1.0, float(want.abs().max())), (nm, err, float(want.abs().max()))
    dx2, dw2, db2 = op(x.detach().to(dev), w.detach().to(dev), dy.to(dev))
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)  # fixed-order reductions: bitwise reproducible


def _block_forward(x, gamma, beta, mask, w, b):
    """`Block.forward` of the reference (models/sr3_dwt.py:288-300) with the Dropout mask made explicit."""
    a = torch.nn.functional.silu(torch.nn.functional.group_norm(x, 1, gamma, beta, eps=1e-5))
    if mask is not None:
        a = a * mask
    return torch.nn.functional.conv2d(a, w, b, padding=1)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape,drop", [((2, 32, 32, 16, 16), 0.2), ((1, 64, 32, 8, 8), 0.0), ((2, 16, 48, 12, 20), 0.2), ((3, 8, 8, 5, 7), 0.2)],
                         ids=["32-32@16-drop", "64-32@8-eval", "16-48@12x20-drop", "8-8@5x7-drop"])
def test_block_backward_matches_autograd(backend, shape, drop):
    """Backward of a whole `Block` (GroupNorm(1) -> Swish -> Dropout -> conv3x3; SURVEY 8(a) a4 under a15) against torch autograd
    on the CPU in fp32, with the dropout mask pinned.  Tolerance 3e-5 relative to each gradient's scale (fp32, different
    summation order; the GroupNorm sums here are fp64)."""
    from ddif import runtime

    dev = _dev(backend)
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x = (torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.3).requires_grad_()
    gamma = (1.0 + 0.2 * torch.randn(Cin, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(Cin, generator=g)).requires_grad_()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).requires_grad_()
    b = torch.zeros(Cout, requires_grad=True)
    mask = None
    if drop > 0:
        mask = (torch.rand(B, Cin, H, W, generator=g) >= drop).float() / (1.0 - drop)
    dy = torch.randn(B, Cout, H, W, generator=g)
    _block_forward(x, gamma, beta, mask, w, b).backward(dy)
    op = ddif_testops.BlockBackward(B, Cin, Cout, H, W, dev)
    args = [t.detach().to(dev) for t in (x, gamma, beta, w, dy)]
    got = op(*args, mask=None if mask is None else mask.to(dev))
    want = {"dx": x.grad, "dgamma": gamma.grad, "dbeta": beta.grad, "dw": w.grad, "db": b.grad, "dy_plane_sums": dy.sum(dim=(2, 3))}
    for nm, ref in want.items():
        err = float((got[nm].cpu() - ref).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref.abs().max())), (nm, err, float(ref.abs().max()))
    again = op(*args, mask=None if mask is None else mask.to(dev))
    assert all(torch.equal(got[k], again[k]) for k in want)  # fixed-order reductions: bitwise reproducible


@pytest.mark.parametrize("backend", BACKENDS)
def test_resnet_block_backward_composed_from_block_backwards(backend):
    """`ResnetBlock` (models/sr3_dwt.py:303-327; res_conv = Identity in the engine configuration): out = block2(block1(x) + tb) + x with
    tb = FeatureWiseAffine's per-sample time bias.  Its backward is two Block backwards chained: d(block1 output) = dx of block2,
    d(tb) = its plane sums, d(x) = dx of block1 + d(out).  Checked against autograd through the same composition."""
    from ddif import runtime

    dev = _dev(backend)
    B, Cc, H, W = 2, 32, 16, 16
    g = torch.Generator().manual_seed(99)
    leaf = lambda t: t.requires_grad_()
    x = leaf(torch.randn(B, Cc, H, W, generator=g))
    tb = leaf(0.3 * torch.randn(B, Cc, generator=g))
    p = []
    for _ in range(2):
        p.append(dict(gamma=leaf(1.0 + 0.2 * torch.randn(Cc, generator=g)), beta=leaf(0.1 * torch.randn(Cc, generator=g)),
                      w=leaf(torch.randn(Cc, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5)), b=leaf(0.05 * torch.randn(Cc, generator=g))))
    mask2 = (torch.rand(B, Cc, H, W, generator=g) >= 0.2).float() / 0.8  # Dropout sits in block2 only (dropout=0 in block1, :318-319)
    h1 = _block_forward(x, p[0]["gamma"], p[0]["beta"], None, p[0]["w"], p[0]["b"]) + tb[:, :, None, None]
    out = _block_forward(h1, p[1]["gamma"], p[1]["beta"], mask2, p[1]["w"], p[1]["b"]) + x
    dout = torch.randn(B, Cc, H, W, generator=g)
    out.backward(dout)
    op = ddif_testops.BlockBackward(B, Cc, Cc, H, W, dev)
    d = lambda t: t.detach().to(dev)
    g2 = op(d(h1), d(p[1]["gamma"]), d(p[1]["beta"]), d(p[1]["w"]), d(dout), mask=mask2.to(dev))
    g1 = op(d(x), d(p[0]["gamma"]), d(p[0]["beta"]), d(p[0]["w"]), g2["dx"])
    dx = g1["dx"] + d(dout)
    checks = [("dx", dx, x.grad), ("dtb", g1["dy_plane_sums"], tb.grad)]
    for k, gk in ((0, g1), (1, g2)):
        checks += [(f"block{k + 1}.dgamma", gk["dgamma"], p[k]["gamma"].grad), (f"block{k + 1}.dbeta", gk["dbeta"], p[k]["beta"].grad),
                   (f"block{k + 1}.dw", gk["dw"], p[k]["w"].grad), (f"block{k + 1}.db", gk["db"], p[k]["b"].grad)]
    for nm, got, ref in checks:
        err = float((got.cpu() - ref).abs().max())
        assert err <= 5e-5 * max(1.0, float(ref.abs().max())), (nm, err, float(ref.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("ks,pro,resample,shape", [(1, "none", "plain", (2, 32, 64, 16, 16)), (3, "none", "plain", (2, 32, 64, 8, 8)),
                                                   (1, "gn", "plain", (2, 128, 384, 8, 8)), (1, "gn_silu", "plain", (2, 128, 64, 16, 16)),
                                                   (3, "silu", "plain", (2, 64, 32, 16, 16)), (3, "none", "down2", (2, 32, 32, 16, 16)),
                                                   (3, "none", "down2", (1, 16, 24, 9, 13)), (3, "none", "up2", (2, 64, 64, 8, 8))],
                         ids=["x_conv-1x1", "ffn0-3x3", "attn-norm-qkv", "cond-body-1x1", "ffn2-silu-3x3", "downsample", "downsample-odd", "upsample"])
def test_conv_op_backward_variants_match_autograd(backend, ks, pro, resample, shape):
    """The other conv-with-prologue shapes of the network through the same op: plain 1x1 (CondInjection.x_conv, ffn.3, attention
    output convs; models/sr3_dwt.py:385-396,528-533), plain 3x3 (ffn.0), GroupNorm -> 1x1 (SelfAttention.norm -> qkv, :338-339,349),
    GroupNorm -> Swish -> 1x1 (CondInjection.body tail, :380-384), SiLU -> 3x3 (ffn.2), Downsample (:276-282, also at odd sizes),
    Upsample (:266-273)."""
    from ddif import runtime

    dev = _dev(backend)
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + ks)
    x = (torch.randn(B, Cin, H, W, generator=g) * 1.2 - 0.2).requires_grad_()
    gamma = (1.0 + 0.2 * torch.randn(Cin, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(Cin, generator=g)).requires_grad_()
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (ks * Cin ** 0.5)).requires_grad_()
    b = torch.zeros(Cout, requires_grad=True)
    a = x
    if pro in ("gn", "gn_silu"):
        a = torch.nn.functional.group_norm(a, 1, gamma, beta, eps=1e-5)
    if pro in ("gn_silu", "silu"):
        a = torch.nn.functional.silu(a)
    if resample == "up2":
        a = torch.nn.functional.interpolate(a, scale_factor=2, mode="nearest")
    y = torch.nn.functional.conv2d(a, w, b, padding=ks // 2, stride=2 if resample == "down2" else 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    op = ddif_testops.BlockBackward(B, Cin, Cout, H, W, dev, ks=ks, pro=pro, resample=resample)
    d = lambda t: t.detach().to(dev)
    got = op(d(x), d(gamma), d(beta), d(w), d(dy))
    want = {"dx": x.grad, "dw": w.grad, "db": b.grad, "dy_plane_sums": dy.sum(dim=(2, 3))}
    if pro in ("gn", "gn_silu"):
        want.update(dgamma=gamma.grad, dbeta=beta.grad)
    for nm, ref in want.items():
        err = float((got[nm].cpu() - ref).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref.abs().max())), (nm, err, float(ref.abs().max()))


def test_ssim_oracle_properties():
    """The SSIM restatement (oracle.ssim_skimage; skimage itself is absent -> parity unpinned) against what the published
    definition fixes: identical images give exactly 1; two constant images give the luminance term alone (both variances are 0);
    symmetry; and the crop of 3 pixels per side means the border never enters."""
    gen = torch.Generator().manual_seed(11)
    a = torch.rand(3, 20, 22, generator=gen)
    b = (a + 0.1 * torch.randn(3, 20, 22, generator=gen)).clamp(0, 1)
    assert abs(O.ssim_skimage(a, a) - 1.0) <= 1e-12
    assert abs(O.ssim_skimage(a, b) - O.ssim_skimage(b, a)) <= 1e-12
    ca, cb = torch.full((2, 9, 9), 0.25), torch.full((2, 9, 9), 0.75)
    c1 = (0.01 * 2.0) ** 2
    want = (2 * 0.25 * 0.75 + c1) / (0.25 ** 2 + 0.75 ** 2 + c1)
    assert abs(O.ssim_skimage(ca, cb) - want) <= 1e-9
    b2 = b.clone()
    b2[:, :3], b2[:, -3:], b2[:, :, :3], b2[:, :, -3:] = 0.0, 0.0, 0.0, 0.0  # (only windows centred in the interior count, and they reach 3 px out)
    inner_a, inner_b = a[:, 3:-3, 3:-3], b[:, 3:-3, 3:-3]
    assert O.ssim_skimage(a, b2) != O.ssim_skimage(a, b)  # the outer ring IS inside the 7x7 windows of the first interior pixels
    assert 0.0 < O.ssim_skimage(inner_a, inner_b) < 1.0


@pytest.mark.parametrize("backend", BACKENDS)
def test_ssim_kernel_matches_oracle(backend):
    from ddif import runtime

    dev = _dev(backend)
    gen = torch.Generator().manual_seed(12)
    gt = torch.rand(3, 8, 33, 35, generator=gen)
    pred = (gt + 0.05 * torch.randn(3, 8, 33, 35, generator=gen)).clamp(0, 1)
    pred[2] = gt[2]
    got = runtime.ssim(gt.to(dev), pred.to(dev)).cpu()
    for b in range(3):
        assert abs(float(got[b]) - O.ssim_skimage(gt[b], pred[b])) <= 2e-6, b
    assert float(got[2]) == 1.0
    got1 = runtime.ssim(gt.to(dev), pred.to(dev), data_range=1.0).cpu()
    assert abs(float(got1[0]) - O.ssim_skimage(gt[0], pred[0], data_range=1.0)) <= 2e-6
    with pytest.raises(runtime.DdifError):
        runtime.ssim(gt[:, :, :6].contiguous().to(dev), pred[:, :, :6].contiguous().to(dev))  # smaller than the 7x7 window


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_adamw_state_is_checkpointable_and_signals_updates(backend):
    """The optimizer's moments are torch tensors (ddif_optim_create_ex borrows them): state_dict() / load_state_dict() continue a run bit for
    bit; and step() bumps the parameters' version counters -- the signature UNetSR3 keys its packed weights on (the fused step
    writes through raw pointers)."""
    from ddif import runtime

    dev = _dev(backend)
    gen = torch.Generator().manual_seed(13)
    shapes = [(5, 3), (7,), (2, 3, 3, 3)]

    def fresh():
        g2 = torch.Generator().manual_seed(14)
        ps = [torch.randn(s, generator=g2).to(dev) for s in shapes]
        gs = [torch.zeros_like(p) for p in ps]
        es = [p.clone() for p in ps]
        return ps, gs, es, runtime.FusedAdamW(ps, gs, es, lr=1e-2, weight_decay=1e-2)

    grads = [[torch.randn(s, generator=gen) for s in shapes] for _ in range(4)]
    ps, gs, es, opt = fresh()
    v0 = [p._version for p in ps]
    for k in range(4):
        for g, src in zip(gs, grads[k]):
            g.copy_(src.to(dev))
        opt.step(max_grad_norm=1.0, ema_mode=2, ema_decay=0.9)
        if k == 1:
            saved = opt.state_dict()
            saved_p, saved_e = [p.clone() for p in ps], [e.clone() for e in es]
    assert all(p._version > v for p, v in zip(ps, v0))
    assert saved["step"] == 2 and float(saved["exp_avg"][0].abs().max()) > 0
    ps2, gs2, es2, opt2 = fresh()
    for p, e, sp, se in zip(ps2, es2, saved_p, saved_e):
        p.copy_(sp)
        e.copy_(se)
    opt2.load_state_dict(saved)
    for k in (2, 3):
        for g, src in zip(gs2, grads[k]):
            g.copy_(src.to(dev))
        opt2.step(max_grad_norm=1.0, ema_mode=2, ema_decay=0.9)
    for a, b in zip(ps + es, ps2 + es2):
        assert torch.equal(a, b)
