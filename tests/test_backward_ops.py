"""Backward ops of the non-convolution pieces of the denoiser against torch autograd through the ORACLE's own
forward formulas (oracle/ddif_oracle.py, itself pinned to the reference) on the CPU in fp32.  The reference's backward is autograd
(`loss.backward()`, diffusion_engine.py:233), so autograd through the same forward expression is the checker.  Emulator on the
CPU (-m "not gpu"), the real library on MI355X (-m gpu)."""
import math

import pytest
import torch
import torch.nn.functional as F

from ddif_testlib import use_emulator, use_gpu_library

BACKENDS = [pytest.param("emu", id="emulated"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _dev(backend):
    if backend == "emu":
        use_emulator()
        return torch.device("cpu")
    use_gpu_library()
    return torch.device("cuda:0")


def _close(got, want, name, rel=3e-5):
    err = float((got.cpu() - want).abs().max())
    assert err <= rel * max(1.0, float(want.abs().max())), (name, err, float(want.abs().max()))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 24, 9, 13), (2, 128, 8, 8)], ids=["64@16", "24@9x13", "128@8"])
def test_depthwise3x3_backward(backend, shape):
    """FastAttnCondInjection.q[0] / kv[0] (models/sr3_dwt.py:507-520): conv3x3, groups = C, no bias."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Cc, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cc, 1, 3, 3, generator=g) / 3).requires_grad_()
    dy = torch.randn(B, Cc, H, W, generator=g)
    F.conv2d(x, w, None, padding=1, groups=Cc).backward(dy)
    dx, dw = runtime.dwconv3x3_backward(x.detach().to(dev), w.detach().to(dev), dy.to(dev))
    _close(dx, x.grad, "dx")
    _close(dw, w.grad, "dw")


@pytest.mark.parametrize("backend", BACKENDS)
def test_film_backward(backend):
    """CondInjection: xc * (1 + scale) + shift with scale, shift = y.chunk(2, dim=1) (models/sr3_dwt.py:393-396)."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, Cc, H, W = 2, 32, 16, 12
    g = torch.Generator().manual_seed(5)
    xc = torch.randn(B, Cc, H, W, generator=g, requires_grad=True)
    ss = torch.randn(B, 2 * Cc, H, W, generator=g, requires_grad=True)
    dout = torch.randn(B, Cc, H, W, generator=g)
    scale, shift = ss.chunk(2, dim=1)
    (xc * (1 + scale) + shift).backward(dout)
    dxc, dss = runtime.film_backward(xc.detach().to(dev), ss.detach().to(dev), dout.to(dev))
    _close(dxc, xc.grad, "dxc", 1e-6)
    _close(dss, ss.grad, "dss", 1e-6)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 128, 8, 8), (1, 64, 4, 8)], ids=["128@8x8", "64@4x8"])
def test_self_attention_core_backward(backend, shape):
    """SelfAttention between its two 1x1 convs (models/sr3_dwt.py:345-358; oracle.self_attention): per-head [q|k|v] interleave,
    scale 1/sqrt(C)."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, Cc, H, W = shape
    heads, d = 8, Cc // 8
    g = torch.Generator().manual_seed(sum(shape))
    qkv = torch.randn(B, 3 * Cc, H, W, generator=g, requires_grad=True)
    dout = torch.randn(B, Cc, H, W, generator=g)
    v4 = qkv.view(B, heads, 3 * d, H * W)
    q, k, v = v4[:, :, :d], v4[:, :, d:2 * d], v4[:, :, 2 * d:]
    a = torch.softmax(torch.einsum("bncp,bncq->bnpq", q, k) / math.sqrt(Cc), dim=-1)
    torch.einsum("bnpq,bncq->bncp", a, v).reshape(B, Cc, H, W).backward(dout)
    dqkv = runtime.selfattn_core_backward(qkv.detach().to(dev), dout.to(dev), heads=heads)
    _close(dqkv, qkv.grad, "dqkv")


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 96, 8, 12), (1, 256, 8, 8)], ids=["64@16", "96@8x12", "256@8"])
def test_linear_attention_core_backward(backend, shape):
    """FastAttnCondInjection between q / kv and attn_out (models/sr3_dwt.py:545-566; oracle.fast_attn_cond_injection):
    softmax over H of q, over W of k, q * 1/sqrt(d), ctx = k v^T, out = ctx^T q."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, qd, H, W = shape
    heads, d = 8, qd // 8
    g = torch.Generator().manual_seed(sum(shape))
    q_pre = (2.0 * torch.randn(B, qd, H, W, generator=g)).requires_grad_()
    kv_pre = (2.0 * torch.randn(B, 2 * qd, H, W, generator=g)).requires_grad_()
    dout = torch.randn(B, qd, H, W, generator=g)
    k, v = kv_pre.chunk(2, dim=1)
    q = q_pre.softmax(dim=-2).reshape(B, heads, d, H * W) * (1.0 / math.sqrt(d))
    k = k.softmax(dim=-1).reshape(B, heads, d, H * W)
    v = v.reshape(B, heads, d, H * W)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, qd, H, W).backward(dout)
    dq, dkv = runtime.linattn_core_backward(q_pre.detach().to(dev), kv_pre.detach().to(dev), dout.to(dev), heads=heads)
    _close(dq, q_pre.grad, "dq_pre")
    _close(dkv, kv_pre.grad, "dkv_pre")


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 64, 64, 64), (1, 96, 64, 64), (2, 128, 32, 32), (1, 192, 16, 16), (2, 256, 8, 8), (1, 64, 10, 12), (1, 128, 40, 24)],
                         ids=["64@64", "96@64", "128@32", "192@16", "256@8", "64@10x12", "128@40x24"])
def test_linear_attention_core_nhwc_forward_and_backward(backend, shape):
    """The same core in the form the native training step runs (csrc/kernels_linattn.h: row / column workgroups, contexts summed from
    per-workgroup partials) at the shapes of the batch-32 step's decoder blocks and at ragged ones (line groups with a short tail):
    output and both gradients against torch autograd in fp64."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, qd, H, W = shape
    heads, d = 8, qd // 8
    g = torch.Generator().manual_seed(sum(shape))
    q_pre = (2.0 * torch.randn(B, H, W, qd, generator=g, dtype=torch.float64)).requires_grad_()
    kv_pre = (2.0 * torch.randn(B, H, W, 2 * qd, generator=g, dtype=torch.float64)).requires_grad_()
    dout = torch.randn(B, H, W, qd, generator=g, dtype=torch.float64)
    k, v = kv_pre[..., :qd], kv_pre[..., qd:]
    q = q_pre.softmax(dim=1).reshape(B, H * W, heads, d) * (1.0 / math.sqrt(d))
    k = k.softmax(dim=2).reshape(B, H * W, heads, d)
    ctx = torch.einsum("bnha,bnhe->bhae", k, v.reshape(B, H * W, heads, d))
    out = torch.einsum("bhae,bnha->bnhe", ctx, q).reshape(B, H, W, qd)
    out.backward(dout)
    f32 = lambda t: t.detach().to(torch.float32).to(dev)
    got, ws = runtime.linattn_nhwc(f32(q_pre), f32(kv_pre))
    _close(got, out.detach().float(), "out", 1e-5)
    dq, dkv = runtime.linattn_nhwc_backward(f32(q_pre), f32(kv_pre), f32(dout), ws)
    _close(dq, q_pre.grad.float(), "dq_pre", 2e-5)
    _close(dkv, kv_pre.grad.float(), "dkv_pre", 2e-5)
    # deterministic: a second run gives the same bits
    got2, ws2 = runtime.linattn_nhwc(f32(q_pre), f32(kv_pre))
    dq2, dkv2 = runtime.linattn_nhwc_backward(f32(q_pre), f32(kv_pre), f32(dout), ws2)
    assert torch.equal(got, got2) and torch.equal(dq, dq2) and torch.equal(dkv, dkv2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_time_mlp_backward_chain(backend):
    """noise_level_mlp = Linear(32,128) -> Swish -> Linear(128,32) and one FeatureWiseAffine Linear(32, C) (models/sr3_dwt.py:59-64,
    241-258; oracle.time_embedding), chained from the library's linear / swish backward ops."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B = 4
    g = torch.Generator().manual_seed(3)
    leaf = lambda *s: (torch.randn(*s, generator=g) / math.sqrt(s[-1])).requires_grad_()
    enc = torch.randn(B, 32, generator=g)
    w1, b1, w3, b3, wf, bf = leaf(128, 32), leaf(128), leaf(32, 128), leaf(32), leaf(64, 32), leaf(64)
    h1 = F.linear(enc, w1, b1)
    h2 = h1 * torch.sigmoid(h1)
    temb = F.linear(h2, w3, b3)
    tb = F.linear(temb, wf, bf)
    dtb = torch.randn(B, 64, generator=g)
    tb.backward(dtb)
    d = lambda t: t.detach().to(dev)
    dtemb, dwf, dbf = runtime.linear_backward(d(temb), d(wf), d(dtb))
    dh2, dw3, db3 = runtime.linear_backward(d(h2), d(w3), dtemb)
    dh1 = runtime.swish_backward(d(h1), dh2)
    _, dw1, db1 = runtime.linear_backward(d(enc), d(w1), dh1)
    for got, ref, nm in ((dwf, wf.grad, "dwf"), (dbf, bf.grad, "dbf"), (dw3, w3.grad, "dw3"), (db3, b3.grad, "db3"), (dw1, w1.grad, "dw1"), (db1, b1.grad, "db1")):
        _close(got, ref, nm, 1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_l1_loss_backward(backend):
    """loss = F.l1_loss(model_out, target) (diffusion/diffusion_ddpm_pan.py:742-749)."""
    import ddif_testops as runtime

    dev = _dev(backend)
    g = torch.Generator().manual_seed(8)
    pred = torch.randn(2, 8, 16, 16, generator=g, requires_grad=True)
    target = torch.randn(2, 8, 16, 16, generator=g)
    target[0, 0, 0, :4] = pred.detach()[0, 0, 0, :4]  # exact ties: sign(0) = 0
    F.l1_loss(pred, target).backward()
    dp = runtime.l1_loss_backward(pred.detach().to(dev), target.to(dev))
    assert torch.equal(dp.cpu(), pred.grad)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 24, 5, 7)], ids=["64@16", "24@5x7"])
def test_groupnorm_alone_backward(backend, shape):
    """FastAttnCondInjection.prenorm_x (models/sr3_dwt.py:540): its output feeds q[0] and attn_res, so the GroupNorm backward runs on
    the sum of both consumers' gradients."""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (1.5 * torch.randn(B, Cc, H, W, generator=g) + 0.4).requires_grad_()
    gamma = (1.0 + 0.2 * torch.randn(Cc, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(Cc, generator=g)).requires_grad_()
    dy = torch.randn(B, Cc, H, W, generator=g)
    F.group_norm(x, 1, gamma, beta, eps=1e-5).backward(dy)
    dx, dg, db = runtime.groupnorm_backward(x.detach().to(dev), gamma.detach().to(dev), dy.to(dev))
    _close(dx, x.grad, "dx")
    _close(dg, gamma.grad, "dgamma")
    _close(db, beta.grad, "dbeta")


@pytest.mark.parametrize("backend", BACKENDS)
def test_cond_conv_weight_gradient_through_zero_padded_channels(backend):
    """CondInjection.body[0] = conv3x3(cond_L: 9 channels -> 4c, no bias) (models/sr3_dwt.py:380): the conv backward needs 4 | Cin,
    so the 9 cond channels (and the weights) are zero-padded to 12; the padded channels' weight gradient is zero, the rest is dW.
    (dx is not needed: cond is data.)"""
    import ddif_testops as runtime

    dev = _dev(backend)
    B, Cin, Cout, H, W = 2, 9, 128, 16, 16
    g = torch.Generator().manual_seed(21)
    cond = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 9).requires_grad_()
    dy = torch.randn(B, Cout, H, W, generator=g)
    F.conv2d(cond, w, None, padding=1).backward(dy)
    pad = lambda t: F.pad(t, (0, 0, 0, 0, 0, 3))  # channel axis 9 -> 12
    op = runtime.BlockBackward(B, 12, Cout, H, W, dev, ks=3, pro="none")
    got = op(pad(cond).to(dev), None, None, pad(w.detach()).to(dev), dy.to(dev), need_dx=False)
    _close(got["dw"][:, :Cin], w.grad, "dw")
    assert float(got["dw"][:, Cin:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------- forward ops of the training graph
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [dict(cin=32, cout=64, ks=3, H=16, W=16), dict(cin=16, cout=32, ks=3, H=12, W=20), dict(cin=64, cout=64, ks=1, H=8, W=8),
                                  dict(cin=32, cout=32, ks=3, H=16, W=16, stride=2), dict(cin=16, cout=24, ks=3, H=9, W=13, stride=2),
                                  dict(cin=64, cout=64, ks=3, H=8, W=8, up2=True), dict(cin=9, cout=128, ks=3, H=16, W=16, bias=False),
                                  dict(cin=32, cout=8, ks=3, H=16, W=16), dict(cin=11, cout=22, ks=1, H=8, W=8)],
                         ids=["3x3", "stem-like", "1x1", "down", "down-odd", "up", "cond-9ch", "final-8", "kv1-11ch"])
def test_training_graph_conv_forward(backend, case):
    """Every conv shape of the network through ddif_testops.conv2d against F.conv2d (channel counts that are not multiples of 4
    are zero-padded inside)."""
    import ddif_testops as DF

    dev = _dev(backend)
    g = torch.Generator().manual_seed(case["cin"] + case["cout"])
    B, H, W, ks = 2, case["H"], case["W"], case["ks"]
    x = torch.randn(B, case["cin"], H, W, generator=g)
    w = torch.randn(case["cout"], case["cin"], ks, ks, generator=g) / (ks * case["cin"] ** 0.5)
    b = torch.randn(case["cout"], generator=g) if case.get("bias", True) else None
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if case.get("up2") else x
    want = F.conv2d(xin, w, b, stride=case.get("stride", 1), padding=ks // 2)
    got = DF.conv2d(x.to(dev), w.to(dev), None if b is None else b.to(dev), stride=case.get("stride", 1), up2=case.get("up2", False))
    _close(got, want, "y", 2e-6)


@pytest.mark.parametrize("backend", BACKENDS)
def test_training_graph_elementwise_and_attention_forward(backend):
    """group_norm (+SiLU, +mask), depthwise conv, FiLM, residual / DropPath add, Linear, Swish and the two attention cores against
    the oracle's formulas (oracle/ddif_oracle.py: resnet_block, cond_injection, fast_attn_cond_injection, self_attention, time_embedding)."""
    import ddif_testops as DF

    dev = _dev(backend)
    g = torch.Generator().manual_seed(11)
    B, Cc, H, W = 2, 64, 8, 12
    x = 1.3 * torch.randn(B, Cc, H, W, generator=g) + 0.2
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    mask = (torch.rand(B, Cc, H, W, generator=g) > 0.2).float() / 0.8
    d = lambda t: t.to(dev)
    _close(DF.group_norm(d(x), d(gamma), d(beta)), F.group_norm(x, 1, gamma, beta, eps=1e-5), "gn", 2e-6)
    _close(DF.group_norm(d(x), d(gamma), d(beta), silu=True, mask=d(mask)), F.silu(F.group_norm(x, 1, gamma, beta, eps=1e-5)) * mask, "gn_silu_mask", 2e-6)
    wd = torch.randn(Cc, 1, 3, 3, generator=g) / 3
    _close(DF.dwconv3x3(d(x), d(wd)), F.conv2d(x, wd, None, padding=1, groups=Cc), "dw", 2e-6)
    ss = torch.randn(B, 2 * Cc, H, W, generator=g)
    sc_, sh_ = ss.chunk(2, dim=1)
    _close(DF.film(d(x), d(ss)), x * (1 + sc_) + sh_, "film", 1e-6)
    f = torch.randn(B, Cc, H, W, generator=g)
    alpha = torch.tensor([0.0, 1.25])
    _close(DF.add(d(x), d(f)), x + f, "add", 1e-7)
    _close(DF.add(d(x), d(f), d(alpha)), x + alpha.view(B, 1, 1, 1) * f, "droppath", 1e-7)
    xe, wl, bl = torch.randn(4, 32, generator=g), torch.randn(128, 32, generator=g) / 6, torch.randn(128, generator=g)
    _close(DF.linear(d(xe), d(wl), d(bl)), F.linear(xe, wl, bl), "linear", 2e-6)
    _close(DF.swish(d(xe)), xe * torch.sigmoid(xe), "swish", 1e-6)
    qkv = torch.randn(2, 3 * 128, 8, 8, generator=g)
    v4 = qkv.view(2, 8, 48, 64)
    q, k, v = v4[:, :, :16], v4[:, :, 16:32], v4[:, :, 32:]
    a = torch.softmax(torch.einsum("bncp,bncq->bnpq", q, k) / math.sqrt(128), dim=-1)
    _close(DF.selfattn_core(d(qkv)), torch.einsum("bnpq,bncq->bncp", a, v).reshape(2, 128, 8, 8), "selfattn", 3e-6)
    qd = 96
    q_pre, kv_pre = 2 * torch.randn(B, qd, H, W, generator=g), 2 * torch.randn(B, 2 * qd, H, W, generator=g)
    kk, vv = kv_pre.chunk(2, dim=1)
    dd = qd // 8
    qs = q_pre.softmax(dim=-2).reshape(B, 8, dd, H * W) / math.sqrt(dd)
    ks_ = kk.softmax(dim=-1).reshape(B, 8, dd, H * W)
    ctx = torch.einsum("bhdn,bhen->bhde", ks_, vv.reshape(B, 8, dd, H * W))
    _close(DF.linattn_core(d(q_pre), d(kv_pre)), torch.einsum("bhde,bhdn->bhen", ctx, qs).reshape(B, qd, H, W), "linattn", 3e-6)
