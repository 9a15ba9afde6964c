"""One training forward + backward of the whole denoiser through the op-by-op tape (tests/train_tape.py TrainGraph) and the native reverse program against the REAL reference (SURVEY.md 8(a) a15,
golden G7: tests/golden/traingrad_wv3_16.npz = tools/make_golden.py traingrad: `UNetSR3` under .train() with the Dropout / DropPath
masks it drew, F.l1_loss against a fixed target, loss.backward(); diffusion_engine.py:230-233).  Checked: the train-mode output, the
norm of the gradient of EVERY parameter (702 tensors) and a handful of full gradients.  Emulator on the CPU, real library on MI355X."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
import ddif_testops
from ddif_testlib import use_emulator, use_gpu_library

BACKENDS = [pytest.param("emu", id="emulated"), pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]


def _dev(backend):
    if backend == "emu":
        use_emulator()
        return torch.device("cpu")
    use_gpu_library()
    return torch.device("cuda:0")


def _case_inputs(case):
    cid, ds, B, H, W, tvals, seed = case
    g = np.load(os.path.join(gc.GOLDEN_DIR, cid + ".npz"))
    C = gc.DATASETS[ds][0]
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=gen)
    sc = torch.randn(B, C, H, W, generator=gen)
    target = torch.rand(B, C, H, W, generator=gen)
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    t = torch.tensor(tvals, dtype=torch.long)
    masks = []
    for k in range(int(g["n_drop"])):
        shp = tuple(int(v) for v in g[f"drop_{k}_shape"])
        bits = np.unpackbits(g[f"drop_{k}"])[: int(np.prod(shp))].reshape(shp)
        masks.append(torch.from_numpy(bits.astype(np.float32)) / (1.0 - float(g["p_drop"])))
    return g, ds, x, sc, target, cond, t, masks, torch.from_numpy(g["paths"])


@pytest.mark.parametrize("backend", BACKENDS)
def test_training_step_gradients_match_the_reference(backend):
    from ddif import runtime
    from train_tape import TrainGraph

    dev = _dev(backend)
    g, ds, x, sc, target, cond, t, masks, paths = _case_inputs(gc.TRAIN_GRAD_CASES[0])
    cfg = gc.cfg_for(ds)
    P = {k: v.to(dev).contiguous() for k, v in gc.weights_for(ds).items()}
    graph = TrainGraph(cfg, dropout=0.2, drop_path=0.2)
    y = graph.forward(P, x.to(dev), t.to(dev), cond.to(dev), sc.to(dev), drop_masks=[m.to(dev) for m in masks], path_scales=[p.to(dev) for p in paths])
    ref_y = torch.from_numpy(g["y"])
    assert float((y.cpu() - ref_y).abs().max()) <= 2e-5  # same bar as the inference forward
    loss = float((y.cpu() - target).abs().mean())
    assert abs(loss - float(g["loss"])) <= 1e-6
    dy = ddif_testops.l1_loss_backward(y, target.to(dev))
    grads = graph.backward(dy)
    names = [str(n) for n in g["names"]]
    norms = g["grad_norms"]
    assert set(names) <= set(P.keys())
    worst = 0.0
    for n, ref in zip(names, norms):
        assert n in grads, f"no gradient for {n}"
        got = float(grads[n].double().norm())
        tol = 2e-4 * max(float(ref), 1e-4)
        worst = max(worst, abs(got - float(ref)) / max(float(ref), 1e-4))
        assert abs(got - float(ref)) <= tol, (n, got, float(ref))
    for k in g.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(g[k])
            got = grads[k[6:]].cpu()
            assert got.shape == ref.shape, k
            err = float((got - ref).abs().max())
            assert err <= 5e-5 * max(float(ref.abs().max()), 1e-5), (k, err, float(ref.abs().max()))
    print("worst relative grad-norm error over %d parameters: %.2e" % (len(names), worst))


@pytest.mark.parametrize("backend", BACKENDS)
def test_native_training_step_gradients_match_the_reference(backend):
    """The NATIVE reverse program (csrc/ddif_train.cpp, ddif_plan_train_forward_backward): train-mode forward over NHWC activations + L1 loss
    + backward in one C-ABI call, against the real reference's `loss.backward()` golden -- output, loss, the gradient norm of all 702
    parameters and the full gradients the golden carries.  Also: two runs give bit-identical gradients (fixed-order reductions)."""
    from ddif_testlib import make_net

    dev = _dev(backend)
    g, ds, x, sc, target, cond, t, masks, paths = _case_inputs(gc.TRAIN_GRAD_CASES[0])
    B, _, H, W = x.shape
    net = make_net(ds, dev).train()
    try:
        plan = net.plan_for(B, H, W, dev, train=True)
        net._net.refresh_from_device(net.named_parameters())  # fills the dgrad packs (and re-packs the forward weights on the device)
        plan.set_cond(cond.to(dev), force=True)
        plan.set_train_masks([m.to(dev) for m in masks], paths)
        grads = {n: torch.full_like(p, float("nan")) for n, p in net.named_parameters()}
        plan.train_bind(list(grads.items()))
        loss, y = plan.train_forward_backward(x.to(dev), t, sc.to(dev), target.to(dev))
        assert float((y.cpu() - torch.from_numpy(g["y"])).abs().max()) <= 2e-5
        assert abs(float(loss) - float(g["loss"])) <= 1e-6
        names = [str(n) for n in g["names"]]
        assert set(names) == set(grads.keys())
        worst = 0.0
        for n, ref in zip(names, g["grad_norms"]):
            got = float(grads[n].double().norm())
            assert got == got, f"gradient of {n} was not written"
            worst = max(worst, abs(got - float(ref)) / max(float(ref), 1e-4))
            assert abs(got - float(ref)) <= 2e-4 * max(float(ref), 1e-4), (n, got, float(ref))
        for k in g.files:
            if k.startswith("grad::"):
                ref = torch.from_numpy(g[k])
                got = grads[k[6:]].cpu()
                assert got.shape == ref.shape, k
                err = float((got - ref).abs().max())
                assert err <= 5e-5 * max(float(ref.abs().max()), 1e-5), (k, err, float(ref.abs().max()))
        first = {n: v.clone() for n, v in grads.items()}
        plan.train_forward_backward(x.to(dev), t, sc.to(dev), target.to(dev))
        for n in first:
            assert torch.equal(first[n], grads[n]), n
        print("native step: worst relative grad-norm error over %d parameters: %.2e" % (len(names), worst))
    finally:
        net.eval()


GPU_ONLY = [pytest.param("gpu", id="mi355x", marks=pytest.mark.gpu)]  # a training step takes ~40 s on the host emulator: the CPU suite keeps the parity test above


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_loss_backward_through_the_drop_in_fills_parameter_grads_and_the_optimizer_steps(backend):
    """The reference's training lines unchanged (diffusion_engine.py:230-241) on the drop-in classes: `loss, recon = diffusion(x, cond=cond);
    loss.backward()` must leave a finite `.grad` on all 702 parameters, then the fused clip + AdamW + EMA step must move the weights.
    t, the self-conditioning branch and the q_sample tables are pinned so that the pass is the golden's forward (recon is compared with
    it); the gradients themselves are compared with the reference in the test above."""
    import random

    from ddif import runtime
    from ddif_testlib import make_diffusion, make_net

    dev = _dev(backend)
    g, ds, x, sc, target, cond, t, masks, paths = _case_inputs(gc.TRAIN_GRAD_CASES[0])
    net = make_net(ds, dev)
    d = make_diffusion(net, 8, 500, 16, dev)
    d.loss_type = "l1"
    net.train()
    net.set_train_masks([m.to(dev) for m in masks], paths.to(dev))
    try:
        # p_losses draws t and decides the self-conditioning branch itself; pin both so that the pass equals the golden's:
        # x_noisy = a * x0 + s * noise with a = 1, s = 0 is reproduced by handing the golden's x as x0 and patching the two tables
        torch.manual_seed(0)
        a_keep, s_keep = d.sqrt_alphas_cumprod.clone(), d.sqrt_one_minus_alphas_cumprod.clone()
        d.sqrt_alphas_cumprod.fill_(1.0)
        d.sqrt_one_minus_alphas_cumprod.fill_(0.0)
        randint, rnd = torch.randint, random.random
        torch.randint = lambda *a, **k: t.to(dev)
        random.random = lambda: 1.0  # no self-conditioning pass: the golden's self_cond is a given tensor, not a model output
        try:
            from ddif.runtime import PlanHandle

            step = PlanHandle.train_step

            def spy(self, x0, noise, a, s, time, self_cond, want_pred=True):  # the golden ran with an explicit self_cond tensor
                return step(self, x0, noise, a, s, time, sc.to(dev), want_pred)

            PlanHandle.train_step = spy
            try:
                loss, recon = d(x.to(dev), cond=cond.to(dev))
            finally:
                PlanHandle.train_step = step
        finally:
            torch.randint, random.random = randint, rnd
            d.sqrt_alphas_cumprod.copy_(a_keep)
            d.sqrt_one_minus_alphas_cumprod.copy_(s_keep)
        assert float((recon.cpu() - torch.from_numpy(g["y"])).abs().max()) <= 2e-5
        loss.backward()
        n_with_grad = sum(1 for _, p in net.named_parameters() if p.grad is not None)
        assert n_with_grad == 702
        assert all(torch.isfinite(p.grad).all() for p in net.parameters())
        assert abs(float(loss.detach()) - float((recon.cpu() - x).abs().mean())) <= 1e-6
        before = {n: p.detach().clone() for n, p in list(net.named_parameters())[:8]}
        params = [p for p in net.parameters()]
        grads = [p.grad for p in params]
        ema = [p.detach().clone() for p in params]
        opt = runtime.FusedAdamW(params, grads, ema, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
        gn = opt.step(max_grad_norm=0.003, ema_mode=1, ema_decay=0.995, return_norm=True)
        assert gn > 0 and np.isfinite(gn)
        assert any(float((p.detach() - before[n]).abs().max()) > 0 for n, p in list(net.named_parameters())[:8])
    finally:
        net.eval()
        net.set_train_masks(None, None)


@pytest.mark.parametrize("backend", BACKENDS)
def test_library_masks_are_split_invariant_reproducible_and_keep_the_right_fraction(backend):
    """Dropout / DropPath masks drawn by the library (one launch for every site, Philox keyed by (seed, site, global tile, NCHW element)):
    the masks of tiles 1..2 drawn in a batch of three equal those drawn by a plan of two tiles starting at tile 1 (what a DDP rank holding
    the second shard draws), the same seed gives the same masks, another seed other ones, values are 0 or 1/(1-p), and about 1-p are kept."""
    from ddif_testlib import make_net

    dev = _dev(backend)
    net = make_net("wv3", dev)
    net.train()
    try:
        big = net.plan_for(3, 16, 16, dev, train=True)
        small = net.plan_for(2, 16, 16, dev, train=True)
        big.random_train_masks(1234, 0, 0.2, 0.2)
        small.random_train_masks(1234, 1, 0.2, 0.2)
        mb, pb = big.train_masks()
        ms, ps = small.train_masks()
        assert len(mb) == len(ms) > 20 and pb.shape[0] == ps.shape[0] > 0
        kept = total = 0
        for a, b in zip(mb, ms):
            assert torch.equal(a[1:], b)
            vals = torch.unique(a)
            assert all(abs(float(v)) < 1e-12 or abs(float(v) - 1.25) < 1e-6 for v in vals)
            kept += int((a > 0).sum())
            total += a.numel()
        assert torch.equal(pb[:, 1:], ps)
        assert abs(kept / total - 0.8) < 0.01
        assert not any(torch.equal(mb[0], m) for m in mb[1:] if m.shape == mb[0].shape)  # sites draw different masks
        big.random_train_masks(1234, 0, 0.2, 0.2)
        mb2, pb2 = big.train_masks()
        assert all(torch.equal(a, b) for a, b in zip(mb, mb2)) and torch.equal(pb, pb2)
        big.random_train_masks(1235, 0, 0.2, 0.2)
        mb3, _ = big.train_masks()
        assert not torch.equal(mb[0], mb3[0])
        # a pinned set of masks goes back in unchanged
        big.set_train_masks(mb, pb)
        mb4, pb4 = big.train_masks()
        assert all(torch.equal(a, b) for a, b in zip(mb, mb4)) and torch.equal(pb, pb4)
    finally:
        net.eval()


def _raw_set(n, C, H, seed):
    """raw-count arrays like the reference's h5 training files: gt / lms at H x H, pan at H x H (values in [0, 2047])"""
    from ddif.synth import synth_tiles

    t = synth_tiles(n, C, 1, H, H, seed=seed)
    return {"gt": (t["gt"] * 2047.0).numpy(), "lms": (t["lms"] * 2047.0).numpy(), "pan": (t["pan"] * 2047.0).numpy()}


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_engine_google_trains_and_validates_on_an_in_memory_set(backend):
    """The reference's training entry point (diffusion_engine.py:52-348) end to end on a tiny in-memory WV3-shaped set: two / three iterations
    (cond assembly, p_losses, loss.backward() through the library, clip + AdamW + EMA), one DDIM-25 validation with the EMA weights.
    Checks what a training loop must deliver: finite losses, every parameter moved, EMA = a copy of the weights before `ema_start_iter`,
    finite validation metrics; and determinism: the same seed gives the same loss sequence."""
    import ddif.diffusion_engine as E2

    dev = _dev(backend)
    train, valid = _raw_set(4, 8, 16, 1), _raw_set(2, 8, 16, 2)

    iters = 3 if backend == "gpu" else 2  # (a training step takes ~40 s on the host emulator)

    def run():
        torch.manual_seed(5)
        import random

        random.seed(5)
        return E2.engine_google(train, valid, dataset_name="wv3", image_n_channel=8, image_size=16, n_steps=50, max_iterations=iters, device=str(dev),
                                batch_size=2, lr_d=1e-3, valid_every=iters, log=lambda *_: None)

    out = run()
    assert out["iterations"] == iters and len(out["loss"]) == iters and all(np.isfinite(out["loss"]))
    net = out["model"]
    moved = sum(1 for (n, p), e in zip(net.named_parameters(), out["ema"]) if torch.equal(p.detach(), e))
    assert moved == len(out["ema"])  # ema_mode 1 (before start_iter): EMA tracks the weights exactly
    assert len(out["validation"]) == 1 and all(np.isfinite(v) for v in out["validation"][0][1].values())
    if backend == "gpu":
        again = run()
        assert again["loss"] == out["loss"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_device_side_weight_refresh_equals_a_host_commit(backend):
    """Training changes the parameters every iteration, on the device.  `ddif_net_refresh` re-packs them there (one launch); the result must
    be what `ddif_net_load` + `ddif_net_commit` (host repack + upload) produce from the same values: the train-mode forward of a refreshed
    network is BIT-equal to that of a freshly committed one, and an inference plan is refused until the weights are committed again."""
    from ddif import runtime
    from ddif_testlib import make_net

    dev = _dev(backend)
    _, ds, x, sc, target, cond, t, masks, paths = _case_inputs(gc.TRAIN_GRAD_CASES[0])
    B, _, H, W = x.shape
    net = make_net(ds, dev).train()
    net.set_train_masks([m.to(dev) for m in masks], paths)
    try:
        y0 = net(x.to(dev), t, cond.to(dev), sc.to(dev)).clone()
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for p in net.parameters():
                p.add_((0.05 * torch.randn(p.shape, generator=g)).to(dev))
        runtime._bump_versions(list(net.parameters()))
        y_ref = net(x.to(dev), t, cond.to(dev), sc.to(dev)).clone()     # train=True and committed -> device refresh
        assert net._net.device_refreshed and float((y_ref - y0).abs().max()) > 0
        fresh = make_net(ds, dev).train()
        fresh.load_state_dict(net.state_dict())
        fresh.set_train_masks([m.to(dev) for m in masks], paths)
        y_host = fresh(x.to(dev), t, cond.to(dev), sc.to(dev))          # first use: host commit
        assert not fresh._net.device_refreshed
        assert torch.equal(y_ref, y_host)
        # an inference plan built BEFORE the refresh is refused afterwards; the drop-in re-commits for eval mode by itself
        stale = net._net.plan(B, H, W)
        with pytest.raises(runtime.DdifError, match="refreshed on the device"):
            stale.set_cond(cond.to(dev), force=True)
        net.eval()
        fresh.eval()
        assert torch.equal(net(x.to(dev), t, cond.to(dev), sc.to(dev)), fresh(x.to(dev), t, cond.to(dev), sc.to(dev)))
    finally:
        net.set_train_masks(None, None)
        net.eval()


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_engine_google_trains_on_after_validation_checkpoints_and_resumes(backend, tmp_path):
    """ADVICE r2 + SURVEY 8f-4.  (1) Validation (DDIM-25 respaces ITS schedule in place) runs on a separate diffusion object holding the EMA
    weights, as the reference validates on `ema_model`: training continues afterwards on the untouched n_steps schedule.  (2) The no-grad
    self-conditioning pass sees the weights the fused optimizer just wrote (packed-weight cache invalidated).  (3) Checkpoints are the
    reference's two bare state_dicts (+ a full-state file): `test_fn(weight_path=<ema file>)` loads one.  (4) Resuming from the full-state
    file reproduces the uninterrupted run bit for bit (weights, EMA, losses)."""
    import random

    import ddif.diffusion_engine as E2

    dev = _dev(backend)
    train, valid = _raw_set(4, 8, 16, 1), _raw_set(2, 8, 16, 2)
    kw = dict(dataset_name="wv3", image_n_channel=8, image_size=16, n_steps=50, device=str(dev), batch_size=2, lr_d=1e-3, valid_every=2,
              ema_start_iter=1, log=lambda *_: None)

    def seeded():
        torch.manual_seed(5)
        random.seed(5)
        torch.cuda.manual_seed(5)

    seeded()
    a_dir = str(tmp_path / "a")
    A = E2.engine_google(train, valid, max_iterations=4, save_dir=a_dir, save_every=2, **kw)
    assert A["diffusion"].num_timesteps == 50 and A["diffusion"].betas.numel() == 50  # (1)
    assert [it for it, _ in A["validation"]] == [2, 4]
    assert all(np.isfinite(v) for _, rec in A["validation"] for v in rec.values()) and "SSIM" in A["validation"][0][1]
    assert all(np.isfinite(A["loss"]))
    # (2) the train-mode no-grad forward follows the optimizer: same inputs and masks, before and after one more fused step
    net, d = A["model"], A["diffusion"]
    net.train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 16, 16, generator=g).to(dev)
    cond = gc.tiles_for("wv3", 2, 16, 16, seed=3)["cond"].to(dev)
    t = torch.tensor([3, 30])

    def nograd_forward():
        torch.manual_seed(77)  # same masks both times
        with torch.no_grad():
            return net(x, t, cond, None).clone()

    y0 = nograd_forward()
    for p in net.parameters():
        p.grad.normal_(generator=None)
    A["optimizer"].step(max_grad_norm=0.0, ema_mode=0)
    y1 = nograd_forward()
    assert float((y1 - y0).abs().max()) > 0
    net.eval()
    # (3) reference checkpoint format
    for it in (2, 4):
        for stem in ("diffusion", "ema_diffusion", "train_state"):
            assert os.path.exists(os.path.join(a_dir, f"{stem}_wv3_iter_{it}.pth")), (stem, it)
    ema_sd = torch.load(os.path.join(a_dir, "ema_diffusion_wv3_iter_4.pth"), map_location="cpu")
    assert set(ema_sd) == set(n for n, _ in net.named_parameters())
    out = E2.test_fn(data=valid, weight_path=os.path.join(a_dir, "ema_diffusion_wv3_iter_4.pth"), n_steps=50, dataset_name="wv3", division=2047.0,
                     device=str(dev), batch_size=2, seed=1)
    assert out["sr"].shape == (2, 8, 16, 16) and np.isfinite(out["sr"]).all()
    # (4) resume from iteration 2 == the uninterrupted run
    st = torch.load(os.path.join(a_dir, "train_state_wv3_iter_2.pth"), map_location="cpu", weights_only=False)
    final = torch.load(os.path.join(a_dir, "train_state_wv3_iter_4.pth"), map_location="cpu", weights_only=False)
    torch.manual_seed(999)  # whatever the process state is now: the checkpoint carries the RNG
    B = E2.engine_google(train, valid, max_iterations=4, resume_state=st, log_every=3, **kw)  # deferred loss read-back: same values
    assert B["iterations"] == 4 and B["loss"] == A["loss"][2:]
    for (n, p), e in zip(B["model"].named_parameters(), B["ema"]):
        assert torch.equal(p.detach().cpu(), final["model"][n]), n
        assert torch.equal(e.cpu(), final["ema"][n]), n


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_native_step_equals_the_op_by_op_tape_at_the_benchmark_tile_size(backend, monkeypatch):
    """The reference golden pins the native step at 16 x 16 tiles.  At the benchmark's 64 x 64 (other band / line-group geometries of the
    weight-gradient, linear-attention and depthwise kernels, all four resolution levels at their real sizes) the native forward + reverse
    program is compared with round 2's op-by-op tape (tests/train_tape.py `tape_train_step`: the stateless NCHW ops of csrc/kernels_bwd_ops.h, each pinned
    against autograd in tests/test_backward_ops.py) on the same inputs, timesteps and pinned masks: loss, prediction and the gradient of every
    one of the 702 parameters."""
    from ddif_testlib import make_diffusion, make_net

    dev = _dev(backend)
    B, C, H = 2, 8, 64
    gen = torch.Generator().manual_seed(123)
    net = make_net("wv3", dev)
    d = make_diffusion(net, C, 500, H, dev)
    d.loss_type = "l1"
    x0 = torch.rand(B, C, H, H, generator=gen).to(dev)
    noise = torch.randn(B, C, H, H, generator=gen).to(dev)
    sc = torch.randn(B, C, H, H, generator=gen).to(dev)
    cond = gc.tiles_for("wv3", B, H, H, seed=9)["cond"].to(dev)
    t = torch.tensor([37, 411], device=dev)
    a, s = d._schedule_rows(t)
    net.train()
    try:
        plan = net.plan_for(B, H, H, dev, train=True)
        plan.random_train_masks(77, 0, 0.2, 0.2)
        masks, paths = plan.train_masks()
        net.set_train_masks([m.clone() for m in masks], paths.clone())

        def run(tape):
            from train_tape import tape_train_step

            for p in net.parameters():
                p.grad = None
            loss, pred = tape_train_step(d, x0, noise, a, s, t, cond, sc) if tape else d._train_step(x0, noise, a, s, t, cond, sc)
            loss.backward()
            return float(loss.detach()), pred.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}

        l_nat, p_nat, g_nat = run(False)
        l_tape, p_tape, g_tape = run(True)
        assert abs(l_nat - l_tape) <= 1e-6 * max(1.0, abs(l_tape))
        assert float((p_nat - p_tape).abs().max()) <= 5e-5
        # (some gradients are analytically zero -- a bias in front of a softmax over a whole line, q.1.bias / the k half of kv.1.bias: both paths
        # return rounding noise there -- so the bar is relative to the tensor's own norm plus a floor relative to the largest gradient of the step)
        gmax = max(float(g.norm()) for g in g_tape.values())
        worst = 0.0
        for n, gt in g_tape.items():
            gn = g_nat[n]
            assert gn.shape == gt.shape and torch.isfinite(gn).all(), n
            diff, den = float((gn - gt).norm()), float(gt.norm())
            assert diff <= 2e-4 * den + 2e-6 * gmax, (n, diff, den, gmax)
            if den > 1e-3 * gmax:
                worst = max(worst, diff / den)
        print("worst relative gradient difference native vs tape at 64x64 (tensors above 1e-3 of the largest norm):", worst)
    finally:
        net.eval()
        net.set_train_masks(None, None)


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_native_training_step_at_the_benchmarked_batch_matches_the_oracle_and_autograd(backend):
    """BASELINE configs[4] is timed at batch 32 of 64 x 64 tiles (`bench.py --config wv3_train_b32`); the reference golden pins the native step at
    B = 2, 16 x 16.  At B = 32 the weight-gradient K splits, the GroupNorm-backward chunking and the linear-attention reduce grids take other
    geometries, so the SAME shape is checked here against the oracle (pinned train-mode restatement, oracle/ddif_oracle.py unet_forward with explicit
    masks = `UNetSR3.forward` under .train(), models/sr3_dwt.py:169-219) + torch autograd on the CPU (= `loss.backward()`,
    diffusion_engine.py:230-233): prediction, loss and the gradient of every one of the 702 parameters.  The masks are the library's own Philox
    draws read back through ddif_plan_train_get_dropout / _droppath and handed to the oracle."""
    from ddif_testlib import make_net
    from oracle import ddif_oracle as O

    dev = _dev(backend)
    ds, B, C, H = "wv3", 32, 8, 64
    gen = torch.Generator().manual_seed(2024)
    x = torch.randn(B, C, H, H, generator=gen)
    sc = torch.randn(B, C, H, H, generator=gen)
    target = torch.rand(B, C, H, H, generator=gen)
    cond = gc.tiles_for(ds, B, H, H, seed=31)["cond"]
    t = torch.randint(0, 3000, (B,), generator=gen)
    net = make_net(ds, dev).train()
    try:
        plan = net.plan_for(B, H, H, dev, train=True)
        net._net.refresh_from_device(net.named_parameters())
        plan.set_cond(cond.to(dev), force=True)
        plan.random_train_masks(4242, 0, 0.2, 0.2)
        masks, paths = plan.train_masks()
        grads = {n: torch.full_like(p, float("nan")) for n, p in net.named_parameters()}
        plan.train_bind(list(grads.items()))
        loss, y = plan.train_forward_backward(x.to(dev), t, sc.to(dev), target.to(dev))
        torch.cuda.synchronize()
        # the CPU side: oracle forward under the same masks, autograd backward (a few seconds on the host cores)
        sd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in gc.weights_for(ds).items()}
        y_ref = O.unet_forward(sd, gc.cfg_for(ds), x, t, cond, sc, drop_masks=[m.cpu() for m in masks], path_scales=[p for p in paths.cpu()])
        loss_ref = (y_ref - target).abs().mean()
        loss_ref.backward()
        assert float((y.cpu() - y_ref.detach()).abs().max()) <= 5e-5
        assert abs(float(loss) - float(loss_ref.detach())) <= 1e-6
        ref = {n: (sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])) for n in grads}
        gmax = max(float(g.norm()) for g in ref.values())
        worst, worst_n = 0.0, ""
        for n, gr in ref.items():
            got = grads[n].cpu()
            assert got.shape == gr.shape and torch.isfinite(got).all(), n
            diff, den = float((got - gr).norm()), float(gr.norm())
            # (analytically-zero gradients -- biases in front of a whole-line softmax -- hold rounding noise on both sides: floor relative to the
            # step's largest gradient, as in the native-vs-tape test below)
            assert diff <= 2e-4 * den + 2e-6 * gmax, (n, diff, den, gmax)
            if den > 1e-3 * gmax and diff / den > worst:
                worst, worst_n = diff / den, n
        print("native step at B=32, 64x64 vs oracle + autograd: worst relative gradient error %.2e (%s), loss %.6f" % (worst, worst_n, float(loss)))
    finally:
        net.eval()


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_two_stream_reverse_pass_is_bit_identical_to_the_one_stream_one(backend, monkeypatch):
    """The reverse pass issues the weight-gradient launches on a side stream, ordered against the gradient chain with events (csrc/ddif_train.cpp).
    A plan built with `DDIF_TRAIN_STREAMS=0` runs the same launches on one stream: loss, prediction and all 702 gradients must be the same BITS,
    three iterations in a row (a missing dependency shows up as a difference or a NaN, not as a tolerance question)."""
    from ddif_testlib import make_net

    dev = _dev(backend)
    B, C, H = 4, 8, 64
    gen = torch.Generator().manual_seed(5)
    x0 = torch.rand(B, C, H, H, generator=gen).to(dev)
    noise = torch.randn(B, C, H, H, generator=gen).to(dev)
    sc = torch.randn(B, C, H, H, generator=gen).to(dev)
    cond = gc.tiles_for("wv3", B, H, H, seed=4)["cond"].to(dev)
    a, s, t = torch.full((B,), 0.8), torch.full((B,), 0.6), torch.tensor([3.0, 100.0, 250.0, 499.0])
    results = []
    for streams in ("1", "0"):
        monkeypatch.setenv("DDIF_TRAIN_STREAMS", streams)
        torch.manual_seed(11)
        net = make_net("wv3", dev)  # same weights both times (make_net seeds its initialisation)
        net.train()
        plan = net.plan_for(B, H, H, dev, train=True)
        named = [(n, torch.zeros_like(p)) for n, p in net.named_parameters()]
        plan.train_bind(named)
        net._net.refresh_from_device(net.named_parameters())
        plan.set_cond(cond, force=True)
        plan.random_train_masks(9, 0, 0.2, 0.2)
        out = []
        for _ in range(3):
            loss, pred = plan.train_step(x0, noise, a, s, t, sc)
            torch.cuda.synchronize()
            out.append((float(loss), pred.clone(), [g.clone() for _, g in named]))
        results.append(out)
        net.eval()
    for (l1, p1, g1), (l0, p0, g0) in zip(*results):
        assert l1 == l0 and torch.equal(p1, p0)
        for (n, _), u, v in zip(named, g1, g0):
            assert torch.isfinite(u).all() and torch.equal(u, v), n


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_a_sampler_on_a_train_mode_plan_joins_the_side_stream_before_it_captures(backend):
    """`set_cond` of a train-mode plan leaves its decoder-only half running on the plan's side stream; the step program waits for it in front
    of the first decoder block.  The samplers replay captured step pairs -- an event wait inside a capture would be illegal -- so they join first.
    A fresh train-mode plan (identity masks) must sample what the inference plan samples (other kernel fusions: tolerance, not bits)."""
    from ddif_testlib import make_net

    dev = _dev(backend)
    B, C, H, T = 2, 8, 32, 6
    net = make_net("wv3", dev)
    cond = gc.tiles_for("wv3", B, H, H, seed=2)["cond"].to(dev)
    gen = torch.Generator().manual_seed(1)
    x_T = torch.randn(B, C, H, H, generator=gen).to(dev)
    noise = torch.randn(T, B, C, H, H, generator=gen).to(dev)
    t_model = [float(i) for i in reversed(range(T))]
    c0, c1, cz = [0.6] * T, [0.4] * T, [0.05] * (T - 1) + [0.0]
    outs = []
    for train in (False, True):
        (net.train() if train else net.eval())
        plan = net.plan_for(B, H, H, dev, train=train)
        plan.set_cond(cond, force=True)
        outs.append(plan.sample_ddpm(t_model, c0, c1, cz, x_T, noise, 0, 0, (0.0, 1.0), dev))
        plan.set_cond(cond, force=True)  # twice in a row: the second call waits for the first one's side work
        outs.append(plan.sample_ddpm(t_model, c0, c1, cz, x_T, noise, 0, 0, (0.0, 1.0), dev))
    net.eval()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])
    assert torch.isfinite(outs[2]).all() and float((outs[2] - outs[0]).abs().max()) <= 1e-4


@pytest.mark.parametrize("backend", GPU_ONLY)
def test_reference_optimizer_lines_run_unchanged_on_the_drop_in(backend):
    """diffusion_engine.py:205-241 verbatim in spirit: torch.optim.AdamW over `denoise_fn.parameters()`, `opt.zero_grad()`,
    `diff_loss.backward()`, `clip_grad_norm_(…, 0.003)`, `opt.step()` -- two iterations; the loss must be finite, the clipped gradient norm
    must respect the bound and the weights must move.  (The fused optimizer of `engine_google` is optional: plain torch optimizers see
    ordinary `.grad` tensors.)"""
    from ddif_testlib import make_diffusion, make_net

    dev = _dev(backend)
    _, ds, x, sc, target, cond, t, masks, paths = _case_inputs(gc.TRAIN_GRAD_CASES[0])
    net = make_net(ds, dev)
    d = make_diffusion(net, 8, 500, 16, dev)
    d.loss_type = "l1"
    net.train()
    try:
        opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4)
        first = next(net.parameters()).detach().clone()
        torch.manual_seed(3)
        for _ in range(2):
            opt.zero_grad()
            loss, recon = d(x.to(dev) * 0.1, cond=cond.to(dev))
            loss.backward()
            total = torch.nn.utils.clip_grad_norm_(net.parameters(), 0.003)
            assert torch.isfinite(loss.detach()) and torch.isfinite(total)
            after = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters()))
            assert float(after) <= 0.003 * (1 + 1e-4)
            opt.step()
        assert float((next(net.parameters()).detach() - first).abs().max()) > 0
    finally:
        net.eval()
