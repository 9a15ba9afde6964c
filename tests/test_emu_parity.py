"""CPU-side checks of the kernel LOGIC: the same csrc/ sources, compiled for the host against the test-only
emulator (tools/hipemu: fibers for threads, rendezvous for shuffles / MFMA), driven through the same C ABI and the
same Python drop-in classes, compared with the oracle.  Small shapes only (the emulator is ~10^4x slower than the
GPU).  These tests never stand in for the -m gpu parity tests; they catch indexing / barrier / host-sequencing bugs
without a GPU."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from ddif_testlib import make_diffusion, make_net, use_emulator
from oracle import ddif_oracle as O


@pytest.fixture(scope="module", autouse=True)
def _lib():
    lib = use_emulator()
    assert lib.emulated
    return lib


_nets = {}


def net_for(ds):
    if ds not in _nets:
        _nets[ds] = make_net(ds, "cpu")
    return _nets[ds]


@pytest.mark.parametrize("cid", ["fwd_wv3_16_b"])
def test_emulated_forward_matches_golden(cid):
    import os

    case = [c for c in gc.FORWARD_CASES if c[0] == cid][0]
    g = np.load(os.path.join(gc.GOLDEN_DIR, cid + ".npz"))
    x, t, cond, sc = gc.forward_inputs(case)
    y = net_for(case[1])(x, t, cond, sc)
    assert float((y - torch.from_numpy(g["y"])).abs().max()) <= 2e-5


def _tiny(ds, B, H, W, seed):
    C = gc.DATASETS[ds][0]
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    g = torch.Generator().manual_seed(seed)
    return C, cond, g


def test_emulated_ddpm_steps_match_oracle():
    ds, B, H, W, T, steps = "wv3", 1, 8, 8, 20, 3
    C, cond, g = _tiny(ds, B, H, W, 3)
    xT = torch.randn(B, C, H, W, generator=g)
    noise = torch.randn(steps, B, C, H, W, generator=g)
    d = make_diffusion(net_for(ds), C, T, H, "cpu")
    plan = d._plan(cond)
    c1, c2 = d.posterior_mean_coef1, d.posterior_mean_coef2
    cz = (0.5 * d.posterior_log_variance_clipped).exp()
    order = list(reversed(range(T)))[:steps]
    out = plan.sample_ddpm([float(i) for i in order], [float(c1[i]) for i in order], [float(c2[i]) for i in order],
                           [float(cz[i]) for i in order], xT, noise, 0, 0, (0.0, 1.0), "cpu")
    it = iter([xT] + list(noise))
    with torch.no_grad():
        ref = O.ddpm_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, O.schedule_tables(O.cosine_betas(T)),
                            noise_fn=lambda s: next(it), max_steps=steps)
    assert float((out - ref).abs().max()) <= 2e-5


def test_emulated_ddim_matches_oracle():
    ds, B, H, W, T = "gf2", 1, 8, 8, 12
    C, cond, g = _tiny(ds, B, H, W, 4)
    xT = torch.randn(B, C, H, W, generator=g)
    d = make_diffusion(net_for(ds), C, T, H, "cpu")
    out = d(cond, mode="ddim_sample", section_counts="ddim3", x_T=xT)
    assert d.num_timesteps == 3
    it = iter([xT] + [torch.zeros(B, C, H, W)] * 3)
    with torch.no_grad():
        ref, _ = O.ddim_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, O.schedule_tables(O.cosine_betas(T)), "ddim3",
                               noise_fn=lambda s: next(it))
    assert float((out - ref).abs().max()) <= 2e-5


def test_emulated_device_rng_statistics():
    """Philox + Box-Muller normals: mean ~ 0, std ~ 1, different draws differ, same key repeats."""
    ds, B, H, W, T = "gf2", 2, 8, 8, 4
    C, cond, _ = _tiny(ds, B, H, W, 5)
    d = make_diffusion(net_for(ds), C, T, H, "cpu")
    plan = d._plan(cond)
    zero = [0.0]
    # one "step" with coef_x0 = coef_xt = 0 and coef_z = 1 returns the noise draw itself
    z1 = plan.sample_ddpm([0.0], zero, zero, [1.0], torch.zeros(B, C, H, W), None, 123, 0, None, "cpu")
    z2 = plan.sample_ddpm([0.0], zero, zero, [1.0], torch.zeros(B, C, H, W), None, 123, 0, None, "cpu")
    z3 = plan.sample_ddpm([0.0], zero, zero, [1.0], torch.zeros(B, C, H, W), None, 124, 0, None, "cpu")
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    assert abs(float(z1.mean())) < 0.2 and 0.8 < float(z1.std()) < 1.2
    hi = plan.sample_ddpm([0.0], zero, zero, [1.0], torch.zeros(B, C, H, W), None, 123, 1, None, "cpu")
    assert torch.equal(hi[0], z1[1])  # tile0 offset = global tile index


def test_unsupported_configurations_fail_loudly():
    from ddif import DdifError
    from ddif.models.sr3_dwt import UNetSR3

    with pytest.raises(DdifError):
        UNetSR3(fourier_features=True)
    net = UNetSR3(in_channel=8, out_channel=8, norm_groups=32, channel_mults=(1, 2), image_size=16)  # reference defaults
    with pytest.raises(DdifError, match="norm_groups"):  # (.train() is torch's default; the train-mode plan refuses the same way)
        net(torch.zeros(1, 8, 16, 16), torch.zeros(1), torch.zeros(1, 20, 16, 16))
    net.eval()
    with pytest.raises(DdifError, match="norm_groups"):
        net(torch.zeros(1, 8, 16, 16), torch.zeros(1), torch.zeros(1, 20, 16, 16))


def test_emulated_dpm_solver_matches_oracle():
    from ddif.solver.dpm_solver import DPM_Solver, ImageSpaceClamp, NoiseScheduleVP, model_wrapper

    ds, H, W, T, steps, order = "gf2", 8, 8, 100, 4, 3
    C, cond, g = _tiny(ds, 1, H, W, 6)
    xT = torch.randn(1, C, H, W, generator=g)
    net = net_for(ds)
    d = make_diffusion(net, C, T, H, "cpu")
    ns = NoiseScheduleVP("discrete", betas=d.betas)
    fn = model_wrapper(net, ns, model_type="x_start", guidance_type="classifier-free", guidance_scale=1.0, condition=cond)
    slv = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=ImageSpaceClamp(cond[:, :C]))
    assert slv._fused_target() is not None
    out = slv.sample(xT, steps=steps, order=order)
    with torch.no_grad():
        ref = O.dpmpp_multistep_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, O.schedule_tables(O.cosine_betas(T))["betas"],
                                       xT, steps, order)
    assert float((out - ref).abs().max()) <= 2e-5


N_LAUNCH_64 = 132  # launches of one network evaluation of a 64 x 64 tile (engine configuration): 146 - 4 (EPI_XF) - 8 (linattn8_fused) - 2 (192-channel block of the 16 x 16 level on linattn_fused)


@pytest.mark.parametrize("ds", ["wv3", "gf2"])
def test_emulated_forward_64x64_runs_the_fused_high_resolution_kernels(ds, monkeypatch):
    """A whole 64 x 64 tile: the f16x2 3x3 convs on their 16 x 16 tiling and the fused linear-attention block (csrc/kernels_lafuse.h) with
    64-row columns (a column spans a wave pair: statistics exchange through LDS), 32-row columns (one wave per column), 64- and 96-channel
    inputs -- against the oracle, and against the three-launch form (`DDIF_LAFUSE=0`) it replaces."""
    B, H = 1, 64
    C = gc.DATASETS[ds][0]
    g = torch.Generator().manual_seed(64)
    x = torch.randn(B, C, H, H, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    cond = gc.tiles_for(ds, B, H, H, seed=65)["cond"]
    with torch.no_grad():
        ref = O.unet_forward(gc.weights_for(ds), gc.cfg_for(ds), x, t, cond, None)
    net = net_for(ds)
    net._net and net._net.plans.clear()
    y = net(x, t, cond).clone()
    assert float((y - ref).abs().max()) <= 2e-5
    # 168 launches per step with the three-launch attention half; the 11 decoder blocks at 64 x 64 / 32 x 32 / 16 x 16 with <= 128 channels take one each (146);
    # round 6: the x_conv + FiLM of the three 64 x 64 encoder blocks and of the first 32 x 32 one ride in their producers' epilogues (EPI_XF): 142
    assert net.plan_for(B, H, H, torch.device("cpu")).num_launches()["step"] == N_LAUNCH_64


def test_emulated_ddpm_32x32_runs_the_sampler_update_in_the_final_conv_epilogue():
    """At >= 32 x 32 the final conv has a split-operand tiling that carries the DDPM / DDIM update in its epilogue (kernels_conv.h EPI_SAMP:
    clamp(x0 + lms) - lms, posterior mean, + sigma z, the next step's counter) -- no separate update / counter launches.  Three steps (an odd
    count: the double-buffered step counter ends on the other parity) with uploaded reference-order noise against the oracle."""
    ds, B, H, T, steps = "wv3", 2, 32, 20, 3
    C, cond, g = _tiny(ds, B, H, H, 3)
    xT = torch.randn(B, C, H, H, generator=g)
    noise = torch.randn(steps, B, C, H, H, generator=g)
    d = make_diffusion(net_for(ds), C, T, H, "cpu")
    plan = d._plan(cond)
    c1, c2 = d.posterior_mean_coef1, d.posterior_mean_coef2
    cz = (0.5 * d.posterior_log_variance_clipped).exp()
    order = list(reversed(range(T)))[:steps]
    out = plan.sample_ddpm([float(i) for i in order], [float(c1[i]) for i in order], [float(c2[i]) for i in order],
                           [float(cz[i]) for i in order], xT, noise, 0, 0, (0.0, 1.0), "cpu")
    it = iter([xT] + [noise[k] for k in range(steps)])
    with torch.no_grad():
        ref = O.ddpm_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, O.schedule_tables(O.cosine_betas(T)), noise_fn=lambda s: next(it), timesteps=order)
    assert float((out - ref).abs().max()) <= 2e-5


def test_f16x2_range_guards_fall_back_to_bf16x3():
    """The two-plane fp16 split pre-scales its operands by fixed powers of two (csrc/ddif_dev.h): a conv takes it only when its weights are inside the
    scaled half range (|w| < 64, checked at commit) and, behind a GroupNorm, when sqrt(N) max|gamma| + max|beta| < 4094 (checked per conv at plan
    build); otherwise the conv stays on bf16x3.  Both guards are driven here: the forward must stay at fp32-class accuracy against the oracle with
    an absurd GroupNorm gain and with a 100x conv weight (outputs of O(100) magnitude: relative bar)."""
    from ddif.models.sr3_dwt import UNetSR3
    from ddif_testlib import CTOR_KEYS

    ds, B, H = "wv3", 1, 16
    C, cond, g = _tiny(ds, B, H, H, 21)
    x = torch.randn(B, C, H, H, generator=g)
    t = torch.tensor([123])
    for key, factor in (("downs.1.res_block.block1.block.0.weight", 500.0), ("downs.1.res_block.block2.block.3.weight", 100.0)):
        sd = {k: v.clone() for k, v in gc.weights_for(ds).items()}
        sd[key] = sd[key] * factor
        cfg = gc.cfg_for(ds)
        net = UNetSR3(**{k: cfg[k] for k in CTOR_KEYS})
        net.load_state_dict(sd)
        net = net.eval()
        y = net(x, t, cond)
        with torch.no_grad():
            ref = O.unet_forward(sd, cfg, x, t, cond, None)
        assert torch.isfinite(y).all()
        assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), key


def test_train_entry_points_refuse_empty_gradient_packs_and_plans_keep_their_net_alive():
    """(1) the dgrad weight packs of a train-mode plan start zeroed and are filled by `ddif_net_refresh`; a backward pass before that used to
    return silently-zero input gradients -- now the train entry points fail with DDIF_ERR_STATE.  (2) a plan holds raw pointers into its net:
    `ddif_net_destroy` before `ddif_plan_destroy` defers the free to the last plan."""
    import ctypes as C

    from ddif import DdifError

    ds, B, H = "wv3", 1, 8
    Cc, cond, g = _tiny(ds, B, H, H, 4)
    net = make_net(ds, "cpu").train()
    try:
        plan = net.plan_for(B, H, H, torch.device("cpu"), train=True)
        plan.set_cond(cond, force=True)
        grads = [(n, torch.zeros_like(p)) for n, p in net.named_parameters()]
        plan.train_bind(grads)
        x = torch.randn(B, Cc, H, H, generator=g)
        with pytest.raises(DdifError, match="ddif_net_refresh"):
            plan.train_forward_backward(x, torch.tensor([5]), None, torch.zeros_like(x))
        net._net.refresh_from_device(net.named_parameters())
        loss, _ = plan.train_forward_backward(x, torch.tensor([5]), None, torch.zeros_like(x))
        assert np.isfinite(float(loss))
    finally:
        net.eval()
    # (2) raw C ABI: destroy the net first, then use and destroy its plan
    net2 = make_net(ds, "cpu")
    p2 = net2.plan_for(B, H, H, torch.device("cpu"))
    lib = net2._net.lib
    nh, ph = net2._net.h, p2.h
    lib.dll.ddif_net_destroy(nh)          # deferred: the plan is alive
    a, b = C.c_int(), C.c_int()
    assert lib.dll.ddif_plan_num_launches(ph, C.byref(a), C.byref(b)) == 0 and a.value > 0
    lib.dll.ddif_plan_destroy(ph)         # frees the plan, then the orphaned net
    net2._net.h = None                    # the Python handles must not free them again
    p2.h = None
    net2._net.plans.clear()


def test_emulated_grid_cap_hook_gives_identical_results():
    """ddif_debug_set_grid_cap (the hook the -m gpu multi-item tests rely on): one workgroup walking every work item
    of every conv launch, across the sample boundaries of a B=3 batch, reproduces the uncapped result bit for bit."""
    from ddif import runtime

    ds, B, H = "gf2", 3, 8
    C, cond, g = _tiny(ds, B, H, H, 9)
    x = torch.randn(B, C, H, H, generator=g)
    t = torch.tensor([3, 500, 999])
    net = net_for(ds)
    net._net and net._net.plans.clear()
    want = net(x, t, cond).clone()
    try:
        runtime.set_debug_grid_cap(1)
        net._net.plans.clear()
        got = net(x, t, cond)
    finally:
        runtime.set_debug_grid_cap(0)
        net._net.plans.clear()
    assert torch.equal(got, want)
    with torch.no_grad():
        ref = O.unet_forward(gc.weights_for(ds), gc.cfg_for(ds), x, t, cond, None)
    assert float((got - ref).abs().max()) <= 2e-5


def test_emulated_train_mode_forward_matches_reference_masks():
    """UNetSR3 under .train(): Dropout(0.2) in every ResnetBlock's block2 and DropPath(0.2) on every decoder FFN, with the masks the
    REFERENCE drew (captured by tools/make_golden.py through forward hooks) uploaded into the train-mode plan."""
    import os

    from ddif import DdifError

    cid, ds, B, H, W, tvals, seed = gc.TRAIN_FWD_CASES[0]
    g = np.load(os.path.join(gc.GOLDEN_DIR, cid + ".npz"))
    C = gc.DATASETS[ds][0]
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=gen)
    sc = torch.randn(B, C, H, W, generator=gen)
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    t = torch.tensor(tvals, dtype=torch.long)
    masks = []
    for k in range(int(g["n_drop"])):
        shp = tuple(int(v) for v in g[f"drop_{k}_shape"])
        bits = np.unpackbits(g[f"drop_{k}"])[: int(np.prod(shp))].reshape(shp)
        masks.append(torch.from_numpy(bits.astype(np.float32)) / (1.0 - float(g["p_drop"])))
    net = make_net(ds, "cpu")
    try:
        net.train()
        net.set_train_masks(masks, torch.from_numpy(g["paths"]))
        y = net(x, t, cond, sc)
        assert float((y - torch.from_numpy(g["y"])).abs().max()) <= 2e-5
        # fresh masks from the library's generator: a different, finite result; identity masks reproduce the eval network
        net.set_train_masks(None, None)
        y2 = net(x, t, cond, sc)
        assert torch.isfinite(y2).all() and float((y2 - y).abs().max()) > 1e-3
        ones = [torch.ones_like(m) for m in masks]
        net.set_train_masks(ones, torch.ones_like(torch.from_numpy(g["paths"])))
        y3 = net(x, t, cond, sc)
        net.eval()
        assert float((y3 - net(x, t, cond, sc)).abs().max()) <= 2e-6
        with pytest.raises(DdifError):
            net.train()
            net.set_train_masks(masks[:3], torch.from_numpy(g["paths"]))
            net(x, t, cond, sc)
    finally:
        net.eval()
        net.set_train_masks(None, None)


def test_emulated_plan_arena_reuses_activation_buffers():
    """The step program's activations live in a liveness-based arena (csrc/ddif_plan.cpp Plan::build): far smaller than one
    buffer per tensor, and -- the real check -- every parity test in this file and in -m gpu runs on the aliased buffers."""
    net = net_for("wv3")
    plan = net.plan_for(2, 16, 16, torch.device("cpu"))
    m = plan.memory()
    assert 0 < m["arena_bytes"] < 0.45 * m["unaliased_bytes"]
    assert m["arena_bytes"] <= m["total_bytes"]
