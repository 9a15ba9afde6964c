"""Pins the CPU oracle (oracle/ddif_oracle.py) against vectors produced by the real reference
(tools/make_golden.py).  CPU only.  Tolerances: the oracle calls the same torch CPU ops as the reference, so
whole-forward agreement is ~1e-6 (summation order inside einsum/reshape paths may differ); schedule tables and
the manifest are exact."""
import json
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from ddif.layout import param_manifest
from oracle import ddif_oracle as O


def _load(name):
    return np.load(os.path.join(gc.GOLDEN_DIR, name + ".npz"))


def _chk(t):
    return np.array([float(t.double().sum()), float(t.double().abs().max())])


@pytest.mark.parametrize("ds", list(gc.DATASETS))
def test_manifest_matches_reference(ds):
    with open(os.path.join(gc.GOLDEN_DIR, f"manifest_{ds}.json")) as f:
        ref = json.load(f)
    mine = param_manifest(gc.cfg_for(ds))
    assert [k for k, _ in mine] == [k for k, _ in ref["keys"]]
    assert [list(s) for _, s in mine] == [s for _, s in ref["keys"]]
    n = sum(int(np.prod(s)) for _, s in mine)
    assert n == ref["n_params"]
    assert {"wv3": 10397208, "gf2": 10250324, "cave": 11332511}[ds] == n


def test_layer_plans_agree():
    from ddif.layout import layer_plan

    for ds in gc.DATASETS:
        a, b = layer_plan(gc.cfg_for(ds)), O.layer_plan(gc.cfg_for(ds))
        assert a == b


@pytest.mark.parametrize("case", gc.FORWARD_CASES, ids=[c[0] for c in gc.FORWARD_CASES])
def test_forward_matches_reference(case):
    g = _load(case[0])
    x, t, cond, sc = gc.forward_inputs(case)
    np.testing.assert_allclose(_chk(x), g["x_chk"], rtol=0, atol=0)  # generators are bit-stable
    np.testing.assert_allclose(_chk(cond), g["cond_chk"], rtol=0, atol=0)
    with torch.no_grad():
        y = O.unet_forward(gc.weights_for(case[1]), gc.cfg_for(case[1]), x, t, cond, sc)
    err = float((y - torch.from_numpy(g["y"])).abs().max())
    assert err <= 5e-6, err


def test_schedule_tables_exact():
    g = _load("schedules")
    for T in gc.SCHEDULE_T:
        tabs = O.schedule_tables(O.cosine_betas(T))
        for k in O.TABLE_NAMES:
            assert np.array_equal(tabs[k].numpy(), g[f"T{T}.{k}"]), (T, k)
    for T in gc.DDIM_FROM:
        tabs = O.schedule_tables(O.cosine_betas(T))
        keep = O.ddim_stride_set(T, "ddim25")
        assert keep == list(g[f"ddim25_from_T{T}.keep"])
        nt = O.schedule_tables(O.respaced_betas(tabs["alphas_cumprod"], keep))
        for k in O.TABLE_NAMES:
            assert np.array_equal(nt[k].numpy(), g[f"ddim25_from_T{T}.{k}"]), (T, k)


def _ddpm(case, record=None, max_steps=None):
    cid, ds, B, H, W, T, seed = case
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    tabs = O.schedule_tables(O.cosine_betas(T))
    torch.manual_seed(seed)
    with torch.no_grad():
        return O.ddpm_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, tabs, record=record, max_steps=max_steps)


@pytest.mark.parametrize("case", [c for c in gc.DDPM_CASES if c[5] * c[3] * c[4] <= 100 * 32 * 32],
                         ids=lambda c: c[0])
def test_ddpm_short_matches_reference(case):
    g = _load(case[0])
    rec = {n: None for n in gc.DDPM_SNAPSHOTS.get(case[0], [])}
    out = _ddpm(case, record=rec)
    for n, v in rec.items():
        assert float((v - torch.from_numpy(g[f"after_{n}"])).abs().max()) <= 1e-5
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5


# (the full T = 1000 chain of the oracle against the reference's golden: ~40 s of CPU on 8 cores -- run by default since round 4)
@pytest.mark.parametrize("case", [c for c in gc.DDPM_CASES if c[0] == "ddpm_wv3_16_T1000"], ids=lambda c: c[0])
def test_ddpm_T1000_matches_reference(case):
    g = _load(case[0])
    out = _ddpm(case)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-4  # north-star atol


@pytest.mark.parametrize("case", gc.DDIM_CASES, ids=lambda c: c[0])
def test_ddim_matches_reference(case):
    cid, ds, B, H, W, T, sect, seed = case
    g = _load(cid)
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    tabs = O.schedule_tables(O.cosine_betas(T))
    torch.manual_seed(seed)
    with torch.no_grad():
        out, nt = O.ddim_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, tabs, sect)
    assert nt["betas"].numel() == int(g["num_timesteps_after"])
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 2e-5


@pytest.mark.parametrize("case", gc.DPM_CASES + gc.DPM_BIG_CASES, ids=lambda c: c[0])  # (+ round 6: GF2 at the benchmarked 64 x 64 tile size, ~10 s)
def test_dpm_solver_matches_reference(case):
    cid, ds, H, W, T, steps, order, seed = case
    g = _load(cid)
    C = gc.DATASETS[ds][0]
    cond = gc.tiles_for(ds, 1, H, W, seed=seed)["cond"]
    tabs = O.schedule_tables(O.cosine_betas(T))
    xT = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        out = O.dpmpp_multistep_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, tabs["betas"], xT, steps, order)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 2e-5


@pytest.mark.parametrize("case", gc.DPM_SKIP_CASES, ids=lambda c: c[0])
def test_dpm_solver_logsnr_matches_reference(case):
    cid, ds, H, W, T, steps, order, seed, skip = case
    g = _load(cid)
    C = gc.DATASETS[ds][0]
    cond = gc.tiles_for(ds, 1, H, W, seed=seed)["cond"]
    tabs = O.schedule_tables(O.cosine_betas(T))
    xT = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        out = O.dpmpp_multistep_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, tabs["betas"], xT, steps, order, skip_type=skip)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 2e-5


@pytest.mark.parametrize("case", gc.DDPM_TRUNC_CASES, ids=lambda c: c[0])
def test_ddpm_cave_T2000_truncated_matches_reference(case):
    """BASELINE config 4's schedule (T = 2000) on a CAVE-shaped tile: first / last 20 steps of the reference's own loop."""
    cid, ds, B, H, W, T, which, n, seed = case
    g = _load(cid)
    cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
    tabs = O.schedule_tables(O.cosine_betas(T))
    order = list(reversed(range(T)))
    steps = order[:n] if which == "first" else order[-n:]
    torch.manual_seed(seed)
    with torch.no_grad():
        out = O.ddpm_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, tabs, timesteps=steps)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5


@pytest.mark.parametrize("case", gc.FORWARD_BIG_CASES, ids=lambda c: c[0])
def test_forward_cave_128_matches_reference(case):
    g = _load(case[0])
    x, t, cond, sc = gc.forward_inputs(case)
    with torch.no_grad():
        y = O.unet_forward(gc.weights_for(case[1]), gc.cfg_for(case[1]), x, t, cond, sc)
    assert float((y - torch.from_numpy(g["y"])).abs().max()) <= 5e-6


def test_dropin_schedule_buffers_match_reference_tables():
    """The drop-in GaussianDiffusion's OWN schedule buffers (the ones its samplers hand to libddif), bit for bit against the
    tables captured from the reference's set_new_noise_schedule / space_new_betas (diffusion_ddpm_pan.py:199-276,583-592)."""
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule

    class _Stub(torch.nn.Module):
        self_condition, pred_var = True, False

    g = _load("schedules")
    for T in gc.SCHEDULE_T:
        d = GaussianDiffusion(_Stub(), image_size=64, channels=8, pred_mode="x_start", loss_type="l1", device="cpu", clamp_range=(0, 1))
        d.set_new_noise_schedule(betas=make_beta_schedule("cosine", T, cosine_s=8e-3), device="cpu")
        for k in O.TABLE_NAMES:
            assert np.array_equal(getattr(d, k).numpy(), g[f"T{T}.{k}"]), (T, k)
    for T in gc.DDIM_FROM:
        d = GaussianDiffusion(_Stub(), image_size=64, channels=8, pred_mode="x_start", loss_type="l1", device="cpu", clamp_range=(0, 1))
        d.set_new_noise_schedule(betas=make_beta_schedule("cosine", T, cosine_s=8e-3), device="cpu")
        use = d.space_timesteps(d.num_timesteps, "ddim25")
        assert sorted(use) == list(g[f"ddim25_from_T{T}.keep"])
        d.space_new_betas(use)
        assert d.num_timesteps == 25
        for k in O.TABLE_NAMES:
            assert np.array_equal(getattr(d, k).numpy(), g[f"ddim25_from_T{T}.{k}"]), (T, k)


@pytest.mark.parametrize("case", gc.LOSS_CASES, ids=lambda c: c[0])
def test_p_losses_matches_reference(case):
    cid, ds, B, H, W, T, tvals, sc_branch, seed = case
    g = _load(cid)
    C = gc.DATASETS[ds][0]
    tiles = gc.tiles_for(ds, B, H, W, seed=seed)
    res = tiles["gt"] - tiles["lms"]
    tabs = O.schedule_tables(O.cosine_betas(T))
    noise = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        loss, recon = O.p_losses_eval(gc.weights_for(ds), gc.cfg_for(ds), tabs, res, tiles["cond"],
                                      torch.tensor(tvals), noise, sc_branch)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert float((recon - torch.from_numpy(g["recon"])).abs().max()) <= 1e-5


def test_psnr_sign_quirk():
    g = _load("psnr")
    gen = torch.Generator().manual_seed(5)
    a = torch.rand(8, 33, 35, generator=gen)
    b = (a + 0.05 * torch.randn(8, 33, 35, generator=gen)).clamp(0, 1)
    assert abs(O.psnr_reference_sign(a, b) - float(g["ref_psnr"])) < 1e-4
    assert O.psnr(a, b) > 0 > O.psnr_reference_sign(a, b)


def test_oracle_train_mode_forward_and_autograd_match_the_reference_gradients():
    """Golden G7 (tools/make_golden.py traingrad: the REAL reference under .train(), its own Dropout / DropPath masks, F.l1_loss, loss.backward()):
    the oracle with those masks made explicit must reproduce the output, the loss and -- through torch autograd -- the gradient norm of all 702
    parameters.  This pins the oracle's train-mode restatement; the product's hand-written backward is checked against the same file
    (tests/test_train_graph.py)."""
    import os

    cid, ds, B, H, W, tvals, seed = gc.TRAIN_GRAD_CASES[0]
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
    paths = [p for p in torch.from_numpy(g["paths"])]
    sd = {k: v.clone().requires_grad_(v.dtype == torch.float32) for k, v in gc.weights_for(ds).items()}
    y = O.unet_forward(sd, gc.cfg_for(ds), x, t, cond, sc, drop_masks=masks, path_scales=paths)
    assert float((y.detach() - torch.from_numpy(g["y"])).abs().max()) <= 2e-6
    loss = (y - target).abs().mean()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-6
    loss.backward()
    for n, ref in zip([str(n) for n in g["names"]], g["grad_norms"]):
        got = float(sd[n].grad.double().norm()) if sd[n].grad is not None else 0.0
        assert abs(got - float(ref)) <= 1e-4 * max(float(ref), 1e-4), (n, got, float(ref))
