"""Host-side engine pieces (CPU) and the test_fn mirror (GPU): cond assembly, Haar identities, tiling."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from ddif import diffusion_engine as E
from ddif.synth import synth_tiles


def test_haar_identities():
    """PyWavelets is not installed in the build image, so the level-1 db1 analysis is pinned by its defining
    identities and the documented example pywt.dwt([1,2,3,4], 'db1') = ([2.1213, 4.9497], [-0.7071, -0.7071])."""
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 8, 12, generator=g)
    ll, (ch, cv, cd) = E.haar_dwt2(x)
    assert ll.shape == (2, 3, 4, 6)
    # orthonormal: energy is preserved
    e = (ll ** 2 + ch ** 2 + cv ** 2 + cd ** 2).sum()
    assert abs(float(e) - float((x ** 2).sum())) < 1e-3
    # constant image: only LL (= 2 * value); row ramp: detail along the rows axis only
    c = torch.full((1, 1, 4, 4), 0.25)
    ll, (ch, cv, cd) = E.haar_dwt2(c)
    assert torch.allclose(ll, torch.full_like(ll, 0.5)) and float(ch.abs().max() + cv.abs().max() + cd.abs().max()) == 0
    ramp = torch.arange(4.0).view(1, 1, 4, 1).expand(1, 1, 4, 4)
    _, (ch, cv, cd) = E.haar_dwt2(ramp)
    assert torch.allclose(ch, torch.full_like(ch, -1.0)) and float(cv.abs().max() + cd.abs().max()) == 0
    # 1-D example from the PyWavelets documentation, as the separable row transform of a 1x4 signal repeated on 2 rows
    s = torch.tensor([[1.0, 2.0, 3.0, 4.0]]).repeat(2, 1).view(1, 1, 2, 4)
    ll, (ch, cv, cd) = E.haar_dwt2(s)
    assert torch.allclose(ll.flatten() / 2 ** 0.5, torch.tensor([2.1213, 4.9497]), atol=1e-4)
    assert torch.allclose(cv.flatten() / 2 ** 0.5, torch.tensor([-0.7071, -0.7071]), atol=1e-4)


def test_assemble_cond_matches_the_fixture_generator():
    t = synth_tiles(2, 8, 1, 16, 16, seed=3)
    cond = E.assemble_cond(t["lms"], t["pan"], None, "wv3")
    assert cond.shape == (2, 20, 16, 16)
    assert torch.equal(cond, t["cond"])
    hs = synth_tiles(1, 31, 3, 16, 16, seed=4, order="hisr")
    assert torch.equal(E.assemble_cond(hs["lms"], hs["pan"], None, "cave"), hs["cond"])


@pytest.mark.parametrize("world", [1, 2])
def test_training_batches_resume_from_the_middle_of_an_epoch(world):
    """a checkpoint rarely falls on an epoch boundary (`save_every=5000` against a data-dependent epoch length).  The data position is
    part of the training state: epoch in force, batch offset inside it and -- with one rank, where the permutation comes from the global torch
    generator like DataLoader's RandomSampler -- the generator state it was drawn from.  An 11-iteration run (7 samples, batch 2: epochs of 4
    batches, the last one short) resumed from EVERY iteration reproduces the rest of the uninterrupted run, for one rank and for rank 1 of 2."""
    data = {k: np.arange(7 * 2, dtype=np.float32).reshape(7, 2) + i for i, k in enumerate(("pan", "lms", "gt"))}
    rank = world - 1

    def run(n_it, resume=None):
        tr = E._Batches(data, 2, True, rank, world, 3)
        seen, states = [], []
        if resume is not None:
            torch.set_rng_state(resume["rng"])
            tr.load_state(resume["data"])
            it = resume["it"]
        else:
            torch.manual_seed(5)
            it = 0
        while it < n_it:
            for _pan, _lms, gt in tr:
                torch.rand(3)  # the iteration's own draws (timesteps, noise) advance the same generator
                seen.append(gt[:, 0].tolist())
                it += 1
                states.append({"rng": torch.get_rng_state(), "data": tr.state(), "it": it})
                if it >= n_it:
                    break
        return seen, states

    full, states = run(11)
    assert len(full) == 11
    for cut in range(1, 11):
        torch.manual_seed(999)  # whatever the process state is at resume time
        tail, _ = run(11, resume=states[cut - 1])
        assert tail == full[cut:], (world, cut)


def test_training_batches_refuse_fewer_samples_than_ranks():
    from ddif import DdifError

    data = {k: np.zeros((1, 2), np.float32) for k in ("pan", "lms", "gt")}
    with pytest.raises(DdifError, match="ranks"):
        E._Batches(data, 1, True, 0, 2, 0)


def test_engine_google_needs_h5py_or_arrays():
    """The reference reads h5 files (diffusion_engine.py:142-143); h5py is not in this image, so a path must fail loudly and name the
    in-memory alternative."""
    from ddif import DdifError

    with pytest.raises(DdifError, match="h5py"):
        E.engine_google("train_wv3.h5", "valid_wv3.h5", dataset_name="wv3", device="cpu")


@pytest.mark.gpu
def test_test_fn_on_in_memory_set():
    from ddif_testlib import use_gpu_library
    from oracle import ddif_oracle as O

    use_gpu_library()
    ds, N, H = "gf2", 3, 16
    t = gc.tiles_for(ds, N, H, H, seed=8)
    div = 1023.0
    data = dict(lms=(t["lms"] * div).numpy(), pan=(t["pan"] * div).numpy(), gt=(t["gt"] * div).numpy())
    out = E.test_fn(None, None, batch_size=2, n_steps=50, device="cuda:0", dataset_name=ds, division=div, data=data,
                    state_dict=gc.weights_for(ds), seed=5)
    assert out["sr"].shape == (N, 4, H, H) and len(out["psnr"]) == 2
    assert float(out["sr"].min()) >= 0 and float(out["sr"].max()) <= div
    # same job through the oracle for the first batch (ddim25 from T=50; eta = 0 -> only x_T is random)
    torch.manual_seed(5)
    xT = torch.randn(2, 4, H, H, device="cuda:0").cpu()
    it = iter([xT] + [torch.zeros(2, 4, H, H)] * 25)
    cond = t["cond"][:2]
    with torch.no_grad():
        ref, _ = O.ddim_sample(gc.weights_for(ds), gc.cfg_for(ds), cond, O.schedule_tables(O.cosine_betas(50)), "ddim25",
                               noise_fn=lambda s: next(it))
    ref_sr = ((ref + cond[:, :4]).clip(0, 1) * div).numpy()
    assert float(np.abs(out["sr"][:2] - ref_sr).max()) <= 1e-4 * div


@pytest.mark.gpu
def test_test_fn_tiled_scene_matches_per_tile_runs_and_oracle(tmp_path):
    """a 64x64 GF2 scene cut into four 32x32 tiles inside test_fn (cond assembled on the whole scene by the
    fused kernel, tiles sampled in two batches of 2, stitched, written as .mat).  Checked against (a) the oracle on every
    tile with the same initial noise and (b) the stitching order."""
    from scipy.io import loadmat

    from ddif.sharding import cut_tiles, stitch_tiles
    from ddif_testlib import use_gpu_library
    from oracle import ddif_oracle as O

    use_gpu_library()
    ds, Hs, tile, div = "gf2", 64, 32, 1023.0
    t = gc.tiles_for(ds, 1, Hs, Hs, seed=21)
    raw = dict(lms=(t["lms"] * div).round().numpy(), pan=(t["pan"] * div).round().numpy(), gt=(t["gt"] * div).round().numpy())
    g = torch.Generator().manual_seed(22)
    xT = torch.randn(4, 4, tile, tile, generator=g)
    mat = os.path.join(tmp_path, "out.mat")
    out = E.test_fn(None, None, batch_size=2, n_steps=50, device="cuda:0", dataset_name=ds, division=div, data=raw,
                    state_dict=gc.weights_for(ds), tile=tile, x_T=xT, save_path=mat)
    assert out["sr"].shape == (1, 4, Hs, Hs) and out["metrics"].shape == (1, 4)
    # oracle: cond of the whole scene (raw data / division, Haar, bilinear), cut the same way, DDIM-25 from T=50 per tile
    cond_scene = O.assemble_cond(torch.from_numpy(raw["lms"]), torch.from_numpy(raw["pan"]), div)[0]
    ctiles = cut_tiles(cond_scene, tile)
    tabs = O.schedule_tables(O.cosine_betas(50))
    refs = []
    for k in range(4):
        it = iter([xT[k:k + 1]] + [torch.zeros(1, 4, tile, tile)] * 25)
        with torch.no_grad():
            r, _ = O.ddim_sample(gc.weights_for(ds), gc.cfg_for(ds), ctiles[k:k + 1], tabs, "ddim25", noise_fn=lambda s: next(it))
        refs.append((r + ctiles[k:k + 1, :4]).clip(0, 1))
    ref_scene = stitch_tiles(torch.cat(refs), 2, 2).numpy() * div
    assert float(np.abs(out["sr"][0] - ref_scene).max()) <= 1e-4 * div
    # metrics column 2 is the reference-sign PSNR of the stitched scene (utils/_metric_legacy.py:341-346)
    gt_n = torch.from_numpy(raw["gt"]) / div
    m = O.analysis_accu(gt_n[0].permute(1, 2, 0), torch.from_numpy(ref_scene / div).float().permute(1, 2, 0), 4)
    assert abs(float(out["metrics"][0, 2]) - m["PSNR"]) <= 1e-3 and abs(float(out["metrics"][0, 0]) - m["SAM"]) <= 1e-2
    saved = loadmat(mat)
    assert saved["sr"].shape == (1, 4, Hs, Hs) and np.allclose(saved["sr"], out["sr"])
