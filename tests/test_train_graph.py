"""One training forward + backward of the whole denoiser through ddif.train.TrainGraph against the REAL reference (SURVEY.md 8(a) a15,
golden G7: tests/golden/traingrad_wv3_16.npz = tools/make_golden.py traingrad: `UNetSR3` under .train() with the Dropout / DropPath
masks it drew, F.l1_loss against a fixed target, loss.backward(); diffusion_engine.py:230-233).  Checked: the train-mode output, the
norm of the gradient of EVERY parameter (702 tensors) and a handful of full gradients.  Emulator on the CPU, real library on MI355X."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
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
    from ddif.train import TrainGraph

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
    dy = runtime.l1_loss_backward(y, target.to(dev))
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
