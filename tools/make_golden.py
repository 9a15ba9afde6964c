#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the REAL reference (imported from /root/reference) on the seeded
cases of tests/golden_cases.py.  Runs only in the build container; the reference never travels.  Nothing but
numeric vectors (inputs' checksums, expected outputs) and the reference's state-dict key/shape manifest is
written into the repository.

    python tools/make_golden.py [--only fwd,sched,ddpm,ddim,dpm,loss,psnr] [--skip-long]
    python tools/make_golden.py --only ddpmbig,dpmbig,ddpmfull     (round 6: the config-exact cases; not part of the default set)
"""
import argparse
import json
import os
import random
import sys
import time
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402

REF = os.environ.get("DDIF_REFERENCE", "/root/reference")


def import_reference():
    class DropPath(nn.Module):  # stand-in for timm.models.layers.DropPath (timm is not installed)
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * (m.div_(keep) if self.scale_by_keep and keep > 0 else m)

    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["timm.models.layers"].DropPath = DropPath
    sys.path.insert(0, REF)
    import builtins

    _print = builtins.print
    builtins.print = lambda *a, **k: None  # the constructor prints "use attn: res 8"
    try:
        from models.sr3_dwt import UNetSR3
        from diffusion import diffusion_ddpm_pan as D
        from solver import dpm_solver as S
    finally:
        builtins.print = _print
    D.tqdm = lambda it, **kw: it
    return UNetSR3, D, S


def build_ref_net(UNetSR3, ds):
    cfg = gc.cfg_for(ds)
    import builtins

    _print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        net = UNetSR3(
            in_channel=cfg["in_channel"], out_channel=cfg["out_channel"], lms_channel=cfg["lms_channel"],
            pan_channel=cfg["pan_channel"], inner_channel=32, norm_groups=1, channel_mults=(1, 2, 2, 4),
            attn_res=(8,), dropout=0.2, image_size=64, self_condition=True)
    finally:
        builtins.print = _print
    net.load_state_dict(gc.weights_for(ds), strict=True)
    net.eval()
    return net


def chk(t):
    return float(t.double().sum()), float(t.double().abs().max())


def save(name, **arrs):
    path = os.path.join(gc.GOLDEN_DIR, name + ".npz")
    np.savez(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB)")


def make_diffusion(D, net, C, T, size):
    d = D.GaussianDiffusion(net, image_size=size, channels=C, pred_mode="x_start", loss_type="l1", device="cpu",
                            clamp_range=(0, 1))
    d.set_new_noise_schedule(betas=D.make_beta_schedule(schedule="cosine", n_timestep=T, cosine_s=8e-3))
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="manifest,fwd,fwdbig,trunc,dpmskip,trainfwd,traingrad,sched,ddpm,ddim,dpm,loss,psnr")
    ap.add_argument("--skip-long", action="store_true")
    args = ap.parse_args()
    only = set(args.only.split(","))
    os.makedirs(gc.GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(8)
    UNetSR3, D, S = import_reference()
    nets = {}

    def net_for(ds):
        if ds not in nets:
            nets[ds] = build_ref_net(UNetSR3, ds)
        return nets[ds]

    if "manifest" in only:
        for ds in gc.DATASETS:
            import builtins
            _print = builtins.print
            builtins.print = lambda *a, **k: None
            cfg = gc.cfg_for(ds)
            fresh = UNetSR3(in_channel=cfg["in_channel"], out_channel=cfg["out_channel"],
                            lms_channel=cfg["lms_channel"], pan_channel=cfg["pan_channel"], inner_channel=32,
                            norm_groups=1, channel_mults=(1, 2, 2, 4), attn_res=(8,), dropout=0.2, image_size=64,
                            self_condition=True)
            builtins.print = _print
            man = [[k, list(v.shape)] for k, v in fresh.state_dict().items()]
            with open(os.path.join(gc.GOLDEN_DIR, f"manifest_{ds}.json"), "w") as f:
                json.dump(dict(n_params=sum(p.numel() for p in fresh.parameters()), keys=man), f)
            print(f"  manifest_{ds}.json: {len(man)} tensors")

    if "fwd" in only:
        for case in gc.FORWARD_CASES:
            x, t, cond, sc = gc.forward_inputs(case)
            with torch.no_grad():
                y = net_for(case[1])(x, t, cond, sc)
            save(case[0], y=y, x_chk=chk(x), cond_chk=chk(cond))

    if "fwdbig" in only:
        for case in gc.FORWARD_BIG_CASES:
            x, t, cond, sc = gc.forward_inputs(case)
            with torch.no_grad():
                y = net_for(case[1])(x, t, cond, sc)
            save(case[0], y=y, x_chk=chk(x), cond_chk=chk(cond))

    if "trunc" in only:
        import builtins

        for cid, ds, B, H, W, T, which, n, seed in gc.DDPM_TRUNC_CASES:
            C = gc.DATASETS[ds][0]
            cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
            d = make_diffusion(D, net_for(ds), C, T, H)
            # the reference loop itself, iterating only the first / last n timesteps: p_sample_loop looks `reversed` up
            # in its module globals, so a module-level stand-in truncates the iteration without touching its code
            D.reversed = (lambda r: iter(list(builtins.reversed(r))[:n])) if which == "first" else (lambda r: iter(list(builtins.reversed(r))[-n:]))
            try:
                torch.manual_seed(seed)
                t0 = time.time()
                out = d(cond, mode="ddpm_sample")
            finally:
                del D.reversed
            print(f"  {cid}: {time.time() - t0:.1f}s")
            save(cid, out=out, cond_chk=chk(cond))

    if "dpmskip" in only:
        for cid, ds, H, W, T, steps, order, seed, skip in gc.DPM_SKIP_CASES:
            C = gc.DATASETS[ds][0]
            cond = gc.tiles_for(ds, 1, H, W, seed=seed)["cond"]
            d = make_diffusion(D, net_for(ds), C, T, H)
            ns = S.NoiseScheduleVP("discrete", betas=d.betas)
            lms = cond[:, :C]
            fn = S.model_wrapper(net_for(ds), ns, model_type="x_start", guidance_type="classifier-free",
                                 guidance_scale=1.0, condition=cond)
            slv = S.DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=lambda x0, t, lms=lms: (x0 + lms).clamp(0, 1.0) - lms)
            xT = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(seed))
            with torch.no_grad():
                out = slv.sample(xT, steps=steps, order=order, skip_type=skip, method="multistep")
            save(cid, out=out, cond_chk=chk(cond))

    if "trainfwd" in only:
        for cid, ds, B, H, W, tvals, seed in gc.TRAIN_FWD_CASES:
            C = gc.DATASETS[ds][0]
            net = net_for(ds)
            g = torch.Generator().manual_seed(seed)
            x = torch.randn(B, C, H, W, generator=g)
            sc = torch.randn(B, C, H, W, generator=g)
            cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
            t = torch.tensor(tvals, dtype=torch.long)
            drops, paths, hooks = [], [], []
            for m in net.modules():  # module order is not execution order: record in the hooks, which fire in execution order
                if isinstance(m, nn.Dropout):
                    hooks.append(m.register_forward_hook(lambda mod, inp, out: drops.append((out != 0) | (inp[0] == 0))))
                elif type(m).__name__ == "DropPath":
                    hooks.append(m.register_forward_hook(lambda mod, inp, out: paths.append((out.flatten(1).abs().sum(1) != 0).float() / (1 - mod.drop_prob))))
            net.train()
            torch.manual_seed(seed)
            try:
                with torch.no_grad():
                    y = net(x, t, cond, sc)
            finally:
                net.eval()
                for hk in hooks:
                    hk.remove()
            arrs = {f"drop_{k}": np.packbits(d.numpy().reshape(-1)) for k, d in enumerate(drops)}
            arrs.update({f"drop_{k}_shape": np.array(d.shape) for k, d in enumerate(drops)})
            save(cid, y=y, n_drop=len(drops), paths=torch.stack(paths), p_drop=0.2, **arrs)

    if "traingrad" in only:
        # G7: one training forward + backward of the REAL reference under .train() (same inputs / seed / masks as trainfwd), L1 loss against
        # a fixed target: the norm of every parameter's gradient, and a handful of full gradients
        for cid, ds, B, H, W, tvals, seed in gc.TRAIN_GRAD_CASES:
            C = gc.DATASETS[ds][0]
            net = net_for(ds)
            g = torch.Generator().manual_seed(seed)
            x = torch.randn(B, C, H, W, generator=g)
            sc = torch.randn(B, C, H, W, generator=g)
            target = torch.rand(B, C, H, W, generator=g)
            cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
            t = torch.tensor(tvals, dtype=torch.long)
            drops, paths, hooks = [], [], []
            for m in net.modules():
                if isinstance(m, nn.Dropout):
                    hooks.append(m.register_forward_hook(lambda mod, inp, out: drops.append(((out != 0) | (inp[0] == 0)).detach())))
                elif type(m).__name__ == "DropPath":
                    hooks.append(m.register_forward_hook(lambda mod, inp, out: paths.append(((out.detach().flatten(1).abs().sum(1) != 0).float() / (1 - mod.drop_prob)))))
            net.train()
            for prm in net.parameters():
                prm.requires_grad_(True)
                prm.grad = None
            torch.manual_seed(seed)
            try:
                y = net(x, t, cond, sc)
                loss = F.l1_loss(y, target)
                loss.backward()
            finally:
                net.eval()
                for hk in hooks:
                    hk.remove()
            names = [k for k, _ in net.named_parameters()]
            norms = np.array([float(prm.grad.norm()) if prm.grad is not None else -1.0 for _, prm in net.named_parameters()], dtype=np.float64)
            full = {}
            for k, prm in net.named_parameters():
                if any(k == f or k.startswith(f) for f in gc.TRAIN_GRAD_FULL):
                    full["grad::" + k] = prm.grad.detach().clone()
            for prm in net.parameters():
                prm.grad = None
            arrs = {f"drop_{k}": np.packbits(d.numpy().reshape(-1)) for k, d in enumerate(drops)}
            arrs.update({f"drop_{k}_shape": np.array(d.shape) for k, d in enumerate(drops)})
            save(cid, y=y.detach(), loss=float(loss), n_drop=len(drops), paths=torch.stack(paths), p_drop=0.2, names=np.array(names), grad_norms=norms, **arrs, **full)

    if "sched" in only:
        out = {}
        for T in gc.SCHEDULE_T:
            d = make_diffusion(D, net_for("wv3"), 8, T, 64)
            for k, v in d.named_buffers():
                if "." not in k:
                    out[f"T{T}.{k}"] = v.clone()
        for T in gc.DDIM_FROM:
            d = make_diffusion(D, net_for("wv3"), 8, T, 64)
            use = d.space_timesteps(d.num_timesteps, "ddim25")
            out[f"ddim25_from_T{T}.keep"] = np.array(sorted(use))
            d.space_new_betas(use)
            for k, v in d.named_buffers():
                if "." not in k:
                    out[f"ddim25_from_T{T}.{k}"] = v.clone()
        save("schedules", **out)

    ddpm_cases = list(gc.DDPM_CASES) if "ddpm" in only else []
    if "ddpmbig" in only:  # round 6: two more configs[1] tiles (about a minute each on 8 cores)
        ddpm_cases += gc.DDPM_BIG_CASES
    if "ddpmfull" in only:  # round 6: the full CAVE 128 x 128 T = 2000 chain (one-off: 15-25 minutes on 8 cores)
        ddpm_cases += gc.DDPM_FULL_CASES
    if ddpm_cases:
        for cid, ds, B, H, W, T, seed in ddpm_cases:
            if args.skip_long and T * H * W > 100 * 32 * 32:
                continue
            C = gc.DATASETS[ds][0]
            cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
            d = make_diffusion(D, net_for(ds), C, T, H)
            snaps = {}
            want = gc.DDPM_SNAPSHOTS.get(cid, [])
            if want:
                orig = d.p_sample
                count = [0]

                def wrapped(*a, **k):
                    r = orig(*a, **k)
                    count[0] += 1
                    if count[0] in want:
                        snaps[f"after_{count[0]}"] = r.clone()
                    return r

                d.p_sample = wrapped
            torch.manual_seed(seed)
            t0 = time.time()
            out = d(cond, mode="ddpm_sample")
            print(f"  {cid}: {time.time() - t0:.1f}s")
            save(cid, out=out, cond_chk=chk(cond), **snaps)

    if "ddim" in only:
        for cid, ds, B, H, W, T, sect, seed in gc.DDIM_CASES:
            C = gc.DATASETS[ds][0]
            cond = gc.tiles_for(ds, B, H, W, seed=seed)["cond"]
            d = make_diffusion(D, net_for(ds), C, T, H)
            torch.manual_seed(seed)
            out = d(cond, mode="ddim_sample", section_counts=sect)
            save(cid, out=out, cond_chk=chk(cond), num_timesteps_after=d.num_timesteps)

    dpm_cases = list(gc.DPM_CASES) if "dpm" in only else []
    if "dpmbig" in only:  # round 6: configs[2] at the benchmarked tile size
        dpm_cases += gc.DPM_BIG_CASES
    if dpm_cases:
        for cid, ds, H, W, T, steps, order, seed in dpm_cases:
            C = gc.DATASETS[ds][0]
            cond = gc.tiles_for(ds, 1, H, W, seed=seed)["cond"]
            d = make_diffusion(D, net_for(ds), C, T, H)
            ns = S.NoiseScheduleVP("discrete", betas=d.betas)
            lms = cond[:, :C]

            def corr(x0, t, lms=lms):
                return (x0 + lms).clamp(0, 1.0) - lms

            fn = S.model_wrapper(net_for(ds), ns, model_type="x_start", guidance_type="classifier-free",
                                 guidance_scale=1.0, condition=cond)
            slv = S.DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=corr)
            g = torch.Generator().manual_seed(seed)
            xT = torch.randn(1, C, H, W, generator=g)
            with torch.no_grad():
                out = slv.sample(xT, steps=steps, order=order, skip_type="time_uniform", method="multistep")
            save(cid, out=out, cond_chk=chk(cond))

    if "loss" in only:
        for cid, ds, B, H, W, T, tvals, sc_branch, seed in gc.LOSS_CASES:
            C = gc.DATASETS[ds][0]
            tiles = gc.tiles_for(ds, B, H, W, seed=seed)
            cond = tiles["cond"]
            res = tiles["gt"] - tiles["lms"]
            d = make_diffusion(D, net_for(ds), C, T, H)
            g = torch.Generator().manual_seed(seed)
            noise = torch.randn(B, C, H, W, generator=g)
            tt = torch.tensor(tvals, dtype=torch.long)
            _ri, _rr = torch.randint, random.random
            D.torch.randint = lambda *a, **k: tt
            D.random.random = (lambda: 0.0) if sc_branch else (lambda: 1.0)
            try:
                with torch.no_grad():
                    loss, recon = d(res, mode="train", noise=noise, cond=cond)
            finally:
                D.torch.randint, D.random.random = _ri, _rr
            save(cid, loss=loss, recon=recon, cond_chk=chk(cond))

    if "psnr" in only:
        import importlib.util

        g = torch.Generator().manual_seed(5)
        a = torch.rand(8, 33, 35, generator=g)
        b = (a + 0.05 * torch.randn(8, 33, 35, generator=g)).clamp(0, 1)
        spec = importlib.util.spec_from_file_location("_ml", os.path.join(REF, "utils", "_metric_legacy.py"))
        ml = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(ml)
            acc = ml.analysis_accu(a.permute(1, 2, 0), b.permute(1, 2, 0), 4, choices=5)  # what AnalysisPanAcc calls (utils/metric.py:27-29)
            save("psnr", ref_psnr=acc["PSNR"], sam=acc["SAM"], ergas=acc["ERGAS"], cc=acc["CC"])
        except Exception as e:  # pragma: no cover
            print("  psnr fixture skipped:", repr(e))


if __name__ == "__main__":
    main()
