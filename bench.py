#!/usr/bin/env python3
"""bench.py -- fused megapixels/sec of the DDIF sampler on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one complete sampler call over one batch of synthetic WV3-shaped tiles already resident in HBM:
set_cond (cond-only precompute) + T=1000 DDPM p_sample steps + (img + lms).clip(0,1) (+ the RCCL all-gather that
stitches the tiles of all ranks when N > 1).  Workload at N=1 = BASELINE.json configs[1]: batch 64 of 64x64x8 tiles,
T=1000 (computed in fp32 -- the parity configuration; bf16 is not used).  Tiles shard across ranks (weak scaling:
every rank samples its own 64 tiles, noise keyed by global tile index).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : dominant kernel class (3x3 implicit-GEMM convolutions on v_mfma_f32_32x32x2_f32) timed with HIP events
                 on the launch stream inside the timed region, algorithmic flops / duration vs the dense fp32 MFMA peak
  cpu_baseline : the CPU oracle (a port of the reference sampler, oracle/ddif_oracle.py) timed on this box's host cores
                 on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between the ranks of a node needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "dif-pan_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

_T0 = time.time()


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix (= vector) peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2516.8  # dense bf16 MFMA (16 x the fp32 rate); the bf16x3 path issues 6 bf16 products per fp32 product
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on (default; what the driver runs)
    "wv3": dict(ds="wv3", C=8, P=1, batch=64, tile=64, T=1000, sampler="ddpm",
                metric="fused megapixels/sec at T=%d, WV3 64x64x8 tiles"),
    # configs[2]: GF2 512x512 scene = 64 tiles of 64x64, DPM-Solver++ 2M, 50 model evaluations (per GPU: 64 tiles at N=1)
    "gf2_dpm50": dict(ds="gf2", C=4, P=1, batch=64, tile=64, T=1000, sampler="dpmpp2m", nfe=50,
                      metric="fused megapixels/sec, GF2 64x64x4 tiles, DPM-Solver++ 2M %d NFE"),
    # configs[3]: CAVE 31-band HSI + 3-band MSI, 128x128 patches, T=2000 DDPM
    "cave128_t2000": dict(ds="cave", C=31, P=3, batch=8, tile=128, T=2000, sampler="ddpm",
                          metric="fused megapixels/sec at T=%d, CAVE 128x128x31 patches"),
    # configs[4]: one training iteration of sr3_dwt on WV3 tiles, batch 32 per GPU, AdamW, DDP over the ranks (engine_google's loop body)
    "wv3_train_b32": dict(ds="wv3", C=8, P=1, batch=32, tile=64, T=3000, sampler="train",
                          metric="training tiles/sec, sr3_dwt on WV3 64x64x8 tiles, batch 32 per GPU (T=%d schedule)"),
}


def build_id():
    """sha1 over the kernel / host sources the library is built from: ties a committed PMC profile to the build it measured."""
    import glob
    import hashlib

    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "dif-pan_amd", "csrc", "*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="wv3", choices=sorted(CONFIGS), help="wv3 = BASELINE.json's metric configuration (default)")
    ap.add_argument("--batch", type=int, default=0, help="tiles per GPU (0: the configuration's own)")
    ap.add_argument("--T", type=int, default=0, help="diffusion steps (0: the configuration's own)")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="development aid: another build of libddif.so to benchmark (A/B of kernel variants)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU time budget of the cpu_baseline leg (split over B=1 and B=8)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: CPUs this process may run on (sched_getaffinity)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import ddif
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule
    from ddif.layout import engine_cfg
    from ddif.models.sr3_dwt import UNetSR3
    from ddif.synth import synth_state_dict, synth_tiles

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ddif hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if args.lib:
        from ddif import runtime as _rt

        _rt.use_library(os.path.abspath(args.lib))
    lib = ddif.get_lib()
    assert not lib.emulated
    cf = CONFIGS[args.config]
    if cf["sampler"] == "train":
        return bench_training(args, cf, rank, world, dev)
    C, P = cf["C"], cf["P"]
    B, H, T = args.batch or cf["batch"], args.tile or cf["tile"], args.T or cf["T"]
    order = "hisr" if cf["ds"] == "cave" else "pan"
    cfg = engine_cfg(C, P)
    keys = ("in_channel", "out_channel", "inner_channel", "lms_channel", "pan_channel", "norm_groups", "channel_mults",
            "attn_res", "res_blocks", "dropout", "image_size", "self_condition")
    sd = synth_state_dict(cfg, 1234)
    net = UNetSR3(**{k: cfg[k] for k in keys})
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    diffusion = GaussianDiffusion(net, image_size=H, channels=C, pred_mode="x_start", loss_type="l1", device=dev,
                                  clamp_range=(0, 1))
    diffusion.set_new_noise_schedule(betas=make_beta_schedule("cosine", T, cosine_s=8e-3), device=dev)

    log("network built (%d params), generating %d synthetic tiles" % (sum(p.numel() for p in net.parameters()), B))
    tiles = synth_tiles(B, C, P, H, H, seed=100 + rank, order=order)
    cond = tiles["cond"].to(dev)
    lms = cond[:, :C].contiguous()
    gathered = torch.empty((world * B, C, H, H), device=dev) if world > 1 else None
    plan = diffusion._plan(cond)  # builds workspaces + runs set_cond once (not timed)
    cost = plan.cost()
    mem = plan.memory()
    torch.cuda.synchronize()
    log("plan ready: %.3f GFLOP and %.1f MB (algorithmic) per denoising step of the batch" % (cost["step_flop"] / 1e9, cost["step_bytes"] / 1e6))

    solver = None
    n_evals = T
    if cf["sampler"] == "dpmpp2m":
        from ddif.solver.dpm_solver import DPM_Solver, ImageSpaceClamp, NoiseScheduleVP, model_wrapper

        n_evals = cf["nfe"]
        ns = NoiseScheduleVP("discrete", betas=diffusion.betas)
        fn = model_wrapper(net, ns, model_type="x_start", guidance_type="classifier-free", guidance_scale=1.0, condition=cond)
        solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=ImageSpaceClamp(lms, 0.0, 1.0))
        assert solver._fused_target() is not None  # the whole solver loop runs inside libddif

    def one_step(seed):
        plan.set_cond(cond, force=True)  # once-per-tile precompute is part of the job
        if solver is not None:
            xT = torch.randn((B, C, H, H), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
            res = solver.sample(xT, steps=n_evals, order=2, skip_type="time_uniform", method="multistep")
        else:
            res = diffusion(cond, mode="ddpm_sample", seed=seed, tile0=rank * B, device_rng=True)
        sr = (res + lms).clip(0, 1)  # diffusion_engine.py:446-447
        if world > 1:
            dist.all_gather_into_tensor(gathered, sr)  # stitch: every rank ends with the whole scene
            return gathered
        return sr

    for w in range(args.warmup):
        tw = time.perf_counter()
        one_step(1000 + w)
        torch.cuda.synchronize()
        log("warmup step %d: %.3f s" % (w, time.perf_counter() - tw))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    prof_every = max(10, n_evals // 20)  # a profiled step has an event pair around every launch: keep it to <= 1 step in 10
    plan.prof_begin(prof_every, 16384)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for k in range(args.steps):
        out = one_step(2000 + k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = plan.prof_collect()
    log("timed region: %.3f s for %d step(s)" % (dt, args.steps))
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out is not None and bool(torch.isfinite(out).all())

    mp_per_step = world * B * H * H / 1e6
    value = mp_per_step * args.steps / dt
    ach_tflops = prof["total_flop"] / (prof["total_ms"] * 1e-3) / 1e12 if prof["total_ms"] > 0 else 0.0
    step_flop_total = cost["step_flop"] * n_evals + cost["cond_flop"]
    # the committed PMC passes are of the default (wv3, B = 64) command: other configurations report no traffic
    traffic, traffic_info = committed_traffic() if (args.config == "wv3" and B == 64) else (None, {"traffic_from_committed_profile": False})
    x3 = "bf16x3" in prof["kernel"]
    # per-class breakdown of the profiled denoising steps (every launch of those steps sits between two HIP events)
    n_prof_steps = max(1, sum(1 for k in range(n_evals) if k % prof_every == 0) * args.steps)
    classes, cls_ms_total = [], 0.0
    for c in prof["classes"]:
        if not c["launches"]:
            continue
        ms = c["total_ms"] / n_prof_steps
        floor_ms = max(c["total_flop"] / (PEAK_F32_MFMA_TFLOPS * 1e12), c["total_bytes"] / (PEAK_HBM_GBS * 1e9)) * 1e3 / n_prof_steps
        cls_ms_total += ms
        classes.append({"class": c["name"], "launches_per_step": c["launches"] / n_prof_steps, "ms_per_step": ms,
                        "tflops": c["total_flop"] / (c["total_ms"] * 1e-3) / 1e12, "algorithmic_gbytes_per_s": c["total_bytes"] / (c["total_ms"] * 1e-3) / 1e9,
                        "floor_ms_per_step": floor_ms, "frac_of_floor": floor_ms / ms if ms > 0 else None})
    if cf["sampler"] == "dpmpp2m":
        metric = cf["metric"] % n_evals
        workload = "GF2 pansharpening, batch %d of %dx%dx%d tiles per GPU, DPM-Solver++ 2M %d NFE (T=%d schedule), fp32" % (B, H, H, C, n_evals, T)
    else:
        metric = cf["metric"] % T
        workload = "%s, batch %d of %dx%dx%d tiles per GPU, T=%d DDPM p_sample, fp32" % ("WV3 pansharpening" if cf["ds"] == "wv3" else "CAVE MHIF", B, H, H, C, T)
    result = {
        "metric": metric,
        "value": value,
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "name": args.config, "tiles_per_gpu": B, "tile": [H, H, C], "T": T, "model_evaluations": n_evals,
                   "sampler": cf["sampler"], "parallelism": "tile-shard x%d" % world,
                   "plan_memory_mb": {"total": mem["total_bytes"] / 1e6, "step_activation_arena": mem["arena_bytes"] / 1e6,
                                      "same_activations_unaliased": mem["unaliased_bytes"] / 1e6},
                   "conv_math": ("fp32 operands split into 3 bf16 planes, 6 exact products on v_mfma_f32_32x32x16_bf16, fp32 accumulate (3x3 convs, "
                                 "wide 1x1 convs, low-resolution levels); exact fp32 MFMA elsewhere") if x3 else "exact fp32 MFMA"},
        "roofline": {
            "bound": "mfma",
            "achieved": ach_tflops,
            "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": ach_tflops / PEAK_F32_MFMA_TFLOPS,
            "peak_note": "dense fp32 matrix peak (guide); `achieved` counts ALGORITHMIC fp32 flops (2*M*N*K, unpadded) of the dominant class",
            "mfma_issued_tflops": (6.0 * ach_tflops) if x3 else ach_tflops,
            "mfma_issued_note": ("every fp32 product is issued as 6 bf16 MFMA products: issued rate vs the dense bf16 peak %.1f TF" % PEAK_BF16_MFMA_TFLOPS) if x3 else "exact fp32 MFMA",
            "frac_of_bf16_mfma_peak_issued": (6.0 * ach_tflops / PEAK_BF16_MFMA_TFLOPS) if x3 else None,
            "traffic": traffic,
            "traffic_unit": "bytes of HBM traffic per launch of the dominant class (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of this command)",
            **traffic_info,
            "algorithmic_bytes_per_launch": (prof["total_bytes"] / prof["launches"]) if prof["launches"] else None,
            "kernel": prof["kernel"],
            "launches_timed": prof["launches"],
            "avg_launch_us": (prof["total_ms"] * 1e3 / prof["launches"]) if prof["launches"] else None,
            "algorithmic_gflop_per_launch": (prof["total_flop"] / prof["launches"] / 1e9) if prof["launches"] else None,
            "whole_step": {
                "ms_per_denoising_step": dt / args.steps * 1e3 / n_evals,
                "tflops": step_flop_total * args.steps / dt / 1e12,
                "frac_of_f32_mfma_peak": step_flop_total * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "hbm_frac": (cost["step_bytes"] * n_evals + cost["cond_bytes"]) * args.steps / dt / 1e9 / PEAK_HBM_GBS,
                "classes_note": "HIP events around every launch of one denoising step in %d; floor = max(flops / 157.3 TF, algorithmic bytes / 8 TB/s)" % prof_every,
                "classes_ms_per_step_sum": cls_ms_total,
                "classes": classes,
            },
        },
        "build_id": build_id(),
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("gpu result: %s" % json.dumps({k: result[k] for k in ("value", "ms_per_step")}))
        cb = cpu_baseline(sd, cfg, tiles["cond"][:8].contiguous(), T, args.cpu_seconds, args.cpu_threads)
        result["cpu_baseline"] = cb
        result["vs_cpu_baseline"] = value / cb["value"]
        result["vs_cpu_baseline_b1"] = value / cb["by_batch"]["1"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def bench_training(args, cf, rank, world, dev):
    """BASELINE configs[4]: one "step" = one training iteration as engine_google runs it (reference diffusion_engine.py:218-241): q_sample +
    self-conditioning draw + forward + loss.backward() through the library's reverse pass + gradient all-reduce over the ranks + fused
    clip / AdamW / EMA.  Weak scaling: every rank trains its own batch; value = tiles per second over all ranks."""
    import random

    import torch
    import torch.distributed as dist

    from ddif import runtime
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule
    from ddif.diffusion_engine import average_gradients
    from ddif.layout import engine_cfg
    from ddif.models.sr3_dwt import UNetSR3
    from ddif.synth import synth_state_dict, synth_tiles

    C, P, B, H, T = cf["C"], cf["P"], args.batch or cf["batch"], args.tile or cf["tile"], args.T or cf["T"]
    cfg = engine_cfg(C, P)
    keys = ("in_channel", "out_channel", "inner_channel", "lms_channel", "pan_channel", "norm_groups", "channel_mults", "attn_res", "res_blocks", "dropout",
            "image_size", "self_condition")
    net = UNetSR3(**{k: cfg[k] for k in keys})
    net.load_state_dict(synth_state_dict(cfg, 1234))
    net = net.to(dev).train()
    d = GaussianDiffusion(net, image_size=H, channels=C, pred_mode="x_start", loss_type="l1", device=dev, clamp_range=(0, 1))
    d.set_new_noise_schedule(betas=make_beta_schedule("cosine", T, cosine_s=8e-3), device=dev)
    tiles = synth_tiles(B, C, P, H, H, seed=100 + rank)
    cond = tiles["cond"].to(dev)
    res = (tiles["gt"].to(dev) - cond[:, :C]).contiguous()
    params = [p for p in net.parameters()]
    grads = [torch.zeros_like(p) for p in params]
    for p, g in zip(params, grads):
        p.grad = g
    ema = [p.detach().clone() for p in params]
    opt = runtime.FusedAdamW(params, grads, ema, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    torch.manual_seed(7 + rank)
    random.seed(7 + rank)

    def one_step():
        for g in grads:
            g.zero_()
        loss, _ = d(res, cond=cond)
        loss.backward()
        if world > 1:
            average_gradients(grads, world)
        opt.step(max_grad_norm=0.003, ema_mode=1, ema_decay=0.995)
        return loss

    for w in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    loss = None
    for k in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(loss.detach()).all())
    if rank == 0:
        line = {"metric": cf["metric"] % T, "value": world * B * args.steps / dt, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "sr3_dwt training iteration (q_sample, 50 %% self-conditioning pass, forward, backward, DDP all-reduce, clip + AdamW + EMA), "
                                       "WV3 %dx%dx%d tiles, batch %d per GPU, fp32" % (H, H, C, B), "name": args.config, "tiles_per_gpu": B, "tile": [H, H, C],
                           "T": T, "parallelism": "ddp x%d" % world},
                "roofline": None, "cpu_baseline": None, "build_id": build_id(),
                "note": "correctness-first training graph (ddif/train.py): per-kernel split in profiles/r02_z_train_kernel_stats_after.csv"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def committed_traffic():
    """HBM bytes per launch of the dominant kernel class.  PMC counters need their own rocprofv3 passes (they cannot be
    collected from inside this process), so this reads the newest summary committed under profiles/ (tools/gpu_full.sh +
    tools/pmc_traffic.py, same `bench.py` command) -- and ONLY reports it when that profile was taken from the build being
    benchmarked (same sha1 over csrc/); otherwise `traffic` is null and the stale file is named."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, {"traffic_from_committed_profile": False}
    bid = build_id()
    for path in reversed(files):  # the set taken from THIS build, whatever its tag sorts like
        try:
            with open(path) as f:
                j = json.load(f)
            if j.get("build_id") == bid:
                return float(j["class_hbm_bytes_per_launch"]), {"traffic_from_committed_profile": True, "traffic_source": "profiles/" + os.path.basename(path)}
        except (OSError, ValueError, KeyError):
            continue
    return None, {"traffic_from_committed_profile": False, "traffic_stale_profile": "profiles/" + os.path.basename(files[-1])}


def usable_cpus():
    """CPUs this process can actually use: min(affinity mask, cgroup CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(sd, cfg, cond8, T, budget_s, threads=0):
    """The oracle (CPU port of the reference DDPM sampler: same torch-CPU op sequence, no hoisting, no fusion) on the host
    cores of this box at B=1 (BASELINE config 1) and at B=8 (the CPU's best operating point): the first n of the
    T steps of each, scaled to T.  `value` is the BETTER of the two (the honest multiple); both are listed."""
    import torch

    from oracle import ddif_oracle as O

    if threads <= 0:
        threads = usable_cpus()
    torch.set_num_threads(threads)
    log("cpu baseline: %d threads (os.cpu_count() = %s)" % (threads, os.cpu_count()))
    tabs = O.schedule_tables(O.cosine_betas(T))
    g = torch.Generator().manual_seed(1)
    by = {}
    for bsz, share in ((1, 0.4), (8, 0.6)):
        cond = cond8[:bsz].contiguous()

        def run(n):
            t0 = time.perf_counter()
            with torch.no_grad():
                O.ddpm_sample(sd, cfg, cond, tabs, noise_fn=lambda s: torch.randn(s, generator=g), max_steps=n)
            return time.perf_counter() - t0

        run(1)  # warm-up (oneDNN primitive creation)
        t_probe = run(2) / 2
        n = int(max(2, min(T, share * budget_s / max(t_probe, 1e-4))))
        dt = run(n)
        per_step = dt / n
        mp = cond.shape[0] * cond.shape[2] * cond.shape[3] / 1e6
        by[str(bsz)] = {"value": mp / (per_step * T), "seconds_per_step": per_step, "steps_timed": n, "seconds_timed": dt}
        log("cpu baseline B=%d: %.3f s/step over %d steps -> %.3e MP/s" % (bsz, per_step, n, by[str(bsz)]["value"]))
    best = max(by, key=lambda k: by[k]["value"])
    return {
        "value": by[best]["value"],
        "unit": "MP/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": "oracle/ddif_oracle.ddpm_sample on tiles %dx%dx%d: first %d of T=%d steps at B=1 (%.1f s) and first %d at B=8 (%.1f s), each scaled to T; value = B=%s (the better)"
                  % (cond8.shape[2], cond8.shape[3], cfg["out_channel"], by["1"]["steps_timed"], T, by["1"]["seconds_timed"], by["8"]["steps_timed"], by["8"]["seconds_timed"], best),
        "best_batch": int(best),
        "by_batch": by,
        "host_cpus": os.cpu_count(),
    }


if __name__ == "__main__":
    main()
