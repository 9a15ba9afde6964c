#!/usr/bin/env python3
"""bench.py -- fused megapixels/sec of the DDIF sampler on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks (`python -m torch.distributed.run ... bench.py --gpus N`:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or, when WORLD_SIZE is not set, this script launches them
ITSELF: the parent process -- which never imports torch and never touches a GPU -- starts `torch.distributed.run` as a child
process, relays rank 0's JSON line and exits with the child's status.

One "step" = one complete sampler call over one batch of synthetic WV3-shaped tiles already resident in HBM:
set_cond (cond-only precompute) + T=1000 DDPM p_sample steps + (img + lms).clip(0,1) (+ the RCCL all-gather that
stitches the tiles of all ranks when N > 1).  Workload at N=1 = BASELINE.json configs[1]: batch 64 of 64x64x8 tiles,
T=1000 (computed in fp32 -- the parity configuration; bf16 is not used).  Tiles shard across ranks (weak scaling:
every rank samples its own 64 tiles, noise keyed by global tile index).

`--config gf2_dpm50` is BASELINE.json configs[2] as stated: ONE 512x512 GF2 scene = 64 tiles split over the N ranks (STRONG
scaling: 64 / N tiles per GPU), DPM-Solver++ 2M 50 NFE, all-gather + stitch of the scene inside the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : dominant kernel class (3x3 implicit-GEMM convolutions, f16x2 split products on v_mfma_f32_32x32x16_f16) timed
                 with HIP events on the launch stream inside the timed region; `achieved` = algorithmic fp32 flops / duration,
                 `peak` = the dense 16-bit MFMA peak / the products the class issues per fp32 product (3 on the f16x2 path, 6 on
                 bf16x3, 16 on the exact fp32 MFMA; reported by the library per launch), so `frac` is the issued fraction of the
                 matrix pipe the kernel actually runs on
  cpu_baseline : the CPU oracle (a port of the reference sampler, oracle/ddif_oracle.py) timed on this box's host cores
                 on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between the ranks of a node needs it on this driver
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (also set by the ddif package; here for the self-launched ranks: profiles/r05/a_kernarg_ab.txt)

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "dif-pan_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

_T0 = time.time()


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix (= vector) peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2516.8  # dense bf16 / fp16 MFMA (16 x the fp32 rate); the f16x2 path issues 3 half products per fp32 product, bf16x3 issues 6
PEAK_HBM_GBS = 8000.0
PROF_STEPS_TARGET = 40  # denoising steps bracketed with events over the whole timed region (every launch of such a step has its own pair)

CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on (default; what the driver runs)
    "wv3": dict(ds="wv3", C=8, P=1, batch=64, tile=64, T=1000, sampler="ddpm", scaling="weak",
                metric="fused megapixels/sec at T=%d, WV3 64x64x8 tiles"),
    # configs[1]'s "bf16": the THROUGHPUT variant -- conv operands rounded once to bf16, one MFMA product, fp32 accumulate (ddif_set_math_mode).  Never the
    # headline and never a parity configuration: the line carries its measured drift against the fp32-class path on the same tiles, seeds and steps.
    "wv3_bf16": dict(ds="wv3", C=8, P=1, batch=64, tile=64, T=1000, sampler="ddpm", scaling="weak", math="bf16",
                     metric="fused megapixels/sec at T=%d, WV3 64x64x8 tiles, bf16 conv operands (throughput variant)"),
    # configs[2]: ONE GF2 512x512 scene = 64 tiles of 64x64 split over the ranks (strong scaling), DPM-Solver++ 2M, 50 model evaluations,
    # all-gather + stitch inside the timed region
    "gf2_dpm50": dict(ds="gf2", C=4, P=1, batch=64, tile=64, T=1000, sampler="dpmpp2m", nfe=50, scaling="strong",
                      metric="fused megapixels/sec, GF2 512x512 scene as 64x64x4 tiles, DPM-Solver++ 2M %d NFE"),
    # configs[3]: CAVE 31-band HSI + 3-band MSI, 128x128 patches, T=2000 DDPM
    "cave128_t2000": dict(ds="cave", C=31, P=3, batch=8, tile=128, T=2000, sampler="ddpm", scaling="weak",
                          metric="fused megapixels/sec at T=%d, CAVE 128x128x31 patches"),
    # configs[4]: one training iteration of sr3_dwt on WV3 tiles, batch 32 per GPU, AdamW, DDP over the ranks (engine_google's loop body)
    "wv3_train_b32": dict(ds="wv3", C=8, P=1, batch=32, tile=64, T=3000, sampler="train", scaling="weak",
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="wv3", choices=sorted(CONFIGS), help="wv3 = BASELINE.json's metric configuration (default)")
    ap.add_argument("--batch", type=int, default=0, help="tiles per GPU (gf2_dpm50: tiles of the whole scene, split over the GPUs); 0: the configuration's own")
    ap.add_argument("--T", type=int, default=0, help="diffusion steps (0: the configuration's own)")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="development aid: another build of libddif.so to benchmark (A/B of kernel variants)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU time budget of the cpu_baseline leg")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: CPUs this process may run on (sched_getaffinity)")
    ap.add_argument("--no-parity", action="store_true", help="skip the golden-vector parity probe of the default line (N = 1, config wv3)")
    ap.add_argument("--no-bracket", action="store_true", help="skip the exact-fp32 bracket run of the default line (a child process with DDIF_F16=0 DDIF_X3=0)")
    ap.add_argument("--no-shares", action="store_true", help="gf2_dpm50 at N = 1: skip the per-rank shares (32 / 16 / 8 tiles) behind projected_strong_scaling")
    return ap.parse_args(argv)


def self_launch(args):
    """`bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree (torch.distributed.run, one rank per GPU) from
    this parent, which has not imported torch and never initialises a GPU (no exec of a GPU-initialised process anywhere), relay rank 0's
    JSON line on stdout and exit non-zero when any rank failed."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log("no WORLD_SIZE in the environment: launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0 or line is None:
        log("multi-GPU launch failed (exit status %d, %s result line)" % (rc, "with a" if line else "no"))
        raise SystemExit(rc if rc != 0 else 1)
    print(line, flush=True)
    raise SystemExit(0)


def main():
    args = parse_args()
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ  # (WORLD_SIZE alone -- exported by a scheduler, no rendezvous -- is not a launcher: ADVICE r5)
    if args.gpus > 1 and not under_launcher:
        self_launch(args)

    import torch
    import torch.distributed as dist

    import ddif
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule
    from ddif.layout import engine_cfg
    from ddif.models.sr3_dwt import UNetSR3
    from ddif.sharding import shard_range, stitch_tiles
    from ddif.synth import synth_state_dict, synth_tiles

    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ddif hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a launcher (WORLD_SIZE in the environment) the process group is RCCL even at world size 1, and every collective of the multi-GPU path
    # runs (an all-gather / all-reduce over one rank): `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` is the RCCL smoke of the
    # 1-GPU lease (tests/test_rccl_world1.py).  `python bench.py` (the driver's N = 1 line) has no process group and no collectives.
    if under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if args.lib:
        from ddif import runtime as _rt

        _rt.use_library(os.path.abspath(args.lib))
    lib = ddif.get_lib()
    assert not lib.emulated
    if os.environ.get("DDIF_BENCH_GRID_CAP"):  # development aid (tools/gpu_r6v.sh): cap the persistent grid of every conv launch (the test hook ddif_debug_set_grid_cap)
        from ddif import runtime as _rt

        _rt.set_debug_grid_cap(int(os.environ["DDIF_BENCH_GRID_CAP"]))
    cf = CONFIGS[args.config]
    bf16 = cf.get("math") == "bf16"
    if bf16:
        from ddif import runtime as _rt

        _rt.set_math_mode("bf16")
    if cf["sampler"] == "train":
        return bench_training(args, cf, rank, world, dev)
    strong = cf["scaling"] == "strong"
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

    if strong:
        # one scene of `total` tiles (the same on every rank), this rank's contiguous block of it; noise keyed by the tile's index in the scene
        total = B
        from ddif.sharding import scene_grid

        ny, nx = scene_grid(total)
        if total % world:
            raise SystemExit("gf2_dpm50: %d scene tiles must be divisible by %d GPUs" % (total, world))
        lo, hi = shard_range(total, rank, world)
        tiles = synth_tiles(total, C, P, H, H, seed=100, order=order)
        tiles = {k: v[lo:hi].contiguous() for k, v in tiles.items()}
        B, tile0 = hi - lo, lo
    else:
        total, ny, nx = world * B, 0, 0
        tiles = synth_tiles(B, C, P, H, H, seed=100 + rank, order=order)
        tile0 = rank * B
    log("network built (%d params), %d synthetic tiles on this rank" % (sum(p.numel() for p in net.parameters()), B))
    cond = tiles["cond"].to(dev)
    lms = cond[:, :C].contiguous()
    gathered = torch.empty((total, C, H, H), device=dev) if (dist.is_initialized() or strong) else None
    plan = diffusion._plan(cond)  # builds workspaces + runs set_cond once (not timed)
    cost = plan.cost()
    mem = plan.memory()
    n_launch = plan.num_launches()
    torch.cuda.synchronize()
    log("plan ready: %.3f GFLOP and %.1f MB (algorithmic) per denoising step of the batch, %d launches per step"
        % (cost["step_flop"] / 1e9, cost["step_bytes"] / 1e6, n_launch["step"]))

    solver = None
    n_evals = T
    if cf["sampler"] == "dpmpp2m":
        from ddif.solver.dpm_solver import DPM_Solver, ImageSpaceClamp, NoiseScheduleVP, model_wrapper

        n_evals = cf["nfe"]
        ns = NoiseScheduleVP("discrete", betas=diffusion.betas)
        fn = model_wrapper(net, ns, model_type="x_start", guidance_type="classifier-free", guidance_scale=1.0, condition=cond)
        solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=ImageSpaceClamp(lms, 0.0, 1.0))
        assert solver._fused_target() is not None  # the whole solver loop runs inside libddif

    ag_pairs = []  # (start, end) HIP events around every all-gather of the timed region (read after the final synchronise: nothing waits on them inside it)

    def one_step(seed, timed=False):
        ag_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if (timed and dist.is_initialized()) else None
        plan.set_cond(cond, force=True)  # once-per-tile precompute is part of the job
        if solver is not None:
            if strong:  # x_T of the whole scene from one seeded stream, this rank's block of it: independent of the GPU count
                xT = torch.randn((total, C, H, H), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))[tile0:tile0 + B].contiguous()
            else:
                xT = torch.randn((B, C, H, H), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
            res = solver.sample(xT, steps=n_evals, order=2, skip_type="time_uniform", method="multistep")
        else:
            res = diffusion(cond, mode="ddpm_sample", seed=seed, tile0=tile0, device_rng=True)
        sr = (res + lms).clip(0, 1)  # diffusion_engine.py:446-447
        if dist.is_initialized():
            if ag_ev is not None:
                ag_ev[0].record()
            dist.all_gather_into_tensor(gathered, sr)  # the only exchange: every rank ends with all tiles
            if ag_ev is not None:
                ag_ev[1].record()
                ag_pairs.append(ag_ev)
            sr = gathered
        if strong:
            return stitch_tiles(sr, ny, nx)  # (C, 512, 512): the fused scene
        return sr

    for w in range(args.warmup):
        tw = time.perf_counter()
        one_step(1000 + w)
        torch.cuda.synchronize()
        log("warmup step %d: %.3f s" % (w, time.perf_counter() - tw))
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    # profiled denoising steps: about PROF_STEPS_TARGET over the whole timed region, at most one step in 10 (a profiled step is launched
    # kernel by kernel with an event pair around each); the event buffer is sized for exactly those steps, and the library reports how
    # many whole steps it recorded (ddif_prof_collect.steps_recorded) -- per-step figures divide by THAT
    per_job = max(1, min(n_evals // 10 if n_evals >= 10 else 1, PROF_STEPS_TARGET // max(1, args.steps)))
    prof_every = -(-n_evals // per_job)
    n_prof_planned = len(range(0, n_evals, prof_every)) * args.steps
    plan.prof_begin(prof_every, n_prof_planned * n_launch["step"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for k in range(args.steps):
        out = one_step(2000 + k, timed=True)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0  # this rank's own time (before the closing barrier): a straggler GPU shows in per_rank_ms_per_step below
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = plan.prof_collect()
    log("timed region: %.3f s for %d step(s); %d of %d planned denoising steps profiled" % (dt, args.steps, prof["steps_recorded"], n_prof_planned))
    rccl = None
    if dist.is_initialized():
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # what the one line the driver keeps must show of a multi-GPU run: how many ranks RCCL really had, every rank's own time, the stitch collective's cost
        mine = torch.tensor([dt_rank / args.steps * 1e3], device=dev, dtype=torch.float64)
        allr = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        per_rank = [float(v) for v in allr.tolist()]
        ag_us = [1e3 * a.elapsed_time(b) for a, b in ag_pairs]
        ag_bytes = gathered.numel() * gathered.element_size()
        ag_mean = sum(ag_us) / len(ag_us) if ag_us else None
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                "per_rank_ms_per_step": {"min": min(per_rank), "max": max(per_rank), "all": per_rank},
                "allgather": {"bytes_gathered": ag_bytes, "us_mean": ag_mean, "us_max": max(ag_us) if ag_us else None, "calls": len(ag_us),
                              "algorithmic_gbytes_per_s": (ag_bytes / (ag_mean * 1e-6) / 1e9) if ag_mean else None,
                              "bus_gbytes_per_s": ((world - 1) / world * ag_bytes / (ag_mean * 1e-6) / 1e9) if ag_mean else None,
                              "note": "all_gather_into_tensor of the fused tiles (the only exchange of the path), HIP events on the launch stream, rank 0; "
                                      "bus = (N-1)/N x bytes / time (ring convention)"}}
    assert out is not None and bool(torch.isfinite(out).all())
    drift = None
    if bf16:
        # the same job (tiles, seed, T steps) once more on the fp32-class path: what the single bf16 product costs in the fused image
        from ddif import runtime as _rt

        seed_last = 2000 + args.steps - 1
        _rt.set_math_mode("split")
        plan_ref = diffusion._plan(cond)
        plan_ref.set_cond(cond, force=True)
        ref = (diffusion(cond, mode="ddpm_sample", seed=seed_last, tile0=tile0, device_rng=True) + lms).clip(0, 1)
        torch.cuda.synchronize()
        _rt.set_math_mode("bf16")
        mine = out[tile0:tile0 + B] if dist.is_initialized() else out
        d = (mine - ref).double()
        mse_t = d.pow(2).mean(dim=(1, 2, 3))
        drift = {"against": "the fp32-class (split-product) path of this library: same tiles, seed %d, T=%d steps, batch %d" % (seed_last, T, B),
                 "max_abs": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()), "rel_l2": float(d.norm() / ref.double().norm()),
                 "psnr_db_min_over_tiles": float((10.0 * torch.log10(1.0 / mse_t.clamp_min(1e-30))).min()), "data_range": [0.0, 1.0]}
        log("bf16 drift against the fp32-class path after %d steps: max %.3e, rms %.3e, worst-tile PSNR %.1f dB" % (T, drift["max_abs"], drift["rms"], drift["psnr_db_min_over_tiles"]))

    mp_per_step = total * H * H / 1e6
    value = mp_per_step * args.steps / dt
    ach_tflops = prof["total_flop"] / (prof["total_ms"] * 1e-3) / 1e12 if prof["total_ms"] > 0 else 0.0
    step_flop_total = cost["step_flop"] * n_evals + cost["cond_flop"]
    # the committed PMC passes are of the default (wv3, B = 64) command: other configurations report no traffic
    traffic, traffic_info = committed_traffic() if (args.config == "wv3" and B == 64) else (None, {"traffic_from_committed_profile": False})
    x3 = "split products" in prof["kernel"] and not bf16
    # products the dominant class issues on the matrix pipe per algorithmic fp32 product (flop-weighted over its launches: 3 = f16x2, 6 = bf16x3,
    # 16 = exact fp32 MFMA in units of the 16-bit rate); its fp32-equivalent ceiling is the dense 16-bit MFMA peak over that
    products = (prof["total_mfma_flop"] / prof["total_flop"]) if prof["total_flop"] > 0 else 16.0
    peak = PEAK_BF16_MFMA_TFLOPS / products
    ms_step = dt / args.steps * 1e3 / n_evals
    # per-class breakdown of the profiled denoising steps (every launch of those steps sits between two HIP events)
    n_rec = prof["steps_recorded"]
    classes, cls_ms_total = [], 0.0
    for c in prof["classes"]:
        if not c["launches"] or not n_rec:
            continue
        ms = c["total_ms"] / n_rec
        floor_ms = max(c["total_mfma_flop"] / (PEAK_BF16_MFMA_TFLOPS * 1e12), c["total_bytes"] / (PEAK_HBM_GBS * 1e9)) * 1e3 / n_rec
        cls_ms_total += ms
        classes.append({"class": c["name"], "launches_per_step": c["launches"] / n_rec, "ms_per_step": ms,
                        "tflops": c["total_flop"] / (c["total_ms"] * 1e-3) / 1e12, "algorithmic_gbytes_per_s": c["total_bytes"] / (c["total_ms"] * 1e-3) / 1e9,
                        "floor_ms_per_step": floor_ms, "frac_of_floor": floor_ms / ms if ms > 0 else None})
    # the table is only printed when it adds up: sum of the classes within 15 % of the measured denoising step, no class above its floor
    classes_ok = bool(classes) and abs(cls_ms_total - ms_step) / ms_step < 0.15 and all(c["frac_of_floor"] is None or c["frac_of_floor"] <= 1.0 for c in classes)
    # ONE scale that does not move with the arithmetic: the step's floor = sum over the classes of max(issued 16-bit MFMA flops / dense peak,
    # algorithmic bytes / 8 TB/s) -- a property of the launch program, not of the timing -- over the measured denoising step
    step_floor_ms = sum(c["floor_ms_per_step"] for c in classes) if classes else None
    step_frac = (step_floor_ms / ms_step) if step_floor_ms else None
    if not classes_ok:
        log("per-class table INCONSISTENT with the step (sum %.3f ms vs %.3f ms per denoising step): not reported" % (cls_ms_total, ms_step))
    # one product per fp32 product: the dominant class's MFMA floor (flops / 2516.8 TF) drops below its HBM floor (algorithmic bytes / 8 TB/s) -- the
    # throughput variant is priced against HBM
    hbm_roofline = None
    if bf16 and prof["total_ms"] > 0 and prof["total_bytes"] / (PEAK_HBM_GBS * 1e9) > prof["total_mfma_flop"] / (PEAK_BF16_MFMA_TFLOPS * 1e12):
        ach_gbs = prof["total_bytes"] / (prof["total_ms"] * 1e-3) / 1e9
        hbm_roofline = {
            "bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gbs / PEAK_HBM_GBS, "traffic": None,
            "peak_note": "dominant class (3x3 convs of the 64^2 / 32^2 levels) with ONE bf16 product per fp32 product: algorithmic bytes / 8 TB/s exceeds flops / %.1f TF" % PEAK_BF16_MFMA_TFLOPS,
            "mfma_products_per_fp32_product": products, "mfma_tflops": ach_tflops, "mfma_frac_of_dense_bf16_peak": ach_tflops * products / PEAK_BF16_MFMA_TFLOPS,
            "algorithmic_bytes_per_launch": (prof["total_bytes"] / prof["launches"]) if prof["launches"] else None,
            "kernel": prof["kernel"], "launches_timed": prof["launches"], "avg_launch_us": (prof["total_ms"] * 1e3 / prof["launches"]) if prof["launches"] else None,
            "step_floor_ms": step_floor_ms, "step_frac": step_frac,
            "whole_step": {"ms_per_denoising_step": ms_step, "tflops": step_flop_total * args.steps / dt / 1e12, "hbm_frac": (cost["step_bytes"] * n_evals + cost["cond_bytes"]) * args.steps / dt / 1e9 / PEAK_HBM_GBS,
                           "profiled_steps": n_rec, "classes_ms_per_step_sum": cls_ms_total if classes_ok else None, "classes": classes if classes_ok else None},
        }
    if cf["sampler"] == "dpmpp2m":
        metric = cf["metric"] % n_evals
        workload = "GF2 pansharpening, one %dx%d scene = %d tiles of %dx%dx%d split over %d GPU(s) (%d per GPU), DPM-Solver++ 2M %d NFE (T=%d schedule), all-gather + stitch, fp32" % (
            ny * H, nx * H, total, H, H, C, world, B, n_evals, T)
    else:
        metric = cf["metric"] % T
        workload = "%s, batch %d of %dx%dx%d tiles per GPU, T=%d DDPM p_sample, %s" % ("WV3 pansharpening" if cf["ds"] == "wv3" else "CAVE MHIF", B, H, H, C, T,
                                                                                        "bf16 conv operands / fp32 accumulate and tensors (throughput variant)" if bf16 else "fp32")
    result = {
        "metric": metric,
        "value": value,
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": cf["scaling"],
        "vs_baseline": None,
        "dtype": "bf16" if bf16 else ("f32-class (f16x2 split)" if x3 else "f32"),
        "data": "synthetic",
        **({"drift": drift, "parity_configuration": False} if bf16 else {}),
        "config": {"workload": workload, "name": args.config, "tiles_per_gpu": B, "tiles_total": total, "tile": [H, H, C], "T": T, "model_evaluations": n_evals,
                   "sampler": cf["sampler"], "parallelism": "tile-shard x%d" % world, "launches_per_denoising_step": n_launch["step"],
                   "plan_memory_mb": {"total": mem["total_bytes"] / 1e6, "step_activation_arena": mem["arena_bytes"] / 1e6,
                                      "same_activations_unaliased": mem["unaliased_bytes"] / 1e6},
                   "conv_math": ("fp32 operands pre-scaled by powers of two and split into 2 fp16 planes, 3 exact products on v_mfma_f32_32x32x16_f16, fp32 accumulate "
                                 "(3x3 convs, low-resolution levels, q.1 of the fused attention block); 3 bf16 planes / 6 products where the operand range is open "
                                 "(per-sample folded attention weights, wide 1x1 convs); exact fp32 MFMA elsewhere") if x3 else
                                ("THROUGHPUT variant: conv operands rounded once to bf16, one product on v_mfma_f32_32x32x16_bf16, fp32 accumulate; tensors in HBM, GroupNorm "
                                 "statistics, attention blocks, per-sample folded attention weights and the sampler update stay fp32 / split-product") if bf16 else "exact fp32 MFMA"},
        "roofline": hbm_roofline if hbm_roofline else {
            "bound": "mfma",
            "achieved": ach_tflops,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": ach_tflops / peak,
            "peak_note": ("`achieved` counts ALGORITHMIC fp32 flops (2*M*N*K, unpadded) of the dominant class; every fp32 product is issued as %.2f 16-bit MFMA products "
                          "(flop-weighted over the class: 3 = f16x2 two-way split, 6 = bf16x3 three-way split), so the ceiling of this class is the dense 16-bit MFMA "
                          "peak %.1f / %.2f = %.1f TF and `frac` = issued 16-bit TFLOP/s / %.1f" % (products, PEAK_BF16_MFMA_TFLOPS, products, peak, PEAK_BF16_MFMA_TFLOPS))
                          if x3 else "dense fp32 matrix peak (guide); exact fp32 MFMA",
            "mfma_products_per_fp32_product": products,
            "mfma_issued_tflops": products * ach_tflops if x3 else ach_tflops,
            "frac_of_f32_mfma_peak_algorithmic": ach_tflops / PEAK_F32_MFMA_TFLOPS,
            "frac_of_f32_note": "secondary: algorithmic TFLOP/s over the 157.3 TF fp32-matrix peak -- NOT a bound of the split-operand paths (it may exceed 1)",
            "traffic": traffic,
            "traffic_unit": "bytes of HBM traffic per launch of the dominant class (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes of this command)",
            **traffic_info,
            "algorithmic_bytes_per_launch": (prof["total_bytes"] / prof["launches"]) if prof["launches"] else None,
            "kernel": prof["kernel"],
            "launches_timed": prof["launches"],
            "avg_launch_us": (prof["total_ms"] * 1e3 / prof["launches"]) if prof["launches"] else None,
            "algorithmic_gflop_per_launch": (prof["total_flop"] / prof["launches"] / 1e9) if prof["launches"] else None,
            "step_floor_ms": step_floor_ms,
            "step_frac": step_frac,
            "step_frac_note": "step_floor_ms = sum over the step's kernel classes of max(issued 16-bit MFMA flops / %.1f TF, algorithmic bytes / 8 TB/s); step_frac = step_floor_ms / ms_per_denoising_step -- the round-over-round scale" % PEAK_BF16_MFMA_TFLOPS,
            "whole_step": {
                "ms_per_denoising_step": ms_step,
                "tflops": step_flop_total * args.steps / dt / 1e12,
                "frac_of_split_peak": step_flop_total * args.steps / dt / 1e12 / peak,
                "frac_of_f32_mfma_peak_algorithmic": step_flop_total * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "hbm_frac": (cost["step_bytes"] * n_evals + cost["cond_bytes"]) * args.steps / dt / 1e9 / PEAK_HBM_GBS,
                "profiled_steps": n_rec,
                "classes_note": "HIP events around every launch of %d whole denoising steps (one in %d); floor = max(issued 16-bit MFMA flops / %.1f TF, algorithmic bytes / 8 TB/s)" % (n_rec, prof_every, PEAK_BF16_MFMA_TFLOPS),
                "classes_ms_per_step_sum": cls_ms_total if classes_ok else None,
                "classes": classes if classes_ok else None,
                **({} if classes_ok else {"classes_error": "class sum %.3f ms vs %.3f ms per step: table withheld" % (cls_ms_total, ms_step)}),
            },
        },
        "build_id": build_id(),
    }
    if rccl is not None:
        result["rccl"] = rccl
    if rank == 0 and world == 1 and strong and not args.no_shares and total == 64:
        result["projected_strong_scaling"] = strong_scaling_shares(diffusion, net, cf, C, P, H, T, n_evals, order, dev, dt / args.steps * 1e3)
    # the probe and the bracket belong to the full default line only: not to shortened profiling runs (--T), and never under a profiler -- the bracket starts a
    # child process, and a process spawned from under rocprofv3's preloaded tool is exactly the exec the GPU boxes forbid (a --pmc pass hung on it in round 6)
    profiled = any(k in os.environ for k in ("ROCPROFILER_REGISTER_LIBRARY", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    full_line = rank == 0 and world == 1 and args.config == "wv3" and T == cf["T"] and not profiled
    if full_line and not args.no_parity:
        result["parity"] = parity_probe(net, dev)
    if full_line and not args.no_bracket and "DDIF_X3" not in os.environ and "DDIF_F16" not in os.environ:
        result["exact_fp32"] = exact_fp32_bracket(B, H)
        if result["exact_fp32"].get("ms_per_denoising_step"):
            result["ms_per_step_exact_fp32"] = result["exact_fp32"]["ms_per_denoising_step"] * T

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("gpu result: %s" % json.dumps({k: result[k] for k in ("value", "ms_per_step")}))
        if solver is not None:  # configs[2]: the oracle's DPM-Solver++ multistep sampler on the host cores
            cb = cpu_baseline_dpmpp(sd, cfg, tiles["cond"][:8].contiguous(), diffusion.betas.cpu(), n_evals, args.cpu_seconds, args.cpu_threads)
        else:
            cb = cpu_baseline(sd, cfg, tiles["cond"][:8].contiguous(), T, args.cpu_seconds, args.cpu_threads)
        result["cpu_baseline"] = cb
        result["vs_cpu_baseline"] = value / cb["value"]
        result["vs_cpu_baseline_b1"] = value / cb["by_batch"]["1"]["value"]
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def parity_probe(net, dev):
    """The benchmarked build and math mode against the REAL reference's own output (VERDICT r5 #3b): the B = 1 golden of BASELINE configs[1]
    (tests/golden/ddpm_wv3_64_T1000.npz -- the reference's p_sample_loop, T = 1000, 64 x 64 x 8, generated by tools/make_golden.py in the build
    container) re-run through the HIP path with the reference's noise stream, outside the timed region.  Data only: no oracle code runs here."""
    import math

    import numpy as np
    import torch

    import golden_cases as gc
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule
    from ddif_testlib import reference_noise_stream

    cid, ds, B1, H, W, T, seed = [c for c in gc.DDPM_CASES if c[0] == "ddpm_wv3_64_T1000"][0]
    path = os.path.join(gc.GOLDEN_DIR, cid + ".npz")
    if not os.path.exists(path):
        return {"golden": cid, "error": "golden vector not found"}
    ref = torch.from_numpy(np.load(path)["out"])
    tiles = gc.tiles_for(ds, B1, H, W, seed=seed)
    d = GaussianDiffusion(net, image_size=H, channels=8, pred_mode="x_start", loss_type="l1", device=dev, clamp_range=(0, 1))
    d.set_new_noise_schedule(betas=make_beta_schedule("cosine", T, cosine_s=8e-3), device=dev)
    xT, noise = reference_noise_stream(seed, (B1, 8, H, W), T)
    t0 = time.perf_counter()
    out = d(tiles["cond"].to(dev), mode="ddpm_sample", x_T=xT.to(dev), noise=noise.to(dev)).cpu()
    dt = time.perf_counter() - t0
    lms = tiles["cond"][:, :8]

    def psnr(a, b):
        mse = float(torch.mean((a.double() - b.double()) ** 2))
        return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)

    sr_hip, sr_ref = (out + lms).clip(0, 1), (ref + lms).clip(0, 1)  # diffusion_engine.py:446-447
    mode = "exact fp32 MFMA" if os.environ.get("DDIF_X3") == "0" else ("bf16x3 split" if os.environ.get("DDIF_F16") == "0" else "f16x2 split (default)")
    p = {"golden": cid, "against": "the reference's own p_sample_loop output (tools/make_golden.py), B = 1, T = %d, 64x64x8, the reference's noise stream" % T,
         "max_abs": float((out - ref).abs().max()), "psnr_diff_db": abs(psnr(sr_hip, tiles["gt"]) - psnr(sr_ref, tiles["gt"])),
         "tolerance": {"max_abs": 1e-4, "psnr_diff_db": 1e-3}, "mode": mode, "seconds": dt}
    p["within_tolerance"] = bool(p["max_abs"] <= 1e-4 and p["psnr_diff_db"] <= 1e-3)
    log("parity probe (%s): max|hip - reference| = %.3e, PSNR difference %.2e dB after %d steps" % (mode, p["max_abs"], p["psnr_diff_db"], T))
    return p


def exact_fp32_bracket(B, H, T=100):
    """The same workload on the EXACT fp32 instruction (v_mfma_f32_32x32x2_f32 everywhere: DDIF_F16=0 DDIF_X3=0, read by the library at load time, hence a
    child process) -- the bracket the f16x2 headline is quoted between ('the line must carry the bracket').  One job of T denoising
    steps outside the timed region; ms per denoising step does not depend on T."""
    import subprocess

    env = dict(os.environ)
    env.update({"DDIF_F16": "0", "DDIF_X3": "0"})
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--T", str(T), "--batch", str(B), "--tile", str(H), "--no-cpu-baseline", "--no-bracket"]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1]
        j = json.loads(line)
        out = {"ms_per_denoising_step": j["ms_per_step"] / T, "T": T, "launches_per_denoising_step": j["config"]["launches_per_denoising_step"], "dtype": j["dtype"],
               "env": "DDIF_F16=0 DDIF_X3=0", "parity": j.get("parity"), "seconds": time.perf_counter() - t0}
        log("exact-fp32 bracket: %.3f ms per denoising step (%d launches)" % (out["ms_per_denoising_step"], out["launches_per_denoising_step"]))
        return out
    except Exception as e:  # the bracket must never take the headline down with it
        log("exact-fp32 bracket failed: %r" % (e,))
        return {"error": repr(e)[:300]}


def strong_scaling_shares(diffusion, net, cf, C, P, H, T, n_evals, order, dev, ms64):
    """gf2_dpm50 at N = 1: what ONE rank of an N-GPU run of the same scene computes (64 / N tiles), measured on this GPU, and the strong-scaling
    speed-up those shares project (compute share only; the 4 MB all-gather + stitch at N > 1 is not in it).  VERDICT r5 #2."""
    import torch

    from ddif.solver.dpm_solver import DPM_Solver, ImageSpaceClamp, NoiseScheduleVP, model_wrapper
    from ddif.synth import synth_tiles

    shares = {"64": ms64}
    tiles_all = synth_tiles(64, C, P, H, H, seed=100, order=order)
    for n in (32, 16, 8):
        cond = tiles_all["cond"][:n].contiguous().to(dev)
        lms = cond[:, :C].contiguous()
        plan = diffusion._plan(cond)
        ns = NoiseScheduleVP("discrete", betas=diffusion.betas)
        fn = model_wrapper(net, ns, model_type="x_start", guidance_type="classifier-free", guidance_scale=1.0, condition=cond)
        solver = DPM_Solver(fn, ns, algorithm_type="dpmsolver++", correcting_x0_fn=ImageSpaceClamp(lms, 0.0, 1.0))

        def job(seed):
            plan.set_cond(cond, force=True)
            xT = torch.randn((64, C, H, H), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))[:n].contiguous()
            return (solver.sample(xT, steps=n_evals, order=2, skip_type="time_uniform", method="multistep") + lms).clip(0, 1)

        job(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(3):
            job(2 + k)
        torch.cuda.synchronize()
        shares[str(n)] = (time.perf_counter() - t0) / 3 * 1e3
    proj = {str(64 // int(k)): ms64 / v for k, v in shares.items()}
    log("per-rank shares (ms per scene job): %s -> projected speed-up by GPU count %s" % (shares, proj))
    return {"ms_per_job_by_tiles_per_gpu": shares, "projected_speedup_by_gpus": proj,
            "note": "each share = the job of ONE rank holding 64 / N tiles of the scene, timed on this one GPU (3 jobs after 1 warm-up); "
                    "speed-up = share(64) / share(64 / N); the all-gather + stitch of N > 1 is not included"}


def bench_training(args, cf, rank, world, dev):
    """BASELINE configs[4]: one "step" = one training iteration as engine_google runs it (reference diffusion_engine.py:218-241): q_sample +
    self-conditioning draw + forward + loss.backward() through the library's reverse pass + gradient all-reduce over the ranks + fused
    clip / AdamW / EMA.  Weak scaling: every rank trains its own batch; value = tiles per second over all ranks."""
    import random

    import torch
    import torch.distributed as dist

    from ddif import runtime
    from ddif.diffusion.diffusion_ddpm_pan import GaussianDiffusion, make_beta_schedule
    from ddif.diffusion_engine import average_gradients, broadcast_parameters, gradient_bucket
    from ddif.layout import engine_cfg
    from ddif.models.sr3_dwt import UNetSR3
    from ddif.synth import synth_state_dict, synth_tiles

    C, P, B, H, T = cf["C"], cf["P"], args.batch or cf["batch"], args.tile or cf["tile"], args.T or cf["T"]
    cfg = engine_cfg(C, P)
    keys = ("in_channel", "out_channel", "inner_channel", "lms_channel", "pan_channel", "norm_groups", "channel_mults", "attn_res", "res_blocks", "dropout",
            "image_size", "self_condition")
    sd = synth_state_dict(cfg, 1234)
    net = UNetSR3(**{k: cfg[k] for k in keys})
    net.load_state_dict(sd)
    net = net.to(dev).train()
    d = GaussianDiffusion(net, image_size=H, channels=C, pred_mode="x_start", loss_type="l1", device=dev, clamp_range=(0, 1))
    d.set_new_noise_schedule(betas=make_beta_schedule("cosine", T, cosine_s=8e-3), device=dev)
    tiles = synth_tiles(B, C, P, H, H, seed=100 + rank)
    cond = tiles["cond"].to(dev)
    res = (tiles["gt"].to(dev) - cond[:, :C]).contiguous()
    params = [p for p in net.parameters()]
    _, grads = gradient_bucket(params)
    for p, g in zip(params, grads):
        p.grad = g
    ema = [p.detach().clone() for p in params]
    if dist.is_initialized():
        broadcast_parameters(params + ema)
    opt = runtime.FusedAdamW(params, grads, ema, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    torch.manual_seed(7 + rank)
    random.seed(7 + rank)
    n_param = sum(p.numel() for p in params)
    comm_ms = [0.0]
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] if dist.is_initialized() else None
    sc_passes = [0]
    qsf = runtime.PlanHandle.q_sample_forward

    def counted(self, *a, **k):  # the no-grad self-conditioning forward of half of the iterations (reference :703-709)
        sc_passes[0] += 1
        return qsf(self, *a, **k)

    runtime.PlanHandle.q_sample_forward = counted

    def one_step(timed=False):
        loss, _ = d.train_step_into(res, cond, grads)  # q_sample + self-conditioning draw + forward + backward: gradients written into `grads`
        if dist.is_initialized():
            if timed:
                ev[0].record()
            average_gradients(grads, world)
            if timed:
                ev[1].record()
                ev[1].synchronize()
                comm_ms[0] += ev[0].elapsed_time(ev[1])
        opt.step(max_grad_norm=0.003, ema_mode=1, ema_decay=0.995)
        net.mark_weights_dirty()
        return loss

    for w in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    sc_passes[0] = 0
    t0 = time.perf_counter()
    loss = None
    for k in range(args.steps):
        loss = one_step(timed=True)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(loss.detach()).all())
    # algorithmic flops of one iteration: forward + backward = 3 x the un-hoisted forward (dgrad + wgrad = 2 x forward), plus one
    # forward per self-conditioning pass actually run
    cost = d._plan(cond, train=True).cost()
    fwd_flop = cost["step_flop"] + cost["cond_flop"]
    step_flop = fwd_flop * (3.0 + sc_passes[0] / max(1, args.steps))
    tf = step_flop * args.steps / dt / 1e12
    if rank == 0:
        line = {"metric": cf["metric"] % T, "value": world * B * args.steps / dt, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "sr3_dwt training iteration (q_sample, 50 %% self-conditioning pass, forward, backward, DDP all-reduce, clip + AdamW + EMA), "
                                       "WV3 %dx%dx%d tiles, batch %d per GPU, fp32" % (H, H, C, B), "name": args.config, "tiles_per_gpu": B, "tile": [H, H, C],
                           "T": T, "parallelism": "ddp x%d" % world, "self_conditioning_passes_in_timed_region": sc_passes[0]},
                "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                             "algorithmic_gflop_per_iteration": step_flop / 1e9,
                             "peak_note": "whole training iteration: algorithmic flops (3 x un-hoisted forward + self-conditioning forwards) / wall time vs the dense fp32 "
                                          "matrix peak; forward / dgrad convs and the 3x3 weight gradients (conv3x3_wgrad_x3_kernel, DDIF_WGRAD_X3=1) run bf16x3 split products, "
                                          "the exact fp32 MFMA weight-gradient kernel remains the fallback for widths that are not a multiple of 8 or exceed 128",
                             "path": "native reverse launch program (csrc/ddif_train.cpp, ddif_plan_train_step): NHWC end to end, device-side weight refresh"},
                "build_id": build_id()}
        if dist.is_initialized():
            line["allreduce"] = {"bytes": 4 * n_param, "ms_per_iteration": comm_ms[0] / args.steps,
                                 "algorithmic_gbytes_per_s": 4 * n_param / (comm_ms[0] / args.steps * 1e-3) / 1e9 if comm_ms[0] > 0 else None,
                                 "bus_gbytes_per_s": 2.0 * (world - 1) / world * 4 * n_param / (comm_ms[0] / args.steps * 1e-3) / 1e9 if comm_ms[0] > 0 else None,
                                 "note": "one flat fp32 bucket, RCCL ring all-reduce (AVG); timed with HIP events around the collective, rank 0"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_training(sd, cfg, tiles, T, args.cpu_seconds, args.cpu_threads)
            line["vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def cpu_baseline_training(sd, cfg, tiles, T, budget_s, threads=0):
    """The oracle's forward (oracle/ddif_oracle.unet_forward, pinned to the reference's training golden) + torch autograd + torch.optim.AdamW on
    the host cores: whole training iterations (q_sample, forward, L1, backward, clip, AdamW) on a batch of 4 tiles, as many as fit the budget."""
    import torch

    from oracle import ddif_oracle as O

    if threads <= 0:
        threads = usable_cpus()
    torch.set_num_threads(threads)
    bsz = 4
    cond = tiles["cond"][:bsz].contiguous()
    x0 = (tiles["gt"][:bsz] - cond[:, : cfg["lms_channel"]]).contiguous()
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    tabs = O.schedule_tables(O.cosine_betas(T))
    g = torch.Generator().manual_seed(3)

    def it():
        t = torch.randint(0, T, (bsz,), generator=g)
        noise = torch.randn(x0.shape, generator=g)
        xt = O.q_sample(tabs, x0, t, noise)
        opt.zero_grad(set_to_none=True)
        pred = O.unet_forward(P, cfg, xt, t, cond, None)
        loss = (x0 - pred).abs().mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(P.values()), 0.003)
        opt.step()

    t0 = time.perf_counter()
    it()  # warm-up (oneDNN primitive creation)
    probe = time.perf_counter() - t0
    n = int(max(1, min(50, (budget_s - probe) / max(probe, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        it()
    dt = time.perf_counter() - t0
    log("cpu baseline (training): %.3f s per iteration of %d tiles over %d iterations, %d threads" % (dt / n, bsz, n, threads))
    return {"value": bsz * n / dt, "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d training iterations (q_sample, oracle forward, L1, autograd backward, clip 0.003, AdamW) on a batch of %d WV3 64x64 tiles, %.1f s; no "
                      "self-conditioning pass, Dropout / DropPath off (both add work on the reference's side)" % (n, bsz, dt),
            "host_cpus": os.cpu_count()}


def committed_traffic():
    """HBM bytes per launch of the dominant kernel class.  PMC counters need their own rocprofv3 passes (they cannot be
    collected from inside this process), so this reads the newest summary committed under profiles/ (tools/gpu_full.sh +
    tools/pmc_traffic.py, same `bench.py` command) -- and ONLY reports it when that profile was taken from the build being
    benchmarked (same sha1 over csrc/); otherwise `traffic` is null and the stale file is named."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*hbm_traffic.json")))
    if not files:
        return None, {"traffic_from_committed_profile": False}
    bid = build_id()
    for path in reversed(files):  # the set taken from THIS build, whatever its tag sorts like
        try:
            with open(path) as f:
                j = json.load(f)
            if j.get("build_id") == bid:
                return float(j["class_hbm_bytes_per_launch"]), {"traffic_from_committed_profile": True, "traffic_source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return None, {"traffic_from_committed_profile": False, "traffic_stale_profile": os.path.relpath(files[-1], ROOT)}


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


def cpu_baseline_dpmpp(sd, cfg, cond8, betas, nfe, budget_s, threads=0):
    """The oracle's DPM-Solver++ 2M sampler (oracle/ddif_oracle.dpmpp_multistep_sample, pinned to the reference's goldens) on the host cores: whole jobs of a
    REDUCED number of evaluations at B = 1 and B = 8, scaled to `nfe` (the cost of the solver is its model evaluations)."""
    import torch

    from oracle import ddif_oracle as O

    if threads <= 0:
        threads = usable_cpus()
    torch.set_num_threads(threads)
    C, H = cfg["out_channel"], cond8.shape[2]
    by = {}
    for bsz, share in ((1, 0.4), (8, 0.6)):
        cond = cond8[:bsz].contiguous()
        xT = torch.randn(bsz, C, H, H, generator=torch.Generator().manual_seed(1))

        def run(n):
            t0 = time.perf_counter()
            with torch.no_grad():
                O.dpmpp_multistep_sample(sd, cfg, cond, betas, xT, steps=n, order=2)
            return time.perf_counter() - t0

        run(2)  # warm-up (oneDNN primitive creation)
        t_probe = run(3) / 3
        n = int(max(3, min(nfe, share * budget_s / max(t_probe, 1e-4))))
        dt = run(n)
        per_eval = dt / n
        mp = cond.shape[0] * H * H / 1e6
        by[str(bsz)] = {"value": mp / (per_eval * nfe), "seconds_per_evaluation": per_eval, "evaluations_timed": n, "seconds_timed": dt}
        log("cpu baseline (DPM-Solver++) B=%d: %.3f s per evaluation over %d -> %.3e MP/s" % (bsz, per_eval, n, by[str(bsz)]["value"]))
    best = max(by, key=lambda k: by[k]["value"])
    return {"value": by[best]["value"], "unit": "MP/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/ddif_oracle.dpmpp_multistep_sample (2M) on tiles %dx%dx%d: %d evaluations at B=1 (%.1f s) and %d at B=8 (%.1f s), each scaled to %d NFE; value = B=%s (the better)"
                      % (H, H, C, by["1"]["evaluations_timed"], by["1"]["seconds_timed"], by["8"]["evaluations_timed"], by["8"]["seconds_timed"], nfe, best),
            "best_batch": int(best), "by_batch": by, "host_cpus": os.cpu_count()}


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
