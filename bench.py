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

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "dif-pan_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

_T0 = time.time()


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix (= vector) peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 16 * 157.3  # dense bf16 MFMA (2.5 PF); the bf16x3 path issues 6 bf16 products per fp32 product
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU (BASELINE config: 64)")
    ap.add_argument("--T", type=int, default=1000, help="diffusion steps (BASELINE config: 1000)")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
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

    lib = ddif.get_lib()
    assert not lib.emulated
    C, P, B, H, T = 8, 1, args.batch, args.tile, args.T
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
    tiles = synth_tiles(B, C, P, H, H, seed=100 + rank)
    cond = tiles["cond"].to(dev)
    lms = cond[:, :C].contiguous()
    gathered = torch.empty((world * B, C, H, H), device=dev) if world > 1 else None
    plan = diffusion._plan(cond)  # builds workspaces + runs set_cond once (not timed)
    cost = plan.cost()
    torch.cuda.synchronize()
    log("plan ready: %.3f GFLOP and %.1f MB (algorithmic) per denoising step of the batch" % (cost["step_flop"] / 1e9, cost["step_bytes"] / 1e6))

    def one_step(seed):
        plan.set_cond(cond, force=True)  # once-per-tile precompute is part of the job
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
    plan.prof_begin(max(1, T // 20), 16384)
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
    step_flop_total = cost["step_flop"] * T + cost["cond_flop"]
    traffic, traffic_src = committed_traffic()
    x3 = "bf16x3" in prof["kernel"]
    peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if x3 else PEAK_F32_MFMA_TFLOPS
    result = {
        "metric": "fused megapixels/sec at T=%d, WV3 64x64x8 tiles" % T,
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
        "config": {"workload": "WV3 pansharpening, batch %d of %dx%dx8 tiles per GPU, T=%d DDPM p_sample, fp32" % (B, H, H, T),
                   "tiles_per_gpu": B, "tile": [H, H, C], "T": T, "sampler": "ddpm", "parallelism": "tile-shard x%d" % world,
                   "conv_math": "bf16x3 split products (3x3 and 32-channel-chunk 1x1 convs) + exact fp32 MFMA (remaining convs, attention)" if x3 else "exact fp32 MFMA"},
        "roofline": {
            "bound": "mfma",
            "achieved": ach_tflops,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": ach_tflops / peak,
            "peak_note": ("fp32-equivalent: dense bf16 MFMA peak 2516.8 TF / 6 split products per fp32 product (bf16x3: hi/mid/lo "
                          "operand split, fp32 accumulate, fp32-class accuracy); achieved counts ALGORITHMIC fp32 flops"
                          if x3 else "dense fp32 MFMA (v_mfma_f32_32x32x2_f32)"),
            "frac_of_f32_mfma_peak": ach_tflops / PEAK_F32_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes of HBM traffic per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes)",
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": (prof["total_bytes"] / prof["launches"]) if prof["launches"] else None,
            "kernel": prof["kernel"],
            "launches_timed": prof["launches"],
            "avg_launch_us": (prof["total_ms"] * 1e3 / prof["launches"]) if prof["launches"] else None,
            "algorithmic_gflop_per_launch": (prof["total_flop"] / prof["launches"] / 1e9) if prof["launches"] else None,
            "whole_job_tflops": step_flop_total * args.steps / dt / 1e12,
            "whole_job_frac_of_f32_peak": step_flop_total * args.steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "whole_job_hbm_frac": (cost["step_bytes"] * T + cost["cond_bytes"]) * args.steps / dt / 1e9 / PEAK_HBM_GBS,
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("gpu result: %s" % json.dumps({k: result[k] for k in ("value", "ms_per_step")}))
        result["cpu_baseline"] = cpu_baseline(sd, cfg, tiles["cond"][:1].contiguous(), T, args.cpu_seconds, args.cpu_threads)
        result["vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def committed_traffic():
    """HBM bytes per launch of the dominant kernel class.  PMC counters need their own rocprofv3 passes (they cannot
    be collected from inside this process), so this reads the newest summary committed under profiles/ -- produced
    from the same `bench.py` command by tools/gpu_full.sh + tools/pmc_traffic.py -- and reports null when none exists."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return float(json.load(f)["class_hbm_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1])
    except (OSError, ValueError, KeyError):
        return None, None


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


def cpu_baseline(sd, cfg, cond1, T, budget_s, threads=0):
    """The oracle (CPU port of the reference DDPM sampler: same torch-CPU op sequence, no hoisting, no fusion) on the
    host cores of this box: B=1 tile (BASELINE config 1), the first n of the T steps, scaled to T."""
    import torch

    from oracle import ddif_oracle as O

    if threads <= 0:
        threads = usable_cpus()
    torch.set_num_threads(threads)
    log("cpu baseline: %d threads (os.cpu_count() = %s)" % (threads, os.cpu_count()))
    tabs = O.schedule_tables(O.cosine_betas(T))
    g = torch.Generator().manual_seed(1)
    shape = tuple(cond1.shape[:1]) + (cfg["out_channel"],) + tuple(cond1.shape[2:])

    def run(n):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.ddpm_sample(sd, cfg, cond1, tabs, noise_fn=lambda s: torch.randn(s, generator=g), max_steps=n)
        return time.perf_counter() - t0

    tw = run(1)  # warm-up (oneDNN primitive creation)
    log("cpu baseline warm-up step: %.2f s" % tw)
    t_probe = run(3) / 3
    log("cpu baseline probe: %.3f s/step" % t_probe)
    n = int(max(3, min(T, budget_s / max(t_probe, 1e-4))))
    dt = run(n)
    per_step = dt / n
    mp = cond1.shape[0] * cond1.shape[2] * cond1.shape[3] / 1e6
    return {
        "value": mp / (per_step * T),
        "unit": "MP/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": "oracle/ddif_oracle.ddpm_sample, B=1 tile 64x64x8, first %d of T=%d steps timed (%.2f s), scaled to T" % (n, T, dt),
        "seconds_per_step": per_step,
        "host_cpus": os.cpu_count(),
    }


if __name__ == "__main__":
    main()
