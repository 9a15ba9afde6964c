"""RCCL at world size 1 on the 1-GPU lease: no multi-GPU node has been available to this project, so the first 8-GPU run would
also have been the first `nccl` initialisation of this code.  Here the process group is RCCL with ONE rank and every collective of the multi-GPU
path runs for real (over one rank): the tile all-gather of `sample_sharded` (reference diffusion_engine.py:373-377 feeds whole scenes),
the flat gradient-bucket all-reduce of the DDP step (41.6 MB for the engine network), and `bench.py --gpus 1` under `torch.distributed.run`
(launched before any GPU call, as the driver does for N > 1).  Everything runs in child processes: the pytest process never joins a group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path[:0] = [%(pkg)r, %(root)r, %(tests)r]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import golden_cases as gc
from ddif_testlib import make_diffusion, make_net, use_gpu_library
use_gpu_library()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from ddif.sharding import sample_sharded, sample_scene_dpmpp
import ddif.diffusion_engine as E
ds, H, T = "gf2", 16, 4
C = gc.DATASETS[ds][0]
cond = gc.tiles_for(ds, 2, H, H, seed=21)["cond"].to(dev)
net = make_net(ds, dev)
d = make_diffusion(net, C, T, H, dev)
alone = sample_sharded(d, cond, mode="ddpm_sample", seed=5)            # no process group: no collective
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
gathered = sample_sharded(d, cond, mode="ddpm_sample", seed=5)         # all_gather_into_tensor over RCCL
assert gathered.shape == alone.shape and torch.equal(gathered, alone)
xT = torch.randn(2, C, H, H, generator=torch.Generator().manual_seed(3)).to(dev)
scene = sample_scene_dpmpp(net, make_diffusion(net, C, 1000, H, dev), cond, xT, steps=3, order=2, grid=(1, 2))
assert scene.shape == (C, H, 2 * H) and bool(torch.isfinite(scene).all())
# the DDP step's flat bucket at the engine network's size (10.4 M parameters = 41.6 MB): all-reduce (AVG) in place, pointers kept
shapes = [(2600000, 4), (1, 128), (7,)]
params = [torch.empty(s, device=dev) for s in shapes]
flat, views = E.gradient_bucket(params)
assert flat.numel() * 4 > 41.5e6
for i, v in enumerate(views):
    v.fill_(float(i + 1))
ptrs = [v.data_ptr() for v in views]
E.average_gradients(views, 1)
torch.cuda.synchronize()
assert [v.data_ptr() for v in views] == ptrs
assert all(bool((v == float(i + 1)).all()) for i, v in enumerate(views))
E.broadcast_parameters([torch.ones(5, device=dev), torch.zeros(3, 3, device=dev)])
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_rccl_world_size_one_runs_the_collectives_of_the_multi_gpu_path():
    code = CHILD % {"pkg": os.path.join(ROOT, "dif-pan_amd"), "root": ROOT, "tests": os.path.join(ROOT, "tests")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def _launch_bench(extra, port):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_the_launcher_at_one_gpu_reports_the_multi_gpu_fields():
    """The driver's N > 1 command line with N = 1: RCCL process group, tile all-gather inside the timed region, max-over-ranks all-reduce of the time."""
    line = _launch_bench(["--steps", "1", "--warmup", "0", "--T", "20"], 29742)
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0 and line["config"]["parallelism"] == "tile-shard x1"
    line = _launch_bench(["--config", "gf2_dpm50", "--batch", "8", "--steps", "1", "--warmup", "0"], 29743)
    assert line["n_gpus"] == 1 and line["scaling"] == "strong"
    line = _launch_bench(["--config", "wv3_train_b32", "--batch", "4", "--steps", "2", "--warmup", "1"], 29744)
    ar = line["allreduce"]
    assert line["n_gpus"] == 1 and ar["bytes"] > 40e6 and ar["ms_per_iteration"] > 0 and ar["algorithmic_gbytes_per_s"] > 0 and "bus_gbytes_per_s" in ar
