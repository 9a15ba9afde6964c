"""One training step (forward + backward of the whole denoiser, tests/train_tape.py TrainGraph) of the config-5 shape on the GPU: wall time per
phase; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.   python3 tools/train_step_bench.py [batch] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dif-pan_amd"), os.path.join(ROOT, "tests"), ROOT]
import torch  # noqa: E402

from ddif import runtime  # noqa: E402
from ddif.layout import engine_cfg  # noqa: E402
from ddif.synth import synth_state_dict, synth_tiles  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from train_tape import TrainGraph  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
cfg = engine_cfg(8, 1)
P = {k: v.to(dev).contiguous() for k, v in synth_state_dict(cfg, 1).items() if v.dtype == torch.float32}
t = synth_tiles(B, 8, 1, 64, 64, seed=3)
x = torch.randn(B, 8, 64, 64, device=dev)
cond = t["cond"].to(dev)
tt = torch.randint(0, 1000, (B,), device=dev)
g = TrainGraph(cfg)
for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = g.forward(P, x, tt, cond, None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    G = g.backward(runtime.l1_loss_backward(y, x))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iteration {it}: forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms, batch {B}, {len(G)} gradients, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB", flush=True)
