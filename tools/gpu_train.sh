#!/bin/bash
# usage: tools/gpu_train.sh <tag>  -- training tests on the GPU, then the time of one training step (config 5 shape: WV3 64x64, batch 32)
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout 1500 python -m pytest tests/test_train_graph.py tests/test_engine.py -m gpu -q -x 2>&1 | tail -6) > $R/gpurun_out/${tag}_train_tests.log 2>&1
cat $R/gpurun_out/${tag}_train_tests.log
python3 - <<'PY' 2>&1 | tee $R/gpurun_out/${tag}_train_step_timing.txt
import sys, time, os
sys.path[:0] = ["dif-pan_amd", "tests", "."]
import torch
from ddif.layout import engine_cfg
from ddif.synth import synth_state_dict, synth_tiles
from ddif.train import TrainGraph
from ddif import runtime
dev = torch.device("cuda:0")
cfg = engine_cfg(8, 1)
P = {k: v.to(dev).contiguous() for k, v in synth_state_dict(cfg, 1).items() if v.dtype == torch.float32}
for B in (8, 32):
    t = synth_tiles(B, 8, 1, 64, 64, seed=3)
    x = torch.randn(B, 8, 64, 64, device=dev); cond = t["cond"].to(dev); tt = torch.randint(0, 1000, (B,), device=dev)
    g = TrainGraph(cfg)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = g.forward(P, x, tt, cond, None)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dy = runtime.l1_loss_backward(y, x)
        G = g.backward(dy)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"training step WV3 64x64 batch {B}: forward {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, {len(G)} gradients, peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
PY
