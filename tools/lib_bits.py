"""Development aid: run a fixed 64 x 64 forward (B = 3, three time values) and a 4-step DDPM chain of three tiles with a given build of libddif.so and save the outputs;
with two saved files, report whether they are bit-identical.  usage: lib_bits.py run <lib.so|-> <out.pt>   |   lib_bits.py cmp <a.pt> <b.pt>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "dif-pan_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch


def run(lib, out):
    import golden_cases as gc
    from ddif import runtime
    from ddif_testlib import make_diffusion, make_net, use_gpu_library

    if lib != "-":
        runtime.use_library(os.path.abspath(lib))
    else:
        use_gpu_library()
    dev = torch.device("cuda:0")
    res = {}
    for ds, B, H in (("wv3", 3, 64), ("gf2", 2, 32)):
        C = gc.DATASETS[ds][0]
        g = torch.Generator().manual_seed(77)
        x = torch.randn(B, C, H, H, generator=g).to(dev)
        t = torch.tensor([900, 12, 433][:B]).to(dev)
        cond = gc.tiles_for(ds, B, H, H, seed=78)["cond"].to(dev)
        net = make_net(ds, dev)
        res["y_%s" % ds] = net(x, t, cond).cpu()
        d = make_diffusion(net, C, 4, H, dev)
        res["out_%s" % ds] = d(cond, mode="ddpm_sample", seed=3, tile0=0, device_rng=True).cpu()
    torch.save(res, out)


if sys.argv[1] == "run":
    run(sys.argv[2], sys.argv[3])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        print(k, "bit-identical" if torch.equal(a[k], b[k]) else "DIFFERENT: max |d| = %.3e" % float((a[k] - b[k]).abs().max()))
