"""The build-time-independent switches libddif reads from the environment, each run against the reference goldens
(-m gpu).  The library reads them once per process, so every case runs a slice of the parity suite in a child process:

  DDIF_X3=0     every conv on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) instead of the bf16x3
                split products -- INTEGRATION.md advertises it as the "bitwise-fmaf build";
  DDIF_GRAPH=0  the sampler loop as plain stream launches instead of hipGraph replay of step pairs;
  DDIF_LR=0     the 8x8 / 16x16 levels on the general conv kernel (kernels_conv.h) instead of the low-resolution
                split-K kernel (kernels_lr.h) -- the golden cases at 16x16 / 32x32 otherwise run almost entirely on the
                latter, so this is what keeps the general kernel's small-tile instantiations covered.

  DDIF_F16=0    the split-operand convs on bf16x3 (six products) instead of f16x2 (three);
  DDIF_LAFUSE=0 the decoder's linear-attention half as three launches instead of the fused block (csrc/kernels_lafuse.h);
  DDIF_S2_F16=0 the Downsample convs and the stem on the exact-fp32 tilings instead of the f16x2 ones (round 5);
  DDIF_WRES=0 / DDIF_XCD=0  the 16-channel-stage tiling of the 32 -> 32 convs / the dispatcher's round-robin work partition instead of the resident-weights
                tiling / the XCD-contiguous partition (round 5): bit-identical results either way.

  DDIF_ATTN_NW=8  the fused bottleneck attention block on eight wavefronts per sample instead of four (round 6: measured neutral, not the default): same values.

  DDIF_XF=0     CondInjection.x_conv + FiLM as a 1x1 launch of its own everywhere instead of riding in the producer conv's epilogue (round 6, EPI_XF).

  DDIF_LA6=0    the 192-channel decoder block of the 16 x 16 level as three launches instead of linattn_fused (round 6).
  DDIF_LA8=0    the decoder's linear-attention half at the 8 x 8 level as three launches instead of the fused kernel of round 6 (csrc/kernels_lafuse8.h).

  DDIF_LA_NW=4 / 8  the fused linear-attention block on four-wave workgroups of 128 pixels / eight-wave ones of 256 everywhere (round 6; by default the plan
                picks four where eight would leave CUs idle): same results either way.

  DDIF_ATTN_SPLIT=1 / 2  the fused bottleneck attention block on ONE / TWO workgroups per sample instead of four (round 6: the query tokens of a sample on up to four CUs).

  DDIF_TILE16=1 / 0  the 3x3 convs of the 64 x 64 / 32 x 32 levels on 16 x 16-pixel tiles (eight waves) / 8 x 16 tiles (four waves) everywhere; by default the plan picks
                the small tiles where the big ones would not fill the CUs (round 6).  Bit-identical results: the big tiles write half-tile statistics partials.

  DDIF_ATTN_F16=0  the qkv conv of the bottleneck attention block on bf16x3 (six products) instead of f16x2 (three; round 6).

  DDIF_LR_ROWS=0  the low-resolution 3x3 convs on the general pixel-item staging where the plan would take the row staging (kernels_lr.h ROWS, round 6): the
                same LDS image and contraction, so bit-identical results (checked below together with the tilings).

All other A/B switches (round 1: wave-specialised conv, VALU attention, unfused depthwise, tile-shape overrides; round 5: the fused feed-forward
kernel DDIF_FFNFUSE and the forked low-resolution region DDIF_SPLIT, both measured slower -- profiles/r04/t_*, r03_b_*) were deleted together with
their code."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SLICE = ("test_forward_matches_reference_golden or test_ddpm_matches_reference_golden and ddpm_wv3_16_T10 "
         "or test_ddim_matches_reference_golden and ddim_gf2 or test_dpm_solver_matches_reference_golden and s10_o2 "
         "or test_forward_matches_oracle_other_sizes and 8x8")


@pytest.mark.parametrize("env", [{"DDIF_X3": "0"}, {"DDIF_GRAPH": "0"}, {"DDIF_X3": "0", "DDIF_GRAPH": "0"}, {"DDIF_LR": "0"}, {"DDIF_F16": "0"}, {"DDIF_LAFUSE": "0"},
                                 {"DDIF_S2_F16": "0"}, {"DDIF_XF": "0"}, {"DDIF_LA8": "0"}],
                         ids=["X3=0", "GRAPH=0", "X3=0+GRAPH=0", "LR=0", "F16=0", "LAFUSE=0", "S2_F16=0", "XF=0", "LA8=0"])
def test_parity_slice_under_switch(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", SLICE, "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail


# the round-6 switches that only change WHICH workgroup / tiling computes a value (results identical): the forward goldens, incl. the 64 x 64 tile that reaches
# every kernel they touch, are enough -- and keep the suite under ten minutes
SLICE_FWD = "test_forward_matches_reference_golden"


# (independent switches share a child process: a child costs ~10 s of interpreter + library start-up whatever it runs)
@pytest.mark.parametrize("env", [{"DDIF_LA_NW": "4", "DDIF_ATTN_SPLIT": "1", "DDIF_TILE16": "1"}, {"DDIF_LA_NW": "8", "DDIF_ATTN_SPLIT": "2", "DDIF_TILE16": "0"},
                                 {"DDIF_ATTN_NW": "8", "DDIF_LA6": "0"}, {"DDIF_ATTN_F16": "0", "DDIF_LR_ROWS": "0"}],
                         ids=["LA_NW=4+ATTN_SPLIT=1+TILE16=1", "LA_NW=8+ATTN_SPLIT=2+TILE16=0", "ATTN_NW=8+LA6=0", "ATTN_F16=0+LR_ROWS=0"])
def test_forward_goldens_under_placement_switch(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", SLICE_FWD, "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail


@pytest.mark.parametrize("env", [{"DDIF_X3": "0"}, {"DDIF_LR": "0"}], ids=["X3=0", "LR=0"])
def test_multi_item_path_under_switch(env):
    """The capped-grid tests once more with DDIF_X3=0 / DDIF_LR=0 (other kernel instantiations, other tile shapes)."""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_batch64.py"), "-m", "gpu", "-x", "-q",
                        "-k", "capped_grid", "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail


def test_training_gradients_on_the_exact_fp32_convs():
    """DDIF_TRAIN_X3=0: the training convs (forward and dgrad) on the exact-fp32 MFMA instead of the bf16x3 split products; the gradient
    parity test against the reference must hold on that path too."""
    e = dict(os.environ)
    e["DDIF_TRAIN_X3"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_train_graph.py"), "-m", "gpu", "-x", "-q",
                        "-k", "gradients_match_the_reference", "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail


def test_resident_weights_tiling_and_xcd_partition_are_bit_identical_to_what_they_replace(tmp_path):
    """DDIF_WRES=0 keeps tiling 27 (16-channel stages, weights re-staged per stage) where the default takes the resident-weights tiling 37 (kernels_conv.h
    MATH = 5: one 32-channel stage per work item, weights in LDS for the whole launch) -- same pack, same accumulation order, same GroupNorm partials:
    a 64 x 64 forward and a 4-step DDPM chain of two tiles must agree BIT FOR BIT between the two, and the default plan must really contain the tiling."""
    code = r"""
import sys
sys.path[:0] = [%r, %r, %r]
import torch
import golden_cases as gc
from ddif_testlib import make_diffusion, make_net, use_gpu_library
use_gpu_library()
dev = torch.device("cuda:0")
ds, B, H = "wv3", 2, 64
C = gc.DATASETS[ds][0]
g = torch.Generator().manual_seed(77)
x = torch.randn(B, C, H, H, generator=g).to(dev)
t = torch.tensor([900, 12]).to(dev)
cond = gc.tiles_for(ds, B, H, H, seed=78)["cond"].to(dev)
net = make_net(ds, dev)
y = net(x, t, cond)
d = make_diffusion(net, C, 4, H, dev)
out = d(cond, mode="ddpm_sample", seed=3, tile0=0, device_rng=True)
torch.save({"y": y.cpu(), "out": out.cpu()}, sys.argv[1])
"""
    import torch

    res = {}
    for flag in ("0", "1", "small", "norows"):
        e = dict(os.environ)
        e["DDIF_WRES"] = "0" if flag == "0" else "1"
        e["DDIF_XCD"] = "0" if flag == "0" else "15"  # ... and the XCD-contiguous work partition (ddif_dev.h wg_work_range): which workgroup computes an item, never what
        # two tiles would not fill the CUs: since round 6 the plan would pick the four-wave 8 x 16 tiling by itself ("small" lets it); DDIF_TILE16=1 keeps the
        # 16 x 16 tilings this test is about.  All three must agree bit for bit: the big tilings write half-tile statistics partials (ConvArgs::st_halves)
        e["DDIF_TILE16"] = "0" if flag == "small" else "1"
        if flag == "norows":  # ... and the general staging of the low-resolution 3x3 convs instead of the row staging (kernels_lr.h ROWS): the same LDS image
            e["DDIF_LR_ROWS"] = "0"
        e["DDIF_DUMP_PLAN"] = "1"
        f = str(tmp_path / ("wres%s.pt" % flag))
        r = subprocess.run([sys.executable, "-c", code % (os.path.join(ROOT, "dif-pan_amd"), ROOT, os.path.join(ROOT, "tests")), f], env=e, cwd=ROOT, capture_output=True,
                           text=True, timeout=1500)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        res[flag] = (torch.load(f), r.stderr.count("cfg=37"), r.stderr.count("cfg=28"), r.stderr.count("_rows "))
    assert res["0"][1] == 0 and res["1"][1] >= 15, (res["0"][1], res["1"][1])  # 14 ResnetBlock convs + the final conv at the 64 x 64 level
    assert res["small"][1] == 0 and res["small"][2] >= 30, res["small"][1:]  # every 3x3 conv of the 64 x 64 / 32 x 32 levels on the 8 x 16 tiling
    assert res["1"][3] >= 40 and res["norows"][3] == 0, (res["1"][3], res["norows"][3])  # the 3x3 convs of the 16 x 16 / 8 x 8 levels stage by rows
    assert torch.equal(res["norows"][0]["y"], res["1"][0]["y"])
    assert torch.equal(res["norows"][0]["out"], res["1"][0]["out"])
    assert torch.equal(res["0"][0]["y"], res["1"][0]["y"])
    assert torch.equal(res["0"][0]["out"], res["1"][0]["out"])
    assert torch.equal(res["small"][0]["y"], res["1"][0]["y"])
    assert torch.equal(res["small"][0]["out"], res["1"][0]["out"])
    assert bool(torch.isfinite(res["1"][0]["out"]).all())
