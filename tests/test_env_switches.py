"""The build-time-independent switches libddif reads from the environment, each run against the reference goldens
(-m gpu).  The library reads them once per process, so every case runs a slice of the parity suite in a child process:

  DDIF_X3=0     every conv on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) instead of the bf16x3
                split products -- INTEGRATION.md advertises it as the "bitwise-fmaf build";
  DDIF_GRAPH=0  the sampler loop as plain stream launches instead of hipGraph replay of step pairs;
  DDIF_LR=0     the 8x8 / 16x16 levels on the general conv kernel (kernels_conv.h) instead of the low-resolution
                split-K kernel (kernels_lr.h) -- the golden cases at 16x16 / 32x32 otherwise run almost entirely on the
                latter, so this is what keeps the general kernel's small-tile instantiations covered.

  DDIF_F16=0    the split-operand convs on bf16x3 (six products) instead of f16x2 (three);
  DDIF_LAFUSE=0 the decoder's linear-attention half as three launches instead of the fused block (csrc/kernels_lafuse.h).

All other A/B switches (round 1: wave-specialised conv, VALU attention, unfused depthwise, tile-shape overrides; round 5: the fused feed-forward
kernel DDIF_FFNFUSE and the forked low-resolution region DDIF_SPLIT, both measured slower -- profiles/r04_t_*, r03_b_*) were deleted together with
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


@pytest.mark.parametrize("env", [{"DDIF_X3": "0"}, {"DDIF_GRAPH": "0"}, {"DDIF_X3": "0", "DDIF_GRAPH": "0"}, {"DDIF_LR": "0"}, {"DDIF_F16": "0"}, {"DDIF_LAFUSE": "0"}],
                         ids=["X3=0", "GRAPH=0", "X3=0+GRAPH=0", "LR=0", "F16=0", "LAFUSE=0"])
def test_parity_slice_under_switch(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-k", SLICE, "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=1500)
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
