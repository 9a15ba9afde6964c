"""The gfx950 library must build without a GPU, load, and export every entry point include/*.h declares -- include/ddif.h, the product surface, and
include/ddif_testops.h, the test-only per-op entry points split off it in round 6 (no compute calls here: that is what the -m gpu tests do)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dif-pan_amd")


def _declared(header="ddif.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"DDIF_API\s+[\w\s\*]+?\b(ddif_\w+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    names = _declared()
    for must in ("ddif_net_create", "ddif_net_load", "ddif_net_commit", "ddif_plan_create", "ddif_plan_set_cond",
                 "ddif_plan_forward", "ddif_plan_sample_ddpm", "ddif_plan_sample_ddim", "ddif_plan_sample_dpmpp",
                 "ddif_plan_q_sample_forward", "ddif_last_error"):
        assert must in names
    # the product surface carries no per-op training-tape entry point any more (round 6): those live in include/ddif_testops.h
    assert not [n for n in names if n.startswith(("ddif_convfwd_", "ddif_convbwd_", "ddif_blockbwd_")) or n.endswith(("_fwd", "_bwd"))]
    assert "ddif_convbwd_run" in _declared("ddif_testops.h") and "ddif_groupnorm_fwd" in _declared("ddif_testops.h")
    # ... and the product package binds none of them
    from ddif import runtime

    src = open(runtime.__file__).read()
    assert not [n for n in _declared("ddif_testops.h") if n in src]


def test_gfx950_library_builds_loads_and_exports_every_symbol():
    subprocess.run(["make", "-C", PKG, "-j8", "all"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(PKG, "lib", "libddif.so"))
    for header in ("ddif.h", "ddif_testops.h"):
        for name in _declared(header):
            assert hasattr(lib, name), f"{name} declared in include/{header} but not exported by libddif.so"
    lib.ddif_is_emulated.restype = ctypes.c_int
    assert lib.ddif_is_emulated() == 0
    lib.ddif_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ddif_version()


def test_product_loader_never_picks_the_emulated_build():
    from ddif import runtime

    assert runtime.DEFAULT_LIB.endswith(os.path.join("lib", "libddif.so"))
    src = open(runtime.__file__).read()
    assert "libddif_emu" not in src  # the emulated build is only ever named by tests/
