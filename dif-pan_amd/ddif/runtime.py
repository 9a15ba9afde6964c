"""ctypes binding of libddif.so (C ABI: include/ddif.h) and thin handle wrappers.

There is NO fallback path: if the gfx950 library is missing or the tensors are not on a GPU the calls raise.
`use_library()` exists so the test-suite can point the binding at the host-emulated build of the same sources
(tools/hipemu); the product never does.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref
from typing import Dict, Optional, Sequence, Tuple

import torch

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(PKG_ROOT, "lib", "libddif.so")


class DdifError(RuntimeError):
    pass


class NetCfg(C.Structure):
    _fields_ = [
        ("in_channel", C.c_int32), ("out_channel", C.c_int32), ("inner_channel", C.c_int32),
        ("lms_channel", C.c_int32), ("pan_channel", C.c_int32), ("norm_groups", C.c_int32),
        ("n_channel_mults", C.c_int32), ("channel_mults", C.c_int32 * 8),
        ("n_attn_res", C.c_int32), ("attn_res", C.c_int32 * 8),
        ("res_blocks", C.c_int32), ("image_size", C.c_int32), ("self_condition", C.c_int32),
    ]


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)


class DdpmTables(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("t_model", _FP), ("coef_x0", _FP), ("coef_xt", _FP), ("coef_z", _FP)]


class DdimTables(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("t_model", _FP), ("sqrt_recip", _FP), ("sqrt_recipm1", _FP),
                ("sqrt_ap", _FP), ("dir_coef", _FP), ("sigma", _FP)]


class DpmTables(C.Structure):
    _fields_ = [("n_evals", C.c_int32), ("order", C.c_int32), ("t_model", _FP), ("alpha", _FP), ("sigma", _FP),
                ("ord", _IP), ("cx", _FP), ("a_phi1", _FP), ("inv_r0", _FP), ("inv_r1", _FP), ("r0_frac", _FP),
                ("inv_r01", _FP), ("a_phi2", _FP), ("a_phi3", _FP)]


class ProfResult(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("total_flop", C.c_double),
                ("total_bytes", C.c_double), ("kernel_name", C.c_char * 128), ("steps_recorded", C.c_int64),
                ("launches_per_step", C.c_int64), ("total_mfma_flop", C.c_double)]


class ProfClass(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("total_flop", C.c_double), ("total_bytes", C.c_double),
                ("name", C.c_char * 64), ("total_mfma_flop", C.c_double)]


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise DdifError(
                f"{path} not found: the HIP library has not been built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C dif-pan_amd`). "
                f"There is no CPU/PyTorch fallback for this path.")
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        vp, i32, u64, f32 = C.c_void_p, C.c_int, C.c_uint64, C.c_float
        d.ddif_last_error.restype = C.c_char_p
        d.ddif_version.restype = C.c_char_p
        d.ddif_is_emulated.restype = C.c_int
        d.ddif_net_create.argtypes = [C.POINTER(vp), C.POINTER(NetCfg), i32]
        d.ddif_net_destroy.argtypes = [vp]
        d.ddif_net_destroy.restype = None
        d.ddif_net_load.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]
        d.ddif_net_commit.argtypes = [vp, vp]
        d.ddif_net_refresh.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(vp), vp]
        d.ddif_net_num_params.argtypes = [vp]
        d.ddif_net_num_params.restype = C.c_int64
        d.ddif_plan_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32]
        d.ddif_plan_create_train.argtypes = [C.POINTER(vp), vp, i32, i32, i32]
        d.ddif_plan_train_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        d.ddif_plan_train_bind.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(vp)]
        d.ddif_plan_train_num_grads.argtypes = [vp, C.POINTER(i32)]
        d.ddif_plan_train_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        d.ddif_plan_train_forward_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        d.ddif_plan_train_site.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        d.ddif_plan_train_set_dropout.argtypes = [vp, i32, vp, vp]
        d.ddif_plan_train_set_droppath.argtypes = [vp, vp, vp]
        d.ddif_plan_train_random_masks.argtypes = [vp, u64, u64, f32, f32, vp]
        d.ddif_plan_train_get_dropout.argtypes = [vp, i32, vp, vp]
        d.ddif_plan_train_get_droppath.argtypes = [vp, vp, vp]
        d.ddif_plan_destroy.argtypes = [vp]
        d.ddif_plan_destroy.restype = None
        d.ddif_plan_set_cond.argtypes = [vp, vp, vp]
        d.ddif_plan_forward.argtypes = [vp, vp, vp, vp, vp, vp]
        d.ddif_plan_sample_ddpm.argtypes = [vp, C.POINTER(DdpmTables), vp, vp, u64, u64, f32, f32, i32, vp, vp]
        d.ddif_plan_sample_ddim.argtypes = [vp, C.POINTER(DdimTables), vp, vp, u64, u64, f32, f32, i32, vp, vp]
        d.ddif_plan_sample_dpmpp.argtypes = [vp, C.POINTER(DpmTables), vp, f32, f32, i32, vp, vp]
        d.ddif_plan_q_sample_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        d.ddif_prof_begin.argtypes = [vp, i32, i32]
        d.ddif_prof_collect.argtypes = [vp, C.POINTER(ProfResult)]
        d.ddif_prof_classes.argtypes = [vp, C.POINTER(ProfClass)]
        d.ddif_plan_cost.argtypes = [vp] + [C.POINTER(C.c_double)] * 4
        d.ddif_plan_num_launches.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        d.ddif_debug_set_grid_cap.argtypes = [i32]
        d.ddif_set_math_mode.argtypes = [i32]
        d.ddif_get_math_mode.argtypes = []
        d.ddif_plan_range_status.argtypes = [vp, vp, C.POINTER(i32)]
        d.ddif_set_f16_raw.argtypes = [i32]
        d.ddif_get_f16_raw.argtypes = []
        d.ddif_plan_memory.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        d.ddif_cond_assemble.argtypes = [vp, vp, f32, i32, i32, i32, i32, i32, i32, vp, vp]
        d.ddif_metrics.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp]
        d.ddif_ssim.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp]
        d.ddif_optim_create.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32]
        d.ddif_optim_create_ex.argtypes = [C.POINTER(vp), i32, C.POINTER(C.c_int64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32]
        d.ddif_optim_destroy.argtypes = [vp]
        d.ddif_optim_destroy.restype = None
        d.ddif_optim_step.argtypes = [vp, f32, f32, f32, f32, f32, C.c_int64, f32, i32, f32, C.POINTER(C.c_float), vp]
        self.emulated = bool(d.ddif_is_emulated())

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.dll.ddif_last_error().decode(errors="replace")
            raise DdifError(f"{what} failed (status {rc}): {msg}")


_LIB: Optional[_Lib] = None


def get_lib() -> _Lib:
    """The gfx950 library.  Raises DdifError when it has not been built -- never falls back."""
    global _LIB
    if _LIB is None:
        _LIB = _Lib(DEFAULT_LIB)
    return _LIB


def use_library(path: str) -> _Lib:
    """Point the binding at another build of libddif (tests: the host-emulated build).  Loud on purpose."""
    global _LIB
    _LIB = _Lib(path)
    if _LIB.emulated:
        print(f"[ddif] WARNING: using the HOST-EMULATED test build {path}; this is not the product path.",
              file=sys.stderr)
    return _LIB


def library_loaded_path() -> Optional[str]:
    return _LIB.path if _LIB else None


def _check_tensor(lib: _Lib, t: torch.Tensor, name: str):
    if t.dtype != torch.float32:
        raise DdifError(f"{name}: expected float32, got {t.dtype}")
    if lib.emulated:
        if t.device.type != "cpu":
            raise DdifError(f"{name}: the emulated test build works on CPU tensors")
    elif t.device.type != "cuda":
        raise DdifError(f"{name} is on {t.device}: the ddif kernels run on the GPU only; there is no CPU fallback "
                        f"(move the tensors to cuda)")


def _check_current_device(t: torch.Tensor, name: str):
    """The stateless ops (include/ddif.h) launch on the CURRENT device: a tensor living on another GPU would be read through the wrong
    context.  One process per GPU makes this always true (torch.cuda.set_device(local_rank)); refuse anything else loudly."""
    if t.device.type == "cuda" and t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise DdifError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                        f"make it current (torch.cuda.set_device / one process per GPU) before calling this op")


def _check_shape(t: torch.Tensor, name: str, expected: tuple):
    """The kernels index with the plan's shapes: a mismatched tensor would read out of bounds on the GPU (the process
    dies with a bare 'Aborted').  The reference raises a shape error from torch in the same situation."""
    if tuple(t.shape) != tuple(expected):
        raise DdifError(f"{name}: expected shape {tuple(expected)}, got {tuple(t.shape)}")


def set_debug_grid_cap(max_workgroups: int):
    """TEST HOOK (include/ddif.h ddif_debug_set_grid_cap): cap the persistent conv grids of plans created afterwards."""
    lib = get_lib()
    lib.check(lib.dll.ddif_debug_set_grid_cap(int(max_workgroups)), "ddif_debug_set_grid_cap")


MATH_MODES = {"split": 0, "bf16": 1}


def set_math_mode(mode: str):
    """Arithmetic of the convs of inference plans created AFTERWARDS (include/ddif.h ddif_set_math_mode): "split" = the fp32-class default
    (the parity configuration), "bf16" = the throughput variant (one bf16 MFMA product, fp32 accumulate; BASELINE configs[1] "bf16")."""
    if mode not in MATH_MODES:
        raise DdifError(f"math mode {mode!r}: expected one of {sorted(MATH_MODES)}")
    lib = get_lib()
    lib.check(lib.dll.ddif_set_math_mode(MATH_MODES[mode]), "ddif_set_math_mode")


def get_math_mode() -> str:
    m = get_lib().dll.ddif_get_math_mode()
    return {v: k for k, v in MATH_MODES.items()}[int(m)]


def set_f16_raw(on: bool):
    """Inference plans created AFTERWARDS run convs WITHOUT a GroupNorm prologue on f16x2 under the plan's range watch (True, default) or on bf16x3 (False:
    full fp32 exponent range, six matrix products instead of three) -- include/ddif.h ddif_set_f16_raw.  PlanHandle switches a plan over by itself when
    the watch fires; this is the process-wide override."""
    get_lib().dll.ddif_set_f16_raw(1 if on else 0)


def get_f16_raw() -> bool:
    return bool(get_lib().dll.ddif_get_f16_raw())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(lib: _Lib, dev: torch.device):
    if lib.emulated or dev.type != "cuda":
        return None
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _farr(vals) -> "C.Array":
    vals = [float(v) for v in vals]
    return (C.c_float * len(vals))(*vals)


def _iarr(vals) -> "C.Array":
    vals = [int(v) for v in vals]
    return (C.c_int32 * len(vals))(*vals)


class NetHandle:
    """ddif_net_t: weights of one UNetSR3, repacked on `device`."""

    def __init__(self, cfg: dict, device: torch.device):
        self.lib = get_lib()
        self.device = torch.device(device)
        c = NetCfg()
        for k in ("in_channel", "out_channel", "inner_channel", "lms_channel", "pan_channel", "norm_groups",
                  "res_blocks", "image_size"):
            setattr(c, k, int(cfg[k]))
        c.self_condition = int(bool(cfg["self_condition"]))
        mults, ares = list(cfg["channel_mults"]), list(cfg["attn_res"])
        if len(mults) > 8 or len(ares) > 8:
            raise DdifError("at most 8 channel_mults / attn_res entries")
        c.n_channel_mults, c.n_attn_res = len(mults), len(ares)
        for i, m in enumerate(mults):
            c.channel_mults[i] = int(m)
        for i, a in enumerate(ares):
            c.attn_res[i] = int(a)
        for k in ("fourier_features", "pred_var"):
            if cfg.get(k):
                raise DdifError(f"{k}=True is not implemented by the HIP path (and there is no fallback)")
        if not cfg.get("with_noise_level_emb", True):
            raise DdifError("with_noise_level_emb=False is not implemented by the HIP path")
        h = C.c_void_p()
        idx = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
        self.lib.check(self.lib.dll.ddif_net_create(C.byref(h), C.byref(c), idx), "ddif_net_create")
        self.h = h
        self.in_channel, self.out_channel = int(cfg["in_channel"]), int(cfg["out_channel"])
        self.cond_channel = 2 * int(cfg["lms_channel"]) + 4 * int(cfg["pan_channel"])
        self.plans: Dict[Tuple[int, int, int], "PlanHandle"] = {}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], freqs: torch.Tensor):
        dll = self.lib.dll
        for k, v in sd.items():
            t = v.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            self.lib.check(dll.ddif_net_load(self.h, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()),
                           f"ddif_net_load({k})")
        f = freqs.detach().to("cpu", torch.float32).contiguous()
        shape = (C.c_int64 * 1)(f.numel())
        self.lib.check(dll.ddif_net_load(self.h, b"noise_level_mlp.0.freqs", C.c_void_p(f.data_ptr()), shape, 1),
                       "ddif_net_load(freqs)")
        self.lib.check(dll.ddif_net_commit(self.h, _stream(self.lib, self.device)), "ddif_net_commit")
        self.device_refreshed = False
        self.plans.clear()  # plans hold pointers into the old weight blob

    def refresh_from_device(self, named_params, data_ptrs=None):
        """Training: rewrite the packed weights in place from the DEVICE parameter tensors (one launch, no host copy): `named_params` =
        [(state-dict key, tensor)].  Existing plans stay valid; only train-mode plans may run afterwards (the inference-only merged FFN
        weights are left stale) until the next load_state_dict.
        `data_ptrs` (optional): the tensors' data pointers in the same order, if the caller has them already -- the argument arrays of the call
        (702 keys, 702 checked tensors: 2.5 ms of Python to build) are kept and reused while the pointers do not move."""
        if not isinstance(named_params, list):
            named_params = list(named_params)
        n = len(named_params)
        if data_ptrs is None:
            data_ptrs = tuple(t.data_ptr() for _, t in named_params)
        cache = getattr(self, "_refresh_args", None)
        if cache is None or cache[0] != data_ptrs or cache[1] is not named_params or os.environ.get("DDIF_PARAM_CACHE", "1") == "0":
            keys = (C.c_char_p * n)(*[k.encode() for k, _ in named_params])
            for k, t in named_params:
                t = t.detach()
                _check_tensor(self.lib, t, k)
                if not t.is_contiguous():
                    raise DdifError(f"{k}: parameters must be contiguous")
            ptrs = (C.c_void_p * n)(*data_ptrs)
            cache = self._refresh_args = (data_ptrs, named_params, keys, ptrs)
        _, _, keys, ptrs = cache
        self.lib.check(self.lib.dll.ddif_net_refresh(self.h, n, keys, ptrs, _stream(self.lib, self.device)), "ddif_net_refresh")
        self.device_refreshed = True
        self.epoch = getattr(self, "epoch", 0) + 1  # the plans' cond-only caches (FiLM bodies, kv contexts, folded attn_out weights) are stale

    def plan(self, B: int, H: int, W: int, train: bool = False) -> "PlanHandle":
        # (an inference plan carries the conv arithmetic it was built under: set_math_mode("bf16") afterwards builds another one)
        key = (int(B), int(H), int(W)) + (("train",) if train else (get_math_mode(),))
        p = self.plans.get(key)
        if p is None:
            p = PlanHandle(self, int(B), int(H), int(W), train=train)
            self.plans[key] = p
        return p

    def __del__(self):
        try:
            self.plans.clear()
            if getattr(self, "h", None):
                self.lib.dll.ddif_net_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _row(v, device):
    """A per-sample row (sqrt(alpha_bar_t), sqrt(1 - alpha_bar_t), t) as B contiguous floats: left on `device` when it already is there (the
    library copies it on the stream -- no host round trip, no synchronisation), else on the host."""
    v = v.detach()
    if v.device == device and device.type == "cuda":
        return v.to(torch.float32).contiguous()
    return v.to("cpu", torch.float32).contiguous()


class PlanHandle:
    """ddif_plan_t for batches of B tiles of H x W."""

    def __init__(self, net: NetHandle, B: int, H: int, W: int, train: bool = False):
        self.net, self.lib = net, net.lib
        self.B, self.H, self.W = B, H, W
        self.train = train
        self.h = None
        self.f16_raw = get_f16_raw()  # this plan's convs on raw inputs: f16x2 under the range watch, or (after the watch fired once) bf16x3
        self.math_mode = get_math_mode()  # the conv arithmetic this plan is built -- and keyed in NetHandle.plans -- under; a rebuild (range fallback) restores it
        self.range_fallbacks = 0
        # the range watch costs ONE stream synchronisation + 4-byte read per inference call (forward / sample_*): DDIF_RANGE_CHECK=0 or plan.range_check = False
        # turns it off for callers that must not block (their activations are then their own responsibility); it is skipped by itself inside a stream capture
        self.range_check = os.environ.get("DDIF_RANGE_CHECK", "1") != "0"
        self._prof_args = None
        self._create()
        self._cond_ref = None
        self._cond_ver = None
        self.net_out_channels = net.out_channel

    def _create(self):
        h = C.c_void_p()
        if self.train:
            self.lib.check(self.lib.dll.ddif_plan_create_train(C.byref(h), self.net.h, self.B, self.H, self.W), "ddif_plan_create_train")
        else:
            prev, prev_mode = get_f16_raw(), get_math_mode()
            try:
                set_f16_raw(self.f16_raw)
                if prev_mode != self.math_mode:
                    set_math_mode(self.math_mode)
                self.lib.check(self.lib.dll.ddif_plan_create(C.byref(h), self.net.h, self.B, self.H, self.W), "ddif_plan_create")
            finally:
                set_f16_raw(prev)
                if prev_mode != self.math_mode:
                    set_math_mode(prev_mode)
        self.h = h

    def _range_overflow(self, device) -> bool:
        """ONE synchronisation + 4-byte read per sampler / forward call (never inside a loop): did a conv of this plan stage a value outside the scaled
        half range of the f16x2 split (include/ddif.h ddif_plan_range_status)?"""
        v = C.c_int(0)
        self.lib.check(self.lib.dll.ddif_plan_range_status(self.h, _stream(self.lib, torch.device(device)), C.byref(v)), "ddif_plan_range_status")
        return bool(v.value)

    def _guarded(self, enqueue, device):
        """Run `enqueue` (one library call of an inference plan); when the plan's range watch fired -- a trained checkpoint with residual-stream
        activations beyond 4094 in front of a conv without GroupNorm -- rebuild the plan with those convs on bf16x3 (full fp32 range), restore its
        cond caches and repeat the call: any checkpoint runs (reference utils/misc.py:89-122), never a NaN image."""
        out = enqueue()
        if self.train or not self.range_check:
            return out
        dev = torch.device(device)
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return out  # a synchronisation would invalidate the caller's capture: the flag stays set (sticky) and the first call outside a capture reads it
        if not self._range_overflow(device):
            return out
        if not self.f16_raw:
            raise DdifError("activations outside the fp16 range although this plan keeps raw-input convs on bf16x3 (status -5, DDIF_ERR_RANGE): an infinite input?")
        import warnings

        warnings.warn("ddif: an activation in front of a conv without GroupNorm left the f16x2 range (|x| >= 4094); this plan is rebuilt with those "
                      "convs on bf16x3 (full fp32 range) and the call is repeated", RuntimeWarning, stacklevel=3)
        cond = getattr(self, "_keep", None)
        self.lib.dll.ddif_plan_destroy(self.h)
        self.h = None
        self.f16_raw = False
        self.range_fallbacks += 1
        self._create()
        self._cond_ref = None
        if cond is not None:
            self.lib.check(self.lib.dll.ddif_plan_set_cond(self.h, _ptr(cond), _stream(self.lib, cond.device)), "ddif_plan_set_cond")
            self._cond_ref = None  # (the next set_cond of the caller re-validates its own tensor)
        if self._prof_args is not None:  # the destroyed plan took its profiling state with it: re-arm (events recorded so far are lost, the caller's collect sees the rest)
            self.lib.check(self.lib.dll.ddif_prof_begin(self.h, *self._prof_args), "ddif_prof_begin")
        out = enqueue()
        if self._range_overflow(device):
            raise DdifError("activations outside the fp16 range after the bf16x3 fallback (status -5, DDIF_ERR_RANGE)")
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dll.ddif_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- train mode: dropout / DropPath masks ----------------------------------------------------------------------------
    def train_sites(self):
        """[(C, H, W)] of every dropout site and the number of DropPath sites (execution order, as in the reference)."""
        nd, npth = C.c_int(), C.c_int()
        self.lib.check(self.lib.dll.ddif_plan_train_info(self.h, C.byref(nd), C.byref(npth)), "ddif_plan_train_info")
        shapes = []
        for k in range(nd.value):
            c, hh, ww = C.c_int(), C.c_int(), C.c_int()
            self.lib.check(self.lib.dll.ddif_plan_train_site(self.h, k, C.byref(c), C.byref(hh), C.byref(ww)), "ddif_plan_train_site")
            shapes.append((c.value, hh.value, ww.value))
        return shapes, npth.value

    def set_train_masks(self, dropout_masks, droppath_scales):
        """Explicit masks: dropout_masks[k] (B,C,H,W) holding 0 or 1/(1-p); droppath_scales (n_sites, B) holding 0 or 1/(1-p)."""
        shapes, npth = self.train_sites()
        if len(dropout_masks) != len(shapes) or tuple(droppath_scales.shape) != (npth, self.B):
            raise DdifError(f"expected {len(shapes)} dropout masks and DropPath scales of shape {(npth, self.B)}")
        for k, (m, shp) in enumerate(zip(dropout_masks, shapes)):
            _check_tensor(self.lib, m, f"dropout_masks[{k}]")
            _check_shape(m, f"dropout_masks[{k}]", (self.B,) + shp)
            m = m.contiguous()
            self.lib.check(self.lib.dll.ddif_plan_train_set_dropout(self.h, k, _ptr(m), _stream(self.lib, m.device)), "ddif_plan_train_set_dropout")
            self._keep_mask = m
        sc = droppath_scales.detach().to("cpu", torch.float32).contiguous()
        self.lib.check(self.lib.dll.ddif_plan_train_set_droppath(self.h, C.c_void_p(sc.data_ptr()), _stream(self.lib, self.net.device)), "ddif_plan_train_set_droppath")

    def random_train_masks(self, seed: int, tile0: int, p_dropout: float, p_droppath: float):
        self.lib.check(self.lib.dll.ddif_plan_train_random_masks(self.h, int(seed), int(tile0), float(p_dropout), float(p_droppath),
                                                                 _stream(self.lib, self.net.device)), "ddif_plan_train_random_masks")

    def train_masks(self):
        """The masks in force: ([dropout mask (B,C,H,W) per site], DropPath scales (n_sites, B)) -- what `set_train_masks` accepts."""
        shapes, npth = self.train_sites()
        dev = torch.device(self.net.device)
        st = _stream(self.lib, dev)
        masks = []
        for k, shp in enumerate(shapes):
            m = torch.empty((self.B,) + shp, dtype=torch.float32, device=dev)
            self.lib.check(self.lib.dll.ddif_plan_train_get_dropout(self.h, k, _ptr(m), st), "ddif_plan_train_get_dropout")
            masks.append(m)
        sc = torch.empty((npth, self.B), dtype=torch.float32, device=dev)
        if npth:
            self.lib.check(self.lib.dll.ddif_plan_train_get_droppath(self.h, _ptr(sc), st), "ddif_plan_train_get_droppath")
        return masks, sc

    # -- native training step ---------------------------------------------------------------------------------------
    def train_bind(self, named_grads):
        """Name the gradient tensors the reverse pass writes: [(state-dict key, contiguous fp32 tensor of the parameter's shape)] for every
        learnable tensor.  They must stay alive and in place while the plan is used."""
        named_grads = list(named_grads)
        n = len(named_grads)
        for k, t in named_grads:
            _check_tensor(self.lib, t, f"grad[{k}]")
            if not t.is_contiguous():
                raise DdifError(f"grad[{k}] must be contiguous")
        keys = (C.c_char_p * n)(*[k.encode() for k, _ in named_grads])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in named_grads])
        self.lib.check(self.lib.dll.ddif_plan_train_bind(self.h, n, keys, ptrs), "ddif_plan_train_bind")
        self._grads_keep = [t for _, t in named_grads]

    def train_step(self, x0, noise, a, s, time, self_cond, want_pred=True):
        """One iteration's device work: q_sample, train-mode forward, L1 loss, backward (gradients -> the bound tensors).
        Returns (loss: 0-d device tensor, pred or None)."""
        img = (self.B, self.net.out_channel, self.H, self.W)
        for nm, t in (("x_start", x0), ("noise", noise)) + ((("self_cond", self_cond),) if self_cond is not None else ()):
            _check_tensor(self.lib, t, nm)
            _check_shape(t, nm, img)
        x0, noise = x0.contiguous(), noise.contiguous()
        sc = None if self_cond is None else self_cond.contiguous()
        a, s, t = _row(a, x0.device), _row(s, x0.device), _row(time, x0.device)
        loss = torch.empty((), dtype=torch.float32, device=x0.device)
        pred = torch.empty_like(x0) if want_pred else None
        self.lib.check(self.lib.dll.ddif_plan_train_step(
            self.h, _ptr(x0), _ptr(noise), C.c_void_p(a.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(t.data_ptr()), _ptr(sc), _ptr(loss), _ptr(pred),
            _stream(self.lib, x0.device)), "ddif_plan_train_step")
        return loss, pred

    def train_forward_backward(self, x, time, self_cond, target):
        """Forward + L1 loss + backward on a given network input / target (parity tests).  Returns (loss, pred)."""
        img = (self.B, self.net.out_channel, self.H, self.W)
        for nm, t in (("x", x), ("target", target)) + ((("self_cond", self_cond),) if self_cond is not None else ()):
            _check_tensor(self.lib, t, nm)
            _check_shape(t, nm, img)
        x, target = x.contiguous(), target.contiguous()
        sc = None if self_cond is None else self_cond.contiguous()
        t = time.detach().to("cpu", torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        pred = torch.empty_like(x)
        self.lib.check(self.lib.dll.ddif_plan_train_forward_backward(self.h, _ptr(x), C.c_void_p(t.data_ptr()), _ptr(sc), _ptr(target), _ptr(loss), _ptr(pred),
                                                                     _stream(self.lib, x.device)), "ddif_plan_train_forward_backward")
        return loss, pred

    # -- cond ---------------------------------------------------------------------------------------------------
    def set_cond(self, cond: torch.Tensor, force: bool = False):
        _check_tensor(self.lib, cond, "cond")
        _check_shape(cond, "cond", (self.B, self.net.cond_channel, self.H, self.W))
        cond_c = cond.contiguous()
        epoch = getattr(self.net, "epoch", 0)
        same = (not force and self._cond_ref is not None and self._cond_ref() is cond
                and self._cond_ver == cond._version and cond_c is cond and getattr(self, "_cond_epoch", epoch) == epoch)
        if same:
            return
        self.lib.check(self.lib.dll.ddif_plan_set_cond(self.h, _ptr(cond_c), _stream(self.lib, cond.device)),
                       "ddif_plan_set_cond")
        self._keep = cond_c  # borrowed by the enqueued work
        self._cond_ref = weakref.ref(cond)
        self._cond_ver = cond._version
        self._cond_epoch = epoch

    # -- network ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, time: torch.Tensor, self_cond: Optional[torch.Tensor]) -> torch.Tensor:
        _check_tensor(self.lib, x, "x")
        _check_shape(x, "x", (self.B, self.net.in_channel, self.H, self.W))
        x = x.contiguous()
        sc = None
        if self_cond is not None:
            _check_tensor(self.lib, self_cond, "self_cond")
            _check_shape(self_cond, "self_cond", (self.B, self.net.out_channel, self.H, self.W))
            sc = self_cond.contiguous()
        t = time.detach().to("cpu", torch.float32).contiguous()
        if t.numel() != self.B:
            t = t.reshape(-1).expand(self.B).contiguous()
        out = torch.empty((self.B, self.net_out_channels, self.H, self.W), dtype=torch.float32, device=x.device)
        def run():
            self.lib.check(self.lib.dll.ddif_plan_forward(self.h, _ptr(x), C.c_void_p(t.data_ptr()), _ptr(sc), _ptr(out),
                                                          _stream(self.lib, x.device)), "ddif_plan_forward")
            return out

        return self._guarded(run, x.device)

    def _check_sampler_inputs(self, x_T, noise, n_steps):
        img = (self.B, self.net.out_channel, self.H, self.W)
        if x_T is not None:
            _check_tensor(self.lib, x_T, "x_T")
            _check_shape(x_T, "x_T", img)
        if noise is not None:
            _check_tensor(self.lib, noise, "noise")
            if noise.dim() != 5 or tuple(noise.shape[1:]) != img or noise.shape[0] < n_steps:
                raise DdifError(f"noise: expected shape (>={n_steps},) + {img}, got {tuple(noise.shape)}")

    # -- samplers -----------------------------------------------------------------------------------------------
    def sample_ddpm(self, t_model, c_x0, c_xt, c_z, x_T, noise, seed, tile0, clamp, device) -> torch.Tensor:
        n = len(t_model)
        keep = [_farr(t_model), _farr(c_x0), _farr(c_xt), _farr(c_z)]
        tabs = DdpmTables(n, *[C.cast(a, _FP) for a in keep])
        self._check_sampler_inputs(x_T, noise, n)
        x_T = None if x_T is None else x_T.contiguous()
        noise = None if noise is None else noise.contiguous()
        out = torch.empty((self.B, self.net_out_channels, self.H, self.W), dtype=torch.float32, device=device)
        lo, hi, do = (clamp[0], clamp[1], 1) if clamp is not None else (0.0, 0.0, 0)
        def run():
            self.lib.check(self.lib.dll.ddif_plan_sample_ddpm(self.h, C.byref(tabs), _ptr(x_T), _ptr(noise), int(seed),
                                                              int(tile0), lo, hi, do, _ptr(out),
                                                              _stream(self.lib, torch.device(device))), "ddif_plan_sample_ddpm")
            return out

        return self._guarded(run, device)

    def sample_ddim(self, t_model, sqrt_recip, sqrt_recipm1, sqrt_ap, dir_coef, sigma, x_T, noise, seed, tile0, clamp,
                    device) -> torch.Tensor:
        n = len(t_model)
        keep = [_farr(v) for v in (t_model, sqrt_recip, sqrt_recipm1, sqrt_ap, dir_coef, sigma)]
        tabs = DdimTables(n, *[C.cast(a, _FP) for a in keep])
        self._check_sampler_inputs(x_T, noise, n)
        x_T = None if x_T is None else x_T.contiguous()
        noise = None if noise is None else noise.contiguous()
        out = torch.empty((self.B, self.net_out_channels, self.H, self.W), dtype=torch.float32, device=device)
        lo, hi, do = (clamp[0], clamp[1], 1) if clamp is not None else (0.0, 0.0, 0)
        def run():
            self.lib.check(self.lib.dll.ddif_plan_sample_ddim(self.h, C.byref(tabs), _ptr(x_T), _ptr(noise), int(seed),
                                                              int(tile0), lo, hi, do, _ptr(out),
                                                              _stream(self.lib, torch.device(device))), "ddif_plan_sample_ddim")
            return out

        return self._guarded(run, device)

    def sample_dpmpp(self, tabs: dict, x_T: torch.Tensor, clamp) -> torch.Tensor:
        self._check_sampler_inputs(x_T, None, 0)
        x_T = x_T.contiguous()
        keep = {k: (_iarr(v) if k == "ord" else _farr(v)) for k, v in tabs.items() if k not in ("n_evals", "order")}
        t = DpmTables()
        t.n_evals, t.order = int(tabs["n_evals"]), int(tabs["order"])
        for k, arr in keep.items():
            setattr(t, k, C.cast(arr, _IP if k == "ord" else _FP))
        out = torch.empty_like(x_T)
        lo, hi, do = (clamp[0], clamp[1], 1) if clamp is not None else (0.0, 0.0, 0)
        def run():
            self.lib.check(self.lib.dll.ddif_plan_sample_dpmpp(self.h, C.byref(t), _ptr(x_T), lo, hi, do, _ptr(out),
                                                               _stream(self.lib, x_T.device)), "ddif_plan_sample_dpmpp")
            return out

        return self._guarded(run, x_T.device)

    def q_sample_forward(self, x0, noise, a, s, time, self_cond) -> torch.Tensor:
        img = (self.B, self.net.out_channel, self.H, self.W)
        for nm, t in (("x_start", x0), ("noise", noise)) + ((("self_cond", self_cond),) if self_cond is not None else ()):
            _check_tensor(self.lib, t, nm)
            _check_shape(t, nm, img)
        x0, noise = x0.contiguous(), noise.contiguous()
        sc = None if self_cond is None else self_cond.contiguous()
        a, s, t = _row(a, x0.device), _row(s, x0.device), _row(time, x0.device)
        out = torch.empty_like(x0)
        def run():
            self.lib.check(self.lib.dll.ddif_plan_q_sample_forward(
                self.h, _ptr(x0), _ptr(noise), C.c_void_p(a.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(t.data_ptr()),
                _ptr(sc), _ptr(out), _stream(self.lib, x0.device)), "ddif_plan_q_sample_forward")
            return out

        return self._guarded(run, x0.device)

    # -- measurement --------------------------------------------------------------------------------------------
    def prof_begin(self, every_n_steps: int, max_events: int):
        self._prof_args = (int(every_n_steps), int(max_events))
        self.lib.check(self.lib.dll.ddif_prof_begin(self.h, every_n_steps, max_events), "ddif_prof_begin")

    def prof_collect(self) -> dict:
        r = ProfResult()
        self.lib.check(self.lib.dll.ddif_prof_collect(self.h, C.byref(r)), "ddif_prof_collect")
        cls = (ProfClass * 6)()
        self.lib.check(self.lib.dll.ddif_prof_classes(self.h, cls), "ddif_prof_classes")
        classes = [dict(name=c.name.decode(), launches=c.launches, total_ms=c.total_ms, total_flop=c.total_flop, total_bytes=c.total_bytes,
                        total_mfma_flop=c.total_mfma_flop) for c in cls]
        return dict(launches=r.launches, total_ms=r.total_ms, total_flop=r.total_flop, total_bytes=r.total_bytes,
                    kernel=r.kernel_name.decode(), classes=classes, steps_recorded=int(r.steps_recorded),
                    launches_per_step=int(r.launches_per_step), total_mfma_flop=r.total_mfma_flop)

    def num_launches(self) -> dict:
        a, b = C.c_int(), C.c_int()
        self.lib.check(self.lib.dll.ddif_plan_num_launches(self.h, C.byref(a), C.byref(b)), "ddif_plan_num_launches")
        return dict(step=a.value, cond=b.value)

    def memory(self) -> dict:
        v = [C.c_int64() for _ in range(3)]
        self.lib.check(self.lib.dll.ddif_plan_memory(self.h, *[C.byref(x) for x in v]), "ddif_plan_memory")
        return dict(total_bytes=v[0].value, arena_bytes=v[1].value, unaliased_bytes=v[2].value)

    def cost(self) -> dict:
        v = [C.c_double() for _ in range(4)]
        self.lib.check(self.lib.dll.ddif_plan_cost(self.h, *[C.byref(x) for x in v]), "ddif_plan_cost")
        return dict(step_flop=v[0].value, step_bytes=v[1].value, cond_flop=v[2].value, cond_bytes=v[3].value)


# ---- kernels either side of the denoising loop (include/ddif.h, csrc/kernels_aux.h) ---------------------------------------
def cond_assemble(lms_raw: torch.Tensor, pan_raw: torch.Tensor, division: float, wavelet_order: int = 0) -> torch.Tensor:
    """cond = cat[lms, pan, bilinear_up2(Haar wavelets)] / division in ONE kernel (reference diffusion_engine.py:221-228 +
    dataset/pan_dataset.py:73-81,139-142).  wavelet_order 0 = [LL,H,D,V] (pan sets), 1 = [LL,H,V,D] (CAVE / Harvard)."""
    lib = get_lib()
    _check_tensor(lib, lms_raw, "lms")
    _check_tensor(lib, pan_raw, "pan")
    B, Cc, H, W = lms_raw.shape
    _check_shape(pan_raw, "pan", (B, pan_raw.shape[1], H, W))
    P = pan_raw.shape[1]
    lms_raw, pan_raw = lms_raw.contiguous(), pan_raw.contiguous()
    out = torch.empty((B, 2 * Cc + 4 * P, H, W), dtype=torch.float32, device=lms_raw.device)
    lib.check(lib.dll.ddif_cond_assemble(_ptr(lms_raw), _ptr(pan_raw), float(division), B, Cc, P, H, W, int(wavelet_order), _ptr(out),
                                         _stream(lib, lms_raw.device)), "ddif_cond_assemble")
    return out


def metrics(gt: torch.Tensor, pred: torch.Tensor, ergas_ratio: float = 4.0) -> torch.Tensor:
    """(B, 4) = [SAM, ERGAS, PSNR (reference sign), CC] per image (reference utils/_metric_legacy.py:299-379)."""
    lib = get_lib()
    _check_tensor(lib, gt, "gt")
    _check_tensor(lib, pred, "pred")
    _check_shape(pred, "pred", tuple(gt.shape))
    B, Cc, H, W = gt.shape
    gt, pred = gt.contiguous(), pred.contiguous()
    out = torch.empty((B, 4), dtype=torch.float32, device=gt.device)
    lib.check(lib.dll.ddif_metrics(_ptr(gt), _ptr(pred), B, Cc, H, W, float(ergas_ratio), _ptr(out), _stream(lib, gt.device)), "ddif_metrics")
    return out


def ssim(gt: torch.Tensor, pred: torch.Tensor, data_range: float = 2.0) -> torch.Tensor:
    """(B,) SSIM per image = skimage.metrics.structural_similarity(gt[b], pred[b], channel_axis=0) with library defaults (reference
    utils/metric.py:153-166).  data_range 2.0 is what skimage derives for float images when none is given (dtype range (-1, 1)), which is
    how the reference calls it.  skimage is not in the build image: parity-unpinned restatement."""
    lib = get_lib()
    _check_tensor(lib, gt, "gt")
    _check_tensor(lib, pred, "pred")
    _check_shape(pred, "pred", tuple(gt.shape))
    B, Cc, H, W = gt.shape
    gt, pred = gt.contiguous(), pred.contiguous()
    out = torch.empty((B,), dtype=torch.float32, device=gt.device)
    lib.check(lib.dll.ddif_ssim(_ptr(gt), _ptr(pred), B, Cc, H, W, float(data_range), _ptr(out), _stream(lib, gt.device)), "ddif_ssim")
    return out


class FusedAdamW:
    """clip_grad_norm_ + torch.optim.AdamW.step + EmaUpdater.update as three launches (reference diffusion_engine.py:237-241).
    `params`, `grads` (and `ema`, optional) are lists of contiguous fp32 tensors that stay alive and in place.  The moments (exp_avg,
    exp_avg_sq) are torch tensors owned by this object, so `state_dict()` / `load_state_dict()` make the optimizer checkpointable.
    The kernels write the parameters through raw pointers: `step()` bumps the tensors' version counters so that anything keyed on
    `Tensor._version` (UNetSR3's packed-weight cache) sees the update."""

    def __init__(self, params, grads, ema=None, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.lib = get_lib()
        self.params, self.grads, self.ema = list(params), list(grads), (list(ema) if ema is not None else None)
        for i, (p, g) in enumerate(zip(self.params, self.grads)):
            _check_tensor(self.lib, p, f"param[{i}]")
            _check_tensor(self.lib, g, f"grad[{i}]")
            if not (p.is_contiguous() and g.is_contiguous()) or p.numel() != g.numel():
                raise DdifError(f"param[{i}] / grad[{i}] must be contiguous and of equal size")
        n = len(self.params)
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        sizes = (C.c_int64 * n)(*[p.numel() for p in self.params])
        pp = (C.c_void_p * n)(*[p.data_ptr() for p in self.params])
        gp = (C.c_void_p * n)(*[g.data_ptr() for g in self.grads])
        mp = (C.c_void_p * n)(*[m.data_ptr() for m in self.exp_avg])
        vp = (C.c_void_p * n)(*[v.data_ptr() for v in self.exp_avg_sq])
        ep = None
        if self.ema is not None:
            ep = (C.c_void_p * n)(*[e.data_ptr() for e in self.ema])
        dev = self.params[0].device
        idx = dev.index if dev.type == "cuda" and dev.index is not None else 0
        h = C.c_void_p()
        self.lib.check(self.lib.dll.ddif_optim_create_ex(C.byref(h), n, sizes, pp, gp, ep, mp, vp, idx), "ddif_optim_create_ex")
        self.h, self.device = h, dev
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.t = 0

    def step(self, max_grad_norm: float = 0.0, ema_mode: int = 0, ema_decay: float = 0.0, return_norm: bool = False):
        self.t += 1
        gn = C.c_float(0.0)
        self.lib.check(self.lib.dll.ddif_optim_step(self.h, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t,
                                                    float(max_grad_norm), int(ema_mode), float(ema_decay),
                                                    C.byref(gn) if return_norm else None, _stream(self.lib, self.device)), "ddif_optim_step")
        _bump_versions(self.params)
        if self.ema is not None and ema_mode:
            _bump_versions(self.ema)
        return gn.value if return_norm else None

    def state_dict(self) -> dict:
        """torch.optim.AdamW-shaped state: per-parameter exp_avg / exp_avg_sq (in `params` order) + the step count and hyper-parameters."""
        return {"step": self.t, "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                "exp_avg": [m.detach().cpu().clone() for m in self.exp_avg], "exp_avg_sq": [v.detach().cpu().clone() for v in self.exp_avg_sq]}

    def load_state_dict(self, sd: dict):
        if len(sd["exp_avg"]) != len(self.exp_avg) or len(sd["exp_avg_sq"]) != len(self.exp_avg_sq):
            raise DdifError("FusedAdamW.load_state_dict: the state holds a different number of tensors")
        with torch.no_grad():
            for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
                if tuple(dst.shape) != tuple(src.shape):
                    raise DdifError("FusedAdamW.load_state_dict: moment shape mismatch")
                dst.copy_(src)  # in place: the handle keeps these pointers
        self.t = int(sd["step"])
        self.lr, self.betas, self.eps, self.weight_decay = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dll.ddif_optim_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _bump_versions(tensors):
    """The library wrote these tensors through raw pointers: tell torch (version counters) so caches keyed on `_version` invalidate."""
    inc = getattr(torch._C, "_increment_version", None)
    if inc is None:  # pragma: no cover
        raise DdifError("torch._C._increment_version is missing: cannot signal in-place parameter updates")
    try:
        inc(list(tensors))  # torch >= 2.3: an iterable of tensors (a bare tensor would be iterated row by row)
    except TypeError:
        for t in tensors:
            inc(t)
