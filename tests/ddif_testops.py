"""TEST SCAFFOLDING (moved out of the product package in round 6): ctypes wrappers of the per-op C-ABI entry points declared in
include/ddif_testops.h -- the stateless forward / backward ops round 2's op-by-op training tape was built from.  The product trains through ONE call
(ddif_plan_train_step, csrc/ddif_train.cpp); these ops stay in libddif.so as an independent cross-check of that reverse program (tests/train_tape.py,
tests/test_backward_ops.py) and nothing in `ddif/` imports this module.

Two groups: the backward ops (`Conv3x3Backward`, `BlockBackward`, `*_backward`, `linattn_nhwc*`; formerly ddif/runtime.py) and the functional forward ops with
torch.nn.functional-like signatures (`conv2d`, `group_norm`, ...; formerly ddif/functional.py).  Tensors are torch fp32 NCHW on the GPU (CPU tensors only with the
emulated test build).  No torch arithmetic in here: torch allocates, views (cat / chunk / pad are data movement) and nothing else."""
import ctypes as C

import torch

from ddif import runtime as R
from ddif.runtime import DdifError, _check_current_device, _check_shape, _check_tensor, _ptr, _stream

_BOUND = set()


def get_lib():
    """The library the product loaded, with the argument types of the test-only entry points bound on first use."""
    lib = R.get_lib()
    if id(lib) not in _BOUND:
        d = lib.dll
        vp, i32 = C.c_void_p, C.c_int
        d.ddif_convbwd_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32]
        d.ddif_convbwd_destroy.argtypes = [vp]
        d.ddif_convbwd_destroy.restype = None
        d.ddif_convbwd_run.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        d.ddif_blockbwd_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32]
        d.ddif_blockbwd_create_ex.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32, i32, i32, i32]
        d.ddif_blockbwd_destroy.argtypes = [vp]
        d.ddif_blockbwd_destroy.restype = None
        d.ddif_blockbwd_run.argtypes = [vp] + [vp] * 13
        d.ddif_dwconv3x3_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
        d.ddif_convfwd_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32, i32, i32, i32, i32]
        d.ddif_convfwd_destroy.argtypes = [vp]
        d.ddif_convfwd_destroy.restype = None
        d.ddif_convfwd_run.argtypes = [vp, vp, vp, vp, vp, vp]
        d.ddif_dwconv3x3_fwd.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        d.ddif_groupnorm_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
        d.ddif_swish_fwd.argtypes = [vp, C.c_int64, vp, vp]
        d.ddif_film_fwd.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
        d.ddif_add_scaled.argtypes = [vp, vp, vp, i32, C.c_int64, vp, vp]
        d.ddif_linear_fwd.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
        d.ddif_selfattn_core_fwd.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        d.ddif_linattn_core_fwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        d.ddif_q_sample.argtypes = [vp, vp, vp, vp, i32, C.c_int64, vp, vp]
        d.ddif_l1_loss_fwd.argtypes = [vp, vp, C.c_int64, vp, vp]
        d.ddif_film_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
        d.ddif_selfattn_core_bwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        d.ddif_linattn_core_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        d.ddif_linattn_nhwc_workspace.argtypes = [i32, i32, i32, i32, i32]
        d.ddif_linattn_nhwc_workspace.restype = C.c_int64
        d.ddif_linattn_nhwc_fwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        d.ddif_linattn_nhwc_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        d.ddif_linear_bwd.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
        d.ddif_groupnorm_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
        d.ddif_swish_bwd.argtypes = [vp, vp, C.c_int64, vp, vp]
        d.ddif_l1_loss_bwd.argtypes = [vp, vp, C.c_int64, C.c_float, vp, vp]
        _BOUND.add(id(lib))
    return lib


class Conv3x3Backward:
    """Backward of nn.Conv2d(Cin, Cout, 3, padding=1) (autograd under loss.backward(), reference diffusion_engine.py:233):
    returns (dx, dw, db) for x (B,Cin,H,W), w (Cout,Cin,3,3), dy (B,Cout,H,W) -- hand-written dgrad / wgrad kernels."""

    def __init__(self, B, Cin, Cout, H, W, device):
        self.lib = get_lib()
        dev = torch.device(device)
        idx = dev.index if dev.type == "cuda" and dev.index is not None else 0
        h = C.c_void_p()
        self.lib.check(self.lib.dll.ddif_convbwd_create(C.byref(h), B, Cin, Cout, H, W, idx), "ddif_convbwd_create")
        self.h, self.shape, self.device = h, (B, Cin, Cout, H, W), dev

    def __call__(self, x, w, dy, need_dx=True, need_dw=True, need_db=True):
        B, Cin, Cout, H, W = self.shape
        for nm, t, shp in (("x", x, (B, Cin, H, W)), ("w", w, (Cout, Cin, 3, 3)), ("dy", dy, (B, Cout, H, W))):
            _check_tensor(self.lib, t, nm)
            _check_shape(t, nm, shp)
        x, w, dy = x.contiguous(), w.contiguous(), dy.contiguous()
        dx = torch.empty_like(x) if need_dx else None
        dw = torch.empty_like(w) if need_dw else None
        db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if need_db else None
        self.lib.check(self.lib.dll.ddif_convbwd_run(self.h, _ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db),
                                                     _stream(self.lib, x.device)), "ddif_convbwd_run")
        return dx, dw, db

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dll.ddif_convbwd_destroy(self.h)
                self.h = None
        except Exception:
            pass


class BlockBackward:
    """Backward of one `Block` of the denoiser (reference models/sr3_dwt.py:288-300: GroupNorm(1 group) -> Swish -> Dropout ->
    conv3x3) as autograd runs it under loss.backward() (diffusion_engine.py:233).  `mask` is the dropout site's mask (0 or
    1/(1-p), what `PlanHandle.train_sites` / `set_train_masks` carry), None in eval mode.  Returns a dict with dx, dgamma,
    dbeta, dw, db and dy_plane_sums (B, Cout) -- the gradient of the time bias FeatureWiseAffine adds behind block1."""

    PRO = {"none": 0, "gn": 1, "gn_silu": 2, "silu": 3}
    RESAMPLE = {"plain": 0, "down2": 1, "up2": 2}

    def __init__(self, B, Cin, Cout, H, W, device, ks=3, pro="gn_silu", resample="plain"):
        self.lib = get_lib()
        dev = torch.device(device)
        idx = dev.index if dev.type == "cuda" and dev.index is not None else 0
        h = C.c_void_p()
        self.lib.check(self.lib.dll.ddif_blockbwd_create_ex(C.byref(h), B, Cin, Cout, H, W, ks, self.PRO[pro], self.RESAMPLE[resample], idx),
                       "ddif_blockbwd_create_ex")
        self.h, self.shape, self.device, self.ks, self.pro = h, (B, Cin, Cout, H, W), dev, ks, pro
        self.out_hw = {"plain": (H, W), "down2": ((H - 1) // 2 + 1, (W - 1) // 2 + 1), "up2": (2 * H, 2 * W)}[resample]

    def __call__(self, x, gamma, beta, w, dy, mask=None, need_dx=True):
        B, Cin, Cout, H, W = self.shape
        named = [("x", x, (B, Cin, H, W)), ("w", w, (Cout, Cin, self.ks, self.ks)), ("dy", dy, (B, Cout) + self.out_hw)]
        if self.pro in ("gn", "gn_silu"):
            named += [("gamma", gamma, (Cin,)), ("beta", beta, (Cin,))]
        else:
            gamma = beta = None
        if mask is not None:
            named.append(("mask", mask, (B, Cin, H, W)))
        for nm, t, shp in named:
            _check_tensor(self.lib, t, nm)
            _check_shape(t, nm, shp)
        x, w, dy = x.contiguous(), w.contiguous(), dy.contiguous()
        gamma = gamma.contiguous() if gamma is not None else None
        beta = beta.contiguous() if beta is not None else None
        mask = mask.contiguous() if mask is not None else None
        f = dict(dtype=torch.float32, device=x.device)
        gn = self.pro in ("gn", "gn_silu")
        out = {"dx": torch.empty_like(x) if need_dx else None, "dgamma": torch.empty((Cin,), **f) if gn else None,
               "dbeta": torch.empty((Cin,), **f) if gn else None,
               "dw": torch.empty_like(w), "db": torch.empty((Cout,), **f), "dy_plane_sums": torch.empty((B, Cout), **f)}
        self.lib.check(self.lib.dll.ddif_blockbwd_run(self.h, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mask), _ptr(w), _ptr(dy), _ptr(out["dx"]),
                                                      _ptr(out["dgamma"]), _ptr(out["dbeta"]), _ptr(out["dw"]), _ptr(out["db"]),
                                                      _ptr(out["dy_plane_sums"]), _stream(self.lib, x.device)), "ddif_blockbwd_run")
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dll.ddif_blockbwd_destroy(self.h)
                self.h = None
        except Exception:
            pass


# ---- stateless backward ops (include/ddif.h "stateless backward ops"): each returns the gradients autograd would produce --------
def _ops_prepare(named):
    lib = get_lib()
    out = []
    for nm, t, shp in named:
        _check_tensor(lib, t, nm)
        _check_shape(t, nm, shp)
        _check_current_device(t, nm)
        out.append(t.contiguous())
    return lib, out


def dwconv3x3_backward(x, w, dy):
    """Depthwise conv3x3 (groups = C, pad 1, no bias; reference models/sr3_dwt.py:507-520): (dx, dw)."""
    B, Cc, H, W = x.shape
    lib, (x, w, dy) = _ops_prepare([("x", x, (B, Cc, H, W)), ("w", w, (Cc, 1, 3, 3)), ("dy", dy, (B, Cc, H, W))])
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    lib.check(lib.dll.ddif_dwconv3x3_bwd(_ptr(x), _ptr(w), _ptr(dy), B, Cc, H, W, _ptr(dx), _ptr(dw), _stream(lib, x.device)), "ddif_dwconv3x3_bwd")
    return dx, dw


def film_backward(xc, scale_shift, dout):
    """CondInjection's xc * (1 + scale) + shift (reference :395-396): (dxc, dscale_shift)."""
    B, Cc, H, W = xc.shape
    lib, (xc, ss, dout) = _ops_prepare([("xc", xc, (B, Cc, H, W)), ("scale_shift", scale_shift, (B, 2 * Cc, H, W)), ("dout", dout, (B, Cc, H, W))])
    dxc, dss = torch.empty_like(xc), torch.empty_like(ss)
    lib.check(lib.dll.ddif_film_bwd(_ptr(xc), _ptr(ss), _ptr(dout), B, Cc, H, W, _ptr(dxc), _ptr(dss), _stream(lib, xc.device)), "ddif_film_bwd")
    return dxc, dss


def selfattn_core_backward(qkv, dout, heads=8):
    """SelfAttention core (reference :345-358): gradient of qkv (B, 3C, H, W) given that of the weighted sum (B, C, H, W)."""
    B, C3, H, W = qkv.shape
    lib, (qkv, dout) = _ops_prepare([("qkv", qkv, (B, C3, H, W)), ("dout", dout, (B, C3 // 3, H, W))])
    dqkv = torch.empty_like(qkv)
    lib.check(lib.dll.ddif_selfattn_core_bwd(_ptr(qkv), _ptr(dout), B, C3 // 3, H, W, heads, _ptr(dqkv), _stream(lib, qkv.device)), "ddif_selfattn_core_bwd")
    return dqkv


def linattn_core_backward(q_pre, kv_pre, dout, heads=8):
    """FastAttnCondInjection core (reference :545-566): (dq_pre, dkv_pre) given the gradient of the attention output."""
    B, qd, H, W = q_pre.shape
    lib, (q_pre, kv_pre, dout) = _ops_prepare([("q_pre", q_pre, (B, qd, H, W)), ("kv_pre", kv_pre, (B, 2 * qd, H, W)), ("dout", dout, (B, qd, H, W))])
    dq, dkv = torch.empty_like(q_pre), torch.empty_like(kv_pre)
    lib.check(lib.dll.ddif_linattn_core_bwd(_ptr(q_pre), _ptr(kv_pre), _ptr(dout), B, qd, H, W, heads, _ptr(dq), _ptr(dkv), _stream(lib, q_pre.device)),
              "ddif_linattn_core_bwd")
    return dq, dkv


def linear_backward(x, w, dy):
    """nn.Linear (reference :59-64, 241-258): (dx, dw, db)."""
    B, nin = x.shape
    nout = w.shape[0]
    lib, (x, w, dy) = _ops_prepare([("x", x, (B, nin)), ("w", w, (nout, nin)), ("dy", dy, (B, nout))])
    dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty((nout,), dtype=torch.float32, device=x.device)
    lib.check(lib.dll.ddif_linear_bwd(_ptr(x), _ptr(w), _ptr(dy), B, nin, nout, _ptr(dx), _ptr(dw), _ptr(db), _stream(lib, x.device)), "ddif_linear_bwd")
    return dx, dw, db


def swish_backward(x, dy):
    lib, (x, dy) = _ops_prepare([("x", x, tuple(x.shape)), ("dy", dy, tuple(x.shape))])
    dx = torch.empty_like(x)
    lib.check(lib.dll.ddif_swish_bwd(_ptr(x), _ptr(dy), x.numel(), _ptr(dx), _stream(lib, x.device)), "ddif_swish_bwd")
    return dx


def l1_loss_backward(pred, target, upstream=1.0):
    """F.l1_loss(pred, target) (mean) backward (reference diffusion/diffusion_ddpm_pan.py:742-749)."""
    lib, (pred, target) = _ops_prepare([("pred", pred, tuple(pred.shape)), ("target", target, tuple(pred.shape))])
    dp = torch.empty_like(pred)
    lib.check(lib.dll.ddif_l1_loss_bwd(_ptr(pred), _ptr(target), pred.numel(), C.c_float(upstream), _ptr(dp), _stream(lib, pred.device)), "ddif_l1_loss_bwd")
    return dp


def groupnorm_backward(x, gamma, dy):
    """GroupNorm(1 group, eps 1e-5) alone (reference models/sr3_dwt.py:540-573 prenorm_x): (dx, dgamma, dbeta)."""
    B, Cc, H, W = x.shape
    lib, (x, gamma, dy) = _ops_prepare([("x", x, (B, Cc, H, W)), ("gamma", gamma, (Cc,)), ("dy", dy, (B, Cc, H, W))])
    dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
    ws = torch.empty((B * (2 * Cc + 4),), dtype=torch.float64, device=x.device)
    lib.check(lib.dll.ddif_groupnorm_bwd(_ptr(x), _ptr(gamma), _ptr(dy), B, Cc, H, W, _ptr(dx), _ptr(dg), _ptr(db), _ptr(ws), _stream(lib, x.device)),
              "ddif_groupnorm_bwd")
    return dx, dg, db


def linattn_nhwc(q_pre, kv_pre, heads=8):
    """The linear attention core as the native training step runs it: q_pre (B,H,W,qd), kv_pre (B,H,W,2qd), NHWC.  Returns (out (B,H,W,qd),
    workspace) -- the workspace carries the contexts `linattn_nhwc_backward` needs."""
    B, H, W, qd = q_pre.shape
    lib, (q_pre, kv_pre) = _ops_prepare([("q_pre", q_pre, (B, H, W, qd)), ("kv_pre", kv_pre, (B, H, W, 2 * qd))])
    n = lib.dll.ddif_linattn_nhwc_workspace(B, qd, H, W, heads)
    if n < 0:
        lib.check(int(n), "ddif_linattn_nhwc_workspace")  # negative = refused: the message is in ddif_last_error
    ws = torch.empty((n,), dtype=torch.float32, device=q_pre.device)
    out = torch.empty_like(q_pre)
    lib.check(lib.dll.ddif_linattn_nhwc_fwd(_ptr(q_pre), _ptr(kv_pre), B, qd, H, W, heads, _ptr(out), _ptr(ws), _stream(lib, q_pre.device)), "ddif_linattn_nhwc_fwd")
    return out, ws


def linattn_nhwc_backward(q_pre, kv_pre, dout, workspace, heads=8):
    B, H, W, qd = q_pre.shape
    lib, (q_pre, kv_pre, dout) = _ops_prepare([("q_pre", q_pre, (B, H, W, qd)), ("kv_pre", kv_pre, (B, H, W, 2 * qd)), ("dout", dout, (B, H, W, qd))])
    dq, dkv = torch.empty_like(q_pre), torch.empty_like(kv_pre)
    lib.check(lib.dll.ddif_linattn_nhwc_bwd(_ptr(q_pre), _ptr(kv_pre), _ptr(dout), B, qd, H, W, heads, _ptr(dq), _ptr(dkv), _ptr(workspace),
                                            _stream(lib, q_pre.device)), "ddif_linattn_nhwc_bwd")
    return dq, dkv


# ---------------------------------------------------------------------------------------------------------------- functional forward ops (+ conv backward)
_CONV_FWD = {}
_CONV_BWD = {}


def clear_caches():
    """Destroy the per-shape conv handles (each owns its NHWC staging buffers on the device)."""
    lib = get_lib()
    for h in _CONV_FWD.values():
        lib.dll.ddif_convfwd_destroy(h)
    _CONV_FWD.clear()
    _CONV_BWD.clear()  # BlockBackward objects free themselves


def _dev_index(t):
    return t.device.index if t.device.type == "cuda" and t.device.index is not None else 0


def _pad_c(t, dim, to):
    """zero-pad a channel axis up to `to` (the conv kernels need 4 | C: the cond convs have 9 / 11 input channels)"""
    if t.shape[dim] == to:
        return t
    shape = list(t.shape)
    shape[dim] = to - t.shape[dim]
    return torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim=dim)


def _c4(c):
    return (c + 3) & ~3


def conv2d(x, w, b=None, stride=1, up2=False):
    """nn.Conv2d(.., ks, stride, padding = ks // 2) (+ nearest x2 in front when up2: Upsample)."""
    lib = get_lib()
    B, Cin, H, W = x.shape
    Cout, _, ks, _ = w.shape
    ci, co = _c4(Cin), _c4(Cout)
    xp, wp = _pad_c(x, 1, ci).contiguous(), _pad_c(_pad_c(w, 1, ci), 0, co).contiguous()
    bp = None if b is None else _pad_c(b, 0, co).contiguous()
    key = (lib.path, str(x.device), B, ci, co, H, W, ks, stride, bool(up2))
    h = _CONV_FWD.get(key)
    if h is None:
        h = C.c_void_p()
        lib.check(lib.dll.ddif_convfwd_create(C.byref(h), B, ci, co, H, W, ks, stride, 1 if up2 else 0, _dev_index(x)), "ddif_convfwd_create")
        _CONV_FWD[key] = h
    Ho, Wo = (2 * H, 2 * W) if up2 else (((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride == 2 else (H, W))
    y = torch.empty((B, co, Ho, Wo), dtype=torch.float32, device=x.device)
    for nm, t in (("x", xp), ("w", wp), ("y", y)):
        _check_tensor(lib, t, nm)
    lib.check(lib.dll.ddif_convfwd_run(h, _ptr(xp), _ptr(wp), _ptr(bp), _ptr(y), _stream(lib, x.device)), "ddif_convfwd_run")
    return y if co == Cout else y[:, :Cout].contiguous()


def conv2d_backward(x_in, w, dy, pro="none", gamma=None, beta=None, mask=None, stride=1, up2=False, need_dx=True):
    """Backward of conv2d(prologue(x_in), w) where prologue is none / GroupNorm / GroupNorm+SiLU(+mask) / SiLU.  Returns a dict with
    dx (of x_in), dw, db, dgamma, dbeta, dy_plane_sums."""
    lib = get_lib()
    B, Cin, H, W = x_in.shape
    Cout, _, ks, _ = w.shape
    ci, co = _c4(Cin), _c4(Cout)
    if (ci != Cin) and pro != "none":
        raise R.DdifError("conv2d_backward: channel padding only for prologue-free convs")
    resample = "up2" if up2 else ("down2" if stride == 2 else "plain")
    key = (lib.path, str(x_in.device), B, ci, co, H, W, ks, pro, resample)
    op = _CONV_BWD.get(key)
    if op is None:
        op = BlockBackward(B, ci, co, H, W, x_in.device, ks=ks, pro=pro, resample=resample)
        _CONV_BWD[key] = op
    g = op(_pad_c(x_in, 1, ci), gamma, beta, _pad_c(_pad_c(w, 1, ci), 0, co), _pad_c(dy, 1, co), mask=mask, need_dx=need_dx)
    if ci != Cin or co != Cout:
        g["dw"] = g["dw"][:Cout, :Cin].contiguous()
        g["db"] = g["db"][:Cout].contiguous()
        g["dy_plane_sums"] = g["dy_plane_sums"][:, :Cout].contiguous()
        if g["dx"] is not None:
            g["dx"] = g["dx"][:, :Cin].contiguous()
    return g


def _launch(name, fn, *args):
    lib = get_lib()
    lib.check(fn(*args), name)


def dwconv3x3(x, w):
    lib = get_lib()
    _check_current_device(x, "x")
    B, Cc, H, W = x.shape
    x, w = x.contiguous(), w.contiguous()
    y = torch.empty_like(x)
    lib.check(lib.dll.ddif_dwconv3x3_fwd(_ptr(x), _ptr(w), B, Cc, H, W, _ptr(y), _stream(lib, x.device)), "ddif_dwconv3x3_fwd")
    return y


def group_norm(x, gamma, beta, silu=False, mask=None):
    lib = get_lib()
    _check_current_device(x, "x")
    B, Cc, H, W = x.shape
    x = x.contiguous()
    mask = None if mask is None else mask.contiguous()
    y = torch.empty_like(x)
    lib.check(lib.dll.ddif_groupnorm_fwd(_ptr(x), _ptr(gamma.contiguous()), _ptr(beta.contiguous()), _ptr(mask), B, Cc, H, W, 1 if silu else 0, _ptr(y),
                                         _stream(lib, x.device)), "ddif_groupnorm_fwd")
    return y


def swish(x):
    lib = get_lib()
    _check_current_device(x, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    lib.check(lib.dll.ddif_swish_fwd(_ptr(x), x.numel(), _ptr(y), _stream(lib, x.device)), "ddif_swish_fwd")
    return y


def film(xc, scale_shift):
    lib = get_lib()
    _check_current_device(xc, "xc")
    B, Cc, H, W = xc.shape
    xc, ss = xc.contiguous(), scale_shift.contiguous()
    _check_shape(ss, "scale_shift", (B, 2 * Cc, H, W))
    out = torch.empty_like(xc)
    lib.check(lib.dll.ddif_film_fwd(_ptr(xc), _ptr(ss), B, Cc, H, W, _ptr(out), _stream(lib, xc.device)), "ddif_film_fwd")
    return out


def add(a, f, alpha=None):
    """a + alpha[b] * f (alpha None: plain residual add)"""
    lib = get_lib()
    _check_current_device(a, "a")
    a, f = a.contiguous(), f.contiguous()
    _check_shape(f, "f", tuple(a.shape))
    out = torch.empty_like(a)
    B = a.shape[0]
    lib.check(lib.dll.ddif_add_scaled(_ptr(a), _ptr(f), _ptr(None if alpha is None else alpha.contiguous()), B, a.numel() // B, _ptr(out),
                                      _stream(lib, a.device)), "ddif_add_scaled")
    return out


def linear(x, w, b=None):
    lib = get_lib()
    _check_current_device(x, "x")
    B, nin = x.shape
    nout = w.shape[0]
    x, w = x.contiguous(), w.contiguous()
    y = torch.empty((B, nout), dtype=torch.float32, device=x.device)
    lib.check(lib.dll.ddif_linear_fwd(_ptr(x), _ptr(w), _ptr(None if b is None else b.contiguous()), B, nin, nout, _ptr(y), _stream(lib, x.device)),
              "ddif_linear_fwd")
    return y


def selfattn_core(qkv, heads=8):
    lib = get_lib()
    _check_current_device(qkv, "qkv")
    B, C3, H, W = qkv.shape
    qkv = qkv.contiguous()
    out = torch.empty((B, C3 // 3, H, W), dtype=torch.float32, device=qkv.device)
    lib.check(lib.dll.ddif_selfattn_core_fwd(_ptr(qkv), B, C3 // 3, H, W, heads, _ptr(out), _stream(lib, qkv.device)), "ddif_selfattn_core_fwd")
    return out


def linattn_core(q_pre, kv_pre, heads=8):
    lib = get_lib()
    _check_current_device(q_pre, "q_pre")
    B, qd, H, W = q_pre.shape
    q_pre, kv_pre = q_pre.contiguous(), kv_pre.contiguous()
    _check_shape(kv_pre, "kv_pre", (B, 2 * qd, H, W))
    out = torch.empty_like(q_pre)
    lib.check(lib.dll.ddif_linattn_core_fwd(_ptr(q_pre), _ptr(kv_pre), B, qd, H, W, heads, _ptr(out), _stream(lib, q_pre.device)), "ddif_linattn_core_fwd")
    return out


def q_sample(x0, noise, a, s):
    """x_t = a[b] * x0 + s[b] * noise (reference diffusion/diffusion_ddpm_pan.py:668-681); a, s: (B,) tensors"""
    lib = get_lib()
    _check_current_device(x0, "x0")
    x0, noise = x0.contiguous(), noise.contiguous()
    a, s = a.to(x0.device, torch.float32).contiguous(), s.to(x0.device, torch.float32).contiguous()
    out = torch.empty_like(x0)
    B = x0.shape[0]
    lib.check(lib.dll.ddif_q_sample(_ptr(x0), _ptr(noise), _ptr(a), _ptr(s), B, x0.numel() // B, _ptr(out), _stream(lib, x0.device)), "ddif_q_sample")
    return out


def l1_loss(pred, target):
    """F.l1_loss(pred, target) (mean): a 0-d tensor on pred's device"""
    lib = get_lib()
    _check_current_device(pred, "pred")
    pred, target = pred.contiguous(), target.contiguous()
    _check_shape(target, "target", tuple(pred.shape))
    out = torch.empty((1,), dtype=torch.float32, device=pred.device)
    lib.check(lib.dll.ddif_l1_loss_fwd(_ptr(pred), _ptr(target), pred.numel(), _ptr(out), _stream(lib, pred.device)), "ddif_l1_loss_fwd")
    return out[0]
