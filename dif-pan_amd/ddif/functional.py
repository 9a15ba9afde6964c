"""Op-by-op forward AND backward of the denoiser's building blocks on libddif (C ABI: include/ddif.h, "forward ops of the
TRAINING graph" + the backward ops) -- what the op-by-op tape of the tests (tests/train_tape.py) strings together.  Signatures follow torch.nn.functional where the
reference uses it (models/sr3_dwt.py); tensors are torch fp32 NCHW on the GPU (CPU tensors only with the emulated test build).
No torch arithmetic in here: torch allocates, views (cat / chunk / pad are data movement) and nothing else."""
import ctypes as C

import torch

from . import runtime as R
from .runtime import _check_current_device, _check_shape, _check_tensor, _ptr, _stream, get_lib

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
        op = R.BlockBackward(B, ci, co, H, W, x_in.device, ks=ks, pro=pro, resample=resample)
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
