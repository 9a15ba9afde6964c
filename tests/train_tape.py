"""Training forward + backward of `UNetSR3` on libddif (reference: models/sr3_dwt.py:169-219 under `.train()`
and `loss.backward()`, diffusion_engine.py:230-233).

The inference plan (csrc/ddif_plan.cpp) fuses GroupNorm / SiLU / FiLM / softmax into conv prologues and epilogues and keeps no
intermediate; a training step needs them.  This module therefore walks the reference's module graph op by op through
`tests/ddif_testops.py` (every op is a C-ABI call into libddif: include/ddif.h "forward ops of the TRAINING graph" and the backward ops),
keeps what the backward pass needs, and then walks it in reverse -- a hand-written autograd tape for exactly this network.  Python
only orders the calls and owns the tensors (as the reference's Python does for torch ops); the arithmetic is in the library.
Correctness first: convs run on the exact-fp32 MFMA kernels with NCHW <-> NHWC conversion around each call; 1x1 convs ride the 3x3
kernels.  `cat` / `chunk` / zero-padding of channel axes are data movement done with torch views; the (B, 32) sinusoidal table of
PositionalEncoding and the bilinear resize of the (constant) cond image are input preparation done with torch.

    g = TrainGraph(cfg)                                    # cfg: ddif.layout.engine_cfg(...)
    y = g.forward(params, x, t, cond, self_cond, drop_masks=None, path_scales=None)   # params: dict key -> tensor (state_dict names)
    grads = g.backward(dy)                                 # dict key -> gradient, every learnable key of `params`
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as TF  # ONLY interpolate() on the constant cond image

import ddif_testops as F  # TEST SCAFFOLDING (tests/ddif_testops.py): the product trains through ddif_plan_train_step (csrc/ddif_train.cpp)
import ddif_testops as R  # (the backward ops lived in ddif/runtime.py until round 6)
from ddif import runtime as RT
from ddif.layout import layer_plan


def _acc(grads: Dict[str, torch.Tensor], key: str, g: torch.Tensor):
    grads[key] = g if key not in grads else F.add(grads[key], g)


class TrainGraph:
    def __init__(self, cfg: dict, dropout: float = 0.2, drop_path: float = 0.2):
        if cfg["norm_groups"] != 1:
            raise R.DdifError("TrainGraph: norm_groups = 1 only (the engine configuration)")
        self.cfg, self.plan = cfg, layer_plan(cfg)
        self.p_drop, self.p_path = dropout, drop_path
        self.tape: List = []

    # ------------------------------------------------------------------------------------------------ helpers
    def _mask(self, like: torch.Tensor) -> Optional[torch.Tensor]:
        """next Dropout mask (0 or 1/(1-p)): the caller's (parity) or a fresh Bernoulli draw from torch's generator, as nn.Dropout does"""
        if self.p_drop <= 0:
            return None
        if self._masks is not None:
            m = self._masks[self._mi].to(like.device)
            self._mi += 1
            return m
        return (torch.rand_like(like) >= self.p_drop).float() / (1.0 - self.p_drop)

    def _path(self, B: int, dev) -> Optional[torch.Tensor]:
        if self.p_path <= 0:
            return None
        if self._paths is not None:
            s = self._paths[self._pi].to(dev)
            self._pi += 1
            return s
        return (torch.rand(B, device=dev) >= self.p_path).float() / (1.0 - self.p_path)

    # ------------------------------------------------------------------------------------------------ modules (forward records a closure for the backward)
    def _time_embedding(self, P, t):
        """PositionalEncoding + noise_level_mlp (models/sr3_dwt.py:223-238, 59-64)"""
        inner = self.cfg["inner_channel"]
        count = inner // 2
        tt = t.to(torch.float32)
        step = torch.arange(count, dtype=torch.float32, device=tt.device) / count
        enc = tt.unsqueeze(1) * torch.exp(-math.log(1e4) * step.unsqueeze(0))
        enc = torch.cat([torch.sin(enc), torch.cos(enc)], dim=-1).contiguous()
        w1, b1, w3, b3 = (P["noise_level_mlp.%s" % k] for k in ("1.weight", "1.bias", "3.weight", "3.bias"))
        h1 = F.linear(enc, w1, b1)
        h2 = F.swish(h1)
        temb = F.linear(h2, w3, b3)

        def bwd(dtemb, G):
            dh2, dw3, db3 = R.linear_backward(h2, w3, dtemb)
            dh1 = R.swish_backward(h1, dh2)
            _, dw1, db1 = R.linear_backward(enc, w1, dh1)
            _acc(G, "noise_level_mlp.1.weight", dw1)
            _acc(G, "noise_level_mlp.1.bias", db1)
            _acc(G, "noise_level_mlp.3.weight", dw3)
            _acc(G, "noise_level_mlp.3.bias", db3)

        return temb, bwd

    def _resblock(self, P, p, x, temb):
        """ResnetBlock (models/sr3_dwt.py:303-327): block2(noise_func(block1(x), t)) + x; Dropout sits in block2"""
        B, Cc = x.shape[0], x.shape[1]
        g1, b1, w1, c1b = P[p + ".block1.block.0.weight"], P[p + ".block1.block.0.bias"], P[p + ".block1.block.3.weight"], P[p + ".block1.block.3.bias"]
        g2, b2, w2, c2b = P[p + ".block2.block.0.weight"], P[p + ".block2.block.0.bias"], P[p + ".block2.block.3.weight"], P[p + ".block2.block.3.bias"]
        wf, bf = P[p + ".noise_func.noise_func.0.weight"], P[p + ".noise_func.noise_func.0.bias"]
        if (p + ".res_conv.weight") in P:
            raise R.DdifError("TrainGraph: res_conv is Identity in the engine configuration")
        a1 = F.group_norm(x, g1, b1, silu=True)
        tb = F.linear(temb, wf, bf)  # (B, C): FeatureWiseAffine adds it per sample and channel
        # conv bias + time bias: the per-sample row rides as a 1x1 "bias image" through the residual add below
        c1 = F.conv2d(a1, w1, c1b)
        h = F.add(c1, tb.view(B, Cc, 1, 1).expand(B, Cc, x.shape[2], x.shape[3]).contiguous())
        mask = self._mask(h)
        a2 = F.group_norm(h, g2, b2, silu=True, mask=mask)
        out = F.add(F.conv2d(a2, w2, c2b), x)

        def bwd(dout, G, dtemb_list):
            r2 = F.conv2d_backward(h, w2, dout, pro="gn_silu", gamma=g2, beta=b2, mask=mask)
            _acc(G, p + ".block2.block.0.weight", r2["dgamma"])
            _acc(G, p + ".block2.block.0.bias", r2["dbeta"])
            _acc(G, p + ".block2.block.3.weight", r2["dw"])
            _acc(G, p + ".block2.block.3.bias", r2["db"])
            dh = r2["dx"]
            r1 = F.conv2d_backward(x, w1, dh, pro="gn_silu", gamma=g1, beta=b1)
            _acc(G, p + ".block1.block.0.weight", r1["dgamma"])
            _acc(G, p + ".block1.block.0.bias", r1["dbeta"])
            _acc(G, p + ".block1.block.3.weight", r1["dw"])
            _acc(G, p + ".block1.block.3.bias", r1["db"])
            dtb = r1["dy_plane_sums"]  # sum over pixels of dh = gradient of the time bias
            dtemb, dwf, dbf = R.linear_backward(temb, wf, dtb)
            _acc(G, p + ".noise_func.noise_func.0.weight", dwf)
            _acc(G, p + ".noise_func.noise_func.0.bias", dbf)
            dtemb_list.append(dtemb)
            return F.add(r1["dx"], dout)

        return out, bwd

    def _self_attention(self, P, p, x):
        """SelfAttention (models/sr3_dwt.py:330-360)"""
        gn, bn, wq, wo, bo = P[p + ".norm.weight"], P[p + ".norm.bias"], P[p + ".qkv.weight"], P[p + ".out.weight"], P[p + ".out.bias"]
        qkv = F.conv2d(F.group_norm(x, gn, bn), wq, None)
        o = F.selfattn_core(qkv)
        out = F.add(F.conv2d(o, wo, bo), x)

        def bwd(dout, G):
            ro = F.conv2d_backward(o, wo, dout, pro="none")
            _acc(G, p + ".out.weight", ro["dw"])
            _acc(G, p + ".out.bias", ro["db"])
            dqkv = R.selfattn_core_backward(qkv, ro["dx"])
            rq = F.conv2d_backward(x, wq, dqkv, pro="gn", gamma=gn, beta=bn)
            _acc(G, p + ".qkv.weight", rq["dw"])
            _acc(G, p + ".norm.weight", rq["dgamma"])
            _acc(G, p + ".norm.bias", rq["dbeta"])
            return F.add(rq["dx"], dout)

        return out, bwd

    def _cond_injection(self, P, p, x, cL):
        """encoder CondInjection, FiLM style (models/sr3_dwt.py:376-396)"""
        w0, g1, b1, w3, b3, wx, bx = (P[p + k] for k in (".body.0.weight", ".body.1.weight", ".body.1.bias", ".body.3.weight", ".body.3.bias", ".x_conv.weight",
                                                         ".x_conv.bias"))
        y0 = F.conv2d(cL, w0, None)
        ss = F.conv2d(F.group_norm(y0, g1, b1, silu=True), w3, b3)
        xc = F.conv2d(x, wx, bx)
        out = F.film(xc, ss)

        def bwd(dout, G):
            dxc, dss = R.film_backward(xc, ss, dout)
            rx = F.conv2d_backward(x, wx, dxc, pro="none")
            _acc(G, p + ".x_conv.weight", rx["dw"])
            _acc(G, p + ".x_conv.bias", rx["db"])
            r3 = F.conv2d_backward(y0, w3, dss, pro="gn_silu", gamma=g1, beta=b1)
            _acc(G, p + ".body.3.weight", r3["dw"])
            _acc(G, p + ".body.3.bias", r3["db"])
            _acc(G, p + ".body.1.weight", r3["dgamma"])
            _acc(G, p + ".body.1.bias", r3["dbeta"])
            r0 = F.conv2d_backward(cL, w0, r3["dx"], pro="none", need_dx=False)
            _acc(G, p + ".body.0.weight", r0["dw"])
            return rx["dx"]

        return out, bwd

    def _fast_attn(self, P, p, x, cL):
        """decoder FastAttnCondInjection (models/sr3_dwt.py:493-577) incl. DropPath on the FFN branch"""
        B = x.shape[0]
        gp, bp = P[p + ".prenorm_x.weight"], P[p + ".prenorm_x.bias"]
        wq0, wq1, bq1 = P[p + ".q.0.weight"], P[p + ".q.1.weight"], P[p + ".q.1.bias"]
        wk0, wk1, bk1 = P[p + ".kv.0.weight"], P[p + ".kv.1.weight"], P[p + ".kv.1.bias"]
        wao, bao = P[p + ".attn_out.weight"], P[p + ".attn_out.bias"]
        has_res = (p + ".attn_res.weight") in P
        wf0, wf2, wf3, bf3 = P[p + ".ffn.0.weight"], P[p + ".ffn.2.weight"], P[p + ".ffn.3.weight"], P[p + ".ffn.3.bias"]
        xn = F.group_norm(x, gp, bp)
        q0 = F.dwconv3x3(xn, wq0)
        q_pre = F.conv2d(q0, wq1, bq1)
        kv0 = F.dwconv3x3(cL, wk0)
        kv_pre = F.conv2d(kv0, wk1, bk1)
        o = F.linattn_core(q_pre, kv_pre)
        ao = F.conv2d(o, wao, bao)
        a = F.add(ao, F.conv2d(xn, P[p + ".attn_res.weight"], P[p + ".attn_res.bias"]) if has_res else xn)
        f0 = F.conv2d(a, wf0, None)
        f2 = F.conv2d(F.swish(f0), wf2, None)
        f3 = F.conv2d(f2, wf3, bf3)
        alpha = self._path(B, x.device)
        out = F.add(a, f3, alpha)

        def bwd(dout, G):
            df3 = dout if alpha is None else F.add(torch.zeros_like(dout), dout, alpha)
            r3 = F.conv2d_backward(f2, wf3, df3, pro="none")
            _acc(G, p + ".ffn.3.weight", r3["dw"])
            _acc(G, p + ".ffn.3.bias", r3["db"])
            r2 = F.conv2d_backward(f0, wf2, r3["dx"], pro="silu")
            _acc(G, p + ".ffn.2.weight", r2["dw"])
            r0 = F.conv2d_backward(a, wf0, r2["dx"], pro="none")
            _acc(G, p + ".ffn.0.weight", r0["dw"])
            da = F.add(dout, r0["dx"])
            rao = F.conv2d_backward(o, wao, da, pro="none")
            _acc(G, p + ".attn_out.weight", rao["dw"])
            _acc(G, p + ".attn_out.bias", rao["db"])
            if has_res:
                rar = F.conv2d_backward(xn, P[p + ".attn_res.weight"], da, pro="none")
                _acc(G, p + ".attn_res.weight", rar["dw"])
                _acc(G, p + ".attn_res.bias", rar["db"])
                dxn = rar["dx"]
            else:
                dxn = da
            dq_pre, dkv_pre = R.linattn_core_backward(q_pre, kv_pre, rao["dx"])
            rq1 = F.conv2d_backward(q0, wq1, dq_pre, pro="none")
            _acc(G, p + ".q.1.weight", rq1["dw"])
            _acc(G, p + ".q.1.bias", rq1["db"])
            dxn2, dwq0 = R.dwconv3x3_backward(xn, wq0, rq1["dx"])
            _acc(G, p + ".q.0.weight", dwq0)
            rk1 = F.conv2d_backward(kv0, wk1, dkv_pre, pro="none")
            _acc(G, p + ".kv.1.weight", rk1["dw"])
            _acc(G, p + ".kv.1.bias", rk1["db"])
            _, dwk0 = R.dwconv3x3_backward(cL, wk0, rk1["dx"])
            _acc(G, p + ".kv.0.weight", dwk0)
            dx, dgp, dbp = R.groupnorm_backward(x, gp, F.add(dxn, dxn2))
            _acc(G, p + ".prenorm_x.weight", dgp)
            _acc(G, p + ".prenorm_x.bias", dbp)
            return dx

        return out, bwd

    # ------------------------------------------------------------------------------------------------ network
    def forward(self, P: Dict[str, torch.Tensor], x, t, cond, self_cond=None, drop_masks=None, path_scales=None):
        """UNetSR3.forward under .train() (models/sr3_dwt.py:169-219, 658-673).  drop_masks / path_scales: the Dropout masks (0 or
        1/(1-p), execution order) and DropPath row scales (n_sites, B) to use instead of fresh draws (parity tests)."""
        cfg = self.cfg
        Cc, Pp = cfg["lms_channel"], cfg["pan_channel"]
        self._masks, self._mi = drop_masks, 0
        self._paths, self._pi = path_scales, 0
        tape = []  # closures, execution order
        if cfg["self_condition"]:
            sc = x if self_cond is None else self_cond
            x = torch.cat([sc, x], dim=1).contiguous()
        temb, temb_bwd = self._time_embedding(P, t)
        c_enc = cond[:, : Cc + Pp].contiguous()
        c_dec = cond[:, -(Cc + 3 * Pp):].contiguous()
        sized = {}

        def resize(c, hw):  # the cond image at a level's resolution, formed once per forward pass (28 uses, 4 sizes x 2 halves)
            key = (id(c), tuple(hw))
            if key not in sized:
                sized[key] = c if tuple(c.shape[-2:]) == tuple(hw) else TF.interpolate(c, size=tuple(hw), mode="bilinear").contiguous()
            return sized[key]

        feats = []
        for i, L in enumerate(self.plan["downs"]):
            p = f"downs.{i}"
            if L["kind"] == "stem":
                xin, w, b = x, P[p + ".weight"], P[p + ".bias"]
                x = F.conv2d(xin, w, b)
                tape.append(("conv", p, (xin, w, 1, False, False)))
            elif L["kind"] == "down":
                xin, w, b = x, P[p + ".conv.weight"], P[p + ".conv.bias"]
                x = F.conv2d(xin, w, b, stride=2)
                tape.append(("conv", p + ".conv", (xin, w, 2, False, True)))
            else:
                x, b1 = self._cond_injection(P, p + ".cond_inj", x, resize(c_enc, x.shape[-2:]))
                tape.append(("mod", b1))
                x, b2 = self._resblock(P, p + ".res_block", x, temb)
                tape.append(("res", b2))
                if L["attn"]:
                    x, b3 = self._self_attention(P, p + ".attn", x)
                    tape.append(("mod", b3))
            feats.append(x)
            tape.append(("push",))
        for i, L in enumerate(self.plan["mid"]):
            p = f"mid.{i}"
            x, b2 = self._resblock(P, p + ".res_block", x, temb)
            tape.append(("res", b2))
            if L["attn"]:
                x, b3 = self._self_attention(P, p + ".attn", x)
                tape.append(("mod", b3))
        for i, L in enumerate(self.plan["ups"]):
            p = f"ups.{i}"
            if L["kind"] == "up":
                xin, w, b = x, P[p + ".conv.weight"], P[p + ".conv.bias"]
                x = F.conv2d(xin, w, b, up2=True)
                tape.append(("conv", p + ".conv", (xin, w, 1, True, True)))
            else:
                skip = feats.pop()
                tape.append(("cat", x.shape[1]))
                x = torch.cat([x, skip], dim=1).contiguous()
                x, b1 = self._fast_attn(P, p + ".cond_inj", x, resize(c_dec, x.shape[-2:]))
                tape.append(("mod", b1))
                x, b2 = self._resblock(P, p + ".res_block", x, temb)
                tape.append(("res", b2))
                if L["attn"]:
                    x, b3 = self._self_attention(P, p + ".attn", x)
                    tape.append(("mod", b3))
        gf, bf, wf, cf = P["final_conv.block.0.weight"], P["final_conv.block.0.bias"], P["final_conv.block.3.weight"], P["final_conv.block.3.bias"]
        xin = x
        out = F.conv2d(F.group_norm(xin, gf, bf, silu=True), wf, cf)
        tape.append(("final", (xin, gf, bf, wf)))
        self.tape, self._temb_bwd = tape, temb_bwd
        return out

    def backward(self, dout: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Gradients of every parameter given d(loss)/d(output) -- loss.backward() of the reference (diffusion_engine.py:233)."""
        G: Dict[str, torch.Tensor] = {}
        dtembs: List[torch.Tensor] = []
        skip_grads: List[torch.Tensor] = []  # gradients flowing into the encoder features through the decoder's cat (a stack, like feats)
        dx = dout
        for ent in reversed(self.tape):
            kind = ent[0]
            if kind == "final":
                xin, gf, bf, wf = ent[1]
                r = F.conv2d_backward(xin, wf, dx, pro="gn_silu", gamma=gf, beta=bf)
                _acc(G, "final_conv.block.0.weight", r["dgamma"])
                _acc(G, "final_conv.block.0.bias", r["dbeta"])
                _acc(G, "final_conv.block.3.weight", r["dw"])
                _acc(G, "final_conv.block.3.bias", r["db"])
                dx = r["dx"]
            elif kind == "mod":
                dx = ent[1](dx, G)
            elif kind == "res":
                dx = ent[1](dx, G, dtembs)
            elif kind == "cat":
                cx = ent[1]
                skip_grads.append(dx[:, cx:].contiguous())
                dx = dx[:, :cx].contiguous()
            elif kind == "push":  # this point's activation was also a skip connection: add what came back through the decoder
                dx = F.add(dx, self._pop_skip(skip_grads))
            elif kind == "conv":
                p, (xin, w, stride, up2, has_dx) = ent[1], ent[2]
                r = F.conv2d_backward(xin, w, dx, pro="none", stride=stride, up2=up2, need_dx=has_dx)
                _acc(G, p + ".weight", r["dw"])
                _acc(G, p + ".bias", r["db"])
                dx = r["dx"]
        dtemb = dtembs[0]
        for d in dtembs[1:]:
            dtemb = F.add(dtemb, d)
        self._temb_bwd(dtemb, G)
        self.tape = []
        return G

    @staticmethod
    def _pop_skip(stack: List[torch.Tensor]) -> torch.Tensor:
        # the decoder pops the encoder features last-in-first-out, so walking the tape backwards meets the "push" points in the
        # order in which their gradients were appended: first appended = last decoder cat = FIRST encoder feature ... i.e. the
        # gradient for the push met first (the deepest encoder feature) is the one appended LAST among those still pending
        return stack.pop()


class TrainStepFn(torch.autograd.Function):
    """Bridges the library's training step into torch autograd so that the reference's training loop runs unchanged:
    `loss, recon = diffusion(x, cond=cond); loss.backward(); clip; optimizer.step()` (diffusion_engine.py:230-241).  The parameters are
    the differentiable inputs; forward = TrainGraph.forward + L1 loss, backward = TrainGraph.backward -> one gradient per parameter."""

    @staticmethod
    def forward(ctx, graph, names, x_noisy, t, cond, self_cond, target, drop_masks, path_scales, *params):
        P = {n: p.detach() for n, p in zip(names, params)}
        y = graph.forward(P, x_noisy, t, cond, self_cond, drop_masks=drop_masks, path_scales=path_scales)
        loss = F.l1_loss(y, target)
        ctx.graph, ctx.names, ctx.y, ctx.target = graph, names, y, target
        ctx.mark_non_differentiable(y)
        return loss, y

    @staticmethod
    def backward(ctx, gloss, _grecon):
        dy = R.l1_loss_backward(ctx.y, ctx.target, upstream=float(gloss))
        G = ctx.graph.backward(dy)
        return (None,) * 9 + tuple(G.get(n) for n in ctx.names)


def tape_train_step(diffusion, x_start, noise, a, s, t, cond, x_self_cond):
    """`GaussianDiffusion._train_step` on the op-by-op tape instead of the native step (same signature, same (loss, pred) result with an autograd
    node behind `loss`): the cross-check test patches this in; the product package never imports it."""
    model = diffusion.model
    named = [(n, p) for n, p in model.named_parameters()]
    graph = TrainGraph(model.cfg, dropout=float(model.cfg["dropout"]), drop_path=model.DROP_PATH_PROB)  # one tape per call (gradient accumulation safe)
    names = tuple(n for n, _ in named)
    x_noisy = F.q_sample(x_start, noise, a, s)
    pinned = getattr(model, "_train_masks", None)
    drop_masks, path_scales = (None, None)
    if pinned is not None:
        drop_masks, paths = pinned
        path_scales = None if paths is None else [paths[k] for k in range(paths.shape[0])]
    loss, pred = TrainStepFn.apply(graph, names, x_noisy, t, cond, x_self_cond, x_start, drop_masks, path_scales, *[p for _, p in named])
    return loss, pred
