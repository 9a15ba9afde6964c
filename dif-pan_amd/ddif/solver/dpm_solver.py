"""Drop-in for the part of the reference `solver/dpm_solver.py` (vendored DPM-Solver v2) that the DDIF hot path
reaches: `NoiseScheduleVP('discrete')` (:6-175), `model_wrapper` (:178-342) and `DPM_Solver.sample(method="multistep")`
(:345-459, 555-588, 804-912, 1055-1253).

Scalar schedule math stays on the host as fp32 torch scalars (the reference evaluates the same expressions as 0-d /
1-element fp32 tensors); the tensor work of a whole sampling run -- every network evaluation, the x_start<->eps
round trip, the corrector clamp and the multistep updates -- is ONE call into libddif (`ddif_plan_sample_dpmpp`)
when the model is a ddif `UNetSR3`.  Any other model / corrector takes the generic loop below (same algorithm, model
called per evaluation).  Out of scope and rejected loudly: continuous schedules, singlestep / adaptive solvers,
classifier guidance, dynamic thresholding.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch

from ..runtime import DdifError


def interpolate_fn(x: torch.Tensor, xp: torch.Tensor, yp: torch.Tensor) -> torch.Tensor:
    """Piecewise-linear y(x) through the keypoints (xp, yp) with linear extrapolation beyond both ends
    (reference :1261-1300).  x: [N, C], xp / yp: [C, K] -> [N, C].  A query equal to a keypoint takes the segment
    on its LEFT (the reference sorts the query in front of the equal keypoint)."""
    N, K = x.shape[0], xp.shape[1]
    out = torch.empty_like(x)
    for c in range(x.shape[1]):
        xs, ys = xp[c], yp[c]
        below = (xs.unsqueeze(0) < x[:, c].unsqueeze(1)).sum(dim=1)  # keypoints strictly below each query
        lo = torch.where(below == 0, torch.zeros_like(below), torch.where(below == K, torch.full_like(below, K - 2), below - 1))
        x0, x1, y0, y1 = xs[lo], xs[lo + 1], ys[lo], ys[lo + 1]
        out[:, c] = y0 + (x[:, c] - x0) * (y1 - y0) / (x1 - x0)
    return out


class NoiseScheduleVP:
    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20.0, dtype=torch.float32):
        if schedule != "discrete":
            raise DdifError(f"NoiseScheduleVP(schedule='{schedule}'): only 'discrete' is in scope of the HIP path")
        self.schedule = schedule
        if betas is not None:
            log_alphas = 0.5 * torch.log(1 - betas.detach().cpu()).cumsum(dim=0)
        else:
            assert alphas_cumprod is not None
            log_alphas = 0.5 * torch.log(alphas_cumprod.detach().cpu())
        self.total_N = len(log_alphas)
        self.T = 1.0
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].reshape((1, -1)).to(dtype=dtype)
        self.log_alpha_array = log_alphas.reshape((1, -1)).to(dtype=dtype)

    def marginal_log_mean_coeff(self, t):
        t = t.detach().cpu()
        return interpolate_fn(t.reshape((-1, 1)), self.t_array, self.log_alpha_array).reshape((-1))

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        la = self.marginal_log_mean_coeff(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))

    def inverse_lambda(self, lamb):
        lamb = lamb.detach().cpu()
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,)), -2.0 * lamb)
        t = interpolate_fn(log_alpha.reshape((-1, 1)), torch.flip(self.log_alpha_array, [1]), torch.flip(self.t_array, [1]))
        return t.reshape((-1,))


class ImageSpaceClamp:
    """The x0 corrector the engine uses: clamp in image space around the up-sampled LMS,
    x0 <- clamp(x0 + lms, lo, hi) - lms  (reference diffusion_engine.py:43-49).  A recognisable object (rather than
    an opaque closure) so `DPM_Solver` can run it inside the fused kernels."""

    def __init__(self, lms: torch.Tensor, lo: float = 0.0, hi: float = 1.0):
        self.lms, self.lo, self.hi = lms, float(lo), float(hi)

    def __call__(self, x0, t=None):
        return (x0 + self.lms).clamp(self.lo, self.hi) - self.lms


def model_wrapper(model, noise_schedule, model_type="noise", model_kwargs={}, guidance_type="uncond", condition=None,
                  unconditional_condition=None, guidance_scale=1.0, classifier_fn=None, classifier_kwargs={}):
    """Reference :178-342.  Returns model_fn(x, t_continuous) -> predicted noise."""
    assert model_type in ["noise", "x_start", "v", "score"]
    assert guidance_type in ["uncond", "classifier", "classifier-free"]
    if guidance_type == "classifier":
        raise DdifError("classifier guidance is out of scope of the HIP path")
    if guidance_type == "classifier-free" and not (guidance_scale == 1.0 or unconditional_condition is None):
        raise DdifError("classifier-free guidance with scale != 1 is out of scope of the HIP path")

    def get_model_input_time(t_continuous):
        return (t_continuous - 1.0 / noise_schedule.total_N) * 1000.0  # float model time (:285-286)

    def bshape(v, x):
        return v.to(x.device).reshape((-1,) + (1,) * (x.dim() - 1))  # broadcasts for any B (the reference: B==1 only)

    def noise_pred_fn(x, t_continuous, cond=None):
        t_input = get_model_input_time(t_continuous)
        out = model(x, t_input, **model_kwargs) if cond is None else model(x, t_input, cond, **model_kwargs)
        if model_type == "noise":
            return out
        a, s = noise_schedule.marginal_alpha(t_continuous), noise_schedule.marginal_std(t_continuous)
        if model_type == "x_start":
            return (x - bshape(a, x) * out) / bshape(s, x)
        if model_type == "v":
            return bshape(a, x) * out + bshape(s, x) * x
        return -bshape(s, x) * out

    def model_fn(x, t_continuous):
        return noise_pred_fn(x, t_continuous, cond=condition if guidance_type == "classifier-free" else None)

    model_fn.ddif = dict(model=model, noise_schedule=noise_schedule, model_type=model_type, model_kwargs=model_kwargs,
                         guidance_type=guidance_type, condition=condition)
    return model_fn


class DPM_Solver:
    def __init__(self, model_fn, noise_schedule, algorithm_type="dpmsolver++", correcting_x0_fn=None,
                 correcting_xt_fn=None, thresholding_max_val=1.0, dynamic_thresholding_ratio=0.995):
        assert algorithm_type in ["dpmsolver", "dpmsolver++"]
        if correcting_x0_fn == "dynamic_thresholding":
            raise DdifError("dynamic thresholding is out of scope of the HIP path")
        self._model_fn = model_fn
        self.model = lambda x, t: model_fn(x, t.expand((x.shape[0])))
        self.noise_schedule = noise_schedule
        self.algorithm_type = algorithm_type
        self.correcting_x0_fn = correcting_x0_fn
        self.correcting_xt_fn = correcting_xt_fn
        self.dynamic_thresholding_ratio = dynamic_thresholding_ratio
        self.thresholding_max_val = thresholding_max_val

    # ---- reference helpers kept for API compatibility ---------------------------------------------------------------
    def noise_prediction_fn(self, x, t):
        return self.model(x, t)

    def data_prediction_fn(self, x, t):
        noise = self.noise_prediction_fn(x, t)
        a, s = self.noise_schedule.marginal_alpha(t), self.noise_schedule.marginal_std(t)
        a, s = a.to(x.device).reshape(-1, *([1] * (x.dim() - 1))), s.to(x.device).reshape(-1, *([1] * (x.dim() - 1)))
        x0 = (x - s * noise) / a
        if self.correcting_x0_fn is not None:
            x0 = self.correcting_x0_fn(x0, t)
        return x0

    def model_fn(self, x, t):
        return self.data_prediction_fn(x, t) if self.algorithm_type == "dpmsolver++" else self.noise_prediction_fn(x, t)

    def get_time_steps(self, skip_type, t_T, t_0, N, device=None):
        if skip_type == "logSNR":
            lt = self.noise_schedule.marginal_lambda(torch.tensor(t_T))
            l0 = self.noise_schedule.marginal_lambda(torch.tensor(t_0))
            return self.noise_schedule.inverse_lambda(torch.linspace(lt.item(), l0.item(), N + 1))
        if skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        if skip_type == "time_quadratic":
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
        raise ValueError(f"Unsupported skip_type {skip_type}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'")

    # ---- per-update scalars (reference :555-588, 804-845, 862-912), fp32 like the reference ----------------------------
    def _update_coefs(self, tprev: List[torch.Tensor], t: torch.Tensor, order: int) -> dict:
        ns = self.noise_schedule
        lam_t, lam_0 = ns.marginal_lambda(t), ns.marginal_lambda(tprev[-1])
        h = lam_t - lam_0
        a_t = torch.exp(ns.marginal_log_mean_coeff(t))
        s_t, s_0 = ns.marginal_std(t), ns.marginal_std(tprev[-1])
        phi1 = torch.expm1(-h)
        c = dict(cx=float(s_t / s_0), a_phi1=float(a_t * phi1), inv_r0=0.0, inv_r1=0.0, r0_frac=0.0, inv_r01=0.0,
                 a_phi2=0.0, a_phi3=0.0)
        if order >= 2:
            h0 = lam_0 - ns.marginal_lambda(tprev[-2])
            r0 = h0 / h
            c["inv_r0"] = float(1.0 / r0)
        if order == 3:
            h1 = ns.marginal_lambda(tprev[-2]) - ns.marginal_lambda(tprev[-3])
            r1 = h1 / h
            phi2 = phi1 / h + 1.0
            phi3 = phi2 / h - 0.5
            c.update(inv_r1=float(1.0 / r1), r0_frac=float(r0 / (r0 + r1)), inv_r01=float(1.0 / (r0 + r1)),
                     a_phi2=float(a_t * phi2), a_phi3=float(a_t * phi3))
        return c

    @staticmethod
    def _apply_update(x, models, c, order):
        m0 = models[-1]
        v = c["cx"] * x - c["a_phi1"] * m0
        if order == 2:
            d1 = c["inv_r0"] * (m0 - models[-2])
            v = v - (0.5 * c["a_phi1"]) * d1
        elif order == 3:
            d10 = c["inv_r0"] * (m0 - models[-2])
            d11 = c["inv_r1"] * (models[-2] - models[-3])
            d1 = d10 + c["r0_frac"] * (d10 - d11)
            d2 = c["inv_r01"] * (d10 - d11)
            v = (v + c["a_phi2"] * d1) - c["a_phi3"] * d2
        return v

    def _fused_target(self):
        """(UNetSR3, cond, clamp) when the whole run can execute inside libddif, else None."""
        meta = getattr(self._model_fn, "ddif", None)
        if meta is None or self.algorithm_type != "dpmsolver++" or self.correcting_xt_fn is not None:
            return None
        from ..models.sr3_dwt import UNetSR3

        if not isinstance(meta["model"], UNetSR3) or meta["model_type"] != "x_start" or meta["model_kwargs"]:
            return None
        if meta["guidance_type"] != "classifier-free" or meta["condition"] is None:
            return None
        cx0 = self.correcting_x0_fn
        if cx0 is None:
            return meta["model"], meta["condition"], None
        if isinstance(cx0, ImageSpaceClamp):
            cond, C = meta["condition"], meta["model"].cfg["out_channel"]
            if cx0.lms.shape == cond[:, :C].shape and cx0.lms.data_ptr() == cond[:, :C].data_ptr() or \
                    torch.equal(cx0.lms, cond[:, :C]):
                return meta["model"], cond, (cx0.lo, cx0.hi)
        return None

    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
               return_intermediate=False):
        if method != "multistep":
            raise DdifError(f"DPM_Solver.sample(method='{method}'): only 'multistep' is in scope of the HIP path")
        if solver_type != "dpmsolver":
            raise DdifError("solver_type='taylor' is out of scope of the HIP path")
        if self.algorithm_type != "dpmsolver++":
            raise DdifError("algorithm_type='dpmsolver' (noise prediction updates) is out of scope of the HIP path")
        ns = self.noise_schedule
        t_0 = 1.0 / ns.total_N if t_end is None else t_end
        t_T = ns.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0 and steps >= order and 1 <= order <= 3
        ts = self.get_time_steps(skip_type, t_T, t_0, steps)
        assert ts.shape[0] - 1 == steps
        ords = []
        for k in range(steps):
            step = k + 1
            if step < order:
                ords.append(step)
            elif lower_order_final and steps < 10:
                ords.append(min(order, steps + 1 - step))
            else:
                ords.append(order)
        coefs = [self._update_coefs([ts[j].reshape(1) for j in range(max(0, k - 2), k + 1)], ts[k + 1].reshape(1), ords[k])
                 for k in range(steps)]
        fused = self._fused_target()
        if fused is not None and not return_intermediate and not denoise_to_zero:
            model, cond, clamp = fused
            B, _, H, W = x.shape
            plan = model.plan_for(B, H, W, x.device)
            plan.set_cond(cond)
            tabs = dict(n_evals=steps, order=order,
                        t_model=[float((ts[k] - 1.0 / ns.total_N) * 1000.0) for k in range(steps)],
                        alpha=[float(ns.marginal_alpha(ts[k].reshape(1))) for k in range(steps)],
                        sigma=[float(ns.marginal_std(ts[k].reshape(1))) for k in range(steps)],
                        ord=ords, cx=[c["cx"] for c in coefs], a_phi1=[c["a_phi1"] for c in coefs],
                        inv_r0=[c["inv_r0"] for c in coefs], inv_r1=[c["inv_r1"] for c in coefs],
                        r0_frac=[c["r0_frac"] for c in coefs], inv_r01=[c["inv_r01"] for c in coefs],
                        a_phi2=[c["a_phi2"] for c in coefs], a_phi3=[c["a_phi3"] for c in coefs])
            with torch.no_grad():
                return plan.sample_dpmpp(tabs, x, clamp)
        # generic loop: same algorithm, the model is called once per evaluation (reference :1179-1221)
        inter = []
        with torch.no_grad():
            models: List[torch.Tensor] = []
            for k in range(steps):
                models = (models + [self.model_fn(x, ts[k].reshape(1))])[-3:]
                x = self._apply_update(x, models, coefs[k], ords[k])
                if self.correcting_xt_fn is not None:
                    x = self.correcting_xt_fn(x, ts[k + 1].reshape(1), k + 1)
                inter.append(x)
            if denoise_to_zero:
                x = self.data_prediction_fn(x, torch.ones((1,)) * t_0)
                inter.append(x)
        return (x, inter) if return_intermediate else x
