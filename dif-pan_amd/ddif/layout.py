"""Structure of the DDIF denoiser: which layers exist for a constructor configuration and what
parameters (reference state-dict key -> shape) they own.

Mirrors the constructor logic of the reference `UNetSR3.__init__` (models/sr3_dwt.py:31-167) and the
checkpoint key layout of SURVEY.md appendix C; `tests/golden/manifest_*.json` (captured from the reference's
own `state_dict()`) pins it.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

DEFAULT_CFG = dict(  # defaults of the reference constructor (models/sr3_dwt.py:33-50)
    in_channel=8, out_channel=3, inner_channel=32, lms_channel=8, pan_channel=1, norm_groups=32,
    channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3, dropout=0, with_noise_level_emb=True,
    image_size=128, self_condition=False, fourier_features=False, fourier_min=7, fourier_max=8,
    fourier_step=1, pred_var=False,
)

N_HEADS = 8  # SelfAttention n_head and FastAttnCondInjection nheads (models/sr3_dwt.py:641,654)


def engine_cfg(channels: int = 8, pan: int = 1, **over) -> dict:
    """Configuration diffusion_engine.py builds for every dataset (diffusion_engine.py:121-133,381-393)."""
    cfg = dict(DEFAULT_CFG)
    cfg.update(in_channel=channels, out_channel=channels, lms_channel=channels, pan_channel=pan, inner_channel=32,
               norm_groups=1, channel_mults=(1, 2, 2, 4), attn_res=(8,), dropout=0.2, image_size=64,
               self_condition=True)
    cfg.update(over)
    return cfg


def layer_plan(cfg: dict) -> dict:
    inner = cfg["inner_channel"]
    mults = tuple(cfg["channel_mults"])
    cin0 = cfg["in_channel"] + (cfg["out_channel"] if cfg["self_condition"] else 0)
    pre, res = inner, cfg["image_size"]
    skip_stack = [pre]
    downs: List[dict] = [dict(kind="stem", cin=cin0, cout=inner)]
    for i, m in enumerate(mults):
        ch = inner * m
        attn = res in cfg["attn_res"]
        for _ in range(cfg["res_blocks"]):
            downs.append(dict(kind="enc", cin=pre, cout=ch, attn=attn))
            skip_stack.append(ch)
            pre = ch
        if i != len(mults) - 1:
            downs.append(dict(kind="down", cin=pre, cout=pre))
            skip_stack.append(pre)
            res //= 2
    mid = [dict(kind="mid", cin=pre, cout=pre, attn=True), dict(kind="mid", cin=pre, cout=pre, attn=False)]
    ups: List[dict] = []
    for i in reversed(range(len(mults))):
        ch = inner * mults[i]
        attn = res in cfg["attn_res"]
        for _ in range(cfg["res_blocks"] + 1):
            sk = skip_stack.pop()
            ups.append(dict(kind="dec", cin=pre + sk, cx=pre, cskip=sk, cout=ch, attn=attn))
            pre = ch
        if i >= 1:
            ups.append(dict(kind="up", cin=pre, cout=pre))
            res *= 2
    return dict(downs=downs, mid=mid, ups=ups, final_in=pre)


def param_manifest(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list identical to the reference module's `state_dict()` order."""
    inner = cfg["inner_channel"]
    C, P = cfg["lms_channel"], cfg["pan_channel"]
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def add(k, *shape):
        out.append((k, tuple(int(s) for s in shape)))

    def resblock(p, c):
        add(p + ".noise_func.noise_func.0.weight", c, inner)
        add(p + ".noise_func.noise_func.0.bias", c)
        for b in ("block1", "block2"):
            add(f"{p}.{b}.block.0.weight", c)
            add(f"{p}.{b}.block.0.bias", c)
            add(f"{p}.{b}.block.3.weight", c, c, 3, 3)
            add(f"{p}.{b}.block.3.bias", c)

    def attn(p, c):
        add(p + ".norm.weight", c)
        add(p + ".norm.bias", c)
        add(p + ".qkv.weight", 3 * c, c, 1, 1)
        add(p + ".out.weight", c, c, 1, 1)
        add(p + ".out.bias", c)

    # registration order inside ResnetBlocWithAttn: res_block, attn, cond_inj (models/sr3_dwt.py:633-656)
    def enc_inj(p, cin, c):
        cd = C + P
        add(p + ".body.0.weight", 4 * c, cd, 3, 3)
        add(p + ".body.1.weight", 4 * c)
        add(p + ".body.1.bias", 4 * c)
        add(p + ".body.3.weight", 2 * c, 4 * c, 1, 1)
        add(p + ".body.3.bias", 2 * c)
        add(p + ".x_conv.weight", c, cin, 1, 1)
        add(p + ".x_conv.bias", c)

    def dec_inj(p, fea, c):
        cd = C + 3 * P
        add(p + ".prenorm_x.weight", fea)
        add(p + ".prenorm_x.bias", fea)
        add(p + ".q.0.weight", fea, 1, 3, 3)
        add(p + ".q.1.weight", fea, fea, 1, 1)
        add(p + ".q.1.bias", fea)
        add(p + ".kv.0.weight", cd, 1, 3, 3)
        add(p + ".kv.1.weight", 2 * fea, cd, 1, 1)
        add(p + ".kv.1.bias", 2 * fea)
        add(p + ".attn_out.weight", c, fea, 1, 1)
        add(p + ".attn_out.bias", c)
        if fea != c:
            add(p + ".attn_res.weight", c, fea, 1, 1)
            add(p + ".attn_res.bias", c)
        add(p + ".ffn.0.weight", 2 * c, c, 3, 3)
        add(p + ".ffn.2.weight", c, 2 * c, 3, 3)
        add(p + ".ffn.3.weight", c, c, 1, 1)
        add(p + ".ffn.3.bias", c)

    add("noise_level_mlp.1.weight", 4 * inner, inner)
    add("noise_level_mlp.1.bias", 4 * inner)
    add("noise_level_mlp.3.weight", inner, 4 * inner)
    add("noise_level_mlp.3.bias", inner)
    plan = layer_plan(cfg)
    for i, L in enumerate(plan["downs"]):
        p = f"downs.{i}"
        if L["kind"] == "stem":
            add(p + ".weight", L["cout"], L["cin"], 3, 3)
            add(p + ".bias", L["cout"])
        elif L["kind"] == "down":
            add(p + ".conv.weight", L["cout"], L["cin"], 3, 3)
            add(p + ".conv.bias", L["cout"])
        else:
            resblock(p + ".res_block", L["cout"])
            if L["attn"]:
                attn(p + ".attn", L["cout"])
            enc_inj(p + ".cond_inj", L["cin"], L["cout"])
    for i, L in enumerate(plan["mid"]):
        p = f"mid.{i}"
        resblock(p + ".res_block", L["cout"])
        if L["attn"]:
            attn(p + ".attn", L["cout"])
    for i, L in enumerate(plan["ups"]):
        p = f"ups.{i}"
        if L["kind"] == "up":
            add(p + ".conv.weight", L["cout"], L["cin"], 3, 3)
            add(p + ".conv.bias", L["cout"])
        else:
            resblock(p + ".res_block", L["cout"])
            if L["attn"]:
                attn(p + ".attn", L["cout"])
            dec_inj(p + ".cond_inj", L["cin"], L["cout"])
    fin = plan["final_in"]
    add("final_conv.block.0.weight", fin)
    add("final_conv.block.0.bias", fin)
    add("final_conv.block.3.weight", cfg["out_channel"], fin, 3, 3)
    add("final_conv.block.3.bias", cfg["out_channel"])
    return out


def time_bias_slots(cfg: dict) -> List[Tuple[str, int]]:
    """(res_block prefix, channels) for every FeatureWiseAffine, in network order."""
    plan = layer_plan(cfg)
    slots = []
    for grp in ("downs", "mid", "ups"):
        for i, L in enumerate(plan[grp]):
            if L["kind"] in ("enc", "mid", "dec"):
                slots.append((f"{grp}.{i}.res_block", L["cout"]))
    return slots
