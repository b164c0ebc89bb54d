"""ORACLE (test infrastructure) — CPU restatement of the reference U-Net.

Functional, dictionary-of-tensors form of `SimpleUnet` (reference gms/diffusion/simple_unet.py:16-72)
written with plain torch-CPU ops.  Parameter names/shapes are the reference's state-dict keys
 so a reference checkpoint can be fed in unchanged.

Extension beyond the reference (no reference oracle, "parity unpinned" for it): `in_channels` != 1
generalises the stem (`down.seq.0.conv`) and head (`out.2`) convolutions, which the reference
hard-codes to one channel (simple_unet.py:93,41).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

MAX_TIMESTEPS = 256  # simple_unet.py:13 — a frequency constant, not a cap on T


# --------------------------------------------------------------------------------------
# parameter inventory (reference state-dict order, simple_unet.py:17-42,87-102,125-144,155-179)
# --------------------------------------------------------------------------------------
def _res_spec(prefix, cin, cout, emb):
    spec = [
        (f"{prefix}.in_layers.0.weight", (cin,)),
        (f"{prefix}.in_layers.0.bias", (cin,)),
        (f"{prefix}.in_layers.2.weight", (cout, cin, 3, 3)),
        (f"{prefix}.in_layers.2.bias", (cout,)),
        (f"{prefix}.emb_layers.1.weight", (cout, emb)),
        (f"{prefix}.emb_layers.1.bias", (cout,)),
        (f"{prefix}.out_layers.0.weight", (cout,)),
        (f"{prefix}.out_layers.0.bias", (cout,)),
        (f"{prefix}.out_layers.3.weight", (cout, cout, 3, 3)),
        (f"{prefix}.out_layers.3.bias", (cout,)),
    ]
    if cin != cout:
        spec += [
            (f"{prefix}.skip_connection.weight", (cout, cin, 1, 1)),
            (f"{prefix}.skip_connection.bias", (cout,)),
        ]
    return spec


ATTN_SPEC = lambda C: [("attn.norm.weight", (C,)), ("attn.norm.bias", (C,)), ("attn.qkv.weight", (3 * C, C, 1, 1)),
                       ("attn.qkv.bias", (3 * C,)), ("attn.proj.weight", (C, C, 1, 1)), ("attn.proj.bias", (C,))]


def param_spec(channels, in_channels=1, attention=False):
    """[(name, shape)] in the order `SimpleUnet(channels, p).state_dict()` yields them.  attention=True appends the
    parameters of the self-attention extension (no counterpart in the reference) behind `turn`."""
    C, E = channels, 2 * channels
    spec = []
    for name, fan_in in (("time_embed", 64), ("cond_w_embed", 64), ("guide_embed", 10)):
        spec += [
            (f"{name}.0.weight", (E, fan_in)),
            (f"{name}.0.bias", (E,)),
            (f"{name}.2.weight", (E, E)),
            (f"{name}.2.bias", (E,)),
        ]
    spec += [("down.seq.0.conv.weight", (C, in_channels, 3, 3)), ("down.seq.0.conv.bias", (C,))]
    for i in (1, 2):
        spec += _res_spec(f"down.seq.{i}", C, C, E)
    spec += [("down.seq.3.conv.weight", (C, C, 3, 3)), ("down.seq.3.conv.bias", (C,))]
    for i in (4, 5):
        spec += _res_spec(f"down.seq.{i}", C, C, E)
    spec += [("down.seq.6.conv.weight", (C, C, 3, 3)), ("down.seq.6.conv.bias", (C,))]
    spec += _res_spec("turn", C, C, E)
    if attention:
        spec += ATTN_SPEC(C)
    for i in range(7):
        if i in (0, 3):
            spec += _res_spec(f"up.seq.{i}.0", 2 * C, C, E)
            spec += [(f"up.seq.{i}.1.conv.weight", (C, C, 3, 3)), (f"up.seq.{i}.1.conv.bias", (C,))]
        else:
            spec += _res_spec(f"up.seq.{i}", 2 * C, C, E)
    spec += [
        ("out.0.weight", (C,)),
        ("out.0.bias", (C,)),
        ("out.2.weight", (in_channels, C, 3, 3)),
        ("out.2.bias", (in_channels,)),
    ]
    return spec


def closed_form_params(channels, in_channels=1, dtype=torch.float32, attention=False):
    """Deterministic parameter fill both the golden generator and every test can regenerate.

    Every tensor gets a distinct phase; the reference's zero-initialised `out_layers.3`
    convolutions (simple_unet.py:172,198-202) are filled too — zeros would hide bugs in half
    the convolutions.  Matrices/filters: amp*sin(0.37*k + 1.3*id), amp = 1.4/sqrt(fan_in);
    GroupNorm weights: 1 + 0.1*sin; biases: 0.1*sin.
    """
    params = OrderedDict()
    for tid, (name, shape) in enumerate(param_spec(channels, in_channels, attention)):
        n = 1
        for s in shape:
            n *= s
        k = torch.arange(n, dtype=torch.float64)
        wave = torch.sin(0.37 * k + 1.3 * tid)
        if len(shape) >= 2:
            fan_in = n // shape[0]
            v = (1.4 / math.sqrt(fan_in)) * wave
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * wave
        else:
            v = 0.1 * wave
        params[name] = v.reshape(shape).to(dtype)
    return params


def reference_init_params(channels, in_channels=1, seed=0, dtype=torch.float32, zero_out_layers=True, attention=False):
    """PyTorch-default-style init (kaiming-uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
    weights and biases, GroupNorm 1/0, `out_layers.3` zeroed as simple_unet.py:172 does unless
    zero_out_layers=False, which keeps every convolution live at its default-init scale)."""
    g = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    fan = {}
    for name, shape in param_spec(channels, in_channels, attention):
        if len(shape) >= 2:
            n = 1
            for s in shape[1:]:
                n *= s
            fan[name[: -len(".weight")]] = n
    for name, shape in param_spec(channels, in_channels, attention):
        base, kind = name.rsplit(".", 1)
        if base in fan:
            bound = 1.0 / math.sqrt(fan[base])
            v = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound
            if (".out_layers.3" in name or name.startswith("attn.proj")) and zero_out_layers:
                v = torch.zeros(shape, dtype=torch.float64)
        else:  # GroupNorm affine
            v = torch.ones(shape, dtype=torch.float64) if kind == "weight" else torch.zeros(shape, dtype=torch.float64)
        params[name] = v.to(dtype)
    return params


# --------------------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------------------
def timestep_embedding(timesteps, dim, max_period):
    """simple_unet.py:205-224 — layout [cos(half) | sin(half)], freqs = exp(-ln(max_period)*k/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _mlp(p, prefix, x):
    """Linear -> SiLU -> Linear (simple_unet.py:20-34)."""
    h = F.linear(x, p[f"{prefix}.0.weight"], p[f"{prefix}.0.bias"])
    return F.linear(F.silu(h), p[f"{prefix}.2.weight"], p[f"{prefix}.2.bias"])


def embed(p, logsnr, guide=None, cond_w=None):
    """simple_unet.py:45-64: time MLP (+ masked guide MLP) (+ cond_w MLP)."""
    emb = _mlp(p, "time_embed", timestep_embedding(logsnr.float(), 64, MAX_TIMESTEPS))
    if guide is not None:
        guide = guide.clone()
        mask = guide == -1                      # simple_unet.py:54 (integer compare, bit-exact)
        guide[mask] = 0                         # :55
        g = _mlp(p, "guide_embed", F.one_hot(guide, num_classes=10).float())
        g = torch.where(mask[:, None], torch.zeros_like(g), g)   # :57 rows zeroed AFTER the MLP
        emb = emb + g
    if cond_w is not None:
        emb = emb + _mlp(p, "cond_w_embed", timestep_embedding(cond_w, 64, 4))
    return emb


def gn_silu(x, w, b, groups=32, eps=1e-5):
    return F.silu(F.group_norm(x, groups, w, b, eps))


def resblock(p, prefix, x, emb, drop=None):
    """simple_unet.py:181-186.  drop = (mask, p): the nn.Dropout(p) of out_layers (:171) with an explicit keep-mask."""
    h = gn_silu(x, p[f"{prefix}.in_layers.0.weight"], p[f"{prefix}.in_layers.0.bias"])
    h = F.conv2d(h, p[f"{prefix}.in_layers.2.weight"], p[f"{prefix}.in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), p[f"{prefix}.emb_layers.1.weight"], p[f"{prefix}.emb_layers.1.bias"])
    h = h + e[..., None, None]
    h = gn_silu(h, p[f"{prefix}.out_layers.0.weight"], p[f"{prefix}.out_layers.0.bias"])
    if drop is not None:
        h = h * drop[0] / (1.0 - drop[1])
    h = F.conv2d(h, p[f"{prefix}.out_layers.3.weight"], p[f"{prefix}.out_layers.3.bias"], padding=1)
    if f"{prefix}.skip_connection.weight" in p:
        x = F.conv2d(x, p[f"{prefix}.skip_connection.weight"], p[f"{prefix}.skip_connection.bias"])
    return x + h


def attention_core(p, a):
    """q, k, v = split(conv1x1(a), 3);  proj(softmax(q k^T / sqrt(C)) v) on an NCHW map `a` - the contraction core of the attention extension.
    It follows the one attention implementation the reference holds, `CausalSelfAttention.forward` (gms/autoregs/pixel_transformer.py:101-122:
    key / query / value = Linear(x), att = softmax(q k^T / sqrt(head size)), y = proj(att v)) with n_head = 1 and an all-ones mask; the rows of
    `attn.qkv.weight` are [query | key | value].  PINNED by tests/golden/attn_core_{64,256}.npz, generated from that class
    (oracle/make_golden.py gen_attn_core)."""
    B, C, H, W = a.shape
    qkv = F.conv2d(a, p["attn.qkv.weight"], p["attn.qkv.bias"]).reshape(B, 3, C, H * W)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]                       # [B, C, N]
    w = torch.softmax(torch.einsum("bci,bcj->bij", q, k) * (C ** -0.5), dim=-1)
    o = torch.einsum("bij,bcj->bci", w, v).reshape(B, C, H, W)
    return F.conv2d(o, p["attn.proj.weight"], p["attn.proj.bias"])


def attention_block(p, x):
    """The self-attention extension (north_star "optional self-attention block", BASELINE config 5).  The reference's SimpleUnet has no
    attention block: its PLACEMENT (lowest resolution, pre-activation, residual) is this function's own definition; its contraction core is the
    reference's own attention arithmetic (`attention_core`):
        a = SiLU(GroupNorm32(x));  out = x + attention_core(a)."""
    return x + attention_core(p, gn_silu(x, p["attn.norm.weight"], p["attn.norm.bias"]))


def attn_core_case(T, C=128, B=2):
    """Inputs of the attention-core fixtures, regenerated by the tests instead of stored: tokens x [B, T, C], upstream gradient dy, and the four
    Linear layers of the reference class as (weight, bias) pairs keyed by the reference's attribute names.  Everything is an exact integer hash of
    the element index -> uniform, zero mean: x of unit variance, weights of unit gain (U(-sqrt(3 / C), sqrt(3 / C)): logits of std ~ 1, a live
    softmax), biases 0.1 U(-1, 1).  (`closed_form_params`' sinusoid fill is no use here: sin(0.37 (i C + j) + phase) is a rank-2 matrix.)"""
    def hashed(n, salt):
        h = (torch.arange(n, dtype=torch.int64) + salt) * 0x9E3779B1 & 0xFFFFFFFF
        h = (h ^ (h >> 15)) * 0x85EBCA77 & 0xFFFFFFFF
        h = (h ^ (h >> 13)) * 0xC2B2AE3D & 0xFFFFFFFF
        h = h ^ (h >> 16)
        return (h.double() + 0.5) / 2147483648.0 - 1.0                # U(-1, 1)
    lin = {}
    for i, name in enumerate(("query", "key", "value", "proj")):
        lin[name] = ((math.sqrt(3.0 / C) * hashed(C * C, 1000003 * (i + 1))).reshape(C, C).float(), (0.1 * hashed(C, 77777 * (i + 1))).float())
    x = (math.sqrt(3.0) * hashed(B * T * C, 17)).reshape(B, T, C).float()
    dy = (math.sqrt(3.0) * hashed(B * T * C, 900001) / (B * T)).reshape(B, T, C).float()
    return x, dy, lin


def unet_forward(p, x, logsnr, guide=None, cond_w=None, taps=None, dropout=None):
    """v_hat = net(z, logsnr, guide, cond_w) — simple_unet.py:44-72.

    `taps`, if a dict, receives named intermediate activations (NCHW) for per-layer checks.
    `dropout` = (masks, p): training-mode nn.Dropout(p) with one explicit NCHW keep-mask per ResBlock name."""
    # `up.seq[3]`'s ResBlock is constructed without the dropout argument (simple_unet.py:138): it never drops
    dm = (lambda n: (dropout[0][n], dropout[1]) if n != "up.seq.3.0" else None) if dropout is not None else (lambda n: None)
    emb = embed(p, logsnr, guide, cond_w)
    t = taps if taps is not None else {}
    t["emb"] = emb
    cache = []
    # Down (simple_unet.py:90-109)
    h = F.conv2d(x, p["down.seq.0.conv.weight"], p["down.seq.0.conv.bias"], padding=1)
    cache.append(h)
    for i in (1, 2):
        h = resblock(p, f"down.seq.{i}", h, emb, dm(f"down.seq.{i}"))
        cache.append(h)
    h = F.conv2d(h, p["down.seq.3.conv.weight"], p["down.seq.3.conv.bias"], stride=2, padding=1)
    cache.append(h)
    for i in (4, 5):
        h = resblock(p, f"down.seq.{i}", h, emb, dm(f"down.seq.{i}"))
        cache.append(h)
    h = F.conv2d(h, p["down.seq.6.conv.weight"], p["down.seq.6.conv.bias"], stride=2, padding=1)
    cache.append(h)
    for i, c in enumerate(cache):
        t[f"down.{i}"] = c
    # turn (:68)
    h = resblock(p, "turn", h, emb, dm("turn"))
    if "attn.qkv.weight" in p:
        h = attention_block(p, h)
    t["turn"] = h
    # Up (:146-152): cat with the reversed cache, ResBlock(2C->C), nearest x2 + conv at idx 0 and 3
    for i in range(7):
        h = torch.cat([h, cache[6 - i]], 1)
        if i in (0, 3):
            h = resblock(p, f"up.seq.{i}.0", h, emb, dm(f"up.seq.{i}.0"))
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, p[f"up.seq.{i}.1.conv.weight"], p[f"up.seq.{i}.1.conv.bias"], padding=1)
        else:
            h = resblock(p, f"up.seq.{i}", h, emb, dm(f"up.seq.{i}"))
        t[f"up.{i}"] = h
    # head (:38-42)
    h = gn_silu(h, p["out.0.weight"], p["out.0.bias"])
    return F.conv2d(h, p["out.2.weight"], p["out.2.bias"], padding=1)


def count_params(channels, in_channels=1, include_cond_w=True):
    n = 0
    for name, shape in param_spec(channels, in_channels):
        if not include_cond_w and name.startswith("cond_w_embed"):
            continue
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
