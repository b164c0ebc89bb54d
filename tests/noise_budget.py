"""CPU experiment (not a test; run by hand): where the bf16 path's output noise comes from, and what changing the storage
type of each class of tensor would buy.  The oracle's forward is re-run with every storage point of the HIP path rounded
to a chosen type in turn (same points as generative_models_amd/diffusion/simple_unet.py: packed conv weights, GroupNorm+SiLU
outputs, conv1 outputs h, 1x1 skip-conv outputs, residual-stream tensors); errors are reported relative to the fp32 forward
on the reference-pinned default-init goldens (tests/golden/definit_c128_s*.npz).

    python tests/noise_budget.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref as U   # noqa: E402


def rnd(t, kind):
    if kind == "f32":
        return t
    if kind == "bf16":
        return t.bfloat16().float()
    if kind == "f16":
        return t.half().float()
    raise ValueError(kind)


def forward(p, x, logsnr, guide, cfg):
    """cfg: {'w','a','h','res','stream'} -> 'f32' | 'bf16' | 'f16'"""
    q = lambda t, k: rnd(t, cfg[k])
    W = lambda n: q(p[n], "w")
    emb = U.embed(p, logsnr, guide, None)

    def res(prefix, x):
        a = q(U.gn_silu(x, p[f"{prefix}.in_layers.0.weight"], p[f"{prefix}.in_layers.0.bias"]), "a")
        h = q(F.conv2d(a, W(f"{prefix}.in_layers.2.weight"), None, padding=1), "h")
        e = F.linear(F.silu(emb), p[f"{prefix}.emb_layers.1.weight"], p[f"{prefix}.emb_layers.1.bias"])
        h = h + (e + p[f"{prefix}.in_layers.2.bias"])[..., None, None]
        a2 = q(U.gn_silu(h, p[f"{prefix}.out_layers.0.weight"], p[f"{prefix}.out_layers.0.bias"]), "a")
        o = F.conv2d(a2, W(f"{prefix}.out_layers.3.weight"), p[f"{prefix}.out_layers.3.bias"], padding=1)
        if f"{prefix}.skip_connection.weight" in p:
            x = q(F.conv2d(x, W(f"{prefix}.skip_connection.weight"), p[f"{prefix}.skip_connection.bias"]), "res")
        return q(x + o, "stream")

    cache = []
    h = q(F.conv2d(x, p["down.seq.0.conv.weight"], p["down.seq.0.conv.bias"], padding=1), "stream")
    cache.append(h)
    for i in (1, 2):
        h = res(f"down.seq.{i}", h); cache.append(h)
    h = q(F.conv2d(h, W("down.seq.3.conv.weight"), p["down.seq.3.conv.bias"], stride=2, padding=1), "stream"); cache.append(h)
    for i in (4, 5):
        h = res(f"down.seq.{i}", h); cache.append(h)
    h = q(F.conv2d(h, W("down.seq.6.conv.weight"), p["down.seq.6.conv.bias"], stride=2, padding=1), "stream"); cache.append(h)
    h = res("turn", h)
    for i in range(7):
        h = torch.cat([h, cache[6 - i]], 1)
        if i in (0, 3):
            h = res(f"up.seq.{i}.0", h)
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = q(F.conv2d(h, W(f"up.seq.{i}.1.conv.weight"), p[f"up.seq.{i}.1.conv.bias"], padding=1), "stream")
        else:
            h = res(f"up.seq.{i}", h)
    a = q(U.gn_silu(h, p["out.0.weight"], p["out.0.bias"]), "a")
    return F.conv2d(a, p["out.2.weight"], p["out.2.bias"], padding=1)


def main():
    torch.set_num_threads(8)
    base = dict(w="bf16", a="bf16", h="bf16", res="bf16", stream="bf16")
    variants = {
        "all bf16 (shipped round 1)": base,
        "h f16": {**base, "h": "f16"},
        "h + res f16": {**base, "h": "f16", "res": "f16"},
        "stream f16": {**base, "stream": "f16"},
        "h + res + stream f16": {**base, "h": "f16", "res": "f16", "stream": "f16"},
        "h + res + stream f32": {**base, "h": "f32", "res": "f32", "stream": "f32"},
        "h + res + stream f16, a f16 (w bf16)": {**base, "h": "f16", "res": "f16", "stream": "f16", "a": "f16"},
        "everything f16": dict(w="f16", a="f16", h="f16", res="f16", stream="f16"),
        "only w bf16": dict(w="bf16", a="f32", h="f32", res="f32", stream="f32"),
        "only a bf16": dict(w="f32", a="bf16", h="f32", res="f32", stream="f32"),
    }
    for name in ("definit_c128_s28.npz", "definit_c128_s32.npz"):
        g = np.load(os.path.join(ROOT, "tests", "golden", name))
        p = U.reference_init_params(128, 1, seed=int(g["init_seed"]), zero_out_layers=False)
        z, l, y = (torch.from_numpy(g[k]) for k in ("z", "logsnr", "guide"))
        ref = torch.from_numpy(g["v"]).double()
        with torch.no_grad():
            exact = forward(p, z, l, y, dict(w="f32", a="f32", h="f32", res="f32", stream="f32")).double()
            print(f"{name}: oracle-structure fp32 vs reference golden: {float((exact - ref).abs().max() / ref.abs().max()):.2e}")
            for vn, cfg in variants.items():
                o = forward(p, z, l, y, cfg).double()
                mx = float((o - ref).abs().max() / ref.abs().max())
                l2 = float((o - ref).norm() / ref.norm())
                print(f"   {vn:42s} max-norm {mx:.2e}   L2 {l2:.2e}")


if __name__ == "__main__":
    main()
