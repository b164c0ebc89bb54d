"""Micro-benchmark of the convolution kernels on single shapes (development tool; not part of the product path).
    python tools/conv_bench.py [--dtype bf16] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    T = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    B, C = a.batch, 128
    shapes = [("3x3 128->128 @28", 28, 1, 3), ("3x3 256->128 @28", 28, 2, 3), ("1x1 256->128 @28", 28, 2, 1),
              ("3x3 128->128 @14", 14, 1, 3), ("3x3 256->128 @14", 14, 2, 3), ("3x3 128->128 @7", 7, 1, 3)]
    for name, S, nsrc, ks in shapes:
        if a.only and a.only not in name:
            continue
        srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(nsrc)]
        cin = nsrc * C
        w = torch.randn((C, cin, ks, ks), device="cuda") / (cin * ks * ks) ** 0.5
        wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
        ops.pack_conv_weight(w, wf, wd)
        bias = torch.zeros(C, device="cuda")
        res = torch.randn((B, S, S, C), device="cuda").to(T)
        flops = 2.0 * B * S * S * C * cin * ks * ks
        t = timed(lambda: ops.conv_igemm(srcs, wf, C, ks, ops.NORMAL, (S, S), bias=bias, residual=res), a.iters)
        dy = torch.randn((B, S, S, C), device="cuda").to(T)
        dw = torch.empty_like(w)
        t2 = timed(lambda: ops.conv_wgrad(dy, srcs, ks, ops.NORMAL, dw), a.iters)
        print(f"{name:22s} fwd {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s | wgrad {t2 * 1e6:8.1f} us {flops / t2 / 1e12:7.1f} TF/s",
              flush=True)
    for S in (28, 14, 7):
        x = torch.randn((B, S, S, C), device="cuda").to(T)
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        t = timed(lambda: ops.gn_silu_fwd(x, g, b, 32), a.iters)
        nbytes = x.numel() * x.element_size()
        print(f"gn_silu_fwd @{S} {t * 1e6:8.1f} us  {2 * nbytes / t / 1e12:.2f} TB/s of (1 read + 1 write)")
        y, mean, rstd = ops.gn_silu_fwd(x, g, b, 32)
        t = timed(lambda: ops.gn_silu_bwd(x, x, g, b, mean, rstd), a.iters)
        print(f"gn_silu_bwd @{S} {t * 1e6:8.1f} us  {3 * nbytes / t / 1e12:.2f} TB/s of (2 reads + 1 write)")


if __name__ == "__main__":
    main()
