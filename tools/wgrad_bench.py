"""3x3 weight-gradient kernels A/B: parity against fp32 torch on a small problem, then in-process timing at the shapes the train
step runs (GMK_WGRAD_KERNEL: 2 = 8-compute-wave slot kernel, 3 = wave-specialised slot kernel).  python tools/wgrad_bench.py"""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

def wgrad(dy, srcs, variant):
    lib.gmk_set_kernel_choice(-1, variant, -1)
    cin = sum(s.shape[3] for s in srcs)
    dw = torch.empty((dy.shape[3], cin, 3, 3), device="cuda")
    ops.conv_wgrad(dy, srcs, 3, ops.NORMAL, dw)
    return dw, lib.gmk_last_kernel()

torch.manual_seed(0)
for B, S, two in ((3, 12, False), (2, 28, True), (5, 7, False), (2, 64, False), (3, 32, True)):
    xs = [torch.randn(B, S, S, 128, device="cuda").bfloat16() for _ in range(2 if two else 1)]
    dy = torch.randn(B, S, S, 128, device="cuda").bfloat16()
    x = torch.cat([t.float() for t in xs], 3).permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(128, x.shape[1], 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x, w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    for v in (2, 3):
        dw, k = wgrad(dy, xs, v)
        err = float((dw - w.grad).abs().max() / w.grad.abs().max())
        print(f"B={B} S={S} two={two} variant {v} (kernel {k}): rel err {err:.2e}", "OK" if err < 2e-3 else "MISMATCH")
        assert err < 2e-3
    a, _ = wgrad(dy, xs, 2); b, _ = wgrad(dy, xs, 3)
print("parity ok")
for B, S, two in ((1024, 28, False), (1024, 28, True), (1024, 14, False), (2048, 32, False), (2048, 32, True), (1024, 64, False), (1024, 7, False)):
    xs = [torch.randn(B, S, S, 128, device="cuda").bfloat16() for _ in range(2 if two else 1)]
    dy = torch.randn(B, S, S, 128, device="cuda").bfloat16()
    flops = 2.0 * B * S * S * 128 * 128 * len(xs) * 9
    res = {}
    for rnd in range(3):
        for v in (2, 3):
            wgrad(dy, xs, v)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): wgrad(dy, xs, v)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            res.setdefault(v, []).append(dt)
    print(f"B={B} {S}x{S} K={1152 * len(xs)}: " + "  ".join(f"variant {v}: {min(t) * 1e6:7.1f} us = {flops / min(t) / 1e12:6.1f} TFLOP/s (reduce included)" for v, t in res.items()))
lib.gmk_set_kernel_choice(-1, -1, -1)
