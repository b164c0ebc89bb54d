"""Stride-2 weight gradient: the slot kernel's four-plane form against the im2col kernel it replaces, per launch, at the train step's sizes.
usage: python tools/wgrad_s2_ab.py [B]      (prints microseconds per launch and the largest relative difference of the two results)"""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
C = 128
for S in (32, 16, 28, 14, 64):
    nb = B if S < 64 else B // 2
    g = torch.Generator(device="cuda").manual_seed(S)
    x = torch.randn((nb, S, S, C), device="cuda", generator=g).half()
    dy = torch.randn((nb, S // 2, S // 2, C), device="cuda", generator=g).bfloat16()
    res = {}
    for name, choice in (("im2col", 1), ("planes", 0)):
        lib.gmk_set_kernel_choice(-1, choice if choice else -1, -1)
        dw = torch.empty((C, C, 3, 3), device="cuda")
        for _ in range(3):
            ops.conv_wgrad(dy, [x], 3, ops.STRIDE2, dw)
        kid = lib.gmk_last_kernel()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_wgrad(dy, [x], 3, ops.STRIDE2, dw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        flops = 2.0 * nb * (S // 2) ** 2 * C * C * 9
        res[name] = dw.clone()
        print(f"{S}->{S // 2} B={nb} {name:7s} kernel {kid}: {us:7.1f} us per launch (with its reduce), {flops / us / 1e6:6.1f} TFLOP/s", flush=True)
    lib.gmk_set_kernel_choice(-1, -1, -1)
    d = float((res["planes"] - res["im2col"]).abs().max() / res["im2col"].abs().max())
    print(f"   largest difference / largest entry: {d:.2e}", flush=True)
