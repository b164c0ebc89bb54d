"""Is the 3x3 halo kernel limited by the board's power management rather than by its instruction stream?  The same launch (32 x 32, B = 2048,
K = 1152, fp16 operands) on random operands, on all-zero operands and on operands with 7 / 8 of the values zeroed: identical instructions and
addresses, only the number of bits that toggle in the multipliers and on the data paths differs.  TFLOP/s each, interleaved."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops

C, S, B = 128, 32, 2048
g = torch.Generator().manual_seed(0)


def make(kind, dt):
    x = torch.randn((B, S, S, C), generator=g)
    w = torch.randn((C, C, 3, 3), generator=g) / 34
    if kind == "zeros":
        x.zero_(); w.zero_()
    elif kind == "sparse":
        x *= (torch.rand(x.shape, generator=g) < 0.125)
    x = x.cuda().to(dt); w = w.cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=dt); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    return x, wf


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


flops = 2.0 * B * S * S * C * C * 9
for dt in (torch.float16, torch.bfloat16):
    sets = {k: make(k, dt) for k in ("random", "sparse", "zeros")}
    for rnd in range(3):
        row = []
        for k, (x, wf) in sets.items():
            t = timed(lambda: ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S)))
            row.append(f"{k} {t * 1e6:6.1f} us = {flops / t / 1e12:6.0f} TFLOP/s")
        print(f"{str(dt)[6:]:9s} " + " | ".join(row), flush=True)
