"""Stride-2 data gradient (GMK_CONV_TRANSPOSED2) at the train step's shapes: four-phase form (default, kernel id 6) against the zero-stuffed
halo form (GMK_CONV_KERNEL=3, id 5), with the residual the net adds; same box, interleaved."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


C = 128
for (B, S) in ((2048, 32), (2048, 16), (1024, 64), (1024, 28), (1024, 14)):
    g = torch.Generator().manual_seed(0)
    dy = torch.randn((B, S // 2, S // 2, C), generator=g).cuda().bfloat16()
    res = torch.randn((B, S, S, C), generator=g).cuda().bfloat16()
    w = (torch.randn((C, C, 3, 3), generator=g) / 34).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    for rnd in range(2):
        for choice in (0, 3):
            lib.gmk_set_kernel_choice(choice, -1, -1)
            for r in (None, res):
                t = timed(lambda: ops.conv_igemm([dy], wd, C, 3, ops.TRANSPOSED2, (S, S), residual=r))
                nb = dy.numel() * 2 + res.numel() * 2 * (2 if r is not None else 1)
                print(f"B={B} {S // 2}->{S} choice {choice} (kernel {lib.gmk_last_kernel()}) residual={r is not None}: {t:7.1f} us  {nb / t / 1e6:5.2f} TB/s", flush=True)
lib.gmk_set_kernel_choice(-1, -1, -1)
