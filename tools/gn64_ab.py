"""GroupNorm backward at 64 x 64 (B = 1024, C = 128): 16-channel-slab hybrid (mode 7) against the 32-channel-slab resident form (mode 9 / 0),
with 0 / 1 / 2 gradient addends and fp16 / bf16 x; results must agree bit for bit in dgamma / dbeta partials up to summation order."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

dev = "cuda"


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
    return e0.elapsed_time(e1) / n * 1e-3


for (B, S, C, G, xdt) in ((1024, 64, 128, 32, torch.float16), (1024, 64, 128, 32, torch.bfloat16), (512, 64, 256, 32, torch.float16)):
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, S, S, C, generator=g).to(dev, xdt)
    dy = torch.randn(B, S, S, C, generator=g).to(dev, torch.bfloat16)
    d1 = torch.randn(B, S, S, C, generator=g).to(dev, torch.bfloat16)
    d2 = torch.randn(B, S, S, C, generator=g).to(dev, torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    nbytes = x.numel() * 2
    lib.gmk_set_kernel_choice(-1, -1, 0)
    y, mean, rstd = ops.gn_silu_fwd(x, gamma, beta, G)
    for nadd in (0, 1, 2):
        kw = dict(dadd1=d1 if nadd >= 1 else None, dadd2=d2 if nadd >= 2 else None)
        ref = None
        for rnd in range(2):
            for m in (7, 9):
                lib.gmk_set_kernel_choice(-1, -1, m)
                dx, dgp, dbp = ops.gn_silu_bwd(dy, x, gamma, beta, mean, rstd, **kw)
                out = [dx.float(), dgp.sum(0) if dgp.dim() > 1 else dgp, dbp.sum(0) if dbp.dim() > 1 else dbp]
                if ref is None:
                    ref = out
                else:
                    errs = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(out, ref)]
                    assert errs[0] < 1e-2 and max(errs[1:]) < 1e-4, (m, errs)
                tb = timed(lambda: ops.gn_silu_bwd(dy, x, gamma, beta, mean, rstd, **kw))
                print(f"B={B} S={S} C={C} x={str(xdt)[6:]} addends={nadd} mode {m}: bwd {tb*1e6:7.1f} us "
                      f"({(3 + nadd)*nbytes/tb/1e12:5.2f} TB/s of {3 + nadd} passes)", flush=True)
lib.gmk_set_kernel_choice(-1, -1, -1)
