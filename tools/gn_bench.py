"""Time the GroupNorm+SiLU forward/backward kernels per workgroup-slab mode (gmk_set_kernel_choice gn = 1 whole sample,
3 = 32-channel slabs, 4 = 64-channel slabs, 0 = automatic: register / hybrid single-read kernels) and check that the modes agree."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

dev = "cuda"
modes = [int(m) for m in sys.argv[1:]] or [1, 3, 4]


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


for (B, S, C, G) in ((1024, 28, 128, 32), (1024, 28, 128, 16), (1024, 14, 128, 32), (1024, 7, 128, 32), (2048, 32, 128, 32),
                     (1000, 28, 128, 32), (1024, 64, 128, 32), (512, 64, 128, 32), (2048, 16, 128, 32), (1024, 32, 128, 32), (2048, 8, 128, 32)):
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, S, S, C, generator=g).to(dev, torch.bfloat16)
    dy = torch.randn(B, S, S, C, generator=g).to(dev, torch.bfloat16)
    d1 = torch.randn(B, S, S, C, generator=g).to(dev, torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    nbytes = x.numel() * 2
    ref = None
    for rnd in range(2):
        for m in modes:
            lib.gmk_set_kernel_choice(-1, -1, m)
            y, mean, rstd = ops.gn_silu_fwd(x, gamma, beta, G)
            dx, dgp, dbp = ops.gn_silu_bwd(dy, x, gamma, beta, mean, rstd, dadd1=d1)
            out = [t.float() for t in (y, mean, rstd, dx, dgp, dbp)]
            if ref is None:
                ref = out
            else:
                errs = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(out, ref)]
                assert max(errs) < 1e-2 and max(errs[1:3] + errs[4:]) < 1e-5, (m, errs)
            tf = timed(lambda: ops.gn_silu_fwd(x, gamma, beta, G))
            tb = timed(lambda: ops.gn_silu_bwd(dy, x, gamma, beta, mean, rstd, dadd1=d1))
            print(f"B={B} S={S} C={C} G={G} mode {m}: fwd {tf*1e6:7.1f} us ({2*nbytes/tf/1e12:5.2f} TB/s of 2 passes)  "
                  f"bwd {tb*1e6:7.1f} us ({4*nbytes/tb/1e12:5.2f} TB/s of 4 passes)", flush=True)
lib.gmk_set_kernel_choice(-1, -1, -1)
