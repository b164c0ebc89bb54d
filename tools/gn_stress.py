"""Randomised cross-check of the GroupNorm+SiLU kernel variants (register-resident forward, hybrid backward, slab / whole-sample
streaming) against the whole-sample streaming kernels, with addends, per-(sample, channel) input addends and dropout."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from generative_models_amd._lib import lib
T = torch.bfloat16
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nfail = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H, W = rng.choice([(7, 7), (8, 8), (10, 10), (12, 12), (14, 14), (16, 16), (20, 20), (24, 24), (25, 25), (28, 28), (26, 32), (30, 30),
                       (32, 32), (40, 40), (64, 64), (9, 64), (28, 20)])
    B = rng.choice([1, 2, 3, 8, 9, 16]); G = rng.choice([32, 16]); C = 128
    g = torch.Generator().manual_seed(100 + it)
    x = (torch.randn((B, H, W, C), generator=g) * 1.5 + 0.3).cuda().to(T)
    dy = torch.randn((B, H, W, C), generator=g).cuda().to(T)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda(); beta = (0.1 * torch.randn(C, generator=g)).cuda()
    d1 = torch.randn((B, H, W, C), generator=g).cuda().to(T) if rng.random() < 0.5 else None
    d2 = torch.randn((B, H, W, C), generator=g).cuda().to(T) if d1 is not None and rng.random() < 0.5 else None
    xa_full = torch.randn((B, 3 * C), generator=g).cuda()
    xadd = xa_full[:, C:2 * C] if rng.random() < 0.5 else None
    drop = (0.2, 1234, 16 * it) if rng.random() < 0.3 else None
    res = {}
    for mode in (1, 0):
        lib.gmk_set_kernel_choice(-1, -1, mode)
        y, mean, rstd = ops.gn_silu_fwd(x, gamma, beta, G, dropout=drop, xadd=xadd)
        xs = torch.empty((B, C), device="cuda")
        dx, dgp, dbp = ops.gn_silu_bwd(dy, x, gamma, beta, mean, rstd, dadd1=d1, dadd2=d2, dxsum=xs, dropout=drop, xadd=xadd)
        res[mode] = [t.float() for t in (y, mean, rstd, dx, dgp, dbp, xs)]
    errs = [float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(res[0], res[1])]
    bad = max(errs[0], errs[3]) > 1.2e-2 or max(errs[1], errs[2]) > 1e-4 or max(errs[4:]) > 2e-3
    nfail += bad
    print(("FAIL " if bad else "ok   ") + f"B={B} {H}x{W} G={G} dadd={(d1 is not None) + (d2 is not None)} xadd={xadd is not None} drop={drop is not None} "
          f"err={[round(e, 5) for e in errs]}", flush=True)
lib.gmk_set_kernel_choice(-1, -1, -1)
print("failures:", nfail)
sys.exit(1 if nfail else 0)
