import sys, torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib
C = 128
for B, S in ((9, 64), (40, 64), (9, 32)):
    g = torch.Generator().manual_seed(B + S)
    x = (torch.randn((B, S, S, C), generator=g) * 1.3 + 0.2).bfloat16().cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda(); beta = (0.3 * torch.randn(C, generator=g)).cuda()
    w = torch.randn((128, C, 3, 3), generator=g).cuda() / (C * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); ops.pack_conv_weight(w, wf, None)
    a, m, r = ops.gn_silu_fwd(x, gamma, beta, 32)
    tsc = torch.empty((B, C), device="cuda"); tsh = torch.empty_like(tsc)
    ops.gn_stats(x, gamma, beta, 32, tsc, tsh)
    for v in (16, 32):
        lib.gmk_set_dev_variant(v)
        out = ops.conv_igemm([x], wf, 128, 3, ops.NORMAL, (S, S), gn=(tsc, tsh))
        ref = ops.conv_igemm([a], wf, 128, 3, ops.NORMAL, (S, S))
        d = (out != ref)
        n = int(d.sum())
        print(B, S, "variant", v, "differing:", n, "max abs diff", float((out.float() - ref.float()).abs().max()))
        if n:
            idx = d.nonzero()
            print("   samples", sorted(set(idx[:, 0].tolist()))[:12], "rows", sorted(set(idx[:, 1].tolist()))[:40], "cols", sorted(set(idx[:, 2].tolist()))[:70])
    lib.gmk_set_dev_variant(0)
