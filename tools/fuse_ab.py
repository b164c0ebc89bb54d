"""Fused GroupNorm convolution vs GroupNorm launch + plain convolution, in-process, at the sampler's shapes.  python tools/fuse_ab.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
C = 128
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6
for B, S, nsrc in ((1024, 28, 1), (1024, 28, 2), (2048, 32, 1), (2048, 32, 2), (2048, 16, 1), (1024, 64, 1)):
    srcs = [torch.randn(B, S, S, C, device="cuda").half() for _ in range(nsrc)]
    ct = C * nsrc
    gamma = torch.ones(ct, device="cuda"); beta = torch.zeros(ct, device="cuda")
    w = torch.randn(128, ct, 3, 3, device="cuda") / (ct * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.float16); ops.pack_conv_weight(w, wf, None)
    gpc = 32 // nsrc
    tsc = torch.empty((B, ct), device="cuda"); tsh = torch.empty_like(tsc)
    def gn_all():
        return [ops.gn_silu_fwd(s, gamma[i * C:(i + 1) * C], beta[i * C:(i + 1) * C], gpc)[0] for i, s in enumerate(srcs)]
    def stats_all():
        for i, s in enumerate(srcs):
            ops.gn_stats(s, gamma[i * C:(i + 1) * C], beta[i * C:(i + 1) * C], gpc, tsc[:, i * C:(i + 1) * C], tsh[:, i * C:(i + 1) * C])
    a = gn_all(); stats_all()
    t_gn, t_st = timeit(gn_all), timeit(stats_all)
    t_conv = timeit(lambda: ops.conv_igemm(a, wf, 128, 3, ops.NORMAL, (S, S)))
    t_fuse = timeit(lambda: ops.conv_igemm(srcs, wf, 128, 3, ops.NORMAL, (S, S), gn=(tsc, tsh)))
    print(f"B={B} {S}x{S} srcs={nsrc}: GroupNorm {t_gn:6.1f} + conv {t_conv:6.1f} = {t_gn + t_conv:6.1f} us | statistics {t_st:6.1f} + fused conv {t_fuse:6.1f} = {t_st + t_fuse:6.1f} us")
