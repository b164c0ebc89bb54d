"""The 3x3 halo kernel with its halo pieces by LDS-DMA (shipped) against the register-staged fill without a transform (GMK_DEV_VARIANT=11): same bits?
time per launch?  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` for the bytes (the two forms are different template instantiations)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (B, S) in ((2048, 32), (1024, 28), (1024, 64), (2048, 16)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, S, S, C), generator=g).cuda().half()
    res = torch.randn((B, S, S, C), generator=g).cuda().half()
    w = (torch.randn((C, C, 3, 3), generator=g) / 34).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.float16)
    ops.pack_conv_weight(w, wf, None)
    outs = {}
    for rnd in range(2):
        for v in (0, 11):
            lib.gmk_set_dev_variant(v)
            for r in (None, res):
                t = timed(lambda: ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S), residual=r), n)
                outs[(v, r is not None)] = ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S), residual=r)
                print(f"B={B} {S}x{S} variant {v:2d} residual={r is not None}: {t:7.1f} us", flush=True)
    lib.gmk_set_dev_variant(0)
    print("  bit-identical:", torch.equal(outs[(0, False)], outs[(11, False)]), torch.equal(outs[(0, True)], outs[(11, True)]), flush=True)
