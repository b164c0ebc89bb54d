"""3x3 halo kernel: XCD-aware tile order (default) against the plain round-robin order (GMK_DEV_VARIANT=9); bit-identical results."""
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
for (B, S, two, dt) in ((2048, 32, False, torch.float16), (2048, 32, True, torch.float16), (2048, 16, False, torch.bfloat16), (1024, 28, False, torch.float16),
                        (1024, 64, False, torch.float16), (2048, 8, False, torch.float16), (1000, 14, False, torch.bfloat16)):
    g = torch.Generator().manual_seed(0)
    srcs = [torch.randn((B, S, S, C), generator=g).cuda().to(dt) for _ in range(2 if two else 1)]
    cin = C * len(srcs)
    w = (torch.randn((C, cin, 3, 3), generator=g) / (3 * cin ** 0.5)).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=dt); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    outs = {}
    for rnd in range(3):
        line = []
        for v in (0, 9):
            lib.gmk_set_dev_variant(v)
            outs[v] = ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S))
            t = timed(lambda: ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S)))
            line.append(f"variant {v}: {t:7.1f} us")
        assert torch.equal(outs[0], outs[9])
        print(f"B={B} {S}x{S} cin={cin} {str(dt)[6:]}: " + "   ".join(line), flush=True)
lib.gmk_set_dev_variant(0)
