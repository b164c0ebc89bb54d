"""3x3 convolution with a residual epilogue (12 of the 26 forward launches): residual tile handed over by the producer waves through LDS
(default) against the consumers' own loads (GMK_DEV_VARIANT=7), same box, interleaved; results must be bit-identical."""
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
for (B, S, two, dt) in ((2048, 32, False, torch.float16), (2048, 32, True, torch.float16), (2048, 16, False, torch.float16), (1024, 28, False, torch.float16),
                        (1024, 28, True, torch.float16), (1024, 64, False, torch.float16), (2048, 8, False, torch.float16), (1000, 14, False, torch.bfloat16),
                        (2048, 32, False, torch.bfloat16)):
    g = torch.Generator().manual_seed(0)
    srcs = [torch.randn((B, S, S, C), generator=g).cuda().to(dt) for _ in range(2 if two else 1)]
    res = torch.randn((B, S, S, C), generator=g).cuda().to(dt)
    cin = C * len(srcs)
    w = (torch.randn((C, cin, 3, 3), generator=g) / (3 * cin ** 0.5)).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=dt); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    bias = torch.randn(C, generator=g).cuda()
    outs = {}
    for rnd in range(2):
        for v in (0, 7):
            lib.gmk_set_dev_variant(v)
            o = ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), bias=bias, residual=res)
            outs[v] = o
            t = timed(lambda: ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), bias=bias, residual=res))
            t0 = timed(lambda: ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), bias=bias))
            print(f"B={B} {S}x{S} cin={cin} {str(dt)[6:]} variant {v} (kernel {lib.gmk_last_kernel()}): with residual {t:7.1f} us, without {t0:7.1f} us", flush=True)
        assert torch.equal(outs[0], outs[7]), "hand-over result differs"
lib.gmk_set_dev_variant(0)
print("bit-identical")
