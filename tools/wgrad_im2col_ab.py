"""im2col weight-gradient kernel (1x1 skip convolution over the concatenated input; 3x3 stride 2).  Default: the workgroups of one pixel range
next to each other on one XCD; the 1x1 form over 128 + 128 input channels runs on its own streaming kernel (conv1x1_stream.hip).
GMK_DEV_VARIANT=61: the (split, tap, block) grid of rounds 1 - 3 (same bits); 47: the im2col kernel for the 1x1 form too.
   python tools/wgrad_im2col_ab.py          (under rocprofv3 --pmc FETCH_SIZE with WG_ONLY=0|61 for the bytes)"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

def run(dy, srcs, k, mode, variant):
    lib.gmk_set_dev_variant(variant)
    try:
        cin = sum(s.shape[3] for s in srcs)
        dw = torch.empty((dy.shape[3], cin, k, k), device="cuda")
        ops.conv_wgrad(dy, srcs, k, mode, dw)
        return dw
    finally:
        lib.gmk_set_dev_variant(0)

NAMES = {0: "default", 47: "im2col kernel", 61: "old block order"}
torch.manual_seed(0)
only = os.environ.get("WG_ONLY")
for B, S, k, mode, two in ((2048, 32, 1, ops.NORMAL, True), (2048, 16, 1, ops.NORMAL, True), (2048, 32, 3, ops.STRIDE2, False), (1024, 64, 1, ops.NORMAL, True),
                           (1024, 28, 1, ops.NORMAL, True)):
    So = S // 2 if mode == ops.STRIDE2 else S
    xs = [torch.randn(B, S, S, 128, device="cuda").half() for _ in range(2 if two else 1)]
    dy = torch.randn(B, So, So, 128, device="cuda").bfloat16()
    if not only:
        a, b = run(dy, xs, k, mode, 0), run(dy, xs, k, mode, 47 if k == 1 else 61)
        assert torch.equal(a, b), "the kernels / block orders disagree"
    res = {}
    for rnd in range(3):
        for v in ((int(only),) if only else ((0, 47) if k == 1 else (0, 61))):
            run(dy, xs, k, mode, v)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run(dy, xs, k, mode, v)
            torch.cuda.synchronize(); res.setdefault(v, []).append((time.perf_counter() - t0) / 10)
    nbytes = sum(x.numel() for x in xs) * 2 + dy.numel() * 2
    print(f"B={B} {S}x{S} k={k} mode={mode} cin={128 * len(xs)}: " + "  ".join(
        f"{NAMES[v]}: {sorted(t)[1] * 1e6:7.1f} us = {nbytes / sorted(t)[1] / 1e9:6.0f} GB/s of its operands" for v, t in res.items()), flush=True)
