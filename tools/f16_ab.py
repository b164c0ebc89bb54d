"""In-process A/B of the kernels that see fp16 activations next to bf16 gradients, against their all-bf16 forms, at the train step's
shapes (B = 2048 at 32 x 32 and its 16 x 16 / 8 x 8 levels, B = 1024 at 64 x 64): slot weight gradient, im2col weight gradient (1x1 skip,
stride 2), GroupNorm forward / backward.  python tools/f16_ab.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

H, BF = torch.float16, torch.bfloat16
def timeit(fn, n=10, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6

torch.manual_seed(0)
print("slot weight gradient (3x3): us bf16-x / fp16-x")
for B, S, two in ((2048, 32, False), (2048, 32, True), (2048, 16, False), (2048, 16, True), (2048, 8, False), (2048, 8, True), (1024, 64, False), (1024, 28, False)):
    dy = torch.randn(B, S, S, 128, device="cuda").to(BF)
    r = {}
    for dt in (BF, H):
        xs = [torch.randn(B, S, S, 128, device="cuda").to(dt) for _ in range(2 if two else 1)]
        dw = torch.empty(128, 128 * len(xs), 3, 3, device="cuda")
        r[dt] = timeit(lambda: ops.conv_wgrad(dy, xs, 3, ops.NORMAL, dw)); k = lib.gmk_last_kernel()
    print(f"  B={B} {S}x{S} K={1152 * (2 if two else 1)} (kernel {k}): {r[BF]:7.1f} / {r[H]:7.1f}  ({(r[H] / r[BF] - 1) * 100:+.1f} %)", flush=True)
print("im2col weight gradient: us bf16-x / fp16-x")
for B, S, ks, mode, two in ((2048, 32, 1, ops.NORMAL, True), (2048, 16, 1, ops.NORMAL, True), (2048, 32, 3, ops.STRIDE2, False), (2048, 16, 3, ops.STRIDE2, False)):
    ho = S // 2 if mode == ops.STRIDE2 else S
    dy = torch.randn(B, ho, ho, 128, device="cuda").to(BF)
    r = {}
    for dt in (BF, H):
        xs = [torch.randn(B, S, S, 128, device="cuda").to(dt) for _ in range(2 if two else 1)]
        dw = torch.empty(128, 128 * len(xs), ks, ks, device="cuda")
        r[dt] = timeit(lambda: ops.conv_wgrad(dy, xs, ks, mode, dw)); k = lib.gmk_last_kernel()
    print(f"  B={B} {S}x{S} k={ks} mode={mode} (kernel {k}): {r[BF]:7.1f} / {r[H]:7.1f}  ({(r[H] / r[BF] - 1) * 100:+.1f} %)", flush=True)
print("GroupNorm forward / backward: us bf16 / fp16 activations")
for B, S, C, G in ((2048, 32, 128, 32), (2048, 16, 128, 32), (2048, 8, 128, 32), (1024, 64, 128, 32), (1024, 28, 128, 32)):
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    dy = torch.randn(B, S, S, C, device="cuda").to(BF)
    rf, rb = {}, {}
    for dt in (BF, H):
        x = torch.randn(B, S, S, C, device="cuda").to(dt)
        y, mean, rstd = ops.gn_silu_fwd(x, gamma, beta, G)
        rf[dt] = timeit(lambda: ops.gn_silu_fwd(x, gamma, beta, G))
        rb[dt] = timeit(lambda: ops.gn_silu_bwd(dy, x, gamma, beta, mean, rstd))
    print(f"  B={B} {S}x{S}: fwd {rf[BF]:7.1f} / {rf[H]:7.1f} ({(rf[H] / rf[BF] - 1) * 100:+.1f} %)   bwd {rb[BF]:7.1f} / {rb[H]:7.1f} ({(rb[H] / rb[BF] - 1) * 100:+.1f} %)", flush=True)
