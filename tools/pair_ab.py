"""Data gradient of the 1x1 skip convolution (gmk_conv1x1_pair): the streaming kernel (weights resident in LDS, 128-pixel tiles; default at the
train step's sizes) against the general LDS-DMA kernel (GMK_DEV_VARIANT=46): same bits (one fp32 dot product per output, same k order inside
the MFMA), time.   python tools/pair_ab.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib
torch.manual_seed(0)
for B, S, dt in ((2048, 32, torch.bfloat16), (2048, 16, torch.bfloat16), (2048, 8, torch.bfloat16), (1024, 28, torch.bfloat16), (1024, 64, torch.bfloat16), (777, 14, torch.bfloat16), (2048, 32, torch.float16)):
    dy = (torch.randn(B, S, S, 128, device="cuda") * 0.5).to(dt)
    w = (torch.randn(256, 128, device="cuda") / 11).to(dt)
    res, tm, kern = {}, {}, {}
    for v in (0, 46):
        lib.gmk_set_dev_variant(v)
        a, b = ops.conv1x1_pair(dy, w, 256); kern[v] = lib.gmk_last_kernel()
        res[v] = torch.cat([a, b], -1)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ops.conv1x1_pair(dy, w, 256)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10)
        tm[v] = sorted(ts)[1]
    lib.gmk_set_dev_variant(0)
    ref = dy.float().reshape(-1, 128) @ w.float().t()
    err = float((res[0].float().reshape(-1, 256) - ref).abs().max() / ref.abs().max())
    nbytes = dy.numel() * 2 * 3
    print(f"B={B} {S}x{S} {str(dt)[6:]}: kernels {kern[0]} / {kern[46]}  identical {torch.equal(res[0], res[46])}  err vs fp32 matmul {err:.2e}  "
          f"streaming {tm[0] * 1e6:7.1f} us ({nbytes / tm[0] / 1e9:5.0f} GB/s)  general {tm[46] * 1e6:7.1f} us ({nbytes / tm[46] / 1e9:5.0f} GB/s)", flush=True)
