"""Is the 3x3 halo kernel sensitive to memory latency?  The same convolution (32 x 32, K = 1152, no residual, fp16 operands) on batches whose
tensors do (B <= 512: in + out <= 134 MB, re-used across the timed launches) and do not (B = 2048: 1.07 GB) stay resident in the 256-MB
Infinity Cache; and with the input rotated over 8 distinct buffers so that no launch finds its input cached.  TFLOP/s per configuration."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops

C, S = 128, 32
w = (torch.randn((C, C, 3, 3)) / 34).cuda()
wf = torch.empty(w.numel(), device="cuda", dtype=torch.float16); wd = torch.empty_like(wf)
ops.pack_conv_weight(w, wf, wd)
for B in (256, 512, 1024, 2048, 4096):
    nbuf = max(1, min(8, 8192 // B))
    xs = [torch.randn((B, S, S, C), device="cuda").half() for _ in range(nbuf)]
    flops = 2.0 * B * S * S * C * C * 9
    for label, pick in (("same buffer", lambda i: xs[0]), (f"{nbuf} rotating buffers", lambda i: xs[i % nbuf])):
        for _ in range(3):
            ops.conv_igemm([pick(0)], wf, C, 3, ops.NORMAL, (S, S))
        torch.cuda.synchronize()
        n = 24
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            ops.conv_igemm([pick(i)], wf, C, 3, ops.NORMAL, (S, S))
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        print(f"B={B:5d} {label:22s}: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TFLOP/s  (tiles per CU {B * S * S / 256 / 256:.1f})", flush=True)
