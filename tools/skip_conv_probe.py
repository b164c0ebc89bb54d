"""1x1 skip convolution (two 128-channel sources -> 128) and its data gradient (128 -> 256 as two halves) at 28x28 / 14x14:
time per kernel choice (GMK_CONV_KERNEL: 0 automatic = LDS-DMA im2col for these sizes, 1 register-staged) against the bytes moved."""
import sys, torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib
T = torch.bfloat16
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for S in (28, 14):
    B, C = 1024, 128
    srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(2)]
    dy = torch.randn((B, S, S, C), device="cuda").to(T)
    w = torch.randn((C, 2 * C, 1, 1), device="cuda") / 16
    wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    big = torch.empty(3 * 10**8, device="cuda", dtype=torch.int8)       # 300 MB to flush the Infinity Cache between launches
    for choice in (0, 1):
        lib.gmk_set_kernel_choice(choice, -1, -1)
        def fwd(): big.fill_(1); ops.conv_igemm(srcs, wf, C, 1, ops.NORMAL, (S, S))
        def dg(): big.fill_(1); ops.conv_igemm([dy], wd, 2 * C, 1, ops.NORMAL, (S, S), n0=0); ops.conv_igemm([dy], wd, 2 * C, 1, ops.NORMAL, (S, S), n0=C)
        def flush(): big.fill_(1)
        tf = timed(flush)
        t1, t2 = timed(fwd) - tf, timed(dg) - tf
        mb = B * S * S * C * 2 / 1e6
        print(f"S={S} choice {choice} (kernel id {lib.gmk_last_kernel()}): fwd {t1*1e6:7.1f} us ({3*mb/t1/1e6:5.2f} TB/s of 3 tensors)   dgrad 2 halves {t2*1e6:7.1f} us ({4*mb/t2/1e6:5.2f} TB/s of 4 tensor passes)")
lib.gmk_set_kernel_choice(-1, -1, -1)
