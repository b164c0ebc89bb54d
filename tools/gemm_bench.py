"""fp32 GEMM (gmk_gemm_f32: the embedding MLPs and the 12 emb_layers of the train step) at the step's shapes: time and TFLOP/s per call,
and a checksum of the result to compare builds (the MFMA sequence per accumulator is fixed: results must not change)."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


g = torch.Generator().manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tot = 0.0
for name, (M, K, N, tA, tB) in {"time_embed.0": (B, 64, 256, False, True), "mlp.2": (B, 256, 256, False, True), "emb_all": (B, 256, 1536, False, True),
                                 "dWcat": (1536, B, 256, True, False), "dsemb": (B, 1536, 256, False, False), "dW2": (256, B, 256, True, False),
                                 "dsh": (B, 256, 256, False, False), "dW0": (256, B, 64, True, False)}.items():
    A = torch.randn((K, M) if tA else (M, K), generator=g).cuda()
    Bm = torch.randn((N, K) if tB else (K, N), generator=g).cuda()
    Av, Bv = (A.t() if tA else A), (Bm.t() if tB else Bm)
    out = ops.gemm(Av, Bv)
    ref = Av.double() @ Bv.double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    t = timed(lambda: ops.gemm(Av, Bv))
    tot += t
    print(f"{name:14s} M={M:5d} K={K:5d} N={N:5d}: {t * 1e6:7.1f} us  {2.0 * M * N * K / t / 1e12:6.1f} TFLOP/s  err {err:.1e}  checksum {float(out.double().sum()):.10e}", flush=True)
print(f"sum of the 8 shapes: {tot * 1e6:.1f} us")
