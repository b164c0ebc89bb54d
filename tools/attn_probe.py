import sys, torch
sys.path.insert(0, ".")
from generative_models_amd import ops
B, C = 5, 128
for N in (64, 256):
    for sc_in in (0.5, 1.0, 1.5):
        g = torch.Generator().manual_seed(N)
        qkv = (torch.randn((B, N, 3 * C), generator=g) * sc_in).bfloat16().cuda()
        q, k, v = (qkv[:, :, i * C:(i + 1) * C].float() for i in range(3))
        scale = C ** -0.5
        Pref = torch.softmax(torch.einsum("bic,bjc->bij", q, k) * scale, dim=-1)
        oref = torch.einsum("bij,bjc->bic", Pref, v)
        err = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
        l2 = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
        for fp8 in (False, True):
            o, P = ops.attention_fwd(qkv, scale, want_p=True, fp8=fp8)
            print(f"N={N} input scale {sc_in} fp8={fp8}: o max-norm {err(o, oref):.3e} L2 {l2(o, oref):.3e}  P max-norm {err(P, Pref):.3e}")
import time
for N, B in ((256, 512), (64, 2048)):
    qkv = torch.randn((B, N, 3 * C), device="cuda").bfloat16()
    for fp8 in (False, True):
        ops.attention_fwd(qkv, C ** -0.5, fp8=fp8); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ops.attention_fwd(qkv, C ** -0.5, fp8=fp8)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"B={B} N={N} fp8={fp8}: {dt * 1e6:.1f} us = {4.0 * B * N * N * C / dt / 1e12:.1f} TFLOP/s")
    S = ops.bgemm_nt(qkv[:, :, :C], qkv[:, :, C:2 * C], out_dtype=torch.float32); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        S = ops.bgemm_nt(qkv[:, :, :C], qkv[:, :, C:2 * C], out_dtype=torch.float32)
        Pm = ops.softmax_fwd(S, C ** -0.5, torch.bfloat16)
        o3 = ops.bgemm_nt(Pm, ops.transpose_last2(qkv[:, :, 2 * C:]))
    torch.cuda.synchronize(); print(f"   three-kernel path: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
