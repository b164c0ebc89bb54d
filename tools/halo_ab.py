"""3x3 halo convolution A/B: GMK_DEV_VARIANT 0 (v_mfma_f32_32x32x16_bf16 consumers) vs 16 (v_mfma_f32_16x16x32_bf16 consumers).
Parity of both against fp32 torch (bias, residual, two sources, nearest-x2 source, ragged batches, half-job tails), then interleaved
in-process timing at the train step's shapes.  python tools/halo_ab.py"""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

def run(variant, srcs, wf, mode, out_hw, **kw):
    lib.gmk_set_dev_variant(variant)
    lib.gmk_set_kernel_choice(3, -1, -1)            # force the halo kernels (also below the automatic tile threshold)
    o = ops.conv_igemm(srcs, wf, 128, 3, mode, out_hw, **kw)
    k = lib.gmk_last_kernel()
    lib.gmk_set_kernel_choice(-1, -1, -1)
    return o, k

torch.manual_seed(0)
cases = [(3, 12, 12, 1, False, True, True), (2, 28, 28, 2, False, True, False), (5, 7, 7, 1, False, False, True), (1, 64, 64, 1, False, True, True),
         (3, 16, 16, 1, True, True, False), (300, 14, 14, 1, False, True, True), (37, 28, 28, 2, False, False, True), (3, 32, 32, 2, False, True, True)]
for B, H, W, nsrc, up, use_bias, use_res in cases:
    hs, ws = (H // 2, W // 2) if up else (H, W)
    xs = [torch.randn(B, hs, ws, 128, device="cuda").bfloat16() for _ in range(nsrc)]
    w = torch.randn(128, 128 * nsrc, 3, 3, device="cuda") / (128 * nsrc * 9) ** 0.5
    wq = w.bfloat16().float()
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); ops.pack_conv_weight(wq, wf, None)
    bias = torch.randn(128, device="cuda") if use_bias else None
    res = torch.randn(B, H, W, 128, device="cuda").bfloat16() if use_res else None
    x = torch.cat([t.float() for t in xs], 3).permute(0, 3, 1, 2)
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    ref = F.conv2d(x, wq, bias, padding=1).permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.float()
    outs = {}
    for v in (0, 16):
        o, k = run(v, xs, wf, ops.UPSAMPLE2 if up else ops.NORMAL, (H, W), bias=bias, residual=res)
        err = float((o.float() - ref).abs().max() / ref.abs().max())
        outs[v] = o
        print(f"B={B} {H}x{W} srcs={nsrc} up={up} bias={use_bias} res={use_res} variant {v:2d} (kernel {k}): rel err {err:.2e}", "OK" if err < 1e-2 else "MISMATCH")
        assert err < 1e-2 and k == 4
    d = float((outs[0].float() - outs[16].float()).abs().max())
    print(f"      variants differ by at most {d:.3e}")
print("parity ok")
for B, S, nsrc, use_res in ((1024, 28, 1, False), (1024, 28, 1, True), (1024, 28, 2, False), (1024, 14, 1, True), (2048, 32, 1, True), (2048, 32, 2, False), (1024, 64, 1, True)):
    xs = [torch.randn(B, S, S, 128, device="cuda").bfloat16() for _ in range(nsrc)]
    w = torch.randn(128, 128 * nsrc, 3, 3, device="cuda") / (128 * nsrc * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); ops.pack_conv_weight(w, wf, None)
    bias = torch.randn(128, device="cuda")
    res = torch.randn(B, S, S, 128, device="cuda").bfloat16() if use_res else None
    flops = 2.0 * B * S * S * 128 * 128 * nsrc * 9
    t = {0: [], 16: []}
    for rnd in range(4):
        for v in (0, 16):
            run(v, xs, wf, ops.NORMAL, (S, S), bias=bias, residual=res)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run(v, xs, wf, ops.NORMAL, (S, S), bias=bias, residual=res)
            torch.cuda.synchronize(); t[v].append((time.perf_counter() - t0) / 10)
    print(f"B={B} {S}x{S} K={1152 * nsrc} res={use_res}: " + "   ".join(f"variant {v:2d}: {min(x) * 1e6:7.1f} us = {flops / min(x) / 1e12:6.1f} TFLOP/s" for v, x in t.items()))
lib.gmk_set_dev_variant(0)
