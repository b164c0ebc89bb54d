"""Randomised cross-check of the LDS-halo / slot kernels against the im2col kernels on odd shapes (tiles spanning images,
partial last tiles, tiny batches, up-sampled inputs, every epilogue combination)."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from generative_models_amd._lib import lib
T = torch.bfloat16
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nfail = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    H = rng.choice([4, 5, 7, 8, 9, 12, 14, 16, 20, 28, 32, 40, 64]); W = rng.choice([4, 6, 7, 8, 12, 14, 16, 28, 30, 32, 48, 64])
    if (H * W) > 2048: H = max(4, 2048 // W)
    B = rng.choice([1, 2, 3, 5, 8, 17]); two = rng.random() < 0.4; up = rng.random() < 0.25 and H % 2 == 0 and W % 2 == 0
    C = 128; cin = 2 * C if two else C
    hs, ws = (H // 2, W // 2) if up else (H, W)
    g = torch.Generator().manual_seed(it)
    srcs = [torch.randn((B, hs, ws, C), generator=g).cuda().to(T) for _ in range(2 if two else 1)]
    w = (torch.randn((C, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf); ops.pack_conv_weight(w, wf, wd)
    kw = {}
    if rng.random() < 0.6: kw["bias"] = torch.randn(C, generator=g).cuda()
    if rng.random() < 0.3: kw["emb"] = torch.randn((B, C), generator=g).cuda()
    if rng.random() < 0.5: kw["residual"] = torch.randn((B, H, W, C), generator=g).cuda().to(T)
    mode = ops.UPSAMPLE2 if up else ops.NORMAL
    outs = {}
    for force, variant in ((3, 0), (3, 3), (2, 0), (1, 0)):
        lib.gmk_set_kernel_choice(force, -1, -1); lib.gmk_set_dev_variant(variant)
        outs[(force, variant)] = (ops.conv_igemm(srcs, wf, C, 3, mode, (H, W), **kw).float(), lib.gmk_last_kernel())
    ref = outs[(1, 0)][0]
    scale = float(ref.abs().max())
    errs = {k: float((v[0] - ref).abs().max()) / scale for k, v in outs.items()}
    dy = torch.randn((B, H, W, C), generator=g).cuda().to(T)
    dws = {}
    for wk in (2, 1):
        lib.gmk_set_kernel_choice(-1, wk, -1)
        dw = torch.empty_like(w); ops.conv_wgrad(dy, srcs, 3, mode, dw); dws[wk] = (dw.clone(), lib.gmk_last_kernel())
    werr = float((dws[2][0] - dws[1][0]).abs().max() / dws[1][0].abs().max())
    bad = max(errs.values()) > 1.2e-2 or werr > 1.2e-2
    nfail += bad
    print(("FAIL " if bad else "ok   ") + f"B={B} {H}x{W} cin={cin} up={int(up)} {sorted(kw)} kernels={[v[1] for v in outs.values()]} "
          f"err={[round(e, 4) for e in errs.values()]} wgrad kernels={[v[1] for v in dws.values()]} err={werr:.4f}", flush=True)
lib.gmk_set_kernel_choice(-1, -1, -1); lib.gmk_set_dev_variant(0)
print("failures:", nfail)
sys.exit(1 if nfail else 0)
