"""GPU tests of the fp16-activation forms of the kernels (round 3).  In the 16-bit mode the forward tensors and forward weight packs
are fp16 (the precision of the reference's fp16 autocast forward, gms/diffusion/diffusion_model.py:68) and the gradients bf16; the
kernels that see both take the two types separately.  Two kinds of checks:
  * against fp32 torch on inputs pre-rounded to the storage types (tolerance 2e-3 of the output scale: one fp16 rounding is 5e-4);
  * bit identity where the arithmetic is the same by construction: a kernel that re-rounds fp16 activations to bf16 must give the
    bits of the all-bf16 kernel when the activations are bf16-representable; fused GroupNorm-apply must equal GroupNorm -> conv."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
H, BF = torch.float16, torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from generative_models_amd import ops as o
    return o


@pytest.fixture(scope="module")
def lib():
    from generative_models_amd._lib import lib as l
    return l


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def rnd(*shape, seed=0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def both16(x):
    """x rounded so that it is exactly representable in bf16 AND fp16 (bf16 values below 2^-17 are not: fp16 subnormals stop at 2^-24)."""
    x = x.to(BF).float()
    return torch.where(x.abs() < 2.0 ** -10, torch.zeros_like(x), x)


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("B,S,two,up,force,variant", [
    (3, 28, False, False, 3, 16), (3, 28, False, False, 3, 32), (2, 32, True, False, 3, 0), (3, 12, False, False, 3, 3),
    (2, 16, False, True, 3, 0), (300, 14, False, False, 0, 0), (1, 64, True, False, 3, 0), (5, 7, False, False, 1, 0),
    (2, 8, True, False, 2, 0)])
def test_conv_forward_with_fp16_operands(ops, lib, B, S, two, up, force, variant):
    """Every forward convolution kernel (halo wave-specialised with both MFMA shapes, half-job tails, 8-compute-wave, register-staged
    im2col, LDS-DMA im2col) on fp16 sources / weights / residual, output fp16, against fp32 torch."""
    C = 128
    hs = S // 2 if up else S
    srcs = [rnd(B, C, hs, hs, seed=1 + i).half().float() for i in range(2 if two else 1)]
    cin = C * len(srcs)
    w = (rnd(C, cin, 3, 3, seed=5) / math.sqrt(cin * 9)).half().float()
    bias = 0.1 * rnd(C, seed=6)
    res = rnd(B, C, S, S, seed=7).half().float()
    x = torch.cat(srcs, 1)
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    ref = F.conv2d(x, w, bias, padding=1) + res
    wf = torch.empty(w.numel(), device="cuda", dtype=H)
    ops.pack_conv_weight(w.cuda(), wf, None)
    try:
        lib.gmk_set_kernel_choice(force, -1, -1)
        lib.gmk_set_dev_variant(variant)
        out = ops.conv_igemm([nhwc(s, H) for s in srcs], wf, C, 3, ops.UPSAMPLE2 if up else ops.NORMAL, (S, S), bias=bias.cuda(),
                             residual=nhwc(res, H))
        kid = lib.gmk_last_kernel()
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)
    assert out.dtype == H
    assert kid == {1: 1, 2: 2}.get(force, 3 if variant == 3 else 4), kid
    assert rel_err(nchw(out), ref) < 2e-3


@pytest.mark.parametrize("residual", [False, True], ids=["plain", "residual-near-max"])
@pytest.mark.parametrize("kernel", ["auto-small", "im2col-reg", "im2col-dma", "halo-ws", "halo-ws-32x32", "halo-8wave", "halo-fused-gn", "skipfold", "stem"])
def test_conv_output_saturates_instead_of_overflowing(ops, lib, kernel, residual):
    """An fp16 output beyond 65504 is stored as the largest finite value, not as inf (the residual stream is unbounded; the reference would
    skip the step through its GradScaler, an unsaturated inf here would poison the next GroupNorm silently).  Saturation rests on
    MODE.FP16_OVFL, set at the top of each kernel that stores fp16 activations (gmk_common.h fp16_saturating_stores): EVERY such kernel is
    driven into overflow here - both im2col kernels, the LDS-halo kernel in its three consumer forms and with the fused GroupNorm producer,
    the folded skip convolution, the stem - at a small size (two tiles: half jobs) and at 32 tiles, with and without a residual near the
    largest finite value (round 3 covered the small im2col path only: the advisor's finding)."""
    C = 128
    B, S = (2, 8) if kernel == "auto-small" else (2, 64)             # 64-pixel rows: 4 rows per tile, 32 tiles at B = 2
    x = torch.full((B, S, S, C), 60000.0, device="cuda", dtype=H)
    w = torch.zeros(C, C, 3, 3)
    w[:, :, 1, 1] = 1.0 / 64                                   # centre tap: every output = 128 * 60000 / 64 = 120000
    wf = torch.empty(w.numel(), device="cuda", dtype=H)
    ops.pack_conv_weight(w.cuda(), wf, None)
    res = torch.full((B, S, S, C), 65000.0, device="cuda", dtype=H) if residual else None
    force, variant = {"auto-small": (0, 0), "im2col-reg": (1, 0), "im2col-dma": (2, 0), "halo-ws": (3, 0), "halo-ws-32x32": (3, 32),
                      "halo-8wave": (3, 3), "halo-fused-gn": (3, 0), "skipfold": (3, 0), "stem": (0, 0)}[kernel]
    try:
        lib.gmk_set_kernel_choice(force, -1, -1)
        lib.gmk_set_dev_variant(variant)
        if kernel == "stem":
            if residual:
                pytest.skip("the stem has no residual input")
            img = torch.full((B, 3, S, S), 1.0, device="cuda")
            ws = torch.full((C, 3, 3, 3), 10000.0, device="cuda")             # 270000 per interior output, 120000 in the corners (4 taps x 3 channels)
            out = ops.stem_fwd(img, ws, torch.zeros(C, device="cuda"), C, H)
        elif kernel == "skipfold":
            if residual:
                pytest.skip("the folded form has no residual input (the skip convolution is the residual)")
            sk = [torch.full((B, S, S, C), 30000.0, device="cuda", dtype=H) for _ in range(2)]
            wsk = torch.full((C, 2 * C, 1, 1), 1.0 / 128)                     # + 2 x 30000: 180000 in all
            wsf = torch.empty(wsk.numel(), device="cuda", dtype=H)
            ops.pack_conv_weight(wsk.cuda(), wsf, None)
            out = ops.conv3x3_skipfold(x, wf, torch.zeros(C, device="cuda"), sk, wsf, torch.zeros(C, device="cuda"))
            assert lib.gmk_last_kernel() == 7
        elif kernel == "halo-fused-gn":
            # raw source 1.0 with scale / shift tables that make silu(x * scale + shift) = silu(60000) = 60000: the producers' own fp16 store
            xr = torch.ones((B, S, S, C), device="cuda", dtype=H)
            tsc = torch.full((B, C), 60000.0, device="cuda"); tsh = torch.zeros((B, C), device="cuda")
            assert ops.conv_gn_fusable([xr])
            out = ops.conv_igemm([xr], wf, C, 3, ops.NORMAL, (S, S), residual=res, gn=(tsc, tsh))
        else:
            out = ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S), residual=res)
            want = {"auto-small": 4, "im2col-reg": 1, "im2col-dma": 2, "halo-ws": 4, "halo-ws-32x32": 4, "halo-8wave": 3}[kernel]
            assert lib.gmk_last_kernel() == want, lib.gmk_last_kernel()
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)
    assert out.dtype == H
    assert bool(torch.isfinite(out).all()) and float(out.float().max()) == 65504.0 and float(out.float().min()) == 65504.0


@pytest.mark.parametrize("B,S,two", [(40, 64, False), (33, 64, True)])
def test_fused_groupnorm_conv_fp16_is_bit_identical(ops, lib, B, S, two):
    """GroupNorm-apply + SiLU inside the convolution's producer waves (fp16 raw input -> fp16 operand) equals GroupNorm kernel -> conv."""
    C = 128
    srcs = [(rnd(B, S, S, C, seed=11 + i) * 1.3 + 0.2).to(H).cuda() for i in range(2 if two else 1)]
    cin = C * len(srcs)
    gamma = (1 + 0.1 * rnd(cin, seed=13)).cuda(); beta = (0.1 * rnd(cin, seed=14)).cuda()
    xadd = rnd(B, cin, seed=15).cuda() if not two else None
    w = rnd(C, cin, 3, 3, seed=16) / math.sqrt(cin * 9)
    wf = torch.empty(w.numel(), device="cuda", dtype=H)
    ops.pack_conv_weight(w.cuda(), wf, None)
    gpc = 32 // len(srcs)
    assert ops.conv_gn_fusable(srcs)
    tsc = torch.empty((B, cin), device="cuda"); tsh = torch.empty_like(tsc)
    ys = []
    for i, s in enumerate(srcs):
        sl = slice(i * C, (i + 1) * C)
        ops.gn_stats(s, gamma[sl], beta[sl], gpc, tsc[:, sl], tsh[:, sl], xadd=xadd)
        ys.append(ops.gn_silu_fwd(s, gamma[sl], beta[sl], gpc, xadd=xadd)[0])
    fused = ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), gn=(tsc, tsh))
    assert lib.gmk_last_kernel() == 4
    plain = ops.conv_igemm(ys, wf, C, 3, ops.NORMAL, (S, S))
    assert torch.equal(fused, plain)


@pytest.mark.parametrize("C,G,S", [(128, 32, 28), (128, 16, 14), (128, 32, 7), (256, 32, 8), (128, 32, 32), (128, 16, 64), (128, 32, 16), (128, 64, 16)])
def test_groupnorm_fp16_forward_and_mixed_backward(ops, C, G, S):
    """Forward fp16 -> fp16; backward with the saved input in fp16 and every gradient tensor in bf16 (register, hybrid and streaming
    kernels by size), against torch autograd on the rounded inputs."""
    B = 3
    x = (rnd(B, C, S, S, seed=21) * 1.5 + 0.3).half().float().requires_grad_(True)
    gamma = (1 + 0.1 * rnd(C, seed=22)).requires_grad_(True); beta = (0.1 * rnd(C, seed=23)).requires_grad_(True)
    xadd = 0.3 * rnd(B, C, seed=24)
    y_ref = F.silu(F.group_norm(x + xadd[:, :, None, None], G, gamma, beta, 1e-5))
    dy = rnd(B, C, S, S, seed=25).bfloat16().float(); add1 = rnd(B, C, S, S, seed=26).bfloat16().float()
    y_ref.backward(dy)
    xd = nhwc(x.detach(), H)
    y, mean, rstd = ops.gn_silu_fwd(xd, gamma.detach().cuda(), beta.detach().cuda(), G, xadd=xadd.cuda())
    assert y.dtype == H and rel_err(nchw(y), y_ref) < 2e-3
    dx, dgp, dbp = ops.gn_silu_bwd(nhwc(dy, BF), xd, gamma.detach().cuda(), beta.detach().cuda(), mean, rstd, dadd1=nhwc(add1, BF),
                                   xadd=xadd.cuda())
    assert dx.dtype == BF
    assert rel_err(nchw(dx), x.grad + add1) < 1e-2
    assert rel_err(dgp.sum(0), gamma.grad) < 5e-3 and rel_err(dbp.sum(0), beta.grad) < 5e-3
    # the same kernels on a bf16-representable x give the bits of the all-bf16 call (only the unpack differs)
    xb = both16(x.detach())
    dx1, g1, b1 = ops.gn_silu_bwd(nhwc(dy, BF), nhwc(xb, H), gamma.detach().cuda(), beta.detach().cuda(), mean, rstd)
    dx2, g2, b2 = ops.gn_silu_bwd(nhwc(dy, BF), nhwc(xb, BF), gamma.detach().cuda(), beta.detach().cuda(), mean, rstd)
    assert torch.equal(dx1, dx2) and torch.equal(g1, g2) and torch.equal(b1, b2)


@pytest.mark.parametrize("B,S,two,mode,ks,force", [
    (6, 28, False, 0, 3, 3), (4, 32, True, 0, 3, 3), (2, 64, False, 0, 3, 3), (3, 16, False, 2, 3, 3), (5, 14, True, 0, 3, 1),
    (4, 16, True, 0, 1, 1), (3, 28, False, 1, 3, 1), (600, 14, False, 0, 3, 0)])
def test_weight_gradient_with_fp16_activations(ops, lib, B, S, two, mode, ks, force):
    """dy bf16 x saved activations fp16.  The kernels re-round the activations to bf16 on their way into LDS, so on bf16-representable
    activations the result must equal the all-bf16 kernel's BIT FOR BIT (slot kernel: LOOK = 1 and 2, nearest-x2 source; im2col kernel:
    3x3, 1x1, stride 2); on general fp16 activations it must agree with fp32 torch on the re-rounded values."""
    C = 128
    hs = S // 2 if mode == 2 else S
    ho = (S - 1) // 2 + 1 if mode == 1 else S
    xs = [rnd(B, hs, hs, C, seed=31 + i) for i in range(2 if two else 1)]
    cin = C * len(xs)
    dy = rnd(B, ho, ho, C, seed=35).to(BF).cuda()
    outs = {}
    try:
        lib.gmk_set_kernel_choice(-1, force, -1)
        for tag, srcs in (("bf16", [both16(x).to(BF).cuda() for x in xs]), ("f16 exact", [both16(x).to(H).cuda() for x in xs]),
                          ("f16", [x.to(H).cuda() for x in xs])):
            dw = torch.empty(C, cin, ks, ks, device="cuda")
            ops.conv_wgrad(dy, srcs, ks, mode, dw)
            outs[tag] = (dw, lib.gmk_last_kernel())
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
    assert outs["bf16"][1] == outs["f16"][1] == (11 if force == 1 else 13)
    assert torch.equal(outs["bf16"][0], outs["f16 exact"][0])
    xr = torch.cat([x.to(H).to(BF).float().permute(0, 3, 1, 2) for x in xs], 1).requires_grad_(False)
    w = torch.zeros(C, cin, ks, ks, requires_grad=True)
    if mode == 2:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    F.conv2d(xr, w, None, stride=2 if mode == 1 else 1, padding=ks // 2).backward(dy.float().cpu().permute(0, 3, 1, 2))
    assert rel_err(outs["f16"][0], w.grad) < 1e-3


@pytest.mark.parametrize("cs,S,B", [(1, 28, 3), (3, 32, 2), (3, 64, 1)])
def test_stem_and_head_with_fp16_activations(ops, cs, S, B):
    C = 128
    x = rnd(B, cs, S, S, seed=41)
    w = rnd(C, cs, 3, 3, seed=42) / 3.0; b = 0.1 * rnd(C, seed=43)
    y = ops.stem_fwd(x.cuda(), w.cuda(), b.cuda(), C, H)
    assert y.dtype == H and rel_err(nchw(y), F.conv2d(x, w, b, padding=1)) < 2e-3
    a = rnd(B, C, S, S, seed=44).half().float()
    wh = rnd(cs, C, 3, 3, seed=45) / math.sqrt(C * 9); bh = 0.1 * rnd(cs, seed=46)
    out = ops.head_fwd(nhwc(a, H), wh.cuda(), bh.cuda())
    assert rel_err(out, F.conv2d(a, wh, bh, padding=1)) < 1e-3
    # head weight gradient: an fp16 tensor is split into bf16 hi + lo exactly (8 + 3 bits), so activations representable in both types give the
    # same bits whichever type stores them - and both agree with autograd to the split of the fp32 gradient operand (2^-17)
    dout = rnd(B, cs, S, S, seed=47).cuda()
    ab = both16(a)
    n = cs * C * 9 + cs
    g1 = ops.head_wgrad(dout, nhwc(ab, H), torch.empty(n, device="cuda")).clone()
    g2 = ops.head_wgrad(dout, nhwc(ab, BF), torch.empty(n, device="cuda")).clone()
    assert torch.equal(g1, g2)
    w0 = torch.zeros(cs, C, 3, 3, requires_grad=True); b0 = torch.zeros(cs, requires_grad=True)
    F.conv2d(ab, w0, b0, padding=1).backward(dout.cpu())
    assert rel_err(g1, torch.cat([w0.grad.reshape(-1), b0.grad])) < 2e-5
    # and with a gradient operand far below fp16's range (a large global batch's 1 / B): nothing is flushed
    tiny = ops.head_wgrad(dout * 1e-9, nhwc(ab, H), torch.empty(n, device="cuda")).clone()
    assert rel_err(tiny * 1e9, g1) < 2e-5


def test_cast16_round_trip(ops):
    x = (rnd(4, 8, 8, 128, seed=51) * 3).to(H).cuda()
    b = ops.cast16(x, BF)
    assert b.dtype == BF and torch.equal(b, x.to(BF))
    assert torch.equal(ops.cast16(b, H), b.to(H))
    big = torch.full((8,), 1e6, device="cuda", dtype=BF)
    assert float(ops.cast16(big, H).float().max()) == 65504.0


def test_net_defaults_to_fp16_activations_and_keeps_the_bf16_path():
    from generative_models_amd.diffusion.simple_unet import SimpleUnet
    n = SimpleUnet(128, 0.0)
    assert n.compute_dtype == BF and n.act_dtype == H
    assert SimpleUnet(128, 0.0, act_dtype=BF).act_dtype == BF
    assert SimpleUnet(128, 0.0, compute_dtype=torch.float32).act_dtype == torch.float32
    with pytest.raises(ValueError):
        SimpleUnet(128, 0.0, compute_dtype=torch.float32, act_dtype=H)
    n = n.cuda()
    g = torch.Generator().manual_seed(0)
    z = torch.randn((2, 1, 16, 16), generator=g).cuda(); l = torch.tensor([0.5, -2.0]).cuda()
    ctx = {}
    out = n.forward_hip(z, l, None, None, ctx=ctx)
    srcs, a, _, h, a2, _ = ctx["down.seq.1"]
    assert srcs[0].dtype == a[0].dtype == h.dtype == a2.dtype == H and out.dtype == torch.float32
    assert n._packs["down.seq.1.in_layers.2"][0].dtype == H and n._packs["down.seq.1.in_layers.2"][1].dtype == BF
    n.backward_hip(ctx, torch.randn_like(out))
    assert bool(torch.isfinite(n.flat_grads).all())
