"""GPU parity tests, per kernel family: libgmk.so (through the C ABI) vs torch-CPU fp32 restatements of the same op
(the reference's arithmetic is stock torch ops).  Tolerances: fp32 path 1e-3 of the output
scale (north_star), bf16 path 1e-2 of the output scale with inputs pre-rounded to bf16; integer / index work
(one-hot, masks, sampler select) bit-exact."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from generative_models_amd import ops as o
    return o


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def q(x, dtype):
    """round a CPU fp32 tensor through `dtype`"""
    return x.to(dtype).float()


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,G,S", [(128, 32, 28), (128, 16, 14), (128, 32, 7), (256, 32, 8), (256, 16, 4), (128, 32, 32), (128, 32, 64),
                                   (128, 64, 12), (128, 128, 8), (128, 64, 32)])      # the last three: 2 / 1 channels per group (narrow widths)
def test_gn_silu_fwd_bwd(ops, dtype, C, G, S):
    B = 3
    x = q(rnd(B, C, S, S, seed=1) * 1.5 + 0.3, dtype).requires_grad_(True)
    gamma = (1 + 0.1 * rnd(C, seed=2)).requires_grad_(True)
    beta = (0.1 * rnd(C, seed=3)).requires_grad_(True)
    y_ref = F.silu(F.group_norm(x, G, gamma, beta, 1e-5))
    dy = q(rnd(B, C, S, S, seed=4), dtype)
    add1 = q(rnd(B, C, S, S, seed=5), dtype)
    y_ref.backward(dy)
    xd = nhwc(x.detach(), dtype)
    y, mean, rstd = ops.gn_silu_fwd(xd, gamma.detach().cuda(), beta.detach().cuda(), G)
    assert rel_err(nchw(y), y_ref) < TOL[dtype]
    m_ref = x.detach().reshape(B, G, -1).mean(-1)
    assert rel_err(mean, m_ref) < 1e-4
    dxsum = torch.zeros((B, C + 128), device="cuda")[:, 64:64 + C]
    dx, dgp, dbp = ops.gn_silu_bwd(nhwc(dy, dtype), xd, gamma.detach().cuda(), beta.detach().cuda(), mean, rstd,
                                   dadd1=nhwc(add1, dtype), dxsum=dxsum)
    dx_ref = x.grad + add1
    assert rel_err(nchw(dx), dx_ref) < TOL[dtype]
    assert rel_err(dgp.sum(0), gamma.grad) < TOL[dtype]
    assert rel_err(dbp.sum(0), beta.grad) < TOL[dtype]
    assert rel_err(dxsum, nchw(dx).sum((2, 3))) < TOL[dtype]      # fp32 sums of the unrounded dx vs sums of the stored dx
    # helpers
    out = torch.empty(C, device="cuda")
    ops.colsum(dgp, out)
    assert rel_err(out, dgp.sum(0)) < 1e-5
    assert rel_err(ops.chansum(xd), nchw(xd).sum((2, 3))) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_sumpool(ops, dtype):
    x = q(rnd(2, 128, 8, 12, seed=7), dtype)
    y = ops.sumpool2x2(nhwc(x, dtype))
    ref = F.avg_pool2d(x, 2) * 4
    assert rel_err(nchw(y), ref) < TOL[dtype]


def conv_ref(mode, srcs, w, bias):
    x = torch.cat(srcs, 1)
    if mode == 0:
        return F.conv2d(x, w, bias, padding=w.shape[-1] // 2)
    if mode == 1:
        return F.conv2d(x, w, bias, stride=2, padding=1)
    if mode == 2:
        return F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    raise ValueError


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode,two,ks,S,B", [(0, False, 3, 28, 2), (0, True, 3, 14, 3), (0, True, 1, 7, 5), (1, False, 3, 28, 2),
                                             (1, False, 3, 14, 3), (2, False, 3, 7, 3), (2, False, 3, 14, 2),
                                             (0, False, 3, 7, 1), (0, False, 3, 8, 37), (0, True, 3, 64, 2), (0, False, 3, 32, 3),
                                             (2, False, 3, 32, 2)])
def test_conv_fwd_dgrad_wgrad(ops, dtype, mode, two, ks, S, B):
    C = 128
    cin = 2 * C if two else C
    srcs = [q(rnd(B, C, S, S, seed=10 + i), dtype).requires_grad_(True) for i in range(2 if two else 1)]
    w = q(rnd(C, cin, ks, ks, seed=20) / math.sqrt(cin * ks * ks), dtype).requires_grad_(True)
    bias = 0.1 * rnd(C, seed=21)
    out_ref = conv_ref(mode, srcs, w, bias)
    ho, wo = out_ref.shape[2:]
    emb = rnd(B, C, seed=22)
    res = q(rnd(B, C, ho, wo, seed=23), dtype)
    full_ref = out_ref + emb[:, :, None, None] + res
    T = dtype
    wf = torch.empty(w.numel(), device="cuda", dtype=T)
    wd = torch.empty(w.numel(), device="cuda", dtype=T)
    ops.pack_conv_weight(w.detach().cuda(), wf, wd)
    sd = [nhwc(s.detach(), T) for s in srcs]
    embd = torch.zeros((B, 3 * C), device="cuda")
    embd[:, C:2 * C] = emb.cuda()
    out = ops.conv_igemm(sd, wf, C, ks, mode, (ho, wo), bias=bias.cuda(), emb=embd[:, C:2 * C], residual=nhwc(res, T))
    assert rel_err(nchw(out), full_ref) < TOL[dtype], "forward"
    # backward
    dy = q(rnd(B, C, ho, wo, seed=30), dtype)
    out_ref.backward(dy)
    dyd = nhwc(dy, T)
    # data gradient
    for i, s in enumerate(srcs):
        if mode == 0:
            dx = ops.conv_igemm([dyd], wd, cin, ks, 0, (S, S), n0=i * C)
        elif mode == 1:
            dx = ops.conv_igemm([dyd], wd, cin, ks, 3, (S, S), n0=i * C)
        else:
            dx = ops.sumpool2x2(ops.conv_igemm([dyd], wd, cin, ks, 0, (2 * S, 2 * S), n0=i * C))
        tol = TOL[dtype] * (2 if (mode == 2 and dtype == torch.bfloat16) else 1)   # one extra bf16 rounding before the pool
        assert rel_err(nchw(dx), s.grad) < tol, f"dgrad src{i}"
    # weight gradient
    dw = torch.empty_like(w.detach()).cuda()
    ops.conv_wgrad(dyd, sd, ks, mode, dw)
    assert rel_err(dw, w.grad) < TOL[dtype], "wgrad"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cs,S,B", [(1, 28, 3), (3, 16, 2), (3, 14, 3), (2, 32, 1), (1, 7, 5), (4, 8, 2)])
def test_stem_head(ops, dtype, cs, S, B):
    C = 128
    x = rnd(B, cs, S, S, seed=40).clamp(-1, 1)
    w = (rnd(C, cs, 3, 3, seed=41) / 3).requires_grad_(True)
    b = 0.1 * rnd(C, seed=42)
    y_ref = F.conv2d(x, w, b, padding=1)
    y = ops.stem_fwd(x.cuda(), w.detach().cuda(), b.cuda(), C, dtype)
    assert rel_err(nchw(y), y_ref) < TOL[dtype]
    dy = q(rnd(B, C, S, S, seed=43), dtype)
    y_ref.backward(dy)
    dw = torch.empty_like(w.detach()).cuda()
    ops.stem_wgrad(x.cuda(), nhwc(dy, dtype), dw)
    assert rel_err(dw, w.grad) < TOL[dtype]
    # head
    a = q(rnd(B, C, S, S, seed=44), dtype).requires_grad_(True)
    wh = (rnd(cs, C, 3, 3, seed=45) / math.sqrt(9 * C)).requires_grad_(True)
    bh = (0.1 * rnd(cs, seed=46)).requires_grad_(True)
    o_ref = F.conv2d(a, wh, bh, padding=1)
    o = ops.head_fwd(nhwc(a.detach(), dtype), wh.detach().cuda(), bh.detach().cuda())
    assert rel_err(o, o_ref) < TOL[dtype]
    do = rnd(B, cs, S, S, seed=47)
    o_ref.backward(do)
    da = ops.head_dgrad(do.cuda(), wh.detach().cuda(), dtype)
    assert rel_err(nchw(da), a.grad) < TOL[dtype]
    dwb = torch.empty(cs * C * 9 + cs, device="cuda")
    ops.head_wgrad(do.cuda(), nhwc(a.detach(), dtype), dwb)
    assert rel_err(dwb[:cs * C * 9].view_as(wh), wh.grad) < TOL[dtype]
    assert rel_err(dwb[cs * C * 9:], bh.grad) < 1e-3


def test_stem_head_random_shapes(ops):
    """The four stem / head launches of the 16-bit mode on 24 random shapes (non-square, rows that fill no 64- or 128-pixel tile, widths on
    both sides of the tiled kernels' W <= 128, 1 - 3 image channels, batches of 1 - 6) against autograd on the same rounded operands."""
    import random
    rng = random.Random(7)
    C = 128
    for case in range(24):
        cs, B = rng.randint(1, 3), rng.randint(1, 6)
        H, W = 2 * rng.randint(2, 24), 2 * rng.randint(2, 70 if case % 6 == 0 else 24)
        if H * W < 32:
            H = 8
        x = rnd(B, cs, H, W, seed=100 + case).clamp(-1, 1)
        w = (rnd(C, cs, 3, 3, seed=200 + case) / 3).requires_grad_(True)
        b = 0.1 * rnd(C, seed=300 + case)
        tag = (case, B, cs, H, W)
        y_ref = F.conv2d(x, w, b, padding=1)
        y = ops.stem_fwd(x.cuda(), w.detach().cuda(), b.cuda(), C, torch.float16)
        assert rel_err(nchw(y), y_ref) < 1e-3, tag
        dy = q(rnd(B, C, H, W, seed=400 + case), torch.bfloat16)
        y_ref.backward(dy)
        dw = torch.empty_like(w.detach()).cuda()
        ops.stem_wgrad(x.cuda(), nhwc(dy, torch.bfloat16), dw)
        assert rel_err(dw, w.grad) < 2e-5, tag
        a = rnd(B, C, H, W, seed=500 + case).half().float().requires_grad_(True)
        wh = (rnd(cs, C, 3, 3, seed=600 + case) / math.sqrt(9 * C)).requires_grad_(True)
        bh = (0.1 * rnd(cs, seed=700 + case)).requires_grad_(True)
        o_ref = F.conv2d(a, wh, bh, padding=1)
        do = rnd(B, cs, H, W, seed=800 + case) / 64
        o_ref.backward(do)
        da = ops.head_dgrad(do.cuda(), wh.detach().cuda(), torch.bfloat16)
        assert rel_err(nchw(da), a.grad) < 5e-3, tag                    # bf16 result: one ulp of the largest entry
        dwb = torch.empty(cs * C * 9 + cs, device="cuda")
        ops.head_wgrad(do.cuda(), nhwc(a.detach(), torch.float16), dwb)
        assert rel_err(dwb[:cs * C * 9].view_as(wh), wh.grad) < 2e-5, tag
        assert rel_err(dwb[cs * C * 9:], bh.grad) < 1e-4, tag                 # a cancelling fp32 sum in another order


def test_embedding_path(ops, golden):
    g = golden("schedule.npz")
    t = torch.from_numpy(g["temb_t"]).cuda()
    te = ops.timestep_embedding(t, ops.timestep_freqs(256, "cuda"))
    assert rel_err(te, torch.from_numpy(g["temb_256"])) < 1e-5
    w = torch.from_numpy(g["temb_w"]).cuda()
    assert rel_err(ops.timestep_embedding(w, ops.timestep_freqs(4, "cuda")), torch.from_numpy(g["temb_4"])) < 1e-5
    # integer label path is bit-exact (row I2)
    guide = torch.tensor([3, -1, 0, 9, -1, 5], device="cuda")
    oh, keep = ops.guide_onehot(guide)
    gg = guide.cpu().clone(); mask = gg == -1; gg[mask] = 0
    assert torch.equal(oh.cpu(), F.one_hot(gg, 10).float())
    assert torch.equal(keep.cpu(), (~mask).float())
    # strided GEMM in all three layouts + fused SiLU / bias / row mask / accumulate
    A, Bm = rnd(37, 70, seed=50), rnd(70, 45, seed=51)
    bias, rs = rnd(45, seed=52), (rnd(37, seed=53) > 0).float()
    ref = rs[:, None] * (F.silu(A) @ Bm + bias)
    out = ops.gemm(A.cuda(), Bm.cuda(), bias=bias.cuda(), rowscale=rs.cuda(), silu_a=True)
    assert rel_err(out, ref) < 1e-5
    out2 = ops.gemm(A.t().contiguous().cuda().t(), Bm.t().contiguous().cuda().t(), out=out.clone(), accumulate=True)
    assert rel_err(out2, ref + A @ Bm) < 1e-5
    out3 = ops.gemm(A.cuda().t(), A.cuda(), silu_b=True)
    assert rel_err(out3, A.t() @ F.silu(A)) < 1e-5
    # long-K shapes take the deterministic split-K path; batched deferred column sums
    A2, B2 = rnd(1024, 200, seed=56), rnd(1024, 70, seed=57)
    assert rel_err(ops.gemm(A2.cuda().t(), B2.cuda(), silu_b=True), A2.t() @ F.silu(B2)) < 1e-5
    parts = [rnd(300 + 7 * k, 40 + 11 * k, seed=60 + k).cuda() for k in range(19)]
    outs = [torch.empty(p_.shape[1], device="cuda") for p_ in parts]
    for p_, o_ in zip(parts, outs):
        ops.colsum(p_, o_, defer=True)
    ops.flush_colsums()
    for p_, o_ in zip(parts, outs):
        assert rel_err(o_, p_.sum(0)) < 1e-5
    pre = rnd(9, 33, seed=54).requires_grad_(True)
    d = rnd(9, 33, seed=55)
    F.silu(pre).backward(d)
    assert rel_err(ops.silu_bwd(d.cuda(), pre.detach().cuda()), pre.grad) < 1e-5


def test_q_sample_and_loss(ops):
    from oracle import diffusion_ref as D
    B, n = 5, 784
    x = rnd(B, 1, 28, 28, seed=60).clamp(-1, 1)
    x[:, :, :7] = -1.0
    eps = rnd(B, 1, 28, 28, seed=61)
    u = torch.tensor([0.0, 0.03, 0.4, 0.77, 0.999])
    logsnr, z = ops.q_sample(x.cuda(), eps.cuda(), u.cuda())
    l_ref = D.logsnr_schedule_cosine(u)
    assert rel_err(logsnr, l_ref) < 1e-5
    assert rel_err(z, D.q_sample(x, l_ref, eps)) < 1e-5
    v = rnd(B, 1, 28, 28, seed=62).requires_grad_(True)
    out = D.model_outputs(v, z.cpu(), logsnr.cpu())
    x_mse = (out["model_x"] - x).square().flatten(1).mean(1)
    e_mse = (out["model_eps"] - eps).square().flatten(1).mean(1)
    loss = torch.maximum(x_mse, e_mse)
    (loss.sum() * 0.25).backward()
    lb, xm, em, dv = ops.v_loss(v.detach().cuda(), z, x.cuda(), eps.cuda(), logsnr, grad_scale=0.25)
    assert rel_err(lb, loss) < 1e-4 and rel_err(xm, x_mse) < 1e-4 and rel_err(em, e_mse) < 1e-4
    assert rel_err(dv, v.grad) < 1e-4


@pytest.mark.parametrize("mode", ["ddim", "cfg", "noisy"])
def test_sampler_step(ops, mode):
    from oracle import diffusion_ref as D
    B = 4
    z = rnd(B, 1, 12, 12, seed=70)
    v = rnd(B, 1, 12, 12, seed=71)
    vu = rnd(B, 1, 12, 12, seed=72)
    w = torch.tensor([0.0, 0.5, 2.0, 3.9])
    noise = rnd(B, 1, 12, 12, seed=73)
    for (i, T) in [(7, 8), (3, 8), (0, 8), (199, 200)]:
        u_t, u_s = D.sampler_times(i, T)
        lt = D.logsnr_schedule_cosine(torch.tensor(u_t)); ls = D.logsnr_schedule_cosine(torch.tensor(u_s))
        ltv = lt.expand(B)
        o = D.model_outputs(v, z, ltv)
        xp, ep = o["model_x"], o["model_eps"]
        if mode == "cfg":
            ou = D.model_outputs(vu, z, ltv)
            ww = D.bcast(w, z.shape)
            e = (1 + ww) * ep + (-ww) * ou["model_eps"]
            xp = torch.clip(D.predict_x_from_eps(z, e, ltv), -1, 1)
            ep = D.predict_eps_from_x(z, xp, ltv)
        if mode == "noisy":
            alpha_st = torch.sqrt((1 + torch.exp(-lt)) / (1 + torch.exp(-ls)))
            r = torch.exp(lt - ls); omr = -torch.expm1(lt - ls)
            zs = r * alpha_st * z + omr * torch.sqrt(torch.sigmoid(ls)) * xp + torch.sqrt(omr * torch.sigmoid(-lt)) * noise
        else:
            zs = torch.sqrt(torch.sigmoid(ls)) * xp + torch.sqrt(torch.sigmoid(-ls)) * ep
        z_ref = xp if i == 0 else zs
        zn, xo, eo = ops.sampler_step(v.cuda(), z.cuda(), float(lt), float(ls), i == 0,
                                      v_uncond=vu.cuda() if mode == "cfg" else None,
                                      cond_w=w.cuda() if mode == "cfg" else None,
                                      noise=noise.cuda() if mode == "noisy" else None, want_pred=True)
        scale = 1e-4 if float(lt) > -15 else 2e-3     # c1 = sqrt(1+e^l) amplifies rounding at the noisiest steps
        assert rel_err(xo, xp) < 1e-4 and rel_err(eo, ep) < scale and rel_err(zn, z_ref) < scale
        if i == 0:
            assert torch.equal(zn, xo)      # the i == 0 select is exact (row I1)


def test_rng_and_adam(ops):
    from oracle import diffusion_ref as D
    n = 1 << 20
    a = ops.rng_normal((n,), 1234, 0, "cuda")
    b = ops.rng_normal((n,), 1234, 0, "cuda")
    assert torch.equal(a, b)
    c = ops.rng_normal((n,), 1234, n // 4, "cuda")
    assert not torch.equal(a, c)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 4).mean()) - 3) < 0.1
    u = ops.rng_uniform((n,), 99, 0, "cuda")
    assert float(u.min()) >= 0 and float(u.max()) < 1 and abs(float(u.mean()) - 0.5) < 2e-3
    # Adam vs torch.optim.Adam, 3 steps, odd length (tail path)
    p0 = rnd(1003, seed=80); p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=3e-4)
    pd = p0.clone().cuda(); m = torch.zeros_like(pd); v = torch.zeros_like(pd)
    for step in range(1, 4):
        g = rnd(1003, seed=80 + step) * 0.01
        p.grad = g.clone(); opt.step()
        ops.adam_step(pd, (g * 8).cuda(), m, v, 3e-4, 0.9, 0.999, 1e-8, step, grad_scale=0.125)
        assert float((pd.cpu() - p.detach()).abs().max()) < 1e-6 * step + 1e-7


@pytest.mark.parametrize("two,S,B,mode", [(False, 28, 2, 0), (True, 14, 3, 0), (False, 64, 2, 0), (True, 64, 1, 0), (False, 32, 2, 2),
                                          (False, 8, 5, 0)])
def test_forced_kernel_variants_agree_with_torch(ops, two, S, B, mode):
    """Small problems pick the im2col kernels automatically: force the slot weight-gradient kernel (both window widths: W <= 61
    and the wide one for 64-pixel rows) and both LDS-halo convolution kernels on them and hold them to the same parity bar."""
    from generative_models_amd._lib import lib
    dtype, C = torch.bfloat16, 128
    cin = 2 * C if two else C
    srcs = [q(rnd(B, C, S, S, seed=110 + i), dtype).requires_grad_(True) for i in range(2 if two else 1)]
    w = q(rnd(C, cin, 3, 3, seed=120) / math.sqrt(cin * 9), dtype).requires_grad_(True)
    out_ref = conv_ref(mode, srcs, w, torch.zeros(C))
    ho, wo = out_ref.shape[2:]
    dy = q(rnd(B, C, ho, wo, seed=130), dtype)
    out_ref.backward(dy)
    sd = [nhwc(t.detach(), dtype) for t in srcs]
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w.detach().cuda(), wf, wd)
    try:
        lib.gmk_set_kernel_choice(-1, 2, -1)                       # slot weight-gradient kernel
        dw = torch.empty_like(w.detach()).cuda()
        ops.conv_wgrad(nhwc(dy, dtype), sd, 3, mode, dw)
        assert lib.gmk_last_kernel() == 12
        assert rel_err(dw, w.grad) < TOL[dtype], "slot wgrad"
        for variant, kid in ((0, 4), (3, 3)):                      # wave-specialised / 8-compute-wave halo kernel
            lib.gmk_set_kernel_choice(3, -1, -1)
            lib.gmk_set_dev_variant(variant)
            out = ops.conv_igemm(sd, wf, C, 3, mode, (ho, wo))
            assert lib.gmk_last_kernel() == kid
            assert rel_err(nchw(out), out_ref) < TOL[dtype], f"halo forward variant {variant}"
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)


@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("S,B,C", [(28, 2, 128), (14, 3, 128), (32, 2, 128), (16, 5, 128), (8, 3, 128), (64, 1, 128), (32, 40, 128), (32, 300, 128),
                                   (16, 1100, 128), (26, 7, 128), (16, 9, 256), (32, 150, 256)])
def test_stride2_wgrad_on_parity_planes(ops, xdt, S, B, C):
    """Weight gradient of `Downsample`'s stride-2 convolution (reference simple_unet.py:81,97,100) on the slot kernel's four-plane form (round 6):
    the input's parity planes against the low-resolution gradient's slots, every tap a constant slot offset into one plane.  Against autograd in
    fp32 on the rounded operands, against the im2col kernel it replaces (the same bf16 x bf16 products, fp32 sums in another order),
    bit-reproducible; small problems take it only when forced, the train step's sizes automatically; odd input sizes keep the im2col kernel."""
    from generative_models_amd._lib import lib
    x = q(rnd(B, C, S, S, seed=340), xdt)
    w = (rnd(C, C, 3, 3, seed=341) / math.sqrt(C * 9)).requires_grad_(True)
    out = F.conv2d(q(x, torch.bfloat16), w, None, stride=2, padding=1)               # the kernels multiply bf16(x)
    dy = q(rnd(*out.shape, seed=342), torch.bfloat16)
    out.backward(dy)
    xd, dyd = nhwc(x, xdt), nhwc(dy, torch.bfloat16)
    dw = torch.empty((C, C, 3, 3), device="cuda")
    big = B * (S // 2 + 1) ** 2 >= 65536 * 128 // C          # >= 8 chunks of 64 slots for each of the 256 / (C / 64) splits the planner asks for
    try:
        if not big:
            ops.conv_wgrad(dyd, [xd], 3, ops.STRIDE2, dw)
            assert lib.gmk_last_kernel() == 11              # automatic: the im2col kernel
            lib.gmk_set_kernel_choice(-1, 3, -1)
        ops.conv_wgrad(dyd, [xd], 3, ops.STRIDE2, dw)
        assert lib.gmk_last_kernel() == 17
        e = rel_err(dw, w.grad)
        assert e < TOL[torch.bfloat16], e
        dw2 = torch.empty_like(dw)
        ops.conv_wgrad(dyd, [xd], 3, ops.STRIDE2, dw2)
        assert torch.equal(dw, dw2)
        lib.gmk_set_kernel_choice(-1, 1, -1)
        dw_old = torch.empty_like(dw)
        ops.conv_wgrad(dyd, [xd], 3, ops.STRIDE2, dw_old)
        assert lib.gmk_last_kernel() == 11
        assert rel_err(dw, dw_old) < 1e-4, rel_err(dw, dw_old)
        assert e <= rel_err(dw_old, w.grad) + 1e-4
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
    if S == 26:                                              # 13 x 13 -> 7 x 7: not twice the output
        xo = nhwc(q(rnd(B, C, 13, 13, seed=343), xdt), xdt)
        dyo = nhwc(q(rnd(B, C, 7, 7, seed=344), torch.bfloat16), torch.bfloat16)
        lib.gmk_set_kernel_choice(-1, 3, -1)
        try:
            ops.conv_wgrad(dyo, [xo], 3, ops.STRIDE2, dw2)
            assert lib.gmk_last_kernel() == 11
        finally:
            lib.gmk_set_kernel_choice(-1, -1, -1)


@pytest.mark.parametrize("S,B", [(28, 3), (14, 5), (32, 2), (64, 1), (8, 37)])
def test_transposed_dgrad_on_halo_kernels(ops, S, B):
    """Data gradient of the stride-2 conv, the older form (GMK_CONV_KERNEL=3): on the halo kernels as a 3x3 conv of the zero-stuffed gradient
    (kernel id 5).  Force that path on small problems (tiles spanning several images at 8x8 / 14x14) and compare with autograd
    and with the im2col gather it replaces."""
    from generative_models_amd._lib import lib
    dtype, C = torch.bfloat16, 128
    x = q(rnd(B, C, S, S, seed=140), dtype).requires_grad_(True)
    w = q(rnd(C, C, 3, 3, seed=141) / math.sqrt(C * 9), dtype).requires_grad_(True)
    out_ref = F.conv2d(x, w, None, stride=2, padding=1)
    dy = q(rnd(*out_ref.shape, seed=142), dtype)
    out_ref.backward(dy)
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w.detach().cuda(), wf, wd)
    dyd = nhwc(dy, dtype)
    try:
        lib.gmk_set_kernel_choice(1, -1, -1)                        # im2col gather (register-staged)
        dx_im2col = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (S, S))
        assert lib.gmk_last_kernel() == 1
        for variant in (0, 3):                                      # wave-specialised / 8-compute-wave halo kernel
            lib.gmk_set_kernel_choice(3, -1, -1)
            lib.gmk_set_dev_variant(variant)
            dx = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (S, S))
            assert lib.gmk_last_kernel() == 5
            assert rel_err(nchw(dx), x.grad) < TOL[dtype], f"variant {variant} vs autograd"
            assert rel_err(nchw(dx), nchw(dx_im2col).cpu()) < 8e-3, f"variant {variant} vs im2col"   # same products, other summation order: <= 2 bf16 ulps
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,B,res", [(28, 3, True), (14, 5, False), (32, 2, True), (64, 1, False), (8, 37, True), (4, 3, False), (16, 70, True)])
def test_transposed_dgrad_as_four_phases(ops, dtype, S, B, res):
    """Data gradient of the stride-2 conv as four output-parity phases on the LDS-DMA kernel (kernel id 6: the path of fp32 tensors and of
    problems the halo kernel declines; GMK_CONV_KERNEL=2 elsewhere) - 1 / 2 / 2 / 4 taps over the gradient's own grid, rows scattered with
    stride 2 - with and without the residual the net adds there (the skip path's gradient).  Against autograd, against the zero-stuffed
    halo form (same products, other summation order), in fp32 and bf16."""
    from generative_models_amd._lib import lib
    C = 128
    x = q(rnd(B, C, S, S, seed=150), dtype).requires_grad_(True)
    w = q(rnd(C, C, 3, 3, seed=151) / math.sqrt(C * 9), dtype).requires_grad_(True)
    out_ref = F.conv2d(x, w, None, stride=2, padding=1)
    dy = q(rnd(*out_ref.shape, seed=152), dtype)
    out_ref.backward(dy)
    r = q(rnd(B, C, S, S, seed=153), dtype) if res else None
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w.detach().cuda(), wf, wd)
    dyd = nhwc(dy, dtype)
    rd = nhwc(r, dtype) if res else None
    ref = x.grad + (r if res else 0)
    dx0 = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (S, S), residual=rd)       # automatic choice
    assert lib.gmk_last_kernel() == 6 if dtype == torch.float32 else lib.gmk_last_kernel() in (5, 6, 9)      # (9: the sub-pixel form, round 5)
    assert rel_err(nchw(dx0), ref) < TOL[dtype]
    try:
        lib.gmk_set_kernel_choice(2, -1, -1)
        dx = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (S, S), residual=rd)
        assert lib.gmk_last_kernel() == 6
        assert rel_err(nchw(dx), ref) < TOL[dtype]
        if dtype == torch.bfloat16 and S >= 8:
            lib.gmk_set_kernel_choice(3, -1, -1)
            dx_h = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (S, S), residual=rd)
            assert lib.gmk_last_kernel() == 5
            assert rel_err(nchw(dx), nchw(dx_h).cpu()) < 8e-3
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)


@pytest.mark.parametrize("S,B,two,mode", [(14, 400, False, 0), (28, 90, False, 0), (14, 330, True, 0), (16, 300, False, 2), (28, 100, False, 3)])
def test_halo_tail_runs_as_half_jobs(ops, S, B, two, mode):
    """More tiles than CUs with a last round of <= 128 tiles: the wave-specialised kernel runs that round as half-channel jobs
    (64 pixels x 64 channels per consumer wave).  Same dot products in the same order as whole jobs, so the result must equal
    the 8-compute-wave kernel's (which has no such split) bit for bit, and agree with a float32 torch convolution."""
    from generative_models_amd._lib import lib
    dtype, C = torch.bfloat16, 128
    cin = 2 * C if two else C
    hs = S // 2 if mode in (2, 3) else S
    g = torch.Generator(device="cpu").manual_seed(7)
    srcs = [torch.randn(B, hs, hs, C, generator=g).to("cuda", dtype) for _ in range(2 if two else 1)]
    w = (torch.randn(C, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    bias = 0.1 * torch.randn(C, generator=g).cuda()
    res = torch.randn(B, S, S, C, generator=g).to("cuda", dtype)
    R = 256 // S
    ntiles = (B * S + R - 1) // R
    assert ntiles > 256 and 0 < ntiles % 256 <= 128, ntiles           # the shape really has a splittable tail
    outs = {}
    try:
        for variant in (0, 3):
            lib.gmk_set_kernel_choice(3, -1, -1)
            lib.gmk_set_dev_variant(variant)
            outs[variant] = ops.conv_igemm(srcs, wf, C, 3, mode, (S, S), bias=bias, residual=res)
            assert lib.gmk_last_kernel() == (5 if mode == 3 else 4 if variant == 0 else 3)
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)
    assert torch.equal(outs[0], outs[3])
    x = torch.cat([t.float() for t in srcs], 3).permute(0, 3, 1, 2)
    wq = w.to(dtype).float()
    if mode == 2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if mode == 3:        # transposed: the packed weights are used as they are on the zero-stuffed input
        z = torch.zeros(B, C, S, S, device="cuda"); z[:, :, ::2, ::2] = x; x = z
    ref = F.conv2d(x, wq, bias, padding=1) + res.float().permute(0, 3, 1, 2)
    got = outs[0].float().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-2


@pytest.mark.parametrize("S,B,G", [(28, 3, 32), (14, 5, 32), (14, 5, 16), (8, 37, 32), (7, 9, 32)])
def test_conv_emits_groupnorm_statistics(ops, S, B, G):
    """The halo convolution's epilogue statistics == statistics of the tensor it stored (tiles spanning samples,
    32-pixel groups straddling two samples, partial last tile)."""
    from generative_models_amd._lib import lib
    C, dtype = 128, torch.bfloat16
    x = nhwc(q(rnd(B, C, S, S, seed=90), dtype), dtype)
    w = q(rnd(C, C, 3, 3, seed=91) / math.sqrt(9 * C), dtype)
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype)
    ops.pack_conv_weight(w.cuda(), wf, None)
    res = nhwc(q(rnd(B, C, S, S, seed=92) * 3, dtype), dtype)
    lib.gmk_set_kernel_choice(3, -1, -1)
    ops.GN_STATS = True
    try:
        out = ops.conv_igemm([x], wf, C, 3, 0, (S, S), bias=(0.5 * rnd(C, seed=93)).cuda(), residual=res, gn_stats=True)
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        ops.GN_STATS = False
    assert getattr(out, "_gn_stats", None) is not None
    gamma, beta = (1 + 0.1 * rnd(C, seed=94)).cuda(), (0.1 * rnd(C, seed=95)).cuda()
    y1, m1, r1 = ops.gn_silu_fwd(out, gamma, beta, G)
    plain = out.clone()                                   # same values, no attached statistics -> 3-pass kernel
    y2, m2, r2 = ops.gn_silu_fwd(plain, gamma, beta, G)
    assert rel_err(m1, m2) < 1e-4 and rel_err(r1, r2) < 1e-4
    assert rel_err(y1.float(), y2.float()) < 1e-2


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_core_ops(ops, dtype):
    """Batched K-contiguous GEMM (strided operands, ragged M/N), transpose, row softmax forward/backward vs torch."""
    g = torch.Generator().manual_seed(7)
    Bn, N, C = 3, 200, 128
    qkv = q(torch.randn((Bn, N, 3 * C), generator=g) * 0.5, dtype)
    d = nhwc if False else None
    t = qkv.cuda().to(dtype)
    qv, kv, vv = t[:, :, :C], t[:, :, C:2 * C], t[:, :, 2 * C:]
    S = ops.bgemm_nt(qv, kv, out_dtype=torch.float32)
    S_ref = torch.einsum("bik,bjk->bij", qkv[:, :, :C], qkv[:, :, C:2 * C])
    assert rel_err(S, S_ref) < TOL[dtype]
    scale = C ** -0.5
    P = ops.softmax_fwd(S, scale, dtype)
    P_ref = torch.softmax(S_ref * scale, -1)
    assert rel_err(P.float(), P_ref) < TOL[dtype]
    vT = ops.transpose_last2(vv)
    assert torch.equal(vT, vv.transpose(1, 2).contiguous())
    o = ops.bgemm_nt(P, vT)
    assert rel_err(o.float(), torch.einsum("bij,bjc->bic", P_ref, qkv[:, :, 2 * C:])) < 2 * TOL[dtype]
    # strided output: write into a column block of a wider tensor
    wide = torch.zeros((Bn, N, 3 * C), device="cuda", dtype=dtype)
    ops.bgemm_nt(P, vT, out=wide[:, :, C:2 * C], alpha=0.5)
    assert rel_err(wide[:, :, C:2 * C].float(), 0.5 * o.float()) < 1e-2 and float(wide[:, :, :C].abs().max()) == 0.0
    dP = torch.randn((Bn, N, N), generator=g)
    dS = ops.softmax_bwd(P, dP.cuda(), scale)
    Pr = P_ref.clone().requires_grad_(True)
    Sr = (S_ref.clone()).requires_grad_(True)
    torch.softmax(Sr * scale, -1).backward(dP)
    assert rel_err(dS.float(), Sr.grad) < 2 * TOL[dtype]


def test_halo_and_slot_kernels_agree_with_im2col_on_random_shapes():
    """tools/conv_stress.py: 30 random (B, H, W, cin, upsample, epilogue) problems — tiles spanning images, partial last
    tiles, single images — run through both LDS-halo kernels, the slot weight-gradient kernel and the im2col kernels."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_stress.py"), "1", "30"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_groupnorm_kernel_variants_agree_on_random_shapes():
    """tools/gn_stress.py: 30 random shapes through the register-resident forward / hybrid backward / slab kernels vs the
    whole-sample streaming kernels, with gradient addends, input addends (xadd) and dropout masks."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gn_stress.py"), "1", "30"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_pack_multi_equals_single(ops):
    """gmk_pack_conv_weights_multi (one launch for a table of convolutions) == gmk_pack_conv_weight per tensor, both dtypes."""
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 128, 3, 3), (128, 256, 1, 1), (384, 128, 1, 1), (128, 256, 3, 3)]
    sizes = [math.prod(s) for s in shapes]
    pad = lambda n: (n + 3) // 4 * 4
    offs, o = [], 5 * 4
    for n in sizes:
        offs.append(o); o += pad(n) + 8
    arena = torch.randn(o, generator=g).cuda()
    for dtype in DTYPES:
        packs = torch.zeros(2 * sum(sizes), device="cuda", dtype=dtype)
        poffs, po = [], 0
        for n in sizes:
            poffs.append(po); po += 2 * n
        ops.pack_conv_weights_multi(arena, packs, (offs, poffs, [s[0] for s in shapes], [s[1] for s in shapes], [s[2] for s in shapes]))
        for shp, n, wo, pofs in zip(shapes, sizes, offs, poffs):
            wf = torch.empty(n, device="cuda", dtype=dtype); wd = torch.empty_like(wf)
            ops.pack_conv_weight(arena[wo:wo + n].view(shp), wf, wd)
            assert torch.equal(packs[pofs:pofs + n], wf) and torch.equal(packs[pofs + n:pofs + 2 * n], wd)


@pytest.mark.parametrize("N", [64, 256])
@pytest.mark.parametrize("fp8", [False, True])
def test_fused_attention_forward(ops, N, fp8):
    """gmk_attention_fwd (BASELINE config 5: 256 tokens x 128 channels; 64 tokens at 32 x 32 inputs): o and the optional P against
    softmax(q k^T / sqrt(C)) v in fp32 on the same bf16 inputs, and against the three-kernel bf16 path it replaces.  The fp8 form
    (both contractions on the e4m3 matrix cores, fp32 accumulation; e4m3 carries 3 mantissa bits, 2^-4 per element) is held to 1e-1
    max-norm / 8e-2 L2 against that path: measured 4.7e-2 ... 8.5e-2 / 3.9e-2 ... 7.3e-2 from soft to sharp softmax (round 4's attention probe);
    the bf16 form measures 2e-3 ... 3.6e-3."""
    B, C = 5, 128
    g = torch.Generator().manual_seed(N + fp8)
    qkv = (torch.randn((B, N, 3 * C), generator=g) * 1.5).bfloat16().cuda()
    q, k, v = (qkv[:, :, i * C:(i + 1) * C].float() for i in range(3))
    scale = C ** -0.5
    Pref = torch.softmax(torch.einsum("bic,bjc->bij", q, k) * scale, dim=-1)
    oref = torch.einsum("bij,bjc->bic", Pref, v)
    o, P = ops.attention_fwd(qkv, scale, want_p=True, fp8=fp8)
    o2, none = ops.attention_fwd(qkv, scale, want_p=False, fp8=fp8)
    assert none is None and torch.equal(o, o2)
    S = ops.bgemm_nt(qkv[:, :, :C], qkv[:, :, C:2 * C], out_dtype=torch.float32)
    Pm = ops.softmax_fwd(S, scale, torch.bfloat16)
    o3 = ops.bgemm_nt(Pm, ops.transpose_last2(qkv[:, :, 2 * C:]))
    err = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
    tol = 1e-1 if fp8 else 1e-2
    assert err(o, oref) < tol and err(o, o3) < tol, (err(o, oref), err(o, o3))
    l2 = float((o.float() - oref).norm() / oref.norm())
    assert l2 < (8e-2 if fp8 else 5e-3), l2
    assert err(P, Pref) < (1e-1 if fp8 else 1e-2), err(P, Pref)
    assert float((P.float().sum(-1) - 1).abs().max()) < 2e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S", [(3, 8), (160, 32), (85, 28)])
def test_conv1x1_pair_equals_two_launches(ops, dtype, B, S):
    """gmk_conv1x1_pair (both halves of the skip connection's data gradient from one read of the output gradient) against two
    gmk_conv_igemm launches with n0 = 0 / 128: the same bits.  (160, 32) runs on the streaming kernel in the 16-bit modes and on the LDS-DMA
    kernel (two output blocks per pixel tile) in fp32, (3, 8) takes the two-launch route inside the library, (85, 28) ends in a partial tile."""
    from generative_models_amd._lib import lib
    C = 128
    dy = q(rnd(B, S, S, C, seed=70), dtype).cuda().to(dtype)
    w = (rnd(2 * C, C, seed=71) / 11).cuda().to(dtype)          # packed [1 tap][256 rows][128]
    a, b = ops.conv1x1_pair(dy, w, 2 * C)
    kernel = lib.gmk_last_kernel()
    ra = ops.conv_igemm([dy], w, 2 * C, 1, ops.NORMAL, (S, S), n0=0)
    rb = ops.conv_igemm([dy], w, 2 * C, 1, ops.NORMAL, (S, S), n0=C)
    assert torch.equal(a, ra) and torch.equal(b, rb)
    # 16-bit problems of at least two 128-pixel tiles per CU: the streaming kernel (weights resident in LDS; conv1x1_stream.hip) - the same bits
    assert kernel == (14 if dtype != torch.float32 and B * S * S >= 128 * 512 else 2 if B * S * S >= 256 * 512 else 1)
    ref = dy.float().reshape(-1, C) @ w.float().t()
    assert rel_err(torch.cat([a, b], -1).reshape(-1, 2 * C), ref.cpu()) < TOL[dtype]


@pytest.mark.parametrize("xdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,S", [(140, 32), (171, 28)])
def test_skip_conv_weight_gradient_streaming_kernel(ops, xdt, B, S):
    """Weight gradient of the 1x1 skip convolution over a concatenated input at sizes where the streaming kernel takes it (conv1x1_stream.hip: one
    workgroup per pixel range owns both 128-channel input blocks): the same bits as the im2col kernel (GMK_DEV_VARIANT=47: same split of the pixels,
    same order inside the MFMA), and autograd's values.  (171, 28): a pixel count that is no multiple of the 64-pixel step."""
    from generative_models_amd._lib import lib
    C = 128
    g = torch.Generator().manual_seed(B + S)
    xs = [torch.randn((B, S, S, C), generator=g).to(xdt).cuda() for _ in range(2)]
    dy = (torch.randn((B, S, S, C), generator=g) / 16).bfloat16().cuda()
    out = {}
    for variant in (0, 47):
        lib.gmk_set_dev_variant(variant)
        try:
            dw = torch.empty((C, 2 * C, 1, 1), device="cuda")
            ops.conv_wgrad(dy, xs, 1, ops.NORMAL, dw)
            out[variant] = (dw.clone(), lib.gmk_last_kernel())
        finally:
            lib.gmk_set_dev_variant(0)
    assert out[0][1] == 15 and out[47][1] == 11
    assert torch.equal(out[0][0], out[47][0])
    # fp16 activations are re-rounded to bf16 on their way to the MFMA (as in every weight-gradient kernel): the reference uses the same values
    xr = torch.cat([x.float().bfloat16().double() for x in xs], -1).reshape(-1, 2 * C)
    ref = dy.double().reshape(-1, C).t() @ xr
    assert rel_err(out[0][0].reshape(C, 2 * C), ref) < 1e-4


@pytest.mark.parametrize("B,S,two,res", [(40, 28, False, True), (40, 28, True, False), (33, 32, True, True), (64, 16, False, True), (9, 64, False, False)])
def test_conv_with_fused_groupnorm_is_bit_identical(ops, B, S, two, res):
    """gmk_gn_stats + gmk_conv_igemm(gn_scale, gn_shift): GroupNorm-apply + SiLU in the convolution's producer waves instead of a
    materialised normalised tensor (simple_unet.py:161-163,169-172).  Same fp32 arithmetic, same bf16 rounding -> the same bits as
    gmk_gn_silu_fwd followed by the plain convolution; statistics equal too.  Shapes: tiles that straddle two samples (28, 32),
    tile = sample (16), 64-pixel rows, one and two sources, with the per-(sample, channel) addend of conv2's GroupNorm."""
    g = torch.Generator().manual_seed(B + S)
    C = 128
    srcs = [(torch.randn((B, S, S, C), generator=g) * 1.3 + 0.2).bfloat16().cuda() for _ in range(2 if two else 1)]
    ctot = C * len(srcs)
    gamma = (1 + 0.2 * torch.randn(ctot, generator=g)).cuda(); beta = (0.3 * torch.randn(ctot, generator=g)).cuda()
    xadd = (0.5 * torch.randn((B, C), generator=g)).cuda() if not two else None
    w = torch.randn((128, ctot, 3, 3), generator=g).cuda() / (ctot * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); ops.pack_conv_weight(w, wf, None)
    bias = torch.randn(128, generator=g).cuda()
    resid = torch.randn((B, S, S, 128), generator=g).bfloat16().cuda() if res else None
    keep_min, ops.GN_FUSE_MIN_HW = ops.GN_FUSE_MIN_HW, 0          # the policy threshold (where fusing pays) is not what is tested here
    try:
        assert ops.conv_gn_fusable(srcs)
        # 14 x 14: a tile of 18 rows can touch three samples -> not fusable
        assert not ops.conv_gn_fusable([torch.zeros((40, 14, 14, C), device="cuda", dtype=torch.bfloat16)])
    finally:
        ops.GN_FUSE_MIN_HW = keep_min
    gpc = 32 // len(srcs)
    a, stats = [], []
    for i, s in enumerate(srcs):
        y, m, r = ops.gn_silu_fwd(s, gamma[i * C:(i + 1) * C], beta[i * C:(i + 1) * C], gpc, xadd=xadd)
        a.append(y); stats.append((m, r))
    ref = ops.conv_igemm(a, wf, 128, 3, ops.NORMAL, (S, S), bias=bias, residual=resid)
    tsc = torch.empty((B, ctot), device="cuda"); tsh = torch.empty_like(tsc)
    for i, s in enumerate(srcs):
        m, r = ops.gn_stats(s, gamma[i * C:(i + 1) * C], beta[i * C:(i + 1) * C], gpc, tsc[:, i * C:(i + 1) * C], tsh[:, i * C:(i + 1) * C], xadd=xadd)
        assert torch.equal(m, stats[i][0]) and torch.equal(r, stats[i][1])
    out = ops.conv_igemm(srcs, wf, 128, 3, ops.NORMAL, (S, S), bias=bias, residual=resid, gn=(tsc, tsh))
    assert torch.equal(out, ref), float((out.float() - ref.float()).abs().max())
    # and against fp32 torch
    x = torch.cat([F.silu(F.group_norm(s.float().permute(0, 3, 1, 2) + (xadd[:, :, None, None] if xadd is not None else 0), gpc,
                                       gamma[i * C:(i + 1) * C], beta[i * C:(i + 1) * C])) for i, s in enumerate(srcs)], 1)
    t = F.conv2d(x, w.bfloat16().float(), bias, padding=1).permute(0, 2, 3, 1)
    if resid is not None:
        t = t + resid.float()
    assert float((out.float() - t).abs().max() / t.abs().max()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("S,B", [(32, 2), (28, 3), (14, 5), (7, 9), (64, 1), (16, 70), (8, 37), (32, 80)])
def test_skip_convolution_folded_into_conv2(ops, dtype, S, B):
    """`skip_connection(x) + h` of the up-path ResBlocks (reference simple_unet.py:174-186) as ONE launch: conv3x3(a2) + conv1x1(cat(x, skip)) + both
    biases.  Against torch fp32 on the rounded operands (1e-2 bar of the 16-bit mode; measured far below: fp32 accumulation of 9 x 128 + 256
    products and ONE rounding) and against the two-launch path it replaces (1x1 convolution -> 16-bit tensor -> residual of conv2), which
    differs only by that intermediate rounding.  Shapes: tiles inside one image (32, 64), tiles spanning images with a ragged last tile
    (28: 252 of 256 pixels per tile; 14, 7, 8), more tiles than CUs (several jobs per workgroup: 16 x 70, 32 x 80)."""
    from generative_models_amd._lib import lib
    C = 128
    a2 = q(rnd(B, C, S, S, seed=210), dtype)
    xs = [q(rnd(B, C, S, S, seed=211 + i) * (1.0 + i), dtype) for i in range(2)]
    w = q(rnd(C, C, 3, 3, seed=220) / math.sqrt(C * 9), dtype)
    wsk = q(rnd(C, 2 * C, 1, 1, seed=221) / math.sqrt(2 * C), dtype)
    bias, bias_sk = rnd(C, seed=222) * 0.1, rnd(C, seed=223) * 0.1
    ref = F.conv2d(a2, w, bias, padding=1) + F.conv2d(torch.cat(xs, 1), wsk, bias_sk)
    ad, xd = nhwc(a2, dtype), [nhwc(t, dtype) for t in xs]
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wsf = torch.empty(wsk.numel(), device="cuda", dtype=dtype)
    ops.pack_conv_weight(w.cuda(), wf, None)
    ops.pack_conv_weight(wsk.cuda(), wsf, None)
    try:
        lib.gmk_set_kernel_choice(3, -1, -1)                      # small problems too (the dispatcher asks for 32 tiles)
        assert ops.conv_skipfold_ok(ad, xd)
        out = ops.conv3x3_skipfold(ad, wf, bias.cuda(), xd, wsf, bias_sk.cuda())
        assert lib.gmk_last_kernel() == 7
        res = ops.conv_igemm(xd, wsf, C, 1, ops.NORMAL, (S, S), bias=bias_sk.cuda())
        two = ops.conv_igemm([ad], wf, C, 3, ops.NORMAL, (S, S), bias=bias.cuda(), residual=res)
        out2 = ops.conv3x3_skipfold(ad, wf, bias.cuda(), xd, wsf, bias_sk.cuda())
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
    e_ref, e_two = rel_err(nchw(out), ref), rel_err(nchw(out), nchw(two))
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)           # spacing of the 16-bit type at the output scale
    assert e_ref < 0.75 * ulp and e_ref <= rel_err(nchw(two), ref) + 1e-6, (e_ref, rel_err(nchw(two), ref))
    assert e_two < 1.5 * ulp, e_two
    assert torch.equal(out, out2)                                   # bit-reproducible


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("S,B", [(32, 2), (16, 70), (32, 80), (16, 3), (32, 300), (28, 3), (28, 400), (8, 37), (8, 1100), (14, 330)])
def test_merged_skip_schedule_is_bit_identical_to_the_thirteen_step_one(ops, dtype, S, B):
    """Round 6: where a tile's halo fits 6 of the 7 fill pieces (with one pad column per row: 32 x 32, 28 x 28, 16 x 16, 14 x 14, 8 x 8) the folded skip convolution runs its dense sub-phases INSIDE the
    tap steps (kMerge: 9 barriers per phase, dense weights in the free seventh piece of the half-buffers, fragment sets swapping roles).  Same products in
    the same order as the 13-step schedule (GMK_DEV_VARIANT=12 keeps it): the same bits - a dense piece overwritten too early, a stale weight tile or
    a missed wait would show here.  Whole and half jobs, several jobs per workgroup, tails."""
    from generative_models_amd._lib import lib
    C = 128
    g = torch.Generator().manual_seed(S * 1000 + B)
    ad = torch.randn((B, S, S, C), generator=g).to(dtype).cuda()
    xd = [torch.randn((B, S, S, C), generator=g).to(dtype).cuda() for _ in range(2)]
    w = (torch.randn((C, C, 3, 3), generator=g) / math.sqrt(C * 9)).cuda()
    wsk = (torch.randn((C, 2 * C, 1, 1), generator=g) / math.sqrt(2 * C)).cuda()
    bias, bias_sk = (0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wsf = torch.empty(wsk.numel(), device="cuda", dtype=dtype)
    ops.pack_conv_weight(w, wf, None); ops.pack_conv_weight(wsk, wsf, None)
    outs = {}
    try:
        lib.gmk_set_kernel_choice(3, -1, -1)
        for variant in (0, 12, 0):
            lib.gmk_set_dev_variant(variant)
            outs.setdefault(variant, []).append(ops.conv3x3_skipfold(ad, wf, bias, xd, wsf, bias_sk))
            assert lib.gmk_last_kernel() == 7
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)
    assert torch.equal(outs[0][0], outs[12][0]) and torch.equal(outs[0][0], outs[0][1])
    assert bool(torch.isfinite(outs[0][0].float()).all()) and float(outs[0][0].float().abs().max()) > 0.1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("S,B,two", [(32, 2, False), (16, 70, False), (32, 80, True), (14, 400, False), (16, 300, True), (32, 300, False), (28, 90, False),
                                     (28, 400, True), (8, 37, False), (8, 1100, True)])
def test_four_slot_weight_ring_is_bit_identical_to_the_three_slot_one(ops, dtype, S, B, two):
    """Round 6: the plain 3x3 halo kernel without a residual runs with 6-piece halos and a FOURTH weight-ring slot where a tile fits 384 slots
    (one pad column per row: 32 x 32, 28 x 28, 16 x 16, 14 x 14, 8 x 8; kRing4: the weight tile of step q + 3 is issued in step q).  Same products, same order as the three-slot ring
    (GMK_DEV_VARIANT=13 keeps it): the same bits.  One and two sources (K = 1152 / 2304), whole and half jobs, tails, several jobs per workgroup."""
    from generative_models_amd._lib import lib
    C = 128
    g = torch.Generator().manual_seed(S * 1000 + B + 7)
    srcs = [torch.randn((B, S, S, C), generator=g).to(dtype).cuda() for _ in range(2 if two else 1)]
    cin = C * len(srcs)
    w = (torch.randn((C, cin, 3, 3), generator=g) / math.sqrt(cin * 9)).cuda()
    bias = (0.1 * torch.randn(C, generator=g)).cuda()
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); ops.pack_conv_weight(w, wf, None)
    outs = {}
    try:
        lib.gmk_set_kernel_choice(3, -1, -1)
        for variant in (0, 13, 0):
            lib.gmk_set_dev_variant(variant)
            outs.setdefault(variant, []).append(ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), bias=bias))
            assert lib.gmk_last_kernel() == 4
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
        lib.gmk_set_dev_variant(0)
    assert torch.equal(outs[0][0], outs[13][0]) and torch.equal(outs[0][0], outs[0][1])
    x = torch.cat([t.float() for t in srcs], 3).permute(0, 3, 1, 2)
    ref = F.conv2d(x[:2].cpu(), w.to(dtype).float().cpu(), bias.cpu(), padding=1)
    assert rel_err(nchw(outs[0][0][:2]), ref) < TOL[torch.bfloat16]


SUBPIXEL_SHAPES = [(7, 9), (8, 37), (16, 3), (16, 300), (32, 2), (14, 5), (14, 330), (32, 66),      # (low-resolution size, batch)
                   (9, 6), (10, 3), (11, 5), (15, 4), (22, 2), (30, 2)]      # odd / non-power-of-two widths: inputs of 36, 40, 44, 60, 88, 120 pixels


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("S,B", SUBPIXEL_SHAPES)
def test_upsample_conv_subpixel(ops, dtype, S, B):
    """`Upsample` (reference simple_unet.py:112-122: F.interpolate(nearest, x2), then Conv2d(C, C, 3, padding=1)) in its sub-pixel form: four output
    parities, each a 2x2-tap convolution of the LOW-resolution tensor with the 3x3 weights that land on one low-resolution pixel pre-summed in fp32
    (16 tap-products per low-resolution pixel instead of 36).  Against torch fp32 on the rounded input and the fp32 weights at the 16-bit mode's bar
    (measured far below it), against the nearest-x2 form of the halo kernel it replaces (the same sum, other rounding order), bit-reproducible.
    Shapes: 7 -> 14 ... 32 -> 64; tiles spanning images with ragged tiles (7, 14), one image per tile (16), several tiles per image (32), more
    tiles than CUs (several jobs per workgroup), fewer tiles than XCDs (16 x 16, B = 3)."""
    from generative_models_amd._lib import lib
    C = 128
    x = q(rnd(B, C, S, S, seed=300), dtype)
    w = rnd(C, C, 3, 3, seed=301) / math.sqrt(C * 9)             # fp32 master weights: the pack sums them before rounding
    bias = rnd(C, seed=302) * 0.1
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    xd = nhwc(x, dtype)
    assert ops.conv_subpixel_ok(B, S, S, C, dtype)
    wsub = torch.empty(16 * C * C, device="cuda", dtype=dtype)
    wsub_d = torch.empty(16 * C * C, device="cuda", dtype=torch.bfloat16)
    ops.pack_upsample_weight(w.cuda(), wsub, wsub_d)
    # the pack itself: [4 (2a + b) + 2 ty + tx][cout][cin] = fp32 sums of the 3x3 taps that meet one low-resolution pixel, rounded once
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    pk = wsub.float().cpu().view(2, 2, 2, 2, C, C)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    want = sum(w[:, :, y, x_] for y in rows[a][ty] for x_ in rows[b][tx])
                    assert torch.equal(pk[a, b, ty, tx], q(want, dtype))
                    assert torch.equal(wsub_d.float().cpu().view(2, 2, 2, 2, C, C)[a, b, ty, tx], q(want, torch.bfloat16).t())
    out = ops.conv_subpixel(xd, wsub, C, ops.SUBPIXEL_UPSAMPLE, bias=bias.cuda())
    assert lib.gmk_last_kernel() == 8 and out.shape == (B, 2 * S, 2 * S, C)
    e = rel_err(nchw(out), ref)
    assert e < TOL[torch.bfloat16], e
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype)
    ops.pack_conv_weight(w.cuda(), wf, None)
    old = ops.conv_igemm([xd], wf, C, 3, ops.UPSAMPLE2, (2 * S, 2 * S), bias=bias.cuda())
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    assert rel_err(nchw(out), nchw(old)) < 3 * ulp
    assert e <= rel_err(nchw(old), ref) + 0.5 * ulp                 # not worse than the form it replaces
    assert torch.equal(out, ops.conv_subpixel(xd, wsub, C, ops.SUBPIXEL_UPSAMPLE, bias=bias.cuda()))


@pytest.mark.parametrize("dtype,res,S,B", [(torch.bfloat16, r, s, b) for r in (True, False) for s, b in SUBPIXEL_SHAPES] +
                         [(torch.float16, True, 16, 3), (torch.float16, False, 14, 5)])       # (gradients are bf16 in the product)
def test_transposed_dgrad_subpixel(ops, dtype, res, S, B):
    """Data gradient of `Downsample` (reference simple_unet.py:75-84: Conv2d(C, C, 3, stride=2, padding=1)) in its sub-pixel form: output parities
    meet 1 / 2 / 2 / 4 taps of the ordinary data-gradient pack over the gradient's own grid - the products of the zero-stuffed halo form without
    its multiplications by zero.  Against autograd, with and without the residual the net adds there (the skip path's gradient), and against the
    zero-stuffed form: the same products accumulated in fp32, so the two differ by summation order only (<= 1 ulp of the 16-bit output)."""
    from generative_models_amd._lib import lib
    C = 128
    x = q(rnd(B, C, 2 * S, 2 * S, seed=310), dtype).requires_grad_(True)
    w = q(rnd(C, C, 3, 3, seed=311) / math.sqrt(C * 9), dtype).requires_grad_(True)
    out_ref = F.conv2d(x, w, None, stride=2, padding=1)
    dy = q(rnd(*out_ref.shape, seed=312), dtype)
    out_ref.backward(dy)
    r = q(rnd(B, C, 2 * S, 2 * S, seed=313), dtype) if res else None
    ref = x.grad + (r if res else 0)
    wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w.detach().cuda(), wf, wd)
    dyd = nhwc(dy, dtype)
    rd = nhwc(r, dtype) if res else None
    assert ops.conv_subpixel_ok(B, S, S, C, dtype)
    dx = ops.conv_subpixel(dyd, wd, C, ops.SUBPIXEL_TRANSPOSED, residual=rd)
    assert lib.gmk_last_kernel() == 9
    assert rel_err(nchw(dx), ref) < TOL[torch.bfloat16]
    try:
        lib.gmk_set_kernel_choice(2, -1, -1)                        # the four phase launches of the LDS-DMA kernel: the same products per output
        dx_p = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (2 * S, 2 * S), residual=rd)
        assert lib.gmk_last_kernel() == 6
    finally:
        lib.gmk_set_kernel_choice(-1, -1, -1)
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    assert rel_err(nchw(dx), nchw(dx_p)) < 1.5 * ulp
    # through the dispatcher: gmk_conv_igemm(GMK_CONV_TRANSPOSED2) takes the sub-pixel form where it is eligible
    dx_auto = ops.conv_igemm([dyd], wd, C, 3, ops.TRANSPOSED2, (2 * S, 2 * S), residual=rd)
    assert lib.gmk_last_kernel() == 9 and torch.equal(dx_auto, dx)


@pytest.mark.parametrize("dtype,S,B", [(torch.bfloat16, s, b) for s, b in SUBPIXEL_SHAPES] + [(torch.float16, 7, 9), (torch.float16, 16, 3)])
def test_upsample_dgrad_subpixel(ops, dtype, S, B):
    """Data gradient of `Upsample` (reference simple_unet.py:112-122; autograd: dgrad of the 3x3 convolution at the high resolution, then the 2x2
    sum-pool that is the backward of F.interpolate(nearest)) as ONE launch: the transpose of the sub-pixel forward - the four parity views of the
    output gradient, 2x2 taps each on the transposed pre-summed matrices, accumulated in fp32 and rounded once.  Against autograd in fp32 on the rounded
    gradient and fp32 weights, and against the two launches it replaces (which round the high-resolution intermediate to 16 bits before the pool)."""
    from generative_models_amd._lib import lib
    C = 128
    x = rnd(B, C, S, S, seed=320).requires_grad_(True)
    w = rnd(C, C, 3, 3, seed=321) / math.sqrt(C * 9)
    out = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=1)
    dy = q(rnd(*out.shape, seed=322), dtype)
    out.backward(dy)
    dyd = nhwc(dy, dtype)
    wsub_d = torch.empty(16 * C * C, device="cuda", dtype=dtype)
    ops.pack_upsample_weight(w.cuda(), None, wsub_d)
    assert ops.conv_subpixel_ok(B, S, S, C, dtype)
    dx = ops.conv_subpixel(dyd, wsub_d, C, ops.SUBPIXEL_UPSAMPLE_DGRAD)
    assert lib.gmk_last_kernel() == 10 and dx.shape == (B, S, S, C)
    e = rel_err(nchw(dx), x.grad)
    assert e < TOL[torch.bfloat16], e
    if dtype == torch.bfloat16:            # (gradients are bf16 in the product; the pooling kernel has no fp16 form)
        wf = torch.empty(w.numel(), device="cuda", dtype=dtype); wd = torch.empty_like(wf)
        ops.pack_conv_weight(w.cuda(), wf, wd)
        old = ops.sumpool2x2(ops.conv_igemm([dyd], wd, C, 3, ops.NORMAL, (2 * S, 2 * S)))
        e_old = rel_err(nchw(old), x.grad)
        ulp = 2.0 ** -7
        assert e <= e_old + 0.5 * ulp, (e, e_old)                   # one rounding instead of two: not worse than the pair it replaces
        assert rel_err(nchw(dx), nchw(old)) < 4 * ulp
    assert torch.equal(dx, ops.conv_subpixel(dyd, wsub_d, C, ops.SUBPIXEL_UPSAMPLE_DGRAD))


@pytest.mark.parametrize("xdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("S,B", [(16, 300), (8, 400), (14, 200), (7, 400), (32, 70), (16, 3), (9, 170), (15, 70), (22, 35)])      # (B = 2048 at 16 x 16: tests/test_gpu_fullsize.py)
def test_upsample_wgrad_subpixel(ops, xdt, S, B):
    """Weight gradient of `Upsample` (reference simple_unet.py:112-122) in the sub-pixel form: the 16 tap gradients of the pre-summed 2x2-tap
    matrices accumulated over the LOW-resolution slots (two dY parity streams and eight accumulators per workgroup), folded onto the 9 taps by a
    deterministic reduce.  Against autograd in fp32 on the rounded operands, against the nearest-x2 slot kernel it replaces (the same products,
    other summation order), bit-reproducible; small problems (too few slot chunks per split) are declined."""
    from generative_models_amd._lib import lib
    C = 128
    x = q(rnd(B, C, S, S, seed=330), xdt)
    w = (rnd(C, C, 3, 3, seed=331) / math.sqrt(C * 9)).requires_grad_(True)
    out = F.conv2d(F.interpolate(q(x, torch.bfloat16), scale_factor=2, mode="nearest"), w, None, padding=1)      # the kernels multiply bf16(x)
    dy = q(rnd(*out.shape, seed=332), torch.bfloat16)
    out.backward(dy)
    xd, dyd = nhwc(x, xdt), nhwc(dy, torch.bfloat16)
    ok = ops.conv_wgrad_subpixel_ok(B, S, S, C, torch.bfloat16)
    if (S, B) == (16, 3):
        assert not ok
        return
    assert ok
    dw = torch.empty((C, C, 3, 3), device="cuda")
    ops.conv_wgrad_subpixel(dyd, xd, dw)
    assert lib.gmk_last_kernel() == 16
    e = rel_err(dw, w.grad)
    assert e < TOL[torch.bfloat16], e
    dw_old = torch.empty_like(dw)
    ops.conv_wgrad(dyd, [xd], 3, ops.UPSAMPLE2, dw_old)
    assert rel_err(dw, dw_old) < 1e-4, rel_err(dw, dw_old)            # the same bf16 x bf16 products, fp32 sums in another order
    assert e <= rel_err(dw_old, w.grad) + 1e-4
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad_subpixel(dyd, xd, dw2)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("N", [64, 128, 256])
@pytest.mark.parametrize("sharp", [1.0, 3.0])
def test_fused_attention_backward(ops, N, sharp):
    """gmk_attention_bwd (BASELINE configs[4]'s block, round 5): (dq | dk | dv) of o = softmax(q k^T / sqrt(C)) v with P recomputed block by block in
    registers - no N x N matrix in HBM - against torch autograd in fp32 on the same bf16 inputs, and against the three-kernel path it replaces
    (batched GEMMs + softmax backward on a stored bf16 P, fp32 dP): measured closer to autograd than that path (P is not rounded to bf16 before dV
    and dS), from soft to sharp softmax; deterministic."""
    B, C = 5, 128
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn((B, N, 3 * C), generator=g) * sharp).bfloat16()
    do = torch.randn((B, N, C), generator=g).bfloat16()
    scale = C ** -0.5
    t = qkv.float().requires_grad_(True)
    q, k, v = (t[:, :, i * C:(i + 1) * C] for i in range(3))
    oref = torch.einsum("bij,bjc->bic", torch.softmax(torch.einsum("bic,bjc->bij", q, k) * scale, dim=-1), v)
    oref.backward(do.float())
    qd, dod = qkv.cuda(), do.cuda()
    o, P = ops.attention_fwd(qd, scale, want_p=True)
    assert ops.attention_bwd_fused_ok(qd)
    d = ops.attention_bwd(qd, o, dod)
    err = lambda a, b: float((a.float().cpu() - b).abs().max() / b.abs().max())
    errs = [err(d[:, :, i * C:(i + 1) * C], t.grad[:, :, i * C:(i + 1) * C]) for i in range(3)]
    assert max(errs) < 1e-2, errs                  # the 16-bit bar at 1x (measured 2.4e-3 ... 6.3e-3)
    # the three-kernel path on the stored P
    dP = ops.bgemm_nt(dod, qd[:, :, 2 * C:], out_dtype=torch.float32)
    dS = ops.softmax_bwd(P, dP, scale)
    d3 = torch.empty_like(qd)
    ops.bgemm_nt(ops.transpose_last2(P), ops.transpose_last2(dod), out=d3[:, :, 2 * C:])
    ops.bgemm_nt(dS, ops.transpose_last2(qd[:, :, C:2 * C]), out=d3[:, :, :C])
    ops.bgemm_nt(ops.transpose_last2(dS), ops.transpose_last2(qd[:, :, :C]), out=d3[:, :, C:2 * C])
    errs3 = [err(d3[:, :, i * C:(i + 1) * C], t.grad[:, :, i * C:(i + 1) * C]) for i in range(3)]
    assert all(a <= b_ + 3e-3 for a, b_ in zip(errs, errs3)), (errs, errs3)
    assert torch.equal(d, ops.attention_bwd(qd, o, dod))
