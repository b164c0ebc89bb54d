"""Full-size (BASELINE.json configs[1]: B=1024, 1x28x28, C=128, bf16) checks through size-independent properties —
the oracle cannot run at this size in seconds, so the HIP path is checked against itself and against invariants:
determinism, batch-splitting equivalence (what data parallelism relies on), linearity of the convolution in its input,
GroupNorm's normalisation invariant, and the sampler's final-step select."""
from functools import partial

import pytest
import torch

pytestmark = pytest.mark.gpu
B, S, C = 1024, 28, 128


@pytest.fixture(scope="module")
def net():
    from generative_models_amd.diffusion.simple_unet import SimpleUnet
    torch.manual_seed(0)
    n = SimpleUnet(C, 0.0, compute_dtype=torch.bfloat16)
    with torch.no_grad():
        for name, p in n.named_parameters():
            if ".out_layers.3.weight" in name:
                p.uniform_(-0.02, 0.02)
    return n.cuda()


def data(seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand((B, 1, S, S), generator=g) * 2 - 1).cuda()
    y = torch.randint(0, 10, (B,), generator=g).cuda()
    u = torch.rand((B,), generator=g).cuda()
    eps = torch.randn((B, 1, S, S), generator=g).cuda()
    return x, y, u, eps


def test_train_step_deterministic_and_batch_split(net):
    from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    d = GaussianDiffusion(mean_type="v", num_steps=1000)
    x, y, u, eps = data()
    out1 = d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps)
    g1 = net.flat_grads.clone(); l1 = out1["loss"].clone()
    out2 = d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps)
    assert torch.equal(out2["loss"], l1) and torch.equal(net.flat_grads, g1)          # bit-for-bit reproducible
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    # gradient of the full batch == sum of the gradients of its two halves (same 1/B scale)
    acc = torch.zeros_like(g1)
    losses = []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        o = d.train_forward_backward(net=partial(net, guide=y[sl]), x=x[sl], grad_scale=1.0 / B, u=u[sl], eps=eps[sl])
        acc += net.flat_grads; losses.append(o["loss"])
    assert torch.equal(torch.cat(losses), l1)                                          # per-sample work is independent
    rel = float((acc - g1).abs().max() / g1.abs().max())
    assert rel < 2e-3, rel                                                             # fp32 summation order only


def test_conv_linearity_and_gn_invariant():
    from generative_models_amd import ops
    T = torch.float32                     # exact-fp32 MFMA path: linearity holds to rounding
    a = torch.randn((B, S, S, C), device="cuda"); b = torch.randn((B, S, S, C), device="cuda")
    w = torch.randn((C, C, 3, 3), device="cuda") / (C * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=T); ops.pack_conv_weight(w, wf, None)
    f = lambda t: ops.conv_igemm([t], wf, C, 3, ops.NORMAL, (S, S))
    lhs = f(a + 2 * b); rhs = f(a) + 2 * f(b)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) < 1e-5
    # halo kernel (bf16) vs exact-fp32 kernel on bf16-representable inputs
    ab = a.bfloat16(); wb = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16)
    wq = w.bfloat16().float(); ops.pack_conv_weight(wq, wb, None)
    wq32 = torch.empty(w.numel(), device="cuda", dtype=T); ops.pack_conv_weight(wq, wq32, None)
    y16 = ops.conv_igemm([ab], wb, C, 3, ops.NORMAL, (S, S)).float()
    y32 = ops.conv_igemm([ab.float()], wq32, C, 3, ops.NORMAL, (S, S))
    assert float((y16 - y32).abs().max() / y32.abs().max()) < 1e-2
    # GroupNorm with gamma=1, beta=0: SiLU^-1 not needed — check the saved statistics against torch on the device
    x = (a * 1.7 + 0.4).bfloat16()
    _, mean, rstd = ops.gn_silu_fwd(x, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), 32)
    xr = x.float().reshape(B, S * S, 32, 4).permute(0, 2, 1, 3).reshape(B, 32, -1)
    assert float((mean - xr.mean(-1)).abs().max()) < 1e-4
    assert float((rstd - (xr.var(-1, unbiased=False) + 1e-5).rsqrt()).abs().max()) < 1e-3


def test_sampler_full_batch(net):
    from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    d = GaussianDiffusion(mean_type="v", num_steps=3, sampler="ddim")
    _, y, _, eps = data(1)
    zs, xs, es = d.sample(net=partial(net, guide=y), init_x=eps)
    assert zs.shape == (3, B, 1, S, S) and torch.equal(zs[-1], xs[-1])
    assert float(xs.abs().max()) <= 1.0 and bool(torch.isfinite(zs).all())
    z2 = d.sample(net=partial(net, guide=y), init_x=eps, record=False)[0][-1]
    assert torch.equal(z2, zs[-1])


def test_forward_is_bit_reproducible_with_the_skip_convs_on_the_side_stream(net):
    """Side-stream determinism: repeated identical forwards with GMK_FWD_SIDE (1x1 skip convolutions beside the
    GroupNorm / conv1 chain) are bit-identical, and equal to the forward without the overlap.  A -DGMK_SHFL_BPERMUTE build fails
    this about every second forward; the shipped DPP / permlane reductions have never failed it."""
    from generative_models_amd import ops
    x, y, u, _ = data(3)
    l = u * 20 - 10
    keep, keep_emb = ops.FWD_SIDE, ops.EMB_SIDE
    try:
        ops.FWD_SIDE = True
        outs = [net.forward_hip(x, l, y, None).clone() for _ in range(8)]
        ops.FWD_SIDE = False
        plain = net.forward_hip(x, l, y, None)
        # the embedding path on the side stream (GMK_EMB_SIDE, default on since round 5): same kernels, joined before the first consumer - same bits
        ops.EMB_SIDE = True
        emb_on = [net.forward_hip(x, l, y, None).clone() for _ in range(4)]
        ops.EMB_SIDE = False
        emb_off = net.forward_hip(x, l, y, None)
    finally:
        ops.FWD_SIDE, ops.EMB_SIDE = keep, keep_emb
    assert all(torch.equal(outs[0], o) for o in outs[1:]) and torch.equal(outs[0], plain)
    assert all(torch.equal(emb_off, o) for o in emb_on) and torch.equal(emb_off, plain)


def test_bias_gather_is_stable_under_stream_overlap(net):
    """Round 6: the 3x3 halo kernels take a tile's bias as the first MFMA's C operand and gather it from two registers per wave with ds_bpermute_b32
    (conv_halo.hip load_bias) - the one ds_bpermute user in csrc/ (rounds 1-2 saw a miscompare, never explained below the ISA, in a GroupNorm build
    whose REDUCTIONS used it, with a second stream active).  With live biases and every overlap on (weight gradients and 1x1 skip convolutions on
    the side stream), 12 identical train passes and 12 forwards at the full batch give the same bits every time (tools/bias_gather_stress.py
    runs 40 + 40 per shape)."""
    from generative_models_amd import ops
    from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    d = GaussianDiffusion(mean_type="v", num_steps=1000)
    x, y, u, eps = data(11)
    saved = net.flat_params.clone()
    keep = ops.FWD_SIDE
    try:
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith(".bias"):
                    p.uniform_(-0.2, 0.2)
        net.mark_params_changed()
        ops.FWD_SIDE = True
        ref = None
        for _ in range(12):
            out = d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps)
            cur = (out["loss"].clone(), net.flat_grads.clone(), net.forward_hip(x, u * 20 - 10, y, None).clone())
            if ref is None:
                ref = cur
            assert all(torch.equal(a, b) for a, b in zip(cur, ref))
        assert bool(torch.isfinite(ref[1]).all()) and float(ref[1].abs().max()) > 0
    finally:
        ops.FWD_SIDE = keep
        with torch.no_grad():
            net.flat_params.copy_(saved)
        net.mark_params_changed()


def test_backward_is_bit_reproducible_with_the_weight_gradients_on_the_side_stream(net):
    """The backward analogue of the test above, for the overlap that is ON by default (ops.WGRAD_STREAM: every weight gradient beside the
    data-gradient chain, the same co-residency pattern): eight identical train passes at the full batch give the same gradient arena bit for
    bit, equal to the arena of a pass with every kernel on ONE stream; and with the chip partitioned between the two streams
    (ops.WGRAD_CUS: other split counts, i.e. another fp32 summation order of the weight-gradient slabs) the same values to rounding."""
    from generative_models_amd import ops
    from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    d = GaussianDiffusion(mean_type="v", num_steps=1000)
    x, y, u, eps = data(7)
    run = lambda: (d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps)["loss"].clone(), net.flat_grads.clone())
    keep, keep_cus = ops.WGRAD_STREAM, ops.WGRAD_CUS
    try:
        ops.WGRAD_STREAM, ops.WGRAD_CUS = True, 0
        runs = [run() for _ in range(8)]
        ops.WGRAD_STREAM = False
        loss1, grads1 = run()
        ops.WGRAD_STREAM, ops.WGRAD_CUS = True, 96
        loss_p, grads_p = run()
        assert ops.get_cu_limit() == 256                         # the partition is undone behind the pass
    finally:
        ops.WGRAD_STREAM, ops.WGRAD_CUS = keep, keep_cus
    assert all(torch.equal(runs[0][0], l) and torch.equal(runs[0][1], g) for l, g in runs[1:])
    assert torch.equal(runs[0][0], loss1) and torch.equal(runs[0][1], grads1)
    assert torch.equal(loss_p, loss1)
    rel = float((grads_p - grads1).abs().max() / grads1.abs().max())
    assert rel < 1e-4, rel


def test_fused_groupnorm_forward_equals_materialised_forward(net):
    """Inference forward with GroupNorm-apply + SiLU inside the convolutions (ops.GN_FUSE, the default) against the forward that
    materialises the normalised tensors (the training path's forward): the same bits, at the full batch."""
    from generative_models_amd import ops
    x, y, u, _ = data(5)
    l = u * 24 - 12
    keep, keep_min = ops.GN_FUSE, ops.GN_FUSE_MIN_HW
    try:
        ops.GN_FUSE, ops.GN_FUSE_MIN_HW = True, 0          # fuse wherever the kernel can (the 28 x 28 level here), not only where it pays
        fused = net.forward_hip(x, l, y, None)
        ops.GN_FUSE = False
        plain = net.forward_hip(x, l, y, None)
        ctx = {}
        train_fwd = net.forward_hip(x, l, y, None, ctx=ctx)
    finally:
        ops.GN_FUSE, ops.GN_FUSE_MIN_HW = keep, keep_min
    assert torch.equal(fused, plain) and torch.equal(fused, train_fwd)


def test_subpixel_forms_are_each_others_adjoints_at_the_headline_size():
    """BASELINE configs[2]'s largest resampling level (B = 2048, 16x16 <-> 32x32, C = 128), where no CPU reference runs in seconds: the three
    sub-pixel forms of `Upsample` (reference simple_unet.py:112-122) and the stride-2 pair (:75-84) are checked through the identities that tie
    them together - the data gradient is the adjoint of the forward, <conv(x), dy> = <x, dgrad(dy)>; the weight gradient is the derivative in
    w, <dW, w> = <conv(x; w), dy>; the transposed form is the adjoint of the stride-2 forward - inner products accumulated in fp64 over 268 M
    products each, 16-bit storage of every result."""
    import math
    from generative_models_amd import ops
    Bh, Sl = 2048, 16
    g = torch.Generator().manual_seed(5)
    x = torch.randn((Bh, Sl, Sl, C), generator=g).cuda().bfloat16()
    dy = torch.randn((Bh, 2 * Sl, 2 * Sl, C), generator=g).cuda().bfloat16()
    w = (torch.randn((C, C, 3, 3), generator=g) / math.sqrt(9 * C)).cuda()
    wsub, wsub_d = torch.empty(16 * C * C, device="cuda", dtype=torch.bfloat16), torch.empty(16 * C * C, device="cuda", dtype=torch.bfloat16)
    ops.pack_upsample_weight(w, wsub, wsub_d)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    assert ops.conv_subpixel_ok(Bh, Sl, Sl, C, torch.bfloat16) and ops.conv_wgrad_subpixel_ok(Bh, Sl, Sl, C, torch.bfloat16)
    y = ops.conv_subpixel(x, wsub, C, ops.SUBPIXEL_UPSAMPLE)
    dx = ops.conv_subpixel(dy, wsub_d, C, ops.SUBPIXEL_UPSAMPLE_DGRAD)
    dw = torch.empty((C, C, 3, 3), device="cuda")
    ops.conv_wgrad_subpixel(dy, x, dw)
    a, b_, c = dot(y, dy), dot(x, dx), dot(dw, w)
    # <y, dy> of independent tensors is a random-walk sum: its size, and the size of what the 16-bit roundings of y / dx / the weight packs (2^-9
    # relative each, independent) add to it, is |y| |dy| / sqrt(N) - expected difference 1 - 2e-3 of that, bar 1e-2
    unit = math.sqrt(float(y.double().pow(2).sum()) * float(dy.double().pow(2).sum()) / y.numel())
    assert abs(a - b_) < 1e-2 * unit, (a, b_, unit)
    assert abs(a - c) < 1e-2 * unit, (a, c, unit)
    # linearity of the forward in its input (fp32 accumulation, one rounding): conv(x1 + x2) = conv(x1) + conv(x2) up to the 16-bit roundings
    x2 = torch.randn((Bh, Sl, Sl, C), generator=g).cuda().bfloat16()
    xs = (x.float() + x2.float()).bfloat16()
    lhs = ops.conv_subpixel(xs, wsub, C, ops.SUBPIXEL_UPSAMPLE).float()
    rhs = y.float() + ops.conv_subpixel(x2, wsub, C, ops.SUBPIXEL_UPSAMPLE).float() + \
        ops.conv_subpixel((xs.float() - x.float() - x2.float()).bfloat16(), wsub, C, ops.SUBPIXEL_UPSAMPLE).float()
    assert float((lhs - rhs).abs().max() / lhs.abs().max()) < 2e-2
    # stride-2 pair: <conv_s2(xh), dyl> = <xh, transposed(dyl)>
    xh = dy                                                          # a high-resolution tensor
    dyl = x                                                          # a low-resolution one
    wf, wd = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16), torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16)
    ops.pack_conv_weight(w, wf, wd)
    ys2 = ops.conv_igemm([xh], wf, C, 3, ops.STRIDE2, (Sl, Sl))
    dxh = ops.conv_subpixel(dyl, wd, C, ops.SUBPIXEL_TRANSPOSED)
    a2, b2 = dot(ys2, dyl), dot(xh, dxh)
    unit2 = math.sqrt(float(ys2.double().pow(2).sum()) * float(dyl.double().pow(2).sum()) / ys2.numel())
    assert abs(a2 - b2) < 1e-2 * unit2, (a2, b2, unit2)


@pytest.mark.parametrize("kind,cond_w", [("ddim", None), ("ddim", 0.5), ("noisy", None)])
def test_sampler_on_two_streams_is_bit_identical(net, kind, cond_w):
    """Round 5: batches of at least 2 x 256 Ki pixels sample as two half-batches on two HIP streams (the start-up and tail of one half's persistent
    kernels overlap with the other half's kernels: -2.4 ... -3.0 % per forward).  The chains of different samples never meet, so the images -
    every intermediate z, x-hat and eps-hat of the recorded trajectory - must be the SAME BITS as on one stream: DDIM, guided DDIM (2B-image
    forwards per half) and the ancestral sampler, whose per-step noise is one Philox draw over the whole batch that the halves slice by counter."""
    from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    _, y, _, eps = data(2)
    outs = []
    for streams in (1, 2):
        d = GaussianDiffusion(mean_type="v", num_steps=4, sampler=kind, seed=11)
        d.SAMPLER_STREAMS = streams
        assert (B // 2) * S * S >= d.STREAM_MIN_PIXELS
        outs.append(d.sample(net=partial(net, guide=y), init_x=eps, cond_w=cond_w, net_cond_w=None if cond_w is None else torch.full((B,), cond_w).cuda()))
        last = d.sample(net=partial(net, guide=y), init_x=eps, cond_w=cond_w, net_cond_w=None if cond_w is None else torch.full((B,), cond_w).cuda(),
                        record=False)[0][-1]
        if kind != "noisy":                      # (the ancestral sampler's second call draws fresh noise)
            assert torch.equal(last, outs[-1][0][-1])
    for a, b in zip(*outs):
        assert a.shape == b.shape and torch.equal(a, b)
    assert bool(torch.isfinite(outs[0][0]).all())
    # ... and with the embedding path on the main stream instead of the side stream (GMK_EMB_SIDE=0) the two-stream images are the same bits again
    from generative_models_amd import ops
    keep = ops.EMB_SIDE
    try:
        ops.EMB_SIDE = not keep
        d = GaussianDiffusion(mean_type="v", num_steps=4, sampler=kind, seed=11)
        other = d.sample(net=partial(net, guide=y), init_x=eps, cond_w=cond_w, net_cond_w=None if cond_w is None else torch.full((B,), cond_w).cuda())
    finally:
        ops.EMB_SIDE = keep
    for a, b in zip(outs[1], other):
        assert torch.equal(a, b)


def test_two_stream_sampler_sees_a_weight_update_made_right_before_it(net):
    """The first forward after a weight update re-packs the convolution weights.  In the two-stream sampler that forward runs on chunk stream 0 and
    clears the host-side flag; chunk stream 1 only waits for the stream that was current at the fork, so the re-pack has to be enqueued THERE, before
    the fork (round 6; before, chunk 1's convolutions could read pack buffers whose re-pack kernel they had never waited for - hidden by host
    issue latency).  A K = 2 sample right after a perturbation + mark_params_changed() must equal the one-stream sample of the same weights, and
    differ from the sample of the old weights."""
    from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    _, y, _, eps = data(5)
    def run(streams):
        d = GaussianDiffusion(mean_type="v", num_steps=2, sampler="ddim", seed=3)
        d.SAMPLER_STREAMS = streams
        return d.sample(net=partial(net, guide=y), init_x=eps, record=False)[0][-1]
    before = run(1).clone()
    name = "up.seq.5.out_layers.3.weight"
    saved = net._pv[name].clone()
    try:
        for _ in range(3):                    # several updates in a row: every one has to be seen by both halves
            with torch.no_grad():
                net._pv[name].mul_(1.25)
            net.mark_params_changed()
            two = run(2).clone()
            one = run(1)
            assert torch.equal(two, one)
            assert not torch.equal(two, before)
            before = two
    finally:
        with torch.no_grad():
            net._pv[name].copy_(saved)
        net.mark_params_changed()


def test_embedding_rows_do_not_depend_on_the_batch_size(net):
    """What the two-stream sampler's bit-identity rests on besides per-sample GroupNorm: a sample's embedding (time / guide MLPs -> the 12 emb_layers, fp32
    GEMMs) is the same bits whatever batch it is computed in - the K-split of the short contractions is decided independently of M (round 5; before,
    a batch of 2048 and one of 1024 differed in the last bit)."""
    g = torch.Generator().manual_seed(3)
    for nb in (24, 100, 256, 512, 1024, 2048, 4096):
        ll = (torch.randn((nb,), generator=g) * 3).cuda()
        yy = torch.randint(-1, 10, (nb,), generator=g).cuda()
        full = net._embed_fwd(ll, yy, None, None)
        for a, b in ((0, nb // 2), (nb // 2, nb), (nb // 4, min(nb, nb // 4 + 100))):
            part = net._embed_fwd(ll[a:b].clone(), yy[a:b].clone(), None, None)
            assert torch.equal(full[a:b], part), (nb, a, b)
