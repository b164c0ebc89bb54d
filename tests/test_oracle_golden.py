"""Pins the oracle (CPU restatement) to the golden vectors captured from the reference itself
(oracle/make_golden.py) and to the known-answer anchors of SURVEY.md Appendix C."""
import numpy as np
import pytest
import torch

from oracle import diffusion_ref as D
from oracle import unet_ref as U

T = torch.from_numpy


def close(a, b, tol=1e-5):
    a = a.detach().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max abs err {err:.3e} (scale {scale:.3g})"


def test_param_inventory():
    assert U.count_params(128) == 6033665
    assert U.count_params(32) == 387137
    assert U.count_params(64) == 1521793
    assert len(U.param_spec(128)) == 160   # reference state_dict (strict load in make_golden.py)


def test_schedule_anchors(golden):
    g = golden("schedule.npz")
    # Appendix C anchors
    assert D.SCHED_B == 4.539992973129278e-05 and D.SCHED_A == 1.5707055269354342
    ls = D.logsnr_schedule_cosine(torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0]))
    np.testing.assert_array_equal(
        ls.numpy(), np.array([20.0, 1.7626187801361084, -0.0, -1.7626190185546875, -20.001096725463867], np.float32))
    np.testing.assert_array_equal(D.logsnr_schedule_cosine(T(g["u"])).numpy(), g["logsnr"])


@pytest.mark.parametrize("steps", [4, 8, 200, 250, 1000])
def test_sampler_index_arithmetic_bit_exact(golden, steps):
    """Row I1: u_t, u_s are fp32 bit patterns of (i+1)/T and i/T."""
    g = golden("schedule.npz")
    ut = np.array([D.sampler_times(i, steps)[0] for i in range(steps)], np.float32)
    us = np.array([D.sampler_times(i, steps)[1] for i in range(steps)], np.float32)
    assert ut.tobytes() == g[f"T{steps}_u_t"].tobytes()
    assert us.tobytes() == g[f"T{steps}_u_s"].tobytes()
    np.testing.assert_array_equal(D.logsnr_schedule_cosine(T(ut)).numpy(), g[f"T{steps}_logsnr_t"])
    np.testing.assert_array_equal(D.logsnr_schedule_cosine(T(us)).numpy(), g[f"T{steps}_logsnr_s"])


def test_embedding_and_algebra(golden):
    g = golden("schedule.npz")
    close(U.timestep_embedding(T(g["temb_t"]), 64, 256), g["temb_256"], 1e-6)
    close(U.timestep_embedding(T(g["temb_w"]), 64, 4), g["temb_4"], 1e-6)
    l, z, e = T(g["alg_logsnr"]), T(g["alg_z"]), T(g["alg_e"])
    close(D.predict_x_from_eps(z, e, l), g["alg_x_from_eps"], 1e-6)
    close(D.predict_eps_from_x(z, e, l), g["alg_eps_from_x"], 1e-6)
    close(D.predict_v_from_x_and_eps(z, e, l), g["alg_v"], 1e-6)
    close(D.predict_x_from_v(z, e, l), g["alg_x_from_v"], 1e-6)
    close(D.q_sample(z, l, torch.zeros_like(z)), g["alg_fw_mean"], 1e-6)
    # Appendix C: diffusion_forward(x=1, logsnr=0) mean = std = 0.70710677
    assert float(D.q_sample(torch.ones(1), torch.zeros(1), torch.zeros(1))) == pytest.approx(0.7071067690849304, abs=1e-7)


@pytest.mark.parametrize("name,C", [("unet_c32_s8.npz", 32), ("unet_c32_s12.npz", 32), ("unet_c32_s16.npz", 32),
                                    ("unet_c128_s28.npz", 128), ("unet_c64_s8.npz", 64), ("unet_c256_s8.npz", 256),
                                    ("unet_c96_s8.npz", 96), ("unet_c192_s8.npz", 192)])
def test_unet_forward(golden, name, C):
    g = golden(name)
    p = U.closed_form_params(C)
    z, l, y = T(g["z"]), T(g["logsnr"]), T(g["guide"])
    with torch.no_grad():
        close(U.unet_forward(p, z, l, guide=y), g["v"], 2e-5)
        close(U.unet_forward(p, z, l), g["v_noguide"], 2e-5)
        close(U.unet_forward(p, z, l, guide=y, cond_w=T(g["cond_w"])), g["v_condw"], 2e-5)


@pytest.mark.parametrize("name,C", [("train_c32_s8.npz", 32), ("train_c32_s16.npz", 32), ("train_c128_s28.npz", 128), ("train_c64_s8.npz", 64), ("train_c96_s8.npz", 96),
                                    ("train_c256_s8.npz", 256)])
def test_training_loss_and_grads(golden, name, C):
    g = golden(name)
    p = {k: v.clone().requires_grad_(True) for k, v in U.closed_form_params(C).items()}
    out = D.training_losses(p, T(g["x0"]), T(g["y"]), T(g["u"]), T(g["eps"]))
    close(out["logsnr"], g["logsnr"], 1e-6)
    close(out["z_t"], g["z_t"], 1e-6)
    close(out["loss"], g["loss_b"], 2e-5)
    out["loss"].mean().backward()
    names = [str(n) for n in g["grad_names"]]
    norms = torch.stack([p[n].grad.norm() if p[n].grad is not None else torch.tensor(0.0) for n in names])
    ref = T(g["grad_norms"])
    # relative per tensor, with a floor for mathematically-zero gradients (conv biases ahead of a GroupNorm;
    # at C=32, one channel per group, the whole embedding path)
    assert bool(((norms - ref).abs() <= 2e-3 * ref.abs() + 1e-5 * ref.abs().max()).all())
    for k in g.files:
        if k.startswith("grad__"):
            close(p[k[6:]].grad, g[k], 2e-4)
        if k.startswith("gradslice__"):
            close(p[k[11:]].grad[:4, :6], g[k], 2e-4)
    # cond_w_embed is not on the non-distillation path: no gradient
    assert p["cond_w_embed.0.weight"].grad is None


@pytest.mark.parametrize("name,C", [("train_c32_s8.npz", 32), ("train_c64_s8.npz", 64)])
def test_adam_two_steps(golden, name, C):
    g = golden(name)
    p = {k: v.clone() for k, v in U.closed_form_params(C).items()}
    p0 = {k: v.clone() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    names = [str(n) for n in g["grad_names"]]
    for step in (1, 2):
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        out = D.training_losses(q, T(g["x0"]), T(g["y"]), T(g["u"]), T(g["eps"]))
        loss = out["loss"].mean()
        close(loss, g[f"loss_step{step}"], 2e-5)
        loss.backward()
        for k in names:
            if q[k].grad is None:
                continue
            p[k], m[k], v2[k] = D.adam_step(p[k], q[k].grad, m[k], v2[k], step)
        dn = torch.stack([(p[k] - p0[k]).norm() for k in names])
        ref = T(g[f"delta_norms_step{step}"])
        # Adam turns a mathematically-zero gradient (rounding noise ~1e-9, e.g. a conv bias ahead of a GroupNorm)
        # into +-lr steps whose sign is noise: compare only tensors with a real gradient.
        gn = T(g["grad_norms"])
        live = gn > 1e-4 * gn.max()
        assert int(live.sum()) > 40
        assert bool((((dn - ref).abs() <= 5e-3 * ref.abs()) | ~live).all())
        close(p["down.seq.0.conv.weight"] - p0["down.seq.0.conv.weight"], g[f"delta_stem_step{step}"], 1e-3)


@pytest.mark.parametrize("name,C,steps", [("sample_c32_s8_T4.npz", 32, 4), ("sample_c32_s12_T8.npz", 32, 8)])
def test_samplers(golden, name, C, steps):
    g = golden(name)
    p = U.closed_form_params(C)
    init, y = T(g["init"]), T(g["y"])
    with torch.no_grad():
        zs, xs, es = D.sample(p, init, y, steps, "ddim")
        close(zs, g["ddim_zs"], 5e-5); close(xs, g["ddim_xs"], 5e-5); close(es, g["ddim_eps"], 5e-5)
        zs, xs, _ = D.sample(p, init, y, steps, "ddim", cond_w=T(g["cfg_w"]))
        close(zs, g["cfg_zs"], 1e-4); close(xs, g["cfg_xs"], 1e-4)
        zs, xs, _ = D.sample(p, init, y, steps, "noisy", noises=T(g["anc_noise"]))
        close(zs, g["anc_zs"], 1e-4); close(xs, g["anc_xs"], 1e-4)
        # final-step select (gaussian_diffusion.py:292): last z equals last x_pred
        assert torch.equal(zs[-1], xs[-1])


@pytest.mark.parametrize("name,C", [("distill_c64_s8.npz", 64), ("distill_c128_s8.npz", 128)])
@pytest.mark.parametrize("mode", ["step1", "step2"])
def test_distillation_losses(golden, name, C, mode):
    """Teacher branches of training_losses (gaussian_diffusion.py:87-91,105-154), incl. the integer time path (I3)."""
    g = golden(name)
    teacher = U.closed_form_params(C)
    student = {k: (0.9 * v).clone().requires_grad_(True) for k, v in teacher.items()}
    kw = dict(u=T(g[f"{mode}_u"])) if mode == "step1" else dict(i_times=T(g[f"{mode}_i"]))
    out = D.distill_losses(student, teacher, T(g["x0"]), T(g["y"]), T(g[f"{mode}_eps"]), T(g[f"{mode}_cond_w"]), 8, mode, **kw)
    close(out["loss"], g[f"{mode}_loss_b"], 5e-5)
    out["loss"].mean().backward()
    names = [str(n) for n in g["grad_names"]]
    norms = torch.stack([student[n].grad.norm() if student[n].grad is not None else torch.tensor(0.0) for n in names])
    ref = T(g[f"{mode}_grad_norms"])
    assert bool(((norms - ref).abs() <= 2e-3 * ref.abs() + 1e-5 * ref.abs().max()).all())
    close(student["cond_w_embed.2.weight"].grad, g[f"{mode}_grad_cond_w_embed"], 5e-4)


@pytest.mark.parametrize("mt", ["eps", "x"])
def test_mean_types(golden, mt):
    """`mean_type` 'eps' / 'x' (gaussian_diffusion.py:58-63): loss, gradient norms, DDIM with and without guidance."""
    g = golden("meantype_c128_s8.npz")
    p = {k: v.clone().requires_grad_(True) for k, v in U.closed_form_params(128).items()}
    out = D.training_losses(p, T(g["x0"]), T(g["y"]), T(g["u"]), T(g["eps"]), mean_type=mt)
    close(out["loss"], g[f"{mt}_loss_b"], 2e-5)
    out["loss"].mean().backward()
    names = [str(n) for n in g["grad_names"]]
    norms = torch.stack([p[n].grad.norm() if p[n].grad is not None else torch.tensor(0.0) for n in names])
    ref = T(g[f"{mt}_grad_norms"])
    assert bool(((norms - ref).abs() <= 2e-3 * ref.abs() + 1e-5 * ref.abs().max()).all())
    close(p["out.2.weight"].grad, g[f"{mt}_grad_out2"], 2e-4)
    q = U.closed_form_params(128)
    with torch.no_grad():
        zs, xs, _ = D.sample(q, T(g["init"]), T(g["y"]), 4, "ddim", mean_type=mt)
        close(zs, g[f"{mt}_ddim_zs"], 5e-5); close(xs, g[f"{mt}_ddim_xs"], 5e-5)
        zs, _, _ = D.sample(q, T(g["init"]), T(g["y"]), 4, "ddim", cond_w=T(g[f"{mt}_cfg_w"]), mean_type=mt)
        close(zs, g[f"{mt}_cfg_zs"], 1e-4)


def test_attention_block_definition_matches_sdpa():
    """The self-attention extension has no reference counterpart: pin the oracle's definition against an independent formulation
    (torch's scaled_dot_product_attention, single head over all channels) and check that a zero-initialised projection makes the
    block the identity."""
    import torch.nn.functional as F
    C, B, H, W = 128, 2, 8, 8
    p = U.closed_form_params(C, attention=True)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((B, C, H, W), generator=g)
    out = U.attention_block(p, x)
    a = U.gn_silu(x, p["attn.norm.weight"], p["attn.norm.bias"])
    qkv = F.conv2d(a, p["attn.qkv.weight"], p["attn.qkv.bias"])
    q, k, v = (t.reshape(B, 1, C, H * W).transpose(2, 3) for t in qkv.chunk(3, dim=1))        # [B, heads=1, N, C]
    o = F.scaled_dot_product_attention(q, k, v).transpose(2, 3).reshape(B, C, H, W)
    ref = x + F.conv2d(o, p["attn.proj.weight"], p["attn.proj.bias"])
    close(out, ref, 2e-5)
    p0 = dict(p); p0["attn.proj.weight"] = torch.zeros_like(p["attn.proj.weight"]); p0["attn.proj.bias"] = torch.zeros_like(p["attn.proj.bias"])
    assert torch.equal(U.attention_block(p0, x), x)
    # the inventory with the extension: 6 more tensors behind `turn`, none without it
    names = [n for n, _ in U.param_spec(C, attention=True)]
    assert len(names) == 166 and names.index("attn.norm.weight") == names.index("turn.out_layers.3.bias") + 1
    assert len(U.param_spec(C)) == 160


@pytest.mark.parametrize("N", [64, 256])
def test_attention_core_is_pinned_to_the_references_own_attention(golden, N):
    """Round 6: the contraction core of the attention extension (qkv -> softmax(q k^T / sqrt(C)) v -> proj) against the ONE attention
    implementation the reference holds - `CausalSelfAttention` (gms/autoregs/pixel_transformer.py:74-122), n_head = 1, n_embed = 128, mask of
    ones; fixtures generated from that class by oracle/make_golden.py gen_attn_core.  `oracle.unet_ref.attention_core` must reproduce its output
    and, through autograd, dx and the gradients of all four Linear layers.  What stays the extension's own definition is the placement
    (GroupNorm + SiLU in front, the residual, the level), which `attention_block` adds around this core."""
    gd = golden(f"attn_core_{N}.npz")
    C, B = int(gd["C"]), int(gd["B"])
    S = int(round(N ** 0.5))
    x, dy, lin = U.attn_core_case(N, C, B)
    p = {"attn.qkv.weight": torch.cat([lin[k][0] for k in ("query", "key", "value")]).reshape(3 * C, C, 1, 1).clone().requires_grad_(True),
         "attn.qkv.bias": torch.cat([lin[k][1] for k in ("query", "key", "value")]).clone().requires_grad_(True),
         "attn.proj.weight": lin["proj"][0].reshape(C, C, 1, 1).clone().requires_grad_(True),
         "attn.proj.bias": lin["proj"][1].clone().requires_grad_(True)}
    full = U.closed_form_params(C, attention=True)
    a = x.transpose(1, 2).reshape(B, C, S, S).clone().requires_grad_(True)      # tokens [B, N, C] -> the NCHW map the block sees
    y = U.attention_core(p, a)
    close(y.reshape(B, C, N).transpose(1, 2), T(gd["y"]), 2e-5)
    y.backward(dy.transpose(1, 2).reshape(B, C, S, S))
    close(a.grad.reshape(B, C, N).transpose(1, 2), T(gd["dx"]), 2e-5)
    gw = p["attn.qkv.weight"].grad.reshape(3, C, C)
    gb = p["attn.qkv.bias"].grad.reshape(3, C)
    for i, k in enumerate(("query", "key", "value")):
        close(gw[i], T(gd[f"d{k}_w"]), 2e-5)
        close(gb[i], T(gd[f"d{k}_b"]), 2e-5)          # (the key bias shifts every logit of a row alike: its gradient is zero up to rounding, ~ 1e-8 here)
    assert float(T(gd["dkey_b"]).abs().max()) < 1e-6 * float(T(gd["dquery_b"]).abs().max())
    close(p["attn.proj.weight"].grad.reshape(C, C), T(gd["dproj_w"]), 2e-5)
    close(p["attn.proj.bias"].grad, T(gd["dproj_b"]), 2e-5)
    assert float(T(gd["dkey_w"]).abs().max()) > 0 and float(T(gd["y"]).std()) > 0.05
    # the block = the extension's placement around that core
    xb = torch.randn((B, C, S, S), generator=torch.Generator().manual_seed(N))
    pd = {k: v.detach() for k, v in p.items()}
    pd.update({"attn.norm.weight": full["attn.norm.weight"], "attn.norm.bias": full["attn.norm.bias"]})
    assert torch.equal(U.attention_block(pd, xb), xb + U.attention_core(pd, U.gn_silu(xb, pd["attn.norm.weight"], pd["attn.norm.bias"])))


@pytest.mark.parametrize("name,C", [("definit_c128_s28.npz", 128), ("definit_c128_s32.npz", 128), ("definit_c256_s16.npz", 256)])
def test_default_init_goldens_pin_the_oracle(golden, name, C):
    """The default-init-scale set (oracle/make_golden.py:gen_default_init) the bf16 bar is held on: forward with / without
    labels, per-sample loss and every gradient norm of the reference, reproduced by the oracle."""
    g = golden(name)
    p = U.reference_init_params(C, 1, seed=int(g["init_seed"]), zero_out_layers=False)
    z, l, y = T(g["z"]), T(g["logsnr"]), T(g["guide"])
    with torch.no_grad():
        close(U.unet_forward(p, z, l, guide=y), g["v"])
        close(U.unet_forward(p, z, l), g["v_noguide"])
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_b = D.training_losses(pr, T(g["x0"]), y, T(g["u"]), T(g["eps"]))["loss"]
    close(loss_b, g["loss_b"], 2e-5)
    loss_b.mean().backward()
    names = [str(n) for n in g["grad_names"]]
    norms = torch.stack([pr[n].grad.norm() if pr[n].grad is not None else torch.tensor(0.0) for n in names])
    ref = T(g["grad_norms"])
    assert float(((norms - ref).abs() / (ref.abs() + 1e-6 * ref.abs().max())).max()) < 2e-3


SIZED = [("sized_c128_1x64.npz", 1, 64), ("sized_c128_3x32.npz", 3, 32), ("sized_c128_3x64.npz", 3, 64)]


@pytest.mark.parametrize("name,cin,S", SIZED)
def test_sized_goldens_pin_the_oracle(golden, name, cin, S):
    """Round 4: the oracle above 32x32 and at 3 image channels against the REFERENCE (oracle/make_golden.py:gen_sized - the unmodified
    reference net at 1x64x64; its stem / head re-assigned to 3 channels, SURVEY §8c O4, at 3x32x32 and 3x64x64): forward with / without
    labels, per-sample loss, every gradient norm, stored gradients and gradient slices, DDIM chains without / with guidance."""
    g = golden(name)
    assert int(g["in_channels"]) == cin and g["z"].shape[1:] == (cin, S, S)
    p = U.reference_init_params(128, cin, seed=int(g["init_seed"]), zero_out_layers=False)
    z, l, y = T(g["z"]), T(g["logsnr"]), T(g["guide"])
    with torch.no_grad():
        close(U.unet_forward(p, z, l, guide=y), g["v"])
        close(U.unet_forward(p, z, l), g["v_noguide"])
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_b = D.training_losses(pr, T(g["x0"]), y, T(g["u"]), T(g["eps"]))["loss"]
    close(loss_b, g["loss_b"], 2e-5)
    loss_b.mean().backward()
    names = [str(n) for n in g["grad_names"]]
    norms = torch.stack([pr[n].grad.norm() if pr[n].grad is not None else torch.tensor(0.0) for n in names])
    ref = T(g["grad_norms"])
    assert float(((norms - ref).abs() / (ref.abs() + 1e-6 * ref.abs().max())).max()) < 2e-3
    for k in g.files:
        if k.startswith("grad__"):
            close(pr[k[6:]].grad, g[k], 2e-4)
        elif k.startswith("gradslice__"):
            close(pr[k[11:]].grad[:4, :6], g[k], 2e-4)
    if "chain_T" in g.files:
        steps, init, yc = int(g["chain_T"]), T(g["chain_init"]), T(g["chain_y"])
        with torch.no_grad():
            zs, xs, _ = D.sample(p, init, yc, steps, "ddim")
            close(zs, g["ddim_zs"], 1e-4); close(xs, g["ddim_xs"], 1e-4)
            zs, _, _ = D.sample(p, init, yc, steps, "ddim", cond_w=T(g["cfg_w"]))
            close(zs, g["cfg_zs"], 1e-4)
