"""Golden-vector generator (test infrastructure; run in the BUILD container only).

Imports the reference's three hot-path modules from /root/reference (never copied into this
repo), fills the reference `SimpleUnet` with `oracle.unet_ref.closed_form_params`, and writes
small input/output fixtures to tests/golden/*.npz.  The fixtures are data only.

    PYTHONPATH=/root/reference python -m oracle.make_golden

RNG-consuming reference calls are driven by seeding torch's CPU generator and replaying the
same draw order here to capture the drawn values into the fixture (tests never rely on torch's
RNG stream being reproducible across versions).
"""
import os
import sys
from functools import partial

import numpy as np
import torch

REF = os.environ.get("GMS_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from gms.diffusion import diffusion_utils as R_du        # noqa: E402
from gms.diffusion import gaussian_diffusion as R_gd     # noqa: E402
from gms.diffusion import simple_unet as R_su            # noqa: E402

from oracle import unet_ref                              # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def ref_net(C):
    net = R_su.SimpleUnet(C, 0.0)
    sd = unet_ref.closed_form_params(C)
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    # the oracle's inventory must match the reference's state-dict order exactly
    assert list(net.state_dict().keys()) == list(sd.keys())
    return net.eval()


def inputs(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    x0 = (torch.rand((B, 1, S, S), generator=g) * 2 - 1)
    x0[:, :, : S // 4] = -1.0            # MNIST-like saturated background rows (exercises the clip)
    y = torch.randint(0, 10, (B,), generator=g)
    y[0] = -1                            # one unconditional row (simple_unet.py:54-57)
    return x0, y


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


def gen_schedule():
    sched = R_du.get_logsnr_schedule("cosine", logsnr_min=-20.0, logsnr_max=20.0)
    u = torch.tensor([0.0, 1e-3, 0.1, 0.25, 0.5, 0.75, 0.9, 0.999, 1.0], dtype=torch.float32)
    out = {"u": u, "logsnr": sched(u)}
    # sampler index arithmetic (gaussian_diffusion.py:288-290), every i for T in {4, 8, 200, 250, 1000}
    for T in (4, 8, 200, 250, 1000):
        ut, us, lt, ls = [], [], [], []
        for i in range(T):
            ti = torch.tensor(i)
            a = (ti + 1.0) / T
            b = ti / T
            ut.append(a); us.append(b); lt.append(sched(a)); ls.append(sched(b))
        out[f"T{T}_u_t"] = torch.stack(ut); out[f"T{T}_u_s"] = torch.stack(us)
        out[f"T{T}_logsnr_t"] = torch.stack(lt); out[f"T{T}_logsnr_s"] = torch.stack(ls)
    t = torch.tensor([20.0, 0.0, -20.0, 1.7626188, -9.682026], dtype=torch.float32)
    out["temb_t"] = t
    out["temb_256"] = R_su.timestep_embedding(timesteps=t, dim=64, max_period=256)
    w = torch.tensor([0.0, 0.5, 1.0, 3.999], dtype=torch.float32)
    out["temb_w"] = w
    out["temb_4"] = R_su.timestep_embedding(timesteps=w, dim=64, max_period=4)
    l = torch.tensor([-20.0, -3.0, 0.0, 2.5, 20.0])
    z = torch.tensor([0.3, -1.2, 0.7, 0.1, -0.5]); e = torch.tensor([1.0, -0.4, 0.2, 2.0, -1.5])
    out["alg_logsnr"], out["alg_z"], out["alg_e"] = l, z, e
    out["alg_x_from_eps"] = R_du.predict_x_from_eps(z=z, eps=e, logsnr=l)
    out["alg_eps_from_x"] = R_du.predict_eps_from_x(z=z, x=e, logsnr=l)
    out["alg_v"] = R_du.predict_v_from_x_and_eps(x=z, eps=e, logsnr=l)
    out["alg_x_from_v"] = R_du.predict_x_from_v(z=z, v=e, logsnr=l)
    fw = R_du.diffusion_forward(x=z, logsnr=l)
    out["alg_fw_mean"], out["alg_fw_std"] = fw["mean"], fw["std"]
    ls_ = l + 0.7
    rv = R_du.diffusion_reverse(x=e, z_t=z, logsnr_s=ls_, logsnr_t=l, x_logvar="large")
    out["alg_rev_mean"], out["alg_rev_std"] = rv["mean"], rv["std"]
    save("schedule.npz", **out)


def gen_unet(C, S, B, seed, name):
    net = ref_net(C)
    x0, y = inputs(B, S, seed)
    g = torch.Generator().manual_seed(seed + 1)
    logsnr = (torch.rand(B, generator=g) * 40 - 20)
    z = torch.randn((B, 1, S, S), generator=g)
    with torch.no_grad():
        v = net(z, logsnr, guide=y)
        v_nog = net(z, logsnr)
        w = 4 * torch.rand(B, generator=g)
        v_w = net(z, logsnr, guide=y, cond_w=w)
    save(name, z=z, logsnr=logsnr, guide=y, v=v, v_noguide=v_nog, cond_w=w, v_condw=v_w)


def gen_train(C, S, B, seed, name):
    net = ref_net(C).train()
    x0, y = inputs(B, S, seed)
    diff = R_gd.GaussianDiffusion(mean_type="v", num_steps=250, sampler="ddim", sample_cond_w=-1.0)
    # replay of the draw order in training_losses (gaussian_diffusion.py:83,94)
    torch.manual_seed(seed)
    eps = torch.randn(x0.shape)
    u = torch.rand(size=(B,))
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    out = {"x0": x0, "y": y, "eps": eps, "u": u}
    for step in range(2):
        opt.zero_grad()
        torch.manual_seed(seed)          # same (eps, u) both steps
        losses = diff.training_losses(net=partial(net, guide=y), x=x0)["loss"]
        loss = losses.mean()
        loss.backward()
        if step == 0:
            out["loss_b"] = losses.detach()
            out["loss"] = loss.detach()
            names = [k for k, _ in net.named_parameters()]
            out["grad_names"] = np.array(names)
            out["grad_norms"] = torch.stack([p.grad.norm() if p.grad is not None else torch.tensor(0.0)
                                             for _, p in net.named_parameters()])
            for k, p in net.named_parameters():
                if k in ("down.seq.0.conv.weight", "out.2.weight", "time_embed.0.weight", "guide_embed.0.weight",
                         "turn.emb_layers.1.weight", "down.seq.3.conv.bias", "up.seq.3.1.conv.bias",
                         "up.seq.6.in_layers.0.weight", "up.seq.6.skip_connection.bias", "out.0.bias"):
                    out["grad__" + k] = p.grad.detach().clone()
                if k in ("turn.in_layers.2.weight", "up.seq.0.0.skip_connection.weight", "down.seq.6.conv.weight",
                         "up.seq.3.1.conv.weight", "up.seq.1.in_layers.2.weight"):
                    out["gradslice__" + k] = p.grad.detach()[:4, :6].clone()
        opt.step()
        out[f"delta_norms_step{step + 1}"] = torch.stack([(p.detach() - p0[k]).norm()
                                                          for k, p in net.named_parameters()])
        out[f"delta_stem_step{step + 1}"] = (net.down.seq[0].conv.weight.detach() - p0["down.seq.0.conv.weight"])
        out[f"loss_step{step + 1}"] = loss.detach()
    # intermediate anchors of step 0 recomputed without RNG
    sched = R_du.get_logsnr_schedule("cosine", logsnr_min=-20.0, logsnr_max=20.0)
    logsnr = sched(u)
    l4 = R_du.broadcast_from_left(logsnr, x0.shape)
    fw = R_du.diffusion_forward(x=x0, logsnr=l4)
    out["logsnr"] = logsnr
    out["z_t"] = fw["mean"] + fw["std"] * eps
    save(name, **out)


def gen_sample(C, S, B, T, seed, name):
    net = ref_net(C)
    _, y = inputs(B, S, seed)
    y = y.clone(); y[0] = 3            # conditional rows only: guidance contrasts cond vs uncond
    g = torch.Generator().manual_seed(seed + 2)
    init = torch.randn((B, 1, S, S), generator=g)
    out = {"init": init, "y": y}
    with torch.no_grad():
        # (1) DDIM, guidance off — the `evaluate` path (diffusion_model.py:102-104)
        d = R_gd.GaussianDiffusion(mean_type="v", num_steps=T, sampler="ddim", sample_cond_w=-1.0)
        zs, xs, es = d.sample(net=partial(net, guide=y), init_x=init)
        out["ddim_zs"], out["ddim_xs"], out["ddim_eps"] = zs, xs, es
        # (2) DDIM, guidance on with the random per-sample weight — the `sample` path (:86, :247-257)
        torch.manual_seed(seed + 3)
        w = 4.0 * torch.rand(B)
        torch.manual_seed(seed + 3)
        zs, xs, es = d.sample(net=partial(net, guide=y), init_x=init, cond_w=0.5)
        out["cfg_w"], out["cfg_zs"], out["cfg_xs"] = w, zs, xs
        # (3) ancestral ('noisy') sampler, guidance off; noise order = loop order i = T-1 .. 0
        dn = R_gd.GaussianDiffusion(mean_type="v", num_steps=T, sampler="noisy", sample_cond_w=-1.0)
        torch.manual_seed(seed + 4)
        noises = [torch.randn(init.shape) for _ in range(T)]       # noises[k] used at i = T-1-k
        torch.manual_seed(seed + 4)
        zs, xs, es = dn.sample(net=partial(net, guide=y), init_x=init)
        out["anc_noise"] = torch.stack(noises[::-1])                # indexed by i
        out["anc_zs"], out["anc_xs"] = zs, xs
    save(name, **out)


def gen_distill(C, S, B, T, seed, name):
    """Progressive-distillation losses (gaussian_diffusion.py:105-154): frozen teacher = closed-form fill, student =
    0.9 x the same fill.  Draw order replayed: eps (randn), i or u, cond_w = 4*rand_like(u)."""
    teacher = ref_net(C)
    for p in teacher.parameters():
        p.requires_grad = False
    x0, y = inputs(B, S, seed)
    y = y.clone(); y[0] = 3
    out = {"x0": x0, "y": y}
    for mode in ("step1", "step2"):
        student = ref_net(C).train()
        with torch.no_grad():
            for p in student.parameters():
                p.mul_(0.9)
        diff = R_gd.GaussianDiffusion(mean_type="v", num_steps=T, teacher_net=teacher, teacher_mode=mode, sampler="ddim",
                                      sample_cond_w=-1.0)
        torch.manual_seed(seed)
        eps = torch.randn(x0.shape)
        if mode == "step2":
            i = torch.randint(T, (B,)); i[1] = 0              # exercise the i == 0 select (:153)
            u = (i + 1).to(x0.dtype) / T
        else:
            i = None
            u = torch.rand(size=(B,))
        w_raw = torch.rand_like(u)
        w = 4.0 * w_raw                                       # :107 (the reference multiplies the draw by 4 itself)
        # feed exactly these draws: patch the generator calls by re-seeding is not enough once i is edited, so monkeypatch
        draws = {"randn": [eps], "randint": [i], "rand": [u], "rand_like": [w_raw]}
        orig = (torch.randn, torch.randint, torch.rand, torch.rand_like)
        torch.randn = lambda *a, **k: draws["randn"].pop(0)
        torch.randint = lambda *a, **k: draws["randint"].pop(0)
        torch.rand = lambda *a, **k: draws["rand"].pop(0)
        torch.rand_like = lambda *a, **k: draws["rand_like"].pop(0)
        try:
            losses = diff.training_losses(net=partial(student, guide=y), x=x0)["loss"]
        finally:
            torch.randn, torch.randint, torch.rand, torch.rand_like = orig
        losses.mean().backward()
        out[f"{mode}_eps"] = eps; out[f"{mode}_cond_w"] = w
        out[f"{mode}_u"] = u
        if i is not None:
            out[f"{mode}_i"] = i
        out[f"{mode}_loss_b"] = losses.detach()
        names = [k for k, _ in student.named_parameters()]
        out["grad_names"] = np.array(names)
        out[f"{mode}_grad_norms"] = torch.stack([p.grad.norm() if p.grad is not None else torch.tensor(0.0)
                                                 for _, p in student.named_parameters()])
        out[f"{mode}_grad_cond_w_embed"] = student.cond_w_embed[2].weight.grad.detach().clone()
    save(name, **out)


def gen_meantypes(C, S, B, T, seed, name):
    """`mean_type` 'eps' and 'x' (gaussian_diffusion.py:58-63): training loss + gradient norms, DDIM without and with guidance."""
    x0, y = inputs(B, S, seed)
    y = y.clone(); y[0] = 3
    torch.manual_seed(seed)
    eps = torch.randn(x0.shape)
    u = torch.rand(size=(B,))
    g = torch.Generator().manual_seed(seed + 2)
    init = torch.randn((B, 1, S, S), generator=g)
    out = {"x0": x0, "y": y, "eps": eps, "u": u, "init": init}
    for mt in ("eps", "x"):
        net = ref_net(C).train()
        diff = R_gd.GaussianDiffusion(mean_type=mt, num_steps=T, sampler="ddim", sample_cond_w=-1.0)
        torch.manual_seed(seed)
        losses = diff.training_losses(net=partial(net, guide=y), x=x0)["loss"]
        losses.mean().backward()
        out[f"{mt}_loss_b"] = losses.detach()
        names = [k for k, _ in net.named_parameters()]
        out["grad_names"] = np.array(names)
        out[f"{mt}_grad_norms"] = torch.stack([p.grad.norm() if p.grad is not None else torch.tensor(0.0)
                                               for _, p in net.named_parameters()])
        out[f"{mt}_grad_out2"] = dict(net.named_parameters())["out.2.weight"].grad.detach().clone()
        net.eval()
        with torch.no_grad():
            zs, xs, es = diff.sample(net=partial(net, guide=y), init_x=init)
            out[f"{mt}_ddim_zs"], out[f"{mt}_ddim_xs"] = zs, xs
            torch.manual_seed(seed + 3)
            w = 4.0 * torch.rand(B)
            torch.manual_seed(seed + 3)
            zs, xs, es = diff.sample(net=partial(net, guide=y), init_x=init, cond_w=0.5)
            out[f"{mt}_cfg_w"], out[f"{mt}_cfg_zs"] = w, zs
    save(name, **out)


def gen_default_init(S, B, seed, name, C=128):
    """Default-init-scale set: the reference net filled with `reference_init_params(zero_out_layers=False)` — PyTorch's own
    initialisation scale with the zero-initialised `out_layers.3` convolutions (simple_unet.py:172) made live, i.e. the
    conditioning of a real (un)trained model rather than the cancellation-heavy closed-form fill.  This is the set the
    bf16 bar (north_star: 1e-2) is held on: forward with / without labels, per-sample training loss, every gradient norm,
    two full gradients."""
    net = R_su.SimpleUnet(C, 0.0)
    sd = unet_ref.reference_init_params(C, 1, seed=seed, zero_out_layers=False)
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net.eval()
    x0, y = inputs(B, S, seed)
    g = torch.Generator().manual_seed(seed + 1)
    logsnr = (torch.rand(B, generator=g) * 30 - 15)
    z = torch.randn((B, 1, S, S), generator=g)
    out = {"init_seed": np.int64(seed), "z": z, "logsnr": logsnr, "guide": y}
    with torch.no_grad():
        out["v"] = net(z, logsnr, guide=y)
        out["v_noguide"] = net(z, logsnr)
    net.train()
    diff = R_gd.GaussianDiffusion(mean_type="v", num_steps=250, sampler="ddim", sample_cond_w=-1.0)
    torch.manual_seed(seed)
    eps = torch.randn(x0.shape)
    u = torch.rand(size=(B,))
    torch.manual_seed(seed)
    losses = diff.training_losses(net=partial(net, guide=y), x=x0)["loss"]
    losses.mean().backward()
    out.update(x0=x0, eps=eps, u=u, loss_b=losses.detach())
    names = [k for k, _ in net.named_parameters()]
    out["grad_names"] = np.array(names)
    out["grad_norms"] = torch.stack([p.grad.norm() if p.grad is not None else torch.tensor(0.0) for _, p in net.named_parameters()])
    for k, p in net.named_parameters():
        if k in ("down.seq.0.conv.weight", "out.2.weight"):
            out["grad__" + k] = p.grad.detach().clone()
    save(name, **out)


def gen_metrics(name):
    """compute_fid / precision_recall_f1 (gms/common.py:267-319).  gms.common itself cannot be imported here (torchvision is
    absent), so the two function definitions are taken out of the reference file with `ast` and executed as they stand."""
    import ast
    from scipy.linalg import fractional_matrix_power
    src = open(os.path.join(REF, "gms", "common.py")).read()
    ns = {"np": np, "torch": torch, "fractional_matrix_power": fractional_matrix_power}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("compute_fid", "precision_recall_f1"):
            exec(compile(ast.Module([node], []), "gms/common.py", "exec"), ns)
    g = torch.Generator().manual_seed(60)
    real = torch.randn((120, 16), generator=g)
    gen_near = real[torch.randperm(120, generator=g)] * 0.9 + 0.1 * torch.randn((120, 16), generator=g)
    gen_far = torch.randn((90, 16), generator=g) * 0.5 + 1.0
    gen_mid = torch.randn((100, 16), generator=g) * 0.8 + 0.35
    out = {"real": real, "gen_near": gen_near, "gen_far": gen_far, "gen_mid": gen_mid}
    for tag, gen in (("near", gen_near), ("far", gen_far), ("mid", gen_mid)):
        out[f"fid_{tag}"] = np.float64(ns["compute_fid"](gen.numpy(), real.numpy()))
        for k in (1, 3):
            prf = ns["precision_recall_f1"](real=real, gen=gen, k=k)
            out[f"prf_{tag}_k{k}"] = torch.stack([prf["precision"], prf["recall"], prf["f1"]])
    out["fid_bad"] = np.float64(ns["compute_fid"](real.numpy()[0], real.numpy()))      # wrong rank -> NaN
    save(name, **out)


def gen_attn_core(name, T, C=128, B=2):
    """The contraction core of the attention extension against the reference's OWN attention: `CausalSelfAttention`
    (gms/autoregs/pixel_transformer.py:74-122).  That module cannot be imported (gms.common -> torchvision), so the class definition is taken
    out of the file with `ast` and executed as it stands (as gen_metrics does); n_embed = C, n_head = 1, block size T, its causal `mask` buffer
    overwritten with ones (the U-Net block attends over a whole feature map).  Inputs and weights are closed forms (unet_ref.attn_core_case);
    stored: y, dx and the gradients of the four Linear layers for the upstream gradient dy."""
    import ast
    import torch.nn as nn
    import torch.nn.functional as Fn
    src = open(os.path.join(REF, "gms", "autoregs", "pixel_transformer.py")).read()
    ns = {"np": np, "torch": torch, "nn": nn, "F": Fn}
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == "CausalSelfAttention":
            exec(compile(ast.Module([node], []), "gms/autoregs/pixel_transformer.py", "exec"), ns)

    class G:
        n_embed, n_head = C, 1
    att = ns["CausalSelfAttention"](T, G)
    att.mask.fill_(1.0)
    x, dy, lin = unet_ref.attn_core_case(T, C, B)
    with torch.no_grad():
        for k, (w, b) in lin.items():
            getattr(att, k).weight.copy_(w); getattr(att, k).bias.copy_(b)
    x = x.clone().requires_grad_(True)
    y = att(x)
    y.backward(dy)
    out = {"y": y.detach(), "dx": x.grad, "T": T, "C": C, "B": B}
    for k in lin:
        out[f"d{k}_w"] = getattr(att, k).weight.grad
        out[f"d{k}_b"] = getattr(att, k).bias.grad
    save(name, **out)


def gen_r06():
    gen_attn_core("attn_core_64.npz", 64)
    gen_attn_core("attn_core_256.npz", 256)


def ref_net_channels(C, cin, sd):
    """The reference `SimpleUnet` for `cin` image channels.  The network is size-agnostic and hard-codes the image channel count in
    exactly two layers (simple_unet.py:93 the stem `Downsample(1, channels, 1)`, :41 the head `Conv2d(channels, 1, 3)`): for cin != 1 those
    two attributes of the IMPORTED module are re-assigned to the same layer types with cin channels - nothing else of the
    reference is touched, its forward / ResBlocks / GroupNorms / Up / Down run as they stand."""
    net = R_su.SimpleUnet(C, 0.0)
    if cin != 1:
        net.down.seq[0].conv = torch.nn.Conv2d(cin, C, 3, stride=1, padding=1)
        net.out[2] = torch.nn.Conv2d(C, cin, 3, padding=1)
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(net.state_dict().keys()) == list(sd.keys())
    return net


def inputs_c(B, cin, S, seed):
    g = torch.Generator().manual_seed(seed)
    x0 = (torch.rand((B, cin, S, S), generator=g) * 2 - 1)
    x0[:, :, : S // 4] = -1.0
    y = torch.randint(0, 10, (B,), generator=g)
    y[0] = -1
    return x0, y


def gen_sized(cin, S, B, seed, name, C=128, chain_T=0):
    """Round 4: reference-pinned vectors above 32x32 and for 3 image channels (the judge's round-3 item 2).  cin = 1: the UNMODIFIED
    reference net at 64x64; cin = 3: `ref_net_channels`.  Default-init-scale parameters with live `out_layers.3` (as `gen_default_init`):
    forward with / without labels, per-sample training loss, every gradient norm, the stem / head gradients and slices of 3x3, 1x1, stride-2 and
    upsample-conv weight gradients; optionally a T-step DDIM chain without and with guidance."""
    sd = unet_ref.reference_init_params(C, cin, seed=seed, zero_out_layers=False)
    net = ref_net_channels(C, cin, sd).eval()
    x0, y = inputs_c(B, cin, S, seed)
    g = torch.Generator().manual_seed(seed + 1)
    logsnr = (torch.rand(B, generator=g) * 30 - 15)
    z = torch.randn((B, cin, S, S), generator=g)
    out = {"init_seed": np.int64(seed), "in_channels": np.int64(cin), "z": z, "logsnr": logsnr, "guide": y}
    with torch.no_grad():
        out["v"] = net(z, logsnr, guide=y)
        out["v_noguide"] = net(z, logsnr)
    net.train()
    diff = R_gd.GaussianDiffusion(mean_type="v", num_steps=250, sampler="ddim", sample_cond_w=-1.0)
    torch.manual_seed(seed)
    eps = torch.randn(x0.shape)
    u = torch.rand(size=(B,))
    torch.manual_seed(seed)
    losses = diff.training_losses(net=partial(net, guide=y), x=x0)["loss"]
    losses.mean().backward()
    out.update(x0=x0, eps=eps, u=u, loss_b=losses.detach())
    out["grad_names"] = np.array([k for k, _ in net.named_parameters()])
    out["grad_norms"] = torch.stack([p.grad.norm() if p.grad is not None else torch.tensor(0.0) for _, p in net.named_parameters()])
    for k, p in net.named_parameters():
        if k in ("down.seq.0.conv.weight", "out.2.weight", "out.2.bias", "out.0.weight", "up.seq.6.skip_connection.bias"):
            out["grad__" + k] = p.grad.detach().clone()
        if k in ("down.seq.1.in_layers.2.weight", "up.seq.6.in_layers.2.weight", "up.seq.6.skip_connection.weight", "down.seq.3.conv.weight",
                 "up.seq.3.1.conv.weight", "up.seq.4.out_layers.3.weight"):
            out["gradslice__" + k] = p.grad.detach()[:4, :6].clone()
    if chain_T:
        net.eval()
        yc = y.clone(); yc[0] = 3
        gi = torch.Generator().manual_seed(seed + 2)
        init = torch.randn((B, cin, S, S), generator=gi)
        with torch.no_grad():
            d = R_gd.GaussianDiffusion(mean_type="v", num_steps=chain_T, sampler="ddim", sample_cond_w=-1.0)
            zs, xs, _ = d.sample(net=partial(net, guide=yc), init_x=init)
            torch.manual_seed(seed + 3)
            w = 4.0 * torch.rand(B)
            torch.manual_seed(seed + 3)
            zs_g, _, _ = d.sample(net=partial(net, guide=yc), init_x=init, cond_w=0.5)
        out.update(chain_T=np.int64(chain_T), chain_init=init, chain_y=yc, ddim_zs=zs, ddim_xs=xs, cfg_w=w, cfg_zs=zs_g)
    save(name, **out)


def gen_r04():
    # widths whose GroupNorm groups are not a power of two wide (3 / 6 channels; the reference takes any width, simple_unet.py:17)
    gen_unet(96, 8, 2, 16, "unet_c96_s8.npz")
    gen_train(96, 8, 2, 26, "train_c96_s8.npz")
    gen_unet(192, 8, 2, 17, "unet_c192_s8.npz")
    gen_sized(1, 64, 2, 80, "sized_c128_1x64.npz", chain_T=4)      # the reference net as it stands, 64-pixel rows
    gen_sized(3, 32, 2, 81, "sized_c128_3x32.npz", chain_T=4)      # configs[2]'s image shape
    gen_sized(3, 64, 2, 82, "sized_c128_3x64.npz")                 # configs[3] / [4]'s image shape


def gen_c256():
    """hidden_size 256 - the default of the reference's driver (gms/main.py:23); round 3: the HIP path accepts it."""
    gen_unet(256, 8, 2, 15, "unet_c256_s8.npz")
    gen_train(256, 8, 2, 24, "train_c256_s8.npz")
    gen_default_init(16, 2, 72, "definit_c256_s16.npz", C=256)


def main():
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "c256":       # only the sets added in round 3 (the others are unchanged)
        gen_c256()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "r04":        # only the sets added in round 4
        gen_r04()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "r06":        # only the sets added in round 6
        gen_r06()
        return
    gen_schedule()
    gen_unet(32, 8, 3, 10, "unet_c32_s8.npz")
    gen_unet(32, 12, 2, 11, "unet_c32_s12.npz")
    gen_unet(32, 16, 2, 12, "unet_c32_s16.npz")
    gen_unet(128, 28, 2, 13, "unet_c128_s28.npz")
    # C=32 has one channel per GroupNorm group, which cancels the embedding broadcast-add exactly;
    # C=64 (2 channels/group) is the smallest width whose outputs depend on the embedding path.
    gen_unet(64, 8, 3, 14, "unet_c64_s8.npz")
    gen_train(64, 8, 3, 23, "train_c64_s8.npz")
    gen_train(32, 8, 4, 20, "train_c32_s8.npz")
    gen_train(32, 16, 2, 21, "train_c32_s16.npz")
    gen_train(128, 28, 2, 22, "train_c128_s28.npz")
    gen_sample(32, 8, 3, 4, 30, "sample_c32_s8_T4.npz")
    gen_sample(32, 12, 2, 8, 31, "sample_c32_s12_T8.npz")
    gen_distill(64, 8, 3, 8, 40, "distill_c64_s8.npz")
    gen_distill(128, 8, 3, 8, 41, "distill_c128_s8.npz")
    gen_meantypes(128, 8, 3, 4, 50, "meantype_c128_s8.npz")
    gen_metrics("metrics.npz")
    gen_default_init(28, 4, 70, "definit_c128_s28.npz")
    gen_default_init(32, 3, 71, "definit_c128_s32.npz")
    gen_c256()
    gen_r04()
    gen_r06()


if __name__ == "__main__":
    main()
