"""Prints the measured parity numbers DESIGN.md quotes (not a test: `python tests/parity_report.py` on a GPU box).

For every reference-pinned default-init fixture (tests/golden/definit_*.npz, generated from the imported reference by
oracle/make_golden.py) and both compute modes: max-norm and L2 relative error of the U-Net output with / without labels, of the
per-sample loss, the worst gradient-norm deviation and the error of the two stored full gradients.
"""
import os
import sys
from functools import partial

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion  # noqa: E402
from generative_models_amd.diffusion.simple_unet import SimpleUnet  # noqa: E402
from oracle import unet_ref as U  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def errs(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())


MODES = ((torch.float32, torch.float32), (torch.bfloat16, torch.float16), (torch.bfloat16, torch.bfloat16))


def mode_name(dtype, act):
    return "fp32" if dtype == torch.float32 else ("fp16 fwd + bf16 bwd" if act == torch.float16 else "all bf16 (rounds 1-2)")


def ragged():
    """The oracle-only shapes of tests/test_gpu_unet.py::test_odd_batches_and_sizes_bf16_vs_fp32_vs_oracle and of the config tests
    (3-channel nets: parity unpinned), forward in training mode (activations kept), per storage mode."""
    for cin, B, S in ((1, 1, 28), (1, 5, 32), (1, 7, 28), (1, 3, 12), (3, 3, 32), (3, 2, 64)):
        params = U.reference_init_params(128, cin, seed=0, zero_out_layers=False)
        g = torch.Generator().manual_seed(B * 100 + S)
        z = torch.randn((B, cin, S, S), generator=g)
        l = torch.rand((B,), generator=g) * 30 - 15
        y = torch.randint(-1, 10, (B,), generator=g)
        with torch.no_grad():
            ref = U.unet_forward(params, z, l, guide=y)
        row = []
        for dtype, act in MODES:
            net = SimpleUnet(128, 0.0, in_channels=cin, compute_dtype=dtype, act_dtype=act)
            net.load_state_dict(params, strict=True)
            net = net.cuda()
            out = net.forward_hip(z.cuda(), l.cuda(), y.cuda(), None, ctx={})
            e = errs(out, ref)
            row.append(f"{mode_name(dtype, act)}: max {e[0]:.2e} L2 {e[1]:.2e}")
        print(f"ragged {cin}x{S}x{S} B={B} vs oracle | " + " | ".join(row), flush=True)


def main():
    T = torch.from_numpy
    for name in ("definit_c128_s28.npz", "definit_c128_s32.npz"):
        g = np.load(os.path.join(GOLD, name), allow_pickle=True)
        params = U.reference_init_params(128, 1, seed=int(g["init_seed"]), zero_out_layers=False)
        for dtype, act in MODES:
            net = SimpleUnet(128, 0.0, compute_dtype=dtype, act_dtype=act)
            net.load_state_dict(params, strict=True)
            net = net.cuda()
            z, l, y = T(g["z"]).cuda(), T(g["logsnr"]).cuda(), T(g["guide"]).cuda()
            with torch.no_grad():
                e_v, e_n = errs(net(z, l, guide=y), T(g["v"])), errs(net(z, l), T(g["v_noguide"]))
            diff = GaussianDiffusion(mean_type="v", num_steps=250)
            x0, u, eps = (T(g[k]).cuda() for k in ("x0", "u", "eps"))
            out = diff.train_forward_backward(net=partial(net, guide=y), x=x0, grad_scale=1.0 / x0.shape[0], u=u, eps=eps)
            e_l = errs(out["loss"], T(g["loss_b"]))
            names = [str(n) for n in g["grad_names"]]
            norms = torch.stack([net.grad(n).norm() for n in names]).cpu()
            ref = T(g["grad_norms"])
            worst = float(((norms - ref).abs() / ref.abs().clamp_min(1e-3 * float(ref.abs().max()))).max())
            eg = {k[6:]: errs(net.grad(k[6:]), T(g[k])) for k in g.files if k.startswith("grad__")}
            print(f"{name} {mode_name(dtype, act):22s} v max {e_v[0]:.2e} L2 {e_v[1]:.2e} | no-label max {e_n[0]:.2e} L2 {e_n[1]:.2e} | "
                  f"loss max {e_l[0]:.2e} | worst grad-norm dev {worst:.2e} | " +
                  " ".join(f"{k} max {v[0]:.2e}" for k, v in eg.items()), flush=True)


def widths():
    """Reference outputs at the other widths (closed-form fill, tests/golden/unet_c*_*.npz): 256 native, 64 / 32 zero-padded to 128 channels."""
    T = torch.from_numpy
    for name, C in (("unet_c256_s8.npz", 256), ("unet_c64_s8.npz", 64), ("unet_c32_s16.npz", 32)):
        g = np.load(os.path.join(GOLD, name), allow_pickle=True)
        params = U.closed_form_params(C)
        row = []
        for dtype, act in MODES:
            net = SimpleUnet(C, 0.0, compute_dtype=dtype, act_dtype=act)
            net.load_state_dict(params, strict=True)
            net = net.cuda()
            z, l, y = T(g["z"]).cuda(), T(g["logsnr"]).cuda(), T(g["guide"]).cuda()
            with torch.no_grad():
                e = errs(net(z, l, guide=y), T(g["v"]))
            row.append(f"{mode_name(dtype, act)}: max {e[0]:.2e} L2 {e[1]:.2e}")
        print(f"{name} (hidden_size {C}, closed-form fill) vs the reference | " + " | ".join(row), flush=True)


def gradient_distribution():
    """What the 16-bit mode's gradient statement looks like tensor by tensor: every parameter gradient of a train pass at
    default-init scale against the oracle's fp32 autograd gradient of the same inputs, three seeds: how many of the 160 tensors are within 1e-2 of
    their largest entry, the quantiles, the worst five by name."""
    from oracle import diffusion_ref as D
    for seed in (0, 1, 2):
        params = U.reference_init_params(128, 1, seed=seed, zero_out_layers=False)
        g = torch.Generator().manual_seed(50 + seed)
        B, S = 4, 28
        x0 = torch.rand((B, 1, S, S), generator=g) * 2 - 1
        y = torch.randint(-1, 10, (B,), generator=g)
        u = torch.rand((B,), generator=g); eps = torch.randn((B, 1, S, S), generator=g)
        pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        D.training_losses(pr, x0, y, u, eps)["loss"].mean().backward()
        net = SimpleUnet(128, 0.0, compute_dtype=torch.bfloat16, act_dtype=torch.float16)
        net.load_state_dict(params, strict=True)
        net = net.cuda()
        diff = GaussianDiffusion(mean_type="v", num_steps=250)
        diff.train_forward_backward(net=partial(net, guide=y.cuda()), x=x0.cuda(), grad_scale=1.0 / B, u=u.cuda(), eps=eps.cuda())
        rows = []
        for n, p in pr.items():
            if p.grad is None or float(p.grad.abs().max()) == 0.0:
                continue
            e = float((net.grad(n).float().cpu() - p.grad).abs().max() / p.grad.abs().max())
            rows.append((e, n))
        rows.sort()
        es = [e for e, _ in rows]
        q = lambda f: es[min(len(es) - 1, int(f * len(es)))]
        gh = torch.cat([net.grad(n).double().cpu().reshape(-1) for _, n in rows]); gr = torch.cat([pr[n].grad.double().reshape(-1) for _, n in rows])
        print(f"16-bit gradients, seed {seed}: {sum(e <= 1e-2 for e in es)} of {len(es)} tensors within 1e-2 (max-norm, of the tensor's largest entry); "
              f"median {q(0.5):.2e}, 90 % {q(0.9):.2e}, worst {es[-1]:.2e}; whole gradient: cosine {float(torch.nn.functional.cosine_similarity(gh, gr, dim=0)):.6f}, "
              f"relative L2 {float((gh - gr).norm() / gr.norm()):.2e}; worst five: " + ", ".join(f"{n} {e:.1e}" for e, n in rows[-5:]), flush=True)


def fp32_split():
    """The fp32 mode's fast form (products as three bf16 MFMAs of hi / lo halves, gmk_set_fp32_exact(0)) on the default-init vectors."""
    from generative_models_amd._lib import lib
    T = torch.from_numpy
    try:
        lib.gmk_set_fp32_exact(0)
        for name in ("definit_c128_s28.npz", "definit_c128_s32.npz"):
            g = np.load(os.path.join(GOLD, name), allow_pickle=True)
            params = U.reference_init_params(128, 1, seed=int(g["init_seed"]), zero_out_layers=False)
            net = SimpleUnet(128, 0.0, compute_dtype=torch.float32)
            net.load_state_dict(params, strict=True)
            net = net.cuda()
            z, l, y = T(g["z"]).cuda(), T(g["logsnr"]).cuda(), T(g["guide"]).cuda()
            with torch.no_grad():
                e_v = errs(net(z, l, guide=y), T(g["v"]))
            diff = GaussianDiffusion(mean_type="v", num_steps=250)
            x0, u, eps = (T(g[k]).cuda() for k in ("x0", "u", "eps"))
            out = diff.train_forward_backward(net=partial(net, guide=y), x=x0, grad_scale=1.0 / x0.shape[0], u=u, eps=eps)
            e_l = errs(out["loss"], T(g["loss_b"]))
            eg = max(errs(net.grad(k[6:]), T(g[k]))[0] for k in g.files if k.startswith("grad__"))
            print(f"{name} fp32 split (3x bf16 MFMA)  v max {e_v[0]:.2e} L2 {e_v[1]:.2e} | loss max {e_l[0]:.2e} | stored gradients max {eg:.2e}", flush=True)
    finally:
        lib.gmk_set_fp32_exact(1)


if __name__ == "__main__":
    main()
    ragged()
    widths()
    fp32_split()
    gradient_distribution()
