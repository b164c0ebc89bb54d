"""Prints how far whole TRAINING TRAJECTORIES stay together (not a test: `python tests/trajectory_report.py [steps]` on a GPU box).

The same run three ways from the same initial parameters and the same per-step (x0, y, u, eps): the CPU oracle (torch fp32 restatement of the
reference: training_losses -> backward -> Adam, oracle/diffusion_ref.py), the HIP path in fp32 mode, and the HIP path in the 16-bit mode
(fp16 forward + bf16 gradients) - B = 32, 1x28x28 (BASELINE configs[0]'s shape), lr 3e-4, default-init-scale weights.  Data: a fixed random
ink mask per class (15 % ink) with uniform ink values, background -1, so that the labels carry information.  Prints the loss of every 10th
step for the three runs, the largest relative deviation of the smoothed loss curves and the relative distance of the final parameters.
"""
import os
import sys
from functools import partial

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion  # noqa: E402
from generative_models_amd.diffusion.optim import FusedAdam  # noqa: E402
from generative_models_amd.diffusion.simple_unet import SimpleUnet  # noqa: E402
from oracle import diffusion_ref as D  # noqa: E402
from oracle import unet_ref as U  # noqa: E402


def batches(steps, B, S=28, seed=0):
    g = torch.Generator().manual_seed(seed)
    masks = torch.rand((10, 1, S, S), generator=g) < 0.15
    for _ in range(steps):
        y = torch.randint(0, 10, (B,), generator=g)
        x0 = torch.where(masks[y], torch.rand((B, 1, S, S), generator=g) * 2 - 1, torch.full((B, 1, S, S), -1.0))
        u = torch.rand((B,), generator=g)
        eps = torch.randn((B, 1, S, S), generator=g)
        yield x0, y, u, eps


def run_oracle(params, steps, B):
    torch.set_num_threads(min(16, os.cpu_count() or 8))      # the GPU box's CPU share
    p = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    losses = []
    for step, (x0, y, u, eps) in enumerate(batches(steps, B), 1):
        q = {k: t.clone().requires_grad_(True) for k, t in p.items()}
        loss = D.training_losses(q, x0, y, u, eps)["loss"].mean()
        loss.backward()
        for k in p:
            if q[k].grad is not None:
                p[k], m[k], v2[k] = D.adam_step(p[k], q[k].grad, m[k], v2[k], step)
        losses.append(float(loss.detach()))
        if step % 10 == 0:
            print(f"oracle step {step}: loss {losses[-1]:.5f}", flush=True)
    return losses, p


def run_hip(params, steps, B, dtype):
    net = SimpleUnet(128, 0.0, compute_dtype=dtype)
    net.load_state_dict(params, strict=True)
    net = net.cuda()
    opt = FusedAdam(net, lr=3e-4)
    diff = GaussianDiffusion(mean_type="v", num_steps=250)
    losses = []
    for x0, y, u, eps in batches(steps, B):
        out = diff.train_forward_backward(net=partial(net, guide=y.cuda()), x=x0.cuda(), grad_scale=1.0 / B, u=u.cuda(), eps=eps.cuda())
        opt.step()
        losses.append(float(out["loss"].mean()))
    return losses, {k: v.detach().float().cpu() for k, v in net.state_dict().items()}


def smooth(x, w=10):
    t = torch.tensor(x)
    return torch.stack([t[max(0, i - w + 1):i + 1].mean() for i in range(len(x))])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    B = 32
    params = U.reference_init_params(128, 1, seed=0, zero_out_layers=True)      # the reference's initialisation (out_layers.3 zeroed)
    lo, po = run_oracle(params, steps, B)
    l32, p32 = run_hip(params, steps, B, torch.float32)
    l16, p16 = run_hip(params, steps, B, torch.bfloat16)
    print(f"{steps} Adam steps, B = {B}, 1x28x28, lr 3e-4; loss of every 10th step: oracle (CPU fp32) | HIP fp32 | HIP 16-bit")
    for i in range(0, steps, 10):
        print(f"  step {i + 1:4d}: {lo[i]:.5f} | {l32[i]:.5f} | {l16[i]:.5f}")
    so, s32, s16 = smooth(lo), smooth(l32), smooth(l16)
    dev = lambda a, b: float(((a - b).abs() / b.abs()).max())
    print(f"first / last-10 mean loss: oracle {lo[0]:.4f} / {float(so[-1]):.4f}, HIP fp32 {l32[0]:.4f} / {float(s32[-1]):.4f}, HIP 16-bit {l16[0]:.4f} / {float(s16[-1]):.4f}")
    print(f"largest relative deviation of the 10-step-smoothed loss curve from the oracle's: HIP fp32 {dev(s32, so):.2e}, HIP 16-bit {dev(s16, so):.2e}")
    print(f"largest per-step relative loss deviation over the first 20 steps: HIP fp32 {dev(torch.tensor(l32[:20]), torch.tensor(lo[:20])):.2e}, "
          f"HIP 16-bit {dev(torch.tensor(l16[:20]), torch.tensor(lo[:20])):.2e}")
    pd = lambda a: float(torch.sqrt(sum(((a[k] - po[k]) ** 2).sum() for k in po) / sum(((po[k] - params[k]) ** 2).sum() for k in po)))
    print(f"distance of the final parameters from the oracle's, relative to the distance the oracle's travelled: HIP fp32 {pd(p32):.3f}, HIP 16-bit {pd(p16):.3f}")


if __name__ == "__main__":
    main()
