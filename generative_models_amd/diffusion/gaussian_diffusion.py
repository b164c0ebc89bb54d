"""Host-side mirror of reference gms/diffusion/gaussian_diffusion.py (`GaussianDiffusion`, :19-296) over the HIP kernels.

Same constructor keywords, same `training_losses(net=, x=)` / `sample(net=, init_x=, cond_w=)` call shapes and
return structure; `net` is the HIP `SimpleUnet` or a `functools.partial` of it carrying `guide=` / `cond_w=` exactly
as the reference passes it (diffusion_model.py:77,85).  All arithmetic is in libgmk.so:

* q_sample + log-SNR schedule          gmk_q_sample     (:94-100, diffusion_utils.py:65-73,198-201)
* v -> x_hat/eps_hat, clip, loss, dL/dv gmk_v_loss       (:61-77,:165-169 and their backward)
* DDIM / ancestral / guidance update   gmk_sampler_step (:174-243,:292)
* RNG                                   counter-based Philox streams (gmk_rng_*), keyed (seed, rank, draw index)

Only `mean_type='v'` (the reference default, diffusion_model.py:21) is implemented on the HIP path; the progressive
distillation branches (:105-154, teacher_net) are SURVEY §8f "next" and raise NotImplementedError.
"""
from functools import partial

import numpy as np
import torch

from .. import ops

# diffusion_utils.py:199-200 with logsnr in [-20, 20]; applied to fp32 values as fp32 scalars
SCHED_B = np.float32(np.arctan(np.exp(-0.5 * 20.0)))
SCHED_A = np.float32(np.arctan(np.exp(-0.5 * -20.0)) - np.arctan(np.exp(-0.5 * 20.0)))


def sampler_times(i, num_steps):
    """Integer loop index -> fp32 (u_t, u_s), bit-exact with gaussian_diffusion.py:288-290."""
    return np.float32(np.float32(i + 1.0) / np.float32(num_steps)), np.float32(np.float32(i) / np.float32(num_steps))


def logsnr_schedule_cosine_host(u):
    """diffusion_utils.py:198-201 for a host scalar, all in fp32."""
    u = np.float32(u)
    return np.float32(-2.0) * np.log(np.tan(SCHED_A * u + SCHED_B, dtype=np.float32), dtype=np.float32)


class PhiloxStream:
    """Counter-based RNG stream: (seed, running counter).  Every draw reserves its own counter range, so a run is
    reproducible from (seed) alone and ranks use disjoint seeds."""

    def __init__(self, seed):
        self.seed = int(seed) & ((1 << 64) - 1)
        self.counter = 0

    def _take(self, n):
        off = self.counter
        self.counter += (n + 3) // 4
        return off

    def normal(self, shape, device):
        n = int(np.prod(shape))
        return ops.rng_normal(tuple(shape), self.seed, self._take(n), device)

    def uniform(self, shape, device):
        n = int(np.prod(shape))
        return ops.rng_uniform(tuple(shape), self.seed, self._take(n), device)


def _unwrap(net):
    kw = {}
    while isinstance(net, partial):
        kw = {**net.keywords, **kw}
        net = net.func
    if hasattr(net, "__self__"):       # bound method (e.g. module.forward)
        net = net.__self__
    return net, kw.get("guide"), kw.get("cond_w")


class _VLoss(torch.autograd.Function):
    """loss_b = max(mse_x, mse_eps) of the clipped v-parameterised prediction, differentiable w.r.t. v."""

    @staticmethod
    def forward(ctx, v, z, x, eps, logsnr):
        loss_b, _, _, dv = ops.v_loss(v.contiguous(), z, x, eps, logsnr, grad_scale=1.0)
        ctx.save_for_backward(dv)
        return loss_b

    @staticmethod
    def backward(ctx, g):
        (dv,) = ctx.saved_tensors
        B = dv.shape[0]
        out = ops.scale_rows(dv.view(B, -1), g.contiguous().float()).view_as(dv)
        return out, None, None, None, None


class GaussianDiffusion:
    def __init__(self, *, mean_type, num_steps, teacher_net=None, teacher_mode=None, sampler="ddim", sample_cond_w=None,
                 seed=0):
        if mean_type != "v":
            raise NotImplementedError(f"HIP path implements mean_type='v' (reference default); got {mean_type!r}")
        if teacher_net is not None:
            raise NotImplementedError("progressive distillation (teacher_net) is not on the HIP path yet")
        self.mean_type = mean_type
        self.num_steps = num_steps
        self.teacher_net = None
        self.sampler = sampler
        self.sample_cond_w = sample_cond_w
        self.loss_weight_type = "snr_trunc"
        self.rng = PhiloxStream(seed)

    # ---- training ----------------------------------------------------------------------------------------
    def _draw(self, x, u, eps):
        if eps is None:
            eps = self.rng.normal(x.shape, x.device)          # :83
        if u is None:
            u = self.rng.uniform((x.shape[0],), x.device)     # :94 continuous time
        return ops.aligned(u.float()), ops.aligned(eps.float())

    def training_losses(self, *, net, x, u=None, eps=None):
        """Reference call shape (:81).  Differentiable through torch.autograd when grad mode is on."""
        assert x.dtype in [torch.float32, torch.float64]
        module, guide, cond_w = _unwrap(net)
        x = ops.aligned(x.float())
        u, eps = self._draw(x, u, eps)
        logsnr, z_t = ops.q_sample(x, eps, u)
        v = module(z_t, logsnr, guide=guide, cond_w=cond_w)
        if torch.is_grad_enabled() and v.requires_grad:
            loss = _VLoss.apply(v, z_t, x, eps, logsnr)
        else:
            loss = ops.v_loss(v, z_t, x, eps, logsnr)[0]
        return {"loss": loss}

    def train_forward_backward(self, *, net, x, grad_scale, u=None, eps=None, on_grads_ready=None):
        """Fused training pass used by DiffusionModel.train_step: forward, loss, dL/dv and the explicit backward
        schedule, leaving d(grad_scale * sum_b loss_b)/d(theta) in `module.flat_grads`.  No autograd graph."""
        module, guide, cond_w = _unwrap(net)
        x = ops.aligned(x.float())
        u, eps = self._draw(x, u, eps)
        logsnr, z_t = ops.q_sample(x, eps, u)
        ctx = {}
        v = module.forward_hip(z_t, logsnr, guide, cond_w, ctx=ctx)
        loss_b, x_mse, eps_mse, dv = ops.v_loss(v, z_t, x, eps, logsnr, grad_scale=grad_scale)
        module.backward_hip(ctx, dv, on_grads_ready=on_grads_ready)
        return {"loss": loss_b, "x_mse": x_mse, "eps_mse": eps_mse, "logsnr": logsnr}

    # ---- sampling ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, *, net, init_x, cond_w=None, record=True, noises=None, net_cond_w=None):
        """:245-296.  Returns (all_zs, all_xs, all_eps) stacked [T, B, C, H, W] like the reference when `record`;
        with record=False only the final z is produced and returned as a 1-tuple-compatible triple
        (z[None], None, None) — the 3*T*B*C*H*W*4-byte trajectory is the dominant cost at large T*B otherwise.
        `noises` ([T, ...], indexed by step i) / `net_cond_w` inject the random draws (tests)."""
        module, guide, _ = _unwrap(net)
        B = init_x.shape[0]
        dev = init_x.device
        if cond_w is not None:
            if net_cond_w is None:
                net_cond_w = 4.0 * self.rng.uniform((B,), dev)       # :247-251
            if isinstance(self.sample_cond_w, torch.Tensor):
                w = self.sample_cond_w.to(dev).float().expand(B).contiguous()
            elif self.sample_cond_w is not None and float(self.sample_cond_w) != -1.0:
                w = torch.full((B,), float(self.sample_cond_w), device=dev)
            else:
                w = ops.aligned(net_cond_w.float())                   # :257
        else:
            w = None
        if self.sampler not in ("ddim", "noisy"):
            raise NotImplementedError(self.sampler)
        if w is not None and guide is None:
            raise ValueError("classifier-free guidance needs class labels (net must carry guide=)")
        z_t = ops.aligned(init_x.float())
        zs, xs, es = [], [], []
        if w is not None:       # conditional + unconditional evaluations share one 2B-image forward (:176-177)
            guide2 = torch.cat([guide, -torch.ones_like(guide)])
        for i in range(self.num_steps)[::-1]:
            u_t, u_s = sampler_times(i, self.num_steps)
            lt, ls = logsnr_schedule_cosine_host(u_t), logsnr_schedule_cosine_host(u_s)
            if w is None:
                lvec = torch.full((B,), float(lt), device=dev)
                v = module.forward_hip(z_t, lvec, guide, None)
                vu = None
            else:
                lvec = torch.full((2 * B,), float(lt), device=dev)
                v2 = module.forward_hip(torch.cat([z_t, z_t]), lvec, guide2, None)
                v, vu = v2[:B], v2[B:]
            noise = None
            if self.sampler == "noisy":
                noise = ops.aligned(noises[i]) if noises is not None else self.rng.normal(z_t.shape, dev)   # :241
            z_t, xp, ep = ops.sampler_step(v, z_t, lt, ls, i == 0, v_uncond=vu, cond_w=w, noise=noise, want_pred=record)
            if record:
                zs.append(z_t); xs.append(xp); es.append(ep)
        if record:
            return torch.stack(zs), torch.stack(xs), torch.stack(es)
        return z_t[None], None, None
