"""ORACLE (test infrastructure) — CPU restatement of the reference diffusion process.

Follows reference gms/diffusion/gaussian_diffusion.py and gms/diffusion/diffusion_utils.py
(line cites on each function).  RNG is injected (u, eps, sampler noise, guidance weights are
arguments) so the HIP path and this restatement can be driven by identical numbers.
"""
import math

import numpy as np
import torch

from . import unet_ref

# diffusion_utils.py:198-201 with logsnr_min=-20, logsnr_max=20 (gaussian_diffusion.py:33-35):
# float64 numpy scalars; torch applies them to fp32 tensors as fp32 scalars.
SCHED_B = float(np.arctan(np.exp(-0.5 * 20.0)))            # 4.539992973129278e-05
SCHED_A = float(np.arctan(np.exp(-0.5 * -20.0))) - SCHED_B  # 1.5707055269354342


def logsnr_schedule_cosine(t):
    """diffusion_utils.py:198-201: -2*log(tan(a*t+b)), evaluated in t's dtype (fp32)."""
    return -2.0 * torch.log(torch.tan(SCHED_A * t + SCHED_B))


def bcast(v, shape):
    """diffusion_utils.py:126-130 broadcast_from_left."""
    return torch.broadcast_to(v.reshape(v.shape + (1,) * (len(shape) - v.ndim)), shape)


def sampler_times(i, num_steps):
    """gaussian_diffusion.py:288-290 — integer loop index -> fp32 (u_t, u_s).

    torch: int64 0-dim tensor; `(i + 1.0) / T` and `i / T` are true divisions producing fp32
    (default dtype).  Bit-exact restatement in numpy fp32."""
    u_t = np.float32(np.float32(i + 1.0) / np.float32(num_steps))
    u_s = np.float32(np.float32(i) / np.float32(num_steps))
    return u_t, u_s


def q_sample(x, logsnr, eps):
    """gaussian_diffusion.py:99-100 + diffusion_utils.py:65-73."""
    l = bcast(logsnr, x.shape)
    return x * torch.sqrt(torch.sigmoid(l)) + torch.sqrt(torch.sigmoid(-l)) * eps


def predict_x_from_eps(z, eps, logsnr):      # diffusion_utils.py:76-82
    l = bcast(logsnr, z.shape)
    return torch.sqrt(1.0 + torch.exp(-l)) * (z - eps * torch.rsqrt(1.0 + torch.exp(l)))


def predict_eps_from_x(z, x, logsnr):        # diffusion_utils.py:85-91
    l = bcast(logsnr, z.shape)
    return torch.sqrt(1.0 + torch.exp(l)) * (z - x * torch.rsqrt(1.0 + torch.exp(-l)))


def predict_v_from_x_and_eps(x, eps, logsnr):  # diffusion_utils.py:94-98
    l = bcast(logsnr, x.shape)
    return torch.sqrt(torch.sigmoid(l)) * eps - torch.sqrt(torch.sigmoid(-l)) * x


def predict_x_from_v(z, v, logsnr):          # diffusion_utils.py:101-105
    l = bcast(logsnr, z.shape)
    return torch.sqrt(torch.sigmoid(l)) * z - torch.sqrt(torch.sigmoid(-l)) * v


def model_outputs(model_output, z, logsnr, mean_type="v"):
    """gaussian_diffusion.py:56-79 after the net call: convert to x, clip, recompute eps and v."""
    if mean_type == "eps":
        model_x = predict_x_from_eps(z, model_output, logsnr)
    elif mean_type == "x":
        model_x = model_output
    elif mean_type == "v":
        model_x = predict_x_from_v(z, model_output, logsnr)
    else:  # 'both' is broken in the reference for a 1-channel net — excluded
        raise NotImplementedError(mean_type)
    model_x = torch.clip(model_x, -1.0, 1.0)
    model_eps = predict_eps_from_x(z, model_x, logsnr)
    model_v = predict_v_from_x_and_eps(model_x, model_eps, logsnr)
    return {"model_x": model_x, "model_eps": model_eps, "model_v": model_v}


def run_model(params, z, logsnr, guide=None, cond_w=None, mean_type="v"):
    """gaussian_diffusion.py:45-79."""
    out = unet_ref.unet_forward(params, z, logsnr, guide=guide, cond_w=cond_w)
    return model_outputs(out, z, logsnr, mean_type)


def training_losses(params, x, y, u, eps, mean_type="v"):
    """gaussian_diffusion.py:81-172, non-teacher branch, with (u, eps) injected.

    Returns dict(loss[B], logsnr[B], z_t, model_x, model_eps)."""
    assert x.dtype in (torch.float32, torch.float64)
    logsnr = logsnr_schedule_cosine(u)
    z_t = q_sample(x, logsnr, eps)
    out = run_model(params, z_t, logsnr, guide=y, mean_type=mean_type)
    x_mse = torch.square(out["model_x"] - x).flatten(1).mean(1)       # mean_flat, diffusion_utils.py:133
    eps_mse = torch.square(out["model_eps"] - eps).flatten(1).mean(1)
    loss = torch.maximum(x_mse, eps_mse)                               # 'snr_trunc', :168-169
    return {"loss": loss, "logsnr": logsnr, "z_t": z_t, "x_mse": x_mse, "eps_mse": eps_mse, **out}


def cf_guidance(params, z_t, eps_pred_t, logsnr_t, cond_w, guide, mean_type="v"):
    """gaussian_diffusion.py:174-187."""
    uncond = run_model(params, z_t, logsnr_t, guide=-torch.ones_like(guide), mean_type=mean_type)
    w = bcast(cond_w, z_t.shape)
    eps = (1 + w) * eps_pred_t + (-w) * uncond["model_eps"]
    x = torch.clip(predict_x_from_eps(z_t, eps, logsnr_t), -1.0, 1.0)
    eps = predict_eps_from_x(z_t, x, logsnr_t)
    return x, eps


def ddim_step(params, logsnr_t, logsnr_s, z_t, guide=None, cond_w=None, mean_type="v"):
    """gaussian_diffusion.py:189-213 (logsnr_t / logsnr_s are 0-dim fp32 tensors)."""
    B = z_t.shape[0]
    lt = torch.broadcast_to(logsnr_t.reshape(()), (B,))
    out = run_model(params, z_t, lt, guide=guide, mean_type=mean_type)
    x_pred, eps_pred = out["model_x"], out["model_eps"]
    if cond_w is not None:
        x_pred, eps_pred = cf_guidance(params, z_t, eps_pred, lt, cond_w, guide, mean_type)
    stdv_s = torch.sqrt(torch.sigmoid(-logsnr_s))
    alpha_s = torch.sqrt(torch.sigmoid(logsnr_s))
    z_s = alpha_s * x_pred + stdv_s * eps_pred
    return z_s, x_pred, eps_pred


def reverse_dpm_step(params, logsnr_t, logsnr_s, z_t, noise, guide=None, cond_w=None, mean_type="v"):
    """gaussian_diffusion.py:215-243 + diffusion_utils.py:34-62 (x_logvar='large'), noise injected.

    `log1mexp` (diffusion_utils.py:108-123) only feeds the unused 'logvar' entry — omitted."""
    B = z_t.shape[0]
    lt = torch.broadcast_to(logsnr_t.reshape(()), (B,))
    out = run_model(params, z_t, lt, guide=guide, mean_type=mean_type)
    x_pred, eps_pred = out["model_x"], out["model_eps"]
    if cond_w is not None:
        x_pred, eps_pred = cf_guidance(params, z_t, eps_pred, lt, cond_w, guide, mean_type)
    alpha_st = torch.sqrt((1.0 + torch.exp(-logsnr_t)) / (1.0 + torch.exp(-logsnr_s)))
    alpha_s = torch.sqrt(torch.sigmoid(logsnr_s))
    r = torch.exp(logsnr_t - logsnr_s)
    one_minus_r = -torch.expm1(logsnr_t - logsnr_s)
    mean = r * alpha_st * z_t + one_minus_r * alpha_s * x_pred
    var = one_minus_r * torch.sigmoid(-logsnr_t)
    z_s = mean + torch.sqrt(var) * noise
    return z_s, x_pred, eps_pred


def sample(params, init_x, guide, num_steps, sampler="ddim", cond_w=None, noises=None,
           mean_type="v", record=True):
    """gaussian_diffusion.py:245-296.  `cond_w` is the resolved per-sample weight tensor or None
    (the policy of :247-257 lives with the caller); `noises[i]` feeds the ancestral sampler."""
    z_t = init_x
    zs, xs, es = [], [], []
    for i in range(num_steps)[::-1]:
        u_t, u_s = sampler_times(i, num_steps)
        logsnr_t = logsnr_schedule_cosine(torch.tensor(u_t))
        logsnr_s = logsnr_schedule_cosine(torch.tensor(u_s))
        if sampler == "ddim":
            z_s, x_pred, eps_pred = ddim_step(params, logsnr_t, logsnr_s, z_t, guide, cond_w, mean_type)
        elif sampler == "noisy":
            z_s, x_pred, eps_pred = reverse_dpm_step(params, logsnr_t, logsnr_s, z_t, noises[i], guide,
                                                     cond_w, mean_type)
        else:
            raise NotImplementedError(sampler)
        z_t = x_pred if i == 0 else z_s        # :292 where(i == 0, x_pred, z_s)
        if record:
            zs.append(z_t); xs.append(x_pred); es.append(eps_pred)
    if record:
        return torch.stack(zs), torch.stack(xs), torch.stack(es)
    return z_t


def ddim_step_vec(params, logsnr_t, logsnr_s, z_t, guide=None, net_cond_w=None, cf_w=None, mean_type="v"):
    """gaussian_diffusion.py:189-213 with per-sample [B] times (the teacher steps inside the distillation loss).
    net_cond_w conditions the net on w (cond_w_embed); cf_w applies classifier-free guidance with weight w."""
    out = model_outputs(unet_ref.unet_forward(params, z_t, logsnr_t, guide=guide, cond_w=net_cond_w), z_t, logsnr_t, mean_type)
    x_pred, eps_pred = out["model_x"], out["model_eps"]
    if cf_w is not None:
        unc = model_outputs(unet_ref.unet_forward(params, z_t, logsnr_t, guide=-torch.ones_like(guide), cond_w=net_cond_w),
                            z_t, logsnr_t, mean_type)
        w = bcast(cf_w, z_t.shape)
        eps = (1 + w) * eps_pred + (-w) * unc["model_eps"]
        x_pred = torch.clip(predict_x_from_eps(z_t, eps, logsnr_t), -1.0, 1.0)
        eps_pred = predict_eps_from_x(z_t, x_pred, logsnr_t)
    ls = bcast(logsnr_s, z_t.shape)
    z_s = torch.sqrt(torch.sigmoid(ls)) * x_pred + torch.sqrt(torch.sigmoid(-ls)) * eps_pred
    return z_s, x_pred, eps_pred


def distill_losses(student, teacher, x, y, eps, cond_w, num_steps, mode, u=None, i_times=None):
    """gaussian_diffusion.py:81-172, teacher branches (:87-91,:105-154), draws injected.
    mode 'step1': u given; 'step2': integer i_times given (u = (i+1)/T)."""
    if mode == "step2":
        u = (i_times + 1).to(x.dtype) / num_steps
    logsnr = logsnr_schedule_cosine(u)
    z_t = q_sample(x, logsnr, eps)
    u_s = u - 1.0 / num_steps
    logsnr_s = logsnr_schedule_cosine(u_s)
    with torch.no_grad():
        if mode == "step1":
            _, x_target, eps_target = ddim_step_vec(teacher, logsnr, logsnr_s, z_t, guide=y, cf_w=cond_w)
        else:
            logsnr_mid = logsnr_schedule_cosine(u - 0.5 / num_steps)
            z_mid, _, _ = ddim_step_vec(teacher, logsnr, logsnr_mid, z_t, guide=y, net_cond_w=cond_w)
            z_teacher, x_pred_teacher, _ = ddim_step_vec(teacher, logsnr_mid, logsnr_s, z_mid, guide=y, net_cond_w=cond_w)
            alpha_s = bcast(torch.sqrt(torch.sigmoid(logsnr_s)), x.shape)
            alpha_t = bcast(torch.sqrt(torch.sigmoid(logsnr)), x.shape)
            frac = bcast(torch.exp(0.5 * (torch.nn.functional.softplus(logsnr) - torch.nn.functional.softplus(logsnr_s))), x.shape)
            x_target = (z_teacher - frac * z_t) / (alpha_s - frac * alpha_t)
            x_target = torch.where(bcast(i_times == 0, x.shape), x_pred_teacher, x_target)
            eps_target = predict_eps_from_x(z_t, x_target, logsnr)
    out = run_model(student, z_t, logsnr, guide=y, cond_w=cond_w)
    x_mse = torch.square(out["model_x"] - x_target).flatten(1).mean(1)
    eps_mse = torch.square(out["model_eps"] - eps_target).flatten(1).mean(1)
    loss = eps_mse if mode == "step1" else torch.maximum(x_mse, eps_mse)
    return {"loss": loss, "x_target": x_target, "eps_target": eps_target, "logsnr": logsnr, "z_t": z_t}


def adam_step(p, g, m, v, step, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as diffusion_model.py:56 uses them (no weight decay, no amsgrad).
    `step` is the 1-based step count AFTER incrementing.  Returns (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
