"""Fused Adam over the flat parameter arena (one kernel launch per step) — replaces the per-tensor
torch.optim.Adam(self.net.parameters(), lr=G.lr) of reference gms/diffusion/diffusion_model.py:56,71.
Same update rule and defaults (betas 0.9/0.999, eps 1e-8, no weight decay, bias-corrected); `grad_scale` folds in
the 1/world factor of data-parallel training.  Parameters without a gradient keep a zero gradient slice, which
leaves them unchanged exactly as torch's skip of `grad is None` does."""
import torch

from .. import ops


class FusedAdam:
    def __init__(self, net, lr=3e-4, betas=(0.9, 0.999), eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, float(lr), betas, float(eps)
        self.step_count = 0
        self.m = self.v = None

    def zero_grad(self):
        self.net.flat_grads.zero_()

    def step(self, grad_scale=1.0):
        p = self.net.flat_params
        if self.m is None or self.m.device != p.device:
            self.m = torch.zeros_like(p)
            self.v = torch.zeros_like(p)
        self.step_count += 1
        ops.adam_step(p, self.net.flat_grads, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.step_count, grad_scale)
        self.net.mark_params_changed()

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count, self.m, self.v, self.lr = sd["step"], sd["m"], sd["v"], sd["lr"]
