"""`DiffusionModel` plugin — drop-in for reference gms/diffusion/diffusion_model.py:14-111 on the HIP path.

Same class name (registry key `diffusion_model`, alias `diffusion`), same `DG` keys and defaults (:15-29), same
method surface the driver calls: train_step(x, y) -> {'loss', 'loss_scale'}, loss(x, y) -> (loss, metrics),
sample(n, y) -> [n, C, S, S] in [-1, 1], evaluate(writer, x, y, epoch), save (inherited).  State-dict keys are
`net.<reference names>`.

How it differs underneath: 16-bit MFMA compute with fp32 master weights - forward activations and forward weight packs in fp16 (the
precision of the reference's fp16 autocast forward), gradients in bf16 - instead of fp16 autocast + GradScaler (bf16 gradients need no
loss scaling; `loss_scale` is reported as 1.0 to keep the metric key); one fused pass
(forward -> loss -> explicit backward -> bucketed RCCL all-reduce -> fused Adam) instead of autograd + per-tensor
optimiser; RNG from counter-based Philox streams on the device.  `make_plugin(base)` builds the same class on top of
the REFERENCE's `gms.common.GM`, which is what a `gms/` checkout needs for `discover_models()` to pick it up
(INTEGRATION.md).
"""
import random
from functools import partial
from pathlib import Path

import os

import torch

from .. import common, ops, parallel
from .gaussian_diffusion import GaussianDiffusion, PhiloxStream
from .optim import FusedAdam
from .simple_unet import SimpleUnet

_DTYPES = {"bf16": torch.bfloat16, "fp32": torch.float32}


def make_plugin(GMBase, AttrDict):
    class DiffusionModel(GMBase):
        DG = AttrDict()  # default G  (diffusion_model.py:15-29)
        DG.binarize = 0
        DG.timesteps = 250
        DG.hidden_size = 128
        DG.dropout = 0.0
        DG.sampler = "ddim"
        DG.mean_type = "v"
        DG.eval_heavy = 1
        DG.class_cond = 1
        DG.sample_cond_w = -1.0
        DG.cf_drop_prob = 0.1
        DG.teacher_path = Path(".")
        DG.teacher_mode = "step1"
        DG.lr_scheduler = "none"
        # additions of the HIP path
        DG.compute_dtype = "bf16"      # 'bf16' (16-bit MFMA mode: bf16 gradients) or 'fp32' (exact-fp32 MFMA, 1e-3 parity mode)
        DG.act_dtype = os.environ.get("GMK_ACT_DTYPE", "fp16")      # 16-bit mode only: storage of forward activations / forward weight packs, 'fp16' (the precision of the
                                       # reference's fp16 autocast forward, diffusion_model.py:68) or 'bf16' (the all-bf16 path, A/B)
        DG.in_channels = 1             # reference: 1 (simple_unet.py:93,41)
        DG.attention = 0               # 1: self-attention block behind `turn` (north_star / BASELINE config 5; not in the reference); 2: the same
                                       # with QK^T / PV on the fp8 matrix cores
        DG.seed = 0

        def __init__(self, G):
            super().__init__(G)
            get = lambda k: G[k] if k in G else self.DG[k]
            cdt = _DTYPES[get("compute_dtype")]
            adt = cdt if cdt == torch.float32 else {"fp16": torch.float16, "bf16": torch.bfloat16}[get("act_dtype")]
            self.net = SimpleUnet(get("hidden_size"), get("dropout"), in_channels=get("in_channels"),
                                  compute_dtype=cdt, attention=int(get("attention")), act_dtype=adt)
            weights_from = Path(G["weights_from"]) if "weights_from" in G else Path(".")
            if Path(get("teacher_path")) != Path(".") and weights_from == Path("."):      # diffusion_model.py:34-43
                print("Loading teacher model")
                self.load_state_dict(torch.load(get("teacher_path"), map_location="cpu"), strict=False)
                self.teacher_net = SimpleUnet(get("hidden_size"), get("dropout"), in_channels=get("in_channels"),
                                              compute_dtype=cdt, attention=int(get("attention")), act_dtype=adt)
                self.teacher_net.load_state_dict(self.net.state_dict())
                self.teacher_net.eval()
                for param in self.teacher_net.parameters():
                    param.requires_grad = False
            else:
                self.teacher_net = None
            seed = int(get("seed")) * 1000 + parallel.rank()
            self.diffusion = GaussianDiffusion(mean_type=get("mean_type"), num_steps=int(get("timesteps")),
                                               sampler=get("sampler"), teacher_net=self.teacher_net,
                                               teacher_mode=get("teacher_mode"), sample_cond_w=get("sample_cond_w"),
                                               seed=seed)
            self.net.drop_seed = seed + 104729              # per-rank dropout masks (only used when dropout > 0)
            self.optimizer = FusedAdam(self.net, lr=G.lr if "lr" in G else 3e-4)
            self.size = 32 if ("pad32" in G and G.pad32) else 28
            self._aux_rng = PhiloxStream(seed + 7919)
            self._sync = None

        # -- training (diffusion_model.py:63-74)
        def train_step(self, x, y):
            B = x.shape[0]
            # classifier-free label drop (:67): mutates the caller's y in place like the reference, but the mask comes from the
            # device RNG inside one small kernel (the reference's CPU-generated mask forces a host sync every step)
            p_drop = float(self.G.cf_drop_prob if "cf_drop_prob" in self.G else self.DG.cf_drop_prob)
            if y.dtype == torch.int64 and y.is_contiguous() and y.data_ptr() % 16 == 0:
                ops.label_drop(y, p_drop, self._aux_rng.seed, self._aux_rng._take(B))
            else:                                   # odd label tensors (a slice, int32): same draw, torch does the masking
                y.masked_fill_(self._aux_rng.uniform((B,), x.device) < p_drop, -1)
            if self._sync is None:
                self._sync = parallel.GradSync(self.net)
            world = parallel.world()
            if self._graphable(x, world):
                return self._train_step_graphed(x, y)
            try:
                out = self.diffusion.train_forward_backward(net=partial(self.net, guide=y), x=x, grad_scale=1.0 / B,
                                                            on_grads_ready=self._sync.hook, join_side_before_ready=False)
            except BaseException:
                self._sync.abort()                  # a raise between hook() and finish() must not leave the process on the carved CU limit
                raise
            self._sync.finish()
            self.optimizer.step(grad_scale=1.0 / world)
            metrics = {"loss": ops.mean(out["loss"])}
            ops.throttle()                          # at most two steps queued on the GPU (see ops.throttle)
            metrics["loss_scale"] = torch.tensor(1.0)
            return metrics

        # -- small batches: the step as a replayed HIP graph ------------------------------------------------------------------
        # The reference's default invocation trains at bs = 32 (BASELINE configs[0]).  At that size a step is ~ 350 kernel launches whose host
        # side (ctypes + torch allocations, 4.6 - 5.0 ms) exceeds the GPU's work (3.5 ms): forward, loss, backward and the loss mean are captured once
        # per input shape and replayed; the random draws (same Philox streams, same order), the label drop on the CALLER's y, the copies into
        # the static inputs and the fused Adam (its step count changes every step) stay outside.  Same kernels, same arguments: the
        # parameters after k steps equal the kernel-by-kernel path's bit for bit (tests/test_gpu_unet.py).  The weight gradients run on the main
        # stream inside the capture (the side stream pays at sizes that fill the chip).  Measured, same box: bs = 32 5.00 -> 3.46 ms per step,
        # bs = 64 4.69 -> 4.14, bs = 128 5.11 -> 5.23 (hence the 64 Ki-pixel limit).  GMK_TRAIN_GRAPH_PIXELS=0 turns it off.
        TRAIN_GRAPH_MAX_PIXELS = int(os.environ.get("GMK_TRAIN_GRAPH_PIXELS", str(64 * 1024)))

        def _graphable(self, x, world):
            return (world == 1 and not parallel.exchanging() and self.teacher_net is None and self.net.dropout == 0.0 and x.is_cuda and x.dim() == 4 and
                    0 < x.shape[0] * x.shape[2] * x.shape[3] <= self.TRAIN_GRAPH_MAX_PIXELS and ops.PROFILE is None)

        def _train_step_graphed(self, x, y):
            B, dev = x.shape[0], x.device
            rng = self.diffusion.rng
            eps = rng.normal(x.shape, dev)                      # the draw order of GaussianDiffusion._prepare: eps, then u
            u = rng.uniform((B,), dev)
            key = (tuple(x.shape), y.dtype)
            graphs = self.__dict__.setdefault("_train_graphs", {})
            ent = graphs.get(key)
            if ent is None:
                xs, ys, es, us = x.float().clone(), y.clone(), eps.clone(), u.clone()
                side_was, side_stream = ops.WGRAD_STREAM, self.net._side
                # nothing outside the capturing stream may be joined from inside the capture; the weight gradients stay on the main stream
                # (GMK_TRAIN_GRAPH_SIDE=1 gives them a side stream of the capture's own: measured slower at these sizes, 3.68 vs 3.46 ms at bs = 32)
                ops.WGRAD_STREAM, self.net._side = side_was and os.environ.get("GMK_TRAIN_GRAPH_SIDE", "0") == "1", None
                try:
                    run = lambda: self.diffusion.train_forward_backward(net=partial(self.net, guide=ys), x=xs, grad_scale=1.0 / B, u=us, eps=es,
                                                                        on_grads_ready=self._sync.hook, join_side_before_ready=False)
                    warm = torch.cuda.Stream(device=dev)        # warm-up off the capture: packs, workspaces, allocator pools (gradients only)
                    warm.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(warm):
                        run()
                    torch.cuda.current_stream().wait_stream(warm)
                    self.net.mark_params_changed()              # the captured forward has to contain the weight re-pack every step needs
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        loss = ops.mean(run()["loss"])
                finally:
                    ops.WGRAD_STREAM, self.net._side = side_was, side_stream
                ent = graphs[key] = (graph, xs, ys, es, us, loss)
                if len(graphs) > 4:
                    graphs.pop(next(iter(graphs)))
            graph, xs, ys, es, us, loss = ent
            xs.copy_(x); ys.copy_(y); es.copy_(eps); us.copy_(u)
            graph.replay()
            self.optimizer.step(grad_scale=1.0)
            metrics = {"loss": loss.clone(), "loss_scale": torch.tensor(1.0)}
            ops.throttle()
            return metrics

        # -- loss (:76-80): differentiable through torch.autograd; used by the driver's test-set pass
        def loss(self, x, y):
            metrics = self.diffusion.training_losses(net=partial(self.net, guide=y), x=x)
            metrics = {key: val.mean() for key, val in metrics.items()}
            return metrics["loss"], metrics

        # -- sampling (:82-87)
        def sample(self, n, y=None):
            with torch.no_grad():
                dev = self.net.flat_params.device
                noise = self._aux_rng.normal((n, self.net.in_channels, self.size, self.size), dev)
                net = partial(self.net, guide=y)
                cond_w = 0.5 if y is not None else None
                return self.diffusion.sample(net=net, init_x=noise, cond_w=cond_w, record=False)[0][-1]

        # -- evaluate (:89-111): 25 class-conditional samples without guidance, trajectories as uint8
        def evaluate(self, writer, x, y, epoch):
            def proc(t):
                t = ((t + 1) * 127.5).clamp(0, 255).to(torch.uint8).cpu()
                if "pad32" in self.G and self.G.pad32:
                    t = t[..., 2:-2, 2:-2]
                return t

            stream = PhiloxStream(0)                                  # :99 torch.manual_seed(0)
            noise = stream.normal((25, self.net.in_channels, self.size, self.size), x.device)
            labels = torch.arange(25, dtype=torch.long, device=x.device) % 10   # :101
            zs, xs, eps = self.diffusion.sample(net=partial(self.net, guide=labels), init_x=noise)
            zs, xs, eps = proc(zs), proc(xs), proc(eps)
            self.last_eval = {"samples": zs[-1], "sampling_process": zs, "eps": eps, "x": xs}
            if writer is not None and self.net.in_channels == 1:        # :105-110, same tags
                common.write_grid(writer, "samples", zs[-1], epoch)
                common.write_gridvid(writer, "sampling_process", zs, epoch)
                common.write_gridvid(writer, "diffusion_model/eps", eps, epoch)
                common.write_gridvid(writer, "diffusion_model/x", xs, epoch)
            random.randint(0, 2 ** 32)                                # :111 keeps the host RNG consumption

    return DiffusionModel


DiffusionModel = make_plugin(common.GM, common.AttrDict)
