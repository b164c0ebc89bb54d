"""Host-side mirror of reference gms/diffusion/gaussian_diffusion.py (`GaussianDiffusion`, :19-296) over the HIP kernels.

Same constructor keywords, same `training_losses(net=, x=)` / `sample(net=, init_x=, cond_w=)` call shapes and
return structure; `net` is the HIP `SimpleUnet` or a `functools.partial` of it carrying `guide=` / `cond_w=` exactly
as the reference passes it (diffusion_model.py:77,85).  All arithmetic is in libgmk.so:

* q_sample + log-SNR schedule          gmk_q_sample     (:94-100, diffusion_utils.py:65-73,198-201)
* v -> x_hat/eps_hat, clip, loss, dL/dv gmk_v_loss       (:61-77,:165-169 and their backward)
* DDIM / ancestral / guidance update   gmk_sampler_step (:174-243,:292)
* RNG                                   counter-based Philox streams (gmk_rng_*), keyed (seed, rank, draw index)

`mean_type` 'v' (the reference default, diffusion_model.py:21), 'eps' and 'x' (:58-63) are kernel arguments; 'both'
(:64-69) splits a 1-channel output along W in the reference and cannot run there — it raises here too.  Progressive
distillation (:87-91,:105-154) is supported: `teacher_net` is a frozen HIP `SimpleUnet`; teacher DDIM
steps run through gmk_ddim_step_vec / gmk_distill_target, the student is conditioned on the guidance weight.
"""
from functools import partial

import os

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


def resolve_guidance(*, sample_cond_w, net_cond_w, kw_cond_w, has_teacher, sampler):
    """Guidance-weight policy of `GaussianDiffusion.sample` (gaussian_diffusion.py:246-257,271-278) as a pure function.
    net_cond_w: the per-sample draw 4*U[0,1) made when the caller passed `cond_w` (else None); kw_cond_w: a `cond_w=` keyword the
    caller's partial already carried.  -> (student_w, guidance_w, use_teacher):
      student_w    conditions the network through `cond_w_embed`
      guidance_w   classifier-free guidance weight (None: one unguided forward per step)
      use_teacher  evaluate the frozen teacher instead of `net` (sampler='teacher_test')
    No teacher: guidance_w = sample_cond_w unless that is -1.0 (then the draw, which is None when the caller did not ask) —
    so `--sample_cond_w 2.0` also guides `evaluate()`, which passes no cond_w (:257).  With a teacher the student is
    conditioned on the draw and never guided (:252-255); 'teacher_test' runs the teacher, guided by the weight the student
    would have been conditioned on (:271-278: `net.keywords['cond_w']`), unguided when there is none."""
    fixed = sample_cond_w is not None and float(sample_cond_w) != -1.0
    if has_teacher:
        student_w, guidance_w = net_cond_w, None
    else:
        student_w, guidance_w = kw_cond_w, (sample_cond_w if fixed else net_cond_w)
    if sampler == "teacher_test":
        if not has_teacher:
            raise ValueError("sampler='teacher_test' needs a teacher_net")
        return None, student_w, True
    return student_w, guidance_w, False


class _VLoss(torch.autograd.Function):
    """loss_b = max(mse_x, mse_eps) of the clipped v-parameterised prediction, differentiable w.r.t. v."""

    @staticmethod
    def forward(ctx, v, z, x, eps, logsnr, loss_type=0, mean_type="v"):
        loss_b, _, _, dv = ops.v_loss(v.contiguous(), z, x, eps, logsnr, grad_scale=1.0, loss_type=loss_type, mean_type=mean_type)
        ctx.save_for_backward(dv)
        return loss_b

    @staticmethod
    def backward(ctx, g):
        (dv,) = ctx.saved_tensors
        B = dv.shape[0]
        out = ops.scale_rows(dv.view(B, -1), g.contiguous().float()).view_as(dv)
        return out, None, None, None, None, None, None


class GaussianDiffusion:
    def __init__(self, *, mean_type, num_steps, teacher_net=None, teacher_mode=None, sampler="ddim", sample_cond_w=None,
                 seed=0):
        if mean_type not in ops.MEAN_TYPES:                     # :70-71
            raise NotImplementedError(mean_type)
        self.mean_type = mean_type
        self.num_steps = num_steps
        self.teacher_net = teacher_net
        self.sampler = sampler
        self.sample_cond_w = sample_cond_w
        self.loss_weight_type = "snr_trunc"
        if self.teacher_net is not None:                      # :39-43
            assert teacher_mode in ["step1", "step2"]
            self.teacher_mode = teacher_mode
            if self.teacher_mode == "step1":
                self.loss_weight_type = "snr"
        self.rng = PhiloxStream(seed)
        self._graphs = {}                   # small-batch sampling: one captured U-Net forward per (net, shape, conditioning kind)

    # ---- small-batch sampling: the forward as a replayed HIP graph ------------------------------------------------------------
    # The reference's own use of the sampler is small: `evaluate` draws 25 images (diffusion_model.py:98-104), `eval_heavy` the test batch
    # size.  Since round 4's dispatch changes a 25-image forward is 0.93 ms of kernels behind 1.1 ms of host launch work (~ 85 launches through
    # ctypes): the host is the bound, and replaying the captured forward removes it.  (Round 3 had measured a graph at 1.34 = 1.34 ms and
    # dropped it - the kernels, then 30 % longer, were the bound.)  Same kernels in the same order: bit-identical images.
    GRAPH_MAX_PIXELS = int(os.environ.get("GMK_SAMPLER_GRAPH_PIXELS", str(64 * 1024)))       # images x H x W of one forward; 0 turns it off

    def _graph_path(self, module, nb, H, W):
        """Does a forward of `nb` images of H x W replay a captured graph in the sampler loop?"""
        return 0 < nb * H * W <= self.GRAPH_MAX_PIXELS and self.num_steps >= 16 and not (module.training and module.dropout > 0.0)

    def _forward_runner(self, module, z, guide, student_w):
        """-> (run(z_t) -> v, lvec): the forward of `module` on a [nb, C, H, W] batch conditioned on `guide` / `student_w`, and the fp32 [nb]
        log-SNR vector it reads (the caller fills it before the first step, the sampler-update kernel writes it afterwards)."""
        nb = z.shape[0]
        dev = z.device
        # (a captured forward would replay ONE dropout mask: training-mode dropout keeps the kernel-by-kernel path)
        small = self._graph_path(module, nb, z.shape[2], z.shape[3])
        if not small:
            lvecs = [torch.empty((nb,), device=dev), torch.empty((nb,), device=dev)]      # two buffers alternate (see `sample`)
            return None, lvecs
        # the host-side freshness check of the packed convolution weights runs at capture time only: do it here, before every replay loop
        # (the pack buffer keeps its address, so a re-pack is seen by the captured kernels)
        module.prepare_forward(dev)
        key = (id(module), module.flat_params.data_ptr(), module._pack_buf.data_ptr(), tuple(z.shape), guide is not None, student_w is not None)
        ent = self._graphs.get(key)
        if ent is None:
            zs, ls = torch.empty_like(z), torch.zeros((nb,), device=dev)
            gs = guide.clone() if guide is not None else None
            ws = student_w.clone() if student_w is not None else None
            zs.copy_(z)
            side = torch.cuda.Stream(device=dev)                  # warm-up off the capture: weight packs, workspaces, allocator pools
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    module.forward_hip(zs, ls, gs, ws)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = module.forward_hip(zs, ls, gs, ws)
            ent = self._graphs[key] = (graph, zs, ls, gs, ws, out)
            if len(self._graphs) > 8:                             # a handful of shapes at most: drop the oldest capture
                self._graphs.pop(next(iter(self._graphs)))
        graph, zs, ls, gs, ws, out = ent
        if gs is not None:
            gs.copy_(guide)
        if ws is not None:
            ws.copy_(student_w)

        def run(z_t):
            zs.copy_(z_t)
            graph.replay()
            return out
        return run, [ls, ls]

    # ---- training ----------------------------------------------------------------------------------------
    def _teacher_eval(self, z, logsnr, guide, cond_w_net, guided):
        """Teacher forward (no grad).  guided: also the unconditional evaluation, batched as one 2B forward (:176-177)."""
        t = self.teacher_net
        if not guided:
            return t.forward_hip(z, logsnr, guide, cond_w_net), None
        B = z.shape[0]
        v2 = t.forward_hip(torch.cat([z, z]), torch.cat([logsnr, logsnr]), torch.cat([guide, -torch.ones_like(guide)]), None)
        return v2[:B], v2[B:]

    def _prepare(self, net, x, u, eps, i_times=None, cond_w=None):
        """Everything ahead of the student's forward pass: draws, q_sample and (distillation) the teacher's targets.
        -> (module, guide, student cond_w, z_t, logsnr, x_target, eps_target, loss_type)"""
        module, guide, _ = _unwrap(net)
        x = ops.aligned(x.float())
        B, dev = x.shape[0], x.device
        distill = self.teacher_net is not None
        if eps is None:
            eps = self.rng.normal(x.shape, dev)                                   # :83
        eps = ops.aligned(eps.float())
        if distill and self.teacher_mode == "step2":                              # :87-91 discrete time
            if i_times is None:
                i_times = (self.rng.uniform((B,), dev) * self.num_steps).long().clamp_(max=self.num_steps - 1)
            i_times = ops.aligned(i_times.long())
            _, u = ops.logsnr_schedule(B, dev, i_times=i_times, num_steps=self.num_steps, want_u=True)
        else:
            if u is None:
                u = self.rng.uniform((B,), dev)                                   # :94 continuous time
            u = ops.aligned(u.float())
        logsnr, z_t = ops.q_sample(x, eps, u)                                     # :95-100
        if not distill:
            return module, guide, None, z_t, logsnr, x, eps, 0
        if cond_w is None:
            cond_w = 4.0 * self.rng.uniform((B,), dev)                            # :107
        cond_w = ops.aligned(cond_w.float())
        with torch.no_grad():
            logsnr_s = ops.logsnr_schedule(B, dev, u=u, shift=1.0 / self.num_steps)           # :118-119
            if self.teacher_mode == "step1":                                      # :121-126 one guided teacher step
                v, vu = self._teacher_eval(z_t, logsnr, guide, None, guided=True)
                _, x_target, eps_target = ops.ddim_step_vec(v, z_t, logsnr, logsnr_s, v_uncond=vu, cond_w=cond_w, mean_type=self.mean_type)
                loss_type = 1
            else:                                                                 # :128-154 two w-conditioned teacher steps
                logsnr_mid = ops.logsnr_schedule(B, dev, u=u, shift=0.5 / self.num_steps)
                v1, _ = self._teacher_eval(z_t, logsnr, guide, cond_w, guided=False)
                z_mid, _, _ = ops.ddim_step_vec(v1, z_t, logsnr, logsnr_mid, mean_type=self.mean_type)
                v2, _ = self._teacher_eval(z_mid, logsnr_mid, guide, cond_w, guided=False)
                z_teacher, x_pred_teacher, _ = ops.ddim_step_vec(v2, z_mid, logsnr_mid, logsnr_s, mean_type=self.mean_type)
                x_target, eps_target = ops.distill_target(z_teacher, z_t, x_pred_teacher, logsnr, logsnr_s, i_times)
                loss_type = 0
        return module, guide, cond_w, z_t, logsnr, x_target, eps_target, loss_type

    def training_losses(self, *, net, x, u=None, eps=None, i_times=None, cond_w=None):
        """Reference call shape (:81).  Differentiable through torch.autograd when grad mode is on."""
        assert x.dtype in [torch.float32, torch.float64]
        module, guide, w, z_t, logsnr, x_t, eps_t, loss_type = self._prepare(net, x, u, eps, i_times, cond_w)
        v = module(z_t, logsnr, guide=guide, cond_w=w)
        if torch.is_grad_enabled() and v.requires_grad:
            loss = _VLoss.apply(v, z_t, x_t, eps_t, logsnr, loss_type, self.mean_type)
        else:
            loss = ops.v_loss(v, z_t, x_t, eps_t, logsnr, loss_type=loss_type, mean_type=self.mean_type)[0]
        return {"loss": loss}

    def train_forward_backward(self, *, net, x, grad_scale, u=None, eps=None, i_times=None, cond_w=None,
                               on_grads_ready=None, join_side_before_ready=True):
        """Fused training pass used by DiffusionModel.train_step: forward, loss, dL/dv and the explicit backward
        schedule, leaving d(grad_scale * sum_b loss_b)/d(theta) in `module.flat_grads`.  No autograd graph."""
        module, guide, w, z_t, logsnr, x_t, eps_t, loss_type = self._prepare(net, x, u, eps, i_times, cond_w)
        ctx = {}
        v = module.forward_hip(z_t, logsnr, guide, w, ctx=ctx)
        loss_b, x_mse, eps_mse, dv = ops.v_loss(v, z_t, x_t, eps_t, logsnr, grad_scale=grad_scale, loss_type=loss_type, mean_type=self.mean_type)
        module.backward_hip(ctx, dv, on_grads_ready=on_grads_ready, join_side_before_ready=join_side_before_ready)
        return {"loss": loss_b, "x_mse": x_mse, "eps_mse": eps_mse, "logsnr": logsnr}

    # ---- sampling ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, *, net, init_x, cond_w=None, record=True, noises=None, net_cond_w=None):
        """:245-296.  Returns (all_zs, all_xs, all_eps) stacked [T, B, C, H, W] like the reference when `record`;
        with record=False only the final z is produced and returned as a 1-tuple-compatible triple
        (z[None], None, None) — the 3*T*B*C*H*W*4-byte trajectory is the dominant cost at large T*B otherwise.
        `noises` ([T, ...], indexed by step i) / `net_cond_w` inject the random draws (tests)."""
        module, guide, kw_cond_w = _unwrap(net)
        B = init_x.shape[0]
        dev = init_x.device
        if cond_w is not None and net_cond_w is None:
            net_cond_w = 4.0 * self.rng.uniform((B,), dev)           # :247-251
        if self.sampler not in ("ddim", "noisy", "teacher_test"):
            raise NotImplementedError(self.sampler)
        student_w, w, use_teacher = resolve_guidance(sample_cond_w=self.sample_cond_w, net_cond_w=net_cond_w, kw_cond_w=kw_cond_w,
                                                     has_teacher=self.teacher_net is not None, sampler=self.sampler)
        if use_teacher:
            module = self.teacher_net
        as_vec = lambda t: (t.to(dev).float().expand(B).contiguous() if isinstance(t, torch.Tensor)
                            else torch.full((B,), float(t), device=dev))
        student_w = ops.aligned(as_vec(student_w)) if student_w is not None else None
        w = ops.aligned(as_vec(w)) if w is not None else None
        if w is not None and guide is None:
            raise ValueError("classifier-free guidance needs class labels (net must carry guide=)")
        z_all = ops.aligned(init_x.float())
        n1 = int(np.prod(init_x.shape[1:]))
        # Large batches run as TWO half-batches on two HIP streams (round 5): the chains of different samples never meet (GroupNorm is per
        # sample, the update is elementwise), and a persistent kernel's start-up and tail - about one tile time per launch, ~ 80 launches per
        # forward - then overlap with the other half's kernels: -2.4 ... -3.0 % per forward at the bench configurations (tools/two_stream_probe.py;
        # four quarters are slower than one batch), the same bits.  Small batches (the captured-graph path) and odd batches stay on one stream.
        # Never together with the captured-graph path (both halves would replay ONE graph on ONE set of static buffers, whatever
        # GMK_SAMPLER_GRAPH_PIXELS is set to), and only at hidden_size 128: the bit-identity of the halves rests on the embedding GEMMs' K-split
        # not depending on the row count, which `gemm_ksplit` guarantees for K <= 256 = 2 x 128 only.
        K = 1
        nb_half = (B // 2) * (2 if w is not None else 1)
        if dev.type == "cuda" and self.SAMPLER_STREAMS >= 2 and B % 2 == 0 and ((B // 2) * n1) % 4 == 0 and (noises is None or isinstance(noises, torch.Tensor)) and \
                (B // 2) * init_x.shape[2] * init_x.shape[3] >= self.STREAM_MIN_PIXELS and module.channels == 128 and \
                not self._graph_path(module, nb_half, init_x.shape[2], init_x.shape[3]):
            K = 2
        bounds = [(k * B // K, (k + 1) * B // K) for k in range(K)]
        cut = lambda t, a, b_: None if t is None else ops.aligned(t[a:b_])
        offs = []                      # Philox counter of each step's noise draw for the WHOLE batch (ancestral sampler): chunks take their slice of it
        need_rng = self.sampler == "noisy" and noises is None
        cur = torch.cuda.current_stream() if dev.type == "cuda" else None
        streams = [cur] if K == 1 else self._chunk_streams(dev, K)
        gens = []
        for k, (a, b_) in enumerate(bounds):
            args = (module, cut(guide, a, b_), cut(student_w, a, b_), cut(w, a, b_), cut(z_all, a, b_),
                    noises if (noises is None or K == 1) else torch.as_tensor(noises)[:, a:b_], record, offs, a * n1 // 4)
            gens.append(self._sample_chunk(*args))
        if K > 1:
            # what a forward builds lazily after a weight update (packed weights, frequency tables) is enqueued HERE, on the stream both chunk
            # streams wait for: the first chunk's forward would otherwise re-pack on ITS stream and clear the host flag, and the second chunk's
            # convolutions would read the pack buffers without ever having waited for that kernel
            module.prepare_forward(dev)
            for st in streams:
                st.wait_stream(cur)

        def advance():
            outs = []
            for k in range(K):
                if K > 1:
                    with torch.cuda.stream(streams[k]):
                        outs.append(next(gens[k]))
                else:
                    outs.append(next(gens[k]))
            return outs
        for _ in range(self.num_steps):
            if need_rng:
                offs.append(self.rng._take(B * n1))
            advance()
            if K > 1:                                   # the throttle's event has to cover both halves
                for st in streams:
                    cur.wait_stream(st)
            ops.throttle()                              # at most two sampler iterations queued on the GPU (see ops.throttle)
        res = advance()                                 # the chunks' results
        if K == 1:
            return res[0]
        for st in streams:
            cur.wait_stream(st)
        for r in res:
            for t in r:
                if t is not None:
                    t.record_stream(cur)
        return tuple(None if res[0][j] is None else torch.cat([r[j] for r in res], dim=1) for j in range(3))

    SAMPLER_STREAMS = int(os.environ.get("GMK_SAMPLER_STREAMS", "2"))            # 1: every batch on one stream (A/B switch)
    STREAM_MIN_PIXELS = 1 << 18                                                  # images x H x W of a half-batch

    def _chunk_streams(self, dev, K):
        if getattr(self, "_streams", None) is None or len(self._streams) < K or self._streams[0].device != dev:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        return self._streams[:K]

    def _sample_chunk(self, module, guide, student_w, w, z_t, noises, record, offs, q0):
        """The sampler loop over one (chunk of a) batch as a generator: one `next` per step (everything it launches goes to the stream current at
        that call), then one more for the result.  offs[it] / q0: Philox counter of step `it`'s whole-batch noise draw / this chunk's offset in it."""
        B = z_t.shape[0]
        dev = z_t.device
        zs, xs, es = [], [], []
        guided = w is not None
        nb = 2 * B if guided else B
        if guided:       # conditional + unconditional evaluations share one 2B-image forward (:176-177)
            guide2 = torch.cat([guide, -torch.ones_like(guide)])
            sw2 = None if student_w is None else torch.cat([student_w, student_w])
            z2 = torch.cat([z_t, z_t])      # once: afterwards the update kernel writes both halves of the next 2B batch itself
        # the network's time vector: filled once here, then by the update kernel (the next logsnr_t is this step's logsnr_s);
        # two buffers alternate so that a forward still queued on the GPU never sees its input overwritten
        first = logsnr_schedule_cosine_host(sampler_times(self.num_steps - 1, self.num_steps)[0])
        # small batches replay a captured forward (one static log-SNR buffer, safe by stream order); large ones launch it kernel by kernel
        graphed, lvecs = self._forward_runner(module, z2 if guided else z_t, guide2 if guided else guide, sw2 if guided else student_w)
        lvecs[0].fill_(float(first))
        for it, i in enumerate(range(self.num_steps)[::-1]):
            u_t, u_s = sampler_times(i, self.num_steps)
            lt, ls = logsnr_schedule_cosine_host(u_t), logsnr_schedule_cosine_host(u_s)
            lvec, lnext = lvecs[it & 1], lvecs[(it + 1) & 1]
            if not guided:
                v = graphed(z_t) if graphed else module.forward_hip(z_t, lvec, guide, student_w)
                vu = None
            else:
                v2 = graphed(z2) if graphed else module.forward_hip(z2, lvec, guide2, sw2)
                v, vu = v2[:B], v2[B:]
            noise = None
            if self.sampler == "noisy":
                noise = ops.aligned(noises[i]) if noises is not None else ops.rng_normal(tuple(z_t.shape), self.rng.seed, offs[it] + q0, dev)   # :241
            z_t, xp, ep = ops.sampler_step(v, z_t, lt, ls, i == 0, v_uncond=vu, cond_w=w, noise=noise, want_pred=record, mean_type=self.mean_type,
                                           dup=guided, logsnr_next=lnext)
            if guided:
                z_t, z2 = z_t
            if record:
                zs.append(z_t); xs.append(xp); es.append(ep)
            yield None
        if record:
            yield torch.stack(zs), torch.stack(xs), torch.stack(es)
        else:
            yield z_t[None], None, None

