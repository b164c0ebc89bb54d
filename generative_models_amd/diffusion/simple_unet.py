"""HIP-backed `SimpleUnet` — host-side mirror of reference gms/diffusion/simple_unet.py:16-72.

Same constructor, call signature and state-dict keys as the reference module, so reference
checkpoints load; every arithmetic op runs in libgmk.so (include/gmk.h).  Differences in *how*:

* parameters are views into ONE flat fp32 arena (`net.flat_params`), gradients into a second one
  (`net.flat_grads`): the optimiser is a single fused kernel over the arena and the data-parallel gradient
  exchange is a handful of large RCCL all-reduces over contiguous ranges instead of 160 small tensors;
* activations are NHWC in `compute_dtype` (bf16 by default, fp32 for the 1e-3 parity mode); convolution weights
  are re-packed into K-contiguous tiles for the MFMA implicit GEMM whenever the parameters change;
* torch.cat([x, skip]) (simple_unet.py:150), F.interpolate(nearest) (:120), the embedding broadcast add (:184)
  and the residual add (:186) are folded into the convolution kernels' gather / epilogue;
* forward and backward are explicit schedules of kernel launches (no autograd graph); `SimpleUnetFunction`
  exposes the pair to torch.autograd so `loss.backward()` style code keeps working.

Extension over the reference: `in_channels` (the reference hard-codes 1, simple_unet.py:93,41).
Widths: `channels` 128 and 256 are native (multiples of the 128-channel MFMA tile).  Every other multiple of 32 up to 256 (the reference takes
any, simple_unet.py:17) runs ZERO-PADDED to 128 (32, 64, 96) or 256 (160, 192, 224) channels: every
parameter lives in its padded shape in the arena (the padding stays exactly zero under Adam: its gradients are exactly zero),
GroupNorm(32, C) over the real channels becomes GroupNorm(128 / (C / 32)) over the padded ones (all-zero groups normalise to zero), and
state_dict() / load_state_dict() / param() / grad() translate to and from the reference's shapes.  A compatibility path (3/4 or 15/16 of
the MFMA work multiplies zeros), there so that reference checkpoints and goldens of those widths run on the HIP kernels; widths above 256
(320 ... 512) raise.  Widths whose GroupNorm groups have 3, 5, 6 or 7 channels use the whole-sample streaming GroupNorm kernels with an explicit group size.
"""
import math
import os
from collections import OrderedDict

import torch
from torch import nn

from .. import ops

MAX_TIMESTEPS = 256  # simple_unet.py:13


# ----------------------------------------------------------------------------------------------------------
# parameter inventory (reference registration order)
# ----------------------------------------------------------------------------------------------------------
def _res_names(prefix, cin, cout, emb):
    out = [
        (f"{prefix}.in_layers.0.weight", (cin,)), (f"{prefix}.in_layers.0.bias", (cin,)),
        (f"{prefix}.in_layers.2.weight", (cout, cin, 3, 3)), (f"{prefix}.in_layers.2.bias", (cout,)),
        (f"{prefix}.emb_layers.1.weight", (cout, emb)), (f"{prefix}.emb_layers.1.bias", (cout,)),
        (f"{prefix}.out_layers.0.weight", (cout,)), (f"{prefix}.out_layers.0.bias", (cout,)),
        (f"{prefix}.out_layers.3.weight", (cout, cout, 3, 3)), (f"{prefix}.out_layers.3.bias", (cout,)),
    ]
    if cin != cout:
        out += [(f"{prefix}.skip_connection.weight", (cout, cin, 1, 1)), (f"{prefix}.skip_connection.bias", (cout,))]
    return out


RES_BLOCKS = ["down.seq.1", "down.seq.2", "down.seq.4", "down.seq.5", "turn", "up.seq.0.0", "up.seq.1", "up.seq.2",
              "up.seq.3.0", "up.seq.4", "up.seq.5", "up.seq.6"]


def param_inventory(C, in_channels=1, attention=False):
    E = 2 * C
    inv = []
    for name, fan_in in (("time_embed", 64), ("cond_w_embed", 64), ("guide_embed", 10)):
        inv += [(f"{name}.0.weight", (E, fan_in)), (f"{name}.0.bias", (E,)),
                (f"{name}.2.weight", (E, E)), (f"{name}.2.bias", (E,))]
    inv += [("down.seq.0.conv.weight", (C, in_channels, 3, 3)), ("down.seq.0.conv.bias", (C,))]
    inv += _res_names("down.seq.1", C, C, E) + _res_names("down.seq.2", C, C, E)
    inv += [("down.seq.3.conv.weight", (C, C, 3, 3)), ("down.seq.3.conv.bias", (C,))]
    inv += _res_names("down.seq.4", C, C, E) + _res_names("down.seq.5", C, C, E)
    inv += [("down.seq.6.conv.weight", (C, C, 3, 3)), ("down.seq.6.conv.bias", (C,))]
    inv += _res_names("turn", C, C, E)
    if attention:        # self-attention extension behind `turn` (no reference counterpart; defined by the tests' CPU restatement `attention_block`)
        inv += [("attn.norm.weight", (C,)), ("attn.norm.bias", (C,)), ("attn.qkv.weight", (3 * C, C, 1, 1)),
                ("attn.qkv.bias", (3 * C,)), ("attn.proj.weight", (C, C, 1, 1)), ("attn.proj.bias", (C,))]
    for i in range(7):
        if i in (0, 3):
            inv += _res_names(f"up.seq.{i}.0", 2 * C, C, E)
            inv += [(f"up.seq.{i}.1.conv.weight", (C, C, 3, 3)), (f"up.seq.{i}.1.conv.bias", (C,))]
        else:
            inv += _res_names(f"up.seq.{i}", 2 * C, C, E)
    inv += [("out.0.weight", (C,)), ("out.0.bias", (C,)), ("out.2.weight", (in_channels, C, 3, 3)),
            ("out.2.bias", (in_channels,))]
    return inv


def _is_cat_dim(name, dim):
    """True if dimension `dim` of parameter `name` runs over torch.cat([x, skip]) channels (the up blocks' first GroupNorm, conv1 and skip
    convolution inputs): in the padded layout the two C-channel halves sit at [0, C) and [CP, CP + C)."""
    if not name.startswith("up.seq."):
        return False
    return (".in_layers.0." in name and dim == 0) or (name.endswith(".in_layers.2.weight") and dim == 1) or \
           (name.endswith(".skip_connection.weight") and dim == 1)


def _pad_indices(name, real_shape, pad_shape, C, CP):
    """Per-dimension index lists: position of every real index inside the padded tensor."""
    out = []
    for d, (r, p) in enumerate(zip(real_shape, pad_shape)):
        if r == p:
            out.append(None)
        elif _is_cat_dim(name, d):
            assert r == 2 * C and p == 2 * CP
            out.append(torch.cat([torch.arange(C), CP + torch.arange(C)]))
        else:
            out.append(torch.arange(r))          # C -> CP, 2C -> 2CP (embedding width): the real entries come first
    return out


def _attach(root, dotted, param):
    """Register `param` under the nested attribute path `dotted`, creating container modules on the way."""
    parts = dotted.split(".")
    mod = root
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, nn.Module())
        mod = mod._modules[part]
    mod.register_parameter(parts[-1], param)


class SimpleUnet(nn.Module):
    def __init__(self, channels, dropout=0.0, in_channels=1, compute_dtype=torch.bfloat16, attention=False, act_dtype=None):
        super().__init__()
        if channels % 32 or not 32 <= channels <= 256:
            raise ValueError(f"the HIP path is built for hidden_size 128 (DiffusionModel's default, every BASELINE config) and 256 (the default of "
                             f"gms/main.py:23); every other multiple of 32 up to 256 runs zero-padded to 128 / 256 channels; got {channels}")
        if attention and channels != 128:
            raise ValueError("the self-attention extension is built for 128 channels (one head over C = 128)")
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout}")
        if not 1 <= in_channels <= 4:
            raise ValueError("in_channels must be 1..4")
        if compute_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("compute_dtype must be torch.bfloat16 or torch.float32")
        # 16-bit mode: forward activations and forward weight packs are fp16 (11 significant bits: the precision of the reference's own fp16
        # autocast forward, diffusion_model.py:68), gradients stay bf16 (fp32's range, so no GradScaler).  act_dtype=torch.bfloat16 (or
        # GMK_ACT_DTYPE=bf16) is the all-bf16 path of rounds 1-2, kept for A/B runs.
        if act_dtype is None:
            act_dtype = torch.float16 if compute_dtype == torch.bfloat16 and os.environ.get("GMK_ACT_DTYPE", "fp16") != "bf16" else compute_dtype
        if act_dtype != compute_dtype and not (compute_dtype == torch.bfloat16 and act_dtype == torch.float16):
            raise ValueError("act_dtype must equal compute_dtype, or be torch.float16 next to compute_dtype=torch.bfloat16")
        # hidden_size: the reference's `channels`; self.channels: the width the kernels run at (narrow nets zero-padded to one 128-channel tile)
        self.hidden_size = channels
        padded = 128 if channels <= 128 else 256                      # the width the kernels run at: whole 128-channel MFMA tiles
        self._narrow = channels != padded                             # zero-padded (32 ... 224 except 128): a compatibility path
        # GroupNorm(32, C) has C / 32 channels per group; in the padded layout that is `padded / cpg` groups where cpg divides it (32, 64:
        # the all-zero groups normalise to zero), and an explicit group size otherwise (96, 160, 192, 224: 3, 5, 6, 7 channels - passed as a
        # NEGATIVE group count, ops.gn_silu_fwd).  _g2: the up blocks' GroupNorm(32, 2C), per C-channel source (16 groups of 2C / 32 channels).
        cpg1, cpg2 = channels // 32, 2 * channels // 32
        self._g1 = padded // cpg1 if padded % cpg1 == 0 and cpg1 in (1, 2, 4, 8, 16) else -cpg1
        self._g2 = padded // cpg2 if padded % cpg2 == 0 and cpg2 in (1, 2, 4, 8, 16) else -cpg2
        self._real_inventory = param_inventory(channels, in_channels, bool(attention))
        channels = padded
        self.channels, self.in_channels, self.compute_dtype, self.act_dtype = channels, in_channels, compute_dtype, act_dtype
        self.dropout = float(dropout)      # nn.Dropout(p) of every ResBlock's out_layers (simple_unet.py:171); training mode only
        self.drop_seed, self._drop_counter = 0x5EEDD0, 0
        self.attention = bool(attention)
        self.attention_fp8 = int(attention) == 2      # attention=2: QK^T / PV on the fp8 matrix cores (BASELINE config 5)
        self._inventory = param_inventory(channels, in_channels, self.attention)
        # arena order: the 12 emb_layers Linear weights, their biases, the 12 conv1 biases (one batched GEMM with two bias
        # tables serves all 12 ResBlocks), then everything else in reference order; every tensor starts on a 16-byte boundary.
        emb_w = [f"{b}.emb_layers.1.weight" for b in RES_BLOCKS]
        emb_b = [f"{b}.emb_layers.1.bias" for b in RES_BLOCKS]
        conv1_b = [f"{b}.in_layers.2.bias" for b in RES_BLOCKS]      # second bias table of the emb_layers GEMM (_embed_fwd)
        shapes = dict(self._inventory)
        front = emb_w + emb_b + conv1_b
        order = front + [n for n, _ in self._inventory if n not in set(front)]
        self._offsets = OrderedDict()
        off = 0
        for n in order:
            self._offsets[n] = off
            off += (math.prod(shapes[n]) + 3) // 4 * 4
        self._arena_size = off
        flat = torch.zeros(off, dtype=torch.float32)
        self._init_values(flat, shapes)
        # registration order = reference order, so state_dict() lists keys exactly like the reference
        for n, shp in self._inventory:
            _attach(self, n, nn.Parameter(flat[self._offsets[n]:self._offsets[n] + math.prod(shp)].view(shp)))
        self._shapes = shapes
        self._real_shapes = dict(self._real_inventory)
        self.padding_numel = sum(math.prod(s) for _, s in self._inventory) - sum(math.prod(s) for _, s in self._real_inventory)
        self._bind(flat)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_params_changed())
        if self._narrow:
            self._register_load_state_dict_pre_hook(self._pad_incoming_state)

    # ---- initialisation: torch defaults for Linear / Conv2d / GroupNorm, `out_layers.3` zeroed (simple_unet.py:172)
    def _init_values(self, flat, shapes):
        real = dict(self._real_inventory)
        for n, shp in self._inventory:
            padded = flat[self._offsets[n]:self._offsets[n] + math.prod(shp)].view(shp)
            view = padded if real[n] == shp else torch.empty(real[n])      # narrow: draw in the reference's shape (its fan-in), then pad
            base, kind = n.rsplit(".", 1)
            wshape = real.get(base + ".weight")
            if len(wshape) >= 2:                      # Linear / Conv2d: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
                if ".out_layers.3" in n or n.startswith("attn.proj"):
                    view.zero_()
                else:
                    view.uniform_(-bound, bound)
            else:                                      # GroupNorm affine
                view.fill_(1.0 if kind == "weight" else 0.0)
            if view is not padded:
                padded.copy_(self._pad(n, view))

    # ---- narrow widths: reference shapes <-> zero-padded arena shapes -----------------------------------------------
    def _pad(self, name, t):
        """Reference-shaped tensor -> its zero-padded arena shape."""
        pshape = self._shapes[name] if hasattr(self, "_shapes") else dict(self._inventory)[name]
        if tuple(t.shape) == tuple(pshape):
            return t
        idx = _pad_indices(name, tuple(t.shape), pshape, self.hidden_size, self.channels)
        out = torch.zeros(pshape, dtype=t.dtype, device=t.device)
        grids = [torch.arange(s, device=t.device) if ix is None else ix.to(t.device) for s, ix in zip(t.shape, idx)]
        out[torch.meshgrid(*grids, indexing="ij")] = t
        return out

    def _unpad(self, name, t):
        """Padded arena tensor -> a reference-shaped copy."""
        rshape = self._real_shapes[name]
        if tuple(t.shape) == tuple(rshape):
            return t
        for d, ix in enumerate(_pad_indices(name, rshape, tuple(t.shape), self.hidden_size, self.channels)):
            if ix is not None:
                t = t.index_select(d, ix.to(t.device))
        return t

    def _pad_incoming_state(self, state_dict, prefix, *unused):
        for n, rshape in self._real_inventory:
            v = state_dict.get(prefix + n)
            if v is not None and tuple(v.shape) == tuple(rshape):
                state_dict[prefix + n] = self._pad(n, v)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        if self._narrow:      # the reference's shapes (copies: the padded parameters are not views of them)
            for n, _ in self._real_inventory:
                sd[prefix + n] = self._unpad(n, sd[prefix + n])
        return sd

    # ---- flat arena management -------------------------------------------------------------------------
    def _bind(self, flat):
        self.flat_params = flat
        self.flat_grads = torch.zeros_like(flat)
        named = dict(self.named_parameters())
        self._pv, self._gv = {}, {}
        for n, shp in self._inventory:
            o, k = self._offsets[n], math.prod(shp)
            p = named[n]
            p.data = flat[o:o + k].view(shp)
            p.grad = self.flat_grads[o:o + k].view(shp)
            self._pv[n] = p.data
            self._gv[n] = p.grad
        self._packs = None
        self._up_packs = None
        self._plist = [named[n] for n, _ in self._inventory]
        self._packed_version = -1
        self._side = None          # side stream of the weight gradients (backward_hip)
        self._cu_part = None       # (CUs of the chip, CUs of the side stream) while a partitioned backward pass is being enqueued
        self._packs_stale = True
        self._freqs = {}

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        # .to()/.cuda() moved every parameter separately: rebuild the arena on the new device and re-point the views
        named = dict(self.named_parameters())
        dev = next(iter(named.values())).device
        flat = torch.zeros(self._arena_size, dtype=torch.float32, device=dev)
        for n, shp in self._inventory:
            o, k = self._offsets[n], math.prod(shp)
            flat[o:o + k].copy_(named[n].data.reshape(-1).float())
        self._bind(flat)
        return self

    def _conv(self, *args, **kw):
        """ops.conv_igemm with this net's width as the output-channel count (the C-ABI default is one 128-channel block)."""
        kw.setdefault("cout", self.channels)
        return ops.conv_igemm(*args, **kw)

    def _upsample_conv(self, name, x, out_hw):
        """`Upsample` (simple_unet.py:112-122): nearest x2 + 3x3 convolution.  Sub-pixel form (four 2x2-tap parities of the low-resolution tensor on
        pre-summed weights, gmk_conv_subpixel) where the kernel takes the shape; else the nearest-x2 addressing of the stride-1 kernels."""
        B, H, W, c = x.shape
        if self._up_packs is not None and ops.conv_subpixel_ok(B, H, W, c, x.dtype):
            return ops.conv_subpixel(x, self._up_packs[name][0], self.channels, ops.SUBPIXEL_UPSAMPLE, bias=self._pv[name + ".bias"])
        return self._conv([x], self._packs[name][0], self.channels, 3, ops.UPSAMPLE2, out_hw, bias=self._pv[name + ".bias"], gn_stats=True)

    def _upsample_wgrad(self, name, dy, x):
        """Weight gradient of `Upsample` on the side stream: the sub-pixel slot correlation where the kernel takes the shape."""
        B, H, W, c = x.shape
        dw = self._gv[name + ".weight"]
        if self._up_packs is not None:
            # the kernel's plan (split count, workspace) depends on the CU limit in force where it is LAUNCHED - the side stream's under a partitioned
            # backward pass - so the eligibility question is asked there too, and the other form is the fallback in the same scope
            def run():
                if ops.conv_wgrad_subpixel_ok(B, H, W, c, dy.dtype):
                    ops.conv_wgrad_subpixel(dy, x, dw)
                else:
                    ops.conv_wgrad(dy, [x], 3, ops.UPSAMPLE2, dw)
            self._on_side(run, (dy, x))
            return dw
        return self._wgrad(dy, [x], 3, ops.UPSAMPLE2, dw)

    def _upsample_dgrad(self, name, dy):
        """Data gradient of `Upsample`: sumpool2x2(dgrad3x3(dy)) - one launch in the sub-pixel form (no high-resolution intermediate), else two."""
        B, H, W, c = dy.shape
        if self._up_packs is not None and ops.conv_subpixel_ok(B, H // 2, W // 2, c, dy.dtype):
            return ops.conv_subpixel(dy, self._up_packs[name][1], self.channels, ops.SUBPIXEL_UPSAMPLE_DGRAD)
        return ops.sumpool2x2(self._conv([dy], self._packs[name][1], self.channels, 3, ops.NORMAL, (H, W)))

    def mark_params_changed(self):
        self._packs_stale = True

    def _version_sum(self):
        """Sum of the parameters' autograd version counters: any in-place update through torch (torch.optim, EMA copy_,
        p.data.mul_ ...) bumps one of them, so the packed convolution weights are refreshed without the caller having to
        say so.  Kernel-side updates (FusedAdam, broadcast) do not bump versions and call mark_params_changed()."""
        return sum(p._version for p in self._plist)

    def param(self, name):
        """The parameter in the reference's shape (a view of the arena; a copy for the zero-padded narrow widths)."""
        return self._unpad(name, self._pv[name]) if self._narrow else self._pv[name]

    def grad(self, name):
        return self._unpad(name, self._gv[name]) if self._narrow else self._gv[name]

    def arena_range(self, names):
        """(start, end) of the contiguous arena range spanned by `names` (used for gradient buckets)."""
        starts = [self._offsets[n] for n in names]
        ends = [self._offsets[n] + (math.prod(self._shapes[n]) + 3) // 4 * 4 for n in names]
        return min(starts), max(ends)

    # ---- weight packs ----------------------------------------------------------------------------------
    def _conv_names(self):
        names = ["down.seq.3.conv", "down.seq.6.conv", "up.seq.0.1.conv", "up.seq.3.1.conv"]
        for b in RES_BLOCKS:
            names += [f"{b}.in_layers.2", f"{b}.out_layers.3"]
            if f"{b}.skip_connection.weight" in self._shapes:
                names.append(f"{b}.skip_connection")
        if self.attention:
            names += ["attn.qkv", "attn.proj"]
        return names

    def _repack(self):
        dev = self.flat_params.device
        if dev.type != "cuda":
            raise RuntimeError("SimpleUnet (HIP) must live on a GPU device: call .to('cuda') first; no CPU fallback")
        names = self._conv_names()
        if self._packs is None:
            total = sum(2 * math.prod(self._shapes[n + ".weight"]) for n in names)
            self._pack_buf = torch.empty(total, device=dev, dtype=self.compute_dtype)
            self._packs = {}
            off = 0
            table = ([], [], [], [], [])
            # forward packs in the activation type (fp16 in the 16-bit mode), data-gradient packs in the gradient type; the attention
            # extension keeps bf16 internals (its input is converted at the block boundary)
            f16 = self.act_dtype == torch.float16
            self._pack_f16 = [int(f16 and not n.startswith("attn.")) for n in names]
            for n, half in zip(names, self._pack_f16):
                shp = self._shapes[n + ".weight"]
                k = math.prod(shp)
                wf = self._pack_buf[off:off + k]
                self._packs[n] = (wf.view(torch.float16) if half else wf, self._pack_buf[off + k:off + 2 * k])
                for col, v in zip(table, (self._offsets[n + ".weight"], off, shp[0], shp[1], shp[2])):
                    col.append(v)
                off += 2 * k
            self._pack_table = table
        ops.pack_conv_weights_multi(self.flat_params, self._pack_buf, self._pack_table, self._pack_f16)      # all convolutions, one launch
        # the two `Upsample` convolutions (simple_unet.py:112-122) also get the pack of their sub-pixel form: 16 pre-summed 2x2-tap matrices
        if self.channels == 128 and self.compute_dtype == torch.bfloat16:
            if self._up_packs is None:      # (forward pack in the activation type, data-gradient pack in the gradient type)
                self._up_packs = {n: (torch.empty(16 * 128 * 128, device=dev, dtype=self.act_dtype),
                                      torch.empty(16 * 128 * 128, device=dev, dtype=self.compute_dtype)) for n in ("up.seq.0.1.conv", "up.seq.3.1.conv")}
            for n, (bf, bd) in self._up_packs.items():
                ops.pack_upsample_weight(self._pv[n + ".weight"], bf, bd)
        self._packs_stale = False
        self._packed_version = self._version_sum()

    # ---- embedding path (simple_unet.py:45-64 + the 12 emb_layers of :166) --------------------------------
    def _freq_table(self, max_period, dev):
        key = (max_period, dev)
        if key not in self._freqs:
            self._freqs[key] = ops.timestep_freqs(max_period, dev)
        return self._freqs[key]

    def _mlp_fwd(self, prefix, x, ctx, rowscale=None, out=None):
        P = self._pv
        h1 = ops.gemm(x, P[f"{prefix}.0.weight"].t(), bias=P[f"{prefix}.0.bias"])
        y = ops.gemm(h1, P[f"{prefix}.2.weight"].t(), out=out, bias=P[f"{prefix}.2.bias"], rowscale=rowscale,
                     silu_a=True, accumulate=out is not None)
        if ctx is not None:
            ctx[prefix] = (x, h1, rowscale)
        return y

    def _mlp_bwd(self, prefix, ctx, demb):
        P, G = self._pv, self._gv
        x, h1, rowscale = ctx[prefix]
        d = ops.scale_rows(demb, rowscale) if rowscale is not None else demb
        ops.gemm(d.t(), h1, out=G[f"{prefix}.2.weight"], silu_b=True)      # dW2 = d^T . silu(h1)
        ops.colsum(d, G[f"{prefix}.2.bias"], defer=True)
        dsh = ops.gemm(d, P[f"{prefix}.2.weight"])
        dh1 = ops.silu_bwd(dsh, h1)
        ops.gemm(dh1.t(), x, out=G[f"{prefix}.0.weight"])
        ops.colsum(dh1, G[f"{prefix}.0.bias"], defer=True)

    def _embed_fwd(self, logsnr, guide, cond_w, ctx):
        dev = logsnr.device
        C = self.channels
        te = ops.timestep_embedding(logsnr.float().contiguous(), self._freq_table(MAX_TIMESTEPS, dev))
        emb = self._mlp_fwd("time_embed", te, ctx)
        if guide is not None:
            onehot, keep = ops.guide_onehot(guide.contiguous())
            self._mlp_fwd("guide_embed", onehot, ctx, rowscale=keep, out=emb)
        if cond_w is not None:
            ce = ops.timestep_embedding(cond_w.float().contiguous(), self._freq_table(4, dev))
            self._mlp_fwd("cond_w_embed", ce, ctx, out=emb)
        # all 12 ResBlock emb_layers (SiLU -> Linear(2C -> C)) as one GEMM over the arena-contiguous weights
        o0 = self._offsets[f"{RES_BLOCKS[0]}.emb_layers.1.weight"]
        wcat = self.flat_params[o0:o0 + 12 * C * 2 * C].view(12 * C, 2 * C)
        b0 = self._offsets[f"{RES_BLOCKS[0]}.emb_layers.1.bias"]
        bcat = self.flat_params[b0:b0 + 12 * C]
        # conv1's bias rides with the embedding: h = conv1(a) + bias + emb_out[..., None, None] (simple_unet.py:183) is a
        # per-(sample, channel) addend, applied by the GroupNorm that consumes h (gmk_gn_silu_fwd xadd) instead of by the
        # MFMA kernel's epilogue, where its loads cost 25 % of a tile
        c0 = self._offsets[f"{RES_BLOCKS[0]}.in_layers.2.bias"]
        emb_all = ops.gemm(emb, wcat.t(), bias=bcat, bias2=self.flat_params[c0:c0 + 12 * C], silu_a=True)
        if ctx is not None:
            ctx["emb"] = emb
            ctx["emb_all"] = emb_all
        return emb_all

    def _embed_bwd(self, ctx, demb_all):
        C = self.channels
        emb = ctx["emb"]
        o0 = self._offsets[f"{RES_BLOCKS[0]}.emb_layers.1.weight"]
        b0 = self._offsets[f"{RES_BLOCKS[0]}.emb_layers.1.bias"]
        wcat = self.flat_params[o0:o0 + 12 * C * 2 * C].view(12 * C, 2 * C)
        gw = self.flat_grads[o0:o0 + 12 * C * 2 * C].view(12 * C, 2 * C)
        gb = self.flat_grads[b0:b0 + 12 * C]
        ops.gemm(demb_all.t(), emb, out=gw, silu_b=True)        # dWcat[n][k] = sum_b dE[b][n] * silu(emb)[b][k]
        ops.colsum(demb_all, gb, defer=True)
        dsemb = ops.gemm(demb_all, wcat)
        demb = ops.silu_bwd(dsemb, emb)
        self._mlp_bwd("time_embed", ctx, demb)
        if "guide_embed" in ctx:
            self._mlp_bwd("guide_embed", ctx, demb)
        if "cond_w_embed" in ctx:
            self._mlp_bwd("cond_w_embed", ctx, demb)

    # ---- ResBlock (simple_unet.py:155-186) ---------------------------------------------------------------
    def _res_fwd(self, name, srcs, emb_all, blk, ctx):
        P, C = self._pv, self.channels
        B, H, W, _ = srcs[0].shape
        gpc = self._g2 if len(srcs) == 2 else self._g1           # GroupNorm(32, cin): 16 groups per C-channel source
        skip = {}
        fwd_side = len(srcs) == 2 and ops.FWD_SIDE and ops.WGRAD_STREAM
        # `skip_connection(x) + h` (simple_unet.py:174-186) as ONE launch where the kernel takes the shape: no 1x1 launch, no skip tensor in HBM
        fold = len(srcs) == 2 and not self._narrow and ops.conv_skipfold_ok(srcs[0], srcs)
        if fold:
            fwd_side = False
        elif len(srcs) == 2:
            # The 1x1 skip convolution (HBM-bound) only needs the block input.  With GMK_FWD_SIDE=1 it runs on the side stream beside
            # GroupNorm / conv1 / GroupNorm and is joined in front of conv2 (-1 % of a forward).  OFF by default: that co-residency
            # (conv_igemm_kernel's LDS staging next to the GroupNorm kernel) is where round 1's unexplained fault lived (docs/EXPERIMENTS.md section 5).
            wfs, _ = self._packs[f"{name}.skip_connection"]
            run_skip = lambda: skip.__setitem__("res", self._conv(srcs, wfs, C, 1, ops.NORMAL, (H, W),
                                                                      bias=P[f"{name}.skip_connection.bias"]))
            if fwd_side:
                self._on_side(run_skip, tuple(srcs))
            else:
                run_skip()
        eadd = emb_all[:, blk * C:(blk + 1) * C]
        wf1, _ = self._packs[f"{name}.in_layers.2"]
        wf2, _ = self._packs[f"{name}.out_layers.3"]
        dropping = self.dropout > 0.0 and self.training and name != "up.seq.3.0"
        def conv2_folded(a2):
            wfs, _ = self._packs[f"{name}.skip_connection"]
            return ops.conv3x3_skipfold(a2, wf2, P[f"{name}.out_layers.3.bias"], srcs, wfs, P[f"{name}.skip_connection.bias"])
        if ctx is None and not dropping and ops.GN_FUSE and not self._narrow and ops.conv_gn_fusable(srcs):
            # Inference (nothing is kept for a backward pass): GroupNorm-apply + SiLU run inside the convolutions' producer waves.
            # A statistics-only launch reads the raw tensor once and leaves the per-(sample, channel) affine tables; the normalised
            # tensors `a` / `a2` of simple_unet.py:161-163,169-172 are never written or read back (bit-identical results).
            tsc = torch.empty((B, len(srcs) * C), device=srcs[0].device, dtype=torch.float32)
            tsh = torch.empty_like(tsc)
            for i, s in enumerate(srcs):
                ops.gn_stats(s, P[f"{name}.in_layers.0.weight"][i * C:(i + 1) * C], P[f"{name}.in_layers.0.bias"][i * C:(i + 1) * C], gpc,
                             tsc[:, i * C:(i + 1) * C], tsh[:, i * C:(i + 1) * C])
            h = self._conv(srcs, wf1, C, 3, ops.NORMAL, (H, W), gn=(tsc, tsh))
            if fold and ops.SKIP_FOLD_OVER_FUSE:      # the fold and the in-convolution GroupNorm do not combine (yet): conv2's input is materialised
                self._emb_ready()
                a2, _, _ = ops.gn_silu_fwd(h, P[f"{name}.out_layers.0.weight"], P[f"{name}.out_layers.0.bias"], self._g1, xadd=eadd)
                return conv2_folded(a2)
            if fold:          # GMK_SKIP_FOLD_FUSE=fuse: conv2 keeps its in-convolution GroupNorm, the skip convolution its own launch
                skip["res"] = self._conv(srcs, self._packs[f"{name}.skip_connection"][0], C, 1, ops.NORMAL, (H, W),
                                         bias=P[f"{name}.skip_connection.bias"])
            t2c = torch.empty((B, C), device=h.device, dtype=torch.float32)
            t2h = torch.empty_like(t2c)
            self._emb_ready()
            ops.gn_stats(h, P[f"{name}.out_layers.0.weight"], P[f"{name}.out_layers.0.bias"], self._g1, t2c, t2h, xadd=eadd)
            if len(srcs) == 2:
                res = skip["res"]
                if fwd_side:
                    self._join_side()
                    res.record_stream(torch.cuda.current_stream())
            else:
                res = srcs[0]
            return self._conv([h], wf2, C, 3, ops.NORMAL, (H, W), bias=P[f"{name}.out_layers.3.bias"], residual=res, gn=(t2c, t2h))
        a, stats1 = [], []
        for i, s in enumerate(srcs):
            g = P[f"{name}.in_layers.0.weight"][i * C:(i + 1) * C]
            b = P[f"{name}.in_layers.0.bias"][i * C:(i + 1) * C]
            y, mean, rstd = ops.gn_silu_fwd(s, g, b, gpc)
            a.append(y); stats1.append((mean, rstd))
        h = self._conv(a, wf1, C, 3, ops.NORMAL, (H, W))      # bias + embedding enter through `xadd` below
        drop = None
        # reference quirk kept: `up.seq[3]`'s ResBlock is built WITHOUT the dropout argument (simple_unet.py:138), so it never drops
        if dropping:      # mask = Philox uniform >= p, regenerated by the backward kernel
            drop = (self.dropout, self.drop_seed, self._drop_counter)
            self._drop_counter += (h.numel() + 3) // 4
        self._emb_ready()
        a2, mean2, rstd2 = ops.gn_silu_fwd(h, P[f"{name}.out_layers.0.weight"], P[f"{name}.out_layers.0.bias"], self._g1, dropout=drop,
                                           xadd=eadd)
        if fold:
            out = conv2_folded(a2)
        else:
            if len(srcs) == 2:
                res = skip["res"]
                if fwd_side:
                    self._join_side()
                    res.record_stream(torch.cuda.current_stream())      # allocated on the side stream, consumed here
            else:
                res = srcs[0]
            out = self._conv([a2], wf2, C, 3, ops.NORMAL, (H, W), bias=P[f"{name}.out_layers.3.bias"], residual=res,
                             gn_stats=True)
        if ctx is not None:
            ctx[name] = (srcs, a, stats1, h, a2, (mean2, rstd2))
            ctx[name + ".dropout"] = drop
        return out

    # ---- self-attention extension (north_star; no reference counterpart — defined by the tests' CPU restatement `attention_block`) ------------
    def _attn_core_fwd(self, a, residual=None, keep=False):
        """The contraction core of the attention block on an NHWC map `a` (compute dtype): proj(softmax(q k^T / sqrt(C)) v) (+ residual),
        (q, k, v) = conv1x1(a); single head over C.  Same arithmetic as the reference's `CausalSelfAttention.forward`
        (gms/autoregs/pixel_transformer.py:101-122) with n_head = 1 and no mask: pinned by tests/golden/attn_core_{64,256}.npz.
        -> (out, saved for `_attn_core_bwd`)."""
        P, C, T = self._pv, self.channels, self.compute_dtype
        B, H, W, _ = a.shape
        N = H * W
        if N % 8 or N > 1024:
            raise ValueError(f"the attention level has {N} tokens: the HIP path needs a multiple of 8, at most 1024 "
                             f"(input sizes 32 / 64: 64 / 256 tokens)")
        qkv = self._conv([a], self._packs["attn.qkv"][0], 3 * C, 1, ops.NORMAL, (H, W), cout=3 * C, bias=P["attn.qkv.bias"])
        t = qkv.view(B, N, 3 * C)
        q, k, v = t[:, :, :C], t[:, :, C:2 * C], t[:, :, 2 * C:]
        if T == torch.bfloat16 and N in (64, 128, 256):       # fused: K / V resident in LDS, no fp32 score matrix in HBM
            # (the fused backward recomputes P in registers: nothing N x N is kept for it)
            o, Pm = ops.attention_fwd(t, C ** -0.5, want_p=keep and not ops.attention_bwd_fused_ok(t), fp8=self.attention_fp8)
            o = o.view(B, H, W, C)
        else:
            S = ops.bgemm_nt(q, k, out_dtype=torch.float32)
            Pm = ops.softmax_fwd(S, C ** -0.5, T)
            o = ops.bgemm_nt(Pm, ops.transpose_last2(v)).view(B, H, W, C)
        out = self._conv([o], self._packs["attn.proj"][0], C, 1, ops.NORMAL, (H, W), bias=P["attn.proj.bias"], residual=residual)
        return out, (a, qkv, Pm, o)

    def _attn_core_bwd(self, saved, dout):
        """Backward of `_attn_core_fwd` for the gradient `dout` of its output: fills the gradients of attn.proj / attn.qkv, -> gradient of `a`."""
        P, G, C = self._pv, self._gv, self.channels
        a, qkv, Pm, o = saved
        B, H, W, _ = a.shape
        N = H * W
        scale = C ** -0.5
        t = qkv.view(B, N, 3 * C)
        q, k, v = t[:, :, :C], t[:, :, C:2 * C], t[:, :, 2 * C:]
        ops.colsum(ops.chansum(dout), G["attn.proj.bias"], defer=True)
        self._wgrad(dout, [o], 1, ops.NORMAL, G["attn.proj.weight"])
        do = self._conv([dout], self._packs["attn.proj"][1], C, 1, ops.NORMAL, (H, W)).view(B, N, C)
        # o = P v,  P = softmax(scale * q k^T)
        if Pm is None:          # fused backward: P recomputed block by block in registers (gmk_attention_bwd), no [B, N, N] tensor anywhere
            dqkv = ops.attention_bwd(t, o.view(B, N, C), do.contiguous()).view(qkv.shape)
        else:
            dP = ops.bgemm_nt(do, v, out_dtype=torch.float32)
            dS = ops.softmax_bwd(Pm, dP, scale)
            dqkv = torch.empty_like(qkv)
            d3 = dqkv.view(B, N, 3 * C)
            ops.bgemm_nt(ops.transpose_last2(Pm), ops.transpose_last2(do), out=d3[:, :, 2 * C:])     # dv = P^T do
            ops.bgemm_nt(dS, ops.transpose_last2(k), out=d3[:, :, :C])                               # dq = dS k
            ops.bgemm_nt(ops.transpose_last2(dS), ops.transpose_last2(q), out=d3[:, :, C:2 * C])     # dk = dS^T q
        # (q, k, v) = conv1x1(a)
        G["attn.qkv.bias"].copy_(dqkv.float().sum((0, 1, 2)))        # 3C-channel bias gradient: tiny, off the hot path
        self._wgrad(dqkv, [a], 1, ops.NORMAL, G["attn.qkv.weight"])
        return self._conv([dqkv], self._packs["attn.qkv"][1], C, 1, ops.NORMAL, (H, W))

    def _attn_fwd(self, x, ctx):
        """x NHWC [B,H,W,C] -> x + core(SiLU(GN(x))): the extension's placement (pre-activation, residual) around `_attn_core_fwd`."""
        P, T = self._pv, self.compute_dtype
        x = ops.cast16(x, T) if x.dtype != T else x        # fp16 stream -> the block's bf16 internals (16-bit mode)
        a, mean, rstd = ops.gn_silu_fwd(x, P["attn.norm.weight"], P["attn.norm.bias"], 32)
        out, saved = self._attn_core_fwd(a, residual=x, keep=ctx is not None)
        if ctx is not None:
            ctx["attn"] = (x, mean, rstd, saved)
        return ops.cast16(out, self.act_dtype) if self.act_dtype != T else out

    def _attn_bwd(self, ctx, dout):
        """-> (dx, per-sample channel sums of dx)."""
        P, G, C = self._pv, self._gv, self.channels
        x, mean, rstd, saved = ctx.pop("attn")
        B = x.shape[0]
        da = self._attn_core_bwd(saved, dout)            # out = x + core(a)
        s = torch.empty((B, C), device=x.device, dtype=torch.float32)
        dx, dgp, dbp = ops.gn_silu_bwd(da, x, P["attn.norm.weight"], P["attn.norm.bias"], mean, rstd, dadd1=dout, dxsum=s)
        ops.colsum(dgp, G["attn.norm.weight"], defer=True); ops.colsum(dbp, G["attn.norm.bias"], defer=True)
        return dx, s

    def _wgrad(self, dy, srcs, ksize, mode, dw):
        """Weight gradient on the side stream (ops.WGRAD_STREAM): it depends only on dy and the saved activations, so it
        runs beside the data-gradient chain and fills the CUs the persistent kernels' tails leave idle."""
        self._on_side(lambda: ops.conv_wgrad(dy, srcs, ksize, mode, dw), (dy, *srcs))
        return dw

    def _on_side(self, fn, tensors):
        """Run fn() on the side stream, ordered behind everything enqueued so far; `tensors` are the inputs it reads."""
        if not ops.WGRAD_STREAM:
            return fn()
        if self._side is None:
            self._side = torch.cuda.Stream(device=tensors[0].device)
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        part = self._cu_part
        if part is not None:               # the side stream's persistent grids take `part[1]` CUs, the chain's the rest (backward_hip)
            ops.set_cu_limit(part[1])
        try:
            with torch.cuda.stream(side):
                fn()
        finally:
            if part is not None:
                ops.set_cu_limit(part[0] - part[1])
        for t in tensors:
            t.record_stream(side)

    def _emb_ready(self):
        """Join the side stream if the embedding path of this forward is still pending there (forward_hip, GMK_EMB_SIDE): its first consumer follows."""
        pend = getattr(self, "_emb_pending", None)
        if pend is not None:
            self._emb_pending = None
            self._join_side()
            pend.record_stream(torch.cuda.current_stream())

    def _join_side(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _res_bwd(self, name, ctx, dout, dout_sum, demb_all, blk, extra_add=None):
        """dout: gradient of the block output (NHWC); dout_sum: its per-sample channel sums [B, C].
        extra_add: per-source optional extra gradient tensors added into the returned source gradients.
        Returns [(dsrc, dsrc_sum)] per source."""
        P, G, C = self._pv, self._gv, self.channels
        srcs, a, stats1, h, a2, (mean2, rstd2) = ctx.pop(name)
        B, H, W, _ = dout.shape
        two = len(srcs) == 2
        # conv2 (out_layers.3)
        ops.colsum(dout_sum, G[f"{name}.out_layers.3.bias"], defer=True)
        self._wgrad(dout, [a2], 3, ops.NORMAL, G[f"{name}.out_layers.3.weight"])
        _, wd2 = self._packs[f"{name}.out_layers.3"]
        da2 = self._conv([dout], wd2, C, 3, ops.NORMAL, (H, W))
        dh, dgp, dbp = ops.gn_silu_bwd(da2, h, P[f"{name}.out_layers.0.weight"], P[f"{name}.out_layers.0.bias"], mean2,
                                       rstd2, dxsum=demb_all[:, blk * C:(blk + 1) * C], dropout=ctx.pop(name + ".dropout", None),
                                       xadd=ctx["emb_all"][:, blk * C:(blk + 1) * C])
        ops.colsum(dgp, G[f"{name}.out_layers.0.weight"], defer=True); ops.colsum(dbp, G[f"{name}.out_layers.0.bias"], defer=True)
        # conv1 (in_layers.2): bias gradient = column sum of the embedding gradient slice (both are sum_hw dh)
        ops.colsum(demb_all[:, blk * C:(blk + 1) * C], G[f"{name}.in_layers.2.bias"], defer=True)
        self._wgrad(dh, a, 3, ops.NORMAL, G[f"{name}.in_layers.2.weight"])
        _, wd1 = self._packs[f"{name}.in_layers.2"]
        if two:
            ops.colsum(dout_sum, G[f"{name}.skip_connection.bias"], defer=True)
            self._wgrad(dout, srcs, 1, ops.NORMAL, G[f"{name}.skip_connection.weight"])
            _, wds = self._packs[f"{name}.skip_connection"]
        outs = []
        gw, gb = G[f"{name}.in_layers.0.weight"], G[f"{name}.in_layers.0.bias"]
        if two and C == 128:
            dskips = ops.conv1x1_pair(dout, wds, 2 * C)                      # both halves from one read of dout (128-channel blocks)
        elif two:
            dskips = tuple(self._conv([dout], wds, 2 * C, 1, ops.NORMAL, (H, W), n0=i * C) for i in range(2))
        else:
            dskips = (dout,)
        for i, s in enumerate(srcs):
            da = self._conv([dh], wd1, len(srcs) * C, 3, ops.NORMAL, (H, W), n0=i * C)
            dskip = dskips[i]
            add2 = extra_add[i] if extra_add is not None else None
            ssum = torch.empty((B, C), device=dout.device, dtype=torch.float32)
            ds, dgp, dbp = ops.gn_silu_bwd(da, s, P[f"{name}.in_layers.0.weight"][i * C:(i + 1) * C],
                                           P[f"{name}.in_layers.0.bias"][i * C:(i + 1) * C], stats1[i][0], stats1[i][1],
                                           dadd1=dskip, dadd2=add2, dxsum=ssum)
            ops.colsum(dgp, gw[i * C:(i + 1) * C], defer=True); ops.colsum(dbp, gb[i * C:(i + 1) * C], defer=True)
            outs.append((ds, ssum))
        return outs

    # ---- whole network ---------------------------------------------------------------------------------
    def prepare_forward(self, dev=None):
        """Everything a forward builds lazily on its first call after a weight update - the packed convolution weights, the frequency tables -
        enqueued on the CURRENT stream.  Callers that fan forwards out over several streams (the two-stream sampler, a captured graph) call this
        before the fork: the host-side freshness flags are cleared by whichever forward comes first, and a forward on another stream would then
        read pack buffers whose re-pack kernel it never waited for."""
        if self._packs_stale or self._packed_version != self._version_sum():
            self._repack()
        dev = self.flat_params.device if dev is None else dev
        self._freq_table(MAX_TIMESTEPS, dev)
        self._freq_table(4, dev)

    def forward_hip(self, x, logsnr, guide=None, cond_w=None, ctx=None):
        """x: NCHW fp32 [B, in_channels, H, W]; returns NCHW fp32.  ctx: dict receiving what backward needs."""
        if self._packs_stale or self._packed_version != self._version_sum():
            self._repack()
        P, C, T = self._pv, self.channels, self.act_dtype
        B, cin, H, W = x.shape
        if cin != self.in_channels or H % 4 or W % 4:
            raise ValueError(f"input {tuple(x.shape)}: need {self.in_channels} channels and H, W divisible by 4")
        x = ops.aligned(x.float())
        logsnr, guide, cond_w = ops.aligned(logsnr), ops.aligned(guide), ops.aligned(cond_w)
        # The embedding path (a dozen tiny fp32 kernels, ~ 0.2 ms of launch-bound work) is first needed by the SECOND GroupNorm of the first ResBlock:
        # with GMK_EMB_SIDE=1 it runs on the side stream beside the stem, the first GroupNorm and conv1, and is joined right before its first use
        if ops.EMB_SIDE and ops.WGRAD_STREAM and x.is_cuda and not torch.cuda.is_current_stream_capturing():
            box = []
            self._on_side(lambda: box.append(self._embed_fwd(logsnr, guide, cond_w, ctx)), tuple(t for t in (logsnr, guide, cond_w) if t is not None))
            emb_all = box[0]
            self._emb_pending = emb_all
        else:
            emb_all = self._embed_fwd(logsnr, guide, cond_w, ctx)
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        t0 = ops.stem_fwd(x, P["down.seq.0.conv.weight"], P["down.seq.0.conv.bias"], C, T)
        t1 = self._res_fwd("down.seq.1", [t0], emb_all, 0, ctx)
        t2 = self._res_fwd("down.seq.2", [t1], emb_all, 1, ctx)
        t3 = self._conv([t2], self._packs["down.seq.3.conv"][0], C, 3, ops.STRIDE2, (H2, W2),
                            bias=P["down.seq.3.conv.bias"])
        t4 = self._res_fwd("down.seq.4", [t3], emb_all, 2, ctx)
        t5 = self._res_fwd("down.seq.5", [t4], emb_all, 3, ctx)
        t6 = self._conv([t5], self._packs["down.seq.6.conv"][0], C, 3, ops.STRIDE2, (H4, W4),
                            bias=P["down.seq.6.conv.bias"])
        t7 = self._res_fwd("turn", [t6], emb_all, 4, ctx)
        if self.attention:
            t7 = self._attn_fwd(t7, ctx)
        u0r = self._res_fwd("up.seq.0.0", [t7, t6], emb_all, 5, ctx)
        u0 = self._upsample_conv("up.seq.0.1.conv", u0r, (H2, W2))
        u1 = self._res_fwd("up.seq.1", [u0, t5], emb_all, 6, ctx)
        u2 = self._res_fwd("up.seq.2", [u1, t4], emb_all, 7, ctx)
        u3r = self._res_fwd("up.seq.3.0", [u2, t3], emb_all, 8, ctx)
        u3 = self._upsample_conv("up.seq.3.1.conv", u3r, (H, W))
        u4 = self._res_fwd("up.seq.4", [u3, t2], emb_all, 9, ctx)
        u5 = self._res_fwd("up.seq.5", [u4, t1], emb_all, 10, ctx)
        u6 = self._res_fwd("up.seq.6", [u5, t0], emb_all, 11, ctx)
        ao, mo, ro = ops.gn_silu_fwd(u6, P["out.0.weight"], P["out.0.bias"], self._g1)
        out = ops.head_fwd(ao, P["out.2.weight"], P["out.2.bias"])
        if ctx is not None:
            ctx["net"] = (x, t2, t5, u0r, u3r, u6, ao, mo, ro)
            ctx["dims"] = (B, H, W)
        return out

    def grad_buckets(self):
        """Contiguous arena ranges [(start, end)] in the order their gradients become final during backward_hip:
        up.seq.4..out, up.seq.0..up.seq.3, down.seq.0..turn, then the embedding MLPs + the 12 emb_layers."""
        names = [n for n, _ in self._inventory]
        def rng(pred):
            return self.arena_range([n for n in names if pred(n)])
        late = ("up.seq.4", "up.seq.5", "up.seq.6", "out.")
        mid = ("up.seq.0", "up.seq.1", "up.seq.2", "up.seq.3")
        emb = ("time_embed", "cond_w_embed", "guide_embed")
        is_emb = lambda n: n.startswith(emb) or ".emb_layers." in n or n.endswith(".in_layers.2.bias")   # the arena's front block
        b0 = rng(lambda n: n.startswith(late) and not is_emb(n))
        b1 = rng(lambda n: n.startswith(mid) and not is_emb(n))
        b2 = rng(lambda n: (n.startswith("down.") or n.startswith("turn.") or n.startswith("attn.")) and not is_emb(n))
        b3 = rng(is_emb)
        buckets = [b0, b1, b2, b3]
        assert sorted(buckets)[0][0] == 0 and sorted(buckets)[-1][1] == self._arena_size
        srt = sorted(buckets)
        assert all(srt[i][1] == srt[i + 1][0] for i in range(3)), "gradient buckets must tile the arena"
        return buckets

    def backward_hip(self, ctx, dout, on_grads_ready=None, join_side_before_ready=True):
        """dout: NCHW fp32 gradient of the network output.  Fills `flat_grads` (overwrites every touched slice).
        on_grads_ready(k): called as soon as every kernel that writes bucket k of grad_buckets() has been ENQUEUED (overlapped gradient
        all-reduce).  A bucket is final only when the side stream's weight gradients are: with join_side_before_ready the current stream
        joins the side stream in front of every callback (a callback may then read the bucket in current-stream order); a consumer
        that orders its own stream behind BOTH streams (parallel.GradSync: the all-reduce runs on a third stream) passes False, and
        the data-gradient chain on the current stream never waits for the weight gradients."""
        if ops.WGRAD_CUS > 0 and ops.WGRAD_STREAM and dout.is_cuda and self._cu_part is None and not (
                torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
            full = ops.get_cu_limit()
            self._cu_part = (full, min(ops.WGRAD_CUS, full - 8))
            ops.set_cu_limit(full - self._cu_part[1])
            try:
                return self.backward_hip(ctx, dout, on_grads_ready, join_side_before_ready)
            finally:
                ops.set_cu_limit(full)
                self._cu_part = None
        ready = on_grads_ready if on_grads_ready is not None else (lambda k: None)
        join = self._join_side if (on_grads_ready is not None and join_side_before_ready) else (lambda: None)
        P, G, C, T = self._pv, self._gv, self.channels, self.compute_dtype
        B, H, W = ctx["dims"]
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        x, t2, t5, u0r, u3r, u6, ao, mo, ro = ctx["net"]
        dev = dout.device
        dout = ops.aligned(dout.float())
        demb_all = torch.empty((B, 12 * C), device=dev, dtype=torch.float32)

        # head: out.2 conv + out.0 GroupNorm/SiLU
        o = self._offsets["out.2.weight"]
        nhead = self.in_channels * C * 9
        assert self._offsets["out.2.bias"] == o + (nhead + 3) // 4 * 4 == o + nhead
        self._on_side(lambda: ops.head_wgrad(dout, ao, self.flat_grads[o:o + nhead + self.in_channels]), (dout, ao))
        dao = ops.head_dgrad(dout, P["out.2.weight"], T)
        s6 = torch.empty((B, C), device=dev, dtype=torch.float32)
        du6, dgp, dbp = ops.gn_silu_bwd(dao, u6, P["out.0.weight"], P["out.0.bias"], mo, ro, dxsum=s6)
        ops.colsum(dgp, G["out.0.weight"], defer=True); ops.colsum(dbp, G["out.0.bias"], defer=True)

        (du5, s5), (dt0a, _) = self._res_bwd("up.seq.6", ctx, du6, s6, demb_all, 11)
        (du4, s4), (dt1a, _) = self._res_bwd("up.seq.5", ctx, du5, s5, demb_all, 10)
        (du3, s3), (dt2a, _) = self._res_bwd("up.seq.4", ctx, du4, s4, demb_all, 9)
        ops.flush_colsums(); join()
        ready(0)
        # up.seq.3.1: nearest x2 + conv
        ops.colsum(s3, G["up.seq.3.1.conv.bias"], defer=True)
        self._upsample_wgrad("up.seq.3.1.conv", du3, u3r)
        du3r = self._upsample_dgrad("up.seq.3.1.conv", du3)
        s3r = ops.chansum(du3r)
        (du2, s2), (dt3a, _) = self._res_bwd("up.seq.3.0", ctx, du3r, s3r, demb_all, 8)
        (du1, s1), (dt4a, _) = self._res_bwd("up.seq.2", ctx, du2, s2, demb_all, 7)
        (du0, s0), (dt5a, _) = self._res_bwd("up.seq.1", ctx, du1, s1, demb_all, 6)
        ops.colsum(s0, G["up.seq.0.1.conv.bias"], defer=True)
        self._upsample_wgrad("up.seq.0.1.conv", du0, u0r)
        du0r = self._upsample_dgrad("up.seq.0.1.conv", du0)
        s0r = ops.chansum(du0r)
        (dt7, s7), (dt6a, _) = self._res_bwd("up.seq.0.0", ctx, du0r, s0r, demb_all, 5)
        if self.attention:
            dt7, s7 = self._attn_bwd(ctx, dt7)
        ops.flush_colsums(); join()
        ready(1)
        ((dt6, s6t),) = self._res_bwd("turn", ctx, dt7, s7, demb_all, 4, extra_add=[dt6a])
        # down.seq.6: stride-2 conv; its data gradient is the transposed gather
        ops.colsum(s6t, G["down.seq.6.conv.bias"], defer=True)
        self._wgrad(dt6, [t5], 3, ops.STRIDE2, G["down.seq.6.conv.weight"])
        dt5 = self._conv([dt6], self._packs["down.seq.6.conv"][1], C, 3, ops.TRANSPOSED2, (H2, W2), residual=dt5a)
        s5t = ops.chansum(dt5)
        ((dt4, s4t),) = self._res_bwd("down.seq.5", ctx, dt5, s5t, demb_all, 3, extra_add=[dt4a])
        ((dt3, s3t),) = self._res_bwd("down.seq.4", ctx, dt4, s4t, demb_all, 2, extra_add=[dt3a])
        ops.colsum(s3t, G["down.seq.3.conv.bias"], defer=True)
        self._wgrad(dt3, [t2], 3, ops.STRIDE2, G["down.seq.3.conv.weight"])
        dt2 = self._conv([dt3], self._packs["down.seq.3.conv"][1], C, 3, ops.TRANSPOSED2, (H, W), residual=dt2a)
        s2t = ops.chansum(dt2)
        ((dt1, s1t),) = self._res_bwd("down.seq.2", ctx, dt2, s2t, demb_all, 1, extra_add=[dt1a])
        ((dt0, s0t),) = self._res_bwd("down.seq.1", ctx, dt1, s1t, demb_all, 0, extra_add=[dt0a])
        ops.colsum(s0t, G["down.seq.0.conv.bias"], defer=True)
        self._on_side(lambda: ops.stem_wgrad(x, dt0, G["down.seq.0.conv.weight"]), (x, dt0))
        ops.flush_colsums(); join()
        ready(2)
        self._embed_bwd(ctx, demb_all)
        ops.flush_colsums()
        ready(3)
        self._join_side()          # the caller (optimiser step, next forward) continues on the current stream

    def zero_grad_arena(self):
        self.flat_grads.zero_()

    def forward(self, x, timesteps, guide=None, cond_w=None):
        """Reference signature (simple_unet.py:44).  Differentiable through SimpleUnetFunction when grad is enabled."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return SimpleUnetFunction.apply(self, x, timesteps, guide, cond_w, *self.parameters())
        return self.forward_hip(x, timesteps, guide, cond_w)


class SimpleUnetFunction(torch.autograd.Function):
    """torch.autograd bridge: forward = forward_hip, backward = backward_hip.  Parameter gradients are ACCUMULATED
    into `p.grad` (the arena views) like autograd would; there is no gradient w.r.t. x (the reference never needs it)."""

    @staticmethod
    def forward(ctx, net, x, timesteps, guide, cond_w, *params):
        store = {}
        out = net.forward_hip(x, timesteps, guide, cond_w, ctx=store)
        ctx.net, ctx.store = net, store
        return out

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        prev = net.flat_grads.clone()
        net.backward_hip(ctx.store, dout)
        net.flat_grads.add_(prev)
        # gradients were written in place into p.grad (views of the arena): return None for every input
        return (None,) * (5 + len(list(net.parameters())))
