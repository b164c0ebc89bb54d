"""Thin torch-tensor wrappers over the C ABI (include/gmk.h).  torch supplies device memory and streams only.

Every wrapper checks shapes/dtypes/contiguity on the host before launching (a kernel fault on this hardware can
take the whole node down) and enqueues on torch's current HIP stream.
"""
import math

import os

import torch

from ._lib import check, lib

F32, BF16, F16 = 0, 1, 2
# fp16 = forward activations / forward weight packs of the 16-bit mode; bf16 = gradients (and the all-bf16 A/B mode)
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_HALF = (torch.bfloat16, torch.float16)
NORMAL, STRIDE2, UPSAMPLE2, TRANSPOSED2 = 0, 1, 2, 3

# bench.py sets this to a list to collect (kernel name, start event, end event, algorithmic FLOPs, algorithmic HBM bytes, multiplied FLOPs) per
# launch of the convolution, weight-gradient, GroupNorm and stem / head kernels; events are recorded on the stream the kernel is launched on.
# "Algorithmic" FLOPs are the REFERENCE op's (9 taps per output pixel); "multiplied" are the products the kernel's MFMAs actually form - they
# differ for the sub-pixel forms (16 tap-products per low-resolution pixel where the reference op counts 36), and only the multiplied
# count may be set against the MFMA peak.
PROFILE = None


KERNEL_NAMES = {1: "conv_igemm_kernel", 2: "conv_igemm_dma_kernel", 3: "conv3x3_halo_kernel", 4: "conv3x3_halo_ws_kernel", 11: "conv_wgrad_kernel",
                12: "conv_wgrad_slots_kernel", 13: "conv_wgrad_slots_ws_kernel", 16: "conv_wgrad_subpixel_ws_kernel", 14: "conv1x1_pair_stream_kernel", 15: "conv1x1_wgrad_stream_kernel",
                # same binary as 4, launched on the zero-stuffed gradient of a stride-2 conv: 4x the algorithmic MFMA work by
                # construction, so the profile keeps it apart from the plain 3x3 convolutions
                5: "conv3x3_halo_ws_kernel[zero-stuffed transposed conv]",
                # conv2 of an up-path ResBlock with its 1x1 skip convolution folded in (K = 9 x 128 + 256): its own line in the profile
                7: "conv3x3_halo_ws_kernel[+1x1 skip]",
                # stride-2 data gradient where the halo kernel declines (fp32, small problems): four output-parity phase launches of the LDS-DMA kernel
                6: "conv_igemm_dma_kernel[stride-2 dgrad phases]",
                # sub-pixel (output-parity) forms: `Upsample` as four 2x2-tap parities on pre-summed weights, the stride-2 data gradient as 1 / 2 / 2 / 4 taps
                # the slot weight-gradient kernel on the four parity planes of a stride-2 convolution's input (round 6): its own line (4 - 16 MFMAs per step, not 36)
                17: "conv_wgrad_slots_ws_kernel[stride-2 planes]",
                8: "conv_subpixel_ws_kernel[upsample]", 9: "conv_subpixel_ws_kernel[transposed]", 10: "conv_subpixel_ws_kernel[upsample dgrad]",
                21: "gn_silu_fwd_reg_kernel", 22: "gn_silu_fwd_kernel", 23: "gn_silu_bwd_hybrid_kernel", 24: "gn_silu_bwd_kernel"}


def instantiation_key(name, dtype):
    """The template instantiation a profiled launch ran as, spelled like the keys of the counter records (tools/kernel_names.py:
    `conv3x3_halo_ws_kernel<f16>`, `<bf16>`, `<f16,+skip>`, `conv_subpixel_ws_kernel<bf16,upsample dgrad>` ...), so that a launch's own
    algorithmic bytes can stand beside its own counter bytes; None for kernels that have one instantiation per name."""
    if dtype not in _HALF:
        return None
    t = "f16" if dtype == torch.float16 else "bf16"
    if name == "conv3x3_halo_ws_kernel":
        return f"{name}<{t}>"
    if name == "conv3x3_halo_ws_kernel[+1x1 skip]":
        return f"conv3x3_halo_ws_kernel<{t},+skip>"
    if name == "conv_wgrad_slots_ws_kernel[stride-2 planes]":
        return "conv_wgrad_slots_ws_kernel<stride-2 planes>"
    if name.startswith("conv_subpixel_ws_kernel["):
        return f"conv_subpixel_ws_kernel<{t},{name[len('conv_subpixel_ws_kernel['):-1]}>"
    return None


def _nbytes(*tensors):
    """Algorithmic bytes of a launch - only evaluated when a profile is being collected (it sits on every launch's path)."""
    if PROFILE is None:
        return 0.0
    return float(sum(t.numel() * t.element_size() for t in tensors if t is not None))


class _Timed:
    """flops / nbytes: ALGORITHMIC work of the launch (each operand tensor read once, each result written once);
    fixed=True: `name` is the kernel's own name (no gmk_last_kernel lookup)."""

    def __init__(self, name, flops, nbytes=0.0, fixed=False, multiplied=None, dtype=None):
        self.on = PROFILE is not None
        if self.on:
            self.name, self.flops, self.nbytes, self.fixed = name, flops, nbytes, fixed
            self.multiplied = flops if multiplied is None else multiplied
            self.dtype = dtype
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.s.record()

    def __exit__(self, *exc):
        if self.on:
            self.e.record()
            name = self.name if self.fixed else KERNEL_NAMES.get(lib.gmk_last_kernel(), self.name)
            PROFILE.append((name, self.s, self.e, self.flops, self.nbytes, self.multiplied, instantiation_key(name, self.dtype)))


_IN_FLIGHT = {}


def throttle(max_in_flight=2):
    """Call once per step (train step, sampler iteration): records an event and waits for the one `max_in_flight` steps back.
    A step's host work (~4 ms of launches) is far shorter than its GPU work, so an unthrottled loop queues dozens of steps
    ahead; every tensor that crossed streams (`record_stream`: the side-stream weight gradients and skip convolutions) then
    stays unavailable to the caching allocator until the GPU catches up, the pool grows by those bytes PER QUEUED STEP, hits the
    device limit and falls into free-and-retry (measured at 3x64x64, B=1024: 305 instead of 117 ms per step, and 5 instead of
    28 sampler steps/s).  Two steps in flight keep the GPU fed and the pool at its steady size."""
    q = _IN_FLIGHT.setdefault(torch.cuda.current_device(), [])
    ev = torch.cuda.Event()
    ev.record()
    q.append(ev)
    if len(q) >= max_in_flight:
        q.pop(0).synchronize()


def dt_code(dtype):
    return _DT[dtype]


# The current HIP stream's raw handle.  torch.cuda.current_stream() builds a Stream object and walks torch's device-index helpers (an
# os.environ lookup among them) on every call - a quarter of the host time of a small-batch step with ~ 350 launches (a host profile of round 3 (docs/EXPERIMENTS.md));
# the raw getter is the same value through one C call.
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _s():
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if t.data_ptr() % 16:
        raise ValueError(f"{name}: data pointer not 16-byte aligned")
    return t


def get_cu_limit():
    return lib.gmk_get_cu_limit()


def set_cu_limit(n):
    """CUs the persistent grids launched from here on may occupy (read by the host at launch time)."""
    check(lib.gmk_set_cu_limit(int(n)), "set_cu_limit")


def aligned(t):
    """Contiguous, 16-byte aligned version of a caller-supplied tensor (a slice of a batch may start anywhere)."""
    if t is None:
        return None
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


def _f32(t, name):
    return _chk(t, torch.float32, name)


# ---- packing ---------------------------------------------------------------------------------------------
def pack_conv_weight(w, w_fwd, w_dgrad):
    _f32(w, "w")
    cout, cin, k, _ = w.shape
    ref = w_fwd if w_fwd is not None else w_dgrad
    for t in (w_fwd, w_dgrad):
        if t is not None:
            _chk(t, name="pack")
            assert t.numel() == w.numel()
    if w_fwd is not None and w_dgrad is not None and w_fwd.dtype != w_dgrad.dtype:      # fp16 forward pack + bf16 data-gradient pack
        pack_conv_weight(w, w_fwd, None)
        pack_conv_weight(w, None, w_dgrad)
        return
    check(lib.gmk_pack_conv_weight(_p(w), _p(w_fwd), _p(w_dgrad), cout, cin, k, _DT[ref.dtype], _s()), "pack_conv_weight")


def pack_conv_weights_multi(arena, packs, table, fwd_f16=None):
    """One launch for every convolution of the net.  table = (w_off, pack_off, cout, cin, ksize) host int lists: tensor e is
    arena[w_off[e]:...] (fp32 [cout][cin][k][k]) -> packs[pack_off[e]:...] = [w_fwd | w_dgrad]; fwd_f16[e] != 0: that entry's
    w_fwd half is written as fp16 (bf16 packs only)."""
    import ctypes
    _f32(arena, "arena"); _chk(packs, name="packs")
    n = len(table[0])
    arrs = [(ctypes.c_int * n)(*[int(v) for v in col]) for col in table]
    flags = (ctypes.c_int * n)(*[int(v) for v in fwd_f16]) if fwd_f16 is not None else None
    check(lib.gmk_pack_conv_weights_multi(_p(arena), _p(packs), n, *arrs, flags, _DT[packs.dtype], _s()), "pack_conv_weights_multi")


# ---- GroupNorm + SiLU ------------------------------------------------------------------------------------
# Let producing convolutions emit GroupNorm statistics from their epilogue (gmk_conv_igemm gn_stats).  OFF by default:
# measured on MI355X the DPP reductions in the conv epilogue cost as much as the statistics pass they save (27.25 vs
# 27.19 ms/step), and with it a sample's result depends (in the last bits) on which tile neighbours it had.
GN_STATS = False
GN_FUSE = os.environ.get("GMK_GN_FUSE", "1") != "0"            # inference: GroupNorm-apply + SiLU inside the consuming convolution (simple_unet._res_fwd)
# ... where it pays (round 3's fused-GroupNorm A/B, docs/EXPERIMENTS.md 7b.7, B x HW = 1 M pixels): at 64 x 64 the statistics-only launch + fused convolution take 1,497 us against
# 1,658 us for GroupNorm + convolution (-10 %); at 32 x 32 / 28 x 28 the producer waves' transform (224 VALU issue slots per K-step
# against the ~190 the consumer's MFMA stream leaves free on the shared SIMD) costs what the normalised tensor's round trip saved
# (758 vs 747 us, 351 vs 341 us), at 16 x 16 more.  GMK_GN_FUSE_MIN_HW overrides the threshold.
GN_FUSE_MIN_HW = int(os.environ.get("GMK_GN_FUSE_MIN_HW", "2048"))
# `skip_connection(x) + h` of the up-path ResBlocks as one launch (gmk_conv3x3_skipfold; GMK_SKIP_FOLD=0: the 1x1 convolution as its own
# launch + residual epilogue, the path of rounds 1-3, kept for A/B).  GMK_SKIP_FOLD_FUSE=fuse: where inference could instead apply conv2's
# GroupNorm inside the convolution (64-pixel rows), prefer that to the fold (the two do not combine yet)
SKIP_FOLD = os.environ.get("GMK_SKIP_FOLD", "1") != "0"
SKIP_FOLD_OVER_FUSE = os.environ.get("GMK_SKIP_FOLD_FUSE", "fold") != "fuse"
EMB_SIDE = os.environ.get("GMK_EMB_SIDE", "1") == "1"          # the embedding path of a forward on the side stream (simple_unet.forward_hip; round 5: +0.7 % DDIM steps/s, +0.2 % train)
FWD_SIDE = os.environ.get("GMK_FWD_SIDE", "0") == "1"          # forward 1x1 skip convolutions on the side stream (simple_unet._res_fwd)
WGRAD_CUS = int(os.environ.get("GMK_WGRAD_CUS", "0"))           # > 0: the side stream's persistent grids take this many CUs, the data-gradient chain's the rest
WGRAD_STREAM = os.environ.get("GMK_WGRAD_STREAM", "1") != "0"    # weight gradients on a side stream beside the data-gradient chain (simple_unet._wgrad)


def _xadd_stride(xadd, B, C):
    """xadd: optional fp32 [B, C] view (unit column stride, any row stride) added to x per (sample, channel) on load."""
    if xadd is None:
        return 0
    assert xadd.dtype == torch.float32 and xadd.shape == (B, C) and xadd.stride(1) == 1 and xadd.is_cuda
    return xadd.stride(0)


def gn_silu_fwd(x, gamma, beta, groups, eps=1e-5, dropout=None, xadd=None):
    """x NHWC [B,H,W,C] -> (y, mean[B,G], rstd[B,G]).  If the convolution that produced x attached its partial
    GroupNorm statistics (x._gn_stats), the statistics pass over x is skipped.  dropout = (p, seed, offset): nn.Dropout(p)
    behind the SiLU with the mask `rng_uniform(x.shape, seed, offset) >= p` (pass the same triple to gn_silu_bwd)."""
    _chk(x, name="x"); _f32(gamma, "gamma"); _f32(beta, "beta")
    B, H, W, C = x.shape
    assert gamma.numel() == C and beta.numel() == C
    y = torch.empty_like(x)
    # groups < 0: -groups channels per group (any size up to 16: the zero-padded widths 96, 160, 192, 224), ceil(C / -groups) groups
    ncol = groups if groups > 0 else (C - groups - 1) // -groups
    mean = torch.empty((B, ncol), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    mean._gn_groups = groups                    # gn_silu_bwd reads the group layout from the statistics it is handed
    st = getattr(x, "_gn_stats", None)
    part, tp, nt = st if st is not None else (None, 0, 0)
    dp, dseed, doff = dropout if dropout is not None else (0.0, 0, 0)
    xs = _xadd_stride(xadd, B, C)
    with _Timed("gn_silu_fwd", 0.0, _nbytes(x, y)):
        check(lib.gmk_gn_silu_fwd(_p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(rstd), B, H * W, C, groups, eps, _p(part), tp, nt,
                                  float(dp), int(dseed), int(doff), _p(xadd), xs, _DT[x.dtype], _s()), "gn_silu_fwd")
    return y, mean, rstd


def gn_stats(x, gamma, beta, groups, tab_scale, tab_shift, eps=1e-5, xadd=None):
    """Statistics-only GroupNorm of x NHWC [B,H,W,C]: fills columns [0, C) of the fp32 table views tab_scale / tab_shift
    ([B, C] slices of a [B, Ctot] table: unit column stride, row stride Ctot) with the affine form of the normalisation,
    y = silu(x * scale + shift), for a convolution that applies it itself (conv_igemm gn=).  -> (mean, rstd)"""
    _chk(x, name="x"); _f32(gamma, "gamma"); _f32(beta, "beta")
    assert x.dtype in _HALF
    B, H, W, C = x.shape
    for t in (tab_scale, tab_shift):
        assert t.dtype == torch.float32 and t.shape == (B, C) and t.stride(1) == 1 and t.is_cuda and t.data_ptr() % 16 == 0
    assert tab_scale.stride(0) == tab_shift.stride(0)
    mean = torch.empty((B, groups), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    xs = _xadd_stride(xadd, B, C)
    with _Timed("gn_stats", 0.0, _nbytes(x)):
        check(lib.gmk_gn_stats(_p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(tab_scale), _p(tab_shift), tab_scale.stride(0), B, H * W, C,
                               groups, eps, _p(xadd), xs, _DT[x.dtype], _s()), "gn_stats")
    return mean, rstd


def conv_gn_fusable(srcs, cout=128):
    """Can conv_igemm(srcs, ..., 3, NORMAL, gn=...) apply the GroupNorm of its (16-bit) sources itself?"""
    s0 = srcs[0]
    c1 = srcs[1].shape[3] if len(srcs) > 1 else 0
    B, H, W, c0 = s0.shape
    return s0.dtype in _HALF and H * W >= GN_FUSE_MIN_HW and bool(lib.gmk_conv_gn_fusable(B, H, W, c0, c1, cout))


def gn_silu_bwd(dy, x, gamma, beta, mean, rstd, dadd1=None, dadd2=None, dxsum=None, dropout=None, xadd=None):
    """-> (dx, dgamma_part[B,C], dbeta_part[B,C]); dxsum (optional fp32 [B, >=C] view with row stride) is filled.
    x (the saved forward input) may be fp16 next to bf16 gradients (dy, addends, dx)."""
    _chk(x, name="x"); _chk(dy, name="dy")
    assert dy.shape == x.shape and (x.dtype == dy.dtype or (x.dtype == torch.float16 and dy.dtype == torch.bfloat16)), (x.dtype, dy.dtype)
    for t in (dadd1, dadd2):
        if t is not None:
            _chk(t, dy.dtype, "dadd"); assert t.shape == x.shape
    B, H, W, C = x.shape
    G = getattr(mean, "_gn_groups", mean.shape[1])
    dx = torch.empty_like(dy)
    dgp = torch.empty((B, C), device=x.device, dtype=torch.float32)
    dbp = torch.empty_like(dgp)
    stride = 0
    if dxsum is not None:
        assert dxsum.dtype == torch.float32 and dxsum.shape == (B, C) and dxsum.stride(1) == 1
        stride = dxsum.stride(0)
    dp, dseed, doff = dropout if dropout is not None else (0.0, 0, 0)
    xs = _xadd_stride(xadd, B, C)
    with _Timed("gn_silu_bwd", 0.0, _nbytes(dy, x, dadd1, dadd2, dx)):
        check(lib.gmk_gn_silu_bwd(_p(dy), _p(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dadd1), _p(dadd2), _p(dx),
                                  _p(dgp), _p(dbp), _p(dxsum), stride, B, H * W, C, G, float(dp), int(dseed), int(doff),
                                  _p(xadd), xs, _DT[dy.dtype], _DT[x.dtype], _s()), "gn_silu_bwd")
    return dx, dgp, dbp


def cast16(x, dtype):
    """fp16 <-> bf16 copy of a contiguous tensor (numel a multiple of 8)."""
    _chk(x, name="x")
    assert x.dtype in _HALF and dtype in _HALF
    if x.dtype == dtype:
        return x
    out = torch.empty_like(x, dtype=dtype)
    check(lib.gmk_cast16(_p(x), _p(out), x.numel(), _DT[x.dtype], _DT[dtype], _s()), "cast16")
    return out


def chansum(x, out=None):
    _chk(x, name="x")
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, C), device=x.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.shape == (B, C) and out.stride(1) == 1
    with _Timed("chansum_kernel", 0.0, _nbytes(x), fixed=True):
        check(lib.gmk_chansum(_p(x), _p(out), out.stride(0), B, H * W, C, _DT[x.dtype], _s()), "chansum")
    return out


_PENDING_COLSUMS = []      # (part, out) pairs deferred into gmk_colsum_multi launches of up to 16


def colsum(part, out, accumulate=False, defer=False):
    """out[c] (+)= sum_r part[r, c]; part fp32 [R, C] with unit column stride, out fp32 [C] (any contiguous view).
    defer=True queues the reduction (same stream order) until flush_colsums(); `part` is kept alive until then."""
    assert part.dtype == torch.float32 and part.dim() == 2 and part.stride(1) == 1 and part.is_cuda
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == part.shape[1]
    if defer and not accumulate:
        _PENDING_COLSUMS.append((part, out))
        if len(_PENDING_COLSUMS) >= 16:
            flush_colsums()
        return out
    check(lib.gmk_colsum(_p(part), part.stride(0), _p(out), part.shape[0], part.shape[1], int(accumulate), _s()), "colsum")
    return out


def flush_colsums():
    import ctypes
    while _PENDING_COLSUMS:
        batch = _PENDING_COLSUMS[:16]
        del _PENDING_COLSUMS[:16]
        n = len(batch)
        parts = (ctypes.c_void_p * n)(*[b[0].data_ptr() for b in batch])
        outs = (ctypes.c_void_p * n)(*[b[1].data_ptr() for b in batch])
        strides = (ctypes.c_int64 * n)(*[b[0].stride(0) for b in batch])
        Rs = (ctypes.c_int * n)(*[b[0].shape[0] for b in batch])
        Cs = (ctypes.c_int * n)(*[b[0].shape[1] for b in batch])
        check(lib.gmk_colsum_multi(n, ctypes.cast(parts, ctypes.c_void_p), ctypes.cast(strides, ctypes.c_void_p),
                                   ctypes.cast(outs, ctypes.c_void_p), ctypes.cast(Rs, ctypes.c_void_p),
                                   ctypes.cast(Cs, ctypes.c_void_p), _s()), "colsum_multi")


def sumpool2x2(x):
    _chk(x, name="x")
    B, H2, W2, C = x.shape
    assert H2 % 2 == 0 and W2 % 2 == 0
    y = torch.empty((B, H2 // 2, W2 // 2, C), device=x.device, dtype=x.dtype)
    with _Timed("sumpool2x2_kernel", 0.0, _nbytes(x, y), fixed=True):
        check(lib.gmk_sumpool2x2(_p(x), _p(y), B, H2 // 2, W2 // 2, C, _DT[x.dtype], _s()), "sumpool2x2")
    return y


# ---- convolutions ----------------------------------------------------------------------------------------
def out_size(mode, hs, ws):
    if mode == NORMAL:
        return hs, ws
    if mode == STRIDE2:
        return (hs - 1) // 2 + 1, (ws - 1) // 2 + 1
    if mode == UPSAMPLE2:
        return 2 * hs, 2 * ws
    raise ValueError(mode)


def conv_igemm(srcs, w, w_rows, ksize, mode, out_hw, n0=0, cout=128, bias=None, emb=None, residual=None, gn_stats=False, gn=None):
    """srcs: list of 1-2 NHWC tensors (same B,H,W); w: packed weights [taps][w_rows][sum C]; -> out NHWC [B,ho,wo,cout]."""
    s0 = _chk(srcs[0], name="src0")
    s1 = _chk(srcs[1], s0.dtype, "src1") if len(srcs) > 1 else None
    B, hs, ws, c0 = s0.shape
    c1 = 0
    if s1 is not None:
        assert s1.shape[:3] == s0.shape[:3]
        c1 = s1.shape[3]
    _chk(w, s0.dtype, "w")
    assert w.numel() == ksize * ksize * w_rows * (c0 + c1), (w.numel(), ksize, w_rows, c0, c1)
    ho, wo = out_hw
    out = torch.empty((B, ho, wo, cout), device=s0.device, dtype=s0.dtype)
    emb_stride = 0
    if emb is not None:
        assert emb.dtype == torch.float32 and emb.shape == (B, cout) and emb.stride(1) == 1 and emb.data_ptr() % 16 == 0
        emb_stride = emb.stride(0)
        assert emb_stride % 4 == 0
    if bias is not None:
        _f32(bias, "bias"); assert bias.numel() == cout
    if residual is not None:
        _chk(residual, s0.dtype, "residual"); assert residual.shape == out.shape
    gsc = gsh = None
    gstride = 0
    if gn is not None:          # (scale, shift) fp32 [B, c0 + c1] tables of gn_stats: the sources are raw, the kernel normalises them
        gsc, gsh = gn
        for t in (gsc, gsh):
            _f32(t, "gn table"); assert t.shape == (B, c0 + c1)
        gstride = c0 + c1
    mpix = B * hs * ws if mode == TRANSPOSED2 else B * ho * wo     # algorithmic work: that of the stride-2 conv
    part, tp, nt = None, 0, 0
    if gn_stats and GN_STATS and ksize == 3 and mode in (NORMAL, UPSAMPLE2) and s0.dtype == torch.bfloat16 and 4 <= wo <= 254:      # (bf16 only)
        r = 256 // wo
        tp, nt = r * wo, (B * ho + r - 1) // r
        part = torch.empty(nt * 8 * 2 * (cout // 4) * 2, device=s0.device, dtype=torch.float32)
    with _Timed("conv_igemm", 2.0 * mpix * cout * (c0 + c1) * ksize * ksize, _nbytes(s0, s1, residual, out), dtype=s0.dtype):
        check(lib.gmk_conv_igemm(_p(s0), _p(s1), c0, c1, B, hs, ws, ho, wo, ksize, mode, _p(w), w_rows, n0, cout,
                                 _p(bias), _p(emb), emb_stride, _p(residual), _p(out), cout, _p(part),
                                 part.numel() * 4 if part is not None else 0, _p(gsc), _p(gsh), gstride, _DT[s0.dtype], _s()),
              "conv_igemm")
    if part is not None and lib.gmk_last_kernel() == 3 and ho * wo >= 32:      # only the 8-compute-wave halo kernel emits statistics
        out._gn_stats = (part, tp, nt)     # consumed by gn_silu_fwd(out, ...)
    return out


SUBPIXEL_UPSAMPLE, SUBPIXEL_TRANSPOSED, SUBPIXEL_UPSAMPLE_DGRAD = 0, 1, 2


def conv_subpixel_ok(B, H, W, c, dtype, cout=128):
    """True if the x2 resampling convolutions over the LOW-resolution grid B x H x W run in their sub-pixel form (gmk_conv_subpixel)."""
    return dtype in _HALF and bool(lib.gmk_conv_subpixel_ok(B, H, W, c, cout, _DT[dtype]))


def pack_upsample_weight(w, out, out_dgrad=None):
    """w: fp32 [Cout][Cin][3][3] (a contiguous arena view) -> out: 16 x Cout x Cin elements, the pre-summed 2x2-tap matrices of the sub-pixel
    `Upsample` (reference simple_unet.py:112-122); out_dgrad: their transposes [16][Cin][Cout] for its data gradient.  Either may be None."""
    _f32(w, "w")
    cout, cin = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 3) and w.is_contiguous()
    for t in (out, out_dgrad):
        if t is not None:
            _chk(t, name="pack"); assert t.numel() == 16 * cout * cin and t.dtype in _HALF
    ref = out if out is not None else out_dgrad
    check(lib.gmk_pack_upsample_weight(_p(w), _p(out), _p(out_dgrad), cout, cin, _DT[ref.dtype],
                                       _DT[(out_dgrad if out_dgrad is not None else ref).dtype], _s()), "pack_upsample_weight")
    return out


def conv_subpixel(src, w, w_rows, mode, bias=None, residual=None, n0=0, cout=128):
    """SUBPIXEL_UPSAMPLE: low-resolution NHWC src [B,H,W,128] -> conv3x3(nearest x2 (src)) [B,2H,2W,cout], w = the forward pack of
    pack_upsample_weight.  SUBPIXEL_TRANSPOSED: the data gradient of a 3x3 stride-2 convolution (src = its output gradient, w = its ordinary
    data-gradient pack) -> [B,2H,2W,cout].  SUBPIXEL_UPSAMPLE_DGRAD: src = the HIGH-resolution output gradient [B,2H,2W,128] of the upsampled
    convolution -> its input gradient [B,H,W,cout] (w = the out_dgrad pack): sumpool2x2(dgrad3x3(src)) in one launch."""
    s0 = _chk(src, name="src")
    B, H, W, c = s0.shape
    down = mode == SUBPIXEL_UPSAMPLE_DGRAD
    if down:
        assert H % 2 == 0 and W % 2 == 0
        H, W = H // 2, W // 2                  # the C ABI takes the LOW-resolution grid in every mode
    _chk(w, s0.dtype, "w")
    taps = 9 if mode == SUBPIXEL_TRANSPOSED else 16
    assert w.numel() == taps * w_rows * c, (w.numel(), taps, w_rows, c)
    out = torch.empty((B, H, W, cout) if down else (B, 2 * H, 2 * W, cout), device=s0.device, dtype=s0.dtype)
    if bias is not None:
        _f32(bias, "bias"); assert bias.numel() == cout
    if residual is not None:
        _chk(residual, s0.dtype, "residual"); assert residual.shape == out.shape
    # algorithmic work: that of the reference's op (9 taps per HIGH-resolution pixel for the upsampled convolution and its data gradient,
    # 9 per LOW-resolution pixel for the transposed one)
    mpix = B * H * W if mode == SUBPIXEL_TRANSPOSED else B * 4 * H * W
    with _Timed("conv_subpixel", 2.0 * mpix * cout * c * 9, _nbytes(s0, residual, out), multiplied=2.0 * B * H * W * cout * c * taps, dtype=s0.dtype):
        check(lib.gmk_conv_subpixel(_p(s0), B, H, W, c, _p(w), w_rows, n0, cout, mode, _p(bias), _p(residual), _p(out), cout,
                                    _DT[s0.dtype], _s()), "conv_subpixel")
    return out


def conv_skipfold_ok(src, skips):
    """True if conv3x3(src) + conv1x1(cat(skips)) of these shapes runs as one launch (gmk_conv3x3_skipfold)."""
    if not SKIP_FOLD or len(skips) != 2 or src.dtype not in (torch.bfloat16, torch.float16):
        return False
    B, H, W, c0 = src.shape
    return skips[0].shape[3] == skips[1].shape[3] and bool(lib.gmk_conv3x3_skipfold_ok(B, H, W, c0, skips[0].shape[3], 128))


def conv3x3_skipfold(src, w, bias, skips, wsk, bias_sk, cout=128):
    """out = conv3x3(src; w) + bias + conv1x1(cat(skips); wsk) + bias_sk in one launch: the `skip_connection(x) + h` of the up-path
    ResBlocks (reference simple_unet.py:174-186) without the skip output ever reaching HBM."""
    s0 = _chk(src, name="src")
    k0, k1 = _chk(skips[0], s0.dtype, "skip0"), _chk(skips[1], s0.dtype, "skip1")
    B, H, W, c0 = s0.shape
    cs = k0.shape[3]
    assert k0.shape == k1.shape == (B, H, W, cs)
    _chk(w, s0.dtype, "w"); _chk(wsk, s0.dtype, "wsk")
    assert w.numel() == 9 * cout * c0 and wsk.numel() == cout * 2 * cs, (w.numel(), wsk.numel())
    _f32(bias, "bias"); _f32(bias_sk, "bias_sk")
    assert bias.numel() == cout and bias_sk.numel() == cout
    out = torch.empty((B, H, W, cout), device=s0.device, dtype=s0.dtype)
    with _Timed("conv_igemm", 2.0 * B * H * W * cout * (9 * c0 + 2 * cs), _nbytes(s0, k0, k1, out), dtype=s0.dtype):
        check(lib.gmk_conv3x3_skipfold(_p(s0), c0, B, H, W, _p(w), cout, 0, cout, _p(bias), _p(k0), _p(k1), cs, _p(wsk), cout, 0,
                                       _p(bias_sk), _p(out), cout, _DT[s0.dtype], _s()), "conv3x3_skipfold")
    return out


def conv1x1_pair(src, w, w_rows, n0=0):
    """Two 128-channel 1x1 convolutions of one source (packed weight rows n0.. and n0+128..) from one pass over it: the skip
    connection's data gradient for both halves of a concatenated input.  -> (out_a, out_b), NHWC [B,H,W,128] each."""
    s0 = _chk(src, name="src")
    B, H, W, c = s0.shape
    _chk(w, s0.dtype, "w")
    assert w.numel() == w_rows * c and n0 + 256 <= w_rows
    oa = torch.empty((B, H, W, 128), device=s0.device, dtype=s0.dtype)
    ob = torch.empty_like(oa)
    with _Timed("conv_igemm", 2.0 * B * H * W * 256 * c, _nbytes(s0, oa, ob)):
        check(lib.gmk_conv1x1_pair(_p(s0), c, B, H, W, _p(w), w_rows, n0, _p(oa), _p(ob), _DT[s0.dtype], _s()), "conv1x1_pair")
    return oa, ob


_WS = {}


def _workspace(nbytes, device, tag="conv"):
    key = (tag, device.index, _s())
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        _WS[key] = ws
    return ws


def conv_wgrad(dy, srcs, ksize, mode, dw):
    """dw (fp32 [cout][sum C][k][k], a contiguous view into the gradient arena) = weight gradient."""
    _chk(dy, name="dy")
    s0 = _chk(srcs[0], name="src0")
    s1 = _chk(srcs[1], s0.dtype, "src1") if len(srcs) > 1 else None
    assert s0.dtype == dy.dtype or (s0.dtype == torch.float16 and dy.dtype == torch.bfloat16), (s0.dtype, dy.dtype)
    B, hs, ws_, c0 = s0.shape
    c1 = s1.shape[3] if s1 is not None else 0
    _, ho, wo, cout = dy.shape
    assert dy.shape[0] == B
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * (c0 + c1) * ksize * ksize
    need = lib.gmk_conv_wgrad_workspace_bytes(B * ho * wo, ksize * ksize, cout, c0 + c1)
    assert need > 0
    wsbuf = _workspace(need, dy.device)
    with _Timed("conv_wgrad", 2.0 * B * ho * wo * cout * (c0 + c1) * ksize * ksize, _nbytes(dy, s0, s1)):
        check(lib.gmk_conv_wgrad(_p(dy), cout, _p(s0), _p(s1), c0, c1, B, hs, ws_, ho, wo, ksize, mode, _p(dw), cout,
                                 _p(wsbuf), wsbuf.numel(), _DT[dy.dtype], _DT[s0.dtype], _s()), "conv_wgrad")
    return dw


def conv_wgrad_subpixel_ok(B, H, W, c, dy_dtype, cout=128):
    """True if the weight gradient of `Upsample` over the LOW-resolution grid B x H x W runs in its sub-pixel form (gmk_conv_wgrad_subpixel)."""
    return dy_dtype == torch.bfloat16 and bool(lib.gmk_conv_wgrad_subpixel_ok(B, H, W, c, cout))


def conv_wgrad_subpixel(dy, x, dw):
    """dw (fp32 [cout][cin][3][3], a contiguous arena view) = weight gradient of conv3x3(nearest x2 (x)) given the high-resolution output
    gradient dy [B,2H,2W,cout] (bf16) and the saved low-resolution input x [B,H,W,cin] (bf16 / fp16)."""
    _chk(dy, torch.bfloat16, "dy")
    _chk(x, name="x")
    assert x.dtype in _HALF
    B, H, W, cin = x.shape
    cout = dy.shape[3]
    assert dy.shape == (B, 2 * H, 2 * W, cout)
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * cin * 9
    need = lib.gmk_conv_wgrad_subpixel_workspace_bytes(B, H, W, cin, cout)
    assert need > 0
    wsbuf = _workspace(need, dy.device)
    with _Timed("conv_wgrad", 2.0 * B * 4 * H * W * cout * cin * 9, _nbytes(dy, x), multiplied=2.0 * B * H * W * cout * cin * 16):
        check(lib.gmk_conv_wgrad_subpixel(_p(dy), cout, _p(x), B, H, W, cin, cout, _p(dw), _p(wsbuf), wsbuf.numel(), _DT[dy.dtype],
                                          _DT[x.dtype], _s()), "conv_wgrad_subpixel")
    return dw


def _expand_name(dtype, cs, W):
    """Profile label of the stem forward / head data gradient launch (smallconv.hip launch_expand: 16-bit results of <= 3 image channels run tiled)."""
    return "expand3x3_tile16_kernel" if dtype in (torch.float16, torch.bfloat16) and cs <= 3 and W <= 128 else "expand3x3_mfma_kernel"


def _wgrad3_name(dtype, cs, W):
    """Profile label of the stem / head weight-gradient launch (smallconv.hip launch_wgrad: 16-bit tensors run tiled)."""
    return "wgrad3x3_tile16_kernel" if dtype in (torch.float16, torch.bfloat16) and cs <= 3 and W <= 128 and W % 2 == 0 else "wgrad3x3_mfma_kernel"


def stem_fwd(x, w, bias, C, dtype):
    _f32(x, "x"); _f32(w, "w"); _f32(bias, "bias")
    B, cin, H, W = x.shape
    assert w.shape == (C, cin, 3, 3)
    y = torch.empty((B, H, W, C), device=x.device, dtype=dtype)
    with _Timed(_expand_name(dtype, cin, W), 2.0 * B * H * W * C * cin * 9, _nbytes(x, y), fixed=True):
        check(lib.gmk_stem_fwd(_p(x), _p(w), _p(bias), _p(y), B, cin, H, W, C, _DT[dtype], _s()), "stem_fwd")
    return y


def stem_wgrad(x, dy, dw):
    _f32(x, "x"); _chk(dy, name="dy")
    B, cin, H, W = x.shape
    C = dy.shape[3]
    nb = lib.gmk_stem_wgrad_blocks(B * H * W)
    part = torch.empty((nb, C * cin * 9), device=x.device, dtype=torch.float32)
    with _Timed(_wgrad3_name(dy.dtype, cin, W), 2.0 * B * H * W * C * cin * 9, _nbytes(x, dy), fixed=True):
        check(lib.gmk_stem_wgrad(_p(x), _p(dy), _p(part), B, cin, H, W, C, _DT[dy.dtype], _s()), "stem_wgrad")
    return colsum(part, dw)


def head_fwd(a, w, bias):
    _chk(a, name="a"); _f32(w, "w"); _f32(bias, "bias")
    B, H, W, C = a.shape
    cout = w.shape[0]
    assert w.shape == (cout, C, 3, 3)
    out = torch.empty((B, cout, H, W), device=a.device, dtype=torch.float32)
    with _Timed("head_fwd_mfma_kernel", 2.0 * B * H * W * C * cout * 9, _nbytes(a, out), fixed=True):
        check(lib.gmk_head_fwd(_p(a), _p(w), _p(bias), _p(out), B, cout, H, W, C, _DT[a.dtype], _s()), "head_fwd")
    return out


def head_dgrad(dout, w, dtype):
    _f32(dout, "dout"); _f32(w, "w")
    B, cout, H, W = dout.shape
    C = w.shape[1]
    da = torch.empty((B, H, W, C), device=dout.device, dtype=dtype)
    with _Timed(_expand_name(dtype, cout, W), 2.0 * B * H * W * C * cout * 9, _nbytes(dout, da), fixed=True):
        check(lib.gmk_head_dgrad(_p(dout), _p(w), _p(da), B, cout, H, W, C, _DT[dtype], _s()), "head_dgrad")
    return da


def head_wgrad(dout, a, dwb):
    """dwb: contiguous fp32 view of [weight (cout*C*9) | bias (cout)] in the gradient arena."""
    _f32(dout, "dout"); _chk(a, name="a")
    B, cout, H, W = dout.shape
    C = a.shape[3]
    n = cout * C * 9 + cout
    assert dwb.numel() == n
    nb = lib.gmk_head_wgrad_blocks(B * H * W)
    part = torch.empty((nb, n), device=a.device, dtype=torch.float32)
    with _Timed(_wgrad3_name(a.dtype, cout, W), 2.0 * B * H * W * C * cout * 9, _nbytes(dout, a), fixed=True):
        check(lib.gmk_head_wgrad(_p(dout), _p(a), _p(part), B, cout, H, W, C, _DT[a.dtype], _s()), "head_wgrad")
    return colsum(part, dwb)


# ---- embedding path --------------------------------------------------------------------------------------
def timestep_freqs(max_period, device):
    """simple_unet.py:215-219 evaluated with the same torch fp32 ops (host), as a 32-entry table."""
    half = 32
    return torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(device)


def timestep_embedding(t, freqs):
    _f32(t, "t"); _f32(freqs, "freqs")
    B = t.numel()
    out = torch.empty((B, 64), device=t.device, dtype=torch.float32)
    check(lib.gmk_timestep_embedding(_p(t), _p(freqs), _p(out), B, _s()), "timestep_embedding")
    return out


def guide_onehot(guide):
    _chk(guide, torch.int64, "guide")
    B = guide.numel()
    onehot = torch.empty((B, 10), device=guide.device, dtype=torch.float32)
    keep = torch.empty((B,), device=guide.device, dtype=torch.float32)
    check(lib.gmk_guide_onehot(_p(guide), _p(onehot), _p(keep), B, _s()), "guide_onehot")
    return onehot, keep


def label_drop(y, p, seed, offset):
    """In place: y[b] = -1 where rng_uniform((B,), seed, offset)[b] < p (diffusion_model.py:67)."""
    _chk(y, torch.int64, "y")
    check(lib.gmk_label_drop(_p(y), y.numel(), float(p), int(seed), int(offset), _s()), "label_drop")
    return y


def mean(x):
    """0-dim fp32 mean of a contiguous fp32 vector (fixed summation order)."""
    _f32(x, "x")
    out = torch.empty((), device=x.device, dtype=torch.float32)
    check(lib.gmk_mean(_p(x), x.numel(), _p(out), _s()), "mean")
    return out


def gemm(A, B, out=None, bias=None, rowscale=None, silu_a=False, silu_b=False, accumulate=False, bias2=None):
    """out[M,N] (+)= rowscale * (bias + bias2 + fa(A) @ fb(B)) for 2-D fp32 views A [M,K], B [K,N] with arbitrary strides."""
    assert A.dtype == torch.float32 and B.dtype == torch.float32 and A.is_cuda and B.is_cuda
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    if out is None:
        assert not accumulate
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    for bv in (bias, bias2):
        if bv is not None:
            assert bv.dtype == torch.float32 and bv.numel() == N and bv.is_contiguous()
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rowscale.numel() == M and rowscale.is_contiguous()
    need = lib.gmk_gemm_f32_workspace_bytes(M, N, K)
    wsb = _workspace(need, A.device, "gemm") if need else None
    check(lib.gmk_gemm_f32(_p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(out), out.stride(0), M, N, K,
                           _p(bias), _p(bias2), _p(rowscale), int(silu_a) | (int(silu_b) << 1), int(accumulate), _p(wsb), need, _s()),
          "gemm_f32")
    return out


def silu_bwd(dpost, pre, rowscale=None):
    _f32(dpost, "dpost"); _f32(pre, "pre")
    assert dpost.shape == pre.shape
    out = torch.empty_like(pre)
    check(lib.gmk_silu_bwd(_p(dpost), _p(pre), _p(rowscale), _p(out), pre.numel(), pre.shape[-1], _s()), "silu_bwd")
    return out


def scale_rows(x, rowscale):
    _f32(x, "x"); _f32(rowscale, "rowscale")
    out = torch.empty_like(x)
    check(lib.gmk_scale_rows(_p(x), _p(rowscale), _p(out), x.numel(), x.shape[-1], _s()), "scale_rows")
    return out


# ---- diffusion algebra -----------------------------------------------------------------------------------
def rng_normal(shape, seed, offset, device):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(lib.gmk_rng_normal(_p(out), out.numel(), seed, offset, _s()), "rng_normal")
    return out


def rng_uniform(shape, seed, offset, device):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(lib.gmk_rng_uniform(_p(out), out.numel(), seed, offset, _s()), "rng_uniform")
    return out


def q_sample(x, eps, u):
    _f32(x, "x"); _f32(eps, "eps"); _f32(u, "u")
    B = x.shape[0]
    n = x.numel() // B
    assert eps.shape == x.shape and u.numel() == B
    logsnr = torch.empty((B,), device=x.device, dtype=torch.float32)
    z = torch.empty_like(x)
    check(lib.gmk_q_sample(_p(x), _p(eps), _p(u), _p(logsnr), _p(z), B, n, _s()), "q_sample")
    return logsnr, z


MEAN_TYPES = {"v": 0, "eps": 1, "x": 2}      # what the network output parametrises (gaussian_diffusion.py:58-73)


def v_loss(v, z, x, eps, logsnr, grad_scale=None, loss_type=0, mean_type="v"):
    """-> (loss_b, x_mse, eps_mse, dv or None).  loss_type 0 'snr_trunc', 1 'snr'."""
    for t, nm in ((v, "v"), (z, "z"), (x, "x"), (eps, "eps"), (logsnr, "logsnr")):
        _f32(t, nm)
    B = x.shape[0]
    n = x.numel() // B
    assert v.shape == x.shape == z.shape == eps.shape and logsnr.numel() == B
    loss_b = torch.empty((B,), device=x.device, dtype=torch.float32)
    xm = torch.empty_like(loss_b); em = torch.empty_like(loss_b)
    dv = torch.empty_like(v) if grad_scale is not None else None
    check(lib.gmk_v_loss(_p(v), _p(z), _p(x), _p(eps), _p(logsnr), _p(loss_b), _p(xm), _p(em), _p(dv),
                         float(grad_scale or 0.0), loss_type, MEAN_TYPES[mean_type], B, n, _s()), "v_loss")
    return loss_b, xm, em, dv


def sampler_step(v, z, logsnr_t, logsnr_s, is_last, v_uncond=None, cond_w=None, noise=None, want_pred=False, mean_type="v",
                 dup=False, logsnr_next=None):
    """dup: z_next is returned as the first half of a [2B, ...] tensor whose second half holds the same values (z2 = returned[1]);
    logsnr_next: fp32 [B] (or [2B] with dup) filled with logsnr_s."""
    _f32(v, "v"); _f32(z, "z")
    B = z.shape[0]
    n = z.numel() // B
    assert v.shape == z.shape
    for t in (v_uncond, noise):
        if t is not None:
            _f32(t, "aux"); assert t.shape == z.shape
    if cond_w is not None:
        _f32(cond_w, "cond_w"); assert cond_w.numel() == B
    z2 = torch.empty((2 * B,) + tuple(z.shape[1:]), device=z.device, dtype=z.dtype) if dup else None
    z_next = z2[:B] if dup else torch.empty_like(z)
    xp = torch.empty_like(z) if want_pred else None
    ep = torch.empty_like(z) if want_pred else None
    if logsnr_next is not None:
        _f32(logsnr_next, "logsnr_next"); assert logsnr_next.numel() == (2 * B if dup else B)
    check(lib.gmk_sampler_step(_p(v), _p(v_uncond), _p(cond_w), _p(z), _p(noise), float(logsnr_t), float(logsnr_s),
                               int(is_last), _p(z_next), _p(xp), _p(ep), _p(z2[B:]) if dup else None, _p(logsnr_next),
                               MEAN_TYPES[mean_type], B, n, _s()), "sampler_step")
    return ((z_next, z2) if dup else z_next), xp, ep


def logsnr_schedule(B, device, u=None, i_times=None, num_steps=1, shift=0.0, want_u=False):
    """logsnr = schedule(u - shift), u given or (i_times + 1) / num_steps.  -> logsnr (, u_shifted)"""
    if u is not None:
        _f32(u, "u")
    if i_times is not None:
        _chk(i_times, torch.int64, "i_times")
    logsnr = torch.empty((B,), device=device, dtype=torch.float32)
    uo = torch.empty((B,), device=device, dtype=torch.float32) if want_u else None
    check(lib.gmk_logsnr_schedule(_p(u), _p(i_times), int(num_steps), float(shift), _p(uo), _p(logsnr), B, _s()),
          "logsnr_schedule")
    return (logsnr, uo) if want_u else logsnr


def ddim_step_vec(v, z, logsnr_t, logsnr_s, v_uncond=None, cond_w=None, mean_type="v"):
    """Teacher DDIM step with per-sample times.  -> (z_s, x_pred, eps_pred)"""
    for t, nm in ((v, "v"), (z, "z"), (logsnr_t, "logsnr_t"), (logsnr_s, "logsnr_s")):
        _f32(t, nm)
    B = z.shape[0]
    n = z.numel() // B
    zs, xp, ep = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
    check(lib.gmk_ddim_step_vec(_p(v), _p(v_uncond), _p(cond_w), _p(z), _p(logsnr_t), _p(logsnr_s), _p(zs), _p(xp), _p(ep),
                                MEAN_TYPES[mean_type], B, n, _s()), "ddim_step_vec")
    return zs, xp, ep


def distill_target(z_teacher, z_t, x_pred_teacher, logsnr, logsnr_s, i_times):
    for t, nm in ((z_teacher, "z_teacher"), (z_t, "z_t"), (x_pred_teacher, "x_pred"), (logsnr, "logsnr"), (logsnr_s, "logsnr_s")):
        _f32(t, nm)
    _chk(i_times, torch.int64, "i_times")
    B = z_t.shape[0]
    n = z_t.numel() // B
    xt, et = torch.empty_like(z_t), torch.empty_like(z_t)
    check(lib.gmk_distill_target(_p(z_teacher), _p(z_t), _p(x_pred_teacher), _p(logsnr), _p(logsnr_s), _p(i_times), _p(xt), _p(et),
                                 B, n, _s()), "distill_target")
    return xt, et


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _f32(t, nm)
    assert p.numel() == g.numel() == m.numel() == v.numel()
    check(lib.gmk_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _s()), "adam_step")


# ---- self-attention core (north_star extension; no reference call site) -----------------------------------------
def bgemm_nt(A, B, out=None, alpha=1.0, out_dtype=None):
    """C[b] = alpha * A[b] @ B[b]^T for batches of K-contiguous matrices (row / batch strides free): A [batch, M, K],
    B [batch, N, K] -> C [batch, M, N].  bf16 operands run on the matrix cores (fp32 accumulation), fp32 operands on FMAs."""
    assert A.dim() == 3 and B.dim() == 3 and A.shape[0] == B.shape[0] and A.shape[2] == B.shape[2] and A.dtype == B.dtype
    assert A.stride(2) == 1 and B.stride(2) == 1 and A.is_cuda and A.dtype in _DT
    batch, M, K = A.shape
    N = B.shape[1]
    od = out_dtype or (out.dtype if out is not None else A.dtype)
    if out is None:
        out = torch.empty((batch, M, N), device=A.device, dtype=od)
    assert out.shape == (batch, M, N) and out.stride(2) == 1 and out.dtype == od
    check(lib.gmk_bgemm_nt(_p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(out), out.stride(0), out.stride(1),
                           batch, M, N, K, float(alpha), _DT[A.dtype], _DT[od], _s()), "bgemm_nt")
    return out


def attention_fwd(qkv, scale, want_p=False, fp8=False):
    """Fused softmax(scale * q k^T) v for qkv [B, N, 3C] (bf16, C = 128, N in {64, 128, 256}) -> (o [B, N, C], P [B, N, N] or None).
    fp8: both contractions on the fp8 (e4m3) matrix cores."""
    _chk(qkv, torch.bfloat16, "qkv")
    B, N, C3 = qkv.shape
    C = C3 // 3
    o = torch.empty((B, N, C), device=qkv.device, dtype=qkv.dtype)
    P = torch.empty((B, N, N), device=qkv.device, dtype=qkv.dtype) if want_p else None
    check(lib.gmk_attention_fwd(_p(qkv), _p(o), _p(P), B, N, C, float(scale), int(bool(fp8)), _s()), "attention_fwd")
    return o, P


def transpose_last2(x, out=None):
    """[batch, R, C] (unit last stride, free row / batch strides) -> contiguous [batch, C, R]."""
    assert x.dim() == 3 and x.stride(2) == 1 and x.is_cuda and x.dtype in _DT
    batch, R, C = x.shape
    if out is None:
        out = torch.empty((batch, C, R), device=x.device, dtype=x.dtype)
    check(lib.gmk_transpose(_p(x), x.stride(0), x.stride(1), _p(out), out.stride(0), out.stride(1), batch, R, C, _DT[x.dtype], _s()),
          "transpose")
    return out


def attention_bwd_fused_ok(qkv):
    """True if the fused backward of the attention core takes this shape (bf16, C = 128, N in {64, 128, 256}): the forward then need not keep P."""
    return qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.shape[2] == 384 and qkv.shape[1] in (64, 128, 256) and \
        os.environ.get("GMK_ATTN_BWD", "fused") != "gemm"


def attention_bwd(qkv, o, do):
    """(dq | dk | dv) [B, N, 3C] of o = softmax(scale q k^T) v given do, with P recomputed in registers (gmk_attention_bwd): no N x N matrix in HBM."""
    _chk(qkv, torch.bfloat16, "qkv"); _chk(o, torch.bfloat16, "o"); _chk(do, torch.bfloat16, "do")
    B, N, C3 = qkv.shape
    C = C3 // 3
    assert o.shape == (B, N, C) and do.shape == (B, N, C)
    dqkv = torch.empty_like(qkv)
    stats = torch.empty((B, N, 2), device=qkv.device, dtype=torch.float32)
    check(lib.gmk_attention_bwd(_p(qkv), _p(o), _p(do), _p(dqkv), _p(stats), B, N, C, float(C ** -0.5), _s()), "attention_bwd")
    return dqkv


def softmax_fwd(S, scale, dtype):
    """softmax(scale * S) over the last dimension; S fp32 contiguous -> P in `dtype`."""
    _f32(S, "S")
    N = S.shape[-1]
    P = torch.empty(S.shape, device=S.device, dtype=dtype)
    check(lib.gmk_softmax_fwd(_p(S), _p(P), S.numel() // N, N, float(scale), _DT[dtype], _s()), "softmax_fwd")
    return P


def softmax_bwd(P, dP, scale):
    """dS = scale * P * (dP - sum(dP * P)) per row; dP fp32, P / dS in P.dtype."""
    _chk(P, name="P"); _f32(dP, "dP")
    assert P.shape == dP.shape
    N = P.shape[-1]
    dS = torch.empty_like(P)
    check(lib.gmk_softmax_bwd(_p(P), _p(dP), _p(dS), P.numel() // N, N, float(scale), _DT[P.dtype], _s()), "softmax_bwd")
    return dS
