"""Sub-pixel forms (conv_subpixel.hip) against the forms they replace, per launch at the train step's shapes, same box, interleaved:
`Upsample` forward = gmk_conv_subpixel(UPSAMPLE) vs the nearest-x2 addressing of the halo kernel; stride-2 data gradient =
gmk_conv_subpixel(TRANSPOSED) vs the zero-stuffed halo form (GMK_CONV_KERNEL=3), with and without the residual the net adds."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


C = 128
for (B, S) in ((2048, 16), (2048, 8), (1024, 14), (1024, 7), (1024, 32), (1024, 16)):      # low-resolution size
    g = torch.Generator().manual_seed(0)
    w = (torch.randn((C, C, 3, 3), generator=g) / 34).cuda()
    bias = torch.zeros(C, device="cuda")
    for act, name in ((torch.float16, "upsample fwd (fp16)"),):
        x = torch.randn((B, S, S, C), generator=g).cuda().to(act)
        wf = torch.empty(w.numel(), device="cuda", dtype=act)
        ops.pack_conv_weight(w, wf, None)
        wsub = torch.empty(16 * C * C, device="cuda", dtype=act)
        ops.pack_upsample_weight(w, wsub)
        for rnd in range(2):
            t_old = timed(lambda: ops.conv_igemm([x], wf, C, 3, ops.UPSAMPLE2, (2 * S, 2 * S), bias=bias))
            t_new = timed(lambda: ops.conv_subpixel(x, wsub, C, ops.SUBPIXEL_UPSAMPLE, bias=bias))
            fl = 2.0 * B * 4 * S * S * C * C * 9
            print(f"B={B} {S}->{2 * S} {name}: nearest-x2 halo {t_old:7.1f} us ({fl / t_old / 1e6:6.0f} TFLOP/s)  sub-pixel {t_new:7.1f} us ({fl / t_new / 1e6:6.0f} algorithmic TFLOP/s)  ratio {t_new / t_old:.3f}", flush=True)
    dyh = torch.randn((B, 2 * S, 2 * S, C), generator=g).cuda().bfloat16()
    wfb0 = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); wd0 = torch.empty_like(wfb0)
    ops.pack_conv_weight(w, wfb0, wd0)
    wsd = torch.empty(16 * C * C, device="cuda", dtype=torch.bfloat16)
    ops.pack_upsample_weight(w, None, wsd)
    for rnd in range(2):
        t_old = timed(lambda: ops.sumpool2x2(ops.conv_igemm([dyh], wd0, C, 3, ops.NORMAL, (2 * S, 2 * S))))
        t_new = timed(lambda: ops.conv_subpixel(dyh, wsd, C, ops.SUBPIXEL_UPSAMPLE_DGRAD))
        print(f"B={B} {2 * S}->{S} upsample dgrad (bf16): dgrad3x3 + sumpool {t_old:7.1f} us  sub-pixel {t_new:7.1f} us  ratio {t_new / t_old:.3f}", flush=True)
    xa = torch.randn((B, S, S, C), generator=g).cuda().half()
    dwa, dwb = torch.empty((C, C, 3, 3), device="cuda"), torch.empty((C, C, 3, 3), device="cuda")
    if ops.conv_wgrad_subpixel_ok(B, S, S, C, torch.bfloat16):
        for rnd in range(2):
            t_old = timed(lambda: ops.conv_wgrad(dyh, [xa], 3, ops.UPSAMPLE2, dwa))
            t_new = timed(lambda: ops.conv_wgrad_subpixel(dyh, xa, dwb))
            print(f"B={B} {S}->{2 * S} upsample wgrad (fp16 x): nearest-x2 slots {t_old:7.1f} us  sub-pixel {t_new:7.1f} us  ratio {t_new / t_old:.3f}", flush=True)
    dy = torch.randn((B, S, S, C), generator=g).cuda().bfloat16()
    res = torch.randn((B, 2 * S, 2 * S, C), generator=g).cuda().bfloat16()
    wfb = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); wd = torch.empty_like(wfb)
    ops.pack_conv_weight(w, wfb, wd)
    for rnd in range(2):
        for r in (None, res):
            lib.gmk_set_kernel_choice(3, -1, -1)
            t_old = timed(lambda: ops.conv_igemm([dy], wd, C, 3, ops.TRANSPOSED2, (2 * S, 2 * S), residual=r))
            k_old = lib.gmk_last_kernel()
            lib.gmk_set_kernel_choice(-1, -1, -1)
            t_new = timed(lambda: ops.conv_subpixel(dy, wd, C, ops.SUBPIXEL_TRANSPOSED, residual=r))
            nb = dy.numel() * 2 + res.numel() * 2 * (2 if r is not None else 1)
            print(f"B={B} {S}->{2 * S} stride-2 dgrad residual={r is not None}: zero-stuffed (kernel {k_old}) {t_old:7.1f} us  sub-pixel {t_new:7.1f} us ({nb / t_new / 1e6:5.2f} TB/s)  ratio {t_new / t_old:.3f}", flush=True)
