"""Halo conv time vs number of tile rounds per workgroup: separates per-launch, per-tile and per-K-step costs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from tools.conv_bench import timed
T = torch.bfloat16
C, S = 128, 28
for nsrc in (1, 2):
    for B in ((1024,) if os.environ.get('GMK_DEV_VARIANT') else (82, 164, 329, 658, 1024, 2048)):
        srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(nsrc)]
        cin = nsrc * C
        w = torch.randn((C, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5
        wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
        ops.pack_conv_weight(w, wf, wd)
        ntiles = (B * S + 8) // 9
        t = timed(lambda: ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S)), 20)
        flops = 2.0 * B * S * S * C * cin * 9
        print(f"cin={cin} B={B:5d} tiles={ntiles:5d} rounds={ntiles/256:6.2f} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s", flush=True)
