"""3x3 conv at the 7x7 / 14x14 levels: automatic kernel choice vs forced halo kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from generative_models_amd._lib import lib
from tools.conv_bench import timed
T = torch.bfloat16
C = 128
for S in (7, 14):
    for nsrc in (1, 2):
        B = 1024
        srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(nsrc)]
        cin = nsrc * C
        w = torch.randn((C, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5
        wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
        ops.pack_conv_weight(w, wf, wd)
        flops = 2.0 * B * S * S * C * cin * 9
        for force in (0, 3, 2):
            lib.gmk_set_kernel_choice(force, -1, -1)
            t = timed(lambda: ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S)), 20)
            print(f"S={S} cin={cin} force={force} kernel={lib.gmk_last_kernel()} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s", flush=True)
lib.gmk_set_kernel_choice(-1, -1, -1)
