import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from tools.conv_bench import timed
T = torch.bfloat16
B, C, S = 1024, 128, 28
for nsrc in (1, 2):
    srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(nsrc)]
    cin = nsrc * C
    w = torch.randn((C, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5
    wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
    ops.pack_conv_weight(w, wf, wd)
    bias = torch.zeros(C, device="cuda")
    res = torch.randn((B, S, S, C), device="cuda").to(T)
    emb = torch.randn((B, C), device="cuda")
    flops = 2.0 * B * S * S * C * cin * 9
    for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+res", dict(bias=bias, residual=res)),
                     ("bias+emb", dict(bias=bias, emb=emb))):
        t = timed(lambda: ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), **kw), 20)
        print(f"cin={cin} {name:10s} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s", flush=True)
