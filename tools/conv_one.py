import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
T = torch.bfloat16
B, C, S = 1024, 128, 28
nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
srcs = [torch.randn((B, S, S, C), device="cuda").to(T) for _ in range(nsrc)]
cin = nsrc * C
w = torch.randn((C, cin, 3, 3), device="cuda") / (cin * 9) ** 0.5
wf = torch.empty(w.numel(), device="cuda", dtype=T); wd = torch.empty_like(wf)
ops.pack_conv_weight(w, wf, wd)
bias = torch.zeros(C, device="cuda")
for _ in range(5):
    ops.conv_igemm(srcs, wf, C, 3, ops.NORMAL, (S, S), bias=bias)
torch.cuda.synchronize()
