"""Time the stem / head convolution kernels (HBM-bound: one pass over the C-channel tensor each)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import ops
from tools.conv_bench import timed
B, S, C = (int(v) for v in (sys.argv[1:4] + ["1024", "28", "128"][len(sys.argv) - 1:]))
for cs in (1, 3):
    T = torch.bfloat16
    x = torch.randn(B, cs, S, S, device="cuda")
    w = torch.randn(C, cs, 3, 3, device="cuda") * 0.1; b = torch.randn(C, device="cuda")
    wh = torch.randn(cs, C, 3, 3, device="cuda") * 0.1; bh = torch.randn(cs, device="cuda")
    a = torch.randn(B, S, S, C, device="cuda").to(T)
    dw = torch.empty(C * cs * 9, device="cuda"); dwb = torch.empty(cs * C * 9 + cs, device="cuda")
    big = a.numel() * 2
    for name, fn in (("stem_fwd", lambda: ops.stem_fwd(x, w, b, C, T)), ("stem_wgrad", lambda: ops.stem_wgrad(x, a, dw)),
                     ("head_fwd", lambda: ops.head_fwd(a, wh, bh)), ("head_dgrad", lambda: ops.head_dgrad(x, wh, T)),
                     ("head_wgrad", lambda: ops.head_wgrad(x, a, dwb))):
        t = timed(fn, 20)
        print(f"B={B} S={S} cs={cs} {name:10s} {t*1e6:8.1f} us  {big/t/1e12:5.2f} TB/s of one pass", flush=True)
