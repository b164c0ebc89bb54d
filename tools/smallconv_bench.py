"""Times the 3-channel stem / head kernels at the BASELINE shapes and prints a digest of every output, so that two runs with
different GMK_DEV_VARIANT values (41 = VALU expand kernel) can be compared for speed AND for bit identity.

    python tools/smallconv_bench.py [cfg2|cfg3]
"""
import hashlib
import sys

import torch

from generative_models_amd import ops

SHAPES = {"cfg2": (2048, 3, 32), "cfg3": (1024, 3, 64), "cfg1": (2048, 1, 28)}


def digest(t):
    return hashlib.sha1(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]


def timed(fn, n=20):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3, out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    B, cs, S = SHAPES[which]
    C = 128
    g = torch.Generator(device="cuda").manual_seed(5)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, w, b = r(B, cs, S, S).clamp(-1, 1), r(C, cs, 3, 3) / 3, 0.1 * r(C)
    wh, bh, do = r(cs, C, 3, 3) / 34, 0.1 * r(cs), r(B, cs, S, S)
    for dtype in (torch.bfloat16, torch.float32):
        a = r(B, S, S, C).to(dtype)
        dy = r(B, S, S, C).to(dtype)
        dw = torch.empty(C, cs, 3, 3, device="cuda")
        dwb = torch.empty(cs * C * 9 + cs, device="cuda")
        out_bytes = B * S * S * C * a.element_size()
        rows = [("stem_fwd", lambda: ops.stem_fwd(x, w, b, C, dtype)),
                ("head_dgrad", lambda: ops.head_dgrad(do, wh, dtype)),
                ("stem_wgrad", lambda: ops.stem_wgrad(x, dy, dw)),
                ("head_fwd", lambda: ops.head_fwd(a, wh, bh)),
                ("head_wgrad", lambda: ops.head_wgrad(do, a, dwb))]
        for name, fn in rows:
            us, out = timed(fn)
            print(f"{which} {str(dtype)[6:]:9s} {name:11s} {us:8.1f} us  {out_bytes / us / 1e6:6.2f} TB/s of the C-channel tensor  {digest(out)}", flush=True)


if __name__ == "__main__":
    main()
