"""Folded skip convolution (gmk_conv3x3_skipfold) against the two launches it replaces (1x1 skip convolution + conv2 with the residual
epilogue), per launch, interleaved on one box.   python tools/fold_ab.py [S B] ...   (default: the up-path levels of configs[1..3])"""
import math
import sys

import torch

sys.path.insert(0, ".")
from generative_models_amd import ops


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    shapes = [(32, 2048), (16, 2048), (8, 2048), (28, 1024), (14, 1024), (7, 1024), (64, 1024)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    C, dt = 128, torch.float16
    for S, B in shapes:
        g = torch.Generator(device="cuda").manual_seed(S)
        a2 = torch.randn((B, S, S, C), device="cuda", generator=g).to(dt)
        xs = [torch.randn((B, S, S, C), device="cuda", generator=g).to(dt) for _ in range(2)]
        w = (torch.randn((9 * C * C,), device="cuda", generator=g) / math.sqrt(9 * C)).to(dt)
        ws = (torch.randn((C * 2 * C,), device="cuda", generator=g) / math.sqrt(2 * C)).to(dt)
        b1, b2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        if not ops.conv_skipfold_ok(a2, xs):
            print(f"{S}x{S} B={B}: not foldable"); continue
        def two():
            r = ops.conv_igemm(xs, ws, C, 1, ops.NORMAL, (S, S), bias=b2)
            return ops.conv_igemm([a2], w, C, 3, ops.NORMAL, (S, S), bias=b1, residual=r)
        def plain():
            return ops.conv_igemm([a2], w, C, 3, ops.NORMAL, (S, S), bias=b1)
        def fold():
            return ops.conv3x3_skipfold(a2, w, b1, xs, ws, b2)
        t = {"two": [], "fold": [], "plain": []}
        for _ in range(3):
            for k, f in (("two", two), ("fold", fold), ("plain", plain)):
                t[k].append(bench(f))
        m = {k: sorted(v)[1] for k, v in t.items()}
        fl = 2.0 * B * S * S * C * (9 * C + 2 * C)
        print(f"{S}x{S} B={B}: two launches {m['two']:7.1f} us | folded {m['fold']:7.1f} us ({fl / m['fold'] / 1e6:6.0f} TFLOP/s) | conv2 alone, no residual {m['plain']:7.1f} us"
              f" | fold saves {m['two'] - m['fold']:6.1f} us = {100 * (1 - m['fold'] / m['two']):.1f} %", flush=True)


if __name__ == "__main__":
    main()
