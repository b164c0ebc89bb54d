"""Fixed cost per launch of the persistent 3x3 kernel: time against the number of tile rounds (32 x 32 images, 8 rows per tile: 4 tiles per image,
256 tiles = one round).  The slope is the tile time, the intercept what every launch pays on top (start-up, first fills, tail, drain)."""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops

C, S = 128, 32
g = torch.Generator().manual_seed(0)
w = (torch.randn((C, C, 3, 3), generator=g) / 34).cuda()
wf = torch.empty(w.numel(), device="cuda", dtype=torch.float16)
ops.pack_conv_weight(w, wf, None)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pts = []
for rounds in (1, 2, 3, 4, 6, 8, 12, 16, 32):
    B = 64 * rounds
    x = torch.randn((B, S, S, C), generator=g).cuda().half()
    t = timed(lambda: ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S)))
    pts.append((rounds, t))
    print(f"{rounds:3d} rounds (B = {B:5d}): {t:8.2f} us per launch  ({t / rounds:6.2f} us per round)", flush=True)
n = len(pts); sx = sum(r for r, _ in pts); sy = sum(t for _, t in pts); sxx = sum(r * r for r, _ in pts); sxy = sum(r * t for r, t in pts)
slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
print(f"least squares: {slope:.2f} us per round + {icpt:.2f} us per launch (back-to-back launches of the same kernel on one stream)")
