"""What the box's HBM delivers to simple streaming kernels (context for the GroupNorm / stem kernels' TB/s in DESIGN.md):
device-to-device copy, read-only sum and write-only fill, at a size inside the 256-MiB Infinity Cache (205 MB = one 1024x28x28x128 16-bit
activation) and at sizes beyond it (537 MB = one 2048x32x32x128 activation, 1.07 GB, 2.1 GB).   python tools/hbm_probe.py"""
import torch
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for n in (1024 * 28 * 28 * 128, 2048 * 32 * 32 * 128, 2 * 2048 * 32 * 32 * 128, 4 * 2048 * 32 * 32 * 128):
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    b = 2 * n
    t_copy, t_read, t_fill = timed(lambda: y.copy_(x)), timed(lambda: x.view(torch.int16).max()), timed(lambda: y.fill_(1.5))
    print(f"{b / 1e6:7.0f} MB: copy {2 * b / t_copy / 1e12:5.2f} TB/s (read + write)   read-only {b / t_read / 1e12:5.2f} TB/s   write-only fill {b / t_fill / 1e12:5.2f} TB/s", flush=True)
    del x, y
