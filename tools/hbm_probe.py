"""What the box's HBM delivers to simple streaming kernels (context for the GroupNorm kernels' TB/s in DESIGN.md):
device-to-device copy, read-only sum and write-only fill of a 205 MB bf16 tensor (= one 1024x28x28x128 activation)."""
import torch
n = 1024 * 28 * 28 * 128
x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: y.copy_(x)); print(f"copy  (read+write {2*n*2/1e6:.0f} MB): {t*1e6:7.1f} us  {2*n*2/t/1e12:.2f} TB/s")
xi = x.view(torch.int16)
t = timed(lambda: xi.sum(dtype=torch.int64)); print(f"sum   (read {n*2/1e6:.0f} MB):       {t*1e6:7.1f} us  {n*2/t/1e12:.2f} TB/s")
t = timed(lambda: y.fill_(1.0)); print(f"fill  (write {n*2/1e6:.0f} MB):      {t*1e6:7.1f} us  {n*2/t/1e12:.2f} TB/s")
big = torch.empty(4 * n, device="cuda", dtype=torch.bfloat16); big2 = torch.empty_like(big)
t = timed(lambda: big2.copy_(big), 10); print(f"copy  (read+write {2*4*n*2/1e6:.0f} MB): {t*1e6:7.1f} us  {2*4*n*2/t/1e12:.2f} TB/s")
