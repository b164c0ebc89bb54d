"""In-process A/B of the U-Net forward (what a sampler step costs) under GMK_DEV_VARIANT values: python tools/fwd_ab.py 0 6"""
import sys, time, torch
sys.path.insert(0, ".")
from generative_models_amd.diffusion.simple_unet import SimpleUnet
from generative_models_amd._lib import lib
variants = [int(a) for a in sys.argv[1:]] or [0]
torch.manual_seed(0)
net = SimpleUnet(128, 0.0).cuda().eval()
B = 1024
z = torch.randn(B, 1, 28, 28, device="cuda"); l = torch.randn(B, device="cuda"); y = torch.randint(0, 10, (B,), device="cuda")
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        lib.gmk_set_dev_variant(v)
        for _ in range(3): net.forward_hip(z, l, y, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net.forward_hip(z, l, y, None)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 20 * 1e3)
for v in variants:
    print("variant", v, "forward ms: min %.3f" % min(res[v]), ["%.3f" % t for t in res[v]])
