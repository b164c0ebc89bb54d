"""Repeats one U-Net forward (B=1024, 1x28x28, bf16) with the side stream on and compares the outputs bit for bit.

Rounds 1-2 used it to chase a miscompare that only a build with ds_bpermute wave reductions showed; the
shipped kernels (DPP / permlane reductions) give identical runs, and tests/test_gpu_fullsize.py keeps that as a regression test.  Usage: python tools/side_stream_det.py [runs]"""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
ops.FWD_SIDE = True          # the overlap under investigation (off by default in the product)
from generative_models_amd.diffusion.simple_unet import SimpleUnet

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.manual_seed(0)
net = SimpleUnet(128, 0.0).cuda().eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if ".out_layers.3.weight" in n:
            p.uniform_(-0.02, 0.02)
net.mark_params_changed()
B = 1024
z = torch.randn(B, 1, 28, 28, device="cuda"); l = torch.randn(B, device="cuda"); y = torch.randint(0, 10, (B,), device="cuda")
outs = [net.forward_hip(z, l, y, None).clone() for _ in range(runs)]
torch.cuda.synchronize()
same = [bool(torch.equal(outs[0], o)) for o in outs]
print("side on :", same.count(True), "of", runs, "identical; max |diff|", max(float((outs[0] - o).abs().max()) for o in outs))
ops.WGRAD_STREAM = False
o2 = [net.forward_hip(z, l, y, None).clone() for _ in range(3)]
print("side off:", [bool(torch.equal(o2[0], o)) for o in o2], "on == off:", bool(torch.equal(o2[0], outs[0])))
sys.exit(0 if all(same) and torch.equal(o2[0], outs[0]) else 1)
