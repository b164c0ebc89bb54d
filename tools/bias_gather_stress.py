"""Stress for the one ds_bpermute user in csrc/ (round 6: the 3x3 halo kernels gather a tile's bias from two registers, conv_halo.hip load_bias):
identical train passes and forwards with every overlap ON (weight gradients and 1x1 skip convolutions on the side stream) must stay bit-identical.
    python tools/bias_gather_stress.py [passes per shape = 40]"""
import sys, torch
sys.path.insert(0, ".")
from functools import partial
from generative_models_amd import ops
from generative_models_amd.diffusion.simple_unet import SimpleUnet
from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for (B, S, cin) in ((1024, 28, 1), (2048, 32, 3), (512, 64, 3)):
    net = SimpleUnet(128, 0.0, in_channels=cin, compute_dtype=torch.bfloat16)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if ".out_layers.3.weight" in n: p.uniform_(-0.02, 0.02)
            if n.endswith(".bias"): p.uniform_(-0.2, 0.2)            # live biases: the gathered values matter
    net = net.cuda()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((B, cin, S, S), generator=g) * 2 - 1).cuda(); y = torch.randint(0, 10, (B,), generator=g).cuda()
    u = torch.rand((B,), generator=g).cuda(); eps = torch.randn((B, cin, S, S), generator=g).cuda()
    d = GaussianDiffusion(mean_type="v", num_steps=1000)
    ops.FWD_SIDE = True
    ref_l = ref_g = ref_f = None
    bad = 0
    for it in range(N):
        out = d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps)
        l, gr = out["loss"].clone(), net.flat_grads.clone()
        f = net.forward_hip(x, u * 20 - 10, y, None).clone()
        if ref_l is None: ref_l, ref_g, ref_f = l, gr, f
        else: bad += int(not (torch.equal(l, ref_l) and torch.equal(gr, ref_g) and torch.equal(f, ref_f)))
    print(f"{cin}x{S}x{S} B={B}: {N} train passes + {N} forwards with the side streams on: {bad} differing", flush=True)
    assert bad == 0
    del net
    torch.cuda.empty_cache()
print("ok")
