"""Sampler step time of one bench config (DDIM, guidance off), N steps after warm-up.  python tools/sampler_probe.py cfg2 60"""
import os, sys, time
import torch
sys.path.insert(0, ".")
import bench
from functools import partial
from generative_models_amd import common
key, n = sys.argv[1], int(sys.argv[2])
cin, S, B, attention, _ = bench.CONFIGS[key]
if len(sys.argv) > 3:                      # batch override: python tools/sampler_probe.py cfg2 40 512
    B = int(sys.argv[3])
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, pad32=0, device="cuda", timesteps=n, bs=B, in_channels=cin, attention=attention)
m = Model(G).cuda().eval(); m.size = S
y = torch.randint(0, 10, (B,), device="cuda")
init = m._aux_rng.normal((B, cin, S, S), "cuda")
m.diffusion.num_steps = 5; m.diffusion.sample(net=partial(m.net, guide=y), init_x=init, record=False)
m.diffusion.num_steps = n
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.diffusion.sample(net=partial(m.net, guide=y), init_x=init, record=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{key} B={B}: {n / dt:.2f} steps/s  ({dt / n * 1e3:.3f} ms per step, {B * n / dt / 1e3:.1f} k image-steps/s)", flush=True)
