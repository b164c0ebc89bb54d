"""Host-side cost of a small-batch train step (bs = 32, 1x28x28: configs[0]'s shape) - where the ~ 5 ms of Python per step go.
    python tools/host_profile.py [B=32]"""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
from generative_models_amd import common
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, pad32=0, device="cuda", timesteps=200, bs=B)
m = Model(G).cuda().train()
x = (torch.rand(B, 1, 28, 28, device="cuda") * 2 - 1); y = torch.randint(0, 10, (B,), device="cuda")
for _ in range(10): m.train_step(x, y.clone())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): m.train_step(x, y.clone())
torch.cuda.synchronize(); print(f"B={B}: {(time.perf_counter() - t0) * 10:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): m.train_step(x, y.clone())
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
