"""Is the train step host-bound?  Compares the time Python needs to enqueue N steps with the time the GPU needs to drain them."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import common
import bench
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, pad32=0, device="cuda", timesteps=1000, bs=1024)
model = Model(G).to("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x, y = bench.synthetic_batch(B, 1, 28, "cuda", 1000)
for _ in range(3):
    model.train_step(x, y.clone())
torch.cuda.synchronize()
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(8):
        model.train_step(x, y.clone())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: enqueue {1e3*(t1-t0)/8:.2f} ms/step, drained after {1e3*(t2-t0)/8:.2f} ms/step", flush=True)
