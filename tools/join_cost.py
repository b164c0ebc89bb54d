"""What did round 2's four main-stream joins per backward pass cost?  One GPU, a no-op bucket callback, the bench's train step:
  (a) no callback at all, (b) callback + join of the weight-gradient stream on the main stream in front of each bucket (round 2's
  data-parallel schedule), (c) callback without joins (round 3: the exchange stream orders itself behind both streams).
python tools/join_cost.py [cfg2|cfg3]"""
import sys, time
from functools import partial
import torch
sys.path.insert(0, ".")
import bench
from generative_models_amd import common, ops

key = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cin, S, B, attention, _ = bench.CONFIGS[key]
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, timesteps=1000, bs=B, in_channels=cin, attention=attention, seed=0)
model = Model(G).cuda(); model.train()
x, y = bench.synthetic_batch(B, cin, S, "cuda", 1)
net, d = model.net, model.diffusion
def step(mode):
    kw = {} if mode == "a" else dict(on_grads_ready=lambda k: None, join_side_before_ready=(mode == "b"))
    d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, **kw)
    model.optimizer.step()
    ops.throttle()
for m in "abc":
    for _ in range(3): step(m)
res = {m: [] for m in "abc"}
for rnd in range(3):
    for m in "abc":
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step(m)
        torch.cuda.synchronize(); res[m].append((time.perf_counter() - t0) / 10 * 1e3)
print(f"{key}: ms/step  (a) no callback {min(res['a']):.3f}   (b) 4 main-stream joins {min(res['b']):.3f}   (c) no joins {min(res['c']):.3f}")
