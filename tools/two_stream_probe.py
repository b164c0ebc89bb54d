"""Does running the U-Net forward as TWO half-batches on two HIP streams hide the per-launch start-up / tail of the persistent kernels?
python tools/two_stream_probe.py cfg2 30     (forward passes of the sampler's net; full batch on one stream vs two halves on two streams)"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from generative_models_amd import common
key, n = sys.argv[1], int(sys.argv[2])
cin, S, B, attention, _ = bench.CONFIGS[key]
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, pad32=0, device="cuda", timesteps=1000, bs=B, in_channels=cin, attention=attention)
m = Model(G).cuda().eval(); m.size = S
net = m.net
y = torch.randint(0, 10, (B,), device="cuda")
x = torch.randn((B, cin, S, S), device="cuda")
l = torch.randn((B,), device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]
h = B // 2


def full():
    return net.forward_hip(x, l, y, None)


def chunked(K):
    c = B // K
    parts = [(x[k * c:(k + 1) * c].contiguous(), l[k * c:(k + 1) * c].contiguous(), y[k * c:(k + 1) * c].contiguous()) for k in range(K)]

    def run():
        cur = torch.cuda.current_stream()
        outs = []
        for k in range(K):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                outs.append(net.forward_hip(*parts[k], None))
        for k in range(K):
            cur.wait_stream(streams[k])
        return outs
    return run


halves = chunked(2)


with torch.no_grad():
    o = full(); oa, ob = halves()
    torch.cuda.synchronize()
    print("same bits:", torch.equal(o[:h], oa), torch.equal(o[h:], ob))
    for rnd in range(3):
        for name, fn in (("one stream, full batch", full), ("two streams, half batches", halves), ("three streams", chunked(3)) if B % 3 == 0 else ("four streams, quarters", chunked(4)), ("four streams, quarters", chunked(4))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{key} B={B} {name}: {dt / n * 1e3:.3f} ms per forward", flush=True)
