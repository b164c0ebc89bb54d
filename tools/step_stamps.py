"""Who waits for whom inside a K-step of the wave-specialised 3x3 halo kernel (round 6).  With a stamp buffer set (gmk_dev_set_stamp_buffer)
the launches run an instrumented instantiation (conv_halo.hip, kStamp): wave 4 (a producer) sums, per K-step index, the shader cycles
(s_memtime) between leaving a barrier and starting the next counted vmcnt wait (`issue`: the step's DMA instructions), in that wait (`land`:
data it issued has not landed) and in the barrier (`bar`: the consumers - or another producer - are not there yet); wave 0 (a consumer)
sums the cycles between barriers (`work`: MFMAs + fragment reads; step 0 includes the previous tile's epilogue) and in the barrier (`bar`:
the producers are not there yet).  The sums are divided by the number of steps of that index and printed in s_memtime counts.

    python tools/step_stamps.py [size=32] [batch=2048]
"""
import sys
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
from generative_models_amd._lib import lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
C = 128
g = torch.Generator().manual_seed(0)
x = torch.randn((B, S, S, C), generator=g).cuda().half()
k0 = torch.randn((B, S, S, C), generator=g).cuda().half()
k1 = torch.randn((B, S, S, C), generator=g).cuda().half()
res = torch.randn((B, S, S, C), generator=g).cuda().bfloat16()
w = (torch.randn((C, C, 3, 3), generator=g) / 34).cuda()
wsk = (torch.randn((C, 2 * C, 1, 1), generator=g) / 16).cuda()
bias = torch.zeros(C, device="cuda")
wf = torch.empty(w.numel(), device="cuda", dtype=torch.float16); ops.pack_conv_weight(w, wf, None)
wd = torch.empty(w.numel(), device="cuda", dtype=torch.bfloat16); ops.pack_conv_weight(w, None, wd)
wskf = torch.empty(wsk.numel(), device="cuda", dtype=torch.float16); ops.pack_conv_weight(wsk, wskf, None)
xb = x.bfloat16()
stamps = torch.zeros(256 * 128, device="cuda", dtype=torch.int32)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def stamped(name, fn, nsteps, labels):
    t_plain = timed(fn)
    lib.gmk_dev_set_stamp_buffer(stamps.data_ptr(), stamps.numel() * 4)
    try:
        t_st = timed(fn)
        stamps.zero_()
        fn()
        torch.cuda.synchronize()
    finally:
        lib.gmk_dev_set_stamp_buffer(None, 0)
    a = stamps.view(256, 2, 64).cpu().double()
    nst = a[:, 0, 48].clamp(min=1)[:, None]                  # producer: steps of each index this workgroup ran (jobs x phases, or jobs)
    iss, land, pbar = a[:, 0, 0:nsteps] / nst, a[:, 0, 16:16 + nsteps] / nst, a[:, 0, 32:32 + nsteps] / nst
    work, cbar = a[:, 1, 0:nsteps] / nst, a[:, 1, 16:16 + nsteps] / nst
    print(f"\n{name}: {t_plain:.1f} us per launch shipped, {t_st:.1f} us stamped; cycles per K-step, mean over {a.shape[0]} workgroups")
    print(f"{'step':>6} | producer: {'issue':>7} {'land':>7} {'bar':>7} | consumer: {'work':>7} {'bar':>7} | step total (consumer)")
    tot = 0.0
    for i in range(nsteps):
        c = float(work[:, i].mean() + cbar[:, i].mean())
        tot += c
        print(f"{labels[i]:>6} | {float(iss[:, i].mean()):17.0f} {float(land[:, i].mean()):7.0f} {float(pbar[:, i].mean()):7.0f} | "
              f"{float(work[:, i].mean()):17.0f} {float(cbar[:, i].mean()):7.0f} | {c:8.0f}")
    epi = a[:, 1, 32:35] / a[:, 1, 35].clamp(min=1)[:, None]
    print(f"epilogue of a tile (consumer wave 0): last step's MFMAs drain {float(epi[:, 0].mean()):.0f}, bias loads (+ hand-over barriers) {float(epi[:, 1].mean()):.0f}, "
          f"conversion + store issue {float(epi[:, 2].mean()):.0f}")
    print(f"sum over the {nsteps} step kinds: {tot:.0f} cycles (consumer view); producer issue {float(iss.mean(0).sum()):.0f}, "
          f"land {float(land.mean(0).sum()):.0f}, bar {float(pbar.mean(0).sum()):.0f}; consumer work {float(work.mean(0).sum()):.0f}, bar {float(cbar.mean(0).sum()):.0f}")
    return t_plain


taps = [f"t{i}" for i in range(9)]
for var, what in ((21, "no epilogue stores"), (31, "stamps of waves 1 / 5"), (32, "stamps of waves 2 / 6"), (33, "stamps of waves 3 / 7")):
    lib.gmk_set_dev_variant(var)
    print(f"\n==== GMK_DEV_VARIANT={var} (stamped instantiation only): {what}")
    stamped(f"plain fp16 forward {S}x{S} B={B} K=1152", lambda: ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S), bias=bias), 9, taps)
lib.gmk_set_dev_variant(0)
print("\n==== shipped schedule")
stamped(f"plain fp16 forward {S}x{S} B={B} K=1152", lambda: ops.conv_igemm([x], wf, C, 3, ops.NORMAL, (S, S), bias=bias), 9, taps)
stamped(f"plain bf16 dgrad + residual {S}x{S} B={B} K=1152", lambda: ops.conv_igemm([xb], wd, C, 3, ops.NORMAL, (S, S), residual=res), 9, taps)
# producer step order of the folded kernel: t0 t1 t2 t3 e0 t4 e1 t5 e2 t6 e3 t7 t8; the consumer indexes taps 0..8 then dense 9..12
prod = ["t0", "t1", "t2", "t3", "e0", "t4", "e1", "t5", "e2", "t6", "e3", "t7", "t8"]
print("\n(folded kernel: the producer columns are in ISSUE order " + " ".join(prod) + ", the consumer columns in the order t0..t8 e0..e3)")
stamped(f"folded skip fp16 {S}x{S} B={B} K=1152+256", lambda: ops.conv3x3_skipfold(x, wf, bias, [k0, k1], wskf, bias), 13,
        [f"{a}/{b}" for a, b in zip(prod, taps + ["e0", "e1", "e2", "e3"])])
